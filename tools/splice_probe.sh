#!/bin/bash
# custom-long bench with a build of the library: time, then FETCH_SIZE / WRITE_SIZE of the splice kernel.
# usage: tools/splice_probe.sh name lib
name="$1"; export SIMMR_HIP_LIB="$2"
python3 bench.py --profile custom-long --reads 1000000 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'bases/s %.4g' % d['value'], 'ms_per_step %.2f' % d['ms_per_step'], 'kernel_ms %.2f' % d['roofline'].get('kernel_ms'))"
tools/pmc_cmd.sh sp_$name k_custom_long_splice "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU" -- python3 bench.py --profile custom-long --reads 1000000 --no-cpu-baseline --steps 1 --warmup 0
