#!/bin/bash
# A/B timing of builds of libsimmr_hip.so on ONE box in ONE call: every named build runs the
# same bench.py command, the whole list is repeated ROUNDS times, and the emit-kernel time of each run is printed.
# usage: tools/ab_bench.sh "<bench.py args>" ROUNDS name=path [name=path ...]
args="$1"; rounds="$2"; shift 2
for r in $(seq 1 "$rounds"); do
  for nv in "$@"; do
    name="${nv%%=*}"; lib="${nv#*=}"
    out=$(SIMMR_HIP_LIB="$lib" timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-mode $args 2>/dev/null | tail -1)
    echo "$name round $r: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.3f ms_per_step=%.3f plan_ms=%.3f value=%.4g subst=%.6f phred=%.4f" % (d["roofline"]["kernel_ms"], d["ms_per_step"], d["plan_ms_per_step"], d["value"], d.get("substitution_rate", 0), d.get("mean_phred", 0)))' 2>&1)"
  done
done
