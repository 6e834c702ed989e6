#!/bin/bash
# Which LDS access of k_emit_philox owns its bank-conflict cycles: the LDS counters of the product
# library and of the differential builds that compile one piece out (`make ablate`: lookup = no jtab gathers, draws = no
# Philox / lookups / plane load, items = one round of items per block), per launch of the default bench command, and of
# the TEXT form.  usage: tools/lds_probe.sh <out-dir-under-gpurun_out> name=lib [name=lib ...]
out="$1"; shift
S="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for nv in "$@"; do
  name="${nv%%=*}"; lib="${nv#*=}"
  export SIMMR_HIP_LIB="$lib"
  echo "== $name (columns, 16-byte slots)"
  tools/pmc_cmd.sh "$out/$name" "k_emit_philox<false, false, true, false" "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --steps 1 --warmup 0
  if [ "$name" = product ]; then
    echo "== $name (TEXT form)"
    tools/pmc_cmd.sh "$out/${name}_text" "k_emit_philox<false, false, true, true" "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --through-fastq --steps 1 --warmup 0
    echo "== $name (columns, compact)"
    tools/pmc_cmd.sh "$out/${name}_compact" "k_emit_philox<false, false, true, false" "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --layout compact --steps 1 --warmup 0
  fi
done
