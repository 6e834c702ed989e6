#!/bin/bash
# PMC passes for one bench.py command (one counter set per run, no tracing: the guide's rule), summed per kernel.
# usage: tools/pmc_emit.sh <out-dir-under-gpurun_out> <kernel-substring> "<bench.py args>"
out="gpurun_out/$1"; kern="$2"; args="$3"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -o p -- python3 bench.py --no-cpu-baseline --no-other-mode --steps 1 --warmup 0 $args > "$out/p$i.log" 2>&1
done
python3 profiles/summarize_pmc.py "$kern" "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt" | grep -v "^#"
