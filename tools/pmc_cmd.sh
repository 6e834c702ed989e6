#!/bin/bash
# PMC passes for an arbitrary command (one counter set per run, no tracing), summed per kernel.
# Every pass runs under its own 4-minute limit: a counter set the hardware cannot collect together makes rocprofv3 abort
# and then hang.
# usage: tools/pmc_cmd.sh <out-dir-under-gpurun_out> <kernel-substring> <counter sets separated by ';'> -- cmd...
out="gpurun_out/$1"; kern="$2"; sets="$3"; shift 4
mkdir -p "$out"
export TMPDIR=/tmp
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -o p -- "$@" > "$out/p$i.log" 2>&1
done
python3 profiles/summarize_pmc.py "$kern" "$out" > "$out/summary.txt" 2>&1
grep -v "^#" "$out/summary.txt"
