#!/bin/bash
# average duration of the kernels whose name contains $1 under rocprofv3 --kernel-trace --stats, for each library given
# usage: tools/kernel_ms.sh <kernel substring> "<bench args>" name=lib [name=lib ...]
kern="$1"; args="$2"; shift 2
export TMPDIR=/tmp
for nv in "$@"; do
  name="${nv%%=*}"; lib="${nv#*=}"
  d=gpurun_out/kms_$name; rm -rf "$d"; mkdir -p "$d"
  SIMMR_HIP_LIB="$lib" rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python3 bench.py --no-cpu-baseline --no-other-mode $args > "$d/log" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "$name: $(grep "$kern" "$f" | cut -d, -f1-4 | tr '\n' ' ')"
done
