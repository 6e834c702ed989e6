#!/bin/bash
# emit-kernel time against the number of workgroups launched per CU (SIMMR_PHILOX_WGS_PER_CU), for one library
lib="$1"; shift
for k in "$@"; do
  out=$(SIMMR_HIP_LIB="$lib" SIMMR_PHILOX_WGS_PER_CU=$k timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-mode 2>/dev/null | tail -1)
  echo "wgs_per_cu=$k: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.3f" % d["roofline"]["kernel_ms"])')"
done
