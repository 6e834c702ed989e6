#!/bin/bash
# k_fastq_write (the framing kernel of the two-step FASTQ path) against its grid: bench.py --profile perfect-short runs
# emit + framing as its untimed `through_fastq` side measurement; one line per SIMMR_FASTQ_GRID_MULT.
for m in "$@"; do
  out=$(SIMMR_FASTQ_GRID_MULT=$m timeout -k 10 300 python bench.py --no-cpu-baseline --profile perfect-short 2>/dev/null | tail -1)
  echo "fastq_mult=$m $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); t=d["through_fastq"]; print("through_fastq_ms=%.3f emit_kernel_ms=%.3f plan_ms=%.3f fastq_plan_ms=%.3f" % (t["ms_per_step"], t["emit_kernel_ms"], t["plan_ms"], t["fastq_plan_ms"]))' 2>&1)"
done
