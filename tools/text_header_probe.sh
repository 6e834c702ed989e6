for envs in "A=1" "SIMMR_FASTQ_HEADERS=1"; do
  for prof in "--profile perfect-short" ""; do
    out=$(env $envs timeout -k 10 300 python bench.py --no-cpu-baseline $prof 2>/dev/null | tail -1)
    echo "[$envs] [$prof] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); t=d["through_fastq"]; print("through_fastq_ms=%.3f emit_kernel_ms=%.3f plan_ms=%.3f fastq_plan_ms=%.3f" % (t["ms_per_step"], t["emit_kernel_ms"], t["plan_ms"], t["fastq_plan_ms"]))' 2>&1)"
  done
done
