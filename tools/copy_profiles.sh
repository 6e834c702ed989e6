#!/bin/bash
# gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/<tag>/ : the summaries that are committed
tag="${1:-r5}"; src="gpurun_out/prof_$tag"; dst="profiles/$tag"; mkdir -p "$dst"
for n in default perfect_short custom_long custom_long_reference custom_short minimal_long through_fastq; do
  cp "$src/kernel_stats_bench_$n.csv" "$dst/kernel_stats_bench_$n.csv"
  cp "$src/bench_bench_${n}_under_rocprof.json" "$dst/bench_${n}_under_rocprof.json"
done
cp "$src/pmc_default.txt" "$dst/pmc_k_emit_philox_slot16.txt"
cp "$src/pmc_compact.txt" "$dst/pmc_k_emit_philox.txt"
cp "$src/pmc_perfect.txt" "$dst/pmc_k_emit_perfect_pe.txt"
cp "$src/pmc_custom_long.txt" "$dst/pmc_k_custom_long_splice_ctr_bench.txt"   # (the counter mode: the custom-long bench's default)
cp "$src/pmc_custom_long_reference.txt" "$dst/pmc_k_custom_long_splice.txt"
(echo "# the TEXT form's launch of bench.py --through-fastq --steps 1 --warmup 0 (the command's one column launch is left out)"; cat "$src/pmc_through_fastq.txt") > "$dst/pmc_k_emit_philox_text.txt"
cp "$src/pmc_lanes.txt" "$dst/pmc_k_emit_lanes.txt" 2>/dev/null
cp "$src/pmc_text_lines.txt" "$dst/pmc_k_emit_text_lines.txt" 2>/dev/null
cp "$src/kernel_stats_bench_through_fastq_whole_lines.csv" "$src/bench_bench_through_fastq_whole_lines_under_rocprof.json" "$dst/" 2>/dev/null
mv "$dst/bench_bench_through_fastq_whole_lines_under_rocprof.json" "$dst/bench_through_fastq_whole_lines_under_rocprof.json" 2>/dev/null
# bases the custom-long command's splice launch wrote = its stream bytes / 2 (seq + qual), from the JSON line of that run
bases=$(python3 -c "import json,re,sys; d=json.load(open('$dst/bench_custom_long_under_rocprof.json')); print(int(re.search(r'(\d+) stream bytes', d['config']['layout']).group(1)) // 2)")
python3 tools/make_pmc_traffic.py "$src" --mix 0.60,0.23,0.17 --mix-slot 0.63,0.21,0.16 --round "$tag" --splice-bases "$bases"
