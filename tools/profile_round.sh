#!/bin/bash
# The round's rocprofv3 evidence in one go (run on the GPU box through gpurun): per-kernel stats of the bench
# commands and the PMC passes of the default and perfect-short benches.  Output: gpurun_out/prof_<tag>/ .
tag="${1:-r5}"; out="gpurun_out/prof_$tag"; mkdir -p "$out"
export TMPDIR=/tmp
stats() {  # name, bench args...
  name="$1"; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name" -o t -- python3 bench.py --no-cpu-baseline --no-overlap-side "$@" > "$out/${name}_bench.log" 2>&1
  f=$(find "$out/$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_$name.csv"
  grep '^{"metric"' "$out/${name}_bench.log" | tail -1 > "$out/bench_${name}_under_rocprof.json"  # (rocprofv3 logs after it)
}
stats bench_default
stats bench_perfect_short --profile perfect-short
stats bench_custom_long --profile custom-long --reads 1000000
stats bench_custom_long_reference --profile custom-long --rng reference --reads 1000000
stats bench_custom_short --profile custom-short --reads 20000000
stats bench_minimal_long --profile minimal-long --reads 10000000
stats bench_through_fastq --through-fastq --no-other-mode
SIMMR_TEXT_FORM=2 stats bench_through_fastq_whole_lines --through-fastq --no-other-mode   # (text_lines.hip, opt-in)
S="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD;GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
tools/pmc_cmd.sh "prof_$tag/pmc_default" "k_emit_philox<false, false, true, false" "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --steps 1 --warmup 0 > "$out/pmc_default.txt"
tools/pmc_cmd.sh "prof_$tag/pmc_compact" "k_emit_philox<false, false, true, false" "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES;SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY" -- python3 bench.py --no-cpu-baseline --no-other-mode --layout compact --steps 1 --warmup 0 > "$out/pmc_compact.txt"
tools/pmc_cmd.sh "prof_$tag/pmc_perfect" k_emit_perfect_pe "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_WAVES" -- python3 bench.py --no-cpu-baseline --profile perfect-short --steps 1 --warmup 0 > "$out/pmc_perfect.txt"
tools/pmc_cmd.sh "prof_$tag/pmc_custom_long" k_custom_long_splice "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" -- python3 bench.py --no-cpu-baseline --no-other-mode --profile custom-long --reads 1000000 --steps 1 --warmup 0 > "$out/pmc_custom_long.txt"
tools/pmc_cmd.sh "prof_$tag/pmc_custom_long_reference" k_custom_long_splice "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" -- python3 bench.py --no-cpu-baseline --no-other-mode --profile custom-long --rng reference --reads 1000000 --steps 1 --warmup 0 > "$out/pmc_custom_long_reference.txt"
# (the TEXT form only: <HAS_EXC, COPY_ONLY, CACHED, TEXT, ...> = <false, false, true, true, ...>; the command's one column launch is left out)
tools/pmc_cmd.sh "prof_$tag/pmc_through_fastq" "k_emit_philox<false, false, true, true" "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES;SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY" -- python3 bench.py --no-cpu-baseline --no-other-mode --through-fastq --steps 1 --warmup 0 > "$out/pmc_through_fastq.txt"
tools/pmc_cmd.sh "prof_$tag/pmc_lanes" k_emit_lanes "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY;SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS;SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;FETCH_SIZE;WRITE_SIZE" -- python3 bench.py --no-cpu-baseline --no-other-mode --rng reference --steps 1 --warmup 0 > "$out/pmc_lanes.txt"
SIMMR_TEXT_FORM=2 tools/pmc_cmd.sh "prof_$tag/pmc_text_lines" k_emit_text_lines "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD;TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- python3 bench.py --no-cpu-baseline --no-other-mode --through-fastq --steps 1 --warmup 0 > "$out/pmc_text_lines.txt"
rm -rf "$out"/bench_*/ "$out"/pmc_*/p*/  # keep the summaries, drop the raw traces
ls "$out"
