#!/bin/bash
# L2 -> memory write requests of k_emit_philox in its three forms (one counter set per run, no tracing): how many of the
# write requests are whole 64-byte ones, how often the write path stalls.  Output: gpurun_out/write_path_probe/*.txt
S="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum;TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum;TCC_REQ_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum;TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_BUSY_avr"
mkdir -p gpurun_out/write_path_probe
tools/pmc_cmd.sh write_path_probe/slot16 k_emit_philox "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --steps 1 --warmup 0 > gpurun_out/write_path_probe/slot16.txt
tools/pmc_cmd.sh write_path_probe/compact k_emit_philox "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --layout compact --steps 1 --warmup 0 > gpurun_out/write_path_probe/compact.txt
tools/pmc_cmd.sh write_path_probe/text k_emit_philox "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --through-fastq --steps 1 --warmup 0 > gpurun_out/write_path_probe/text_plus_slot16_columns.txt
tools/pmc_cmd.sh write_path_probe/perfect k_emit_perfect_pe "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --profile perfect-short --steps 1 --warmup 0 > gpurun_out/write_path_probe/perfect.txt
rm -rf gpurun_out/write_path_probe/*/p*/
for f in slot16 compact text_plus_slot16_columns perfect; do echo "== $f"; cat gpurun_out/write_path_probe/$f.txt; done
