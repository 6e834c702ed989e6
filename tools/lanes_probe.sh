#!/bin/bash
# k_emit_lanes (the bit-exact minimal-short emit) under counters: VERDICT r4 item 4(a).  One counter set per pass.
# usage: tools/lanes_probe.sh <out-dir-under-gpurun_out> [library]
out="$1"; lib="$2"
[ -n "$lib" ] && export SIMMR_HIP_LIB="$lib"
S="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY;SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS;SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM;SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES;FETCH_SIZE;WRITE_SIZE;TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
tools/pmc_cmd.sh "$out" "k_emit_lanes" "$S" -- python3 bench.py --no-cpu-baseline --no-other-mode --rng reference --steps 1 --warmup 0
