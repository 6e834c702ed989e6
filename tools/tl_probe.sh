#!/bin/bash
# The whole-line TEXT kernel (text_lines.hip) under counters and against its timing ablations, one box, one call.
# usage: tools/tl_probe.sh <out-dir-under-gpurun_out>
out="$1"
mkdir -p "gpurun_out/$out"
export SIMMR_TEXT_FORM=2
B="python3 bench.py --no-cpu-baseline --no-other-mode --through-fastq"
for lib in "" variants/libsimmr_hip_tl_nostore.so variants/libsimmr_hip_tl_plainwrite.so; do
  name="${lib:-product}"
  [ -n "$lib" ] && [ ! -f "simmr_amd/csrc/$lib" ] && continue
  for r in 1 2; do
    o=$(env ${lib:+SIMMR_HIP_LIB=$PWD/simmr_amd/csrc/$lib} timeout -k 10 200 $B --steps 10 --warmup 2 2>/dev/null | tail -1)
    echo "$name round $r: $(echo "$o" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.3f ms_per_step=%.3f" % (d["roofline"]["kernel_ms"], d["ms_per_step"]))' 2>&1)"
  done
done | tee "gpurun_out/$out/ablations.log"
S="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT;SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES;SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY;TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN"
tools/pmc_cmd.sh "$out/pmc" "k_emit_text_lines" "$S" -- $B --steps 1 --warmup 0
