for v in full hotstore nostores; do
  if [ $v = full ]; then export SIMMR_HIP_LIB=simmr_amd/csrc/libsimmr_hip.so; else export SIMMR_HIP_LIB=simmr_amd/csrc/variants/libsimmr_hip_$v.so; fi
  echo "== $v"
  tools/pmc_cmd.sh clk_$v k_emit_philox "GRBM_GUI_ACTIVE" -- python bench.py --no-cpu-baseline --no-other-mode --steps 3 --warmup 1
  grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/clk_$v/p1.log
done
