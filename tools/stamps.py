#!/usr/bin/env python3
"""Phase shares of k_emit_philox from the in-kernel cycle stamps of the diagnostic build
(make -C simmr_amd/csrc stamps).  usage: SIMMR_HIP_LIB=simmr_amd/csrc/variants/libsimmr_hip_stamps.so python tools/stamps.py [reads]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from simmr_amd import MinimalShortErrorProfile, _abi  # noqa: E402
from simmr_amd.engine import Engine, Reads  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)
eng.stage_synthetic(0, [100_000_000], 2)
prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
info = eng.pe_plan(0, prof, reads, 42, 0, reads // 2, (0, 0))
out = Reads.allocate(info.n_reads, info.total_bases, eng.device, qual_offset=33)
lib = _abi.load()
buf = (C.c_uint64 * 16)()
names = ["top barrier", "plan loads + record + metadata", "scan + map", "barrier after map", "staging loads -> LDS",
         "barrier after staging", "whole groups", "partial groups + loop"]
for it in range(2):
    eng.pe_plan(0, prof, reads, 42, 0, reads // 2, (0, 0))
    lib.simmr_debug_stamps(buf)  # clear
    eng.pe_emit(0, out)
    torch.cuda.synchronize()
    assert lib.simmr_debug_stamps(buf) == 0
    tot = sum(buf[:8])
    print(f"run {it}: emit kernel {eng.last_emit_kernel_ms():.3f} ms; wave-cycles by phase:")
    for i, n in enumerate(names):
        print(f"  {n:34s} {buf[i]:>16d}  {100.0 * buf[i] / tot:5.1f} %")
