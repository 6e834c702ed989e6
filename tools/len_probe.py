#!/usr/bin/env python3
"""Emit-kernel time of the counter mode for fixed read lengths (probe for the store path: L = 160 has no partial
groups and every 16-byte store is aligned; L = 150 has one partial group per read).  usage: python tools/len_probe.py [pairs]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from simmr_amd import MinimalShortErrorProfile, _abi  # noqa: E402
from simmr_amd.engine import Engine, Reads  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
eng = Engine(0)
eng.stage_synthetic(0, [100_000_000], 2)
cases = ((150, 15.0), (150, 0.0), (160, 0.0), (144, 0.0), (151, 0.0))
if len(sys.argv) > 3:
    cases = ((int(sys.argv[2]), float(sys.argv[3])),)
for L, std in cases:
    prof = MinimalShortErrorProfile(read_length=L, insert_size=L, read_length_std=std, insert_size_std=0.0,
                                    rng_mode=_abi.RNG_PHILOX).pod()
    info = eng.pe_plan(0, prof, 2 * pairs, 42, 0, pairs, (0, 0))
    out = Reads.allocate(info.n_reads, info.total_bases, eng.device, qual_offset=33)
    ms = []
    for it in range(3):
        eng.pe_plan(0, prof, 2 * pairs, 42, 0, pairs, (0, 0))
        eng.pe_emit(0, out)
        torch.cuda.synchronize()
        ms.append(eng.last_emit_kernel_ms())
    best = min(ms)
    print(f"L={L} std={std}: emit {best:.3f} ms, {info.total_bases / best / 1e6:.1f} Gbases/s, "
          f"{info.total_bases * 2.25 / best / 1e9:.3f} TB/s algorithmic")
    del out
    torch.cuda.empty_cache()
