#!/usr/bin/env python3
"""One GPU's share of BASELINE configs[3] (125 genomes of 5 Mbp x 1 M reads of minimal-short, counter mode) in ONE plan
(simmr_pe_plan_multi): reads per second of plan + emit, columns resident in HBM.  usage: tools/c4_share_timing.py [slot16] [full]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from simmr_amd import MinimalShortErrorProfile, _abi
from simmr_amd.engine import Engine

eng = Engine(0)
n_g, per = 125, 1_000_000
slots = list(range(100, 100 + n_g))
for g, s in enumerate(slots):
    eng.stage_synthetic(s, [5_000_000], 1000 + g)
full = "full" in sys.argv[1:]  # the plan's draws from Philox counters too (SIMMR_RNG_PHILOX_FULL)
prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL if full else _abi.RNG_PHILOX).pod()
if "slot16" in sys.argv[1:]:
    eng.set_read_slots(16)
print("rng:", "philox-full" if full else "philox (plan from the reference's streams)", "| layout:", "slot16" if "slot16" in sys.argv[1:] else "compact")
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = eng.simulate_pe_reads_multi(slots, [per] * n_g, prof, 42, qual_offset=33)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("run %d: %d reads in %.2f ms = %.3g reads/s (emit kernel %.2f ms, plan %.2f ms)" % (it, r.n_reads, dt * 1e3, r.n_reads / dt, eng.last_emit_kernel_ms(), eng.last_plan_ms()))
    del r
