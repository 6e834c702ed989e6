"""Ad-hoc timing of the custom-model long-read path (BASELINE config 5 shape: k = 7, every ACGT 7-mer
observed, 1000 modelled positions): python profiles/microbench/custom_long_timing.py [reads]"""
import sys
import time

import torch

from simmr_amd import CustomShortErrorProfile, Engine
from tests import _model


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    eng = Engine(0)
    eng.stage_synthetic(0, [100_000_000], 2)
    blob = _model.synthetic_long_model(kmer_size=7, n_positions=1000, seed=1, n_kmers=4 ** 7,
                                       lengths=(10000, 30000, 500))
    pod = CustomShortErrorProfile(blob)
    p = pod.pod()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.simulate_long_reads([0], [reads], p, 42, qual_offset=33)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"iter {it}: {reads} reads, {out.total_bases / 1e9:.2f} Gbases in {dt * 1e3:.1f} ms "
              f"(emit kernel {eng.last_emit_kernel_ms():.1f} ms) -> {out.total_bases / dt / 1e9:.1f} Gbases/s", flush=True)
        del out
    c = eng.counters()
    print("counters", list(c))


if __name__ == "__main__":
    main()
