"""Ad-hoc timing of the custom-model long-read path (BASELINE config 5 shape: k = 7, every ACGT 7-mer
observed, 1000 modelled positions): python profiles/microbench/custom_long_timing.py [reads] [per-read]
("per-read": 64 genomes x 10 Mbp with 1/(g+1) abundances, per-read lengths, uniform starts)"""
import sys
import time

import torch

from simmr_amd import CustomShortErrorProfile, Engine
from simmr_amd import model_io as _model


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    per_read = len(sys.argv) > 2 and sys.argv[2] == "per-read"
    n_genomes = 64 if per_read else 1  # BASELINE config 5: 64 genomes, abundances ~ 1 / (g + 1)
    eng = Engine(0)
    for g in range(n_genomes):
        eng.stage_synthetic(g, [10_000_000 if per_read else 100_000_000], 2 + g)
    w = [1.0 / (g + 1) for g in range(n_genomes)]
    counts = [int(-(-reads * x // sum(w))) for x in w]
    blob = _model.synthetic_long_model(kmer_size=7, n_positions=1000, seed=1, n_kmers=4 ** 7,
                                       lengths=(10000, 30000, 500))
    pod = CustomShortErrorProfile(blob)
    p = pod.pod()
    if per_read:
        from simmr_amd import _abi
        p.length_mode = _abi.LEN_PER_READ
        p.long_start_mode = _abi.START_UNIFORM
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.simulate_long_reads(list(range(n_genomes)), counts, p, 42, qual_offset=33)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"iter {it}: {reads} reads, {out.total_bases / 1e9:.2f} Gbases in {dt * 1e3:.1f} ms "
              f"(emit kernel {eng.last_emit_kernel_ms():.1f} ms) -> {out.total_bases / dt / 1e9:.1f} Gbases/s", flush=True)
        del out
    c = eng.counters()
    print("counters", list(c))


if __name__ == "__main__":
    main()
