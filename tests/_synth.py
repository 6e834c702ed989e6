"""Deterministic synthetic references: word k (32 bases) of the
2-bit plane is SplitMix64 output k of the seed; contigs start on 64-base
boundaries of that plane.  numpy mirror of simmr_stage_synthetic."""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_words(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synthetic_contigs(contig_lens, seed: int):
    """list of uint8 ASCII arrays, one per contig"""
    bases = []
    off = 0
    for n in contig_lens:
        bases.append(off)
        off += (int(n) + 63) // 64 * 64
    words = splitmix64_words(seed, max(off // 32, 1))
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    codes = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[codes[b:b + int(n)]] for b, n in zip(bases, contig_lens)]


def write_fasta(path, contigs, names=None, width=80):
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            name = names[i] if names else f"synth_{i}"
            f.write(b">" + name.encode() + b"\n")
            b = c.tobytes()
            for j in range(0, len(b), width):
                f.write(b[j:j + width] + b"\n")
