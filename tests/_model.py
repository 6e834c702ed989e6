"""bincode 1.3.3 (default options) writer for shared::encoding::ErrorModelParams
(shared/src/encoding.rs:82-117) — used to build synthetic custom error models
for the tests (the reference ships no model file)."""
import struct

import numpy as np


def _bins(density, ranges, num_bins=None, bin_width=1):
    out = struct.pack("<QQ", len(ranges) if num_bins is None else num_bins, bin_width)
    out += struct.pack("<Q", len(density)) + b"".join(struct.pack("<d", float(x)) for x in density)
    out += struct.pack("<Q", len(ranges)) + b"".join(struct.pack("<II", int(a), int(b)) for a, b in ranges)
    return out


def three_bit_encode(kmer: str) -> int:
    code = 0
    for i, ch in enumerate(kmer):
        code |= "ACGTN".index(ch) << (3 * i)
    return code


def serialize_model(quality_bins, read_length_bins, insert_size_bins=None, probabilities=(), kmer_size=7,
                    bin_size=1, insert_size_mean=150.0, insert_size_std=75.0, read_length_mean=150.0,
                    read_length_std=15.0, is_long=False) -> bytes:
    """quality_bins: list of (density, ranges) per read position; *_bins: (density, ranges)."""
    out = struct.pack("<Q", bin_size)
    out += struct.pack("<Q", len(quality_bins)) + b"".join(_bins(d, r) for d, r in quality_bins)
    out += struct.pack("<B", 3) + struct.pack("<Q", kmer_size)
    out += struct.pack("<Q", len(probabilities))
    for kmer, alts in probabilities:
        out += struct.pack("<I", kmer) + struct.pack("<Q", len(alts))
        out += b"".join(struct.pack("<If", int(a), float(w)) for a, w in alts)
    out += struct.pack("<dd", insert_size_mean, insert_size_std)
    if insert_size_bins is None:
        out += b"\x00"
    else:
        out += b"\x01" + _bins(*insert_size_bins)
    out += struct.pack("<dd", read_length_mean, read_length_std) + _bins(*read_length_bins)
    out += struct.pack("<B", 1 if is_long else 0)
    return out


def synthetic_short_model(n_positions=120, seed=3, mean_len=140, sd_len=12, mean_insert=200, sd_insert=40):
    """Shaped like simmrd output (simmrd/src/probability.rs:119-166): per position
    one-score bins (i, i) for scores 0..69 with KDE-like densities; read length and
    insert size as 5-wide bins."""
    rng = np.random.default_rng(seed)
    quality = []
    for p in range(n_positions):
        centre = 36.0 - 12.0 * p / n_positions + rng.normal(0, 0.5)
        x = np.arange(70)
        dens = np.exp(-0.5 * ((x - centre) / (4.0 + 3.0 * p / n_positions)) ** 2) + 1e-4
        dens[rng.integers(0, 70, 3)] = 0.0  # empty bins do occur
        quality.append((dens / dens.sum(), [(i, i) for i in range(70)]))

    def hist(mean, sd, lo, hi, width):
        edges = list(range(lo, hi, width))
        centres = np.array([e + width / 2 for e in edges])
        d = np.exp(-0.5 * ((centres - mean) / sd) ** 2)
        return d / d.sum(), [(e, e + width - 1) for e in edges]
    return serialize_model(quality, hist(mean_len, sd_len, 80, 200, 5), hist(mean_insert, sd_insert, 40, 400, 10),
                           insert_size_mean=float(mean_insert), insert_size_std=float(sd_insert),
                           read_length_mean=float(mean_len), read_length_std=float(sd_len))
