"""bincode 1.3.3 (default options) writer for shared::encoding::ErrorModelParams
(shared/src/encoding.rs:82-117), the wire format of a `simmrd` error model: little-endian fixed-width
integers, usize / lengths as u64, Option as a u8 tag, bool as u8, tuples inline, fields in declaration
order.  The library reads the same bytes (csrc/custom_model.hpp).  The reference ships no model file, so
the synthetic generators below stand in for `simmrd` output in the tests and in bench.py."""
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
        out += struct.pack("<I", int(kmer) & 0xFFFFFFFF) + struct.pack("<Q", len(alts))
        out += b"".join(struct.pack("<If", int(a) & 0xFFFFFFFF, float(w)) for a, w in alts)
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


def synthetic_long_model(kmer_size=7, n_positions=300, seed=5, n_kmers=3000, lengths=(1500, 6000, 100),
                         with_n=True, deletion=False, read_length_mean=None, read_length_std=None, max_alts=6):
    """A long-read model (is_long) with k-mer probabilities shaped like simmrd's: per observed k-mer a
    list of (alternate, weight) with the k-mer itself dominant and a few substituted variants.  Some keys
    contain an N (three_bit_encode_kmer accepts it), one key is listed twice (the HashMap keeps the last)."""
    rng = np.random.default_rng(seed)
    quality = []
    for p in range(n_positions):
        centre = 22.0 - 8.0 * p / n_positions + rng.normal(0, 0.5)
        x = np.arange(50)
        dens = np.exp(-0.5 * ((x - centre) / 5.0) ** 2) + 1e-4
        quality.append((dens / dens.sum(), [(i, i) for i in range(50)]))
    k = kmer_size
    n_all = 4 ** k
    picks = rng.choice(n_all, size=min(n_kmers, n_all), replace=False)
    probs = []

    def code_of(idx):  # ACGT k-mer number -> 3-bit code
        c = 0
        for j in range(k):
            c |= ((idx >> (2 * j)) & 3) << (3 * j)
        return c
    for idx in picks:
        key = code_of(int(idx))
        n_alt = int(rng.integers(1, max_alts + 1))  # simmrd keeps up to --max-alt-kmers (default 20) per k-mer
        alts = [(key, float(np.float32(rng.uniform(5, 50))))]
        for _ in range(n_alt - 1):
            pos = int(rng.integers(0, k))
            base = int(rng.integers(0, 4))
            alt = (key & ~(7 << (3 * pos))) | (base << (3 * pos))
            alts.append((alt, float(np.float32(rng.uniform(0.0, 3.0)))))
        order = rng.permutation(len(alts))
        probs.append((key, [alts[i] for i in order]))
    if with_n:
        for _ in range(40):  # keys with one N; alternates resolve the N to a base (no deletion)
            key = code_of(int(rng.integers(0, n_all)))
            pos = int(rng.integers(0, k))
            keyn = (key & ~(7 << (3 * pos))) | (4 << (3 * pos))
            probs.append((keyn, [(key, 1.0), (key ^ (1 << (3 * ((pos + 1) % k))), 0.5)]))
    # a repeated key: the later list wins
    first_key = probs[0][0]
    probs.append((first_key, [(first_key ^ 1, 2.0), (first_key, 1.0)]))
    if deletion:
        # every ACGT k-mer starting with "AC" may lose a base: alternate with an N field (decoded with skip_n)
        for rest in range(4 ** max(k - 2, 0)):
            key = code_of((rest << 4) | 0b0100)
            probs.append((key, [(key, 1.0), ((key & ~7) | 4, 1.0)]))
    lo, hi, width = lengths
    edges = list(range(lo, hi, width))
    centres = np.array([e + width / 2 for e in edges])
    d = np.exp(-0.5 * ((centres - (lo + hi) / 2) / ((hi - lo) / 5)) ** 2)
    return serialize_model(quality, (d / d.sum(), [(e, e + width - 1) for e in edges]), None,
                           probabilities=probs, kmer_size=k, insert_size_mean=0.0, insert_size_std=0.0,
                           read_length_mean=float((lo + hi) / 2 if read_length_mean is None else read_length_mean),
                           read_length_std=float((hi - lo) / 5 if read_length_std is None else read_length_std), is_long=True)
