"""Oracle restatement of the custom (empirical) profile: bincode model reader,
alias-table PDFs (rand_distr WeightedAliasIndex), rand Uniform samplers, and the
reference's one unit test for the k-mer splice (custom_long.rs:300-343)."""
import ctypes as C

import numpy as np
import pytest

from tests import _model, _oracle


@pytest.fixture(scope="module")
def lib(oracle):
    oracle.orc_custom_new.restype = C.c_void_p
    oracle.orc_custom_new.argtypes = [C.c_char_p, C.c_uint64]
    oracle.orc_custom_simulate_errors.restype = C.c_int64
    oracle.orc_custom_simulate_errors.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p]
    oracle.orc_custom_model.restype = C.c_void_p
    oracle.orc_custom_model.argtypes = [C.c_void_p]
    oracle.orc_custom_get_read_length.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint16)]
    oracle.orc_custom_get_insert_size.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint16)]
    oracle.orc_custom_minimum_genome_size.restype = C.c_uint16
    oracle.orc_custom_minimum_genome_size.argtypes = [C.c_void_p]
    oracle.orc_custom_simulate_phred_scores.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    return oracle


@pytest.mark.provenance("reference-held")
def test_reference_simulate_errors_unit_test(lib):
    # custom_long.rs:300-343: 3-mer map {ACC -> CAT (1.0), ATC -> ATC, TCG -> TGT}; "ACCCG" -> "CATGT"
    e = _model.three_bit_encode
    probs = [(e("ACC"), [(e("ACC"), 0.0), (e("CAT"), 1.0)]), (e("ATC"), [(e("ATC"), 1.0)]), (e("TCG"), [(e("TGT"), 1.0)])]
    blob = _model.serialize_model([([1.0], [(30, 30)])], ([1.0], [(100, 100)]), probabilities=probs, kmer_size=3)
    c = lib.orc_custom_new(blob, len(blob))
    assert c
    for seed in (0, 1, 42, 2 ** 63):
        out = C.create_string_buffer(8)
        n = lib.orc_custom_simulate_errors(lib.orc_custom_model(c), b"ACCCG", 5, seed, out)
        assert n == 5 and out.raw[:5] == b"CATGT"
    # an N in the alternate k-mer is a deletion (three_bit_decode_kmer(.., skip_n = true)); the loop
    # bound keeps using the ORIGINAL length (custom_short.rs:475), so the reference then slices out of
    # range and panics -- the oracle reports that as -1 instead of inventing a behaviour
    probs = [(e("ACC"), [(e("ANC"), 1.0)])]
    blob = _model.serialize_model([([1.0], [(30, 30)])], ([1.0], [(100, 100)]), probabilities=probs, kmer_size=3)
    c = lib.orc_custom_new(blob, len(blob))
    out = C.create_string_buffer(8)
    assert lib.orc_custom_simulate_errors(lib.orc_custom_model(c), b"ACCGT", 5, 9, out) == -1


def test_model_roundtrip_and_pdfs(lib):
    blob = _model.synthetic_short_model()
    c = lib.orc_custom_new(blob, len(blob))
    assert c
    assert lib.orc_custom_new(blob[:-3], len(blob) - 3) is None  # truncated model is rejected
    assert lib.orc_custom_minimum_genome_size(c) == 2 * 140 + 200
    v = C.c_uint16()
    lens, ins = [], []
    for seed in range(4000):
        assert lib.orc_custom_get_read_length(c, seed, C.byref(v)) == 0
        lens.append(v.value)
        assert lib.orc_custom_get_insert_size(c, seed, C.byref(v)) == 0
        ins.append(v.value)
    lens, ins = np.array(lens), np.array(ins)
    assert 80 <= lens.min() and lens.max() < 200 and abs(lens.mean() - 140) < 2 and abs(lens.std() - 12) < 2
    assert 40 <= ins.min() and ins.max() < 400 and abs(ins.mean() - 200) < 5
    # per-position Phred: same seed at every position (custom_short.rs:349), position-specific PDFs,
    # positions past the model reuse the last PDF
    q = np.zeros(200, dtype=np.uint8)
    means = np.zeros(200)
    for seed in range(300):
        assert lib.orc_custom_simulate_phred_scores(c, 200, seed, C.c_void_p(q.ctypes.data)) == 0
        assert q.max() < 70
        means += q
    means /= 300
    assert means[:10].mean() > means[100:120].mean() + 5
    assert abs(means[125:200].mean() - means[119]) < 1.5


def test_alias_sampler_distribution(lib):
    w = np.array([0.05, 0.0, 0.5, 0.25, 0.2])
    a = (C.c_uint8 * 256)()  # opaque orc_alias storage
    assert lib.orc_alias_new(C.c_void_p(w.ctypes.data), 5, a) == 0
    r = _oracle.Rng()
    lib.orc_rng_seed_from_u64(C.byref(r), 11)
    lib.orc_alias_sample.restype = C.c_uint32
    n = 200000
    counts = np.bincount([lib.orc_alias_sample(a, C.byref(r)) for _ in range(n)], minlength=5)
    assert counts[1] == 0
    assert np.abs(counts / n - w).max() < 0.005


def test_long_path_with_a_custom_model(oracle):
    """simulate.rs:497-503 with CustomShortErrorProfile: qualities from the per-position PDFs (one value from
    position n_quality - 1 on), then the k-mer splice, then an identity simulate_point_mutations.  With one
    alternate per k-mer the splice needs no random choice, so a plain Python walk predicts it."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _synth
    rng = np.random.default_rng(4)
    k = 4
    table = {}
    probs = []
    for idx in rng.choice(4 ** k, size=120, replace=False):
        key = sum(((int(idx) >> (2 * j)) & 3) << (3 * j) for j in range(k))
        alt = key ^ (int(rng.integers(1, 4)) << (3 * int(rng.integers(0, k))))
        alt = sum((((alt >> (3 * j)) & 7) & 3) << (3 * j) for j in range(k))  # keep the fields in ACGT
        probs.append((key, [(alt, 2.5)]))
        table[key] = alt
    quality = [([0.2, 0.5, 0.3], [(10 + p % 7, 12 + p % 7), (20, 20), (30, 33)]) for p in range(25)]
    blob = _model.serialize_model(quality, ([1.0], [(900, 900)]), probabilities=probs, kmer_size=k,
                                  read_length_mean=900.0, insert_size_mean=0.0, is_long=True)
    prof = CustomShortErrorProfile(blob)
    assert prof.is_long_read()
    contigs = _synth.synthetic_contigs([5000, 1200], 3)
    g = _oracle.HostGenome(contigs)
    res = _oracle.simulate_long(oracle, [g], [60], prof.pod(), 11)
    out = res.trimmed()
    # one length for the run (simulate.rs:358): get_random_read_length = floor(N(read_length_mean, read_length_std))
    assert out["seq_off"][-1] == out["seq"].size and 800 < res.const_len < 1000 and (np.diff(out["seq_off"]) <= res.const_len).all()
    lut = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    for r in range(60):
        s, e = int(out["start"][r]), int(out["end"][r])
        seq = bytearray(contigs[int(out["contig"][r])][s:e].tobytes())
        n = len(seq)
        for i in range(n - k + 1):
            key = sum(lut[seq[i + j]] << (3 * j) for j in range(k))
            if key in table:
                alt = table[key]
                seq[i:i + k] = bytes(b"ACGT"[(alt >> (3 * j)) & 7] for j in range(k))
        got = out["seq"][out["seq_off"][r]:out["seq_off"][r + 1]]
        assert got.tobytes() == bytes(seq), r
        q = out["qual"][out["seq_off"][r]:out["seq_off"][r + 1]]
        assert (q[24:] == q[24]).all() and set(np.unique(q)) <= set(range(10, 34))
    # the per-read length extension: every read draws its own floor(Normal(mean, std)) length
    pod = prof.pod()
    pod.length_mode = 1
    per = _oracle.simulate_long(oracle, [g], [40], pod, 11).trimmed()
    lens = np.diff(per["seq_off"].astype(np.int64))
    assert len(set(lens.tolist())) > 10 and 500 < lens.mean() < 1000  # reads on the 1200 nt sequence are re-cut at its end


def test_counter_mode_splice_specification(oracle):
    """SIMMR_RNG_PHILOX with a custom long-read model (include/simmr_hip.h; oracle/custom.c: ctr_splice_tables,
    orc_custom_simulate_errors_philox) — the specification the HIP kernel is compared with bit for bit, checked here for
    what it promises: (1) with one alternate per k-mer no draw matters, so both modes give the reference walk's bytes;
    (2) lengths, positions and qualities are the reference mode's; (3) on a model of single-base k-mers, where a visited
    base is replaced independently of its neighbours, every alternate's frequency follows its weight (chi-square), for
    dominant, balanced and self-less lists alike; (4) a shard of a run equals that range of the whole run."""
    from scipy.stats import chisquare
    from simmr_amd import CustomShortErrorProfile, _abi
    from tests import _synth
    lut = np.full(256, 255, np.uint8)
    for ch, v in {65: 0, 67: 1, 71: 2, 84: 3}.items():
        lut[ch] = v
    contigs = _synth.synthetic_contigs([400_000], 8)
    g = _oracle.HostGenome(contigs)
    # (1) deterministic lists
    k = 3
    probs = [(sum(((i >> (2 * j)) & 3) << (3 * j) for j in range(k)), [(sum((((i * 7 + 3) >> (2 * j)) & 3) << (3 * j) for j in range(k)), 1.5)])
             for i in range(0, 64, 3)]
    q3 = [([1.0], [(30, 30)])] * 3
    det = _model.serialize_model(q3, ([1.0], [(700, 700)]), probabilities=probs, kmer_size=k, read_length_mean=700.0,
                                 insert_size_mean=0.0, is_long=True)
    a = _oracle.simulate_long(oracle, [g], [50], CustomShortErrorProfile(det, _abi.RNG_REFERENCE).pod(), 3).trimmed()
    b = _oracle.simulate_long(oracle, [g], [50], CustomShortErrorProfile(det, _abi.RNG_PHILOX).pod(), 3).trimmed()
    for col in ("seq", "qual", "seq_off", "start", "end"):
        assert np.array_equal(a[col], b[col]), col
    # (3) the law, on lists with a dominant self, a balanced list, a list without self, a list of one
    w = [[8.0, 1.0, 0.5, 0.5], [1.0, 1.0, 1.0, 1.0], [0.0, 3.0, 0.0, 4.0], None]  # (G's list has no G: (C, 3), (A, 0), (T, 4))
    probs = [(0, [(j, w[0][j]) for j in range(4)]), (1, [(j, w[1][j]) for j in range(4)]), (2, [(j, w[2][j]) for j in (1, 0, 3)]),
             (3, [(3, 0.25)])]
    one = _model.serialize_model(q3, ([1.0], [(6000, 6000)]), probabilities=probs, kmer_size=1, read_length_mean=6000.0,
                                 insert_size_mean=0.0, is_long=True)
    ref_pod, ctr_pod = CustomShortErrorProfile(one, _abi.RNG_REFERENCE).pod(), CustomShortErrorProfile(one, _abi.RNG_PHILOX).pod()
    ra = _oracle.simulate_long(oracle, [g], [120], ref_pod, 9).trimmed()
    rb = _oracle.simulate_long(oracle, [g], [120], ctr_pod, 9).trimmed()
    for col in ("qual", "seq_off", "start", "end", "contig", "read_id"):  # (2)
        assert np.array_equal(ra[col], rb[col]), col
    assert not np.array_equal(ra["seq"], rb["seq"])
    src = lut[np.concatenate([contigs[0][int(rb["start"][r]):int(rb["end"][r])] for r in range(120)])]
    for out in (ra, rb):
        dst = lut[out["seq"]]
        assert src.size == dst.size > 600_000
        for c0 in range(3):
            obs = np.bincount(dst[src == c0], minlength=4).astype(float)
            pr = np.array(w[c0], dtype=np.float64)
            keep = pr > 0
            assert obs[~keep].sum() == 0
            assert chisquare(obs[keep], pr[keep] / pr.sum() * obs.sum()).pvalue > 1e-4, (c0, obs)
        assert (dst[src == 3] == 3).all()
    # (4) a shard of the counter-mode run
    part = _oracle.simulate_long(oracle, [g], [120], ctr_pod, 9, first=37, count=40).trimmed()
    lo, hi = int(rb["seq_off"][37]), int(rb["seq_off"][77])
    assert np.array_equal(part["seq"], rb["seq"][lo:hi]) and np.array_equal(part["qual"], rb["qual"][lo:hi])


def _splice_law_from_tables(T24, thr, alias, alt, n):
    """P(alternate code) that the counter mode's two-level draw of one k-mer ENCODES, exactly (a count of the 2^32 words X):
    level 1, X >> 8 < T24 -> self; level 2, Z = (X - (T24 << 8)) << (24 - e) over the remaining 2^(e + 8) words,
    column c = (Z n) >> 32, ((Z n) & 0xffffffff) >> 8 < thr[c] ? alternate c : alternate alias[c]
    (include/simmr_hip.h; Z = t * 2^(24 - e), t = 0 .. 2^(e + 8) - 1: the t of a column and of its lower part are ranges)."""
    rest = (1 << 24) - T24          # 2^e level-1 values go to level 2
    e = rest.bit_length() - 1
    assert rest == 1 << e and 1 <= e <= 24
    S = 1 << (24 - e)
    K = 1 << (e + 8)
    ceil_div = lambda a, b: -(-a // b)
    counts = {}
    for c in range(n):
        lo = ceil_div(c << 32, S * n)
        mid = min(ceil_div((c << 32) + (int(thr[c]) << 8), S * n), ceil_div((c + 1) << 32, S * n))
        hi = min(ceil_div((c + 1) << 32, S * n), K)
        lo, mid = min(lo, K), min(mid, K)
        counts[int(alt[c])] = counts.get(int(alt[c]), 0) + max(mid - lo, 0)
        counts[int(alt[int(alias[c])])] = counts.get(int(alt[int(alias[c])]), 0) + max(hi - mid, 0)
    assert sum(counts.values()) == K
    return counts


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_counter_mode_splice_tables_encode_the_reference_law(oracle, which):
    """The table builder is the one piece the product and the CPU specification share line for line, so a bit-for-bit
    comparison of their outputs cannot find an error in it.  Here the law the tables encode is counted exactly
    over the 2^32 words of a draw and compared with the reference's law P(alternate j) = w_j / sum(w)
    (custom_short.rs:497-503) for random lists — dominant self, no self, self listed twice, zero weights, one entry, forty —
    for the oracle's builder (oracle/custom.c) and for the product's (csrc/custom_model.hpp through libsimmr_host.so)."""
    if which == "oracle":
        fn = oracle.orc_ctr_splice_tables
    else:
        import os
        from pathlib import Path
        host = C.CDLL(os.environ.get("SIMMR_HOST_LIB") or str(Path(__file__).resolve().parent.parent / "simmr_amd" / "host" / "libsimmr_host.so"))
        fn = host.simmr_host_ctr_splice_tables
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(2025)
    worst = 0.0
    for trial in range(400):
        n = int(rng.choice([1, 2, 3, 4, 6, 9, 20, 40]))
        self_code = 1000
        alt = rng.choice(np.arange(1, 900), size=n, replace=False).astype(np.uint32)
        w = rng.uniform(0.0, 3.0, n).astype(np.float32)
        w[rng.random(n) < 0.15] = 0.0
        shape = trial % 5
        if shape != 1 and n >= 1:          # a self entry (shape 1: none at all)
            alt[0] = self_code
            w[0] = np.float32(rng.choice([0.2, 5.0, 50.0, 5000.0]))
        if shape == 2 and n >= 3:          # self listed twice
            alt[2] = self_code
            w[2] = np.float32(1.0)
        if shape == 3:                     # nearly all weight on self: level 2 is the smallest power of two
            w[1:] *= np.float32(1e-5)
        if w.sum() == 0:
            w[-1] = np.float32(1.0)
        has_self = 0 if trial % 11 == 0 else 1   # a k-mer with an N has no "self", whatever its list says
        thr = np.zeros(n, np.uint32)
        alias = np.zeros(n, np.uint32)
        T24 = fn(alt.ctypes.data, w.ctypes.data, n, self_code, has_self, thr.ctypes.data, alias.ctypes.data)
        assert 0 <= T24 < 1 << 24 and (alias < n).all() and (thr <= 1 << 24).all()
        counts = _splice_law_from_tables(int(T24), thr, alias, alt, n)
        if T24:
            assert has_self and self_code in alt
            counts[self_code] = counts.get(self_code, 0) + (int(T24) << 8)
        W = float(w.astype(np.float64).sum())
        for code in set(int(a) for a in alt):
            want = float(w[alt == code].astype(np.float64).sum()) / W
            got = counts.get(code, 0) / 2.0 ** 32
            worst = max(worst, abs(got - want))
            assert abs(got - want) <= (n + 1) * 2.0 ** -23, (trial, n, code, got, want, T24)
        assert sum(counts.values()) == 1 << 32
    assert worst > 0.0  # (the tables are integer: some rounding must show)
