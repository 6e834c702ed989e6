"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit for bit.

Everything the reference makes deterministic under --seed is compared with
equality: bases, qualities, CSR offsets, start/end, contig, read ids, flags.
Where the reference itself falls back to OS entropy in a seeded run (mate-2
Option<u64> == None, simulate.rs:266,270) both sides use the documented
substitute (include/simmr_hip.h) and the reads are flagged.
"""
import numpy as np
import pytest


from simmr_amd import (MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectLongErrorProfile,
                       PerfectShortErrorProfile, _abi)
from tests import _oracle, _synth

pytestmark = pytest.mark.gpu

COLS = ("seq_off", "start", "end", "contig", "read_id", "flags", "qual", "seq")


def assert_same(dev: dict, ora: dict, cols=COLS, what=""):
    for c in cols:
        a, b = dev[c], ora[c]
        assert a.shape == b.shape, f"{what}{c}: shape {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            bad = np.flatnonzero(a != b)
            raise AssertionError(f"{what}{c}: {bad.size} of {a.size} differ, first at {bad[:8]}: "
                                 f"dev={a[bad[:8]]} oracle={b[bad[:8]]}")


@pytest.fixture(scope="module")
def genome_1m(engine):
    contigs = _synth.synthetic_contigs([1_000_000], 1)
    engine.stage_synthetic(0, [1_000_000], 1)
    return _oracle.HostGenome(contigs)


@pytest.fixture(scope="module")
def genome_multi(engine):
    lens = [300_000, 90_001, 30_017, 70_000, 64, 123_457]
    contigs = _synth.synthetic_contigs(lens, 7)
    # contig 4 is too small for any profile: the reference filters it (main.rs:117-162)
    keep = [0, 1, 2, 3, 5]
    engine.stage_genome(1, [contigs[i] for i in keep])
    return _oracle.HostGenome([contigs[i] for i in keep])


def test_staging_roundtrip(engine, genome_1m, genome_multi):
    got = engine.unstage(0, 0, 0, 1_000_000)
    assert np.array_equal(got, genome_1m.contigs[0])
    for c, ref in enumerate(genome_multi.contigs):
        assert np.array_equal(engine.unstage(1, c, 0, ref.size), ref)
    assert engine.genome_info(1) == (5, sum(c.size for c in genome_multi.contigs))


def test_staging_exceptions(engine):
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5000)].copy()
    seq[100:180] = ord("N")
    seq[1000] = ord("-")
    seq[4999] = ord("N")
    engine.stage_genome(2, [seq, seq[:777].copy()])
    assert np.array_equal(engine.unstage(2, 0, 0, 5000), seq)
    assert np.array_equal(engine.unstage(2, 1, 3, 700), seq[3:703])


# BASELINE.json configs[0]: perfect-short, 1 Mbp synthetic, --num-reads 10000, seed 42
def test_c1_perfect_short_bit_exact(engine, oracle, genome_1m):
    prof = PerfectShortErrorProfile(150, 150).pod()
    dev = engine.simulate_pe_reads_from_genome(0, prof, 10000, 42)
    assert dev.n_reads == 10000 and dev.total_bases == 10000 * 150
    ora = _oracle.simulate_pe(oracle, genome_1m, prof, 10000, 42)
    assert_same(dev.to_host(), ora.trimmed())
    assert (dev.to_host()["qual"] == 60).all()
    assert np.array_equal(dev.to_host()["read_id"], np.repeat(np.arange(5000, dtype=np.uint32), 2))


@pytest.mark.parametrize("L,I", [(20, 20), (33, 5), (150, 600), (7, 3), (250, 100), (16, 16)])
def test_perfect_short_lengths(engine, oracle, genome_multi, L, I):
    prof = PerfectShortErrorProfile(L, I).pod()
    dev = engine.simulate_pe_reads_from_genome(1, prof, 3001, 7, read_id_base=17, qual_offset=33)
    ora = _oracle.simulate_pe(oracle, genome_multi, prof, 3001, 7, read_id_base=17, qual_offset=33)
    assert_same(dev.to_host(), ora.trimmed())


def test_perfect_short_exceptions_revcomp(engine, oracle):
    rng = np.random.default_rng(11)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3000)].copy()
    seq[rng.integers(0, 3000, 300)] = ord("N")
    seq[rng.integers(0, 3000, 100)] = ord("-")
    engine.stage_genome(3, [seq])
    g = _oracle.HostGenome([seq])
    prof = PerfectShortErrorProfile(50, 70).pod()
    dev = engine.simulate_pe_reads_from_genome(3, prof, 2000, 3)
    ora = _oracle.simulate_pe(oracle, g, prof, 2000, 3)
    assert_same(dev.to_host(), ora.trimmed())


def test_pe_sharding_matches_whole(engine, oracle, genome_multi):
    prof = PerfectShortErrorProfile(100, 150).pod()
    whole = _oracle.simulate_pe(oracle, genome_multi, prof, 4000, 99, read_id_base=5).trimmed()
    for first, count in [(0, 700), (700, 1), (701, 1299), (1990, 10 ** 9)]:
        dev = engine.simulate_pe_reads_from_genome(1, prof, 4000, 99, first=first, count=count, read_id_base=5)
        n = min(count, 2000 - first)
        d = dev.to_host()
        assert dev.n_reads == 2 * n
        lo, hi = 2 * first, 2 * (first + n)
        base = whole["seq_off"][lo]
        assert np.array_equal(d["seq_off"], whole["seq_off"][lo:hi + 1] - base)
        for c in ("start", "end", "contig", "read_id", "flags"):
            assert np.array_equal(d[c], whole[c][lo:hi]), c
        assert np.array_equal(d["seq"], whole["seq"][base:whole["seq_off"][hi]])


# minimal-short: every draw comes from the reference's own ChaCha12 streams
@pytest.mark.parametrize("seed,reads", [(42, 10000), (1, 2222)])
def test_minimal_short_bit_exact(engine, oracle, genome_1m, seed, reads):
    prof = MinimalShortErrorProfile().pod()
    dev = engine.simulate_pe_reads_from_genome(0, prof, reads, seed)
    ora = _oracle.simulate_pe(oracle, genome_1m, prof, reads, seed)
    assert dev.total_bases == ora.total_bases
    assert_same(dev.to_host(), ora.trimmed())


@pytest.mark.parametrize("L,I,q", [(150, 150, 20), (300, 500, 30), (40, 10, 5), (600, 300, 45)])
def test_minimal_short_params(engine, oracle, genome_multi, L, I, q):
    prof = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=q).pod()
    dev = engine.simulate_pe_reads_from_genome(1, prof, 1500, 5, qual_offset=33)
    ora = _oracle.simulate_pe(oracle, genome_multi, prof, 1500, 5, qual_offset=33, max_len=4096)
    assert_same(dev.to_host(), ora.trimmed())


def test_minimal_short_exceptions(engine, oracle):
    rng = np.random.default_rng(12)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 20000)].copy()
    seq[rng.integers(0, 20000, 2000)] = ord("N")
    seq[5000:5400] = ord("N")
    engine.stage_genome(3, [seq])
    g = _oracle.HostGenome([seq])
    prof = MinimalShortErrorProfile(mean_phred_score=12).pod()
    dev = engine.simulate_pe_reads_from_genome(3, prof, 3000, 8)
    ora = _oracle.simulate_pe(oracle, g, prof, 3000, 8)
    assert_same(dev.to_host(), ora.trimmed())


def test_minimal_short_tolerances(engine, genome_1m):
    """substitution rate 0.013404 +-2 % at mean Phred 30, Phred mean 29.5."""
    prof = MinimalShortErrorProfile().pod()
    engine.counters_reset()
    dev = engine.simulate_pe_reads_from_genome(0, prof, 400_000, 2024)
    c = engine.counters()
    assert c[_abi.CNT_READS] == 400_000 and c[_abi.CNT_BASES] == dev.total_bases
    rate = c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES]
    assert abs(rate / 0.013404 - 1) < 0.02, rate
    assert abs(c[_abi.CNT_QUAL_SUM] / c[_abi.CNT_BASES] - 29.5) < 0.05
    # substitutions really are in the output: compare with the reference bases
    d = dev.to_host()
    ref = genome_1m.contigs[0]
    fwd = np.flatnonzero((d["flags"] & 1) == 0)[:20000]
    mism = tot = 0
    for r in fwd:
        s, e = int(d["start"][r]), int(d["end"][r])
        o = int(d["seq_off"][r])
        mism += int((d["seq"][o:o + e - s] != ref[s:e]).sum())
        tot += e - s
    assert abs(mism / tot / 0.013404 - 1) < 0.06


# long reads
@pytest.mark.parametrize("cls", [MinimalLongErrorProfile, PerfectLongErrorProfile])
def test_long_reference_mode_bit_exact(engine, oracle, genome_multi, genome_1m, cls):
    prof = cls().pod()
    reads = [37, 0, 25]
    engine.stage_genome(4, genome_1m.contigs)
    dev = engine.simulate_long_reads([1, 4, 0], reads, prof, 42, read_id_base=3)
    ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m, genome_1m], reads, prof, 42, read_id_base=3)
    d, o = dev.to_host(), ora.trimmed()
    # the oracle numbers genomes by position in the list; the engine by staged index
    o["genome"] = np.array([1, 4, 0], dtype=np.uint32)[o["genome"]]
    assert_same(d, o, cols=COLS + ("genome",))


def test_long_per_read_mode_bit_exact(engine, oracle, genome_multi):
    prof = MinimalLongErrorProfile(gamma_mean=8000.0, gamma_std=6000.0, length_mode=_abi.LEN_PER_READ).pod()
    dev = engine.simulate_long_reads([1], [300], prof, 77)
    ora = _oracle.simulate_long(oracle, [genome_multi], [300], prof, 77)
    d, o = dev.to_host(), ora.trimmed()
    o["genome"][:] = 1
    assert_same(d, o, cols=COLS + ("genome",))
    lens = np.diff(d["seq_off"].astype(np.int64))
    assert 5000 < lens.mean() < 11000


def test_long_sharding(engine, oracle, genome_multi, genome_1m):
    prof = MinimalLongErrorProfile().pod()
    reads = [40, 30]
    whole = _oracle.simulate_long(oracle, [genome_multi, genome_1m], reads, prof, 9).trimmed()
    for first, count in [(0, 10), (35, 10), (40, 30), (69, 5)]:
        dev = engine.simulate_long_reads([1, 0], reads, prof, 9, first=first, count=count)
        d = dev.to_host()
        n = min(count, 70 - first)
        base = whole["seq_off"][first]
        assert np.array_equal(d["seq_off"], whole["seq_off"][first:first + n + 1] - base)
        assert np.array_equal(d["seq"], whole["seq"][base:whole["seq_off"][first + n]])
        assert np.array_equal(d["qual"], whole["qual"][base:whole["seq_off"][first + n]])
        assert np.array_equal(d["read_id"], whole["read_id"][first:first + n])


# ---- SIMMR_RNG_PHILOX: counter mode (north_star's design for the per-base draws) ----
def test_philox_mode_matches_its_specification(engine, oracle, genome_multi, genome_1m):
    """Same bytes as the CPU restatement of the mode (oracle/philox.c): positions and
    lengths still come from the reference streams, per-base draws from Philox4x32-10."""
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    for gidx, g, reads, seed in ((1, genome_multi, 3001, 5), (0, genome_1m, 8000, 42)):
        dev = engine.simulate_pe_reads_from_genome(gidx, prof, reads, seed, qual_offset=33)
        ora = _oracle.simulate_pe(oracle, g, prof, reads, seed, qual_offset=33)
        assert_same(dev.to_host(), ora.trimmed())
    lp = MinimalLongErrorProfile(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ,
                                 rng_mode=_abi.RNG_PHILOX, mean_phred_score=20).pod()
    dev = engine.simulate_long_reads([1], [120], lp, 3)
    ora = _oracle.simulate_long(oracle, [genome_multi], [120], lp, 3)
    d, o = dev.to_host(), ora.trimmed()
    o["genome"][:] = 1
    assert_same(d, o, cols=COLS + ("genome",))


@pytest.mark.parametrize("L,I,q", [(20, 20, 30), (7, 3, 10), (150, 600, 45), (333, 100, 2), (16, 16, 60)])
def test_philox_mode_edge_shapes(engine, oracle, genome_multi, L, I, q):
    prof = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=q, rng_mode=_abi.RNG_PHILOX).pod()
    dev = engine.simulate_pe_reads_from_genome(1, prof, 2501, 13, first=100, count=900, read_id_base=7)
    ora = _oracle.simulate_pe(oracle, genome_multi, prof, 2501, 13, first=100, count=900, read_id_base=7, max_len=4096)
    assert_same(dev.to_host(), ora.trimmed())


def test_philox_mode_quality_offset_wraps(engine, oracle, genome_multi):
    """q + 33 is an unchecked u8 add in the reference (util.rs:46-50): with a mean Phred of 240 most
    encoded qualities wrap; the bytes and the QUAL_SUM counter (raw Phred) must still be right."""
    prof = MinimalShortErrorProfile(read_length=37, insert_size=50, mean_phred_score=240, rng_mode=_abi.RNG_PHILOX).pod()
    for qoff in (33, 0, 200):
        engine.counters_reset()
        dev = engine.simulate_pe_reads_from_genome(1, prof, 2000, 3, qual_offset=qoff)
        ora = _oracle.simulate_pe(oracle, genome_multi, prof, 2000, 3, qual_offset=qoff)
        d, o = dev.to_host(), ora.trimmed()
        assert_same(d, o)
        raw = (d["qual"].astype(np.int64) - qoff) % 256
        assert engine.counters()[_abi.CNT_QUAL_SUM] == raw.sum()


def test_philox_mode_exceptions(engine, oracle):
    rng = np.random.default_rng(21)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 30000)].copy()
    seq[rng.integers(0, 30000, 3000)] = ord("N")
    seq[rng.integers(0, 30000, 500)] = ord("-")
    engine.stage_genome(3, [seq])
    g = _oracle.HostGenome([seq])
    prof = MinimalShortErrorProfile(mean_phred_score=8, rng_mode=_abi.RNG_PHILOX).pod()
    dev = engine.simulate_pe_reads_from_genome(3, prof, 3000, 8)
    ora = _oracle.simulate_pe(oracle, g, prof, 3000, 8)
    assert_same(dev.to_host(), ora.trimmed())
    from simmr_amd import SimmrError
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    keep = CustomShortErrorProfile(_model.synthetic_short_model(n_positions=40, seed=5))
    cp = keep.pod()
    cp.rng_mode = _abi.RNG_PHILOX  # the paired-end path of a custom model has no base-by-base draws: no counter mode
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(3, cp, 10, 1)
    assert ei.value.code == _abi.EINVAL


def test_philox_mode_tolerances(engine, genome_1m):
    """The tolerances BASELINE.json / SURVEY §8d state for the statistical profiles:
    substitution rate within 2 % of the analytic 0.013404 (mean Phred 30), Phred mean
    29.5 +- 0.05 and sd ~10, substitution target uniform over the 3 alternatives."""
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    engine.counters_reset()
    dev = engine.simulate_pe_reads_from_genome(0, prof, 1_000_000, 2024)
    c = engine.counters()
    rate = c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES]
    assert abs(rate / 0.013404 - 1) < 0.02, rate
    assert abs(c[_abi.CNT_QUAL_SUM] / c[_abi.CNT_BASES] - 29.5) < 0.05
    d = dev.to_host()
    q = d["qual"].astype(np.float64)
    assert abs(q.std() - 10.0) < 0.1
    # exact Phred histogram vs the Normal(30, 10) floor distribution (z-test per bin)
    from math import erf, sqrt
    cdf = lambda x: 0.5 * (1 + erf((x - 30.0) / 10.0 / sqrt(2)))
    hist = np.bincount(d["qual"], minlength=256)
    n = hist.sum()
    for qv in range(5, 60, 5):
        p = cdf(qv + 1) - cdf(qv)
        assert abs(hist[qv] / n - p) < 5 * sqrt(p / n), qv
    # forward mates: compare with the reference bases -> which alternatives were chosen
    ref = genome_1m.contigs[0]
    counts = np.zeros((4, 4), dtype=np.int64)
    lut = np.full(256, 4, dtype=np.int64)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    fwd = np.flatnonzero((d["flags"] & 1) == 0)[:100000]
    for r in fwd:
        s, e2, o = int(d["start"][r]), int(d["end"][r]), int(d["seq_off"][r])
        a, b = lut[ref[s:e2]], lut[d["seq"][o:o + e2 - s]]
        m = a != b
        np.add.at(counts, (a[m], b[m]), 1)
    subs = counts.sum()
    assert subs > 100000
    for a in range(4):
        row = counts[a][[x for x in range(4) if x != a]]
        exp = row.sum() / 3
        chi2 = ((row - exp) ** 2 / exp).sum()
        assert chi2 < 13.8, (a, row)  # chi-square, 2 dof, p > 0.001


# ---- SIMMR_RNG_PHILOX_FULL: the plan's draws from Philox counters too (include/simmr_hip.h) ----
def test_philox_full_matches_its_specification(engine, oracle, genome_multi, genome_1m):
    """Bit for bit the CPU restatement of the mode (oracle/rand08.c: the generator's second word source; simulate.c:
    orc_pe_outer_ctr): contigs, seeds, lengths, windows, mate-2 seeds, qualities and bases — whole runs, shards of them,
    several genomes in one plan, long reads with per-read lengths."""
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
    for gidx, g, reads, seed in ((1, genome_multi, 3001, 5), (0, genome_1m, 8000, 42)):
        dev = engine.simulate_pe_reads_from_genome(gidx, prof, reads, seed, qual_offset=33)
        ora = _oracle.simulate_pe(oracle, g, prof, reads, seed, qual_offset=33)
        assert_same(dev.to_host(), ora.trimmed())
    # a shard is a function of its pair indices alone: no position in a stream to hand over
    whole = _oracle.simulate_pe(oracle, genome_multi, prof, 5000, 77, read_id_base=3).trimmed()
    for first, count in ((0, 10), (1234, 700), (2499, 1), (2400, 5000)):
        part = engine.simulate_pe_reads_from_genome(1, prof, 5000, 77, first=first, count=count, read_id_base=3).to_host()
        n = min(count, 2500 - first)
        a, b = int(whole["seq_off"][2 * first]), int(whole["seq_off"][2 * (first + n)])
        assert np.array_equal(part["seq"], whole["seq"][a:b]) and np.array_equal(part["qual"], whole["qual"][a:b])
        for col in ("start", "end", "contig", "read_id", "flags"):
            assert np.array_equal(part[col], whole[col][2 * first:2 * (first + n)]), col
    # edge shapes (the three window cases of simulate.rs:241-258, tiny reads, long inserts)
    for L, I, q in ((20, 20, 30), (7, 3, 10), (150, 600, 45), (333, 100, 2)):
        pe = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=q, rng_mode=_abi.RNG_PHILOX_FULL).pod()
        dev = engine.simulate_pe_reads_from_genome(1, pe, 2501, 13, first=100, count=900, read_id_base=7)
        ora = _oracle.simulate_pe(oracle, genome_multi, pe, 2501, 13, first=100, count=900, read_id_base=7, max_len=4096)
        assert_same(dev.to_host(), ora.trimmed())
    # several genomes in one plan (simmr_pe_plan_multi): the per-genome runs of the specification, concatenated
    engine.stage_genome(4, genome_1m.contigs)
    reads = [1200, 0, 801]
    d = engine.simulate_pe_reads_multi([1, 4, 0], reads, prof, 9, qual_offset=33).to_host()
    parts, base = [], 0
    for g, n in zip((genome_multi, genome_1m, genome_1m), reads):
        parts.append(_oracle.simulate_pe(oracle, g, prof, n, 9, read_id_base=base, qual_offset=33).trimmed())
        base += n // 2
    assert np.array_equal(d["seq"], np.concatenate([p["seq"] for p in parts]))
    assert np.array_equal(d["qual"], np.concatenate([p["qual"] for p in parts]))
    for col in ("start", "end", "contig", "read_id", "flags"):
        assert np.array_equal(d[col], np.concatenate([p[col] for p in parts])), col
    # long reads with per-read lengths, both Phred laws
    for cls, kw in ((MinimalLongErrorProfile, dict(mean_phred_score=20)), (PerfectLongErrorProfile, {})):
        lp = cls(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX_FULL, **kw).pod()
        dev = engine.simulate_long_reads([1, 0], [120, 75], lp, 3, first=11, count=150)
        ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m], [120, 75], lp, 3, first=11, count=150)
        dd, oo = dev.to_host(), ora.trimmed()
        oo["genome"] = np.array([1, 0], dtype=np.uint32)[oo["genome"]]
        assert_same(dd, oo, cols=COLS + ("genome",))


def test_philox_full_refusals(engine, genome_multi):
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from tests import _model
    for pod, plan in ((PerfectShortErrorProfile().pod(), "pe"), (CustomShortErrorProfile(_model.synthetic_short_model()).pod(), "pe"),
                      (CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=5, n_kmers=100)).pod(), "long"),
                      (MinimalLongErrorProfile().pod(), "long")):  # (the last: the reference's one constant length)
        pod.rng_mode = _abi.RNG_PHILOX_FULL
        with pytest.raises(SimmrError) as ei:
            engine.pe_plan(1, pod, 100, 1) if plan == "pe" else engine.long_plan([1], [10], pod, 1)
        assert ei.value.code == _abi.EINVAL and "SIMMR_RNG_PHILOX" in ei.value.msg, ei.value.msg


def test_philox_full_tolerances(engine, genome_1m, genome_multi):
    """The laws of the plan's draws (the per-base laws are SIMMR_RNG_PHILOX's, test_philox_mode_tolerances): read length
    floor(N(150, 15)), insert from the same z (minimal_short.rs:33-67), the first mate's start uniform on
    [0, size - required), contigs uniform whatever their size (simulate.rs:181), half of the mate-2 seeds drawn and half
    substituted (Option<u64>, simulate.rs:266), and the run's substitution rate and Phred mean."""
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
    engine.counters_reset()
    d = engine.simulate_pe_reads_from_genome(0, prof, 1_000_000, 2024).to_host()
    c = engine.counters()
    assert abs(c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES] / 0.013404 - 1) < 0.02
    assert abs(c[_abi.CNT_QUAL_SUM] / c[_abi.CNT_BASES] - 29.5) < 0.05
    lens = np.diff(d["seq_off"].astype(np.int64))[0::2]
    n = lens.size
    assert n == 500_000 and abs(lens.mean() - 149.5) < 0.1 and abs(lens.std() - 15.0) < 0.1  # floor() takes half a base
    # start of mate 1 ~ U[0, size - required): mean, variance and a coarse histogram
    size, required = 1_000_000, 450
    st = d["start"][0::2].astype(np.float64) / (size - required)
    assert abs(st.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n) and abs(st.var() - 1 / 12) < 0.001
    hist = np.bincount((st * 20).astype(int), minlength=20)
    assert (np.abs(hist - n / 20) < 5 * np.sqrt(n / 20)).all()
    # mate-2 seeds: Some / None with probability 1/2 each
    subst = ((d["flags"][1::2] & _abi.FLAG_QSEED_SUBST) != 0).mean()
    assert abs(subst - 0.5) < 5 * np.sqrt(0.25 / n)
    # contigs of a genome are drawn uniformly, not by size
    dm = engine.simulate_pe_reads_from_genome(1, prof, 200_000, 5).to_host()
    cnt = np.bincount(dm["contig"][0::2], minlength=len(genome_multi.contigs))
    assert (np.abs(cnt - 100_000 / cnt.size) < 5 * np.sqrt(100_000 / cnt.size)).all()
    # and the two counter modes are different runs of the same law
    d1 = engine.simulate_pe_reads_from_genome(0, MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod(), 1_000_000, 2024).to_host()
    l1 = np.diff(d1["seq_off"].astype(np.int64))[0::2]
    assert not np.array_equal(d1["start"], d["start"]) and abs(l1.mean() - lens.mean()) < 0.15


# custom-short (empirical PDFs): bincode model -> alias tables on both sides
@pytest.mark.parametrize("n_positions,seed", [(120, 42), (60, 7)])
def test_custom_short_bit_exact(engine, oracle, genome_multi, n_positions, seed):
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    prof = CustomShortErrorProfile(_model.synthetic_short_model(n_positions=n_positions, seed=seed))
    pod = prof.pod()
    engine.counters_reset()
    dev = engine.simulate_pe_reads_from_genome(1, pod, 4000, seed, qual_offset=33)
    ora = _oracle.simulate_pe(oracle, genome_multi, pod, 4000, seed, qual_offset=33)
    assert_same(dev.to_host(), ora.trimmed())
    d = dev.to_host()
    lens = np.diff(d["seq_off"].astype(np.int64))
    assert 80 <= lens.min() and lens.max() < 200 and abs(lens.mean() - 140) < 3
    c = engine.counters()
    assert c[_abi.CNT_SUBSTITUTIONS] == 0 and c[_abi.CNT_BASES] == dev.total_bases


def test_custom_short_rejection_heavy_model(engine, oracle, genome_multi):
    """Bin ranges of about 2^31 + 1 scores make Uniform<u32>::new_inclusive reject every second word, so some
    samples run past the 16 staged words of their stream and take the exact on-demand path."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    wide = [(0, 2 ** 31 + 5), (7, 2 ** 31 + 900), (40, 40)]
    quality = [([0.45, 0.45, 0.10], wide)] * 90
    blob = _model.serialize_model(quality, ([0.5, 0.5], [(70, 79), (80, 89)]), ([1.0], [(60, 160)]),
                                  read_length_mean=80.0, insert_size_mean=110.0)
    pod = CustomShortErrorProfile(blob).pod()
    engine.counters_reset()
    dev = engine.simulate_pe_reads_from_genome(1, pod, 6000, 3, qual_offset=33)
    ora = _oracle.simulate_pe(oracle, genome_multi, pod, 6000, 3, qual_offset=33)
    assert_same(dev.to_host(), ora.trimmed())
    raw = (dev.to_host()["qual"].astype(np.int64) - 33) % 256
    assert engine.counters()[_abi.CNT_QUAL_SUM] == raw.sum()


def test_custom_short_rejects_bad_models(engine, genome_multi):
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from tests import _model
    blob = _model.synthetic_short_model()
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(1, CustomShortErrorProfile(blob[:-5]).pod(), 100, 1)
    assert ei.value.code == _abi.EINVAL
    long_blob = _model.serialize_model([([1.0], [(30, 30)])], ([1.0], [(100, 100)]), is_long=True)
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(1, CustomShortErrorProfile(long_blob).pod(), 100, 1)
    assert ei.value.code == _abi.EINVAL and "long reads" in ei.value.msg
    # simmrd writes one density more than bin ranges (probability.rs:162-166): picking it panics
    # in the reference (index out of bounds); here it is an error, not a silent value
    q = [([0.0, 1.0], [(30, 30)])] * 50
    bad = _model.serialize_model(q, ([1.0], [(100, 100)]), read_length_mean=100.0, insert_size_mean=0.0)
    with pytest.raises(SimmrError) as ei:
        engine.simulate_pe_reads_from_genome(1, CustomShortErrorProfile(bad).pod(), 100, 1)
    assert ei.value.code == _abi.ERANGE


# custom model on the long-read path: empirical qualities + the k-mer splice of simulate_errors.
# rng: the splice's draws from the read's StdRng stream (the reference's bits) or from Philox counters (SIMMR_RNG_PHILOX:
# the same walk, the same tables, bit-exact against its own restatement orc_custom_simulate_errors_philox)
RNG_MODES = pytest.mark.parametrize("rng_mode", [_abi.RNG_REFERENCE, _abi.RNG_PHILOX], ids=["reference", "philox"])


@RNG_MODES
@pytest.mark.parametrize("k,seed,n_positions", [(7, 42, 300), (3, 7, 40), (10, 5, 5000), (1, 9, 10)])
def test_custom_long_bit_exact(engine, oracle, genome_multi, genome_1m, k, seed, n_positions, rng_mode):
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    blob = _model.synthetic_long_model(kmer_size=k, n_positions=n_positions, seed=seed,
                                       n_kmers=3000 if k >= 6 else 4 ** k)
    prof = CustomShortErrorProfile(blob, rng_mode)
    assert prof.is_long_read()
    pod = prof.pod()
    engine.stage_genome(4, genome_1m.contigs)
    engine.counters_reset()
    reads = [90, 0, 70]
    dev = engine.simulate_long_reads([1, 4, 0], reads, pod, seed, read_id_base=11, qual_offset=33)
    ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m, genome_1m], reads, pod, seed, read_id_base=11,
                                qual_offset=33)
    d, o = dev.to_host(), ora.trimmed()
    o["genome"] = np.array([1, 4, 0], dtype=np.uint32)[o["genome"]]
    assert_same(d, o, cols=COLS + ("genome",))
    # the splice really edits bases, and the counter says how many
    g = {1: genome_multi, 0: genome_1m}
    mism = 0
    for r in range(len(d["start"])):
        ref = g[int(d["genome"][r])].contigs[int(d["contig"][r])][int(d["start"][r]):int(d["end"][r])]
        mism += int((d["seq"][d["seq_off"][r]:d["seq_off"][r + 1]] != ref).sum())
    c = engine.counters()
    assert c[_abi.CNT_SUBSTITUTIONS] == mism and (mism > 0 or k == 10)
    assert c[_abi.CNT_QUAL_SUM] == ((d["qual"].astype(np.int64) - 33) % 256).sum()


def test_custom_long_emit_refuses_a_model_without_its_counter_mode_tables(genome_1m, monkeypatch):
    """The guard in front of the custom long-read kernels (engine.hip: custom_long_tables_missing): a
    profile whose device tables for the requested mode are not set must be refused with SIMMR_EINVAL before anything is
    launched — the first build of the splice's counter mode faulted the GPU on a nil table address (LAB.md, round 5).
    SIMMR_FAULT_INJECT=null_ctr_tables (read when the engine is made) drops the counter mode's three tables from the
    profile; the reference mode, whose tables are still there, keeps working on the same engine."""
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from simmr_amd.engine import Engine
    from tests import _model
    monkeypatch.setenv("SIMMR_FAULT_INJECT", "null_ctr_tables")
    eng = Engine(0)
    try:
        eng.stage_genome(0, genome_1m.contigs)
        for k in (7, 10):  # the fixed-stride form with its LDS table, and the two-load form
            blob = _model.synthetic_long_model(kmer_size=k, n_positions=300, seed=42, n_kmers=3000)
            with pytest.raises(SimmrError) as ei:
                eng.simulate_long_reads([0], [40], CustomShortErrorProfile(blob, _abi.RNG_PHILOX).pod(), 3)
            assert ei.value.code == _abi.EINVAL and "kmer_" in str(ei.value) and "nothing was launched" in str(ei.value)
            reads = eng.simulate_long_reads([0], [40], CustomShortErrorProfile(blob, _abi.RNG_REFERENCE).pod(), 3)
            assert reads.n_reads == 40
    finally:
        eng.close()


@RNG_MODES
@pytest.mark.parametrize("k,max_alts", [(7, 21), (6, 32), (7, 40), (5, 255)])
def test_custom_long_alternate_lists_of_simmrd_size(engine, oracle, genome_1m, k, max_alts, rng_mode):
    """simmrd keeps up to --max-alt-kmers alternates per k-mer (default 20, a u8): lists of up to 32 go through the
    fixed-stride column table of the fast splice kernel, longer ones through the two-load kernel."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    blob = _model.synthetic_long_model(kmer_size=k, n_positions=50, seed=17 + max_alts, n_kmers=4 ** k, max_alts=max_alts,
                                       lengths=(900, 2500, 100))
    pod = CustomShortErrorProfile(blob, rng_mode).pod()
    dev = engine.simulate_long_reads([0], [150], pod, 77, read_id_base=0, qual_offset=33)
    ora = _oracle.simulate_long(oracle, [genome_1m], [150], pod, 77, read_id_base=0, qual_offset=33)
    assert_same(dev.to_host(), ora.trimmed(), cols=COLS)


def test_custom_model_tables_follow_the_model(engine, genome_1m):
    """The engine keeps the tables of the last custom model (parsed, built and uploaded once per model): another model
    must replace them, the first one must come back exactly, and the same bytes on the other path must still be refused."""
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from tests import _model
    pa = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=6, n_positions=40, seed=101, n_kmers=4 ** 6, lengths=(600, 1500, 100)))
    pb = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=6, n_positions=40, seed=102, n_kmers=4 ** 6, lengths=(600, 1500, 100)))
    a, b = pa.pod(), pb.pod()  # (the PODs point into the profiles' model bytes)
    run = lambda pod: engine.simulate_long_reads([0], [64], pod, 5, read_id_base=0, qual_offset=33).to_host()
    ra, rb, ra2, rb2 = run(a), run(b), run(a), run(b)
    for col in ("seq", "qual", "seq_off", "start", "end"):
        assert np.array_equal(ra[col], ra2[col]) and np.array_equal(rb[col], rb2[col]), col
    assert not (ra["seq"].shape == rb["seq"].shape and np.array_equal(ra["qual"], rb["qual"]))
    with pytest.raises(SimmrError) as ei:  # a long-read model on the paired-end path (main.rs:30-33), cached or not
        engine.simulate_pe_reads_from_genome(0, a, 100, 1)
    assert ei.value.code == _abi.EINVAL
    assert np.array_equal(run(a)["seq"], ra["seq"])


@RNG_MODES
def test_custom_long_exceptions_and_sharding(engine, oracle, rng_mode):
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    rng = np.random.default_rng(21)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60000)].copy()
    seq[rng.integers(0, 60000, 1500)] = ord("N")
    seq[rng.integers(0, 60000, 300)] = ord("-")
    seq[30000:30050] = ord("N")
    short = seq[:2500].copy()  # shorter than most drawn lengths: reads are re-cut at the sequence end
    engine.stage_genome(3, [seq, short])
    g = _oracle.HostGenome([seq, short])
    pod = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=4, n_positions=100, seed=2, n_kmers=256,
                                                             lengths=(300, 2400, 50)), rng_mode).pod()
    whole = _oracle.simulate_long(oracle, [g], [400], pod, 6).trimmed()
    dev = engine.simulate_long_reads([3], [400], pod, 6)
    d = dev.to_host()
    whole["genome"][:] = 3
    assert_same(d, whole, cols=COLS + ("genome",))
    for first, count in [(0, 17), (123, 200), (399, 5)]:
        part = engine.simulate_long_reads([3], [400], pod, 6, first=first, count=count).to_host()
        n = min(count, 400 - first)
        base = whole["seq_off"][first]
        assert np.array_equal(part["seq_off"], whole["seq_off"][first:first + n + 1] - base)
        assert np.array_equal(part["seq"], whole["seq"][base:whole["seq_off"][first + n]])
        assert np.array_equal(part["qual"], whole["qual"][base:whole["seq_off"][first + n]])


@RNG_MODES
@pytest.mark.parametrize("uniform_start", [False, True])
def test_custom_long_per_read_lengths(engine, oracle, genome_multi, genome_1m, uniform_start, rng_mode):
    """SIMMR_LEN_PER_READ with a custom model: every read draws floor(Normal(read_length_mean, read_length_std))
    from its own StdRng, as the reference does without --seed (custom_short.rs:286-301, simulate.rs:358)."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    prof = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=6, n_positions=2500, seed=12, n_kmers=4 ** 6,
                                                               lengths=(800, 5200, 100)), rng_mode)
    pod = prof.pod()
    pod.length_mode = _abi.LEN_PER_READ
    pod.long_start_mode = _abi.START_UNIFORM if uniform_start else _abi.START_REFERENCE
    engine.stage_genome(4, genome_1m.contigs)
    reads = [150, 100]
    dev = engine.simulate_long_reads([1, 4], reads, pod, 23, qual_offset=33)
    ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m], reads, pod, 23, qual_offset=33)
    d, o = dev.to_host(), ora.trimmed()
    o["genome"] = np.array([1, 4], dtype=np.uint32)[o["genome"]]
    assert_same(d, o, cols=COLS + ("genome",))
    lens = np.diff(d["seq_off"].astype(np.int64))
    assert lens.std() > 300 and abs(lens.mean() - 3000) < 250  # N(3000, 880)
    if uniform_start:  # starts spread over the sequences instead of their first read_length bases (quirk Q6)
        assert (d["start"].astype(np.int64) > 10_000).mean() > 0.5
    else:
        assert (d["start"].astype(np.int64) < lens).all()
    # a shard of the same run
    part = engine.simulate_long_reads([1, 4], reads, pod, 23, first=140, count=30, qual_offset=33).to_host()
    base = int(d["seq_off"][140])
    assert np.array_equal(part["seq"], d["seq"][base:int(d["seq_off"][170])])
    assert np.array_equal(part["qual"], d["qual"][base:int(d["seq_off"][170])])


@RNG_MODES
def test_custom_long_reads_shorter_than_a_kmer(engine, oracle, rng_mode):
    """Reads re-cut at the end of a tiny sequence can be shorter than k (even empty): simulate_errors then visits
    no k-mer (custom_short.rs:475-477) and the read is a plain copy."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 47)].copy()
    seq[11] = ord("N")
    engine.stage_genome(3, [seq])
    g = _oracle.HostGenome([seq])
    blob = _model.synthetic_long_model(kmer_size=7, n_positions=12, seed=3, n_kmers=4 ** 7, read_length_mean=40.0,
                                       read_length_std=2.0)
    pod = CustomShortErrorProfile(blob, rng_mode).pod()
    for seed in (1, 2, 3):
        dev = engine.simulate_long_reads([3], [300], pod, seed, qual_offset=33).to_host()
        ora = _oracle.simulate_long(oracle, [g], [300], pod, seed, qual_offset=33).trimmed()
        ora["genome"][:] = 3
        assert_same(dev, ora, cols=COLS + ("genome",))
        lens = np.diff(dev["seq_off"].astype(np.int64))
        assert (lens < 7).any() and (lens >= 7).any()


@RNG_MODES
def test_custom_long_error_paths(engine, oracle, genome_multi, rng_mode):
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from tests import _model
    # a deletion: the reference panics on its next slice; both sides answer ERANGE
    blob = _model.synthetic_long_model(kmer_size=5, n_positions=20, seed=1, n_kmers=100, deletion=True)
    pod = CustomShortErrorProfile(blob, rng_mode).pod()
    with pytest.raises(SimmrError) as ei:
        engine.simulate_long_reads([1], [50], pod, 3)
    assert ei.value.code == _abi.ERANGE and "simulate_errors" in ei.value.msg
    with pytest.raises(RuntimeError) as oi:
        _oracle.simulate_long(oracle, [genome_multi], [50], pod, 3)
    assert f"oracle error {_abi.ERANGE}" in str(oi.value)
    good = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=5, n_positions=20, seed=1, n_kmers=100))
    # a short-read model has is_long_read() == false and never reaches simulate_long_reads
    with pytest.raises(SimmrError) as ei:
        engine.long_plan([1], [10], CustomShortErrorProfile(_model.synthetic_short_model()).pod(), 3)
    assert ei.value.code == _abi.EINVAL
    # a 3-bit code of 11 bases does not fit the model's u32 keys
    with pytest.raises(SimmrError) as ei:
        engine.long_plan([1], [10], CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=11, n_kmers=10)).pod(), 3)
    assert ei.value.code == _abi.ENOTSUP
    # weights WeightedAliasIndex::new refuses: the reference unwraps the Err when the k-mer is visited
    probs = [(c, [(c, 0.0), (c ^ 1, 0.0)]) for c in range(4)]
    zero = _model.serialize_model([([1.0], [(30, 30)])] * 3, ([1.0], [(700, 700)]), probabilities=probs, kmer_size=1,
                                  read_length_mean=700.0, insert_size_mean=0.0, is_long=True)
    with pytest.raises(SimmrError) as ei:
        engine.simulate_long_reads([1], [5], CustomShortErrorProfile(zero, rng_mode).pod(), 3)
    assert ei.value.code == _abi.ERANGE
    # the counter mode of a custom model is the splice's: the paired-end path has no base-by-base draws to offer it
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(1, CustomShortErrorProfile(_model.synthetic_short_model(), _abi.RNG_PHILOX).pod(), 10, 3)
    assert ei.value.code == _abi.EINVAL and "SIMMR_RNG_PHILOX" in ei.value.msg


def test_custom_long_counter_mode_tolerances(engine, genome_1m):
    """SIMMR_RNG_PHILOX with a custom long-read model: lengths, positions and qualities are the reference mode's bit for
    bit (the same seeds, the same streams); the splice draws every visited k-mer's alternate from the same alias tables
    with other bits.  Law: the edited fraction of the bases agrees with the reference mode's within 4 % (relative; ~6e4
    edits per run), and — on a model whose k-mers are single bases, where a visited base is replaced independently of
    its neighbours — every alternate's frequency agrees with its weight (chi-square per k-mer, p > 1e-4)."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    blob = _model.synthetic_long_model(kmer_size=6, n_positions=200, seed=31, n_kmers=4 ** 6, lengths=(4000, 9000, 100))
    runs = {}
    for rng_mode in (_abi.RNG_REFERENCE, _abi.RNG_PHILOX):
        engine.counters_reset()
        d = engine.simulate_long_reads([0], [1500], CustomShortErrorProfile(blob, rng_mode).pod(), 5, qual_offset=33).to_host()
        c = engine.counters()
        runs[rng_mode] = (d, c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES], c[_abi.CNT_ACGT_BASES])
    (a, ra, na), (b, rb, nb) = runs[_abi.RNG_REFERENCE], runs[_abi.RNG_PHILOX]
    for col in ("seq_off", "start", "end", "contig", "read_id", "qual"):
        assert np.array_equal(a[col], b[col]), col
    assert na == nb > 5_000_000 and not np.array_equal(a["seq"], b["seq"])
    assert ra > 0.005 and abs(rb / ra - 1) < 0.04, (ra, rb)
    # single-base k-mers: P(base c becomes alternate j) = w[c][j] / sum(w[c])
    w = np.array([[8.0, 1.0, 0.5, 0.5], [0.25, 6.0, 0.25, 1.5], [1.0, 1.0, 5.0, 1.0], [0.1, 0.2, 0.3, 9.4]], dtype=np.float32)
    probs = [(c, [(j, float(w[c][j])) for j in range(4)]) for c in range(4)]
    one = _model.serialize_model([([1.0], [(30, 30)])] * 3, ([1.0], [(6000, 6000)]), probabilities=probs, kmer_size=1,
                                 read_length_mean=6000.0, insert_size_mean=0.0, is_long=True)
    d = engine.simulate_long_reads([0], [400], CustomShortErrorProfile(one, _abi.RNG_PHILOX).pod(), 9).to_host()
    ref = np.concatenate([genome_1m.contigs[int(d["contig"][r])][int(d["start"][r]):int(d["end"][r])] for r in range(400)])
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    lut = np.full(256, 255, np.uint8)
    for ch, v in code.items():
        lut[ch] = v
    src, dst = lut[ref], lut[d["seq"]]
    assert src.size == dst.size > 2_000_000 and src.max() < 4 and dst.max() < 4
    from scipy.stats import chisquare
    for c0 in range(4):
        obs = np.bincount(dst[src == c0], minlength=4).astype(float)
        pr = w[c0].astype(np.float64)
        exp = pr / pr.sum() * obs.sum()
        assert chisquare(obs, exp).pvalue > 1e-4, (c0, obs, exp)


def test_error_paths(engine, genome_multi):
    from simmr_amd import SimmrError
    with pytest.raises(SimmrError) as ei:  # contig 2 (30 017) <= 2*20000+... required
        engine.pe_plan(1, PerfectShortErrorProfile(20000, 20000).pod(), 10, 1)
    assert ei.value.code == _abi.EGENOME
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(1, PerfectShortErrorProfile(30000, 30000).pod(), 10, 1)
    assert ei.value.code == _abi.ERANGE  # u16 overflow of minimum_genome_size
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(63, PerfectShortErrorProfile().pod(), 10, 1)  # a slot no test stages
    assert ei.value.code == _abi.EINVAL
    with pytest.raises(SimmrError) as ei:
        engine.pe_plan(1, MinimalLongErrorProfile().pod(), 10, 1)
    assert ei.value.code == _abi.EINVAL
    # empty and odd inputs (simulate.rs:179: num_reads / 2 pairs)
    assert engine.pe_plan(1, PerfectShortErrorProfile().pod(), 1, 1).n_reads == 0
    assert engine.pe_plan(1, PerfectShortErrorProfile().pod(), 7, 1).n_reads == 6
    out = engine.simulate_pe_reads_from_genome(1, PerfectShortErrorProfile().pod(), 0, 1)
    assert out.n_reads == 0 and out.total_bases == 0


# ---- multi-GPU seek in the outer stream (simmr_outer_summarize / simmr_pe_plan_at) ----
def test_outer_stream_seek(engine, oracle, genome_multi, genome_1m):
    from simmr_amd.simulate import compose_outer_summaries, outer_slot_floor
    from tests.test_multi_rank_cpu import outer_accept_bits, replay_outer
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    for gidx, g, seed, per_rank, world in ((0, genome_1m, 42, 20_000, 4), (1, genome_multi, 9, 15_000, 3)):
        nc = len(g.contigs)
        total = per_rank * world
        acc = outer_accept_bits(oracle, nc, seed, outer_slot_floor(nc, total) + 64)
        pieces = [(gidx, nc, j * per_rank, (j + 1) * per_rank) if j + 1 < world else None for j in range(world)]
        summaries = []
        for p in pieces:
            if p is None:
                summaries.append((0, 0, 0, 1))
                continue
            lo, hi = outer_slot_floor(nc, p[2]), outer_slot_floor(nc, p[3])
            got = engine.outer_summarize(gidx, seed, lo, hi - lo)
            (u0, e0), (u1, e1) = replay_outer(acc, lo, hi, 0), replay_outer(acc, lo, hi, 1)
            assert got == (u0, u1, e0, e1)
            summaries.append(got)
        whole = engine.simulate_pe_reads_from_genome(gidx, prof, 2 * total, seed, qual_offset=33).to_host()
        for r in range(1, world):
            first = r * per_rank
            start = compose_outer_summaries(pieces, summaries, (gidx, nc, first))
            assert start != (0, 0) and start[1] <= first
            part = engine.simulate_pe_reads_from_genome(gidx, prof, 2 * total, seed, first=first, count=per_rank,
                                                        read_id_base=5, qual_offset=33, start=start).to_host()
            a, b = 2 * first, 2 * (first + per_rank)
            base = whole["seq_off"][a]
            assert np.array_equal(part["seq_off"], whole["seq_off"][a:b + 1] - base)
            for col in ("start", "end", "contig", "flags"):
                assert np.array_equal(part[col], whole[col][a:b]), col
            assert np.array_equal(part["read_id"], whole["read_id"][a:b] + 5)
            assert np.array_equal(part["seq"], whole["seq"][base:whole["seq_off"][b]])
            assert np.array_equal(part["qual"], whole["qual"][base:whole["seq_off"][b]])
    from simmr_amd import SimmrError
    with pytest.raises(SimmrError):  # a position past the shard's first pair
        engine.pe_plan(0, prof, 1000, 42, 10, 20, start=(64, 11))
    with pytest.raises(SimmrError):  # slot ranges are whole ChaCha blocks
        engine.outer_summarize(0, 42, 4, 16)


# ---- simulate_pe_reads over several genomes in one plan (simmr_pe_plan_multi) ----
@pytest.mark.parametrize("kind", ["perfect", "minimal", "philox", "perfect17"])
def test_pe_plan_multi_equals_per_genome_calls(engine, oracle, genome_multi, genome_1m, kind):
    """Same reads, ids and order as the reference's loop over genomes (simulate.rs:121-150), checked against
    the oracle genome by genome; genomes 0 and 4 have one sequence and share one outer list, a genome
    without reads and one with an exception plane are in the middle."""
    rng = np.random.default_rng(5)
    exc = _synth.synthetic_contigs([50_000], 33)[0].copy()
    exc[rng.integers(0, 50_000, 4000)] = ord("N")
    engine.stage_genome(4, [exc])
    hosts = {0: genome_1m, 1: genome_multi, 4: _oracle.HostGenome([exc])}
    order, reads = [0, 1, 4, 1], [3000, 2501, 0, 1801]
    if kind != "perfect17":
        order, reads = order + [4], reads + [2200]
    prof = {"perfect": PerfectShortErrorProfile(), "minimal": MinimalShortErrorProfile(),
            "philox": MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX, mean_phred_score=25),
            "perfect17": PerfectShortErrorProfile(read_length=17, insert_size=40)}[kind].pod()
    seed = 77
    parts, base = [], 0
    for gi, n in zip(order, reads):
        o = _oracle.simulate_pe(oracle, hosts[gi], prof, n, seed, read_id_base=base, qual_offset=33).trimmed()
        o["genome"] = np.full(o["read_id"].size, gi, np.uint32)
        parts.append(o)
        base += n // 2
    lens = np.concatenate([np.diff(p["seq_off"].astype(np.int64)) for p in parts])
    whole = {c: np.concatenate([p[c] for p in parts]) for c in ("seq", "qual", "start", "end", "contig", "genome", "read_id", "flags")}
    whole["seq_off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n_pairs = base
    engine.counters_reset()
    dev = engine.simulate_pe_reads_multi(order, reads, prof, seed, qual_offset=33)
    assert_same(dev.to_host(), whole, cols=COLS + ("genome",))
    c = engine.counters()
    assert c[_abi.CNT_READS] == 2 * n_pairs and c[_abi.CNT_BASES] == lens.sum()
    for first, count in ((0, 10), (1400, 300), (2700, 1500), (n_pairs - 7, 50)):  # ranges across genome borders
        d = engine.simulate_pe_reads_multi(order, reads, prof, seed, first=first, count=count, qual_offset=33).to_host()
        a, b = 2 * first, 2 * min(first + count, n_pairs)
        o0 = whole["seq_off"][a]
        assert np.array_equal(d["seq_off"], whole["seq_off"][a:b + 1] - o0)
        for col in ("start", "end", "contig", "genome", "read_id", "flags"):
            assert np.array_equal(d[col], whole[col][a:b]), (col, first)
        assert np.array_equal(d["seq"], whole["seq"][o0:whole["seq_off"][b]])
        assert np.array_equal(d["qual"], whole["qual"][o0:whole["seq_off"][b]])
    engine.stage_genome(4, [exc[:400]])  # too small for 2 * 150 + 150
    if kind != "perfect17":
        from simmr_amd import SimmrError
        with pytest.raises(SimmrError) as ei:
            engine.pe_plan_multi(order, reads, prof, seed)
        assert ei.value.code == _abi.EGENOME
        engine.pe_plan_multi(order, reads, prof, seed, first=0, count=100)  # that genome is outside the shard


def test_pe_plan_multi_edges(engine, genome_multi, genome_1m):
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    # nothing to do: no reads at all, an empty shard, a shard past the end
    for reads, first, count in (([0, 0], 0, _abi.U64_MAX), ([1, 1], 0, _abi.U64_MAX), ([100, 60], 20, 0), ([100, 60], 500, 10)):
        info = engine.pe_plan_multi([0, 1], reads, prof, 5, first, count)
        assert info.n_units == 0 and info.n_reads == 0 and info.total_bases == 0
        out = engine.simulate_pe_reads_multi([0, 1], reads, prof, 5, first=first, count=count, qual_offset=33)
        assert out.n_reads == 0 and int(out.seq_off[0].item()) == 0
    # one genome through the multi plan == the single-genome plan (ids start at 0 in both)
    a = engine.simulate_pe_reads_multi([1], [3001], prof, 8, qual_offset=33).to_host()
    b = engine.simulate_pe_reads_from_genome(1, prof, 3001, 8, qual_offset=33).to_host()
    for col in COLS:
        assert np.array_equal(a[col], b[col]), col
    assert (a["genome"] == 1).all()
    from simmr_amd import CustomShortErrorProfile, SimmrError
    from tests import _model
    with pytest.raises(SimmrError) as ei:  # custom profiles are planned genome by genome
        engine.pe_plan_multi([1], [100], CustomShortErrorProfile(_model.synthetic_short_model()).pod(), 1)
    assert ei.value.code == _abi.ENOTSUP


def test_long_reads_uniform_start(engine, oracle, genome_multi, genome_1m):
    """SIMMR_START_UNIFORM: the start is drawn over the whole sequence instead of the
    reference's [0, read_length); same streams otherwise, bit-exact against the oracle in both length modes."""
    for kw in ({}, {"length_mode": _abi.LEN_PER_READ, "gamma_mean": 9000.0, "gamma_std": 5000.0}):
        for rng_mode in (_abi.RNG_REFERENCE, _abi.RNG_PHILOX):
            prof = MinimalLongErrorProfile(uniform_start=True, rng_mode=rng_mode, **kw).pod()
            dev = engine.simulate_long_reads([1, 0], [60, 40], prof, 11).to_host()
            ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m], [60, 40], prof, 11).trimmed()
            assert_same(dev, ora)
            lens = (dev["end"] - dev["start"]).astype(np.int64)
            assert (dev["start"] > lens).any()  # impossible with the reference's start < read_length
            sizes = np.array([c.size for c in genome_multi.contigs])
            g1 = dev["genome"] == 1
            assert (dev["end"][g1] <= sizes[dev["contig"][g1]]).all()
    # the reference's quirk for comparison: with the constant 20 000-base length every start is below it
    ref = engine.simulate_long_reads([1, 0], [60, 40], MinimalLongErrorProfile().pod(), 11).to_host()
    assert (ref["start"] < 65536).all()


def test_philox_mode_perfect_long(engine, oracle, genome_multi, genome_1m):
    """Counter mode for perfect-long (perfect_long.rs:60-119): same joint-alias machinery with that profile's Phred law;
    bit-exact against its restatement, and the Phred histogram agrees with the bit-exact mode's."""
    prof = PerfectLongErrorProfile(rng_mode=_abi.RNG_PHILOX, length_mode=_abi.LEN_PER_READ, gamma_mean=3000.0, gamma_std=2000.0).pod()
    dev = engine.simulate_long_reads([1, 0], [150, 100], prof, 21).to_host()
    ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m], [150, 100], prof, 21).trimmed()
    assert_same(dev, ora)
    prof.rng_mode = _abi.RNG_REFERENCE
    ref = engine.simulate_long_reads([1, 0], [150, 100], prof, 21).to_host()
    assert np.array_equal(ref["seq_off"], dev["seq_off"])  # lengths and positions do not depend on the mode
    n = dev["qual"].size
    assert n > 500_000
    ha, hb = np.bincount(dev["qual"], minlength=64) / n, np.bincount(ref["qual"], minlength=64) / n
    assert ha[41:].sum() == 0 and hb[41:].sum() == 0 and abs(ha[40] - hb[40]) < 0.005
    assert np.abs(ha - hb).max() < 0.005


def test_counter_allreduce_through_the_abi(engine, genome_multi):
    """simmr_comm_* / simmr_allreduce_counts (RCCL bound at run time): without a communicator the sum is a no-op;
    a one-rank communicator leaves the counters as they are (more ranks need more GPUs than this box has)."""
    import torch
    prof = PerfectShortErrorProfile(50, 70).pod()
    engine.counters_reset()
    engine.simulate_pe_reads_from_genome(1, prof, 2000, 3)
    t = torch.zeros(_abi.N_COUNTERS, dtype=torch.int64, device=engine.device)
    engine.counters_to(t)
    torch.cuda.synchronize()
    before = t.clone()
    engine.allreduce_counts(t)  # no communicator yet
    torch.cuda.synchronize()
    assert torch.equal(t, before) and int(t[_abi.CNT_READS]) == 2000
    cid = engine.comm_unique_id()
    assert len(cid) == _abi.COMM_ID_BYTES
    engine.comm_init(cid, 0, 1)
    engine.allreduce_counts(t)
    torch.cuda.synchronize()
    assert torch.equal(t, before)
