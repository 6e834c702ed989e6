"""16-byte read slots (SIMMR_SLOT16, include/simmr_hip.h: simmr_engine_set_read_slots / simmr_reads_out).

The opt-in layout of the counter-mode emit kernel: every read owns ceil(L / 16) * 16 bytes on a 16-byte boundary of
seq[] and qual[], qualities and forward bases left-aligned, a reverse-complemented mate's bases right-aligned, padding
0.  The READS must be the compact layout's reads byte for byte — i.e. equal to the CPU restatement (oracle/) — and the
raw columns must follow the rules the header states; FASTQ framing from either layout gives one text."""
import numpy as np
import pytest

from simmr_amd import (MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectLongErrorProfile,
                       PerfectShortErrorProfile, SimmrError, _abi)
from tests import _oracle, _synth
from tests.test_gpu_cli import FMT
from tests.test_gpu_parity import COLS, assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome_1m(engine):
    contigs = _synth.synthetic_contigs([1_000_000], 1)
    engine.stage_synthetic(0, [1_000_000], 1)
    return _oracle.HostGenome(contigs)


@pytest.fixture(scope="module")
def genome_multi(engine):
    contigs = _synth.synthetic_contigs([300_000, 90_001, 30_017, 70_000, 123_457], 7)
    engine.stage_genome(1, contigs)
    return _oracle.HostGenome(contigs)


@pytest.fixture()
def slots(engine):
    engine.set_read_slots(16)
    try:
        yield engine
    finally:
        engine.set_read_slots(0)


def check_raw_layout(reads):
    """the rules of include/simmr_hip.h for the columns as they lie in HBM"""
    assert reads.slot_bytes == 16
    r = reads.raw_to_host()
    n = reads.n_reads
    L = np.abs(r["end"].astype(np.int64) - r["start"].astype(np.int64))
    Lp = (L + 15) // 16 * 16
    slot = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(Lp, out=slot[1:])
    assert int(slot[n]) == reads.total_bases == int(r["seq_off"][n])
    first = r["seq_off"][:n].astype(np.int64)
    rev = (r["flags"] & _abi.FLAG_REVCOMP) != 0
    assert np.array_equal(first & ~np.int64(15), slot[:n])              # qualities start on the slot
    assert np.array_equal(first - slot[:n], np.where(rev, Lp - L, 0))   # bases: right-aligned iff reverse-complemented
    used_s = np.zeros(reads.total_bases + 1, dtype=np.int64)
    used_q = np.zeros(reads.total_bases + 1, dtype=np.int64)
    np.add.at(used_s, first, 1); np.add.at(used_s, first + L, -1)
    np.add.at(used_q, slot[:n], 1); np.add.at(used_q, slot[:n] + L, -1)
    pad_s, pad_q = np.cumsum(used_s)[:-1] == 0, np.cumsum(used_q)[:-1] == 0
    assert not r["seq"][pad_s].any() and not r["qual"][pad_q].any()     # padding is 0
    assert int(pad_s.sum()) == int(pad_q.sum()) == int((Lp - L).sum())
    return r


def test_slot16_pairs_equal_the_specification(slots, oracle, genome_multi, genome_1m):
    eng = slots
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    for gidx, g, reads, seed in ((1, genome_multi, 3001, 5), (0, genome_1m, 8000, 42)):
        info = eng.pe_plan(gidx, prof, reads, seed)
        assert info.slot_bytes == 16 and info.total_bases % 16 == 0
        dev = eng.simulate_pe_reads_from_genome(gidx, prof, reads, seed, qual_offset=33)
        check_raw_layout(dev)
        ora = _oracle.simulate_pe(oracle, g, prof, reads, seed, qual_offset=33)
        assert_same(dev.to_host(), ora.trimmed())


@pytest.mark.parametrize("L,I,q", [(20, 20, 30), (7, 3, 10), (150, 600, 45), (333, 100, 2), (16, 16, 60), (32, 40, 30), (15, 15, 20)])
def test_slot16_edge_shapes(slots, oracle, genome_multi, L, I, q):
    prof = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=q, rng_mode=_abi.RNG_PHILOX).pod()
    dev = slots.simulate_pe_reads_from_genome(1, prof, 2501, 13, first=100, count=900, read_id_base=7)
    check_raw_layout(dev)
    ora = _oracle.simulate_pe(oracle, genome_multi, prof, 2501, 13, first=100, count=900, read_id_base=7, max_len=4096)
    assert_same(dev.to_host(), ora.trimmed())


def test_slot16_long_reads_and_counters(slots, oracle, genome_multi, genome_1m):
    eng = slots
    for cls in (MinimalLongErrorProfile, PerfectLongErrorProfile):
        lp = cls(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX).pod()
        eng.counters_reset()
        dev = eng.simulate_long_reads([1, 0], [150, 100], lp, 3, qual_offset=33)
        c_slot = eng.counters()
        check_raw_layout(dev)
        ora = _oracle.simulate_long(oracle, [genome_multi, genome_1m], [150, 100], lp, 3, qual_offset=33)
        assert_same(dev.to_host(), ora.trimmed())
        eng.set_read_slots(0)
        eng.counters_reset()
        compact = eng.simulate_long_reads([1, 0], [150, 100], lp, 3, qual_offset=33)
        assert compact.slot_bytes == 0
        assert np.array_equal(eng.counters(), c_slot)  # the run counters do not depend on the layout
        assert_same(dev.to_host(), compact.to_host(), cols=COLS + ("genome",))
        eng.set_read_slots(16)


def test_slot16_exception_bases_and_multi_genome_plan(slots, oracle, genome_multi, genome_1m):
    eng = slots
    rng = np.random.default_rng(21)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 30000)].copy()
    seq[rng.integers(0, 30000, 3000)] = ord("N")
    seq[rng.integers(0, 30000, 500)] = ord("-")
    eng.stage_genome(3, [seq])
    g = _oracle.HostGenome([seq])
    prof = MinimalShortErrorProfile(mean_phred_score=8, rng_mode=_abi.RNG_PHILOX).pod()
    dev = eng.simulate_pe_reads_from_genome(3, prof, 3000, 8)
    check_raw_layout(dev)
    assert_same(dev.to_host(), _oracle.simulate_pe(oracle, g, prof, 3000, 8).trimmed())
    # several genomes in one plan (k_multi_units + the kernel form that looks the genome up per read)
    eng.counters_reset()
    multi = eng.simulate_pe_reads_multi([1, 3, 0], [2000, 1001, 3000], prof, 17)
    check_raw_layout(multi)
    c_slot = eng.counters()
    eng.set_read_slots(0)
    eng.counters_reset()
    compact = eng.simulate_pe_reads_multi([1, 3, 0], [2000, 1001, 3000], prof, 17)
    eng.set_read_slots(16)
    assert np.array_equal(eng.counters(), c_slot)
    assert_same(multi.to_host(), compact.to_host(), cols=COLS + ("genome",))


def test_slot16_quality_offset_wraps(slots, oracle, genome_multi):
    prof = MinimalShortErrorProfile(read_length=37, insert_size=50, mean_phred_score=240, rng_mode=_abi.RNG_PHILOX).pod()
    for qoff in (33, 200):
        slots.counters_reset()
        dev = slots.simulate_pe_reads_from_genome(1, prof, 2000, 3, qual_offset=qoff)
        d = dev.to_host()
        assert_same(d, _oracle.simulate_pe(oracle, genome_multi, prof, 2000, 3, qual_offset=qoff).trimmed())
        raw = (d["qual"].astype(np.int64) - qoff) % 256
        assert slots.counters()[_abi.CNT_QUAL_SUM] == raw.sum()


def test_slot16_stays_inside_exact_capacity(slots, genome_multi):
    """seq / qual of exactly total_bases bytes between canaries; shards whose last read is a partial group"""
    import torch
    from simmr_amd.engine import Reads
    eng = slots
    for pod, long_mode in ((MinimalShortErrorProfile(read_length=41, insert_size=60, rng_mode=_abi.RNG_PHILOX).pod(), False),
                           (MinimalLongErrorProfile(gamma_mean=900.0, gamma_std=700.0, length_mode=_abi.LEN_PER_READ,
                                                    rng_mode=_abi.RNG_PHILOX).pod(), True)):
        if long_mode:
            padded = eng.simulate_long_reads([1], [333], pod, 7, qual_offset=33).raw_to_host()
            info = eng.long_plan([1], [333], pod, 7)
        else:
            padded = eng.simulate_pe_reads_from_genome(1, pod, 2601, 7, qual_offset=33).raw_to_host()
            info = eng.pe_plan(1, pod, 2601, 7)
        PAD, tb = 256, int(info.total_bases)
        r = Reads.allocate(info.n_reads, tb, eng.device, 33, slot_bytes=16)
        bufs = []
        for name in ("seq", "qual"):
            buf = torch.full((tb + 2 * PAD,), 0xA5, dtype=torch.uint8, device=eng.device)
            setattr(r, name, buf[PAD:PAD + tb])
            bufs.append(buf)
        assert r.pod().seq_capacity == tb
        (eng.long_emit if long_mode else eng.pe_emit)(0, r)
        torch.cuda.synchronize()
        for b in bufs:
            assert bool((b[:PAD] == 0xA5).all()) and bool((b[PAD + tb:] == 0xA5).all())
        exact = r.raw_to_host()
        for col in ("seq", "qual", "seq_off", "start", "end", "contig", "read_id", "flags"):
            assert np.array_equal(exact[col], padded[col]), col


def test_slot16_refusals(engine, genome_multi):
    from simmr_amd import CustomShortErrorProfile, model_io
    from simmr_amd.engine import Reads
    with pytest.raises(SimmrError) as ei:
        engine.set_read_slots(8)
    assert ei.value.code == _abi.EINVAL
    ph = MinimalShortErrorProfile(read_length=41, insert_size=60, rng_mode=_abi.RNG_PHILOX).pod()
    engine.set_read_slots(16)
    try:
        keep = CustomShortErrorProfile(model_io.synthetic_short_model(n_positions=40, seed=5))
        # slots are a preference: plans whose emit kernel writes the compact layout only say so, and emit into it
        for pod in (PerfectShortErrorProfile().pod(), MinimalShortErrorProfile().pod(), keep.pod()):
            info = engine.pe_plan(1, pod, 1000, 1)
            assert info.slot_bytes == 0
            engine.pe_emit(0, Reads.allocate(info.n_reads, info.total_bases, engine.device, 33, slot_bytes=0))
        assert engine.long_plan([1], [10], MinimalLongErrorProfile().pod(), 1).slot_bytes == 0
        assert engine.pe_plan(1, ph, 1000, 1).slot_bytes == 16
        # a caller that expects the compact layout is not handed slots (and the other way round)
        info = engine.pe_plan(1, ph, 1000, 1)
        out = Reads.allocate(info.n_reads, info.total_bases, engine.device, 33, slot_bytes=0)
        with pytest.raises(SimmrError) as ei:
            engine.pe_emit(0, out)
        assert ei.value.code == _abi.EINVAL
    finally:
        engine.set_read_slots(0)
    info = engine.pe_plan(1, ph, 1000, 1)  # the setting is taken at plan time
    assert info.slot_bytes == 0
    out = Reads.allocate(info.n_reads, info.total_bases + 4096, engine.device, 33, slot_bytes=16)
    with pytest.raises(SimmrError) as ei:
        engine.pe_emit(0, out)
    assert ei.value.code == _abi.EINVAL


def test_slot16_fastq_text_is_the_compact_text(slots, genome_multi, genome_1m):
    """simmr_fastq_plan / simmr_fastq_emit read either layout; the text straight from the plan does not depend on it"""
    eng = slots
    n = len(genome_multi.contigs)
    names = [(1, "genome-one", ["chrA something long", "b", "c c", "d" * 40, "e"][:n]), (0, "7700123", ["synth_1M"])]
    for prof, reads in ((MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod(), 3001),
                        (MinimalShortErrorProfile(read_length=9, insert_size=5, rng_mode=_abi.RNG_PHILOX).pod(), 333),
                        (MinimalShortErrorProfile(read_length=37, insert_size=80, rng_mode=_abi.RNG_PHILOX).pod(), 1500)):
        for fmt in (FMT, "@{:read_id:}"):
            eng.set_read_slots(0)
            want = eng.fastq(eng.simulate_pe_reads_from_genome(1, prof, reads, 11, qual_offset=33), fmt, names, True).cpu().numpy().tobytes()
            eng.set_read_slots(16)
            dev = eng.simulate_pe_reads_from_genome(1, prof, reads, 11, qual_offset=33)
            assert dev.slot_bytes == 16
            assert eng.fastq(dev, fmt, names, True).cpu().numpy().tobytes() == want
            eng.pe_plan(1, prof, reads, 11)
            assert eng.fastq_direct(fmt, names, 0).cpu().numpy().tobytes() == want
    lp = MinimalLongErrorProfile(gamma_mean=2500.0, gamma_std=2000.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX).pod()
    eng.set_read_slots(0)
    want = eng.fastq(eng.simulate_long_reads([1, 0], [70, 45], lp, 5, qual_offset=33), FMT, names, False).cpu().numpy().tobytes()
    eng.set_read_slots(16)
    assert eng.fastq(eng.simulate_long_reads([1, 0], [70, 45], lp, 5, qual_offset=33), FMT, names, False).cpu().numpy().tobytes() == want
