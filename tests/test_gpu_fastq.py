"""Device FASTQ framing (simmr_fastq_plan / simmr_fastq_emit) against a restatement of
fastq.rs:32-121 applied to the same SoA columns: byte-identical records, including the
chained String::replace of the header template."""
import numpy as np
import pytest

from simmr_amd import MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectShortErrorProfile, SimmrError, _abi
from tests import _synth
from tests.test_gpu_cli import FMT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome_1m(engine):
    from tests import _oracle
    contig = _synth.synthetic_contigs([1_000_000], 1)[0]
    engine.stage_genome(0, [contig])
    return _oracle.HostGenome([contig])


@pytest.fixture(scope="module")
def genome_multi(engine):
    from tests import _oracle
    contigs = _synth.synthetic_contigs([300_000, 90_001, 30_017, 70_000, 123_457], 7)
    engine.stage_genome(1, contigs)
    return _oracle.HostGenome(contigs)


def _check(engine, reads, names, fmt, paired):
    got = engine.fastq(reads, fmt, names, paired).cpu().numpy().tobytes()
    d = reads.to_host()
    by_slot = {slot: (gid, sids) for slot, gid, sids in names}
    want = bytearray()
    for r in range(reads.n_reads):
        gid, sids = by_slot[int(d["genome"][r])]
        h = fmt  # fastq.rs:34-56: the replace calls in the reference's order
        for k, v in (("{:genome_id:}", gid), ("{:read_id:}", str(int(d["read_id"][r]))),
                     ("{:sequence_id:}", sids[int(d["contig"][r])]), ("{:start_position:}", str(int(d["start"][r]))),
                     ("{:end_position:}", str(int(d["end"][r]))), ("{:reverse_complement:}", "t" if d["flags"][r] & 1 else "f"),
                     ("{:pair:}", "2" if (paired and r & 1) else "1")):
            h = h.replace(k, v)
        a, b = int(d["seq_off"][r]), int(d["seq_off"][r + 1])
        want += h.encode() + b"\n" + d["seq"][a:b].tobytes() + b"\n+\n" + d["qual"][a:b].tobytes() + b"\n"
    assert len(got) == len(want)
    if got != bytes(want):
        g, w = np.frombuffer(got, np.uint8), np.frombuffer(bytes(want), np.uint8)
        i = int(np.flatnonzero(g != w)[0])
        raise AssertionError(f"first difference at byte {i}: {got[max(0, i - 60):i + 20]!r} vs {bytes(want)[max(0, i - 60):i + 20]!r}")


@pytest.mark.parametrize("fmt", [
    FMT,
    "@{:read_id:}",
    "{:pair:}{:pair:}x{:reverse_complement:}{:genome_id:}{:genome_id:} {:end_position:}-{:start_position:} {:sequence_id:}{:",
    "@r{:read_id:}/{:pair:} {:s{:reverse_complement:}art_position:} {:unknown:} {{:pair:}:read_id:}",
    "",
])
def test_fastq_pe(engine, genome_multi, fmt):
    names = [(1, "genome-one", ["chrA something long", "b", "c c", "d" * 40][: len(genome_multi.contigs)] +
              ["x%d" % i for i in range(max(0, len(genome_multi.contigs) - 4))])]
    for prof, n in ((PerfectShortErrorProfile().pod(), 2001), (MinimalShortErrorProfile(read_length=37, insert_size=80).pod(), 1500),
                    (MinimalShortErrorProfile(read_length=9, insert_size=5).pod(), 333)):
        reads = engine.simulate_pe_reads_from_genome(1, prof, n, 11, read_id_base=4_294_000_000 if n == 333 else 0, qual_offset=33)
        _check(engine, reads, names, fmt, True)


def test_fastq_long_multi_genome(engine, genome_multi, genome_1m):
    lp = MinimalLongErrorProfile(gamma_mean=2500.0, gamma_std=2000.0, length_mode=_abi.LEN_PER_READ).pod()
    reads = engine.simulate_long_reads([1, 0], [70, 45], lp, 5, qual_offset=33)
    names = [(1, "g1", ["ctg%d" % i for i in range(len(genome_multi.contigs))]), (0, "7700123", ["synth_1M"])]
    _check(engine, reads, names, FMT, False)
    _check(engine, reads, names, "@{:sequence_id:}", False)


def test_fastq_longest_headers(engine, genome_multi):
    """Headers close to the 255-byte limit: the header slots then need more LDS than the default launch limit."""
    n = len(genome_multi.contigs)
    names = [(1, "g" * 60, ["contig %d " % i + "x" * 100 for i in range(n)])]
    reads = engine.simulate_pe_reads_from_genome(1, MinimalShortErrorProfile(read_length=40, insert_size=60).pod(), 3000, 2, qual_offset=33)
    _check(engine, reads, names, FMT, True)


def test_fastq_left_to_the_host(engine, genome_multi):
    reads = engine.simulate_pe_reads_from_genome(1, PerfectShortErrorProfile().pod(), 100, 1, qual_offset=33)
    n = len(genome_multi.contigs)
    for names, fmt in (([(1, "id{with}braces", ["c"] * n)], FMT), ([(1, "g", ["{:pair:}"] + ["c"] * (n - 1))], FMT),
                       ([(1, "g", ["c" * 300] * n)], FMT), ([(1, "g", ["c"] * n)], "{:pair:}x" * 13),
                       ([(0, "wrong slot", ["c"])], FMT),
                       # the chain of replacements would complete "{:read_id:}" out of the template's own braces
                       ([(1, "id", ["c"] * n)], "@{:read_{:genome_id:}:}/{:pair:}")):
        with pytest.raises(SimmrError) as ei:
            engine.fastq(reads, fmt, names, True)
        assert ei.value.code == _abi.ENOTSUP
    with pytest.raises(SimmrError) as ei:  # a genome index the engine has never seen (0xffffffff + 1 must not wrap)
        engine.fastq(reads, FMT, [(0xffffffff, "g", ["c"])], True)
    assert ei.value.code == _abi.EINVAL
    # an empty shard is an empty file
    empty = engine.simulate_pe_reads_from_genome(1, PerfectShortErrorProfile().pod(), 100, 1, first=50, count=0, qual_offset=33)
    assert engine.fastq(empty, FMT, [(1, "g", ["c"] * n)], True).numel() == 0


def test_fastq_random_templates(engine, genome_multi):
    """Random header templates built from placeholders, placeholder fragments, braces and text: the device's
    compiled template must equal the reference's chain of String::replace calls (fastq.rs:34-56) on every one."""
    rng = np.random.default_rng(77)
    pieces = ["{:genome_id:}", "{:read_id:}", "{:sequence_id:}", "{:start_position:}", "{:end_position:}",
              "{:reverse_complement:}", "{:pair:}", "{:", ":}", "{", "}", ":", "{:pair", "read_id:}", "{:s", "art_position:}",
              "@", " ", "|", "sp=", "x", "{:genome_id:", "{:{:pair:}"]
    n = len(genome_multi.contigs)
    names = [(1, "G-1", ["ctg_%d z" % i for i in range(n)])]
    reads = engine.simulate_pe_reads_from_genome(1, MinimalShortErrorProfile(read_length=33, insert_size=50).pod(), 400, 6, qual_offset=33)
    for _ in range(40):
        fmt = "".join(rng.choice(pieces, size=int(rng.integers(0, 12))))
        try:
            _check(engine, reads, names, fmt, True)
        except SimmrError as ex:  # more than 24 pieces / 256 literal bytes: left to the host
            assert ex.code == _abi.ENOTSUP


# ---- simmr_fastq_plan_direct / simmr_emit_fastq: the text straight from the plan ------------------------------------
def _two_step(engine, reads, names, fmt, paired):
    return engine.fastq(reads, fmt, names, paired).cpu().numpy().tobytes()


def _same_text(got, want, what=""):
    assert len(got) == len(want), (what, len(got), len(want))
    if got != want:
        g, w = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        i = int(np.flatnonzero(g != w)[0])
        raise AssertionError(f"{what}first difference at byte {i}: {got[max(0, i - 70):i + 20]!r} vs {want[max(0, i - 70):i + 20]!r}")


PE_NAMES = lambda g: [(1, "genome-one", ["chrA something long", "b", "c c", "d" * 40][: len(g.contigs)] +
                       ["x%d" % i for i in range(max(0, len(g.contigs) - 4))])]


@pytest.mark.parametrize("fmt", [FMT, "@{:read_id:}", "", "{:pair:}{:pair:}x{:reverse_complement:}{:genome_id:} {:end_position:}-{:start_position:} {:sequence_id:}{:"])
def test_fastq_direct_pe_equals_two_steps(engine, genome_multi, fmt):
    """Counter mode (the emit kernel writes into the text) and the profiles served through the engine's own columns:
    the same bytes as simmr_pe_emit + simmr_fastq_plan + simmr_fastq_emit, the same counters."""
    names = PE_NAMES(genome_multi)
    profs = [(MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod(), 6001, 0),
             (MinimalShortErrorProfile(read_length=37, insert_size=80, rng_mode=_abi.RNG_PHILOX).pod(), 1500, 0),
             (MinimalShortErrorProfile(read_length=9, insert_size=5, rng_mode=_abi.RNG_PHILOX).pod(), 333, 4_294_000_000),
             (MinimalShortErrorProfile(read_length=16, insert_size=16, rng_mode=_abi.RNG_PHILOX).pod(), 700, 9),
             (MinimalShortErrorProfile(read_length=333, insert_size=100, mean_phred_score=2, rng_mode=_abi.RNG_PHILOX).pod(), 900, 0),
             (PerfectShortErrorProfile().pod(), 2001, 0),
             (MinimalShortErrorProfile(read_length=37, insert_size=80).pod(), 800, 3)]
    for prof, n, idb in profs:
        engine.counters_reset()
        reads = engine.simulate_pe_reads_from_genome(1, prof, n, 11, first=2, count=n // 2 - 5, read_id_base=idb, qual_offset=33)
        c2 = engine.counters()
        want = _two_step(engine, reads, names, fmt, True)
        engine.counters_reset()
        engine.pe_plan(1, prof, n, 11, 2, n // 2 - 5)
        got = engine.fastq_direct(fmt, names, idb).cpu().numpy().tobytes()
        _same_text(got, want, f"L={prof.read_length} ")
        assert np.array_equal(engine.counters(), c2)


def test_fastq_direct_long_and_multi_genome(engine, genome_multi, genome_1m):
    names = [(1, "g1", ["ctg%d" % i for i in range(len(genome_multi.contigs))]), (0, "7700123", ["synth_1M"])]
    for lp in (MinimalLongErrorProfile(gamma_mean=2500.0, gamma_std=2000.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX).pod(),
               MinimalLongErrorProfile(gamma_mean=2500.0, gamma_std=2000.0, length_mode=_abi.LEN_PER_READ).pod()):
        reads = engine.simulate_long_reads([1, 0], [70, 45], lp, 5, first=3, count=100, read_id_base=17, qual_offset=33)
        for fmt in (FMT, "@{:sequence_id:}"):
            want = _two_step(engine, reads, names, fmt, False)
            engine.long_plan([1, 0], [70, 45], lp, 5, 3, 100)
            got = engine.fastq_direct(fmt, names, 17).cpu().numpy().tobytes()
            _same_text(got, want)
    # pairs of two genomes in one plan
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    reads = engine.simulate_pe_reads_multi([1, 0], [3000, 2000], prof, 17, qual_offset=33)
    want = _two_step(engine, reads, names, FMT, True)
    engine.pe_plan_multi([1, 0], [3000, 2000], prof, 17)
    _same_text(engine.fastq_direct(FMT, names, 0).cpu().numpy().tobytes(), want)


def test_fastq_direct_longest_headers_and_refusals(engine, genome_multi):
    n = len(genome_multi.contigs)
    prof = MinimalShortErrorProfile(read_length=40, insert_size=60, rng_mode=_abi.RNG_PHILOX).pod()
    names = [(1, "g" * 60, ["contig %d " % i + "x" * 100 for i in range(n)])]
    reads = engine.simulate_pe_reads_from_genome(1, prof, 3000, 2, qual_offset=33)
    want = _two_step(engine, reads, names, FMT, True)
    engine.pe_plan(1, prof, 3000, 2)
    _same_text(engine.fastq_direct(FMT, names).cpu().numpy().tobytes(), want)
    for bad_names, fmt in (([(1, "id{with}braces", ["c"] * n)], FMT), ([(1, "g", ["c" * 300] * n)], FMT),
                           ([(1, "g", ["c"] * n)], "{:pair:}x" * 13), ([(0, "wrong slot", ["c"])], FMT)):
        with pytest.raises(SimmrError) as ei:
            engine.fastq_direct(fmt, bad_names)
        assert ei.value.code == _abi.ENOTSUP
    # an empty shard is an empty file; emit without a direct plan is a state error
    engine.pe_plan(1, prof, 100, 1, 50, 0)
    assert engine.fastq_direct(FMT, [(1, "g", ["c"] * n)]).numel() == 0
    engine.pe_plan(1, prof, 100, 1)
    import torch
    with pytest.raises(SimmrError) as ei:
        engine.emit_fastq(torch.empty(16, dtype=torch.uint8, device=engine.device))
    assert ei.value.code in (_abi.ESTATE, _abi.ERANGE)


def test_fastq_direct_exception_bases(engine):
    """N / '-' runs in the genome (the HAS_EXC instantiations of the text-writing emit kernel), short and long reads."""
    from tests import _oracle
    rng = np.random.default_rng(44)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60000)].copy()
    seq[rng.integers(0, 60000, 5000)] = ord("N")
    seq[rng.integers(0, 60000, 900)] = ord("-")
    seq[7000:7600] = ord("N")
    engine.stage_genome(12, [seq, seq[:21000].copy()])
    names = [(12, "exc-genome", ["first", "second one"])]
    prof = MinimalShortErrorProfile(mean_phred_score=9, rng_mode=_abi.RNG_PHILOX).pod()
    reads = engine.simulate_pe_reads_from_genome(12, prof, 4001, 5, first=1, count=1990, read_id_base=3, qual_offset=33)
    want = _two_step(engine, reads, names, FMT, True)
    engine.pe_plan(12, prof, 4001, 5, 1, 1990)
    _same_text(engine.fastq_direct(FMT, names, 3).cpu().numpy().tobytes(), want, "pairs ")
    # perfect-short goes into the text through the copy-only form of the same kernel (no draws, every quality 60)
    pp = PerfectShortErrorProfile().pod()
    engine.counters_reset()
    reads = engine.simulate_pe_reads_from_genome(12, pp, 3001, 6, first=3, count=1400, read_id_base=9, qual_offset=33)
    c2 = engine.counters()
    want = _two_step(engine, reads, names, FMT, True)
    engine.counters_reset()
    engine.pe_plan(12, pp, 3001, 6, 3, 1400)
    _same_text(engine.fastq_direct(FMT, names, 9).cpu().numpy().tobytes(), want, "perfect pairs ")
    c1 = engine.counters()
    # (k_emit_perfect_pe leaves the ACGT counter alone when the genome has N / '-' runs; the text-writing form counts them)
    keep = [i for i in range(_abi.N_COUNTERS) if i != _abi.CNT_ACGT_BASES]
    assert np.array_equal(c1[keep], c2[keep])
    assert int(c1[_abi.CNT_ACGT_BASES]) == sum(sum(line.count(b) for b in b"ACGT") for line in want.split(b"\n")[1::4])
    lp = MinimalLongErrorProfile(gamma_mean=2500.0, gamma_std=2000.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX).pod()
    reads = engine.simulate_long_reads([12], [60], lp, 8, qual_offset=33)
    want = _two_step(engine, reads, names, FMT, False)
    engine.long_plan([12], [60], lp, 8)
    _same_text(engine.fastq_direct(FMT, names, 0).cpu().numpy().tobytes(), want, "long ")
