"""The whole-line form of the FASTQ text kernel (simmr_amd/csrc/text_lines.hip, the default of simmr_emit_fastq for paired
plans of short reads) against the item form it replaces (k_emit_philox<TEXT>, SIMMR_TEXT_FORM=1): the same bytes and the
same run counters for every read length a segment can hold, every header shape, shards that start anywhere in a run,
genomes with N / '-' runs, several genomes in one plan and perfect-short; exact-capacity destinations between canaries;
destinations the whole-line form does not take (not 16-byte aligned) fall back to the item form.

The item form itself is pinned to the oracle through the column path (tests/test_gpu_fastq.py: text == framing of the
emitted columns; tests/test_gpu_parity.py: columns == oracle), and those tests now run the whole-line form."""
import os

import numpy as np
import pytest

from simmr_amd import MinimalShortErrorProfile, PerfectShortErrorProfile, _abi
from tests import _synth
from tests.test_gpu_cli import FMT

pytestmark = pytest.mark.gpu


def _engine_with_text_form(form):
    """An engine whose simmr_emit_fastq runs one form of the text kernel (the knob is read once, when the engine is made)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from simmr_amd.engine import Engine
    old = os.environ.get("SIMMR_TEXT_FORM")
    os.environ["SIMMR_TEXT_FORM"] = form
    try:
        return Engine(0)
    finally:
        if old is None:
            del os.environ["SIMMR_TEXT_FORM"]
        else:
            os.environ["SIMMR_TEXT_FORM"] = old


@pytest.fixture(scope="module")
def item_engine():
    e = _engine_with_text_form("1")
    yield e
    e.close()


@pytest.fixture(scope="module")
def engine():
    """(shadows the session's engine: this file's subject is the whole-line form, whichever form is the library's default)"""
    e = _engine_with_text_form("2")
    yield e
    e.close()


def _stage_both(engine, item_engine, slot, contigs):
    engine.stage_genome(slot, contigs)
    item_engine.stage_genome(slot, contigs)


@pytest.fixture(scope="module")
def genomes(engine, item_engine):
    contigs = _synth.synthetic_contigs([300_000, 90_001, 30_017, 70_000, 123_457], 7)
    _stage_both(engine, item_engine, 1, contigs)
    one = _synth.synthetic_contigs([1_000_000], 1)
    _stage_both(engine, item_engine, 0, one)
    rng = np.random.default_rng(44)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60000)].copy()
    seq[rng.integers(0, 60000, 5000)] = ord("N")
    seq[rng.integers(0, 60000, 900)] = ord("-")
    seq[7000:7600] = ord("N")
    _stage_both(engine, item_engine, 12, [seq, seq[:21000].copy()])
    return {1: len(contigs), 0: 1, 12: 2}


def _names(slot, n):
    return [(slot, "0b5e3c9e-7d1c-4c1a-9c55-2f7d3a1b6e42", ["contig_%d with a description" % i for i in range(n)])]


def _same(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    if got != want:
        g, w = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        i = int(np.flatnonzero(g != w)[0])
        raise AssertionError(f"{what}: first difference at byte {i} of {len(got)} ({int((g != w).sum())} differ): "
                             f"{got[max(0, i - 70):i + 24]!r} vs {want[max(0, i - 70):i + 24]!r}")


def _both(engine, item_engine, plan, fmt, names, idb, what):
    texts, counters = [], []
    for e in (engine, item_engine):
        e.counters_reset()
        plan(e)
        texts.append(e.fastq_direct(fmt, names, idb).cpu().numpy().tobytes())
        counters.append(e.counters())
    _same(texts[0], texts[1], what)
    assert np.array_equal(counters[0], counters[1]), what
    return texts[0]


# (read length, its standard deviation): minimal_short.rs:33-40 draws every pair's length from Normal(read_length, std);
# the last rows stay below TL_MAXL = 256 bases (a plan with a longer read goes to the item form as a whole)
LENGTHS = [(1, 0.4), (2, 1.0), (7, 3.0), (15, 1.0), (16, 0.6), (17, 2.0), (31, 6.0), (33, 15.0), (100, 15.0), (150, 15.0), (140, 22.0),
           (200, 10.0), (240, 3.0), (250, 1.0), (256, 0.0)]


@pytest.mark.parametrize("rng_mode", [_abi.RNG_PHILOX, _abi.RNG_PHILOX_FULL])
def test_every_length_and_shard_offset(engine, item_engine, genomes, rng_mode):
    """Read lengths around every item boundary (1 .. 256 bases: one to sixteen items per read), shards that begin at odd
    pairs (blocks, segments and lines then start anywhere), more than one block, the last block partial."""
    names = _names(1, genomes[1])
    seen = set()
    for k, (L, sd) in enumerate(LENGTHS):
        prof = MinimalShortErrorProfile(read_length=L, read_length_std=sd, insert_size=max(L // 2, 1), insert_size_std=max(L / 4.0, 0.5),
                                        mean_phred_score=20 + (k % 3) * 9, rng_mode=rng_mode).pod()
        n, first = 2 * (700 + 37 * k), 1 + 3 * k
        count = n // 2 - first - (k % 4)
        text = _both(engine, item_engine, lambda e: e.pe_plan(1, prof, n, 100 + k, first, count), FMT, names, 5 * k, f"L={L} ")
        lines = text.split(b"\n")
        assert len(lines) == 4 * 2 * count + 1 and lines[-1] == b""
        assert all(len(a) == len(b) for a, b in zip(lines[1::4], lines[3::4]))
        lens = {len(x) for x in lines[1::4]}
        assert max(lens) <= 256, "this row was meant for the whole-line form"
        seen |= lens
    assert {1, 15, 16, 17, 32, 33, 150, 208, 240, 256} <= seen


@pytest.mark.parametrize("fmt", [FMT, "@{:read_id:}", "", "{:pair:}{:pair:}x{:reverse_complement:}{:genome_id:} {:end_position:}-{:start_position:} {:sequence_id:}{:",
                                 "@" + "{:genome_id:}|" * 5 + "{:sequence_id:}"])
def test_header_shapes(engine, item_engine, genomes, fmt):
    """Headers from none at all to ~250 bytes (the slot pitch, the tasks per read and the LDS the kernel asks for follow)."""
    names = _names(1, genomes[1])
    for L, n in ((150, 6002), (37, 3000), (9, 900)):
        prof = MinimalShortErrorProfile(read_length=L, insert_size=2 * L, rng_mode=_abi.RNG_PHILOX).pod()
        _both(engine, item_engine, lambda e: e.pe_plan(1, prof, n, 11, 2, n // 2 - 5), fmt, names, 4_294_000_000 if L == 9 else 0, f"L={L} ")


def test_exception_bases_multi_genome_and_perfect_short(engine, item_engine, genomes):
    prof = MinimalShortErrorProfile(mean_phred_score=9, rng_mode=_abi.RNG_PHILOX).pod()
    _both(engine, item_engine, lambda e: e.pe_plan(12, prof, 4001, 5, 1, 1990), FMT, _names(12, 2), 3, "N / '-' runs ")
    pp = PerfectShortErrorProfile().pod()
    _both(engine, item_engine, lambda e: e.pe_plan(12, pp, 3001, 6, 3, 1400), FMT, _names(12, 2), 9, "perfect-short, N runs ")
    _both(engine, item_engine, lambda e: e.pe_plan(1, pp, 5000, 6), "@{:read_id:}/{:pair:}", _names(1, genomes[1]), 0, "perfect-short ")
    full = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
    names = _names(1, genomes[1]) + [(0, "7700123", ["synth_1M"])]
    _both(engine, item_engine, lambda e: e.pe_plan_multi([1, 0], [3000, 2000], full, 17), FMT, names, 0, "two genomes in one plan ")
    # a quality offset / mean that can escape to level 2 of the draw (mean Phred 2: escapes are frequent)
    low = MinimalShortErrorProfile(read_length=100, insert_size=50, mean_phred_score=2, rng_mode=_abi.RNG_PHILOX).pod()
    _both(engine, item_engine, lambda e: e.pe_plan(1, low, 3000, 8), FMT, _names(1, genomes[1]), 0, "mean Phred 2 ")


def test_exact_capacity_between_canaries_and_unaligned_destinations(engine, item_engine, genomes):
    """The text of a shard into a destination of exactly its size cut out of a canary-filled buffer: at a 64-byte-aligned
    place, at 16-byte-aligned places that are not line-aligned (the first and last chunks of segments and blocks are then
    partial in every way), and at an odd address (the whole-line form does not take it: the item form writes it)."""
    import torch
    names = _names(1, genomes[1])
    prof = MinimalShortErrorProfile(read_length=151, insert_size=60, rng_mode=_abi.RNG_PHILOX_FULL).pod()
    PAD = 4096
    item_engine.pe_plan(1, prof, 3000, 21, 7, 1400)
    want = item_engine.fastq_direct(FMT, names, 1).cpu().numpy().tobytes()
    for off in (0, 16, 48, 1, 7):
        engine.pe_plan(1, prof, 3000, 21, 7, 1400)
        total = engine.fastq_plan_direct(FMT, names, 1)
        assert total == len(want)
        buf = torch.full((PAD + off + total + PAD,), 0xEE, dtype=torch.uint8, device=engine.device)
        assert buf.data_ptr() % 64 == 0
        engine.emit_fastq(buf[PAD + off: PAD + off + total])
        host = buf.cpu().numpy()
        assert (host[: PAD + off] == 0xEE).all() and (host[PAD + off + total:] == 0xEE).all(), f"offset {off}: a canary was written"
        _same(host[PAD + off: PAD + off + total].tobytes(), want, f"destination offset {off} ")


def test_one_pair_and_tiny_shards(engine, item_engine, genomes):
    names = _names(0, 1)
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
    for count in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129):
        _both(engine, item_engine, lambda e: e.pe_plan(0, prof, 1000, 3, 5, count), FMT, names, 0, f"{count} pairs ")
