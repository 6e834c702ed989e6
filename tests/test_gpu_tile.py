"""GPU parity of the tile form of the counter-mode emit kernel (simmr_amd/csrc/emit_tile.hip).

The tile form measured slower than the item kernel (profiles/r3/tile_form_*) and is not in the product library: it is
compiled into the `make extras` build only, and this module runs only against that build (tests/conftest.py:
needs_extras).  There it is opt-in (SIMMR_PHILOX_FORM=2), for
paired plans whose reads are at most TILE_MAXL bases; here it and its corners are forced:
block sizes from one pair to 32, tiles too small for their block (the direct-store path inside the kernel, for
some or for all blocks), the item kernel on the same plan (SIMMR_PHILOX_FORM=1), reads of fewer than 16 bases
(every item partial), read lengths around TILE_MAXL (the engine must pick the item kernel by itself), output
buffers at every byte alignment, and exact capacity between canaries.  Expected bytes: the CPU restatement of
the mode (oracle/philox.c), bit for bit; reference law: minimal_short.rs:83-140, simulate.rs:260-299.
"""
import numpy as np
import pytest

from simmr_amd import MinimalShortErrorProfile, _abi
from tests import _oracle, _synth
from tests.test_gpu_parity import assert_same

from tests.conftest import needs_extras

pytestmark = [pytest.mark.gpu, needs_extras]

LENS = [300_000, 90_001, 30_017, 70_000, 123_457]

KNOBS = [
    {},                                                     # the tile form with its default block
    {"SIMMR_PHILOX_FORM": "1"},                             # item kernel
    {"SIMMR_TILE_UPB": "17"},
    {"SIMMR_TILE_UPB": "1"},
    {"SIMMR_TILE_UPB": "5", "SIMMR_TILE_WGS_PER_CU": "1"},
    {"SIMMR_TILE_UPB": "32", "SIMMR_TILE_CAP": "256"},      # no block fits: direct stores from the tile kernel
    {"SIMMR_TILE_UPB": "16", "SIMMR_TILE_CAP": "4800"},     # some blocks fit, some do not (mean 150 -> 4800 bytes)
    {"SIMMR_TILE_UPB": "31", "SIMMR_TILE_WGS_PER_CU": "2"},
]


def _engine_with(monkeypatch, knobs):
    from simmr_amd.engine import Engine
    for k in ("SIMMR_PHILOX_FORM", "SIMMR_TILE_UPB", "SIMMR_TILE_CAP", "SIMMR_TILE_WGS_PER_CU"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SIMMR_PHILOX_FORM", "2")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    return Engine(0)


@pytest.fixture(scope="module")
def host_genome():
    return _oracle.HostGenome(_synth.synthetic_contigs(LENS, 7))


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a[6:]}={b}" for a, b in k.items()) or "default")
def test_tile_forms_equal_the_specification(oracle, host_genome, monkeypatch, knobs):
    e = _engine_with(monkeypatch, knobs)
    try:
        e.stage_genome(0, host_genome.contigs)
        for L, I, q, reads, seed in ((150, 150, 30, 20001, 42), (20, 20, 30, 3000, 5), (7, 3, 10, 2222, 6),
                                     (333, 100, 2, 1500, 7), (16, 16, 60, 1999, 8), (490, 200, 25, 1400, 9),
                                     (520, 100, 30, 700, 10), (31, 64, 40, 2601, 11)):
            prof = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=q, rng_mode=_abi.RNG_PHILOX).pod()
            e.counters_reset()
            dev = e.simulate_pe_reads_from_genome(0, prof, reads, seed, first=3, count=reads // 2 - 7, read_id_base=11,
                                                  qual_offset=33)
            ora = _oracle.simulate_pe(oracle, host_genome, prof, reads, seed, first=3, count=reads // 2 - 7, read_id_base=11,
                                      qual_offset=33, max_len=4096)
            d, o = dev.to_host(), ora.trimmed()
            assert_same(d, o, what=f"L={L} ")
            c = e.counters()
            assert c[_abi.CNT_READS] == d["start"].size and c[_abi.CNT_BASES] == d["seq"].size
            assert c[_abi.CNT_QUAL_SUM] == (d["qual"].astype(np.int64) - 33).sum()
    finally:
        e.close()


def test_tile_form_exceptions_and_multi_genome(oracle, monkeypatch):
    """N / '-' runs (the HAS_EXC instantiation) and a plan over several genomes (the not-CACHED one)."""
    rng = np.random.default_rng(33)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 40000)].copy()
    seq[rng.integers(0, 40000, 4000)] = ord("N")
    seq[rng.integers(0, 40000, 700)] = ord("-")
    seq[5000:5400] = ord("N")
    clean = _synth.synthetic_contigs([50_000, 20_000], 3)
    for knobs in ({}, {"SIMMR_TILE_UPB": "11"}, {"SIMMR_TILE_UPB": "7", "SIMMR_TILE_CAP": "1600"}):
        e = _engine_with(monkeypatch, knobs)
        try:
            e.stage_genome(0, [seq, seq[:9000].copy()])
            e.stage_genome(1, clean)
            g0, g1 = _oracle.HostGenome([seq, seq[:9000].copy()]), _oracle.HostGenome(clean)
            prof = MinimalShortErrorProfile(mean_phred_score=8, rng_mode=_abi.RNG_PHILOX).pod()
            dev = e.simulate_pe_reads_from_genome(0, prof, 5000, 8, qual_offset=33)
            ora = _oracle.simulate_pe(oracle, g0, prof, 5000, 8, qual_offset=33)
            assert_same(dev.to_host(), ora.trimmed(), what="exc ")
            # two genomes in one plan: ids run on, the genome column says which
            dev = e.simulate_pe_reads_multi([0, 1], [3000, 2000], prof, 17, qual_offset=33)
            d = dev.to_host()
            o0 = _oracle.simulate_pe(oracle, g0, prof, 3000, 17, qual_offset=33).trimmed()
            o1 = _oracle.simulate_pe(oracle, g1, prof, 2000, 17, read_id_base=1500, qual_offset=33).trimmed()
            n0, b0 = o0["start"].size, o0["seq"].size
            for col in ("start", "end", "contig", "read_id", "flags"):
                assert np.array_equal(d[col], np.concatenate([o0[col], o1[col]])), col
            assert np.array_equal(d["seq"], np.concatenate([o0["seq"], o1["seq"]]))
            assert np.array_equal(d["qual"], np.concatenate([o0["qual"], o1["qual"]]))
            assert np.array_equal(d["genome"], np.concatenate([np.zeros(n0, np.uint32), np.ones(d["start"].size - n0, np.uint32)]))
            assert np.array_equal(d["seq_off"][: n0 + 1], o0["seq_off"]) and np.array_equal(d["seq_off"][n0:], o1["seq_off"] + b0)
        finally:
            e.close()


@pytest.mark.parametrize("shift_q,shift_s", [(0, 0), (1, 15), (7, 3), (15, 8), (4, 4)])
def test_tile_form_any_buffer_alignment(oracle, host_genome, monkeypatch, shift_q, shift_s):
    """The flush writes aligned 16-byte chunks of whatever addresses the caller's buffers have; bytes in front of
    and behind the planned range stay untouched."""
    import torch
    from simmr_amd.engine import Reads
    engine = _engine_with(monkeypatch, {})
    try:
        _alignment_case(engine, oracle, host_genome, shift_q, shift_s, torch, Reads)
    finally:
        engine.close()


def _alignment_case(engine, oracle, host_genome, shift_q, shift_s, torch, Reads):
    engine.stage_genome(9, host_genome.contigs)
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    info = engine.pe_plan(9, prof, 9001, 77)
    out = Reads.allocate(info.n_reads, info.total_bases, engine.device, qual_offset=33)
    tb = info.total_bases
    raw_s = torch.full((tb + 64,), 0xA5, dtype=torch.uint8, device=engine.device)
    raw_q = torch.full((tb + 64,), 0x5A, dtype=torch.uint8, device=engine.device)
    out.seq = raw_s[16 + shift_s: 16 + shift_s + tb]
    out.qual = raw_q[16 + shift_q: 16 + shift_q + tb]
    engine.pe_emit(0, out)
    ora = _oracle.simulate_pe(oracle, host_genome, prof, 9001, 77, qual_offset=33).trimmed()
    assert_same(out.to_host(), ora)
    hs, hq = raw_s.cpu().numpy(), raw_q.cpu().numpy()
    assert (hs[: 16 + shift_s] == 0xA5).all() and (hs[16 + shift_s + tb:] == 0xA5).all()
    assert (hq[: 16 + shift_q] == 0x5A).all() and (hq[16 + shift_q + tb:] == 0x5A).all()
