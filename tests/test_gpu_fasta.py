"""FASTA record bodies normalised and packed on the device (simmr_stage_fasta) against the host restatement of
needletail 0.4.1 normalize(false) as genome.rs:93-137 applies it (simmr_amd/host: simmr_host_normalize, itself
pinned by the reference's genome_tests.rs fixture in tests/test_host_cpp.py)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def host_normalize(raw: bytes) -> bytes:
    lib = C.CDLL(str(ROOT / "simmr_amd" / "host" / "libsimmr_host.so"))
    lib.simmr_host_normalize.restype = C.c_void_p
    lib.simmr_host_normalize.argtypes = [C.c_char_p, C.c_uint64]
    lib.simmr_host_free.argtypes = [C.c_void_p]
    p = lib.simmr_host_normalize(raw, len(raw))
    try:
        return C.string_at(p)
    finally:
        lib.simmr_host_free(p)


def bodies():
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b"ACGTacgtNnUu-.~RYKMSWBDHVX*\t \r", dtype=np.uint8)

    def body(n, width, crlf=False):
        seq = alphabet[rng.choice(alphabet.size, n, p=np.r_[np.full(8, 0.11), np.full(alphabet.size - 8, 0.12 / (alphabet.size - 8))])]
        out = bytearray()
        for i in range(0, n, width):
            out += seq[i:i + width].tobytes() + (b"\r\n" if crlf else b"\n")
        return bytes(out)
    return [body(70_001, 80), b"", body(15, 60), b"\n\n  \n", body(3_000, 7, crlf=True), body(1_048_576, 61), b"ACGT", body(1023, 1023),
            body(1024, 1 << 20), body(1025, 60)]


@pytest.mark.parametrize("contiguous", [False, True])
def test_stage_fasta_equals_host_normalize(engine, contiguous):
    raw = bodies()
    want = [host_normalize(b) for b in raw]
    min_size = 0 if contiguous else 14
    counts, n_staged = engine.stage_fasta(20, raw, contiguous=contiguous, min_size=min_size)
    assert counts == [len(w) for w in want]
    if contiguous:
        assert n_staged == 1
        whole = b"".join(w + b"N" for w in want)  # genome.rs:121-137
        n_contigs, size = engine.genome_info(20)
        assert n_contigs == 1 and size == sum(len(w) for w in want)  # Seq.size does not count the separators
        assert engine.unstage(20, 0, 0, len(whole)).tobytes() == whole
    else:
        kept = [w for w in want if len(w) > min_size]  # main.rs:117-162
        assert n_staged == len(kept) and len(kept) < len(want)
        n_contigs, size = engine.genome_info(20)
        assert n_contigs == len(kept) and size == sum(len(w) for w in kept)
        for c, w in enumerate(kept):
            assert engine.unstage(20, c, 0, len(w)).tobytes() == w, c


def test_stage_fasta_then_simulate(engine, oracle):
    """A genome staged from raw FASTA bytes simulates the same reads as the same genome staged from the host-normalised text."""
    from simmr_amd import MinimalShortErrorProfile
    from tests import _oracle
    raw = bodies()[:1] + bodies()[5:6]
    norm = [np.frombuffer(host_normalize(b), dtype=np.uint8) for b in raw]
    engine.stage_fasta(20, raw)
    engine.stage_genome(21, norm)
    prof = MinimalShortErrorProfile().pod()
    a = engine.simulate_pe_reads_from_genome(20, prof, 3000, 5, qual_offset=33).to_host()
    b = engine.simulate_pe_reads_from_genome(21, prof, 3000, 5, qual_offset=33).to_host()
    for col in ("seq", "qual", "seq_off", "start", "end", "contig", "flags"):
        assert np.array_equal(a[col], b[col]), col
    o = _oracle.simulate_pe(oracle, _oracle.HostGenome(norm), prof, 3000, 5, qual_offset=33).trimmed()
    assert np.array_equal(a["seq"], o["seq"]) and np.array_equal(a["qual"], o["qual"])


def test_stage_fasta_nothing_left(engine):
    from simmr_amd import SimmrError, PerfectShortErrorProfile
    counts, n_staged = engine.stage_fasta(22, [b"ACGT\nAC\n", b"\n"], min_size=100)
    assert counts == [6, 0] and n_staged == 0
    with pytest.raises(SimmrError):  # the slot is not staged
        engine.pe_plan(22, PerfectShortErrorProfile().pod(), 10, 1)
