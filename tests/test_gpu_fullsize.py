"""BASELINE.json configs[1] at FULL size (100 M reads of minimal-short 150 bp PE on a
100 Mbp synthetic genome) checked through size-independent properties, all
evaluated on the device: determinism (same seed -> same checksums), sharded ==
whole (position-sensitive checksums of the shard byte ranges), CSR consistency,
alphabet / quality ranges, mate geometry, id sequence, substitution / Phred
statistics.  Both generator modes."""
import numpy as np
import pytest

from simmr_amd import MinimalShortErrorProfile, PerfectShortErrorProfile, _abi

pytestmark = pytest.mark.gpu

N_READS = 100_000_000
GENOME = 100_000_000


def colsum256(t):
    """position-mod-256 sensitive checksum of a uint8 CUDA tensor (int64[256])"""
    import torch
    n = t.numel() // 256 * 256
    out = torch.zeros(256, dtype=torch.int64, device=t.device)
    step = 1 << 30
    for a in range(0, n, step):
        b = min(n, a + step)
        out += t[a:b].view(-1, 256).sum(dim=0, dtype=torch.int64)
    if n < t.numel():
        out[: t.numel() - n] += t[n:].to(torch.int64)
    return out


def checksum_range(t, a, b):
    """colsum256 of t[a:b], phase-aligned to absolute positions"""
    import torch
    cs = colsum256(t[a:b])
    return torch.roll(cs, a % 256)


@pytest.fixture(scope="module")
def big(engine):
    engine.stage_synthetic(5, [GENOME], 2)
    return engine


@pytest.mark.parametrize("rng_mode", [_abi.RNG_PHILOX_FULL, _abi.RNG_PHILOX, _abi.RNG_REFERENCE], ids=["philox-full", "philox", "reference"])
def test_c2_full_size_properties(big, rng_mode):
    import torch
    eng = big
    prof = MinimalShortErrorProfile(rng_mode=rng_mode).pod()
    eng.counters_reset()
    whole = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, qual_offset=33)
    c = eng.counters()
    n, tb = whole.n_reads, whole.total_bases
    assert n == N_READS and c[_abi.CNT_READS] == N_READS and c[_abi.CNT_BASES] == tb
    off = whole.seq_off[: n + 1]
    lens = off[1:] - off[:-1]
    # CSR: monotone, starts at 0, ends at total; mates share a length; lengths ~ floor(N(150, 15))
    assert int(off[0]) == 0 and int(off[-1]) == tb and bool((lens >= 0).all())
    assert bool((lens[0::2] == lens[1::2]).all())
    lm = lens.double().mean().item()
    assert abs(lm - 149.5) < 0.05 and abs(lens.double().std().item() - 15.0) < 0.1
    # geometry: forward mates ascend, mate 2 is stored with start > end (simulate.rs:295-296), |end-start| == len
    st, en, fl = whole.start[:n], whole.end[:n], whole.flags[:n]
    assert bool(((fl[0::2] & 1) == 0).all()) and bool(((fl[1::2] & 1) == 1).all())
    assert bool(((en[0::2] - st[0::2]) == lens[0::2]).all()) and bool(((st[1::2] - en[1::2]) == lens[1::2]).all())
    assert int(st.min()) >= 0 and int(torch.maximum(st, en).max()) <= GENOME
    # ids: one per pair, in generation order; single contig
    ids = whole.read_id[:n]
    assert bool((ids[0::2] == ids[1::2]).all()) and bool((ids[0::2] == torch.arange(n // 2, device=ids.device, dtype=ids.dtype)).all())
    assert int(whole.contig[:n].max()) == 0
    # alphabet and quality range (+33; the ziggurat tail does reach Phred 94 = 6.4 sigma at this size)
    seq, qual = whole.seq[:tb], whole.qual[:tb]
    hist = torch.bincount(seq[: 2_000_000_000].to(torch.int64), minlength=256)
    assert int(hist.sum() - hist[[65, 67, 71, 84]].sum()) == 0
    assert int(qual.min()) >= 33 and int(qual.max()) <= 33 + 105  # 1.5e10 draws reach ~6.4 sigma
    # statistics from the run counters (SURVEY §8d tolerances)
    rate = c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES]
    assert abs(rate / 0.013404 - 1) < 0.005, rate
    assert abs(c[_abi.CNT_QUAL_SUM] / c[_abi.CNT_BASES] - 29.5) < 0.01
    # the substitutions are really in the bytes: mismatches of forward mates against the staged reference
    # (reads are picked by INDEX: picking by start position would bias the sample — in the reference
    #  algorithm the first u64 of StdRng(pe_seed) decides both fwd_start and the first Phred draw)
    ref = torch.from_numpy(eng.unstage(5, 0, 0, GENOME)).to(seq.device)
    sel = torch.arange(0, 400_000, 2, device=seq.device)
    o, s0, ln = off[sel], st[sel], lens[sel]
    k = torch.arange(64, device=seq.device)
    got = seq[(o[:, None] + k[None, :]).clamp(max=tb - 1)]
    want = ref[(s0[:, None] + k[None, :])]
    valid = k[None, :] < ln[:, None]
    mism = ((got != want) & valid).sum().item() / valid.sum().item()
    assert abs(mism / 0.013404 - 1) < 0.03, mism
    # determinism + sharding: two shards reproduce the whole run byte for byte (checksums)
    cs_seq, cs_qual = colsum256(seq), colsum256(qual)
    half = N_READS // 4  # pairs
    a = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, first=0, count=half, qual_offset=33)
    cut = int(off[2 * half])
    assert a.total_bases == cut
    assert bool((checksum_range(seq, 0, cut) == colsum256(a.seq[:cut])).all())
    assert bool((checksum_range(qual, 0, cut) == colsum256(a.qual[:cut])).all())
    del a
    b = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, first=half, count=N_READS, qual_offset=33, read_id_base=0)
    assert b.total_bases == tb - cut and b.n_reads == N_READS - 2 * half
    assert bool((checksum_range(seq, cut, tb) == torch.roll(colsum256(b.seq[: tb - cut]), cut % 256)).all())
    assert bool((checksum_range(qual, cut, tb) == torch.roll(colsum256(b.qual[: tb - cut]), cut % 256)).all())
    assert bool((b.start[: b.n_reads] == st[2 * half:]).all()) and bool((b.read_id[: b.n_reads] == ids[2 * half:]).all())
    del b
    again = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, qual_offset=33)
    assert bool((colsum256(again.seq[:tb]) == cs_seq).all()) and bool((colsum256(again.qual[:tb]) == cs_qual).all())
    # a different seed gives a different run
    del again
    other = eng.simulate_pe_reads_from_genome(5, prof, 2_000_000, 43, qual_offset=33)
    assert not bool((other.start[:1000] == st[:1000]).all())


def test_c1_perfect_short_full_roundtrip(big):
    """perfect-short at 100 M reads: every read equals the reference slice (mate 2: its
    reverse complement) — checked for ALL reads on the device with gathers."""
    import torch
    eng = big
    prof = PerfectShortErrorProfile().pod()
    r = eng.simulate_pe_reads_from_genome(5, prof, 20_000_000, 42)
    n, tb = r.n_reads, r.total_bases
    assert tb == n * 150 and bool((r.qual[:tb] == 60).all())
    ref = torch.from_numpy(eng.unstage(5, 0, 0, GENOME)).to(r.seq.device)
    comp = torch.zeros(256, dtype=torch.uint8, device=ref.device)
    comp[[65, 67, 71, 84]] = torch.tensor([84, 71, 67, 65], dtype=torch.uint8, device=ref.device)
    seq = r.seq[:tb].view(n, 150)
    k = torch.arange(150, device=ref.device)
    for a in range(0, n, 2_000_000):
        b = min(n, a + 2_000_000)
        st, en = r.start[a:b], r.end[a:b]
        fwd = ref[(st[0::2, None] + k[None, :])]
        assert bool((seq[a:b:2] == fwd).all())
        rc = comp[ref[(st[1::2, None] - 1 - k[None, :])].long()]
        assert bool((seq[a + 1:b:2] == rc).all())


@pytest.mark.parametrize("rng_mode", [0, 1], ids=["reference", "philox"])
def test_c5_custom_long_properties(big, rng_mode):
    """(rng_mode 1 = SIMMR_RNG_PHILOX: the splice's draws from Philox counters; the properties are the same, and the
    edited fraction stays in the band the reference mode's falls in.)
    BASELINE configs[4] in the large: a custom (simmrd-shaped) long-read model, per-read lengths, 500 k reads
    (10 Gbases), checked on the device through size-independent properties: determinism, sharded == whole, the
    constant quality tail, substitutions only where the counters say, alphabet."""
    import torch
    from simmr_amd import CustomShortErrorProfile, model_io
    eng = big
    n_pos = 1000
    prof = CustomShortErrorProfile(model_io.synthetic_long_model(kmer_size=7, n_positions=n_pos, seed=1, n_kmers=4 ** 7,
                                                                  read_length_mean=20000.0, read_length_std=4000.0), rng_mode)
    pod = prof.pod()
    pod.length_mode = _abi.LEN_PER_READ
    pod.long_start_mode = _abi.START_UNIFORM
    n_reads = 500_000
    eng.counters_reset()
    whole = eng.simulate_long_reads([5], [n_reads], pod, 42, qual_offset=33)
    c = eng.counters()
    tb = whole.total_bases
    assert whole.n_reads == n_reads and c[_abi.CNT_READS] == n_reads and c[_abi.CNT_BASES] == tb
    off = whole.seq_off[: n_reads + 1]
    lens = off[1:] - off[:-1]
    assert int(off[0]) == 0 and int(off[-1]) == tb
    assert abs(lens.double().mean().item() - 20000) < 100 and abs(lens.double().std().item() - 4000) < 100
    assert bool((whole.end[:n_reads] - whole.start[:n_reads] == lens).all())
    # alphabet: the synthetic genome has no N, and an alternate with an N would have been refused
    seq = whole.seq[:tb]
    hist = torch.bincount(seq.to(torch.int64), minlength=256)
    assert int(hist[[65, 67, 71, 84]].sum()) == tb
    # qualities: from position n_pos - 1 on one value per read (custom_short.rs:339-350): compare every read's
    # last quality with the one at position n_pos - 1
    at = (off[:-1] + (n_pos - 1)).clamp(max=tb - 1)
    longer = lens > n_pos
    assert bool((whole.qual[at][longer] == whole.qual[(off[1:] - 1)][longer]).all())
    qsum = int(whole.qual[:tb].to(torch.int64).sum().item()) - 33 * tb
    assert c[_abi.CNT_QUAL_SUM] == qsum
    # the substitution counter against the bytes, on the first 2000 reads
    ref = eng.unstage(5, 0, 0, GENOME) if hasattr(eng, "unstage") else None
    if ref is not None:
        refb = torch.from_numpy(np.frombuffer(ref, dtype=np.uint8).copy()).to(seq.device)
        mism = 0
        for r in range(2000):
            a, b = int(off[r]), int(off[r + 1])
            s0 = int(whole.start[r])
            mism += int((seq[a:b] != refb[s0:s0 + (b - a)]).sum().item())
        assert 0.05 < mism / int(off[2000]) < 0.15
    assert 0.05 < c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES] < 0.15
    # both modes draw from the same law: 0.0970 edited bases per base with this model (10 Gbases: the two agree to 1e-4)
    assert abs(c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES] - 0.0970) < 0.0005
    # determinism and sharding: checksums of a shard's byte range
    cs_whole = checksum_range(whole.seq, int(off[123_456]), int(off[223_456]))
    cq_whole = checksum_range(whole.qual, int(off[123_456]), int(off[223_456]))
    part = eng.simulate_long_reads([5], [n_reads], pod, 42, first=123_456, count=100_000, qual_offset=33)
    assert part.total_bases == int(off[223_456]) - int(off[123_456])
    shift = int(off[123_456]) % 256
    assert torch.equal(torch.roll(colsum256(part.seq[:part.total_bases]), shift), cs_whole)
    assert torch.equal(torch.roll(colsum256(part.qual[:part.total_bases]), shift), cq_whole)


def test_c3_minimal_long_properties(big):
    """BASELINE configs[2]: minimal-long, gamma(8000, 6000) per-read lengths, 2 M reads (16 Gbases) in counter mode:
    lengths follow the gamma law, reads lie inside the sequence, the substitution rate and Phred mean are the
    reference's, a shard equals the same range of the whole run."""
    import torch
    from simmr_amd import MinimalLongErrorProfile
    eng = big
    prof = MinimalLongErrorProfile(gamma_mean=8000.0, gamma_std=6000.0, length_mode=_abi.LEN_PER_READ,
                                   rng_mode=_abi.RNG_PHILOX).pod()
    n_reads = 2_000_000
    eng.counters_reset()
    whole = eng.simulate_long_reads([5], [n_reads], prof, 42, qual_offset=33)
    c = eng.counters()
    tb = whole.total_bases
    off = whole.seq_off[: n_reads + 1]
    lens = off[1:] - off[:-1]
    assert c[_abi.CNT_READS] == n_reads and c[_abi.CNT_BASES] == tb and int(off[-1]) == tb
    # floor(Gamma(shape (8/6)^2, scale 6000^2/8000)) saturated to u16: mean a little under 8000, sd about 6000
    assert 7800 < lens.double().mean().item() < 8050 and 5700 < lens.double().std().item() < 6100
    assert int(lens.max()) <= 65535 and bool((whole.end[:n_reads] <= GENOME).all())
    rate = c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES]
    assert abs(rate / 0.013404 - 1) < 0.02 and abs(c[_abi.CNT_QUAL_SUM] / tb - 29.5) < 0.05
    a, b = 700_000, 900_000
    cs = checksum_range(whole.seq, int(off[a]), int(off[b]))
    cq = checksum_range(whole.qual, int(off[a]), int(off[b]))
    part = eng.simulate_long_reads([5], [n_reads], prof, 42, first=a, count=b - a, qual_offset=33)
    shift = int(off[a]) % 256
    assert torch.equal(torch.roll(colsum256(part.seq[:part.total_bases]), shift), cs)
    assert torch.equal(torch.roll(colsum256(part.qual[:part.total_bases]), shift), cq)


def test_c4_many_genomes_one_plan(engine):
    """One GPU's share of BASELINE configs[3]: 125 genomes (5 Mbp each) x 1 M reads of minimal-short in ONE plan:
    every genome gets its reads in order, ids run through, a later shard equals the same range of the whole."""
    import torch
    eng = engine
    n_g, per = 125, 1_000_000
    slots = list(range(100, 100 + n_g))
    for g, s in enumerate(slots):
        eng.stage_synthetic(s, [5_000_000], 1000 + g)
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    eng.counters_reset()
    whole = eng.simulate_pe_reads_multi(slots, [per] * n_g, prof, 42, qual_offset=33)
    n = whole.n_reads
    assert n == n_g * per and eng.counters()[_abi.CNT_READS] == n
    genome = whole.genome[:n].to(torch.int64)
    expect = torch.arange(n, device=genome.device) // per + 100
    assert bool((genome == expect).all())
    ids = whole.read_id[:n].to(torch.int64)
    assert bool((ids == torch.arange(n, device=ids.device) // 2).all())  # simulate.rs:85-89: one id per pair, in order
    off = whole.seq_off[: n + 1]
    # same seed for every genome (simulate.rs:137): genomes with one sequence each draw the same pe_seed list, hence
    # the same lengths
    l0 = (off[1:per + 1] - off[:per])
    l7 = (off[7 * per + 1:8 * per + 1] - off[7 * per:8 * per])
    assert bool((l0 == l7).all())
    a_pair, n_pair = 3 * (per // 2) + 1234, 400_000  # a shard that crosses a genome boundary
    part = eng.simulate_pe_reads_multi(slots, [per] * n_g, prof, 42, first=a_pair, count=n_pair, qual_offset=33)
    a, b = int(off[2 * a_pair]), int(off[2 * (a_pair + n_pair)])
    assert part.total_bases == b - a
    shift = a % 256
    assert torch.equal(torch.roll(colsum256(part.seq[:part.total_bases]), shift), checksum_range(whole.seq, a, b))
    assert torch.equal(torch.roll(colsum256(part.qual[:part.total_bases]), shift), checksum_range(whole.qual, a, b))


def test_c2_full_size_slot16_is_the_compact_run(big):
    """The 16-byte slot layout (SIMMR_SLOT16, the bench's default layout) at BASELINE configs[1]'s full size: the
    layout rules of include/simmr_hip.h over all 100 M reads, padding 0, the same reads as the compact layout — every
    metadata column, the run counters, the byte sums of both streams (padding adds nothing) over the whole run, and
    byte-for-byte equality of three windows of 2 M reads —, all evaluated on the device."""
    import torch
    eng = big
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    eng.counters_reset()
    comp = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, qual_offset=33)
    c_comp = eng.counters()
    n = comp.n_reads
    c_off = comp.seq_off[: n + 1].clone()
    c_len = c_off[1:] - c_off[:-1]
    c_sum_s = int(comp.seq[: comp.total_bases].sum(dtype=torch.int64))
    c_sum_q = int(comp.qual[: comp.total_bases].sum(dtype=torch.int64))
    cols = {k: getattr(comp, k)[:n].clone() for k in ("start", "end", "contig", "genome", "read_id", "flags")}
    wins = [(0, 2_000_000), (n // 2 - 1_000_000, n // 2 + 1_000_000), (n - 2_000_000, n)]
    k = torch.arange(256, device=c_off.device)

    def window_bytes(stream, first, lens, a, b):
        idx = first[a:b, None] + k[None, :]
        live = k[None, :] < lens[a:b, None]
        return torch.where(live, stream[idx.clamp(max=stream.numel() - 1)], torch.zeros((), dtype=torch.uint8, device=stream.device))
    want = [(window_bytes(comp.seq, c_off[:-1], c_len, a, b), window_bytes(comp.qual, c_off[:-1], c_len, a, b)) for a, b in wins]
    del comp
    eng.set_read_slots(16)
    try:
        eng.counters_reset()
        slot = eng.simulate_pe_reads_from_genome(5, prof, N_READS, 42, qual_offset=33)
        c_slot = eng.counters()
    finally:
        eng.set_read_slots(0)
    assert slot.slot_bytes == 16 and slot.n_reads == n
    assert list(c_slot) == list(c_comp)
    for name, col in cols.items():
        assert bool((getattr(slot, name)[:n] == col).all()), name
    L = (slot.end[:n] - slot.start[:n]).abs()
    assert bool((L == c_len).all())
    Lp = (L + 15) // 16 * 16
    place = torch.zeros(n + 1, dtype=torch.int64, device=L.device)
    torch.cumsum(Lp, 0, out=place[1:])
    tb = slot.total_bases
    assert int(place[n]) == tb == int(slot.seq_off[n]) and tb % 16 == 0
    first = slot.seq_off[:n]
    rev = (slot.flags[:n] & _abi.FLAG_REVCOMP) != 0
    assert bool(((first & ~15) == place[:n]).all())                               # a read's slot starts on 16 bytes
    assert bool(((first - place[:n]) == torch.where(rev, Lp - L, torch.zeros_like(L))).all())  # right-aligned iff reverse
    # padding adds nothing: the sums of all bytes are the compact run's
    assert int(slot.seq[:tb].sum(dtype=torch.int64)) == c_sum_s
    assert int(slot.qual[:tb].sum(dtype=torch.int64)) == c_sum_q
    # ... and it is where the rules put it: the last 16 bytes of every slot hold L - (Lp - 16) live bytes
    tail = Lp - L                                                                # padding bytes per read (0..15)
    j = torch.arange(16, device=L.device)
    sel = torch.arange(0, n, 97, device=L.device)                                # a million reads, both mates
    q_last = slot.qual[(place[sel] + Lp[sel] - 16)[:, None] + j[None, :]]
    assert not bool((q_last * (j[None, :] >= (16 - tail[sel])[:, None])).any())
    s_edge = torch.where(rev[sel], place[sel], place[sel] + Lp[sel] - 16)        # reverse mates pad in FRONT
    s_pad = torch.where(rev[sel][:, None], j[None, :] < tail[sel][:, None], j[None, :] >= (16 - tail[sel])[:, None])
    assert not bool((slot.seq[s_edge[:, None] + j[None, :]] * s_pad).any())
    # the reads themselves, byte for byte, in three windows
    for (a, b), (ws, wq) in zip(wins, want):
        assert bool((window_bytes(slot.seq, first, L, a, b) == ws).all())
        assert bool((window_bytes(slot.qual, place[:n], L, a, b) == wq).all())


def test_whole_line_text_is_the_item_form_text_at_full_size(big):
    """BASELINE configs[1] at full size through both forms of the text kernel (simmr_emit_fastq: the item form, the library's
    default, and the whole-line form of text_lines.hip, SIMMR_TEXT_FORM=2): 41 GB of FASTQ text each, compared on the device
    byte for byte, with the run counters; and the text's own structure — every fourth line a '+', header lines start with '@'
    — through position-independent counts."""
    import os
    import torch
    from simmr_amd.engine import Engine
    from tests.test_gpu_cli import FMT
    names = [(5, "0b5e3c9e-7d1c-4c1a-9c55-2f7d3a1b6e42", ["synthetic_100Mbp"])]
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
    texts, counters = [], []
    for form in ("1", "2"):
        old = os.environ.get("SIMMR_TEXT_FORM")
        os.environ["SIMMR_TEXT_FORM"] = form
        try:
            eng = Engine(0)
        finally:
            if old is None:
                del os.environ["SIMMR_TEXT_FORM"]
            else:
                os.environ["SIMMR_TEXT_FORM"] = old
        try:
            eng.stage_synthetic(5, [GENOME], 2)
            eng.counters_reset()
            eng.pe_plan(5, prof, N_READS, 42)
            texts.append(eng.fastq_direct(FMT, names, 0))
            counters.append(eng.counters())
        finally:
            eng.close()
    a, b = texts
    assert a.numel() == b.numel() > 40_000_000_000
    step = 1 << 30
    for lo in range(0, a.numel(), step):
        assert torch.equal(a[lo:lo + step], b[lo:lo + step]), f"first difference in bytes [{lo}, {lo + step})"
    assert np.array_equal(counters[0], counters[1]) and counters[0][_abi.CNT_READS] == N_READS
    # structure: 4 lines per record; one '@' line start and one "\n+\n" per record at least (quality bytes may also be '@' or '+')
    n_nl = sum(int((b[lo:lo + step] == 10).sum()) for lo in range(0, b.numel(), step))
    assert n_nl == 4 * N_READS and int(b[0]) == ord("@") and int(b[-1]) == 10
