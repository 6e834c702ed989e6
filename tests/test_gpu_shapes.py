"""The BASELINE.json configurations in their own SHAPE, where one GPU holds them:
  C4  a thousand staged genomes in one plan, a rank's shard of the global pair range (ranks 0, 3, 7 of 8), checked
      against per-genome runs of the CPU restatement on the genomes at the shard's borders and in its middle;
  C3  10 M long reads of gamma(8000, 6000) (80 Gbases, 160 GB of output) through size-independent properties;
  C5  one GPU's share of 50 M custom-model long reads over 64 staged genomes with 1/(g+1) abundances
      (CustomAbundanceProfile): the whole share is planned at once and emitted range by range;
and that an engine's work really lands on the HIP stream it was given (simmr_engine_set_stream)."""
import numpy as np
import pytest

from simmr_amd import (CustomAbundanceProfile, MinimalLongErrorProfile, MinimalShortErrorProfile, _abi)
from simmr_amd.simulate import GenomeRef, determine_reads, split_range
from tests import _oracle, _synth
from tests.test_gpu_fullsize import checksum_range, colsum256

ROOT = __import__("pathlib").Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome_multi(engine):
    lens = [300_000, 90_001, 30_017, 70_000, 123_457]
    contigs = _synth.synthetic_contigs(lens, 7)
    engine.stage_genome(1, contigs)
    return _oracle.HostGenome(contigs)


@pytest.mark.parametrize("rng_mode", [_abi.RNG_REFERENCE, _abi.RNG_PHILOX, _abi.RNG_PHILOX_FULL], ids=["reference", "philox", "philox-full"])
def test_c4_thousand_genomes_rank_shards(engine, oracle, rng_mode):
    """1000 genomes (one sequence of 30 kbp each, SplitMix64(1000 + g)), uniform abundance, 2 M reads: rank r of 8
    takes pairs [r N / 8, (r + 1) N / 8) of the global pair index through ONE plan over all genomes."""
    n_g, per = 1000, 2000  # reads per genome (uniform.rs: ceil(N / G))
    slots = list(range(1000, 1000 + n_g))
    for g, s in enumerate(slots):
        engine.stage_synthetic(s, [30_000], 1000 + g)
    prof = MinimalShortErrorProfile(rng_mode=rng_mode).pod()
    reads = [per] * n_g
    total_pairs = n_g * (per // 2)
    for rank in (0, 3, 7):
        first, count = split_range(total_pairs, rank, 8)
        dev = engine.simulate_pe_reads_multi(slots, reads, prof, 42, first=first, count=count, qual_offset=33).to_host()
        assert dev["read_id"].size == 2 * count
        # ids and genomes of the whole shard follow from the global pair index alone
        pair = first + np.arange(count)
        assert np.array_equal(dev["read_id"], np.repeat(pair, 2).astype(np.uint32))  # simulate.rs:85-89
        assert np.array_equal(dev["genome"], np.repeat(pair // (per // 2) + 1000, 2).astype(np.uint32))
        g_lo, g_hi = first // (per // 2), (first + count - 1) // (per // 2)
        for g in sorted({g_lo, (g_lo + g_hi) // 2, g_hi}):
            host = _oracle.HostGenome(_synth.synthetic_contigs([30_000], 1000 + g))
            a = max(first, g * (per // 2)) - g * (per // 2)            # pairs [a, b) of genome g are in the shard
            b = min(first + count, (g + 1) * (per // 2)) - g * (per // 2)
            ora = _oracle.simulate_pe(oracle, host, prof, per, 42, first=a, count=b - a,
                                      read_id_base=g * (per // 2), qual_offset=33).trimmed()
            r0 = 2 * (g * (per // 2) + a - first)                      # first read of that range in the shard
            r1 = r0 + 2 * (b - a)
            o0 = int(dev["seq_off"][r0])
            assert np.array_equal(dev["seq_off"][r0:r1 + 1] - o0, ora["seq_off"]), (rank, g)
            for col in ("start", "end", "contig", "read_id", "flags"):
                assert np.array_equal(dev[col][r0:r1], ora[col]), (rank, g, col)
            o1 = int(dev["seq_off"][r1])
            assert np.array_equal(dev["seq"][o0:o1], ora["seq"]), (rank, g)
            assert np.array_equal(dev["qual"][o0:o1], ora["qual"]), (rank, g)


def test_c3_ten_million_long_reads(engine):
    """BASELINE configs[2] at its stated size: 10 M reads of gamma(8000, 6000) on the 100 Mbp genome, counter mode."""
    import torch
    eng = engine
    eng.stage_synthetic(5, [100_000_000], 2)
    prof = MinimalLongErrorProfile(gamma_mean=8000.0, gamma_std=6000.0, length_mode=_abi.LEN_PER_READ,
                                   rng_mode=_abi.RNG_PHILOX).pod()
    n_reads = 10_000_000
    eng.counters_reset()
    whole = eng.simulate_long_reads([5], [n_reads], prof, 42, qual_offset=33)
    torch.cuda.synchronize()
    c = eng.counters()
    tb = whole.total_bases
    off = whole.seq_off[: n_reads + 1]
    lens = off[1:] - off[:-1]
    assert c[_abi.CNT_READS] == n_reads and c[_abi.CNT_BASES] == tb and int(off[0]) == 0 and int(off[-1]) == tb
    assert 7.7e10 < tb < 8.1e10
    assert 7800 < lens.double().mean().item() < 8050 and 5700 < lens.double().std().item() < 6100
    assert int(lens.max()) <= 65535 and int(lens.min()) >= 1
    assert bool((whole.end[:n_reads] - whole.start[:n_reads] == lens).all()) and bool((whole.end[:n_reads] <= 100_000_000).all())
    rate = c[_abi.CNT_SUBSTITUTIONS] / c[_abi.CNT_ACGT_BASES]
    assert abs(rate / 0.013404 - 1) < 0.02 and abs(c[_abi.CNT_QUAL_SUM] / tb - 29.5) < 0.05
    assert bool((whole.read_id[:n_reads].to(torch.int64) == torch.arange(n_reads, device=off.device)).all())
    # alphabet and quality range on a slice from the far end of the 160 GB
    a, b = int(off[n_reads - 20_000]), tb
    hist = torch.bincount(whole.seq[a:b].to(torch.int64), minlength=256)
    assert int(hist[[65, 67, 71, 84]].sum()) == b - a
    assert int(whole.qual[a:b].min()) >= 33 and int(whole.qual[a:b].max()) <= 33 + 93
    # a shard deep inside equals the same range of the whole run
    s0, s1 = 9_200_000, 9_300_000
    cs = checksum_range(whole.seq, int(off[s0]), int(off[s1]))
    cq = checksum_range(whole.qual, int(off[s0]), int(off[s1]))
    base = int(off[s0])
    n_part = int(off[s1]) - base
    del whole
    torch.cuda.empty_cache()
    part = eng.simulate_long_reads([5], [n_reads], prof, 42, first=s0, count=s1 - s0, qual_offset=33)
    assert part.total_bases == n_part
    assert torch.equal(torch.roll(colsum256(part.seq[:n_part]), base % 256), cs)
    assert torch.equal(torch.roll(colsum256(part.qual[:n_part]), base % 256), cq)


def test_c5_share_of_one_gpu(engine):
    """BASELINE configs[4], one GPU's share: 64 genomes of 10 Mbp, abundances 1/(g + 1) through CustomAbundanceProfile
    (normalised by custom.rs:29-38), 50 M reads in the run, rank 0 of 8 = 6.25 M reads of the custom long-read model.
    The share is PLANNED at once (the run layout over the genomes, lengths, positions) and emitted in ranges of
    1 M reads (the whole share would be 250 GB of output); the ranges tile the plan."""
    import torch
    from simmr_amd import CustomShortErrorProfile, model_io
    eng = engine
    n_g = 64
    slots = list(range(300, 300 + n_g))
    for g, s in enumerate(slots):
        eng.stage_synthetic(s, [10_000_000], 500 + g)
    keep = CustomShortErrorProfile(model_io.synthetic_long_model(kmer_size=7, n_positions=200, seed=1, n_kmers=4 ** 7,
                                                                 read_length_mean=20000.0, read_length_std=4000.0))
    pod = keep.pod()
    pod.length_mode = _abi.LEN_PER_READ
    pod.long_start_mode = _abi.START_UNIFORM
    refs = [GenomeRef(s, 10_000_000, f"g{g}.fna", f"id{g}", 1) for g, s in enumerate(slots)]
    ab = determine_reads(50_000_000, refs, keep, CustomAbundanceProfile([1.0 / (g + 1) for g in range(n_g)]), False)
    reads = [r for r, _ in ab]
    total = sum(reads)
    assert 50_000_000 <= total <= 50_000_000 + n_g and reads[0] > reads[1] > reads[-1] > 0
    first, count = split_range(total, 0, 8)
    info = eng.long_plan(slots, reads, pod, 42, first, count)
    assert info.n_reads == count and 1.2e11 < info.total_bases < 1.3e11   # ~20 kb each
    bounds = np.cumsum([0] + reads)
    done_bases = 0
    for a in range(0, count, 1_000_000):
        n = min(1_000_000, count - a)
        part = eng.simulate_long_reads(slots, reads, pod, 42, first=first + a, count=n, qual_offset=33)
        tb = part.total_bases
        off = part.seq_off[: n + 1]
        lens = off[1:] - off[:-1]
        assert abs(lens.double().mean().item() - 20000) < 150 and int(off[-1]) == tb
        gidx = np.searchsorted(bounds, first + a + np.arange(0, n, 50_000), side="right") - 1
        got = part.genome[:n][::50_000].cpu().numpy()
        assert np.array_equal(got, np.array(slots)[gidx])          # reads go to the genomes in run order
        ids = part.read_id[:n].to(torch.int64)
        assert bool((ids == first + a + torch.arange(n, device=ids.device)).all())
        assert bool((part.end[:n] - part.start[:n] == lens).all()) and bool((part.end[:n] <= 10_000_000).all())
        hist = torch.bincount(part.seq[: min(tb, 1 << 28)].to(torch.int64), minlength=256)
        assert int(hist[[65, 67, 71, 84]].sum()) == min(tb, 1 << 28)
        done_bases += tb
        del part
        torch.cuda.empty_cache()
    assert done_bases == info.total_bases                           # the ranges tile the planned share


def test_engine_set_stream_puts_the_work_on_that_stream(engine):
    """simmr_engine_set_stream: the emit kernels queue up behind earlier work of the stream the engine was given and not
    behind work of a stream it was not given.  Decided by stream ORDER, not by wall-clock thresholds: with a long spin
    ahead of the emit on the engine's stream, a copy of the output enqueued on the other stream right after the emit call
    still sees the zeros the buffer was filled with (the spin is measured with events and must dwarf that copy); with the
    spin on the stream the engine does not use, the same copy — now behind the emit in its own stream — sees the reads."""
    import torch
    from simmr_amd.engine import Reads
    eng = engine
    eng.stage_synthetic(7, [5_000_000], 9)
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod()
    side, main = torch.cuda.Stream(), torch.cuda.default_stream()
    info = eng.pe_plan(7, prof, 2_000_000, 3)
    out = Reads.allocate(info.n_reads, info.total_bases, eng.device, qual_offset=33)
    spin = 400_000_000  # device clock cycles
    peek = torch.empty(4096, dtype=torch.uint8).pin_memory()

    def emit_with_a_spin_on_side():
        """zero the output, spin on `side`, emit on the engine's stream, copy the head of seq[] through `main`;
        returns (what that copy saw, the spin's duration in ms)"""
        out.seq.zero_()
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            s0.record(side)
            torch.cuda._sleep(spin)
            s1.record(side)
        eng.pe_emit(0, out)  # on the engine's stream
        with torch.cuda.stream(main):
            peek.copy_(out.seq[:4096], non_blocking=True)
        main.synchronize()
        seen = peek.numpy().copy()
        torch.cuda.synchronize()
        return seen, s0.elapsed_time(s1)

    try:
        with torch.cuda.stream(side):
            eng.use_current_torch_stream()
        eng.pe_plan(7, prof, 2_000_000, 3)
        seen, spin_ms = emit_with_a_spin_on_side()
        if spin_ms > 20.0:  # (else the spin is no cover for a 4 KB copy on this box: nothing to conclude)
            assert not seen.any(), "the emit kernels did not wait for the earlier work of the engine's stream"
        first = out.to_host()
        assert first["seq"][:4096].all()                 # ... and ran after it
        with torch.cuda.stream(main):
            eng.use_current_torch_stream()
        eng.pe_plan(7, prof, 2_000_000, 3)
        seen, _ = emit_with_a_spin_on_side()
        assert np.array_equal(seen, first["seq"][:4096])  # behind the emit in its own stream, whatever `side` is doing
        again = out.to_host()
        for col in ("seq", "qual", "seq_off", "read_id"):
            assert np.array_equal(first[col], again[col]), col
    finally:
        with torch.cuda.stream(torch.cuda.default_stream()):
            eng.use_current_torch_stream()


# ---- the output contract: seq / qual hold exactly total_bases bytes -------------------------------------------
def _exact_reads(engine, info, qual_offset):
    """simmr_reads_out whose seq / qual are EXACTLY total_bases long, cut out of larger canary-filled buffers."""
    import torch
    from simmr_amd.engine import Reads
    PAD = 256
    tb = int(info.total_bases)
    r = Reads.allocate(info.n_reads, tb, engine.device, qual_offset)
    bufs = []
    for name in ("seq", "qual"):
        buf = torch.full((tb + 2 * PAD,), 0xA5, dtype=torch.uint8, device=engine.device)
        setattr(r, name, buf[PAD:PAD + tb])
        bufs.append(buf)
    assert r.pod().seq_capacity == tb
    return r, bufs, PAD, tb


def _canaries_intact(bufs, PAD, tb):
    for b in bufs:
        assert bool((b[:PAD] == 0xA5).all()) and bool((b[PAD + tb:] == 0xA5).all())


@pytest.mark.parametrize("path", ["perfect-short", "minimal-short", "philox-short", "custom-short",
                                  "minimal-long", "philox-long", "perfect-long", "custom-long"])
def test_emit_stays_inside_exact_capacity(engine, genome_multi, path, monkeypatch):
    """include/simmr_hip.h promises that seq and qual need total_bases bytes, not a byte more: every emit kernel is run
    with buffers of exactly that size between canaries (lengths that are no multiple of 16, so the last chunk is a
    partial one), and must reproduce the padded run."""
    import torch
    from simmr_amd import (CustomShortErrorProfile, PerfectLongErrorProfile, PerfectShortErrorProfile, model_io)
    from simmr_amd.engine import Engine
    eng = engine
    keep = None
    long_mode = path.endswith("long")
    if path == "perfect-short":
        pod = PerfectShortErrorProfile(read_length=37, insert_size=45).pod()
    elif path == "minimal-short":
        pod = MinimalShortErrorProfile(read_length=41, insert_size=60).pod()
    elif path == "philox-short":
        pod = MinimalShortErrorProfile(read_length=41, insert_size=60, rng_mode=_abi.RNG_PHILOX).pod()
    elif path == "custom-short":
        keep = CustomShortErrorProfile(model_io.synthetic_short_model(n_positions=40, seed=5, mean_len=45, sd_len=7,
                                                                      mean_insert=70, sd_insert=10))
        pod = keep.pod()
    elif path == "minimal-long":
        pod = MinimalLongErrorProfile(gamma_mean=900.0, gamma_std=700.0, length_mode=_abi.LEN_PER_READ).pod()
    elif path == "philox-long":
        pod = MinimalLongErrorProfile(gamma_mean=900.0, gamma_std=700.0, length_mode=_abi.LEN_PER_READ,
                                      rng_mode=_abi.RNG_PHILOX).pod()
    elif path == "perfect-long":
        pod = PerfectLongErrorProfile(gamma_mean=900.0, gamma_std=700.0, length_mode=_abi.LEN_PER_READ).pod()
    else:
        keep = CustomShortErrorProfile(model_io.synthetic_long_model(kmer_size=5, n_positions=60, seed=2, n_kmers=600,
                                                                     with_n=False, read_length_mean=700.0, read_length_std=200.0))
        pod = keep.pod()
        pod.length_mode = _abi.LEN_PER_READ
        pod.long_start_mode = _abi.START_UNIFORM
    try:
        if long_mode:
            info = eng.long_plan([1], [333], pod, 7)
            padded = eng.simulate_long_reads([1], [333], pod, 7, qual_offset=33).to_host()
            info = eng.long_plan([1], [333], pod, 7)
        else:
            info = eng.pe_plan(1, pod, 2601, 7)
            padded = eng.simulate_pe_reads_from_genome(1, pod, 2601, 7, qual_offset=33).to_host()
            info = eng.pe_plan(1, pod, 2601, 7)
        r, bufs, PAD, tb = _exact_reads(eng, info, 33)
        (eng.long_emit if long_mode else eng.pe_emit)(0, r)
        torch.cuda.synchronize()
        _canaries_intact(bufs, PAD, tb)
        exact = r.to_host()
        for col in ("seq", "qual", "seq_off", "start", "end", "contig", "read_id", "flags"):
            assert np.array_equal(exact[col], padded[col]), col
    finally:
        if eng is not engine:
            eng.close()


@pytest.mark.parametrize("variant", ["ablate_lines", "ablate_all16"])
def test_store_moving_ablation_builds_stay_inside_the_streams(variant):
    """The two measurement builds of k_emit_philox that MOVE stores (make -C simmr_amd/csrc ablate: whole-line stores,
    partial groups as 16 bytes) write wrong bytes by design, but never outside the total_bases bytes of seq / qual: run
    once between canaries of exactly that size, in a process of its own (the library is chosen per process).  Skipped
    when the variant libraries are not built (they are not part of the product)."""
    import subprocess
    import sys
    lib = ROOT / "simmr_amd" / "csrc" / "variants" / f"libsimmr_hip_{variant}.so"
    if not lib.exists():
        pytest.skip(f"{lib.name} is not built (make -C simmr_amd/csrc ablate)")
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from simmr_amd import MinimalShortErrorProfile, _abi
from simmr_amd.engine import Engine, Reads
eng = Engine(0)
eng.stage_synthetic(0, [400_000], 5)
for reads, L in ((2601, 41), (20001, 150), (9, 150), (3, 20)):
    pod = MinimalShortErrorProfile(read_length=L, insert_size=60, rng_mode=_abi.RNG_PHILOX).pod()
    info = eng.pe_plan(0, pod, reads, 7)
    PAD, tb = 4096, int(info.total_bases)
    r = Reads.allocate(info.n_reads, tb, eng.device, 33)
    bufs = []
    for name in ("seq", "qual"):
        buf = torch.full((tb + 2 * PAD,), 0xA5, dtype=torch.uint8, device=eng.device)
        setattr(r, name, buf[PAD:PAD + tb])
        bufs.append(buf)
    eng.pe_emit(0, r)
    torch.cuda.synchronize()
    for b in bufs:
        assert bool((b[:PAD] == 0xA5).all()) and bool((b[PAD + tb:] == 0xA5).all()), (reads, L)
print("inside")
''' % str(ROOT)
    import os
    env = dict(os.environ, SIMMR_HIP_LIB=str(lib))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "inside" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("knobs", [{"SIMMR_PHILOX_WGS_PER_CU": "1", "SIMMR_GRID_MULT": "1"},
                                   {"SIMMR_PHILOX_WGS_PER_CU": "7", "SIMMR_GRID_MULT": "3"},
                                   {"SIMMR_PHILOX_WGS_PER_CU": "4096", "SIMMR_GRID_MULT": "512"}])
def test_results_do_not_depend_on_the_grid(engine, monkeypatch, knobs):
    """The emit kernels are grid-stride loops whose launch size is a tuning knob ("the grid"), and the
    run counters are summed from per-workgroup rows: reads, metadata and counters must be the same for one workgroup per
    CU, for an odd number, and for one block per workgroup — both generators, both layouts of the counter mode,
    perfect-short, long reads, and the FASTQ text straight from the plan."""
    from simmr_amd import MinimalLongErrorProfile, PerfectShortErrorProfile
    from simmr_amd.engine import Engine
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)  # (read once, at engine creation)
    other = Engine(0)
    try:
        names = [(21, "g21", ["c%d" % i for i in range(3)])]
        for eng in (engine, other):
            eng.stage_synthetic(21, [400_000, 70_001, 150_000], 4)
        runs = [("philox", MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod(), 0),
                ("philox slot16", MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX).pod(), 16),
                ("reference", MinimalShortErrorProfile().pod(), 0),
                ("perfect", PerfectShortErrorProfile().pod(), 0)]
        for what, prof, slot in runs:
            got = []
            for eng in (engine, other):
                eng.set_read_slots(slot)
                try:
                    eng.counters_reset()
                    r = eng.simulate_pe_reads_from_genome(21, prof, 60_001, 5, first=7, count=29_000, read_id_base=3, qual_offset=33)
                    c = eng.counters()
                    eng.pe_plan(21, prof, 60_001, 5, 7, 29_000)
                    text = eng.fastq_direct("@{:read_id:}/{:pair:} {:sequence_id:} {:start_position:}", names, 3).cpu().numpy().tobytes() if slot == 0 else b""
                finally:
                    eng.set_read_slots(0)
                got.append((r.raw_to_host() if slot else r.to_host(), c, text))
            (a, ca, ta), (b, cb, tb) = got
            assert np.array_equal(ca, cb), what
            assert ta == tb, what
            if slot:
                for k in a:
                    assert np.array_equal(a[k], b[k]), (what, k)
            else:
                from tests.test_gpu_parity import assert_same
                assert_same(a, b)
        lp = MinimalLongErrorProfile(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX).pod()
        got = []
        for eng in (engine, other):
            eng.counters_reset()
            r = eng.simulate_long_reads([21], [900], lp, 9, qual_offset=33)
            got.append((r.to_host(), eng.counters()))
        from tests.test_gpu_parity import assert_same
        assert_same(got[0][0], got[1][0])
        assert np.array_equal(got[0][1], got[1][1])
    finally:
        other.close()


def test_plan_overlap_gives_the_same_reads(oracle):
    """simmr_engine_set_plan_overlap: plan calls on the engine's own stream into a second set of plan buffers, so that plan
    k + 1 runs while emit k is still on the device.  Shards of one run are planned and emitted back to back without a
    synchronisation in between — paired-end in all three rng modes, several genomes in one plan, long reads, a custom
    long-read model — and every shard must be what the same calls give without the overlap (= the oracle's)."""
    import torch
    from simmr_amd import CustomShortErrorProfile, MinimalLongErrorProfile
    from simmr_amd.engine import Engine, Reads
    from tests import _model
    eng = Engine(0)
    try:
        lens = [60_000, 45_000, 52_000]
        eng.stage_synthetic(0, lens, 5)
        eng.stage_synthetic(1, [70_000], 6)
        g0 = _oracle.HostGenome(_synth.synthetic_contigs(lens, 5))
        g1 = _oracle.HostGenome(_synth.synthetic_contigs([70_000], 6))
        eng.set_plan_overlap(True)
        reads, seed = 40_000, 77
        for mode in (_abi.RNG_PHILOX_FULL, _abi.RNG_PHILOX, _abi.RNG_REFERENCE):
            prof = MinimalShortErrorProfile(rng_mode=mode).pod()
            whole = _oracle.simulate_pe(oracle, g0, prof, reads, seed, qual_offset=33).trimmed()
            outs = []
            for k in range(8):  # eight shards of 2500 pairs, nothing waited for between them
                info = eng.pe_plan(0, prof, reads, seed, 2500 * k, 2500)
                out = Reads.allocate(info.n_reads, info.total_bases, eng.device, qual_offset=33, slot_bytes=info.slot_bytes)
                out.total_bases = int(info.total_bases)
                eng.pe_emit(0, out)  # (ids: pair 0 of the genome is id 0; the plan knows its first pair)
                outs.append(out)
            torch.cuda.synchronize()
            got = [o.to_host() for o in outs]
            assert np.array_equal(np.concatenate([d["seq"] for d in got]), whole["seq"]), mode
            assert np.array_equal(np.concatenate([d["qual"] for d in got]), whole["qual"]), mode
            for col in ("start", "end", "contig", "read_id", "flags"):
                assert np.array_equal(np.concatenate([d[col] for d in got]), whole[col]), (mode, col)
        # several genomes in one plan, alternating with single-genome plans of another profile
        prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL).pod()
        prof2 = MinimalShortErrorProfile(read_length=80, insert_size=120, mean_phred_score=20, rng_mode=_abi.RNG_PHILOX_FULL).pod()
        a = eng.simulate_pe_reads_multi([0, 1], [9000, 7000], prof, 5, qual_offset=33)
        b = eng.simulate_pe_reads_from_genome(1, prof2, 6000, 9, qual_offset=33)
        c = eng.simulate_pe_reads_multi([1, 0], [3000, 5000], prof, 6, qual_offset=33)
        torch.cuda.synchronize()
        ob = _oracle.simulate_pe(oracle, g1, prof2, 6000, 9, qual_offset=33).trimmed()
        assert np.array_equal(b.to_host()["seq"], ob["seq"]) and np.array_equal(b.to_host()["qual"], ob["qual"])
        for dev, order, rd, sd in ((a, (g0, g1), (9000, 7000), 5), (c, (g1, g0), (3000, 5000), 6)):
            parts, base = [], 0
            for g, n in zip(order, rd):
                parts.append(_oracle.simulate_pe(oracle, g, prof, n, sd, read_id_base=base, qual_offset=33).trimmed())
                base += n // 2
            d = dev.to_host()
            assert np.array_equal(d["seq"], np.concatenate([p["seq"] for p in parts]))
            assert np.array_equal(d["qual"], np.concatenate([p["qual"] for p in parts]))
        # long reads, and a custom long-read model (its emit kernels use the error word of their buffer set)
        lp = MinimalLongErrorProfile(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ, rng_mode=_abi.RNG_PHILOX_FULL).pod()
        cp = CustomShortErrorProfile(_model.synthetic_long_model(kmer_size=6, n_positions=100, seed=4, n_kmers=4 ** 6, lengths=(800, 3000, 100)),
                                     _abi.RNG_PHILOX).pod()
        runs = [eng.simulate_long_reads([0, 1], [150, 90], p, 3, first=f, count=60, qual_offset=33) for p in (lp, cp) for f in (0, 60, 120, 180)]
        torch.cuda.synchronize()
        i = 0
        for p in (lp, cp):
            for f in (0, 60, 120, 180):
                o = _oracle.simulate_long(oracle, [g0, g1], [150, 90], p, 3, first=f, count=60, qual_offset=33).trimmed()
                d = runs[i].to_host(); i += 1
                assert np.array_equal(d["seq"], o["seq"]) and np.array_equal(d["qual"], o["qual"]), (f,)
        # plan -> fastq_plan_direct -> emit_fastq back to back (the FASTQ sizing buffers exist once; every user
        # of them synchronises before it returns, engine.hip: "INVARIANT the overlap rests on")
        from tests.test_gpu_cli import FMT
        names = [(0, "genome-zero", ["c0", "c1 with words", "c2"])]
        texts = []
        for k in range(6):
            eng.pe_plan(0, prof, reads, seed, 2500 * k, 2500)
            texts.append(eng.fastq_direct(FMT, names, 7))
        torch.cuda.synchronize()
        eng.set_plan_overlap(False)
        for k in range(6):
            eng.pe_plan(0, prof, reads, seed, 2500 * k, 2500)
            want = eng.fastq_direct(FMT, names, 7).cpu().numpy().tobytes()
            assert texts[k].cpu().numpy().tobytes() == want, k
        d = eng.simulate_pe_reads_from_genome(1, prof2, 6000, 9, qual_offset=33).to_host()
        assert np.array_equal(d["seq"], ob["seq"])
    finally:
        eng.close()
