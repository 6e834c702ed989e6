"""Randomized parity sweep: a few hundred random (genome shape, profile, parameters, seed, shard) combinations,
device against the oracle, every column bit for bit.  Complements the hand-picked cases of test_gpu_parity.py."""
import numpy as np
import pytest

from simmr_amd import MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectShortErrorProfile, SimmrError, _abi
from tests import _oracle, _synth
from tests.test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


import os

# SIMMR_SWEEP_SEEDS=1,2,3: other sweeps for a soak run
SWEEP_SEEDS = [int(x) for x in os.environ.get("SIMMR_SWEEP_SEEDS", "7,11,2024").split(",")]


@pytest.mark.parametrize("sweep_seed", SWEEP_SEEDS)
def test_random_configurations_match_the_oracle(engine, oracle, sweep_seed):
    rng = np.random.default_rng(sweep_seed)
    e, lib = engine, oracle
    n_ok = 0
    SLOT = 30
    for it in range(220):
        nc = int(rng.integers(1, 6))
        lens = [int(rng.integers(700, 60_000)) for _ in range(nc)]
        contigs = _synth.synthetic_contigs(lens, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.4:
            c = contigs[0].copy(); k = rng.integers(0, c.size, c.size // 20); c[k] = ord("N"); c[rng.integers(0, c.size, 30)] = ord("-"); contigs[0] = c
        e.stage_genome(SLOT, contigs); host = _oracle.HostGenome(contigs)
        seed = int(rng.integers(0, 1 << 62)); qoff = int(rng.choice([0, 33]))
        kind = rng.integers(0, 4)
        if kind <= 1:
            L = int(rng.integers(1, 260)); I = int(rng.integers(0, 400)); mq = int(rng.integers(0, 70))
            if kind == 0:
                prof = PerfectShortErrorProfile(L, I).pod()
            else:
                prof = MinimalShortErrorProfile(read_length=L, insert_size=I, mean_phred_score=mq, rng_mode=int(rng.integers(0, 3))).pod()  # reference / philox / philox-full
            reads = int(rng.integers(0, 1500)); first = int(rng.integers(0, reads // 2 + 2)); count = int(rng.integers(0, 800))
            if 2 * L + I >= min(lens):
                continue
            run_dev = lambda: e.simulate_pe_reads_from_genome(SLOT, prof, reads, seed, first=first, count=count, read_id_base=3, qual_offset=qoff)
            run_ora = lambda: _oracle.simulate_pe(lib, host, prof, reads, seed, first=first, count=count, read_id_base=3, qual_offset=qoff, max_len=70000)
        else:
            gm = float(rng.integers(300, 4000)); gs = gm * float(rng.uniform(0.3, 0.9))
            cls = MinimalLongErrorProfile
            rm = int(rng.integers(0, 3))  # (philox-full with the reference's one constant length: refused on both sides)
            prof = cls(gamma_mean=gm, gamma_std=gs, length_mode=int(rng.integers(0, 2)), rng_mode=rm, uniform_start=bool(rng.integers(0, 2)), mean_phred_score=int(rng.integers(0, 60))).pod()
            if kind == 3:
                prof.kind = _abi.PERFECT_LONG
            if min(lens) <= 20000 and max(lens) <= 20000:
                continue
            reads = int(rng.integers(0, 60)); first = int(rng.integers(0, reads + 1)); count = int(rng.integers(0, 40))
            run_dev = lambda: e.simulate_long_reads([SLOT], [reads], prof, seed, first=first, count=count, read_id_base=1, qual_offset=qoff)
            run_ora = lambda: _oracle.simulate_long(lib, [host], [reads], prof, seed, first=first, count=count, read_id_base=1, qual_offset=qoff)
        # both sides refuse some configurations (a contig too small for the drawn length, ...): the refusal must be mutual
        dev = ora = None
        try:
            ora = run_ora()
        except RuntimeError as ex:
            ora_err = str(ex)
        try:
            dev = run_dev()
        except SimmrError as ex:
            dev_err = str(ex)
        assert (dev is None) == (ora is None), (f"it{it} kind{kind}: device " + ("refused: " + dev_err if dev is None else "ran") +
                                                 ", oracle " + ("refused: " + ora_err if ora is None else "ran"))
        if dev is None:
            continue
        d, o = dev.to_host(), ora.trimmed()
        assert_same(d, o, what=f"it{it} kind{kind} ")
        n_ok += 1
    assert n_ok > 150


def test_random_multi_genome_plans(engine, oracle):
    """simmr_pe_plan_multi over random genome lists, read counts, profiles and shards against the reference's loop
    over genomes restated with the oracle (simulate.rs:121-150: same seed per genome, one global id counter)."""
    rng = np.random.default_rng(99)
    hosts = {}
    for slot, lens in ((40, [30_000]), (41, [9_000, 25_000, 14_000]), (42, [50_000]), (43, [8_000, 8_500])):
        contigs = _synth.synthetic_contigs(lens, 100 + slot)
        if slot == 42:
            contigs[0] = contigs[0].copy()
            contigs[0][rng.integers(0, 50_000, 2500)] = ord("N")
        engine.stage_genome(slot, contigs)
        hosts[slot] = _oracle.HostGenome(contigs)
    n_ok = 0
    for it in range(60):
        order = [int(x) for x in rng.choice([40, 41, 42, 43], size=int(rng.integers(1, 7)))]
        reads = [int(rng.integers(0, 900)) for _ in order]
        L, I = int(rng.integers(5, 200)), int(rng.integers(0, 300))
        prof = [PerfectShortErrorProfile(L, I), MinimalShortErrorProfile(read_length=L, insert_size=I),
                MinimalShortErrorProfile(read_length=L, insert_size=I, rng_mode=_abi.RNG_PHILOX),
                MinimalShortErrorProfile(read_length=L, insert_size=I, rng_mode=_abi.RNG_PHILOX_FULL)][int(rng.integers(0, 4))].pod()
        seed = int(rng.integers(0, 1 << 60))
        parts, base = [], 0
        for gi, n in zip(order, reads):
            o = _oracle.simulate_pe(oracle, hosts[gi], prof, n, seed, read_id_base=base, qual_offset=33, max_len=70000).trimmed()
            o["genome"] = np.full(o["read_id"].size, gi, np.uint32)
            parts.append(o)
            base += n // 2
        lens = np.concatenate([np.diff(p["seq_off"].astype(np.int64)) for p in parts])
        whole = {c: np.concatenate([p[c] for p in parts]) for c in ("seq", "qual", "start", "end", "contig", "genome", "read_id", "flags")}
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        first, count = int(rng.integers(0, base + 2)), int(rng.integers(0, base + 2))
        d = engine.simulate_pe_reads_multi(order, reads, prof, seed, first=first, count=count, qual_offset=33).to_host()
        a, b = 2 * min(first, base), 2 * min(first + count, base)
        assert np.array_equal(d["seq_off"], off[a:b + 1] - off[a]), it
        for col in ("start", "end", "contig", "genome", "read_id", "flags"):
            assert np.array_equal(d[col], whole[col][a:b]), (it, col)
        assert np.array_equal(d["seq"], whole["seq"][int(off[a]):int(off[b])]), it
        assert np.array_equal(d["qual"], whole["qual"][int(off[a]):int(off[b])]), it
        n_ok += 1
    assert n_ok == 60


@pytest.mark.parametrize("rng_mode", [0, 1], ids=["reference", "philox"])
def test_random_custom_long_models(engine, oracle, rng_mode):
    """Random long-read models (k-mer size, sparsity of the k-mer table, modelled positions, length law) on random
    genomes with N / '-' runs: qualities and the k-mer splice against the oracle; a refusal must be mutual."""
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    rng = np.random.default_rng(99 + (SWEEP_SEEDS[0] if SWEEP_SEEDS != [7, 11, 2024] else 0))
    SLOT = 31
    n_ok = n_refused = 0
    for it in range(40):
        k = int(rng.integers(1, 9))
        n_kmers = int(min(4 ** k, rng.choice([4, 60, 1000, 20000])))
        lo = int(rng.integers(50, 800)); hi = lo + int(rng.integers(200, 3000))
        blob = _model.synthetic_long_model(kmer_size=k, n_positions=int(rng.integers(1, 1500)), seed=int(rng.integers(0, 1 << 30)),
                                           n_kmers=n_kmers, lengths=(lo, hi, 50), deletion=(it % 8 == 3))
        prof = CustomShortErrorProfile(blob, rng_mode)  # 1 = SIMMR_RNG_PHILOX: the splice's draws from Philox counters
        pod = prof.pod()
        nc = int(rng.integers(1, 4))
        lens = [int(rng.integers(hi + 500, 30_000)) for _ in range(nc)]
        contigs = _synth.synthetic_contigs(lens, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.5:
            c = contigs[0].copy()
            c[rng.integers(0, c.size, c.size // 15)] = ord("N")
            c[rng.integers(0, c.size, 20)] = ord("-")
            s = int(rng.integers(0, c.size - 40)); c[s:s + 30] = ord("N")
            contigs[0] = c
        engine.stage_genome(SLOT, contigs)
        host = _oracle.HostGenome(contigs)
        seed = int(rng.integers(0, 1 << 62)); qoff = int(rng.choice([0, 33]))
        reads = int(rng.integers(1, 200)); first = int(rng.integers(0, reads)); count = int(rng.integers(1, 150))
        dev = ora = None
        try:
            ora = _oracle.simulate_long(oracle, [host], [reads], pod, seed, first=first, count=count, read_id_base=5, qual_offset=qoff)
        except RuntimeError:
            pass
        try:
            dev = engine.simulate_long_reads([SLOT], [reads], pod, seed, first=first, count=count, read_id_base=5, qual_offset=qoff)
        except SimmrError:
            pass
        assert (dev is None) == (ora is None), f"it{it}: device {'refused' if dev is None else 'ran'}, oracle {'refused' if ora is None else 'ran'}"
        if dev is None:
            n_refused += 1
            continue
        o = ora.trimmed(); o["genome"][:] = SLOT
        assert_same(dev.to_host(), o, cols=("seq_off", "start", "end", "contig", "read_id", "flags", "qual", "seq", "genome"), what=f"it{it} k{k} ")
        n_ok += 1
    assert n_ok >= 25 and n_refused >= 3
