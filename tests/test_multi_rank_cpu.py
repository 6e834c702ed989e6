"""N > 1 path on CPU (gloo, world_size 2): the host sharding logic of
simmr_amd.simulate driven with an oracle-backed stand-in for the GPU engine
(test infrastructure), checked against the single-rank whole run, plus the one
collective of the path (all-reduce of the run counters)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simmr_amd import (ExactAbundanceProfile, MinimalLongErrorProfile, MinimalShortErrorProfile,
                       PerfectShortErrorProfile, UniformAbundanceProfile)
from simmr_amd.simulate import (GenomeRef, all_reduce_counters, pe_shards, simulate_long_reads,
                                simulate_pe_reads, split_range)
from tests import _oracle, _synth

GENOMES = [([40_000, 30_000], 3), ([25_000], 4), ([60_000, 22_000, 21_000], 5)]


class OracleBackend:
    """Engine look-alike that computes shards with the CPU oracle (tests only)."""

    def __init__(self):
        self.lib = _oracle.load()
        self.genomes = [_oracle.HostGenome(_synth.synthetic_contigs(lens, seed)) for lens, seed in GENOMES]

    def simulate_pe_reads_from_genome(self, idx, pod, reads, seed, first=0, count=(1 << 64) - 1, read_id_base=0,
                                      qual_offset=0, start=(0, 0)):
        if start != (0, 0):  # the oracle always walks from slot 0: only check the claimed position
            self.seeks = getattr(self, "seeks", 0) + 1
            acc = outer_accept_bits(self.lib, len(self.genomes[idx].contigs), seed, start[0])
            assert replay_outer(acc, 0, start[0], 0) == (start[1], 0) and start[1] <= first
        return _oracle.simulate_pe(self.lib, self.genomes[idx], pod, reads, seed, first, count, read_id_base,
                                   qual_offset=qual_offset)

    def outer_summarize(self, idx, seed, slot_first, slot_count):
        acc = outer_accept_bits(self.lib, len(self.genomes[idx].contigs), seed, slot_first + slot_count)
        (u0, e0), (u1, e1) = (replay_outer(acc, slot_first, slot_first + slot_count, s) for s in (0, 1))
        return u0, u1, e0, e1

    def simulate_long_reads(self, idxs, reads, pod, seed, first=0, count=(1 << 64) - 1, read_id_base=0,
                            qual_offset=0):
        return _oracle.simulate_long(self.lib, [self.genomes[i] for i in idxs], reads, pod, seed, first, count,
                                     read_id_base, qual_offset=qual_offset)


def outer_accept_bits(lib, n_contigs, seed, n_slots):
    """Per u64 slot of StdRng(seed): would gen_range(0..n_contigs) accept it (rand 0.8.5 sample_single)?
    Drawn with the oracle's generator (tests only)."""
    import ctypes as C
    from tests._oracle import Rng
    r = Rng()
    lib.orc_rng_seed_from_u64(C.byref(r), seed)
    zone = ((n_contigs << (64 - n_contigs.bit_length())) - 1) & ((1 << 64) - 1)
    acc = np.zeros(n_slots, np.uint8)
    for i in range(n_slots):
        v = lib.orc_next_u64(C.byref(r))
        acc[i] = ((v * n_contigs) & ((1 << 64) - 1)) <= zone
    return acc


def replay_outer(acc, lo, hi, state):
    """simulate.rs:172-184 over slots [lo, hi) entered in `state` (0 = contig draw, 1 = pe_seed draw):
    (pairs completed, state after)."""
    units = 0
    for i in range(lo, hi):
        if state == 0:
            state = 1 if acc[i] else 0
        else:
            state, units = 0, units + 1
    return units, state


def test_outer_stream_seek_composition():
    """Host side of the multi-GPU seek: ranks summarize disjoint slot ranges, the summaries compose to a
    position (slot, pair) at or before every rank's first pair, and that position is right."""
    from simmr_amd.simulate import compose_outer_summaries, outer_slot_floor
    lib = _oracle.load()
    for n_contigs, seed, per_rank, world in ((1, 42, 9000, 4), (3, 7, 12000, 3), (2, 5, 7000, 5)):
        total = per_rank * world
        n_slots = outer_slot_floor(n_contigs, total) + 8
        acc = outer_accept_bits(lib, n_contigs, seed, n_slots)
        pieces = [(0, n_contigs, j * per_rank, (j + 1) * per_rank) if j + 1 < world else None for j in range(world)]
        summaries = []
        for p in pieces:
            if p is None:
                summaries.append((0, 0, 0, 1))
                continue
            lo, hi = outer_slot_floor(n_contigs, p[2]), outer_slot_floor(n_contigs, p[3])
            assert lo % 8 == 0 and hi % 8 == 0 and lo <= hi
            (u0, e0), (u1, e1) = replay_outer(acc, lo, hi, 0), replay_outer(acc, lo, hi, 1)
            summaries.append((u0, u1, e0, e1))
        for r in range(world):
            first = r * per_rank
            slot, unit = compose_outer_summaries(pieces, summaries, (0, n_contigs, first))
            assert unit <= first and (r == 0) == ((slot, unit) == (0, 0))
            assert replay_outer(acc, 0, slot, 0) == (unit, 0)  # pair `unit` starts at `slot`
            if r > 0:
                assert first - unit < 0.03 * first + 3000  # and it is close to the shard
        # a piece missing from the chain: composition stops there, the position stays valid
        holes = list(pieces)
        holes[1] = None
        slot, unit = compose_outer_summaries(holes, summaries, (0, n_contigs, (world - 1) * per_rank))
        assert replay_outer(acc, 0, slot, 0) == (unit, 0) and unit <= per_rank


def _refs():
    return [GenomeRef(i, sum(l), f"g{i}.fna", f"id{i}", len(l)) for i, (l, _) in enumerate(GENOMES)]


def test_split_and_shards_cover_everything():
    for total in (0, 1, 7, 1000, 10 ** 9 + 7):
        for world in (1, 2, 3, 8):
            parts = [split_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    reads = [10, 7, 0, 25]  # pairs 5, 3, 0, 12
    seen = [0] * 4
    for r in range(3):
        for g, (first, count, base) in enumerate(pe_shards(reads, r, 3)):
            assert base == [0, 5, 8, 8][g]
            seen[g] += count
    assert seen == [5, 3, 0, 12]


PE_READS = 90_000  # 15 000 pairs per genome: outer_slot_floor > 0, so rank 1 starts from a composed position != (0, 0)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = OracleBackend()
    prof = MinimalShortErrorProfile()
    res = simulate_pe_reads(be, PE_READS, _refs(), prof, UniformAbundanceProfile(), 42, rank, world)
    payload = []
    counters = torch.zeros(4, dtype=torch.int64)
    for (path, uuid, reads, abund, r) in res:
        if r is None:
            payload.append(None)
            continue
        d = r.trimmed()
        payload.append({k: v.copy() for k, v in d.items()})
        counters[0] += r.n_reads
        counters[1] += r.total_bases
    meta, lr = simulate_long_reads(be, 40, _refs(), MinimalLongErrorProfile(), ExactAbundanceProfile(), 9, rank, world)
    counters[2] += lr.n_reads
    counters[3] += lr.total_bases
    all_reduce_counters(counters)
    q.put((rank, payload, {k: v.copy() for k, v in lr.trimmed().items()}, counters.tolist(), getattr(be, "seeks", 0)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    be = OracleBackend()
    whole = simulate_pe_reads(be, PE_READS, _refs(), MinimalShortErrorProfile(), UniformAbundanceProfile(), 42, 0, 1)
    tot_reads = tot_bases = 0
    for g, (path, uuid, reads, abund, r) in enumerate(whole):
        assert reads == PE_READS // 3 and abund == 100.0 / 3
        w = r.trimmed()
        tot_reads += r.n_reads
        tot_bases += r.total_bases
        parts = [got[rk][1][g] for rk in range(world) if got[rk][1][g] is not None]
        for col in ("start", "end", "contig", "read_id", "flags", "seq", "qual"):
            assert np.array_equal(np.concatenate([p[col] for p in parts]), w[col]), (g, col)
        lens = np.concatenate([np.diff(p["seq_off"].astype(np.int64)) for p in parts])
        assert np.array_equal(lens, np.diff(w["seq_off"].astype(np.int64)))
    # read ids run across genomes in generation order (simulate.rs:85-89)
    ids = np.concatenate([r.trimmed()["read_id"] for *_, r in whole])
    assert np.array_equal(ids, np.repeat(np.arange(PE_READS // 2, dtype=np.uint32), 2))
    # long reads: one stream across genomes, shard = global read-index range
    meta, lw = simulate_long_reads(be, 40, _refs(), MinimalLongErrorProfile(), ExactAbundanceProfile(), 9, 0, 1)
    w = lw.trimmed()
    assert lw.n_reads == 120
    for col in ("start", "end", "contig", "genome", "read_id", "seq", "qual"):
        assert np.array_equal(np.concatenate([got[rk][2][col] for rk in range(world)]), w[col]), col
    # the all-reduced counters equal the whole-run totals on every rank
    for rk in range(world):
        assert got[rk][3] == [tot_reads, tot_bases, lw.n_reads, lw.total_bases]
    # rank 1 starts inside genome 1: it planned from a composed position other than (0, 0)
    assert got[0][4] == 0 and got[1][4] == 1


def test_full_counter_mode_needs_no_seek():
    """SIMMR_RNG_PHILOX_FULL: a pair's outer draws are a function of its index, so a rank that starts in the middle of a
    genome plans its shard without the summaries / all-gather of the reference's stream (simulate.py skips the seek: no
    process group exists here, a collective would raise) — and the shards of 1, 2, 3 and 5 ranks are the whole run."""
    from simmr_amd import _abi
    be = OracleBackend()
    prof = MinimalShortErrorProfile(rng_mode=_abi.RNG_PHILOX_FULL)
    whole = simulate_pe_reads(be, PE_READS, _refs(), prof, UniformAbundanceProfile(), 42, 0, 1)
    for world in (2, 3, 5):
        per_rank = [simulate_pe_reads(be, PE_READS, _refs(), prof, UniformAbundanceProfile(), 42, rk, world) for rk in range(world)]
        for g, (*_, r) in enumerate(whole):
            w = r.trimmed()
            parts = [per_rank[rk][g][4].trimmed() for rk in range(world) if per_rank[rk][g][4] is not None]
            for col in ("start", "end", "contig", "read_id", "flags", "seq", "qual"):
                assert np.array_equal(np.concatenate([p[col] for p in parts]), w[col]), (world, g, col)
    assert getattr(be, "seeks", 0) == 0


# ---- eight ranks, a stream a billion slots long ---------------------------------------------------------------
class PeriodicStream:
    """An outer stream whose accept bits repeat with period P (a stand-in for StdRng: nothing can replay 1e9 real
    slots in a CPU test).  The loop of simulate.rs:172-184 over a range of slots is a two-state transducer, so the
    summary of any range composes from the summaries of whole periods and two partial ones — in O(P + log n)."""

    # 6 pairs per 14 slots = 7/3 slots per pair, what gen_range(0..3) (3 of 4 slots accepted) takes on average
    PATTERN = (0, 0, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1)

    @staticmethod
    def step(bits, state):
        units = 0
        for a in bits:
            if state == 0:
                state = 1 if a else 0
            else:
                state, units = 0, units + 1
        return units, state

    def summary(self, lo, hi):
        """(units0, units1, end0, end1) of slots [lo, hi)."""
        P = len(self.PATTERN)
        out = []
        for s0 in (0, 1):
            units, state, i = 0, s0, lo
            while i < hi and i % P:  # to the next period boundary
                u, state = self.step((self.PATTERN[i % P],), state)
                units, i = units + u, i + 1
            n_per = (hi - i) // P
            if n_per > 0:
                # after one period from either state; the state sequence over periods becomes periodic at once
                per = {st: self.step(self.PATTERN, st) for st in (0, 1)}
                seen, k = {}, 0
                while k < n_per and state not in seen:
                    seen[state] = (k, units)
                    u, state = per[state]
                    units, k = units + u, k + 1
                if k < n_per:  # a cycle of period states: skip whole cycles
                    k0, u0 = seen[state]
                    clen, cunits = k - k0, units - u0
                    reps = (n_per - k) // clen
                    units, k = units + reps * cunits, k + reps * clen
                    while k < n_per:
                        u, state = per[state]
                        units, k = units + u, k + 1
                i += n_per * P
            while i < hi:
                u, state = self.step((self.PATTERN[i % P],), state)
                units, i = units + u, i + 1
            out.append((units, state))
        return out[0][0], out[1][0], out[0][1], out[1][1]

    def outer_summarize(self, idx, seed, slot_first, slot_count):
        return self.summary(slot_first, slot_first + slot_count)


def _worker8(rank, world, port, q, per_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simmr_amd.simulate import seek_outer_stream
    be = PeriodicStream()
    pieces = [(0, 3, j * per_rank, (j + 1) * per_rank) if j + 1 < world else None for j in range(world)]
    slot, unit = seek_outer_stream(be, pieces, (0, 3, rank * per_rank), 42)
    q.put((rank, slot, unit))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_ranks_seek_a_billion_slots_in():
    """bench.py's N = 8 control flow for the outer stream (every rank summarizes its own slots, one all-gather of four
    numbers, composition on the host) at BASELINE config 4's size per rank: 62.5 M pairs per rank, so the last rank
    starts 437.5 M pairs = a billion slots into the stream.  Every rank's position must be exact and close."""
    world, per_rank = 8, 62_500_000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, per_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = PeriodicStream()
    for rank, slot, unit in got:
        first = rank * per_rank
        assert unit <= first
        u0, _, e0, _ = ref.summary(0, slot)
        assert (u0, e0) == (unit, 0), rank          # pair `unit` starts exactly at `slot`
        if rank == 0:
            assert (slot, unit) == (0, 0)
        else:
            assert first - unit < 0.03 * first + 3000, (rank, first, unit)
    assert got[-1][1] > 1_000_000_000               # the last rank really is a billion slots in
