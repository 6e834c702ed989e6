"""Host-side mirror of simmr/src/simulate.rs's two entry points, with the
read-index sharding that lets N GPUs split one run.

    simulate_pe_reads    simulate.rs:110-150
    simulate_long_reads  simulate.rs:323-406

Same argument order and meaning as the reference (num_reads, genomes,
error_profile, abundance_profile, seed) plus (rank, world).  `backend` is an
`Engine` (one per GPU); every read byte comes from the HIP kernels behind the C
ABI.  The only collective of the path is the all-reduce of the run counters.

Output layout: the entry points ask the engine for 16-byte read slots
(`read_slots=16`, SIMMR_SLOT16 of include/simmr_hip.h) — the layout the
counter-mode emit kernel writes 20 % faster than gap-free streams — and get
them wherever the plan's emit kernel offers them, else the compact layout; the
returned `Reads` knows which (`slot_bytes`), and `Reads.to_host()` hands every
consumer the same compact form.  `read_slots=0` asks for compact streams.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import _abi
from ._abi import U64_MAX
from .profiles import AbundanceProfile, ErrorProfile


@dataclass
class GenomeRef:
    """What the host needs to know about one staged genome (genome.rs:26-41)."""
    index: int          # staged genome index on the engine
    size: int           # Genome.size (sum of Seq.size)
    filepath: str = ""
    uuid: str = ""
    n_contigs: int = 1  # Genome.num_seqs (the range of the contig draw, simulate.rs:181)


def _prefer_layout(backend, read_slots: int) -> None:
    """simmr_engine_set_read_slots: the layout of the plans the entry points below are about to make.  The setting stays on
    the engine after they return (Engine.set_read_slots says what that means for direct plan calls)."""
    if hasattr(backend, "set_read_slots"):
        backend.set_read_slots(read_slots)


def split_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [first, first+count) share of `total` units for `rank`."""
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi - lo


def determine_reads(num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                    abundance_profile: AbundanceProfile, paired: bool) -> List[Tuple[int, float]]:
    """simulate.rs:121-132 / :334-343."""
    ab = abundance_profile.determine_abundances(num_reads, len(genomes))
    if abundance_profile.is_size_aware():
        # PE: error_profile.get_read_length(seed) (unused by the arithmetic); long: 20_000
        ab = abundance_profile.adjust_for_size([g.size for g in genomes], ab,
                                               150 if paired else 20_000, paired)
    return ab


def pe_shards(genome_reads: Sequence[int], rank: int, world: int):
    """Per-genome (first_pair, n_pairs, read_id_base) of this rank's share of the
    global pair-index space (genomes concatenated in order; ids as the
    reference's global AtomicU32 hands them out, simulate.rs:85-89)."""
    pairs = [r // 2 for r in genome_reads]  # simulate.rs:179
    total = sum(pairs)
    lo, n = split_range(total, rank, world)
    hi = lo + n
    out = []
    base = 0
    for p in pairs:
        a, b = max(lo, base), min(hi, base + p)
        out.append((a - base, b - a, base) if b > a else (0, 0, base))
        base += p
    return out


def outer_slot_floor(n_contigs: int, unit: int) -> int:
    """A block-aligned slot that lies at or before the slot where pair `unit` of a genome's outer
    stream starts (simulate.rs:172-184: per pair, a contig index by rejection sampling —
    gen_range(0..n_contigs), 1 / p_acc slots on average — and one slot for pe_seed).  Pure
    arithmetic, identical on every rank; 2 % + 4096 slots below the expectation, which is more
    than 20 standard deviations of the rejection count for any unit."""
    if unit <= 0:
        return 0
    zone = ((n_contigs << (64 - n_contigs.bit_length())) - 1) & ((1 << 64) - 1)
    p_acc = (zone + 1) / 2.0 ** 64
    est = 0.98 * unit * (1.0 / p_acc + 1.0) - 4096.0
    return max(0, int(est) // 8 * 8)


def seek_outer_stream(backend, pieces, want, seed: int):
    """Positions of the outer streams at the start of this rank's shard, without walking them
    from slot 0 on every rank.

    `pieces[j]` = (genome index, n_contigs, a, b) or None: the part [a, b) of a genome's pairs that
    rank j covers and that ends before the genome does (only then a later rank starts inside the
    same genome).  Rank j summarizes the slots [floor(a), floor(b)) of that genome's stream once
    (`backend.outer_summarize`: pairs completed and end state for either entry state), the ranks
    exchange the four numbers (all_gather), and each rank composes the summaries of the pieces in
    front of its own start.  `want` = (genome index, n_contigs, first pair) of this rank.
    Returns (slot, pair): pair `pair` (<= first pair) starts at slot `slot`."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    # RCCL moves device tensors; gloo (CPU tests, single-GPU rehearsal) host tensors
    dev = torch.device("cuda", torch.cuda.current_device()) if world > 1 and dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(4, dtype=torch.int64)
    if pieces[rank] is not None:
        g, nc, a, b = pieces[rank]
        lo, hi = outer_slot_floor(nc, a), outer_slot_floor(nc, b)
        mine = torch.tensor(backend.outer_summarize(g, seed, lo, hi - lo), dtype=torch.int64)
    if world > 1:
        mine = mine.to(dev)
        gathered = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(gathered, mine)
        gathered = [t.cpu() for t in gathered]
    else:
        gathered = [mine]
    return compose_outer_summaries(pieces, [tuple(int(x) for x in t) for t in gathered], want)


def compose_outer_summaries(pieces, summaries, want):
    """The pure part of seek_outer_stream: `summaries[j]` = (units0, units1, end0, end1) of piece j."""
    g, nc, first = want
    if first == 0:
        return 0, 0
    # the pieces of genome g in front of `first`, in order: [0, b1) [b1, b2) ... [.., first)
    chain = sorted((p[2], p[3], j) for j, p in enumerate(pieces) if p is not None and p[0] == g and p[3] <= first)
    units, state = 0, 0
    best = (0, 0)
    at = 0
    for a, b, j in chain:
        if a != at:
            break  # a gap: what has been composed so far still stands
        u0, u1, e0, e1 = summaries[j]
        units += u1 if state else u0
        state = e1 if state else e0
        at = b
        # in state 1 the pending pair gets its pe_seed from the next slot: the pair after it starts one slot later
        if units + state <= first:
            best = (outer_slot_floor(nc, b) + state, units + state)
    return best


def simulate_pe_reads(backend, num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                      abundance_profile: AbundanceProfile, seed: Optional[int], rank: int = 0, world: int = 1,
                      qual_offset: int = 0, read_slots: int = 16):
    """Returns one tuple per genome, like simulate.rs:119:
    (filepath, uuid, genome_reads, abundance, reads-of-this-rank or None)."""
    _prefer_layout(backend, read_slots)
    ab = determine_reads(num_reads, genomes, error_profile, abundance_profile, True)
    reads_per_genome = [r for r, _ in ab]
    shards = pe_shards(reads_per_genome, rank, world)
    pod = error_profile.pod()
    # Only the first genome of a rank's range can start in the middle of that genome's outer stream.
    start = {}
    # (RNG_PHILOX_FULL: a pair's outer draws are a function of its index — there is no stream to seek in)
    if world > 1 and seed is not None and hasattr(backend, "outer_summarize") and pod.rng_mode != _abi.RNG_PHILOX_FULL:
        pieces = []
        for j in range(world):
            mid = [(gi, sh) for gi, sh in enumerate(pe_shards(reads_per_genome, j, world))
                   if sh[1] > 0 and sh[0] + sh[1] < reads_per_genome[gi] // 2]
            gi, sh = mid[-1] if mid else (None, None)
            pieces.append(None if gi is None else (genomes[gi].index, genomes[gi].n_contigs, sh[0], sh[0] + sh[1]))
        firsts = [(gi, sh) for gi, sh in enumerate(shards) if sh[1] > 0]
        if firsts:
            gi, sh = firsts[0]
            want = (genomes[gi].index, genomes[gi].n_contigs, sh[0])
        else:
            gi, want = None, (0, 1, 0)
        pos = seek_outer_stream(backend, pieces, want, seed)  # collective: every rank calls it
        if gi is not None:
            start[gi] = pos
    out = []
    for gi, (g, (reads, abund), (first, count, id_base)) in enumerate(zip(genomes, ab, shards)):
        res = None
        if count > 0:
            # the SAME seed for every genome (simulate.rs:137,172)
            kw = {"start": start[gi]} if start.get(gi, (0, 0)) != (0, 0) else {}
            res = backend.simulate_pe_reads_from_genome(g.index, pod, reads, seed, first=first, count=count,
                                                        read_id_base=id_base, qual_offset=qual_offset, **kw)
        out.append((g.filepath, g.uuid, reads, abund, res))
    return out


def simulate_pe_reads_batched(backend, num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                              abundance_profile: AbundanceProfile, seed: Optional[int], rank: int = 0,
                              world: int = 1, qual_offset: int = 0, read_slots: int = 16):
    """simulate_pe_reads (simulate.rs:110-150) with all genomes in ONE device plan
    (`simmr_pe_plan_multi`): same reads, ids and order as the per-genome loop, returned like
    simulate_long_reads as (per-genome metadata, reads of this rank's range of the global pair
    index).  For runs over many genomes (BASELINE config 4) this removes the per-genome launch
    and synchronisation cost."""
    _prefer_layout(backend, read_slots)
    ab = determine_reads(num_reads, genomes, error_profile, abundance_profile, True)
    reads_per_genome = [r for r, _ in ab]
    first, count = split_range(sum(r // 2 for r in reads_per_genome), rank, world)
    reads = backend.simulate_pe_reads_multi([g.index for g in genomes], reads_per_genome, error_profile.pod(), seed,
                                            first=first, count=count, qual_offset=qual_offset)
    meta = [(g.filepath, g.uuid, r, a) for g, (r, a) in zip(genomes, ab)]
    return meta, reads


def simulate_long_reads(backend, num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                        abundance_profile: AbundanceProfile, seed: Optional[int], rank: int = 0, world: int = 1,
                        qual_offset: int = 0, read_slots: int = 16):
    """One StdRng stream spans all genomes (simulate.rs:348): the shard is a
    range of global read indices.  Returns (per-genome metadata, reads)."""
    _prefer_layout(backend, read_slots)
    ab = determine_reads(num_reads, genomes, error_profile, abundance_profile, False)
    total = sum(r for r, _ in ab)
    first, count = split_range(total, rank, world)
    reads = backend.simulate_long_reads([g.index for g in genomes], [r for r, _ in ab], error_profile.pod(),
                                        seed, first=first, count=count, read_id_base=0, qual_offset=qual_offset)
    meta = [(g.filepath, g.uuid, r, a) for g, (r, a) in zip(genomes, ab)]
    return meta, reads


def all_reduce_counters(counters_tensor):
    """The path's single collective: sum of the run counters over all ranks
    (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters_tensor)
    return counters_tensor
