"""Host-side mirror of simmr/src/simulate.rs's two entry points, with the
read-index sharding that lets N GPUs split one run.

    simulate_pe_reads    simulate.rs:110-150
    simulate_long_reads  simulate.rs:323-406

Same argument order and meaning as the reference (num_reads, genomes,
error_profile, abundance_profile, seed) plus (rank, world).  `backend` is an
`Engine` (one per GPU); every read byte comes from the HIP kernels behind the C
ABI.  The only collective of the path is the all-reduce of the run counters.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from ._abi import U64_MAX
from .profiles import AbundanceProfile, ErrorProfile


@dataclass
class GenomeRef:
    """What the host needs to know about one staged genome (genome.rs:26-41)."""
    index: int          # staged genome index on the engine
    size: int           # Genome.size (sum of Seq.size)
    filepath: str = ""
    uuid: str = ""


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


def simulate_pe_reads(backend, num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                      abundance_profile: AbundanceProfile, seed: Optional[int], rank: int = 0, world: int = 1,
                      qual_offset: int = 0):
    """Returns one tuple per genome, like simulate.rs:119:
    (filepath, uuid, genome_reads, abundance, reads-of-this-rank or None)."""
    ab = determine_reads(num_reads, genomes, error_profile, abundance_profile, True)
    shards = pe_shards([r for r, _ in ab], rank, world)
    pod = error_profile.pod()
    out = []
    for g, (reads, abund), (first, count, id_base) in zip(genomes, ab, shards):
        res = None
        if count > 0:
            # the SAME seed for every genome (simulate.rs:137,172)
            res = backend.simulate_pe_reads_from_genome(g.index, pod, reads, seed, first=first, count=count,
                                                        read_id_base=id_base, qual_offset=qual_offset)
        out.append((g.filepath, g.uuid, reads, abund, res))
    return out


def simulate_long_reads(backend, num_reads: int, genomes: Sequence[GenomeRef], error_profile: ErrorProfile,
                        abundance_profile: AbundanceProfile, seed: Optional[int], rank: int = 0, world: int = 1,
                        qual_offset: int = 0):
    """One StdRng stream spans all genomes (simulate.rs:348): the shard is a
    range of global read indices.  Returns (per-genome metadata, reads)."""
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
