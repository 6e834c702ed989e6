"""Thin Python host over the C ABI (include/simmr_hip.h).

PyTorch is used only for device memory, streams and torch.distributed; every
byte of simulated read content is produced by the HIP kernels in
simmr_amd/csrc through libsimmr_hip.so.  No fallback path exists.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._abi import (ErrorProfilePOD, PlanInfo, Range, ReadsOut, SimmrError, U64_MAX)


def _torch():
    import torch  # deferred: importing the package must not need a GPU
    return torch


@dataclass
class Reads:
    """SoA replacement of Vec<SimulatedRead> (simulate.rs:27-75) in HBM.

    PE shards interleave mates: read r is mate (r & 1) of pair (r >> 1)."""
    seq: "object"
    qual: "object"
    seq_off: "object"
    start: "object"
    end: "object"
    contig: "object"
    genome: "object"
    read_id: "object"
    flags: "object"
    n_reads: int
    total_bases: int
    qual_offset: int = 0
    slot_bytes: int = 0  # 0: compact streams; 16: SIMMR_SLOT16 (include/simmr_hip.h, simmr_reads_out)

    @classmethod
    def allocate(cls, n_reads: int, total_bases: int, device, qual_offset: int = 0, slot_bytes: int = 0) -> "Reads":
        torch = _torch()
        n = max(int(n_reads), 1)
        # the library needs exactly total_bases bytes (include/simmr_hip.h); the slack only keeps torch views of
        # whole 16-byte rows possible for the tests' checksums
        nb = (int(total_bases) + 15) // 16 * 16 + 16
        return cls(
            seq=torch.empty(nb, dtype=torch.uint8, device=device),
            qual=torch.empty(nb, dtype=torch.uint8, device=device),
            seq_off=torch.empty(n + 1, dtype=torch.int64, device=device),
            start=torch.empty(n, dtype=torch.int64, device=device),
            end=torch.empty(n, dtype=torch.int64, device=device),
            contig=torch.empty(n, dtype=torch.int32, device=device),
            genome=torch.empty(n, dtype=torch.int32, device=device),
            read_id=torch.empty(n, dtype=torch.int32, device=device),
            flags=torch.empty(n, dtype=torch.uint8, device=device),
            n_reads=int(n_reads), total_bases=int(total_bases), qual_offset=int(qual_offset),
            slot_bytes=int(slot_bytes))

    def pod(self) -> ReadsOut:
        o = ReadsOut()
        o.seq = self.seq.data_ptr()
        o.qual = self.qual.data_ptr()
        o.seq_off = self.seq_off.data_ptr()
        o.start = self.start.data_ptr()
        o.end = self.end.data_ptr()
        o.contig = self.contig.data_ptr()
        o.genome = self.genome.data_ptr()
        o.read_id = self.read_id.data_ptr()
        o.flags = self.flags.data_ptr()
        o.seq_capacity = self.seq.numel()
        o.reads_capacity = self.start.numel()
        o.qual_offset = self.qual_offset
        o.slot_bytes = self.slot_bytes
        return o

    def to_host(self) -> dict:
        """numpy copies, trimmed to the planned sizes.  Reads emitted into 16-byte slots (SIMMR_SLOT16) come back in
        the compact form — seq / qual without the padding, CSR seq_off — so that a consumer sees one layout; the
        columns as they lie in HBM are `raw_to_host()`."""
        h = self.raw_to_host()
        if self.slot_bytes != 16:
            return h
        n = self.n_reads
        a, b = h["start"].astype(np.int64), h["end"].astype(np.int64)
        L = np.abs(b - a)                                   # a read's length in this layout
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(L, out=off[1:])
        first = h["seq_off"][:n].astype(np.int64)           # first base; the qualities start at first & ~15
        within = np.arange(int(off[n]), dtype=np.int64) - np.repeat(off[:n], L)
        h["seq"] = h["seq"][np.repeat(first, L) + within]
        h["qual"] = h["qual"][np.repeat(first & ~np.int64(15), L) + within]
        h["seq_off"] = off.astype(np.uint64)
        return h

    def raw_to_host(self) -> dict:
        n, tb = self.n_reads, self.total_bases
        return {
            "seq": self.seq[:tb].cpu().numpy(),
            "qual": self.qual[:tb].cpu().numpy(),
            "seq_off": self.seq_off[: n + 1].cpu().numpy().astype(np.uint64),
            "start": self.start[:n].cpu().numpy().astype(np.uint64),
            "end": self.end[:n].cpu().numpy().astype(np.uint64),
            "contig": self.contig[:n].cpu().numpy().astype(np.uint32),
            "genome": self.genome[:n].cpu().numpy().astype(np.uint32),
            "read_id": self.read_id[:n].cpu().numpy().astype(np.uint32),
            "flags": self.flags[:n].cpu().numpy(),
        }


class Engine:
    """One engine == one GPU == one host thread (include/simmr_hip.h)."""

    def __init__(self, device: int = 0):
        self.lib = _abi.load()
        h = C.c_void_p()
        rc = self.lib.simmr_engine_create(int(device), C.byref(h))
        if rc != 0:
            raise SimmrError(rc, (self.lib.simmr_last_error(None) or b"").decode())
        self._h = h
        self.device_index = int(device)
        self._keep = []

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.simmr_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise SimmrError(rc, (self.lib.simmr_last_error(self._h) or b"").decode())

    @property
    def device(self):
        return _torch().device("cuda", self.device_index)

    def use_current_torch_stream(self):
        s = _torch().cuda.current_stream(self.device)
        self._check(self.lib.simmr_engine_set_stream(self._h, C.c_void_p(s.cuda_stream)))

    def set_read_slots(self, slot_bytes: int):
        """Layout of the reads that plans made from now on emit: 0 compact, 16 = SIMMR_SLOT16 (simmr_engine_set_read_slots).
        The setting is the ENGINE's and it sticks: simmr_amd.simulate's entry points select SIMMR_SLOT16 and leave it
        selected, so direct pe_plan / long_plan calls made on the same engine afterwards also get the slot layout, in which
        seq_off[r + 1] - seq_off[r] is not read r's length (PlanInfo.slot_bytes / Reads.slot_bytes say which layout a plan
        emits; the C ABI's own default is compact)."""
        self._check(self.lib.simmr_engine_set_read_slots(self._h, int(slot_bytes)))

    def set_plan_overlap(self, on: bool):
        """The plan of the next shard beside the emit of this one (simmr_engine_set_plan_overlap): plan calls on a stream
        of the engine's own, into a second set of the plan buffers."""
        self._check(self.lib.simmr_engine_set_plan_overlap(self._h, 1 if on else 0))

    # -- staging ------------------------------------------------------------
    def stage_genome(self, genome_idx: int, contigs: Sequence, sizes: Optional[Sequence[int]] = None):
        """contigs: normalised ASCII sequences (bytes or uint8 arrays), Seq.seq
        of genome.rs:17-23; sizes: Seq.size when it differs (--contiguous)."""
        arrs = [np.ascontiguousarray(np.frombuffer(c, dtype=np.uint8) if isinstance(c, (bytes, bytearray))
                                     else np.asarray(c, dtype=np.uint8)) for c in contigs]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_uint64 * n)(*[a.size for a in arrs])
        szs = (C.c_uint64 * n)(*[int(s) for s in sizes]) if sizes is not None else None
        self._check(self.lib.simmr_stage_genome(self._h, genome_idx, n, ptrs, lens, szs))

    def stage_fasta(self, genome_idx: int, bodies: Sequence[bytes], contiguous: bool = False, min_size: int = 0):
        """Raw FASTA record bodies (the bytes between a header line and the next header) -> staged genome;
        normalisation (needletail normalize(false), genome.rs:114) and packing happen on the device.
        Returns (bases per record, number of sequences staged)."""
        arrs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in bodies]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_uint64 * n)(*[a.size for a in arrs])
        counts = (C.c_uint64 * n)()
        staged = C.c_uint32(0)
        self._check(self.lib.simmr_stage_fasta(self._h, genome_idx, n, ptrs, lens, 1 if contiguous else 0, int(min_size),
                                               counts, C.byref(staged)))
        return [int(x) for x in counts], int(staged.value)

    def stage_synthetic(self, genome_idx: int, contig_lens: Sequence[int], splitmix_seed: int):
        n = len(contig_lens)
        lens = (C.c_uint64 * n)(*[int(x) for x in contig_lens])
        self._check(self.lib.simmr_stage_synthetic(self._h, genome_idx, n, lens, C.c_uint64(splitmix_seed)))

    def unstage(self, genome_idx: int, contig: int, first: int, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.uint8)
        self._check(self.lib.simmr_unstage_contig(self._h, genome_idx, contig, first, count,
                                                  C.c_void_p(out.ctypes.data)))
        return out

    def genome_info(self, genome_idx: int):
        n = C.c_uint32()
        t = C.c_uint64()
        self._check(self.lib.simmr_genome_info(self._h, genome_idx, C.byref(n), C.byref(t)))
        return n.value, t.value

    # -- paired-end (simulate.rs:165-302) -------------------------------------
    def pe_plan(self, genome_idx: int, profile: ErrorProfilePOD, genome_reads: int,
                seed: Optional[int], first: int = 0, count: int = U64_MAX,
                start: Tuple[int, int] = (0, 0)) -> PlanInfo:
        """`start` = (slot, pair): a known position of the genome's outer stream at or before
        `first` (see simulate.seek_outer_stream); (0, 0) walks the stream from its beginning."""
        info = PlanInfo()
        if start == (0, 0):
            self._check(self.lib.simmr_pe_plan(self._h, genome_idx, C.byref(profile), genome_reads,
                                               0 if seed is None else 1, 0 if seed is None else seed,
                                               Range(first, count), C.byref(info)))
        else:
            if seed is None:
                raise ValueError("seeking needs a seed")
            self._check(self.lib.simmr_pe_plan_at(self._h, genome_idx, C.byref(profile), genome_reads, seed,
                                                  Range(first, count), start[0], start[1], C.byref(info)))
        return info

    def outer_summarize(self, genome_idx: int, seed: int, slot_first: int, slot_count: int):
        """(units0, units1, end0, end1) of slots [slot_first, slot_first + slot_count) of the genome's
        outer stream (simulate.rs:172-184), see include/simmr_hip.h."""
        s = _abi.OuterSummary()
        self._check(self.lib.simmr_outer_summarize(self._h, genome_idx, seed, slot_first, slot_count, C.byref(s)))
        return int(s.units[0]), int(s.units[1]), int(s.end_state[0]), int(s.end_state[1])

    def pe_plan_multi(self, genome_idx: Sequence[int], genome_reads: Sequence[int], profile: ErrorProfilePOD,
                      seed: Optional[int], first: int = 0, count: int = U64_MAX) -> PlanInfo:
        """simulate_pe_reads over several genomes in one plan; [first, first + count) is a range of the
        global pair index (genomes concatenated in order)."""
        n = len(genome_idx)
        gi = (C.c_uint32 * n)(*[int(x) for x in genome_idx])
        gr = (C.c_uint64 * n)(*[int(x) for x in genome_reads])
        info = PlanInfo()
        self._check(self.lib.simmr_pe_plan_multi(self._h, n, gi, gr, C.byref(profile), 0 if seed is None else 1,
                                                 0 if seed is None else seed, Range(first, count), C.byref(info)))
        return info

    def simulate_pe_reads_multi(self, genome_idx, genome_reads, profile, seed, first=0, count=U64_MAX,
                                qual_offset=0) -> Reads:
        info = self.pe_plan_multi(genome_idx, genome_reads, profile, seed, first, count)
        out = Reads.allocate(info.n_reads, info.total_bases, self.device, qual_offset, info.slot_bytes)
        self.pe_emit(0, out)
        return out

    def pe_emit(self, read_id_base: int, out: Reads):
        pod = out.pod()
        self._check(self.lib.simmr_pe_emit(self._h, read_id_base, C.byref(pod)))

    def simulate_pe_reads_from_genome(self, genome_idx: int, profile: ErrorProfilePOD, genome_reads: int,
                                      seed: Optional[int], first: int = 0, count: int = U64_MAX,
                                      read_id_base: int = 0, qual_offset: int = 0,
                                      start: Tuple[int, int] = (0, 0)) -> Reads:
        info = self.pe_plan(genome_idx, profile, genome_reads, seed, first, count, start)
        out = Reads.allocate(info.n_reads, info.total_bases, self.device, qual_offset, info.slot_bytes)
        self.pe_emit(read_id_base, out)
        return out

    # -- long reads (simulate.rs:323-523) --------------------------------------
    def long_plan(self, genome_idx: Sequence[int], genome_reads: Sequence[int], profile: ErrorProfilePOD,
                  seed: Optional[int], first: int = 0, count: int = U64_MAX) -> PlanInfo:
        n = len(genome_idx)
        gi = (C.c_uint32 * n)(*[int(x) for x in genome_idx])
        gr = (C.c_uint64 * n)(*[int(x) for x in genome_reads])
        info = PlanInfo()
        self._check(self.lib.simmr_long_plan(self._h, n, gi, gr, C.byref(profile),
                                             0 if seed is None else 1, 0 if seed is None else seed,
                                             Range(first, count), C.byref(info)))
        return info

    def long_emit(self, read_id_base: int, out: Reads):
        pod = out.pod()
        self._check(self.lib.simmr_long_emit(self._h, read_id_base, C.byref(pod)))

    def simulate_long_reads(self, genome_idx, genome_reads, profile, seed, first=0, count=U64_MAX,
                            read_id_base=0, qual_offset=0) -> Reads:
        info = self.long_plan(genome_idx, genome_reads, profile, seed, first, count)
        out = Reads.allocate(info.n_reads, info.total_bases, self.device, qual_offset, info.slot_bytes)
        self.long_emit(read_id_base, out)
        return out

    # -- FASTQ framing (fastq.rs:14-124) ------------------------------------------
    def fastq(self, reads: Reads, header_format: str, names, paired: bool):
        """FASTQ text of `reads` as a CUDA uint8 tensor.  `names` = [(engine genome slot, genome id,
        [sequence id per contig]), ...]; qualities must have been emitted with qual_offset=33.
        Raises SimmrError(ENOTSUP) for the cases the library leaves to the host writer."""
        torch = _torch()
        n = len(names)
        gi = (C.c_uint32 * max(n, 1))(*[int(x[0]) for x in names])
        gid = (C.c_char_p * max(n, 1))(*[str(x[1]).encode() for x in names])
        nc = (C.c_uint32 * max(n, 1))(*[len(x[2]) for x in names])
        flat = [str(sid).encode() for x in names for sid in x[2]]
        sids = (C.c_char_p * max(len(flat), 1))(*flat)
        fn = _abi.FastqNames(n, gi, gid, nc, sids)
        pod = reads.pod()
        total = C.c_uint64(0)
        self._check(self.lib.simmr_fastq_plan(self._h, header_format.encode(), C.byref(fn), C.byref(pod),
                                              reads.n_reads, 1 if paired else 0, C.byref(total)))
        out = torch.empty(max(total.value, 1), dtype=torch.uint8, device=self.device)
        self._check(self.lib.simmr_fastq_emit(self._h, C.byref(pod), C.c_void_p(out.data_ptr()), total.value))
        return out[: total.value]

    def _fastq_names(self, names):
        n = len(names)
        gi = (C.c_uint32 * max(n, 1))(*[int(x[0]) for x in names])
        gid = (C.c_char_p * max(n, 1))(*[str(x[1]).encode() for x in names])
        nc = (C.c_uint32 * max(n, 1))(*[len(x[2]) for x in names])
        flat = [str(sid).encode() for x in names for sid in x[2]]
        sids = (C.c_char_p * max(len(flat), 1))(*flat)
        fn = _abi.FastqNames(n, gi, gid, nc, sids)
        fn._keep = (gi, gid, nc, sids)
        return fn

    def fastq_plan_direct(self, header_format: str, names, read_id_base: int = 0) -> int:
        """Sizes the FASTQ text of the CURRENT plan's shard (simmr_fastq_plan_direct); returns its bytes."""
        fn = self._fastq_names(names)
        total = C.c_uint64(0)
        self._check(self.lib.simmr_fastq_plan_direct(self._h, header_format.encode(), C.byref(fn), int(read_id_base),
                                                     C.byref(total)))
        return int(total.value)

    def emit_fastq(self, out):
        """Writes the text planned by fastq_plan_direct into the CUDA uint8 tensor `out` (simmr_emit_fastq)."""
        self._check(self.lib.simmr_emit_fastq(self._h, C.c_void_p(out.data_ptr()), int(out.numel())))

    def fastq_direct(self, header_format: str, names, read_id_base: int = 0):
        """FASTQ text of the current plan's shard without the columns in between, as a CUDA uint8 tensor."""
        torch = _torch()
        total = self.fastq_plan_direct(header_format, names, read_id_base)
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=self.device)
        self.emit_fastq(out)
        return out[:total]

    def last_fastq_plan_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.simmr_last_fastq_plan_ms(self._h, C.byref(ms)))
        return ms.value

    # -- counters / timing --------------------------------------------------------
    def counters(self) -> np.ndarray:
        host = (C.c_uint64 * _abi.N_COUNTERS)()
        self._check(self.lib.simmr_counters(self._h, None, host))
        return np.array(list(host), dtype=np.uint64)

    def counters_to(self, tensor):
        """Copy the counters into a CUDA int64 tensor (for one all-reduce)."""
        self._check(self.lib.simmr_counters(self._h, C.c_void_p(tensor.data_ptr()), None))

    def counters_reset(self):
        self._check(self.lib.simmr_counters_reset(self._h))

    # -- the counters across GPUs without torch.distributed: RCCL through the C ABI ------
    def comm_unique_id(self) -> bytes:
        """rank 0: an id to hand to the other ranks (ncclGetUniqueId)"""
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        rc = self.lib.simmr_comm_unique_id(buf)
        if rc != 0:
            raise SimmrError(rc, "simmr_comm_unique_id failed (is librccl loadable?)")
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(bytes(comm_id), _abi.COMM_ID_BYTES)
        self._check(self.lib.simmr_comm_init(self._h, buf, int(rank), int(world)))

    def allreduce_counts(self, tensor):
        """in-place sum over the ranks of a CUDA int64 tensor (a no-op without a communicator)"""
        self._check(self.lib.simmr_allreduce_counts(self._h, C.c_void_p(tensor.data_ptr()), int(tensor.numel())))

    def last_emit_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.simmr_last_emit_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def emit_kernel_ms_mean(self, last_n: int) -> float:
        """Mean HIP-event time of the last `last_n` emits (one synchronisation; simmr_emit_kernel_ms_mean)."""
        ms = C.c_float()
        self._check(self.lib.simmr_emit_kernel_ms_mean(self._h, int(last_n), C.byref(ms)))
        return ms.value

    def last_plan_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.simmr_last_plan_ms(self._h, C.byref(ms)))
        return ms.value
