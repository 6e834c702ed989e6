"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from simmr_amd._abi import ErrorProfilePOD, ReadsOut

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "liboracle.so"


class Rng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64), ("results", C.c_uint32 * 64),
                ("index", C.c_uint32), ("words_used", C.c_uint64), ("ctr", C.c_uint32)]


class Genome(C.Structure):
    _fields_ = [("n_contigs", C.c_uint32), ("seq", C.POINTER(C.c_void_p)),
                ("len", C.POINTER(C.c_uint64)), ("size", C.POINTER(C.c_uint64))]


class PePlan(C.Structure):
    _fields_ = [("read_length", C.c_uint32), ("insert_size", C.c_uint32), ("fwd_start", C.c_uint64),
                ("fwd_end", C.c_uint64), ("rev_end", C.c_uint64), ("rev_start", C.c_uint64),
                ("qseed2", C.c_uint64), ("mseed2", C.c_uint64), ("flags2", C.c_uint8)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle")])


_rr = None


def load_rocrand_pin():
    """oracle/librocrand_pin.so: rocRAND's own Philox4x32-10 engine class run on the host (oracle/rocrand_pin.hip)."""
    global _rr
    if _rr is None:
        so = ROOT / "oracle" / "librocrand_pin.so"
        if not so.exists():
            build()
        _rr = C.CDLL(str(so))
        _rr.rr_philox_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        _rr.rr_philox_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    return _rr


def load():
    global _lib
    if _lib is not None:
        return _lib
    # SIMMR_ORACLE_LIB: another build of the same sources (`make -C oracle asan` for sanitizer runs)
    import os
    alt = os.environ.get("SIMMR_ORACLE_LIB")
    if not alt and not LIB.exists():
        build()
    lib = C.CDLL(alt or str(LIB))
    P = C.POINTER
    lib.orc_next_u64.restype = C.c_uint64
    lib.orc_next_u32.restype = C.c_uint32
    lib.orc_gen_f32.restype = C.c_float
    lib.orc_gen_f64.restype = C.c_double
    lib.orc_open01_f64.restype = C.c_double
    lib.orc_open01_f32.restype = C.c_float
    lib.orc_standard_normal.restype = C.c_double
    lib.orc_normal_f64.restype = C.c_double
    lib.orc_normal_f64.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.orc_normal_f32.restype = C.c_float
    lib.orc_normal_f32.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.orc_gamma_f32.argtypes = [C.c_void_p, C.c_float, C.c_float, P(C.c_float)]
    lib.orc_rng_seed_from_u64.argtypes = [C.c_void_p, C.c_uint64]
    lib.orc_gen_range_u64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, P(C.c_uint64)]
    lib.orc_gen_range_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, P(C.c_uint32)]
    lib.orc_chacha_block.argtypes = [P(C.c_uint32), C.c_uint64, C.c_uint32, P(C.c_uint32)]
    lib.orc_zig_norm_x.restype = P(C.c_double)
    lib.orc_zig_norm_f.restype = P(C.c_double)
    lib.orc_complement.restype = C.c_uint8
    lib.orc_complement.argtypes = [C.c_uint8]
    lib.orc_encode_quality_score.restype = C.c_uint8
    lib.orc_encode_quality_score.argtypes = [C.c_uint8]
    for f in ("orc_convert_phred_to_probability", "orc_convert_phred_to_accuracy"):
        getattr(lib, f).restype = C.c_float
        getattr(lib, f).argtypes = [C.c_uint8]
    for f in ("orc_convert_probability_to_phred", "orc_convert_accuracy_to_phred"):
        getattr(lib, f).restype = C.c_uint8
        getattr(lib, f).argtypes = [C.c_float]
    lib.orc_entropy_substitute.restype = C.c_uint64
    lib.orc_entropy_substitute.argtypes = [C.c_uint64, C.c_uint32]
    lib.orc_per_read_seed.restype = C.c_uint64
    lib.orc_per_read_seed.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_pe_outer.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                 P(C.c_uint64)]
    lib.orc_pe_plan_pair.argtypes = [P(ErrorProfilePOD), C.c_uint64, C.c_uint64, P(PePlan)]
    lib.orc_simulate_pe_reads_from_genome.argtypes = [
        P(Genome), P(ErrorProfilePOD), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
        P(ReadsOut), P(C.c_uint64), C.c_int]
    lib.orc_simulate_long_reads.argtypes = [
        P(Genome), C.c_uint32, P(C.c_uint64), P(ErrorProfilePOD), C.c_int, C.c_uint64, C.c_uint64,
        C.c_uint64, C.c_uint32, P(ReadsOut), P(C.c_uint64), P(C.c_uint32), C.c_int]
    lib.orc_set_faithful_cost.argtypes = [C.c_int]
    lib.orc_set_faithful_cost.restype = None
    lib.orc_profile_simulate_phred_scores.argtypes = [P(ErrorProfilePOD), C.c_uint64, C.c_uint64, C.c_void_p]
    lib.orc_profile_simulate_point_mutations.argtypes = [P(ErrorProfilePOD), C.c_void_p, C.c_void_p,
                                                         C.c_uint64, C.c_uint64, C.c_void_p]
    for f in ("orc_profile_get_read_length", "orc_profile_get_random_read_length",
              "orc_profile_get_insert_size"):
        getattr(lib, f).argtypes = [P(ErrorProfilePOD), C.c_uint64, P(C.c_uint16)]
    lib.orc_profile_minimum_genome_size.argtypes = [P(ErrorProfilePOD), P(C.c_uint16)]
    lib.orc_uniform_determine_abundances.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.orc_exact_determine_abundances.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.orc_custom_determine_abundances.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.orc_adjust_for_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.orc_two_bit_encode_kmer.argtypes = [C.c_char_p, C.c_uint32, P(C.c_uint32)]
    lib.orc_two_bit_decode_kmer.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p]
    _lib = lib
    return lib


class HostGenome:
    """keeps the numpy contigs alive while C holds pointers to them"""

    def __init__(self, contigs, sizes=None):
        self.contigs = [np.ascontiguousarray(np.asarray(c, dtype=np.uint8)) for c in contigs]
        n = len(self.contigs)
        self._ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in self.contigs])
        self._lens = (C.c_uint64 * n)(*[c.size for c in self.contigs])
        self._sizes = (C.c_uint64 * n)(*([int(s) for s in sizes] if sizes is not None else [c.size for c in self.contigs]))
        self.c = Genome(n, C.cast(self._ptrs, C.POINTER(C.c_void_p)), self._lens, self._sizes)


class HostReads:
    """host SoA with the layout of simmr_reads_out"""

    def __init__(self, n_reads, seq_capacity, qual_offset=0):
        n = max(int(n_reads), 1)
        self.seq = np.zeros(max(int(seq_capacity), 1), dtype=np.uint8)
        self.qual = np.zeros(max(int(seq_capacity), 1), dtype=np.uint8)
        self.seq_off = np.zeros(n + 1, dtype=np.uint64)
        self.start = np.zeros(n, dtype=np.uint64)
        self.end = np.zeros(n, dtype=np.uint64)
        self.contig = np.zeros(n, dtype=np.uint32)
        self.genome = np.zeros(n, dtype=np.uint32)
        self.read_id = np.zeros(n, dtype=np.uint32)
        self.flags = np.zeros(n, dtype=np.uint8)
        o = ReadsOut()
        for name in ("seq", "qual", "seq_off", "start", "end", "contig", "genome", "read_id", "flags"):
            setattr(o, name, getattr(self, name).ctypes.data)
        o.seq_capacity = int(seq_capacity)
        o.reads_capacity = n
        o.qual_offset = qual_offset
        self.pod = o
        self.n_reads = int(n_reads)
        self.total_bases = 0

    def trimmed(self):
        n, tb = self.n_reads, self.total_bases
        return {"seq": self.seq[:tb], "qual": self.qual[:tb], "seq_off": self.seq_off[:n + 1],
                "start": self.start[:n], "end": self.end[:n], "contig": self.contig[:n],
                "genome": self.genome[:n], "read_id": self.read_id[:n], "flags": self.flags[:n]}


def simulate_pe(lib, genome: HostGenome, profile: ErrorProfilePOD, genome_reads, seed, first=0,
                count=(1 << 64) - 1, read_id_base=0, max_len=1024, threads=1, qual_offset=0, out=None):
    n_pairs = genome_reads // 2
    first = min(first, n_pairs)
    count = min(count, n_pairs - first)
    if out is None:
        out = HostReads(2 * count, 2 * count * max_len, qual_offset)
    tb = C.c_uint64()
    rc = lib.orc_simulate_pe_reads_from_genome(C.byref(genome.c), C.byref(profile), genome_reads, seed, first,
                                               count, read_id_base, C.byref(out.pod), C.byref(tb), threads)
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {lib.orc_last_error().decode()}")
    out.total_bases = tb.value
    return out


def simulate_long(lib, genomes, genome_reads, profile, seed, first=0, count=(1 << 64) - 1, read_id_base=0,
                  threads=1, has_seed=True, qual_offset=0):
    n = len(genomes)
    garr = (Genome * n)(*[g.c for g in genomes])
    gr = (C.c_uint64 * n)(*[int(x) for x in genome_reads])
    total = sum(int(x) for x in genome_reads)
    first = min(first, total)
    count = min(count, total - first)
    out = HostReads(count, count * 65535 + 16, qual_offset)
    tb = C.c_uint64()
    cl = C.c_uint32()
    rc = lib.orc_simulate_long_reads(garr, n, gr, C.byref(profile), 1 if has_seed else 0, seed, first, count,
                                     read_id_base, C.byref(out.pod), C.byref(tb), C.byref(cl), threads)
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {lib.orc_last_error().decode()}")
    out.total_bases = tb.value
    out.const_len = cl.value
    return out
