"""ctypes mirror of include/simmr_hip.h and loader for libsimmr_hip.so.

This is plumbing only: the product is the C-ABI shared library built from
simmr_amd/csrc (hand-written HIP for gfx950).  There is no Python or CPU
fallback — if the library is missing, `load()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parent
LIB_PATH = ROOT / "csrc" / "libsimmr_hip.so"

# status codes
OK, EINVAL, ENOMEM, ENODEV, ERANGE, ESTATE, EGENOME, ENOTSUP = 0, -22, -12, -19, -34, -1, -61, -95

# enum simmr_profile_kind
PERFECT_SHORT, MINIMAL_SHORT, PERFECT_LONG, MINIMAL_LONG, CUSTOM = range(5)
COMM_ID_BYTES = 128
# enum simmr_rng_mode
RNG_REFERENCE, RNG_PHILOX, RNG_PHILOX_FULL = 0, 1, 2
# enum simmr_length_mode
LEN_REFERENCE, LEN_PER_READ = 0, 1
# enum simmr_long_start_mode
START_REFERENCE, START_UNIFORM = 0, 1

FLAG_REVCOMP, FLAG_QSEED_SUBST, FLAG_MSEED_SUBST, FLAG_REDRAWN = 1, 2, 4, 8

(CNT_READS, CNT_BASES, CNT_ACGT_BASES, CNT_SUBSTITUTIONS, CNT_OUTER_REJECTS, CNT_REDRAWN,
 CNT_SEED_SUBST, CNT_QUAL_SUM) = range(8)
N_COUNTERS = 8

U64_MAX = (1 << 64) - 1


class ErrorProfilePOD(C.Structure):
    """struct simmr_error_profile"""
    _fields_ = [
        ("kind", C.c_uint32),
        ("rng_mode", C.c_uint32),
        ("length_mode", C.c_uint32),
        ("read_length", C.c_uint16),
        ("insert_size", C.c_uint16),
        ("mean_phred", C.c_uint8),
        ("long_start_mode", C.c_uint8),
        ("reserved0", C.c_uint8 * 2),
        ("read_length_std", C.c_double),
        ("insert_size_std", C.c_double),
        ("gamma_shape", C.c_float),
        ("gamma_scale", C.c_float),
        ("custom_model", C.c_void_p),
        ("custom_model_bytes", C.c_uint64),
    ]


class Range(C.Structure):
    """struct simmr_range"""
    _fields_ = [("first", C.c_uint64), ("count", C.c_uint64)]


class PlanInfo(C.Structure):
    """struct simmr_plan_info"""
    _fields_ = [
        ("n_units", C.c_uint64),
        ("n_reads", C.c_uint64),
        ("total_bases", C.c_uint64),
        ("seed_used", C.c_uint64),
        ("outer_slots", C.c_uint64),
        ("const_read_length", C.c_uint32),
        ("slot_bytes", C.c_uint32),
    ]


class ReadsOut(C.Structure):
    """struct simmr_reads_out (pointers are raw addresses: device for the HIP
    library, host for the test oracle)."""
    _fields_ = [
        ("seq", C.c_void_p),
        ("qual", C.c_void_p),
        ("seq_off", C.c_void_p),
        ("start", C.c_void_p),
        ("end", C.c_void_p),
        ("contig", C.c_void_p),
        ("genome", C.c_void_p),
        ("read_id", C.c_void_p),
        ("flags", C.c_void_p),
        ("seq_capacity", C.c_uint64),
        ("reads_capacity", C.c_uint64),
        ("qual_offset", C.c_uint32),
        ("slot_bytes", C.c_uint32),
    ]


class OuterSummary(C.Structure):
    """struct simmr_outer_summary"""
    _fields_ = [("units", C.c_uint64 * 2), ("end_state", C.c_uint32 * 2)]


class FastqNames(C.Structure):
    """struct simmr_fastq_names"""
    _fields_ = [
        ("n_genomes", C.c_uint32),
        ("genome_idx", C.POINTER(C.c_uint32)),
        ("genome_id", C.POINTER(C.c_char_p)),
        ("n_contigs", C.POINTER(C.c_uint32)),
        ("sequence_id", C.POINTER(C.c_char_p)),
    ]


# every symbol include/simmr_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "simmr_abi_version": (C.c_int, []),
    "simmr_engine_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "simmr_engine_destroy": (None, [C.c_void_p]),
    "simmr_last_error": (C.c_char_p, [C.c_void_p]),
    "simmr_engine_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "simmr_engine_set_read_slots": (C.c_int, [C.c_void_p, C.c_uint32]),
    "simmr_engine_set_plan_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "simmr_stage_genome": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_void_p),
                                     _P(C.c_uint64), _P(C.c_uint64)]),
    "simmr_stage_fasta": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_void_p), _P(C.c_uint64), C.c_int,
                                    C.c_uint64, _P(C.c_uint64), _P(C.c_uint32)]),
    "simmr_stage_synthetic": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_uint64),
                                        C.c_uint64]),
    "simmr_unstage_contig": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_uint64, C.c_void_p]),
    "simmr_genome_info": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(C.c_uint64)]),
    "simmr_pe_plan": (C.c_int, [C.c_void_p, C.c_uint32, _P(ErrorProfilePOD), C.c_uint64, C.c_int,
                                C.c_uint64, Range, _P(PlanInfo)]),
    "simmr_outer_summarize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, _P(OuterSummary)]),
    "simmr_pe_plan_at": (C.c_int, [C.c_void_p, C.c_uint32, _P(ErrorProfilePOD), C.c_uint64, C.c_uint64, Range,
                                   C.c_uint64, C.c_uint64, _P(PlanInfo)]),
    "simmr_pe_plan_multi": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(C.c_uint64), _P(ErrorProfilePOD),
                                      C.c_int, C.c_uint64, Range, _P(PlanInfo)]),
    "simmr_pe_emit": (C.c_int, [C.c_void_p, C.c_uint32, _P(ReadsOut)]),
    "simmr_long_plan": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(C.c_uint64),
                                  _P(ErrorProfilePOD), C.c_int, C.c_uint64, Range, _P(PlanInfo)]),
    "simmr_long_emit": (C.c_int, [C.c_void_p, C.c_uint32, _P(ReadsOut)]),
    "simmr_counters": (C.c_int, [C.c_void_p, C.c_void_p, _P(C.c_uint64)]),
    "simmr_counters_reset": (C.c_int, [C.c_void_p]),
    "simmr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "simmr_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "simmr_allreduce_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "simmr_last_emit_kernel_ms": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "simmr_emit_kernel_ms_mean": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]),
    "simmr_last_plan_ms": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "simmr_entropy_substitute": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "simmr_fastq_plan": (C.c_int, [C.c_void_p, C.c_char_p, _P(FastqNames), _P(ReadsOut), C.c_uint64, C.c_int,
                                   _P(C.c_uint64)]),
    "simmr_fastq_emit": (C.c_int, [C.c_void_p, _P(ReadsOut), C.c_void_p, C.c_uint64]),
    "simmr_fastq_plan_direct": (C.c_int, [C.c_void_p, C.c_char_p, _P(FastqNames), C.c_uint32, _P(C.c_uint64)]),
    "simmr_emit_fastq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "simmr_last_fastq_plan_ms": (C.c_int, [C.c_void_p, _P(C.c_float)]),
}

_lib = None


class SimmrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"simmr error {code}: {msg}")
        self.code = code
        self.msg = msg


def load(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen libsimmr_hip.so and bind every declared symbol.  Raises if the
    HIP extension has not been built — there is deliberately no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 / HSA
    # runtime.  If libsimmr_hip.so is dlopen'ed first it binds (RTLD_NOW) to
    # /opt/rocm's copy and the second runtime to initialise finds no device.
    # Loading torch first puts its runtime in the global scope, and ours binds
    # to that one.  Without torch in the process the /opt/rocm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # SIMMR_HIP_LIB: another build of the same library (A/B timing of a kernel change on one box)
    p = Path(path or os.environ.get("SIMMR_HIP_LIB") or LIB_PATH)
    if not p.exists():
        raise ImportError(
            f"{p} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C simmr_amd/csrc)")
    lib = C.CDLL(str(p))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
