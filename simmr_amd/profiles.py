"""Host-side mirror of simmr's plug-in surface for the accelerated path.

Same names and argument meaning as the reference traits:
  ErrorProfile      simmr/src/error_profiles/base.rs:6-32
  AbundanceProfile  simmr/src/abundance_profiles/base.rs:10-69
The per-read methods of ErrorProfile (simulate_phred_scores, ...) run on the
device inside the emit kernels; what lives here is the parameterisation the
reference does in cli.rs:229-320 plus the host-only abundance arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._abi import ErrorProfilePOD


class ErrorProfile:
    kind: int = -1

    def pod(self) -> ErrorProfilePOD:  # flattened form handed to the C ABI
        raise NotImplementedError

    def minimum_genome_size(self) -> int:
        raise NotImplementedError

    def is_long_read(self) -> bool:
        raise NotImplementedError


@dataclass
class PerfectShortErrorProfile(ErrorProfile):
    """error_profiles/perfect_short.rs:8-64"""
    read_length: int = 150
    insert_size: int = 150
    kind = _abi.PERFECT_SHORT

    def pod(self):
        p = ErrorProfilePOD()
        p.kind = self.kind
        p.read_length = self.read_length
        p.insert_size = self.insert_size
        return p

    def get_read_length(self, seed=None):
        return self.read_length

    def get_insert_size(self, seed=None):
        return self.insert_size

    def minimum_genome_size(self):
        return (2 * self.read_length + self.insert_size) & 0xFFFF  # u16 arithmetic, :56-59

    def is_long_read(self):
        return False


@dataclass
class MinimalShortErrorProfile(ErrorProfile):
    """error_profiles/minimal_short.rs:16-150; stds as cli.rs:239-240 fixes them."""
    read_length: int = 150
    insert_size: int = 150
    mean_phred_score: int = 30
    insert_size_std: float = 75.0
    read_length_std: float = 15.0
    rng_mode: int = _abi.RNG_REFERENCE
    kind = _abi.MINIMAL_SHORT

    def pod(self):
        p = ErrorProfilePOD()
        p.kind = self.kind
        p.rng_mode = self.rng_mode
        p.read_length = self.read_length
        p.insert_size = self.insert_size
        p.mean_phred = self.mean_phred_score
        p.read_length_std = self.read_length_std
        p.insert_size_std = self.insert_size_std
        return p

    def minimum_genome_size(self):
        return (2 * self.read_length + self.insert_size) & 0xFFFF  # :142-145

    def is_long_read(self):
        return False


def gamma_params(mean: float, std: float) -> Tuple[float, float]:
    """shape = (mean/std).powf(2.0), scale = std.powf(2.0)/mean in f32
    (minimal_long.rs:64-69)."""
    m, s = np.float32(mean), np.float32(std)
    shape = np.power(m / s, np.float32(2.0), dtype=np.float32)
    scale = np.power(s, np.float32(2.0), dtype=np.float32) / m
    return float(shape), float(np.float32(scale))


@dataclass
class MinimalLongErrorProfile(ErrorProfile):
    """error_profiles/minimal_long.rs:17-159.  The reference hard-codes the
    gamma at mean 20 000 / sd 15 000; gamma_mean/gamma_std expose it."""
    mean_phred_score: int = 30
    gamma_mean: float = 20000.0
    gamma_std: float = 15000.0
    length_mode: int = _abi.LEN_REFERENCE
    rng_mode: int = _abi.RNG_REFERENCE
    uniform_start: bool = False  # SIMMR_START_UNIFORM: starts over the whole sequence (extension)
    kind = _abi.MINIMAL_LONG

    def pod(self):
        p = ErrorProfilePOD()
        p.kind = self.kind
        p.rng_mode = self.rng_mode
        p.length_mode = self.length_mode
        p.long_start_mode = _abi.START_UNIFORM if self.uniform_start else _abi.START_REFERENCE
        p.mean_phred = self.mean_phred_score
        p.gamma_shape, p.gamma_scale = gamma_params(self.gamma_mean, self.gamma_std)
        return p

    def minimum_genome_size(self):
        return 20000  # :152-154

    def is_long_read(self):
        return True


@dataclass
class PerfectLongErrorProfile(MinimalLongErrorProfile):
    """error_profiles/perfect_long.rs:17-136 (not actually error free: Q8)."""
    kind = _abi.PERFECT_LONG


class CustomShortErrorProfile(ErrorProfile):
    """error_profiles/custom_short.rs: empirical read-length / insert-size / per-position
    quality distributions from a simmrd model (bincode ErrorModelParams).  `model` is the
    file's bytes; they are handed to the library, which builds the alias tables."""
    kind = _abi.CUSTOM

    def __init__(self, model: bytes, rng_mode: int = _abi.RNG_REFERENCE):
        import ctypes
        self.model = bytes(model)
        self._buf = ctypes.create_string_buffer(self.model, len(self.model))
        # RNG_PHILOX: the draws of the k-mer splice (simulate_errors) from Philox counters; long-read models only
        self.rng_mode = rng_mode

    def pod(self):
        import ctypes
        p = ErrorProfilePOD()
        p.kind = self.kind
        p.rng_mode = self.rng_mode
        p.custom_model = ctypes.cast(self._buf, ctypes.c_void_p).value
        p.custom_model_bytes = len(self.model)
        return p

    def is_long_read(self):
        # custom_short.rs:540-542: the model's is_long flag, the last byte of the bincode struct
        # (main.rs:30-33 refuses such a model for the custom-short CLI value; the library takes it on the long-read path)
        return len(self.model) > 0 and self.model[-1] != 0


# ---------------------------------------------------------------------------
class AbundanceProfile:
    def is_size_aware(self) -> bool:
        raise NotImplementedError

    def determine_abundances(self, total_reads: int, num_genomes: int) -> List[Tuple[int, float]]:
        raise NotImplementedError

    def adjust_for_size(self, genome_sizes: Sequence[int], read_abundances, read_length: int, paired: bool):
        """uniform.rs:79-94 == custom.rs:80-95 (read_length/paired only feed the
        unused total_coverage there)."""
        total_reads = 0.0
        for n, _ in read_abundances:
            total_reads += float(n)
        total_adjusts = 0.0
        for s, (_, a) in zip(genome_sizes, read_abundances):
            total_adjusts += float(s) * a
        return [(int(math.ceil(total_reads * ((a * float(s)) / total_adjusts))), a)
                for s, (_, a) in zip(genome_sizes, read_abundances)]


@dataclass
class UniformAbundanceProfile(AbundanceProfile):
    """abundance_profiles/uniform.rs:13-96"""
    size_adjusted: bool = False

    def is_size_aware(self):
        return self.size_adjusted

    def determine_abundances(self, total_reads, num_genomes):
        per = int(math.ceil(float(total_reads) / float(num_genomes)))
        return [(per, 100.0 / float(num_genomes))] * num_genomes


@dataclass
class ExactAbundanceProfile(AbundanceProfile):
    """abundance_profiles/exact.rs:12-36"""

    def is_size_aware(self):
        return False

    def determine_abundances(self, total_reads, num_genomes):
        return [(int(total_reads), 100.0 / float(num_genomes))] * num_genomes

    def adjust_for_size(self, genome_sizes, read_abundances, read_length, paired):
        return list(read_abundances)


@dataclass
class CustomAbundanceProfile(AbundanceProfile):
    """abundance_profiles/custom.rs:15-97"""
    abundances: List[float] = field(default_factory=list)
    size_adjusted: bool = False

    def is_size_aware(self):
        return self.size_adjusted

    def determine_abundances(self, total_reads, num_genomes):
        total = 0.0
        for a in self.abundances:
            total += a
        if total < 0.99 or total > 1.01:
            return [(int(math.ceil(float(total_reads) * (a / total))), a / total) for a in self.abundances]
        return [(int(math.ceil(float(total_reads) * a)), a) for a in self.abundances]
