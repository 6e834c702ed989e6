"""simmr_amd — MI355X-native per-read sampling / mutation path of genomicsoup/simmr.

The product is simmr_amd/csrc/libsimmr_hip.so (hand-written HIP for gfx950
behind the C ABI of include/simmr_hip.h).  This package is the thin host-side
mirror of the reference's plug-in surface plus ctypes plumbing.
"""
from . import _abi
from ._abi import SimmrError
from .profiles import (AbundanceProfile, CustomAbundanceProfile, CustomShortErrorProfile, ErrorProfile, ExactAbundanceProfile,
                       MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectLongErrorProfile,
                       PerfectShortErrorProfile, UniformAbundanceProfile)

__all__ = ["_abi", "SimmrError", "Engine", "Reads", "ErrorProfile", "AbundanceProfile",
           "PerfectShortErrorProfile", "MinimalShortErrorProfile", "PerfectLongErrorProfile",
           "MinimalLongErrorProfile", "CustomShortErrorProfile", "UniformAbundanceProfile", "ExactAbundanceProfile",
           "CustomAbundanceProfile"]


def __getattr__(name):
    if name in ("Engine", "Reads"):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
