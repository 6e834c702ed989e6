"""Synthetic custom error models for the tests: the bincode writer and the generators live in the package
(simmr_amd/model_io.py, the writer half of SURVEY §8f3)."""
from simmr_amd.model_io import *  # noqa: F401,F403
from simmr_amd.model_io import _bins  # noqa: F401
