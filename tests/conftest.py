import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "provenance(kind): where a known-answer test's expected values come from "
                            "(reference-held / public third-party / self-generated; tests/test_oracle_kat.py)")


@pytest.fixture(scope="session", autouse=True)
def _hip_library_is_built():
    """A fresh checkout has no binaries (they are git-ignored): build the C-ABI library once, as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  The product itself never
    builds or falls back: simmr_amd._abi.load() raises when the library is missing."""
    import subprocess
    if not (ROOT / "simmr_amd" / "csrc" / "libsimmr_hip.so").exists():
        subprocess.check_call(["make", "-s", "-C", str(ROOT / "simmr_amd" / "csrc")])


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from simmr_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
