"""CPU-only checks of the host layer: the C-ABI library loads and exports every
symbol include/simmr_hip.h declares (no compute without a GPU), the host
mirrors of the abundance profiles agree with the oracle and with the
reference's own unit tests, and the product never reaches into oracle/."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from simmr_amd import (CustomAbundanceProfile, ExactAbundanceProfile, MinimalLongErrorProfile,
                       MinimalShortErrorProfile, PerfectShortErrorProfile, UniformAbundanceProfile, _abi)
from simmr_amd.profiles import gamma_params

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "simmr_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|void|uint64_t|const char\*)\s+(simmr_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    lib = _abi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.simmr_abi_version() == 1


def test_struct_layouts_match_header():
    # offsets the C compiler produces for the header structs (checked with gcc below)
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "simmr_hip.h"
int main(void){
 printf("%zu %zu %zu %zu\n", sizeof(simmr_error_profile), sizeof(simmr_range), sizeof(simmr_plan_info), sizeof(simmr_reads_out));
 printf("%zu %zu %zu %zu %zu\n", offsetof(simmr_error_profile, read_length), offsetof(simmr_error_profile, mean_phred),
        offsetof(simmr_error_profile, read_length_std), offsetof(simmr_error_profile, gamma_shape), offsetof(simmr_error_profile, custom_model));
 printf("%zu %zu %zu\n", offsetof(simmr_reads_out, flags), offsetof(simmr_reads_out, seq_capacity), offsetof(simmr_reads_out, qual_offset));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(src)
        subprocess.check_call(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/t", f"{d}/t.c"])
        out = subprocess.check_output([f"{d}/t"]).decode().split()
    got = list(map(int, out))
    P, R = _abi.ErrorProfilePOD, _abi.ReadsOut
    want = [C.sizeof(P), C.sizeof(_abi.Range), C.sizeof(_abi.PlanInfo), C.sizeof(R),
            P.read_length.offset, P.mean_phred.offset, P.read_length_std.offset, P.gamma_shape.offset,
            P.custom_model.offset, R.flags.offset, R.seq_capacity.offset, R.qual_offset.offset]
    assert got == want


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _abi.load()
    h = C.c_void_p()
    rc = lib.simmr_engine_create(0, C.byref(h))
    assert rc == _abi.ENODEV and not h.value
    assert b"device" in lib.simmr_last_error(None).lower()
    rc = lib.simmr_engine_create(-1, C.byref(h))  # there is no CPU backend
    assert rc == _abi.ENODEV


def test_missing_library_fails_loudly(tmp_path):
    """No fallback: without the built HIP library the loader raises."""
    with pytest.raises(ImportError) as ei:
        _abi.load(tmp_path / "libsimmr_hip.so")
    assert "build the HIP extension" in str(ei.value)


def test_entropy_substitute_matches_oracle(oracle):
    lib = _abi.load()
    for x in (0, 1, 42, 2 ** 64 - 1, 9713269763989775522):
        for w in (1, 2, 3):
            assert lib.simmr_entropy_substitute(x, w) == oracle.orc_entropy_substitute(x, w)


def test_product_does_not_touch_the_oracle():
    for p in (ROOT / "simmr_amd").rglob("*"):
        if p.is_file() and p.suffix in (".py", ".hip", ".hpp", ".cpp", ".h", "") and p.name != "libsimmr_hip.so":
            txt = p.read_text(errors="ignore")
            assert "oracle" not in txt.lower() or p.name in ("_abi.py", "engine.py") and "liboracle" not in txt, p
    import subprocess
    deps = subprocess.check_output(["ldd", str(_abi.LIB_PATH)]).decode()
    assert "oracle" not in deps


# ---- AbundanceProfile (host-only arithmetic, must be exact: ceil!) -----------
def _orc_abund(oracle, fn, *args, n):
    reads = np.zeros(n, dtype=np.uint64)
    ab = np.zeros(n)
    getattr(oracle, fn)(*args, C.c_void_p(reads.ctypes.data), C.c_void_p(ab.ctypes.data))
    return list(zip(map(int, reads), map(float, ab)))


def test_uniform_profile_reference_unit_test():
    # abundance_profile_tests.rs:7-30
    ab = UniformAbundanceProfile(False).determine_abundances(100, 5)
    assert ab == [(20, 20.0)] * 5


@pytest.mark.parametrize("total,n", [(100, 5), (1000, 3), (10 ** 9, 1000), (7, 7), (1, 64), (50_000_000, 64)])
def test_uniform_exact_vs_oracle(oracle, total, n):
    assert UniformAbundanceProfile().determine_abundances(total, n) == \
        _orc_abund(oracle, "orc_uniform_determine_abundances", total, n, n=n)
    assert ExactAbundanceProfile().determine_abundances(total, n) == \
        _orc_abund(oracle, "orc_exact_determine_abundances", total, n, n=n)


@pytest.mark.parametrize("n,norm", [(64, False), (5, True), (3, False)])
def test_custom_vs_oracle(oracle, n, norm):
    a = np.array([1.0 / (g + 1) for g in range(n)])
    if norm:
        a = a / a.sum()
    got = CustomAbundanceProfile(list(a)).determine_abundances(50_000_000, n)
    want = _orc_abund(oracle, "orc_custom_determine_abundances", C.c_void_p(a.ctypes.data), 50_000_000, n, n=n)
    assert got == want


def test_adjust_for_size_vs_oracle(oracle):
    sizes = np.array([5_000_000, 1_234_567, 99_999, 10_000_000], dtype=np.uint64)
    prof = UniformAbundanceProfile(True)
    base = prof.determine_abundances(1_000_001, 4)
    got = prof.adjust_for_size(list(map(int, sizes)), base, 150, True)
    rin = np.array([r for r, _ in base], dtype=np.uint64)
    ain = np.array([a for _, a in base])
    rout = np.zeros(4, dtype=np.uint64)
    aout = np.zeros(4)
    oracle.orc_adjust_for_size(C.c_void_p(sizes.ctypes.data), C.c_void_p(rin.ctypes.data), C.c_void_p(ain.ctypes.data), 4,
                               C.c_void_p(rout.ctypes.data), C.c_void_p(aout.ctypes.data))
    assert got == list(zip(map(int, rout), map(float, aout)))
    assert ExactAbundanceProfile().adjust_for_size([1, 2], [(5, 50.0), (5, 50.0)], 150, True) == [(5, 50.0), (5, 50.0)]


def test_profile_pods():
    p = MinimalShortErrorProfile().pod()
    assert (p.kind, p.read_length, p.insert_size, p.mean_phred) == (_abi.MINIMAL_SHORT, 150, 150, 30)
    assert (p.read_length_std, p.insert_size_std) == (15.0, 75.0)  # cli.rs:239-240
    assert PerfectShortErrorProfile().minimum_genome_size() == 450
    assert PerfectShortErrorProfile(30000, 30000).minimum_genome_size() == (90000 & 0xFFFF)  # u16 wrap, Q7
    assert MinimalLongErrorProfile().minimum_genome_size() == 20000 and MinimalLongErrorProfile().is_long_read()
    shape, scale = gamma_params(20000.0, 15000.0)
    assert abs(shape - 16 / 9) < 1e-6 and abs(scale - 11250.0) < 1e-2


def test_integration_doc_binds_every_symbol():
    """INTEGRATION.md's Rust block declares every function of include/simmr_hip.h."""
    import re
    root = Path(__file__).resolve().parent.parent
    header = (root / "include" / "simmr_hip.h").read_text()
    doc = (root / "INTEGRATION.md").read_text()
    names = set(re.findall(r"^(?:int|void|const char\*|uint64_t)\s+(simmr_[a-z0-9_]+)\(", header, flags=re.M))
    assert len(names) >= 20
    missing = [n for n in sorted(names) if f"pub fn {n}(" not in doc]
    assert not missing, missing


def test_missing_rccl_answers_enodev_without_a_gpu():
    """A host whose librccl cannot be loaded gets SIMMR_ENODEV from the communicator entry points, not a crash
    (engine.hip:rccl_api reads dlerror() once).  Own process: the loader result is cached per process."""
    import subprocess, sys
    code = (
        "import ctypes as C\n"
        "from simmr_amd import _abi\n"
        "lib = _abi.load()\n"
        "buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()\n"
        "rc = lib.simmr_comm_unique_id(buf)\n"
        "rc2 = lib.simmr_comm_unique_id(buf)\n"
        "print(rc, rc2)\n")
    env = dict(__import__("os").environ, SIMMR_RCCL_LIB="/nonexistent/librccl-missing.so", PYTHONPATH=str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == [str(_abi.ENODEV)] * 2


def test_header_states_the_shipped_philox_specification():
    """include/simmr_hip.h is what a maintainer reads: it must describe the generator the library implements (version 3:
    24 bits per base, three calls per 16 bases, two levels), not an earlier one."""
    text = (ROOT / "include" / "simmr_hip.h").read_text()
    for needle in ("0x73696D6D", "0x72000003", "24 bits per base", "3g + {0, 1, 2}", "level 2", "oracle/philox.c"):
        assert needle in text, needle
    assert "one output word per base" not in text and "base index / 4" not in text
    # the same constants in the specification's restatement and in the kernel
    assert "0x73696D6D" in (ROOT / "oracle" / "philox.c").read_text().upper().replace("0X", "0x")
    assert "0x73696D6Du" in (ROOT / "simmr_amd" / "csrc" / "rng_device.hpp").read_text()  # (philox4x32_10, used by kernels.hip)
    for needle in ("SIMMR_RNG_PHILOX_FULL", "(w >> 2, 3, 0x73696D6D, 0x72000003)", "4 | (p >> 32) << 8"):  # the full counter mode
        assert needle in text, needle
    # the k-mer splice of a custom long-read model: the shipped one-word form, not the two-word form
    # of its first commit; the same statement in the oracle and next to the product's table builder
    for needle in ("(i >> 2, 2, 0x73696D6D, 0x72000003)", "X = word i & 3", "T24 = 2^24 - 2^e", "Z = (X - (T24 << 8)) << (24 - e)"):
        assert needle in text, needle
    for stale in ("i >> 1, 2", "A = 2 (i & 1)", "m = B * n", "min(floor(2^24 P(self))"):
        assert stale not in text, stale
    for f in ("oracle/custom.c", "simmr_amd/csrc/custom_model.hpp"):
        assert "2^24 - 2^e" in (ROOT / f).read_text(), f
    assert "(uint32_t)(i >> 2), 2u, 0x73696D6Du, 0x72000003u" in (ROOT / "oracle" / "custom.c").read_text()
    assert "philox4x32_10(i >> 2, 2u," in (ROOT / "simmr_amd" / "csrc" / "kernels.hip").read_text()
