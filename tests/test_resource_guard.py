"""Build-time guard for registers, scratch and occupancy.

Round 3's TEXT form of k_emit_philox went from 127 to 130 VGPRs with one commit — three waves per SIMD instead of four
on a kernel that spends two thirds of its time waiting — and nobody saw it among 58 instantiations.  This test compiles
the library's kernels for gfx950 with `-Rpass-analysis=kernel-resource-usage` (no GPU needed; about a minute) and holds
the kernels the bench lines run to the figures DESIGN.md documents (section 4, "Kernels").
"""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def kernels():
    import resource_usage
    return resource_usage.collect()


def _named(kernels, pattern):
    return [k for k in kernels if re.search(pattern, k["name"])]


def test_counter_mode_item_kernel_every_instantiation(kernels):
    """k_emit_philox: at most 30 instantiations (27 today: engine.hip philox_kernel / philox_text_kernel), each of them —
    the column forms in both layouts, the TEXT forms, the copy-only forms — at four waves per SIMD or more without a
    byte of scratch."""
    ks = _named(kernels, r"k_emit_philox<")
    assert 0 < len(ks) <= 30, len(ks)
    for k in ks:
        assert k["occupancy"] >= 4 and k["scratch"] == 0 and k["vgpr"] <= 128 and k["agpr"] == 0, k
    # <HAS_EXC, COPY_ONLY, CACHED, TEXT, ESCQ, SLOT, COARSE>: the three forms of the default bench line and its side lines
    flags = {tuple(re.search(r"k_emit_philox<([^>]*)>", k["name"]).group(1).split(", ")): k for k in ks}
    slot16 = flags[("false", "false", "true", "false", "true", "true", "true")]
    compact = flags[("false", "false", "true", "false", "true", "false", "true")]
    text = flags[("false", "false", "true", "true", "true", "false", "true")]
    assert slot16["vgpr"] <= 104 and compact["vgpr"] <= 104 and text["vgpr"] <= 116, (slot16, compact, text)
    # four workgroups (one wave per SIMD each) fit a CU's 160 KB of LDS with room for the TEXT form's dynamic header slots
    assert all(k["lds"] <= 30 * 1024 for k in ks)
    # TEXT implies COARSE (no per-record offsets exist for the text)
    assert all(f[6] == "true" for f in flags if f[3] == "true")


def test_whole_line_text_kernel(kernels):
    """k_emit_text_lines (text_lines.hip, what simmr_emit_fastq runs for paired short reads): nine instantiations, three
    waves per SIMD without scratch, and — the binding number — static LDS that leaves room for the header slots of the
    reference's default header format beside THREE workgroups per CU (160 KB): 128 slots x 136 bytes = 17 408 bytes."""
    ks = _named(kernels, r"k_emit_text_lines<")
    assert 0 < len(ks) <= 9, len(ks)
    for k in ks:
        assert k["occupancy"] >= 3 and k["scratch"] == 0 and k["vgpr"] <= 168 and k["agpr"] == 0, k
    flags = {tuple(re.search(r"k_emit_text_lines<([^>]*)>", k["name"]).group(1).split(", ")): k for k in ks}
    bench = flags[("false", "false", "true", "true")]  # <HAS_EXC, COPY_ONLY, CACHED, ESCQ>: bench.py --through-fastq
    assert 3 * (bench["lds"] + 17408) <= 160 * 1024, bench


def test_other_bench_kernels(kernels):
    for name, occ, scratch in (("k_emit_perfect_pe<", 7, 0), ("k_plan_pe<", 6, 0), ("k_outer_classify", 8, 0),
                               ("k_outer_scan", 8, 0), ("k_outer_emit", 8, 0), ("k_fastq_size_plan", 8, 0),
                               ("k_emit_lanes<", 4, 32),               # bit-exact mode: 32 bytes per lane, documented
                               (r"k_custom_long_splice<(true|false), (true|false), false>", 4, 0),  # the k-mer splice on the reference's streams: four waves per SIMD around its LDS rows
                               # its counter mode: six waves per SIMD (two workgroups of 768 lanes around the 64 KB table); the
                               # cap costs the fixed-stride forms 16 / 36 bytes of spills around the group loop, none in the step loop
                               (r"k_custom_long_splice<(true|false), (true|false), true>", 6, 36),
                               ("k_plan_long_per_read", 5, 0), ("k_fastq_write", 6, 0)):
        ks = _named(kernels, name)
        assert ks, name
        for k in ks:
            assert k["occupancy"] >= occ and k["scratch"] <= scratch, (name, k)


def test_library_size(kernels):
    """The forms that were measured and lost are not in the tree any more (git history keeps them)."""
    assert len(kernels) <= 88, len(kernels)
    assert not _named(kernels, r"k_emit_philox_tile|k_emit_stream|k_fastq_headers")
