"""Regenerates tests/golden/oracle_golden.json from the CPU oracle.

The reference is Rust and cannot run in this image, so these vectors are
outputs of the ORACLE (oracle/*.c), frozen so drift is detected on CPU-only
runs.  ecoli_partial_7920.txt is the sequence data of the reference's own test
fixture simmr/src/tests/data/GCF_000005845.2_ASM584v2_genomic.partial.fna
(newlines removed)."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from simmr_amd import MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectShortErrorProfile  # noqa: E402
from tests import _oracle, _synth  # noqa: E402

lib = _oracle.load()
cases = [
    dict(name="c1_perfect_short", kind="pe", profile="perfect-short", contig_lens=[1_000_000], genome_seed=1, reads=10000, seed=42),
    dict(name="minimal_short_1m", kind="pe", profile="minimal-short", contig_lens=[1_000_000], genome_seed=1, reads=10000, seed=42),
    dict(name="minimal_short_multi", kind="pe", profile="minimal-short", contig_lens=[50_000, 20_000, 9_000], genome_seed=3, reads=2000, seed=7),
    dict(name="minimal_long", kind="long", profile="minimal-long", contig_lens=[300_000, 90_000, 30_000], genome_seed=7, reads=50, seed=42),
]
for c in cases:
    g = _oracle.HostGenome(_synth.synthetic_contigs(c["contig_lens"], c["genome_seed"]))
    if c["kind"] == "pe":
        cls = {"perfect-short": PerfectShortErrorProfile, "minimal-short": MinimalShortErrorProfile}[c["profile"]]
        out = _oracle.simulate_pe(lib, g, cls().pod(), c["reads"], c["seed"])
    else:
        out = _oracle.simulate_long(lib, [g], [c["reads"]], MinimalLongErrorProfile().pod(), c["seed"])
    d = out.trimmed()
    c["sha256"] = {k: hashlib.sha256(np.ascontiguousarray(d[k]).tobytes()).hexdigest()
                   for k in ("seq", "qual", "seq_off", "start", "end", "contig", "read_id", "flags")}
    c["start_head"] = [int(x) for x in d["start"][:8]]
    c["seq_head"] = d["seq"][:60].tobytes().decode()
(Path(__file__).parent / "oracle_golden.json").write_text(json.dumps({"cases": cases}, indent=1))
print("wrote", len(cases), "cases")
