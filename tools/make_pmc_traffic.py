#!/usr/bin/env python3
"""profiles/<round>/pmc_traffic.json from the PMC summaries of tools/profile_round.sh, stamped with the hash of the kernel
sources it was collected from (bench.py reports its numbers only while that hash still matches).

usage: tools/make_pmc_traffic.py <dir with pmc_default.txt [pmc_compact.txt pmc_perfect.txt pmc_custom_long.txt]>
       [--mix plain,vop3_sdwa,mad_u64] [--mix-slot plain,vop3_sdwa,mad_u64] [--round r5] [--splice-bases N]
--splice-bases: bases the custom-long bench's splice launch wrote (the command's stream bytes / 2): the record of the
counter-mode splice then carries its write amplification on its own, WRITE_SIZE / bases
(pmc_default.txt = the default command, i.e. the 16-byte read slots; pmc_compact.txt = the same with --layout compact)"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def read(path):
    d = {}
    for line in Path(path).read_text().splitlines():
        parts = line.split()
        if len(parts) == 2 and not line.startswith("#"):
            try:
                d[parts[0]] = float(parts[1])
            except ValueError:
                pass
    return d


def record(d, workload, note):
    r = {"workload": workload}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        r["FETCH_SIZE_KB"], r["WRITE_SIZE_KB"] = d["FETCH_SIZE"], d["WRITE_SIZE"]
        r["bytes_raw"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        r["bytes_fetch_doubled"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    for k, v in d.items():
        if k.startswith("SQ_") or k.startswith("GRBM_"):
            r[k] = v
    r["note"] = note
    return r


def main():
    src = Path(sys.argv[1])
    out = {"source_sha256": bench.kernel_source_hash(), "sources": list(bench.KERNEL_SOURCES)}
    mixes = {}
    for flag, key in (("--mix", "k_emit_philox"), ("--mix-slot", "k_emit_philox_slot16")):
        if flag in sys.argv:
            a, b, c = (float(x) for x in sys.argv[sys.argv.index(flag) + 1].split(","))
            mixes[key] = {"plain": a, "vop3_sdwa": b, "mad_u64": c}
    for name, key, workload in (("pmc_default.txt", "k_emit_philox_slot16", "bench.py default (minimal-short 150 bp PE, 100 Mbp, 100 M reads, counter mode, 16-byte read slots)"),
                                ("pmc_compact.txt", "k_emit_philox", "bench.py --layout compact (the same run with byte streams without gaps)"),
                                ("pmc_perfect.txt", "k_emit_perfect_pe", "bench.py --profile perfect-short"),
                                ("pmc_custom_long.txt", "k_custom_long_splice_ctr", "bench.py --profile custom-long --reads 1000000 (counter mode of the splice)"),
                                ("pmc_custom_long_reference.txt", "k_custom_long_splice", "bench.py --profile custom-long --rng reference --reads 1000000"),
                                ("pmc_through_fastq.txt", "k_emit_philox_text", "bench.py --through-fastq: the TEXT form of the counter-mode kernel (simmr_emit_fastq), its launches only")):
        f = src / name
        if f.exists():
            out[key] = record(read(f), workload, "separate --pmc passes of the bench command with --steps 1 --warmup 0 "
                                                 "(tools/profile_round.sh), per launch")
            if key in mixes:
                out[key]["valu_class_mix"] = mixes[key]
    if "--splice-bases" in sys.argv:
        bases = float(sys.argv[sys.argv.index("--splice-bases") + 1])
        for key in ("k_custom_long_splice_ctr", "k_custom_long_splice"):
            if key in out and "WRITE_SIZE_KB" in out[key] and bases > 0:
                out[key]["bases_written"] = bases
                out[key]["write_bytes_per_base"] = out[key]["WRITE_SIZE_KB"] * 1024.0 / bases  # (1.0 = every base stored once)
                out[key]["fetch_bytes_per_base"] = out[key]["FETCH_SIZE_KB"] * 1024.0 / bases
    tag = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r5"
    dst = ROOT / "profiles" / tag / "pmc_traffic.json"
    dst.parent.mkdir(parents=True, exist_ok=True)
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(dst, out["source_sha256"][:12])


if __name__ == "__main__":
    main()
