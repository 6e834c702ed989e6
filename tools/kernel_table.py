#!/usr/bin/env python3
"""The kernel table of DESIGN.md section 4 and of profiles/README.md, DERIVED from the committed profile files instead of
copied by hand (hand-copied figures had drifted from their sources).

    python3 tools/kernel_table.py r5            # markdown table on stdout
    python3 tools/kernel_table.py r5 --check    # also verifies that DESIGN.md / profiles/README.md quote it verbatim
    python3 tools/kernel_table.py r5 --update   # rewrites the quoted table in both documents

For every bench command of the round (profiles/<round>/kernel_stats_bench_<cmd>.csv = the rocprofv3 --kernel-trace --stats
summary of that command, profiles/<round>/bench_<cmd>_under_rocprof.json = the JSON line the same run printed) it lists
each kernel that took at least 1 % of the command's GPU time: calls, average duration, and — for the kernel(s) the line's
`roofline` object names — the algorithmic bytes per launch, the achieved rate and the fraction of the 8 TB/s
HBM peak, recomputed here from the CSV's average (NOT from the JSON's HIP-event figure, which is quoted beside it so that
the two can be compared: the contract asks that they agree)."""
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PEAK = 8000.0  # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


def short(name):
    """k_emit_philox<false, false, true, true, ...>(args) -> k_emit_philox<F,F,T,T,..>"""
    m = re.match(r"(?:void )?(?:simmr::)?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    base, targs = m.group(1), m.group(2) or ""
    targs = targs.replace("true", "T").replace("false", "F").replace(", ", ",")
    return base + targs


def roofline_kernels(label):
    """'k_custom_long_qual + k_custom_long_splice<CTR>' -> ['k_custom_long_qual', 'k_custom_long_splice']"""
    return [re.match(r"[A-Za-z_0-9]+", p.strip()).group(0) for p in label.split("+")]


def rows(tag):
    d = ROOT / "profiles" / tag
    out = []
    for csv_path in sorted(d.glob("kernel_stats_bench_*.csv")):
        cmd = csv_path.stem[len("kernel_stats_bench_"):]
        js = d / f"bench_{cmd}_under_rocprof.json"
        line = json.loads(js.read_text()) if js.exists() and js.read_text().strip() else {}
        roof = line.get("roofline", {})
        names = roofline_kernels(roof.get("kernel", "")) if roof.get("kernel") else []
        ks = list(csv.DictReader(csv_path.open()))
        total = sum(float(k["TotalDurationNs"]) for k in ks) or 1.0
        # the roofline's kernels: the instantiation with the most calls among those the label names (a command also
        # launches other forms of the same template once, as side measurements)
        chosen = {}
        for base in names:
            cand = [k for k in ks if short(k["Name"]).split("<")[0] == base]
            if cand:
                chosen[base] = max(cand, key=lambda k: (int(k["Calls"]), float(k["TotalDurationNs"])))  # the timed steps' form
        roof_ms = sum(float(k["AverageNs"]) for k in chosen.values()) / 1e6
        for k in ks:
            share = float(k["TotalDurationNs"]) / total
            if share < 0.01:
                continue
            r = {"cmd": cmd, "kernel": short(k["Name"]), "calls": int(k["Calls"]), "avg_ms": float(k["AverageNs"]) / 1e6,
                 "share": share, "alg_gb": None, "rate": None, "frac": None, "event_ms": None}
            if any(k is c for c in chosen.values()) and roof.get("alg_bytes_per_launch"):
                r["alg_gb"] = roof["alg_bytes_per_launch"] / 1e9
                r["rate"] = r["alg_gb"] / roof_ms * 1e3  # GB/s over ALL kernels the roofline names (their sum is one pass)
                r["frac"] = r["rate"] / PEAK
                r["event_ms"] = roof.get("kernel_ms")
            out.append(r)
    return out


def table(tag):
    lines = [f"<!-- tools/kernel_table.py {tag}: derived from profiles/{tag}/kernel_stats_bench_*.csv and bench_*_under_rocprof.json -->",
             "| bench command | kernel (>= 1 % of the command's GPU time) | calls | avg ms (rocprofv3) | share | algorithmic GB / launch | GB/s | of 8 TB/s | HIP-event ms (bench.py) |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in rows(tag):
        f = lambda v, fmt: "" if v is None else format(v, fmt)
        lines.append(f"| {r['cmd']} | `{r['kernel']}` | {r['calls']} | {r['avg_ms']:.3f} | {100 * r['share']:.1f} % | {f(r['alg_gb'], '.2f')} | "
                     f"{f(r['rate'], '.0f')} | {f(r['frac'], '.3f')} | {f(r['event_ms'], '.3f')} |")
    lines.append(f"<!-- end of tools/kernel_table.py {tag} -->")
    return "\n".join(lines)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r5"
    t = table(tag)
    print(t)
    if "--update" in sys.argv:  # rewrite the block between the markers in both documents
        for rel in ("DESIGN.md", "profiles/README.md"):
            f = ROOT / rel
            txt = f.read_text()
            a, b = txt.index(f"<!-- tools/kernel_table.py {tag}:"), txt.index(f"<!-- end of tools/kernel_table.py {tag} -->")
            f.write_text(txt[:a] + t + txt[b + len(f"<!-- end of tools/kernel_table.py {tag} -->"):])
    if "--check" in sys.argv:
        bad = [p for p in ("DESIGN.md", "profiles/README.md") if t not in (ROOT / p).read_text()]
        if bad:
            print("NOT quoted verbatim in: " + ", ".join(bad), file=sys.stderr)
            sys.exit(1)


if __name__ == "__main__":
    main()
