#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of libsimmr_hip.so, as the compiler reports them
(`-Rpass-analysis=kernel-resource-usage`, gfx950 cross-compile: runs without a GPU).

    python3 tools/resource_usage.py                 # table on stdout
    python3 tools/resource_usage.py --json out.json # the same as JSON

tests/test_resource_guard.py asserts the documented figures of the kernels the bench lines run against this."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "simmr_amd" / "csrc"
FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill",
          "SGPRs Spill": "sgpr_spill"}


def collect(extra_flags=()):
    """[{name (demangled), vgpr, scratch, occupancy, lds, ...}] for engine.hip compiled with the Makefile's flags."""
    flags = subprocess.run(["make", "-s", "-C", str(CSRC), "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    cmd = flags + list(extra_flags) + ["-c", "engine.hip", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, cwd=str(CSRC), capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("compile failed:\n" + p.stderr[-4000:])
    kernels, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            kernels.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1) in FIELDS:
            cur[FIELDS[m.group(1)]] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in kernels),
                           capture_output=True, text=True, check=True).stdout.splitlines()
    for k, n in zip(kernels, names):
        k["name"] = re.sub(r"^void ", "", n)
    return kernels


def main():
    ks = collect()
    if "--json" in sys.argv:
        Path(sys.argv[sys.argv.index("--json") + 1]).write_text(json.dumps(ks, indent=1))
    print(f"{len(ks)} kernels")
    print(f"{'VGPR':>5} {'AGPR':>5} {'scr':>5} {'occ':>4} {'LDS':>7}  kernel")
    for k in sorted(ks, key=lambda k: k["name"]):
        print(f"{k['vgpr']:>5} {k['agpr']:>5} {k['scratch']:>5} {k['occupancy']:>4} {k['lds']:>7}  {k['name'][:200]}")


if __name__ == "__main__":
    main()
