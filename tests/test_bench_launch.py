"""`python3 bench.py --gpus N` must start itself: without a launcher in front of it the script
starts its N ranks as child processes, rank 0 prints ONE JSON line with the fields the contract asks for, and the exit
code is the children's.  Run here on CPU: gloo, and an oracle-backed stand-in for the engine (tests/bench_stub/)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(extra, timeout=420):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["PYTHONPATH"] = str(ROOT / "tests" / "bench_stub") + os.pathsep + env.get("PYTHONPATH", "")
    env["SIMMR_BENCH_STUB"] = "1"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--backend", "gloo", "--rehearse-device", "0", "--reads", "6000",
           "--genome-bases", "300000", "--steps", "2", "--warmup", "1", "--no-other-mode", "--layout", "compact"] + extra
    return subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=timeout)


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_2_in_plain_form_starts_its_own_ranks():
    r = _line(_run(["--gpus", "2"]))
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["world_size_seen"] == 2 and r["config"]["backend"].startswith("gloo")
    assert r["config"]["reads_per_gpu"] == 6000
    # the whole job's reads: both ranks' shards went through the all-reduce of the counters
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 12000) < 1e-6 * 12000
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r["roofline"])
    assert r["roofline"]["alg_bytes_per_launch"] > 0 and r["roofline"]["kernel_ms"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    assert "reference_toolchain" in r


@pytest.mark.timeout(600)
def test_long_read_line_carries_cpu_baseline():
    """BASELINE config 3's line (and any N): cpu_baseline from the oracle's simulate_long_reads, with the
    faithful-cost figure (the reference's per-read genome clones, simulate.rs:362-375) beside it."""
    r = _line(_run(["--gpus", "1", "--profile", "minimal-long", "--reads", "400", "--cpu-sample-long-reads", "200"]))
    assert r["n_gpus"] == 1 and r["config"]["world_size_seen"] == 1
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["faithful_cost"]["value"] > 0
    assert "simulate.rs:362-375" in cb["faithful_cost"]["sample"] and cb["single_thread_value"] > 0


def test_a_failing_rank_is_the_exit_code():
    p = _run(["--gpus", "2", "--layout", "slot16"], timeout=300)  # the stand-in refuses the slot layout
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
