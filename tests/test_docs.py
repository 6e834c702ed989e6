"""The documents quote what the profile files say (hand-copied figures had drifted)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_kernel_table_is_quoted_verbatim():
    """DESIGN.md section 4 and profiles/README.md carry the table `tools/kernel_table.py r5` derives from
    profiles/r5/kernel_stats_bench_*.csv and bench_*_under_rocprof.json, character for character."""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_table.py"), "r5", "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("| default |") >= 3 and "k_emit_text_lines" in r.stdout


def test_every_file_the_round5_section_names_exists():
    import re
    text = (ROOT / "profiles" / "README.md").read_text()
    sec = text[text.index("## Round 5"):text.index("## Round 4")]
    names = set(re.findall(r"`(?:r5/)?([A-Za-z0-9_]+\.(?:txt|log|csv|json|md))`", sec))
    missing = [n for n in names if not (ROOT / "profiles" / "r5" / n).exists()]
    assert not missing, missing
