#!/bin/bash
# BASELINE config 3 at the reference's own gamma(20000, 15000) through the CLI: 10 M long reads (about 400 GB of FASTQ)
# generated range by range and drained to /dev/null.  usage: tools/c3_cli_run.sh [reads] [chunk-reads] [extra simmr-hip flags, e.g. "--rng philox"]
reads="${1:-10000000}"; chunk="${2:-}"; extra="${3:-}"
make -s -C simmr_amd/host
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(2)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 100_000_000)]
with open("/tmp/c3_genome.fna", "wb") as f:
    f.write(b">synth_100M\n")
    lines = seq.reshape(-1, 80)
    out = np.empty((lines.shape[0], 81), dtype=np.uint8); out[:, :80] = lines; out[:, 80] = 10
    f.write(out.tobytes())
open("/tmp/c3_genomes.tsv", "w").write("path\tid\n/tmp/c3_genome.fna\tc3-genome\n")
PY
echo "genome written"
args="--genome-file /tmp/c3_genomes.tsv --output /dev/null --num-reads $reads --seed 42 --error-profile minimal-long --per-read-lengths"
[ -n "$chunk" ] && args="$args --device-chunk-reads $chunk"
args="$args $extra"
t0=$(date +%s%N)
timeout -k 10 500 simmr_amd/host/simmr-hip $args
rc=$?
echo "exit=$rc reads=$reads $extra wall_ms=$(( ($(date +%s%N) - t0) / 1000000 ))"
