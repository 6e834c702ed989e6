#!/bin/bash
# End to end through the CLI on short reads: N reads of 150 bp PE on a 50 Mbp FASTA, FASTQ + TSVs to tmpfs / /dev/null.
# usage: tools/cli_short_run.sh [reads] [extra simmr-hip flags, e.g. "--rng philox"] ; prints one line per error profile (process start to exit).
reads="${1:-10000000}"; extra="${2:-}"
make -s -C simmr_amd/host
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(3)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50_000_000)]
with open("/tmp/cli_genome.fna", "wb") as f:
    f.write(b">synth_50M\n")
    lines = seq.reshape(-1, 80)
    out = np.empty((lines.shape[0], 81), dtype=np.uint8); out[:, :80] = lines; out[:, 80] = 10
    f.write(out.tobytes())
open("/tmp/cli_genomes.tsv", "w").write("path\tid\n/tmp/cli_genome.fna\tcli-genome\n")
PY
for prof in perfect-short minimal-short; do
  for dst in /dev/null /dev/shm/cli_out.fastq; do
    t0=$(date +%s%N)
    timeout -k 10 300 simmr_amd/host/simmr-hip --genome-file /tmp/cli_genomes.tsv --output $dst --num-reads $reads --seed 42 --error-profile $prof $extra > /tmp/cli_run.log 2>&1
    rc=$?
    sz=$(stat -c %s $dst 2>/dev/null || echo 0)
    echo "profile=$prof $extra output=$dst exit=$rc reads=$reads bytes=$sz wall_ms=$(( ($(date +%s%N) - t0) / 1000000 ))"
    rm -f /dev/shm/cli_out.fastq*
  done
done
