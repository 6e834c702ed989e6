#!/bin/bash
# One GPU's share of BASELINE config 5 through the CLI: 64 genomes x 10 Mbp with abundances ~ 1 / (g + 1) (custom abundance
# profile), a simmrd-shaped long-read model (k = 7, every 7-mer listed), 50 M / 8 = 6.25 M long reads of ~N(20000, 4000)
# bases (about 250 GB of FASTQ), generated range by range and drained to /dev/null.
# usage: tools/c5_cli_run.sh [reads] [extra simmr-hip flags, e.g. "--rng philox"]
reads="${1:-6250000}"; extra="${2:-}"
make -s -C simmr_amd/host
if [ ! -f /tmp/c5_genomes.tsv ]; then
python3 - <<'PY'
import numpy as np
from simmr_amd import model_io
lines = ["path\tid\tabundance"]
for g in range(64):
    rng = np.random.default_rng(1000 + g)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 10_000_000)]
    with open(f"/tmp/c5_g{g}.fna", "wb") as f:
        f.write(f">c5_genome{g}_contig0\n".encode())
        rows = seq.reshape(-1, 80)
        out = np.empty((rows.shape[0], 81), dtype=np.uint8); out[:, :80] = rows; out[:, 80] = 10
        f.write(out.tobytes())
    lines.append(f"/tmp/c5_g{g}.fna\tc5-genome-{g}\t{1.0 / (g + 1)}")
open("/tmp/c5_genomes.tsv", "w").write("\n".join(lines) + "\n")
open("/tmp/c5_model.bin", "wb").write(model_io.synthetic_long_model(kmer_size=7, n_positions=1000, seed=1, n_kmers=4 ** 7,
                                                                   read_length_mean=20000.0, read_length_std=4000.0))
PY
echo "genomes and model written"
fi
args="--genome-file /tmp/c5_genomes.tsv --output /dev/null --num-reads $reads --seed 42 --error-profile custom-long --custom-profile /tmp/c5_model.bin --abundance-profile custom --per-read-lengths --uniform-start $extra"
t0=$(date +%s%N)
timeout -k 10 500 simmr_amd/host/simmr-hip $args
rc=$?
echo "exit=$rc reads=$reads $extra wall_ms=$(( ($(date +%s%N) - t0) / 1000000 ))"
