"""End-to-end on the GPU box: the simmr-hip CLI (C++ host over the C ABI) must
write the FASTQ / metadata TSV the reference would write for the same seed —
expected bytes are built here from the CPU oracle plus a restatement of
fastq.rs:32-121 and files.rs:100-134."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from simmr_amd import MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectShortErrorProfile
from tests import _oracle, _synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "simmr_amd" / "host" / "simmr-hip"
FMT = ("@{:read_id:}|{:genome_id:}/{:pair:} metadata:sid={:sequence_id:}|sp={:start_position:}"
       "|ep={:end_position:}|rc={:reverse_complement:}")


def fastq_of(d, n_reads, names, genome_id, paired, fmt=FMT):
    out = bytearray()
    for r in range(n_reads):
        rc = bool(d["flags"][r] & 1)
        h = fmt
        for k, v in (("{:genome_id:}", genome_id), ("{:read_id:}", str(int(d["read_id"][r]))),
                     ("{:sequence_id:}", names[int(d["contig"][r])]), ("{:start_position:}", str(int(d["start"][r]))),
                     ("{:end_position:}", str(int(d["end"][r]))), ("{:reverse_complement:}", "t" if rc else "f"),
                     ("{:pair:}", "2" if (paired and r & 1) else "1")):
            h = h.replace(k, v)
        a, b = int(d["seq_off"][r]), int(d["seq_off"][r + 1])
        out += h.encode() + b"\n" + d["seq"][a:b].tobytes() + b"\n+\n" + d["qual"][a:b].tobytes() + b"\n"
    return bytes(out)


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "simmr_amd" / "host")])
    d = tmp_path_factory.mktemp("cli")
    specs = [([120_000, 300, 40_000], 21, ["chr1 test genome", "tiny", "plasmid pX"]), ([90_000], 22, ["only"])]
    genomes = []
    for gi, (lens, seed, names) in enumerate(specs):
        contigs = _synth.synthetic_contigs(lens, seed)
        contigs[0] = contigs[0].copy()
        contigs[0][1000:1100] = ord("N")
        _synth.write_fasta(d / f"g{gi}.fna", contigs, names)
        genomes.append((contigs, names))
    (d / "genomes.tsv").write_text("path\tid\n" + "".join(f"{d}/g{gi}.fna\tgenome{gi}\n" for gi in range(2)))
    return d, genomes


@pytest.mark.parametrize("profile,cls", [("perfect-short", PerfectShortErrorProfile), ("minimal-short", MinimalShortErrorProfile)])
def test_cli_pe_fastq_bytes(workdir, oracle, profile, cls):
    d, genomes = workdir
    out = d / f"{profile}.fq"
    subprocess.check_call([str(EXE), "--genome-file", str(d / "genomes.tsv"), "--output", str(out), "--num-reads", "3001",
                           "--seed", "42", "--error-profile", profile])
    expected = bytearray()
    id_base = 0
    for gi, (contigs, names) in enumerate(genomes):
        keep = [i for i, c in enumerate(contigs) if c.size > 450]  # main.rs:117-162 size filter
        g = _oracle.HostGenome([contigs[i] for i in keep])
        reads = 1501  # uniform: ceil(3001 / 2)
        o = _oracle.simulate_pe(oracle, g, cls().pod(), reads, 42, read_id_base=id_base, qual_offset=33)
        expected += fastq_of(o.trimmed(), o.n_reads, [names[i] for i in keep], f"genome{gi}", True)
        id_base += reads // 2
    assert out.read_bytes() == bytes(expected)
    meta = (d / f"{profile}.fq.tsv").read_text().split("\n")
    assert meta[0] == "genome_id\tfilepath\tnum_reads\tabundance"
    assert meta[1] == f"genome0\t{d}/g0.fna\t1501\t50" and meta[2] == f"genome1\t{d}/g1.fna\t1501\t50"


def test_cli_custom_short(workdir, oracle):
    from simmr_amd import CustomShortErrorProfile
    from tests import _model
    d, genomes = workdir
    blob = _model.synthetic_short_model()
    (d / "model.bin").write_bytes(blob)
    out = d / "custom.fq"
    subprocess.check_call([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(out), "--num-reads", "1200",
                           "--seed", "5", "--error-profile", "custom-short", "--custom-profile", str(d / "model.bin"),
                           "--read-header-format", "@{:read_id:}/{:pair:} sp={:start_position:} ep={:end_position:}"])
    contigs, names = genomes[1]
    prof = CustomShortErrorProfile(blob)
    o = _oracle.simulate_pe(oracle, _oracle.HostGenome(contigs), prof.pod(), 1200, 5, qual_offset=33)
    exp = fastq_of(o.trimmed(), o.n_reads, names, "x", True, fmt="@{:read_id:}/{:pair:} sp={:start_position:} ep={:end_position:}")
    assert out.read_bytes() == exp


def test_cli_long_fastq_bytes(workdir, oracle):
    d, genomes = workdir
    out = d / "long.fq"
    subprocess.check_call([str(EXE), "--genome-file", str(d / "genomes.tsv"), "--output", str(out), "--num-reads", "25",
                           "--seed", "7", "--error-profile", "minimal-long", "--abundance-profile", "exact",
                           "--read-header-format", "@r{:read_id:} {:genome_id:} {:sequence_id:} {:start_position:}-{:end_position:}"])
    hosts, names_all = [], []
    for contigs, names in genomes:
        keep = [i for i, c in enumerate(contigs) if c.size > 20000]
        hosts.append(_oracle.HostGenome([contigs[i] for i in keep]))
        names_all.append([names[i] for i in keep])
    o = _oracle.simulate_long(oracle, hosts, [25, 25], MinimalLongErrorProfile().pod(), 7, qual_offset=33)
    dd = o.trimmed()
    expected = bytearray()
    for r in range(o.n_reads):
        g = int(dd["genome"][r])
        a, b = int(dd["seq_off"][r]), int(dd["seq_off"][r + 1])
        h = f"@r{int(dd['read_id'][r])} genome{g} {names_all[g][int(dd['contig'][r])]} {int(dd['start'][r])}-{int(dd['end'][r])}"
        expected += h.encode() + b"\n" + dd["seq"][a:b].tobytes() + b"\n+\n" + dd["qual"][a:b].tobytes() + b"\n"
    assert out.read_bytes() == bytes(expected)


@pytest.mark.parametrize("rng", ["reference", "philox"])
def test_cli_custom_long_with_abundances(workdir, oracle, rng):
    """(rng = philox: `--rng philox`, the splice's draws from Philox counters — the file is then the mode's specification,
    oracle/custom.c: orc_custom_simulate_errors_philox, framed as FASTQ.)
    BASELINE config 5 in small: a simmrd long-read model, a genome TSV with abundances, the long-read path
    (`custom-long` is an extension of this CLI: the reference's enum has no value that reaches
    CustomShortErrorProfile::simulate_errors, cli.rs:62-70, main.rs:30-33)."""
    from simmr_amd import CustomShortErrorProfile
    from simmr_amd.profiles import CustomAbundanceProfile
    from tests import _model
    d, genomes = workdir
    blob = _model.synthetic_long_model(kmer_size=6, n_positions=400, seed=8, n_kmers=4 ** 6, lengths=(1500, 6000, 100))
    (d / "long_model.bin").write_bytes(blob)
    (d / "abund.tsv").write_text("path\tid\tabundance\n" + f"{d}/g0.fna\tgA\t0.7\n{d}/g1.fna\tgB\t0.3\n")
    out = d / "custom_long.fq"
    fmt = "@{:read_id:} {:genome_id:}|{:sequence_id:}|{:start_position:}|{:end_position:}"
    subprocess.check_call([str(EXE), "--genome-file", str(d / "abund.tsv"), "--output", str(out), "--num-reads", "41",
                           "--seed", "19", "--error-profile", "custom-long", "--custom-profile", str(d / "long_model.bin"),
                           "--abundance-profile", "custom", "--read-header-format", fmt] + (["--rng", "philox"] if rng == "philox" else []))
    from simmr_amd import _abi
    prof = CustomShortErrorProfile(blob, _abi.RNG_PHILOX if rng == "philox" else _abi.RNG_REFERENCE)
    required = 2 * 3750  # custom_short.rs:535-538 with the model's means (3750, 0)
    hosts, names_all = [], []
    for contigs, names in genomes:
        keep = [i for i, c in enumerate(contigs) if c.size > required]
        hosts.append(_oracle.HostGenome([contigs[i] for i in keep]))
        names_all.append([names[i] for i in keep])
    counts = [n for n, _ in CustomAbundanceProfile([0.7, 0.3]).determine_abundances(41, 2)]
    assert counts == [29, 13]
    o = _oracle.simulate_long(oracle, hosts, counts, prof.pod(), 19, qual_offset=33)
    dd = o.trimmed()
    expected = bytearray()
    for r in range(o.n_reads):
        g = int(dd["genome"][r])
        a, b = int(dd["seq_off"][r]), int(dd["seq_off"][r + 1])
        h = f"@{int(dd['read_id'][r])} {'gA' if g == 0 else 'gB'}|{names_all[g][int(dd['contig'][r])]}|{int(dd['start'][r])}|{int(dd['end'][r])}"
        expected += h.encode() + b"\n" + dd["seq"][a:b].tobytes() + b"\n+\n" + dd["qual"][a:b].tobytes() + b"\n"
    assert out.read_bytes() == bytes(expected)
    meta = (d / "custom_long.fq.tsv").read_text().split("\n")
    assert meta[1] == f"gA\t{d}/g0.fna\t29\t0.7" and meta[2] == f"gB\t{d}/g1.fna\t13\t0.3"
    # a short-read model is refused for custom-long, as a long-read model is for custom-short (main.rs:30-33)
    (d / "short_model.bin").write_bytes(_model.synthetic_short_model())
    r = subprocess.run([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(d / "x.fq"), "--error-profile", "custom-long",
                        "--custom-profile", str(d / "short_model.bin")], capture_output=True)
    assert r.returncode != 0 and b"short reads" in r.stderr
    r = subprocess.run([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(d / "x.fq"), "--error-profile", "custom-short",
                        "--custom-profile", str(d / "long_model.bin")], capture_output=True)
    assert r.returncode != 0 and b"long reads" in r.stderr


def test_cli_contiguous(workdir, oracle):
    d, genomes = workdir
    out = d / "contig.fq"
    subprocess.check_call([str(EXE), "--genome", str(d / "g0.fna"), "--output", str(out), "--num-reads", "400",
                           "--seed", "3", "--contiguous", "--read-header-format", "@{:read_id:}/{:pair:} {:sequence_id:}"])
    contigs, _ = genomes[0]
    whole = np.concatenate([np.concatenate([c, np.frombuffer(b"N", dtype=np.uint8)]) for c in contigs])
    g = _oracle.HostGenome([whole], sizes=[sum(c.size for c in contigs)])
    o = _oracle.simulate_pe(oracle, g, PerfectShortErrorProfile().pod(), 400, 3, qual_offset=33)
    exp = fastq_of(o.trimmed(), o.n_reads, ["whole genome"], "x", True, fmt="@{:read_id:}/{:pair:} {:sequence_id:}")
    assert out.read_bytes() == exp


def test_cli_device_and_host_fastq_agree(workdir):
    """The FASTQ is framed on the device by default; --host-fastq keeps the C++ restatement of
    fastq.rs as a second, independently written writer.  Same bytes, also for an odd template."""
    d, _ = workdir
    fmt = "@{:read_id:}/{:pair:} {:genome_id:}|{:sequence_id:} {:start_position:}..{:end_position:} {:reverse_complement:}{:"
    outs = []
    for extra in ([], ["--host-fastq"]):
        out = d / ("agree%d.fq" % len(extra))
        subprocess.check_call([str(EXE), "--genome-file", str(d / "genomes.tsv"), "--output", str(out), "--num-reads", "2000",
                               "--seed", "3", "--error-profile", "minimal-short", "--read-header-format", fmt] + extra)
        outs.append(out.read_bytes())
    assert len(outs[0]) > 600_000 and outs[0] == outs[1]


def test_cli_device_and_host_normalize_agree(workdir, tmp_path):
    """FASTA bodies are normalised, size-filtered and packed on the device by default; --host-normalize keeps the
    C++ restatement of genome.rs:89-162 + needletail normalize.  Same output, also for a messy FASTA."""
    d, _ = workdir
    rng = np.random.default_rng(4)
    letters = np.frombuffer(b"ACGTacgtNnRYuU.-", dtype=np.uint8)
    seq = letters[rng.choice(letters.size, 61_000, p=np.r_[np.full(4, 0.22), np.full(12, 0.01)])].tobytes()
    messy = tmp_path / "messy.fna"
    with open(messy, "wb") as f:
        f.write(b">first record with spaces\r\n")
        for i in range(0, 60_000, 70):
            f.write(seq[i:i + 70] + b"\r\n")
        f.write(b">short one\n" + seq[60_000:60_200] + b"\n\n>third\n" + seq[200:40_200] + b"\n")
    tsv = tmp_path / "g.tsv"
    tsv.write_text(f"path\tid\n{messy}\tmessy\n{d}/g1.fna\tgenome1\n")
    for extra_common in ([], ["--contiguous"]):
        outs = []
        for extra in ([], ["--host-normalize"]):
            out = tmp_path / ("norm%d%d.fq" % (len(extra_common), len(extra)))
            subprocess.check_call([str(EXE), "--genome-file", str(tsv), "--output", str(out), "--num-reads", "3000", "--seed", "9",
                                   "--error-profile", "minimal-short"] + extra_common + extra)
            outs.append(out.read_bytes())
        assert len(outs[0]) > 900_000 and outs[0] == outs[1]


def test_cli_device_chunks_do_not_change_the_output(workdir):
    """--device-chunk-reads: a run generated range by range (here a thousand, 334 and 7 reads at a time — the production
    default is what fits the free device memory) writes the same FASTQ and metadata as the run generated in one pass:
    perfect-short, minimal-short in both generators' modes, a custom short-read model (planned genome by genome),
    long reads with a seed (one run-wide length, simulate.rs:358) and per-read lengths, and a custom long-read model
    with abundances (BASELINE config 5's shape); with the device text and with the host writer."""
    from tests import _model
    d, _ = workdir
    (d / "long_model2.bin").write_bytes(_model.synthetic_long_model(kmer_size=6, n_positions=400, seed=8, n_kmers=4 ** 6, lengths=(1500, 6000, 100)))
    (d / "short_model2.bin").write_bytes(_model.synthetic_short_model())
    (d / "abund2.tsv").write_text("path\tid\tabundance\n" + f"{d}/g0.fna\tgA\t0.7\n{d}/g1.fna\tgB\t0.3\n")
    runs = [
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "5001", "--seed", "7", "--error-profile", "perfect-short"], ["1000", "334"]),
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "5001", "--seed", "7", "--error-profile", "minimal-short"], ["1000", "7"]),
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "3000", "--seed", "7", "--error-profile", "minimal-short", "--host-fastq"], ["1000"]),
        # (genome ids from the TSV: a genome given with --genome gets a random uuid per run, util.rs:124-129)
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "1200", "--seed", "3", "--error-profile", "custom-short",
          "--custom-profile", str(d / "short_model2.bin")], ["334"]),
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "61", "--seed", "11", "--error-profile", "minimal-long"], ["7"]),
        (["--genome-file", str(d / "genomes.tsv"), "--num-reads", "61", "--seed", "11", "--error-profile", "perfect-long",
          "--per-read-lengths", "--gamma", "3000,2500"], ["7", "20"]),
        (["--genome-file", str(d / "abund2.tsv"), "--num-reads", "41", "--seed", "19", "--error-profile", "custom-long",
          "--custom-profile", str(d / "long_model2.bin"), "--abundance-profile", "custom"], ["7"]),
    ]
    for i, (argv, chunks) in enumerate(runs):
        whole = d / f"chunk_ref_{i}.fq"
        subprocess.check_call([str(EXE), "--output", str(whole), "--device-chunk-reads", "1000000000"] + argv)
        assert whole.stat().st_size > 0
        for c in chunks:
            out = d / f"chunk_{i}_{c}.fq"
            subprocess.check_call([str(EXE), "--output", str(out), "--device-chunk-reads", c] + argv)
            assert out.read_bytes() == whole.read_bytes(), (argv, c)
            assert (d / f"chunk_{i}_{c}.fq.tsv").read_text() == (d / f"chunk_ref_{i}.fq.tsv").read_text()


def test_cli_several_engines_write_the_single_engine_file(workdir):
    """--devices a,b,...: the whole node behind the reference's one call (main.rs:180-206) — one engine per entry, the run's
    ranges dealt to them in turn on host threads, the text appended in range order.  On the one GPU of this box two and three
    engines on device 0 must write the bytes (and metadata) of the single-engine run: all profiles of the device-text path,
    one pass per engine and several, and the refusals."""
    from tests import _model
    d, _ = workdir
    (d / "long_model3.bin").write_bytes(_model.synthetic_long_model(kmer_size=6, n_positions=400, seed=8, n_kmers=4 ** 6, lengths=(1500, 6000, 100)))
    (d / "short_model3.bin").write_bytes(_model.synthetic_short_model())
    (d / "abund3.tsv").write_text("path\tid\tabundance\n" + f"{d}/g0.fna\tgA\t0.7\n{d}/g1.fna\tgB\t0.3\n")
    runs = [
        ["--genome-file", str(d / "genomes.tsv"), "--num-reads", "5001", "--seed", "7", "--error-profile", "perfect-short"],
        ["--genome-file", str(d / "genomes.tsv"), "--num-reads", "5001", "--seed", "7", "--error-profile", "minimal-short"],
        ["--genome-file", str(d / "genomes.tsv"), "--num-reads", "5001", "--seed", "7", "--error-profile", "minimal-short", "--rng", "philox-full"],
        ["--genome-file", str(d / "genomes.tsv"), "--num-reads", "1200", "--seed", "3", "--error-profile", "custom-short",
         "--custom-profile", str(d / "short_model3.bin")],
        ["--genome-file", str(d / "genomes.tsv"), "--num-reads", "61", "--seed", "11", "--error-profile", "perfect-long",
         "--per-read-lengths", "--gamma", "3000,2500"],
        ["--genome-file", str(d / "abund3.tsv"), "--num-reads", "41", "--seed", "19", "--error-profile", "custom-long",
         "--custom-profile", str(d / "long_model3.bin"), "--abundance-profile", "custom", "--rng", "philox"],
    ]
    for i, argv in enumerate(runs):
        one = d / f"dev_one_{i}.fq"
        subprocess.check_call([str(EXE), "--output", str(one), "--device", "0"] + argv)
        assert one.stat().st_size > 0
        for devices, chunk in (("0,0", None), ("0,0,0", "334" if i < 4 else "7")):
            out = d / f"dev_{i}_{devices.count(',')}.fq"
            extra = ["--device-chunk-reads", chunk] if chunk else []
            subprocess.check_call([str(EXE), "--output", str(out), "--devices", devices] + extra + argv)
            assert out.read_bytes() == one.read_bytes(), (argv, devices)
            assert (d / f"{out.name}.tsv").read_text() == (d / f"{one.name}.tsv").read_text()
    r = subprocess.run([str(EXE), "--output", str(d / "x.fq"), "--devices", "0,0", "--host-fastq"] + runs[0], capture_output=True)
    assert r.returncode != 0 and b"--host-fastq" in r.stderr
    r = subprocess.run([str(EXE), "--output", str(d / "x.fq"), "--devices", "0,99"] + runs[0], capture_output=True)
    assert r.returncode != 0 and b"cannot create engine" in r.stderr
    r = subprocess.run([str(EXE), "--output", str(d / "x.fq"), "--devices", "0,,1"] + runs[0], capture_output=True)
    assert r.returncode == 2


def test_cli_a_genome_without_a_pair_is_skipped(workdir, oracle):
    """A scope with zero units — a genome whose abundance share is below one pair, or --num-reads 1 — writes nothing and
    the run goes on to the next genome, as the reference's loop does (simulate.rs:179: num_reads / 2 pairs)."""
    from simmr_amd import CustomShortErrorProfile
    from simmr_amd.profiles import CustomAbundanceProfile
    from tests import _model
    d, genomes = workdir
    blob = _model.synthetic_short_model()
    (d / "model0.bin").write_bytes(blob)
    # genome A gets one read = no pair, genome B everything else (custom-short plans genome by genome)
    (d / "skew.tsv").write_text("path\tid\tabundance\n" + f"{d}/g0.fna\tgA\t0.001\n{d}/g1.fna\tgB\t0.999\n")
    counts = [n for n, _ in CustomAbundanceProfile([0.001, 0.999]).determine_abundances(600, 2)]
    assert counts[0] == 1
    fmt = "@{:read_id:}/{:pair:} {:genome_id:} sp={:start_position:}"
    out = d / "skew.fq"
    subprocess.check_call([str(EXE), "--genome-file", str(d / "skew.tsv"), "--output", str(out), "--num-reads", "600", "--seed", "5",
                           "--error-profile", "custom-short", "--custom-profile", str(d / "model0.bin"),
                           "--abundance-profile", "custom", "--read-header-format", fmt])
    contigs, names = genomes[1]
    o = _oracle.simulate_pe(oracle, _oracle.HostGenome(contigs), CustomShortErrorProfile(blob).pod(), counts[1], 5, qual_offset=33)
    assert out.read_bytes() == fastq_of(o.trimmed(), o.n_reads, names, "gB", True, fmt=fmt)
    # and a whole run without a pair
    for profile in ("perfect-short", "minimal-short"):
        one = d / f"one_{profile}.fq"
        r = subprocess.run([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(one), "--num-reads", "1", "--seed", "5",
                            "--error-profile", profile], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert not one.exists() or one.stat().st_size == 0


@pytest.mark.parametrize("profile,cls", [("minimal-short", MinimalShortErrorProfile), ("minimal-long", MinimalLongErrorProfile)])
@pytest.mark.parametrize("mode", ["philox", "philox-full"])
def test_cli_rng_philox(workdir, oracle, profile, cls, mode):
    """--rng philox (extension flag): the counter mode through the CLI.  The text straight from the plan (the TEXT form of
    the counter-mode kernel) and the --host-fastq path (columns in 16-byte read slots, closed up on the host by
    DeviceOut::to_host) write the same file, and that file is the mode's specification (oracle/philox.c) framed as
    fastq.rs:32-121 frames it.  Without the flag the reference's own streams are walked (the other CLI tests)."""
    from simmr_amd import _abi
    rng_mode = _abi.RNG_PHILOX_FULL if mode == "philox-full" else _abi.RNG_PHILOX  # (philox-full: the plan from counters too)
    d, genomes = workdir
    long_mode = profile.endswith("long")
    n = "40" if long_mode else "4001"
    extra = ["--per-read-lengths", "--gamma", "3000,2500"] if long_mode else []
    outs = []
    for tag, flags in (("dev", []), ("host", ["--host-fastq"]), ("chunk", ["--device-chunk-reads", "33"])):
        out = d / f"{mode}_{profile}_{tag}.fq"
        subprocess.check_call([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(out), "--num-reads", n, "--seed", "42",
                               "--error-profile", profile, "--rng", mode, "--read-header-format",
                               "@{:read_id:}/{:pair:} {:sequence_id:} {:start_position:}-{:end_position:} {:reverse_complement:}"] + extra + flags)
        outs.append(out.read_bytes())
    assert len(outs[0]) > 50_000 and outs[0] == outs[1] == outs[2]
    contigs, names = genomes[1]
    g = _oracle.HostGenome(contigs)
    if long_mode:
        pod = cls(gamma_mean=3000.0, gamma_std=2500.0, length_mode=_abi.LEN_PER_READ, rng_mode=rng_mode).pod()
        o = _oracle.simulate_long(oracle, [g], [40], pod, 42, qual_offset=33)
    else:
        o = _oracle.simulate_pe(oracle, g, cls(rng_mode=rng_mode).pod(), 4001, 42, qual_offset=33)
    exp = fastq_of(o.trimmed(), o.n_reads, names, "x", not long_mode,
                   fmt="@{:read_id:}/{:pair:} {:sequence_id:} {:start_position:}-{:end_position:} {:reverse_complement:}")
    assert outs[0] == exp
    r = subprocess.run([str(EXE), "--genome", str(d / "g1.fna"), "--output", str(d / "x.fq"), "--rng", "xorshift"], capture_output=True)
    assert r.returncode != 0
