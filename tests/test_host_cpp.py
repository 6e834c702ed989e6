"""C++ host layer (simmr_amd/host): FASTA ingest + normalisation, genome TSV,
FASTQ header interpolation, metadata float formatting, CLI surface — CPU only.
Mirrors the reference's genome_tests.rs and the formats of fastq.rs / files.rs."""
import ctypes as C
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "simmr_amd" / "host"
GOLDEN = Path(__file__).parent / "golden"
DEFAULT_FMT = ("@{:read_id:}|{:genome_id:}/{:pair:} metadata:sid={:sequence_id:}|sp={:start_position:}"
               "|ep={:end_position:}|rc={:reverse_complement:}")


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", str(HOST), "libsimmr_host.so"])
    import os
    # SIMMR_HOST_LIB: another build of the same sources (a -fsanitize=address,undefined build for sanitizer runs)
    lib = C.CDLL(os.environ.get("SIMMR_HOST_LIB") or str(HOST / "libsimmr_host.so"))
    for f in ("simmr_host_normalize", "simmr_host_format_f64", "simmr_host_format_header",
              "simmr_host_load_fasta", "simmr_host_parse_genome_file"):
        getattr(lib, f).restype = C.c_void_p
    lib.simmr_host_format_f64.argtypes = [C.c_double]
    lib.simmr_host_normalize.argtypes = [C.c_char_p, C.c_uint64]
    lib.simmr_host_format_header.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64,
                                             C.c_uint64, C.c_int, C.c_int]
    lib.simmr_host_load_fasta.argtypes = [C.c_char_p, C.c_int]
    lib.simmr_host_parse_genome_file.argtypes = [C.c_char_p]
    lib.simmr_host_free.argtypes = [C.c_void_p]

    def s(ptr):
        v = C.string_at(ptr).decode()
        lib.simmr_host_free(ptr)
        return v
    lib.s = s
    return lib


def test_genome_from_fasta_reference_unit_test(host):
    # genome_tests.rs:7-20 on the reference's own fixture (data copied to tests/golden)
    out = host.s(host.simmr_host_load_fasta(str(GOLDEN / "sample.fna").encode(), 0)).split("\n")
    assert out[0] == "2\t320"
    h1 = out[1].split("\t")
    h2 = out[2].split("\t")
    assert (h1[0], h1[1], h2[0], h2[1]) == ("header1", "160", "header2", "160")
    assert h1[3].startswith("AGCTTTTCATTCTGACTGCAACGGGCAATATGTCTCTG") and len(h1[3]) == 160
    # --contiguous: one 'whole genome' sequence, 'N' after every record, size excludes the separators
    out = host.s(host.simmr_host_load_fasta(str(GOLDEN / "sample.fna").encode(), 1)).split("\n")
    assert out[0] == "1\t320"
    w = out[1].split("\t")
    assert w[0] == "whole genome" and w[1] == "320" and w[2] == "322"
    assert w[3][160] == "N" and w[3][321] == "N" and w[3][:160] == h1[3]


def test_normalize_needletail_semantics(host):
    raw = b"acgtn ACGTN-\n.~uUxRyY*\r\n\tGG"
    assert host.s(host.simmr_host_normalize(raw, len(raw))) == "ACGTNACGTN---TTNNNNNGG"


def test_fasta_edge_cases(host, tmp_path):
    p = tmp_path / "a.fna"
    p.write_bytes(b">id one two\r\nACGT\r\nacgt\r\n>empty\n>last\nNN--")
    out = host.s(host.simmr_host_load_fasta(str(p).encode(), 0)).split("\n")
    assert out[0] == "3\t12"
    assert out[1].split("\t")[:3] == ["id one two", "8", "8"] and out[1].endswith("ACGTACGT")
    assert out[2].split("\t")[:3] == ["empty", "0", "0"]
    assert out[3].split("\t") == ["last", "4", "4", "NN--"]
    assert host.s(host.simmr_host_load_fasta(str(tmp_path / "missing").encode(), 0)).startswith("ERR\t")
    (tmp_path / "bad").write_bytes(b"ACGT\n")
    assert host.s(host.simmr_host_load_fasta(str(tmp_path / "bad").encode(), 0)).startswith("ERR\t")


def test_genome_file_variants(host, tmp_path):
    p = tmp_path / "g.tsv"
    p.write_text("path\tid\tabundance\n/a/b.fna\tg1\t0.25\n/c.fna\t\t\n")
    assert host.s(host.simmr_host_parse_genome_file(str(p).encode())) == "/a/b.fna\tg1\t0.25\n/c.fna\t<none>\t<none>\n"
    p.write_text("abundance\tgenome_id\tfilepath\n1e-3\tx\t/z.fna\n")  # any column order, serde aliases
    assert host.s(host.simmr_host_parse_genome_file(str(p).encode())) == "/z.fna\tx\t0.001\n"
    p.write_text("/plain/one.fna\n/plain/two.fna\n")  # plain list (extension; the reference mis-detects it)
    assert host.s(host.simmr_host_parse_genome_file(str(p).encode())) == "/plain/one.fna\t<none>\t<none>\n/plain/two.fna\t<none>\t<none>\n"


@pytest.mark.parametrize("v,s", [(100.0, "100"), (20.0, "20"), (33.333333333333336, "33.333333333333336"),
                                 (0.1, "0.1"), (1e-7, "0.0000001"), (1.5e21, "1500000000000000000000"),
                                 (0.015625, "0.015625"), (2.5, "2.5"), (1 / 3, "0.3333333333333333")])
def test_f64_display_like_rust(host, v, s):
    assert host.s(host.simmr_host_format_f64(v)) == s


def test_header_interpolation(host):
    h = host.s(host.simmr_host_format_header(DEFAULT_FMT.encode(), b"abc123", 7, b"NC_000913.3 Escherichia coli", 10,
                                             160, 0, 1))
    assert h == "@7|abc123/1 metadata:sid=NC_000913.3 Escherichia coli|sp=10|ep=160|rc=f"
    h = host.s(host.simmr_host_format_header(b"@{:read_id:}/{:pair:} {:read_id:}", b"g", 5, b"s", 9, 3, 1, 2))
    assert h == "@5/2 5"


def test_cli_surface():
    subprocess.check_call(["make", "-s", "-C", str(HOST), "simmr-hip"])
    exe = str(HOST / "simmr-hip")
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--genome", "--genome-file", "--output", "--num-reads", "--read-length", "--read-length-std",
                 "--insert-size", "--mean-phred-score", "--error-profile", "--abundance-profile", "--custom-profile",
                 "--with-ani", "--read-header-format", "--seed", "--size-adjusted", "--contiguous"):
        assert flag in r.stdout, flag
    assert subprocess.run([exe, "--output", "x"], capture_output=True).returncode == 2       # genomes group required
    assert subprocess.run([exe, "--genome", "a"], capture_output=True).returncode == 2       # --output required
    assert subprocess.run([exe, "--genome", "a", "--output", "x", "--error-profile", "ont"], capture_output=True).returncode == 2
    assert subprocess.run([exe, "--genome", "/nonexistent.fna", "--output", "/tmp/x.fq"], capture_output=True).returncode == 1
