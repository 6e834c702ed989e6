// host.cpp — implementation of simmr_host.hpp (no GPU code, no simulation
// arithmetic).  Each function cites the reference lines it mirrors.
#include "simmr_host.hpp"

#include "../csrc/custom_model.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <random>
#include <sstream>

namespace simmr_host {

// ------------------------------------------------------------------ genome.rs

std::string normalize(const std::string& raw) {
  // needletail 0.4.1 sequence::normalize(seq, iupac = false)
  std::string out;
  out.reserve(raw.size());
  for (unsigned char c : raw) {
    switch (c) {
      case 'A': case 'C': case 'G': case 'T': case 'N': case '-': out.push_back((char)c); break;
      case 'a': out.push_back('A'); break;
      case 'c': out.push_back('C'); break;
      case 'g': out.push_back('G'); break;
      case 't': case 'u': case 'U': out.push_back('T'); break;
      case '.': case '~': out.push_back('-'); break;
      case ' ': case '\t': case '\r': case '\n': break;  // whitespace and line endings are dropped
      default: out.push_back('N'); break;                // everything else is an N
    }
  }
  return out;
}

uint64_t generate_id() {
  // util.rs:124-129: Uuid::new_v4().as_u64_pair().0 — the version nibble (4) sits in bits 15..12
  static std::random_device rd;
  uint64_t r = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
  return (r & ~0xF000ULL) | 0x4000ULL;
}

std::string uuid_from_u64(uint64_t u) {
  char buf[32];
  snprintf(buf, sizeof buf, "%llx", (unsigned long long)u);
  return buf;
}

bool scan_fasta(const std::string& filepath, FastaRecords* out, std::string* err) {
  std::ifstream f(filepath, std::ios::binary);
  if (!f) { *err = "No such file or directory (os error 2)"; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  out->data = ss.str();
  const std::string& data = out->data;
  if (data.empty()) { *err = "Failed to read the first two bytes. Is the file empty?"; return false; }
  if (data.size() >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b) {
    *err = "compressed FASTA is not supported by this host layer";
    return false;
  }
  if (data[0] != '>') { *err = "Bad starting byte found, expected '>' (FASTA records only)"; return false; }
  size_t pos = 0;
  while (pos < data.size()) {
    // header line
    size_t eol = data.find('\n', pos);
    if (eol == std::string::npos) eol = data.size();
    std::string header = data.substr(pos + 1, eol - pos - 1);
    if (!header.empty() && header.back() == '\r') header.pop_back();
    // sequence lines until a line starting with '>'
    size_t p = std::min(eol + 1, data.size());
    size_t next = p;
    for (;;) {
      if (next >= data.size()) { next = data.size(); break; }
      if (data[next] == '>') break;
      size_t e = data.find('\n', next);
      if (e == std::string::npos) { next = data.size(); break; }
      next = e + 1;
    }
    out->ids.push_back(header);                          // genome.rs:112 record.id()
    out->body.emplace_back(p, next - p);
    pos = next;
  }
  return true;
}

bool Genome::from_fasta(const std::string& filepath, bool contiguous, Genome* out, std::string* err) {
  FastaRecords recs;
  if (!scan_fasta(filepath, &recs, err)) return false;
  std::vector<Seq> sequences;
  for (size_t c = 0; c < recs.ids.size(); c++) {
    Seq s;
    s.id = recs.ids[c];
    s.uuid = generate_id();                              // genome.rs:118
    s.seq = normalize(recs.data.substr(recs.body[c].first, recs.body[c].second));  // genome.rs:114 normalize(false)
    s.size = s.seq.size();
    sequences.push_back(std::move(s));
  }
  Genome g;
  g.uuid = uuid_from_u64(generate_id());                 // genome.rs:124,140
  g.filepath = filepath;
  g.contiguous = contiguous;
  uint64_t total = 0;
  for (const Seq& s : sequences) total += s.seq.size();
  g.size = total;
  if (contiguous) {                                      // genome.rs:121-137
    Seq whole;
    whole.id = "whole genome";
    whole.uuid = generate_id();
    for (const Seq& s : sequences) { whole.seq += s.seq; whole.seq.push_back('N'); }
    whole.size = total;                                  // the 'N' separators are NOT counted
    g.sequence.push_back(std::move(whole));
    g.num_seqs = 1;
  } else {
    g.num_seqs = sequences.size();
    g.sequence = std::move(sequences);
  }
  *out = std::move(g);
  return true;
}

// ------------------------------------------------------------------- files.rs

static std::vector<std::string> split_tabs(const std::string& line) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    size_t b = line.find('\t', a);
    if (b == std::string::npos) { out.push_back(line.substr(a)); break; }
    out.push_back(line.substr(a, b - a));
    a = b + 1;
  }
  return out;
}

bool parse_genome_file(const std::string& filepath, std::vector<GenomeRecord>* out, std::string* err) {
  std::ifstream f(filepath);
  if (!f) { *err = "Genome file does not exist"; return false; }  // files.rs:57-59
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (!line.empty()) lines.push_back(line);
  }
  out->clear();
  if (lines.empty()) return true;
  // files.rs:32-45 decides "simple" only when the first line is at most one
  // character long, so every real file is read with a header row (csv crate,
  // tab delimiter; serde aliases path|filepath, id|genome_id|uuid).  A first line
  // that names no path column is accepted as a plain list (extension).
  std::vector<std::string> head = split_tabs(lines[0]);
  int c_path = -1, c_uuid = -1, c_ab = -1;
  for (size_t i = 0; i < head.size(); i++) {
    if (head[i] == "filepath" || head[i] == "path") c_path = (int)i;
    else if (head[i] == "uuid" || head[i] == "id" || head[i] == "genome_id") c_uuid = (int)i;
    else if (head[i] == "abundance") c_ab = (int)i;
  }
  size_t first_row = 1;
  if (c_path < 0) { c_path = 0; c_uuid = 1; c_ab = 2; first_row = 0; }  // positional, no header
  for (size_t r = first_row; r < lines.size(); r++) {
    std::vector<std::string> cols = split_tabs(lines[r]);
    GenomeRecord rec;
    if ((size_t)c_path >= cols.size()) { *err = "genome file row " + std::to_string(r + 1) + " has no path"; return false; }
    rec.filepath = cols[c_path];
    if (c_uuid >= 0 && (size_t)c_uuid < cols.size() && !cols[c_uuid].empty()) rec.uuid = cols[c_uuid];
    if (c_ab >= 0 && (size_t)c_ab < cols.size() && !cols[c_ab].empty()) {
      char* end = nullptr;
      double v = strtod(cols[c_ab].c_str(), &end);
      if (end == cols[c_ab].c_str() || *end != 0) { *err = "invalid abundance '" + cols[c_ab] + "'"; return false; }
      rec.abundance = v;
    }
    out->push_back(rec);
  }
  return true;
}

std::string format_f64_display(double v) {
  if (v != v) return "NaN";
  if (isinf(v)) return v < 0 ? "-inf" : "inf";
  if (v == 0) return signbit(v) ? "-0" : "0";
  char buf[64];
  int prec = 1;
  for (; prec <= 17; prec++) {
    snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
    if (strtod(buf, nullptr) == v) break;
  }
  // buf = d.ddddde[+-]xx
  std::string s(buf);
  bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  size_t epos = s.find('e');
  int exp10 = atoi(s.c_str() + epos + 1);
  std::string digits;
  for (size_t i = 0; i < epos; i++) if (s[i] != '.') digits.push_back(s[i]);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  int nd = (int)digits.size();
  if (exp10 >= nd - 1) {
    out = digits + std::string(exp10 - (nd - 1), '0');
  } else if (exp10 >= 0) {
    out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
  } else {
    out = "0." + std::string(-exp10 - 1, '0') + digits;
  }
  return neg ? "-" + out : out;
}

bool write_metadata(const std::vector<MetadataRow>& rows, const std::string& output, std::string* err) {
  // files.rs:100-134
  remove(output.c_str());
  FILE* f = fopen(output.c_str(), "wb");
  if (!f) { *err = std::string("cannot open ") + output; return false; }
  fputs("genome_id\tfilepath\tnum_reads\tabundance\n", f);
  for (const MetadataRow& r : rows)
    fprintf(f, "%s\t%s\t%llu\t%s\n", r.genome_id.c_str(), r.filepath.c_str(), (unsigned long long)r.num_reads,
            format_f64_display(r.abundance).c_str());
  fclose(f);
  return true;
}

// ------------------------------------------------------------------- fastq.rs

static void replace_all(std::string& s, const char* pat, const std::string& with) {
  const size_t n = strlen(pat);
  size_t pos = 0;
  while ((pos = s.find(pat, pos)) != std::string::npos) {
    s.replace(pos, n, with);
    pos += with.size();
  }
}

std::string format_header(const std::string& header_format, const std::string& genome_id, uint32_t read_id,
                          const std::string& sequence_id, uint64_t start, uint64_t end, bool revcomp,
                          int pair) {
  // the chained String::replace calls of fastq.rs:34-56, in the same order
  std::string h = header_format;
  replace_all(h, "{:genome_id:}", genome_id);
  replace_all(h, "{:read_id:}", std::to_string(read_id));
  replace_all(h, "{:sequence_id:}", sequence_id);
  replace_all(h, "{:start_position:}", std::to_string(start));
  replace_all(h, "{:end_position:}", std::to_string(end));
  replace_all(h, "{:reverse_complement:}", revcomp ? "t" : "f");
  replace_all(h, "{:pair:}", pair == 1 ? "1" : "2");
  return h;
}

bool write_to_fastq(const std::string& genome_uuid, const Genome& genome, const HostReads& reads,
                    uint64_t first, uint64_t count, const std::string& output,
                    const std::string& header_format, bool append, std::string* err) {
  FILE* f = fopen(output.c_str(), append ? "ab" : "wb");
  if (!f) { *err = std::string("cannot open ") + output; return false; }
  std::vector<char> buf(1 << 20);
  setvbuf(f, buf.data(), _IOFBF, buf.size());
  for (uint64_t r = first; r < first + count; r++) {
    const uint64_t o = reads.seq_off[r];
    const bool rc = (reads.flags[r] & SIMMR_FLAG_REVCOMP) != 0;
    const uint64_t len = rc ? reads.start[r] - reads.end[r] : reads.end[r] - reads.start[r];
    // long reads are never pair 2; for pairs the mate is the parity of the read index
    const int pair = (reads.paired && (r & 1)) ? 2 : 1;
    const std::string& sid = genome.sequence[reads.contig[r]].id;
    std::string h = format_header(header_format, genome_uuid, reads.read_id[r], sid, reads.start[r],
                                  reads.end[r], rc, pair);
    fwrite(h.data(), 1, h.size(), f);
    fputc('\n', f);
    fwrite(reads.seq.data() + o, 1, len, f);
    fputs("\n+\n", f);
    fwrite(reads.qual.data() + o, 1, len, f);  // util::encode_quality_scores: +33, applied on the device
    fputc('\n', f);
  }
  fclose(f);
  return true;
}

// --------------------------------------------------------------- error profiles

static simmr_error_profile zero_pod() {
  simmr_error_profile p;
  memset(&p, 0, sizeof p);
  return p;
}
simmr_error_profile PerfectShortErrorProfile::pod() const {
  simmr_error_profile p = zero_pod();
  p.kind = SIMMR_PERFECT_SHORT; p.read_length = read_length; p.insert_size = insert_size;
  return p;
}
simmr_error_profile MinimalShortErrorProfile::pod() const {
  simmr_error_profile p = zero_pod();
  p.kind = SIMMR_MINIMAL_SHORT; p.read_length = read_length; p.insert_size = insert_size;
  p.mean_phred = mean_phred_score; p.read_length_std = read_length_std; p.insert_size_std = insert_size_std;
  return p;
}
simmr_error_profile MinimalLongErrorProfile::pod() const {
  simmr_error_profile p = zero_pod();
  p.kind = SIMMR_MINIMAL_LONG; p.mean_phred = mean_phred_score; p.length_mode = length_mode;
  p.long_start_mode = long_start_mode;
  // minimal_long.rs:64-69: shape = (mean / std_dev).powf(2.0); scale = std_dev.powf(2.0) / mean (f32)
  p.gamma_shape = powf(gamma_mean / gamma_std, 2.0f);
  p.gamma_scale = powf(gamma_std, 2.0f) / gamma_mean;
  return p;
}
simmr_error_profile PerfectLongErrorProfile::pod() const {
  simmr_error_profile p = MinimalLongErrorProfile::pod();
  p.kind = SIMMR_PERFECT_LONG;
  return p;
}

std::unique_ptr<CustomShortErrorProfile> CustomShortErrorProfile::from_path(const std::string& path, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *err = "No such file or directory (os error 2)"; return nullptr; }
  auto p = std::make_unique<CustomShortErrorProfile>();
  p->model.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  simmr::ModelHost m;
  if (!simmr::parse_model(p->model.data(), p->model.size(), &m, err)) return nullptr;
  p->read_length_mean = m.read_length_mean;
  p->insert_size_mean = m.insert_size_mean;
  p->is_long = m.is_long;
  return p;
}
simmr_error_profile CustomShortErrorProfile::pod() const {
  simmr_error_profile p = zero_pod();
  p.kind = SIMMR_CUSTOM;
  p.custom_model = model.data();
  p.custom_model_bytes = model.size();
  p.length_mode = length_mode;          // long-read models only (extensions --per-read-lengths / --uniform-start)
  p.long_start_mode = long_start_mode;
  return p;
}
uint16_t CustomShortErrorProfile::minimum_genome_size() const {
  const double v = 2.0 * read_length_mean + insert_size_mean;  // `as u16` saturates
  if (!(v == v) || v <= 0.0) return 0;
  return v >= 65535.0 ? 65535 : (uint16_t)v;
}

// ----------------------------------------------------------- abundance profiles

Abundances AbundanceProfile::adjust_for_size(const std::vector<Genome>& genomes, const Abundances& ra,
                                             uint64_t, bool) const {
  // uniform.rs:79-94 == custom.rs:80-95 (total_coverage is computed there but unused)
  double total_reads = 0.0, total_adjusts = 0.0;
  for (const auto& x : ra) total_reads += (double)x.first;
  for (size_t i = 0; i < genomes.size() && i < ra.size(); i++) total_adjusts += (double)genomes[i].size * ra[i].second;
  Abundances out;
  for (size_t i = 0; i < genomes.size() && i < ra.size(); i++)
    out.emplace_back((uint64_t)ceil(total_reads * ((ra[i].second * (double)genomes[i].size) / total_adjusts)), ra[i].second);
  return out;
}
Abundances UniformAbundanceProfile::determine_abundances(uint64_t total_reads, uint64_t num_genomes) const {
  const uint64_t per = (uint64_t)ceil((double)total_reads / (double)num_genomes);  // uniform.rs:28
  return Abundances(num_genomes, {per, 100.0 / (double)num_genomes});
}
Abundances ExactAbundanceProfile::determine_abundances(uint64_t total_reads, uint64_t num_genomes) const {
  return Abundances(num_genomes, {total_reads, 100.0 / (double)num_genomes});        // exact.rs:17-24
}
Abundances CustomAbundanceProfile::determine_abundances(uint64_t total_reads, uint64_t) const {
  double total = 0.0;                                                                // custom.rs:26-48
  for (double a : abundances) total += a;
  Abundances out;
  if (total < 0.99 || total > 1.01)
    for (double a : abundances) out.emplace_back((uint64_t)ceil((double)total_reads * (a / total)), a / total);
  else
    for (double a : abundances) out.emplace_back((uint64_t)ceil((double)total_reads * a), a);
  return out;
}

// ------------------------------------------------------------------------ cli.rs

std::string usage() {
  return "simmr-hip — simmr's read simulation on MI355X (same flags as simmr, cli.rs:93-220)\n"
         "  --genome <FILE>...            Filepath to a genome to use for simulations\n"
         "  --genome-file <FILE>          TSV of genome filepaths and metadata (path, uuid|id, abundance)\n"
         "  --output <FILE>               FASTQ output containing simulated reads (required)\n"
         "  --num-reads <N>               Number of reads to simulate [default: 1000]\n"
         "  --read-length <N>             Individual read length (nt) [default: 150]\n"
         "  --read-length-std <F>         Standard deviation of read lengths [default: 10]\n"
         "  --insert-size <N>             Insert size for PE reads (nt) [default: 150]\n"
         "  --mean-phred-score <N>        Average Phred quality score [default: 30]\n"
         "  --error-profile <P>           minimal-short | minimal-long | perfect-short | perfect-long | custom-short [default: perfect-short]\n"
         "                                (extension: custom-long = a simmrd long-read model through the long-read path,\n"
         "                                 which the reference's own enum cannot select, cli.rs:62-70)\n"
         "  --abundance-profile <P>       exact | uniform | custom [default: uniform]\n"
         "  --custom-profile <FILE>       Filepath to a custom error profile (simmrd model) for custom-short\n"
         "  --with-ani <N>                [not implemented, as in the reference]\n"
         "  --read-header-format <FMT>    header template ({:genome_id:} {:read_id:} {:pair:} {:sequence_id:} ...)\n"
         "  --seed <N>                    Random seed\n"
         "  --size-adjusted               Adjust by genome size\n"
         "  --contiguous                  Treat separate sequences in a genome as one contiguous sequence\n"
         "extensions: --device <N>  --devices <a,b,...>  --gamma <mean,std>  --per-read-lengths  --uniform-start  --host-fastq  --host-normalize  --device-chunk-reads <N>  --rng <reference|philox|philox-full>\n";
}

static bool parse_u64(const std::string& s, uint64_t max, uint64_t* out) {
  if (s.empty()) return false;
  char* end = nullptr;
  unsigned long long v = strtoull(s.c_str(), &end, 10);
  if (*end != 0 || s[0] == '-' || v > max) return false;
  *out = v;
  return true;
}

bool parse_cli_args(int argc, const char* const* argv, CliArgs* a, std::string* err, bool* help) {
  *help = false;
  for (int i = 1; i < argc; i++) {
    std::string arg = argv[i], val;
    bool has_val = false;
    size_t eq = arg.find('=');
    if (arg.rfind("--", 0) == 0 && eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); has_val = true; }
    auto need = [&](std::string* dst) -> bool {
      if (has_val) { *dst = val; return true; }
      if (i + 1 >= argc) { *err = "a value is required for '" + arg + "'"; return false; }
      *dst = argv[++i];
      return true;
    };
    std::string v;
    uint64_t u;
    if (arg == "--help" || arg == "-h") { *help = true; return true; }
    else if (arg == "--genome") { if (!need(&v)) return false; a->genome.push_back(v); }
    else if (arg == "--genome-file") { if (!need(&v)) return false; a->genome_file = v; }
    else if (arg == "--output") { if (!need(&v)) return false; a->output = v; }
    else if (arg == "--num-reads") { if (!need(&v) || !parse_u64(v, UINT64_MAX, &u)) { *err = "invalid value for --num-reads"; return false; } a->num_reads = u; }
    else if (arg == "--read-length") { if (!need(&v) || !parse_u64(v, 65535, &u)) { *err = "invalid value for --read-length"; return false; } a->read_length = (uint16_t)u; }
    else if (arg == "--read-length-std") { if (!need(&v)) return false; a->read_length_std = atof(v.c_str()); }
    else if (arg == "--insert-size") { if (!need(&v) || !parse_u64(v, 65535, &u)) { *err = "invalid value for --insert-size"; return false; } a->insert_size = (uint16_t)u; }
    else if (arg == "--mean-phred-score") { if (!need(&v) || !parse_u64(v, 255, &u)) { *err = "invalid value for --mean-phred-score"; return false; } a->mean_phred_score = (uint8_t)u; }
    else if (arg == "--error-profile") {
      if (!need(&v)) return false;
      if (v == "minimal-short") a->error_profile = ErrorProfileKind::MinimalShort;
      else if (v == "minimal-long") a->error_profile = ErrorProfileKind::MinimalLong;
      else if (v == "perfect-short") a->error_profile = ErrorProfileKind::PerfectShort;
      else if (v == "perfect-long") a->error_profile = ErrorProfileKind::PerfectLong;
      else if (v == "custom-short") a->error_profile = ErrorProfileKind::CustomShort;
      else if (v == "custom-long") a->error_profile = ErrorProfileKind::CustomLong;  // extension, see usage()
      else { *err = "invalid value '" + v + "' for '--error-profile'"; return false; }
    } else if (arg == "--abundance-profile") {
      if (!need(&v)) return false;
      if (v == "exact") a->abundance_profile = AbundanceProfileKind::Exact;
      else if (v == "uniform") a->abundance_profile = AbundanceProfileKind::Uniform;
      else if (v == "custom") a->abundance_profile = AbundanceProfileKind::Custom;
      else { *err = "invalid value '" + v + "' for '--abundance-profile'"; return false; }
    }
    else if (arg == "--custom-profile") { if (!need(&v)) return false; a->custom_profile = v; }
    else if (arg == "--with-ani") { if (!need(&v) || !parse_u64(v, 255, &u)) { *err = "invalid value for --with-ani"; return false; } a->with_ani = (uint8_t)u; }
    else if (arg == "--read-header-format") { if (!need(&v)) return false; a->read_header_format = v; }
    else if (arg == "--seed") { if (!need(&v) || !parse_u64(v, UINT64_MAX, &u)) { *err = "invalid value for --seed"; return false; } a->seed = u; }
    else if (arg == "--size-adjusted") a->size_adjusted = true;
    else if (arg == "--contiguous") a->contiguous = true;
    else if (arg == "--host-fastq") a->host_fastq = true;
    else if (arg == "--host-normalize") a->host_normalize = true;
    else if (arg == "--device-chunk-reads") { if (!need(&v) || !parse_u64(v, UINT64_MAX, &u) || u == 0) { *err = "invalid value for --device-chunk-reads"; return false; } a->device_chunk_reads = u; }
    else if (arg == "--devices") {
      if (!need(&v)) return false;
      a->devices.clear();
      size_t pos = 0;
      while (pos <= v.size()) {
        const size_t comma = std::min(v.find(',', pos), v.size());
        if (!parse_u64(v.substr(pos, comma - pos), 1023, &u)) { *err = "invalid value for --devices (a comma-separated list of device ordinals)"; return false; }
        a->devices.push_back((int)u);
        pos = comma + 1;
      }
      if (a->devices.empty() || a->devices.size() > 64) { *err = "invalid value for --devices"; return false; }
    }
    else if (arg == "--device") { if (!need(&v) || !parse_u64(v, 1023, &u)) { *err = "invalid value for --device"; return false; } a->device = (int)u; }
    else if (arg == "--gamma") {
      if (!need(&v)) return false;
      float m = 0, s = 0;
      if (sscanf(v.c_str(), "%f,%f", &m, &s) != 2 || !(m > 0) || !(s > 0)) { *err = "--gamma expects mean,std"; return false; }
      a->gamma = std::make_pair(m, s);
    }
    else if (arg == "--per-read-lengths") a->per_read_lengths = true;
    else if (arg == "--rng") {
      if (!need(&v)) return false;
      if (v == "philox") { a->rng_philox = true; a->rng_philox_full = false; }
      else if (v == "philox-full") { a->rng_philox = true; a->rng_philox_full = true; }
      else if (v == "reference") { a->rng_philox = false; a->rng_philox_full = false; }
      else { *err = "invalid value for --rng (reference, philox, philox-full)"; return false; }
    }
    else if (arg == "--uniform-start") a->uniform_start = true;
    else { *err = "Found argument '" + arg + "' which wasn't expected"; return false; }
  }
  // cli.rs:88-92: ArgGroup "genomes" is required, and --output has no default
  if (a->genome.empty() && !a->genome_file) { *err = "one of --genome / --genome-file is required"; return false; }
  if (!a->genome.empty() && a->genome_file) { *err = "--genome and --genome-file cannot be used together"; return false; }
  if (a->output.empty()) { *err = "--output is required"; return false; }
  return true;
}

std::unique_ptr<ErrorProfile> determine_error_profile(const CliArgs& args, std::string* err) {
  switch (args.error_profile) {
    case ErrorProfileKind::PerfectShort: {  // cli.rs:231-234
      auto p = std::make_unique<PerfectShortErrorProfile>();
      p->read_length = args.read_length; p->insert_size = args.insert_size;
      return p;
    }
    case ErrorProfileKind::MinimalShort: {  // cli.rs:235-241: stds fixed at 75 / 15
      auto p = std::make_unique<MinimalShortErrorProfile>();
      p->read_length = args.read_length; p->insert_size = args.insert_size;
      p->mean_phred_score = args.mean_phred_score; p->insert_size_std = 75.0; p->read_length_std = 15.0;
      return p;
    }
    case ErrorProfileKind::PerfectLong: {  // cli.rs:283
      auto p = std::make_unique<PerfectLongErrorProfile>();
      if (args.gamma) { p->gamma_mean = args.gamma->first; p->gamma_std = args.gamma->second; }
      if (args.per_read_lengths) p->length_mode = SIMMR_LEN_PER_READ;
      if (args.uniform_start) p->long_start_mode = SIMMR_START_UNIFORM;
      return p;
    }
    case ErrorProfileKind::MinimalLong: {  // cli.rs:284-297
      auto p = std::make_unique<MinimalLongErrorProfile>();
      p->mean_phred_score = args.mean_phred_score;
      p->read_length = args.read_length < 400 ? 20000 : args.read_length;
      p->read_length_std = args.read_length < 400 ? args.read_length_std : 5000.0;
      if (args.gamma) { p->gamma_mean = args.gamma->first; p->gamma_std = args.gamma->second; }
      if (args.per_read_lengths) p->length_mode = SIMMR_LEN_PER_READ;
      if (args.uniform_start) p->long_start_mode = SIMMR_START_UNIFORM;
      return p;
    }
    case ErrorProfileKind::CustomLong:    // extension: the same object, driven through simulate_long_reads
    case ErrorProfileKind::CustomShort: {  // cli.rs:255-272
      if (!args.custom_profile) { *err = "--custom-profile is required with --error-profile custom-short / custom-long"; return nullptr; }
      std::string e2;
      auto p = CustomShortErrorProfile::from_path(*args.custom_profile, &e2);
      if (!p) { *err = "Error parsing custom error profile: " + e2; return nullptr; }
      if (args.error_profile == ErrorProfileKind::CustomLong) {
        if (args.per_read_lengths) p->length_mode = SIMMR_LEN_PER_READ;
        if (args.uniform_start) p->long_start_mode = SIMMR_START_UNIFORM;
      }
      return p;
    }
  }
  *err = "unknown error profile";
  return nullptr;
}

std::unique_ptr<AbundanceProfile> determine_abundance_profile(const CliArgs& args,
                                                              std::optional<std::vector<double>> abundances) {
  switch (args.abundance_profile) {  // cli.rs:306-320
    case AbundanceProfileKind::Exact: return std::make_unique<ExactAbundanceProfile>();
    case AbundanceProfileKind::Uniform: {
      auto p = std::make_unique<UniformAbundanceProfile>();
      p->size_adjusted = args.size_adjusted;
      return p;
    }
    case AbundanceProfileKind::Custom: {
      auto p = std::make_unique<CustomAbundanceProfile>();
      p->size_adjusted = args.size_adjusted;
      p->abundances = abundances.value_or(std::vector<double>());
      return p;
    }
  }
  return nullptr;
}

}  // namespace simmr_host

// ---- plain-C views for the CPU tests (ctypes): no GPU, no simulation ------------
extern "C" {
using namespace simmr_host;

// returns a malloc'ed string (caller frees with simmr_host_free)
static char* dup_str(const std::string& s) {
  char* p = (char*)malloc(s.size() + 1);
  memcpy(p, s.c_str(), s.size() + 1);
  return p;
}
void simmr_host_free(void* p) { free(p); }
// the PRODUCT's builder of the counter mode's splice tables (csrc/custom_model.hpp: what engine.hip uploads), for the CPU
// test that enumerates the law the tables encode (the test tree's custom-profile specification tests)
uint32_t simmr_host_ctr_splice_tables(const uint32_t* alt, const float* w, uint32_t n, uint32_t self_code, int has_self,
                                      uint32_t* thr, uint32_t* alias) {
  return simmr::ctr_splice_tables(alt, w, n, self_code, has_self != 0, thr, alias);
}
char* simmr_host_normalize(const char* raw, uint64_t n) { return dup_str(normalize(std::string(raw, n))); }
char* simmr_host_format_f64(double v) { return dup_str(format_f64_display(v)); }
char* simmr_host_format_header(const char* fmt, const char* genome_id, uint32_t read_id, const char* seq_id,
                               uint64_t start, uint64_t end, int revcomp, int pair) {
  return dup_str(format_header(fmt, genome_id, read_id, seq_id, start, end, revcomp != 0, pair));
}
// Loads a FASTA; writes a description "n_seqs\tsize\n" + per sequence "id\tsize\tlen\tseq\n"
char* simmr_host_load_fasta(const char* path, int contiguous) {
  Genome g;
  std::string err;
  if (!Genome::from_fasta(path, contiguous != 0, &g, &err)) return dup_str("ERR\t" + err);
  std::string out = std::to_string(g.num_seqs) + "\t" + std::to_string(g.size) + "\n";
  for (const Seq& s : g.sequence)
    out += s.id + "\t" + std::to_string(s.size) + "\t" + std::to_string(s.seq.size()) + "\t" + s.seq + "\n";
  return dup_str(out);
}
char* simmr_host_parse_genome_file(const char* path) {
  std::vector<GenomeRecord> recs;
  std::string err;
  if (!parse_genome_file(path, &recs, &err)) return dup_str("ERR\t" + err);
  std::string out;
  for (const auto& r : recs)
    out += r.filepath + "\t" + (r.uuid ? *r.uuid : std::string("<none>")) + "\t" +
           (r.abundance ? format_f64_display(*r.abundance) : std::string("<none>")) + "\n";
  return dup_str(out);
}
}
