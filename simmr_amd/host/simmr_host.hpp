// simmr_host.hpp — C++ host layer above the C ABI: the callers and data formats
// either side of the accelerated path, mirroring the reference's own modules
// (same names, argument meaning and error behaviour):
//   genome.rs   -> Seq, Genome, Genome::from_fasta
//   files.rs    -> GenomeRecord, parse_genome_file, write_metadata
//   fastq.rs    -> write_to_fastq (header template interpolation)
//   error_profiles/*.rs, abundance_profiles/*.rs -> ErrorProfile, AbundanceProfile
//   cli.rs      -> CliArgs, determine_error_profile, determine_abundance_profile
// No simulation arithmetic lives here: reads come from libsimmr_hip.so.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "../../include/simmr_hip.h"

namespace simmr_host {

// ---------------------------------------------------------------- genome.rs
struct Seq {  // genome.rs:17-23
  std::string id;    // FASTA header line (needletail record.id())
  uint64_t uuid = 0;
  std::string seq;   // normalised sequence
  uint64_t size = 0;
};

struct Genome {  // genome.rs:26-41
  std::string uuid;
  std::string filepath;
  std::vector<Seq> sequence;
  uint64_t size = 0;
  uint64_t num_seqs = 0;
  std::optional<double> abundance;
  bool contiguous = false;

  // genome.rs:89-162.  Returns an error string on failure (Result<Genome, String>).
  static bool from_fasta(const std::string& filepath, bool contiguous, Genome* out, std::string* err);
};

// The record structure of a FASTA file without its sequences: what Genome::from_fasta does before it
// normalises (genome.rs:93-112).  `data` is the file; record c has header ids[c] and the raw body
// data[body[c].first, body[c].first + body[c].second) — the input of simmr_stage_fasta, which normalises
// and packs on the device.
struct FastaRecords {
  std::string data;
  std::vector<std::string> ids;
  std::vector<std::pair<size_t, size_t>> body;
};
bool scan_fasta(const std::string& filepath, FastaRecords* out, std::string* err);

// needletail 0.4.1 Sequence::normalize(iupac = false) as used at genome.rs:114
std::string normalize(const std::string& raw);
// util.rs:124-129: first 64 bits of a random UUID v4
uint64_t generate_id();
std::string uuid_from_u64(uint64_t u);  // genome.rs:69-73: format!("{:x}", u)

// ---------------------------------------------------------------- files.rs
struct GenomeRecord {  // files.rs:19-26
  std::string filepath;
  std::optional<std::string> uuid;
  std::optional<double> abundance;
};
bool parse_genome_file(const std::string& filepath, std::vector<GenomeRecord>* out, std::string* err);
struct MetadataRow {
  std::string genome_id, filepath;
  uint64_t num_reads;
  double abundance;
};
bool write_metadata(const std::vector<MetadataRow>& rows, const std::string& output, std::string* err);
// Rust `{}` for f64: shortest digits that round-trip, never scientific notation
std::string format_f64_display(double v);

// ---------------------------------------------------------------- fastq.rs
// Host copy of one shard's SoA (simmr_reads_out columns).
struct HostReads {
  uint64_t n_reads = 0;
  bool paired = false;
  std::vector<uint8_t> seq, qual;  // qual already +33 (qual_offset = 33)
  std::vector<uint64_t> seq_off, start, end;
  std::vector<uint32_t> contig, genome, read_id;
  std::vector<uint8_t> flags;
};
// header_format.replace(...) chain of fastq.rs:34-56 for one read
std::string format_header(const std::string& header_format, const std::string& genome_id, uint32_t read_id,
                          const std::string& sequence_id, uint64_t start, uint64_t end, bool revcomp,
                          int pair);
// fastq.rs:14-124 for reads [first, first+count) of `reads`, all from `genome`
bool write_to_fastq(const std::string& genome_uuid, const Genome& genome, const HostReads& reads,
                    uint64_t first, uint64_t count, const std::string& output,
                    const std::string& header_format, bool append, std::string* err);

// ------------------------------------------------------- error_profiles/*.rs
class ErrorProfile {  // error_profiles/base.rs:6-32 (the per-read methods run on the device)
 public:
  virtual ~ErrorProfile() = default;
  virtual simmr_error_profile pod() const = 0;
  virtual uint16_t minimum_genome_size() const = 0;
  virtual bool is_long_read() const = 0;
};
struct PerfectShortErrorProfile : ErrorProfile {  // perfect_short.rs
  uint16_t read_length = 150, insert_size = 150;
  simmr_error_profile pod() const override;
  uint16_t minimum_genome_size() const override { return (uint16_t)(2u * read_length + insert_size); }
  bool is_long_read() const override { return false; }
};
struct MinimalShortErrorProfile : ErrorProfile {  // minimal_short.rs
  uint16_t read_length = 150, insert_size = 150;
  uint8_t mean_phred_score = 30;
  double insert_size_std = 75.0, read_length_std = 15.0;
  simmr_error_profile pod() const override;
  uint16_t minimum_genome_size() const override { return (uint16_t)(2u * read_length + insert_size); }
  bool is_long_read() const override { return false; }
};
struct MinimalLongErrorProfile : ErrorProfile {  // minimal_long.rs
  uint8_t mean_phred_score = 30;
  uint16_t read_length = 20000;  // unused by the reference (minimal_long.rs:64-65 hard-codes the gamma)
  double read_length_std = 5000.0;
  float gamma_mean = 20000.0f, gamma_std = 15000.0f;
  uint32_t length_mode = SIMMR_LEN_REFERENCE;
  uint8_t long_start_mode = SIMMR_START_REFERENCE;
  simmr_error_profile pod() const override;
  uint16_t minimum_genome_size() const override { return 20000; }
  bool is_long_read() const override { return true; }
};
struct PerfectLongErrorProfile : MinimalLongErrorProfile {  // perfect_long.rs
  simmr_error_profile pod() const override;
};

struct CustomShortErrorProfile : ErrorProfile {  // custom_short.rs (model read by cli.rs:255-272)
  std::vector<uint8_t> model;  // bincode ErrorModelParams, handed to the library as is
  double read_length_mean = 0, insert_size_mean = 0;
  bool is_long = false;
  uint32_t length_mode = SIMMR_LEN_REFERENCE;       // custom-long only
  uint8_t long_start_mode = SIMMR_START_REFERENCE;  // custom-long only
  // shared/src/encoding.rs:268-281 deserialize_model_from_path
  static std::unique_ptr<CustomShortErrorProfile> from_path(const std::string& path, std::string* err);
  simmr_error_profile pod() const override;
  uint16_t minimum_genome_size() const override;  // custom_short.rs:535-538
  bool is_long_read() const override { return is_long; }
};

// --------------------------------------------------- abundance_profiles/*.rs
using Abundances = std::vector<std::pair<uint64_t, double>>;
class AbundanceProfile {  // abundance_profiles/base.rs:10-69
 public:
  virtual ~AbundanceProfile() = default;
  virtual bool is_size_aware() const = 0;
  virtual Abundances determine_abundances(uint64_t total_reads, uint64_t num_genomes) const = 0;
  virtual Abundances adjust_for_size(const std::vector<Genome>& genomes, const Abundances& read_abundances,
                                     uint64_t read_length, bool paired) const;
};
struct UniformAbundanceProfile : AbundanceProfile {  // uniform.rs
  bool size_adjusted = false;
  bool is_size_aware() const override { return size_adjusted; }
  Abundances determine_abundances(uint64_t total_reads, uint64_t num_genomes) const override;
};
struct ExactAbundanceProfile : AbundanceProfile {  // exact.rs
  bool is_size_aware() const override { return false; }
  Abundances determine_abundances(uint64_t total_reads, uint64_t num_genomes) const override;
  Abundances adjust_for_size(const std::vector<Genome>&, const Abundances& a, uint64_t, bool) const override { return a; }
};
struct CustomAbundanceProfile : AbundanceProfile {  // custom.rs
  bool size_adjusted = false;
  std::vector<double> abundances;
  bool is_size_aware() const override { return size_adjusted; }
  Abundances determine_abundances(uint64_t total_reads, uint64_t num_genomes) const override;
};

// ---------------------------------------------------------------- cli.rs
enum class ErrorProfileKind { MinimalShort, MinimalLong, PerfectShort, PerfectLong, CustomShort, CustomLong /* extension */ };
enum class AbundanceProfileKind { Exact, Uniform, Custom };
struct CliArgs {  // cli.rs:93-220, same flags and defaults
  std::vector<std::string> genome;
  std::optional<std::string> genome_file;
  std::string output;
  uint64_t num_reads = 1000;
  uint16_t read_length = 150;
  double read_length_std = 10.0;
  uint16_t insert_size = 150;
  uint8_t mean_phred_score = 30;
  ErrorProfileKind error_profile = ErrorProfileKind::PerfectShort;
  AbundanceProfileKind abundance_profile = AbundanceProfileKind::Uniform;
  std::optional<std::string> custom_profile;
  std::optional<uint8_t> with_ani;
  std::string read_header_format =
      "@{:read_id:}|{:genome_id:}/{:pair:} metadata:sid={:sequence_id:}|sp={:start_position:}|ep={:end_position:}|rc={:reverse_complement:}";
  std::optional<uint64_t> seed;
  bool size_adjusted = false;
  bool contiguous = false;
  // extensions of this implementation (not reference flags)
  int device = 0;
  std::vector<int> devices;  // --devices a,b,...: one engine per entry (an ordinal may repeat), the run's ranges dealt to them in turn
  bool host_normalize = false;  // --host-normalize: normalise FASTA on the host instead of the device
  bool host_fastq = false;  // --host-fastq: frame the FASTQ on the host instead of the device
  uint64_t device_chunk_reads = 0;  // --device-chunk-reads: reads generated per device pass (0: what fits the free device memory)
  std::optional<std::pair<float, float>> gamma;  // --gamma mean,std
  bool uniform_start = false;                    // --uniform-start (SIMMR_START_UNIFORM)
  bool per_read_lengths = false;                 // --per-read-lengths (SIMMR_LEN_PER_READ)
  bool rng_philox = false;  // --rng philox: the counter mode for the per-base draws (SIMMR_RNG_PHILOX; statistical parity,
                            // BASELINE.json north_star).  Default `reference`: the reference's own streams, byte-identical output
  bool rng_philox_full = false;  // --rng philox-full: the plan's draws from Philox counters too (SIMMR_RNG_PHILOX_FULL): minimal-short,
                                 // and minimal-long / perfect-long with --per-read-lengths
};
// returns false and fills err on a usage error (clap would exit(2)); help=true for --help
bool parse_cli_args(int argc, const char* const* argv, CliArgs* out, std::string* err, bool* help);
std::string usage();
std::unique_ptr<ErrorProfile> determine_error_profile(const CliArgs& args, std::string* err);   // cli.rs:229-301
std::unique_ptr<AbundanceProfile> determine_abundance_profile(const CliArgs& args,
                                                              std::optional<std::vector<double>> abundances);  // :306-320

}  // namespace simmr_host
