// device_types.hpp — plain structs shared by the host engine and the kernels.
#pragma once
#include <stdint.h>

#include "../../include/simmr_hip.h"

namespace simmr {

// profile kinds as the kernels see them (same values as enum simmr_profile_kind)
#define SIMMR_K_PERFECT_SHORT 0u
#define SIMMR_K_MINIMAL_SHORT 1u
#define SIMMR_K_PERFECT_LONG 2u
#define SIMMR_K_MINIMAL_LONG 3u

// bits of the device error word
#define SIMMR_ERRBIT_GENOME 1u /* contig too small for the profile (simulate.rs:220) */
#define SIMMR_ERRBIT_SLICE 2u  /* slice would leave the contig (a Rust panic)       */

// One sequence record of a staged genome (genome.rs:17-23 `Seq`).
struct ContigDev {
  uint64_t base;  // first base of this contig in the genome's packed plane (multiple of 64)
  uint64_t len;   // Seq.seq.len()
  uint64_t size;  // Seq.size
};

// One staged genome: 2-bit code plane + optional 1-bit exception plane.
// Both planes have 16 bytes of addressable padding in front and 96 behind.
struct GenomeDev {
  const uint32_t* packed;
  const uint32_t* mask;  // nullptr when the genome is pure ACGT
  const ContigDev* contigs;
  uint32_t n_contigs;
  uint32_t has_exc;
};

#define SIMMR_K_CUSTOM 4u
#define PHILOX_ESC 1024u /* level-1 answer of the counter mode: draw again at level 2 */
#define SIMMR_ERRBIT_FASTQ 8u /* a FASTQ header does not fit, or a genome / contig index has no name */
#define SIMMR_ERRBIT_KMER 16u /* simulate_errors chose an alternate the reference cannot splice (deletion / bad code / bad weights) */
#define SIMMR_ERRBIT_PDF 4u /* custom PDF picked a bin without a range (a reference panic) or ran out of words */
#define SIMMR_NOTEBIT_LONGREAD 32u /* not an error: a planned pair has reads longer than LONGREAD_MAXL (selects the TEXT form of the emit kernel) */
#define LONGREAD_MAXL 256u /* longest read the whole-line TEXT kernel takes (text_lines.hip: TL_MAXL) */

// One CustomPDF entry (custom_short.rs:28-35): WeightedAliasIndex<f64> + per-bin Uniform<u32>,
// flattened.  Built on the host exactly as rand_distr 0.4.3 / rand 0.8.5 build them.
struct PdfDev {
  uint32_t n;          // number of densities (alias table size)
  uint32_t idx_zone;   // Uniform<u32>::new(0, n): zone = u32::MAX - ((2^32 - n) % n)
  uint32_t n_bins;     // number of bin ranges
  uint32_t off;        // offset of this PDF in odds[] / alias[]
  uint32_t off_bins;   // offset in bin_low[] / bin_range[] / bin_zone[]
  uint32_t pad;
  double w_scale;      // Uniform<f64>::new(0, weight_sum).scale
};
struct alignas(16) Rec16 { uint32_t x, y, z, w; };
struct CustomDev {
  const PdfDev* pdfs;  // [0] read length, [1] insert size (n == 0 if absent), [2 + p] quality of position p
  const double* odds;
  const uint32_t* alias;
  const uint32_t* bin_low;
  const uint32_t* bin_range;  // 0 = full u32 range
  const uint32_t* bin_zone;
  const Rec16* col_rec;  // the same tables packed for one load per lookup: {odds lo, odds hi, alias, -}
  const Rec16* bin_rec;  // {range, zone, low, -}
  uint32_t n_quality;
  uint32_t pad;
  // simulate_errors (custom_short.rs:455-516), long reads only
  const Rec16* kmer_direct;  // by 2-bit k-mer code: {first record, n alternates (0 = absent, ~0 = the reference panics), Uniform(0, n) zone, -}
  const Rec16* kmer_slots;   // k-mers with an N, open addressing by 3-bit code: {key (~0 = empty), first record, n (~0 = panics), zone}
  const Rec16* kmer_recs;    // per alias column c: {odds f32, alternate c, alternate alias(c) (2 bits per base, bit 31 = not ACGT), scale f32}
  uint32_t kmer_mask;       // slots - 1
  uint32_t kmer_size;
  // fixed-stride form of the direct entries (k <= 7, short lists; null otherwise): the counts go to LDS
  const uint8_t* kmer_cnt8;  // by 2-bit k-mer code: n alternates (0 = absent, 255 = the reference panics)
  const Rec16* kmer_cols;    // [code * kmer_stride + c], same records as kmer_recs
  uint32_t kmer_stride;
  uint32_t pad2;
  // SIMMR_RNG_PHILOX (custom_model.hpp: ctr_splice_tables): kmer_direct[code].w = the k-mer's level-1 threshold T24;
  // level-2 columns parallel to kmer_recs / kmer_cols: {threshold in 2^24ths, alternate c, alternate alias(c), -}
  const Rec16* kmer_recs_ctr;
  const Rec16* kmer_cols_ctr;
  const uint32_t* kmer_tab32;  // by 2-bit k-mer code: T24 << 8 | n alternates (0 = absent, 255 = the reference panics); null without kmer_cols
};

// Device form of simmr_error_profile, with host-derived constants.
struct ProfileDev {
  uint32_t kind;
  uint32_t rng_mode;
  uint32_t read_length;
  uint32_t insert_size;
  uint32_t required;  // minimum_genome_size(), u16 arithmetic done on the host
  float mean_phred_f;
  float pl_mean;      // perfect-long: convert_phred_to_accuracy(20)
  float gamma_shape, gamma_scale;
  double read_length_std, insert_size_std;
  CustomDev custom;
  uint32_t long_start_uniform;   // SIMMR_START_UNIFORM
  // SIMMR_RNG_PHILOX: level 1 = 1024 columns of a 24-bit draw, philox_t1[c] = T | A << 16 and
  // philox_t1[1024 + c] = B (T 16384ths of column c answer outcome A, the rest B; outcome = q | s << 8 or PHILOX_ESC);
  // level 2 = 1024 alias entries thr22 | alias << 22 over the residual law behind the escape cells
  const uint32_t* philox_t1;
  const uint32_t* philox_t2;
  uint32_t philox_qmax;          // largest Phred either table can return
  uint32_t philox_qmax1;         // largest Phred a level-1 column can answer (the template ESCQ of k_emit_philox)
};

struct Key8 {
  uint32_t k[8];
};

struct OuterParams {
  Key8 key;        // ChaCha12 key of the outer StdRng (PCG32 expansion of the seed)
  uint64_t range;  // gen_range(0..range)
  uint64_t zone;   // (range << lz) - 1
};

struct OuterPrefix {
  uint64_t base;   // units emitted before this workgroup
  uint32_t state;  // 0 = NEED_IDX, 1 = NEED_SEED at its first slot
  uint32_t pad;
};

struct OuterScanResult {
  uint64_t total_units;   // units in the scanned blocks when they start in NEED_IDX
  uint64_t wg_lo, wg_hi;
  uint64_t end_slot;
  uint32_t end_state;
  uint32_t end_state1;    // the same two numbers for a start in NEED_SEED
  uint64_t total_units1;
};

// Per-unit plan columns (unit = pair or long read).
struct PlanArrays {
  uint32_t* len;    // read length L (both mates) / long-read length; the unit writes 2L resp. L bytes to seq[]
  uint64_t* a;      // fwd_start / read_start
  uint64_t* b;      // rev_end (mate-2 slice start) / read_end
  uint64_t* qs2;    // mate-2 Phred seed   (nullptr when unused)
  uint64_t* ms2;    // mate-2 mutation seed
  uint8_t* flags;
};

struct OutCols {
  uint64_t* seq_off;
  uint64_t* start;
  uint64_t* end;
  uint32_t* contig;
  uint32_t* genome;
  uint32_t* read_id;
  uint8_t* flags;
};

// A maximal range of consecutive long reads drawn from one genome.
struct LongGenomeRun {
  uint64_t first_read;     // global index of the run's first read
  uint64_t n_reads;
  uint64_t max_size;       // largest Seq.size in the genome
  const uint32_t* usable;  // usable-sequence index -> contig index (reference mode)
  uint32_t n_usable;
  uint32_t genome;
};

// The run's counters on the device: row 0 holds the SIMMR_N_COUNTERS counters a reader sees; SIMMR_CNT_SHARDS more rows
// behind it take the emit kernels' adds (row = wave or workgroup number modulo the rows), so that thousands of waves do
// not queue on eight addresses at the end of a kernel — k_counters_fold sums the rows into row 0 when the counters are
// read (simmr_counters).  Measured with k_emit_philox at 64 workgroups per CU: 10.6 -> 10.4 ms, and 18 -> 10.5 ms at 256.
#define SIMMR_CNT_SHARDS 64u
#if defined(__HIPCC__)
__device__ __forceinline__ void shard_add(unsigned long long* counters, uint32_t idx, unsigned long long v) {
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  atomicAdd(&counters[(1u + (w & (SIMMR_CNT_SHARDS - 1u))) * SIMMR_N_COUNTERS + idx], v);
}
#endif

}  // namespace simmr
