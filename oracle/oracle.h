/*
 * oracle.h — CPU ORACLE for the simmr hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of genomicsoup/simmr's per-read
 * sampling / mutation path (simmr/src/simulate.rs + error_profiles/ +
 * abundance_profiles/ + util.rs) together with the arithmetic of the pinned,
 * un-vendored crates it calls (Cargo.lock:656-693):
 *     rand 0.8.5, rand_chacha 0.3.1, rand_core 0.6.3, rand_distr 0.4.3.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  Nothing under simmr_amd/ links, imports
 * or executes it.
 *
 * Parity pinning: the reference is Rust and cannot be built in
 * this image (no cargo/rustc), and its two simulate tests are #[ignore]d with
 * stale ground truth.  The oracle is pinned by
 *   - the public ChaCha20 / ChaCha12 zero-key vectors,
 *   - rand 0.8's StdRng unit-test value,
 *   - the two reference-authored RNG values in comments of
 *     simmr/src/tests/simulate_tests.rs:27 and :75,
 *   - the live reference unit tests (util_tests.rs, abundance_profile_tests.rs,
 *     error_profile_tests.rs, genome_tests.rs, shared/src/encoding.rs:288-314).
 * Read CONTENT downstream of the RNG (fwd_start for a given pe_seed, Normal /
 * Gamma outputs) has no reference-authored golden vector: that part is
 * "parity unpinned" beyond the crate algorithms restated here.  The ziggurat
 * tables are regenerated from rand_distr's published generator formulas (the
 * literal tables are not in /root/reference).
 */
#ifndef SIMMR_ORACLE_H
#define SIMMR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/simmr_hip.h" /* POD types only (profile, reads_out, flags) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- rand_core 0.6.3 BlockRng<ChaCha12Core> == rand 0.8.5 StdRng */
typedef struct orc_rng {
  uint32_t key[8];
  uint64_t counter;     /* block counter of the next refill                  */
  uint32_t results[64]; /* rand_chacha 0.3.1 buffers 4 blocks per refill     */
  uint32_t index;       /* next unread word, 64 = empty                      */
  uint64_t words_used;  /* oracle-only bookkeeping: words consumed so far    */
  uint32_t ctr;         /* 1: SIMMR_RNG_PHILOX_FULL's word stream W(seed) instead of ChaCha12 (key[0..1] = seed) */
} orc_rng;
/* Which word source orc_rng_seed_from_u64 gives the generators it makes ON THIS THREAD: 0 = the reference's (PCG32
 * expansion + ChaCha12), 1 = SIMMR_RNG_PHILOX_FULL's W(seed): word w = word w & 3 of the Philox4x32-10 block with key =
 * seed and counter (w >> 2, 3, 'simm', 'r\0\0\3').  The plan functions of simulate.c switch it on around their draws. */
void orc_set_stream_kind(int kind);

void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint32_t rounds, uint32_t out[16]);
void orc_pcg32_expand(uint64_t state, uint32_t key_out[8]);
void orc_rng_from_seed(orc_rng* r, const uint8_t seed[32]);
void orc_rng_seed_from_u64(orc_rng* r, uint64_t state);
uint32_t orc_next_u32(orc_rng* r);
uint64_t orc_next_u64(orc_rng* r);
/* gen_range(lo..hi): returns 0 on success, -1 on an empty range (panic in Rust) */
int orc_gen_range_u64(orc_rng* r, uint64_t lo, uint64_t hi, uint64_t* out);
int orc_gen_range_u32(orc_rng* r, uint32_t lo, uint32_t hi, uint32_t* out);
float orc_gen_f32(orc_rng* r);
double orc_gen_f64(orc_rng* r);
int orc_gen_bool(orc_rng* r);
int orc_gen_option_u64(orc_rng* r, uint64_t* out); /* 1 = Some(*out), 0 = None */
double orc_open01_f64(orc_rng* r);
float orc_open01_f32(orc_rng* r);
double orc_standard_normal(orc_rng* r);
double orc_normal_f64(orc_rng* r, double mean, double std);
float orc_normal_f32(orc_rng* r, float mean, float std);
/* Gamma<f32>::new(shape, scale).sample(): -1 if shape <= 1 (not restated) */
int orc_gamma_f32(orc_rng* r, float shape, float scale, float* out);
const double* orc_zig_norm_x(void); /* 257 entries */
const double* orc_zig_norm_f(void); /* 257 entries */

/* ---------------- simmr/src/util.rs */
uint8_t orc_complement(uint8_t n);
void orc_reverse_complement(const uint8_t* in, uint64_t n, uint8_t* out);
uint8_t orc_encode_quality_score(uint8_t s);
float orc_convert_phred_to_probability(uint8_t score);
uint8_t orc_convert_probability_to_phred(float prob);
float orc_convert_phred_to_accuracy(uint8_t score);
uint8_t orc_convert_accuracy_to_phred(float acc);
/* shared/src/encoding.rs two_bit_encode_kmer / decode */
int orc_two_bit_encode_kmer(const uint8_t* kmer, uint32_t k, uint32_t* out);
int orc_two_bit_decode_kmer(uint32_t code, uint32_t k, uint8_t* out);

uint64_t orc_entropy_substitute(uint64_t x, uint32_t which);
uint64_t orc_per_read_seed(uint64_t seed, uint64_t read_index);

/* ---------------- ErrorProfile trait methods (error_profiles/) */
int orc_profile_minimum_genome_size(const simmr_error_profile* p, uint16_t* out);
int orc_profile_is_long_read(const simmr_error_profile* p);
int orc_profile_get_read_length(const simmr_error_profile* p, uint64_t seed, uint16_t* out);
int orc_profile_get_random_read_length(const simmr_error_profile* p, uint64_t seed, uint16_t* out);
int orc_profile_get_insert_size(const simmr_error_profile* p, uint64_t seed, uint16_t* out);
int orc_profile_simulate_phred_scores(const simmr_error_profile* p, uint64_t len, uint64_t seed,
                                      uint8_t* out);
int orc_profile_simulate_point_mutations(const simmr_error_profile* p, const uint8_t* seq,
                                         const uint8_t* qual, uint64_t len, uint64_t seed,
                                         uint8_t* out);

/* ---------------- AbundanceProfile (abundance_profiles/) */
void orc_uniform_determine_abundances(uint64_t total_reads, uint64_t num_genomes,
                                      uint64_t* reads_out, double* abund_out);
void orc_exact_determine_abundances(uint64_t total_reads, uint64_t num_genomes,
                                    uint64_t* reads_out, double* abund_out);
void orc_custom_determine_abundances(const double* abundances, uint64_t total_reads,
                                     uint64_t num_genomes, uint64_t* reads_out, double* abund_out);
void orc_adjust_for_size(const uint64_t* genome_sizes, const uint64_t* reads_in,
                         const double* abund_in, uint64_t num_genomes, uint64_t* reads_out,
                         double* abund_out);

/* ---------------- simulate.rs */
typedef struct orc_genome {
  uint32_t n_contigs;
  const uint8_t* const* seq; /* Seq.seq  */
  const uint64_t* len;       /* Seq.seq.len() */
  const uint64_t* size;      /* Seq.size */
} orc_genome;

/* simulate_pe_reads_from_genome's outer loop (simulate.rs:172-184): fills
 * contig_idx[i], pe_seed[i] for pairs [first, first+count).  *slots = u64
 * draws consumed up to the end of the last returned pair. */
/* SIMMR_RNG_PHILOX_FULL: the outer draws of pairs [first, first + count) of a genome's run, one Philox block per pair */
int orc_pe_outer_ctr(uint64_t n_contigs, uint64_t seed, uint64_t first, uint64_t count, uint32_t* contig_idx, uint64_t* pe_seed);
int orc_pe_outer(uint64_t n_contigs, uint64_t seed, uint64_t first, uint64_t count,
                 uint32_t* contig_idx, uint64_t* pe_seed, uint64_t* slots);

/* Per-pair plan (simulate.rs:211-258). */
typedef struct orc_pe_plan {
  uint32_t read_length;
  uint32_t insert_size;
  uint64_t fwd_start, fwd_end, rev_end, rev_start;
  uint64_t qseed2, mseed2; /* seeds used for mate 2 (drawn or substituted) */
  uint8_t flags2;          /* SIMMR_FLAG_* of mate 2                         */
} orc_pe_plan;
int orc_pe_plan_pair(const simmr_error_profile* p, uint64_t contig_size, uint64_t pe_seed,
                     orc_pe_plan* plan);

/* Whole shard: pairs [first, first+count) of one genome, SoA out (HOST
 * pointers; same layout as simmr_reads_out).  threads > 1 parallelises over
 * pairs (cpu_baseline only; results identical).  Returns 0 or a negative code;
 * *total_bases receives the bytes written to seq/qual. */
int orc_simulate_pe_reads_from_genome(const orc_genome* g, const simmr_error_profile* p,
                                      uint64_t genome_reads, uint64_t seed, uint64_t first,
                                      uint64_t count, uint32_t read_id_base,
                                      const simmr_reads_out* out, uint64_t* total_bases,
                                      int threads);

/* simulate_long_reads (simulate.rs:323-406). shard = global read index range. */
int orc_simulate_long_reads(const orc_genome* genomes, uint32_t n_genomes,
                            const uint64_t* genome_reads, const simmr_error_profile* p,
                            int has_seed, uint64_t seed, uint64_t first, uint64_t count,
                            uint32_t read_id_base, const simmr_reads_out* out,
                            uint64_t* total_bases, uint32_t* const_len, int threads);

/* cpu_baseline only: also pay the reference's per-read clones of the genome (simulate.rs:362-375); results unchanged */
void orc_set_faithful_cost(int on);

/* ---------------- custom (empirical) profile: custom_short.rs + its crates (custom.c) */
typedef struct orc_uniform_u32 { uint32_t low, range, z; } orc_uniform_u32;
typedef struct orc_uniform_f64 { double low, scale; } orc_uniform_f64;
typedef struct orc_alias {
  uint32_t n;
  double* odds;       /* no_alias_odds */
  uint32_t* aliases;
  orc_uniform_u32 uniform_index;
  orc_uniform_f64 uniform_weight;
} orc_alias;
typedef struct orc_bins {
  uint64_t num_bins, bin_width, n_density, n_ranges;
  double* density;
  uint32_t *range_lo, *range_hi;
} orc_bins;
typedef struct orc_model { /* shared/src/encoding.rs:102-117 ErrorModelParams */
  uint64_t bin_size, n_quality;
  orc_bins* quality;
  uint8_t bit_encoding;
  uint64_t kmer_size, n_prob;
  uint32_t* prob_kmer; uint64_t* prob_n; uint32_t** prob_alt; float** prob_w;
  double insert_size_mean, insert_size_std;
  uint8_t has_insert_bins;
  orc_bins insert_bins;
  double read_length_mean, read_length_std;
  orc_bins read_length_bins;
  uint8_t is_long;
} orc_model;
typedef struct orc_pdf { orc_alias alias; uint32_t n_bins; orc_uniform_u32* bins; } orc_pdf;

void orc_uniform_u32_new_inclusive(uint32_t low, uint32_t high, orc_uniform_u32* u);
uint32_t orc_uniform_u32_sample(const orc_uniform_u32* u, orc_rng* r);
void orc_uniform_f64_new(double low, double high, orc_uniform_f64* u);
double orc_uniform_f64_sample(const orc_uniform_f64* u, orc_rng* r);
int orc_alias_new(const double* weights, uint32_t n, orc_alias* a);
void orc_alias_free(orc_alias* a);
uint32_t orc_alias_sample(const orc_alias* a, orc_rng* r);
int orc_model_parse(const uint8_t* bytes, uint64_t n, orc_model* m);
int orc_pdf_new(const orc_bins* b, orc_pdf* p);
int orc_pdf_sample(const orc_pdf* p, uint64_t seed, uint32_t* out);
typedef struct orc_custom orc_custom;
orc_custom* orc_custom_new(const uint8_t* bytes, uint64_t n);
const orc_model* orc_custom_model(const orc_custom* c);
int orc_custom_get_read_length(const orc_custom* c, uint64_t seed, uint16_t* out);
int orc_custom_get_insert_size(const orc_custom* c, uint64_t seed, uint16_t* out);
uint16_t orc_custom_minimum_genome_size(const orc_custom* c);
int orc_custom_simulate_phred_scores(const orc_custom* c, uint64_t len, uint64_t seed, uint8_t* out);
int64_t orc_custom_simulate_errors(const orc_model* m, const uint8_t* seq, uint64_t len, uint64_t seed, uint8_t* out);
/* the same walk with the counter mode's draws (SIMMR_RNG_PHILOX with a custom long-read model) */
int64_t orc_custom_simulate_errors_philox(const orc_model* m, const uint8_t* seq, uint64_t len, uint64_t seed, uint8_t* out);
double orc_gaussian_kde(double x, const double* xs, uint64_t n, double bandwidth);  /* custom_long.rs:36-44 */

/* ---------------- SIMMR_RNG_PHILOX mode (philox.c): counter-based per-base draws */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
#define ORC_PHILOX_ESC 1024u /* level-1 answer "draw again at level 2" */
uint32_t orc_philox_tables(uint32_t kind, uint8_t mean_phred, uint64_t t1[1024], uint32_t t2[1024]);
void orc_philox_read(const simmr_error_profile* p, const uint8_t* seq, uint64_t len, uint64_t key64,
                     uint8_t* qual_out, uint8_t* seq_out);

/* counters over a finished SoA (same definitions as enum simmr_counter where
 * derivable from outputs + the packed reference) */
const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
/* the counter mode's two-level tables of one k-mer's alternates (custom.c: ctr_splice_tables); returns T24 */
uint32_t orc_ctr_splice_tables(const uint32_t* alt, const float* w, uint32_t n, uint32_t self_code, int has_self,
                               uint32_t* thr, uint32_t* alias);

#endif
