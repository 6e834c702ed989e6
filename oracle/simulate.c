/*
 * simulate.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * CPU restatement of simmr's simulation core and its profile plug-ins:
 *   simmr/src/simulate.rs                       (A1-A7 of SURVEY.md §8a)
 *   simmr/src/error_profiles/{perfect_short,minimal_short,perfect_long,
 *                             minimal_long}.rs  (A8-A11)
 *   simmr/src/abundance_profiles/               (A13)
 *   simmr/src/util.rs                           (A14-A15)
 * Written from the behaviour of those files; every function cites the lines it
 * follows.  Outputs use the SoA layout of include/simmr_hip.h so the HIP path
 * can be compared with memcmp.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle.h"

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

/* ------------------------------------------------------------------ util.rs */

/* util.rs:15-27 */
uint8_t orc_complement(uint8_t n) {
  switch (n) {
    case 'A': return 'T'; case 'a': return 't';
    case 'T': return 'A'; case 't': return 'a';
    case 'C': return 'G'; case 'c': return 'g';
    case 'G': return 'C'; case 'g': return 'c';
    default: return n;
  }
}
/* util.rs:32-37 */
void orc_reverse_complement(const uint8_t* in, uint64_t n, uint8_t* out) {
  for (uint64_t i = 0; i < n; i++) out[i] = orc_complement(in[n - 1 - i]);
}
/* util.rs:46-50 (unchecked u8 add; wraps in release builds) */
uint8_t orc_encode_quality_score(uint8_t s) { return (uint8_t)(s + 33); }
/* util.rs:69-71 */
float orc_convert_phred_to_probability(uint8_t score) { return powf(10.0f, -((float)score / 10.0f)); }
/* saturating float -> u8 `as` cast (NaN -> 0) */
static uint8_t sat_u8(float f) {
  if (!(f == f)) return 0;
  if (f <= 0.0f) return 0;
  if (f >= 255.0f) return 255;
  return (uint8_t)f;
}
static uint16_t sat_u16_f32(float f) {
  if (!(f == f)) return 0;
  if (f <= 0.0f) return 0;
  if (f >= 65535.0f) return 65535;
  return (uint16_t)f;
}
static uint16_t sat_u16_f64(double f) {
  if (!(f == f)) return 0;
  if (f <= 0.0) return 0;
  if (f >= 65535.0) return 65535;
  return (uint16_t)f;
}
/* util.rs:83-85 */
uint8_t orc_convert_probability_to_phred(float prob) { return sat_u8(-10.0f * log10f(prob)); }
/* util.rs:96-98 */
float orc_convert_phred_to_accuracy(uint8_t score) { return 1.0f - orc_convert_phred_to_probability(score); }
/* util.rs:109-111 */
uint8_t orc_convert_accuracy_to_phred(float acc) { return sat_u8(roundf(-10.0f * log10f(1.0f - acc))); }

/* shared/src/encoding.rs:146-177 with esize = 2 */
int orc_two_bit_encode_kmer(const uint8_t* kmer, uint32_t k, uint32_t* out) {
  uint32_t e = 0;
  for (uint32_t i = 0; i < k; i++) {
    uint32_t v;
    switch (kmer[i]) {
      case 'A': v = 0; break; case 'C': v = 1; break;
      case 'G': v = 2; break; case 'T': v = 3; break;
      default: return -1;
    }
    e = (e & ~(3u << (2 * i))) | (v << (2 * i));
  }
  *out = e;
  return 0;
}
/* shared/src/encoding.rs:185-210 with esize = 2 */
int orc_two_bit_decode_kmer(uint32_t code, uint32_t k, uint8_t* out) {
  static const char L[4] = {'A', 'C', 'G', 'T'};
  for (uint32_t i = 0; i < k; i++) out[i] = (uint8_t)L[(code >> (2 * i)) & 3];
  return 0;
}

/* Documented substitution for OS entropy (see include/simmr_hip.h): SplitMix64
 * finaliser of x + which * golden. */
uint64_t orc_entropy_substitute(uint64_t x, uint32_t which) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL * (uint64_t)which;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
uint64_t orc_per_read_seed(uint64_t seed, uint64_t read_index) {
  return orc_entropy_substitute(seed ^ (read_index * 0x9E3779B97F4A7C15ULL), 3);
}

/* ------------------------------------------------------- ErrorProfile impls */

/* CustomShortErrorProfile objects are built once per model buffer (cli.rs:255-272) */
static orc_custom* custom_of(const simmr_error_profile* p) {
  /* keyed by the model CONTENT (FNV-1a), not by the caller's buffer address */
  static uint64_t key_hash[8], key_n[8];
  static orc_custom* val[8];
  static int used = 0;
  orc_custom* r = NULL;
  uint64_t h = 1469598103934665603ULL;
  const uint8_t* b = (const uint8_t*)p->custom_model;
  for (uint64_t i = 0; i < p->custom_model_bytes; i++) { h ^= b[i]; h *= 1099511628211ULL; }
#pragma omp critical(orc_custom_cache)
  {
    for (int i = 0; i < used; i++) if (key_hash[i] == h && key_n[i] == p->custom_model_bytes) r = val[i];
    if (!r) {
      r = orc_custom_new(b, p->custom_model_bytes);
      if (r) { int slot = used < 8 ? used++ : 7; key_hash[slot] = h; key_n[slot] = p->custom_model_bytes; val[slot] = r; }
    }
  }
  return r;
}

int orc_profile_is_long_read(const simmr_error_profile* p) {
  /* perfect_short.rs:61, minimal_short.rs:147, perfect_long.rs:133, minimal_long.rs:156, custom_short.rs:540-542 */
  if (p->kind == SIMMR_CUSTOM) { orc_custom* c = custom_of(p); return c ? orc_custom_model(c)->is_long : 0; }
  return p->kind == SIMMR_PERFECT_LONG || p->kind == SIMMR_MINIMAL_LONG;
}

/* perfect_short.rs:56-59, minimal_short.rs:142-145 (u16 arithmetic; a release
 * build wraps), perfect_long.rs:129-131, minimal_long.rs:152-154 */
int orc_profile_minimum_genome_size(const simmr_error_profile* p, uint16_t* out) {
  switch (p->kind) {
    case SIMMR_PERFECT_SHORT:
    case SIMMR_MINIMAL_SHORT:
      *out = (uint16_t)(2u * p->read_length + p->insert_size);
      return 0;
    case SIMMR_PERFECT_LONG:
    case SIMMR_MINIMAL_LONG:
      *out = 20000;
      return 0;
    case SIMMR_CUSTOM: {
      orc_custom* c = custom_of(p);
      if (!c) FAIL(SIMMR_EINVAL, "cannot parse the custom model");
      *out = orc_custom_minimum_genome_size(c);
      return 0;
    }
    default: FAIL(SIMMR_EINVAL, "profile kind %u not restated", p->kind);
  }
}

static int gamma_length(const simmr_error_profile* p, uint64_t seed, uint16_t* out) {
  /* minimal_long.rs:58-73 / perfect_long.rs:40-55: Gamma(shape, scale).floor() as u16 */
  orc_rng r; orc_rng_seed_from_u64(&r, seed);
  float g;
  if (orc_gamma_f32(&r, p->gamma_shape, p->gamma_scale, &g)) FAIL(SIMMR_EINVAL, "gamma shape <= 1");
  *out = sat_u16_f32(floorf(g));
  return 0;
}

int orc_profile_get_read_length(const simmr_error_profile* p, uint64_t seed, uint16_t* out) {
  switch (p->kind) {
    case SIMMR_PERFECT_SHORT: *out = p->read_length; return 0; /* perfect_short.rs:22-24 */
    case SIMMR_MINIMAL_SHORT: {                                 /* minimal_short.rs:33-42 */
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      *out = sat_u16_f64(floor(orc_normal_f64(&r, (double)p->read_length, p->read_length_std)));
      return 0;
    }
    case SIMMR_PERFECT_LONG: *out = 20000; return 0;            /* perfect_long.rs:32-34 */
    case SIMMR_MINIMAL_LONG: return gamma_length(p, seed, out); /* minimal_long.rs:37-53 */
    case SIMMR_CUSTOM: {                                         /* custom_short.rs:237-244 */
      orc_custom* c = custom_of(p);
      if (!c || orc_custom_get_read_length(c, seed, out)) FAIL(SIMMR_EINVAL, "custom read-length PDF failed");
      return 0;
    }
    default: FAIL(SIMMR_EINVAL, "profile kind %u not restated", p->kind);
  }
}

int orc_profile_get_random_read_length(const simmr_error_profile* p, uint64_t seed, uint16_t* out) {
  switch (p->kind) {
    case SIMMR_PERFECT_SHORT: *out = p->read_length; return 0; /* perfect_short.rs:26-28 */
    case SIMMR_MINIMAL_SHORT: return orc_profile_get_read_length(p, seed, out); /* :47-56 */
    case SIMMR_PERFECT_LONG:
    case SIMMR_MINIMAL_LONG: return gamma_length(p, seed, out);
    case SIMMR_CUSTOM: { /* custom_short.rs:286-301: Normal<f64>(read_length_mean, read_length_std), NOT the PDF */
      orc_custom* c = custom_of(p);
      if (!c) FAIL(SIMMR_EINVAL, "cannot parse the custom model");
      const orc_model* m = orc_custom_model(c);
      if (!isfinite(m->read_length_std)) FAIL(SIMMR_ERANGE, "Normal::new(..).unwrap() panics: read_length_std is not finite");
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      *out = sat_u16_f64(floor(orc_normal_f64(&r, m->read_length_mean, m->read_length_std)));
      return 0;
    }
    default: FAIL(SIMMR_EINVAL, "profile kind %u not restated", p->kind);
  }
}

int orc_profile_get_insert_size(const simmr_error_profile* p, uint64_t seed, uint16_t* out) {
  switch (p->kind) {
    case SIMMR_PERFECT_SHORT: *out = p->insert_size; return 0; /* perfect_short.rs:30-32 */
    case SIMMR_MINIMAL_SHORT: {                                 /* minimal_short.rs:58-67 */
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      *out = sat_u16_f64(floor(orc_normal_f64(&r, (double)p->insert_size, p->insert_size_std)));
      return 0;
    }
    case SIMMR_CUSTOM: {                                         /* custom_short.rs:263-270 */
      orc_custom* c = custom_of(p);
      if (!c || orc_custom_get_insert_size(c, seed, out)) FAIL(SIMMR_EINVAL, "custom insert-size PDF failed");
      return 0;
    }
    default: FAIL(SIMMR_EINVAL, "get_insert_size() panics for long-read profiles"); /* minimal_long.rs:29-31 */
  }
}

int orc_profile_simulate_phred_scores(const simmr_error_profile* p, uint64_t len, uint64_t seed,
                                      uint8_t* out) {
  switch (p->kind) {
    case SIMMR_PERFECT_SHORT: /* perfect_short.rs:42-44 */
      memset(out, 60, len);
      return 0;
    case SIMMR_MINIMAL_SHORT: /* minimal_short.rs:83-102 */
    case SIMMR_MINIMAL_LONG: { /* minimal_long.rs:78-99 */
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      for (uint64_t i = 0; i < len; i++)
        out[i] = sat_u8(floorf(orc_normal_f32(&r, (float)p->mean_phred, 10.0f)));
      return 0;
    }
    case SIMMR_PERFECT_LONG: { /* perfect_long.rs:60-78 */
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      float mean = orc_convert_phred_to_accuracy(20);
      for (uint64_t i = 0; i < len; i++) {
        float acc = orc_normal_f32(&r, mean, 0.05f);
        acc = fminf(acc, 0.9999f);
        out[i] = orc_convert_accuracy_to_phred(acc);
      }
      return 0;
    }
    case SIMMR_CUSTOM: {                                         /* custom_short.rs:332-353 */
      orc_custom* c = custom_of(p);
      if (!c || orc_custom_simulate_phred_scores(c, len, seed, out)) FAIL(SIMMR_EINVAL, "custom quality PDF failed");
      return 0;
    }
    default: FAIL(SIMMR_EINVAL, "profile kind %u not restated", p->kind);
  }
}

int orc_profile_simulate_point_mutations(const simmr_error_profile* p, const uint8_t* seq,
                                         const uint8_t* qual, uint64_t len, uint64_t seed,
                                         uint8_t* out) {
  if (p->kind == SIMMR_PERFECT_SHORT || p->kind == SIMMR_CUSTOM) { /* perfect_short.rs:46-54, custom_short.rs:522-529 */
    memmove(out, seq, len);
    return 0;
  }
  /* minimal_short.rs:104-140, minimal_long.rs:106-140, perfect_long.rs:85-119 */
  static const uint8_t ALT_A[3] = {'C', 'G', 'T'}, ALT_C[3] = {'A', 'G', 'T'},
                       ALT_T[3] = {'A', 'C', 'G'}, ALT_G[3] = {'A', 'C', 'T'};
  orc_rng r; orc_rng_seed_from_u64(&r, seed);
  for (uint64_t i = 0; i < len; i++) {
    uint8_t nt = seq[i];
    if (orc_gen_f32(&r) > orc_convert_phred_to_accuracy(qual[i])) {
      const uint8_t* alt = NULL;
      switch (nt) {
        case 'A': alt = ALT_A; break; case 'C': alt = ALT_C; break;
        case 'T': alt = ALT_T; break; case 'G': alt = ALT_G; break;
        default: break;
      }
      if (alt) { uint32_t k; orc_gen_range_u32(&r, 0, 3, &k); nt = alt[k]; }
    }
    out[i] = nt;
  }
  return 0;
}

/* --------------------------------------------------- AbundanceProfile impls */

/* uniform.rs:18-36 */
void orc_uniform_determine_abundances(uint64_t total_reads, uint64_t num_genomes,
                                      uint64_t* reads_out, double* abund_out) {
  uint64_t per = (uint64_t)ceil((double)total_reads / (double)num_genomes);
  double a = 100.0 / (double)num_genomes;
  for (uint64_t i = 0; i < num_genomes; i++) { reads_out[i] = per; abund_out[i] = a; }
}
/* exact.rs:17-24 */
void orc_exact_determine_abundances(uint64_t total_reads, uint64_t num_genomes,
                                    uint64_t* reads_out, double* abund_out) {
  double a = 100.0 / (double)num_genomes;
  for (uint64_t i = 0; i < num_genomes; i++) { reads_out[i] = total_reads; abund_out[i] = a; }
}
/* custom.rs:20-49 */
void orc_custom_determine_abundances(const double* abundances, uint64_t total_reads,
                                     uint64_t num_genomes, uint64_t* reads_out, double* abund_out) {
  double total = 0.0;
  for (uint64_t i = 0; i < num_genomes; i++) total += abundances[i];
  if (total < 0.99 || total > 1.01) {
    for (uint64_t i = 0; i < num_genomes; i++) {
      reads_out[i] = (uint64_t)ceil((double)total_reads * (abundances[i] / total));
      abund_out[i] = abundances[i] / total;
    }
  } else {
    for (uint64_t i = 0; i < num_genomes; i++) {
      reads_out[i] = (uint64_t)ceil((double)total_reads * abundances[i]);
      abund_out[i] = abundances[i];
    }
  }
}
/* uniform.rs:79-94 == custom.rs:80-95 */
void orc_adjust_for_size(const uint64_t* genome_sizes, const uint64_t* reads_in,
                         const double* abund_in, uint64_t num_genomes, uint64_t* reads_out,
                         double* abund_out) {
  double total_reads = 0.0, total_adjusts = 0.0;
  for (uint64_t i = 0; i < num_genomes; i++) total_reads += (double)reads_in[i];
  for (uint64_t i = 0; i < num_genomes; i++) total_adjusts += (double)genome_sizes[i] * abund_in[i];
  for (uint64_t i = 0; i < num_genomes; i++) {
    reads_out[i] = (uint64_t)ceil(total_reads * ((abund_in[i] * (double)genome_sizes[i]) / total_adjusts));
    abund_out[i] = abund_in[i];
  }
}

/* ---------------------------------------------------------------- simulate.rs */

/* simulate.rs:172-184: rng = StdRng::seed_from_u64(seed); per pair:
 * gen_range(0..num_seqs) then gen::<u64>(). */
int orc_pe_outer(uint64_t n_contigs, uint64_t seed, uint64_t first, uint64_t count,
                 uint32_t* contig_idx, uint64_t* pe_seed, uint64_t* slots) {
  if (n_contigs == 0) FAIL(SIMMR_EGENOME, "genome has no sequences");
  orc_rng r; orc_rng_seed_from_u64(&r, seed);
  for (uint64_t i = 0; i < first + count; i++) {
    uint64_t idx;
    orc_gen_range_u64(&r, 0, n_contigs, &idx);
    uint64_t s = orc_next_u64(&r);
    if (i >= first) { contig_idx[i - first] = (uint32_t)idx; pe_seed[i - first] = s; }
  }
  if (slots) *slots = r.words_used / 2;
  return 0;
}

/* SIMMR_RNG_PHILOX_FULL (include/simmr_hip.h): the outer stream as one Philox block per pair — pair i of the genome's run
 * takes the block with key = seed and counter (i & 0xffffffff, 4 | (i >> 32) << 8, 'simm', 'r\0\0\3') = (w0, w1, w2, w3):
 * contig = ((w0 | w1 << 32) * n_contigs) >> 64, pe_seed = w2 | w3 << 32. */
int orc_pe_outer_ctr(uint64_t n_contigs, uint64_t seed, uint64_t first, uint64_t count, uint32_t* contig_idx, uint64_t* pe_seed) {
  if (n_contigs == 0) FAIL(SIMMR_EGENOME, "genome has no sequences");
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (uint64_t i = first; i < first + count; i++) {
    const uint32_t ctr[4] = {(uint32_t)i, 4u | ((uint32_t)(i >> 32) << 8), 0x73696D6Du, 0x72000003u};
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    const unsigned __int128 m = (unsigned __int128)((uint64_t)w[0] | ((uint64_t)w[1] << 32)) * n_contigs;
    contig_idx[i - first] = (uint32_t)(uint64_t)(m >> 64);
    pe_seed[i - first] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  }
  return 0;
}

/* simulate.rs:211-258 and the two Option<u64> draws at :266,:270 */
static int pe_plan_pair_draws(const simmr_error_profile* p, uint64_t size, uint64_t pe_seed, orc_pe_plan* pl);
int orc_pe_plan_pair(const simmr_error_profile* p, uint64_t size, uint64_t pe_seed, orc_pe_plan* pl) {
  /* SIMMR_RNG_PHILOX_FULL: every generator made below is W(its seed) instead of StdRng */
  orc_set_stream_kind(p->rng_mode == SIMMR_RNG_PHILOX_FULL);
  const int rc = pe_plan_pair_draws(p, size, pe_seed, pl);
  orc_set_stream_kind(0);
  return rc;
}
static int pe_plan_pair_draws(const simmr_error_profile* p, uint64_t size, uint64_t pe_seed, orc_pe_plan* pl) {
  uint16_t L16, I16, req16;
  int rc;
  if ((rc = orc_profile_get_read_length(p, pe_seed, &L16))) return rc;  /* :211 */
  if ((rc = orc_profile_get_insert_size(p, pe_seed, &I16))) return rc;  /* :212 */
  if ((rc = orc_profile_minimum_genome_size(p, &req16))) return rc;     /* :216 */
  uint64_t L = L16, I = I16, required = req16;
  if (size <= required)                                                 /* :220-225 */
    FAIL(SIMMR_EGENOME, "Genome size (%llunt) is smaller than the required length (%llu)",
         (unsigned long long)size, (unsigned long long)required);
  orc_rng r; orc_rng_seed_from_u64(&r, pe_seed);                        /* :227-230 */
  uint64_t fwd_start;
  orc_gen_range_u64(&r, 0, size - required, &fwd_start);                /* :233 */
  uint64_t fwd_end = fwd_start + L;
  uint64_t rev_end, rev_start;
  uint8_t flags = SIMMR_FLAG_REVCOMP;
  if (fwd_start + I >= size || fwd_start + I + L >= size) {             /* :241-247 */
    uint64_t nre;
    if (orc_gen_range_u64(&r, fwd_start, size - required, &nre)) FAIL(SIMMR_ERANGE, "empty re-draw range");
    rev_end = nre; rev_start = nre + L;
    flags |= SIMMR_FLAG_REDRAWN;
  } else if ((int32_t)((uint32_t)(fwd_start + I) - (uint32_t)L) < 0) {  /* :250-251 */
    rev_end = 0; rev_start = L;
  } else {                                                              /* :253-256 */
    rev_end = fwd_start + I - L; rev_start = fwd_start + I;
  }
  /* :266 and :270: two `rng.gen::<Option<u64>>()` draws, in this order */
  uint64_t qs, ms;
  if (!orc_gen_option_u64(&r, &qs)) { qs = orc_entropy_substitute(pe_seed, 1); flags |= SIMMR_FLAG_QSEED_SUBST; }
  if (!orc_gen_option_u64(&r, &ms)) { ms = orc_entropy_substitute(pe_seed, 2); flags |= SIMMR_FLAG_MSEED_SUBST; }
  pl->read_length = (uint32_t)L; pl->insert_size = (uint32_t)I;
  pl->fwd_start = fwd_start; pl->fwd_end = fwd_end; pl->rev_end = rev_end; pl->rev_start = rev_start;
  pl->qseed2 = qs; pl->mseed2 = ms; pl->flags2 = flags;
  return 0;
}

static void put_meta(const simmr_reads_out* o, uint64_t r, uint64_t start, uint64_t end,
                     uint32_t contig, uint32_t genome, uint32_t id, uint8_t flags) {
  if (o->start) o->start[r] = start;
  if (o->end) o->end[r] = end;
  if (o->contig) o->contig[r] = contig;
  if (o->genome) o->genome[r] = genome;
  if (o->read_id) o->read_id[r] = id;
  if (o->flags) o->flags[r] = flags;
}

/* simulate.rs:260-299 for one planned pair; writes reads 2*k and 2*k+1. */
static int pe_emit_pair(const orc_genome* g, const simmr_error_profile* p, uint32_t contig,
                        uint64_t pe_seed, const orc_pe_plan* pl, uint64_t k, uint32_t id,
                        const simmr_reads_out* o) {
  const uint8_t* seq = g->seq[contig];
  uint64_t len = g->len[contig], L = pl->read_length;
  if (pl->fwd_end > len || pl->rev_start > len) FAIL(SIMMR_ERANGE, "slice out of bounds (Rust panic)");
  uint64_t o1 = o->seq_off[2 * k], o2 = o->seq_off[2 * k + 1];
  uint8_t* tmp = (uint8_t*)malloc(L ? L : 1);
  if (!tmp) FAIL(SIMMR_ENOMEM, "oom");
  int rc;
  if (p->rng_mode != SIMMR_RNG_REFERENCE && p->kind == SIMMR_MINIMAL_SHORT) {
    /* counter modes (philox.c): one key per read = its Phred seed */
    orc_philox_read(p, seq + pl->fwd_start, L, pe_seed, o->qual + o1, o->seq + o1);
    orc_philox_read(p, seq + pl->rev_end, L, pl->qseed2, o->qual + o2, tmp);
    orc_reverse_complement(tmp, L, o->seq + o2);
    free(tmp);
    if (o->qual_offset)
      for (uint64_t i = 0; i < 2 * L; i++) o->qual[o1 + i] = (uint8_t)(o->qual[o1 + i] + o->qual_offset);
    put_meta(o, 2 * k, pl->fwd_start, pl->fwd_end, contig, 0, id, 0);
    put_meta(o, 2 * k + 1, pl->rev_start, pl->rev_end, contig, 0, id, (uint8_t)(pl->flags2 & ~SIMMR_FLAG_MSEED_SUBST));
    return 0;
  }
  /* forward mate: quality :265, mutations :269 — both re-seeded with pe_seed */
  rc = orc_profile_simulate_phred_scores(p, L, pe_seed, o->qual + o1);
  if (!rc) rc = orc_profile_simulate_point_mutations(p, seq + pl->fwd_start, o->qual + o1, L, pe_seed, o->seq + o1);
  /* reverse mate: quality :266, mutations :270, then reverse complement :283 */
  if (!rc) rc = orc_profile_simulate_phred_scores(p, L, pl->qseed2, o->qual + o2);
  if (!rc) rc = orc_profile_simulate_point_mutations(p, seq + pl->rev_end, o->qual + o2, L, pl->mseed2, tmp);
  if (!rc) orc_reverse_complement(tmp, L, o->seq + o2);
  free(tmp);
  if (rc) return rc;
  if (o->qual_offset)
    for (uint64_t i = 0; i < 2 * L; i++) o->qual[o1 + i] = (uint8_t)(o->qual[o1 + i] + o->qual_offset);
  uint8_t f2 = pl->flags2;
  if (p->kind == SIMMR_PERFECT_SHORT) f2 &= (uint8_t)~(SIMMR_FLAG_QSEED_SUBST | SIMMR_FLAG_MSEED_SUBST);
  if (p->kind == SIMMR_CUSTOM) f2 &= (uint8_t)~SIMMR_FLAG_MSEED_SUBST; /* the mutation seed is drawn but unused */
  put_meta(o, 2 * k, pl->fwd_start, pl->fwd_end, contig, 0, id, 0);              /* :287-292 */
  put_meta(o, 2 * k + 1, pl->rev_start, pl->rev_end, contig, 0, id, f2);          /* :293-298 */
  return 0;
}

int orc_simulate_pe_reads_from_genome(const orc_genome* g, const simmr_error_profile* p,
                                      uint64_t genome_reads, uint64_t seed, uint64_t first,
                                      uint64_t count, uint32_t read_id_base,
                                      const simmr_reads_out* out, uint64_t* total_bases,
                                      int threads) {
  if (p->rng_mode == SIMMR_RNG_PHILOX_FULL && p->kind != SIMMR_MINIMAL_SHORT)
    FAIL(SIMMR_EINVAL, "SIMMR_RNG_PHILOX_FULL on the paired-end path: minimal-short only");
  uint64_t n_pairs = genome_reads / 2; /* simulate.rs:179 */
  if (first > n_pairs) first = n_pairs;
  if (count > n_pairs - first) count = n_pairs - first;
  if (out->reads_capacity < 2 * count) FAIL(SIMMR_ERANGE, "reads_capacity too small");
  uint32_t* cidx = (uint32_t*)malloc(sizeof(uint32_t) * (count ? count : 1));
  uint64_t* seeds = (uint64_t*)malloc(sizeof(uint64_t) * (count ? count : 1));
  orc_pe_plan* plans = (orc_pe_plan*)malloc(sizeof(orc_pe_plan) * (count ? count : 1));
  if (!cidx || !seeds || !plans) { free(cidx); free(seeds); free(plans); FAIL(SIMMR_ENOMEM, "oom"); }
  int rc = p->rng_mode == SIMMR_RNG_PHILOX_FULL ? orc_pe_outer_ctr(g->n_contigs, seed, first, count, cidx, seeds)
                                                : orc_pe_outer(g->n_contigs, seed, first, count, cidx, seeds, NULL);
  int err = 0;
  if (!rc) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) if (threads > 1)
    for (int64_t k = 0; k < (int64_t)count; k++) {
      int r2 = orc_pe_plan_pair(p, g->size[cidx[k]], seeds[k], &plans[k]);
      if (r2) {
#pragma omp critical
        { if (!err) err = r2; }
      }
    }
    rc = err;
  }
  if (!rc) {
    uint64_t off = 0;
    for (uint64_t k = 0; k < count; k++) {
      out->seq_off[2 * k] = off; off += plans[k].read_length;
      out->seq_off[2 * k + 1] = off; off += plans[k].read_length;
    }
    out->seq_off[2 * count] = off;
    if (total_bases) *total_bases = off;
    if (off > out->seq_capacity) { rc = SIMMR_ERANGE; snprintf(g_err, sizeof g_err, "seq_capacity too small"); }
  }
  if (!rc) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) if (threads > 1)
    for (int64_t k = 0; k < (int64_t)count; k++) {
      /* simulate.rs:274: ids are handed out in generation order */
      int r2 = pe_emit_pair(g, p, cidx[k], seeds[k], &plans[k], (uint64_t)k,
                            read_id_base + (uint32_t)(first + (uint64_t)k), out);
      if (r2) {
#pragma omp critical
        { if (!err) err = r2; }
      }
    }
    rc = err;
  }
  free(cidx); free(seeds); free(plans);
  return rc;
}

/* ---------------------------------------------------------------- long reads */

typedef struct long_unit {
  uint32_t genome, contig;
  uint64_t read_seed;
  uint32_t read_length; /* the drawn length (u16) */
  uint64_t start, end;
} long_unit;

/* simulate.rs:478-491 */
static int long_window(uint64_t size, uint32_t read_length, uint64_t read_seed, int uniform_start, uint64_t* start,
                       uint64_t* end) {
  if (size <= read_length) FAIL(SIMMR_EGENOME, "Genome size is smaller than the read length"); /* :471-476 */
  orc_rng r; orc_rng_seed_from_u64(&r, read_seed);
  uint64_t s, e;
  /* :484 draws in [0, read_length); SIMMR_START_UNIFORM (an extension) in [0, size - read_length) */
  if (orc_gen_range_u64(&r, 0, uniform_start ? size - read_length : (uint64_t)read_length, &s))
    FAIL(SIMMR_ERANGE, "read_length == 0 (Rust panic)");
  e = s + read_length;                                                                    /* :485 */
  if (e >= size) orc_gen_range_u64(&r, s, size, &e);                                      /* :488-491 */
  *start = s; *end = e;
  return 0;
}

/* bench.py's cpu_baseline only ("faithful cost", BASELINE.md section 3 run B4): when set, every long read also
 * pays what the reference pays on top of the algorithm — `usable_seqs` is a Vec of CLONES of every sequence longer
 * than the read (simulate.rs:362-367) and the chosen one is cloned again (:375).  Results do not change. */
static int g_faithful_cost = 0;
void orc_set_faithful_cost(int on) { g_faithful_cost = on; }
static void pay_reference_clones(const orc_genome* G, uint32_t chosen, uint32_t read_length) {
  for (uint32_t c = 0; c <= G->n_contigs; c++) {
    uint32_t src = c < G->n_contigs ? c : chosen;
    if (c < G->n_contigs && !(G->size[c] > read_length)) continue;
    uint8_t* copy = (uint8_t*)malloc(G->len[src] ? G->len[src] : 1);
    if (!copy) continue;
    memcpy(copy, G->seq[src], G->len[src]);
    __asm__ volatile("" : : "r"(copy) : "memory"); /* the copy is made even though nobody reads it */
    free(copy);
  }
}

int orc_simulate_long_reads(const orc_genome* genomes, uint32_t n_genomes,
                            const uint64_t* genome_reads, const simmr_error_profile* p,
                            int has_seed, uint64_t seed, uint64_t first, uint64_t count,
                            uint32_t read_id_base, const simmr_reads_out* out,
                            uint64_t* total_bases, uint32_t* const_len, int threads) {
  uint64_t total = 0;
  for (uint32_t g = 0; g < n_genomes; g++) total += genome_reads[g];
  if (first > total) first = total;
  if (count > total - first) count = total - first;
  if (out->reads_capacity < count) FAIL(SIMMR_ERANGE, "reads_capacity too small");
  int per_read = (!has_seed) || p->length_mode == SIMMR_LEN_PER_READ;
  if (p->kind == SIMMR_CUSTOM && !orc_profile_is_long_read(p)) FAIL(SIMMR_EINVAL, "a short-read custom model on the long-read path");
  if (p->rng_mode == SIMMR_RNG_PHILOX_FULL && (p->kind == SIMMR_CUSTOM || !per_read))
    FAIL(SIMMR_EINVAL, "SIMMR_RNG_PHILOX_FULL: minimal-long / perfect-long with per-read lengths only");
  long_unit* units = (long_unit*)calloc(count ? count : 1, sizeof(long_unit));
  if (!units) FAIL(SIMMR_ENOMEM, "oom");
  int rc = 0;
  if (const_len) *const_len = 0;

  if (!per_read) {
    /* simulate.rs:348-388 with Some(seed): get_random_read_length(seed) is one
     * value for the whole run (:358); one StdRng across all genomes (:348). */
    uint16_t L0;
    rc = orc_profile_get_random_read_length(p, seed, &L0);
    if (!rc) {
      if (const_len) *const_len = L0;
      orc_rng r; orc_rng_seed_from_u64(&r, seed);
      uint64_t gi = 0; /* global read index */
      for (uint32_t g = 0; g < n_genomes && !rc; g++) {
        const orc_genome* G = &genomes[g];
        /* :362-367 usable_seqs = contigs with size > read_length, original order */
        uint32_t n_us = 0;
        uint32_t* us = (uint32_t*)malloc(sizeof(uint32_t) * (G->n_contigs ? G->n_contigs : 1));
        for (uint32_t c = 0; c < G->n_contigs; c++) if (G->size[c] > L0) us[n_us++] = c;
        if (genome_reads[g] > 0 && n_us == 0) {
          free(us);
          rc = SIMMR_EGENOME;
          snprintf(g_err, sizeof g_err, "genome %u: no sequence longer than read length %u (reference loops forever, simulate.rs:370)", g, L0);
          break;
        }
        for (uint64_t k = 0; k < genome_reads[g]; k++, gi++) {
          if (gi >= first + count) break;
          uint64_t idx; orc_gen_range_u64(&r, 0, n_us, &idx);  /* :375 */
          uint64_t rs = orc_next_u64(&r);                       /* :378 */
          if (gi >= first) {
            long_unit* u = &units[gi - first];
            u->genome = g; u->contig = us[idx]; u->read_seed = rs; u->read_length = L0;
          }
        }
        free(us);
      }
    }
  } else {
    /* Extension (include/simmr_hip.h SIMMR_LEN_PER_READ): what the reference
     * does without --seed — every call draws fresh entropy — made reproducible
     * with a private StdRng per read, keyed by mix(seed, global read index). */
    uint64_t gi = 0;
    for (uint32_t g = 0; g < n_genomes && !rc; g++) {
      const orc_genome* G = &genomes[g];
      uint64_t maxsz = 0;
      for (uint32_t c = 0; c < G->n_contigs; c++) if (G->size[c] > maxsz) maxsz = G->size[c];
      for (uint64_t k = 0; k < genome_reads[g]; k++, gi++) {
        if (gi >= first + count) break;
        if (gi < first) continue;
        orc_set_stream_kind(p->rng_mode == SIMMR_RNG_PHILOX_FULL);  /* the read's own generator: W(seed) in the full counter mode */
        orc_rng r; orc_rng_seed_from_u64(&r, orc_per_read_seed(seed, gi));
        orc_set_stream_kind(0);
        long_unit* u = &units[gi - first];
        for (int tries = 0;; tries++) {
          uint32_t L;
          if (p->kind == SIMMR_CUSTOM) { /* get_random_read_length of the custom profile (custom_short.rs:286-301) */
            const orc_model* m = orc_custom_model(custom_of(p));
            L = sat_u16_f64(floor(orc_normal_f64(&r, m->read_length_mean, m->read_length_std)));
          } else {
            float gl;
            if (orc_gamma_f32(&r, p->gamma_shape, p->gamma_scale, &gl)) { rc = SIMMR_EINVAL; break; }
            L = sat_u16_f32(floorf(gl));
          }
          if (L == 0 || maxsz <= L) {
            if (tries > 1000) { rc = SIMMR_EGENOME; snprintf(g_err, sizeof g_err, "no usable sequence"); break; }
            continue; /* :370-372 try a new length */
          }
          uint64_t n_us = 0;
          for (uint32_t c = 0; c < G->n_contigs; c++) if (G->size[c] > L) n_us++;
          uint64_t idx; orc_gen_range_u64(&r, 0, n_us, &idx);
          uint64_t rs = orc_next_u64(&r);
          uint32_t c = 0;
          for (uint64_t seen = 0; c < G->n_contigs; c++) if (G->size[c] > L) { if (seen == idx) break; seen++; }
          u->genome = g; u->contig = c; u->read_seed = rs; u->read_length = L;
          break;
        }
        if (rc) break;
      }
    }
  }

  int err = 0;
  if (!rc) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) if (threads > 1)
    for (int64_t k = 0; k < (int64_t)count; k++) {
      long_unit* u = &units[k];
      orc_set_stream_kind(p->rng_mode == SIMMR_RNG_PHILOX_FULL);
      int r2 = long_window(genomes[u->genome].size[u->contig], u->read_length, u->read_seed,
                           p->long_start_mode == SIMMR_START_UNIFORM, &u->start, &u->end);
      orc_set_stream_kind(0);
      if (r2) {
#pragma omp critical
        { if (!err) err = r2; }
      }
    }
    rc = err;
  }
  if (!rc) {
    uint64_t off = 0;
    for (uint64_t k = 0; k < count; k++) { out->seq_off[k] = off; off += units[k].end - units[k].start; }
    out->seq_off[count] = off;
    if (total_bases) *total_bases = off;
    if (off > out->seq_capacity) { rc = SIMMR_ERANGE; snprintf(g_err, sizeof g_err, "seq_capacity too small"); }
  }
  if (!rc) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1) if (threads > 1)
    for (int64_t k = 0; k < (int64_t)count; k++) {
      const long_unit* u = &units[k];
      const orc_genome* G = &genomes[u->genome];
      uint64_t n = u->end - u->start, o1 = out->seq_off[k];
      int r2 = 0;
      if (g_faithful_cost) pay_reference_clones(G, u->contig, u->read_length);
      if (u->end > G->len[u->contig]) { r2 = SIMMR_ERANGE; }
      /* :497 quality over end-start; :500 simulate_errors = copy; :503 mutations */
      if (!r2 && p->rng_mode != SIMMR_RNG_REFERENCE && (p->kind == SIMMR_MINIMAL_LONG || p->kind == SIMMR_PERFECT_LONG)) {
        orc_philox_read(p, G->seq[u->contig] + u->start, n, u->read_seed, out->qual + o1, out->seq + o1);
      } else if (!r2 && p->kind == SIMMR_CUSTOM) {
        /* :497 quality; :500 simulate_errors (the k-mer splice); :503 simulate_point_mutations = copy */
        r2 = orc_profile_simulate_phred_scores(p, n, u->read_seed, out->qual + o1);
        if (!r2) {
          int64_t m = (p->rng_mode == SIMMR_RNG_PHILOX ? orc_custom_simulate_errors_philox : orc_custom_simulate_errors)(
              orc_custom_model(custom_of(p)), G->seq[u->contig] + u->start, n, u->read_seed, out->seq + o1);
          /* < 0: the reference panics; < n: a deletion in the last k-mer leaves fewer bases than qualities */
          if (m != (int64_t)n) r2 = SIMMR_ERANGE;
        }
      } else {
      if (!r2) r2 = orc_profile_simulate_phred_scores(p, n, u->read_seed, out->qual + o1);
      if (!r2) r2 = orc_profile_simulate_point_mutations(p, G->seq[u->contig] + u->start, out->qual + o1, n, u->read_seed, out->seq + o1);
      }
      if (!r2 && out->qual_offset)
        for (uint64_t i = 0; i < n; i++) out->qual[o1 + i] = (uint8_t)(out->qual[o1 + i] + out->qual_offset);
      if (!r2) put_meta(out, (uint64_t)k, u->start, u->end, u->contig, u->genome,
                        read_id_base + (uint32_t)(first + (uint64_t)k), 0);
      if (r2) {
#pragma omp critical
        { if (!err) err = r2; }
      }
    }
    rc = err;
  }
  free(units);
  return rc;
}
