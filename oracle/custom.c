/*
 * custom.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * Restatement of the custom (empirical) error profile:
 *   simmr/src/error_profiles/custom_short.rs   CustomPDF, CustomShortErrorProfile
 *   shared/src/encoding.rs:82-117,244-281      Bins, ErrorModelParams, bincode (de)serialisation
 *   shared/src/encoding.rs:146-210             3-bit k-mer encode / decode
 * and of the crate arithmetic it calls (not under /root/reference):
 *   rand 0.8.5   Uniform<u32>::new / new_inclusive + sample, Uniform<f64>::new + sample
 *   rand_distr 0.4.3  WeightedAliasIndex<f64|f32>::new + sample (Vose alias method)
 *   bincode 1.3.3 default options: little-endian, fixed-width ints, u64 lengths
 * Parity of this file is unpinned beyond the reference's own test
 * custom_long.rs:300-343 ("ACCCG" -> "CATGT"), see tests/test_oracle_custom.py.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---------------------------------------------------------------- rand Uniform */

/* UniformInt<u32>::new_inclusive(low, high) */
void orc_uniform_u32_new_inclusive(uint32_t low, uint32_t high, orc_uniform_u32* u) {
  uint32_t range = high - low + 1u; /* wrapping; 0 means the full u32 range */
  uint32_t ints_to_reject = range > 0 ? (uint32_t)((0xFFFFFFFFu - range + 1u) % range) : 0;
  u->low = low; u->range = range; u->z = ints_to_reject;
}
/* UniformInt<u32>::sample */
uint32_t orc_uniform_u32_sample(const orc_uniform_u32* u, orc_rng* r) {
  if (u->range > 0) {
    const uint32_t zone = 0xFFFFFFFFu - u->z;
    for (;;) {
      uint32_t v = orc_next_u32(r);
      uint64_t m = (uint64_t)v * u->range;
      if ((uint32_t)m <= zone) return u->low + (uint32_t)(m >> 32);
    }
  }
  return orc_next_u32(r);
}
static double f64_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t bits_f64(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
/* UniformFloat<f64>::new(low, high) */
void orc_uniform_f64_new(double low, double high, orc_uniform_f64* u) {
  const double max_rand = f64_bits((0xFFFFFFFFFFFFFFFFULL >> 12) | 0x3FF0000000000000ULL) - 1.0;
  double scale = high - low;
  while (scale * max_rand + low >= high) scale = f64_bits(bits_f64(scale) - 1);
  u->low = low; u->scale = scale;
}
/* UniformFloat<f64>::sample */
double orc_uniform_f64_sample(const orc_uniform_f64* u, orc_rng* r) {
  double v12 = f64_bits((orc_next_u64(r) >> 12) | 0x3FF0000000000000ULL);
  double v01 = v12 - 1.0;
  return v01 * u->scale + u->low;
}
static float f32_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t bits_f32(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static void uniform_f32_new(float low, float high, float* lo, float* sc) {
  const float max_rand = f32_bits((0xFFFFFFFFu >> 9) | 0x3F800000u) - 1.0f;
  float scale = high - low;
  while (scale * max_rand + low >= high) scale = f32_bits(bits_f32(scale) - 1);
  *lo = low; *sc = scale;
}

/* ------------------------------------------------- rand_distr WeightedAliasIndex */

static double pairwise_sum_f64(const double* v, size_t n) {
  if (n <= 32) { double s = 0.0; for (size_t i = 0; i < n; i++) s += v[i]; return s; }
  size_t mid = n / 2;
  return pairwise_sum_f64(v, mid) + pairwise_sum_f64(v + mid, n - mid);
}

/* WeightedAliasIndex<f64>::new(weights) */
int orc_alias_new(const double* weights, uint32_t n, orc_alias* a) {
  memset(a, 0, sizeof *a);
  if (n == 0) return -1;
  double wsum = pairwise_sum_f64(weights, n);
  if (wsum > 1.7976931348623157e308) wsum = 1.7976931348623157e308;
  if (wsum == 0.0) return -2;
  for (uint32_t i = 0; i < n; i++) if (!(weights[i] >= 0.0)) return -3;
  a->n = n;
  a->odds = (double*)malloc(sizeof(double) * n);
  a->aliases = (uint32_t*)calloc(n, sizeof(uint32_t));
  const double nd = (double)n;
  for (uint32_t i = 0; i < n; i++) a->odds[i] = weights[i] * nd;
  /* the three intrusive stacks of rand_distr's `Aliases` */
  uint32_t smalls = 0xFFFFFFFFu, bigs = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n; i++) {
    if (a->odds[i] < wsum) { a->aliases[i] = smalls; smalls = i; }
    else { a->aliases[i] = bigs; bigs = i; }
  }
  while (smalls != 0xFFFFFFFFu && bigs != 0xFFFFFFFFu) {
    uint32_t s = smalls; smalls = a->aliases[s];
    uint32_t b = bigs; bigs = a->aliases[b];
    a->aliases[s] = b;
    a->odds[b] = a->odds[b] - wsum + a->odds[s];
    if (a->odds[b] < wsum) { a->aliases[b] = smalls; smalls = b; }
    else { a->aliases[b] = bigs; bigs = b; }
  }
  while (smalls != 0xFFFFFFFFu) { uint32_t s = smalls; smalls = a->aliases[s]; a->odds[s] = wsum; }
  while (bigs != 0xFFFFFFFFu) { uint32_t b = bigs; bigs = a->aliases[b]; a->odds[b] = wsum; }
  orc_uniform_u32_new_inclusive(0, n - 1, &a->uniform_index); /* Uniform::new(0, n) */
  orc_uniform_f64_new(0.0, wsum, &a->uniform_weight);
  return 0;
}
void orc_alias_free(orc_alias* a) { free(a->odds); free(a->aliases); memset(a, 0, sizeof *a); }
/* WeightedAliasIndex::sample */
uint32_t orc_alias_sample(const orc_alias* a, orc_rng* r) {
  uint32_t c = orc_uniform_u32_sample(&a->uniform_index, r);
  if (orc_uniform_f64_sample(&a->uniform_weight, r) < a->odds[c]) return c;
  return a->aliases[c];
}

float orc_pairwise_sum_f32(const float* v, uint32_t n);
/* WeightedAliasIndex<f32> one-shot: new(weights) then sample (custom_short.rs:497-503) */
/* the conditions under which WeightedAliasIndex::<f32>::new(weights) is an Err (which the reference unwraps) */
static int alias_f32_check(const float* w, uint32_t n, float* wsum_out) {
  if (n == 0) return -1;
  /* WeightedAliasIndex::new: every weight in [0, f32::MAX / n] (NaN fails), the sum clamped to f32::MAX, not 0 */
  const float maxw = 3.40282347e38f / (float)n;
  for (uint32_t i = 0; i < n; i++) if (!(0.0f <= w[i] && w[i] <= maxw)) return -3;
  float wsum = orc_pairwise_sum_f32(w, n);
  if (wsum > 3.40282347e38f) wsum = 3.40282347e38f;
  if (wsum == 0.0f) return -2;
  *wsum_out = wsum;
  return 0;
}
static int alias_f32_sample_once(const float* w, uint32_t n, orc_rng* r, uint32_t* out) {
  float wsum;
  int chk = alias_f32_check(w, n, &wsum);
  if (chk) return chk;
  float* odds = (float*)malloc(sizeof(float) * n);
  uint32_t* al = (uint32_t*)calloc(n, sizeof(uint32_t));
  const float nf = (float)n;
  for (uint32_t i = 0; i < n; i++) odds[i] = w[i] * nf;
  uint32_t smalls = 0xFFFFFFFFu, bigs = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n; i++) {
    if (odds[i] < wsum) { al[i] = smalls; smalls = i; } else { al[i] = bigs; bigs = i; }
  }
  while (smalls != 0xFFFFFFFFu && bigs != 0xFFFFFFFFu) {
    uint32_t s = smalls; smalls = al[s];
    uint32_t b = bigs; bigs = al[b];
    al[s] = b;
    odds[b] = odds[b] - wsum + odds[s];
    if (odds[b] < wsum) { al[b] = smalls; smalls = b; } else { al[b] = bigs; bigs = b; }
  }
  while (smalls != 0xFFFFFFFFu) { uint32_t s = smalls; smalls = al[s]; odds[s] = wsum; }
  while (bigs != 0xFFFFFFFFu) { uint32_t b = bigs; bigs = al[b]; odds[b] = wsum; }
  orc_uniform_u32 ui; orc_uniform_u32_new_inclusive(0, n - 1, &ui);
  float lo, sc; uniform_f32_new(0.0f, wsum, &lo, &sc);
  uint32_t c = orc_uniform_u32_sample(&ui, r);
  float v12 = f32_bits((orc_next_u32(r) >> 9) | 0x3F800000u);
  float x = (v12 - 1.0f) * sc + lo;
  *out = x < odds[c] ? c : al[c];
  free(odds); free(al);
  return 0;
}
float orc_pairwise_sum_f32(const float* v, uint32_t n) {
  if (n <= 32) { float s = 0.0f; for (uint32_t i = 0; i < n; i++) s += v[i]; return s; }
  uint32_t mid = n / 2;
  return orc_pairwise_sum_f32(v, mid) + orc_pairwise_sum_f32(v + mid, n - mid);
}

/* ------------------------------------------------------ bincode ErrorModelParams */

typedef struct { const uint8_t* p; uint64_t n, pos; int bad; } rd_t;
static uint64_t rd_u64(rd_t* r) { if (r->pos + 8 > r->n) { r->bad = 1; return 0; } uint64_t v; memcpy(&v, r->p + r->pos, 8); r->pos += 8; return v; }
static uint32_t rd_u32(rd_t* r) { if (r->pos + 4 > r->n) { r->bad = 1; return 0; } uint32_t v; memcpy(&v, r->p + r->pos, 4); r->pos += 4; return v; }
static uint8_t rd_u8(rd_t* r) { if (r->pos + 1 > r->n) { r->bad = 1; return 0; } return r->p[r->pos++]; }
static double rd_f64(rd_t* r) { uint64_t v = rd_u64(r); return f64_bits(v); }
static float rd_f32(rd_t* r) { uint32_t v = rd_u32(r); return f32_bits(v); }

static int rd_bins(rd_t* r, orc_bins* b) {
  memset(b, 0, sizeof *b);
  b->num_bins = rd_u64(r);
  b->bin_width = rd_u64(r);
  b->n_density = rd_u64(r);
  if (r->bad || b->n_density > (r->n - r->pos) / 8) { r->bad = 1; return -1; }
  b->density = (double*)malloc(sizeof(double) * (b->n_density ? b->n_density : 1));
  for (uint64_t i = 0; i < b->n_density; i++) b->density[i] = rd_f64(r);
  b->n_ranges = rd_u64(r);
  if (r->bad || b->n_ranges > (r->n - r->pos) / 8) { r->bad = 1; return -1; }
  b->range_lo = (uint32_t*)malloc(4 * (b->n_ranges ? b->n_ranges : 1));
  b->range_hi = (uint32_t*)malloc(4 * (b->n_ranges ? b->n_ranges : 1));
  for (uint64_t i = 0; i < b->n_ranges; i++) { b->range_lo[i] = rd_u32(r); b->range_hi[i] = rd_u32(r); }
  return r->bad ? -1 : 0;
}

/* bincode::deserialize::<ErrorModelParams> (shared/src/encoding.rs:268-281, struct at :102-117) */
int orc_model_parse(const uint8_t* bytes, uint64_t n, orc_model* m) {
  memset(m, 0, sizeof *m);
  rd_t r = {bytes, n, 0, 0};
  m->bin_size = rd_u64(&r);
  m->n_quality = rd_u64(&r);
  if (r.bad || m->n_quality > n) return -1;
  m->quality = (orc_bins*)calloc(m->n_quality ? m->n_quality : 1, sizeof(orc_bins));
  for (uint64_t i = 0; i < m->n_quality; i++) if (rd_bins(&r, &m->quality[i])) return -1;
  m->bit_encoding = rd_u8(&r);
  m->kmer_size = rd_u64(&r);
  m->n_prob = rd_u64(&r);
  if (r.bad || m->n_prob > n) return -1;
  m->prob_kmer = (uint32_t*)malloc(4 * (m->n_prob ? m->n_prob : 1));
  m->prob_n = (uint64_t*)malloc(8 * (m->n_prob ? m->n_prob : 1));
  m->prob_alt = (uint32_t**)calloc(m->n_prob ? m->n_prob : 1, sizeof(uint32_t*));
  m->prob_w = (float**)calloc(m->n_prob ? m->n_prob : 1, sizeof(float*));
  for (uint64_t i = 0; i < m->n_prob; i++) {
    m->prob_kmer[i] = rd_u32(&r);
    uint64_t k = rd_u64(&r);
    if (r.bad || k > n) return -1;
    m->prob_n[i] = k;
    m->prob_alt[i] = (uint32_t*)malloc(4 * (k ? k : 1));
    m->prob_w[i] = (float*)malloc(4 * (k ? k : 1));
    for (uint64_t j = 0; j < k; j++) { m->prob_alt[i][j] = rd_u32(&r); m->prob_w[i][j] = rd_f32(&r); }
  }
  m->insert_size_mean = rd_f64(&r);
  m->insert_size_std = rd_f64(&r);
  m->has_insert_bins = rd_u8(&r);
  if (m->has_insert_bins) { if (rd_bins(&r, &m->insert_bins)) return -1; }
  m->read_length_mean = rd_f64(&r);
  m->read_length_std = rd_f64(&r);
  if (rd_bins(&r, &m->read_length_bins)) return -1;
  m->is_long = rd_u8(&r);
  return (r.bad || r.pos != n) ? -1 : 0;
}

/* ------------------------------------------------------------------- CustomPDF */

/* CustomPDF::new (custom_short.rs:60-86) for ONE Bins */
int orc_pdf_new(const orc_bins* b, orc_pdf* p) {
  memset(p, 0, sizeof *p);
  if (orc_alias_new(b->density, (uint32_t)b->n_density, &p->alias)) return -1;
  p->n_bins = (uint32_t)b->n_ranges;
  p->bins = (orc_uniform_u32*)malloc(sizeof(orc_uniform_u32) * (p->n_bins ? p->n_bins : 1));
  for (uint32_t i = 0; i < p->n_bins; i++) orc_uniform_u32_new_inclusive(b->range_lo[i], b->range_hi[i], &p->bins[i]);
  return 0;
}
/* CustomPDF::sample / sample_with_index body (custom_short.rs:108-151): fresh StdRng(seed) per call */
int orc_pdf_sample(const orc_pdf* p, uint64_t seed, uint32_t* out) {
  orc_rng r; orc_rng_seed_from_u64(&r, seed);
  uint32_t bin = orc_alias_sample(&p->alias, &r);
  if (bin >= p->n_bins) return -1; /* index out of bounds panic (71 densities vs 70 ranges) */
  *out = orc_uniform_u32_sample(&p->bins[bin], &r);
  return 0;
}

/* three_bit_encode_kmer / three_bit_decode_kmer(skip_n = true) (shared/src/encoding.rs:146-210) */
static int enc3(const uint8_t* k, uint64_t n, uint32_t* out) {
  uint32_t e = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint32_t v;
    switch (k[i]) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; case 'N': v = 4; break; default: return -1; }
    e |= v << (3 * i);
  }
  *out = e;
  return 0;
}

/* CustomShortErrorProfile::simulate_errors (custom_short.rs:455-516): sequential, in-place splice.
 * out must hold len bytes; returns the new length (can only shrink) or < 0. */
/* The counter mode's draw of a visited k-mer's alternate: the specification (include/simmr_hip.h, enum simmr_rng_mode;
 * the product builds the same tables in simmr_amd/csrc/custom_model.hpp).  The law is the reference's —
 * P(alternate j) = w_j / sum(w) (custom_short.rs:497-503) — split in two levels so that the common outcome, "the k-mer
 * stays what it is", needs no table, from ONE 32-bit word X per position:
 *   level 1: X >> 8 < T24 answers "self"; T24 = 2^24 - 2^e, 2^e the smallest power of two (1 <= e <= 24) of 2^24ths that
 *            holds 1 - p_s, p_s = P(an alternate equal to the k-mer itself);
 *   level 2: Z = (X - (T24 << 8)) << (24 - e) is a full word again; m = Z n (64 bits), column c = m >> 32 of an alias table
 *            (Vose, n columns, thresholds in 2^24ths against (m & 0xffffffff) >> 8) over the residual law
 *            r_j = (p_j - [j is self] (T24 / 2^24) p_j / p_s) / (1 - T24 / 2^24).
 * All in f64, sums in list order.  A k-mer with an N has no "self" (its alternates with an N are errors, not draws). */
static uint32_t ctr_splice_tables(const uint32_t* alt, const float* w, uint32_t n, uint32_t self_code, int has_self,
                                  uint32_t* thr, uint32_t* alias) {
  double W = 0.0;
  for (uint32_t j = 0; j < n; j++) W += (double)w[j];
  double ps = 0.0;
  if (has_self) for (uint32_t j = 0; j < n; j++) if (alt[j] == self_code) ps += (double)w[j] / W;
  /* level 2 takes 2^e of the 2^24 level-1 values, the smallest power of two that holds 1 - p_s (1 <= e <= 24): a draw
   * that lands there is rescaled to a full word by a shift */
  const double need = (1.0 - ps) * 16777216.0;
  uint32_t e = 1;
  while (e < 24u && (double)(1u << e) < need) e++;
  const uint32_t T24 = 16777216u - (1u << e);
  const double lvl1 = (double)T24 / 16777216.0, rest = 1.0 - lvl1;
  double* odds = (double*)malloc(sizeof(double) * (n ? n : 1));
  for (uint32_t j = 0; j < n; j++) {
    const double p = (double)w[j] / W;
    double q = (has_self && alt[j] == self_code && ps > 0.0) ? p - lvl1 * (p / ps) : p;
    if (q < 0.0) q = 0.0;
    odds[j] = q / rest * (double)n;
  }
  uint32_t smalls = 0xFFFFFFFFu, bigs = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n; i++) {
    if (odds[i] < 1.0) { alias[i] = smalls; smalls = i; } else { alias[i] = bigs; bigs = i; }
  }
  while (smalls != 0xFFFFFFFFu && bigs != 0xFFFFFFFFu) {
    const uint32_t sm = smalls; smalls = alias[sm];
    const uint32_t g = bigs; bigs = alias[g];
    alias[sm] = g;
    odds[g] = odds[g] - 1.0 + odds[sm];
    if (odds[g] < 1.0) { alias[g] = smalls; smalls = g; } else { alias[g] = bigs; bigs = g; }
  }
  while (smalls != 0xFFFFFFFFu) { const uint32_t sm = smalls; smalls = alias[sm]; odds[sm] = 1.0; alias[sm] = sm; }
  while (bigs != 0xFFFFFFFFu) { const uint32_t g = bigs; bigs = alias[g]; odds[g] = 1.0; alias[g] = g; }
  for (uint32_t c = 0; c < n; c++) {
    double v = floor(odds[c] * 16777216.0 + 0.5);
    if (v > 16777216.0) v = 16777216.0;
    if (!(v >= 0.0)) v = 0.0;
    thr[c] = (uint32_t)v;
  }
  free(odds);
  return T24;
}

/* the same tables for the test tree (tests/test_oracle_custom.py: the law they encode, enumerated exactly) */
uint32_t orc_ctr_splice_tables(const uint32_t* alt, const float* w, uint32_t n, uint32_t self_code, int has_self,
                               uint32_t* thr, uint32_t* alias) {
  return ctr_splice_tables(alt, w, n, self_code, has_self, thr, alias);
}

/* `philox` != 0: SIMMR_RNG_PHILOX for a custom long-read model — the walk is the reference's; the alternate of the k-mer
 * visited at position i is drawn from ONE word, X = word i & 3 of the Philox4x32-10 block with key = the read's seed and
 * counter (i >> 2, 2, 'simm', 'r\0\0\3'), through the two levels of ctr_splice_tables.  Tolerance parity: the law of every
 * draw is the reference's, the bits are not. */
static int64_t simulate_errors_walk(const orc_model* m, const uint8_t* seq, uint64_t len, uint64_t seed, uint8_t* out, int philox) {
  orc_rng r; orc_rng_seed_from_u64(&r, seed);
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  memcpy(out, seq, len);
  uint64_t cur = len;
  const uint64_t k = m->kmer_size;
  for (uint64_t i = 0; i < len; i++) {
    if (i + k > len) break;          /* custom_short.rs:475-477: bound uses the ORIGINAL length */
    if (i + k > cur) return -1;      /* slice panic in the reference once the sequence shrank */
    uint32_t code;
    if (enc3(out + i, k, &code)) continue;
    uint64_t e = m->n_prob;
    for (uint64_t j = 0; j < m->n_prob; j++) if (m->prob_kmer[j] == code) e = j; /* HashMap: last insert wins */
    if (e == m->n_prob) continue;
    uint32_t pick;
    uint32_t alt;
    if (philox) {
      const uint32_t n = (uint32_t)m->prob_n[e];
      float wsum;
      if (alias_f32_check(m->prob_w[e], n, &wsum)) return -2;  /* the same lists are unusable in both modes */
      const uint32_t ctr[4] = {(uint32_t)(i >> 2), 2u, 0x73696D6Du, 0x72000003u};
      uint32_t w4[4];
      orc_philox4x32_10(ctr, key, w4);
      const uint32_t X = w4[i & 3];
      int has_n = 0;
      for (uint64_t j = 0; j < k; j++) has_n |= ((code >> (3 * j)) & 7u) == 4u;
      uint32_t* thr = (uint32_t*)malloc(4 * n);
      uint32_t* al = (uint32_t*)malloc(4 * n);
      uint32_t* codes = (uint32_t*)malloc(4 * n);  /* what of an alternate is ever decoded: its first k fields */
      const uint32_t kmask = k >= 10 ? 0x3fffffffu : ((1u << (3 * k)) - 1u);
      for (uint32_t j = 0; j < n; j++) codes[j] = m->prob_alt[e][j] & kmask;
      const uint32_t T24 = ctr_splice_tables(codes, m->prob_w[e], n, code, !has_n, thr, al);
      free(codes);
      if ((X >> 8) < T24) {
        alt = code;
      } else {
        uint32_t eb = 0;  /* the level-2 region holds 2^eb level-1 values: 2^24 - T24 */
        while ((1u << eb) < 16777216u - T24) eb++;
        const uint32_t Z = (X - (T24 << 8)) << (24u - eb);
        const uint64_t mm = (uint64_t)Z * n;
        const uint32_t c = (uint32_t)(mm >> 32), f = (uint32_t)mm >> 8;
        alt = m->prob_alt[e][f < thr[c] ? c : al[c]];
      }
      free(thr); free(al);
    } else {
      if (alias_f32_sample_once(m->prob_w[e], (uint32_t)m->prob_n[e], &r, &pick)) return -2;
      alt = m->prob_alt[e][pick];
    }
    uint8_t dec[16]; uint64_t nd = 0;
    for (uint64_t j = 0; j < k; j++) {
      uint32_t v = (alt >> (3 * j)) & 7u;
      if (v < 4) dec[nd++] = (uint8_t)"ACGT"[v];
      else if (v == 4) continue;     /* skip_n: an N in the alternate k-mer is a deletion */
      else return -3;
    }
    memmove(out + i + nd, out + i + k, cur - (i + k));
    memcpy(out + i, dec, nd);
    cur = cur - k + nd;
  }
  return (int64_t)cur;
}
int64_t orc_custom_simulate_errors(const orc_model* m, const uint8_t* seq, uint64_t len, uint64_t seed, uint8_t* out) {
  return simulate_errors_walk(m, seq, len, seed, out, 0);
}
int64_t orc_custom_simulate_errors_philox(const orc_model* m, const uint8_t* seq, uint64_t len, uint64_t seed, uint8_t* out) {
  return simulate_errors_walk(m, seq, len, seed, out, 1);
}

/* ---------------------------------------------- CustomShortErrorProfile (custom_short.rs:155-543) */
struct orc_custom {
  orc_model model;
  orc_pdf* quality; /* one CustomPDF entry per read position */
  orc_pdf read_length, insert_size;
};

orc_custom* orc_custom_new(const uint8_t* bytes, uint64_t n) {
  orc_custom* c = (orc_custom*)calloc(1, sizeof *c);
  if (orc_model_parse(bytes, n, &c->model)) { free(c); return NULL; }
  c->quality = (orc_pdf*)calloc(c->model.n_quality ? c->model.n_quality : 1, sizeof(orc_pdf));
  for (uint64_t i = 0; i < c->model.n_quality; i++)
    if (orc_pdf_new(&c->model.quality[i], &c->quality[i])) return NULL;
  if (orc_pdf_new(&c->model.read_length_bins, &c->read_length)) return NULL;
  if (c->model.has_insert_bins && orc_pdf_new(&c->model.insert_bins, &c->insert_size)) return NULL;
  return c;
}
const orc_model* orc_custom_model(const orc_custom* c) { return &c->model; }
/* get_read_length (custom_short.rs:237-244): PDF sample `as u16` */
int orc_custom_get_read_length(const orc_custom* c, uint64_t seed, uint16_t* out) {
  uint32_t v;
  if (orc_pdf_sample(&c->read_length, seed, &v)) return -1;
  *out = (uint16_t)v;
  return 0;
}
/* get_insert_size (custom_short.rs:263-270) */
int orc_custom_get_insert_size(const orc_custom* c, uint64_t seed, uint16_t* out) {
  if (!c->model.has_insert_bins) { *out = 0; return 0; }
  uint32_t v;
  if (orc_pdf_sample(&c->insert_size, seed, &v)) return -1;
  *out = (uint16_t)v;
  return 0;
}
/* minimum_genome_size (custom_short.rs:535-538): (2.0 * mean_len + mean_insert) as u16 */
uint16_t orc_custom_minimum_genome_size(const orc_custom* c) {
  double v = 2.0 * c->model.read_length_mean + c->model.insert_size_mean;
  if (!(v == v) || v <= 0.0) return 0;
  if (v >= 65535.0) return 65535;
  return (uint16_t)v;
}
/* simulate_phred_scores (custom_short.rs:332-353): position i uses PDF min(i, n-1),
 * EVERY position re-seeds with the same seed (sample_with_index). */
int orc_custom_simulate_phred_scores(const orc_custom* c, uint64_t len, uint64_t seed, uint8_t* out) {
  const uint64_t n = c->model.n_quality;
  if (n == 0 && len > 0) return -1;
  for (uint64_t i = 0; i < len; i++) {
    uint32_t v;
    if (orc_pdf_sample(&c->quality[i >= n ? n - 1 : i], seed, &v)) return -1;
    out[i] = (uint8_t)v;
  }
  return 0;
}

/* gaussian (custom_long.rs:36-44): the normal kernel density estimate at x over the points xs with the given
 * bandwidth, terms added in slice order as the iterator sum does.  (The rest of CustomLongErrorProfile is not
 * restated: nothing in the reference constructs that profile — no constructor, not in cli.rs:62-70 — see DESIGN.md
 * section 6; this function is here for the one known-answer test the reference holds for it, custom_long.rs:264-272.) */
double orc_gaussian_kde(double x, const double* xs, uint64_t n, double bandwidth) {
  double sum = 0.0;
  for (uint64_t i = 0; i < n; i++) {
    const double f = (x - xs[i]) / bandwidth;
    sum += exp(-0.5 * (f * f));
  }
  return sum / (sqrt(2.0 * 3.14159265358979323846264338327950288) * (double)n * bandwidth);
}
