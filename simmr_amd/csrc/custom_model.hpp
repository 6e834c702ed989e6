// custom_model.hpp — host side of the custom (empirical) error profile:
// reads the bincode `ErrorModelParams` a simmrd run writes
// (shared/src/encoding.rs:82-117,244-281) and builds, exactly as the reference
// does at start-up (CustomPDF::new, custom_short.rs:60-86), the alias tables
// (rand_distr 0.4.3 WeightedAliasIndex<f64>::new) and per-bin uniform samplers
// (rand 0.8.5 Uniform<u32>::new_inclusive) in the flat form the kernels read.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "device_types.hpp"

namespace simmr {

struct BinsHost {
  uint64_t num_bins = 0, bin_width = 0;
  std::vector<double> density;
  std::vector<std::pair<uint32_t, uint32_t>> ranges;
};

struct ModelHost {
  uint64_t bin_size = 0;
  std::vector<BinsHost> quality;
  uint8_t bit_encoding = 0;
  uint64_t kmer_size = 0;
  std::vector<std::pair<uint32_t, std::vector<std::pair<uint32_t, float>>>> probabilities;
  double insert_size_mean = 0, insert_size_std = 0;
  bool has_insert_bins = false;
  BinsHost insert_bins;
  double read_length_mean = 0, read_length_std = 0;
  BinsHost read_length_bins;
  bool is_long = false;
};

class BincodeReader {
 public:
  BincodeReader(const uint8_t* p, uint64_t n) : p_(p), n_(n) {}
  bool ok() const { return !bad_; }
  bool at_end() const { return pos_ == n_; }
  uint64_t u64() { uint64_t v = 0; take(&v, 8); return v; }
  uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
  uint8_t u8() { uint8_t v = 0; take(&v, 1); return v; }
  double f64() { double v = 0; take(&v, 8); return v; }
  float f32() { float v = 0; take(&v, 4); return v; }
  uint64_t len(uint64_t elem_bytes) {  // Vec length, checked against what is left
    uint64_t v = u64();
    if (bad_ || v > (n_ - pos_) / (elem_bytes ? elem_bytes : 1)) { bad_ = true; return 0; }
    return v;
  }
 private:
  void take(void* dst, uint64_t k) {
    if (bad_ || pos_ + k > n_) { bad_ = true; return; }
    memcpy(dst, p_ + pos_, k);
    pos_ += k;
  }
  const uint8_t* p_;
  uint64_t n_, pos_ = 0;
  bool bad_ = false;
};

inline bool read_bins(BincodeReader& r, BinsHost* b) {
  b->num_bins = r.u64();
  b->bin_width = r.u64();
  uint64_t nd = r.len(8);
  b->density.resize(nd);
  for (uint64_t i = 0; i < nd; i++) b->density[i] = r.f64();
  uint64_t nr = r.len(8);
  b->ranges.resize(nr);
  for (uint64_t i = 0; i < nr; i++) { b->ranges[i].first = r.u32(); b->ranges[i].second = r.u32(); }
  return r.ok();
}

// bincode::deserialize::<ErrorModelParams>: fields in declaration order
inline bool parse_model(const uint8_t* bytes, uint64_t n, ModelHost* m, std::string* err) {
  BincodeReader r(bytes, n);
  m->bin_size = r.u64();
  uint64_t nq = r.len(32);
  m->quality.resize(nq);
  for (uint64_t i = 0; i < nq && r.ok(); i++) read_bins(r, &m->quality[i]);
  m->bit_encoding = r.u8();
  m->kmer_size = r.u64();
  uint64_t np = r.len(12);
  m->probabilities.resize(np);
  for (uint64_t i = 0; i < np && r.ok(); i++) {
    m->probabilities[i].first = r.u32();
    uint64_t k = r.len(8);
    m->probabilities[i].second.resize(k);
    for (uint64_t j = 0; j < k; j++) { m->probabilities[i].second[j].first = r.u32(); m->probabilities[i].second[j].second = r.f32(); }
  }
  m->insert_size_mean = r.f64();
  m->insert_size_std = r.f64();
  m->has_insert_bins = r.u8() != 0;
  if (m->has_insert_bins) read_bins(r, &m->insert_bins);
  m->read_length_mean = r.f64();
  m->read_length_std = r.f64();
  read_bins(r, &m->read_length_bins);
  m->is_long = r.u8() != 0;
  if (!r.ok() || !r.at_end()) { *err = "Error parsing custom error profile: unexpected end of file or trailing bytes"; return false; }
  return true;
}

// Flat tables for the device
struct PdfTables {
  std::vector<PdfDev> pdfs;
  std::vector<double> odds;
  std::vector<uint32_t> alias, bin_low, bin_range, bin_zone;
};

inline double pairwise_sum(const double* v, size_t n) {  // rand_distr AliasableWeight::sum for floats
  if (n <= 32) { double s = 0.0; for (size_t i = 0; i < n; i++) s += v[i]; return s; }
  size_t mid = n / 2;
  return pairwise_sum(v, mid) + pairwise_sum(v + mid, n - mid);
}

// Appends one CustomPDF entry (alias table + bins); an empty Bins gives n == 0.
inline bool append_pdf(const BinsHost& b, PdfTables* t, std::string* err) {
  PdfDev d{};
  const uint32_t n = (uint32_t)b.density.size();
  d.n = n;
  d.off = (uint32_t)t->odds.size();
  d.off_bins = (uint32_t)t->bin_low.size();
  d.n_bins = (uint32_t)b.ranges.size();
  if (n > 0) {
    double wsum = pairwise_sum(b.density.data(), n);
    if (wsum > 1.7976931348623157e308) wsum = 1.7976931348623157e308;
    for (double w : b.density) if (!(w >= 0.0)) { *err = "custom model: negative or NaN density (WeightedError::InvalidWeight)"; return false; }
    if (wsum == 0.0) { *err = "custom model: all densities are zero (WeightedError::AllWeightsZero)"; return false; }
    std::vector<double> odds(n);
    std::vector<uint32_t> al(n, 0);
    for (uint32_t i = 0; i < n; i++) odds[i] = b.density[i] * (double)n;
    uint32_t smalls = 0xFFFFFFFFu, bigs = 0xFFFFFFFFu;  // the intrusive stacks of `Aliases`
    for (uint32_t i = 0; i < n; i++) {
      if (odds[i] < wsum) { al[i] = smalls; smalls = i; } else { al[i] = bigs; bigs = i; }
    }
    while (smalls != 0xFFFFFFFFu && bigs != 0xFFFFFFFFu) {
      const uint32_t s = smalls; smalls = al[s];
      const uint32_t g = bigs; bigs = al[g];
      al[s] = g;
      odds[g] = odds[g] - wsum + odds[s];
      if (odds[g] < wsum) { al[g] = smalls; smalls = g; } else { al[g] = bigs; bigs = g; }
    }
    while (smalls != 0xFFFFFFFFu) { const uint32_t s = smalls; smalls = al[s]; odds[s] = wsum; }
    while (bigs != 0xFFFFFFFFu) { const uint32_t g = bigs; bigs = al[g]; odds[g] = wsum; }
    // Uniform::new(0u32, n) and Uniform::new(0.0, weight_sum)
    d.idx_zone = 0xFFFFFFFFu - (uint32_t)((0x100000000ULL - n) % n);
    uint64_t mb = (0xFFFFFFFFFFFFFFFFULL >> 12) | 0x3FF0000000000000ULL;
    double max_rand; memcpy(&max_rand, &mb, 8); max_rand -= 1.0;
    double scale = wsum - 0.0;
    while (scale * max_rand + 0.0 >= wsum) { uint64_t sb; memcpy(&sb, &scale, 8); sb -= 1; memcpy(&scale, &sb, 8); }
    d.w_scale = scale;
    t->odds.insert(t->odds.end(), odds.begin(), odds.end());
    t->alias.insert(t->alias.end(), al.begin(), al.end());
  }
  for (const auto& r : b.ranges) {  // Uniform::new_inclusive(start, end)
    if (r.first > r.second) { *err = "custom model: bin range with start > end (Uniform::new_inclusive panics)"; return false; }
    const uint32_t range = r.second - r.first + 1u;
    t->bin_low.push_back(r.first);
    t->bin_range.push_back(range);
    t->bin_zone.push_back(range ? 0xFFFFFFFFu - (uint32_t)((0x100000000ULL - range) % range) : 0xFFFFFFFFu);
  }
  t->pdfs.push_back(d);
  return true;
}

// Tables for simulate_errors (custom_short.rs:455-516).  The reference rebuilds a HashMap of the model's
// k-mer probabilities per call (:462-467) and a WeightedAliasIndex<f32> per visited k-mer (:497-499); both
// depend on the model only, so they are built once here.  A k-mer of ACGT only is looked up in a direct
// table indexed by its 2-bit code (4^k entries); one with an N (three_bit_encode_kmer accepts it) in a small
// open-addressing table keyed by the model's 3-bit code.  Per entry: the alias columns with the alternates
// already resolved and packed to 2 bits per base (bit 31 = the alternate has an N or an invalid field).
struct KmerTables {
  std::vector<Rec16> direct;  // {first record, n alternates (0 = not in the model, ~0 = unusable weights), zone, -}
  std::vector<Rec16> slots;   // {3-bit key (~0 = empty), first record, n alternates (~0 = unusable weights), zone}
  std::vector<Rec16> recs;    // {odds f32, alternate c, alternate alias(c), Uniform(0, sum) scale f32}
  uint32_t mask = 0;
  // The same columns at a fixed stride, for models whose k-mer codes fit a byte table in LDS (k <= 7) and whose lists
  // are short: a visited k-mer then costs ONE dependent global load (its column) instead of two (entry, then column).
  std::vector<uint8_t> cnt8;  // by 2-bit k-mer code: n alternates (0 = not in the model, 255 = unusable weights)
  std::vector<Rec16> cols;    // [code * stride + c]
  uint32_t stride = 0;        // 0 = no fixed-stride tables
  // SIMMR_RNG_PHILOX (the counter mode's two-level draw, ctr_splice_tables below): direct[code].w = the level-1 threshold
  // T24 of the k-mer; level-2 columns parallel to recs / cols: {threshold in 2^24ths, alternate c, alternate alias(c), -}
  std::vector<Rec16> recs_ctr, cols_ctr;
  std::vector<uint32_t> tab32;  // by 2-bit k-mer code: T24 << 8 | cnt8 (the fixed-stride form's LDS table)
};

// The counter mode's draw of a visited k-mer's alternate (include/simmr_hip.h, enum simmr_rng_mode; restated in
// the test tree's CPU specification).  The law is the reference's — P(alternate j) = w_j / sum(w) — split in two levels so
// that the common outcome, "the k-mer stays what it is", needs no table access, from ONE 32-bit word X per position:
//   level 1: X >> 8 < T24 answers "self", T24 = 2^24 - 2^e with 2^e the smallest power of two (1 <= e <= 24) of 2^24ths
//            that holds 1 - P(self);
//   level 2 (X >> 8 >= T24): Z = (X - (T24 << 8)) << (24 - e), a full word again, picks column (Z n) >> 32 of an alias table
//            (Vose, n columns, thresholds in 2^24ths against ((Z n) & 0xffffffff) >> 8) over the residual law
//            r_j = (p_j - [j is self] (T24 / 2^24) p_j / p_s) / (1 - T24 / 2^24).
// All in f64, sums in list order; `has_self` false for k-mers with an N (their alternates with an N are errors, not draws).
inline uint32_t ctr_splice_tables(const uint32_t* alt, const float* w, uint32_t n, uint32_t self_code, bool has_self,
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
  std::vector<double> odds(n);
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
  return T24;
}
constexpr uint32_t KMER_FAST_MAX_K = 7;          // 4^7 bytes = 16 KB of LDS
constexpr uint32_t KMER_FAST_MAX_STRIDE = 32;    // 4^7 * 32 * 16 B = 8 MB at most

inline float pairwise_sum_f32(const float* v, size_t n) {
  if (n <= 32) { float s = 0.0f; for (size_t i = 0; i < n; i++) s += v[i]; return s; }
  size_t mid = n / 2;
  return pairwise_sum_f32(v, mid) + pairwise_sum_f32(v + mid, n - mid);
}
inline uint32_t f32_to_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
inline float bits_to_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
inline uint32_t kmer_hash(uint32_t key) { return (key * 0x9E3779B1u) >> 7; }

inline bool build_kmer_tables(const ModelHost& m, KmerTables* t, std::string* err) {
  if (m.kmer_size == 0 || m.kmer_size > 10) {
    *err = "custom model: kmer_size must be 1..10 (a 3-bit code of more bases does not fit the model's u32 keys)";
    return false;
  }
  const uint32_t K = (uint32_t)m.kmer_size;
  const uint32_t kmask = K >= 10 ? 0x3fffffffu : ((1u << (3 * K)) - 1u);
  // 3-bit code -> 2 bits per base; bit 31 if a field is not ACGT
  auto pack2 = [&](uint32_t code3) {
    uint32_t v = 0;
    for (uint32_t j = 0; j < K; j++) {
      const uint32_t f = (code3 >> (3 * j)) & 7u;
      if (f >= 4) v |= 0x80000000u;
      v |= (f & 3u) << (2 * j);
    }
    return v;
  };
  // HashMap::from_iter over the list: a repeated key keeps its LAST entry
  std::vector<size_t> last;
  {
    std::vector<std::pair<uint32_t, size_t>> order;
    for (size_t i = 0; i < m.probabilities.size(); i++) order.emplace_back(m.probabilities[i].first, i);
    std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (size_t i = 0; i < order.size(); i++)
      if (i + 1 == order.size() || order[i + 1].first != order[i].first) last.push_back(order[i].second);
  }
  size_t n_hashed = 0;
  for (size_t e : last) {
    const uint32_t key = m.probabilities[e].first;
    if (key <= kmask && (pack2(key) & 0x80000000u)) n_hashed++;
  }
  uint32_t n_slots = 16;
  while ((uint64_t)n_slots < 2 * (uint64_t)n_hashed + 1) n_slots <<= 1;
  t->mask = n_slots - 1;
  t->slots.assign(n_slots, Rec16{0xFFFFFFFFu, 0u, 0u, 0u});
  t->direct.assign((size_t)1 << (2 * K), Rec16{0u, 0u, 0u, 0u});
  t->recs.clear();
  t->recs_ctr.clear();
  for (size_t e : last) {
    const uint32_t key = m.probabilities[e].first;
    if (key > kmask) continue;  // not the code of any k-mer of this size
    bool window_can_hold = true;  // a window only ever holds A, C, G, T, N when it is encodable
    for (uint32_t j = 0; j < K; j++) window_can_hold = window_can_hold && ((key >> (3 * j)) & 7u) <= 4u;
    if (!window_can_hold) continue;
    const auto& alts = m.probabilities[e].second;
    const uint32_t n = (uint32_t)alts.size();
    const uint32_t first = (uint32_t)t->recs.size();
    uint32_t n_field = 0xFFFFFFFFu, zone = 0;
    // WeightedAliasIndex::<f32>::new: n == 0, a weight outside [0, f32::MAX / n] or a zero sum is an Err,
    // which the reference unwraps: n_field stays ~0 = "panics when visited"
    bool ok = n > 0;
    std::vector<float> w(n);
    for (uint32_t i = 0; i < n; i++) w[i] = alts[i].second;
    const float maxw = n ? 3.40282347e38f / (float)n : 0.0f;
    for (uint32_t i = 0; i < n && ok; i++) ok = (0.0f <= w[i]) && (w[i] <= maxw);
    float wsum = 0.0f;
    if (ok) {
      wsum = pairwise_sum_f32(w.data(), n);
      if (wsum > 3.40282347e38f) wsum = 3.40282347e38f;
      ok = wsum != 0.0f;
    }
    if (ok) {
      std::vector<float> odds(n);
      std::vector<uint32_t> al(n, 0);
      const float nf = (float)n;
      for (uint32_t i = 0; i < n; i++) odds[i] = w[i] * nf;
      uint32_t smalls = 0xFFFFFFFFu, bigs = 0xFFFFFFFFu;
      for (uint32_t i = 0; i < n; i++) {
        if (odds[i] < wsum) { al[i] = smalls; smalls = i; } else { al[i] = bigs; bigs = i; }
      }
      while (smalls != 0xFFFFFFFFu && bigs != 0xFFFFFFFFu) {
        const uint32_t s = smalls; smalls = al[s];
        const uint32_t g = bigs; bigs = al[g];
        al[s] = g;
        odds[g] = odds[g] - wsum + odds[s];
        if (odds[g] < wsum) { al[g] = smalls; smalls = g; } else { al[g] = bigs; bigs = g; }
      }
      while (smalls != 0xFFFFFFFFu) { const uint32_t s = smalls; smalls = al[s]; odds[s] = wsum; }
      while (bigs != 0xFFFFFFFFu) { const uint32_t g = bigs; bigs = al[g]; odds[g] = wsum; }
      // Uniform::new(0.0f32, weight_sum): the scale is lowered until the largest draw stays below the sum
      const float max_rand = bits_to_f32((0xFFFFFFFFu >> 9) | 0x3F800000u) - 1.0f;
      float scale = wsum;
      while (scale * max_rand + 0.0f >= wsum) scale = bits_to_f32(f32_to_bits(scale) - 1u);
      n_field = n;
      zone = 0xFFFFFFFFu - (uint32_t)((0x100000000ULL - n) % n);  // Uniform::new(0u32, n)
      // a column left on a stack keeps its link in al[] and odds == sum: its alias is never taken
      for (uint32_t c = 0; c < n; c++)
        t->recs.push_back(Rec16{f32_to_bits(odds[c]), pack2(alts[c].first & kmask),
                                pack2(alts[al[c] < n ? al[c] : c].first & kmask), f32_to_bits(scale)});
    }
    const uint32_t p2 = pack2(key);
    uint32_t t24 = 0;
    if (ok) {  // the counter mode's tables of this k-mer
      std::vector<uint32_t> codes(n), thr(n), al2(n);
      for (uint32_t c = 0; c < n; c++) codes[c] = alts[c].first & kmask;
      t24 = ctr_splice_tables(codes.data(), w.data(), n, key, !(p2 & 0x80000000u), thr.data(), al2.data());
      for (uint32_t c = 0; c < n; c++) t->recs_ctr.push_back(Rec16{thr[c], pack2(codes[c]), pack2(codes[al2[c]]), 0u});
    }
    if (!(p2 & 0x80000000u)) {
      t->direct[p2] = Rec16{first, n_field, zone, t24};
    } else {
      uint32_t h = kmer_hash(key) & t->mask;
      while (t->slots[h].x != 0xFFFFFFFFu) h = (h + 1) & t->mask;
      t->slots[h] = Rec16{key, first, n_field, zone};
    }
  }
  if (t->recs.empty()) { t->recs.push_back(Rec16{0u, 0u, 0u, 0u}); t->recs_ctr.push_back(Rec16{0u, 0u, 0u, 0u}); }
  // fixed-stride copy of the direct entries' columns
  t->cnt8.clear(); t->cols.clear(); t->cols_ctr.clear(); t->tab32.clear(); t->stride = 0;
  if (K <= KMER_FAST_MAX_K) {
    uint32_t longest = 1;
    for (const Rec16& d : t->direct) if (d.y != 0xFFFFFFFFu && d.y > longest) longest = d.y;
    if (longest <= KMER_FAST_MAX_STRIDE) {
      t->stride = longest;
      t->cnt8.assign(t->direct.size(), 0);
      t->cols.assign(t->direct.size() * (size_t)longest, Rec16{0u, 0u, 0u, 0u});
      t->cols_ctr.assign(t->direct.size() * (size_t)longest, Rec16{0u, 0u, 0u, 0u});
      t->tab32.assign(t->direct.size(), 0u);
      for (size_t code = 0; code < t->direct.size(); code++) {
        const Rec16& d = t->direct[code];
        if (d.y == 0xFFFFFFFFu) { t->cnt8[code] = 255; t->tab32[code] = 255u; continue; }
        t->cnt8[code] = (uint8_t)d.y;
        t->tab32[code] = (d.w << 8) | d.y;
        for (uint32_t c = 0; c < d.y; c++) t->cols[code * (size_t)longest + c] = t->recs[d.x + c];
        for (uint32_t c = 0; c < d.y; c++) t->cols_ctr[code * (size_t)longest + c] = t->recs_ctr[d.x + c];
      }
    }
  }
  return true;
}

}  // namespace simmr
