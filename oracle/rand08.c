/*
 * rand08.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * CPU restatement of the third-party arithmetic simmr's hot path depends on.
 * The crates are NOT under /root/reference (un-vendored, Cargo.lock:656-693):
 *   rand_core  0.6.3  — SeedableRng::seed_from_u64 (PCG32 expansion), BlockRng
 *   rand_chacha 0.3.1 — ChaCha12Core, 4-block (64-word) buffer
 *   rand       0.8.5  — StdRng = ChaCha12Rng; Standard, UniformInt::sample_single,
 *                       Open01, SliceRandom::choose
 *   rand_distr 0.4.3  — StandardNormal (ziggurat), Normal, Gamma (Marsaglia-Tsang)
 * Each function restates the published algorithm of that version; the call
 * sites in the reference are cited.  Pinned by tests/test_oracle_kat.py.
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define QR(a, b, c, d)   \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);

/* ChaCha block function, djb variant as rand_chacha uses it: 64-bit block
 * counter in words 12-13, 64-bit stream id (0 for StdRng) in words 14-15. */
void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint32_t rounds, uint32_t out[16]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
  for (int i = 0; i < 8; i++) s[4 + i] = key[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
  s[14] = 0; s[15] = 0;
  memcpy(x, s, sizeof x);
  for (uint32_t r = 0; r < rounds; r += 2) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
    QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
    QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

/* rand_core 0.6.3 SeedableRng::seed_from_u64: PCG32 (XSH-RR) stream fills the
 * 32-byte seed four bytes at a time, little-endian.
 * Reference call sites: simulate.rs:172-175,227-230,348-351,478-481 and every
 * profile method (e.g. minimal_short.rs:34-37). */
void orc_pcg32_expand(uint64_t state, uint32_t key_out[8]) {
  const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key_out[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
}

static _Thread_local int g_stream_kind = 0;
void orc_set_stream_kind(int kind) { g_stream_kind = kind; }
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);  /* philox.c */

void orc_rng_from_seed(orc_rng* r, const uint8_t seed[32]) {
  r->ctr = 0;
  for (int i = 0; i < 8; i++)
    r->key[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 |
                (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
  r->counter = 0;
  r->index = 64;
  r->words_used = 0;
}

void orc_rng_seed_from_u64(orc_rng* r, uint64_t state) {
  r->ctr = g_stream_kind ? 1u : 0u;
  if (r->ctr) { memset(r->key, 0, sizeof r->key); r->key[0] = (uint32_t)state; r->key[1] = (uint32_t)(state >> 32); }
  else
  orc_pcg32_expand(state, r->key);
  r->counter = 0;
  r->index = 64;
  r->words_used = 0;
}

/* BlockRng::generate_and_set: refill 4 consecutive blocks. */
static void refill(orc_rng* r, uint32_t index) {
  if (r->ctr) {  /* the next 64 words of W(seed): sixteen Philox blocks (r->counter counts 16-word units as for ChaCha) */
    for (uint32_t b = 0; b < 16; b++) {
      const uint32_t c[4] = {(uint32_t)(4 * r->counter + b), 3u, 0x73696D6Du, 0x72000003u};
      orc_philox4x32_10(c, r->key, r->results + 4 * b);
    }
  } else
  for (int b = 0; b < 4; b++) orc_chacha_block(r->key, r->counter + b, 12, r->results + 16 * b);
  r->counter += 4;
  r->index = index;
}

/* BlockRng::next_u32 */
uint32_t orc_next_u32(orc_rng* r) {
  if (r->index >= 64) refill(r, 0);
  r->words_used += 1;
  return r->results[r->index++];
}

/* BlockRng::next_u64 — two consecutive words lo|hi<<32, also across a refill. */
uint64_t orc_next_u64(orc_rng* r) {
  r->words_used += 2;
  uint32_t index = r->index;
  if (index < 63) {
    r->index += 2;
    return (uint64_t)r->results[index + 1] << 32 | r->results[index];
  } else if (index >= 64) {
    refill(r, 2);
    return (uint64_t)r->results[1] << 32 | r->results[0];
  } else {
    uint64_t x = r->results[63];
    refill(r, 1);
    uint64_t y = r->results[0];
    return (y << 32) | x;
  }
}

/* rand 0.8.5 UniformInt<u64/usize>::sample_single (gen_range(lo..hi)):
 * widening multiply with the conservative zone (range << lz) - 1.
 * Call sites: simulate.rs:182,233,245,375,484,489. */
int orc_gen_range_u64(orc_rng* r, uint64_t lo, uint64_t hi, uint64_t* out) {
  if (!(lo < hi)) return -1; /* "cannot sample empty range" panic */
  uint64_t range = hi - lo;  /* (hi-1) - lo + 1 */
  uint64_t zone = (range << __builtin_clzll(range)) - 1;
  for (;;) {
    uint64_t v = orc_next_u64(r);
    unsigned __int128 m = (unsigned __int128)v * range;
    uint64_t h = (uint64_t)(m >> 64), l = (uint64_t)m;
    if (l <= zone) { *out = lo + h; return 0; }
  }
}

/* Same for u32 (SliceRandom::choose's gen_index uses the u32 path for
 * len <= u32::MAX; minimal_short.rs:122-125). */
int orc_gen_range_u32(orc_rng* r, uint32_t lo, uint32_t hi, uint32_t* out) {
  if (!(lo < hi)) return -1;
  uint32_t range = hi - lo;
  uint32_t zone = (range << __builtin_clz(range)) - 1;
  for (;;) {
    uint32_t v = orc_next_u32(r);
    uint64_t m = (uint64_t)v * range;
    uint32_t h = (uint32_t)(m >> 32), l = (uint32_t)m;
    if (l <= zone) { *out = lo + h; return 0; }
  }
}

/* Standard: f32 = 24 high bits * 2^-24; f64 = 53 high bits * 2^-53 */
float orc_gen_f32(orc_rng* r) { return (float)(orc_next_u32(r) >> 8) * (1.0f / 16777216.0f); }
double orc_gen_f64(orc_rng* r) {
  return (double)(orc_next_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}
/* Standard bool: sign bit of next_u32 */
int orc_gen_bool(orc_rng* r) { return (int32_t)orc_next_u32(r) < 0; }
/* Standard Option<T>: if gen::<bool>() { Some(gen()) } else { None }
 * — this is what `rng.gen()` resolves to at simulate.rs:266,270. */
int orc_gen_option_u64(orc_rng* r, uint64_t* out) {
  if (orc_gen_bool(r)) { *out = orc_next_u64(r); return 1; }
  return 0;
}

static inline double f64_from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline float f32_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

/* Open01: mantissa bits with exponent 0 -> [1,2), minus (1 - eps/2) -> (0,1) */
double orc_open01_f64(orc_rng* r) {
  uint64_t v = orc_next_u64(r);
  return f64_from_bits((v >> 12) | 0x3FF0000000000000ULL) - (1.0 - 2.220446049250313e-16 / 2.0);
}
float orc_open01_f32(orc_rng* r) {
  uint32_t v = orc_next_u32(r);
  return f32_from_bits((v >> 9) | 0x3F800000u) - (1.0f - 1.1920929e-7f / 2.0f);
}

/* rand_distr 0.4.3 ziggurat tables for N(0,1): regenerated with the crate's
 * own generator formulas (utils/ziggurat_tables.py: r = 3.654152885361009,
 * v = 0.00492867323399, 256 layers). */
#define ZIG_NORM_R 3.654152885361008796
static double ZX[257], ZF[257];
static int zig_ready = 0;
static double norm_f(double x) { return exp(-x * x / 2.0); }
static double norm_f_inv(double y) { return sqrt(-2.0 * log(y)); }
static void zig_init(void) {
  if (zig_ready) return;
  const double r = 3.6541528853610088, v = 0.00492867323399;
  ZX[0] = v / norm_f(r);
  ZX[1] = r;
  for (int i = 2; i < 256; i++) {
    double last = ZX[i - 1];
    ZX[i] = norm_f_inv(v / last + norm_f(last));
  }
  ZX[256] = 0.0;
  for (int i = 0; i < 257; i++) ZF[i] = norm_f(ZX[i]);
  __atomic_store_n(&zig_ready, 1, __ATOMIC_RELEASE);
}
const double* orc_zig_norm_x(void) { zig_init(); return ZX; }
const double* orc_zig_norm_f(void) { zig_init(); return ZF; }

/* StandardNormal::sample -> utils::ziggurat(symmetric = true) */
double orc_standard_normal(orc_rng* r) {
  zig_init();
  for (;;) {
    uint64_t bits = orc_next_u64(r);
    unsigned i = (unsigned)(bits & 0xff);
    double u = f64_from_bits((bits >> 12) | 0x4000000000000000ULL) - 3.0; /* [-1,1) */
    double x = u * ZX[i];
    if (fabs(x) < ZX[i + 1]) return x;
    if (i == 0) { /* zero_case: tail beyond R */
      double xx = 1.0, yy = 0.0;
      while (-2.0 * yy < xx * xx) {
        double x_ = orc_open01_f64(r);
        double y_ = orc_open01_f64(r);
        xx = log(x_) / ZIG_NORM_R;
        yy = log(y_);
      }
      return u < 0.0 ? xx - ZIG_NORM_R : ZIG_NORM_R - xx;
    }
    if (ZF[i + 1] + (ZF[i] - ZF[i + 1]) * orc_gen_f64(r) < exp(-x * x / 2.0)) return x;
  }
}

/* Normal<f64>::sample = mean + std_dev * z  (minimal_short.rs:40,65) */
double orc_normal_f64(orc_rng* r, double mean, double std) {
  double z = orc_standard_normal(r);
  return mean + std * z;
}
/* Normal<f32>::sample: StandardNormal for f32 = (f64 sample) as f32
 * (minimal_short.rs:90-96, minimal_long.rs:88-94, perfect_long.rs:68-72) */
float orc_normal_f32(orc_rng* r, float mean, float std) {
  float z = (float)orc_standard_normal(r);
  volatile float prod = std * z; /* no fused multiply-add: Rust does not contract */
  return mean + prod;
}

/* Gamma<f32>::new(shape, scale) with shape > 1 -> GammaLargeShape
 * (Marsaglia & Tsang 2000).  minimal_long.rs:58-73, perfect_long.rs:40-55. */
int orc_gamma_f32(orc_rng* r, float shape, float scale, float* out) {
  if (!(shape > 1.0f)) return -1;
  const float d = shape - (float)(1.0 / 3.0);
  const float c = 1.0f / sqrtf(9.0f * d);
  for (;;) {
    float x = (float)orc_standard_normal(r);
    volatile float cx = c * x;
    float v_cbrt = 1.0f + cx;
    if (v_cbrt <= 0.0f) continue;
    volatile float v2 = v_cbrt * v_cbrt;
    volatile float v = v2 * v_cbrt;
    float u = orc_open01_f32(r);
    volatile float x_sqr = x * x;
    volatile float t0 = (float)0.0331 * x_sqr;
    volatile float t1 = t0 * x_sqr;
    if (u < 1.0f - t1) { volatile float dv = d * v; *out = dv * scale; return 0; }
    volatile float a0 = 0.5f * x_sqr;
    volatile float a1 = 1.0f - v;
    volatile float a2 = a1 + logf(v);
    volatile float a3 = d * a2;
    if (logf(u) < a0 + a3) { volatile float dv = d * v; *out = dv * scale; return 0; }
  }
}
