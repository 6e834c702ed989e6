// rng_device.hpp — device-side arithmetic of the generators simmr's hot path
// draws from, written for gfx950 (wave64).
//
// What it has to reproduce (the crates are third-party, pinned in the
// reference's Cargo.lock:656-693; call sites are in simmr/src/simulate.rs and
// simmr/src/error_profiles/*.rs):
//   StdRng::seed_from_u64  = PCG32 XSH-RR expansion of a u64 into a 256-bit key
//   StdRng                 = ChaCha12, 64-bit block counter, stream id 0,
//                            words consumed consecutively (next_u32 / next_u64)
//   gen_range              = widening multiply + conservative zone rejection
//   Standard f32/f64/bool/Option<u64>, Open01, ziggurat StandardNormal,
//   Normal, Gamma (Marsaglia-Tsang).
//
// Layout choices are GPU-first: ChaCha is counter based, so a wave computes
// many blocks of one stream at once (one block per lane) into an LDS window and
// the sequential "how many words did the previous draw eat" dependency is
// resolved afterwards with ballots — see kernels.hip.  LaneRng below is the
// one-lane-one-stream form used where a stream is only a few words long
// (per-pair planning).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.hpp"

namespace simmr {

#define SIMMR_DEV __device__ __forceinline__

SIMMR_DEV uint32_t rotl32(uint32_t x, int n) { return __builtin_rotateleft32(x, n); }

using Key = Key8;

// rand_core 0.6.3 seed_from_u64: 8 PCG32 steps -> 8 little-endian key words.
SIMMR_DEV Key pcg32_expand(uint64_t state) {
  const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
  Key key;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key.k[i] = __builtin_rotateright32(xs, rot);
  }
  return key;
}

#define SIMMR_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);

// One ChaCha12 block (rand_chacha 0.3.1 ChaCha12Core): out[16].
SIMMR_DEV void chacha12_block(const Key& key, uint64_t counter, uint32_t out[16]) {
  const uint32_t c0 = 0x61707865u, c1 = 0x3320646eu, c2 = 0x79622d32u, c3 = 0x6b206574u;
  uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3;
  uint32_t x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3];
  uint32_t x8 = key.k[4], x9 = key.k[5], x10 = key.k[6], x11 = key.k[7];
  const uint32_t n0 = (uint32_t)counter, n1 = (uint32_t)(counter >> 32);
  uint32_t x12 = n0, x13 = n1, x14 = 0, x15 = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    SIMMR_QR(x0, x4, x8, x12) SIMMR_QR(x1, x5, x9, x13)
    SIMMR_QR(x2, x6, x10, x14) SIMMR_QR(x3, x7, x11, x15)
    SIMMR_QR(x0, x5, x10, x15) SIMMR_QR(x1, x6, x11, x12)
    SIMMR_QR(x2, x7, x8, x13) SIMMR_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + c0; out[1] = x1 + c1; out[2] = x2 + c2; out[3] = x3 + c3;
  out[4] = x4 + key.k[0]; out[5] = x5 + key.k[1]; out[6] = x6 + key.k[2]; out[7] = x7 + key.k[3];
  out[8] = x8 + key.k[4]; out[9] = x9 + key.k[5]; out[10] = x10 + key.k[6]; out[11] = x11 + key.k[7];
  out[12] = x12 + n0; out[13] = x13 + n1; out[14] = x14; out[15] = x15;
}

// gen_range zone for a u64 range (rand 0.8.5 UniformInt::sample_single).
SIMMR_DEV uint64_t zone64(uint64_t range) { return (range << __builtin_clzll(range)) - 1; }

// Documented substitute for OS entropy (include/simmr_hip.h).
__host__ __device__ inline uint64_t entropy_substitute(uint64_t x, uint32_t which) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL * (uint64_t)which;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t per_read_seed(uint64_t seed, uint64_t read_index) {
  return entropy_substitute(seed ^ (read_index * 0x9E3779B97F4A7C15ULL), 3);
}
__host__ __device__ inline uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
  uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Tables shared by every kernel (device global memory, L1/L2 resident).
struct Tables {
  double zig_x[257];   // rand_distr ZIG_NORM_X
  double zig_f[257];   // rand_distr ZIG_NORM_F
  float acc[256];      // util::convert_phred_to_accuracy(q), host libm powf
  float pl_thresh[64]; // perfect-long: thresholds on (1 - acc) between Phred values
  uint32_t pl_first;   // Phred value of the first interval
  uint32_t pl_count;   // number of thresholds
};

#define SIMMR_ZIG_R 3.654152885361008796

// Philox4x32-10 (Random123 constants; rocRAND's rocrand_philox4x32_10 with subsequence 'simm' | 'r\0\0\3' << 32): the
// block with key (k0, k1) and counter (c0, c1, 'simm', 'r\0\0\3').  Two v_mad_u64_u32 and two v_bitop3_b32 per round.
SIMMR_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
SIMMR_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c2 = 0x73696D6Du, c3 = 0x72000003u;
#if defined(SIMMR_ABLATE_PHILOX)
  out[0] = c0 * M0 + k0; out[1] = (c0 ^ c1) * M1 + k1; out[2] = out[0] ^ c2 ^ (k1 + W0); out[3] = out[1] ^ c3 ^ (k0 + W1);
  return;
#endif
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;  // one v_mad_u64_u32 each
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    c0 = xor3(h1, c1, k0); c1 = l1; c2 = xor3(h0, c3, k1); c3 = l0;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------
// LaneRng: BlockRng<ChaCha12Core> semantics for ONE lane, one block buffered in
// a lane-private LDS row (17-word pitch -> conflict free).  Words are consumed
// strictly consecutively, exactly like rand_core's BlockRng (the reference's
// 4-block refill is only buffering).
template <bool CTR>
struct LaneRngT {
  Key key;
  uint64_t next_block;  // counter of the block that will be generated next
  uint32_t idx;         // next unread word in buf, 16 = empty
  uint32_t* buf;        // 16 words of LDS owned by this lane
  uint32_t words_used;
  // CTR — SIMMR_RNG_PHILOX_FULL (include/simmr_hip.h): the same consumer over the word stream W(s) — word w = word w & 3
  // of the Philox block with key = s and counter (w >> 2, 3, ..) — buffered eight words (two blocks) at a time: a
  // pair's plan takes eight.  cap = words per refill (16 for ChaCha12); k0 / k1 = the key.  (A compile-time switch: with
  // both word sources in one kernel the plan kernels lost a third of their waves to the registers of the one not in use.)
  static constexpr uint32_t cap = CTR ? 8u : 16u;
  uint32_t k0, k1;

  SIMMR_DEV void seed_from_u64(uint64_t s, uint32_t* lds_row) {
    if (CTR) { k0 = (uint32_t)s; k1 = (uint32_t)(s >> 32); } else { key = pcg32_expand(s); }
    next_block = 0;
    idx = cap;
    buf = lds_row;
    words_used = 0;
  }
  // same seed again (the reference re-creates StdRng::seed_from_u64(seed) for every
  // profile call): keep the key, and block 0 if it is still the buffered one
  SIMMR_DEV void restart() {
    if (next_block == 1) { idx = 0; } else { next_block = 0; idx = cap; }
    words_used = 0;
  }
  SIMMR_DEV void refill() {
    if (CTR) {  // (next_block counts refills: Philox blocks 2 next_block and 2 next_block + 1)
      uint32_t o[8];
      philox4x32_10(2u * (uint32_t)next_block, 3u, k0, k1, o);
      philox4x32_10(2u * (uint32_t)next_block + 1u, 3u, k0, k1, o + 4);
#pragma unroll
      for (int i = 0; i < 8; i++) buf[i] = o[i];
    } else {
      uint32_t o[16];
      chacha12_block(key, next_block, o);
#pragma unroll
      for (int i = 0; i < 16; i++) buf[i] = o[i];
    }
    next_block++;
    idx = 0;
  }
  SIMMR_DEV uint32_t next_u32() {
    if (idx >= cap) refill();
    words_used++;
    return buf[idx++];
  }
  SIMMR_DEV uint64_t next_u64() {
    uint32_t lo = next_u32();
    uint32_t hi = next_u32();
    return ((uint64_t)hi << 32) | lo;
  }
  // gen_range(lo..hi) on usize/u64; caller guarantees lo < hi.
  SIMMR_DEV uint64_t gen_range_u64(uint64_t lo, uint64_t hi) {
    uint64_t range = hi - lo;
    uint64_t zone = zone64(range);
    for (;;) {
      uint64_t v = next_u64();
      uint64_t l = v * range;
      if (l <= zone) return lo + __umul64hi(v, range);
    }
  }
  SIMMR_DEV bool gen_bool() { return (int32_t)next_u32() < 0; }
  SIMMR_DEV double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  SIMMR_DEV double open01_f64() {
    return __longlong_as_double((long long)((next_u64() >> 12) | 0x3FF0000000000000ULL)) -
           (1.0 - 2.220446049250313e-16 / 2.0);
  }
  SIMMR_DEV float open01_f32() {
    return __uint_as_float((next_u32() >> 9) | 0x3F800000u) - (1.0f - 1.1920929e-7f / 2.0f);
  }
  // rand_distr StandardNormal (ziggurat, f64)
  SIMMR_DEV double standard_normal(const Tables* __restrict__ T) {
    for (;;) {
      uint64_t bits = next_u64();
      uint32_t i = (uint32_t)bits & 0xffu;
      double u = __longlong_as_double((long long)((bits >> 12) | 0x4000000000000000ULL)) - 3.0;
      double x = __dmul_rn(u, T->zig_x[i]);
      if (fabs(x) < T->zig_x[i + 1]) return x;
      if (i == 0) {
        double xx = 1.0, yy = 0.0;
        while (__dmul_rn(-2.0, yy) < __dmul_rn(xx, xx)) {
          double x_ = open01_f64();
          double y_ = open01_f64();
          xx = log(x_) / SIMMR_ZIG_R;
          yy = log(y_);
        }
        return u < 0.0 ? xx - SIMMR_ZIG_R : SIMMR_ZIG_R - xx;
      }
      double f1 = T->zig_f[i + 1], f0 = T->zig_f[i];
      double t = __dadd_rn(f1, __dmul_rn(__dsub_rn(f0, f1), gen_f64()));
      if (t < exp(__dmul_rn(__dmul_rn(-x, x), 0.5))) return x;
    }
  }
  // rand_distr Gamma<f32> large-shape branch (shape > 1), Marsaglia-Tsang.
  SIMMR_DEV float gamma_f32(const Tables* __restrict__ T, float shape, float scale) {
    const float d = __fsub_rn(shape, (float)(1.0 / 3.0));
    const float c = __fdiv_rn(1.0f, __fsqrt_rn(__fmul_rn(9.0f, d)));
    for (;;) {
      float x = (float)standard_normal(T);
      float v_cbrt = __fadd_rn(1.0f, __fmul_rn(c, x));
      if (v_cbrt <= 0.0f) continue;
      float v = __fmul_rn(__fmul_rn(v_cbrt, v_cbrt), v_cbrt);
      float u = open01_f32();
      float x_sqr = __fmul_rn(x, x);
      if (u < __fsub_rn(1.0f, __fmul_rn(__fmul_rn((float)0.0331, x_sqr), x_sqr)))
        return __fmul_rn(__fmul_rn(d, v), scale);
      float rhs = __fadd_rn(__fmul_rn(0.5f, x_sqr),
                            __fmul_rn(d, __fadd_rn(__fsub_rn(1.0f, v), logf(v))));
      if (logf(u) < rhs) return __fmul_rn(__fmul_rn(d, v), scale);
    }
  }
};
using LaneRng = LaneRngT<false>;  // the reference's streams

// saturating float -> integer `as` casts of Rust (NaN -> 0)
SIMMR_DEV uint32_t sat_u8_f32(float f) {
  if (!(f == f) || f <= 0.0f) return 0;
  if (f >= 255.0f) return 255;
  return (uint32_t)f;
}
SIMMR_DEV uint32_t sat_u16_f32(float f) {
  if (!(f == f) || f <= 0.0f) return 0;
  if (f >= 65535.0f) return 65535;
  return (uint32_t)f;
}
SIMMR_DEV uint32_t sat_u16_f64(double f) {
  if (!(f == f) || f <= 0.0) return 0;
  if (f >= 65535.0) return 65535;
  return (uint32_t)f;
}

}  // namespace simmr
