// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// simmr hot path: reference staging, outer seed stream, per-unit planning,
// offset scan and the emit kernels.  See DESIGN.md for the data layout and the
// roofline of each kernel.  Reference behaviour being reproduced:
//   simmr/src/simulate.rs:165-302 (paired-end), :323-523 (long reads)
//   simmr/src/error_profiles/{perfect_short,minimal_short,perfect_long,minimal_long}.rs
//   simmr/src/util.rs:15-37 (reverse complement), :69-111 (Phred conversions)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "device_types.hpp"
#include "rng_device.hpp"

namespace simmr {

// ===========================================================================
// 1. Reference staging
// ===========================================================================

// ASCII -> (2-bit code, exception bit).  Exceptions keep a 1-bit payload in the
// code plane: 0 = 'N', 1 = '-'.  Lower-case and U follow needletail's
// normalize(false) as used at genome.rs:114.
SIMMR_DEV uint32_t ascii_to_code3(uint32_t ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    case '-': case '.': case '~': return 4 | 1;
    default: return 4 | 0;  // 'N' and everything needletail maps to N
  }
}

// One thread packs 32 bases: two code words + one exception word.
// `dst_base` (in bases, multiple of 64) is where ascii[0] lands.
extern "C" __global__ void __launch_bounds__(256)
k_pack_ascii(const uint8_t* __restrict__ ascii, uint64_t n, uint64_t dst_base,
             uint32_t* __restrict__ packed, uint32_t* __restrict__ mask,
             uint32_t* __restrict__ any_exc) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t b0 = t * 32;
  if (b0 >= n) return;
  uint32_t w0 = 0, w1 = 0, m = 0;
  uint32_t cnt = (uint32_t)((n - b0) < 32 ? (n - b0) : 32);
  for (uint32_t i = 0; i < cnt; i++) {
    uint32_t c = ascii_to_code3(ascii[b0 + i]);
    uint32_t code = c & 3u;
    if (i < 16) w0 |= code << (2 * i); else w1 |= code << (2 * (i - 16));
    m |= (c >> 2) << i;
  }
  uint64_t d = dst_base + b0;
  packed[d >> 4] = w0;
  packed[(d >> 4) + 1] = w1;
  mask[d >> 5] = m;
  if (m) atomicOr(any_exc, 1u);
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a fence plus a barrier, and the fence
// makes every wave wait for its outstanding GLOBAL stores (s_waitcnt vmcnt(0)) although no other wave will read
// them; a kernel that streams its output and synchronises per block then pays the store latency at every barrier.
SIMMR_DEV void lds_barrier() {
#if defined(SIMMR_ABLATE_FULL_BARRIER)
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// exclusive scan of one u32 per thread over a 256-thread workgroup (tid = threadIdx.x; a caller inside a long loop
// may pass a copy the compiler cannot see through, so that the lane arithmetic is not kept in registers across the loop)
template <bool LDS_ONLY = false>
SIMMR_DEV uint32_t wg_exclusive_scan_u32(uint32_t v, uint32_t* lds4, uint32_t* total, uint32_t tid) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += o;
  }
  if (lane == 63) lds4[wave] = inc;
  if (LDS_ONLY) lds_barrier(); else __syncthreads();
  const uint32_t t0 = lds4[0], t1 = lds4[1], t2 = lds4[2], t3 = lds4[3];
  const uint32_t pre = (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
  *total = t0 + t1 + t2 + t3;
  return pre + inc - v;
}
SIMMR_DEV uint32_t wg_exclusive_scan_u32(uint32_t v, uint32_t* lds4, uint32_t* total) {
  return wg_exclusive_scan_u32(v, lds4, total, threadIdx.x);
}
// two 32-bit values scanned together (one barrier): v = lo | hi << 32, no carry from lo into hi as long as the
// workgroup's sum of lo stays below 2^32; lds4: four u64
SIMMR_DEV uint64_t wg_exclusive_scan_2x32(uint64_t v, uint64_t* lds4, uint64_t* total, uint32_t tid) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += o;
  }
  if (lane == 63) lds4[wave] = inc;
  lds_barrier();
  const uint64_t t0 = lds4[0], t1 = lds4[1], t2 = lds4[2], t3 = lds4[3];
  const uint64_t pre = (wave > 0 ? t0 : 0ull) + (wave > 1 ? t1 : 0ull) + (wave > 2 ? t2 : 0ull);
  *total = t0 + t1 + t2 + t3;
  return pre + inc - v;
}

// ---------------------------------------------------------------------------
// FASTA bodies -> planes on the device (simmr_stage_fasta): needletail 0.4.1
// sequence::normalize(seq, iupac = false) as genome.rs:114 applies it, fused with
// the packing.  Raw record bodies sit in `raw` at offsets that are multiples of
// FASTA_TILE, padded with '\n'; a workgroup owns one tile of FASTA_TILE raw bytes.
//   class of a raw byte: 0..3 = A C G T (also a c g t, and u / U -> T), 4 = N
//   (N itself and every byte that is none of the others), 5 = '-' (also . ~),
//   0xff = dropped (space, tab, CR, LF).
// ---------------------------------------------------------------------------
#define FASTA_TILE 1024u

SIMMR_DEV uint32_t fasta_class(uint32_t c) {
  switch (c) {
    case 'A': case 'a': return 0u;
    case 'C': case 'c': return 1u;
    case 'G': case 'g': return 2u;
    case 'T': case 't': case 'U': case 'u': return 3u;
    case '-': case '.': case '~': return 5u;
    case ' ': case '\t': case '\r': case '\n': return 0xffu;
    default: return 4u;
  }
}

// bases kept per tile
extern "C" __global__ void __launch_bounds__(256)
k_fasta_count(const uint8_t* __restrict__ raw, uint64_t n_tiles, uint64_t* __restrict__ kept) {
  __shared__ uint32_t lds4[4];
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint32_t w = reinterpret_cast<const uint32_t*>(raw + t * FASTA_TILE)[threadIdx.x];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) c += fasta_class((w >> (8 * i)) & 0xffu) != 0xffu ? 1u : 0u;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) kept[t] = (uint64_t)lds4[0] + lds4[1] + lds4[2] + lds4[3];
  }
}

struct FastaRecord {
  uint64_t tile0;  // first tile of the record
  uint64_t dst;    // plane position (bases) of the record's first base; ~0 = record not staged
};

// prefix[t] = bases kept in tiles before t (all records); the tile's bases go to
// rec.dst + (prefix[t] - prefix[rec.tile0]) ...
extern "C" __global__ void __launch_bounds__(256)
k_fasta_pack(const uint8_t* __restrict__ raw, uint64_t n_tiles, const uint64_t* __restrict__ prefix,
             const FastaRecord* __restrict__ recs, uint32_t n_recs, uint32_t* __restrict__ packed,
             uint32_t* __restrict__ mask, uint32_t* __restrict__ any_exc) {
  __shared__ uint8_t codes[FASTA_TILE + 64];
  __shared__ uint32_t lds4[4];
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    uint32_t lo = 0, hi = n_recs;  // record of this tile: last one with tile0 <= t
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (recs[mid].tile0 <= t) lo = mid; else hi = mid; }
    const FastaRecord rec = recs[lo];
    if (rec.dst == ~0ull) continue;  // below the minimum size: not staged (main.rs:117-162)
    const uint64_t dst0 = rec.dst + (prefix[t] - prefix[rec.tile0]);
    const uint32_t w = reinterpret_cast<const uint32_t*>(raw + t * FASTA_TILE)[threadIdx.x];
    uint32_t cls[4], c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { cls[i] = fasta_class((w >> (8 * i)) & 0xffu); c += cls[i] != 0xffu ? 1u : 0u; }
    __syncthreads();  // the previous tile's words have been built
    uint32_t n_kept;
    uint32_t at = wg_exclusive_scan_u32(c, lds4, &n_kept);
#pragma unroll
    for (int i = 0; i < 4; i++) if (cls[i] != 0xffu) codes[at++] = (uint8_t)cls[i];
    __syncthreads();
    if (n_kept == 0) continue;
    // plane words touched by bases [dst0, dst0 + n_kept): the first and the last may be shared with a neighbour
    const uint64_t cw0 = dst0 >> 4, cw1 = (dst0 + n_kept - 1) >> 4;
    for (uint64_t wd = cw0 + threadIdx.x; wd <= cw1; wd += 256) {
      uint32_t v = 0;
      for (uint32_t i = 0; i < 16; i++) {
        const uint64_t b = wd * 16 + i;
        if (b >= dst0 && b < dst0 + n_kept) {
          const uint32_t cl = codes[b - dst0];
          v |= (cl < 4u ? cl : cl - 4u) << (2 * i);  // exception bases keep 0 ('N') / 1 ('-') in the code plane
        }
      }
      if (wd == cw0 || wd == cw1) atomicOr(&packed[wd], v); else packed[wd] = v;
    }
    const uint64_t mw0 = dst0 >> 5, mw1 = (dst0 + n_kept - 1) >> 5;
    for (uint64_t wd = mw0 + threadIdx.x; wd <= mw1; wd += 256) {
      uint32_t v = 0;
      for (uint32_t i = 0; i < 32; i++) {
        const uint64_t b = wd * 32 + i;
        if (b >= dst0 && b < dst0 + n_kept) v |= (codes[b - dst0] >= 4u ? 1u : 0u) << i;
      }
      if (v) { atomicOr(&mask[wd], v); atomicOr(any_exc, 1u); }
    }
  }
}

// genome.rs:121-137 (--contiguous): an 'N' after every record
extern "C" __global__ void __launch_bounds__(256)
k_fasta_separators(const uint64_t* __restrict__ pos, uint32_t n, uint32_t* __restrict__ mask, uint32_t* __restrict__ any_exc) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  atomicOr(&mask[pos[i] >> 5], 1u << (pos[i] & 31u));  // code plane stays 0 = 'N'
  atomicOr(any_exc, 1u);
}

// prefix values at the records' first tiles (and the grand total) for the host
extern "C" __global__ void __launch_bounds__(256)
k_fasta_gather(const uint64_t* __restrict__ prefix, const uint64_t* __restrict__ tile0, uint32_t n, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = prefix[tile0[i]];
}

// Synthetic reference: u64 word k of the packed plane = SplitMix64 output k.
extern "C" __global__ void __launch_bounds__(256)
k_synth(uint64_t* __restrict__ packed64, uint64_t n_words64, uint64_t seed) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_words64) packed64[t] = splitmix64_at(seed, t);
}

// Packed -> ASCII (tests / debugging only; not on the hot path).
extern "C" __global__ void __launch_bounds__(256)
k_unpack(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ mask, uint64_t base,
         uint64_t n, uint8_t* __restrict__ dst) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint64_t p = base + t;
  uint32_t code = (packed[p >> 4] >> ((p & 15) * 2)) & 3u;
  if (mask) code |= ((mask[p >> 5] >> (p & 31)) & 1u) << 2;
  dst[t] = (uint8_t)"ACGTN-N-"[code];
}

// ===========================================================================
// 2. Outer seed stream (simulate.rs:172-184 and :348-378)
//
// The reference walks ONE StdRng sequentially: gen_range(0..n) (which rejects
// and redraws — half of the time when n == 1) and then gen::<u64>() per unit.
// ChaCha is counter based, so every 8-slot block is generated independently;
// the sequential part is a two-state transducer (NEED_IDX / NEED_SEED) whose
// per-block transfer functions compose associatively:
//   classify  -> per-block function, reduced per workgroup
//   scan      -> one workgroup resolves workgroup start states + unit bases
//   emit      -> every block re-derives its words and writes its units.
// ===========================================================================

// f: state -> (state, units emitted).  e0/c0 for start NEED_IDX, e1/c1 for
// start NEED_SEED.  Packed: e0 | e1<<1 | c0<<2 (15 bits) | c1<<17 (15 bits).
SIMMR_DEV uint32_t fs_make(uint32_t e0, uint32_t c0, uint32_t e1, uint32_t c1) {
  return e0 | (e1 << 1) | (c0 << 2) | (c1 << 17);
}
SIMMR_DEV uint32_t fs_compose(uint32_t f, uint32_t g) {  // f first, then g
  uint32_t ge0 = g & 1u, ge1 = (g >> 1) & 1u, gc0 = (g >> 2) & 0x7fffu, gc1 = g >> 17;
  uint32_t fe0 = f & 1u, fe1 = (f >> 1) & 1u, fc0 = (f >> 2) & 0x7fffu, fc1 = f >> 17;
  uint32_t e0 = fe0 ? ge1 : ge0, c0 = fc0 + (fe0 ? gc1 : gc0);
  uint32_t e1 = fe1 ? ge1 : ge0, c1 = fc1 + (fe1 ? gc1 : gc0);
  return fs_make(e0, c0, e1, c1);
}
#define FS_IDENTITY 0x2u /* e0 = 0, e1 = 1, no units */

struct BlockWords {
  uint64_t v[8];
};

SIMMR_DEV void outer_block(const OuterParams& P, uint64_t b, BlockWords& w, uint32_t& accmask) {
  uint32_t o[16];
  chacha12_block(P.key, b, o);
  accmask = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    w.v[j] = ((uint64_t)o[2 * j + 1] << 32) | o[2 * j];
    uint64_t lo = w.v[j] * P.range;
    accmask |= (lo <= P.zone ? 1u : 0u) << j;
  }
}

// transfer function of one block; slots below `skip` are not part of the run.
SIMMR_DEV uint32_t outer_summary(uint32_t accmask, uint32_t skip) {
  uint32_t s0 = 0, c0 = 0, s1 = 1, c1 = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if ((uint32_t)j < skip) continue;
    uint32_t a = (accmask >> j) & 1u;
    if (s0) { c0++; s0 = 0; } else { s0 = a; }
    if (s1) { c1++; s1 = 0; } else { s1 = a; }
  }
  return fs_make(s0, c0, s1, c1);
}

// ordered reduction / exclusive scan of transfer functions inside a 256-thread
// workgroup.  Returns this thread's exclusive prefix; *total = whole workgroup.
SIMMR_DEV uint32_t wg_fs_exclusive_scan(uint32_t f, uint32_t* lds4, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = f;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc = fs_compose(o, inc);
  }
  uint32_t exc = __shfl_up(inc, 1, 64);
  if (lane == 0) exc = FS_IDENTITY;
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint32_t pre = FS_IDENTITY, tot = FS_IDENTITY;
  for (uint32_t wv = 0; wv < 4; wv++) {
    uint32_t t = lds4[wv];
    if (wv < wave) pre = fs_compose(pre, t);
    tot = fs_compose(tot, t);
  }
  __syncthreads();
  *total = tot;
  return fs_compose(pre, exc);
}

extern "C" __global__ void __launch_bounds__(256)
k_outer_classify(OuterParams P, uint64_t first_block, uint64_t n_blocks, uint32_t first_skip,
                 uint32_t* __restrict__ last_idx, uint32_t* __restrict__ wg_sums) {
  __shared__ uint32_t lds4[4];
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t f = FS_IDENTITY;
  if (t < n_blocks) {
    BlockWords w;
    uint32_t acc;
    outer_block(P, first_block + t, w, acc);
    f = outer_summary(acc, t == 0 ? first_skip : 0);
    last_idx[t] = (uint32_t)__umul64hi(w.v[7], P.range);
  }
  uint32_t total;
  (void)wg_fs_exclusive_scan(f, lds4, &total);
  if (threadIdx.x == 0) wg_sums[blockIdx.x] = total;
}

// One workgroup: resolves (start state, unit base) of every classify workgroup,
// the totals, and which workgroups hold units [first, first+count).
extern "C" __global__ void __launch_bounds__(256)
k_outer_scan(const uint32_t* __restrict__ wg_sums, uint64_t n_wg, uint64_t first, uint64_t count,
             OuterPrefix* __restrict__ wg_prefix, OuterScanResult* __restrict__ res) {
  __shared__ uint64_t s_c0[256], s_c1[256];
  __shared__ uint32_t s_e0[256], s_e1[256];
  __shared__ uint64_t s_base[256];
  __shared__ uint32_t s_state[256];
  const uint32_t t = threadIdx.x;
  // chunks are multiples of 4 entries so that they can be read 16 bytes at a time: the loop is a chain of
  // dependent-latency loads, and four entries per load make it four times shorter
  const uint64_t chunk = (((n_wg + 255) / 256) + 3) & ~(uint64_t)3;
  const uint64_t lo = (uint64_t)t * chunk < n_wg ? (uint64_t)t * chunk : n_wg, hi = (lo + chunk < n_wg) ? lo + chunk : n_wg;
  // compose my chunk with 64-bit counts
  uint32_t e0 = 0, e1 = 1;
  uint64_t c0 = 0, c1 = 0;
  auto compose = [&](uint32_t g) {
    uint32_t ge0 = g & 1u, ge1 = (g >> 1) & 1u;
    uint64_t gc0 = (g >> 2) & 0x7fffu, gc1 = g >> 17;
    c0 += e0 ? gc1 : gc0; e0 = e0 ? ge1 : ge0;
    c1 += e1 ? gc1 : gc0; e1 = e1 ? ge1 : ge0;
  };
  {
    uint64_t w = lo;
    for (; w + 4 <= hi; w += 4) {
      const uint4 g4 = *reinterpret_cast<const uint4*>(wg_sums + w);
      compose(g4.x); compose(g4.y); compose(g4.z); compose(g4.w);
    }
    for (; w < hi; w++) compose(wg_sums[w]);
  }
  s_e0[t] = e0; s_e1[t] = e1; s_c0[t] = c0; s_c1[t] = c1;
  __syncthreads();
  if (t == 0) {
    uint32_t st = 0;  // the run starts in NEED_IDX
    uint64_t base = 0;
    for (int i = 0; i < 256; i++) {
      s_state[i] = st; s_base[i] = base;
      base += st ? s_c1[i] : s_c0[i];
      st = st ? s_e1[i] : s_e0[i];
    }
    res->total_units = base;
    res->end_state = st;
    uint32_t st1 = 1;  // the same composition for a range entered in NEED_SEED (simmr_outer_summarize)
    uint64_t base1 = 0;
    for (int i = 0; i < 256; i++) {
      base1 += st1 ? s_c1[i] : s_c0[i];
      st1 = st1 ? s_e1[i] : s_e0[i];
    }
    res->total_units1 = base1;
    res->end_state1 = st1;
  }
  __syncthreads();
  uint32_t st = s_state[t];
  uint64_t base = s_base[t];
  const uint64_t last = first + count;  // exclusive
  auto emit = [&](uint64_t w, uint32_t g) {
    OuterPrefix p; p.base = base; p.state = st; p.pad = 0;
    wg_prefix[w] = p;
    uint64_t gc = st ? (uint64_t)(g >> 17) : (uint64_t)((g >> 2) & 0x7fffu);
    uint32_t ge = st ? ((g >> 1) & 1u) : (g & 1u);
    uint64_t nb = base + gc;
    if (count > 0) {
      if (base <= first && first < nb) res->wg_lo = w;
      if (base < last && last <= nb) res->wg_hi = w;
    }
    base = nb; st = ge;
  };
  {
    uint64_t w = lo;
    for (; w + 4 <= hi; w += 4) {
      const uint4 g4 = *reinterpret_cast<const uint4*>(wg_sums + w);
      emit(w, g4.x); emit(w + 1, g4.y); emit(w + 2, g4.z); emit(w + 3, g4.w);
    }
    for (; w < hi; w++) emit(w, wg_sums[w]);
  }
}

extern "C" __global__ void __launch_bounds__(256)
k_outer_emit(OuterParams P, uint64_t first_block, uint64_t n_blocks, uint32_t first_skip,
             uint64_t wg_lo, const uint32_t* __restrict__ last_idx,
             const OuterPrefix* __restrict__ wg_prefix, uint64_t first, uint64_t count,
             uint32_t* __restrict__ out_idx, uint64_t* __restrict__ out_seed,
             uint64_t* __restrict__ end_slot) {
  __shared__ uint32_t lds4[4];
  const uint64_t wg = wg_lo + blockIdx.x;
  const uint64_t t = wg * 256 + threadIdx.x;
  uint32_t f = FS_IDENTITY, acc = 0, skip = 0;
  BlockWords w;
  if (t < n_blocks) {
    outer_block(P, first_block + t, w, acc);
    skip = (t == 0) ? first_skip : 0;
    f = outer_summary(acc, skip);
  }
  uint32_t total;
  uint32_t pre = wg_fs_exclusive_scan(f, lds4, &total);
  if (t >= n_blocks) return;
  OuterPrefix wp = wg_prefix[wg];
  uint32_t st = wp.state ? ((pre >> 1) & 1u) : (pre & 1u);
  uint64_t c = wp.base + (wp.state ? (uint64_t)(pre >> 17) : (uint64_t)((pre >> 2) & 0x7fffu));
  uint32_t idx = 0;
  if (st) idx = last_idx[t - 1];  // the accepted index draw sits in the previous block
  const uint64_t last = first + count;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if ((uint32_t)j < skip) continue;
    if (st) {
      if (c >= first && c < last) {
        out_idx[c - first] = idx;
        out_seed[c - first] = w.v[j];
        if (c == last - 1) *end_slot = (first_block + t) * 8 + (uint64_t)j + 1;
      }
      c++;
      st = 0;
    } else if ((acc >> j) & 1u) {
      idx = (uint32_t)__umul64hi(w.v[j], P.range);
      st = 1;
    }
  }
}

// ===========================================================================
// 2b. Custom (empirical) PDFs: CustomPDF::sample (custom_short.rs:108-151) =
//     WeightedAliasIndex<f64>::sample, then Uniform<u32>::sample of the chosen bin.
// ===========================================================================
template <typename Rng>
SIMMR_DEV uint32_t pdf_sample_lane(Rng& rng, const CustomDev& C, const PdfDev pdf, bool* bad) {
  uint32_t c;
  for (;;) {  // uniform_index.sample
    const uint64_t m = (uint64_t)rng.next_u32() * pdf.n;
    if ((uint32_t)m <= pdf.idx_zone) { c = (uint32_t)(m >> 32); break; }
  }
  const double v01 = __longlong_as_double((long long)((rng.next_u64() >> 12) | 0x3FF0000000000000ULL)) - 1.0;
  const double x = __dmul_rn(v01, pdf.w_scale);
  const uint32_t bin = (x < C.odds[pdf.off + c]) ? c : C.alias[pdf.off + c];
  if (bin >= pdf.n_bins) { *bad = true; return 0; }
  const uint32_t range = C.bin_range[pdf.off_bins + bin], low = C.bin_low[pdf.off_bins + bin];
  if (range == 0) return rng.next_u32();
  const uint32_t zone = C.bin_zone[pdf.off_bins + bin];
  for (;;) {
    const uint64_t m = (uint64_t)rng.next_u32() * range;
    if ((uint32_t)m <= zone) return low + (uint32_t)(m >> 32);
  }
}

// Same draw sequence, reading the words of one StdRng stream staged in LDS
// (W[0 .. nw)); *ovf is set when the window is too short.
SIMMR_DEV uint32_t pdf_sample_words(const uint32_t* __restrict__ W, uint32_t nw, const CustomDev& C,
                                    const PdfDev pdf, bool* bad, bool* ovf) {
  uint32_t k = 0, c = 0;
  for (;;) {
    if (k >= nw) { *ovf = true; return 0; }
    const uint64_t m = (uint64_t)W[k++] * pdf.n;
    if ((uint32_t)m <= pdf.idx_zone) { c = (uint32_t)(m >> 32); break; }
  }
  if (k + 2 > nw) { *ovf = true; return 0; }
  const uint64_t bits = ((uint64_t)W[k + 1] << 32) | W[k];
  k += 2;
  const double v01 = __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ULL)) - 1.0;
  const double x = __dmul_rn(v01, pdf.w_scale);
  const uint32_t bin = (x < C.odds[pdf.off + c]) ? c : C.alias[pdf.off + c];
  if (bin >= pdf.n_bins) { *bad = true; return 0; }
  const uint32_t range = C.bin_range[pdf.off_bins + bin], low = C.bin_low[pdf.off_bins + bin];
  if (range == 0) { if (k >= nw) { *ovf = true; return 0; } return W[k]; }
  const uint32_t zone = C.bin_zone[pdf.off_bins + bin];
  for (;;) {
    if (k >= nw) { *ovf = true; return 0; }
    const uint64_t m = (uint64_t)W[k++] * range;
    if ((uint32_t)m <= zone) return low + (uint32_t)(m >> 32);
  }
}

// The same draws from the stream itself, for the sample that runs past the staged window (a dozen
// consecutive rejections; kept for exactness, not for speed): blocks are generated on demand.
__device__ __attribute__((noinline)) uint32_t pdf_sample_stream(uint64_t seed, const CustomDev& C, const PdfDev pdf,
                                                                bool* bad) {
  const Key key = pcg32_expand(seed);
  uint32_t blk[16];
  uint32_t cur = 0xffffffffu, k = 0;
  auto word = [&](uint32_t i) {
    if ((i >> 4) != cur) { cur = i >> 4; chacha12_block(key, cur, blk); }
    return blk[i & 15u];
  };
  const uint32_t cap = 1u << 16;  // words; a stream that rejects this often is a broken model
  uint32_t c = 0;
  for (;;) {
    if (k >= cap) { *bad = true; return 0; }
    const uint64_t m = (uint64_t)word(k++) * pdf.n;
    if ((uint32_t)m <= pdf.idx_zone) { c = (uint32_t)(m >> 32); break; }
  }
  const uint32_t lo = word(k), hi = word(k + 1);
  k += 2;
  const uint64_t bits = ((uint64_t)hi << 32) | lo;
  const double v01 = __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ULL)) - 1.0;
  const double x = __dmul_rn(v01, pdf.w_scale);
  const uint32_t bin = (x < C.odds[pdf.off + c]) ? c : C.alias[pdf.off + c];
  if (bin >= pdf.n_bins) { *bad = true; return 0; }
  const uint32_t range = C.bin_range[pdf.off_bins + bin], low = C.bin_low[pdf.off_bins + bin];
  if (range == 0) return word(k);
  const uint32_t zone = C.bin_zone[pdf.off_bins + bin];
  for (;;) {
    if (k >= cap) { *bad = true; return 0; }
    const uint64_t m = (uint64_t)word(k++) * range;
    if ((uint32_t)m <= zone) return low + (uint32_t)(m >> 32);
  }
}

// ===========================================================================
// 3. Per-unit planning (simulate.rs:211-258 for pairs, :478-491 for long reads)
//    One lane per unit; each lane owns a LaneRng (one ChaCha block in LDS).
// ===========================================================================

#define PLAN_THREADS 256

// simulate_pe_reads over several genomes in one plan (simulate.rs:110-150): every genome re-creates
// the outer StdRng with the same seed (simulate.rs:137,172), so genomes with the same number of
// sequences draw the same (contig, pe_seed) list; it is generated once per class and looked up here.
struct MultiGenome {
  uint64_t base;     // global index of the genome's first pair
  uint64_t cls_off;  // start of its class's list in cls_contig / cls_seed
  uint32_t slot;     // engine genome slot
  uint32_t pad;
};

// SIMMR_RNG_PHILOX_FULL: the outer draws (simulate.rs:172-186) as one Philox block per pair — no stream, no scan, no kernel
// of their own: k_plan_pe<true> makes them where it plans the pair.  Pair p of its genome's run takes the block with key =
// the run's seed and counter (p & 0xffffffff, 4 | (p >> 32) << 8, ..): contig = ((w0 | w1 << 32) * num_seqs) >> 64,
// pe_seed = w2 | w3 << 32.  mg == null: one genome (n_contigs0), pairs first .. ; else the plan over several genomes
// (k_multi_units' search).  The three columns are written for the emit kernels, which read them as in every other mode.
struct OuterCtrArgs {
  const MultiGenome* mg;
  uint32_t n_genomes, n_contigs0;
  uint64_t seed, first;
  uint32_t* u_genome_w;
  uint32_t* u_contig_w;
  uint64_t* u_seed_w;
};
SIMMR_DEV void outer_ctr_pair(const OuterCtrArgs& oc, const GenomeDev* __restrict__ genomes, uint64_t k, uint32_t genome0,
                              uint32_t* genome, uint32_t* contig, uint64_t* pe_seed) {
  uint64_t p = oc.first + k;
  uint64_t nc = oc.n_contigs0;
  *genome = genome0;
  if (oc.mg) {
    uint32_t lo = 0, hi = oc.n_genomes;  // last genome with base <= the global pair index
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (oc.mg[mid].base <= p) lo = mid; else hi = mid; }
    const MultiGenome g = oc.mg[lo];
    p -= g.base;
    *genome = g.slot;
    oc.u_genome_w[k] = g.slot;
    nc = genomes[g.slot].n_contigs;
  }
  uint32_t w[4];
  philox4x32_10((uint32_t)p, 4u | ((uint32_t)(p >> 32) << 8), (uint32_t)oc.seed, (uint32_t)(oc.seed >> 32), w);
  *contig = (uint32_t)__umul64hi((uint64_t)w[0] | ((uint64_t)w[1] << 32), nc);
  *pe_seed = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  oc.u_contig_w[k] = *contig;
  oc.u_seed_w[k] = *pe_seed;
}

extern "C" __global__ void __launch_bounds__(256)
k_multi_units(const MultiGenome* __restrict__ mg, uint32_t n_genomes, uint64_t first, uint64_t n_units,
              const uint32_t* __restrict__ cls_contig, const uint64_t* __restrict__ cls_seed,
              uint32_t* __restrict__ u_genome, uint32_t* __restrict__ u_contig, uint64_t* __restrict__ u_seed) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_units) return;
  const uint64_t gp = first + k;  // global pair index
  uint32_t lo = 0, hi = n_genomes;  // last genome with base <= gp (genomes without pairs share a base: take the last)
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (mg[mid].base <= gp) lo = mid; else hi = mid; }
  const MultiGenome g = mg[lo];
  const uint64_t p = gp - g.base;
  u_genome[k] = g.slot;
  u_contig[k] = cls_contig[g.cls_off + p];
  u_seed[k] = cls_seed[g.cls_off + p];
}

template <bool CTR>
SIMMR_DEV uint32_t k_plan_pe_unit(const ProfileDev& prof, const GenomeDev* __restrict__ genomes, uint32_t genome, uint64_t k,
                                  const uint32_t* __restrict__ u_contig, const uint64_t* __restrict__ u_seed,
                                  const uint32_t* __restrict__ u_genome, const PlanArrays& pl, const Tables* __restrict__ T,
                                  uint32_t* __restrict__ err, uint32_t* rows, bool given, uint32_t genome_v, uint32_t contig_v,
                                  uint64_t seed_v);
#define SCAN_THREADS 256
#define SCAN_ITEMS 8 /* per thread */

// CTR: SIMMR_RNG_PHILOX_FULL — the pair's generators are the word streams W(seed) (rng_device.hpp: LaneRngT<true>)
template <bool CTR>
__global__ void __launch_bounds__(PLAN_THREADS)
k_plan_pe(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint32_t genome, uint64_t n_units,
          const uint32_t* __restrict__ u_contig, const uint64_t* __restrict__ u_seed,
          const uint32_t* __restrict__ u_genome, PlanArrays pl, const Tables* __restrict__ T,
          uint32_t* __restrict__ err, unsigned long long* __restrict__ tile_bytes, uint32_t slot_round,
          unsigned long long* __restrict__ wave_bytes, OuterCtrArgs oc) {
  __shared__ uint32_t rows[PLAN_THREADS * 17];
  uint64_t k = (uint64_t)blockIdx.x * PLAN_THREADS + threadIdx.x;
  uint32_t genome_v = genome, contig_v = 0;
  uint64_t seed_v = 0;
  if (CTR && k < n_units) outer_ctr_pair(oc, genomes, k, genome, &genome_v, &contig_v, &seed_v);  // (the pair's outer draws)
  const uint32_t planned = k < n_units ? k_plan_pe_unit<CTR>(prof, genomes, genome, k, u_contig, u_seed, u_genome, pl, T, err, rows,
                                                            CTR, genome_v, contig_v, seed_v) : 0u;
  // The bytes this workgroup's pairs will write, added to the sum of their tile of the offset scan (SCAN_THREADS *
  // SCAN_ITEMS units: a multiple of this workgroup's 256), so that the scan needs no pass of its own to reduce them.
  // Or (wave_bytes: the plans the counter-mode emit kernel serves) the bytes of every 64 pairs by themselves: that kernel
  // takes a block's first output byte from the scan of THESE and places the block's reads with a scan of its own, so the
  // per-pair offsets (50 M entries, 0.4 ms) are never made.
  if (tile_bytes || wave_bytes) {
    unsigned long long s = 2ull * ((planned + slot_round) & ~slot_round);  // (slot_round = 15 for 16-byte read slots, else 0)
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63u) == 0) {
      // (a wave whose first pair lies beyond the shard has no entry: the array holds ceil(n_units / 64) sums)
      if (wave_bytes) { if (k < n_units) wave_bytes[k >> 6] = s; }
      else if (s) atomicAdd(&tile_bytes[((uint64_t)blockIdx.x * PLAN_THREADS) / (SCAN_THREADS * SCAN_ITEMS)], s);
    }
  }
}

// one pair (the body of k_plan_pe); returns its read length L, 0 for a pair that cannot be planned
template <bool CTR>
SIMMR_DEV uint32_t k_plan_pe_unit(const ProfileDev& prof, const GenomeDev* __restrict__ genomes, uint32_t genome, uint64_t k,
                                  const uint32_t* __restrict__ u_contig, const uint64_t* __restrict__ u_seed,
                                  const uint32_t* __restrict__ u_genome, const PlanArrays& pl, const Tables* __restrict__ T,
                                  uint32_t* __restrict__ err, uint32_t* rows, bool given, uint32_t genome_v, uint32_t contig_v,
                                  uint64_t seed_v) {
  // (given: the outer draws were made by the caller — SIMMR_RNG_PHILOX_FULL — instead of read from the columns)
  const GenomeDev G = genomes[given ? genome_v : (u_genome ? u_genome[k] : genome)];  // u_genome: several genomes in one plan
  const uint32_t contig_k = given ? contig_v : u_contig[k];
  const uint64_t size = G.contigs[contig_k].size;
  const uint64_t pe_seed = given ? seed_v : u_seed[k];
  LaneRngT<CTR> rng;
  rng.seed_from_u64(pe_seed, rows + threadIdx.x * 17);  // (CTR: W(pe_seed), rng_device.hpp)
  uint64_t L = prof.read_length, I = prof.insert_size;
  if (prof.kind == SIMMR_K_MINIMAL_SHORT) {
    // minimal_short.rs:33-42 and :58-67: both re-seed with pe_seed, so both see
    // the same standard-normal draw.
    double z = rng.standard_normal(T);
    L = sat_u16_f64(floor(__dadd_rn((double)prof.read_length, __dmul_rn(prof.read_length_std, z))));
    I = sat_u16_f64(floor(__dadd_rn((double)prof.insert_size, __dmul_rn(prof.insert_size_std, z))));
    rng.restart();  // simulate.rs:227: fresh StdRng::seed_from_u64(pe_seed)
  } else if (prof.kind == SIMMR_K_CUSTOM) {
    // custom_short.rs:237-270: each getter samples its PDF with a fresh StdRng(pe_seed), `as u16`
    bool bad = false;
    L = pdf_sample_lane(rng, prof.custom, prof.custom.pdfs[0], &bad) & 0xffffu;
    I = 0;
    if (prof.custom.pdfs[1].n) {
      rng.restart();
      I = pdf_sample_lane(rng, prof.custom, prof.custom.pdfs[1], &bad) & 0xffffu;
    }
    if (bad) atomicOr(err, SIMMR_ERRBIT_PDF);
    rng.restart();
  }
  const uint64_t required = prof.required;
  if (size <= required) { atomicOr(err, SIMMR_ERRBIT_GENOME); return 0u; }
  uint8_t flags = SIMMR_FLAG_REVCOMP;
  uint64_t fs = rng.gen_range_u64(0, size - required);  // simulate.rs:233
  uint64_t re;
  if (fs + I >= size || fs + I + L >= size) {           // simulate.rs:241-247
    re = rng.gen_range_u64(fs, size - required);
    flags |= SIMMR_FLAG_REDRAWN;
  } else if ((int32_t)((uint32_t)(fs + I) - (uint32_t)L) < 0) {  // simulate.rs:250-251
    re = 0;
  } else {
    re = fs + I - L;                                    // simulate.rs:253-256
  }
  // simulate.rs:266,270: rng.gen::<Option<u64>>() twice (bool, then u64 if Some)
  uint64_t qs, ms;
  if (rng.gen_bool()) qs = rng.next_u64(); else { qs = entropy_substitute(pe_seed, 1); flags |= SIMMR_FLAG_QSEED_SUBST; }
  if (CTR) ms = 0;  // (mate 2's mutation seed: the stream's last draw, read by no kernel of the counter modes, its flag cleared below)
  else if (rng.gen_bool()) ms = rng.next_u64(); else { ms = entropy_substitute(pe_seed, 2); flags |= SIMMR_FLAG_MSEED_SUBST; }
  if (prof.kind == SIMMR_K_PERFECT_SHORT) flags &= (uint8_t)~(SIMMR_FLAG_QSEED_SUBST | SIMMR_FLAG_MSEED_SUBST);
  if (prof.kind == SIMMR_K_CUSTOM || prof.rng_mode != SIMMR_RNG_REFERENCE) flags &= (uint8_t)~SIMMR_FLAG_MSEED_SUBST;  // drawn but never used
  // Rust would panic on an out-of-range slice; never silently read out of bounds.
  const uint64_t len = G.contigs[contig_k].len;
  if (fs + L > len || re + L > len) { atomicOr(err, SIMMR_ERRBIT_SLICE); L = 0; }
  if (L > LONGREAD_MAXL) atomicOr(err, SIMMR_NOTEBIT_LONGREAD);
  pl.len[k] = (uint32_t)L;
  pl.a[k] = fs;
  pl.b[k] = re;
  pl.flags[k] = flags;
  if (pl.qs2) pl.qs2[k] = qs;
  if (pl.ms2) pl.ms2[k] = ms;  // not kept when no kernel will read it (counter mode, custom profiles)
  return (uint32_t)L;
}

// the run of consecutive long reads that holds global read index gi: the last r with runs[r].first_read <= gi
// (binary search: a run per genome, and BASELINE config 4 has a thousand genomes)
SIMMR_DEV uint32_t find_run(const LongGenomeRun* __restrict__ runs, uint32_t n_runs, uint64_t gi) {
  uint32_t lo = 0, hi = n_runs;  // invariant: runs[lo].first_read <= gi (runs[0].first_read is the plan's first read or earlier)
  while (hi - lo > 1u) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (runs[mid].first_read <= gi) lo = mid; else hi = mid;
  }
  return lo;
}

// long reads, reference mode: contig + read_seed come from the outer stream,
// the length is the run-wide constant (simulate.rs:358 with Some(seed)).
extern "C" __global__ void __launch_bounds__(PLAN_THREADS)
k_plan_long_ref(const GenomeDev* __restrict__ genomes, const LongGenomeRun* __restrict__ runs,
                uint32_t n_runs, uint64_t first_unit, uint64_t n_units, uint32_t L0, uint32_t uniform_start,
                uint32_t* __restrict__ u_contig, uint32_t* __restrict__ u_genome,
                const uint64_t* __restrict__ u_seed, PlanArrays pl, uint32_t* __restrict__ err) {
  __shared__ uint32_t rows[PLAN_THREADS * 17];
  uint64_t k = (uint64_t)blockIdx.x * PLAN_THREADS + threadIdx.x;
  if (k >= n_units) return;
  const uint64_t gi = first_unit + k;  // global read index
  const LongGenomeRun run = runs[find_run(runs, n_runs, gi)];
  const GenomeDev G = genomes[run.genome];
  const uint32_t contig = run.usable[u_contig[k]];  // idx-th usable sequence (simulate.rs:375)
  const uint64_t size = G.contigs[contig].size;
  LaneRng rng;
  rng.seed_from_u64(u_seed[k], rows + threadIdx.x * 17);
  // simulate.rs:484 draws the start in [0, L0); SIMMR_START_UNIFORM in [0, size - L0) (usable: size > L0)
  uint64_t s = rng.gen_range_u64(0, uniform_start ? size - L0 : (uint64_t)L0);
  uint64_t e = s + L0;                            // :485
  if (e >= size) e = rng.gen_range_u64(s, size);  // :488-491 (never taken with a uniform start)
  if (e > G.contigs[contig].len) { atomicOr(err, SIMMR_ERRBIT_SLICE); e = s; }
  u_contig[k] = contig;
  u_genome[k] = run.genome;
  pl.len[k] = (uint32_t)(e - s);
  pl.a[k] = s;
  pl.b[k] = e;
  pl.flags[k] = 0;
}

// long reads, per-read mode (include/simmr_hip.h SIMMR_LEN_PER_READ): length,
// contig and read seed all come from StdRng(per_read_seed(seed, read index)).
template <bool CTR>  // (CTR: SIMMR_RNG_PHILOX_FULL, as in k_plan_pe)
__global__ void __launch_bounds__(PLAN_THREADS)
k_plan_long_per_read(ProfileDev prof, const GenomeDev* __restrict__ genomes,
                     const LongGenomeRun* __restrict__ runs, uint32_t n_runs, uint64_t seed,
                     uint64_t first_unit, uint64_t n_units, uint32_t* __restrict__ u_contig,
                     uint32_t* __restrict__ u_genome, uint64_t* __restrict__ u_seed, PlanArrays pl,
                     const Tables* __restrict__ T, uint32_t* __restrict__ err) {
  __shared__ uint32_t rows[PLAN_THREADS * 17];
  uint64_t k = (uint64_t)blockIdx.x * PLAN_THREADS + threadIdx.x;
  if (k >= n_units) return;
  const uint64_t gi = first_unit + k;
  const LongGenomeRun run = runs[find_run(runs, n_runs, gi)];
  const GenomeDev G = genomes[run.genome];
  LaneRngT<CTR> rng;
  rng.seed_from_u64(per_read_seed(seed, gi), rows + threadIdx.x * 17);
  uint32_t L = 0, contig = 0;
  uint64_t read_seed = 0;
  for (int tries = 0;; tries++) {
    if (prof.kind == SIMMR_K_CUSTOM)  // custom_short.rs:286-301 (the model's mean rides in insert_size_std, see k_const_length)
      L = sat_u16_f64(floor(__dadd_rn(prof.insert_size_std, __dmul_rn(prof.read_length_std, rng.standard_normal(T)))));
    else
      L = sat_u16_f32(floorf(rng.gamma_f32(T, prof.gamma_shape, prof.gamma_scale)));
    if (L == 0 || run.max_size <= L) {
      if (tries > 1000) { atomicOr(err, SIMMR_ERRBIT_GENOME); L = 0; break; }
      continue;
    }
    uint64_t n_us = 0;
    for (uint32_t c = 0; c < G.n_contigs; c++) n_us += (G.contigs[c].size > L) ? 1u : 0u;
    uint64_t idx = rng.gen_range_u64(0, n_us);
    read_seed = rng.next_u64();
    uint64_t seen = 0;
    for (contig = 0; contig < G.n_contigs; contig++)
      if (G.contigs[contig].size > L) { if (seen == idx) break; seen++; }
    break;
  }
  uint64_t s = 0, e = 0;
  if (L) {
    const uint64_t size = G.contigs[contig].size;
    rng.seed_from_u64(read_seed, rows + threadIdx.x * 17);
    s = rng.gen_range_u64(0, prof.long_start_uniform ? size - L : (uint64_t)L);
    e = s + L;
    if (e >= size) e = rng.gen_range_u64(s, size);
    if (e > G.contigs[contig].len) { atomicOr(err, SIMMR_ERRBIT_SLICE); e = s; }
  }
  u_contig[k] = contig;
  u_genome[k] = run.genome;
  u_seed[k] = read_seed;
  pl.len[k] = (uint32_t)(e - s);
  pl.a[k] = s;
  pl.b[k] = e;
  pl.flags[k] = 0;
}

// get_random_read_length(seed) for the whole run (simulate.rs:358): one lane.
extern "C" __global__ void k_const_length(ProfileDev prof, uint64_t seed,
                                          const Tables* __restrict__ T, uint32_t* __restrict__ out) {
  __shared__ uint32_t row[17];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  LaneRng rng;
  rng.seed_from_u64(seed, row);
  if (prof.kind == SIMMR_K_CUSTOM) {
    // custom_short.rs:286-301: get_random_read_length is Normal<f64>(read_length_mean, read_length_std),
    // floor, `as u16` (the PDF only serves get_read_length); the model's values ride in read_length_std / insert_size_std
    const double z = rng.standard_normal(T);
    *out = sat_u16_f64(floor(__dadd_rn(prof.insert_size_std, __dmul_rn(prof.read_length_std, z))));
    return;
  }
  *out = sat_u16_f32(floorf(rng.gamma_f32(T, prof.gamma_shape, prof.gamma_scale)));
}

// ===========================================================================
// 4. Exclusive scan of per-unit byte counts -> CSR unit offsets
// ===========================================================================
SIMMR_DEV uint64_t wg_exclusive_scan_u64(uint64_t v, uint64_t* lds4, uint64_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += o;
  }
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint64_t pre = 0, tot = 0;
  for (uint32_t wv = 0; wv < 4; wv++) {
    uint64_t t = lds4[wv];
    if (wv < wave) pre += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return pre + inc - v;
}

// T = uint64_t (byte counts) or uint32_t (read lengths, `scale` reads per unit: the bytes a unit writes; every length is
// first rounded up to a multiple of round + 1: 16-byte read slots, SIMMR_SLOT16)
template <typename T>
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_reduce(const T* __restrict__ in, uint64_t n, uint32_t scale, uint32_t round, uint64_t* __restrict__ wg_tot) {
  __shared__ uint64_t lds4[4];
  uint64_t base = ((uint64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += (((uint64_t)in[base + i] + round) & ~(uint64_t)round) * scale;
  uint64_t tot;
  (void)wg_exclusive_scan_u64(s, lds4, &tot);
  if (threadIdx.x == 0) wg_tot[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of wg_tot in place; writes the grand total.
extern "C" __global__ void __launch_bounds__(SCAN_THREADS)
k_scan_tops(uint64_t* __restrict__ wg_tot, uint64_t n_wg, uint64_t* __restrict__ grand) {
  __shared__ uint64_t lds4[4];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n_wg; base += SCAN_THREADS) {
    uint64_t i = base + threadIdx.x;
    uint64_t v = (i < n_wg) ? wg_tot[i] : 0;
    uint64_t tot;
    uint64_t ex = wg_exclusive_scan_u64(v, lds4, &tot);
    uint64_t carry = carry_s;
    if (i < n_wg) wg_tot[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *grand = carry_s;
}

template <typename T>
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_apply(const T* __restrict__ in, uint64_t n, uint32_t scale, uint32_t round, const uint64_t* __restrict__ wg_tot,
             uint64_t* __restrict__ out /* n + 1 */) {
  __shared__ uint64_t lds4[4];
  uint64_t base = ((uint64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? (((uint64_t)in[base + i] + round) & ~(uint64_t)round) * scale : 0; s += v[i]; }
  uint64_t tot;
  uint64_t ex = wg_exclusive_scan_u64(s, lds4, &tot) + wg_tot[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
    if (base + i == n - 1) out[n] = ex;
  }
}

// ===========================================================================
// 5. Emit: metadata columns
// ===========================================================================
extern "C" __global__ void __launch_bounds__(256)
k_write_meta(uint32_t paired, uint64_t n_units, uint64_t first_unit, uint32_t read_id_base,
             uint32_t genome_const, PlanArrays pl, const uint64_t* __restrict__ u_off,
             const uint32_t* __restrict__ u_contig, const uint32_t* __restrict__ u_genome,
             OutCols o) {
  uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k > n_units) return;
  if (k == n_units) {  // closing CSR offset
    o.seq_off[paired ? 2 * n_units : n_units] = u_off[n_units];
    return;
  }
  const uint64_t off = u_off[k];
  const uint32_t L = pl.len[k];
  const uint32_t id = read_id_base + (uint32_t)(first_unit + k);  // simulate.rs:85-89,274
  const uint32_t contig = u_contig[k];
  const uint32_t genome = u_genome ? u_genome[k] : genome_const;
  if (paired) {
    const uint64_t fs = pl.a[k], re = pl.b[k];
    const uint64_t r0 = 2 * k, r1 = 2 * k + 1;
    o.seq_off[r0] = off;
    o.seq_off[r1] = off + L;
    if (o.start) { o.start[r0] = fs; o.start[r1] = re + L; }  // simulate.rs:289,295
    if (o.end) { o.end[r0] = fs + L; o.end[r1] = re; }        // simulate.rs:290,296
    if (o.contig) { o.contig[r0] = contig; o.contig[r1] = contig; }
    if (o.genome) { o.genome[r0] = genome; o.genome[r1] = genome; }
    if (o.read_id) { o.read_id[r0] = id; o.read_id[r1] = id; }
    if (o.flags) { o.flags[r0] = 0; o.flags[r1] = pl.flags[k]; }
  } else {
    o.seq_off[k] = off;
    if (o.start) o.start[k] = pl.a[k];  // simulate.rs:515
    if (o.end) o.end[k] = pl.b[k];      // simulate.rs:516
    if (o.contig) o.contig[k] = contig;
    if (o.genome) o.genome[k] = genome;
    if (o.read_id) o.read_id[k] = id;
    if (o.flags) o.flags[k] = pl.flags[k];
  }
}

// ===========================================================================
// 6. Emit: perfect-short pairs (perfect_short.rs:42-54) — pure data movement.
//
// Every thread produces one ALIGNED 16-byte chunk of the compact seq[] stream
// and the matching chunk of qual[].  A chunk may straddle two reads (L is not a
// multiple of 16): both pieces are gathered as 2-bit codes (mate 2 complement-
// reversed in the code domain), spliced with shifts, and only then expanded to
// ASCII with v_perm_b32 — so all stores are global_store_dwordx4.
// ===========================================================================

// 16 bases (32 bits of codes) starting at absolute base position p (may be
// slightly negative: the plane has front padding).
// The planes live in device memory: say so (a pointer that came out of a struct or out of LDS is a
// generic pointer, and a flat load is slower than a global one).
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef v4u32 __attribute__((aligned(1))) v4u32_unaligned;
// A store into the output streams and columns.  They are written once and read by nobody on the device, so where a wave
// writes WHOLE lines the stores are nontemporal (`nt`): the lines are not kept in L2 / MALL behind the write.  Same-box
// A/B (profiles/r3/ab_nt_stores_*): k_emit_perfect_pe 6.3 against 7.4 ms per 100 M reads, k_emit_philox in the slot
// layout 11.4 against 11.9 — and in the compact layout, whose 16-byte stores straddle lines and whose read ends are
// written bytewise, 15.3 against 12.7: there the stores stay plain.  -DSIMMR_PLAIN_STORES: the A side.
// (a macro, not a function template: the pointee types carry `aligned(1)`, which template deduction would drop)
#if defined(SIMMR_PLAIN_STORES)
#define stream_store(p, v) (*(p) = (v))
#else
#define stream_store(p, v) __builtin_nontemporal_store((v), (p))
#endif
typedef const __attribute__((address_space(1))) u64_unaligned* global_u64_unaligned_ptr;
SIMMR_DEV uint64_t load_plane_u64(const uint32_t* __restrict__ plane, int64_t word) {
  return *(global_u64_unaligned_ptr)(plane + word);  // words `word` and `word + 1`, 4-byte aligned
}
SIMMR_DEV uint32_t fetch_codes16(const uint32_t* __restrict__ packed, int64_t p) {
  return (uint32_t)(load_plane_u64(packed, p >> 4) >> ((uint32_t)(p & 15) * 2u));
}
SIMMR_DEV uint32_t fetch_mask16(const uint32_t* __restrict__ mask, int64_t p) {
  return (uint32_t)(load_plane_u64(mask, p >> 5) >> (uint32_t)(p & 31)) & 0xffffu;
}
// reverse the order of the sixteen 2-bit groups
SIMMR_DEV uint32_t reverse_groups16(uint32_t x) {
  const uint32_t y = __builtin_bitreverse32(x);
  // each bit pair swapped back: odd result bits from y << 1, even ones from y >> 1 — one three-input select (v_bitop3_b32,
  // truth table of c ? a : b with a = y >> 1, b = y << 1, c = 0x55555555)
  return __builtin_amdgcn_bitop3_b32(y >> 1, y << 1, 0x55555555u, 0xe4);
}
// ~reverse_groups16(x): the complement folded into the select's truth table
SIMMR_DEV uint32_t reverse_complement_groups16(uint32_t x) {
  const uint32_t y = __builtin_bitreverse32(x);
  return __builtin_amdgcn_bitop3_b32(y >> 1, y << 1, 0x55555555u, 0x1b);
}
// low bit of each of the sixteen 2-bit codes -> 16 bits
SIMMR_DEV uint32_t c16_odd(uint32_t c) {
  c &= 0x55555555u;
  c = (c | (c >> 1)) & 0x33333333u;
  c = (c | (c >> 2)) & 0x0F0F0F0Fu;
  c = (c | (c >> 4)) & 0x00FF00FFu;
  c = (c | (c >> 8)) & 0x0000FFFFu;
  return c;
}
// 16 mask bits -> 16 two-bit groups (each bit duplicated)
SIMMR_DEV uint32_t spread16(uint32_t m) {
  m = (m | (m << 8)) & 0x00FF00FFu;
  m = (m | (m << 4)) & 0x0F0F0F0Fu;
  m = (m | (m << 2)) & 0x33333333u;
  m = (m | (m << 1)) & 0x55555555u;
  return m * 3u;
}
// 4 codes (8 bits) + 4 exception bits -> 4 ASCII bytes
SIMMR_DEV uint32_t expand4(uint32_t c8, uint32_t m4) {
  uint32_t sel = (c8 | (c8 << 6) | (c8 << 12) | (c8 << 18)) & 0x03030303u;
  uint32_t ms = (m4 | (m4 << 7) | (m4 << 14) | (m4 << 21)) & 0x01010101u;
  sel |= ms << 2;
  // selector 0-3 -> "ACGT" (src1), 4-7 -> "N-N-" (src0)
  return __builtin_amdgcn_perm(0x2D4E2D4Eu, 0x54474341u, sel);
}

struct PieceSrc {
  int64_t pos;   // absolute base position of output byte 0's source
  uint32_t rev;  // mate 2: byte k comes from pos - k, complemented
};

SIMMR_DEV void gather_piece(const GenomeDev& G, const PieceSrc& s, uint32_t k, uint32_t& codes,
                            uint32_t& exc) {
  // codes for output bytes k .. k+15 of this read (garbage past the read end)
  if (!s.rev) {
    int64_t p = s.pos + (int64_t)k;
    codes = fetch_codes16(G.packed, p);
    exc = G.has_exc ? fetch_mask16(G.mask, p) : 0u;
  } else {
    int64_t p = s.pos - (int64_t)k - 15;
    uint32_t c = fetch_codes16(G.packed, p);
    codes = ~reverse_groups16(c);  // complement = 3 - code
    if (G.has_exc) {
      uint32_t m = fetch_mask16(G.mask, p);
      exc = __builtin_bitreverse32(m) >> 16;
      codes ^= spread16(exc);  // exceptions ('N', '-') are their own complement
    } else {
      exc = 0u;
    }
  }
}

#define PERFECT_GROUP 256u /* reads per workgroup iteration: 256*L is a multiple of 16 */
#define PERFECT_UNROLL 2

// A workgroup takes 256 consecutive reads (128 pairs): their 256*L output bytes start at a multiple
// of 16, so the group is a whole number of aligned 16-byte chunks except at the end of the shard.
// Each thread first writes the source position of one read to LDS (one coalesced pass over the plan
// columns, one contig-table lookup per read instead of one per chunk); the chunk loop then needs
// only LDS and the packed plane.  Chunk -> (read, offset) advances incrementally (no division).
// MULTI: the plan spans several genomes (u_genome[] per pair); the planes of a read then come from LDS.
template <bool MULTI>
__global__ void __launch_bounds__(256)
k_emit_perfect_pe(const GenomeDev* __restrict__ genomes, uint32_t genome, const uint32_t* __restrict__ u_genome,
                  uint32_t any_exc, uint64_t n_units,
                  uint32_t L, PlanArrays pl, const uint32_t* __restrict__ u_contig,
                  uint8_t* __restrict__ seq, uint8_t* __restrict__ qual, uint32_t qual_byte,
                  uint64_t first_unit, uint32_t read_id_base, OutCols o, unsigned long long* __restrict__ counters) {
  __shared__ int64_t r_pos[PERFECT_GROUP];  // absolute base position of output byte 0's source
  __shared__ uint32_t asc[256];             // four 2-bit codes -> four ASCII bytes (genomes without exceptions)
  {
    const uint32_t t = threadIdx.x, acgt = 0x54474341u;  // "ACGT"
    asc[t] = ((acgt >> (8 * (t & 3u))) & 0xffu) | (((acgt >> (8 * ((t >> 2) & 3u))) & 0xffu) << 8) |
             (((acgt >> (8 * ((t >> 4) & 3u))) & 0xffu) << 16) | (((acgt >> (8 * (t >> 6))) & 0xffu) << 24);
  }
  __shared__ const uint32_t* r_packed[MULTI ? PERFECT_GROUP : 1];
  __shared__ const uint32_t* r_mask[MULTI ? PERFECT_GROUP : 1];
  GenomeDev G = genomes[genome];
  if (MULTI) G.has_exc = any_exc;  // uniform: does any genome of the plan have an exception plane
  // the planes of read rr of the current group
  auto planes = [&](uint32_t rr) {
    GenomeDev R = G;
    if (MULTI) { R.packed = r_packed[rr]; R.mask = r_mask[rr]; R.has_exc = r_mask[rr] != nullptr; }
    return R;
  };
  const uint64_t n_reads = 2 * n_units;
  const uint64_t n_groups = (n_reads + PERFECT_GROUP - 1) / PERFECT_GROUP;
  const uint32_t q4 = qual_byte * 0x01010101u;
  const uint4 qv = make_uint4(q4, q4, q4, q4);
  // this thread's chunks are cl = threadIdx.x + 256 j: byte lb = 16 cl advances by 4096 per step
  const uint32_t step_r = 4096u / L, step_k = 4096u % L;
  const uint32_t rl0 = (threadIdx.x << 4) / L, k00 = (threadIdx.x << 4) - rl0 * L;
  for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint64_t r_base = g * PERFECT_GROUP;
    const uint32_t n_in = (n_reads - r_base) < PERFECT_GROUP ? (uint32_t)(n_reads - r_base) : PERFECT_GROUP;
    const uint32_t gbytes = n_in * L;  // < 2^24
    const uint32_t n_chunks = (gbytes + 15u) >> 4;
    const uint64_t gbyte0 = r_base * L;  // multiple of 16
    __syncthreads();  // the previous group's chunks are done with r_pos
    if (threadIdx.x < n_in) {
      const uint64_t r = r_base + threadIdx.x;
      const uint64_t u = r >> 1;
      const uint32_t contig = u_contig[u];
      const uint32_t rev = (uint32_t)(r & 1u);
      const uint64_t pos = rev ? pl.b[u] : pl.a[u];
      const uint32_t gslot = MULTI ? u_genome[u] : genome;
      const GenomeDev Gr = MULTI ? genomes[gslot] : G;
      if (MULTI) { r_packed[threadIdx.x] = Gr.packed; r_mask[threadIdx.x] = Gr.has_exc ? Gr.mask : nullptr; }
      r_pos[threadIdx.x] = (int64_t)(Gr.contigs[contig].base + (rev ? pos + L - 1 : pos));  // mate 2: byte k comes from pos - k
      // metadata columns of this read (k_write_meta is not launched for this kernel)
      stream_store(&o.seq_off[r], (uint64_t)(r * L));
      if (r + 1 == n_reads) o.seq_off[n_reads] = n_reads * L;  // closing CSR offset
      if (o.start) stream_store(&o.start[r], (uint64_t)(rev ? pos + L : pos));  // simulate.rs:289,295
      if (o.end) stream_store(&o.end[r], (uint64_t)(rev ? pos : pos + L));      // simulate.rs:290,296
      if (o.contig) stream_store(&o.contig[r], contig);
      if (o.genome) stream_store(&o.genome[r], gslot);
      if (o.read_id) stream_store(&o.read_id[r], read_id_base + (uint32_t)(first_unit + u));  // simulate.rs:85-89,274
      if (o.flags) stream_store(&o.flags[r], rev ? (uint8_t)pl.flags[u] : (uint8_t)0);
    }
    __syncthreads();
    uint32_t rl = rl0, k0 = k00;
    if (L >= 16u) {
      // A chunk has at most two pieces.  PERFECT_UNROLL chunks per thread are in flight at once: all their
      // packed-plane loads are issued before the first is consumed.
      for (uint32_t cl0 = threadIdx.x; cl0 < n_chunks; cl0 += 256 * PERFECT_UNROLL) {
        uint64_t raw[PERFECT_UNROLL][2], rawm[PERFECT_UNROLL][2];
        uint32_t sh[PERFECT_UNROLL][2], shm[PERFECT_UNROLL][2], crl[PERFECT_UNROLL], cna[PERFECT_UNROLL];
#pragma unroll
        for (int j = 0; j < PERFECT_UNROLL; j++) {
          const uint32_t cl = cl0 + 256u * j;
          crl[j] = rl;
          cna[j] = (L - k0) < 16u ? (L - k0) : 16u;
          raw[j][0] = raw[j][1] = rawm[j][0] = rawm[j][1] = 0;
          sh[j][0] = sh[j][1] = shm[j][0] = shm[j][1] = 0;
          if (cl < n_chunks) {
#pragma unroll
            for (int pc = 0; pc < 2; pc++) {
              const uint32_t r = rl + pc;
              if (pc == 1 && !(cna[j] < 16u && r < n_in)) continue;
              const int64_t pos = r_pos[r];
              const int64_t p = (r & 1u) ? pos - (int64_t)(pc ? 0u : k0) - 15 : pos + (int64_t)(pc ? 0u : k0);
              raw[j][pc] = load_plane_u64(MULTI ? r_packed[r] : G.packed, p >> 4);
              sh[j][pc] = (uint32_t)(p & 15) * 2u;
              if (G.has_exc) {
                const uint32_t* mk = MULTI ? r_mask[r] : G.mask;
                if (mk) rawm[j][pc] = load_plane_u64(mk, p >> 5);
                shm[j][pc] = (uint32_t)(p & 31);
              }
            }
          }
          rl += step_r;
          k0 += step_k;
          if (k0 >= L) { k0 -= L; rl++; }
        }
#pragma unroll
        for (int j = 0; j < PERFECT_UNROLL; j++) {
          const uint32_t cl = cl0 + 256u * j;
          if (cl >= n_chunks) continue;
          uint32_t pc_codes[2], pc_exc[2];
#pragma unroll
          for (int pc = 0; pc < 2; pc++) {
            uint32_t c = (uint32_t)(raw[j][pc] >> sh[j][pc]);
            uint32_t m = G.has_exc ? ((uint32_t)(rawm[j][pc] >> shm[j][pc]) & 0xffffu) : 0u;
            if ((crl[j] + pc) & 1u) {  // mate 2: complement-reverse in the code domain
              c = ~reverse_groups16(c);
              if (G.has_exc) { m = __builtin_bitreverse32(m) >> 16; c ^= spread16(m); }
            }
            pc_codes[pc] = c;
            pc_exc[pc] = m;
          }
          uint32_t codes = pc_codes[0], exc = pc_exc[0];
          const uint32_t na = cna[j];
          if (na < 16u) {
            codes = (codes & ((1u << (2 * na)) - 1u)) | (pc_codes[1] << (2 * na));
            exc = (exc & ((1u << na) - 1u)) | (pc_exc[1] << na);
          }
          uint4 out;
          if (G.has_exc) {
            out.x = expand4(codes & 0xffu, exc & 0xfu);
            out.y = expand4((codes >> 8) & 0xffu, (exc >> 4) & 0xfu);
            out.z = expand4((codes >> 16) & 0xffu, (exc >> 8) & 0xfu);
            out.w = expand4(codes >> 24, (exc >> 12) & 0xfu);
          } else {
            out = make_uint4(asc[codes & 0xffu], asc[(codes >> 8) & 0xffu], asc[(codes >> 16) & 0xffu], asc[codes >> 24]);
          }
          const uint32_t lb = cl << 4;
          const uint64_t byte0 = gbyte0 + lb;
          if (lb + 16u <= gbytes) {
            stream_store(reinterpret_cast<v4u32*>(seq + byte0), (v4u32{out.x, out.y, out.z, out.w}));
            stream_store(reinterpret_cast<v4u32*>(qual + byte0), (v4u32{qv.x, qv.y, qv.z, qv.w}));
          } else {  // last partial chunk of the shard
            const uint32_t words[4] = {out.x, out.y, out.z, out.w};
            for (uint32_t i = 0; lb + i < gbytes; i++) {
              seq[byte0 + i] = (uint8_t)(words[i >> 2] >> (8 * (i & 3)));
              qual[byte0 + i] = (uint8_t)qual_byte;
            }
          }
        }
      }
      continue;
    }
    for (uint32_t cl = threadIdx.x; cl < n_chunks; cl += 256) {  // L < 16: a chunk can span several reads
      const uint32_t lb = cl << 4;
      const uint32_t na = (L - k0) < 16u ? (L - k0) : 16u;  // bytes taken from read r_base + rl
      uint32_t codes, exc;
      gather_piece(planes(rl), PieceSrc{r_pos[rl], rl & 1u}, k0, codes, exc);
      if (na < 16u) {
        uint32_t filled = na, r = rl + 1;
        codes &= (1u << (2 * filled)) - 1u;
        exc &= (1u << filled) - 1u;
        while (filled < 16u && r < n_in) {
          uint32_t c2, e2;
          gather_piece(planes(r), PieceSrc{r_pos[r], r & 1u}, 0, c2, e2);
          const uint32_t take = (16u - filled) < L ? (16u - filled) : L;
          if (take < 16u) { c2 &= (1u << (2 * take)) - 1u; e2 &= (1u << take) - 1u; }
          codes |= c2 << (2 * filled);
          exc |= e2 << filled;
          filled += take;
          r++;
        }
      }
      uint4 out;
      out.x = expand4(codes & 0xffu, exc & 0xfu);
      out.y = expand4((codes >> 8) & 0xffu, (exc >> 4) & 0xfu);
      out.z = expand4((codes >> 16) & 0xffu, (exc >> 8) & 0xfu);
      out.w = expand4(codes >> 24, (exc >> 12) & 0xfu);
      const uint64_t byte0 = gbyte0 + lb;
      if (lb + 16u <= gbytes) {
        stream_store(reinterpret_cast<v4u32*>(seq + byte0), (v4u32{out.x, out.y, out.z, out.w}));
        stream_store(reinterpret_cast<v4u32*>(qual + byte0), (v4u32{qv.x, qv.y, qv.z, qv.w}));
      } else {  // last partial chunk of the shard
        const uint32_t words[4] = {out.x, out.y, out.z, out.w};
        for (uint32_t i = 0; lb + i < gbytes; i++) {
          seq[byte0 + i] = (uint8_t)(words[i >> 2] >> (8 * (i & 3)));
          qual[byte0 + i] = (uint8_t)qual_byte;
        }
      }
      rl += step_r;
      k0 += step_k;
      if (k0 >= L) { k0 -= L; rl++; }
    }
  }
  // perfect-short has no per-base draws: every counter follows from the plan
  if (blockIdx.x == 0 && threadIdx.x == 0 && counters) {
    const uint64_t bases = n_reads * L;
    shard_add(counters, SIMMR_CNT_READS, (unsigned long long)n_reads);
    shard_add(counters, SIMMR_CNT_BASES, (unsigned long long)bases);
    shard_add(counters, SIMMR_CNT_QUAL_SUM, (unsigned long long)(bases * 60u));  // perfect_short.rs:42-44
    if (!G.has_exc) shard_add(counters, SIMMR_CNT_ACGT_BASES, (unsigned long long)bases);
  }
}

// (section 7 — the first-generation wave-per-unit kernel k_emit_stream — lost its A/B in round 1 and left the tree in round 5;
// git history keeps it.  What the lane-per-read kernel still uses of it:)

// Phred value of one accepted standard-normal draw.
SIMMR_DEV uint32_t phred_of_z(const ProfileDev& prof, const Tables* __restrict__ T, double x) {
  const float z = (float)x;  // StandardNormal for f32 = f64 sample cast down
  if (prof.kind == SIMMR_K_PERFECT_LONG) {
    // perfect_long.rs:68-77: acc = N(0.99, 0.05).min(0.9999); round(-10 log10(1 - acc)).
    // The log is replaced by the host-libm-derived threshold table on (1 - acc).
    float acc = __fadd_rn(prof.pl_mean, __fmul_rn(0.05f, z));
    acc = fminf(acc, 0.9999f);
    const float d = __fsub_rn(1.0f, acc);
    uint32_t q = T->pl_first;
    for (uint32_t i = 0; i < T->pl_count; i++) q += (d <= T->pl_thresh[i]) ? 1u : 0u;
    return q;
  }
  // minimal_short.rs:90-101 / minimal_long.rs:88-98: floor(mean + 10 z) as u8
  return sat_u8_f32(floorf(__fadd_rn(prof.mean_phred_f, __fmul_rn(10.0f, z))));
}


// ===========================================================================
// 8. Emit, lane-per-read form (the fast path for short reads)
//
// Every LANE owns one read (mate) and walks its own StdRng streams exactly as
// the reference does, one u64 (Phred) or one u32 (mutation) per step:
//   phase Q: ChaCha12 block b of the Phred stream in registers, 8 ziggurat
//            steps, qualities stored 8 bytes at a time;
//   phase M: ChaCha12 block b of the mutation stream, 16 substitution steps,
//            bases stored 8 bytes at a time (mate 2 complement-reversed).
// All 64 lanes compute ChaCha blocks of 64 different streams at once (100 %
// lane use, no LDS traffic, no cross-lane dependency); the block loop is
// wave-uniform because every step consumes exactly one slot.  Reads are
// processed in order of length (k_len_* below) so the lanes of a wave finish
// together; the output position of a read does not depend on that order.
// The ziggurat wedge test exp(-x^2/2) is decided by rigorous Taylor bounds
// around the layer's table value and only falls back to exp() inside a
// 1e-13-wide band, so decisions equal the reference's.
// ===========================================================================

#define LBINS 1024u

extern "C" __global__ void __launch_bounds__(256)
k_len_hist(const uint32_t* __restrict__ len, uint64_t n, uint32_t shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[LBINS];
  for (uint32_t i = threadIdx.x; i < LBINS; i += 256) h[i] = 0;
  __syncthreads();
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) {
    uint32_t L = len[k] >> shift;
    atomicAdd(&h[L < LBINS ? L : LBINS - 1], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < LBINS; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// single workgroup: hist -> exclusive prefix (cursor)
extern "C" __global__ void __launch_bounds__(256)
k_len_scan(uint32_t* __restrict__ hist_cursor) {
  __shared__ uint64_t lds4[4];
  uint32_t v[4], s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { v[i] = hist_cursor[threadIdx.x * 4 + i]; s += v[i]; }
  uint64_t tot;
  uint32_t ex = (uint32_t)wg_exclusive_scan_u64(s, lds4, &tot);
#pragma unroll
  for (int i = 0; i < 4; i++) { hist_cursor[threadIdx.x * 4 + i] = ex; ex += v[i]; }
}

extern "C" __global__ void __launch_bounds__(256)
k_len_scatter(const uint32_t* __restrict__ len, uint64_t n, uint32_t shift, uint32_t* __restrict__ cursor,
              uint32_t* __restrict__ order) {
  __shared__ uint32_t h[LBINS];
  __shared__ uint32_t base[LBINS];
  const uint64_t per_wg = 256ull * 16ull;
  for (uint64_t start = (uint64_t)blockIdx.x * per_wg; start < n; start += (uint64_t)gridDim.x * per_wg) {
    for (uint32_t i = threadIdx.x; i < LBINS; i += 256) h[i] = 0;
    __syncthreads();
    uint32_t bin[16], rank[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      uint64_t k = start + (uint64_t)i * 256 + threadIdx.x;
      bin[i] = 0xffffffffu;
      if (k < n) {
        uint32_t L = len[k] >> shift;
        bin[i] = L < LBINS ? L : LBINS - 1;
        rank[i] = atomicAdd(&h[bin[i]], 1u);
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LBINS; i += 256)
      if (h[i]) base[i] = atomicAdd(&cursor[i], h[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      uint64_t k = start + (uint64_t)i * 256 + threadIdx.x;
      if (bin[i] != 0xffffffffu) order[base[bin[i]] + rank[i]] = (uint32_t)k;
    }
    __syncthreads();
  }
}


SIMMR_DEV void store_bytes(uint8_t* __restrict__ p, uint64_t v, uint32_t n) {
  if (n == 8) {
    *reinterpret_cast<u64_unaligned*>(p) = v;
  } else {
    for (uint32_t i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * i));
  }
}

SIMMR_DEV uint32_t cvt_u32_f32_sat(float f) {
  uint32_t r;
  asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
SIMMR_DEV uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

struct LaneQ {
  uint32_t i;      // next base index
  uint32_t st;     // 0 FRESH, 1 WEDGE (need the f64 draw), 2 TAIL_X, 3 TAIL_Y
  uint32_t pidx;   // layer of the pending wedge
  double px;       // pending x (WEDGE) / u (TAIL) 
  double tx;       // TAIL: x_ draw waiting for its y_
  uint32_t qsum;   // <= 65535 * 255
};

// Per-lane output staging: a 128-byte ring in LDS holding the two 64-byte
// global sectors the lane is currently filling.  Sectors are flushed whole
// (global_store_dwordx4 x4 on a 64-byte aligned address) once per ChaCha
// block, so HBM sees full-sector writes instead of scattered 8-byte pieces.
#define RING_PITCH 136u
struct OutRing {
  uint8_t* ring;    // LDS row of this lane
  uint8_t* g;       // global address of byte 0 of this read
  uint32_t a0;      // (address of byte 0) & 127
  uint32_t L;
  uint32_t mark;    // forward: bytes [0, mark) are flushed; reverse: bytes [mark, L) are flushed
};

SIMMR_DEV void ring_init(OutRing& r, uint8_t* lds_row, uint8_t* g, uint32_t L, bool descending) {
  r.ring = lds_row; r.g = g; r.L = L;
  r.a0 = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 127u);
  r.mark = descending ? L : 0u;
}
SIMMR_DEV void ring_put(const OutRing& r, uint32_t k, uint32_t byte) { r.ring[(r.a0 + k) & 127u] = (uint8_t)byte; }

// copy bytes [lo, hi) of the read (all inside one 64-byte sector) ring -> global
SIMMR_DEV void ring_copy(const OutRing& r, uint32_t lo, uint32_t hi) {
  const uint32_t p = r.a0 + lo;
  if (hi - lo == 64u && (p & 63u) == 0u) {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(r.ring + (p & 127u));
    uint4* dst = reinterpret_cast<uint4*>(r.g + lo);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t a = src[2 * c], b = src[2 * c + 1];
      dst[c] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
  } else {
    uint32_t k = lo;
    while (k < hi) {
      const uint32_t q = r.a0 + k;
      if ((q & 7u) == 0u && k + 8u <= hi) {
        *reinterpret_cast<uint64_t*>(r.g + k) = *reinterpret_cast<const uint64_t*>(r.ring + (q & 127u));
        k += 8;
      } else {
        r.g[k] = r.ring[q & 127u];
        k++;
      }
    }
  }
}
// forward stream: `done` bytes [0, done) are in the ring or flushed
SIMMR_DEV void ring_flush_fwd(OutRing& r, uint32_t done) {
  while (r.mark < done) {
    const uint32_t sector_end = (((r.a0 + r.mark) >> 6) + 1u) * 64u - r.a0;
    const uint32_t kend = sector_end < r.L ? sector_end : r.L;
    if (done < kend) break;
    ring_copy(r, r.mark, kend);
    r.mark = kend;
  }
}
// descending stream: bytes [low, L) are in the ring or flushed
SIMMR_DEV void ring_flush_rev(OutRing& r, uint32_t low) {
  while (r.mark > 0u) {
    const int32_t sector_start = (int32_t)(((r.a0 + r.mark - 1u) >> 6) * 64u) - (int32_t)r.a0;
    const uint32_t ks = sector_start > 0 ? (uint32_t)sector_start : 0u;
    if (low > ks) break;
    ring_copy(r, ks, r.mark);
    r.mark = ks;
  }
}

// rare: exact wedge decision and the tail loop (rand_distr zero_case)
__device__ __attribute__((noinline)) bool wedge_exact(double t, double x) {
  return t < exp(__dmul_rn(__dmul_rn(-x, x), 0.5));
}
// Decision of the ziggurat wedge test  f1 + (f0 - f1) r < exp(-x^2/2)  in f64:
// Taylor bounds of pdf(x) = F[i] e^d, d = (X[i]^2 - x^2)/2 in [0, 0.73], with a
// 1e-13 guard band, exp() only inside the band.
__device__ __attribute__((noinline)) bool wedge_f64(const Tables* __restrict__ T, uint32_t zi, double x,
                                                    uint64_t bits) {
  const double r = (double)(bits >> 11) * (1.0 / 9007199254740992.0);
  const double f0 = T->zig_f[zi], f1 = T->zig_f[zi + 1];
  const double t = __dadd_rn(f1, __dmul_rn(__dsub_rn(f0, f1), r));
  const double Xi = T->zig_x[zi], ax = fabs(x);
  const double d = (Xi - ax) * (Xi + ax) * 0.5;
  const double d2 = d * d;
  const double lo = f0 * (1.0 + d + 0.5 * d2 + d2 * d * (1.0 / 6.0));
  const double hi = lo + f0 * (0.06 * d2 * d2);
  if (t < lo * (1.0 - 1e-13)) return true;
  if (t > hi * (1.0 + 1e-13)) return false;
  return wedge_exact(t, x);
}

__device__ __attribute__((noinline)) bool tail_try(double x_, double y_, double u, double* out) {
  const double xx = log(x_) / SIMMR_ZIG_R;
  const double yy = log(y_);
  if (__dmul_rn(-2.0, yy) < __dmul_rn(xx, xx)) return false;  // loop again
  *out = u < 0.0 ? xx - SIMMR_ZIG_R : SIMMR_ZIG_R - xx;
  return true;
}

template <bool PL>
SIMMR_DEV void q_emit(LaneQ& s, double x, const ProfileDev& prof, const Tables* __restrict__ T,
                      const OutRing& ring, uint32_t qoff) {
  // minimal_short.rs:90-101 / minimal_long.rs:88-98: floor(Normal<f32>(mean, 10).sample()) as u8;
  // perfect_long.rs:68-77 through the threshold table (phred_of_z)
  // floor(..) as u8 (saturating, NaN -> 0): v_cvt_u32_f32 truncates, sends NaN and negatives to 0 and saturates
  // upwards; below zero floor and truncation differ, but both end at 0, so the floor needs no instruction
  const uint32_t q = PL ? phred_of_z(prof, T, x)
                        : min_u32(cvt_u32_f32_sat(__fadd_rn(prof.mean_phred_f, __fmul_rn(10.0f, (float)x))), 255u);
  s.qsum += q;
  ring_put(ring, s.i, q + qoff);
  s.i++;
}

template <bool PL>
SIMMR_DEV void q_step(LaneQ& s, uint64_t bits, const double2* __restrict__ zx2,
                      const float2* __restrict__ zf2f, const Tables* __restrict__ T,
                      const ProfileDev& prof, const OutRing& ring, uint32_t qoff) {
  if (s.st == 0) {
    const uint32_t zi = (uint32_t)bits & 0xffu;
    const double u = __longlong_as_double((long long)((bits >> 12) | 0x4000000000000000ULL)) - 3.0;
    const double2 X = zx2[zi];
    const double x = __dmul_rn(u, X.x);
    if (fabs(x) < X.y) {
      q_emit<PL>(s, x, prof, T, ring, qoff);
    } else if (zi == 0) {
      s.st = 2; s.px = u;
    } else {
      s.st = 1; s.px = x; s.pidx = zi;
    }
  } else if (s.st == 1) {
    // f_tab[i+1] + (f_tab[i] - f_tab[i+1]) * gen::<f64>() < pdf(x):
    // an f32 estimate settles it unless the two sides are within 1e-4 (the
    // f32 error is < 1e-5); only then the f64 decision runs.
    const float2 Ff = zf2f[s.pidx];  // {F[i], F[i+1]} rounded to f32
    const float rf = (float)(uint32_t)(bits >> 40) * (1.0f / 16777216.0f);
    const float tf = Ff.y + (Ff.x - Ff.y) * rf;
    const float xf = (float)s.px;
    const float pf = __builtin_amdgcn_exp2f(xf * xf * -0.72134752044448170f);  // exp(-x^2/2)
    bool accept;
    if (tf < pf * (1.0f - 1e-4f)) accept = true;
    else if (tf > pf * (1.0f + 1e-4f)) accept = false;
    else accept = wedge_f64(T, s.pidx, s.px, bits);
    s.st = 0;
    if (accept) q_emit<PL>(s, s.px, prof, T, ring, qoff);
  } else if (s.st == 2) {
    s.tx = __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ULL)) - (1.0 - 2.220446049250313e-16 / 2.0);
    s.st = 3;
  } else {
    const double y_ = __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ULL)) - (1.0 - 2.220446049250313e-16 / 2.0);
    double x;
    if (tail_try(s.tx, y_, s.px, &x)) { s.st = 0; q_emit<PL>(s, x, prof, T, ring, qoff); }
    else s.st = 2;
  }
}

struct LaneM {
  uint32_t i;       // next base index (forward-strand slice order)
  uint32_t st;      // 0 FRESH, 1 CHOOSE (base i mutates, waiting for an accepted u32)
  uint32_t creg;    // 16 reference codes of bases (i & ~15) ..
  uint32_t ereg;    // their exception bits
  uint64_t qreg, qreg2;    // 16 qualities of bases (i & ~15) ..
  uint64_t qnext, qnext2;  // prefetched: the following 16 qualities
  uint32_t cnext, enext;  // prefetched: the following 16 codes / exception bits
  uint32_t n_subst, n_acgt;
  uint32_t lut;     // "ACGT", or "TGCA" for mate 2 (complemented; written back to front)
  uint32_t wpos;    // ring position of the next output byte (mod 128); moves by wdir
  uint32_t wdir;    // +1, or -1 for mate 2 (simulate.rs:283: reverse complement AFTER mutation)
};

template <bool HAS_EXC>
SIMMR_DEV void m_emit(LaneM& s, uint32_t code, const OutRing& ring) {
  // code 0-3 = ACGT, 4 = 'N', 5 = '-' (only with an exception plane)
  uint32_t ch = __builtin_amdgcn_perm(0u, s.lut, code | 0x0c0c0c00u);  // byte `code` of the lut
  if (HAS_EXC && code >= 4u) ch = code == 4u ? 'N' : '-';
  ring.ring[s.wpos & 127u] = (uint8_t)ch;
  s.wpos += s.wdir;
  s.i++;
}

__device__ __attribute__((noinline)) uint64_t load_q_tail(const uint8_t* __restrict__ qsrc, uint32_t i, uint32_t L) {
  uint64_t v = 0;
  for (uint32_t b = 0; i + b < L; b++) v |= (uint64_t)qsrc[i + b] << (8 * b);
  return v;
}

SIMMR_DEV uint64_t load_q8(const uint8_t* __restrict__ qsrc, uint32_t i, uint32_t L) {
  if (i + 8 <= L) return *reinterpret_cast<const u64_unaligned*>(qsrc + i);
  if (i >= L) return 0;
  if (L >= 8) return *reinterpret_cast<const u64_unaligned*>(qsrc + L - 8) >> (8 * (i + 8 - L));
  return load_q_tail(qsrc, i, L);
}

// 16 qualities from byte i on: one 16-byte load (global_load_dwordx4 at any byte address) away from the read's end
SIMMR_DEV void load_q16(const uint8_t* __restrict__ qsrc, uint32_t i, uint32_t L, uint64_t& lo, uint64_t& hi) {
  if (i + 16u <= L) {
    const v4u32 v = *reinterpret_cast<const v4u32_unaligned*>(qsrc + i);
    lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
    hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
  } else {
    lo = load_q8(qsrc, i, L);
    hi = load_q8(qsrc, i + 8u, L);
  }
}

template <bool HAS_EXC>
SIMMR_DEV void m_step(LaneM& s, uint32_t w, uint32_t L, uint32_t rev, const uint32_t* __restrict__ packed,
                      const uint32_t* __restrict__ mask, uint64_t src,
                      const uint32_t* __restrict__ thr, const uint8_t* __restrict__ qsrc, uint32_t qoff,
                      const OutRing& ring) {
  // One word, read both ways by every lane — as the f32 test of base i (state 0) and as a draw of choose() for the
  // base that is waiting (state 1) — and the state picks: some lane of a wave is in state 1 in two steps of three, so
  // the two readings as separate branches cost more than both of them straight.
  const uint32_t i = s.i;
  const bool choosing = s.st != 0u;
  // registers are refilled one chunk ahead so the loads overlap the steps
  if (!choosing && (i & 15u) == 0) {
    s.qreg = s.qnext; s.qreg2 = s.qnext2;
    load_q16(qsrc, i + 16, L, s.qnext, s.qnext2);
    s.creg = s.cnext;
    s.cnext = fetch_codes16(packed, (int64_t)(src + i + 16));
    if (HAS_EXC) { s.ereg = s.enext; s.enext = mask ? fetch_mask16(mask, (int64_t)(src + i + 16)) : 0u; }
  }
  // byte (i & 7) of qreg, zero-extended, in one v_perm_b32
  const uint64_t qh = (i & 8u) ? s.qreg2 : s.qreg;
  const uint32_t q = (__builtin_amdgcn_perm((uint32_t)(qh >> 32), (uint32_t)qh, (i & 7u) | 0x0c0c0c00u) - qoff) & 0xffu;
  const uint32_t c = (s.creg >> (2 * (i & 15u))) & 3u;
  uint32_t code = c;
  uint32_t exc = 0;
  if (HAS_EXC) {
    exc = (s.ereg >> (i & 15u)) & 1u;
    if (exc) code = 4u + (c & 1u);
    s.n_acgt += (exc || choosing) ? 0u : 1u;
  }
  // state 0: gen::<f32>() > accuracy(q)  <=>  (w >> 8) > floor(acc * 2^24)
  const bool mutate = (w >> 8) > thr[q] && !exc;
  // state 1: choose(&[3 alternatives]) = gen_range(0..3u32), zone 0xBFFFFFFF
  const uint64_t m = (uint64_t)w * 3u;
  const bool drawn = (uint32_t)m <= 0xBFFFFFFFu;
  const uint32_t k = (uint32_t)(m >> 32);
  const uint32_t alt = k + (k >= c ? 1u : 0u);
  const bool emit = choosing ? drawn : !mutate;
  s.n_subst += (choosing && drawn) ? 1u : 0u;
  s.st = choosing ? (drawn ? 0u : 1u) : (mutate ? 1u : 0u);
  if (emit) m_emit<HAS_EXC>(s, choosing ? alt : code, ring);
}

#define LANES_WG 512
template <bool HAS_EXC, bool PAIRED, bool PL>
__global__ void __launch_bounds__(LANES_WG)
k_emit_lanes(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint32_t genome, uint64_t n_units,
             const uint32_t* __restrict__ order, PlanArrays pl, const uint64_t* __restrict__ u_off,
             const uint32_t* __restrict__ u_contig, const uint32_t* __restrict__ u_genome,
             const uint64_t* __restrict__ u_seed,
             uint8_t* __restrict__ seq, uint8_t* __restrict__ qual, uint32_t qual_offset,
             const Tables* __restrict__ T, unsigned long long* __restrict__ counters) {
  __shared__ double2 zx2[256];
  __shared__ float2 zf2f[256];
  __shared__ uint32_t thr[256];
  __shared__ __attribute__((aligned(16))) uint8_t rings[LANES_WG * RING_PITCH];
  if (threadIdx.x < 256) {
    const uint32_t t = threadIdx.x;
    zx2[t] = make_double2(T->zig_x[t], T->zig_x[t + 1]);
    zf2f[t] = make_float2((float)T->zig_f[t], (float)T->zig_f[t + 1]);
    thr[t] = (uint32_t)floorf(T->acc[t] * 16777216.0f);
  }
  __syncthreads();
  uint8_t* my_ring = rings + threadIdx.x * RING_PITCH;
  const uint64_t n_tasks = PAIRED ? 2 * n_units : n_units;
  uint64_t qsum_tot = 0;
  uint32_t subst_tot = 0, acgt_tot = 0;
  const uint64_t stride = (uint64_t)gridDim.x * LANES_WG;
  for (uint64_t task0 = (uint64_t)blockIdx.x * LANES_WG + (threadIdx.x & ~63u); task0 < n_tasks; task0 += stride) {
    const uint64_t task = task0 + (threadIdx.x & 63u);
    const bool live = task < n_tasks;
    uint32_t L = 0, rev = 0;
    uint64_t seed_q = 0, seed_m = 0, src = 0, off = 0;
    const uint32_t* packed = nullptr;
    const uint32_t* mask = nullptr;
    if (live) {
      const uint64_t ti = PAIRED ? (task >> 1) : task;
      // longest first: the tail of the grid-stride loop then holds the shortest reads
      const uint64_t u = order ? (uint64_t)order[n_units - 1 - ti] : ti;
      rev = PAIRED ? (uint32_t)(task & 1u) : 0u;
      L = pl.len[u];
      off = u_off[u] + (rev ? L : 0u);
      const GenomeDev G = genomes[u_genome ? u_genome[u] : genome];
      packed = G.packed;
      mask = G.mask;
      src = G.contigs[u_contig[u]].base + (rev ? pl.b[u] : pl.a[u]);
      seed_q = rev ? pl.qs2[u] : u_seed[u];  // long reads re-seed everything with read_seed (simulate.rs:497-503)
      seed_m = rev ? pl.ms2[u] : u_seed[u];
    }
    // ---- phase Q: simulate_phred_scores (minimal_short.rs:83-102)
    {
      const Key key = pcg32_expand(seed_q);
      LaneQ s;
      s.i = 0; s.st = 0; s.pidx = 0; s.px = 0.0; s.tx = 0.0; s.qsum = 0;
      OutRing ring;
      ring_init(ring, my_ring, qual + off, L, false);
      for (uint32_t blk = 0; __any(s.i < L); blk++) {
        uint32_t w[16];
        chacha12_block(key, blk, w);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (s.i < L) q_step<PL>(s, ((uint64_t)w[2 * j + 1] << 32) | w[2 * j], zx2, zf2f, T, prof, ring, qual_offset);
        }
        // Sectors are flushed by the whole wave at the same blocks (every 8th: 64 slots, about one sector per lane),
        // not by each lane at the block in which its own sector fills: flushing every block ran the flush code with
        // an eighth of the lanes.  A lane holds < 64 unflushed bytes after a flush and adds <= 64 until the next: the
        // 128-byte ring never wraps onto them.  A finished lane flushes its last, partial sector at once.
        if ((blk & 7u) == 7u || s.i >= L) ring_flush_fwd(ring, s.i);
      }
      qsum_tot += s.qsum;
    }
    // ---- phase M: simulate_point_mutations (minimal_short.rs:104-140) + output
    {
      const Key key = pcg32_expand(seed_m);
      LaneM s;
      s.i = 0; s.st = 0; s.creg = 0; s.ereg = 0; s.qreg = 0; s.qreg2 = 0; s.n_subst = 0; s.n_acgt = 0;
      const uint8_t* qsrc = qual + off;
      OutRing ring;
      ring_init(ring, my_ring, seq + off, L, rev != 0);
      s.lut = rev ? 0x41434754u : 0x54474341u;  // "TGCA" / "ACGT"
      s.wpos = ring.a0 + (rev ? L - 1u : 0u);   // forward mate: byte i; mate 2: byte L-1-i
      s.wdir = rev ? 0xffffffffu : 1u;
      load_q16(qsrc, 0, L, s.qnext, s.qnext2);
      s.cnext = live ? fetch_codes16(packed, (int64_t)src) : 0u;
      s.enext = (HAS_EXC && mask) ? fetch_mask16(mask, (int64_t)src) : 0u;
      for (uint32_t blk = 0; __any(s.i < L); blk++) {
        uint32_t w[16];
        chacha12_block(key, blk, w);
#pragma unroll
        for (int j = 0; j < 16; j++)
          if (s.i < L) m_step<HAS_EXC>(s, w[j], L, rev, packed, mask, src, thr, qsrc, qual_offset, ring);
        if ((blk & 3u) == 3u || s.i >= L) {  // every 4th block: 64 words, about one sector per lane (as above)
          if (rev) ring_flush_rev(ring, L - s.i); else ring_flush_fwd(ring, s.i);
        }
      }
      subst_tot += s.n_subst;
      acgt_tot += HAS_EXC ? s.n_acgt : L;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    subst_tot += __shfl_down(subst_tot, d, 64);
    acgt_tot += __shfl_down(acgt_tot, d, 64);
    qsum_tot += __shfl_down(qsum_tot, d, 64);
  }
  if ((threadIdx.x & 63u) == 0 && counters) {
    shard_add(counters, SIMMR_CNT_SUBSTITUTIONS, (unsigned long long)subst_tot);
    shard_add(counters, SIMMR_CNT_ACGT_BASES, (unsigned long long)acgt_tot);
    shard_add(counters, SIMMR_CNT_QUAL_SUM, (unsigned long long)qsum_tot);
  }
}

// ===========================================================================
// 9. Emit, counter mode (SIMMR_RNG_PHILOX — the design BASELINE.json's north_star
//    prescribes for the per-base draws; statistical parity).
//
// Philox4x32-10 keyed by the read's Phred seed.  A base draws its Phred score and
// its substitution together from their joint law over the 1024 outcomes (q, s):
// s = 0 no substitution, s = 1..3 the base becomes "ACGT"[(code + s) & 3].  The
// draw has two levels (DESIGN.md section 4 states the specification): 24 bits per base
// settle it through a 1024-column alias table in all but E of 2^24 cells (E ~ 120),
// and a base that lands on one of those escape cells takes one full word of
// another Philox call against the residual law.  Three calls serve 16 bases, and
// since no draw depends on another the work item is "16 consecutive bases of one
// read".  An escape is rare enough (one item in 9 000) to be repaired where it is
// found.
//
// A workgroup takes 128 units (256 mates, or 128 long reads), writes one 32-byte
// record per read to LDS (key, output offset, source word address, length) together
// with the prefix of their item counts, and deals the items to its 256 lanes in
// read order: consecutive lanes hold consecutive groups, so a wave's 16-byte quality
// and base stores are contiguous and every 64-byte line of the output is written
// in one go.  (Measured in round 2: writing the partial group at the end of each
// read in a pass of its own makes the kernel's write traffic 45 GB instead of 33 GB
// per 100 M reads — the lines are written twice, from two store instructions that
// are too far apart to be merged in L2 — and the kernel HBM-bound at the time it
// has now.)
// Instruction choices follow profiles/microbench/valu_asm_rates*: v_mad_u64_u32
// (2.1 x a v_add_u32) gives both halves of a Philox product, v_bitop3_b32 the
// three-way xor of a round; one 8-byte LDS entry per column holds the word the
// draw is compared with (the column's index is its top ten bits, as in the draw,
// so no field is extracted first) and the two results as 16-bit halves;
// v_alignbit_b32 shifts s into the packed substitution word, an SDWA move drops
// the quality byte into place; the substitutions are one SWAR add modulo 4 on the
// packed 2-bit codes; blocks of short reads keep an item -> read map in LDS (one
// byte load), otherwise a branch-free binary search over the item prefix finds the
// read.
// Named -D switches compile pieces out for differential timing (Makefile, `make ablate`):
// SIMMR_ABLATE_PHILOX (words from one multiply instead of the ten rounds), SIMMR_ABLATE_LOOKUP
// (no table lookups), SIMMR_ABLATE_STORES (no global stores of bases / qualities), SIMMR_ABLATE_CODES
// (no load from the reference plane), SIMMR_ABLATE_META (no metadata columns), SIMMR_ABLATE_ITEMS (one
// round of items per block), SIMMR_ABLATE_ALL16 / SIMMR_ABLATE_ALIGN16 (partial groups stored as
// 16 bytes, except the shard's last ones / every store aligned down to 16 bytes: wrong bytes, inside the buffers), SIMMR_ABLATE_NOP
// (without the wait states between the compare and the select), SIMMR_ABLATE_HOTSTORE (the same store
// instructions, all landing in the first 64 KB of the two streams: store issue without the DRAM write
// path), SIMMR_ABLATE_LINES (every store instruction of a wave writes sixteen whole 64-byte lines, clamped to the
// streams' total_bases).  None of them changes an index, a pointer into a table or a loop bound; the two that
// move stores are bounded by the planned size (tests/test_gpu_shapes.py runs them between canaries when built).
// ===========================================================================
#include "fastq_format.hpp"  // (inside namespace simmr) header formatting, for the TEXT form of k_emit_philox

// (xor3, philox4x32_10: rng_device.hpp — the plan kernels of SIMMR_RNG_PHILOX_FULL draw from it too)

#define PHILOX_UNITS 128u
#define PHILOX_READS 256u  /* 128 pairs x 2 mates */
#define PHILOX_MAP_ITEMS 4096u
#define PHILOX_CBASE 64u /* contig bases kept in LDS by the CACHED kernels */
#define FQ_GROUP 128u /* TEXT: headers formatted at a time (LDS slots; they share their memory with the item map, see `owner`) */

// bytes a + b with per-byte wrap-around (u8 add of util.rs:46-50)
SIMMR_DEV uint32_t add_bytes(uint32_t a, uint32_t b) {
  return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
// keep the low `nb` (0..4) bytes of x
SIMMR_DEV uint32_t low_bytes(uint32_t x, int nb) {
  return nb >= 4 ? x : (nb <= 0 ? 0u : (x & ((1u << (8 * nb)) - 1u)));
}
// one 16-byte store at any byte address (global_store_dwordx4; one address-unit access per lane instead of two)
template <bool NT = false>
SIMMR_DEV void store16(uint8_t* __restrict__ d, uint64_t lo, uint64_t hi) {
  v4u32 v;
  v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
  if (NT) stream_store(reinterpret_cast<v4u32_unaligned*>(d), (v4u32_unaligned)v);
  else *reinterpret_cast<v4u32_unaligned*>(d) = v;
}
// the slot layout's 16-byte stores: nontemporal (other cache-policy bits lose to `nt`: LAB.md, round 3)
SIMMR_DEV void slot_store16(uint8_t* __restrict__ base, uint32_t off, uint64_t lo, uint64_t hi) { store16<true>(base + off, lo, hi); }
// store the low n (< 16) bytes of the 128-bit value (lo, hi)
SIMMR_DEV void store_tail(uint8_t* __restrict__ d, uint64_t lo, uint64_t hi, uint32_t n) {
  uint64_t v = lo;
  uint32_t p = 0;
  typedef uint32_t __attribute__((aligned(1))) u32_un;
  typedef uint16_t __attribute__((aligned(1))) u16_un;
  if (n & 8u) { *reinterpret_cast<u64_unaligned*>(d) = (u64_unaligned)lo; v = hi; p = 8; }
  if (n & 4u) { *reinterpret_cast<u32_un*>(d + p) = (u32_un)(uint32_t)v; v >>= 32; p += 4; }
  if (n & 2u) { *reinterpret_cast<u16_un*>(d + p) = (u16_un)(uint16_t)v; v >>= 16; p += 2; }
  if (n & 1u) d[p] = (uint8_t)v;
}

// the low n (< 16) bytes of two 128-bit values to two places: one branch per bit of n for both (half the exec-mask
// bookkeeping of two store_tail calls; the same stores)
SIMMR_DEV void store_tail2(uint8_t* __restrict__ d0, uint64_t lo0, uint64_t hi0, uint8_t* __restrict__ d1, uint64_t lo1, uint64_t hi1,
                            uint32_t n) {
  uint64_t v0 = lo0, v1 = lo1;
  uint32_t p = 0;
  typedef uint32_t __attribute__((aligned(1))) u32_un;
  typedef uint16_t __attribute__((aligned(1))) u16_un;
  if (n & 8u) { *reinterpret_cast<u64_unaligned*>(d0) = (u64_unaligned)lo0; *reinterpret_cast<u64_unaligned*>(d1) = (u64_unaligned)lo1; v0 = hi0; v1 = hi1; p = 8; }
  if (n & 4u) {
    *reinterpret_cast<u32_un*>(d0 + p) = (u32_un)(uint32_t)v0;
    *reinterpret_cast<u32_un*>(d1 + p) = (u32_un)(uint32_t)v1;
    v0 >>= 32; v1 >>= 32; p += 4;
  }
  if (n & 2u) {
    *reinterpret_cast<u16_un*>(d0 + p) = (u16_un)(uint16_t)v0;
    *reinterpret_cast<u16_un*>(d1 + p) = (u16_un)(uint16_t)v1;
    v0 >>= 16; v1 >>= 16; p += 2;
  }
  if (n & 1u) { d0[p] = (uint8_t)v0; d1[p] = (uint8_t)v1; }
}

// Per-read record of a block: 32 bytes, read with two ds_read_b128.
struct alignas(16) PhRec {
  uint32_t k0, k1;  // Philox key = the read's Phred seed
  uint32_t dst;     // first output byte of the read, relative to the block's first output byte
  uint32_t lw;      // L | (2 * (source position & 15)) << 16 | rev << 31   (L <= 65535: u16 lengths)
  uint64_t wa;      // address of the 2-bit plane word that holds the read's first source base
  uint32_t gs;      // first item of the read among the block's items
  uint32_t pad;
};

// One base: R holds the 24-bit draw F in its upper 24 bits (the low byte is whatever the bit string has there).
// Level-1 LDS entry of column c = F >> 14 (T 16384ths answer A, the rest B; results r = enc(q) << 8 | esc << 2 | s):
//   x = c << 22 | T << 8:  R < x  <=>  F & 0x3fff < T, because the top ten bits are equal on both sides and x has
//       zeros in the low byte; T = 16384 is stored as T = 0 with B = A
//   y = r(A) | r(B) << 16
// v_cndmask_b32_sdwa picks the half in place.  A VALU write of VCC needs two wait states before a VALU reads it as
// a mask on gfx950 (the compiler pads its own code; inside an asm statement that is this statement's job).
SIMMR_DEV uint32_t philox_pick(uint32_t R, const uint2* __restrict__ jtab) {
#if defined(SIMMR_ABLATE_LOOKUP)
  return ((R >> 8) & 3u) | ((R >> 10) & 0x3f00u) | 0x2100u;
#else
  const uint2 e = jtab[R >> 22];
  uint32_t x;
#if defined(SIMMR_ABLATE_NOP)
  asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\t"
#else
  asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\t"
      "s_nop 1\n\t"
#endif
      "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"
      : "=v"(x) : "v"(R), "v"(e.x), "v"(e.y) : "vcc");
  return x;
#endif
}

// An item in which a base escaped (one item in 9 000): its 16 draws again, four bases at a time, the escaped ones
// from the residual law (counter (b >> 2, 1), word b & 3).  A loop, not unrolled: it must not cost the item loop
// registers.  t1 = the level-1 table as uploaded (T | A << 16, then B), t2 = the level-2 table.
SIMMR_DEV void philox_repair(const uint32_t k0, const uint32_t k1, const uint32_t ci, const uint32_t* __restrict__ t1,
                             const uint32_t* __restrict__ t2, const uint32_t qoff, uint32_t& ss, uint32_t qr[4]) {
  ss = 0u;
#pragma nounroll
  for (uint32_t g4 = 0; g4 < 4u; g4++) {
    // words 3 g4 .. 3 g4 + 2 of the group's twelve: calls (3 g4) >> 2 and (3 g4 + 2) >> 2
    uint32_t wa[4], wb[4];
    philox4x32_10(3u * ci + ((3u * g4) >> 2), 0u, k0, k1, wa);
    philox4x32_10(3u * ci + ((3u * g4 + 2u) >> 2), 0u, k0, k1, wb);
    // g4 = 0: (a0 a1 a2), 1: (a3 b0 b1), 2: (a2 a3 b0), 3: (b1 b2 b3)  [a = first call above, b = second]
    const uint32_t w0 = g4 == 0u ? wa[0] : g4 == 1u ? wa[3] : g4 == 2u ? wa[2] : wb[1];
    const uint32_t w1 = g4 == 0u ? wa[1] : g4 == 1u ? wb[0] : g4 == 2u ? wa[3] : wb[2];
    const uint32_t w2 = g4 == 0u ? wa[2] : g4 == 1u ? wb[1] : g4 == 2u ? wb[0] : wb[3];
    const uint32_t F[4] = {w0 & 0xffffffu, (w0 >> 24) | ((w1 & 0xffffu) << 8), (w1 >> 16) | ((w2 & 0xffu) << 16), w2 >> 8};
    uint32_t w2nd[4];
    philox4x32_10(4u * ci + g4, 1u, k0, k1, w2nd);
    uint32_t q4 = 0u;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const uint32_t col = F[h] >> 14;
      const uint32_t e = t1[col];  // T | A << 16; B in the second half of the table
      uint32_t o = (F[h] & 0x3fffu) < (e & 0xffffu) ? (e >> 16) : t1[1024u + col];
      if (o == PHILOX_ESC) {
        const uint32_t W = w2nd[h];
        const uint32_t e2 = t2[W >> 22];
        o = (W & 0x3fffffu) < (e2 & 0x3fffffu) ? (W >> 22) : (e2 >> 22);
      }
      ss |= (o >> 8) << (2u * (4u * g4 + (uint32_t)h));
      q4 |= (((o & 0xffu) + qoff) & 0xffu) << (8 * h);
    }
    if (g4 == 0u) qr[0] = q4;
    if (g4 == 1u) qr[1] = q4;
    if (g4 == 2u) qr[2] = q4;
    if (g4 == 3u) qr[3] = q4;
  }
}

// COPY_ONLY: the same item machinery without the draws: bases of the planned reads (mate 2 reverse-complemented)
// with coalesced stores, for the profiles whose qualities another kernel writes (custom-short).
// CACHED: every pair comes from one genome (u_genome == null) with at most PHILOX_CBASE contigs, whose bases sit in
// LDS: a record then needs no load that depends on another load's result.
// TEXT: bases and qualities go straight into FASTQ text (simmr_emit_fastq; fastq.rs:58-66): `seq` is the text, read rd's
// record starts where the block's scan of the record lengths puts it (off64[w] = first byte of record 64 w; a record is
// header + 1 + L + 3 + L + 1 bytes with a header of hlen[rd] bytes), so its bases start at record + hlen[rd] + 1 and its
// qualities L + 3 bytes further ("\n+\n"); the lane that holds a read's first qualities also writes those three bytes
// (one 4-byte store that ends in its own first quality).  The headers are written here as well, block by block: after
// the prologue the threads that hold the block's reads format them into LDS slots, 128 at a time, and all 256 copy the
// slots out in 16-byte windows — the run of a read is the '\n' that ends the record before it, its header and the
// header's '\n'.  TEXT implies COARSE (no per-record offsets exist for it).
// ESCQ: see `esc_q` below (the host checks the condition: qual_offset + philox_qmax1 <= 127, level-1 answers only:
// an escaped item is drawn again in full by philox_repair, whatever its level-2 answers are)
// SLOT: 16-byte read slots (SIMMR_SLOT16, include/simmr_hip.h): every read's place in both streams is ceil(L / 16) * 16
// bytes on a 16-byte boundary (u_off is the scan of the padded lengths), qualities and forward bases left-aligned, the
// bases of a reverse-complemented mate right-aligned — so EVERY item, the partial group at a read's end included, is one
// whole aligned 16-byte store per stream (padding written as 0) and the byte ladder of store_tail2 is gone.
// Four waves per SIMD is this kernel's register budget (128 VGPRs): said to the compiler so that no instantiation slips over
// it unnoticed (round 3's TEXT + COARSE form had: 130 VGPRs, three waves); tests/test_resource_guard.py holds every
// instantiation to occupancy 4 without scratch.
#define PHILOX_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4)))
template <bool HAS_EXC, bool COPY_ONLY, bool CACHED, bool TEXT = false, bool ESCQ = false, bool SLOT = false, bool COARSE = false>
__global__ void __launch_bounds__(256) PHILOX_OCCUPANCY
k_emit_philox(ProfileDev prof, uint32_t paired, const GenomeDev* __restrict__ genomes, uint32_t genome_const,
              uint64_t n_units, PlanArrays pl, const uint64_t* __restrict__ u_off,
              const uint32_t* __restrict__ u_contig, const uint32_t* __restrict__ u_genome,
              const uint64_t* __restrict__ u_seed, uint8_t* __restrict__ seq, uint8_t* __restrict__ qual,
              uint32_t qual_offset, uint64_t first_unit, uint32_t read_id_base, OutCols o,
              unsigned long long* __restrict__ counters,
              const uint8_t* __restrict__ hlen = nullptr, const FqTemplate* __restrict__ fq_tp = nullptr, FqTables fq_tb = FqTables{},
              uint32_t fq_lit_bytes = 0, uint32_t fq_hpitch = 0, uint32_t fq_wshift = 0,
              const uint64_t* __restrict__ off64 = nullptr) {
  // off64 != null ("coarse" plans, engine.hip): u_off does not exist; off64[w] = first output byte of pair 64 w (the scan
  // of the plan kernel's per-wave byte sums), and a block places its reads with a scan of their (padded) lengths
  constexpr bool coarse = COARSE && !TEXT;
  // TEXT: off64[w] = first byte of record 64 w, and a block places its records with a scan of their lengths
  // (header + 1 + L + 3 + L + 1, fastq.rs:58-66) before anything else
  static_assert(!TEXT || COARSE, "the TEXT form places its own records");
  // COPY_ONLY with TEXT: perfect-short straight into FASTQ text (perfect_short.rs:42-44: every quality is 60): the copied
  // bases, a constant quality line, headers and counters as in the drawing form
  constexpr bool FULL = !COPY_ONLY || TEXT;  // this launch writes qualities, headers / metadata and all run counters
  const uint32_t const_q4 = (((60u + (qual_offset & 0xffu)) & 0xffu) * 0x01010101u);
#define PL(x) (x)
#define COL_STORE(p, v) do { if (SLOT) stream_store((p), (v)); else *(p) = (v); } while (0)  /* the metadata columns, as the streams */
  // TEXT: header slots of FQ_GROUP reads at a time (dynamic LDS, FQ_GROUP * fq_hpitch bytes), the template and its literals
  extern __shared__ __attribute__((aligned(16))) uint8_t fq_slots[];
  __shared__ __attribute__((aligned(16))) uint8_t fq_lit[TEXT ? FQ_LIT_MAX + 8 : 1];
  __shared__ FqSeg fq_segs[TEXT ? FQ_MAX_SEGS : 1];
  __shared__ uint64_t fq_run_at[TEXT ? FQ_GROUP : 1];
  __shared__ uint32_t fq_run_len[TEXT ? FQ_GROUP : 1];
  uint32_t fq_n_segs = 0;
  if (TEXT) {
    for (uint32_t i = threadIdx.x; i < fq_lit_bytes; i += 256) fq_lit[i] = fq_tb.blob[i];
    fq_n_segs = fq_stage_template(fq_tp, fq_segs);
  }
  __shared__ uint2 jtab[COPY_ONLY ? 1 : 1024];  // level-1 columns (philox_pick)
  __shared__ uint32_t asc[256];  // four 2-bit codes -> four ASCII bytes
  __shared__ PhRec recs[PHILOX_READS];
  __shared__ uint64_t x_src[HAS_EXC ? PHILOX_READS : 1];          // first source base (exception-plane lookups)
  __shared__ const uint32_t* x_mask[HAS_EXC ? PHILOX_READS : 1];  // exception plane of the read's genome or null
  __shared__ uint32_t r_gs[PHILOX_READS + 1];  // first item of each read, ~0 past the last read
  // item -> read, when the block has few enough items.  TEXT: the map lives in the dynamic LDS under the header slots —
  // the slots are dead once the block's headers are copied out (a barrier closes the header phase), the map is dead
  // until then — which is what lets 128 slots and four workgroups per CU fit (engine.hip sizes it: max of the two)
  __shared__ uint8_t owner_static[TEXT ? 1 : PHILOX_MAP_ITEMS];
  uint8_t* const owner = TEXT ? fq_slots : owner_static;
  __shared__ uint64_t cbase[CACHED ? PHILOX_CBASE : 1];
  // byte masks (0xff) of the first n bytes of 16; SLOT: a second row at +32 with the LAST n bytes (a reverse mate's live bytes)
  __shared__ uint4 nmask[SLOT ? 64 : 17];
  __shared__ uint32_t nmask2[17]; // the low 2n bits
  __shared__ uint32_t lds4[4];
  __shared__ uint64_t lds4w[COARSE ? 4 : 1];  // (TEXT too)
  const uint32_t qoff = qual_offset & 0xffu;
  // Where an escaped base (one item in 9 000) is noticed.  When every encoded quality the level-1 table can answer is
  // below 128 (the usual case: Phred + 33), an escape cell answers the byte 0xff and one test of the item's four quality
  // words finds it; otherwise the answers carry a flag bit that is or-ed together base by base (sixteen more
  // instructions per item).
  constexpr bool esc_q = ESCQ;
  {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t c = t; !COPY_ONLY && c < 1024u; c += 256u) {
      const uint32_t e = prof.philox_t1[c];
      uint32_t T = e & 0xffffu;
      const uint32_t A = e >> 16;
      uint32_t B = prof.philox_t1[1024u + c];
      if (T >= 16384u) { T = 0u; B = A; }
      // an escape cell answers quality byte 0xff when no real answer has bit 7 set (esc_q), else flag bit 2 next to s
      auto res = [&](uint32_t oc) { return oc == PHILOX_ESC ? (esc_q ? 0xff00u : ((qoff << 8) | 4u)) : (((((oc & 0xffu) + qoff) & 0xffu) << 8) | (oc >> 8)); };
      jtab[c] = make_uint2((c << 22) | (T << 8), res(A) | (res(B) << 16));
    }
    if (t <= 16u) {
      auto bytes = [](int k) { return k >= 4 ? 0xffffffffu : (k <= 0 ? 0u : ((1u << (8 * k)) - 1u)); };
      // (v_dot4_u32_u8 against them sums the live quality bytes 255-fold in one instruction per word: qsum is divided once,
      // at the end; and the same masks zero the padding of the slot layout with a plain `and`, no multiply)
      nmask[t] = make_uint4(bytes((int)t), bytes((int)t - 4), bytes((int)t - 8), bytes((int)t - 12));
      if (SLOT) {
        const int d = 16 - (int)t;  // the low d bytes are padding
        nmask[32u + t] = make_uint4(~bytes(d), ~bytes(d - 4), ~bytes(d - 8), ~bytes(d - 12));
      }
      nmask2[t] = t >= 16u ? 0xffffffffu : ((1u << (2u * t)) - 1u);
    }
    const uint32_t acgt = 0x54474341u;  // "ACGT"
    asc[t] = ((acgt >> (8 * (t & 3u))) & 0xffu) | (((acgt >> (8 * ((t >> 2) & 3u))) & 0xffu) << 8) |
             (((acgt >> (8 * ((t >> 4) & 3u))) & 0xffu) << 16) | (((acgt >> (8 * (t >> 6))) & 0xffu) << 24);
  }
  typedef const __attribute__((address_space(1))) ContigDev* global_contig_ptr;  // (a pointer out of a struct is generic: flat loads)
  const uint32_t* packed0 = nullptr;
  const uint32_t* mask0 = nullptr;
  if (CACHED) {
    const GenomeDev* G0 = genomes + genome_const;
    packed0 = G0->packed;
    mask0 = (HAS_EXC && G0->has_exc) ? G0->mask : nullptr;
    const uint32_t nc = G0->n_contigs < PHILOX_CBASE ? G0->n_contigs : PHILOX_CBASE;
    if (threadIdx.x < nc) cbase[threadIdx.x] = ((global_contig_ptr)G0->contigs)[threadIdx.x].base;
  }
  const uint4* rec4 = reinterpret_cast<const uint4*>(recs);
  uint64_t qsum = 0;  // adds encoded qualities; the offset is taken off at the end (every base is drawn exactly once: p_bases)
  uint32_t n_subst = 0, n_acgt = 0;
  // Plan-derived counters, gathered while the read records are written — kept out of the vector registers the item loop
  // needs: the flag counts are ballots (wave-uniform: scalar registers), the bases of this thread's reads a 32-bit sum
  // that moves to LDS long before it could overflow, wrapped qualities (never, with the offsets FASTQ uses) an LDS count.
  uint32_t p_bases32 = 0;
  uint32_t s_redrawn = 0, s_seedsubst = 0;  // per wave
  __shared__ unsigned long long spill_bases, spill_wrap;
  if (threadIdx.x == 0) { spill_bases = 0ull; spill_wrap = 0ull; }
  const uint64_t n_reads = paired ? 2 * n_units : n_units;
  const bool q_nowrap = qoff + prof.philox_qmax <= 255u;  // then no encoded quality wraps
  const uint32_t rpu = paired ? 2u : 1u;
  const uint64_t n_blocks = (n_units + PHILOX_UNITS - 1) / PHILOX_UNITS;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    // (the thread's number as a value the compiler cannot see through: addresses made from it in the per-block part are
    // then computed where they are used instead of being kept in registers across the item loop — they were what the
    // TEXT forms spilled)
    uint32_t tix = threadIdx.x;
    asm volatile("" : "+v"(tix));
    const uint64_t u0 = blk * PHILOX_UNITS;
    const uint32_t nu = (n_units - u0) < PHILOX_UNITS ? (uint32_t)(n_units - u0) : PHILOX_UNITS;
    const uint32_t nr = nu * rpu;
    // the block's first output byte (same for every lane: a scalar load)
    const uint64_t out0 = TEXT ? off64[(paired ? 2 * u0 : u0) >> 6] : (coarse ? off64[u0 >> 6] : u_off[u0]);
    uint8_t* const seq_blk = seq + out0;
    uint8_t* const qual_blk = (TEXT ? seq : qual) + out0;
    lds_barrier();  // the previous block's items are done with the records
    uint32_t g = 0;
    uint32_t n_items = 0, ex = 0;
    uint64_t rec_place = 0;  // TEXT: this thread's record, relative to the block's first
    if (TEXT) {
      uint32_t L0 = 0, h0 = 0;
      const bool on = tix < nr;
      if (on) {
        const uint64_t u = u0 + (paired ? (tix >> 1) : tix);
        L0 = PL(pl.len[u]);
        h0 = hlen[paired ? 2 * u + (tix & 1u) : u];
      }
      uint64_t tot2;
      const uint64_t ex2 = wg_exclusive_scan_2x32((uint64_t)((L0 + 15u) >> 4) | ((uint64_t)(on ? h0 + 2u * L0 + 5u : 0u) << 32), lds4w, &tot2, tix);
      ex = (uint32_t)ex2; n_items = (uint32_t)tot2; rec_place = ex2 >> 32;
    }
    uint32_t my_Lp = 0, my_pad = 0;  // this thread's read: its place in the streams, and (SLOT, reverse mate) the padding in front
    uint64_t my_rd = 0, my_dst = 0;
    // TEXT: what this thread's read shows in its header and cannot be had again from the thread's number (the read's
    // number, its id and which end of the window is the start can: they are put together where the header is formatted)
    uint64_t h_pos = 0, h_rec = 0;
    uint32_t h_L = 0, h_genome = 0, h_contig = 0, h_flags = 0;
    if (tix < nr) {
      const uint32_t t = tix;
      const uint64_t u = u0 + (paired ? (t >> 1) : t);
      const uint32_t rev = paired ? (t & 1u) : 0u;
      const uint32_t L = PL(pl.len[u]);
      g = (L + 15u) >> 4;
      const uint32_t contig = PL(u_contig[u]);
      const uint32_t genome = (!CACHED && u_genome) ? u_genome[u] : genome_const;
      const uint64_t rd = paired ? 2 * u + rev : u;
      const uint32_t Lp = SLOT ? ((L + 15u) & ~15u) : L;  // the read's place in the streams
      const uint64_t my_rec = TEXT ? out0 + rec_place : 0u;
      const uint64_t dst = TEXT ? my_rec + hlen[rd] + 1u : (coarse ? out0 : u_off[u] + (rev ? Lp : 0u));  // (coarse: after the scan below)
      my_Lp = Lp; my_pad = (SLOT && rev) ? Lp - L : 0u; my_rd = rd; my_dst = dst;
      const uint64_t pos = rev ? PL(pl.b[u]) : PL(pl.a[u]);  // first source base of this read on the contig
      const uint64_t key = COPY_ONLY ? 0ull : (rev ? PL(pl.qs2[u]) : PL(u_seed[u]));  // (no draws, no key; qs2 may not exist)
      uint64_t cb;
      const uint32_t* packed;
      const uint32_t* mk = nullptr;
      if (CACHED) {
        cb = cbase[contig & (PHILOX_CBASE - 1u)];
        packed = packed0;
        mk = mask0;
      } else {
        const GenomeDev* G = genomes + genome;
        cb = ((global_contig_ptr)G->contigs)[contig].base;
        packed = G->packed;
        if (HAS_EXC) mk = G->has_exc ? G->mask : nullptr;
      }
      const uint64_t src = cb + pos;
      PhRec rc;
      rc.k0 = (uint32_t)key; rc.k1 = (uint32_t)(key >> 32);
      rc.dst = (uint32_t)(dst - out0);
      rc.lw = (L & 0xffffu) | ((2u * (uint32_t)(src & 15u)) << 16) | (rev << 31);
      rc.wa = (uint64_t)(uintptr_t)(packed + (src >> 4));
      rc.gs = 0; rc.pad = 0;
      recs[t] = rc;
      if (HAS_EXC) { x_src[t] = src; x_mask[t] = mk; }
#if defined(SIMMR_ABLATE_META)
      if (false) {
#else
      if (FULL) {
#endif
        // metadata columns of this read (the other emit kernels leave them to k_write_meta)
        const uint32_t fl = PL(pl.flags[u]);
        if (TEXT) { h_pos = pos; h_L = L; h_genome = genome; h_contig = contig; h_flags = (paired && !rev) ? 0u : fl; h_rec = my_rec; }
        if (!TEXT) {
          if (paired) {
            if (o.start) COL_STORE(&o.start[rd], (uint64_t)(rev ? pos + L : pos));  // simulate.rs:289,295
            if (o.end) COL_STORE(&o.end[rd], (uint64_t)(rev ? pos : pos + L));      // simulate.rs:290,296
          } else {
            if (o.start) COL_STORE(&o.start[rd], (uint64_t)pos);                  // simulate.rs:515
            if (o.end) COL_STORE(&o.end[rd], (uint64_t)pl.b[u]);                  // simulate.rs:516
          }
          if (o.contig) COL_STORE(&o.contig[rd], contig);
          if (o.genome) COL_STORE(&o.genome[rd], genome);
          if (o.read_id) COL_STORE(&o.read_id[rd], read_id_base + (uint32_t)(first_unit + u));  // simulate.rs:85-89,274
          if (o.flags) COL_STORE(&o.flags[rd], (paired && !rev) ? (uint8_t)0 : (uint8_t)fl);
        }
        if (!rev) {
          p_bases32 += paired ? 2u * L : L;  // (a pair's reads are u16 lengths; a long read can be longer: the sum moves to LDS at 2^31 either way)
          if (p_bases32 >= 0x80000000u) { atomicAdd(&spill_bases, (unsigned long long)p_bases32); p_bases32 = 0u; }
        }
        s_redrawn += (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_REDRAWN)));
        s_seedsubst += (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_QSEED_SUBST))) +
                       (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_MSEED_SUBST)));
      }
    }
    if (TEXT) {
      // headers of the block's reads, FQ_GROUP at a time (128: two waves format while two wait; 64 was the most that fit
      // beside four workgroups per CU before the slots shared their memory with the item map): the threads that hold them
      // format, everybody copies
      const uint32_t W = 1u << fq_wshift;  // 16-byte windows per run (covers the longest)
      for (uint32_t half = 0; half * FQ_GROUP < nr; half++) {
        lds_barrier();  // the slots are free (and, the first time, template and literals are staged)
        if (tix < nr && (tix / FQ_GROUP) == half) {
          uint8_t* h = fq_slots + (tix & (FQ_GROUP - 1u)) * fq_hpitch;
          // the header's fields (fastq.rs:34-56), the same values the column form writes as metadata
          const uint32_t h_rev = paired ? (tix & 1u) : 0u;
          const uint64_t h_u = u0 + (paired ? (tix >> 1) : tix);
          const uint64_t h_rd = (uint64_t)rpu * u0 + tix;
          FqFields hf;
          hf.start = h_rev ? h_pos + h_L : h_pos;  // simulate.rs:289,295 / :515
          hf.end = h_rev ? h_pos : h_pos + h_L;    // simulate.rs:290,296 / :516 (a long read's end is its start + its length: k_plan_long_*)
          hf.genome = h_genome; hf.contig = h_contig; hf.flags = h_flags; hf.L = h_L;
          hf.read_id = read_id_base + (uint32_t)(first_unit + h_u);  // simulate.rs:85-89,274
          const uint32_t lead = h_rd > 0 ? 1u : 0u;
          const uint32_t at = fq_format_header(h, lead, fq_segs, fq_n_segs, fq_tb, fq_lit, hf, (paired && (tix & 1u)) ? '2' : '1');
          fq_run_at[tix & (FQ_GROUP - 1u)] = h_rec - lead;
          fq_run_len[tix & (FQ_GROUP - 1u)] = at;
          if (hf.L == 0) { uint8_t* p = seq + h_rec + (at - lead); p[0] = '\n'; p[1] = '+'; p[2] = '\n'; }  // (no item writes it)
          if (h_rd + 1 == n_reads) seq[off64[(n_reads + 63u) >> 6] - 1] = '\n';
        }
        lds_barrier();
        const uint32_t n_runs = nr - FQ_GROUP * half < FQ_GROUP ? nr - FQ_GROUP * half : FQ_GROUP;
        for (uint32_t wi = tix; wi < (n_runs << fq_wshift); wi += 256u) {
          const uint32_t i = wi >> fq_wshift, piece = wi & (W - 1u);
          const uint32_t n = fq_run_len[i];
          uint8_t* d = seq + fq_run_at[i];
          const uint8_t* sl = fq_slots + i * fq_hpitch;
          if (n >= 16u) {  // the last window ends where the run ends (it overlaps its neighbour with the same bytes)
            if (piece * 16u < n) {
              const uint32_t w0 = (piece + 1u) * 16u <= n ? piece * 16u : n - 16u;
#if defined(SIMMR_ABLATE_LINES)  /* timing only: the window stores of a wave as sixteen whole lines inside the text */
              {
                const uint64_t tot = off64[(n_reads + 63u) >> 6];
                const uint64_t last = tot >= 1024u ? ((tot - 1024u) & ~1023ull) : 0u;
                uint64_t a = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(fq_run_at[i] >> 10)) << 10;
                a = a < last ? a : last;
                if (tot >= 1024u) *reinterpret_cast<v4u32*>(seq + a + 16u * (tix & 63u)) = fq_read16(sl, w0);
              }
#else
              *reinterpret_cast<v4u32_unaligned*>(d + w0) = (v4u32_unaligned)fq_read16(sl, w0);
#endif
            }
          } else if (piece == 0u) {
            for (uint32_t j = 0; j < n; j++) d[j] = sl[j];
          }
        }
      }
    }
    if (TEXT) lds_barrier();  // the header slots are read out: their memory becomes the item map
    if (TEXT) {
      // (scanned above)
    } else if (coarse) {  // the reads' places too: the scan of their (padded) lengths rides in the upper half
      uint64_t tot2;
      const uint64_t ex2 = wg_exclusive_scan_2x32((uint64_t)g | ((uint64_t)my_Lp << 32), lds4w, &tot2, tix);
      ex = (uint32_t)ex2; n_items = (uint32_t)tot2;
      if (tix < nr) { recs[tix].dst = (uint32_t)(ex2 >> 32); my_dst = out0 + (ex2 >> 32); }
    } else {
      ex = wg_exclusive_scan_u32<true>(g, lds4, &n_items, tix);
    }
#if !defined(SIMMR_ABLATE_META)
    if (!TEXT && !COPY_ONLY && tix < nr) {
      COL_STORE(&o.seq_off[my_rd], (uint64_t)(my_dst + my_pad));  // first base (SLOT: a reverse mate's bases are right-aligned)
      if (my_rd + 1 == n_reads) o.seq_off[n_reads] = coarse ? off64[(n_units + 63u) >> 6] : u_off[n_units];  // closing CSR offset
    }
#endif
    if (tix < nr) recs[tix].gs = ex;
    r_gs[tix] = tix < nr ? ex : 0xffffffffu;
    if (tix == 0) r_gs[PHILOX_READS] = 0xffffffffu;
    // short reads: every read writes its index over its items, so an item finds its read with one LDS load
    const bool use_map = n_items <= PHILOX_MAP_ITEMS;
    if (use_map && tix < nr) {
      // (8 <= g <= 16, every 150 bp-class read: two 8-byte LDS writes at any byte address, the second ending where the
      // read's items end, instead of g byte writes in a loop)
      const uint64_t t8 = (uint64_t)tix * 0x0101010101010101ull;
      if (g >= 8u && g <= 16u) {
        *reinterpret_cast<u64_unaligned*>(owner + ex) = t8;
        *reinterpret_cast<u64_unaligned*>(owner + ex + g - 8u) = t8;
      } else {
        for (uint32_t j = 0; j < g; j++) owner[ex + j] = (uint8_t)tix;
      }
    }
    lds_barrier();
#if defined(SIMMR_ABLATE_ITEMS)
    const uint32_t i_end = n_items < 256u ? n_items : 256u;  // one round instead of all
#else
    const uint32_t i_end = n_items;
#endif
    // long reads (64 items and more each on average): a lane's items ascend 256 apart, so its read moves on by one
    // now and then — one look at the next read's first item instead of the eight dependent ones of the search
    const bool walk = !use_map && n_items >= nr * 64u;
    uint32_t r_walk = 0;
    // read of an item: last r with r_gs[r] <= item (items are asked for in ascending order per lane)
    auto locate = [&](const uint32_t it) -> uint32_t {
      uint32_t r = 0;
      if (use_map) {
        r = owner[it];
      } else if (walk) {
        while (r_gs[r_walk + 1u] <= it) r_walk++;  // (r_gs[nr ..] = ~0)
        r = r_walk;
      } else {
#pragma unroll
        for (uint32_t step = PHILOX_READS / 2; step; step >>= 1)
          if (r_gs[r + step] <= it) r += step;
      }
      return r;
    };
    // the plane word of an item: two 32-bit words at the read's word address + 4 * (item's group)
    auto plane_word = [&](const uint32_t it, const uint32_t r) -> uint64_t {
      const uint4 rb = rec4[2 * r + 1];
      const uint64_t wa = ((uint64_t)rb.x | ((uint64_t)rb.y << 32)) + 4ull * (it - rb.z);
#if defined(SIMMR_ABLATE_CODES)
      return wa * 0x9E3779B97F4A7C15ull;  // timing only: no load from the plane
#else
      return *reinterpret_cast<global_u64_unaligned_ptr>(wa);
#endif
    };
    // One item ahead: the NEXT item's read and plane word are fetched before this item's stores are issued, so the wait for
    // that load does not stand behind them in the in-order counter.  Only in the copy-only forms, which have nothing but
    // that wait between two items and registers to spare (the drawing forms gain nothing from it: LAB.md, round 3).
    constexpr bool prefetch = COPY_ONLY;
    uint32_t r_next = 0;
    uint64_t raw_next = 0;
    if (prefetch && threadIdx.x < i_end) { r_next = locate(threadIdx.x); raw_next = plane_word(threadIdx.x, r_next); }
    for (uint32_t item = threadIdx.x; item < i_end; item += 256u) {
      const uint32_t r = prefetch ? r_next : locate(item);
      const uint64_t raw = prefetch ? raw_next : plane_word(item, r);
      const uint4 ra = rec4[2 * r], rb = rec4[2 * r + 1];
      const uint32_t k0 = ra.x, k1 = ra.y, lw = ra.w;
      const uint32_t L = lw & 0xffffu, rev = lw >> 31;
      const uint32_t ci = item - rb.z;
      const uint32_t b0 = ci << 4;
      const uint32_t n = (L - b0) < 16u ? (L - b0) : 16u;
      // the 16 source bases: 32 bits at bit offset 2 * (source position & 15) of two plane words
      uint32_t codes = (uint32_t)(raw >> ((lw >> 16) & 31u));
      uint32_t exc = 0u;
      if (HAS_EXC) { const uint32_t* mk = x_mask[r]; if (mk) exc = fetch_mask16(mk, (int64_t)(x_src[r] + b0)); }
      // per base: 24 bits -> (Phred, substitution shift s); qualities packed as bytes, s as 2-bit fields
      uint32_t qr[4] = {0, 0, 0, 0}, ss = 0;
      if (COPY_ONLY && TEXT) { qr[0] = const_q4; qr[1] = const_q4; qr[2] = const_q4; qr[3] = const_q4; }
      if (!COPY_ONLY) {
        uint32_t w[12];
#pragma unroll
        for (int c = 0; c < 3; c++) philox4x32_10(3u * ci + (uint32_t)c, 0u, k0, k1, w + 4 * c);
        bool escaped;
        {  // the sixteen lookups
          constexpr bool FLAGGED = !ESCQ;
          uint32_t ea = 0;
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            const uint32_t w0 = w[3 * g4], w1 = w[3 * g4 + 1], w2 = w[3 * g4 + 2];
            const uint32_t R[4] = {w0 << 8, __builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 8), w2};
            uint32_t x[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
              x[h] = philox_pick(R[h], jtab);
              ss = __builtin_amdgcn_alignbit(x[h], ss, 2);  // ascending, so base j ends at bits 2j of ss
              if (FLAGGED) ea |= x[h];
            }
            // the four quality bytes (byte 1 of each x) with two v_perm_b32 and an or: no SDWA write into a partly
            // preserved register (four of those in a row on one VGPR need a wait state each on gfx940+, which the
            // compiler cannot see inside asm statements)
            qr[g4] = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0501u) | __builtin_amdgcn_perm(x[3], x[2], 0x05010c0cu);
          }
          escaped = FLAGGED ? (ea & 4u) != 0u : ((qr[0] | qr[1] | qr[2] | qr[3]) & 0x80808080u) != 0u;
        }
        if (escaped) philox_repair(k0, k1, ci, prof.philox_t1, prof.philox_t2, qoff, ss, qr);
      }
      // only live ACGT bases mutate (minimal_short.rs:120-128)
      // masks of the n live bases of the item: byte masks for the four quality words, 2-bit-field mask
      const uint4 bm = nmask[n];
      const uint32_t live2 = nmask2[n];
      if (HAS_EXC) ss &= ~spread16(exc);
      ss &= live2;
      n_subst += __builtin_popcount((ss | (ss >> 1)) & 0x55555555u);
      // (without exception bases and outside the copy-only form, ACGT bases = bases = p_bases: nothing to count per item)
      if (HAS_EXC) n_acgt += __builtin_popcount(~spread16(exc) & live2 & 0x55555555u);
      else if (COPY_ONLY && !TEXT) n_acgt += n;
      if (!COPY_ONLY) {
        uint32_t qs = __builtin_amdgcn_udot4(qr[0], bm.x, 0u, false);
        qs = __builtin_amdgcn_udot4(qr[1], bm.y, qs, false);
        qs = __builtin_amdgcn_udot4(qr[2], bm.z, qs, false);
        qs = __builtin_amdgcn_udot4(qr[3], bm.w, qs, false);
        qsum += qs;  // 255 x the sum of the live bytes (<= 16 * 255 * 255 per item)
      }
      if (!COPY_ONLY && !q_nowrap) {
        uint32_t nw = 0;
        for (uint32_t j = 0; j < n; j++) nw += ((qr[j >> 2] >> (8 * (j & 3u))) & 0xffu) < qoff ? 1u : 0u;
        if (nw) atomicAdd(&spill_wrap, (unsigned long long)nw);
      }
      // substitutions in the 2-bit code domain: code' = (code + s) mod 4, 16 bases at once
      // (two-bit addition without the field masks: the low bits add as xor, their carry = and goes into the high bit;
      // v_bitop3_b32 takes three inputs, so this is and-and, shift, xor-xor: three instructions for nine)
      codes = xor3(codes, ss, __builtin_amdgcn_bitop3_b32(codes, ss, 0x55555555u, 0x80) << 1);
      uint32_t o_s = ra.z + b0;
      const uint32_t o_q = TEXT ? o_s + L + 3u : o_s;  // (TEXT: the qualities' line follows the bases' line and "+\n")
      if (rev) {
        // mate 2 is reverse-complemented after mutation (simulate.rs:283), still in the code domain:
        // base b0+j -> byte L-1-(b0+j); the 16-n dead groups fall off the low end
        codes = reverse_complement_groups16(codes);
        if (HAS_EXC) { exc = __builtin_bitreverse32(exc) >> 16; codes ^= spread16(exc); }
        const uint32_t dead = 16u - n;
        if (SLOT) {
          // right-aligned: the group's 16 bytes end at slot byte Lp - b0, the dead groups stay at the low end (zeroed below)
          o_s = ra.z + (((L + 15u) & ~15u) - b0 - 16u);
        } else {
          if (dead) { codes >>= 2 * dead; if (HAS_EXC) exc >>= dead; }
          o_s = ra.z + (L - b0 - n);
        }
      }
      uint32_t s0, s1, s2, s3;
      if (HAS_EXC) {
        s0 = expand4(codes & 0xffu, exc & 0xfu); s1 = expand4((codes >> 8) & 0xffu, (exc >> 4) & 0xfu);
        s2 = expand4((codes >> 16) & 0xffu, (exc >> 8) & 0xfu); s3 = expand4(codes >> 24, (exc >> 12) & 0xfu);
      } else {
        s0 = asc[codes & 0xffu]; s1 = asc[(codes >> 8) & 0xffu]; s2 = asc[(codes >> 16) & 0xffu]; s3 = asc[codes >> 24];
      }
      if (SLOT) {
        // padding bytes are 0: the qualities' and a forward read's bases' high 16 - n bytes, a reverse mate's low ones
        // (every item, without a branch: the masks of n = 16 are all ones)
        qr[0] &= bm.x; qr[1] &= bm.y; qr[2] &= bm.z; qr[3] &= bm.w;
        const uint4 sm = nmask[(rev << 5) + n];
        s0 &= sm.x; s1 &= sm.y; s2 &= sm.z; s3 &= sm.w;
      }
      // qualities are already offset-encoded, forward order
      const uint64_t q_lo = (uint64_t)qr[0] | ((uint64_t)qr[1] << 32), q_hi = (uint64_t)qr[2] | ((uint64_t)qr[3] << 32);
      const uint64_t s_lo = (uint64_t)s0 | ((uint64_t)s1 << 32), s_hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
      if (prefetch && item + 256u < i_end) { r_next = locate(item + 256u); raw_next = plane_word(item + 256u, r_next); }
#if defined(SIMMR_ABLATE_STORES)
      asm volatile("" :: "v"(q_lo), "v"(q_hi), "v"(s_lo), "v"(s_hi), "v"(o_q), "v"(o_s));  // alive, not stored
#else
#if defined(SIMMR_ABLATE_LINES)
      // timing only: every 16-byte store instruction of a wave writes sixteen WHOLE 64-byte lines (the wave's first
      // lane's place rounded down to 1 KB, then lane by lane); wrong places, and clamped so that the 1 KB stays
      // inside the total_bases bytes of the streams (a shard of less than 1 KB writes nothing at all)
      const uint64_t abl_total = TEXT ? off64[(n_reads + 63u) >> 6] : (coarse ? off64[(n_units + 63u) >> 6] : u_off[n_units]);
      const uint64_t abl_last = abl_total >= 1024u ? ((abl_total - 1024u) & ~1023ull) : 0u;
      uint64_t abl_q = ((uint64_t)__builtin_amdgcn_readfirstlane(o_q) + out0) & ~1023ull;
      uint64_t abl_s = ((uint64_t)__builtin_amdgcn_readfirstlane(o_s) + out0) & ~1023ull;
      abl_q = abl_q < abl_last ? abl_q : abl_last;
      abl_s = abl_s < abl_last ? abl_s : abl_last;
      if (abl_total < 1024u) continue;
      uint8_t* qd = qual + abl_q + 16u * (threadIdx.x & 63u);
      uint8_t* sd = seq + abl_s + 16u * (threadIdx.x & 63u);
#elif defined(SIMMR_ABLATE_HOTSTORE)
      // timing only: every store lands in the first 64 KB of the buffers (same instructions, no DRAM write traffic)
      uint8_t* qd = qual + ((o_q + (uint32_t)out0) & 0xffffu);
      uint8_t* sd = seq + ((o_s + (uint32_t)out0) & 0xffffu);
#elif defined(SIMMR_ABLATE_ALIGN16)
      // timing only: every store lands on the aligned 16 bytes below its place (still inside the buffers)
      uint8_t* qd = (uint8_t*)((uintptr_t)(qual_blk + o_q) & ~(uintptr_t)15);
      uint8_t* sd = (uint8_t*)((uintptr_t)(seq_blk + o_s) & ~(uintptr_t)15);
#else
      uint8_t* qd = qual_blk + o_q;
      uint8_t* sd = seq_blk + o_s;
#endif
#if !defined(SIMMR_ABLATE_LINES)
      if (TEXT && ci == 0u)  // "\n+\n" and, once more, the first quality
#else
      if (false)
#endif
        *reinterpret_cast<uint32_t __attribute__((aligned(1)))*>(qd - 3) = 0x000a2b0au | ((uint32_t)q_lo << 24);
#if defined(SIMMR_ABLATE_ALL16) && defined(SIMMR_ABLATE_LINES)
      if (true) {  // timing only (the clamped whole-line places above hold 16 bytes for every lane)
#elif defined(SIMMR_ABLATE_ALL16)
      // timing only: partial groups store 16 bytes too (they overwrite the head of the next read) — except where those
      // 16 bytes would leave the streams: the last groups of the shard keep their exact stores
      if (n == 16u || (uint64_t)(o_q > o_s ? o_q : o_s) + out0 + 16u <= (coarse ? off64[(n_units + 63u) >> 6] : u_off[n_units])) {
#else
      if (SLOT || n == 16u) {
#endif
#if !defined(SIMMR_ABLATE_LINES) && !defined(SIMMR_ABLATE_HOTSTORE) && !defined(SIMMR_ABLATE_ALIGN16)
        if (SLOT) {  // (whole aligned lines per wave in the slot layout: nontemporal)
          if (FULL) slot_store16(qual_blk, o_q, q_lo, q_hi);
          slot_store16(seq_blk, o_s, s_lo, s_hi);
        } else
#endif
        {
          if (FULL) store16<SLOT>(qd, q_lo, q_hi);
          store16<SLOT>(sd, s_lo, s_hi);
        }
      } else if (FULL) {
        store_tail2(qd, q_lo, q_hi, sd, s_lo, s_hi, n);
      } else {
        store_tail(sd, s_lo, s_hi, n);
      }
#endif
    }
  }
  // sum of the raw Phred values (per lane modulo 2^64: a lane that wrote records but drew few bases goes "negative";
  // the sum over the lanes is exact)
  // (the masks of the dot products are 0xff bytes: qsum is a multiple of 255, divided exactly by the inverse of 255 modulo 2^64)
  __syncthreads();  // (the spilled counts)
  uint64_t p_bases = (uint64_t)p_bases32 + (threadIdx.x == 0 ? (uint64_t)spill_bases : 0ull);
  const uint64_t n_wrap = threadIdx.x == 0 ? (uint64_t)spill_wrap : 0ull;
  qsum = qsum * 0xFEFEFEFEFEFEFEFFull + 256ull * n_wrap - (uint64_t)qoff * p_bases;
  if (COPY_ONLY && TEXT) qsum = 60ull * p_bases;  // perfect_short.rs:42-44
  uint64_t acgt = (HAS_EXC || !FULL) ? (uint64_t)n_acgt : p_bases;
  for (int d = 32; d > 0; d >>= 1) {
    n_subst += __shfl_down(n_subst, d, 64);
    acgt += __shfl_down(acgt, d, 64);
    qsum += __shfl_down(qsum, d, 64);
    p_bases += __shfl_down(p_bases, d, 64);
  }
  // One add per counter and WORKGROUP, into one of SIMMR_CNT_SHARDS rows behind the run's counters (k_counters_fold sums
  // the rows when the counters are read): every wave adding to the same eight addresses served the adds one at a time —
  // with 64 workgroups per CU (16 384 workgroups: what the kernel runs best with) that was 460 000 adds to seven addresses.
  __shared__ unsigned long long wsum[4][SIMMR_N_COUNTERS];
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long* w = wsum[threadIdx.x >> 6];
    w[SIMMR_CNT_READS] = (FULL && blockIdx.x == 0 && threadIdx.x == 0) ? (unsigned long long)n_reads : 0ull;
    w[SIMMR_CNT_BASES] = !FULL ? 0ull : (unsigned long long)p_bases;
    w[SIMMR_CNT_ACGT_BASES] = (unsigned long long)acgt;
    w[SIMMR_CNT_SUBSTITUTIONS] = COPY_ONLY ? 0ull : (unsigned long long)n_subst;
    w[SIMMR_CNT_OUTER_REJECTS] = 0ull;
    w[SIMMR_CNT_REDRAWN] = !FULL ? 0ull : (unsigned long long)s_redrawn;
    w[SIMMR_CNT_SEED_SUBST] = !FULL ? 0ull : (unsigned long long)s_seedsubst;
    w[SIMMR_CNT_QUAL_SUM] = !FULL ? 0ull : (unsigned long long)qsum;
  }
  __syncthreads();
  if (threadIdx.x < SIMMR_N_COUNTERS && counters) {
    const unsigned long long v = wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
    if (v) atomicAdd(&counters[(1u + (blockIdx.x & (SIMMR_CNT_SHARDS - 1u))) * SIMMR_N_COUNTERS + threadIdx.x], v);
  }
}

// Rows 1 .. SIMMR_CNT_SHARDS of the counter block (what k_emit_philox's workgroups add to) summed into row 0 — the run's
// counters, which every other kernel adds to directly — and cleared.  One wave; launched where the counters are read.
extern "C" __global__ void __launch_bounds__(64)
k_counters_fold(unsigned long long* __restrict__ counters) {
  static_assert(SIMMR_CNT_SHARDS == 64, "one lane per shard row");
  unsigned long long* row = counters + (1u + threadIdx.x) * SIMMR_N_COUNTERS;
#pragma unroll
  for (uint32_t k = 0; k < SIMMR_N_COUNTERS; k++) {
    unsigned long long v = row[k];
    row[k] = 0ull;
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if (threadIdx.x == 0 && v) counters[k] += v;
  }
}

// ===========================================================================
// 9b. Emit: custom-short pairs (custom_short.rs:332-353, :522-529).
//
// simulate_phred_scores re-seeds the SAME StdRng at every position, so every
// quality of a read is a function of the first few words of one stream (w0 picks
// the alias column, (w1, w2) the f64 against the column's odds, w3 the score
// inside the bin) and of the position's PDF.  One lane per PAIR: it generates
// the first ChaCha12 block of both mates' streams once and then walks the two
// reads (same length, hence the same position at the same time) in groups of
// 16 positions.  All lanes of a wave are at the same position, i.e. the same
// PDF: its header comes from LDS, its tables are read once per wave (lane l
// holds entries l and 64 + l) and every lane fetches its column / bin with
// ds_bpermute.  (The item form dealt "16 positions of one read" to lanes like
// k_emit_philox and was bound by scattered table loads, 26 ms per 20 M reads;
// one lane per read with two 64-lane gathers per position kept the address
// unit busy for the whole kernel, 14.4 ms.)  The kernel writes qualities only:
// the bases are a plain copy (mate 2 complement-reversed) and come from
// k_emit_philox<.., COPY_ONLY>, whose stores are coalesced.  A sample whose
// words are rejected (rare) is drawn again from the stream itself, so the
// result is exact in every case.
// ===========================================================================
#define CUSTOM2_WORDS 16u
#define CUSTOM_LDS_PDFS 512u /* PDF headers kept in LDS (32 bytes each); later positions are read from memory */

// All lanes of a wave are at the same read position, so the header of that position's PDF is one value
// per wave.  Through the vector memory path it still costs a full load per field group (the compiler cannot
// use scalar loads next to the kernel's byte stores), and those loads kept the address unit busy for the
// whole kernel; from LDS it is a broadcast read.
SIMMR_DEV void stage_pdf_headers(const CustomDev& C, PdfDev* __restrict__ s_pdfs) {
  const uint32_t n = (2u + C.n_quality) < CUSTOM_LDS_PDFS ? (2u + C.n_quality) : CUSTOM_LDS_PDFS;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_pdfs[i] = C.pdfs[i];
  __syncthreads();
}
// the PDF of read position p: min(p, n_quality - 1) (custom_short.rs:339-345)
SIMMR_DEV PdfDev quality_pdf(const CustomDev& C, const PdfDev* __restrict__ s_pdfs, uint32_t p) {
  const uint32_t pi = 2u + (p < C.n_quality ? p : C.n_quality - 1u);
  return pi < CUSTOM_LDS_PDFS ? s_pdfs[pi] : C.pdfs[pi];
}

extern "C" __global__ void __launch_bounds__(256)
k_emit_custom_pe(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint32_t genome, uint64_t n_units,
                 PlanArrays pl, const uint64_t* __restrict__ u_off, const uint32_t* __restrict__ u_contig,
                 const uint64_t* __restrict__ u_seed, uint8_t* __restrict__ seq, uint8_t* __restrict__ qual,
                 uint32_t qual_offset, unsigned long long* __restrict__ counters, uint32_t* __restrict__ err) {
  __shared__ PdfDev s_pdfs[CUSTOM_LDS_PDFS];  // PDF headers: wave-uniform reads from LDS instead of the vector memory path
  const GenomeDev G = genomes[genome];
  const CustomDev C = prof.custom;
  stage_pdf_headers(C, s_pdfs);
  uint64_t qsum = 0;
  bool bad = false;
  const uint32_t qoff = qual_offset & 0xffu;
  const uint32_t lane = threadIdx.x & 63u;
  // One lane per PAIR: both mates have the same length, so they are at the same position, i.e. the same PDF, at
  // the same time, and one read of the PDF's tables serves 128 reads.
  for (uint64_t u0 = (uint64_t)blockIdx.x * 256; u0 < n_units; u0 += (uint64_t)gridDim.x * 256) {
    const uint64_t u = u0 + threadIdx.x;
    uint32_t L = 0;
    uint64_t off = 0, src0[2] = {0, 0}, key[2] = {0, 0};
    if (u < n_units) {
      L = pl.len[u];
      off = u_off[u];  // mate 1 at off, mate 2 at off + L
      const uint64_t base = G.contigs[u_contig[u]].base;
      src0[0] = base + pl.a[u];
      src0[1] = base + pl.b[u];
      key[0] = u_seed[u];   // simulate.rs:262: StdRng(pe_seed)
      key[1] = pl.qs2[u];   // simulate.rs:266: StdRng(the drawn seed)
    }
    uint32_t w0[2], w3[2];
    double v01[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      uint32_t o[16];
      chacha12_block(pcg32_expand(key[m]), 0, o);
      w0[m] = o[0]; w3[m] = o[3];
      v01[m] = __longlong_as_double((long long)(((((uint64_t)o[2] << 32) | o[1]) >> 12) | 0x3FF0000000000000ULL)) - 1.0;
    }
    for (uint32_t b0 = 0; __any(b0 < L); b0 += 16) {
      // no lane leaves the position loop early: every lane holds a slice of the current PDF's tables
      const uint32_t n = b0 >= L ? 0u : ((L - b0) < 16u ? (L - b0) : 16u);
      uint64_t q_lo[2] = {0, 0}, q_hi[2] = {0, 0};
#pragma unroll 2
      for (uint32_t j = 0; j < 16; j++) {
        if (!__any(j < n)) break;
        // wave-uniform: position b0 + j samples PDF min(position, n_quality - 1) (custom_short.rs:339-350)
        const PdfDev pdf = quality_pdf(C, s_pdfs, b0 + j);
        const bool narrow = pdf.n <= 128u && pdf.n_bins <= 128u;  // wave-uniform
        // The PDF's tables are read ONCE per wave, lane l taking entries l and 64 + l (coalesced), and every lane
        // then fetches its own column / bin from the lane that holds it (ds_bpermute): a 64-lane gather costs the
        // address unit one access per lane, and two of them per position and read were what bounded this kernel.
        Rec16 c0 = {0u, 0u, 0u, 0u}, c1 = c0, k0 = c0, k1 = c0;
        if (narrow) {
          if (lane < pdf.n) c0 = C.col_rec[pdf.off + lane];
          if (lane + 64u < pdf.n) c1 = C.col_rec[pdf.off + 64u + lane];
          if (lane < pdf.n_bins) k0 = C.bin_rec[pdf.off_bins + lane];
          if (lane + 64u < pdf.n_bins) k1 = C.bin_rec[pdf.off_bins + 64u + lane];
        }
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const uint64_t mm = (uint64_t)w0[m] * pdf.n;
          const uint32_t col = (uint32_t)(mm >> 32);  // < pdf.n
          uint32_t q = 0;
          bool fast = (uint32_t)mm <= pdf.idx_zone;
          if (narrow) {
            const int sc = (int)((col & 63u) << 2);
            uint32_t ox = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c0.x), oy = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c0.y),
                     oz = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c0.z);
            if (pdf.n > 64u) {  // wave-uniform
              const uint32_t x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c1.x), y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c1.y),
                             z1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sc, (int)c1.z);
              if (col >= 64u) { ox = x1; oy = y1; oz = z1; }
            }
            const double odds = __longlong_as_double((long long)(((uint64_t)oy << 32) | ox));  // {odds, alias}
            const uint32_t bin = (__dmul_rn(v01[m], pdf.w_scale) < odds) ? col : oz;
            fast = fast && bin < pdf.n_bins;
            const int sk = (int)((bin & 63u) << 2);
            uint32_t rx = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k0.x), ry = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k0.y),
                     rz = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k0.z);
            if (pdf.n_bins > 64u) {  // wave-uniform
              const uint32_t x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k1.x), y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k1.y),
                             z1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sk, (int)k1.z);
              if ((bin & 127u) >= 64u) { rx = x1; ry = y1; rz = z1; }
            }
            const uint64_t m2 = (uint64_t)w3[m] * rx;  // {range, zone, low}
            if (rx == 0) q = w3[m];
            else if ((uint32_t)m2 <= ry) q = rz + (uint32_t)(m2 >> 32);
            else fast = false;
          } else if (fast) {  // tables wider than two entries per lane: gather
            const Rec16 cr = C.col_rec[pdf.off + col];
            const double odds = __longlong_as_double((long long)(((uint64_t)cr.y << 32) | cr.x));
            const uint32_t bin = (__dmul_rn(v01[m], pdf.w_scale) < odds) ? col : cr.z;
            fast = bin < pdf.n_bins;
            if (fast) {
              const Rec16 br = C.bin_rec[pdf.off_bins + bin];
              const uint64_t m2 = (uint64_t)w3[m] * br.x;
              if (br.x == 0) q = w3[m];
              else if ((uint32_t)m2 <= br.y) q = br.z + (uint32_t)(m2 >> 32);
              else fast = false;
            }
          }
          if (j < n) {
            // a rejected word or a bin without a range: the general routine on the stream itself (and its error reporting)
            if (!fast) q = pdf_sample_stream(key[m], C, pdf, &bad);
            q &= 0xffu;  // `as u8`
            qsum += q;
            const uint64_t enc = (q + qoff) & 0xffu;
            if (j < 8u) q_lo[m] |= enc << (8u * j); else q_hi[m] |= enc << (8u * (j - 8u));
          }
        }
      }
      if (n == 0u) continue;
      // the bases (a plain copy, mate 2 reverse-complemented: custom_short.rs:522-529, simulate.rs:283) are written by
      // k_emit_philox<.., COPY_ONLY>, whose lanes are consecutive 16-byte pieces of the output
#pragma unroll
      for (int m = 0; m < 2; m++) {
        uint8_t* qd = qual + off + (m ? L : 0u) + b0;
        if (n == 16u) store16(qd, q_lo[m], q_hi[m]);
        else store_tail(qd, q_lo[m], q_hi[m], n);
      }
    }
  }
  if (bad) atomicOr(err, SIMMR_ERRBIT_PDF);
  for (int d = 32; d > 0; d >>= 1) qsum += __shfl_down(qsum, d, 64);
  if ((threadIdx.x & 63u) == 0 && counters) shard_add(counters, SIMMR_CNT_QUAL_SUM, (unsigned long long)qsum);
}

// ===========================================================================
// 9c. Emit: a custom model on the long-read path (simulate.rs:497-503 with
//     CustomShortErrorProfile: simulate_phred_scores :332-353, simulate_errors
//     :455-516, simulate_point_mutations = copy :522-529).
//
// Two kernels, one lane per read each (k_custom_long_qual, k_custom_long_splice: the splice then
// fits 70 VGPRs).  Qualities as in 9b; positions from n_quality - 1 on share
// one PDF and one seed, i.e. one value, which is drawn once and stored 16 bytes
// at a time.  simulate_errors walks the read k-mer by k-mer, each visited k-mer
// replaced IN PLACE by an alternate before the next one is read, so k-mer i is
// the last k - 1 bases of the alternate chosen at i - 1 plus source base
// i + k - 1: the lane keeps that window in a register (2 bits per base plus a
// flag bit per base for N and for '-') and shifts one source base in per step.
// The reference's per-call HashMap and per-k-mer WeightedAliasIndex<f32> are
// the precomputed tables of CustomDev (the host builds them with the same f32
// arithmetic): a k-mer of ACGT indexes a direct table with its window value,
// one with an N probes a small hash table; the draws come from the read's own
// StdRng stream, one block at a time in the lane's LDS row.  An alternate with an
// 'N' field is a deletion: the reference then panics on its next slice (or, at
// the last k-mer, returns fewer bases than qualities), so the lane raises
// SIMMR_ERRBIT_KMER instead.
// ===========================================================================
// (the row is passed as an LDS pointer: through a generic one the sixteen stores compile to flat_store_dword, which
// take the vector-memory path — 2 per step of the splice, as many as its loads)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __attribute__((noinline)) void refill_words(const Key key, uint32_t blk, lds_u32* __restrict__ row, uint32_t stride = 1u) {
  uint32_t o[16];
  chacha12_block(key, (uint64_t)blk, o);
#pragma unroll
  for (int i = 0; i < 16; i++) row[(uint32_t)i * stride] = o[i];
}

// qualities of the long reads (simulate_phred_scores, custom_short.rs:332-353): one lane per read
extern "C" __global__ void __launch_bounds__(256)
k_custom_long_qual(ProfileDev prof, uint64_t n_units, const uint32_t* __restrict__ order, PlanArrays pl,
                   const uint64_t* __restrict__ u_off,
                   const uint64_t* __restrict__ u_seed, uint8_t* __restrict__ qual, uint32_t qual_offset,
                   unsigned long long* __restrict__ counters, uint32_t* __restrict__ err) {
  __shared__ uint32_t words[256][CUSTOM2_WORDS + 1];
  __shared__ PdfDev s_pdfs[CUSTOM_LDS_PDFS];
  const CustomDev C = prof.custom;
  stage_pdf_headers(C, s_pdfs);
  uint64_t qsum = 0;
  bool bad = false;
  const uint32_t qoff = qual_offset & 0xffu;
  uint32_t* const row = words[threadIdx.x];
  for (uint64_t r0 = (uint64_t)blockIdx.x * 256; r0 < n_units; r0 += (uint64_t)gridDim.x * 256) {
    // reads in order of length (k_len_*), longest first: the lanes of a wave then finish together
    const uint64_t ti = r0 + threadIdx.x;
    const uint64_t u = ti < n_units ? (order ? (uint64_t)order[n_units - 1 - ti] : ti) : n_units;
    uint32_t n = 0;
    uint64_t off = 0, seed = 0;
    if (u < n_units) { n = pl.len[u]; off = u_off[u]; seed = u_seed[u]; }  // read_seed (simulate.rs:497)
    uint32_t o[16];
    chacha12_block(pcg32_expand(seed), 0, o);
#pragma unroll
    for (int i = 0; i < 16; i++) row[i] = o[i];
    // ---- simulate_phred_scores
    {
      const uint32_t w0 = o[0], w3 = o[3];
      const double v01 = __longlong_as_double((long long)(((((uint64_t)o[2] << 32) | o[1]) >> 12) | 0x3FF0000000000000ULL)) - 1.0;
      auto sample = [&](const PdfDev pdf) -> uint32_t {
        const uint64_t m = (uint64_t)w0 * pdf.n;
        uint32_t q = 0;
        bool fast = (uint32_t)m <= pdf.idx_zone;
        if (fast) {
          const uint32_t col = (uint32_t)(m >> 32);
          const Rec16 cr = C.col_rec[pdf.off + col];
          const double odds = __longlong_as_double((long long)(((uint64_t)cr.y << 32) | cr.x));
          const uint32_t bin = (__dmul_rn(v01, pdf.w_scale) < odds) ? col : cr.z;
          fast = bin < pdf.n_bins;
          if (fast) {
            const Rec16 br = C.bin_rec[pdf.off_bins + bin];
            const uint64_t m2 = (uint64_t)w3 * br.x;
            if (br.x == 0) q = w3;
            else if ((uint32_t)m2 <= br.y) q = br.z + (uint32_t)(m2 >> 32);
            else fast = false;
          }
        }
        if (!fast) {
          bool ovf = false;
          q = pdf_sample_words(row, CUSTOM2_WORDS, C, pdf, &bad, &ovf);
          if (ovf) q = pdf_sample_stream(seed, C, pdf, &bad);
        }
        return q & 0xffu;  // `as u8`
      };
      uint32_t q_last = 0;  // the value of every position >= n_quality - 1 (custom_short.rs:339-350)
      if (n >= C.n_quality) q_last = sample(quality_pdf(C, s_pdfs, C.n_quality - 1u));
      const uint64_t fill = ((q_last + qoff) & 0xffu) * 0x0101010101010101ULL;
      // positions below b_fill (the first 16-group that lies entirely at or after n_quality - 1) are sampled one by
      // one, lane per read; the constant rest of every read is then written by the whole wave, 1 KB per store
      const uint32_t b_fill = (C.n_quality - 1u + 15u) & ~15u;
      for (uint32_t b0 = 0; b0 < b_fill && __any(b0 < n); b0 += 16) {
        if (b0 >= n) continue;
        const uint32_t cnt = (n - b0) < 16u ? (n - b0) : 16u;
        uint64_t q_lo = 0, q_hi = 0;
        uint32_t redo = 0u;
        // Every position draws from the same four words, so the sixteen samples of a group do not depend on one
        // another: eight at a time, first all alias columns are loaded, then all bins (two round trips per eight
        // samples instead of two per sample; the kernel was bound by that chain).  A sample that leaves the straight
        // path (a rejected draw, a bad bin) is redone the general way.
#pragma nounroll
        for (uint32_t h = 0; h < 16u; h += 8u) {
          Rec16 rec[8];
          uint32_t colv[8], ok = 0u;
#pragma unroll
          for (uint32_t j = 0; j < 8u; j++) {
            const PdfDev pdf = quality_pdf(C, s_pdfs, b0 + h + j);
            const uint64_t m = (uint64_t)w0 * pdf.n;
            colv[j] = (uint32_t)(m >> 32);
            const bool f = h + j < cnt && (uint32_t)m <= pdf.idx_zone;
            ok |= (f ? 1u : 0u) << j;
            rec[j] = C.col_rec[f ? pdf.off + colv[j] : 0u];
          }
#pragma unroll
          for (uint32_t j = 0; j < 8u; j++) {
            const PdfDev pdf = quality_pdf(C, s_pdfs, b0 + h + j);
            const double odds = __longlong_as_double((long long)(((uint64_t)rec[j].y << 32) | rec[j].x));
            const uint32_t bin = (__dmul_rn(v01, pdf.w_scale) < odds) ? colv[j] : rec[j].z;
            const bool f = ((ok >> j) & 1u) != 0u && bin < pdf.n_bins;
            ok &= ~((f ? 0u : 1u) << j);
            rec[j] = C.bin_rec[f ? pdf.off_bins + bin : 0u];
          }
#pragma unroll
          for (uint32_t j = 0; j < 8u; j++) {
            const Rec16 br = rec[j];
            const uint64_t m2 = (uint64_t)w3 * br.x;
            bool f = ((ok >> j) & 1u) != 0u;
            uint32_t q = br.x == 0u ? w3 : br.z + (uint32_t)(m2 >> 32);
            if (br.x != 0u && (uint32_t)m2 > br.y) f = false;
            if (h + j < cnt) {
              if (!f) { redo |= 1u << (h + j); q = 0u; }
              q &= 0xffu;  // `as u8`
              qsum += q;
              const uint64_t enc = f ? (q + qoff) & 0xffu : 0u;
              if (h == 0u) q_lo |= enc << (8u * j); else q_hi |= enc << (8u * j);
            }
          }
        }
#pragma nounroll
        while (redo) {  // rare: one copy of the general sampler
          const uint32_t j = (uint32_t)__builtin_ctz(redo);
          redo &= redo - 1u;
          const uint32_t q = sample(quality_pdf(C, s_pdfs, b0 + j));
          qsum += q;
          const uint64_t enc = (q + qoff) & 0xffu;
          if (j < 8u) q_lo |= enc << (8u * j); else q_hi |= enc << (8u * (j - 8u));
        }
        uint8_t* qd = qual + off + b0;
        if (cnt == 16u) {
          *reinterpret_cast<u64_unaligned*>(qd) = q_lo;
          *reinterpret_cast<u64_unaligned*>(qd + 8) = q_hi;
        } else {
          store_tail(qd, q_lo, q_hi, cnt);
        }
      }
      if (n > b_fill) qsum += (uint64_t)q_last * (n - b_fill);
      const uint32_t lane = threadIdx.x & 63u;
      for (uint32_t src = 0; src < 64u; src++) {
        const uint32_t n_s = __shfl(n, (int)src, 64);
        if (n_s <= b_fill) continue;  // wave-uniform
        const uint64_t off_s = __shfl(off, (int)src, 64), fill_s = __shfl(fill, (int)src, 64);
        // The constant rest of a read is most of what this kernel writes (20 GB per million reads of 20 kb: the kernel is
        // bound by these stores, not by the modelled positions — both rewrites of the sampling loop that round 5 tried left
        // it slower, profiles/r5/ab_custom_qual_*.log).  Lanes own ALIGNED 16-byte chunks of device memory, lane l the
        // chunk l of every kilobyte from the line the run starts in: a wave's store instruction is sixteen whole 64-byte
        // lines (nontemporal: nobody reads them on the device), and only the run's first and last chunk are written bytewise.
        uint8_t* const run_a = qual + off_s + b_fill;                      // first byte of the run
        uint8_t* const run_b = qual + off_s + n_s;                         // one past its last
        uint8_t* const line0 = (uint8_t*)((uintptr_t)run_a & ~(uintptr_t)63);
        for (uint8_t* q16 = line0 + 16u * lane; q16 < run_b; q16 += 1024u) {
          if (q16 >= run_a && q16 + 16 <= run_b) {
            v4u32 v;
            v.x = (uint32_t)fill_s; v.y = v.x; v.z = v.x; v.w = v.x;
            stream_store(reinterpret_cast<v4u32*>(q16), v);
          } else if (q16 + 16 > run_a) {
            uint8_t* lo = q16 < run_a ? run_a : q16;
            uint8_t* hi = q16 + 16 < run_b ? q16 + 16 : run_b;
            for (uint8_t* b = lo; b < hi; b++) *b = (uint8_t)fill_s;
          }
        }
      }
    }
  }
  if (bad) atomicOr(err, SIMMR_ERRBIT_PDF);
  for (int d = 32; d > 0; d >>= 1) qsum += __shfl_down(qsum, d, 64);
  if ((threadIdx.x & 63u) == 0 && counters) shard_add(counters, SIMMR_CNT_QUAL_SUM, (unsigned long long)qsum);
}

// bases of the long reads (simulate_errors, custom_short.rs:455-516): one lane per read.
// FAST (k <= 7 and short alternate lists, CustomDev::kmer_cols): the per-k-mer counts sit in LDS as bytes and the
// columns at a fixed stride, so a visited k-mer costs one dependent global load (its column) instead of two (entry,
// then column) — the kernel is bound by that chain and by the rate of per-lane line requests, not by arithmetic.  One
// workgroup of 1024 lanes per CU then shares the 16 KB count table: LDS = word rows [32][1024] (word-major: lanes
// of a wave in consecutive banks whatever word each is at) + the Uniform(0, n) zones + the counts.
//
// CTR: SIMMR_RNG_PHILOX with a custom long-read model (include/simmr_hip.h; restated on the CPU by the test tree:
// orc_custom_simulate_errors_philox).  The walk is the same; a visited k-mer's alternate is drawn from ONE word of
// Philox4x32-10 keyed by the read's seed — position i takes word i & 3 of the block with counter
// (i >> 2, 2, 'simm', 'r\0\0\3') — in TWO LEVELS (custom_model.hpp: ctr_splice_tables): X >> 8 below the k-mer's
// threshold T24 answers "the k-mer stays what it is" without any table access; only the rest (the model's error rate: a
// tenth of the visited k-mers) goes to an alias column, chosen and resolved by the same word rescaled (T24 leaves a
// power of two of values above it, so the rescaling is a shift).  That is the point of the mode:
// the reference's draw needs its column for EVERY visited k-mer, sixty-four different cache lines per wave-step, and the
// reference-mode kernel is bound by exactly that (the same kernel with Philox words but the reference's one-level draw
// took the same 86 ms per 20 Gbases; with one lane in ten loading, 43: profiles/r4/ab_splice_ctr_*.log).  Nothing of a
// stream is kept either: no LDS rows (the reference mode's 128 KB per workgroup), no refill in the middle of a chain, no
// position to take back; the LDS holds T24 << 8 | count per k-mer (4^k words).
constexpr uint32_t SPLICE_FAST_LANES = 1024u;
__host__ __device__ inline uint32_t splice_fast_lds_bytes(uint32_t k) { return 32u * SPLICE_FAST_LANES * 4u + 256u * 4u + (1u << (2u * k)); }
__host__ __device__ inline uint32_t splice_ctr_lds_bytes(uint32_t k) { return 4u << (2u * k); }  // (+ 16 bytes per lane: the store stash, engine.hip)
#if !defined(SPLICE_CTR_LANES)
#define SPLICE_CTR_LANES 768
#endif
constexpr uint32_t SPLICE_CTR_LANES_MAX = SPLICE_CTR_LANES;  // two workgroups beside each other when the table takes 64 KB (k = 7)
template <bool HAS_EXC, bool FAST, bool CTR = false>
#if !defined(SPLICE_CTR_WAVES)
#define SPLICE_CTR_WAVES 6  /* waves per SIMD asked of the compiler for the counter mode's instantiations: two workgroups of 768 lanes per CU */
#endif
__global__ void __launch_bounds__(CTR ? SPLICE_CTR_LANES_MAX : (FAST ? 1024 : 256), CTR ? SPLICE_CTR_WAVES : 1)
k_custom_long_splice(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint64_t n_units,
                     const uint32_t* __restrict__ order, PlanArrays pl,
                     const uint64_t* __restrict__ u_off, const uint32_t* __restrict__ u_contig,
                     const uint32_t* __restrict__ u_genome, const uint64_t* __restrict__ u_seed,
                     uint8_t* __restrict__ seq, unsigned long long* __restrict__ counters, uint32_t* __restrict__ err) {
  const uint32_t NT = CTR ? blockDim.x : (FAST ? SPLICE_FAST_LANES : 256u);
  constexpr uint32_t RS = (FAST && !CTR) ? SPLICE_FAST_LANES : 1u;  // distance between two words of a lane's row
  __shared__ uint32_t words[(FAST || CTR) ? 1 : 256][33];  // per lane: two blocks of its stream
  extern __shared__ uint32_t splice_lds[];        // FAST: rows, zones, counts (CTR: the counts)
  const CustomDev C = prof.custom;
  const uint32_t K = C.kmer_size;  // 1..10 (checked on the host)
  uint32_t n_acgt = 0, n_subst = 0;
  bool bad_kmer = false;
  lds_u32* const row = (lds_u32*)(FAST ? splice_lds + threadIdx.x : words[CTR ? 0u : threadIdx.x]);  // (CTR: not used)
  const uint32_t* const s_zone = splice_lds + (CTR ? 0u : 32u * SPLICE_FAST_LANES);
  const uint8_t* const s_cnt8 = reinterpret_cast<const uint8_t*>(s_zone + (CTR ? 0u : 256u));
  const uint32_t* const s_tab = splice_lds;  // CTR: T24 << 8 | count
  __shared__ uint32_t s_asc[CTR ? 256 : 1];  // CTR: four 2-bit codes -> four ASCII bytes
  if (CTR) {
    const uint32_t acgt = 0x54474341u;  // "ACGT"
    for (uint32_t t = threadIdx.x; t < 256u; t += NT)
      s_asc[t] = ((acgt >> (8 * (t & 3u))) & 0xffu) | (((acgt >> (8 * ((t >> 2) & 3u))) & 0xffu) << 8) |
                 (((acgt >> (8 * ((t >> 4) & 3u))) & 0xffu) << 16) | (((acgt >> (8 * (t >> 6))) & 0xffu) << 24);
    if (!FAST) __syncthreads();
  }
  if (FAST && CTR) {
    const uint32_t n_w = 1u << (2u * K);
    for (uint32_t w = threadIdx.x; w < n_w; w += NT) splice_lds[w] = C.kmer_tab32[w];
    __syncthreads();
  } else if (FAST) {
    uint32_t* zone_w = splice_lds + (CTR ? 0u : 32u * SPLICE_FAST_LANES);
    if (!CTR && threadIdx.x < 256u) {
      const uint32_t c = threadIdx.x;  // Uniform::new(0u32, c): the largest accepted low word
      zone_w[c] = c ? 0xFFFFFFFFu - (uint32_t)((0x100000000ULL - c) % c) : 0u;
    }
    uint32_t* cnt_w = zone_w + (CTR ? 0u : 256u);
    const uint32_t n_w = (1u << (2u * K)) >> 2;  // 4^K bytes
    const uint32_t* src_w = reinterpret_cast<const uint32_t*>(C.kmer_cnt8);
    for (uint32_t w = threadIdx.x; w < n_w; w += NT) cnt_w[w] = src_w[w];
    __syncthreads();
  }
  for (uint64_t r0 = (uint64_t)blockIdx.x * NT; r0 < n_units; r0 += (uint64_t)gridDim.x * NT) {
    const uint64_t ti = r0 + threadIdx.x;
    const uint64_t u = ti < n_units ? (order ? (uint64_t)order[n_units - 1 - ti] : ti) : n_units;
    uint32_t n = 0;
    uint64_t off = 0, src0 = 0, seed = 0;
    const uint32_t* packed = nullptr;
    const uint32_t* mask = nullptr;
    if (u < n_units) {
      n = pl.len[u];
      off = u_off[u];
      const GenomeDev* G = genomes + u_genome[u];
      packed = G->packed;
      mask = (HAS_EXC && G->has_exc) ? G->mask : nullptr;
      src0 = G->contigs[u_contig[u]].base + pl.a[u];
      seed = u_seed[u];  // read_seed re-seeds every per-read generator (simulate.rs:497-503)
    }
    const Key key = CTR ? Key{} : pcg32_expand(seed);
    if (!CTR) refill_words(key, 0, row, RS);
    const uint32_t pk0 = (uint32_t)seed, pk1 = (uint32_t)(seed >> 32);  // CTR: the Philox key
    // ---- simulate_errors
    {
      // StdRng(read_seed): the row holds two blocks of the stream, block b at words (b & 1) * 16.  Every 8
      // steps all lanes that are about to run out generate their next block TOGETHER (a step takes two
      // words, so lanes that drifted apart still refill in the same call); a lane that needs more in
      // between (rejected draws) generates on its own.
      uint32_t wpos = 0, have = 1;  // next word; blocks [0, have) were generated
      auto next_word = [&]() -> uint32_t {
        if ((wpos >> 4) >= have) { refill_words(key, have, row + (have & 1u) * 16u * RS, RS); have++; }
        const uint32_t w = row[(wpos & 31u) * RS];
        wpos++;
        return w;
      };
      // bases i .. i+K-1 of the edited sequence: 2 bits per base plus one flag bit per base for N and for '-';
      // owin / oexc: the same positions of the original (for the counters)
      uint32_t win = 0, nm = 0, dm = 0, owin = 0, oexc = 0;
      if (n) {  // lanes without a read have no plane to read from
        const uint32_t c16 = fetch_codes16(packed, (int64_t)src0);  // K <= 10 bases
        const uint32_t e16 = (HAS_EXC && mask) ? fetch_mask16(mask, (int64_t)src0) : 0u;
        const uint32_t k = K < n ? K : n;
        const uint32_t keep = (1u << k) - 1u;
        const uint32_t exc = e16 & keep;
        win = c16 & ((1u << (2u * k)) - 1u) & ~spread16(exc);
        if (HAS_EXC) { nm = exc & ~c16_odd(c16); dm = exc & c16_odd(c16); }  // exception code 4 + (code & 1): N / '-'
      }
      owin = win; oexc = nm | dm;
      bool dead = false;  // after an error the lane only copies
      uint8_t* const sd = seq + off;
      const uint32_t top2 = 2u * (K - 1u), top1 = K - 1u;
      // FAST: the source bases come 256 at a time (17 words, shifted to the lane's bit offset once): a lane's loads
      // then touch every 64-byte line of its stretch about twice instead of sixteen times — the lines do not survive in
      // L2 between two loads of a lane (75 GB fetched per 20 Gbases with one 8-byte load per group)
      // (CTR: 64 at a time — four words live instead of sixteen; its workgroups are 256 lanes, eight or more per CU)
#if !defined(SPLICE_CTR_SRC_WORDS)
#define SPLICE_CTR_SRC_WORDS 4
#endif
      constexpr uint32_t SW = !FAST ? 1u : (CTR ? (uint32_t)SPLICE_CTR_SRC_WORDS : 16u);
      uint32_t sw[SW];
      uint4 held = make_uint4(0u, 0u, 0u, 0u);  // (reference mode, FAST: an even group waiting for its odd neighbour's store)
      for (uint32_t i0 = 0; __any(i0 < n); i0 += 16u) {
        // the 16 source bases that enter the window during this group: positions i0 + K .. i0 + K + 15
        // (at most K + 31 bases past the read, K + 271 with the 256-base chunks: inside the plane's back padding)
        uint32_t s16;
        if (FAST) {
          if ((i0 & (16u * SW - 1u)) == 0u && i0 < n) {
            const int64_t p = (int64_t)(src0 + i0 + K);
            const __attribute__((address_space(1))) uint32_t* q = (const __attribute__((address_space(1))) uint32_t*)(packed + (p >> 4));
            const uint32_t sh = 2u * (uint32_t)(p & 15);
            uint32_t prev = q[0];
#pragma unroll
            for (int j = 0; j < (int)SW; j++) {
              const uint32_t next = q[j + 1];
              sw[j] = __builtin_amdgcn_alignbit(next, prev, sh);
              prev = next;
            }
          }
          const uint32_t g = (i0 >> 4) & (SW - 1u);  // the same for every lane: one indexed register read
          s16 = sw[0];
#pragma unroll
          for (uint32_t j = 1; j < SW; j++) s16 = g == j ? sw[FAST ? j : 0] : s16;
        } else {
          s16 = i0 < n ? fetch_codes16(packed, (int64_t)(src0 + i0 + K)) : 0u;
        }
        const uint32_t x16 = (HAS_EXC && mask && i0 < n) ? fetch_mask16(mask, (int64_t)(src0 + i0 + K)) : 0u;
        uint32_t out[4] = {0u, 0u, 0u, 0u};  // 16 output bases
        bool todo = true;  // this group still has to be walked the general way
        if (FAST) {
          // A group that lies inside every read of the wave (reads are dealt in order of length, so that is all but the
          // last of a wave's groups), with no exception base in sight: sixteen steps without a branch per lane.  What
          // the straight line cannot do — a rejected index draw (1e-9), a k-mer whose weights the reference panics on,
          // an alternate with an N — is only noticed; such a lane takes its state back and walks the group again below.
          const bool inside = i0 + 16u + K <= n && !dead && (!HAS_EXC || (nm | dm | oexc | x16) == 0u);
          if (__all(inside)) {
            const uint32_t win_0 = win, owin_0 = owin, wpos_0 = wpos, subst_0 = n_subst;
            const uint32_t kmask2 = (1u << (2u * K)) - 1u;
            const char* const cols_bytes = reinterpret_cast<const char*>(CTR ? C.kmer_cols_ctr : C.kmer_cols);
            const uint32_t stride16 = C.kmer_stride << 4;
            bool rare = false;
            if (CTR) {
              // The counter mode's group as a ROLLED loop over its four Philox blocks (one word per step): nothing but the
              // window, two code registers and one block lives across a block's steps, so the kernel fits the registers of
              // six waves per SIMD.  The 16 output bases are gathered as 2-bit codes (v_alignbit shifts a step's code in)
              // and become ASCII once per group through the LDS table; the substitutions are counted once per group from the
              // codes of the output and of the original.
              uint32_t oc = 0u, orig = 0u, sbits = s16;
              uint32_t rarev = 0u;  // bit 31: something the straight line cannot do was met
#pragma nounroll
              for (uint32_t qd = 0; qd < 4u; qd++) {
                uint32_t bw[4];
                philox4x32_10((i0 >> 2) + qd, 2u, pk0, pk1, bw);  // (i0 is a multiple of 16)
#pragma unroll
                for (uint32_t h = 0; h < 4u; h++) {
                  const uint32_t X = bw[h];
                  const uint32_t e32 = s_tab[win];
                  const uint32_t cnt = e32 & 0xffu;
                  const uint32_t tx = e32 & 0xffffff00u;  // T24 << 8
                  // level 1: X >> 8 < T24  <=>  X < T24 << 8; level 2 only for a k-mer of the model that did not stay
                  const bool lvl2 = (cnt - 1u < 254u) & (X >= tx);
                  // level 2: the 2^(e + 8) values from T24 << 8 up, shifted to a full word (24 - e = the leading zeros of
                  // ~(T24 << 8 | 0xff), whose set bits are bits 8 .. e + 7; e >= 1)
                  const uint32_t Z = (X - tx) << __builtin_clz(~(e32 | 0xffu));
                  // No branch: every lane loads — the lanes that stay at level 1 all the same record (one line for the
                  // whole wave), the others their column — so that the load is in flight while the block's other steps'
                  // arithmetic runs; nine wave-steps in ten have a lane at level 2 anyway.
                  const uint64_t m = (uint64_t)Z * cnt;
                  const uint32_t off = lvl2 ? __umul24(win, stride16) + ((uint32_t)(m >> 32) << 4) : 0u;
                  const Rec16 rec = *reinterpret_cast<const Rec16*>(cols_bytes + off);  // (a nontemporal load here: twice the time)
                  const uint32_t alt = ((uint32_t)m >> 8) < rec.x ? rec.y : rec.z;
                  rarev |= lvl2 ? alt : (cnt == 255u ? 0x80000000u : 0u);
                  win = lvl2 ? (alt & kmask2) : win;
                  oc = __builtin_amdgcn_alignbit(win, oc, 2);      // (the low base of the window enters at the top)
                  orig = __builtin_amdgcn_alignbit(owin, orig, 2);
                  const uint32_t c2 = (sbits & 3u) << top2;
                  sbits >>= 2;
                  win = (win >> 2) | c2;
                  owin = (owin >> 2) | c2;
                }
              }
              out[0] = s_asc[oc & 0xffu]; out[1] = s_asc[(oc >> 8) & 0xffu]; out[2] = s_asc[(oc >> 16) & 0xffu]; out[3] = s_asc[oc >> 24];
              const uint32_t dx = oc ^ orig;
              n_subst += (uint32_t)__builtin_popcount((dx | (dx >> 1)) & 0x55555555u);
              rare = rare | ((int32_t)rarev < 0);
            } else
#pragma nounroll
            for (uint32_t t8 = 0; t8 < 16u; t8 += 8u) {
#pragma unroll
            for (uint32_t tt = 0; tt < 8u; tt++) {
              const uint32_t t = t8 + tt;
              // (the words of this step are there whatever the refill below does: after the last one at least 17 were
              // ready and 8 steps take 16)
              const uint32_t cnt = s_cnt8[win];
              const uint32_t zn = s_zone[cnt];
              const uint32_t w1 = row[(wpos & 31u) * RS], w2 = row[((wpos + 1u) & 31u) * RS];
              const uint64_t m = (uint64_t)w1 * cnt;
              const bool hit = cnt - 1u < 254u;
              // every lane loads (a lane that is not on a k-mer of the model: column 0 of its row, not used): a 32-bit
              // byte offset from the table's base, so the address is one multiply-add away from the draw
              const uint32_t c16 = hit ? (uint32_t)(m >> 32) << 4 : 0u;
              const Rec16 rec = *reinterpret_cast<const Rec16*>(cols_bytes + (__umul24(win, stride16) + c16));
              if (tt == 0u) {
                // The refill of the lanes' next block goes HERE, behind the first step's column load: its 600
                // instructions run while that load — the longest link of the step's chain — is in flight (round 4;
                // in front of the step it stood between two chains and overlapped nothing).
                asm volatile("" ::: "memory");  // (the load stays in front of the refill; nothing waits for it here)
                const bool need = have < (wpos >> 4) + 2u;  // then at least 17 words are ready: 8 steps take 16
                if (__any(need)) {
                  if (need) {  // (inline: a call here makes everything that lives across the group callee-saved)
                    uint32_t o[16];
                    chacha12_block(key, (uint64_t)have, o);
                    lds_u32* const dst = row + (have & 1u) * 16u * RS;
#pragma unroll
                    for (uint32_t j = 0; j < 16u; j++) dst[j * RS] = o[j];
                    have++;
                  }
                }
              }
              const float v12 = __uint_as_float((w2 >> 9) | 0x3F800000u);
              const float x = __fadd_rn(__fmul_rn(__fsub_rn(v12, 1.0f), __uint_as_float(rec.w)), 0.0f);
              const uint32_t alt = x < __uint_as_float(rec.x) ? rec.y : rec.z;
              rare = rare | (cnt == 255u) | (hit & (((uint32_t)m > zn) | ((int32_t)alt < 0)));
              win = hit ? (alt & kmask2) : win;
              wpos += hit ? 2u : 0u;
              const uint32_t code = win & 3u;
              {
                const uint32_t chs = __builtin_amdgcn_perm(0u, 0x54474341u, code | 0x0c0c0c00u) << (8u * (tt & 3u));
                if (t8 == 0u) { if (tt < 4u) out[0] |= chs; else out[1] |= chs; }
                else { if (tt < 4u) out[2] |= chs; else out[3] |= chs; }
              }
              n_subst += code != (owin & 3u) ? 1u : 0u;
              const uint32_t c2 = ((s16 >> (2u * t)) & 3u) << top2;
              win = (win >> 2) | c2;
              owin = (owin >> 2) | c2;
            }
            }
            n_acgt += 16u;
#if defined(SIMMR_TEST_SPLICE_REDO)
            rare = rare || ((i0 >> 4) & 1u) != 0u;  // test build: every other group is taken back and walked again
#endif
            todo = rare;
            if (rare) {
              // the words of this group are generated again where they are needed (have = the block wpos lies in)
              win = win_0; owin = owin_0; wpos = wpos_0; n_subst = subst_0; n_acgt -= 16u; have = wpos_0 >> 4;
              out[0] = 0u; out[1] = 0u; out[2] = 0u; out[3] = 0u;
            }
          }
        }
        if (todo) {
        auto general_step = [&](const uint32_t t) {
          const uint32_t i = i0 + t;
          if (!CTR && (t & 7u) == 0u) {
            const bool need = i < n && have < (wpos >> 4) + 2u;
            if (__any(need)) {
              if (need) { refill_words(key, have, row + (have & 1u) * 16u * RS, RS); have++; }
            }
          }
          if (i < n) {
            // three_bit_encode_kmer fails on anything but ACGTN (encoding.rs:149-176): a '-' skips the k-mer
            if (i + K <= n && !dead && (!HAS_EXC || dm == 0u)) {
              uint32_t first = 0, cnt = 0, zone = 0;
              uint32_t t24s = 0u;  // CTR: the k-mer's level-1 threshold << 8 (a k-mer with an N has none)
              const Rec16* recs = CTR ? C.kmer_recs_ctr : C.kmer_recs;
              if ((!HAS_EXC || nm == 0u) && FAST) {
                if (CTR) { const uint32_t e32 = s_tab[win]; cnt = e32 & 0xffu; t24s = e32 & 0xffffff00u; }
                else { cnt = s_cnt8[win]; zone = s_zone[cnt]; }
                recs = CTR ? C.kmer_cols_ctr : C.kmer_cols;
                first = win * C.kmer_stride;
                if (cnt == 255u) cnt = 0xFFFFFFFFu;
              } else if (!HAS_EXC || nm == 0u) {
                const Rec16 d = C.kmer_direct[win];
                first = d.x; cnt = d.y; zone = d.z;
                t24s = d.w << 8;
              } else {
                uint32_t key3 = 0;  // the model's code of a k-mer with an N
                for (uint32_t j = 0; j < K; j++)
                  key3 |= (((nm >> j) & 1u) ? 4u : ((win >> (2u * j)) & 3u)) << (3u * j);
                uint32_t h = ((key3 * 0x9E3779B1u) >> 7) & C.kmer_mask;
                Rec16 slot = C.kmer_slots[h];
                while (slot.x != key3 && slot.x != 0xFFFFFFFFu) { h = (h + 1u) & C.kmer_mask; slot = C.kmer_slots[h]; }
                if (slot.x == key3) { first = slot.y; cnt = slot.z; zone = slot.w; }
              }
              if (cnt == 0xFFFFFFFFu) {
                bad_kmer = true; dead = true;  // WeightedAliasIndex::new(..).unwrap() panics
              } else if (cnt != 0u) {
                uint32_t alt;
                if (CTR) {
                  uint32_t gw[4];
                  philox4x32_10(i >> 2, 2u, pk0, pk1, gw);
                  const uint32_t X = (i & 2u) ? ((i & 1u) ? gw[3] : gw[2]) : ((i & 1u) ? gw[1] : gw[0]);
                  if (X < t24s) {
                    alt = win;  // level 1: the k-mer stays what it is
                  } else {
                    const uint32_t Z = (X - t24s) << __builtin_clz(~(t24s | 0xffu));
                    const uint64_t m = (uint64_t)Z * cnt;
                    const Rec16 rec = recs[first + (uint32_t)(m >> 32)];
                    alt = ((uint32_t)m >> 8) < rec.x ? rec.y : rec.z;
                  }
                } else {
                  uint32_t c;
                  for (;;) {  // uniform_index.sample
                    const uint64_t m = (uint64_t)next_word() * cnt;
                    if ((uint32_t)m <= zone) { c = (uint32_t)(m >> 32); break; }
                  }
                  const Rec16 rec = recs[first + c];
                  const float v12 = __uint_as_float((next_word() >> 9) | 0x3F800000u);
                  const float x = __fadd_rn(__fmul_rn(__fsub_rn(v12, 1.0f), __uint_as_float(rec.w)), 0.0f);
                  alt = x < __uint_as_float(rec.x) ? rec.y : rec.z;
                }
                if (alt & 0x80000000u) { bad_kmer = true; dead = true; }  // an N = a deletion, 5-7 = decode error
                else { win = alt; nm = 0u; }
              }
            }
            const uint32_t code = win & 3u, ocode = owin & 3u;
            uint32_t ch = __builtin_amdgcn_perm(0u, 0x54474341u, code | 0x0c0c0c00u);  // "ACGT"[code]
            bool differs = code != ocode;
            if (HAS_EXC) {
              const uint32_t en = nm & 1u, ed = dm & 1u, oe = oexc & 1u;
              if (en) ch = 'N';
              if (ed) ch = '-';
              differs = (en | ed) ? false : (oe != 0u || code != ocode);  // an exception is only ever replaced, never created
              n_acgt += oe ? 0u : 1u;
            } else {
              n_acgt++;
            }
            n_subst += differs ? 1u : 0u;
            {
              const uint32_t chs = ch << (8u * (t & 3u));  // (t is the same for every lane)
              if ((t >> 2) == 0u) out[0] |= chs; else if ((t >> 2) == 1u) out[1] |= chs;
              else if ((t >> 2) == 2u) out[2] |= chs; else out[3] |= chs;
            }
            win >>= 2; owin >>= 2;
            if (HAS_EXC) { nm >>= 1; dm >>= 1; oexc >>= 1; }
            if (i + K < n) {
              const uint32_t c2 = (s16 >> (2u * t)) & 3u;
              if (HAS_EXC) {
                const uint32_t ex = (x16 >> t) & 1u;
                const uint32_t cc = ex ? 0u : c2;
                win |= cc << top2; owin |= cc << top2;
                nm |= (ex & ~c2 & 1u) << top1; dm |= (ex & c2 & 1u) << top1;
                oexc |= ex << top1;
              } else {
                win |= c2 << top2; owin |= c2 << top2;
              }
            }
          }
        };
        if (FAST) {  // the cold path of the fast kernel: one copy of the step
#pragma nounroll
          for (uint32_t t = 0; t < 16u; t++) general_step(t);
        } else {
#pragma unroll
          for (uint32_t t = 0; t < 16u; t++) general_step(t);
        }
        }
        if (i0 < n) {
          const uint64_t lo = (uint64_t)out[0] | ((uint64_t)out[1] << 32), hi = (uint64_t)out[2] | ((uint64_t)out[3] << 32);
#if !defined(SIMMR_SPLICE_NO_PAIRS)
          if (FAST && (CTR || !HAS_EXC)) {  // (the reference mode's form for genomes with N runs has no register left: 126 of 128)
            // Two groups per store (round 5): a lane's 16-byte store at any byte address dirties 1.5 32-byte sectors on
            // average, and they went to memory before the lane's next store reached them — 2.96 bytes written per base
            // (profiles/r5/pmc_traffic.json).  An even group now waits in the lane's 16 bytes of LDS (behind the k-mer table)
            // and goes out together with its odd neighbour: 32 contiguous bytes, two sectors.  (The reference mode's form has
            // no LDS left — its word rows take 128 KB — but registers to spare at four waves per SIMD: the group waits there.)
            uint4* const stash = reinterpret_cast<uint4*>(splice_lds + (1u << (2u * K))) + threadIdx.x;
            const bool odd = ((i0 >> 4) & 1u) != 0u, full = i0 + 16u <= n;
            if (!odd && i0 + 16u < n) {  // a full group with a group behind it
              if (CTR) *stash = make_uint4(out[0], out[1], out[2], out[3]);
              else held = make_uint4(out[0], out[1], out[2], out[3]);
            } else {
              if (odd) {
                uint4 pv = held;
                if (CTR) pv = *stash;
                store16(sd + i0 - 16u, (uint64_t)pv.x | ((uint64_t)pv.y << 32), (uint64_t)pv.z | ((uint64_t)pv.w << 32));
              }
              if (full) store16(sd + i0, lo, hi); else store_tail(sd + i0, lo, hi, n - i0);
            }
          } else
#endif
          if (i0 + 16u <= n) {
#if defined(SIMMR_SPLICE_NT)
            __builtin_nontemporal_store(lo, reinterpret_cast<u64_unaligned*>(sd + i0));
            __builtin_nontemporal_store(hi, reinterpret_cast<u64_unaligned*>(sd + i0 + 8u));
#else
            *reinterpret_cast<u64_unaligned*>(sd + i0) = lo;
            *reinterpret_cast<u64_unaligned*>(sd + i0 + 8u) = hi;
#endif
          } else {
            store_tail(sd + i0, lo, hi, n - i0);
          }
        }
      }
    }
  }
  if (bad_kmer) atomicOr(err, SIMMR_ERRBIT_KMER);
  for (int d = 32; d > 0; d >>= 1) {
    n_acgt += __shfl_down(n_acgt, d, 64);
    n_subst += __shfl_down(n_subst, d, 64);
  }
  if ((threadIdx.x & 63u) == 0 && counters) {
    shard_add(counters, SIMMR_CNT_ACGT_BASES, (unsigned long long)n_acgt);
    shard_add(counters, SIMMR_CNT_SUBSTITUTIONS, (unsigned long long)n_subst);
  }
}

// perfect-short has no per-base draws: its counters come from the plan.
extern "C" __global__ void __launch_bounds__(256)
k_count_plan(uint32_t paired, uint64_t n_units, PlanArrays pl, uint32_t const_q, uint32_t acgt_all,
             unsigned long long* __restrict__ counters) {
  __shared__ uint64_t lds4[4];
  uint64_t bases = 0, redrawn = 0, subst = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_units; k += (uint64_t)gridDim.x * 256) {
    bases += (uint64_t)pl.len[k] * (paired ? 2u : 1u);
    const uint32_t f = pl.flags[k];
    redrawn += (f & SIMMR_FLAG_REDRAWN) ? 1 : 0;
    subst += ((f & SIMMR_FLAG_QSEED_SUBST) ? 1 : 0) + ((f & SIMMR_FLAG_MSEED_SUBST) ? 1 : 0);
  }
  uint64_t tb, tr, ts;
  (void)wg_exclusive_scan_u64(bases, lds4, &tb);
  (void)wg_exclusive_scan_u64(redrawn, lds4, &tr);
  (void)wg_exclusive_scan_u64(subst, lds4, &ts);
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) atomicAdd(&counters[SIMMR_CNT_READS], (unsigned long long)(paired ? 2 * n_units : n_units));
    atomicAdd(&counters[SIMMR_CNT_BASES], (unsigned long long)tb);
    if (tr) atomicAdd(&counters[SIMMR_CNT_REDRAWN], (unsigned long long)tr);
    if (ts) atomicAdd(&counters[SIMMR_CNT_SEED_SUBST], (unsigned long long)ts);
    if (acgt_all) atomicAdd(&counters[SIMMR_CNT_ACGT_BASES], (unsigned long long)tb);
    if (const_q) atomicAdd(&counters[SIMMR_CNT_QUAL_SUM], (unsigned long long)(tb * const_q));
  }
}

}  // namespace simmr
