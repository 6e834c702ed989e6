// fastq_kernels.hip — FASTQ framing on the device (gfx950), byte-identical to
// simmr/src/fastq.rs:14-124 for every header template the host could compile
// (see compile_header_format in engine.hip).  Included by engine.hip.
//
// Record of read r (fastq.rs:58-66 / 93-103):
//     header '\n' bases '\n' '+' '\n' qualities '\n'
// Two passes: k_fastq_size gives every record's length (the header length depends
// on the decimal widths of read id / start / end and on the sequence id), an
// exclusive scan gives the record offsets, k_fastq_write fills the buffer.
//
// k_fastq_write: a wavefront takes 32 reads at a time.  Lane l formats the header
// of read l into LDS (template literals come from an LDS copy, ids in 8-byte
// pieces, decimals in 32-bit arithmetic when they fit).  A record is three byte
// runs — header+'\n' (from LDS), bases+"\n+\n", qualities+'\n' (from the SoA
// columns) — cut into unaligned 16-byte windows; the windows of all 32 records are
// dealt to the 64 lanes (prefix of window counts in LDS, branch-free search);
// a lane resolves several windows without a branch (one LDS and one global load
// each, the unused one at a harmless address) so their loads are in flight together.  The last window of a run
// ends exactly at the run's end (it overlaps its neighbour with identical bytes),
// so there are no partial stores; the constant trailer bytes are shifted into it.
#pragma once

namespace simmr {

#include "fastq_format.hpp"


extern "C" __global__ void __launch_bounds__(256)
k_fastq_size(FqLenCoef tp, FqTables tb, FqReads rd, uint64_t n_reads, uint64_t* __restrict__ rec_len,
             uint32_t* __restrict__ err) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_reads) return;
  const FqFields f = fq_fields(rd, r);
  const uint32_t h = fq_header_len(tp, tb, f);
  if (h >= FQ_HMAX) { atomicOr(err, SIMMR_ERRBIT_FASTQ); rec_len[r] = 0; return; }  // header + '\n' must fit the LDS slot
  // the longest header sizes the LDS slots of k_fastq_write (an atomic only when this one is longer than any seen so
  // far: atomics on one address are served one at a time, 1.5 M of them took 15 ms)
  if (h > *(volatile uint32_t*)(err + 1)) atomicMax(err + 1, h);
  const uint64_t L = f.L;
  rec_len[r] = (uint64_t)h + 1u + L + 3u + L + 1u;
}

// The same from the plan (simmr_fastq_plan_direct); hlen[r] = the header's bytes, for the emit kernel that writes into the text.
extern "C" __global__ void __launch_bounds__(256)
k_fastq_size_plan(FqLenCoef tp, FqTables tb, FqPlan pn, uint64_t n_reads, uint64_t* __restrict__ rec_len,
                  uint8_t* __restrict__ hlen, uint32_t* __restrict__ err, unsigned long long* __restrict__ tile_bytes,
                  unsigned long long* __restrict__ wave_bytes) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool on = r < n_reads;
  uint32_t h = 0;
  FqFields f{};
  if (on) { f = fq_fields(pn, r); h = fq_header_len(tp, tb, f); }
  const bool bad = h >= FQ_HMAX;  // (also a genome / contig without a name)
  uint32_t wmax = bad ? 0u : h;  // the longest header sizes the LDS slots of the emit kernel's header phase: one atomic per wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(wmax, d, 64); wmax = o > wmax ? o : wmax; }
  if ((threadIdx.x & 63u) == 0 && wmax > *(volatile uint32_t*)(err + 1)) atomicMax(err + 1, wmax);  // (rarely: see k_fastq_size)
  const uint64_t len = (on && !bad) ? (uint64_t)h + 1u + (uint64_t)f.L + 3u + (uint64_t)f.L + 1u : 0u;
  {  // the record lengths of this wave, added to the sum of their tile of the scan that follows (engine.hip: scan_presummed)
     // — or, for the emit kernel that places its own records (wave_bytes: k_emit_philox<TEXT, COARSE>), left as the bytes
     // of these 64 records: their scan is all that kernel asks for, and no record offset is made per read
    unsigned long long s = len;
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63u) == 0) {
      if (wave_bytes) { if (on) wave_bytes[r >> 6] = s; }  // (ceil(n_reads / 64) entries: idle waves of the last block have none)
      else if (s) atomicAdd(&tile_bytes[((uint64_t)blockIdx.x * 256u) / (SCAN_THREADS * SCAN_ITEMS)], s);
    }
  }
  if (!on) return;
  if (bad) atomicOr(err, SIMMR_ERRBIT_FASTQ);
  hlen[r] = bad ? 0 : (uint8_t)h;
  if (rec_len) rec_len[r] = len;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
typedef const __attribute__((address_space(1))) u32x4_unaligned* global_u128_unaligned_ptr;

// the 16 bytes x >> (8 * t) with the low t bytes of `fill` in the t vacated top bytes, 0 <= t <= 3
SIMMR_DEV u32x4 fq_shift_in(u32x4 x, uint32_t t, uint32_t fill) {
  const uint32_t sh = 8u * t;
  u32x4 y;
  y.x = __builtin_amdgcn_alignbit(x.y, x.x, sh);
  y.y = __builtin_amdgcn_alignbit(x.z, x.y, sh);
  y.z = __builtin_amdgcn_alignbit(x.w, x.z, sh);
  // (fill << (32 - sh)) without a shift by 32 when t == 0
  y.w = (x.w >> sh) | (uint32_t)(((uint64_t)(fill & ((1u << sh) - 1u)) << 32) >> sh);
  return y;
}

struct FqRead {  // what the copy phase needs to know about a record
  uint64_t rec;    // byte offset of the record in the output
  uint64_t so;     // first base / quality in seq[] / qual[]
  uint32_t H, L;   // header length including the '\n'; read length
};

#define FQ_UNROLL 4 /* windows per lane in flight */

extern "C" __global__ void __launch_bounds__(256)
k_fastq_write(const FqTemplate* __restrict__ tp, FqTables tb, FqReads rd, uint64_t n_reads, uint32_t paired, uint32_t lit_bytes,
              uint32_t hpitch, const uint64_t* __restrict__ rec_off, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t hdr_all[];  // [4 waves][FQ_BATCH][hpitch], hpitch = 4 * an odd number (fq_slot_pitch)
  __shared__ __attribute__((aligned(16))) uint8_t lit[FQ_LIT_MAX + 8];
  __shared__ FqRead recs[4][FQ_BATCH];
  __shared__ uint32_t wpre[4][FQ_BATCH + 1];  // first window of each record of the batch
  __shared__ FqSeg segs[FQ_MAX_SEGS];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* hdr = hdr_all + (size_t)wave * FQ_BATCH * hpitch;
  for (uint32_t i = threadIdx.x; i < lit_bytes; i += 256) lit[i] = tb.blob[i];
  const uint32_t n_segs = fq_stage_template(tp, segs);
  __syncthreads();
  const uint64_t n_batches = (n_reads + FQ_BATCH - 1) / FQ_BATCH;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + wave, n_waves = (uint64_t)gridDim.x * 4;
  for (uint64_t batch = wave_id; batch < n_batches; batch += n_waves) {
    const uint64_t r0 = batch * FQ_BATCH;
    const uint32_t nb = (n_reads - r0) < FQ_BATCH ? (uint32_t)(n_reads - r0) : FQ_BATCH;
    // ---- phase 1: lane l formats the header of read r0 + l (fastq.rs:34-56) into LDS
    uint32_t nwin = 0;
    if (lane < nb) {
      const uint64_t r = r0 + lane;
      uint8_t* h = hdr + lane * hpitch;
      const uint64_t so = rd.seq_off[r];
      const FqFields f = fq_fields(rd, r);
      const uint32_t H = fq_format_header(h, 0u, segs, n_segs, tb, lit, f, (paired && (r & 1u)) ? '2' : '1'), L = f.L;
      recs[wave][lane] = FqRead{rec_off[r], so, H, L};
      // windows: runs shorter than 16 bytes are one bytewise "window"
      nwin = (H >= 16u ? (H + 15u) >> 4 : 1u) + (L >= 16u ? ((L + 3u + 15u) >> 4) + ((L + 1u + 15u) >> 4) : 2u);
    }
    uint32_t inc = nwin;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < (int)FQ_BATCH; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc += o;
    }
    if (lane < FQ_BATCH) wpre[wave][lane + 1] = inc;
    if (lane == 0) wpre[wave][0] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS only: s_waitcnt 0 would also wait for the previous batch's stores to be acknowledged)
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = wpre[wave][FQ_BATCH];
    // ---- phase 2: every lane moves windows of the batch's records.  A window of a run of n_src source
    // bytes + k trailer bytes starts at 16 * piece, except the last one, which ends at the end of the run (it
    // overlaps its neighbour with identical bytes: every window is exact, trailer bytes included); source
    // loads never leave the run.  Runs shorter than 16 bytes are one "window" copied bytewise afterwards.
    for (uint32_t win0 = lane; win0 < total; win0 += 64 * FQ_UNROLL) {
      u32x4 v[FQ_UNROLL];
      uint8_t* dst[FQ_UNROLL];
      uint32_t slow[FQ_UNROLL];
#pragma unroll
      for (int u = 0; u < FQ_UNROLL; u++) {
        const uint32_t win = win0 + 64u * u;
        const bool on = win < total;
        uint32_t i = 0;  // record of this window: last i with wpre[i] <= win (lanes past nb hold the total)
#pragma unroll
        for (uint32_t step = FQ_BATCH / 2; step; step >>= 1)
          if (wpre[wave][i + step] <= win) i += step;
        if (!on) i = 0;
        const FqRead R = recs[wave][i];
        const uint32_t k0 = win - wpre[wave][i];
        const uint32_t wA = R.H >= 16u ? (R.H + 15u) >> 4 : 1u;
        const uint32_t npB = R.L >= 16u ? (R.L + 3u + 15u) >> 4 : 1u;
        const bool isA = k0 < wA, isB = !isA && k0 - wA < npB;
        // the run: n_src source bytes, kt trailer bytes, `piece` of `np` windows, destination offset in the record
        const uint32_t n_src = isA ? R.H : R.L, kt = isA ? 0u : (isB ? 3u : 1u);
        const uint32_t fill = isB ? 0x0a2b0au : 0x0au;
        const uint32_t piece = isA ? k0 : (isB ? k0 - wA : k0 - wA - npB);
        const uint32_t np = isA ? wA : (isB ? npB : (R.L >= 16u ? (R.L + 1u + 15u) >> 4 : 1u));
        const uint32_t run_off = isA ? 0u : (isB ? R.H : R.H + R.L + 3u);
        const bool fast = on && n_src >= 16u;
        const uint32_t n = n_src + kt;
        const uint32_t w = fast ? (piece + 1 < np ? piece * 16u : n - 16u) : 0u;
        const bool shifted = fast && w + 16u > n_src;  // the window reaches into the trailer
        const uint32_t t = shifted ? w + 16u - n_src : 0u;
        const uint32_t ld = shifted ? n_src - 16u : w;
        // both loads are always issued (no branch): the one that is not needed reads a harmless address
        const uint8_t* gsrc = (isB ? rd.seq + R.so : rd.qual + (rd.slot16 ? (R.so & ~15ull) : R.so)) + ((fast && !isA) ? ld : 0u);
        const u32x4 vg = *(global_u128_unaligned_ptr)(fast && !isA ? gsrc : rd.seq);
        const u32x4 vl = fq_read16(hdr + i * hpitch, (fast && isA) ? w : 0u);  // (aligned words funnelled: no replay in the LDS)
        v[u] = isA ? vl : fq_shift_in(vg, t, fill);
        dst[u] = fast ? out + R.rec + run_off + w : nullptr;
        slow[u] = (on && !fast) ? (i | (isA ? 0x100u : isB ? 0x200u : 0x400u)) : 0u;
      }
#pragma unroll
      for (int u = 0; u < FQ_UNROLL; u++)
        if (dst[u]) *reinterpret_cast<u32x4_unaligned*>(dst[u]) = v[u];
#pragma unroll
      for (int u = 0; u < FQ_UNROLL; u++) {
        if (!slow[u]) continue;  // a run shorter than 16 bytes
        const uint32_t i = slow[u] & 0xffu;
        const FqRead R = recs[wave][i];
        uint8_t* rec = out + R.rec;
        if (slow[u] & 0x100u) {
          for (uint32_t j = 0; j < R.H; j++) rec[j] = hdr[i * hpitch + j];
        } else if (slow[u] & 0x200u) {
          uint8_t* recB = rec + R.H;
          for (uint32_t j = 0; j < R.L; j++) recB[j] = rd.seq[R.so + j];
          recB[R.L] = '\n'; recB[R.L + 1] = '+'; recB[R.L + 2] = '\n';
        } else {
          uint8_t* recC = rec + R.H + R.L + 3u;
          for (uint32_t j = 0; j < R.L; j++) recC[j] = rd.qual[(rd.slot16 ? (R.so & ~15ull) : R.so) + j];
          recC[R.L] = '\n';
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next batch overwrites the LDS slots
  }
}

}  // namespace simmr
