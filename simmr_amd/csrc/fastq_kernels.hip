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

#define FQ_MAX_SEGS 24
#define FQ_HMAX 256u  /* longest header, including the '\n' */
#define FQ_LIT_MAX 256u /* template literals kept in LDS (they are part of a header, so < FQ_HMAX) */
#define FQ_BATCH 64u  /* reads per wave iteration */

enum FqKind : uint32_t {
  FQ_LITERAL = 0, FQ_GENOME_ID, FQ_READ_ID, FQ_SEQUENCE_ID, FQ_START, FQ_END, FQ_REVCOMP, FQ_PAIR
};

struct FqSeg { uint32_t kind, off, len; };  // literal: bytes blob[off, off + len)
struct FqTemplate { uint32_t n_segs; FqSeg segs[FQ_MAX_SEGS]; };

struct FqTables {
  const uint8_t* blob;       // template literals first, then genome ids and sequence ids; 8 bytes of padding
  const uint32_t* g_id_off;  // per engine genome slot
  const uint32_t* g_id_len;
  const uint32_t* g_cbase;   // first row of the genome's contigs in c_off / c_len
  const uint32_t* g_ncontig; // 0 for a slot without names
  const uint32_t* c_off;
  const uint32_t* c_len;
  uint32_t n_slots;
};

struct FqReads {  // the SoA columns simmr_*_emit filled (device pointers)
  const uint8_t* seq;
  const uint8_t* qual;
  const uint64_t* seq_off;
  const uint64_t* start;
  const uint64_t* end;
  const uint32_t* contig;
  const uint32_t* genome;
  const uint32_t* read_id;
  const uint8_t* flags;
};

SIMMR_DEV uint32_t dec_digits(uint64_t v) {
  uint32_t n = 1;
  while (v >= 10u) { v /= 10u; n++; }
  return n;
}

// bytes of the header of read r (without the '\n'); 0xffffffff if a table index is out of range
SIMMR_DEV uint32_t fq_header_len(const FqTemplate& tp, const FqTables& tb, const FqReads& rd, uint64_t r) {
  const uint32_t g = rd.genome[r];
  if (g >= tb.n_slots || rd.contig[r] >= tb.g_ncontig[g]) return 0xffffffffu;
  uint32_t n = 0;
  for (uint32_t s = 0; s < tp.n_segs; s++) {
    const FqSeg sg = tp.segs[s];
    switch (sg.kind) {
      case FQ_LITERAL: n += sg.len; break;
      case FQ_GENOME_ID: n += tb.g_id_len[g]; break;
      case FQ_READ_ID: n += dec_digits(rd.read_id[r]); break;
      case FQ_SEQUENCE_ID: n += tb.c_len[tb.g_cbase[g] + rd.contig[r]]; break;
      case FQ_START: n += dec_digits(rd.start[r]); break;
      case FQ_END: n += dec_digits(rd.end[r]); break;
      default: n += 1; break;  // 't' / 'f', '1' / '2'
    }
  }
  return n;
}

SIMMR_DEV uint32_t fq_put_dec(uint8_t* dst, uint32_t at, uint64_t v) {
  if ((v >> 32) == 0) {  // the usual case: no 64-bit division
    uint32_t x = (uint32_t)v, n = 1;
    for (uint32_t t = x; t >= 10u; t /= 10u) n++;
    for (uint32_t i = n; i-- > 0;) { dst[at + i] = (uint8_t)('0' + x % 10u); x /= 10u; }
    return at + n;
  }
  const uint32_t n = dec_digits(v);
  for (uint32_t i = n; i-- > 0;) { dst[at + i] = (uint8_t)('0' + (uint32_t)(v % 10u)); v /= 10u; }
  return at + n;
}
// LDS -> LDS
SIMMR_DEV uint32_t fq_put_bytes(uint8_t* dst, uint32_t at, const uint8_t* src, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) dst[at + i] = src[i];
  return at + n;
}
// device memory -> LDS in 8-byte pieces (the blob is padded; the slot has FQ_HPITCH - FQ_HMAX spare bytes)
SIMMR_DEV uint32_t fq_put_global(uint8_t* dst, uint32_t at, const uint8_t* __restrict__ src, uint32_t n) {
  for (uint32_t i = 0; i < n; i += 8) {
    const uint64_t v = *(global_u64_unaligned_ptr)(src + i);
    *reinterpret_cast<u64_unaligned*>(dst + at + i) = v;
  }
  return at + n;
}

extern "C" __global__ void __launch_bounds__(256)
k_fastq_size(FqTemplate tp, FqTables tb, FqReads rd, uint64_t n_reads, uint64_t* __restrict__ rec_len,
             uint32_t* __restrict__ err) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_reads) return;
  const uint32_t h = fq_header_len(tp, tb, rd, r);
  if (h >= FQ_HMAX) { atomicOr(err, SIMMR_ERRBIT_FASTQ); rec_len[r] = 0; return; }  // header + '\n' must fit the LDS slot
  atomicMax(err + 1, h);  // the longest header sizes the LDS slots of k_fastq_write
  const uint64_t L = rd.seq_off[r + 1] - rd.seq_off[r];
  rec_len[r] = (uint64_t)h + 1u + L + 3u + L + 1u;
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
k_fastq_write(FqTemplate tp, FqTables tb, FqReads rd, uint64_t n_reads, uint32_t paired, uint32_t lit_bytes,
              uint32_t hpitch, const uint64_t* __restrict__ rec_off, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t hdr_all[];  // [4 waves][FQ_BATCH][hpitch], hpitch % 16 == 0
  __shared__ __attribute__((aligned(16))) uint8_t lit[FQ_LIT_MAX + 8];
  __shared__ FqRead recs[4][FQ_BATCH];
  __shared__ uint32_t wpre[4][FQ_BATCH + 1];  // first window of each record of the batch
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* hdr = hdr_all + (size_t)wave * FQ_BATCH * hpitch;
  for (uint32_t i = threadIdx.x; i < lit_bytes; i += 256) lit[i] = tb.blob[i];
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
      const uint32_t g = rd.genome[r];
      const uint32_t row = tb.g_cbase[g] + rd.contig[r];
      const uint32_t gid_off = tb.g_id_off[g], gid_len = tb.g_id_len[g], sid_off = tb.c_off[row], sid_len = tb.c_len[row];
      const uint64_t start = rd.start[r], end = rd.end[r], so = rd.seq_off[r], so1 = rd.seq_off[r + 1];
      const uint32_t id = rd.read_id[r], fl = rd.flags[r];
      uint32_t at = 0;
      for (uint32_t s = 0; s < tp.n_segs; s++) {
        const FqSeg sg = tp.segs[s];
        switch (sg.kind) {
          case FQ_LITERAL: at = fq_put_bytes(h, at, lit + sg.off, sg.len); break;
          case FQ_GENOME_ID: at = fq_put_global(h, at, tb.blob + gid_off, gid_len); break;
          case FQ_READ_ID: at = fq_put_dec(h, at, id); break;
          case FQ_SEQUENCE_ID: at = fq_put_global(h, at, tb.blob + sid_off, sid_len); break;
          case FQ_START: at = fq_put_dec(h, at, start); break;
          case FQ_END: at = fq_put_dec(h, at, end); break;
          case FQ_REVCOMP: h[at++] = (fl & SIMMR_FLAG_REVCOMP) ? 't' : 'f'; break;
          default: h[at++] = (paired && (r & 1u)) ? '2' : '1'; break;  // mates are interleaved
        }
      }
      h[at] = '\n';
      const uint32_t H = at + 1, L = (uint32_t)(so1 - so);
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
    __builtin_amdgcn_s_waitcnt(0);
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
        const uint8_t* gsrc = (isB ? rd.seq : rd.qual) + R.so + ((fast && !isA) ? ld : 0u);
        const u32x4 vg = *(global_u128_unaligned_ptr)(fast && !isA ? gsrc : rd.seq);
        const u32x4 vl = *reinterpret_cast<const u32x4_unaligned*>(hdr + i * hpitch + ((fast && isA) ? w : 0u));
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
          for (uint32_t j = 0; j < R.L; j++) recC[j] = rd.qual[R.so + j];
          recC[R.L] = '\n';
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next batch overwrites the LDS slots
  }
}

}  // namespace simmr
