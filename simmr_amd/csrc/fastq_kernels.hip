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
// dealt to the 64 lanes (prefix of window counts in LDS, branch-free search), so
// every lane has an independent load -> store in flight.  The last window of a run
// ends exactly at the run's end (it overlaps its neighbour with identical bytes),
// so there are no partial stores; the constant trailer bytes are shifted into it.
#pragma once

namespace simmr {

#define FQ_MAX_SEGS 24
#define FQ_HMAX 256u  /* longest header, including the '\n' */
#define FQ_HPITCH 264u /* LDS bytes per header slot: 8-byte copies may run past the end */
#define FQ_LIT_MAX 256u /* template literals kept in LDS (they are part of a header, so < FQ_HMAX) */
#define FQ_BATCH 32u  /* reads per wave iteration */

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
  const uint64_t L = rd.seq_off[r + 1] - rd.seq_off[r];
  rec_len[r] = (uint64_t)h + 1u + L + 3u + L + 1u;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
typedef const __attribute__((address_space(1))) u32x4_unaligned* global_u128_unaligned_ptr;

// the 16 bytes x >> (8 * t) with the low t bytes of `fill` in the t vacated top bytes, 1 <= t <= 3
SIMMR_DEV u32x4 fq_shift_in(u32x4 x, uint32_t t, uint32_t fill) {
  const uint32_t sh = 8u * t;
  u32x4 y;
  y.x = __builtin_amdgcn_alignbit(x.y, x.x, sh);
  y.y = __builtin_amdgcn_alignbit(x.z, x.y, sh);
  y.z = __builtin_amdgcn_alignbit(x.w, x.z, sh);
  y.w = (x.w >> sh) | (fill << (32u - sh));
  return y;
}

// One unaligned 16-byte window of the run src[0, n_src) + k trailer bytes (`fill`, first byte lowest; k <= 3),
// written to dst.  Window `piece` of `n_pieces` starts at 16 * piece, except the last one, which ends at the
// end of the run (so it overlaps its neighbour — with identical bytes: every window is exact, including the
// trailer bytes it covers).  Needs n_src >= 16; source loads never leave src[0, n_src).
SIMMR_DEV void fq_copy_window(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n_src, uint32_t k,
                              uint32_t fill, uint32_t piece, uint32_t n_pieces) {
  const uint32_t n = n_src + k;
  const uint32_t w = piece + 1 < n_pieces ? piece * 16u : n - 16u;
  u32x4 v;
  if (w + 16u <= n_src) {
    v = *(global_u128_unaligned_ptr)(src + w);
  } else {
    const uint32_t t = w + 16u - n_src;  // trailer bytes inside this window
    v = fq_shift_in(*(global_u128_unaligned_ptr)(src + n_src - 16u), t, fill & (0xffffffffu >> (32u - 8u * t)));
  }
  *reinterpret_cast<u32x4_unaligned*>(dst + w) = v;
}

struct FqRead {  // what the copy phase needs to know about a record
  uint64_t rec;    // byte offset of the record in the output
  uint64_t so;     // first base / quality in seq[] / qual[]
  uint32_t H, L;   // header length including the '\n'; read length
};

extern "C" __global__ void __launch_bounds__(256)
k_fastq_write(FqTemplate tp, FqTables tb, FqReads rd, uint64_t n_reads, uint32_t paired, uint32_t lit_bytes,
              const uint64_t* __restrict__ rec_off, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t hdr[4][FQ_BATCH][FQ_HPITCH];
  __shared__ __attribute__((aligned(16))) uint8_t lit[FQ_LIT_MAX + 8];
  __shared__ FqRead recs[4][FQ_BATCH];
  __shared__ uint32_t wpre[4][FQ_BATCH + 1];  // first window of each record of the batch
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
      uint8_t* h = hdr[wave][lane];
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
    // ---- phase 2: every lane moves one window of one record
    for (uint32_t win = lane; win < total; win += 64) {
      uint32_t i = 0;  // record of this window: last i with wpre[i] <= win (lanes past nb hold the total)
#pragma unroll
      for (uint32_t step = FQ_BATCH / 2; step; step >>= 1)
        if (wpre[wave][i + step] <= win) i += step;
      const FqRead R = recs[wave][i];
      uint32_t k = win - wpre[wave][i];
      uint8_t* rec = out + R.rec;
      const uint32_t wA = R.H >= 16u ? (R.H + 15u) >> 4 : 1u;
      if (k < wA) {  // run A: header + '\n' from LDS
        const uint8_t* h = hdr[wave][i];
        if (R.H >= 16u) {
          const uint32_t w = k + 1 < wA ? k * 16u : R.H - 16u;
          *reinterpret_cast<u32x4_unaligned*>(rec + w) = *reinterpret_cast<const u32x4_unaligned*>(h + w);
        } else {
          for (uint32_t j = 0; j < R.H; j++) rec[j] = h[j];
        }
        continue;
      }
      k -= wA;
      uint8_t* recB = rec + R.H;
      uint8_t* recC = recB + R.L + 3u;
      if (R.L >= 16u) {  // runs B and C: bases + "\n+\n", qualities + '\n'
        const uint32_t npB = (R.L + 3u + 15u) >> 4, npC = (R.L + 1u + 15u) >> 4;
        if (k < npB) fq_copy_window(recB, rd.seq + R.so, R.L, 3u, 0x0a2b0au, k, npB);
        else fq_copy_window(recC, rd.qual + R.so, R.L, 1u, 0x0au, k - npB, npC);
      } else if (k == 0) {
        for (uint32_t j = 0; j < R.L; j++) recB[j] = rd.seq[R.so + j];
        recB[R.L] = '\n'; recB[R.L + 1] = '+'; recB[R.L + 2] = '\n';
      } else {
        for (uint32_t j = 0; j < R.L; j++) recC[j] = rd.qual[R.so + j];
        recC[R.L] = '\n';
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next batch overwrites the LDS slots
  }
}

}  // namespace simmr
