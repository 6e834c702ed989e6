// fastq_format.hpp — the pieces of FASTQ framing (simmr/src/fastq.rs:32-121) that more than one kernel uses: the compiled
// header template, the id tables, what a header can show of a read, and the routines that format one header into an
// LDS slot.  Included by kernels.hip (the counter-mode emit kernel writes headers itself when it emits FASTQ text)
// and by fastq_kernels.hip (framing of emitted columns).  Needs SIMMR_DEV, u64_unaligned, global_u64_unaligned_ptr
// (kernels.hip) and PlanArrays (device_types.hpp).
// Included INSIDE namespace simmr by both.
#pragma once

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
  uint32_t slot16;  // SIMMR_SLOT16 columns: a read's length is |end - start|, its qualities start at seq_off & ~15
};

// What a header can show of a read (fastq.rs:34-56), from the emitted columns or straight from the plan.
struct FqFields {
  uint64_t start, end;  // ReadMetadata.start_pos / end_pos
  uint32_t genome, contig, read_id, flags;
  uint32_t L;           // bases
};
SIMMR_DEV FqFields fq_fields(const FqReads& rd, uint64_t r) {
  const uint64_t a = rd.start[r], b = rd.end[r];
  return FqFields{a, b, rd.genome[r], rd.contig[r], rd.read_id[r], rd.flags[r],
                  rd.slot16 ? (uint32_t)(a < b ? b - a : a - b) : (uint32_t)(rd.seq_off[r + 1] - rd.seq_off[r])};
}
// The plan of the shard about to be emitted (simmr_fastq_plan_direct): the same values simmr_*_emit would write into
// the columns (k_write_meta / the emit kernels' prologues; simulate.rs:274,289-296,515-516).
struct FqPlan {
  PlanArrays pl;
  const uint32_t* u_contig;
  const uint32_t* u_genome;  // null: genome_const
  uint64_t first_unit;
  uint32_t read_id_base, genome_const, paired;
};
SIMMR_DEV FqFields fq_fields(const FqPlan& p, uint64_t r) {
  const uint64_t u = p.paired ? (r >> 1) : r;
  const uint32_t rev = p.paired ? (uint32_t)(r & 1u) : 0u;
  const uint32_t L = p.pl.len[u];
  FqFields f;
  if (p.paired) {
    const uint64_t pos = rev ? p.pl.b[u] : p.pl.a[u];
    f.start = rev ? pos + L : pos;
    f.end = rev ? pos : pos + L;
    f.flags = rev ? p.pl.flags[u] : 0u;
  } else {
    f.start = p.pl.a[u];
    f.end = p.pl.b[u];
    f.flags = p.pl.flags[u];
  }
  f.genome = p.u_genome ? p.u_genome[u] : p.genome_const;
  f.contig = p.u_contig[u];
  f.read_id = p.read_id_base + (uint32_t)(p.first_unit + u);
  f.L = L;
  return f;
}

SIMMR_DEV uint32_t dec_digits(uint64_t v) {
  if ((v >> 32) == 0) {  // the usual case: nine compares, no division
    const uint32_t x = (uint32_t)v;
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
           (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
  }
  uint32_t n = 1;
  while (v >= 10u) { v /= 10u; n++; }
  return n;
}

// bytes of the header of read r (without the '\n'); 0xffffffff if a table index is out of range
// (the template is read through a pointer to device memory: indexing a by-value kernel argument with a loop counter
// makes every thread copy the whole struct to scratch first — 58 GB of traffic per 100 M reads, 17 ms, measured)
SIMMR_DEV uint32_t fq_header_len(const FqTemplate* __restrict__ tp, const FqTables& tb, const FqFields& f) {
  const uint32_t g = f.genome;
  if (g >= tb.n_slots || f.contig >= tb.g_ncontig[g]) return 0xffffffffu;
  uint32_t n = 0;
  const uint32_t n_segs = tp->n_segs;
  for (uint32_t s = 0; s < n_segs; s++) {
    const FqSeg sg = tp->segs[s];
    switch (sg.kind) {
      case FQ_LITERAL: n += sg.len; break;
      case FQ_GENOME_ID: n += tb.g_id_len[g]; break;
      case FQ_READ_ID: n += dec_digits(f.read_id); break;
      case FQ_SEQUENCE_ID: n += tb.c_len[tb.g_cbase[g] + f.contig]; break;
      case FQ_START: n += dec_digits(f.start); break;
      case FQ_END: n += dec_digits(f.end); break;
      default: n += 1; break;  // 't' / 'f', '1' / '2'
    }
  }
  return n;
}

// Decimal digits of v at dst[at ...] (LDS).  Below 10^8 — every position of a genome under 100 Mbp, most read ids — the
// eight digits are made in registers and go out as ONE 8-byte store with the leading zeros shifted off (the bytes
// behind the number are overwritten by the next piece of the header; every slot has that much slack).
SIMMR_DEV uint32_t fq_four_digits(uint32_t y) {  // y < 10000 -> its four digits as bytes, most significant first in memory
  const uint32_t a = y / 100u, b = y - a * 100u;
  const uint32_t a1 = a / 10u, a0 = a - a1 * 10u, b1 = b / 10u, b0 = b - b1 * 10u;
  return a1 | (a0 << 8) | (b1 << 16) | (b0 << 24);
}
SIMMR_DEV uint32_t fq_put_dec(uint8_t* dst, uint32_t at, uint64_t v) {
  if (v < 100000000ull) {
    const uint32_t x = (uint32_t)v;
    const uint32_t hi = x / 10000u, lo = x - hi * 10000u;
    uint64_t p = ((uint64_t)fq_four_digits(hi) | ((uint64_t)fq_four_digits(lo) << 32)) + 0x3030303030303030ull;
    const uint32_t n = dec_digits(x);
    p >>= 8u * (8u - n);
    *reinterpret_cast<u64_unaligned*>(dst + at) = p;
    return at + n;
  }
  if ((v >> 32) == 0) {  // no 64-bit division
    uint32_t x = (uint32_t)v, n = 1;
    for (uint32_t t = x; t >= 10u; t /= 10u) n++;
    for (uint32_t i = n; i-- > 0;) { dst[at + i] = (uint8_t)('0' + x % 10u); x /= 10u; }
    return at + n;
  }
  const uint32_t n = dec_digits(v);
  for (uint32_t i = n; i-- > 0;) { dst[at + i] = (uint8_t)('0' + (uint32_t)(v % 10u)); v /= 10u; }
  return at + n;
}
// LDS -> LDS, eight bytes at a time (src is the same for every lane: one broadcast read per piece; both buffers have
// eight spare bytes, and what is written past n is overwritten by the header's next piece)
SIMMR_DEV uint32_t fq_put_bytes(uint8_t* dst, uint32_t at, const uint8_t* src, uint32_t n) {
  for (uint32_t i = 0; i < n; i += 8) *reinterpret_cast<u64_unaligned*>(dst + at + i) = *reinterpret_cast<const u64_unaligned*>(src + i);
  return at + n;
}
// device memory -> LDS in 8-byte pieces (the blob is padded; the slot has spare bytes behind the longest header).
// The first 48 bytes are fetched before the first is stored: one memory latency for an id, not one per piece (a
// header has two ids; piece by piece they were a third of the header kernel's time).
SIMMR_DEV uint32_t fq_put_global(uint8_t* dst, uint32_t at, const uint8_t* __restrict__ src, uint32_t n) {
  uint64_t v[6];
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) v[k] = (8u * k < n) ? *(global_u64_unaligned_ptr)(src + 8u * k) : 0ull;
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) if (8u * k < n) *reinterpret_cast<u64_unaligned*>(dst + at + 8u * k) = v[k];
  for (uint32_t i = 48; i < n; i += 8) {
    const uint64_t w = *(global_u64_unaligned_ptr)(src + i);
    *reinterpret_cast<u64_unaligned*>(dst + at + i) = w;
  }
  return at + n;
}

#if defined(FQH_ABLATE_FORMAT)
SIMMR_DEV uint32_t fq_header_len_lds(const FqSeg* segs, uint32_t n_segs, const FqTables& tb, const FqFields& f) {
  const uint32_t g = f.genome;
  uint32_t n = 0;
  for (uint32_t s = 0; s < n_segs; s++) {
    const FqSeg sg = segs[s];
    switch (sg.kind) {
      case FQ_LITERAL: n += sg.len; break;
      case FQ_GENOME_ID: n += tb.g_id_len[g]; break;
      case FQ_READ_ID: n += dec_digits(f.read_id); break;
      case FQ_SEQUENCE_ID: n += tb.c_len[tb.g_cbase[g] + f.contig]; break;
      case FQ_START: n += dec_digits(f.start); break;
      case FQ_END: n += dec_digits(f.end); break;
      default: n += 1; break;
    }
  }
  return n;
}
#endif

// the header of a read into an LDS slot at h[at ...] (fastq.rs:34-56); returns the position behind it
// `segs` / `n_segs`: the template's pieces, staged in LDS by the caller (fq_stage_template)
SIMMR_DEV uint32_t fq_format_header(uint8_t* h, uint32_t at, const FqSeg* segs, uint32_t n_segs, const FqTables& tb, const uint8_t* lit,
                                    const FqFields& f, uint8_t pair_char) {
  const uint32_t g = f.genome;
  const uint32_t row = tb.g_cbase[g] + f.contig;
  const uint32_t gid_off = tb.g_id_off[g], gid_len = tb.g_id_len[g], sid_off = tb.c_off[row], sid_len = tb.c_len[row];
  for (uint32_t s = 0; s < n_segs; s++) {
    const FqSeg sg = segs[s];
    switch (sg.kind) {
      case FQ_LITERAL: at = fq_put_bytes(h, at, lit + sg.off, sg.len); break;
      case FQ_GENOME_ID: at = fq_put_global(h, at, tb.blob + gid_off, gid_len); break;
      case FQ_READ_ID: at = fq_put_dec(h, at, f.read_id); break;
      case FQ_SEQUENCE_ID: at = fq_put_global(h, at, tb.blob + sid_off, sid_len); break;
      case FQ_START: at = fq_put_dec(h, at, f.start); break;
      case FQ_END: at = fq_put_dec(h, at, f.end); break;
      case FQ_REVCOMP: h[at++] = (f.flags & SIMMR_FLAG_REVCOMP) ? 't' : 'f'; break;
      default: h[at++] = pair_char; break;  // mates are interleaved
    }
  }
  return at;
}

// the template's pieces from device memory into LDS, once per workgroup (a scalar load per piece and header was a chain of
// fourteen memory latencies per batch of 64 headers)
SIMMR_DEV uint32_t fq_stage_template(const FqTemplate* __restrict__ tp, FqSeg* segs) {
  const uint32_t n = tp->n_segs < FQ_MAX_SEGS ? tp->n_segs : FQ_MAX_SEGS;
  if (threadIdx.x < n) segs[threadIdx.x] = tp->segs[threadIdx.x];
  return n;
}
