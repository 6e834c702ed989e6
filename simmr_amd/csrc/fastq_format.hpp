// fastq_format.hpp — the pieces of FASTQ framing (simmr/src/fastq.rs:32-121) that more than one kernel uses: the compiled
// header template, the id tables, what a header can show of a read, and the routines that format one header into an
// LDS slot.  Included by kernels.hip (the counter-mode emit kernel writes headers itself when it emits FASTQ text)
// and by fastq_kernels.hip (framing of emitted columns).  Needs SIMMR_DEV, u64_unaligned, global_u64_unaligned_ptr
// (kernels.hip) and PlanArrays (device_types.hpp).
// Included INSIDE namespace simmr by both.
#pragma once

#define FQ_MAX_SEGS 24
#define FQ_HMAX 256u  /* longest header, including the '\n' */
#define FQ_LIT_MAX 448u /* template literals kept in LDS, each padded to a multiple of 8 bytes (they are part of a header: < FQ_HMAX + 7 * FQ_MAX_SEGS) */
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

// A header's length without a walk over the template: the template's constant bytes and how often each field occurs
// (the host fills this from the compiled template, engine.hip: fq_len_coef) — the sizing kernels were latency chains of
// one scalar load per template piece and wave (1.3 ms per 100 M reads, 7 us per wave)
struct FqLenCoef { uint32_t h0, n_gid, n_rid, n_sid, n_start, n_end; };
SIMMR_DEV uint32_t fq_header_len(const FqLenCoef& c, const FqTables& tb, const FqFields& f) {
  const uint32_t g = f.genome;
  if (g >= tb.n_slots || f.contig >= tb.g_ncontig[g]) return 0xffffffffu;
  uint32_t n = c.h0;
  if (c.n_gid) n += c.n_gid * tb.g_id_len[g];
  if (c.n_sid) n += c.n_sid * tb.c_len[tb.g_cbase[g] + f.contig];
  if (c.n_rid) n += c.n_rid * dec_digits(f.read_id);
  if (c.n_start) n += c.n_start * dec_digits(f.start);
  if (c.n_end) n += c.n_end * dec_digits(f.end);
  return n;
}

// ---- a header into an LDS slot, through ALIGNED 32-bit stores ------------------------------------------------------
// A DS access off its natural alignment is replayed at 64 cycles per wave-instruction (cdna_hip_programming.md,
// guideline 17): the first form of these routines stored a header's pieces as 8-byte LDS writes at whatever byte the
// header had reached — sixteen of them per header — and the TEXT form of the emit kernel spent 3.0e9 of its 6.2e9
// LDS-array cycles per launch in that replay (SQ_LDS_UNALIGNED_STALL, profiles/r4/lds_probe.log).  The writer below
// keeps the bytes of the unfinished word in a register and stores whole words at 4-byte-aligned addresses only (slot
// bases are 4-byte aligned: fq_slot_pitch is an odd number of words, which also keeps the lanes of a wave on different
// banks).  Every piece rewrites the word it continues, so the slot is complete after the last piece, and it writes up
// to two words past its bytes (zeros, or the next piece's start): every slot has that slack.
struct FqW {
  uint32_t* w;   // the slot, as words
  uint32_t at;   // bytes written
  uint32_t acc;  // the bytes of the word at (at & ~3): its low (at & 3) bytes are the header's, the rest 0
};
SIMMR_DEV FqW fq_begin(uint8_t* slot) { return FqW{reinterpret_cast<uint32_t*>(slot), 0u, 0u}; }
// the low n (1..8) bytes of v
SIMMR_DEV void fq_put8(FqW& o, uint64_t v, uint32_t n) {
  const uint32_t f = o.at & 3u, sh = 8u * f;
  if (n < 8u) v &= (1ull << (8u * n)) - 1ull;
  const uint64_t t = v << sh;
  const uint32_t w0 = o.acc | (uint32_t)t, w1 = (uint32_t)(t >> 32);
  const uint32_t w2 = f ? ((uint32_t)(v >> 32) >> (32u - sh)) : 0u;
  uint32_t* p = o.w + (o.at >> 2);
  p[0] = w0; p[1] = w1; p[2] = w2;
  const uint32_t k = (f + n) >> 2;
  o.acc = k == 0u ? w0 : (k == 1u ? w1 : w2);
  o.at += n;
  asm volatile("" ::: "memory");  // piece by piece: interleaving the pieces of a header buys nothing and costs the kernel its registers
}
SIMMR_DEV void fq_put1(FqW& o, uint32_t c) {
  const uint32_t f = o.at & 3u;
  o.acc |= c << (8u * f);
  o.w[o.at >> 2] = o.acc;
  o.at++;
  if (f == 3u) o.acc = 0u;
}

// Decimal digits of v.  Below 10^8 — every position of a genome under 100 Mbp, most read ids — the eight digits are made
// in registers and go out as one piece with the leading zeros shifted off.
SIMMR_DEV uint32_t fq_four_digits(uint32_t y) {  // y < 10000 -> its four digits as bytes, most significant first in memory
  const uint32_t a = y / 100u, b = y - a * 100u;
  const uint32_t a1 = a / 10u, a0 = a - a1 * 10u, b1 = b / 10u, b0 = b - b1 * 10u;
  return a1 | (a0 << 8) | (b1 << 16) | (b0 << 24);
}
SIMMR_DEV uint64_t fq_eight_digits(uint32_t x) {  // x < 10^8 -> its eight digits as ASCII bytes, most significant first in memory
  const uint32_t hi = x / 10000u, lo = x - hi * 10000u;
  return ((uint64_t)fq_four_digits(hi) | ((uint64_t)fq_four_digits(lo) << 32)) + 0x3030303030303030ull;
}
SIMMR_DEV void fq_put_dec(FqW& o, uint64_t v) {
  if (v < 100000000ull) {
    const uint32_t x = (uint32_t)v;
    const uint32_t n = dec_digits(x);
    fq_put8(o, fq_eight_digits(x) >> (8u * (8u - n)), n);
    return;
  }
  if ((v >> 32) == 0) {  // below 2^32 (read ids of a run of a billion reads): one or two digits, then eight
    const uint32_t x = (uint32_t)v, hi = x / 100000000u, lo = x - hi * 100000000u;  // hi = 1..42
    const uint32_t h1 = hi / 10u, h0 = hi - h1 * 10u;
    fq_put8(o, h1 ? (uint64_t)(('0' + h1) | (('0' + h0) << 8)) : (uint64_t)('0' + h0), h1 ? 2u : 1u);
    fq_put8(o, fq_eight_digits(lo), 8u);
    return;
  }
  // positions beyond 4 Gbases: v = c * 10^16 + d * 10^8 + b (divisions by constants only: a division by a run-time
  // divisor is a hundred instructions and thirty registers, inlined once per field)
  const uint64_t a = v / 100000000ull;
  const uint32_t b = (uint32_t)(v - a * 100000000ull);
  const uint32_t c = (uint32_t)(a / 100000000ull), d = (uint32_t)(a - (uint64_t)c * 100000000ull);  // c < 1845
  if (c) {
    const uint32_t n = dec_digits(c);
    fq_put8(o, (uint64_t)((fq_four_digits(c) + 0x30303030u) >> (8u * (4u - n))), n);
    fq_put8(o, fq_eight_digits(d), 8u);
  } else {
    const uint32_t n = dec_digits(d);
    fq_put8(o, fq_eight_digits(d) >> (8u * (8u - n)), n);
  }
  fq_put8(o, fq_eight_digits(b), 8u);
}
// a template literal, LDS -> LDS, eight bytes at a time (src is the same for every lane — a broadcast read — and starts
// on an 8-byte boundary: compile_header_format pads the literals to that; the buffer has eight spare bytes)
SIMMR_DEV void fq_put_bytes(FqW& o, const uint8_t* src, uint32_t n) {
  for (uint32_t i = 0; i < n; i += 8u) fq_put8(o, *reinterpret_cast<const uint64_t*>(src + i), n - i < 8u ? n - i : 8u);
}
// an id, device memory -> LDS in 8-byte pieces (the blob is padded).  The first 48 bytes are fetched before the first is
// stored: one memory latency for an id, not one per piece (a header has two ids).
SIMMR_DEV void fq_put_global(FqW& o, const uint8_t* __restrict__ src, uint32_t n) {
  uint64_t v[6];
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) v[k] = (8u * k < n) ? *(global_u64_unaligned_ptr)(src + 8u * k) : 0ull;
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) if (8u * k < n) fq_put8(o, v[k], n - 8u * k < 8u ? n - 8u * k : 8u);
  for (uint32_t i = 48; i < n; i += 8) fq_put8(o, *(global_u64_unaligned_ptr)(src + i), n - i < 8u ? n - i : 8u);
}

// the header of a read and its '\n' into an LDS slot (fastq.rs:34-56); returns the bytes written.  `lead`: the run starts
// with the '\n' that ends the record before (the TEXT form of the emit kernel).
// `segs` / `n_segs`: the template's pieces, staged in LDS by the caller (fq_stage_template)
// `at0`: the header starts at byte at0 (< 8) of the slot, whose words up to there are zero (text_lines.hip: slots congruent
// to the text modulo 8); the value returned then counts from the slot's first byte as well.
SIMMR_DEV uint32_t fq_format_header(uint8_t* slot, uint32_t lead, const FqSeg* segs, uint32_t n_segs, const FqTables& tb,
                                    const uint8_t* lit, const FqFields& f, uint32_t pair_char, uint32_t at0 = 0u) {
  FqW o = fq_begin(slot);
  o.at = at0;
  if (lead) fq_put1(o, '\n');
  const uint32_t g = f.genome;
  const uint32_t row = tb.g_cbase[g] + f.contig;
  const uint32_t gid_off = tb.g_id_off[g], gid_len = tb.g_id_len[g], sid_off = tb.c_off[row], sid_len = tb.c_len[row];
#if defined(SIMMR_ABLATE_HALF_SEGS)  /* timing only: what formatting costs if a lane wrote half of a header (wrong text, same places) */
  n_segs = (n_segs + 1u) / 2u;
#endif
  for (uint32_t s = 0; s < n_segs; s++) {
    const FqSeg sg = segs[s];
    // (one copy of each routine: the two ids share one, the three numbers one, the two letters one — inlined per field
    // they made the header code three times as long and pushed the emit kernel into scratch)
    if (sg.kind == FQ_LITERAL) {
      fq_put_bytes(o, lit + sg.off, sg.len);
    } else if (sg.kind == FQ_GENOME_ID || sg.kind == FQ_SEQUENCE_ID) {
      const bool gid = sg.kind == FQ_GENOME_ID;
      fq_put_global(o, tb.blob + (gid ? gid_off : sid_off), gid ? gid_len : sid_len);
    } else if (sg.kind == FQ_READ_ID || sg.kind == FQ_START || sg.kind == FQ_END) {
      fq_put_dec(o, sg.kind == FQ_READ_ID ? (uint64_t)f.read_id : (sg.kind == FQ_START ? f.start : f.end));
    } else {
      fq_put1(o, sg.kind == FQ_REVCOMP ? ((f.flags & SIMMR_FLAG_REVCOMP) ? 't' : 'f') : pair_char);  // mates are interleaved
    }
  }
  fq_put1(o, '\n');
  return o.at;
}

// The same with the two ids' places looked up and their first 48 bytes fetched beforehand (text_lines.hip: the four dependent
// global loads of the routine above — contig row, id offsets, then each id's bytes where the template has it — cost a
// batch of headers more time than its thousand instructions; there the rows are read in the block's prologue and the bytes
// of both ids are requested together before the first piece is written).
struct FqIds { uint32_t gid_off, gid_len, sid_off, sid_len; };
SIMMR_DEV FqIds fq_ids(const FqTables& tb, uint32_t genome, uint32_t contig) {
  const uint32_t row = tb.g_cbase[genome] + contig;
  return FqIds{tb.g_id_off[genome], tb.g_id_len[genome], tb.c_off[row], tb.c_len[row]};
}
SIMMR_DEV void fq_fetch_id(const uint8_t* __restrict__ blob, uint32_t off, uint32_t n, uint64_t v[6]) {
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) v[k] = (8u * k < n) ? *(global_u64_unaligned_ptr)(blob + off + 8u * k) : 0ull;
}
SIMMR_DEV void fq_put_fetched(FqW& o, const uint64_t v[6], const uint8_t* __restrict__ src, uint32_t n) {
#pragma unroll
  for (uint32_t k = 0; k < 6; k++) if (8u * k < n) fq_put8(o, v[k], n - 8u * k < 8u ? n - 8u * k : 8u);
  for (uint32_t i = 48; i < n; i += 8) fq_put8(o, *(global_u64_unaligned_ptr)(src + i), n - i < 8u ? n - i : 8u);
}
SIMMR_DEV uint32_t fq_format_header_fetched(uint8_t* slot, const FqSeg* segs, uint32_t n_segs, const FqTables& tb, const uint8_t* lit,
                                            const FqFields& f, uint32_t pair_char, uint32_t at0, const FqIds& ids,
                                            const uint64_t gidb[6], const uint64_t sidb[6]) {
  FqW o = fq_begin(slot);
  o.at = at0;
  for (uint32_t s = 0; s < n_segs; s++) {
    const FqSeg sg = segs[s];
    if (sg.kind == FQ_LITERAL) {
      fq_put_bytes(o, lit + sg.off, sg.len);
    } else if (sg.kind == FQ_GENOME_ID || sg.kind == FQ_SEQUENCE_ID) {
      const bool gid = sg.kind == FQ_GENOME_ID;
      uint64_t v[6];
#pragma unroll
      for (uint32_t k = 0; k < 6; k++) v[k] = gid ? gidb[k] : sidb[k];
      fq_put_fetched(o, v, tb.blob + (gid ? ids.gid_off : ids.sid_off), gid ? ids.gid_len : ids.sid_len);
    } else if (sg.kind == FQ_READ_ID || sg.kind == FQ_START || sg.kind == FQ_END) {
      fq_put_dec(o, sg.kind == FQ_READ_ID ? (uint64_t)f.read_id : (sg.kind == FQ_START ? f.start : f.end));
    } else {
      fq_put1(o, sg.kind == FQ_REVCOMP ? ((f.flags & SIMMR_FLAG_REVCOMP) ? 't' : 'f') : pair_char);
    }
  }
  fq_put1(o, '\n');
  return o.at;
}

// 16 bytes at byte `b` of an LDS slot (4-byte-aligned base) as five aligned words funnelled together: an unaligned
// ds_read_b128 is replayed at 64 cycles per wave-instruction
SIMMR_DEV v4u32 fq_read16(const uint8_t* slot, uint32_t b) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(slot) + (b >> 2);
  const uint32_t x0 = w[0], x1 = w[1], x2 = w[2], x3 = w[3], x4 = w[4];
  const uint32_t sh = b & 3u;  // v_alignbyte_b32: (hi:lo) >> 8 * sh
  v4u32 v;
  v.x = __builtin_amdgcn_alignbyte(x1, x0, sh); v.y = __builtin_amdgcn_alignbyte(x2, x1, sh);
  v.z = __builtin_amdgcn_alignbyte(x3, x2, sh); v.w = __builtin_amdgcn_alignbyte(x4, x3, sh);
  return v;
}

// the template's pieces from device memory into LDS, once per workgroup (a scalar load per piece and header was a chain of
// fourteen memory latencies per batch of 64 headers)
SIMMR_DEV uint32_t fq_stage_template(const FqTemplate* __restrict__ tp, FqSeg* segs) {
  const uint32_t n = tp->n_segs < FQ_MAX_SEGS ? tp->n_segs : FQ_MAX_SEGS;
  if (threadIdx.x < n) segs[threadIdx.x] = tp->segs[threadIdx.x];
  return n;
}
