// text_lines.hip — FASTQ text straight from a paired plan (simmr_emit_fastq; fastq.rs:32-121, main.rs:201-206),
// written to device memory in WHOLE 64-byte lines.  Included by engine.hip after kernels.hip.
//
// Why a second form of the TEXT kernel.  k_emit_philox<.., TEXT> stores each item's sixteen qualities and sixteen bases
// where the text has them — any byte address — and copies the headers out in 16-byte windows: 15.6 L2 requests per read
// where 6.4 whole lines would carry its 410 bytes (profiles/r3/write_path_probe/), and the kernel ran at the request
// rate of the L2 (~8e10 per second), not at its instruction stream's (LAB.md, round 4 item 6).  The address unit takes a
// 16-byte-per-lane store four lanes at a time, so only a lane <-> ALIGNED-chunk assignment over the text itself turns a
// wave's store into sixteen whole lines.  Here that assignment is made in LDS:
//
//   * a block is 128 pairs = 256 reads, as in k_emit_philox (same prologue: one thread per read writes its record);
//   * headers are formatted 128 at a time by the threads that hold them, into LDS slots whose byte offset is congruent
//     to the header's place in the text modulo 8 (so a slot moves into the text image as aligned 8-byte pieces);
//   * the 128 reads of a phase are dealt to the four waves, 32 consecutive reads = one contiguous piece of text each
//     (a SEGMENT).  A wave owns a ring of TL_RING bytes of LDS that is the image of its segment's text between what it
//     has flushed and what its items have reached: ring offset = text address modulo TL_RING, so a ring chunk is an
//     aligned 16-byte chunk of device memory;
//   * a ROUND is the next 64 items (16 bases of one read each) of the segment, one per lane — items, not whole reads, so
//     every lane is busy whatever the read length.  An item ORs its sixteen qualities and sixteen bases into the ring at
//     their text positions: the bytes are shifted to the word grid with v_perm_b32 and go out as five ds_or_b32 per
//     line (the ring is zero wherever nothing has been written, and pieces that share a word add up whatever their
//     order; no lane needs its neighbour's bytes);
//   * the reads whose first item fell into the round get their header ('@...' + '\n', from the slots), their "\n+\n" and
//     their closing '\n' ORed in by task lanes (a handful of 8-byte pieces per read);
//   * then the wave flushes every whole line below the first byte that is still to come: lane l reads chunk l of the
//     run (ds_read_b128), zeroes it, and stores it nontemporally — one store instruction = one aligned kilobyte.
//   Only the first and the last chunk of a segment can be shared with a neighbour (another wave's or workgroup's text):
//   those two are stored bytewise.  Nothing waits for another wave between the two barriers of a header phase.
//
// Byte-identical to the item form by construction of the same draws (the item code below is k_emit_philox's), and
// tested so: tests/test_gpu_text_lines.py compares the two forms byte for byte (read lengths 1-256, header shapes, N runs,
// several genomes, perfect-short, exact-capacity destinations), tests/test_gpu_fullsize.py at 100 M reads.
// MEASURED SLOWER than the item form (24.0 against 19.0-20.5 ms per 100 M reads: 7.6 instead of 15.6 L2 requests per read,
// but 37 % more instructions on a path that is bound by instruction issue; LAB.md round 5): the
// library runs it only when an engine is made with SIMMR_TEXT_FORM=2.
// Covers: paired plans of the counter modes and perfect-short with every read <= TL_MAXL bases (the plan kernel notes
// longer ones: SIMMR_NOTEBIT_LONGREAD) into a 16-byte-aligned buffer.  Everything else keeps the item form.
#pragma once

namespace simmr {

#define TL_RING 4096u      /* bytes of text image per wave (power of two) */
#define TL_GUARD 32u       /* an item's five words may run past the ring's end: they land here and are merged into chunks 0 and 1 */
#ifndef TL_SEG_READS
#define TL_SEG_READS 32u   /* reads of a segment */
#endif
#define TL_GROUP (4u * TL_SEG_READS) /* headers formatted at a time = 4 waves x TL_SEG_READS */
#define TL_MAXL 256u       /* longest read (<= 16 items per read: the segment's item map has TL_SEG_READS * 16 bytes) */
#define TL_MAP (TL_SEG_READS * 16u)
static_assert(TL_MAXL == LONGREAD_MAXL, "the plan kernel's note bit is this kernel's precondition");

// -DTL_DIAG (measurement build): time per section of the kernel, summed over waves, in s_memtime ticks (10 ns on gfx9):
// TL_T(k) adds the time since the previous stamp to section k; engine.hip prints the table after the launch.
#if defined(TL_DIAG)
__device__ unsigned long long tl_diag[16];
#define TL_T(k) do { const uint64_t t_now = __builtin_readcyclecounter(); tl_acc[k] += (uint32_t)(t_now - tl_t0); tl_t0 = t_now; } while (0)
#else
#define TL_T(k) do { } while (0)
#endif

struct alignas(16) TlRec {
  uint32_t k0, k1;  // Philox key = the read's Phred seed
  uint32_t dst;     // first base of the read in block coordinates x (x = byte in the text - the block's 64-byte-aligned origin)
  uint32_t lw;      // L | (2 * (source position & 15)) << 16 | rev << 31
};

// (-DSIMMR_ABLATE_TL_OR, timing only: plain LDS writes in their place — wrong bytes, the same addresses)
SIMMR_DEV void tl_or32(uint32_t* p, uint32_t v) {
#if defined(SIMMR_ABLATE_TL_OR)
  *(volatile uint32_t*)p = v;
#else
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_or_b32, nothing returned
#endif
}
SIMMR_DEV void tl_or64(uint64_t* p, uint64_t v) {
#if defined(SIMMR_ABLATE_TL_OR)
  *(volatile uint64_t*)p = v;
#else
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_or_b64
#endif
}
// sixteen bytes (w0 = the first four) ORed into the ring at text position p (any byte): five aligned words
SIMMR_DEV void tl_or16(uint32_t* __restrict__ ringw, uint32_t p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  // byte i of word k of the shifted run = byte i + 4 - a of (w_k : w_{k-1}); v_perm_b32 selects from {S0 = bytes 4..7, S1 = bytes 0..3}
  const uint32_t sel = 0x07060504u - (p & 3u) * 0x01010101u;
  uint32_t* d = ringw + ((p & (TL_RING - 1u)) >> 2);
  tl_or32(d + 0, __builtin_amdgcn_perm(w0, 0u, sel));
  tl_or32(d + 1, __builtin_amdgcn_perm(w1, w0, sel));
  tl_or32(d + 2, __builtin_amdgcn_perm(w2, w1, sel));
  tl_or32(d + 3, __builtin_amdgcn_perm(w3, w2, sel));
  tl_or32(d + 4, __builtin_amdgcn_perm(0u, w3, sel));
}
// bytes [lo, hi) of a 16-byte chunk (the two chunks of a segment that a neighbour shares)
SIMMR_DEV void tl_store_bytes(uint8_t* __restrict__ d, v4u32 v, uint32_t lo, uint32_t hi) {
  for (uint32_t b = lo; b < hi; b++) {
    const uint32_t w = b < 4u ? v.x : (b < 8u ? v.y : (b < 12u ? v.z : v.w));
    d[b] = (uint8_t)(w >> (8u * (b & 3u)));
  }
}
// what the wave's lanes wrote to LDS is visible to the wave's lanes (LDS serves a wave's instructions in order; this
// keeps the compiler from moving accesses across the point)
SIMMR_DEV void tl_wave_sync() {
#if defined(TL_FENCE)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
#endif
}

// pitch8: bytes of a header slot (a multiple of 8, an odd number of 8-byte words; engine.hip: tl_slot_pitch);
// t9 = 16-byte tasks per slot = ceil(pitch8 / 16), inv_t9 = 65536 / t9 + 1
template <bool HAS_EXC, bool COPY_ONLY, bool CACHED, bool ESCQ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
k_emit_text_lines(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint32_t genome_const, uint64_t n_units, PlanArrays pl,
                  const uint32_t* __restrict__ u_contig, const uint32_t* __restrict__ u_genome, const uint64_t* __restrict__ u_seed,
                  uint8_t* text, uint32_t qual_offset, uint64_t first_unit, uint32_t read_id_base,
                  unsigned long long* __restrict__ counters, const uint8_t* __restrict__ hlen, const FqTemplate* __restrict__ fq_tp,
                  FqTables fq_tb, uint32_t fq_lit_bytes, uint32_t pitch8, uint32_t t9, uint32_t inv_t9,
                  const uint64_t* __restrict__ off64) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tl_slots[];  // TL_GROUP * pitch8
  __shared__ __attribute__((aligned(16))) uint8_t fq_lit[FQ_LIT_MAX + 8];
  __shared__ FqSeg fq_segs[FQ_MAX_SEGS];
  __shared__ uint2 jtab[COPY_ONLY ? 1 : 1024];  // level-1 columns (philox_pick)
  __shared__ uint32_t asc[256];                 // four 2-bit codes -> four ASCII bytes
  __shared__ TlRec recA[PHILOX_READS];
  // the 2-bit plane word that holds the read's first source base: its index in the one genome's plane (CACHED), or its address
  typedef typename std::conditional<CACHED, uint32_t, uint64_t>::type WaT;
  __shared__ WaT recW[PHILOX_READS];
  __shared__ uint16_t r_gs[PHILOX_READS + 2];   // first item of each read among the block's items; the block's items past the last read
  __shared__ uint8_t r_h[PHILOX_READS];         // header bytes
  __shared__ uint64_t x_src[HAS_EXC ? PHILOX_READS : 1];
  __shared__ const uint32_t* x_mask[HAS_EXC ? PHILOX_READS : 1];
  __shared__ uint64_t cbase[CACHED ? PHILOX_CBASE : 1];
  __shared__ uint4 nmask[17];
  __shared__ uint32_t nmask2[17];
  __shared__ uint64_t lds4w[4];
  __shared__ __attribute__((aligned(16))) uint32_t ring_all[4][(TL_RING + TL_GUARD) / 4];
  __shared__ __attribute__((aligned(8))) uint8_t owner_all[4][TL_MAP];
  __shared__ unsigned long long spill_bases, spill_wrap;
  // the two chunks of a segment that a neighbour shares, kept until the segment's rounds are over (per wave: first, last)
  __shared__ __attribute__((aligned(16))) uint4 edge_v[4][2];
  __shared__ uint32_t edge_m[4][2];  // chunk's x | first byte << 24 | (one past the last byte & 15) << 28; ~0 = none
  const uint32_t qoff = qual_offset & 0xffu;
  const uint32_t const_q4 = (((60u + qoff) & 0xffu) * 0x01010101u);  // perfect_short.rs:42-44
  constexpr bool esc_q = ESCQ;
  {
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < fq_lit_bytes; i += 256) fq_lit[i] = fq_tb.blob[i];
#pragma unroll
    for (uint32_t c = t; !COPY_ONLY && c < 1024u; c += 256u) {  // (as k_emit_philox)
      const uint32_t e = prof.philox_t1[c];
      uint32_t T = e & 0xffffu;
      const uint32_t A = e >> 16;
      uint32_t B = prof.philox_t1[1024u + c];
      if (T >= 16384u) { T = 0u; B = A; }
      auto res = [&](uint32_t oc) { return oc == PHILOX_ESC ? (esc_q ? 0xff00u : ((qoff << 8) | 4u)) : (((((oc & 0xffu) + qoff) & 0xffu) << 8) | (oc >> 8)); };
      jtab[c] = make_uint2((c << 22) | (T << 8), res(A) | (res(B) << 16));
    }
    if (t <= 16u) {
      auto bytes = [](int k) { return k >= 4 ? 0xffffffffu : (k <= 0 ? 0u : ((1u << (8 * k)) - 1u)); };
      nmask[t] = make_uint4(bytes((int)t), bytes((int)t - 4), bytes((int)t - 8), bytes((int)t - 12));
      nmask2[t] = t >= 16u ? 0xffffffffu : ((1u << (2u * t)) - 1u);
    }
    const uint32_t acgt = 0x54474341u;  // "ACGT"
    asc[t] = ((acgt >> (8 * (t & 3u))) & 0xffu) | (((acgt >> (8 * ((t >> 2) & 3u))) & 0xffu) << 8) |
             (((acgt >> (8 * ((t >> 4) & 3u))) & 0xffu) << 16) | (((acgt >> (8 * (t >> 6))) & 0xffu) << 24);
    uint32_t* rz = &ring_all[0][0];
    for (uint32_t i = t; i < 4u * (TL_RING + TL_GUARD) / 4u; i += 256u) rz[i] = 0u;  // the rings start empty = zero
    if (t == 0) { spill_bases = 0ull; spill_wrap = 0ull; }
    if (t < 8u) edge_m[t >> 1][t & 1u] = 0xffffffffu;
  }
  const uint32_t fq_n_segs = fq_stage_template(fq_tp, fq_segs);
  typedef const __attribute__((address_space(1))) ContigDev* global_contig_ptr;
  const uint32_t* packed0 = nullptr;
  const uint32_t* mask0 = nullptr;
  if (CACHED) {
    const GenomeDev* G0 = genomes + genome_const;
    packed0 = G0->packed;
    mask0 = (HAS_EXC && G0->has_exc) ? G0->mask : nullptr;
    const uint32_t nc = G0->n_contigs < PHILOX_CBASE ? G0->n_contigs : PHILOX_CBASE;
    if (threadIdx.x < nc) cbase[threadIdx.x] = ((global_contig_ptr)G0->contigs)[threadIdx.x].base;
  }
#if defined(TL_DIAG)
  uint32_t tl_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tl_t0 = __builtin_readcyclecounter();
#endif
  uint64_t qsum = 0;
  uint32_t n_subst = 0, n_acgt = 0;
  uint32_t p_bases32 = 0;
  uint32_t s_redrawn = 0, s_seedsubst = 0;  // per wave
  const uint64_t n_reads = 2 * n_units;
  const bool q_nowrap = qoff + prof.philox_qmax <= 255u;
  const uint64_t n_blocks = (n_units + PHILOX_UNITS - 1) / PHILOX_UNITS;
  const uint32_t t8 = pitch8 >> 3;  // 8-byte words per slot
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    uint32_t tix = threadIdx.x;
    asm volatile("" : "+v"(tix));  // (see k_emit_philox: per-block addresses are made where they are used)
    const uint64_t u0 = blk * PHILOX_UNITS;
    const uint32_t nu = (n_units - u0) < PHILOX_UNITS ? (uint32_t)(n_units - u0) : PHILOX_UNITS;
    const uint32_t nr = 2u * nu;
    const uint64_t out0 = off64[(2 * u0) >> 6];  // first byte of the block's first record
    // block coordinates: x = byte in the text - (the block's first byte rounded down to a line); a ring chunk at
    // x & (TL_RING - 1) is then an aligned 16-byte chunk of device memory, and x & 63 is the byte's place in its line
    const uint32_t delta = (uint32_t)((uintptr_t)(text + out0) & 63u);
    uint8_t* const gbase = text + out0 - delta;
    lds_barrier();  // the previous block is done with records, slots and maps
    // ---- the block's reads: records, item counts, places (one thread per read) ----
    uint32_t L0 = 0, h0 = 0;
    const bool on = tix < nr;
    if (on) {
      L0 = pl.len[u0 + (tix >> 1)];
      h0 = hlen[2 * u0 + tix];
    }
    uint64_t tot2;
    const uint64_t ex2 = wg_exclusive_scan_2x32((uint64_t)((L0 + 15u) >> 4) | ((uint64_t)(on ? h0 + 2u * L0 + 5u : 0u) << 32), lds4w, &tot2, tix);
    const uint32_t n_items = (uint32_t)tot2;
    const uint32_t x_rec = delta + (uint32_t)(ex2 >> 32);       // this thread's record
    const uint32_t x_blk_end = delta + (uint32_t)(tot2 >> 32);  // one past the block's last byte
    r_gs[tix] = on ? (uint16_t)(uint32_t)ex2 : (uint16_t)n_items;
    if (tix < 2u) r_gs[PHILOX_READS + tix] = (uint16_t)n_items;
    // what this thread's read shows in its header
    uint64_t h_pos = 0;
    uint32_t h_genome = 0, h_contig = 0, h_flags = 0;
    FqIds h_ids = FqIds{0u, 0u, 0u, 0u};  // where its genome's and its contig's ids are (read here, in the shadow of the plan's rows)
    if (on) {
      const uint32_t t = tix;
      const uint64_t u = u0 + (t >> 1);
      const uint32_t rev = t & 1u;
      const uint32_t L = L0;
      const uint32_t contig = u_contig[u];
      const uint32_t genome = (!CACHED && u_genome) ? u_genome[u] : genome_const;
      const uint64_t pos = rev ? pl.b[u] : pl.a[u];
      const uint64_t key = COPY_ONLY ? 0ull : (rev ? pl.qs2[u] : u_seed[u]);
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
      TlRec rc;
      rc.k0 = (uint32_t)key; rc.k1 = (uint32_t)(key >> 32);
      rc.dst = x_rec + h0 + 1u;
      rc.lw = (L & 0xffffu) | ((2u * (uint32_t)(src & 15u)) << 16) | (rev << 31);
      recA[t] = rc;
      recW[t] = CACHED ? (WaT)(src >> 4) : (WaT)(uintptr_t)(packed + (src >> 4));
      r_h[t] = (uint8_t)h0;
      if (HAS_EXC) { x_src[t] = src; x_mask[t] = mk; }
      const uint32_t fl = pl.flags[u];
      h_pos = pos; h_genome = genome; h_contig = contig; h_flags = rev ? fl : 0u;
      h_ids = fq_ids(fq_tb, genome, contig);
      if (!rev) {
        p_bases32 += 2u * L;
        if (p_bases32 >= 0x80000000u) { atomicAdd(&spill_bases, (unsigned long long)p_bases32); p_bases32 = 0u; }
      }
      s_redrawn += (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_REDRAWN)));
      s_seedsubst += (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_QSEED_SUBST))) +
                     (uint32_t)__builtin_popcountll(__ballot(!rev && (fl & SIMMR_FLAG_MSEED_SUBST)));
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tix >> 6), lane = tix & 63u;
    uint32_t* const ringw = &ring_all[wave][0];
    uint8_t* const ringb = reinterpret_cast<uint8_t*>(ringw);
    uint8_t* const owner = &owner_all[wave][0];
    for (uint32_t phase = 0; phase * TL_GROUP < nr; phase++) {
      TL_T(0);
      lds_barrier();  // the slots are free; (the first time) records, template and literals are staged
      TL_T(1);
      if (on && (tix / TL_GROUP) == phase) {
        // header + '\n' into this read's slot, at the byte offset its place in the text has modulo 8 (the slot is zero elsewhere)
        uint8_t* h = tl_slots + (tix & (TL_GROUP - 1u)) * pitch8;
        uint64_t* hz = reinterpret_cast<uint64_t*>(h);
        uint64_t gidb[6], sidb[6];  // the first 48 bytes of both ids, requested together: one memory latency per batch of headers
        fq_fetch_id(fq_tb.blob, h_ids.gid_off, h_ids.gid_len, gidb);
        fq_fetch_id(fq_tb.blob, h_ids.sid_off, h_ids.sid_len, sidb);
        for (uint32_t k = 0; k < t8; k++) hz[k] = 0ull;
        const uint32_t h_rev = tix & 1u;
        FqFields hf;
        hf.start = h_rev ? h_pos + L0 : h_pos;  // simulate.rs:289,295
        hf.end = h_rev ? h_pos : h_pos + L0;    // simulate.rs:290,296
        hf.genome = h_genome; hf.contig = h_contig; hf.flags = h_flags; hf.L = L0;
        hf.read_id = read_id_base + (uint32_t)(first_unit + u0 + (tix >> 1));  // simulate.rs:85-89,274
        (void)fq_format_header_fetched(h, fq_segs, fq_n_segs, fq_tb, fq_lit, hf, h_rev ? '2' : '1', x_rec & 7u, h_ids, gidb, sidb);
      }
      TL_T(2);
      lds_barrier();
      TL_T(3);
      // ---- this wave's segment: reads R0 .. R0 + nsr of the block, one contiguous piece of text ----
      const uint32_t R0 = phase * TL_GROUP + wave * TL_SEG_READS;
      if (R0 >= nr) continue;
      const uint32_t nsr = nr - R0 < TL_SEG_READS ? nr - R0 : TL_SEG_READS;
      const uint32_t it0 = r_gs[R0];                 // the segment's first item among the block's
      const uint32_t n_it = r_gs[R0 + nsr] - it0;    // its items
      // lane j < nsr holds read R0 + j (wave-uniform values are read off these with v_readlane)
      uint32_t my_gs = 0xffffu, my_g = 0, my_dst = 0, my_x = 0, my_lim = 0, my_rev = 0;
      if (lane < nsr) {
        const uint32_t r = R0 + lane;
        const TlRec rc = recA[r];
        const uint32_t L = rc.lw & 0xffffu, h = r_h[r];
        my_gs = r_gs[r] - it0;
        my_g = (L + 15u) >> 4;
        my_dst = rc.dst;
        my_x = rc.dst - h - 1u;
        my_rev = rc.lw >> 31;
        // the last byte a round that touches this read can write, with the slack of an item's five words and of the
        // slot's zero tail: it must stay inside the ring's window [F, F + TL_RING)
        const uint32_t rec_end = my_dst + 2u * L + 4u + 24u, slot_end = (my_x & ~7u) + pitch8 + 8u;
        my_lim = rec_end > slot_end ? rec_end : slot_end;
        // item -> read of the segment: every read writes its index over its items
        const uint64_t j8 = (uint64_t)lane * 0x0101010101010101ull;
        if (my_g >= 8u) {  // (8 <= g <= 16: two 8-byte writes, the second ending where the read's items end)
          *reinterpret_cast<u64_unaligned*>(owner + my_gs) = j8;
          *reinterpret_cast<u64_unaligned*>(owner + my_gs + my_g - 8u) = j8;
        } else {
          for (uint32_t k = 0; k < my_g; k++) owner[my_gs + k] = (uint8_t)lane;
        }
      }
      const uint32_t x_begin = __builtin_amdgcn_readlane(my_x, 0);
      const uint32_t x_end = (R0 + nsr < nr) ? (recA[R0 + nsr].dst - r_h[R0 + nsr] - 1u) : x_blk_end;
      tl_wave_sync();
      // ---- rounds.  Order inside a round (vmcnt counts a wave's loads AND stores, in order, so a wait for a load also
      // waits for every store issued before it): items of round k (their plane words were fetched in round k - 1) ->
      // tasks -> the NEXT round's extent and the fetch of its plane words -> this round's flush.  The one vector-memory
      // wait of a round then stands in front of the first use of the plane word, behind the Philox rounds and the
      // lookups: the stores of the round before and the load have had that long.  (First form of this loop: fetch at the
      // top of the round, right behind the flush's stores — a wait for HBM write acknowledgements per round, 32 ms.)
      uint32_t s = 0, j_s = 0, F = x_begin & ~63u;
      uint32_t e = 0, j_e = 0;
      // what round takes items [s, e) and starts the reads [j_s, j_e), for a window that begins at F
      auto extent = [&](const uint32_t s, const uint32_t j_s, const uint32_t F, uint32_t& e, uint32_t& j_e) {
        const uint32_t e0 = s + 64u < n_it ? s + 64u : n_it;
        const uint64_t viol = __ballot(lane >= j_s && lane < nsr && my_lim > F + TL_RING);
        const uint32_t jv = viol ? (uint32_t)__builtin_ctzll(viol) : nsr;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(lane < nsr && (my_gs < e0 || e0 == n_it)));
        j_e = cnt < jv ? cnt : jv;
        e = e0;
        if (j_e < cnt) { const uint32_t gv = __builtin_amdgcn_readlane(my_gs, j_e); e = gv < e0 ? gv : e0; }
      };
      // this lane's item of a round: its read, its place in the read, the read's record and the plane word
      struct Fetch { uint32_t r, ci; uint4 ra; uint64_t raw; };
      auto fetch = [&](const uint32_t s, const uint32_t e) -> Fetch {
        Fetch f;
        f.r = R0; f.ci = 0; f.ra = make_uint4(0u, 0u, 0u, 0u); f.raw = 0ull;
        if (s + lane < e) {
          const uint32_t item = s + lane;
          f.r = R0 + owner[item];
          f.ra = *reinterpret_cast<const uint4*>(&recA[f.r]);
          f.ci = item - ((uint32_t)r_gs[f.r] - it0);
#if defined(SIMMR_ABLATE_CODES)
          f.raw = ((uint64_t)recW[f.r] + f.ci) * 0x9E3779B97F4A7C15ull;
#else
          // (`text` is deliberately not __restrict__: the compiler must then keep this load in front of the flush's stores
          // instead of sinking it to its use in the next round)
          const uint64_t wa = CACHED ? (uint64_t)(uintptr_t)packed0 + 4ull * ((uint64_t)recW[f.r] + f.ci) : (uint64_t)recW[f.r] + 4ull * f.ci;
          f.raw = *reinterpret_cast<global_u64_unaligned_ptr>(wa);
#endif
        }
        return f;
      };
      extent(s, j_s, F, e, j_e);
      Fetch cur = fetch(s, e);
      for (;;) {
        if (e == s && j_e == j_s) break;  // (cannot happen: engine.hip sizes the ring so that a read always fits)
        TL_T(4);
        // ---- how far the text will be complete after this round (every whole line below that point can go) ----
        const bool last = (e == n_it) && (j_e == nsr);
        uint32_t C;
        if (last) {
          C = x_end;
        } else {
          const uint32_t jl = j_e - 1u;  // (a round that is not the last started a read or is inside one: j_e >= 1)
          const uint32_t gl = __builtin_amdgcn_readlane(my_gs, jl), ng = __builtin_amdgcn_readlane(my_g, jl);
          if (e > gl && e < gl + ng) {  // inside read jl: its bases are complete up to item e (a reverse mate's fill from the end)
            const uint32_t dl = __builtin_amdgcn_readlane(my_dst, jl);
            C = __builtin_amdgcn_readlane(my_rev, jl) ? dl : dl + 16u * (e - gl);
          } else {
            C = j_e < nsr ? __builtin_amdgcn_readlane(my_x, j_e) : x_end;
          }
        }
        const uint32_t limit = last ? x_end : (C & ~63u);
        const uint32_t F2 = limit > F ? limit : F;
        // ---- the next round's extent, item records and plane words: asked for now, so that the two LDS lookups and the
        // memory load behind them run under this round's items (what they depend on — where this round ends — is known)
        uint32_t e2 = e, j_e2 = j_e;
        Fetch nxt;
        nxt.r = R0; nxt.ci = 0; nxt.ra = make_uint4(0u, 0u, 0u, 0u); nxt.raw = 0ull;
        if (!last) {
          extent(e, j_e, F2, e2, j_e2);
          nxt = fetch(e, e2);
        }
        TL_T(7);
        // ---- items ----
        if (s + lane < e) {
          const uint32_t r = cur.r;
          const uint4 ra = cur.ra;
          const uint32_t k0 = ra.x, k1 = ra.y, lw = ra.w;
          const uint32_t L = lw & 0xffffu, rev = lw >> 31;
          const uint32_t ci = cur.ci;
          const uint32_t b0 = ci << 4;
          const uint32_t n = (L - b0) < 16u ? (L - b0) : 16u;
          uint32_t exc = 0u;
          if (HAS_EXC) { const uint32_t* mk = x_mask[r]; if (mk) exc = fetch_mask16(mk, (int64_t)(x_src[r] + b0)); }
          uint32_t qr[4] = {const_q4, const_q4, const_q4, const_q4}, ss = 0;
          if (!COPY_ONLY) {
            uint32_t w[12];
#pragma unroll
            for (int c = 0; c < 3; c++) philox4x32_10(3u * ci + (uint32_t)c, 0u, k0, k1, w + 4 * c);
            bool escaped;
            {  // the sixteen lookups (k_emit_philox)
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
                  ss = __builtin_amdgcn_alignbit(x[h], ss, 2);
                  if (FLAGGED) ea |= x[h];
                }
                qr[g4] = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0501u) | __builtin_amdgcn_perm(x[3], x[2], 0x05010c0cu);
              }
              escaped = FLAGGED ? (ea & 4u) != 0u : ((qr[0] | qr[1] | qr[2] | qr[3]) & 0x80808080u) != 0u;
            }
            if (escaped) philox_repair(k0, k1, ci, prof.philox_t1, prof.philox_t2, qoff, ss, qr);
          }
          // the plane word is first looked at here, behind the draws (the wait for it, and for the stores in front of it)
          uint32_t raw_lo = (uint32_t)cur.raw, raw_hi = (uint32_t)(cur.raw >> 32);
          asm volatile("" : "+v"(raw_lo), "+v"(raw_hi), "+v"(ss));
          uint32_t codes = (uint32_t)((((uint64_t)raw_hi << 32) | raw_lo) >> ((lw >> 16) & 31u));
          const uint4 bm = nmask[n];
          const uint32_t live2 = nmask2[n];
          if (HAS_EXC) ss &= ~spread16(exc);
          ss &= live2;
          n_subst += __builtin_popcount((ss | (ss >> 1)) & 0x55555555u);
          if (HAS_EXC) n_acgt += __builtin_popcount(~spread16(exc) & live2 & 0x55555555u);
          if (!COPY_ONLY) {
            uint32_t qs = __builtin_amdgcn_udot4(qr[0], bm.x, 0u, false);
            qs = __builtin_amdgcn_udot4(qr[1], bm.y, qs, false);
            qs = __builtin_amdgcn_udot4(qr[2], bm.z, qs, false);
            qs = __builtin_amdgcn_udot4(qr[3], bm.w, qs, false);
            qsum += qs;  // 255 x the sum of the live bytes
          }
          if (!COPY_ONLY && !q_nowrap) {
            uint32_t nw = 0;
            for (uint32_t k = 0; k < n; k++) nw += ((qr[k >> 2] >> (8 * (k & 3u))) & 0xffu) < qoff ? 1u : 0u;
            if (nw) atomicAdd(&spill_wrap, (unsigned long long)nw);
          }
          codes = xor3(codes, ss, __builtin_amdgcn_bitop3_b32(codes, ss, 0x55555555u, 0x80) << 1);
          uint32_t p_s = ra.z + b0;                     // where the bases go (block coordinates)
          const uint32_t p_q = ra.z + L + 3u + b0;      // the qualities' line follows the bases' line and "+\n"
          if (rev) {
            // mate 2 is reverse-complemented after mutation (simulate.rs:283): base b0 + j -> byte L - 1 - (b0 + j)
            codes = reverse_complement_groups16(codes);
            if (HAS_EXC) { exc = __builtin_bitreverse32(exc) >> 16; codes ^= spread16(exc); }
            const uint32_t dead = 16u - n;
            if (dead) { codes >>= 2 * dead; if (HAS_EXC) exc >>= dead; }
            p_s = ra.z + (L - b0 - n);
          }
          uint32_t s0, s1, s2, s3;
          if (HAS_EXC) {
            s0 = expand4(codes & 0xffu, exc & 0xfu); s1 = expand4((codes >> 8) & 0xffu, (exc >> 4) & 0xfu);
            s2 = expand4((codes >> 16) & 0xffu, (exc >> 8) & 0xfu); s3 = expand4(codes >> 24, (exc >> 12) & 0xfu);
          } else {
            s0 = asc[codes & 0xffu]; s1 = asc[(codes >> 8) & 0xffu]; s2 = asc[(codes >> 16) & 0xffu]; s3 = asc[codes >> 24];
          }
#if !defined(SIMMR_ABLATE_STORES)
          // only the item's n live bytes may reach the image (what lies behind them is another piece's)
          tl_or16(ringw, p_q, qr[0] & bm.x, qr[1] & bm.y, qr[2] & bm.z, qr[3] & bm.w);
          tl_or16(ringw, p_s, s0 & bm.x, s1 & bm.y, s2 & bm.z, s3 & bm.w);
#else
          asm volatile("" :: "v"(qr[0]), "v"(qr[1]), "v"(qr[2]), "v"(qr[3]), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(p_q), "v"(p_s));
#endif
        }
        TL_T(5);
        // ---- the reads that start in this round: header + '\n' from the slot, "\n+\n", the closing '\n' ----
        {
          const uint32_t n_task = (j_e - j_s) * t9;
          for (uint32_t tk = lane; tk < n_task; tk += 64u) {
            const uint32_t q = (tk * inv_t9) >> 16, c = tk - q * t9;
            const uint32_t r = R0 + j_s + q;
            const uint32_t dst = recA[r].dst, L = recA[r].lw & 0xffffu, h = r_h[r];
            const uint32_t x8 = (dst - h - 1u) & ~7u;
            const uint64_t* sl = reinterpret_cast<const uint64_t*>(tl_slots + (r & (TL_GROUP - 1u)) * pitch8);
            uint64_t* ring64 = reinterpret_cast<uint64_t*>(ringw);
            const uint64_t v0 = sl[2u * c];
            const uint64_t v1 = 2u * c + 1u < t8 ? sl[2u * c + 1u] : 0ull;
            tl_or64(ring64 + (((x8 + 16u * c) & (TL_RING - 1u)) >> 3), v0);
            tl_or64(ring64 + (((x8 + 16u * c + 8u) & (TL_RING - 1u)) >> 3), v1);
            if (c == 0u) {  // "\n+\n" behind the bases
              const uint32_t p = dst + L, sh = 8u * (p & 3u);
              uint32_t* d = ringw + ((p & (TL_RING - 1u)) >> 2);
              tl_or32(d, 0x000a2b0au << sh);
              tl_or32(d + 1, sh ? (0x000a2b0au >> (32u - sh)) : 0u);
            } else if (c == 1u) {  // '\n' behind the qualities
              const uint32_t p = dst + 2u * L + 3u;
              tl_or32(ringw + ((p & (TL_RING - 1u)) >> 2), 0x0au << (8u * (p & 3u)));
            }
          }
        }
        TL_T(6);
        // ---- the next round's item records and plane words take the place of this round's, in front of this round's stores
        // (a register move that waits for the load — it was asked for at the round's top, and the stores of the round before
        // are older still)
        if (!last) cur = nxt;
        tl_wave_sync();
        // ---- flush ----
        // (at most TL_RING bytes = four passes of 64 chunks; the passes' LDS reads are issued together, then their stores:
        // one LDS latency per flush instead of one per pass)
        {
          v4u32 v[4];
#pragma unroll
          for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t cx = F + 16u * lane + 1024u * k;
            v[k] = v4u32{0u, 0u, 0u, 0u};
            if (cx < limit) {
              const uint32_t ro = cx & (TL_RING - 1u);
              v4u32* rp = reinterpret_cast<v4u32*>(ringb + ro);
              v[k] = *rp;
              *rp = v4u32{0u, 0u, 0u, 0u};
              if (ro < TL_GUARD) {  // what ran past the ring's end belongs here
                v4u32* gp = reinterpret_cast<v4u32*>(ringb + TL_RING + ro);
                v[k] |= *gp;
                *gp = v4u32{0u, 0u, 0u, 0u};
              }
            }
          }
#pragma unroll
          for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t cx = F + 16u * lane + 1024u * k;
            if (cx < limit) {
              const uint32_t lo = cx < x_begin ? x_begin - cx : 0u;
              const uint32_t hi = limit - cx < 16u ? limit - cx : 16u;
#if defined(SIMMR_ABLATE_STORES)
              asm volatile("" :: "v"(v[k].x), "v"(v[k].y), "v"(v[k].z), "v"(v[k].w), "v"(lo), "v"(hi));
#else
              if (lo == 0u && hi == 16u) {
                stream_store(reinterpret_cast<v4u32*>(gbase + cx), v[k]);
              } else if (lo < hi) {
                // a chunk a neighbour shares (the segment's first or last): kept for the end of the segment — byte stores
                // inside this loop cost every round a wait for all outstanding stores (the compiler guards the loop's
                // registers against them at the loop's head)
                const uint32_t kk = lo ? 0u : 1u;
                *reinterpret_cast<v4u32*>(&edge_v[wave][kk]) = v[k];
                edge_m[wave][kk] = cx | (lo << 24) | ((hi & 15u) << 28);
              }
#endif
            }
          }
        }
        if (last) { TL_T(8); }
        if (last) break;
        F = F2; s = e; j_s = j_e; e = e2; j_e = j_e2;
        TL_T(8);
        tl_wave_sync();  // (the zeroes are in place before the next round's pieces)
      }
      // the segment's shared chunks, bytewise (lane 0: the first chunk, lane 1: the last)
      tl_wave_sync();
      TL_T(8);
      if (lane < 2u) {
        const uint32_t m = edge_m[wave][lane];
        if (m != 0xffffffffu) {
          const uint32_t cx = m & 0xffffffu, lo = (m >> 24) & 15u, hi = ((m >> 28) & 15u) ? ((m >> 28) & 15u) : 16u;
          tl_store_bytes(gbase + cx, *reinterpret_cast<const v4u32*>(&edge_v[wave][lane]), lo, hi);
          edge_m[wave][lane] = 0xffffffffu;
        }
      }
    }
  }
#if defined(TL_DIAG)
  TL_T(9);
  if ((threadIdx.x & 63u) == 0) {
    for (int k = 0; k < 10; k++) atomicAdd(&tl_diag[k], (unsigned long long)tl_acc[k]);
    atomicAdd(&tl_diag[10], 1ull);
  }
#endif
  // ---- run counters (as k_emit_philox) ----
  __syncthreads();
  uint64_t p_bases = (uint64_t)p_bases32 + (threadIdx.x == 0 ? (uint64_t)spill_bases : 0ull);
  const uint64_t n_wrap = threadIdx.x == 0 ? (uint64_t)spill_wrap : 0ull;
  qsum = qsum * 0xFEFEFEFEFEFEFEFFull + 256ull * n_wrap - (uint64_t)qoff * p_bases;
  if (COPY_ONLY) qsum = 60ull * p_bases;  // perfect_short.rs:42-44
  uint64_t acgt = HAS_EXC ? (uint64_t)n_acgt : p_bases;
  for (int d = 32; d > 0; d >>= 1) {
    n_subst += __shfl_down(n_subst, d, 64);
    acgt += __shfl_down(acgt, d, 64);
    qsum += __shfl_down(qsum, d, 64);
    p_bases += __shfl_down(p_bases, d, 64);
  }
  __shared__ unsigned long long wsum[4][SIMMR_N_COUNTERS];
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long* w = wsum[threadIdx.x >> 6];
    w[SIMMR_CNT_READS] = (blockIdx.x == 0 && threadIdx.x == 0) ? (unsigned long long)n_reads : 0ull;
    w[SIMMR_CNT_BASES] = (unsigned long long)p_bases;
    w[SIMMR_CNT_ACGT_BASES] = (unsigned long long)acgt;
    w[SIMMR_CNT_SUBSTITUTIONS] = COPY_ONLY ? 0ull : (unsigned long long)n_subst;
    w[SIMMR_CNT_OUTER_REJECTS] = 0ull;
    w[SIMMR_CNT_REDRAWN] = (unsigned long long)s_redrawn;
    w[SIMMR_CNT_SEED_SUBST] = (unsigned long long)s_seedsubst;
    w[SIMMR_CNT_QUAL_SUM] = (unsigned long long)qsum;
  }
  __syncthreads();
  if (threadIdx.x < SIMMR_N_COUNTERS && counters) {
    const unsigned long long v = wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
    if (v) atomicAdd(&counters[(1u + (blockIdx.x & (SIMMR_CNT_SHARDS - 1u))) * SIMMR_N_COUNTERS + threadIdx.x], v);
  }
}

}  // namespace simmr
