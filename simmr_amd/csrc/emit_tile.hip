// emit_tile.hip — counter-mode emit for short paired reads, tile form (gfx950).  Included by engine.hip after
// kernels.hip, whose Philox / lookup / repair helpers and 32-byte read records it shares; the specification of
// the draws is the one stated in section 9 of kernels.hip and DESIGN.md section 4, so the
// output is bit-identical to k_emit_philox.  Reference behaviour: simulate.rs:260-299 (qualities, substitutions,
// reverse complement of mate 2 after mutation), minimal_short.rs:83-140.
//
// What round 2 measured about the item kernel: its instruction stream and its stores add
// up instead of overlapping; a third of the non-draw instructions serve the partial 16-base group at the end of
// each read (byte ladder, masks) and the mate-2 path, which every wave executes in every round because the
// items of a block are dealt in read order so that a wave's 16-byte stores are contiguous.  Here a workgroup
// builds the block's piece of the two output streams in LDS and writes it out afterwards:
//
//   * the block = `upb` pairs whose bases / qualities are contiguous ranges [out0, out0 + bytes) of seq[] / qual[];
//   * item phase: the block's items (16 bases of one read) are dealt in CLASS order — full groups of forward
//     mates, full groups of reverse mates, then the partial groups — so a wave-round is almost always of one
//     class and runs code specialised for it (no masks, no ladder, no reverse-complement select); an item drops
//     its 16 qualities and 16 bases into the LDS tiles with ds_write_b128 at the byte offsets they have in the
//     output (LDS takes any alignment), partial groups with an exact byte ladder that only their rounds execute;
//   * flush phase: after one barrier every lane moves ALIGNED 16-byte chunks from LDS to memory, so every store
//     instruction of a wave writes sixteen whole 64-byte lines (the two edge chunks of a block go bytewise);
//     while one workgroup of a CU flushes, the others are in their item phase — stores and draws overlap
//     across workgroups, which they did not inside one instruction stream.
//
// The three parts of a block run as a pipeline inside the workgroup (first measurement of the plain sequence
// prologue -> items -> flush: 17.6 ms against the item kernel's 13.5, and 8.4 ms with the items compiled out, i.e.
// the chain "load the plan rows, wait, scan, barrier" of every 64-read block was what the workgroup waited for):
//   * wave 3 starts a block's item phase by issuing the NEXT block's plan rows as global_load_lds_dword (memory
//     to LDS without registers: nine dwords per read); they arrive while the four waves draw;
//   * wave-rounds of items are handed out through an LDS counter, so a wave that waits takes fewer;
//   * after the barrier that ends the item phase, waves 0-2 flush the tiles while wave 3 turns the prefetched
//     rows into the next block's records, owner map and metadata columns (one read per lane, a wave-level scan);
//   * a second barrier, and the next item phase begins.  No wave waits for a global load between two blocks.
//
// A block whose bytes do not fit the tiles (possible: lengths are N(150, 15) draws) takes the same item code with
// direct global stores, like k_emit_philox.  The engine launches this kernel only for paired plans whose longest
// read is at most TILE_MAXL, which bounds a block's items by the size of the owner map.
#pragma once

namespace simmr {

// TILE_MAXL (device_types.hpp) = 511: 31 full groups per read, so three 12/12/8-bit counts share one scan word

template <uint32_t N> struct TileTag { static constexpr uint32_t value = N; };

#if defined(TILE_DIAG)
// diagnostic build only (make variant NAME=tile_diag DEFS=-DTILE_DIAG): s_memtime stamps summed per phase, [wave class][phase]
__device__ unsigned long long tile_diag[16];
#define TILE_STAMP(var) const uint64_t var = __builtin_readcyclecounter()
#else
#define TILE_STAMP(var)
#endif

#define TILE_READS 64u /* reads per block at most: one lane of the prologue wave per read */
typedef const __attribute__((address_space(1))) void* tile_gptr;
typedef __attribute__((address_space(3))) void* tile_lptr;

// upb = pairs per block (<= 32); cap = bytes per tile (multiple of 16).  Dynamic LDS: two tiles of cap + 32 bytes.
template <bool HAS_EXC, bool CACHED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_emit_philox_tile(ProfileDev prof, const GenomeDev* __restrict__ genomes, uint32_t genome_const, uint64_t n_units,
                   PlanArrays pl, const uint64_t* __restrict__ u_off, const uint32_t* __restrict__ u_contig,
                   const uint32_t* __restrict__ u_genome, const uint64_t* __restrict__ u_seed,
                   uint8_t* __restrict__ seq, uint8_t* __restrict__ qual, uint32_t qual_offset, uint64_t first_unit,
                   uint32_t read_id_base, OutCols o, unsigned long long* __restrict__ counters, uint32_t upb,
                   uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tile_mem[];
  __shared__ uint2 jtab[1024];   // level-1 columns (philox_pick)
  __shared__ uint32_t asc[256];  // four 2-bit codes -> four ASCII bytes
  __shared__ PhRec recs2[2][TILE_READS];  // the block being drawn and the next one (written while this one is drawn)
  __shared__ uint64_t x_src[HAS_EXC ? 2 * TILE_READS : 1];
  __shared__ const uint32_t* x_mask[HAS_EXC ? 2 * TILE_READS : 1];
  __shared__ uint8_t owner[TILE_READS * 32u];  // class-ordered item -> read
  __shared__ uint64_t cbase[CACHED ? PHILOX_CBASE : 1];
  __shared__ uint4 nmask[17];
  __shared__ uint32_t nmask2[17];
  // the next block's plan rows, one dword column each (global_load_lds_dword: lane l writes word l)
  __shared__ uint32_t row_len[64], row_contig[64], row_flags[64], row_genome[64];
  __shared__ uint32_t row_off_lo[64], row_off_hi[64], row_pos_lo[64], row_pos_hi[64], row_key_lo[64], row_key_hi[64];
  // what the prologue wave tells the others about the block: {nA, nAB, n_items, bytes, out0 lo, out0 hi}; the round counter
  __shared__ uint32_t blkinfo2[2][8];
  __shared__ uint32_t wr_next;
  const uint32_t qoff = qual_offset & 0xffu;
  const uint32_t tstride = cap + 32u;
  uint8_t* const qtile = tile_mem;
  uint8_t* const stile = tile_mem + tstride;
  {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t c = t; c < 1024u; c += 256u) {
      const uint32_t e = prof.philox_t1[c];
      uint32_t T = e & 0xffffu;
      const uint32_t A = e >> 16;
      uint32_t B = prof.philox_t1[1024u + c];
      if (T >= 16384u) { T = 0u; B = A; }
      auto res = [&](uint32_t oc) { return oc == PHILOX_ESC ? ((qoff << 8) | 4u) : (((((oc & 0xffu) + qoff) & 0xffu) << 8) | (oc >> 8)); };
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
  }
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
  uint64_t qsum = 0;  // adds encoded qualities; the offset is taken off at the end (every base is drawn exactly once: p_bases)
  uint32_t n_subst = 0, n_acgt = 0, n_wrap = 0;  // (n_acgt only counts where there are exceptions; else it is p_bases)
  uint64_t p_bases = 0;
  uint32_t p_redrawn = 0, p_seedsubst = 0;
  const uint64_t n_reads = 2 * n_units;
  const bool q_nowrap = qoff + prof.philox_qmax <= 255u;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t n_blocks = (n_units + upb - 1) / upb;

  // ---- wave 3, one read per lane: fetch a block's plan rows into LDS (asynchronously) ...
  auto fetch_rows = [&](const uint64_t blk) {
    const uint64_t u0 = blk * upb;
    const uint32_t nu = (n_units - u0) < upb ? (uint32_t)(n_units - u0) : upb;
    uint32_t lane = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane));  // (opaque: the nine per-lane addresses are not to be kept in registers across the item phase)
    if (lane >= 2u * nu) return;
    const uint64_t u = u0 + (lane >> 1);
    const uint32_t rev = lane & 1u;
    auto dma = [&](const void* g, uint32_t* col) {
      __builtin_amdgcn_global_load_lds((tile_gptr)g, (tile_lptr)col, 4, 0, 0);
    };
    dma(pl.len + u, row_len);
    dma(u_contig + u, row_contig);
    dma(reinterpret_cast<const uint32_t*>(pl.flags) + (u >> 2), row_flags);  // the aligned word that holds the pair's flag byte
    if (!CACHED && u_genome) dma(u_genome + u, row_genome);
    // forward mates fetch where the pair's bytes start, reverse mates where the next pair's do (u_off has n_units + 1 entries)
    const uint32_t* po = reinterpret_cast<const uint32_t*>(u_off + u + rev);
    dma(po, row_off_lo); dma(po + 1, row_off_hi);
    const uint32_t* pp = reinterpret_cast<const uint32_t*>(rev ? pl.b + u : pl.a + u);
    dma(pp, row_pos_lo); dma(pp + 1, row_pos_hi);
    const uint32_t* pk = reinterpret_cast<const uint32_t*>(rev ? pl.qs2 + u : u_seed + u);
    dma(pk, row_key_lo); dma(pk + 1, row_key_hi);
  };
  // ---- ... and turn them into the block's records, owner map, metadata columns and block header
  auto prologue = [&](const uint64_t blk, const uint32_t buf) {
    PhRec* const recs = recs2[buf];
    uint32_t* const blkinfo = blkinfo2[buf];
    const uint64_t u0 = blk * upb;
    const uint32_t nu = (n_units - u0) < upb ? (uint32_t)(n_units - u0) : upb;
    const uint32_t nr = 2u * nu;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are in LDS (this wave issued the loads)
    uint32_t lane = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane));  // (opaque, as in fetch_rows)
    uint32_t v = 0, nf = 0, part = 0, L = 0, rev = lane & 1u;
    uint64_t start = 0;  // first output byte of the read
    if (lane < nr) {
      L = row_len[lane];
      const uint64_t off = (uint64_t)row_off_lo[lane] | ((uint64_t)row_off_hi[lane] << 32);
      start = rev ? off - L : off;
    }
    // the block's piece of either stream: from the first read's start to the last read's end
    const uint64_t out0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)start)) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(start >> 32)) << 32);
    const uint64_t endv = start + L;
    const uint64_t out1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)endv, nr - 1u)) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(endv >> 32), nr - 1u) << 32);
    if (lane < nr) {
      const uint64_t u = u0 + (lane >> 1);
      nf = L >> 4;
      part = (L & 15u) ? 1u : 0u;
      v = (rev ? (nf << 12) : nf) | (part << 24);
      const uint32_t contig = row_contig[lane];
      const uint32_t genome = (!CACHED && u_genome) ? row_genome[lane] : genome_const;
      const uint64_t pos = (uint64_t)row_pos_lo[lane] | ((uint64_t)row_pos_hi[lane] << 32);
      const uint32_t fl = (row_flags[lane] >> (8u * ((uint32_t)u & 3u))) & 0xffu;
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
      rc.k0 = row_key_lo[lane]; rc.k1 = row_key_hi[lane];
      rc.dst = (uint32_t)(start - out0);
      rc.lw = (L & 0xffffu) | ((2u * (uint32_t)(src & 15u)) << 16) | (rev << 31);
      rc.wa = (uint64_t)(uintptr_t)(packed + (src >> 4));
      rc.gs = 0; rc.pad = 0;
      recs[lane] = rc;
      if (HAS_EXC) { x_src[buf * TILE_READS + lane] = src; x_mask[buf * TILE_READS + lane] = mk; }
      // metadata columns of this read
      const uint64_t rd = 2 * u + rev;
      o.seq_off[rd] = start;
      if (rd + 1 == n_reads) o.seq_off[n_reads] = endv;  // closing CSR offset
      if (o.start) o.start[rd] = rev ? pos + L : pos;  // simulate.rs:289,295
      if (o.end) o.end[rd] = rev ? pos : pos + L;      // simulate.rs:290,296
      if (o.contig) o.contig[rd] = contig;
      if (o.genome) o.genome[rd] = genome;
      if (o.read_id) o.read_id[rd] = read_id_base + (uint32_t)(first_unit + u);  // simulate.rs:85-89,274
      if (o.flags) o.flags[rd] = rev ? (uint8_t)fl : 0;
      if (!rev) {
        p_bases += 2ull * L;
        p_redrawn += (fl & SIMMR_FLAG_REDRAWN) ? 1u : 0u;
        p_seedsubst += ((fl & SIMMR_FLAG_QSEED_SUBST) ? 1u : 0u) + ((fl & SIMMR_FLAG_MSEED_SUBST) ? 1u : 0u);
      }
    }
    // class-ordered item index: [full groups of forward mates][of reverse mates][partial groups]
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc += up;
    }
    const uint32_t tot = __builtin_amdgcn_readlane(inc, 63), ex = inc - v;
    const uint32_t nA = tot & 0xfffu, nAB = nA + ((tot >> 12) & 0xfffu), n_items = nAB + (tot >> 24);
    if (lane < nr) {
      recs[lane].gs = rev ? nA + ((ex >> 12) & 0xfffu) : (ex & 0xfffu);  // its first full group among the block's items
      recs[lane].pad = part ? nAB + (ex >> 24) : 0xffffffffu;             // its partial group
    }
    if (lane == 0) {
      blkinfo[0] = nA; blkinfo[1] = nAB; blkinfo[2] = n_items; blkinfo[3] = (uint32_t)(out1 - out0);
      blkinfo[4] = (uint32_t)out0; blkinfo[5] = (uint32_t)(out0 >> 32); blkinfo[6] = nr;
    }
  };
  // the item -> read map of a block from its records, four lanes per read (the map is free between two item phases)
  auto build_owner = [&](const uint32_t buf) {
    const uint32_t r = threadIdx.x >> 2, q = threadIdx.x & 3u;
    if (r >= blkinfo2[buf][6]) return;
    const PhRec rc = recs2[buf][r];
    const uint32_t nf = (rc.lw & 0xffffu) >> 4;
    for (uint32_t j = q; j < nf; j += 4u) owner[rc.gs + j] = (uint8_t)r;
    if (q == 0u && rc.pad != 0xffffffffu) owner[rc.pad] = (uint8_t)r;
  };

  if (wave == 3u) {
    fetch_rows(blockIdx.x);
    prologue(blockIdx.x, 0u);
    if (blockIdx.x + (uint64_t)gridDim.x < n_blocks) fetch_rows(blockIdx.x + (uint64_t)gridDim.x);
    if (lane == 0) wr_next = 0u;
  }
#if defined(TILE_STAGGER)
  {  // the workgroups that share a CU start a quarter of a block period apart (measurement: do their phases then interleave?)
    const uint32_t ph = (blockIdx.x / ((gridDim.x + 3u) / 4u)) & 3u;
    for (uint32_t j = 0; j < ph; j++) __builtin_amdgcn_s_sleep(TILE_STAGGER);
  }
#endif
  __syncthreads();  // tables, the first block's records and header are in place
  build_owner(0u);
  lds_barrier();
#if defined(TILE_DIAG)
  uint64_t d_pro = 0, d_item = 0, d_wa = 0, d_flush = 0, d_own = 0, d_wb = 0;
#endif
  uint32_t cur = 0;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x, cur ^= 1u) {
    const uint4* const rec4 = reinterpret_cast<const uint4*>(recs2[cur]);
    const uint32_t nA = blkinfo2[cur][0], nAB = blkinfo2[cur][1], n_items = blkinfo2[cur][2], bytes = blkinfo2[cur][3];
    const uint64_t out0 = (uint64_t)blkinfo2[cur][4] | ((uint64_t)blkinfo2[cur][5] << 32);
    const bool fits = bytes <= cap;
    uint8_t* const seq_blk = seq + out0;
    uint8_t* const qual_blk = qual + out0;
    const uint32_t a0q = (uint32_t)((uintptr_t)qual_blk & 15u), a0s = (uint32_t)((uintptr_t)seq_blk & 15u);
    const uint64_t next = blk + gridDim.x;
    TILE_STAMP(t0);
    if (wave == 3u && next < n_blocks) {
      // the next block's records from the rows fetched one block ago, then the rows of the block after it
      prologue(next, cur ^ 1u);
      if (next + gridDim.x < n_blocks) fetch_rows(next + gridDim.x);
    }
    // ---- item phase -----------------------------------------------------------------------------------------
    // `direct`: global stores (the block does not fit the tiles); otherwise LDS at the output's own byte offsets
    auto item = [&](auto any_part_t, auto any_fwd_t, auto any_rev_t, auto direct_t, const uint32_t i) {
      constexpr bool ANY_PART = decltype(any_part_t)::value != 0, ANY_FWD = decltype(any_fwd_t)::value != 0,
                     ANY_REV = decltype(any_rev_t)::value != 0, DIRECT = decltype(direct_t)::value != 0;
      const uint32_t r = owner[i];
      const uint4 ra = rec4[2 * r], rb = rec4[2 * r + 1];
      const uint32_t k0 = ra.x, k1 = ra.y, lw = ra.w;
      const uint32_t L = lw & 0xffffu;
      const bool rev = ANY_REV && (!ANY_FWD || (lw >> 31));
      const bool is_part = ANY_PART && i >= nAB;
      const uint32_t ci = is_part ? (L >> 4) : (i - rb.z);
      const uint32_t b0 = ci << 4;
      const uint32_t n = is_part ? (L & 15u) : 16u;
      const uint64_t wa = ((uint64_t)rb.x | ((uint64_t)rb.y << 32)) + 4ull * ci;
      uint32_t codes = (uint32_t)(*reinterpret_cast<global_u64_unaligned_ptr>(wa) >> ((lw >> 16) & 31u));
      uint32_t exc = 0u;
      if (HAS_EXC) { const uint32_t* mk = x_mask[cur * TILE_READS + r]; if (mk) exc = fetch_mask16(mk, (int64_t)(x_src[cur * TILE_READS + r] + b0)); }
      uint32_t qr[4] = {0, 0, 0, 0}, ss = 0, ea = 0;
      uint32_t w[12];
#pragma unroll
      for (int c = 0; c < 3; c++) philox4x32_10(3u * ci + (uint32_t)c, 0u, k0, k1, w + 4 * c);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const uint32_t w0 = w[3 * g4], w1 = w[3 * g4 + 1], w2 = w[3 * g4 + 2];
        const uint32_t R[4] = {w0 << 8, __builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 8), w2};
        uint32_t x[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
          x[h] = philox_pick(R[h], jtab);
          ss = __builtin_amdgcn_alignbit(x[h], ss, 2);  // ascending, so base j ends at bits 2j of ss
          ea |= x[h];
        }
        // the four quality bytes (byte 1 of each x) with v_perm_b32: no SDWA write of a partly preserved register
        const uint32_t lo = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0501u);  // bytes: x0.b1, x1.b1, 0, 0
        const uint32_t hi = __builtin_amdgcn_perm(x[3], x[2], 0x05010c0cu);  // bytes: 0, 0, x2.b1, x3.b1
        qr[g4] = lo | hi;
      }
      if (ea & 4u) philox_repair(k0, k1, ci, prof.philox_t1, prof.philox_t2, qoff, ss, qr);
      if (HAS_EXC) ss &= ~spread16(exc);
      if (ANY_PART) {
        const uint4 bm = nmask[n];
        ss &= nmask2[n];
        uint32_t qs = __builtin_amdgcn_sad_u8(qr[0] & bm.x, 0u, 0u);
        qs = __builtin_amdgcn_sad_u8(qr[1] & bm.y, 0u, qs);
        qs = __builtin_amdgcn_sad_u8(qr[2] & bm.z, 0u, qs);
        qs = __builtin_amdgcn_sad_u8(qr[3] & bm.w, 0u, qs);
        qsum += qs;
        if (HAS_EXC) n_acgt += __builtin_popcount(~spread16(exc) & nmask2[n] & 0x55555555u);
      } else {
        uint32_t qs = __builtin_amdgcn_sad_u8(qr[0], 0u, 0u);
        qs = __builtin_amdgcn_sad_u8(qr[1], 0u, qs);
        qs = __builtin_amdgcn_sad_u8(qr[2], 0u, qs);
        qs = __builtin_amdgcn_sad_u8(qr[3], 0u, qs);
        qsum += qs;
        if (HAS_EXC) n_acgt += __builtin_popcount(~spread16(exc) & 0x55555555u);
      }
      n_subst += __builtin_popcount((ss | (ss >> 1)) & 0x55555555u);
      if (!q_nowrap) {
        for (uint32_t j = 0; j < n; j++) n_wrap += ((qr[j >> 2] >> (8 * (j & 3u))) & 0xffu) < qoff ? 1u : 0u;
      }
      codes = (((codes & 0x33333333u) + (ss & 0x33333333u)) & 0x33333333u) |
              (((codes & 0xccccccccu) + (ss & 0xccccccccu)) & 0xccccccccu);
      const uint64_t q_lo = (uint64_t)qr[0] | ((uint64_t)qr[1] << 32), q_hi = (uint64_t)qr[2] | ((uint64_t)qr[3] << 32);
      const uint32_t o_q = ra.z + b0;
      uint32_t o_s = o_q;
      if (ANY_REV) {
        // mate 2 is reverse-complemented after mutation (simulate.rs:283), still in the code domain
        uint32_t rc = ~reverse_groups16(codes);
        uint32_t rexc = 0u;
        if (HAS_EXC) { rexc = __builtin_bitreverse32(exc) >> 16; rc ^= spread16(rexc); }
        if (ANY_PART) { const uint32_t dead = 16u - n; rc >>= 2u * dead; if (HAS_EXC) rexc >>= dead; }  // (dead <= 15)
        codes = rev ? rc : codes;
        if (HAS_EXC) exc = rev ? rexc : exc;
        o_s = rev ? ra.z + (L - b0 - n) : o_q;
      }
      uint32_t s0, s1, s2, s3;
      if (HAS_EXC) {
        s0 = expand4(codes & 0xffu, exc & 0xfu); s1 = expand4((codes >> 8) & 0xffu, (exc >> 4) & 0xfu);
        s2 = expand4((codes >> 16) & 0xffu, (exc >> 8) & 0xfu); s3 = expand4(codes >> 24, (exc >> 12) & 0xfu);
      } else {
        s0 = asc[codes & 0xffu]; s1 = asc[(codes >> 8) & 0xffu]; s2 = asc[(codes >> 16) & 0xffu]; s3 = asc[codes >> 24];
      }
      const uint64_t s_lo = (uint64_t)s0 | ((uint64_t)s1 << 32), s_hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
      uint8_t* const qd = DIRECT ? qual_blk + o_q : qtile + a0q + o_q;
      uint8_t* const sd = DIRECT ? seq_blk + o_s : stile + a0s + o_s;
#if defined(TILE_ABLATE_LDSW)
      if (!DIRECT) { asm volatile("" :: "v"(q_lo), "v"(q_hi), "v"(s_lo), "v"(s_hi), "v"(qd), "v"(sd)); return; }  // timing only
#endif
      if (!ANY_PART || n == 16u) {
        store16(qd, q_lo, q_hi);
        store16(sd, s_lo, s_hi);
      } else {
        store_tail(qd, q_lo, q_hi, n);
        store_tail(sd, s_lo, s_hi, n);
      }
    };
    TILE_STAMP(t1);
#if defined(TILE_ABLATE_ITEMS)
    const uint32_t n_wr = 0;  // timing only: prologue and flush
#else
    const uint32_t n_wr = (n_items + 63u) >> 6;
#endif
    for (;;) {
      uint32_t wr = 0;
      if (lane == 0) wr = atomicAdd(&wr_next, 1u);
      wr = __builtin_amdgcn_readfirstlane(wr);
      if (wr >= n_wr) break;
      const uint32_t i_lo = wr << 6, i_hi = (i_lo + 64u) < n_items ? (i_lo + 64u) : n_items;  // items [i_lo, i_hi)
      const uint32_t i = i_lo + lane;
      if (i >= i_hi) continue;
      if (!fits) { item(TileTag<1>(), TileTag<1>(), TileTag<1>(), TileTag<1>(), i); continue; }
      if (i_hi > nAB) item(TileTag<1>(), TileTag<1>(), TileTag<1>(), TileTag<0>(), i);
      else if (i_hi <= nA) item(TileTag<0>(), TileTag<1>(), TileTag<0>(), TileTag<0>(), i);
      else if (i_lo >= nA) item(TileTag<0>(), TileTag<0>(), TileTag<1>(), TileTag<0>(), i);
      else item(TileTag<0>(), TileTag<1>(), TileTag<1>(), TileTag<0>(), i);
    }
    TILE_STAMP(t2);
    lds_barrier();  // the tiles are complete; nobody reads this block's owner map any more
    TILE_STAMP(t3);
    if (wave == 3u) {
      if (lane == 0) wr_next = 0u;
    } else if (fits) {
#if !defined(TILE_ABLATE_FLUSH)
      // ---- flush: aligned 16-byte chunks of both tiles, dealt to the 192 lanes of waves 0-2
      const uint32_t nq = (a0q + bytes + 15u) >> 4, ns = (a0s + bytes + 15u) >> 4;
      uint8_t* const gq = qual_blk - a0q;
      uint8_t* const gs = seq_blk - a0s;
      for (uint32_t c = threadIdx.x; c < nq + ns; c += 192u) {
        const bool is_s = c >= nq;
        const uint32_t cc = is_s ? c - nq : c;
        const uint32_t a0 = is_s ? a0s : a0q;
        const uint8_t* const src = (is_s ? stile : qtile) + 16u * cc;
        uint8_t* const dst = (is_s ? gs : gq) + 16u * cc;
        const uint4 val = *reinterpret_cast<const uint4*>(src);
        const uint32_t lo = 16u * cc < a0 ? a0 - 16u * cc : 0u;                    // first valid byte of the chunk
        const uint32_t end = a0 + bytes - 16u * cc, hi = end < 16u ? end : 16u;  // one past the last
        if (lo == 0u && hi == 16u) {
          *reinterpret_cast<uint4*>(dst) = val;
        } else {
          const uint64_t v_lo = (uint64_t)val.x | ((uint64_t)val.y << 32), v_hi = (uint64_t)val.z | ((uint64_t)val.w << 32);
          for (uint32_t j = lo; j < hi; j++) dst[j] = (uint8_t)((j < 8u ? v_lo : v_hi) >> (8u * (j & 7u)));
        }
      }
#endif
    }
    TILE_STAMP(t4);
    if (next < n_blocks) build_owner(cur ^ 1u);
    TILE_STAMP(t5);
    lds_barrier();  // the next block's map is written, the tiles are free
#if defined(TILE_DIAG)
    { const uint64_t t6 = __builtin_readcyclecounter();
      d_pro += t1 - t0; d_item += t2 - t1; d_wa += t3 - t2; d_flush += t4 - t3; d_own += t5 - t4; d_wb += t6 - t5; }
#endif
  }
#if defined(TILE_DIAG)
  if (lane == 0) {
    const uint32_t k = wave == 3u ? 8u : 0u;
    atomicAdd(&tile_diag[k + 0], d_pro); atomicAdd(&tile_diag[k + 1], d_item); atomicAdd(&tile_diag[k + 2], d_wa);
    atomicAdd(&tile_diag[k + 3], d_flush); atomicAdd(&tile_diag[k + 4], d_own); atomicAdd(&tile_diag[k + 5], d_wb);
    atomicAdd(&tile_diag[k + 6], 1ull);
  }
#endif
  qsum = qsum + 256ull * n_wrap - (uint64_t)qoff * p_bases;  // sum of the raw Phred values (modulo 2^64 per lane: the sum over lanes is exact)
  uint64_t acgt = HAS_EXC ? (uint64_t)n_acgt : p_bases;
  for (int d = 32; d > 0; d >>= 1) {
    n_subst += __shfl_down(n_subst, d, 64);
    acgt += __shfl_down(acgt, d, 64);
    qsum += __shfl_down(qsum, d, 64);
    p_bases += __shfl_down(p_bases, d, 64);
    p_redrawn += __shfl_down(p_redrawn, d, 64);
    p_seedsubst += __shfl_down(p_seedsubst, d, 64);
  }
  if (lane == 0 && counters) {
    shard_add(counters, SIMMR_CNT_SUBSTITUTIONS, (unsigned long long)n_subst);
    shard_add(counters, SIMMR_CNT_ACGT_BASES, (unsigned long long)acgt);
    shard_add(counters, SIMMR_CNT_QUAL_SUM, (unsigned long long)qsum);
    if (p_bases) shard_add(counters, SIMMR_CNT_BASES, (unsigned long long)p_bases);
    if (p_redrawn) shard_add(counters, SIMMR_CNT_REDRAWN, (unsigned long long)p_redrawn);
    if (p_seedsubst) shard_add(counters, SIMMR_CNT_SEED_SUBST, (unsigned long long)p_seedsubst);
    if (blockIdx.x == 0 && threadIdx.x == 0) shard_add(counters, SIMMR_CNT_READS, (unsigned long long)n_reads);
  }
}

}  // namespace simmr
