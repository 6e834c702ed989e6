/*
 * philox.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * CPU restatement of the SIMMR_RNG_PHILOX mode of include/simmr_hip.h — the
 * counter-based mode BASELINE.json's north_star prescribes for the per-base
 * draws (tolerance parity only; it is NOT the reference's generator).  The
 * specification lives here and in DESIGN.md §4:
 *   - Philox4x32-10 (Salmon et al., SC'11; Random123 constants), key = the
 *     read's Phred seed (pe_seed / drawn-or-substituted mate-2 seed / read_seed),
 *     counter = (b >> 1, 0, 'simm', 'r\0\0\1'); words 0,1 serve base b even,
 *     words 2,3 base b odd.
 *   - word A: Phred = alias-table sample of P(floor(N(mean,10)) sat. to u8):
 *       idx = A >> 24, frac = (A >> 8) & 0xffff, q = frac < thr[idx] ? idx : alias[idx]
 *   - word B: substitution iff (B >> 8) > floor(accuracy(q) * 2^24) — the same
 *     24-bit test as gen::<f32>() > accuracy (minimal_short.rs:119) — and the
 *     base is ACGT; the replacement is the k-th of the three other bases,
 *     k = ((((A & 0xff) << 8) | (B & 0xff)) * 3) >> 16.
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* table[i] = thr17 | alias << 24 with thr17 in [0, 65536] */
void orc_philox_phred_table(uint8_t mean_phred, uint32_t table[256]) {
  double P[256], cdf_prev = 0.0;
  const double mean = (double)mean_phred;
  for (int q = 0; q < 256; q++) {
    /* P(floor(mean + 10 z) saturated == q) */
    double upper = (q == 255) ? 1.0 : 0.5 * erfc(-(((double)(q + 1) - mean) / 10.0) / 1.4142135623730951);
    P[q] = upper - cdf_prev;
    if (P[q] < 0.0) P[q] = 0.0;
    cdf_prev = upper;
  }
  /* Vose alias method, worklists as LIFO stacks filled in increasing index order */
  double odds[256];
  int alias[256], smalls[256], bigs[256], ns = 0, nb = 0;
  for (int i = 0; i < 256; i++) { odds[i] = P[i] * 256.0; alias[i] = i; }
  for (int i = 0; i < 256; i++) { if (odds[i] < 1.0) smalls[ns++] = i; else bigs[nb++] = i; }
  while (ns > 0 && nb > 0) {
    int s = smalls[--ns], b = bigs[--nb];
    alias[s] = b;
    odds[b] = odds[b] - 1.0 + odds[s];
    if (odds[b] < 1.0) smalls[ns++] = b; else bigs[nb++] = b;
  }
  while (ns > 0) odds[smalls[--ns]] = 1.0;
  while (nb > 0) odds[bigs[--nb]] = 1.0;
  for (int i = 0; i < 256; i++) {
    double t = floor(odds[i] * 65536.0);
    uint32_t thr = t >= 65536.0 ? 65536u : (t <= 0.0 ? 0u : (uint32_t)t);
    table[i] = thr | ((uint32_t)alias[i] << 24);
  }
}

/* One read: qualities for bases [0, len) and the mutated copy of `seq` (forward-strand slice order). */
void orc_philox_read(const simmr_error_profile* p, const uint8_t* seq, uint64_t len, uint64_t key64,
                     uint8_t* qual_out, uint8_t* seq_out) {
  uint32_t table[256];
  orc_philox_phred_table(p->mean_phred, table);
  const uint32_t key[2] = {(uint32_t)key64, (uint32_t)(key64 >> 32)};
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint64_t b = 0; b < len; b++) {
    if ((b & 1) == 0) {
      const uint32_t ctr[4] = {(uint32_t)(b >> 1), 0u, 0x73696D6Du, 0x72000001u};
      orc_philox4x32_10(ctr, key, w);
    }
    const uint32_t A = w[(b & 1) * 2], B = w[(b & 1) * 2 + 1];
    const uint32_t e = table[A >> 24];
    const uint32_t q = (((A >> 8) & 0xffffu) < (e & 0x1ffffu)) ? (A >> 24) : (e >> 24);
    qual_out[b] = (uint8_t)q;
    uint8_t nt = seq[b];
    const uint32_t thr = (uint32_t)floorf(orc_convert_phred_to_accuracy((uint8_t)q) * 16777216.0f);
    int code = nt == 'A' ? 0 : nt == 'C' ? 1 : nt == 'G' ? 2 : nt == 'T' ? 3 : -1;
    if ((B >> 8) > thr && code >= 0) {
      uint32_t k = (((((A & 0xffu) << 8) | (B & 0xffu)) * 3u) >> 16);
      uint32_t alt = k + (k >= (uint32_t)code ? 1u : 0u);
      nt = (uint8_t)"ACGT"[alt];
    }
    seq_out[b] = nt;
  }
}
