/*
 * philox.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * CPU restatement of the SIMMR_RNG_PHILOX mode of include/simmr_hip.h — the
 * counter-based mode BASELINE.json's north_star prescribes for the per-base
 * draws (tolerance parity only; it is NOT the reference's generator).  The
 * specification lives here and in DESIGN.md §4 (version 3 of the mode: a
 * two-level draw, 24 bits per base on the common outcomes and one full word on
 * the rare ones):
 *   - Philox4x32-10 (Salmon et al., SC'11; Random123 constants), key = the
 *     read's Phred seed (pe_seed / drawn-or-substituted mate-2 seed / read_seed).
 *   - the joint law of minimal_short.rs:83-140 over the 1024 outcomes o = q | s << 8:
 *       P(q)      = P(floor(N(mean,10)) saturated to u8 == q)                    (minimal profiles)
 *                 = P(round(-10 log10(1 - min(N(0.99, 0.05), 0.9999))) == q)      (perfect-long,
 *                   perfect_long.rs:60-78: q = 40 carries the mass of the 0.9999 cap)
 *       p_q       = P(gen::<f32>() > accuracy(q)) = (2^24 - 1 - t) / 2^24,
 *                   t = min(floor(accuracy(q) * 2^24), 2^24 - 1)   (the reference's 24-bit test)
 *       w(q, 0)   = P(q) (1 - p_q),   w(q, s) = P(q) p_q / 3 for s = 1, 2, 3
 *     is split exactly into  w(o) = c(o) / 2^24 + (E / 2^24) r(o):  c(o) = floor(2^24 w(o)) cells of a
 *     24-bit draw, E = 2^24 - sum c(o) "escape" cells (E ~ 120), r(o) = (2^24 w(o) - c(o)) / E the residual law.
 *   - level 1: 24 bits per base.  The 16 bases of group g = b >> 4 share the 384 bits of three calls, counters
 *     (3g, 0, 'simm', 'r\0\0\3'), (3g + 1, ...), (3g + 2, ...), output words in order = a little-endian bit
 *     string; base b takes bits [24 (b & 15), 24 (b & 15) + 24) = F.  F picks column idx = F >> 14 of a
 *     1024-column alias table whose thresholds count 16384ths: f = F & 0x3fff, outcome = f < T[idx] ? A[idx]
 *     : B[idx] (integer Vose construction over the cells c(o) and the E escape cells).
 *   - level 2, only when level 1 answered "escape" (7e-6 of the bases): counter (b >> 2, 1, 'simm', 'r\0\0\3'),
 *     output word b & 3 = W; idx = W >> 22, frac = W & 0x3fffff, o = frac < thr22[idx] ? idx : alias[idx] over
 *     the residual law r(o).
 *   - s > 0 and the base is ACGT: the base becomes "ACGT"[(code + s) & 3] — each of
 *     the three other bases with probability 1/3, as SliceRandom::choose does
 *     (minimal_short.rs:121-128); non-ACGT bases are left alone.
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

/* P(Phred == q) of the profile's quality law, in closed form */
static double phred_cdf_upper(uint32_t kind, double mean, int q) { /* P(Phred <= q) */
  if (q >= 255) return 1.0;
  if (kind == SIMMR_PERFECT_LONG) {
    /* q = round(-10 log10 d), d = 1 - min(acc, 0.9999), acc ~ N(0.99, 0.05):
     * Phred <= q  <=>  d > 10^-((q + 0.5) / 10)  <=>  acc < 1 - 10^-((q + 0.5) / 10); the cap puts everything
     * above 0.9999 on d = 1e-4, i.e. on q = 40 */
    if (q >= 40) return 1.0;
    const double acc_hi = 1.0 - pow(10.0, -((double)q + 0.5) / 10.0);
    return 0.5 * erfc(-((acc_hi - 0.99) / 0.05) / 1.4142135623730951);
  }
  /* floor(mean + 10 z) saturated: Phred <= q  <=>  mean + 10 z < q + 1 */
  return 0.5 * erfc(-(((double)(q + 1) - mean) / 10.0) / 1.4142135623730951);
}

/* w[o] = the joint law over o = q | s << 8 */
static void joint_law(uint32_t kind, uint8_t mean_phred, double w[1024]) {
  double cdf_prev = 0.0;
  const double mean = (double)mean_phred;
  for (int q = 0; q < 256; q++) {
    double upper = phred_cdf_upper(kind, mean, q);
    double P = upper - cdf_prev;
    if (P < 0.0) P = 0.0;
    cdf_prev = upper;
    float tf = floorf(orc_convert_phred_to_accuracy((uint8_t)q) * 16777216.0f);
    double t = tf > 16777215.0f ? 16777215.0 : (double)tf;
    double pq = (16777215.0 - t) / 16777216.0;
    w[q] = P * (1.0 - pq);
    for (int s = 1; s < 4; s++) w[q + 256 * s] = P * pq / 3.0;
  }
}

/* Vose alias method over 1024 columns of odds[] (mean 1), worklists as LIFO stacks filled in increasing
 * index order; table[i] = thr22 | alias << 22 with thr22 in [0, 2^22 - 1] */
static void alias22(double odds[1024], uint32_t table[1024]) {
  enum { N = 1024 };
  int alias[N], smalls[N], bigs[N];
  int ns = 0, nb = 0;
  for (int i = 0; i < N; i++) alias[i] = i;
  for (int i = 0; i < N; i++) { if (odds[i] < 1.0) smalls[ns++] = i; else bigs[nb++] = i; }
  while (ns > 0 && nb > 0) {
    int s = smalls[--ns], b = bigs[--nb];
    alias[s] = b;
    odds[b] = odds[b] - 1.0 + odds[s];
    if (odds[b] < 1.0) smalls[ns++] = b; else bigs[nb++] = b;
  }
  while (ns > 0) odds[smalls[--ns]] = 1.0;
  while (nb > 0) odds[bigs[--nb]] = 1.0;
  for (int i = 0; i < N; i++) {
    double t = floor(odds[i] * 4194304.0);
    uint32_t thr = t >= 4194303.0 ? 4194303u : (t <= 0.0 ? 0u : (uint32_t)t);
    table[i] = thr | ((uint32_t)alias[i] << 22);
  }
}

/* The two tables of the mode.
 *   t1[col] = T | A << 16 | B << 32: T in [0, 16384] 16384ths of the column that answer A, the rest answer B;
 *             A, B = outcome q | s << 8, or ORC_PHILOX_ESC (1024) = "draw again at level 2"
 *   t2[idx] = thr22 | alias << 22 over the residual law (all zero when there is no escape cell)
 * returns E, the number of escape cells among the 2^24 */
uint32_t orc_philox_tables(uint32_t kind, uint8_t mean_phred, uint64_t t1[1024], uint32_t t2[1024]) {
  enum { N = 1024, CELLS = 16777216, UNIT = 16384 };
  double w[N], odds[N];
  int64_t c[N];
  joint_law(kind, mean_phred, w);
  int64_t sum = 0;
  int nz = 0;
  for (int o = 0; o < N; o++) {
    c[o] = (int64_t)floor(w[o] * (double)CELLS);
    sum += c[o];
    if (c[o] > 0) nz++;
  }
  int64_t E = (int64_t)CELLS - sum;
  /* a column per outcome with cells, one more for the escape: give up the smallest outcomes if that is too many */
  while (E < 0 || nz + (E > 0 ? 1 : 0) > N) {
    int m = -1;
    for (int o = 0; o < N; o++) if (c[o] > 0 && (m < 0 || c[o] < c[m])) m = o;
    E += c[m]; c[m] = 0; nz--;
  }
  /* integer Vose: column k < n holds entry k (outcomes with cells in increasing order, then the escape) */
  int64_t wt[N];
  uint32_t prim[N];
  int alias[N], smalls[N], bigs[N], T[N];
  int n = 0;
  for (int o = 0; o < N; o++) if (c[o] > 0) { prim[n] = (uint32_t)o; wt[n] = c[o]; n++; }
  if (E > 0) { prim[n] = ORC_PHILOX_ESC; wt[n] = E; n++; }
  for (int k = n; k < N; k++) { prim[k] = prim[0]; wt[k] = 0; }
  int ns = 0, nb = 0;
  for (int k = 0; k < N; k++) { alias[k] = k; T[k] = UNIT; }
  for (int k = 0; k < N; k++) { if (wt[k] < UNIT) smalls[ns++] = k; else bigs[nb++] = k; }
  while (ns > 0 && nb > 0) {
    int s = smalls[--ns], b = bigs[--nb];
    alias[s] = b;
    T[s] = (int)wt[s];
    wt[b] -= UNIT - wt[s];
    if (wt[b] < UNIT) smalls[ns++] = b; else bigs[nb++] = b;
  }
  for (int k = 0; k < N; k++) t1[k] = (uint64_t)T[k] | ((uint64_t)prim[k] << 16) | ((uint64_t)prim[alias[k]] << 32);
  /* residual law */
  if (E > 0) {
    for (int o = 0; o < N; o++) odds[o] = (w[o] * (double)CELLS - (double)c[o]) / (double)E * (double)N;
    alias22(odds, t2);
  } else {
    memset(t2, 0, N * sizeof(uint32_t));
  }
  return (uint32_t)E;
}

/* One read: qualities for bases [0, len) and the mutated copy of `seq` (forward-strand slice order). */
void orc_philox_read(const simmr_error_profile* p, const uint8_t* seq, uint64_t len, uint64_t key64,
                     uint8_t* qual_out, uint8_t* seq_out) {
  static _Thread_local uint64_t t1[1024];
  static _Thread_local uint32_t t2[1024];
  static _Thread_local int table_for = -1;
  const int want = (int)p->mean_phred | (p->kind == SIMMR_PERFECT_LONG ? 0x100 : 0);
  if (table_for != want) { orc_philox_tables(p->kind, p->mean_phred, t1, t2); table_for = want; }
  const uint32_t key[2] = {(uint32_t)key64, (uint32_t)(key64 >> 32)};
  uint32_t w[13] = {0};  /* the 12 words of a group, one more so that a 64-bit window can be read at word 11 */
  for (uint64_t b = 0; b < len; b++) {
    if ((b & 15) == 0) {
      for (uint32_t c = 0; c < 3; c++) {
        const uint32_t ctr[4] = {(uint32_t)(3 * (b >> 4) + c), 0u, 0x73696D6Du, 0x72000003u};
        orc_philox4x32_10(ctr, key, w + 4 * c);
      }
    }
    const uint32_t bit = 24u * (uint32_t)(b & 15);
    const uint64_t win = (uint64_t)w[bit >> 5] | ((uint64_t)w[(bit >> 5) + 1] << 32);
    const uint32_t F = (uint32_t)(win >> (bit & 31)) & 0xffffffu;
    const uint64_t e = t1[F >> 14];
    const uint32_t f = F & 0x3fffu;
    uint32_t o = f < (uint32_t)(e & 0xffffu) ? (uint32_t)(e >> 16) & 0xffffu : (uint32_t)(e >> 32);
    if (o == ORC_PHILOX_ESC) {
      const uint32_t ctr2[4] = {(uint32_t)(b >> 2), 1u, 0x73696D6Du, 0x72000003u};
      uint32_t w2[4];
      orc_philox4x32_10(ctr2, key, w2);
      const uint32_t W = w2[b & 3];
      const uint32_t e2 = t2[W >> 22];
      o = ((W & 0x3fffffu) < (e2 & 0x3fffffu)) ? (W >> 22) : (e2 >> 22);
    }
    const uint32_t q = o & 0xffu, sft = o >> 8;
    qual_out[b] = (uint8_t)q;
    uint8_t nt = seq[b];
    int code = nt == 'A' ? 0 : nt == 'C' ? 1 : nt == 'G' ? 2 : nt == 'T' ? 3 : -1;
    if (sft != 0 && code >= 0) nt = (uint8_t)"ACGT"[((uint32_t)code + sft) & 3u];
    seq_out[b] = nt;
  }
}
