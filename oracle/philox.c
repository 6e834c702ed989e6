/*
 * philox.c — ORACLE (test infrastructure only; see oracle.h).
 *
 * CPU restatement of the SIMMR_RNG_PHILOX mode of include/simmr_hip.h — the
 * counter-based mode BASELINE.json's north_star prescribes for the per-base
 * draws (tolerance parity only; it is NOT the reference's generator).  The
 * specification lives here and in DESIGN.md §4:
 *   - Philox4x32-10 (Salmon et al., SC'11; Random123 constants), key = the
 *     read's Phred seed (pe_seed / drawn-or-substituted mate-2 seed / read_seed),
 *     counter = (b >> 2, 0, 'simm', 'r\0\0\1'); base b takes output word b & 3.
 *   - that one word W draws the Phred score and the substitution together from
 *     the joint law of minimal_short.rs:83-140 with an alias table over the 1024
 *     outcomes o = q | s << 8:
 *       P(q)      = P(floor(N(mean,10)) saturated to u8 == q)                    (minimal profiles)
 *                 = P(round(-10 log10(1 - min(N(0.99, 0.05), 0.9999))) == q)      (perfect-long,
 *                   perfect_long.rs:60-78: q = 40 carries the mass of the 0.9999 cap)
 *       p_q       = P(gen::<f32>() > accuracy(q)) = (2^24 - 1 - t) / 2^24,
 *                   t = min(floor(accuracy(q) * 2^24), 2^24 - 1)   (the reference's 24-bit test)
 *       w(q, 0)   = P(q) (1 - p_q),   w(q, s) = P(q) p_q / 3 for s = 1, 2, 3
 *       idx = W >> 22, frac = W & 0x3fffff, o = frac < thr22[idx] ? idx : alias[idx]
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

/* table[i] = thr22 | alias << 22 with thr22 in [0, 2^22 - 1], i = q | s << 8 */
void orc_philox_joint_table(uint32_t kind, uint8_t mean_phred, uint32_t table[1024]) {
  enum { N = 1024 };
  double odds[N];
  int alias[N], smalls[N], bigs[N];
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
    odds[q] = P * (1.0 - pq) * (double)N;
    for (int s = 1; s < 4; s++) odds[q + 256 * s] = P * pq / 3.0 * (double)N;
  }
  /* Vose alias method, worklists as LIFO stacks filled in increasing index order */
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

/* One read: qualities for bases [0, len) and the mutated copy of `seq` (forward-strand slice order). */
void orc_philox_read(const simmr_error_profile* p, const uint8_t* seq, uint64_t len, uint64_t key64,
                     uint8_t* qual_out, uint8_t* seq_out) {
  static _Thread_local uint32_t table[1024];
  static _Thread_local int table_for = -1;
  const int want = (int)p->mean_phred | (p->kind == SIMMR_PERFECT_LONG ? 0x100 : 0);
  if (table_for != want) { orc_philox_joint_table(p->kind, p->mean_phred, table); table_for = want; }
  const uint32_t key[2] = {(uint32_t)key64, (uint32_t)(key64 >> 32)};
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint64_t b = 0; b < len; b++) {
    if ((b & 3) == 0) {
      const uint32_t ctr[4] = {(uint32_t)(b >> 2), 0u, 0x73696D6Du, 0x72000001u};
      orc_philox4x32_10(ctr, key, w);
    }
    const uint32_t W = w[b & 3];
    const uint32_t e = table[W >> 22];
    const uint32_t o = ((W & 0x3fffffu) < (e & 0x3fffffu)) ? (W >> 22) : (e >> 22);
    const uint32_t q = o & 0xffu, sft = o >> 8;
    qual_out[b] = (uint8_t)q;
    uint8_t nt = seq[b];
    int code = nt == 'A' ? 0 : nt == 'C' ? 1 : nt == 'G' ? 2 : nt == 'T' ? 3 : -1;
    if (sft != 0 && code >= 0) nt = (uint8_t)"ACGT"[((uint32_t)code + sft) & 3u];
    seq_out[b] = nt;
  }
}
