// engine.hip — host side of the C ABI declared in include/simmr_hip.h.
//
// Owns device memory for staged genomes, the plan of the current shard and the
// scratch of the outer-stream transducer; enqueues the kernels of kernels.hip
// on one HIP stream.  No CPU compute path exists here: every entry point that
// produces reads needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <mutex>
#include <vector>

#include "kernels.hip"
#include "text_lines.hip"
#include "fastq_kernels.hip"
#include "custom_model.hpp"

using namespace simmr;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
    cap = want;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct GenomeHost {
  bool staged = false;
  std::vector<ContigDev> contigs;
  uint64_t total_size = 0;   // sum of Seq.size (Genome.size, genome.rs:135)
  uint64_t plane_bases = 0;  // bases in the packed plane (contigs padded to 64)
  uint64_t max_size = 0;
  bool has_exc = false;
  DevBuf packed, mask, d_contigs;
};

constexpr size_t FRONT_PAD_WORDS = 4;  // 16 bytes in front of each plane
constexpr size_t BACK_PAD_WORDS = 24;  // the custom-long splice reads 17 words from the word of base (read end + k - 1)

enum PlanKind { PLAN_NONE = 0, PLAN_PE = 1, PLAN_LONG = 2 };

}  // namespace

// which form simmr_emit_fastq runs when SIMMR_TEXT_FORM does not say: the one that measures faster (LAB.md, round 5)
#ifndef TEXT_FORM_DEFAULT
#define TEXT_FORM_DEFAULT 1
#endif
struct simmr_engine {
  int device = -1;
  int n_cu = 0;
  hipStream_t stream = nullptr;
  void* comm = nullptr;  // ncclComm_t of simmr_comm_init (RCCL, bound at run time)
  std::string err;
  std::vector<GenomeHost> genomes;
  DevBuf d_genomes;  // GenomeDev[genomes.size()]
  DevBuf d_tables, d_counters, d_err, d_scalars;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  // The event pairs of the last emits (ev_c / ev_d = the pair of the emit in progress): simmr_emit_kernel_ms_mean averages them
  // with ONE synchronisation, where asking after every emit (simmr_last_emit_kernel_ms) would wait for each.
  static constexpr int EMIT_RING = 64;
  hipEvent_t ring_c[EMIT_RING] = {}, ring_d[EMIT_RING] = {};
  uint64_t n_emits = 0;
  float last_emit_ms = 0.f, last_plan_ms = 0.f, last_fastq_plan_ms = 0.f;

  // current plan
  int plan_kind = PLAN_NONE;
  ProfileDev prof{};
  uint32_t plan_genome = 0;
  uint64_t plan_first = 0, plan_units = 0, plan_total_bases = 0;
  uint32_t read_slots = 0;  // simmr_engine_set_read_slots: the layout of the plans to come (0 compact, 16 SIMMR_SLOT16)
  uint32_t plan_slot = 0;   // ... and of the plan in force
  bool plan_short_ok = false;  // the current paired plan has no read longer than LONGREAD_MAXL (text_lines.hip takes it)
  bool plan_coarse = false; // pairs for the counter-mode kernel: u_off64 (first byte of every 64th pair) instead of u_off
  DevBuf w_bytes, u_off64, fq_off64;
  uint32_t splice_lds_set[2] = {0, 0};  // dynamic-LDS limit already set on this device for k_custom_long_splice<exc, fast>
  uint32_t splice_ctr_lds_set[2] = {0, 0};  // the same for the counter mode's instantiations <exc, fast, CTR>
  bool fq_coarse = false;  // the direct FASTQ plan in force has fq_off64 (first byte of every 64th record) instead of fq_off
  bool plan_paired = false;
  bool plan_multi = false;    // paired-end plan over several genomes (u_genome per pair)
  bool plan_any_exc = false;  // some genome of the plan has an exception plane
  DevBuf m_genomes, m_contig, m_seed;
  DevBuf u_contig, u_genome, u_seed, u_len, u_a, u_b, u_qs2, u_ms2, u_flags, u_off;
  DevBuf scan_tmp, u_order, len_hist;
  bool plan_sorted = false;
  // simmr_engine_set_plan_overlap: the plan calls run on a stream of their own and write a SECOND set of the buffers an
  // emit reads (plan columns, offsets, order, error word, counter-mode tables), so that the plan of the next shard runs
  // while the emit of this one is still on the device — one is bound by latency, the other by VALU issue.  plan_sets_swap
  // exchanges the two sets; mark[s] = the work on the caller's stream that read set s when set s was last given up.
  bool overlap = false;
  hipStream_t plan_stream = nullptr;
  hipEvent_t ev_plan_done = nullptr, ev_mark[2] = {nullptr, nullptr};
  bool mark_valid[2] = {false, false};
  int cur_set = 0;
  DevBuf s_w_bytes, s_u_off64, s_m_genomes, s_u_contig, s_u_genome, s_u_seed, s_u_len, s_u_a, s_u_b, s_u_qs2, s_u_ms2, s_u_flags,
      s_u_off, s_u_order, s_d_err, s_d_runs, s_d_usable, s_ph_table;
  // measurement knobs, read ONCE when the engine is made (a stray variable cannot change a running engine's launches)
  bool tl_debug = false;                // SIMMR_TL_DEBUG: print the occupancy of k_emit_text_lines launches
  bool inject_null_ctr_tables = false;  // SIMMR_FAULT_INJECT=null_ctr_tables: test switch, see custom_long_tables_missing
  int text_form = TEXT_FORM_DEFAULT;  // SIMMR_TEXT_FORM: 1 = the item form (k_emit_philox<TEXT>) always, 2 = the whole-line kernel (text_lines.hip) wherever it applies: same-box A/B
  uint32_t philox_wgs_per_cu = 128;  // SIMMR_PHILOX_WGS_PER_CU (item kernel), clamped to 1..4096.  More workgroups than the 4 per CU that
                                    // are resident: 11.25 ms at 8, 10.6 at 32, 10.4 at 64-256, 11.1 at one block per workgroup
                                    // (profiles/r3/ab_wgs_per_cu_*: the workgroups of a CU stop running their phases in step)
  // The other grid-stride emit kernels: workgroups per CU as a multiple of what is resident.  Every one of them ran its
  // best with far more workgroups than fit at a time (profiles/r3/ab_grid_sweep.log: k_emit_lanes 107.6 ms at 1 x, 98.5 at
  // 4 x, 94.7 at 64 x; k_emit_perfect_pe 6.65 / 5.99 / 5.78; k_emit_custom_pe 10.99 / 10.58 / 10.44 at 16 x; the splice
  // kernel does not care).  SIMMR_GRID_MULT overrides all of them (measurement knob, 1..512).
  uint32_t lanes_mult = 64, perfect_mult = 64, custom_pe_mult = 16, custom_long_mult = 1, fastq_mult = 4;  // (k_fastq_write: 34.6 / 32.8 / 33.2 / 33.4 / 34.4 ms per step at 1 / 4 / 16 / 64 / 256 x, profiles/r3/fastq_grid_probe.log)
  int splice_variant = 0;       // SIMMR_SPLICE_VARIANT: 1 = the two-load splice kernel on every model
  // outer-stream scratch
  DevBuf o_last_idx, o_wg_sums, o_wg_prefix, o_result;
  // long-read runs
  DevBuf d_runs, d_usable;
  // custom profile tables
  DevBuf c_pdfs, c_odds, c_alias, c_low, c_range, c_zone, c_colrec, c_binrec, c_kslots, c_krecs, c_kdirect, c_kcnt8, c_kcols, c_krecs_ctr, c_kcols_ctr, c_ktab32, ph_table;
  // the custom model whose tables those buffers hold (make_custom_profile)
  bool custom_cached = false, custom_long = false;
  uint64_t custom_hash = 0, custom_bytes = 0;
  std::vector<uint8_t> custom_model_copy;  // the bytes those tables were built from (a hash match is confirmed with memcmp)
  ProfileDev custom_prof{};
  // FASTQ framing
  DevBuf fq_blob, fq_gid_off, fq_gid_len, fq_cbase, fq_ncontig, fq_coff, fq_clen, fq_len, fq_off;
  FqTemplate fq_tpl{};
  uint64_t fq_reads = 0, fq_total = 0;
  uint32_t fq_slots = 0, fq_lit_bytes = 0, fq_hpitch = 272;
  bool fq_paired = false, fq_ready = false;
  bool fq_direct = false;          // planned by simmr_fastq_plan_direct (sizes from the plan, for simmr_emit_fastq)
  uint32_t fq_read_id_base = 0, fq_maxhdr = 0;
  uint32_t text_lines_lds_set[16] = {0};  // dynamic LDS granted to each instantiation of k_emit_text_lines so far
  DevBuf fq_hlen;                  // header bytes per read (direct form)
  DevBuf fq_tpl_dev;               // the compiled header template, read by the kernels through a pointer
  DevBuf fd_seq, fd_qual, fd_seq_off, fd_start, fd_end, fd_contig, fd_genome, fd_read_id, fd_flags;  // columns of the unfused fallback

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

#define HIP_TRY(e, call)                                                                  \
  do {                                                                                    \
    hipError_t _s = (call);                                                               \
    if (_s != hipSuccess)                                                                 \
      return (e)->fail(SIMMR_ENODEV, "%s failed: %s", #call, hipGetErrorString(_s));      \
  } while (0)

namespace {

// ---- host copies of the small pieces of arithmetic the host itself needs ----
Key8 host_pcg32_expand(uint64_t state) {
  const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
  Key8 key;
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key.k[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  return key;
}

uint8_t host_sat_u8(float f) {
  if (!(f == f) || f <= 0.0f) return 0;
  if (f >= 255.0f) return 255;
  return (uint8_t)f;
}
// util.rs:109-111 applied to d = 1 - acc, with the host libm
uint8_t host_phred_of_err(float d) { return host_sat_u8(roundf(-10.0f * log10f(d))); }

void build_tables(Tables* T) {
  // rand_distr 0.4.3 ziggurat tables for N(0,1), from the crate's generator
  // formulas (r = 3.654152885361009, v = 0.00492867323399, 256 layers)
  const double r = 3.6541528853610088, v = 0.00492867323399;
  auto f = [](double x) { return exp(-x * x / 2.0); };
  T->zig_x[0] = v / f(r);
  T->zig_x[1] = r;
  for (int i = 2; i < 256; i++) {
    double last = T->zig_x[i - 1];
    T->zig_x[i] = sqrt(-2.0 * log(v / last + f(last)));
  }
  T->zig_x[256] = 0.0;
  for (int i = 0; i < 257; i++) T->zig_f[i] = f(T->zig_x[i]);
  // util.rs:69-71,96-98 with the platform libm, as the Rust std does
  for (int q = 0; q < 256; q++) T->acc[q] = 1.0f - powf(10.0f, -((float)q / 10.0f));
  // perfect-long Phred as thresholds on d = 1 - acc: q(d) = pl_first + #{i: d <= thresh[i]}
  T->pl_first = host_phred_of_err(3.0e38f);
  T->pl_count = 0;
  for (uint32_t q = T->pl_first + 1; q < T->pl_first + 64; q++) {
    // largest positive float d with phred(d) >= q (phred is non-increasing in d)
    uint32_t lo = 1u, hi = 0x7f7fffffu;  // bit patterns of positive finite floats
    float flo; memcpy(&flo, &lo, 4);
    if (host_phred_of_err(flo) < q) break;
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo + 1) / 2;
      float fm; memcpy(&fm, &mid, 4);
      if (host_phred_of_err(fm) >= q) lo = mid; else hi = mid - 1;
    }
    float t; memcpy(&t, &lo, 4);
    T->pl_thresh[T->pl_count++] = t;
  }
  for (uint32_t i = T->pl_count; i < 64; i++) T->pl_thresh[i] = -1.0f;
}

inline uint32_t grid_for(uint64_t n, uint32_t per_block) {
  uint64_t g = (n + per_block - 1) / per_block;
  if (g == 0) g = 1;
  return (uint32_t)g;
}

int sync_check(simmr_engine* e, const char* what) {
  hipError_t s = hipStreamSynchronize(e->stream);
  if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "%s: %s", what, hipGetErrorString(s));
  s = hipGetLastError();
  if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "%s: %s", what, hipGetErrorString(s));
  return SIMMR_OK;
}

int refresh_genome_table(simmr_engine* e) {
  std::vector<GenomeDev> tab(e->genomes.size());
  for (size_t i = 0; i < e->genomes.size(); i++) {
    GenomeHost& g = e->genomes[i];
    GenomeDev d{};
    if (g.staged) {
      d.packed = g.packed.as<uint32_t>() + FRONT_PAD_WORDS;
      d.mask = g.has_exc ? g.mask.as<uint32_t>() + FRONT_PAD_WORDS : nullptr;
      d.contigs = g.d_contigs.as<ContigDev>();
      d.n_contigs = (uint32_t)g.contigs.size();
      d.has_exc = g.has_exc ? 1u : 0u;
    }
    tab[i] = d;
  }
  if (!e->d_genomes.ensure(sizeof(GenomeDev) * std::max<size_t>(tab.size(), 1)))
    return e->fail(SIMMR_ENOMEM, "genome table allocation failed");
  HIP_TRY(e, hipMemcpyAsync(e->d_genomes.p, tab.data(), sizeof(GenomeDev) * tab.size(),
                            hipMemcpyHostToDevice, e->stream));
  return sync_check(e, "genome table upload");
}

int layout_genome(simmr_engine* e, GenomeHost& g, uint32_t n_contigs, const uint64_t* contig_len,
                  const uint64_t* contig_size) {
  g.contigs.resize(n_contigs);
  uint64_t base = 0;
  g.total_size = 0;
  g.max_size = 0;
  for (uint32_t c = 0; c < n_contigs; c++) {
    g.contigs[c].base = base;
    g.contigs[c].len = contig_len[c];
    g.contigs[c].size = contig_size ? contig_size[c] : contig_len[c];
    g.total_size += g.contigs[c].size;
    g.max_size = std::max(g.max_size, g.contigs[c].size);
    base += (contig_len[c] + 63) & ~63ULL;
  }
  g.plane_bases = base;
  const size_t pwords = FRONT_PAD_WORDS + base / 16 + BACK_PAD_WORDS;
  const size_t mwords = FRONT_PAD_WORDS + base / 32 + BACK_PAD_WORDS;
  if (!g.packed.ensure(pwords * 4) || !g.mask.ensure(mwords * 4) ||
      !g.d_contigs.ensure(sizeof(ContigDev) * std::max<uint32_t>(n_contigs, 1)))
    return e->fail(SIMMR_ENOMEM, "genome allocation failed (%zu bytes)", pwords * 4 + mwords * 4);
  HIP_TRY(e, hipMemsetAsync(g.packed.p, 0, pwords * 4, e->stream));
  HIP_TRY(e, hipMemsetAsync(g.mask.p, 0, mwords * 4, e->stream));
  HIP_TRY(e, hipMemcpyAsync(g.d_contigs.p, g.contigs.data(), sizeof(ContigDev) * n_contigs,
                            hipMemcpyHostToDevice, e->stream));
  return SIMMR_OK;
}

int check_genome(simmr_engine* e, uint32_t idx) {
  if (idx >= e->genomes.size() || !e->genomes[idx].staged)
    return e->fail(SIMMR_EINVAL, "genome %u is not staged", idx);
  return SIMMR_OK;
}

// ---- custom (empirical) profile: parse the model, build + upload the PDF tables --------
template <class T> int upload_vec(simmr_engine* e, DevBuf& b, const std::vector<T>& v) {
  if (!b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T))) return e->fail(SIMMR_ENOMEM, "custom table allocation failed");
  if (!v.empty()) HIP_TRY(e, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, e->stream));
  return SIMMR_OK;
}

// 64 bits over the model's bytes (FNV-1a on 8-byte words): the key of the table cache below
uint64_t model_hash(const void* p, uint64_t n) {
  const uint8_t* b = (const uint8_t*)p;
  uint64_t h = 0xcbf29ce484222325ULL ^ n;
  uint64_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001b3ULL; h ^= h >> 29; }
  for (; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ULL;
  return h;
}

// ---- the plan of the next shard beside the emit of this one (simmr_engine_set_plan_overlap) ----
static void plan_sets_swap(simmr_engine* e) {
  std::swap(e->w_bytes, e->s_w_bytes); std::swap(e->u_off64, e->s_u_off64); std::swap(e->m_genomes, e->s_m_genomes);
  std::swap(e->u_contig, e->s_u_contig); std::swap(e->u_genome, e->s_u_genome); std::swap(e->u_seed, e->s_u_seed);
  std::swap(e->u_len, e->s_u_len); std::swap(e->u_a, e->s_u_a); std::swap(e->u_b, e->s_u_b); std::swap(e->u_qs2, e->s_u_qs2);
  std::swap(e->u_ms2, e->s_u_ms2); std::swap(e->u_flags, e->s_u_flags); std::swap(e->u_off, e->s_u_off);
  std::swap(e->u_order, e->s_u_order); std::swap(e->d_err, e->s_d_err); std::swap(e->d_runs, e->s_d_runs);
  std::swap(e->d_usable, e->s_d_usable); std::swap(e->ph_table, e->s_ph_table);
  e->cur_set ^= 1;
}
// INVARIANT the overlap rests on: only the eighteen buffers swapped above exist twice.  The scratch that plan
// calls and emits share WITHOUT a second copy — scan_tmp, len_hist, the outer-stream scratch o_*, d_scalars, m_contig / m_seed,
// the FASTQ sizing buffers fq_* and the custom model's tables c_* — is safe only because every user of it ends in a
// host-side synchronisation of its stream before it returns: every plan call (read_err_word / sync_check), simmr_fastq_plan
// and simmr_fastq_plan_direct (the size readback), simmr_outer_summarize, and a custom emit (it reads the error word
// back).  A new user of that scratch that returns without synchronising breaks the overlap silently.
// tests/test_gpu_shapes.py::test_plan_overlap_gives_the_same_reads runs plan -> fastq_plan_direct -> emit_fastq back to back.
// One per plan call.  With the overlap on: marks what the caller's stream has been given so far (all of it may read the
// set in force), takes the other set, lets the plan stream wait for whatever read THAT set when it was given up, and
// makes the plan stream the engine's stream for the duration of the call; on the way out — whichever way — the caller's
// stream is put back and made to wait for the plan.
struct PlanScope {
  simmr_engine* e;
  hipStream_t main = nullptr;
  bool on = false;
  explicit PlanScope(simmr_engine* eng) : e(eng) {
    if (!e->overlap || !e->plan_stream) return;
    main = e->stream;
    if (hipEventRecord(e->ev_mark[e->cur_set], main) != hipSuccess) return;
    e->mark_valid[e->cur_set] = true;
    plan_sets_swap(e);
    if (e->mark_valid[e->cur_set]) (void)hipStreamWaitEvent(e->plan_stream, e->ev_mark[e->cur_set], 0);
    if (!e->d_err.p && (!e->d_err.ensure(64) || hipMemset(e->d_err.p, 0, 64) != hipSuccess)) { /* the plan's own checks report it */ }
    e->stream = e->plan_stream;
    on = true;
  }
  ~PlanScope() {
    if (!on) return;
    (void)hipEventRecord(e->ev_plan_done, e->plan_stream);
    e->stream = main;
    (void)hipStreamWaitEvent(main, e->ev_plan_done, 0);
  }
};

int make_custom_profile(simmr_engine* e, const simmr_error_profile* p, bool want_long, ProfileDev* out) {
  if (!p->custom_model || p->custom_model_bytes == 0) return e->fail(SIMMR_EINVAL, "custom profile without a model");
  // A job plans many shards with one model: parsing it, building the alias and k-mer tables and uploading them
  // (4 ms for the benchmark's model) is done once per model, not once per plan.  The tables live in buffers only
  // this function writes.
  const uint64_t mh = model_hash(p->custom_model, p->custom_model_bytes);
  if (e->custom_cached && e->custom_hash == mh && e->custom_bytes == p->custom_model_bytes && e->custom_long == want_long &&
      e->custom_model_copy.size() == p->custom_model_bytes &&
      memcmp(e->custom_model_copy.data(), p->custom_model, p->custom_model_bytes) == 0) {
    ProfileDev d = e->custom_prof;
    if (want_long) {
      if (p->long_start_mode > SIMMR_START_UNIFORM) return e->fail(SIMMR_EINVAL, "unknown long_start_mode %u", p->long_start_mode);
      d.long_start_uniform = p->long_start_mode == SIMMR_START_UNIFORM ? 1u : 0u;
    }
    *out = d;
    return SIMMR_OK;
  }
  e->custom_cached = false;
  // (the tables below are rewritten: with the plan overlap on, an emit of the previous model may still be reading them)
  if (e->overlap) (void)hipDeviceSynchronize();
  ModelHost m;
  std::string err;
  if (!parse_model((const uint8_t*)p->custom_model, p->custom_model_bytes, &m, &err)) return e->fail(SIMMR_EINVAL, "%s", err.c_str());
  // main.rs:30-33: a custom-short profile must not be a long-read model.  The long-read path takes the
  // profile through is_long_read() (custom_short.rs:540-542, main.rs:180-186), i.e. only an is_long model.
  if (m.is_long && !want_long)
    return e->fail(SIMMR_EINVAL, "You specified a custom short-read error profile but the provided error profile is for long reads");
  if (!m.is_long && want_long)
    return e->fail(SIMMR_EINVAL, "a short-read custom model was passed to the long-read path (is_long_read() is false)");
  if (m.quality.empty()) return e->fail(SIMMR_EINVAL, "custom model has no quality distributions");
  PdfTables t;
  if (!append_pdf(m.read_length_bins, &t, &err)) return e->fail(SIMMR_EINVAL, "%s", err.c_str());
  if (!append_pdf(m.has_insert_bins ? m.insert_bins : BinsHost(), &t, &err)) return e->fail(SIMMR_EINVAL, "%s", err.c_str());
  for (const BinsHost& b : m.quality) if (!append_pdf(b, &t, &err)) return e->fail(SIMMR_EINVAL, "%s", err.c_str());
  if (t.pdfs[0].n == 0) return e->fail(SIMMR_EINVAL, "custom model has an empty read-length distribution");
  for (size_t i = 2; i < t.pdfs.size(); i++) if (t.pdfs[i].n == 0) return e->fail(SIMMR_EINVAL, "custom model has an empty quality distribution");
  // every host table is built before the first (asynchronous) upload
  KmerTables kt;
  if (want_long && !build_kmer_tables(m, &kt, &err)) return e->fail(SIMMR_ENOTSUP, "%s", err.c_str());
  if (want_long && !std::isfinite(m.read_length_std))
    return e->fail(SIMMR_ERANGE, "custom model: read_length_std is not finite (Normal::new(..).unwrap() panics)");
  // the emit kernel's copies: one 16-byte record per alias column / per bin
  std::vector<Rec16> colrec(t.odds.size()), binrec(t.bin_low.size());
  for (size_t i = 0; i < t.odds.size(); i++) {
    uint64_t bits;
    memcpy(&bits, &t.odds[i], 8);
    colrec[i] = Rec16{(uint32_t)bits, (uint32_t)(bits >> 32), t.alias[i], 0u};
  }
  for (size_t i = 0; i < t.bin_low.size(); i++) binrec[i] = Rec16{t.bin_range[i], t.bin_zone[i], t.bin_low[i], 0u};
  // declared after the vectors, so it runs before they die: whatever path leaves this function, no copy is still reading them
  struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{e->stream};
  int rc;
  if ((rc = upload_vec(e, e->c_pdfs, t.pdfs)) || (rc = upload_vec(e, e->c_odds, t.odds)) ||
      (rc = upload_vec(e, e->c_alias, t.alias)) || (rc = upload_vec(e, e->c_low, t.bin_low)) ||
      (rc = upload_vec(e, e->c_range, t.bin_range)) || (rc = upload_vec(e, e->c_zone, t.bin_zone)))
    return rc;
  if ((rc = upload_vec(e, e->c_colrec, colrec)) || (rc = upload_vec(e, e->c_binrec, binrec))) return rc;
  if (want_long) {  // simulate_errors only runs for long reads (simulate.rs:500)
    if ((rc = upload_vec(e, e->c_kslots, kt.slots)) || (rc = upload_vec(e, e->c_krecs, kt.recs)) ||
        (rc = upload_vec(e, e->c_kdirect, kt.direct)))
      return rc;
    if (kt.stride && ((rc = upload_vec(e, e->c_kcnt8, kt.cnt8)) || (rc = upload_vec(e, e->c_kcols, kt.cols)))) return rc;
    if ((rc = upload_vec(e, e->c_krecs_ctr, kt.recs_ctr))) return rc;  // the counter mode's level-2 columns
    if (kt.stride && ((rc = upload_vec(e, e->c_kcols_ctr, kt.cols_ctr)) || (rc = upload_vec(e, e->c_ktab32, kt.tab32)))) return rc;
  }
  if ((rc = sync_check(e, "custom table upload"))) return rc;  // the host vectors go out of scope
  ProfileDev d{};
  d.kind = SIMMR_K_CUSTOM;
  d.rng_mode = SIMMR_RNG_REFERENCE;  // (make_profile puts the caller's mode in: the tables are the same for both)
  // custom_short.rs:535-538: (2.0 * read_length_mean + insert_size_mean) as u16 (saturating)
  const double req = 2.0 * m.read_length_mean + m.insert_size_mean;
  d.required = !(req == req) || req <= 0.0 ? 0u : (req >= 65535.0 ? 65535u : (uint32_t)req);
  d.custom.pdfs = e->c_pdfs.as<PdfDev>();
  d.custom.odds = e->c_odds.as<double>();
  d.custom.alias = e->c_alias.as<uint32_t>();
  d.custom.bin_low = e->c_low.as<uint32_t>();
  d.custom.bin_range = e->c_range.as<uint32_t>();
  d.custom.bin_zone = e->c_zone.as<uint32_t>();
  d.custom.col_rec = e->c_colrec.as<Rec16>();
  d.custom.bin_rec = e->c_binrec.as<Rec16>();
  d.custom.n_quality = (uint32_t)m.quality.size();
  if (want_long) {
    // get_random_read_length (custom_short.rs:286-301)
    d.read_length_std = m.read_length_std;
    d.insert_size_std = m.read_length_mean;  // k_const_length reads the mean from this slot
    if (p->long_start_mode > SIMMR_START_UNIFORM) return e->fail(SIMMR_EINVAL, "unknown long_start_mode %u", p->long_start_mode);
    d.long_start_uniform = p->long_start_mode == SIMMR_START_UNIFORM ? 1u : 0u;
    d.custom.kmer_direct = e->c_kdirect.as<Rec16>();
    d.custom.kmer_slots = e->c_kslots.as<Rec16>();
    d.custom.kmer_recs = e->c_krecs.as<Rec16>();
    d.custom.kmer_mask = kt.mask;
    d.custom.kmer_size = (uint32_t)m.kmer_size;
    d.custom.kmer_stride = kt.stride;
    d.custom.kmer_cnt8 = kt.stride ? e->c_kcnt8.as<uint8_t>() : nullptr;
    d.custom.kmer_cols = kt.stride ? e->c_kcols.as<Rec16>() : nullptr;
    d.custom.kmer_recs_ctr = e->c_krecs_ctr.as<Rec16>();
    d.custom.kmer_cols_ctr = kt.stride ? e->c_kcols_ctr.as<Rec16>() : nullptr;
    d.custom.kmer_tab32 = kt.stride ? e->c_ktab32.as<uint32_t>() : nullptr;
  }
  if (e->inject_null_ctr_tables) {  // SIMMR_FAULT_INJECT=null_ctr_tables (tests/test_gpu_parity.py: the emit must refuse, not launch)
    d.custom.kmer_recs_ctr = nullptr; d.custom.kmer_cols_ctr = nullptr; d.custom.kmer_tab32 = nullptr;
  }
  e->custom_prof = d;
  e->custom_hash = mh;
  e->custom_bytes = p->custom_model_bytes;
  e->custom_long = want_long;
  e->custom_model_copy.assign((const uint8_t*)p->custom_model, (const uint8_t*)p->custom_model + p->custom_model_bytes);
  e->custom_cached = true;
  *out = d;
  return SIMMR_OK;
}

// ---- profile validation (cli.rs:229-301 semantics) -------------------------
int make_profile(simmr_engine* e, const simmr_error_profile* p, bool want_long, ProfileDev* out) {
  if (!p) return e->fail(SIMMR_EINVAL, "profile is NULL");
  if (p->kind > SIMMR_CUSTOM) return e->fail(SIMMR_EINVAL, "unknown profile kind %u", p->kind);
  if (p->rng_mode > SIMMR_RNG_PHILOX_FULL) return e->fail(SIMMR_EINVAL, "unknown rng_mode %u", p->rng_mode);
  // the full counter mode (the plan's draws from Philox counters too): the parametric profiles that draw per base
  if (p->rng_mode == SIMMR_RNG_PHILOX_FULL && (p->kind == SIMMR_CUSTOM || p->kind == SIMMR_PERFECT_SHORT))
    return e->fail(SIMMR_EINVAL, "SIMMR_RNG_PHILOX_FULL covers minimal-short, minimal-long and perfect-long (a custom model's lengths "
                                 "and qualities, and perfect-short as a whole, are the reference's streams)");
  // a custom model has one place where draws are made base by base: the k-mer splice of its long-read path
  // (simulate_errors, custom_short.rs:455-516; the qualities use the same few words at every position)
  if (p->rng_mode != SIMMR_RNG_REFERENCE && p->kind == SIMMR_CUSTOM && !want_long)
    return e->fail(SIMMR_EINVAL, "SIMMR_RNG_PHILOX covers the profiles that draw base by base: minimal-short, minimal-long, "
                                 "perfect-long and the k-mer splice of a custom long-read model");
  if (p->kind == SIMMR_CUSTOM) {
    const int rc = make_custom_profile(e, p, want_long, out);
    if (rc == SIMMR_OK) out->rng_mode = p->rng_mode;  // (the cached tables serve both modes)
    return rc;
  }
  const bool is_long = p->kind == SIMMR_PERFECT_LONG || p->kind == SIMMR_MINIMAL_LONG;
  if (is_long != want_long)
    return e->fail(SIMMR_EINVAL, want_long ? "a short-read profile was passed to the long-read path"
                                           : "a long-read profile was passed to the paired-end path");
  ProfileDev d{};
  d.kind = p->kind;
  d.rng_mode = p->rng_mode;
  d.read_length = p->read_length;
  d.insert_size = p->insert_size;
  d.mean_phred_f = (float)p->mean_phred;
  d.pl_mean = 1.0f - powf(10.0f, -((float)20 / 10.0f));  // convert_phred_to_accuracy(20)
  if (p->long_start_mode > SIMMR_START_UNIFORM) return e->fail(SIMMR_EINVAL, "unknown long_start_mode %u", p->long_start_mode);
  d.long_start_uniform = p->long_start_mode == SIMMR_START_UNIFORM ? 1u : 0u;
  d.gamma_shape = p->gamma_shape;
  d.gamma_scale = p->gamma_scale;
  d.read_length_std = p->read_length_std;
  d.insert_size_std = p->insert_size_std;
  if (is_long) {
    d.required = 20000;  // minimal_long.rs:152-154
    if (!(p->gamma_shape > 1.0f) || !(p->gamma_scale > 0.0f))
      return e->fail(SIMMR_EINVAL, "gamma shape must be > 1 and scale > 0 (got %g, %g)",
                     (double)p->gamma_shape, (double)p->gamma_scale);
  } else {
    // perfect_short.rs:56-59: 2 * read_length + insert_size in u16.  The
    // reference overflows silently; reject instead.
    uint32_t req = 2u * p->read_length + p->insert_size;
    if (req > 65535u)
      return e->fail(SIMMR_ERANGE, "2*read_length + insert_size = %u overflows the reference's u16", req);
    if (p->read_length == 0) return e->fail(SIMMR_EINVAL, "read_length is 0");
    d.required = req;
    if (p->kind == SIMMR_MINIMAL_SHORT &&
        (!(p->read_length_std >= 0.0) || !(p->insert_size_std >= 0.0)))
      return e->fail(SIMMR_EINVAL, "negative standard deviation");
  }
  if (p->rng_mode != SIMMR_RNG_REFERENCE && p->kind != SIMMR_PERFECT_SHORT) {
    // The two tables of the counter mode.  The joint law over the 1024
    // outcomes o = q | s << 8, w(q,0) = P(q)(1 - p_q), w(q,s) = P(q) p_q / 3 (P(q) = the profile's Phred law,
    // p_q = the probability of the reference's 24-bit test gen::<f32>() > accuracy(q), minimal_short.rs:83-140),
    // is split exactly into c(o) = floor(2^24 w(o)) cells of a 24-bit draw plus E escape cells that lead to a
    // full-word draw from the residual law (2^24 w(o) - c(o)) / E.
    constexpr int N = 1024, CELLS = 16777216, UNIT = 16384;
    std::vector<double> w(N), odds(N);
    double prev = 0.0;
    for (int q = 0; q < 256; q++) {
      // P(Phred <= q): minimal profiles floor(N(mean, 10)) saturated to u8; perfect-long (perfect_long.rs:60-78)
      // round(-10 log10(1 - min(N(0.99, 0.05), 0.9999))), whose cap puts everything above 0.9999 on q = 40
      double upper;
      if (q == 255) upper = 1.0;
      else if (p->kind == SIMMR_PERFECT_LONG)
        upper = q >= 40 ? 1.0 : 0.5 * erfc(-(((1.0 - pow(10.0, -((double)q + 0.5) / 10.0)) - 0.99) / 0.05) / 1.4142135623730951);
      else
        upper = 0.5 * erfc(-(((double)(q + 1) - (double)p->mean_phred) / 10.0) / 1.4142135623730951);
      double P = upper - prev;
      if (P < 0.0) P = 0.0;
      prev = upper;
      const float acc = 1.0f - powf(10.0f, -((float)q / 10.0f));  // util.rs:69-71,96-98
      const float tf = floorf(acc * 16777216.0f);
      const double t = tf > 16777215.0f ? 16777215.0 : (double)tf;
      const double pq = (16777215.0 - t) / 16777216.0;
      w[q] = P * (1.0 - pq);
      for (int sft = 1; sft < 4; sft++) w[q + 256 * sft] = P * pq / 3.0;
    }
    std::vector<int64_t> cells(N), wt(N);
    int64_t sum = 0;
    int nz = 0;
    for (int o = 0; o < N; o++) {
      cells[o] = (int64_t)floor(w[o] * (double)CELLS);
      sum += cells[o];
      if (cells[o] > 0) nz++;
    }
    int64_t E = (int64_t)CELLS - sum;
    while (E < 0 || nz + (E > 0 ? 1 : 0) > N) {  // one column per outcome with cells and one for the escape
      int m = -1;
      for (int o = 0; o < N; o++) if (cells[o] > 0 && (m < 0 || cells[o] < cells[m])) m = o;
      E += cells[m]; cells[m] = 0; nz--;
    }
    // level 1: integer Vose over the entries (outcomes with cells in increasing order, then the escape), 16384 cells
    // per column, LIFO worklists filled in increasing column order
    std::vector<uint32_t> prim(N);
    std::vector<int> alias(N), smalls(N), bigs(N), T(N);
    int n = 0, ns = 0, nb = 0;
    for (int o = 0; o < N; o++) if (cells[o] > 0) { prim[n] = (uint32_t)o; wt[n] = cells[o]; n++; }
    if (E > 0) { prim[n] = PHILOX_ESC; wt[n] = E; n++; }
    for (int k = n; k < N; k++) { prim[k] = prim[0]; wt[k] = 0; }
    for (int k = 0; k < N; k++) { alias[k] = k; T[k] = UNIT; }
    for (int k = 0; k < N; k++) { if (wt[k] < UNIT) smalls[ns++] = k; else bigs[nb++] = k; }
    while (ns > 0 && nb > 0) {
      const int sm = smalls[--ns], bg = bigs[--nb];
      alias[sm] = bg;
      T[sm] = (int)wt[sm];
      wt[bg] -= UNIT - wt[sm];
      if (wt[bg] < UNIT) smalls[ns++] = bg; else bigs[nb++] = bg;
    }
    // [0, N): level 1, T | A << 16; [N, 2N): level 1, B; [2N, 3N): level 2 (thr22 | alias << 22)
    std::vector<uint32_t> table(3 * N, 0u);
    d.philox_qmax = 0;
    auto see = [&](uint32_t o) { if (o != PHILOX_ESC && (o & 255u) > d.philox_qmax) d.philox_qmax = o & 255u; };
    for (int k = 0; k < N; k++) {
      table[k] = (uint32_t)T[k] | (prim[k] << 16);
      table[N + k] = prim[alias[k]];
      if (T[k] > 0) see(prim[k]);
      if (T[k] < UNIT) see(prim[alias[k]]);
    }
    d.philox_qmax1 = d.philox_qmax;  // (so far only level-1 answers have been seen)
    // level 2: Vose's method on the residual law, LIFO worklists filled in increasing index order
    if (E > 0) {
      for (int o = 0; o < N; o++) odds[o] = (w[o] * (double)CELLS - (double)cells[o]) / (double)E * (double)N;
      ns = nb = 0;
      for (int i = 0; i < N; i++) alias[i] = i;
      for (int i = 0; i < N; i++) { if (odds[i] < 1.0) smalls[ns++] = i; else bigs[nb++] = i; }
      while (ns > 0 && nb > 0) {
        const int sm = smalls[--ns], bg = bigs[--nb];
        alias[sm] = bg;
        odds[bg] = odds[bg] - 1.0 + odds[sm];
        if (odds[bg] < 1.0) smalls[ns++] = bg; else bigs[nb++] = bg;
      }
      while (ns > 0) odds[smalls[--ns]] = 1.0;
      while (nb > 0) odds[bigs[--nb]] = 1.0;
      for (int i = 0; i < N; i++) {
        const double t = floor(odds[i] * 4194304.0);
        const uint32_t thr = t >= 4194303.0 ? 4194303u : (t <= 0.0 ? 0u : (uint32_t)t);
        table[2 * N + i] = thr | ((uint32_t)alias[i] << 22);
        if (thr > 0) see((uint32_t)i);
        if (thr < 4194303u) see((uint32_t)alias[i]);
      }
    }
    int rc = upload_vec(e, e->ph_table, table);
    if (rc || (rc = sync_check(e, "philox table upload"))) return rc;
    d.philox_t1 = e->ph_table.as<uint32_t>();
    d.philox_t2 = e->ph_table.as<uint32_t>() + 2 * N;
  }
  *out = d;
  return SIMMR_OK;
}

uint64_t os_entropy_u64() {
  std::random_device rd;
  return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
}

// ---- outer stream ------------------------------------------------------------
// Runs the transducer over one StdRng(seed) run that starts at `start_slot`
// with gen_range(0..range): classifies enough blocks to hold `n_total` units,
// writes (idx, seed) of units [emit_first, emit_first+emit_count) to
// out_idx/out_seed, and returns the slot following unit n_total-1.
int run_outer(simmr_engine* e, uint64_t seed, uint64_t range, uint64_t start_slot, uint64_t n_total,
              uint64_t emit_first, uint64_t emit_count, uint32_t* out_idx, uint64_t* out_seed,
              uint64_t* end_slot) {
  if (n_total == 0) { *end_slot = start_slot; return SIMMR_OK; }
  OuterParams P;
  P.key = host_pcg32_expand(seed);
  P.range = range;
  P.zone = (range << __builtin_clzll(range)) - 1;
  const double p_acc = ((double)P.zone + 1.0) / 18446744073709551616.0;
  const double per_unit = 1.0 / p_acc + 1.0;
  const double var_unit = (1.0 - p_acc) / (p_acc * p_acc);
  const uint64_t first_block = start_slot >> 3;
  const uint32_t first_skip = (uint32_t)(start_slot & 7);
  double want = (double)n_total * per_unit + 6.0 * sqrt((double)n_total * var_unit) + 64.0;
  uint64_t n_blocks = ((uint64_t)want + first_skip + 7) / 8 + 1;
  OuterScanResult res{};
  for (int attempt = 0;; attempt++) {
    if (attempt > 8) return e->fail(SIMMR_ESTATE, "outer stream did not yield %llu units", (unsigned long long)n_total);
    const uint64_t n_wg = (n_blocks + 255) / 256;
    if (n_wg > 0x7fffffffULL) return e->fail(SIMMR_ERANGE, "outer stream too long for one launch");
    if (!e->o_last_idx.ensure(n_blocks * 4) || !e->o_wg_sums.ensure(n_wg * 4) ||
        !e->o_wg_prefix.ensure(n_wg * sizeof(OuterPrefix)) || !e->o_result.ensure(sizeof(OuterScanResult)))
      return e->fail(SIMMR_ENOMEM, "outer stream scratch allocation failed");
    HIP_TRY(e, hipMemsetAsync(e->o_result.p, 0, sizeof(OuterScanResult), e->stream));
    hipLaunchKernelGGL(k_outer_classify, dim3((uint32_t)n_wg), dim3(256), 0, e->stream, P, first_block,
                       n_blocks, first_skip, e->o_last_idx.as<uint32_t>(), e->o_wg_sums.as<uint32_t>());
    // wg range of the units to emit; the closing unit is looked up separately below
    hipLaunchKernelGGL(k_outer_scan, dim3(1), dim3(256), 0, e->stream, e->o_wg_sums.as<uint32_t>(), n_wg,
                       emit_first, emit_count, e->o_wg_prefix.as<OuterPrefix>(),
                       e->o_result.as<OuterScanResult>());
    HIP_TRY(e, hipMemcpyAsync(&res, e->o_result.p, sizeof res, hipMemcpyDeviceToHost, e->stream));
    int rc = sync_check(e, "outer stream classify/scan");
    if (rc) return rc;
    if (res.total_units >= n_total) break;
    n_blocks = n_blocks + n_blocks / 4 + 1024;
  }
  uint64_t* d_end = &e->o_result.as<OuterScanResult>()->end_slot;
  if (emit_count > 0) {
    const uint64_t n_launch = res.wg_hi - res.wg_lo + 1;
    hipLaunchKernelGGL(k_outer_emit, dim3((uint32_t)n_launch), dim3(256), 0, e->stream, P, first_block,
                       n_blocks, first_skip, res.wg_lo, e->o_last_idx.as<uint32_t>(),
                       e->o_wg_prefix.as<OuterPrefix>(), emit_first, emit_count, out_idx, out_seed, d_end);
  }
  if (emit_count == 0 || emit_first + emit_count != n_total) {
    // locate the workgroup of the closing unit with a second scan query, then
    // let that workgroup record the end slot (it writes no units: count = 0 range trick
    // is avoided by pointing the emit range at the closing unit with null outputs)
    HIP_TRY(e, hipMemsetAsync(e->o_result.p, 0, sizeof(OuterScanResult), e->stream));
    const uint64_t n_wg = (n_blocks + 255) / 256;
    hipLaunchKernelGGL(k_outer_scan, dim3(1), dim3(256), 0, e->stream, e->o_wg_sums.as<uint32_t>(), n_wg,
                       n_total - 1, (uint64_t)1, e->o_wg_prefix.as<OuterPrefix>(),
                       e->o_result.as<OuterScanResult>());
    OuterScanResult r2{};
    HIP_TRY(e, hipMemcpyAsync(&r2, e->o_result.p, sizeof r2, hipMemcpyDeviceToHost, e->stream));
    int rc = sync_check(e, "outer stream end lookup");
    if (rc) return rc;
    if (!e->scan_tmp.ensure(64)) return e->fail(SIMMR_ENOMEM, "scratch allocation failed");
    hipLaunchKernelGGL(k_outer_emit, dim3(1), dim3(256), 0, e->stream, P, first_block, n_blocks, first_skip,
                       r2.wg_lo, e->o_last_idx.as<uint32_t>(), e->o_wg_prefix.as<OuterPrefix>(),
                       n_total - 1, (uint64_t)1, e->scan_tmp.as<uint32_t>(),
                       e->scan_tmp.as<uint64_t>() + 2, d_end);
  }
  uint64_t es = 0;
  HIP_TRY(e, hipMemcpyAsync(&es, d_end, 8, hipMemcpyDeviceToHost, e->stream));
  int rc = sync_check(e, "outer stream emit");
  if (rc) return rc;
  *end_slot = es;
  return SIMMR_OK;
}

// processing order by unit length (granularity 2^shift), so the lanes of a wave finish together
int sort_by_length(simmr_engine* e, uint64_t count, uint32_t shift) {
  e->plan_sorted = false;
  if (count == 0) return SIMMR_OK;
  if (count > 0xffffffffULL) return e->fail(SIMMR_ERANGE, "more than 2^32 units in one shard");
  if (!e->u_order.ensure(count * 4) || !e->len_hist.ensure(LBINS * 4))
    return e->fail(SIMMR_ENOMEM, "order allocation failed");
  HIP_TRY(e, hipMemsetAsync(e->len_hist.p, 0, LBINS * 4, e->stream));
  const uint32_t g1 = (uint32_t)std::min<uint64_t>(grid_for(count, 256), (uint64_t)e->n_cu * 8);
  hipLaunchKernelGGL(k_len_hist, dim3(g1), dim3(256), 0, e->stream, e->u_len.as<uint32_t>(), count, shift,
                     e->len_hist.as<uint32_t>());
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, e->stream, e->len_hist.as<uint32_t>());
  const uint32_t g2 = (uint32_t)std::min<uint64_t>(grid_for(count, 4096), (uint64_t)e->n_cu * 8);
  hipLaunchKernelGGL(k_len_scatter, dim3(g2), dim3(256), 0, e->stream, e->u_len.as<uint32_t>(), count, shift,
                     e->len_hist.as<uint32_t>(), e->u_order.as<uint32_t>());
  e->plan_sorted = true;
  return SIMMR_OK;
}

// exclusive scan of scale * in[i] (n entries of type T) -> `out` (n + 1 u64 entries); returns the total
template <typename T>
int scan_scaled(simmr_engine* e, DevBuf& in, uint64_t n, uint32_t scale, DevBuf& out, uint64_t* total, uint32_t round = 0) {
  if (!out.ensure((n + 1) * 8)) return e->fail(SIMMR_ENOMEM, "offset allocation failed");
  if (n == 0) {
    HIP_TRY(e, hipMemsetAsync(out.p, 0, 8, e->stream));
    *total = 0;
    return SIMMR_OK;
  }
  const uint32_t per_wg = SCAN_THREADS * SCAN_ITEMS;
  const uint64_t n_wg = (n + per_wg - 1) / per_wg;
  if (!e->scan_tmp.ensure((n_wg + 2) * 8)) return e->fail(SIMMR_ENOMEM, "scan scratch allocation failed");
  uint64_t* wg_tot = e->scan_tmp.as<uint64_t>();
  uint64_t* grand = wg_tot + n_wg;
  hipLaunchKernelGGL(k_scan_reduce<T>, dim3((uint32_t)n_wg), dim3(SCAN_THREADS), 0, e->stream,
                     (const T*)in.as<T>(), n, scale, round, wg_tot);
  hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(SCAN_THREADS), 0, e->stream, wg_tot, n_wg, grand);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3((uint32_t)n_wg), dim3(SCAN_THREADS), 0, e->stream,
                     (const T*)in.as<T>(), n, scale, round, (const uint64_t*)wg_tot, out.as<uint64_t>());
  HIP_TRY(e, hipMemcpyAsync(total, grand, 8, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e, "offset scan");
}
int scan_u64(simmr_engine* e, DevBuf& in, uint64_t n, DevBuf& out, uint64_t* total) { return scan_scaled<uint64_t>(e, in, n, 1u, out, total); }

// The same scan when the kernel that PRODUCED in[] has already added every entry (times scale) to the sum of its tile:
// tile_sums_begin() zeroes the tile sums and hands them to the producer, scan_presummed() runs the two remaining passes.
unsigned long long* tile_sums_begin(simmr_engine* e, uint64_t n) {
  const uint32_t per_wg = SCAN_THREADS * SCAN_ITEMS;
  const uint64_t n_wg = (std::max<uint64_t>(n, 1) + per_wg - 1) / per_wg;
  if (!e->scan_tmp.ensure((n_wg + 2) * 8)) return nullptr;
  if (hipMemsetAsync(e->scan_tmp.p, 0, (n_wg + 2) * 8, e->stream) != hipSuccess) return nullptr;
  return e->scan_tmp.as<unsigned long long>();
}
template <typename T>
int scan_presummed(simmr_engine* e, DevBuf& in, uint64_t n, uint32_t scale, DevBuf& out, uint64_t* total, uint32_t round = 0) {
  if (!out.ensure((n + 1) * 8)) return e->fail(SIMMR_ENOMEM, "offset allocation failed");
  if (n == 0) {
    HIP_TRY(e, hipMemsetAsync(out.p, 0, 8, e->stream));
    *total = 0;
    return SIMMR_OK;
  }
  const uint32_t per_wg = SCAN_THREADS * SCAN_ITEMS;
  const uint64_t n_wg = (n + per_wg - 1) / per_wg;
  uint64_t* wg_tot = e->scan_tmp.as<uint64_t>();
  uint64_t* grand = wg_tot + n_wg;
  hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(SCAN_THREADS), 0, e->stream, wg_tot, n_wg, grand);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3((uint32_t)n_wg), dim3(SCAN_THREADS), 0, e->stream,
                     (const T*)in.as<T>(), n, scale, round, (const uint64_t*)wg_tot, out.as<uint64_t>());
  HIP_TRY(e, hipMemcpyAsync(total, grand, 8, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e, "offset scan");
}
// exclusive scan of the bytes each unit writes (reads_per_unit * u_len) -> u_off
int scan_offsets(simmr_engine* e, uint64_t n, uint32_t reads_per_unit, uint64_t* total, uint32_t round = 0) {
  return scan_scaled<uint32_t>(e, e->u_len, n, reads_per_unit, e->u_off, total, round);
}

int ensure_plan_arrays(simmr_engine* e, uint64_t n, bool need_seeds2, bool need_genome) {
  const uint64_t m = std::max<uint64_t>(n, 1);
  bool ok = e->u_contig.ensure(m * 4) && e->u_seed.ensure(m * 8) && e->u_len.ensure(m * 4) &&
            e->u_a.ensure(m * 8) && e->u_b.ensure(m * 8) &&
            e->u_flags.ensure(m);
  if (need_seeds2) ok = ok && e->u_qs2.ensure(m * 8) && e->u_ms2.ensure(m * 8);
  if (need_genome) ok = ok && e->u_genome.ensure(m * 4);
  if (!ok) return e->fail(SIMMR_ENOMEM, "plan allocation failed for %llu units", (unsigned long long)n);
  return SIMMR_OK;
}

PlanArrays plan_arrays(simmr_engine* e, bool seeds2) {
  PlanArrays pl;
  pl.len = e->u_len.as<uint32_t>();
  pl.a = e->u_a.as<uint64_t>();
  pl.b = e->u_b.as<uint64_t>();
  pl.qs2 = seeds2 ? e->u_qs2.as<uint64_t>() : nullptr;
  pl.ms2 = seeds2 ? e->u_ms2.as<uint64_t>() : nullptr;
  pl.flags = e->u_flags.as<uint8_t>();
  return pl;
}

int read_err_word(simmr_engine* e, uint32_t* w) {
  HIP_TRY(e, hipMemcpyAsync(w, e->d_err.p, 4, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e, "error word readback");
}

// The layout a plan for `prof` gets (simmr_engine_set_read_slots): *round = 15 for 16-byte read slots, else 0.  Slots
// are a preference: the counter-mode item kernel (k_emit_philox) writes them; a plan whose emit kernel does not gets the
// compact layout and says so in simmr_plan_info.slot_bytes, which is what the caller sizes and labels its buffers from.
int plan_slot_round(simmr_engine* e, const ProfileDev& prof, uint32_t* round) {
  const bool offered = !(prof.kind == SIMMR_K_CUSTOM || prof.kind == SIMMR_K_PERFECT_SHORT || prof.rng_mode == SIMMR_RNG_REFERENCE);
  *round = (e->read_slots == SIMMR_SLOT16 && offered) ? 15u : 0u;
  return SIMMR_OK;
}

// Pair plans whose emit kernel is the counter-mode item kernel get no per-pair offsets: the plan kernel leaves the bytes
// of every 64 pairs, their scan (u_off64) places the emit kernel's blocks, and a block places its own reads
// (k_emit_philox: `coarse`).
// k_plan_pe for `count` pairs.  (Round 3 built a register form of it — the pair's ChaCha12 block in registers, every
// usual draw a fixed word, the 1-2 % of the pairs whose draws take another way left to a second kernel: bit-exact in
// the whole suite and no faster, 1.22 + 1.31 ms against 1.39: the kernel's thousand instructions per pair are PCG32 and
// ChaCha12 either way, and the left-over pairs cost 5000 apiece once no neighbour shares their path.
// profiles/r3/plan_register_form_kernels.txt)
// `oc`: SIMMR_RNG_PHILOX_FULL — the pairs' outer draws are made by the plan kernel itself (kernels.hip: outer_ctr_pair)
int launch_plan_pe(simmr_engine* e, const ProfileDev& prof, uint32_t genome, uint64_t count, const uint32_t* u_genome,
                   const PlanArrays& pw, unsigned long long* tiles, uint32_t slot_round, unsigned long long* wave_bytes,
                   const OuterCtrArgs& oc = OuterCtrArgs{}) {
  auto plan_kern = prof.rng_mode == SIMMR_RNG_PHILOX_FULL ? k_plan_pe<true> : k_plan_pe<false>;
  hipLaunchKernelGGL(plan_kern, dim3(grid_for(count, PLAN_THREADS)), dim3(PLAN_THREADS), 0, e->stream, prof,
                     e->d_genomes.as<GenomeDev>(), genome, count, e->u_contig.as<uint32_t>(), e->u_seed.as<uint64_t>(), u_genome, pw,
                     e->d_tables.as<Tables>(), e->d_err.as<uint32_t>(), tiles, slot_round, wave_bytes, oc);
  return SIMMR_OK;
}

bool plan_is_coarse(simmr_engine* e, const ProfileDev& prof) {
  return prof.rng_mode != SIMMR_RNG_REFERENCE && prof.kind != SIMMR_K_CUSTOM && prof.kind != SIMMR_K_PERFECT_SHORT;
}

// the event pair of the emit about to be launched (a ring of EMIT_RING pairs, made on first use)
hipError_t next_emit_events(simmr_engine* e) {
  const int i = (int)(e->n_emits % simmr_engine::EMIT_RING);
  if (!e->ring_c[i]) {
    if (e->n_emits == 0) { e->ring_c[0] = e->ev_c; e->ring_d[0] = e->ev_d; }
    else {
      // both events exist before either is published: a slot never holds a start event without its end event
      hipEvent_t c = nullptr, d = nullptr;
      hipError_t s = hipEventCreate(&c);
      if (s != hipSuccess) return s;
      if ((s = hipEventCreate(&d)) != hipSuccess) { (void)hipEventDestroy(c); return s; }
      e->ring_c[i] = c; e->ring_d[i] = d;
    }
  }
  e->ev_c = e->ring_c[i]; e->ev_d = e->ring_d[i];
  e->n_emits++;
  return hipSuccess;
}

int check_out(simmr_engine* e, const simmr_reads_out* out, uint64_t n_reads, uint64_t total) {
  if (!out) return e->fail(SIMMR_EINVAL, "out is NULL");
  if (!out->seq_off) return e->fail(SIMMR_EINVAL, "out->seq_off is NULL");
  if (total > 0 && (!out->seq || !out->qual)) return e->fail(SIMMR_EINVAL, "out->seq / out->qual is NULL");
  if (out->seq_capacity < total)
    return e->fail(SIMMR_ERANGE, "seq_capacity %llu < %llu bytes planned",
                   (unsigned long long)out->seq_capacity, (unsigned long long)total);
  if (out->reads_capacity < n_reads)
    return e->fail(SIMMR_ERANGE, "reads_capacity %llu < %llu reads planned",
                   (unsigned long long)out->reads_capacity, (unsigned long long)n_reads);
  if (out->qual_offset > 255u) return e->fail(SIMMR_EINVAL, "qual_offset too large");
  if ((out->slot_bytes <= 1u ? 0u : out->slot_bytes) != e->plan_slot)
    return e->fail(SIMMR_EINVAL, "out->slot_bytes = %u, but the plan was made for %s (simmr_engine_set_read_slots, simmr_plan_info.slot_bytes)",
                   out->slot_bytes, e->plan_slot ? "16-byte read slots" : "the compact layout");
  return SIMMR_OK;
}

OutCols out_cols(const simmr_reads_out* out) {
  OutCols o;
  o.seq_off = out->seq_off; o.start = out->start; o.end = out->end; o.contig = out->contig;
  o.genome = out->genome; o.read_id = out->read_id; o.flags = out->flags;
  return o;
}

// the instantiation of the counter-mode item kernel for (exception plane, contig bases in LDS, escapes noticed through
// the quality byte, 16-byte read slots, block offsets from the coarse scan)
// Paired plans are coarse (the block places its reads) and may keep their contig bases in LDS; long-read plans have
// per-read offsets and never do.  The flag-bit form (ESCQ = false: some offset-encoded level-1 answer is above 127, i.e. a
// mean Phred far above what the profiles of the reference use) exists in its most general shape only — with the
// exception plane, without the contig cache — which serves every plan.  16 instantiations (tests/test_resource_guard.py).
using PhiloxKernel = decltype(&k_emit_philox<false, false, false, false, false, false, false>);
template <bool EXC, bool CACHED, bool ESCQ, bool COARSE>
static PhiloxKernel philox_kernel2(bool slot) {
  return slot ? k_emit_philox<EXC, false, CACHED, false, ESCQ, true, COARSE> : k_emit_philox<EXC, false, CACHED, false, ESCQ, false, COARSE>;
}
static PhiloxKernel philox_kernel(bool exc, bool cached, bool escq, bool slot, bool coarse) {
  if (!escq) return coarse ? philox_kernel2<true, false, false, true>(slot) : philox_kernel2<true, false, false, false>(slot);
  if (!coarse) return exc ? philox_kernel2<true, false, true, false>(slot) : philox_kernel2<false, false, true, false>(slot);
  if (exc) return cached ? philox_kernel2<true, true, true, true>(slot) : philox_kernel2<true, false, true, true>(slot);
  return cached ? philox_kernel2<false, true, true, true>(slot) : philox_kernel2<false, false, true, true>(slot);
}
// the same for the TEXT form (simmr_emit_fastq): always coarse; `copy_only` = perfect-short (no draws)
static PhiloxKernel philox_text_kernel(bool exc, bool cached, bool escq, bool copy_only) {
  if (copy_only) return cached ? (exc ? k_emit_philox<true, true, true, true, false, false, true> : k_emit_philox<false, true, true, true, false, false, true>)
                               : (exc ? k_emit_philox<true, true, false, true, false, false, true> : k_emit_philox<false, true, false, true, false, false, true>);
  if (!escq) return k_emit_philox<true, false, false, true, false, false, true>;
  return cached ? (exc ? k_emit_philox<true, false, true, true, true, false, true> : k_emit_philox<false, false, true, true, true, false, true>)
                : (exc ? k_emit_philox<true, false, false, true, true, false, true> : k_emit_philox<false, false, false, true, true, false, true>);
}

// Every device table the custom long-read kernels dereference for this mode and form must exist BEFORE they are launched:
// a kernel handed a null table base faults the GPU (round 4, gpurun_out/ctr1_*: "Memory access fault ... on address (nil)"
// from the first build of the splice's counter mode; LAB.md, round 5, has what is known about it), and a fault on this
// pool can reset every GPU of the host.  Returns the name of the first table that is missing, or null.
static const char* custom_long_tables_missing(const ProfileDev& prof, bool fast, bool ctr) {
  const CustomDev& c = prof.custom;
  if (!c.pdfs) return "pdfs";
  if (!c.col_rec) return "col_rec";
  if (!c.bin_rec) return "bin_rec";
  if (!c.kmer_direct) return "kmer_direct";
  if (!c.kmer_slots) return "kmer_slots";
  if (fast) {
    if (ctr) { if (!c.kmer_cols_ctr) return "kmer_cols_ctr"; if (!c.kmer_tab32) return "kmer_tab32"; }
    else { if (!c.kmer_cols) return "kmer_cols"; if (!c.kmer_cnt8) return "kmer_cnt8"; }
  }
  if (ctr ? !c.kmer_recs_ctr : !c.kmer_recs) return ctr ? "kmer_recs_ctr" : "kmer_recs";
  return nullptr;
}

// the whole-line form (text_lines.hip)
using TextLinesKernel = decltype(&k_emit_text_lines<false, false, false, false>);
static TextLinesKernel text_lines_kernel(bool exc, bool cached, bool escq, bool copy_only) {
  if (copy_only) return cached ? (exc ? k_emit_text_lines<true, true, true, false> : k_emit_text_lines<false, true, true, false>)
                               : (exc ? k_emit_text_lines<true, true, false, false> : k_emit_text_lines<false, true, false, false>);
  if (!escq) return k_emit_text_lines<true, false, false, false>;
  return cached ? (exc ? k_emit_text_lines<true, false, true, true> : k_emit_text_lines<false, false, true, true>)
                : (exc ? k_emit_text_lines<true, false, false, true> : k_emit_text_lines<false, false, false, true>);
}
// bytes of a header slot of that kernel: the header at any offset below 8, its '\n', what FqW may write past its bytes; a multiple of 8 and an odd number of 8-byte words (the lanes of a wave then start on different banks)
static uint32_t tl_slot_pitch(uint32_t max_header) {
  uint32_t w = (7u + max_header + 1u + 12u + 7u) / 8u;  // (a piece of fq_put8 writes three whole words: up to ten bytes past the header's end)
  if (!(w & 1u)) w++;
  return 8u * w;
}

}  // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int simmr_abi_version(void) { return SIMMR_ABI_VERSION; }

uint64_t simmr_entropy_substitute(uint64_t x, uint32_t which) { return entropy_substitute(x, which); }

const char* simmr_last_error(const simmr_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int simmr_engine_create(int device_ordinal, simmr_engine** out) {
  if (!out) { g_create_error = "out is NULL"; return SIMMR_EINVAL; }
  *out = nullptr;
  int n = 0;
  hipError_t s = hipGetDeviceCount(&n);
  if (s != hipSuccess || n <= 0) {
    g_create_error = std::string("no HIP device available: ") + (s == hipSuccess ? "device count is 0" : hipGetErrorString(s));
    return SIMMR_ENODEV;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    g_create_error = "device ordinal out of range (there is no CPU backend)";
    return SIMMR_ENODEV;
  }
  hipDeviceProp_t prop;
  if ((s = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess) {
    g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(s);
    return SIMMR_ENODEV;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only";
    return SIMMR_ENODEV;
  }
  if ((s = hipSetDevice(device_ordinal)) != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(s);
    return SIMMR_ENODEV;
  }
  simmr_engine* e = new simmr_engine();
  e->device = device_ordinal;
  e->n_cu = prop.multiProcessorCount;
  // Four measurement knobs (grid multiples, workgroups per CU, the splice variant) + SIMMR_TEXT_FORM (which kernel writes the
  // FASTQ text: same-box A/B) + SIMMR_FAULT_INJECT (test switch), read once here; the defaults are what the sweeps of LAB.md found.
  if (const char* v = getenv("SIMMR_GRID_MULT"))
    e->lanes_mult = e->perfect_mult = e->custom_pe_mult = e->custom_long_mult = e->fastq_mult =
        (uint32_t)std::min<unsigned long long>(512, std::max<unsigned long long>(1, strtoull(v, nullptr, 10)));
  if (const char* v = getenv("SIMMR_FASTQ_GRID_MULT")) e->fastq_mult = (uint32_t)std::min<unsigned long long>(512, std::max<unsigned long long>(1, strtoull(v, nullptr, 10)));
  if (const char* v = getenv("SIMMR_TEXT_FORM")) e->text_form = atoi(v);
  e->tl_debug = getenv("SIMMR_TL_DEBUG") != nullptr;
  if (const char* v = getenv("SIMMR_FAULT_INJECT")) e->inject_null_ctr_tables = strcmp(v, "null_ctr_tables") == 0;
  if (const char* v = getenv("SIMMR_PHILOX_WGS_PER_CU")) e->philox_wgs_per_cu = (uint32_t)std::min<unsigned long long>(4096, std::max<unsigned long long>(1, strtoull(v, nullptr, 10)));
  if (const char* v = getenv("SIMMR_SPLICE_VARIANT")) e->splice_variant = atoi(v);
  bool ok = e->d_tables.ensure(sizeof(Tables)) && e->d_counters.ensure(8 * SIMMR_N_COUNTERS * (1 + SIMMR_CNT_SHARDS)) &&
            e->d_err.ensure(64) && e->d_scalars.ensure(256);
  ok = ok && hipEventCreate(&e->ev_a) == hipSuccess && hipEventCreate(&e->ev_b) == hipSuccess &&
       hipEventCreate(&e->ev_c) == hipSuccess && hipEventCreate(&e->ev_d) == hipSuccess;
  if (ok) {
    Tables* T = new Tables();
    build_tables(T);
    ok = hipMemcpy(e->d_tables.p, T, sizeof(Tables), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(e->d_counters.p, 0, 8 * SIMMR_N_COUNTERS * (1 + SIMMR_CNT_SHARDS)) == hipSuccess &&
         hipMemset(e->d_err.p, 0, 64) == hipSuccess;
    delete T;
  }
  if (!ok) {
    g_create_error = "engine allocation failed";
    simmr_engine_destroy(e);
    return SIMMR_ENOMEM;
  }
  *out = e;
  return SIMMR_OK;
}

// ---- the run counters across GPUs: one all-reduce over RCCL ------------------------------
// RCCL is bound at run time (dlopen), so that a process which already carries one (PyTorch bundles its own
// librccl) keeps a single copy, and a single-GPU user needs none.
namespace {
struct RcclId { char internal[128]; };  // ncclUniqueId
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static std::string err;
  static std::once_flag once;
  std::call_once(once, [] {
    // SIMMR_RCCL_LIB names the library to load instead of the default candidates (a test points it at a missing
    // file to check the failure path; a deployment may pin one copy of RCCL with it)
    const char* forced = getenv("SIMMR_RCCL_LIB");
    std::vector<const char*> names;
    if (forced && *forced) names = {forced};
    else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string last = "?";
    for (const char* name : names) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
      const char* m = dlerror();  // glibc clears the message once it has been read: read it once
      if (m) last = m;
    }
    if (!api.lib) {
      err = std::string("cannot load librccl: ") + last;
    } else {
      api.GetUniqueId = (int (*)(RcclId*))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(api.lib, "ncclCommInitRank");
      api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclAllReduce");
      api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
      api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
        api.lib = nullptr;
      }
    }
  });
  if (!api.lib) { if (why) *why = err; return nullptr; }
  return &api;
}
void comm_release(simmr_engine* e) {
  if (!e->comm) return;
  if (RcclApi* a = rccl_api(nullptr)) (void)a->CommDestroy(e->comm);
  e->comm = nullptr;
}
}  // namespace

int simmr_comm_unique_id(uint8_t* id128) {
  if (!id128) return SIMMR_EINVAL;
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (!a) return SIMMR_ENODEV;
  RcclId id;
  if (a->GetUniqueId(&id) != 0) return SIMMR_ENODEV;
  memcpy(id128, id.internal, sizeof id.internal);
  return SIMMR_OK;
}

int simmr_comm_init(simmr_engine* e, const uint8_t* id128, int rank, int world) {
  if (!e) return SIMMR_EINVAL;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return e->fail(SIMMR_EINVAL, "bad communicator arguments (rank %d of %d)", rank, world);
  HIP_TRY(e, hipSetDevice(e->device));
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (!a) return e->fail(SIMMR_ENODEV, "%s", why.c_str());
  comm_release(e);
  RcclId id;
  memcpy(id.internal, id128, sizeof id.internal);
  const int rc = a->CommInitRank(&e->comm, world, id, rank);
  if (rc != 0) {
    e->comm = nullptr;
    return e->fail(SIMMR_ENODEV, "ncclCommInitRank: %s", a->GetErrorString ? a->GetErrorString(rc) : "error");
  }
  return SIMMR_OK;
}

int simmr_allreduce_counts(simmr_engine* e, uint64_t* counts_device, uint32_t n) {
  if (!e) return SIMMR_EINVAL;
  if (!e->comm) return SIMMR_OK;  // one GPU: nothing to add
  if (!counts_device) return e->fail(SIMMR_EINVAL, "counts_device is NULL");
  HIP_TRY(e, hipSetDevice(e->device));
  RcclApi* a = rccl_api(nullptr);
  const int rc = a->AllReduce(counts_device, counts_device, n, /* ncclUint64 */ 5, /* ncclSum */ 0, e->comm, e->stream);
  if (rc != 0) return e->fail(SIMMR_ENODEV, "ncclAllReduce: %s", a->GetErrorString ? a->GetErrorString(rc) : "error");
  return SIMMR_OK;
}

void simmr_engine_destroy(simmr_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  comm_release(e);
  (void)hipDeviceSynchronize();
  if (e->plan_stream) (void)hipStreamDestroy(e->plan_stream);
  if (e->ev_plan_done) (void)hipEventDestroy(e->ev_plan_done);
  for (int i = 0; i < 2; i++) if (e->ev_mark[i]) (void)hipEventDestroy(e->ev_mark[i]);
  for (DevBuf* b : {&e->s_w_bytes, &e->s_u_off64, &e->s_m_genomes, &e->s_u_contig, &e->s_u_genome, &e->s_u_seed, &e->s_u_len, &e->s_u_a,
                    &e->s_u_b, &e->s_u_qs2, &e->s_u_ms2, &e->s_u_flags, &e->s_u_off, &e->s_u_order, &e->s_d_err, &e->s_d_runs,
                    &e->s_d_usable, &e->s_ph_table})
    b->release();
  for (auto& g : e->genomes) { g.packed.release(); g.mask.release(); g.d_contigs.release(); }
  DevBuf* bufs[] = {&e->d_genomes, &e->d_tables, &e->d_counters, &e->d_err, &e->d_scalars, &e->u_contig,
                    &e->u_genome, &e->u_seed, &e->u_len, &e->u_a, &e->u_b, &e->u_qs2,
                    &e->u_ms2, &e->u_flags, &e->u_off, &e->scan_tmp, &e->o_last_idx, &e->o_wg_sums,
                    &e->o_wg_prefix, &e->o_result, &e->d_runs, &e->d_usable, &e->u_order, &e->len_hist, &e->c_pdfs, &e->c_odds, &e->c_alias, &e->c_low, &e->c_range, &e->c_zone, &e->c_colrec, &e->c_binrec, &e->c_kslots, &e->c_krecs, &e->c_kdirect, &e->c_kcnt8, &e->c_kcols, &e->c_krecs_ctr, &e->c_kcols_ctr, &e->c_ktab32, &e->ph_table,
                    &e->fq_blob, &e->fq_gid_off, &e->fq_gid_len, &e->fq_cbase, &e->fq_ncontig, &e->fq_coff, &e->fq_clen,
                    &e->fq_len, &e->fq_off, &e->m_genomes, &e->m_contig, &e->m_seed, &e->w_bytes, &e->u_off64, &e->fq_off64};
  for (DevBuf* b : bufs) b->release();
  if (e->ev_a) (void)hipEventDestroy(e->ev_a);
  if (e->ev_b) (void)hipEventDestroy(e->ev_b);
  if (e->n_emits == 0) { e->ring_c[0] = e->ev_c; e->ring_d[0] = e->ev_d; }
  for (int i = 0; i < simmr_engine::EMIT_RING; i++) {
    if (e->ring_c[i]) (void)hipEventDestroy(e->ring_c[i]);
    if (e->ring_d[i]) (void)hipEventDestroy(e->ring_d[i]);
  }
  delete e;
}

int simmr_engine_set_read_slots(simmr_engine* e, uint32_t slot_bytes) {
  if (!e) return SIMMR_EINVAL;
  if (slot_bytes > 1u && slot_bytes != SIMMR_SLOT16) return e->fail(SIMMR_EINVAL, "read slots are 0 (compact) or 16 bytes");
  e->read_slots = slot_bytes == SIMMR_SLOT16 ? SIMMR_SLOT16 : 0u;
  return SIMMR_OK;
}

int simmr_engine_set_stream(simmr_engine* e, void* hip_stream) {
  if (!e) return SIMMR_EINVAL;
  e->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return SIMMR_OK;
}

int simmr_engine_set_plan_overlap(simmr_engine* e, int on) {
  if (!e) return SIMMR_EINVAL;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipDeviceSynchronize());  // (nothing of either set is in flight across the switch)
  if (on && !e->plan_stream) {
    HIP_TRY(e, hipStreamCreateWithFlags(&e->plan_stream, hipStreamNonBlocking));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_plan_done, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_mark[i], hipEventDisableTiming));
  }
  e->overlap = on != 0;
  e->mark_valid[0] = e->mark_valid[1] = false;
  return SIMMR_OK;
}

// ---- staging ------------------------------------------------------------------
int simmr_stage_genome(simmr_engine* e, uint32_t genome_idx, uint32_t n_contigs,
                       const uint8_t* const* contig_ascii, const uint64_t* contig_len,
                       const uint64_t* contig_size) {
  if (!e) return SIMMR_EINVAL;
  if (!contig_ascii || !contig_len || n_contigs == 0) return e->fail(SIMMR_EINVAL, "empty genome");
  HIP_TRY(e, hipSetDevice(e->device));
  if (genome_idx >= e->genomes.size()) e->genomes.resize(genome_idx + 1);
  GenomeHost& g = e->genomes[genome_idx];
  g.staged = false;
  int rc = layout_genome(e, g, n_contigs, contig_len, contig_size);
  if (rc) return rc;
  const uint64_t CHUNK = 64ull << 20;  // bases per upload, multiple of 64
  DevBuf stage;
  uint64_t maxlen = 0;
  for (uint32_t c = 0; c < n_contigs; c++) maxlen = std::max(maxlen, contig_len[c]);
  if (!stage.ensure(std::min(maxlen, CHUNK) + 64)) return e->fail(SIMMR_ENOMEM, "staging buffer allocation failed");
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  uint32_t* d_any = e->d_err.as<uint32_t>() + 1;
  for (uint32_t c = 0; c < n_contigs; c++) {
    for (uint64_t off = 0; off < contig_len[c]; off += CHUNK) {
      const uint64_t n = std::min(CHUNK, contig_len[c] - off);
      HIP_TRY(e, hipMemcpyAsync(stage.p, contig_ascii[c] + off, n, hipMemcpyHostToDevice, e->stream));
      hipLaunchKernelGGL(k_pack_ascii, dim3(grid_for((n + 31) / 32, 256)), dim3(256), 0, e->stream,
                         stage.as<uint8_t>(), n, g.contigs[c].base + off,
                         g.packed.as<uint32_t>() + FRONT_PAD_WORDS, g.mask.as<uint32_t>() + FRONT_PAD_WORDS,
                         d_any);
      rc = sync_check(e, "k_pack_ascii");  // the staging buffer is reused
      if (rc) { stage.release(); return rc; }
    }
  }
  stage.release();
  uint32_t any = 0;
  HIP_TRY(e, hipMemcpyAsync(&any, d_any, 4, hipMemcpyDeviceToHost, e->stream));
  rc = sync_check(e, "staging");
  if (rc) return rc;
  g.has_exc = any != 0;
  g.staged = true;
  return refresh_genome_table(e);
}

int simmr_stage_fasta(simmr_engine* e, uint32_t genome_idx, uint32_t n_records, const uint8_t* const* body,
                      const uint64_t* body_len, int contiguous, uint64_t min_size, uint64_t* base_count,
                      uint32_t* n_staged) {
  if (!e) return SIMMR_EINVAL;
  if (!body || !body_len || !base_count || n_records == 0) return e->fail(SIMMR_EINVAL, "empty FASTA");
  HIP_TRY(e, hipSetDevice(e->device));
  if (genome_idx >= e->genomes.size()) e->genomes.resize(genome_idx + 1);
  GenomeHost& g = e->genomes[genome_idx];
  g.staged = false;
  if (n_staged) *n_staged = 0;
  // raw bodies on the device, every record at a multiple of FASTA_TILE, the gaps filled with '\n'
  std::vector<uint64_t> tile0(n_records + 1, 0);
  for (uint32_t c = 0; c < n_records; c++) tile0[c + 1] = tile0[c] + (body_len[c] + FASTA_TILE - 1) / FASTA_TILE;
  const uint64_t n_tiles = tile0[n_records];
  DevBuf raw, kept, prefix, d_tile0, d_gather, d_recs, d_sep;
  auto release = [&]() { raw.release(); kept.release(); prefix.release(); d_tile0.release(); d_gather.release(); d_recs.release(); d_sep.release(); };
  int rc = SIMMR_OK;
  if (!raw.ensure(std::max<uint64_t>(n_tiles, 1) * FASTA_TILE) || !kept.ensure(std::max<uint64_t>(n_tiles, 1) * 8) ||
      !d_tile0.ensure((n_records + 1) * 8) || !d_gather.ensure((n_records + 1) * 8) ||
      !d_recs.ensure(n_records * sizeof(FastaRecord)) || !d_sep.ensure(n_records * 8)) {
    release();
    return e->fail(SIMMR_ENOMEM, "FASTA staging allocation failed (%llu bytes)", (unsigned long long)(n_tiles * FASTA_TILE));
  }
#define FASTA_TRY(call) do { hipError_t _s = (call); if (_s != hipSuccess) { release(); return e->fail(SIMMR_ENODEV, "%s failed: %s", #call, hipGetErrorString(_s)); } } while (0)
  FASTA_TRY(hipMemsetAsync(raw.p, '\n', std::max<uint64_t>(n_tiles, 1) * FASTA_TILE, e->stream));
  for (uint32_t c = 0; c < n_records; c++)
    if (body_len[c])
      FASTA_TRY(hipMemcpyAsync(raw.as<uint8_t>() + tile0[c] * FASTA_TILE, body[c], body_len[c], hipMemcpyHostToDevice, e->stream));
  FASTA_TRY(hipMemcpyAsync(d_tile0.p, tile0.data(), (n_records + 1) * 8, hipMemcpyHostToDevice, e->stream));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_tiles, 1), (uint64_t)e->n_cu * 16);
  if (n_tiles) hipLaunchKernelGGL(k_fasta_count, dim3(grid), dim3(256), 0, e->stream, raw.as<uint8_t>(), n_tiles, kept.as<uint64_t>());
  uint64_t total = 0;
  if ((rc = scan_u64(e, kept, n_tiles, prefix, &total))) { release(); return rc; }  // prefix has n_tiles + 1 entries
  std::vector<uint64_t> pre(n_records + 1, 0);
  hipLaunchKernelGGL(k_fasta_gather, dim3(grid_for(n_records + 1, 256)), dim3(256), 0, e->stream, prefix.as<uint64_t>(),
                     d_tile0.as<uint64_t>(), n_records + 1, d_gather.as<uint64_t>());
  FASTA_TRY(hipMemcpyAsync(pre.data(), d_gather.p, (n_records + 1) * 8, hipMemcpyDeviceToHost, e->stream));
  if ((rc = sync_check(e, "FASTA base counts"))) { release(); return rc; }
  for (uint32_t c = 0; c < n_records; c++) base_count[c] = pre[c + 1] - pre[c];
  // layout: the staged sequences and where every record's bases go
  std::vector<FastaRecord> recs(n_records);
  std::vector<uint64_t> lens, sizes, seps;
  if (contiguous) {  // genome.rs:121-137
    uint64_t at = 0;
    for (uint32_t c = 0; c < n_records; c++) { recs[c] = FastaRecord{tile0[c], at}; at += base_count[c]; seps.push_back(at); at += 1; }
    lens.push_back(at);
    sizes.push_back(total);
  } else {
    for (uint32_t c = 0; c < n_records; c++)
      if (base_count[c] > min_size) { lens.push_back(base_count[c]); sizes.push_back(base_count[c]); }
  }
  if (lens.empty()) { release(); return SIMMR_OK; }
  if ((rc = layout_genome(e, g, (uint32_t)lens.size(), lens.data(), sizes.data()))) { release(); return rc; }
  if (!contiguous) {
    uint32_t k = 0;
    for (uint32_t c = 0; c < n_records; c++) recs[c] = FastaRecord{tile0[c], base_count[c] > min_size ? g.contigs[k++].base : ~0ull};
  }
  FASTA_TRY(hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  uint32_t* d_any = e->d_err.as<uint32_t>() + 1;
  FASTA_TRY(hipMemcpyAsync(d_recs.p, recs.data(), n_records * sizeof(FastaRecord), hipMemcpyHostToDevice, e->stream));
  if (n_tiles)
    hipLaunchKernelGGL(k_fasta_pack, dim3(grid), dim3(256), 0, e->stream, raw.as<uint8_t>(), n_tiles, prefix.as<uint64_t>(),
                       d_recs.as<FastaRecord>(), n_records, g.packed.as<uint32_t>() + FRONT_PAD_WORDS,
                       g.mask.as<uint32_t>() + FRONT_PAD_WORDS, d_any);
  if (!seps.empty()) {
    FASTA_TRY(hipMemcpyAsync(d_sep.p, seps.data(), seps.size() * 8, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_fasta_separators, dim3(grid_for(seps.size(), 256)), dim3(256), 0, e->stream, d_sep.as<uint64_t>(),
                       (uint32_t)seps.size(), g.mask.as<uint32_t>() + FRONT_PAD_WORDS, d_any);
  }
  uint32_t any = 0;
  FASTA_TRY(hipMemcpyAsync(&any, d_any, 4, hipMemcpyDeviceToHost, e->stream));
#undef FASTA_TRY
  rc = sync_check(e, "FASTA staging");  // the host vectors and the raw copy may go now
  release();
  if (rc) return rc;
  g.has_exc = any != 0;
  g.staged = true;
  if (n_staged) *n_staged = (uint32_t)lens.size();
  return refresh_genome_table(e);
}

int simmr_stage_synthetic(simmr_engine* e, uint32_t genome_idx, uint32_t n_contigs,
                          const uint64_t* contig_len, uint64_t splitmix_seed) {
  if (!e) return SIMMR_EINVAL;
  if (!contig_len || n_contigs == 0) return e->fail(SIMMR_EINVAL, "empty genome");
  HIP_TRY(e, hipSetDevice(e->device));
  if (genome_idx >= e->genomes.size()) e->genomes.resize(genome_idx + 1);
  GenomeHost& g = e->genomes[genome_idx];
  g.staged = false;
  int rc = layout_genome(e, g, n_contigs, contig_len, nullptr);
  if (rc) return rc;
  const uint64_t n64 = g.plane_bases / 32;
  if (n64)
    hipLaunchKernelGGL(k_synth, dim3(grid_for(n64, 256)), dim3(256), 0, e->stream,
                       reinterpret_cast<uint64_t*>(g.packed.as<uint32_t>() + FRONT_PAD_WORDS), n64,
                       splitmix_seed);
  rc = sync_check(e, "k_synth");
  if (rc) return rc;
  g.has_exc = false;
  g.staged = true;
  return refresh_genome_table(e);
}

int simmr_unstage_contig(simmr_engine* e, uint32_t genome_idx, uint32_t contig, uint64_t first,
                         uint64_t count, uint8_t* dst_host) {
  if (!e) return SIMMR_EINVAL;
  int rc = check_genome(e, genome_idx);
  if (rc) return rc;
  GenomeHost& g = e->genomes[genome_idx];
  if (contig >= g.contigs.size() || first + count > g.contigs[contig].len)
    return e->fail(SIMMR_ERANGE, "unstage range outside the contig");
  if (count == 0) return SIMMR_OK;
  DevBuf tmp;
  if (!tmp.ensure(count)) return e->fail(SIMMR_ENOMEM, "unstage buffer allocation failed");
  hipLaunchKernelGGL(k_unpack, dim3(grid_for(count, 256)), dim3(256), 0, e->stream,
                     g.packed.as<uint32_t>() + FRONT_PAD_WORDS,
                     g.has_exc ? g.mask.as<uint32_t>() + FRONT_PAD_WORDS : (const uint32_t*)nullptr,
                     g.contigs[contig].base + first, count, tmp.as<uint8_t>());
  hipError_t s = hipMemcpyAsync(dst_host, tmp.p, count, hipMemcpyDeviceToHost, e->stream);
  rc = (s == hipSuccess) ? sync_check(e, "k_unpack") : e->fail(SIMMR_ENODEV, "unstage copy: %s", hipGetErrorString(s));
  tmp.release();
  return rc;
}

int simmr_genome_info(const simmr_engine* e, uint32_t genome_idx, uint32_t* n_contigs, uint64_t* total_size) {
  if (!e || genome_idx >= e->genomes.size() || !e->genomes[genome_idx].staged) return SIMMR_EINVAL;
  if (n_contigs) *n_contigs = (uint32_t)e->genomes[genome_idx].contigs.size();
  if (total_size) *total_size = e->genomes[genome_idx].total_size;
  return SIMMR_OK;
}

// ---- paired-end -----------------------------------------------------------------
int simmr_outer_summarize(simmr_engine* e, uint32_t genome_idx, uint64_t seed, uint64_t slot_first,
                          uint64_t slot_count, simmr_outer_summary* out) {
  if (!e) return SIMMR_EINVAL;
  if (!out) return e->fail(SIMMR_EINVAL, "out is NULL");
  if ((slot_first | slot_count) & 7u) return e->fail(SIMMR_EINVAL, "slot ranges are multiples of 8 (one ChaCha block)");
  HIP_TRY(e, hipSetDevice(e->device));
  int rc = check_genome(e, genome_idx);
  if (rc) return rc;
  out->units[0] = out->units[1] = 0;
  out->end_state[0] = 0;
  out->end_state[1] = 1;
  if (slot_count == 0) return SIMMR_OK;
  OuterParams P;
  P.key = host_pcg32_expand(seed);
  P.range = e->genomes[genome_idx].contigs.size();
  P.zone = (P.range << __builtin_clzll(P.range)) - 1;
  const uint64_t n_blocks = slot_count >> 3, n_wg = (n_blocks + 255) / 256;
  if (n_wg > 0x7fffffffULL) return e->fail(SIMMR_ERANGE, "slot range too long for one launch");
  if (!e->o_last_idx.ensure(n_blocks * 4) || !e->o_wg_sums.ensure(n_wg * 4) ||
      !e->o_wg_prefix.ensure(n_wg * sizeof(OuterPrefix)) || !e->o_result.ensure(sizeof(OuterScanResult)))
    return e->fail(SIMMR_ENOMEM, "outer stream scratch allocation failed");
  HIP_TRY(e, hipMemsetAsync(e->o_result.p, 0, sizeof(OuterScanResult), e->stream));
  hipLaunchKernelGGL(k_outer_classify, dim3((uint32_t)n_wg), dim3(256), 0, e->stream, P, slot_first >> 3, n_blocks, 0u,
                     e->o_last_idx.as<uint32_t>(), e->o_wg_sums.as<uint32_t>());
  hipLaunchKernelGGL(k_outer_scan, dim3(1), dim3(256), 0, e->stream, e->o_wg_sums.as<uint32_t>(), n_wg, (uint64_t)0,
                     (uint64_t)0, e->o_wg_prefix.as<OuterPrefix>(), e->o_result.as<OuterScanResult>());
  OuterScanResult res{};
  HIP_TRY(e, hipMemcpyAsync(&res, e->o_result.p, sizeof res, hipMemcpyDeviceToHost, e->stream));
  if ((rc = sync_check(e, "outer stream summary"))) return rc;
  out->units[0] = res.total_units;
  out->units[1] = res.total_units1;
  out->end_state[0] = res.end_state;
  out->end_state[1] = res.end_state1;
  return SIMMR_OK;
}

static int pe_plan_impl(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile, uint64_t genome_reads,
                        int has_seed, uint64_t seed, simmr_range shard, uint64_t start_slot, uint64_t start_unit,
                        simmr_plan_info* info);

int simmr_pe_plan(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile,
                  uint64_t genome_reads, int has_seed, uint64_t seed, simmr_range shard,
                  simmr_plan_info* info) {
  return pe_plan_impl(e, genome_idx, profile, genome_reads, has_seed, seed, shard, 0, 0, info);
}

int simmr_pe_plan_at(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile,
                     uint64_t genome_reads, uint64_t seed, simmr_range shard, uint64_t start_slot,
                     uint64_t start_unit, simmr_plan_info* info) {
  return pe_plan_impl(e, genome_idx, profile, genome_reads, 1, seed, shard, start_slot, start_unit, info);
}

static int pe_plan_impl(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile, uint64_t genome_reads,
                        int has_seed, uint64_t seed, simmr_range shard, uint64_t start_slot, uint64_t start_unit,
                        simmr_plan_info* info) {
  if (!e) return SIMMR_EINVAL;
  e->plan_kind = PLAN_NONE;
  e->fq_direct = false;  // (a direct FASTQ plan belongs to the plan it was made for)
  HIP_TRY(e, hipSetDevice(e->device));
  PlanScope plan_scope(e);  // (simmr_engine_set_plan_overlap: the other buffer set, the plan stream)
  int rc = check_genome(e, genome_idx);
  if (rc) return rc;
  ProfileDev prof;
  if ((rc = make_profile(e, profile, false, &prof))) return rc;
  uint32_t slot_round = 0;
  if ((rc = plan_slot_round(e, prof, &slot_round))) return rc;
  GenomeHost& g = e->genomes[genome_idx];
  // simulate.rs:220-225: a sequence not larger than minimum_genome_size() is an
  // Err that the caller unwrap()s (simulate.rs:186) — any such sequence can be drawn.
  for (size_t c = 0; c < g.contigs.size(); c++)
    if (g.contigs[c].size <= prof.required)
      return e->fail(SIMMR_EGENOME, "Genome size (%llunt) is smaller than the required length (%u)",
                     (unsigned long long)g.contigs[c].size, prof.required);
  const uint64_t n_pairs = genome_reads / 2;  // simulate.rs:179
  uint64_t first = std::min(shard.first, n_pairs);
  uint64_t count = std::min(shard.count, n_pairs - first);
  if (start_unit > first)
    return e->fail(SIMMR_EINVAL, "start_unit %llu is past the first pair of the shard (%llu)", (unsigned long long)start_unit,
                   (unsigned long long)first);
  if (!has_seed) seed = os_entropy_u64();  // simulate.rs:174 from_entropy()
  const bool seeds2 = prof.kind != SIMMR_K_PERFECT_SHORT;
  if ((rc = ensure_plan_arrays(e, count, seeds2, false))) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_a, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  uint64_t end_slot = 0, total = 0;
  bool presummed = false;
  const bool coarse = plan_is_coarse(e, prof);
  if (count > 0) {
    // the stream is entered at pair start_unit (slot start_slot): units are counted from there
    if (prof.rng_mode == SIMMR_RNG_PHILOX_FULL) {
      // one Philox block per pair, made by the plan kernel: nothing to walk, nothing to seek in (start_slot / start_unit have no meaning)
      rc = SIMMR_OK;
    } else {
      rc = run_outer(e, seed, g.contigs.size(), start_slot, first - start_unit + count, first - start_unit, count,
                     e->u_contig.as<uint32_t>(), e->u_seed.as<uint64_t>(), &end_slot);
    }
    if (rc) return rc;
    PlanArrays pw = plan_arrays(e, seeds2);
    // the mutation seed of mate 2 is only read by the kernels that walk the reference's mutation stream
    if (prof.kind == SIMMR_K_CUSTOM || prof.rng_mode != SIMMR_RNG_REFERENCE) pw.ms2 = nullptr;
    // the plan kernel adds each pair's bytes to its tile of the offset scan (sort_by_length uses the same scratch first)
    presummed = prof.kind != SIMMR_K_PERFECT_SHORT && !coarse &&
                !(prof.kind == SIMMR_K_MINIMAL_SHORT && prof.rng_mode == SIMMR_RNG_REFERENCE);
    unsigned long long* tiles = presummed ? tile_sums_begin(e, count) : nullptr;
    if (presummed && !tiles) return e->fail(SIMMR_ENOMEM, "scan scratch allocation failed");
    if (coarse && !e->w_bytes.ensure(((count + 63) / 64) * 8)) return e->fail(SIMMR_ENOMEM, "offset allocation failed");
    const OuterCtrArgs oc{nullptr, 0u, (uint32_t)g.contigs.size(), seed, first, nullptr, e->u_contig.as<uint32_t>(), e->u_seed.as<uint64_t>()};
    if ((rc = launch_plan_pe(e, prof, genome_idx, count, (const uint32_t*)nullptr, pw, tiles, slot_round,
                             coarse ? e->w_bytes.as<unsigned long long>() : (unsigned long long*)nullptr, oc)))
      return rc;
  }
  e->plan_sorted = false;
  if (prof.kind == SIMMR_K_MINIMAL_SHORT && prof.rng_mode == SIMMR_RNG_REFERENCE &&
      (rc = sort_by_length(e, count, 0)))
    return rc;
  if (prof.kind == SIMMR_K_PERFECT_SHORT && count > 0) {
    total = count * 2ull * prof.read_length;  // constant lengths (perfect_short.rs:22-40): read r starts at r * L
  } else if (coarse) {  // the first output byte of every 64th pair is all the counter-mode emit kernel asks for
    if ((rc = scan_u64(e, e->w_bytes, (count + 63) / 64, e->u_off64, &total))) return rc;
  } else if ((rc = presummed ? scan_presummed<uint32_t>(e, e->u_len, count, 2u, e->u_off, &total, slot_round) : scan_offsets(e, count, 2u, &total, slot_round))) {
    return rc;
  }
  HIP_TRY(e, hipEventRecord(e->ev_b, e->stream));
  uint32_t errw = 0;
  if ((rc = read_err_word(e, &errw))) return rc;
  if (errw & SIMMR_ERRBIT_GENOME) return e->fail(SIMMR_EGENOME, "a sequence is smaller than the required length");
  if (errw & SIMMR_ERRBIT_SLICE)
    return e->fail(SIMMR_ERANGE, "a read would extend past its sequence (the reference panics on this slice)");
  if (errw & SIMMR_ERRBIT_PDF)
    return e->fail(SIMMR_ERANGE, "a custom PDF selected a density without a bin range (the reference panics: index out of bounds)");
  (void)hipEventElapsedTime(&e->last_plan_ms, e->ev_a, e->ev_b);
  e->plan_kind = PLAN_PE;
  e->prof = prof;
  e->plan_genome = genome_idx;
  e->plan_slot = slot_round ? SIMMR_SLOT16 : 0u;
  e->plan_coarse = coarse;
  e->plan_short_ok = !(errw & SIMMR_NOTEBIT_LONGREAD);
  e->plan_first = first;
  e->plan_units = count;
  e->plan_total_bases = total;
  e->plan_paired = true;
  e->plan_multi = false;
  e->plan_any_exc = g.has_exc;
  if (info) {
    memset(info, 0, sizeof *info);
    info->n_units = count;
    info->n_reads = 2 * count;
    info->total_bases = total;
    info->seed_used = seed;
    info->outer_slots = end_slot;
    info->slot_bytes = e->plan_slot;
  }
  return SIMMR_OK;
}

// simulate_pe_reads (simulate.rs:110-150) for several genomes in one plan: the shard is a range of the
// global pair index (genomes concatenated in the given order, ids as the reference's global counter).
int simmr_pe_plan_multi(simmr_engine* e, uint32_t n_genomes, const uint32_t* genome_idx, const uint64_t* genome_reads,
                        const simmr_error_profile* profile, int has_seed, uint64_t seed, simmr_range shard,
                        simmr_plan_info* info) {
  if (!e) return SIMMR_EINVAL;
  e->plan_kind = PLAN_NONE;
  e->fq_direct = false;  // (a direct FASTQ plan belongs to the plan it was made for)
  if (n_genomes == 0 || !genome_idx || !genome_reads) return e->fail(SIMMR_EINVAL, "simmr_pe_plan_multi: no genomes");
  HIP_TRY(e, hipSetDevice(e->device));
  PlanScope plan_scope(e);  // (simmr_engine_set_plan_overlap: the other buffer set, the plan stream)
  int rc;
  ProfileDev prof;
  if ((rc = make_profile(e, profile, false, &prof))) return rc;
  uint32_t slot_round = 0;
  if ((rc = plan_slot_round(e, prof, &slot_round))) return rc;
  if (prof.kind == SIMMR_K_CUSTOM)
    return e->fail(SIMMR_ENOTSUP, "a custom profile is planned one genome at a time (simmr_pe_plan)");
  // global pair ranges of the genomes (simulate.rs:179: num_reads / 2 pairs each)
  std::vector<uint64_t> base(n_genomes + 1, 0);
  for (uint32_t g = 0; g < n_genomes; g++) {
    if ((rc = check_genome(e, genome_idx[g]))) return rc;
    base[g + 1] = base[g] + genome_reads[g] / 2;
  }
  const uint64_t n_pairs = base[n_genomes];
  const uint64_t first = std::min(shard.first, n_pairs);
  const uint64_t count = std::min(shard.count, n_pairs - first);
  if (!has_seed) seed = os_entropy_u64();  // one draw for the call; the reference draws one per genome (simulate.rs:174)
  // genomes that overlap the shard; one outer list per distinct number of sequences
  struct Cls { uint64_t range, need, off; };
  std::vector<Cls> classes;
  std::vector<MultiGenome> mg(n_genomes);
  bool any_exc = false;
  for (uint32_t g = 0; g < n_genomes; g++) {
    const GenomeHost& G = e->genomes[genome_idx[g]];
    mg[g] = MultiGenome{base[g], 0, genome_idx[g], 0};
    const uint64_t lo = std::max(first, base[g]), hi = std::min(first + count, base[g + 1]);
    if (hi <= lo) continue;
    // simulate.rs:220-225: any sequence not larger than minimum_genome_size() can be drawn and is an error
    for (size_t c = 0; c < G.contigs.size(); c++)
      if (G.contigs[c].size <= prof.required)
        return e->fail(SIMMR_EGENOME, "Genome size (%llunt) is smaller than the required length (%u)",
                       (unsigned long long)G.contigs[c].size, prof.required);
    any_exc = any_exc || G.has_exc;
    const uint64_t range = G.contigs.size(), need = hi - base[g];  // local pairs [0, need)
    size_t ci = 0;
    while (ci < classes.size() && classes[ci].range != range) ci++;
    if (ci == classes.size()) classes.push_back(Cls{range, 0, 0});
    classes[ci].need = std::max(classes[ci].need, need);
    mg[g].pad = (uint32_t)ci;
  }
  uint64_t list_total = 0;
  for (Cls& c : classes) { c.off = list_total; list_total += c.need; }
  for (uint32_t g = 0; g < n_genomes; g++) mg[g].cls_off = classes.empty() ? 0 : classes[mg[g].pad].off;
  const bool seeds2 = prof.kind != SIMMR_K_PERFECT_SHORT;
  if ((rc = ensure_plan_arrays(e, count, seeds2, true))) return rc;
  if (!e->m_contig.ensure(std::max<uint64_t>(list_total, 1) * 4) || !e->m_seed.ensure(std::max<uint64_t>(list_total, 1) * 8))
    return e->fail(SIMMR_ENOMEM, "outer list allocation failed");
  if ((rc = upload_vec(e, e->m_genomes, mg))) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_a, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  uint64_t end_slot = 0, total = 0;
  bool presummed = false;
  const bool coarse = plan_is_coarse(e, prof);
  const bool full = prof.rng_mode == SIMMR_RNG_PHILOX_FULL;
  for (const Cls& c : classes) {  // run_outer synchronises, so `mg` has been uploaded when it returns
    if (full) break;  // (no outer lists: every pair's block is made where the pair is planned)
    if ((rc = run_outer(e, seed, c.range, 0, c.need, 0, c.need, e->m_contig.as<uint32_t>() + c.off,
                        e->m_seed.as<uint64_t>() + c.off, &end_slot)))
      return rc;
  }
  if (count > 0) {
    if (!full)  // (the full counter mode: the plan kernel finds each pair's genome and makes its outer draws)
    hipLaunchKernelGGL(k_multi_units, dim3(grid_for(count, 256)), dim3(256), 0, e->stream, e->m_genomes.as<MultiGenome>(),
                       n_genomes, first, count, e->m_contig.as<uint32_t>(), e->m_seed.as<uint64_t>(),
                       e->u_genome.as<uint32_t>(), e->u_contig.as<uint32_t>(), e->u_seed.as<uint64_t>());
    PlanArrays pw = plan_arrays(e, seeds2);
    if (prof.kind == SIMMR_K_CUSTOM || prof.rng_mode != SIMMR_RNG_REFERENCE) pw.ms2 = nullptr;
    presummed = prof.kind != SIMMR_K_PERFECT_SHORT && !coarse && !(prof.kind == SIMMR_K_MINIMAL_SHORT && prof.rng_mode == SIMMR_RNG_REFERENCE);
    unsigned long long* tiles = presummed ? tile_sums_begin(e, count) : nullptr;
    if (presummed && !tiles) return e->fail(SIMMR_ENOMEM, "scan scratch allocation failed");
    if (coarse && !e->w_bytes.ensure(((count + 63) / 64) * 8)) return e->fail(SIMMR_ENOMEM, "offset allocation failed");
    const OuterCtrArgs oc{e->m_genomes.as<MultiGenome>(), n_genomes, 0u, seed, first, e->u_genome.as<uint32_t>(),
                          e->u_contig.as<uint32_t>(), e->u_seed.as<uint64_t>()};
    if ((rc = launch_plan_pe(e, prof, 0u, count, (const uint32_t*)e->u_genome.as<uint32_t>(), pw, tiles, slot_round,
                             coarse ? e->w_bytes.as<unsigned long long>() : (unsigned long long*)nullptr, oc)))
      return rc;
  }
  e->plan_sorted = false;
  if (prof.kind == SIMMR_K_MINIMAL_SHORT && prof.rng_mode == SIMMR_RNG_REFERENCE && (rc = sort_by_length(e, count, 0)))
    return rc;
  if (prof.kind == SIMMR_K_PERFECT_SHORT && count > 0) {
    total = count * 2ull * prof.read_length;
  } else if (coarse) {
    if ((rc = scan_u64(e, e->w_bytes, (count + 63) / 64, e->u_off64, &total))) return rc;
  } else if ((rc = presummed ? scan_presummed<uint32_t>(e, e->u_len, count, 2u, e->u_off, &total, slot_round) : scan_offsets(e, count, 2u, &total, slot_round))) {
    return rc;
  }
  HIP_TRY(e, hipEventRecord(e->ev_b, e->stream));
  uint32_t errw = 0;
  if ((rc = read_err_word(e, &errw))) return rc;  // also: the host vector `mg` may go out of scope now
  if (errw & SIMMR_ERRBIT_GENOME) return e->fail(SIMMR_EGENOME, "a sequence is smaller than the required length");
  if (errw & SIMMR_ERRBIT_SLICE)
    return e->fail(SIMMR_ERANGE, "a read would extend past its sequence (the reference panics on this slice)");
  (void)hipEventElapsedTime(&e->last_plan_ms, e->ev_a, e->ev_b);
  e->plan_kind = PLAN_PE;
  e->prof = prof;
  e->plan_genome = genome_idx[0];
  e->plan_slot = slot_round ? SIMMR_SLOT16 : 0u;
  e->plan_coarse = coarse;
  e->plan_short_ok = !(errw & SIMMR_NOTEBIT_LONGREAD);
  e->plan_first = first;
  e->plan_units = count;
  e->plan_total_bases = total;
  e->plan_paired = true;
  e->plan_multi = true;
  e->plan_any_exc = any_exc;
  if (info) {
    memset(info, 0, sizeof *info);
    info->n_units = count;
    info->n_reads = 2 * count;
    info->total_bases = total;
    info->seed_used = seed;
    info->outer_slots = end_slot;
    info->slot_bytes = e->plan_slot;
  }
  return SIMMR_OK;
}

static int emit_common(simmr_engine* e, uint32_t read_id_base, const simmr_reads_out* out) {
  const uint64_t n_units = e->plan_units;
  const bool paired = e->plan_paired;
  const uint64_t n_reads = paired ? 2 * n_units : n_units;
  int rc = check_out(e, out, n_reads, e->plan_total_bases);
  if (rc) return rc;
  const bool seeds2 = paired && e->prof.kind != SIMMR_K_PERFECT_SHORT;
  PlanArrays pl = plan_arrays(e, seeds2);
  const uint32_t* u_genome = (paired && !e->plan_multi) ? nullptr : e->u_genome.as<uint32_t>();
  // the Philox and perfect-short emit kernels write the metadata columns and the plan counters themselves
  const bool fused = n_units > 0 && (e->prof.kind == SIMMR_K_PERFECT_SHORT ||
                                     (e->prof.kind != SIMMR_K_CUSTOM && e->prof.rng_mode != SIMMR_RNG_REFERENCE));
  if (!fused)
    hipLaunchKernelGGL(k_write_meta, dim3(grid_for(n_units + 1, 256)), dim3(256), 0, e->stream,
                       paired ? 1u : 0u, n_units, e->plan_first, read_id_base, e->plan_genome, pl,
                       e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), u_genome, out_cols(out));
  unsigned long long* counters = e->d_counters.as<unsigned long long>();
  HIP_TRY(e, next_emit_events(e));
  HIP_TRY(e, hipEventRecord(e->ev_c, e->stream));
  if (n_units > 0) {
    if (e->prof.kind == SIMMR_K_PERFECT_SHORT) {
      const uint64_t groups = (n_reads + PERFECT_GROUP - 1) / PERFECT_GROUP;
      const uint32_t grid = (uint32_t)std::min<uint64_t>(groups, (uint64_t)e->n_cu * 8 * e->perfect_mult);
      auto kern = e->plan_multi ? k_emit_perfect_pe<true> : k_emit_perfect_pe<false>;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, e->stream, e->d_genomes.as<GenomeDev>(),
                         e->plan_genome, u_genome, e->plan_any_exc ? 1u : 0u, n_units, e->prof.read_length, pl,
                         e->u_contig.as<uint32_t>(), out->seq,
                         out->qual, 60u + out->qual_offset, e->plan_first, read_id_base, out_cols(out), counters);
    } else if (e->prof.rng_mode != SIMMR_RNG_REFERENCE && e->prof.kind != SIMMR_K_CUSTOM) {  // (a custom model's counter mode: below)
      bool exc = false;
      if (paired) exc = e->plan_any_exc;
      else for (const auto& g : e->genomes) exc = exc || (g.staged && g.has_exc);
      // pairs of one genome with few contigs: the contig bases live in LDS (no dependent load per record)
      const bool cached = paired && !e->plan_multi && e->plan_genome < e->genomes.size() &&
                          e->genomes[e->plan_genome].contigs.size() <= PHILOX_CBASE;
      {
        const uint64_t blocks = (n_units + PHILOX_UNITS - 1) / PHILOX_UNITS;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)e->n_cu * e->philox_wgs_per_cu);
        // an escaped base is noticed through its quality byte when no real one has bit 7 set (kernels.hip: esc_q)
        bool escq = (out->qual_offset & 0xffu) + e->prof.philox_qmax1 <= 127u;
#if defined(SIMMR_NO_ESCQ)
        escq = false;  // test build: the flag-bit form on every input
#endif
        const bool coarse = paired && e->plan_coarse;
        auto kern = philox_kernel(exc, cached, escq, e->plan_slot != 0, coarse);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, e->stream, e->prof, paired ? 1u : 0u,
                           e->d_genomes.as<GenomeDev>(), e->plan_genome, n_units, pl, e->u_off.as<uint64_t>(),
                           e->u_contig.as<uint32_t>(), u_genome, e->u_seed.as<uint64_t>(), out->seq, out->qual,
                           out->qual_offset, e->plan_first, read_id_base, out_cols(out), counters,
                           (const uint8_t*)nullptr, (const FqTemplate*)nullptr, FqTables{}, 0u, 0u, 0u,
                           coarse ? (const uint64_t*)e->u_off64.as<uint64_t>() : (const uint64_t*)nullptr);
      }
    } else if (e->prof.kind == SIMMR_K_CUSTOM && !paired) {
      {
        const bool fast0 = e->prof.custom.kmer_stride != 0 && e->splice_variant != 1;
        if (const char* missing = custom_long_tables_missing(e->prof, fast0, e->prof.rng_mode != SIMMR_RNG_REFERENCE))
          return e->fail(SIMMR_EINVAL, "custom long-read emit refused: device table `%s` of the model is not set for rng_mode %u "
                                       "(nothing was launched)", missing, e->prof.rng_mode);
      }
      HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
      bool exc = false;
      for (const auto& g : e->genomes) exc = exc || (g.staged && g.has_exc);
      const uint64_t blocks = (n_reads + 255) / 256;
      const uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)e->n_cu * 8 * e->custom_long_mult);
      const uint32_t* order = e->plan_sorted ? e->u_order.as<uint32_t>() : (const uint32_t*)nullptr;
      hipLaunchKernelGGL(k_custom_long_qual, dim3(grid), dim3(256), 0, e->stream, e->prof, n_units, order, pl,
                         e->u_off.as<uint64_t>(), e->u_seed.as<uint64_t>(), out->qual, out->qual_offset, counters,
                         e->d_err.as<uint32_t>());
      bool fast = e->prof.custom.kmer_stride != 0;
      if (e->splice_variant == 1) fast = false;  // the two-load kernel (A/B timing)
      if (e->prof.rng_mode != SIMMR_RNG_REFERENCE) {
        // the counter mode of the splice (kernels.hip section 9c, CTR): the LDS holds one word per k-mer (4^k words);
        // with k = 7 (64 KB) two workgroups of 768 lanes share a CU — six waves per SIMD, what the kernel's registers allow —
        // with smaller tables 256-lane workgroups do
        auto kern = fast ? (exc ? k_custom_long_splice<true, true, true> : k_custom_long_splice<false, true, true>)
                         : (exc ? k_custom_long_splice<true, false, true> : k_custom_long_splice<false, false, true>);
        const uint32_t tab = fast ? splice_ctr_lds_bytes(e->prof.custom.kmer_size) : 0u;
        const uint32_t lanes = tab > 16384u ? SPLICE_CTR_LANES_MAX : 256u;
        const uint32_t lds = fast ? tab + 16u * lanes : 0u;  // the k-mer table + 16 bytes per lane (an even group waiting for its odd neighbour's store)
        if (lds > 32768u && lds > e->splice_ctr_lds_set[exc ? 1 : 0]) {  // (64 KB of table + the kernel's static LDS: over the default limit)
          HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          e->splice_ctr_lds_set[exc ? 1 : 0] = lds;
        }
        // (many more workgroups than are resident: reads come longest first, and the tail of the launch is short ones)
        const uint32_t cgrid = (uint32_t)std::min<uint64_t>((n_reads + lanes - 1) / lanes, (uint64_t)e->n_cu * 64 * e->custom_long_mult);
        hipLaunchKernelGGL(kern, dim3(cgrid), dim3(lanes), lds, e->stream, e->prof, e->d_genomes.as<GenomeDev>(), n_units, order,
                           pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), e->u_genome.as<uint32_t>(),
                           e->u_seed.as<uint64_t>(), out->seq, counters, e->d_err.as<uint32_t>());
      } else if (fast) {
        // one workgroup of 1024 lanes per CU around the LDS count table (kernels.hip section 9c)
        auto kern = exc ? k_custom_long_splice<true, true> : k_custom_long_splice<false, true>;
        const uint32_t lds = splice_fast_lds_bytes(e->prof.custom.kmer_size);
        // not once per emit: the limit is per device and grows with the model's k (132 KB + 4^k), so each engine keeps
        // the largest size it has set per kernel variant and sets it again only when a model needs more
        if (lds > e->splice_lds_set[exc ? 1 : 0]) {
          HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          e->splice_lds_set[exc ? 1 : 0] = lds;
        }
        const uint32_t fgrid = (uint32_t)std::min<uint64_t>((n_reads + SPLICE_FAST_LANES - 1) / SPLICE_FAST_LANES, (uint64_t)e->n_cu * 8 * e->custom_long_mult);
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(SPLICE_FAST_LANES), lds, e->stream, e->prof, e->d_genomes.as<GenomeDev>(),
                           n_units, order, pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), e->u_genome.as<uint32_t>(),
                           e->u_seed.as<uint64_t>(), out->seq, counters, e->d_err.as<uint32_t>());
      } else {
        auto kern = exc ? k_custom_long_splice<true, false> : k_custom_long_splice<false, false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, e->stream, e->prof, e->d_genomes.as<GenomeDev>(), n_units, order,
                           pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), e->u_genome.as<uint32_t>(),
                           e->u_seed.as<uint64_t>(), out->seq, counters, e->d_err.as<uint32_t>());
      }
    } else if (e->prof.kind == SIMMR_K_CUSTOM) {
      HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
      const uint64_t blocks = (n_units + 255) / 256;  // one lane per pair
      const uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)e->n_cu * 8 * e->custom_pe_mult);
      // qualities: one lane per pair (k_emit_custom_pe); bases: the item kernel without draws (coalesced stores)
      hipLaunchKernelGGL(k_emit_custom_pe, dim3(grid), dim3(256), 0, e->stream, e->prof, e->d_genomes.as<GenomeDev>(),
                         e->plan_genome, n_units, pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(),
                         e->u_seed.as<uint64_t>(), out->seq, out->qual, out->qual_offset, counters,
                         e->d_err.as<uint32_t>());
      const uint64_t cblocks = (n_units + PHILOX_UNITS - 1) / PHILOX_UNITS;
      const uint32_t cgrid = (uint32_t)std::min<uint64_t>(cblocks, (uint64_t)e->n_cu * e->philox_wgs_per_cu);
      auto copy = e->plan_any_exc ? k_emit_philox<true, true, false> : k_emit_philox<false, true, false>;
      hipLaunchKernelGGL(copy, dim3(cgrid), dim3(256), 0, e->stream, e->prof, 1u, e->d_genomes.as<GenomeDev>(),
                         e->plan_genome, n_units, pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(),
                         (const uint32_t*)nullptr, e->u_seed.as<uint64_t>(), out->seq, out->qual, out->qual_offset,
                         e->plan_first, read_id_base, out_cols(out), counters, (const uint8_t*)nullptr,
                         (const FqTemplate*)nullptr, FqTables{}, 0u, 0u, 0u, (const uint64_t*)nullptr);
    } else {
      // lane-per-read kernel: template on (exception plane present, paired, perfect-long Phred)
      bool exc = false;
      if (paired) exc = e->plan_any_exc;
      else for (const auto& g : e->genomes) exc = exc || (g.staged && g.has_exc);
      const bool pl_kind = e->prof.kind == SIMMR_K_PERFECT_LONG;
      using KernT = void (*)(ProfileDev, const GenomeDev*, uint32_t, uint64_t, const uint32_t*, PlanArrays,
                             const uint64_t*, const uint32_t*, const uint32_t*, const uint64_t*, uint8_t*, uint8_t*,
                             uint32_t, const Tables*, unsigned long long*);
      KernT kern;
      if (paired) kern = exc ? k_emit_lanes<true, true, false> : k_emit_lanes<false, true, false>;
      else if (pl_kind) kern = exc ? k_emit_lanes<true, false, true> : k_emit_lanes<false, false, true>;
      else kern = exc ? k_emit_lanes<true, false, false> : k_emit_lanes<false, false, false>;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, LANES_WG, 0) != hipSuccess || per_cu < 1) per_cu = 1;
      const uint64_t n_tasks = paired ? 2 * n_units : n_units;
      const uint64_t wgs = (n_tasks + LANES_WG - 1) / LANES_WG;
      const uint32_t grid = (uint32_t)std::min<uint64_t>(wgs, (uint64_t)e->n_cu * (uint64_t)per_cu * e->lanes_mult);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(LANES_WG), 0, e->stream, e->prof, e->d_genomes.as<GenomeDev>(),
                         e->plan_genome, n_units, e->plan_sorted ? e->u_order.as<uint32_t>() : (const uint32_t*)nullptr,
                         pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), u_genome, e->u_seed.as<uint64_t>(),
                         out->seq, out->qual, out->qual_offset, e->d_tables.as<Tables>(), counters);
    }
  }
  HIP_TRY(e, hipEventRecord(e->ev_d, e->stream));
  if (n_units > 0 && !fused) {
    const bool perfect = e->prof.kind == SIMMR_K_PERFECT_SHORT;
    const bool acgt_all = perfect && !e->plan_any_exc;
    hipLaunchKernelGGL(k_count_plan, dim3(std::min<uint32_t>(grid_for(n_units, 256), (uint32_t)e->n_cu * 4)), dim3(256), 0, e->stream, paired ? 1u : 0u,
                       n_units, pl, perfect ? 60u : 0u, acgt_all ? 1u : 0u, counters);
  }
  hipError_t s = hipGetLastError();
  if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "emit launch failed: %s", hipGetErrorString(s));
  if (n_units > 0 && e->prof.kind == SIMMR_K_CUSTOM) {
    uint32_t errw = 0;
    if ((rc = read_err_word(e, &errw))) return rc;
    if (errw & SIMMR_ERRBIT_PDF)
      return e->fail(SIMMR_ERANGE, "a custom quality PDF selected a density without a bin range (the reference panics: index out of bounds)");
    if (errw & SIMMR_ERRBIT_KMER)
      return e->fail(SIMMR_ERANGE, "simulate_errors chose an alternate k-mer with a deletion or an invalid code, or met unusable weights "
                                   "(the reference panics, or returns fewer bases than qualities, custom_short.rs:475-509)");
  }
  return SIMMR_OK;
}

int simmr_pe_emit(simmr_engine* e, uint32_t read_id_base, const simmr_reads_out* out) {
  if (!e) return SIMMR_EINVAL;
  if (e->plan_kind != PLAN_PE) return e->fail(SIMMR_ESTATE, "simmr_pe_emit called without a paired-end plan");
  HIP_TRY(e, hipSetDevice(e->device));
  return emit_common(e, read_id_base, out);
}

// ---- long reads -------------------------------------------------------------------
int simmr_long_plan(simmr_engine* e, uint32_t n_genomes, const uint32_t* genome_idx,
                    const uint64_t* genome_reads, const simmr_error_profile* profile, int has_seed,
                    uint64_t seed, simmr_range shard, simmr_plan_info* info) {
  if (!e) return SIMMR_EINVAL;
  e->plan_kind = PLAN_NONE;
  e->fq_direct = false;  // (a direct FASTQ plan belongs to the plan it was made for)
  HIP_TRY(e, hipSetDevice(e->device));
  PlanScope plan_scope(e);  // (simmr_engine_set_plan_overlap: the other buffer set, the plan stream)
  if (!genome_idx || !genome_reads || n_genomes == 0) return e->fail(SIMMR_EINVAL, "no genomes");
  int rc;
  ProfileDev prof;
  if ((rc = make_profile(e, profile, true, &prof))) return rc;
  uint32_t slot_round = 0;
  if ((rc = plan_slot_round(e, prof, &slot_round))) return rc;
  uint64_t total_reads = 0;
  for (uint32_t g = 0; g < n_genomes; g++) {
    if ((rc = check_genome(e, genome_idx[g]))) return rc;
    total_reads += genome_reads[g];
  }
  uint64_t first = std::min(shard.first, total_reads);
  uint64_t count = std::min(shard.count, total_reads - first);
  const bool per_read = !has_seed || profile->length_mode == SIMMR_LEN_PER_READ;
  if (prof.rng_mode == SIMMR_RNG_PHILOX_FULL && !per_read)
    return e->fail(SIMMR_EINVAL, "SIMMR_RNG_PHILOX_FULL plans long reads with SIMMR_LEN_PER_READ only (the one constant length of a "
                                 "seeded reference run, simulate.rs:358, is a property of the reference's stream)");
  if (!has_seed) seed = os_entropy_u64();
  if ((rc = ensure_plan_arrays(e, count, false, true))) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_a, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));

  // runs of consecutive reads per genome (simulate.rs:353-356)
  std::vector<LongGenomeRun> runs;
  {
    uint64_t acc = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
      if (genome_reads[g] == 0) continue;
      LongGenomeRun r{};
      r.first_read = acc;
      r.n_reads = genome_reads[g];
      r.genome = genome_idx[g];
      r.max_size = e->genomes[genome_idx[g]].max_size;
      runs.push_back(r);
      acc += genome_reads[g];
    }
  }
  uint32_t L0 = 0;
  uint64_t end_slot = 0, total = 0;
  if (count > 0 && !per_read) {
    // get_random_read_length(seed): the same value for every read (simulate.rs:358)
    uint32_t* d_L0 = e->d_scalars.as<uint32_t>();
    hipLaunchKernelGGL(k_const_length, dim3(1), dim3(64), 0, e->stream, prof, seed, e->d_tables.as<Tables>(), d_L0);
    HIP_TRY(e, hipMemcpyAsync(&L0, d_L0, 4, hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_check(e, "k_const_length"))) return rc;
    if (L0 == 0) return e->fail(SIMMR_ERANGE, "the run-wide read length drew 0 (the reference panics in gen_range(0..0))");
    // usable sequences per genome (simulate.rs:362-367)
    std::vector<uint32_t> usable;
    std::vector<size_t> uoff(runs.size());
    for (size_t r = 0; r < runs.size(); r++) {
      const GenomeHost& g = e->genomes[runs[r].genome];
      uoff[r] = usable.size();
      uint32_t n = 0;
      for (size_t c = 0; c < g.contigs.size(); c++)
        if (g.contigs[c].size > L0) { usable.push_back((uint32_t)c); n++; }
      runs[r].n_usable = n;
      if (n == 0)
        return e->fail(SIMMR_EGENOME,
                       "genome %u has no sequence longer than the read length %u (the reference loops forever, simulate.rs:370)",
                       runs[r].genome, L0);
    }
    if (!e->d_usable.ensure(std::max<size_t>(usable.size(), 1) * 4)) return e->fail(SIMMR_ENOMEM, "usable table allocation failed");
    HIP_TRY(e, hipMemcpyAsync(e->d_usable.p, usable.data(), usable.size() * 4, hipMemcpyHostToDevice, e->stream));
    for (size_t r = 0; r < runs.size(); r++) runs[r].usable = e->d_usable.as<uint32_t>() + uoff[r];
    // ONE StdRng across all genomes (simulate.rs:348): walk it segment by
    // segment; a segment is a maximal range of runs with the same gen_range bound.
    uint64_t slot = 0;
    size_t r = 0;
    while (r < runs.size()) {
      size_t r2 = r;
      uint64_t seg_reads = 0;
      while (r2 < runs.size() && runs[r2].n_usable == runs[r].n_usable) { seg_reads += runs[r2].n_reads; r2++; }
      const uint64_t seg_first = runs[r].first_read;
      if (seg_first >= first + count) break;  // nothing of the shard lies beyond
      // intersection of the shard with this segment, relative to the segment
      const uint64_t lo = std::max(first, seg_first), hi = std::min(first + count, seg_first + seg_reads);
      const uint64_t e_first = lo > seg_first ? lo - seg_first : 0;
      const uint64_t e_count = hi > lo ? hi - lo : 0;
      // units needed from this segment: all of it if the shard continues past it
      const uint64_t n_total = (first + count > seg_first + seg_reads) ? seg_reads : (hi - seg_first);
      rc = run_outer(e, seed, runs[r].n_usable, slot, n_total, e_first, e_count,
                     e->u_contig.as<uint32_t>() + (lo - first), e->u_seed.as<uint64_t>() + (lo - first), &slot);
      if (rc) return rc;
      r = r2;
    }
    end_slot = slot;
  }
  if (!e->d_runs.ensure(std::max<size_t>(runs.size(), 1) * sizeof(LongGenomeRun))) return e->fail(SIMMR_ENOMEM, "run table allocation failed");
  HIP_TRY(e, hipMemcpyAsync(e->d_runs.p, runs.data(), runs.size() * sizeof(LongGenomeRun), hipMemcpyHostToDevice, e->stream));
  if (count > 0) {
    if (!per_read) {
      hipLaunchKernelGGL(k_plan_long_ref, dim3(grid_for(count, PLAN_THREADS)), dim3(PLAN_THREADS), 0, e->stream,
                         e->d_genomes.as<GenomeDev>(), e->d_runs.as<LongGenomeRun>(), (uint32_t)runs.size(), first,
                         count, L0, prof.long_start_uniform, e->u_contig.as<uint32_t>(), e->u_genome.as<uint32_t>(),
                         e->u_seed.as<uint64_t>(), plan_arrays(e, false), e->d_err.as<uint32_t>());
    } else {
      auto plan_kern = prof.rng_mode == SIMMR_RNG_PHILOX_FULL ? k_plan_long_per_read<true> : k_plan_long_per_read<false>;
      hipLaunchKernelGGL(plan_kern, dim3(grid_for(count, PLAN_THREADS)), dim3(PLAN_THREADS), 0,
                         e->stream, prof, e->d_genomes.as<GenomeDev>(), e->d_runs.as<LongGenomeRun>(),
                         (uint32_t)runs.size(), seed, first, count, e->u_contig.as<uint32_t>(),
                         e->u_genome.as<uint32_t>(), e->u_seed.as<uint64_t>(), plan_arrays(e, false),
                         e->d_tables.as<Tables>(), e->d_err.as<uint32_t>());
    }
  }
  e->plan_sorted = false;
  // (the lane-per-read kernels: the reference's streams, and a custom model in either mode)
  if ((prof.rng_mode == SIMMR_RNG_REFERENCE || prof.kind == SIMMR_K_CUSTOM) && (rc = sort_by_length(e, count, 6))) return rc;
  if ((rc = scan_offsets(e, count, 1u, &total, slot_round))) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_b, e->stream));
  uint32_t errw = 0;
  if ((rc = read_err_word(e, &errw))) return rc;
  if (errw & SIMMR_ERRBIT_GENOME) return e->fail(SIMMR_EGENOME, "no usable sequence for a drawn read length");
  if (errw & SIMMR_ERRBIT_SLICE)
    return e->fail(SIMMR_ERANGE, "a read would extend past its sequence (the reference panics on this slice)");
  (void)hipEventElapsedTime(&e->last_plan_ms, e->ev_a, e->ev_b);
  e->plan_kind = PLAN_LONG;
  e->prof = prof;
  e->plan_genome = 0;
  e->plan_slot = slot_round ? SIMMR_SLOT16 : 0u;
  e->plan_coarse = false;
  e->plan_short_ok = false;
  e->plan_first = first;
  e->plan_units = count;
  e->plan_total_bases = total;
  e->plan_paired = false;
  if (info) {
    memset(info, 0, sizeof *info);
    info->n_units = count;
    info->n_reads = count;
    info->total_bases = total;
    info->seed_used = seed;
    info->outer_slots = end_slot;
    info->const_read_length = per_read ? 0 : L0;
    info->slot_bytes = e->plan_slot;
  }
  return SIMMR_OK;
}

int simmr_long_emit(simmr_engine* e, uint32_t read_id_base, const simmr_reads_out* out) {
  if (!e) return SIMMR_EINVAL;
  if (e->plan_kind != PLAN_LONG) return e->fail(SIMMR_ESTATE, "simmr_long_emit called without a long-read plan");
  HIP_TRY(e, hipSetDevice(e->device));
  return emit_common(e, read_id_base, out);
}

// ---- FASTQ framing (fastq.rs:14-124) -------------------------------------------------
namespace {
// fastq.rs:34-56 chains String::replace over the template in this order.  The chain equals one left-to-right scan
// for the seven patterns unless a replacement completes a placeholder of a LATER step.  Ids are refused when they hold
// a brace, so the braces of such a placeholder would have to be literal text of the template with a replaced value
// between them; the values of read_id / start / end are digits and those of reverse_complement / pair are t, f, 1, 2 —
// no placeholder name of a later step can contain them — which leaves genome_id (step 1) and sequence_id (step 3):
// "{:read_{:genome_id:}:}" with the genome id "id" becomes "{:read_id:}", and step 2 replaces it.  *risky says that
// the template has a genome_id or sequence_id field with a '{' in literal text before it and a '}' in literal text
// after it; the caller then answers SIMMR_ENOTSUP and the host writer, which replays the chain, takes over.
bool compile_header_format(const char* fmt, std::vector<uint8_t>* blob, FqTemplate* tp, bool* risky) {
  *risky = false;
  static const struct { const char* pat; uint32_t kind; } pats[] = {
      {"{:genome_id:}", FQ_GENOME_ID}, {"{:read_id:}", FQ_READ_ID}, {"{:sequence_id:}", FQ_SEQUENCE_ID},
      {"{:start_position:}", FQ_START}, {"{:end_position:}", FQ_END}, {"{:reverse_complement:}", FQ_REVCOMP},
      {"{:pair:}", FQ_PAIR}};
  tp->n_segs = 0;
  const size_t n = strlen(fmt);
  size_t i = 0, lit0 = 0;
  auto push = [&](uint32_t kind, uint32_t off, uint32_t len) {
    if (tp->n_segs >= FQ_MAX_SEGS) return false;
    tp->segs[tp->n_segs++] = FqSeg{kind, off, len};
    return true;
  };
  auto flush = [&](size_t end) {
    if (end == lit0) return true;
    while (blob->size() & 7u) blob->push_back(0);  // literals start on 8-byte boundaries: aligned 8-byte LDS reads (fq_put_bytes)
    const uint32_t off = (uint32_t)blob->size();
    blob->insert(blob->end(), fmt + lit0, fmt + end);
    return push(FQ_LITERAL, off, (uint32_t)(end - lit0));
  };
  while (i < n) {
    bool hit = false;
    if (fmt[i] == '{')
      for (const auto& p : pats) {
        const size_t m = strlen(p.pat);
        if (strncmp(fmt + i, p.pat, m) == 0) {
          if (!flush(i) || !push(p.kind, 0, 0)) return false;
          i += m;
          lit0 = i;
          hit = true;
          break;
        }
      }
    if (!hit) i++;
  }
  if (!flush(n)) return false;
  for (uint32_t a = 0; a < tp->n_segs; a++) {
    if (tp->segs[a].kind != FQ_GENOME_ID && tp->segs[a].kind != FQ_SEQUENCE_ID) continue;
    bool open_before = false, close_after = false;
    for (uint32_t b = 0; b < tp->n_segs; b++) {
      if (tp->segs[b].kind != FQ_LITERAL) continue;
      const uint8_t* t = blob->data() + tp->segs[b].off;
      for (uint32_t k = 0; k < tp->segs[b].len; k++) {
        if (b < a && t[k] == '{') open_before = true;
        if (b > a && t[k] == '}') close_after = true;
      }
    }
    if (open_before && close_after) *risky = true;
  }
  return true;
}
bool has_brace(const char* s) { return strchr(s, '{') || strchr(s, '}'); }
}  // namespace

// the header template and the id tables of simmr_fastq_plan / simmr_fastq_plan_direct, compiled and uploaded
static int fq_prepare(simmr_engine* e, const char* header_format, const simmr_fastq_names* names, uint32_t* lit_bytes_out,
                      uint32_t* n_slots_out) {
  std::vector<uint8_t> blob;
  bool risky = false;
  if (!compile_header_format(header_format, &blob, &e->fq_tpl, &risky))
    return e->fail(SIMMR_ENOTSUP, "header format has more than %d pieces", FQ_MAX_SEGS);
  if (risky)
    return e->fail(SIMMR_ENOTSUP, "the header format has braces around a genome / sequence id: the reference's chain of "
                                  "replacements may build a placeholder out of them (fastq.rs:34-56)");
  const uint32_t lit_bytes = (uint32_t)blob.size();  // the literals come first in the blob
  if (lit_bytes > FQ_LIT_MAX) return e->fail(SIMMR_ENOTSUP, "header format has more than %u literal bytes", FQ_LIT_MAX);
  uint32_t n_slots = 0;
  for (uint32_t g = 0; g < names->n_genomes; g++) {
    // a name for a genome index the engine has never seen is a caller error (and index + 1 below must not wrap)
    if (names->genome_idx[g] >= e->genomes.size())
      return e->fail(SIMMR_EINVAL, "simmr_fastq_plan: genome index %u was never staged", names->genome_idx[g]);
    n_slots = std::max(n_slots, names->genome_idx[g] + 1);
  }
  std::vector<uint32_t> gid_off(std::max(n_slots, 1u), 0), gid_len(std::max(n_slots, 1u), 0), cbase(std::max(n_slots, 1u), 0),
      ncontig(std::max(n_slots, 1u), 0), coff, clen;
  size_t row = 0;
  for (uint32_t g = 0; g < names->n_genomes; g++) {
    const uint32_t slot = names->genome_idx[g];
    const char* id = names->genome_id[g];
    if (!id) return e->fail(SIMMR_EINVAL, "genome id %u is NULL", g);
    if (has_brace(id)) return e->fail(SIMMR_ENOTSUP, "genome id '%s' contains a brace", id);
    gid_off[slot] = (uint32_t)blob.size();
    gid_len[slot] = (uint32_t)strlen(id);
    blob.insert(blob.end(), id, id + strlen(id));
    cbase[slot] = (uint32_t)coff.size();
    ncontig[slot] = names->n_contigs[g];
    for (uint32_t c = 0; c < names->n_contigs[g]; c++, row++) {
      const char* sid = names->sequence_id[row];
      if (!sid) return e->fail(SIMMR_EINVAL, "sequence id %zu is NULL", row);
      if (has_brace(sid)) return e->fail(SIMMR_ENOTSUP, "sequence id '%s' contains a brace", sid);
      coff.push_back((uint32_t)blob.size());
      clen.push_back((uint32_t)strlen(sid));
      blob.insert(blob.end(), sid, sid + strlen(sid));
    }
  }
  if (blob.size() > 0xfffffff0ull) return e->fail(SIMMR_ENOTSUP, "names exceed 4 GiB");
  blob.insert(blob.end(), 8, 0);  // ids are read in 8-byte pieces
  int rc;
  if ((rc = upload_vec(e, e->fq_blob, blob)) || (rc = upload_vec(e, e->fq_gid_off, gid_off)) ||
      (rc = upload_vec(e, e->fq_gid_len, gid_len)) || (rc = upload_vec(e, e->fq_cbase, cbase)) ||
      (rc = upload_vec(e, e->fq_ncontig, ncontig)) || (rc = upload_vec(e, e->fq_coff, coff)) ||
      (rc = upload_vec(e, e->fq_clen, clen)))
    return rc;
  if (!e->fq_tpl_dev.ensure(sizeof(FqTemplate))) return e->fail(SIMMR_ENOMEM, "template allocation failed");
  HIP_TRY(e, hipMemcpyAsync(e->fq_tpl_dev.p, &e->fq_tpl, sizeof(FqTemplate), hipMemcpyHostToDevice, e->stream));
  if ((rc = sync_check(e, "fastq table upload"))) return rc;  // the host vectors go out of scope
  *lit_bytes_out = lit_bytes;
  *n_slots_out = n_slots;
  return SIMMR_OK;
}

// what the sizing kernels need of the compiled template (fastq_format.hpp: FqLenCoef)
static FqLenCoef fq_len_coef(const FqTemplate& t) {
  FqLenCoef c{0u, 0u, 0u, 0u, 0u, 0u};
  for (uint32_t s = 0; s < t.n_segs; s++) {
    switch (t.segs[s].kind) {
      case FQ_LITERAL: c.h0 += t.segs[s].len; break;
      case FQ_GENOME_ID: c.n_gid++; break;
      case FQ_READ_ID: c.n_rid++; break;
      case FQ_SEQUENCE_ID: c.n_sid++; break;
      case FQ_START: c.n_start++; break;
      case FQ_END: c.n_end++; break;
      default: c.h0 += 1u; break;  // 't' / 'f', '1' / '2'
    }
  }
  return c;
}

// bytes between the LDS header slots of neighbouring lanes: an ODD number of words, so that the 64 lanes of a wave,
// each writing byte k of its own header, hit 64 different banks (a multiple of 16 bytes made every byte store a
// 4-way bank conflict: SQ_LDS_BANK_CONFLICT was 70 % of the LDS-busy cycles of the header kernel)
static uint32_t fq_slot_pitch(uint32_t bytes) {
  uint32_t p = (bytes + 3u) & ~3u;
  if (((p >> 2) & 1u) == 0) p += 4u;
  return p;
}

static FqTables fq_tables(const simmr_engine* e, uint32_t n_slots) {
  return FqTables{e->fq_blob.as<uint8_t>(), e->fq_gid_off.as<uint32_t>(), e->fq_gid_len.as<uint32_t>(),
                  e->fq_cbase.as<uint32_t>(), e->fq_ncontig.as<uint32_t>(), e->fq_coff.as<uint32_t>(),
                  e->fq_clen.as<uint32_t>(), n_slots};
}

int simmr_fastq_plan(simmr_engine* e, const char* header_format, const simmr_fastq_names* names,
                     const simmr_reads_out* reads, uint64_t n_reads, int paired, uint64_t* total_bytes) {
  if (!e) return SIMMR_EINVAL;
  e->fq_ready = false;
  e->fq_direct = false;
  if (!header_format || !names || !reads || !total_bytes) return e->fail(SIMMR_EINVAL, "simmr_fastq_plan: NULL argument");
  if (!reads->seq_off || !reads->start || !reads->end || !reads->contig || !reads->genome || !reads->read_id ||
      !reads->flags || (n_reads > 0 && (!reads->seq || !reads->qual)))
    return e->fail(SIMMR_EINVAL, "simmr_fastq_plan needs every column of simmr_reads_out");
  HIP_TRY(e, hipSetDevice(e->device));
  uint32_t lit_bytes = 0, n_slots = 0;
  int rc;
  if ((rc = fq_prepare(e, header_format, names, &lit_bytes, &n_slots))) return rc;
  if (!e->fq_len.ensure(std::max<uint64_t>(n_reads, 1) * 8)) return e->fail(SIMMR_ENOMEM, "record length allocation failed");
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  const FqTables tb = fq_tables(e, n_slots);
  const FqReads rd{reads->seq, reads->qual, reads->seq_off, reads->start, reads->end, reads->contig, reads->genome,
                   reads->read_id, reads->flags, reads->slot_bytes == SIMMR_SLOT16 ? 1u : 0u};
  if (n_reads > 0)
    hipLaunchKernelGGL(k_fastq_size, dim3(grid_for(n_reads, 256)), dim3(256), 0, e->stream, fq_len_coef(e->fq_tpl), tb, rd, n_reads,
                       e->fq_len.as<uint64_t>(), e->d_err.as<uint32_t>());
  uint64_t total = 0;
  if ((rc = scan_u64(e, e->fq_len, n_reads, e->fq_off, &total))) return rc;  // also waits for the uploads
  uint32_t errw2[2] = {0, 0};  // error bits, longest header
  HIP_TRY(e, hipMemcpyAsync(errw2, e->d_err.p, 8, hipMemcpyDeviceToHost, e->stream));
  if ((rc = sync_check(e, "fastq size readback"))) return rc;
  if (errw2[0] & SIMMR_ERRBIT_FASTQ)
    return e->fail(SIMMR_ENOTSUP, "a FASTQ header is longer than %u bytes, or a read names a genome / contig without an id", FQ_HMAX - 1);
  e->fq_hpitch = fq_slot_pitch(errw2[1] + 1u + 8u + 16u);  // header, '\n', slack of the 8-byte id copies and 16-byte window reads
  e->fq_reads = n_reads;
  e->fq_total = total;
  e->fq_slots = n_slots;
  e->fq_lit_bytes = lit_bytes;
  e->fq_paired = paired != 0;
  e->fq_ready = true;
  *total_bytes = total;
  return SIMMR_OK;
}

int simmr_fastq_emit(simmr_engine* e, const simmr_reads_out* reads, uint8_t* dst, uint64_t dst_capacity) {
  if (!e) return SIMMR_EINVAL;
  if (!e->fq_ready || e->fq_direct) return e->fail(SIMMR_ESTATE, "simmr_fastq_emit called without simmr_fastq_plan");
  if (!reads) return e->fail(SIMMR_EINVAL, "reads is NULL");
  if (dst_capacity < e->fq_total)
    return e->fail(SIMMR_ERANGE, "dst_capacity %llu < %llu bytes planned", (unsigned long long)dst_capacity,
                   (unsigned long long)e->fq_total);
  if (e->fq_reads == 0) return SIMMR_OK;
  if (!dst) return e->fail(SIMMR_EINVAL, "dst is NULL");
  HIP_TRY(e, hipSetDevice(e->device));
  const FqTables tb = fq_tables(e, e->fq_slots);
  const FqReads rd{reads->seq, reads->qual, reads->seq_off, reads->start, reads->end, reads->contig, reads->genome,
                   reads->read_id, reads->flags, reads->slot_bytes == SIMMR_SLOT16 ? 1u : 0u};
  const uint64_t n_batches = (e->fq_reads + FQ_BATCH - 1) / FQ_BATCH;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n_batches + 3) / 4, (uint64_t)e->n_cu * 8 * e->fastq_mult);
  const uint32_t hdr_lds = 4 * FQ_BATCH * e->fq_hpitch;  // up to 68 KB with 255-byte headers: above the default limit
  if (hdr_lds > 48 * 1024)
    HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_fastq_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hdr_lds));
  HIP_TRY(e, next_emit_events(e));
  HIP_TRY(e, hipEventRecord(e->ev_c, e->stream));
  hipLaunchKernelGGL(k_fastq_write, dim3(grid), dim3(256), hdr_lds, e->stream, e->fq_tpl_dev.as<FqTemplate>(), tb, rd,
                     e->fq_reads, e->fq_paired ? 1u : 0u, e->fq_lit_bytes, e->fq_hpitch, e->fq_off.as<uint64_t>(), dst);
  HIP_TRY(e, hipEventRecord(e->ev_d, e->stream));
  hipError_t s = hipGetLastError();
  if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "fastq launch failed: %s", hipGetErrorString(s));
  return SIMMR_OK;
}

// ---- FASTQ text straight from the plan (include/simmr_hip.h) --------------------------------------------------
static FqPlan fq_plan_view(simmr_engine* e) {
  const bool paired = e->plan_paired;
  const bool seeds2 = paired && e->prof.kind != SIMMR_K_PERFECT_SHORT;
  FqPlan pn;
  pn.pl = plan_arrays(e, seeds2);
  pn.u_contig = e->u_contig.as<uint32_t>();
  pn.u_genome = (paired && !e->plan_multi) ? nullptr : e->u_genome.as<uint32_t>();
  pn.first_unit = e->plan_first;
  pn.read_id_base = e->fq_read_id_base;
  pn.genome_const = e->plan_genome;
  pn.paired = paired ? 1u : 0u;
  return pn;
}

// does the current plan's emit kernel write into FASTQ text?  The counter-mode item kernel does, and so does its copy-only
// form for perfect-short (no draws at all: the planned bases, a constant quality line; perfect_short.rs:24-44)
static bool fq_direct_kernel(const simmr_engine* e) {
  if (e->prof.kind == SIMMR_K_PERFECT_SHORT) return true;
  return e->prof.kind != SIMMR_K_CUSTOM && e->prof.rng_mode != SIMMR_RNG_REFERENCE;
}

int simmr_fastq_plan_direct(simmr_engine* e, const char* header_format, const simmr_fastq_names* names,
                            uint32_t read_id_base, uint64_t* total_bytes) {
  if (!e) return SIMMR_EINVAL;
  e->fq_ready = false;
  e->fq_direct = false;
  if (!header_format || !names || !total_bytes) return e->fail(SIMMR_EINVAL, "simmr_fastq_plan_direct: NULL argument");
  if (e->plan_kind == PLAN_NONE) return e->fail(SIMMR_ESTATE, "simmr_fastq_plan_direct called without a plan");
  HIP_TRY(e, hipSetDevice(e->device));
  uint32_t lit_bytes = 0, n_slots = 0;
  int rc;
  if ((rc = fq_prepare(e, header_format, names, &lit_bytes, &n_slots))) return rc;
  const uint64_t n_reads = e->plan_paired ? 2 * e->plan_units : e->plan_units;
  if (!e->fq_len.ensure(std::max<uint64_t>(n_reads, 1) * 8) || !e->fq_hlen.ensure(std::max<uint64_t>(n_reads, 1)))
    return e->fail(SIMMR_ENOMEM, "record length allocation failed");
  HIP_TRY(e, hipEventRecord(e->ev_a, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->d_err.p, 0, 64, e->stream));
  e->fq_read_id_base = read_id_base;
  const FqTables tb = fq_tables(e, n_slots);
  // When the emit kernel writes into the text and formats the headers itself it also places its own records: it asks for
  // the first byte of every 64th record only (fq_off64, the scan of the size kernel's per-wave sums), not for fq_off.
  const bool coarse = fq_direct_kernel(e);
  const uint64_t n_w = (n_reads + 63) / 64;
  unsigned long long* tiles = coarse ? nullptr : tile_sums_begin(e, n_reads);
  if (!coarse && !tiles) return e->fail(SIMMR_ENOMEM, "scan scratch allocation failed");
  if (coarse && !e->w_bytes.ensure(std::max<uint64_t>(n_w, 1) * 8)) return e->fail(SIMMR_ENOMEM, "offset allocation failed");
  if (n_reads > 0)
    hipLaunchKernelGGL(k_fastq_size_plan, dim3(grid_for(n_reads, 256)), dim3(256), 0, e->stream, fq_len_coef(e->fq_tpl), tb, fq_plan_view(e),
                       n_reads, coarse ? (uint64_t*)nullptr : e->fq_len.as<uint64_t>(), e->fq_hlen.as<uint8_t>(), e->d_err.as<uint32_t>(), tiles,
                       coarse ? e->w_bytes.as<unsigned long long>() : (unsigned long long*)nullptr);
  uint64_t total = 0;
  if (coarse) { if ((rc = scan_u64(e, e->w_bytes, n_w, e->fq_off64, &total))) return rc; }
  else if ((rc = scan_presummed<uint64_t>(e, e->fq_len, n_reads, 1u, e->fq_off, &total))) return rc;
  e->fq_coarse = coarse;
  HIP_TRY(e, hipEventRecord(e->ev_b, e->stream));
  uint32_t errw2[2] = {0, 0};  // error bits, longest header
  HIP_TRY(e, hipMemcpyAsync(errw2, e->d_err.p, 8, hipMemcpyDeviceToHost, e->stream));
  if ((rc = sync_check(e, "fastq size readback"))) return rc;
  if (errw2[0] & SIMMR_ERRBIT_FASTQ)
    return e->fail(SIMMR_ENOTSUP, "a FASTQ header is longer than %u bytes, or a read names a genome / contig without an id", FQ_HMAX - 1);
  (void)hipEventElapsedTime(&e->last_fastq_plan_ms, e->ev_a, e->ev_b);
  e->fq_hpitch = fq_slot_pitch(errw2[1] + 2u + 8u + 16u);  // the previous record's '\n', header, '\n', slack of the 8-byte id copies
  e->fq_maxhdr = errw2[1];
  e->fq_reads = n_reads;
  e->fq_total = total;
  e->fq_slots = n_slots;
  e->fq_lit_bytes = lit_bytes;
  e->fq_paired = e->plan_paired;
  e->fq_ready = true;
  e->fq_direct = true;
  *total_bytes = total;
  return SIMMR_OK;
}

int simmr_emit_fastq(simmr_engine* e, uint8_t* dst, uint64_t dst_capacity) {
  if (!e) return SIMMR_EINVAL;
  if (!e->fq_ready || !e->fq_direct || e->plan_kind == PLAN_NONE)
    return e->fail(SIMMR_ESTATE, "simmr_emit_fastq called without simmr_fastq_plan_direct on the current plan");
  if (dst_capacity < e->fq_total)
    return e->fail(SIMMR_ERANGE, "dst_capacity %llu < %llu bytes planned", (unsigned long long)dst_capacity,
                   (unsigned long long)e->fq_total);
  const uint64_t n_units = e->plan_units, n_reads = e->fq_reads;
  if (n_reads == 0) return SIMMR_OK;
  if (!dst) return e->fail(SIMMR_EINVAL, "dst is NULL");
  HIP_TRY(e, hipSetDevice(e->device));
  const bool paired = e->plan_paired;
  const bool direct_kernel = fq_direct_kernel(e);
  const FqTables tb = fq_tables(e, e->fq_slots);
  if (!direct_kernel) {
    // No emit kernel of this profile writes into text: the columns are built in buffers of the engine and framed from
    // there (the same kernels as simmr_*_emit + simmr_fastq_emit; qualities with the FASTQ offset, util.rs:46-57).
    const uint64_t tb_bytes = std::max<uint64_t>(e->plan_total_bases, 1), nr = std::max<uint64_t>(n_reads, 1);
    if (!e->fd_seq.ensure(tb_bytes) || !e->fd_qual.ensure(tb_bytes) || !e->fd_seq_off.ensure((nr + 1) * 8) || !e->fd_start.ensure(nr * 8) ||
        !e->fd_end.ensure(nr * 8) || !e->fd_contig.ensure(nr * 4) || !e->fd_genome.ensure(nr * 4) || !e->fd_read_id.ensure(nr * 4) ||
        !e->fd_flags.ensure(nr))
      return e->fail(SIMMR_ENOMEM, "column allocation failed (%llu bases, %llu reads)", (unsigned long long)tb_bytes, (unsigned long long)nr);
    simmr_reads_out cols{};
    cols.seq = e->fd_seq.as<uint8_t>(); cols.qual = e->fd_qual.as<uint8_t>(); cols.seq_off = e->fd_seq_off.as<uint64_t>();
    cols.start = e->fd_start.as<uint64_t>(); cols.end = e->fd_end.as<uint64_t>(); cols.contig = e->fd_contig.as<uint32_t>();
    cols.genome = e->fd_genome.as<uint32_t>(); cols.read_id = e->fd_read_id.as<uint32_t>(); cols.flags = e->fd_flags.as<uint8_t>();
    cols.seq_capacity = tb_bytes; cols.reads_capacity = nr; cols.qual_offset = 33;
    int rc = emit_common(e, e->fq_read_id_base, &cols);
    if (rc) return rc;
    const FqReads rd{cols.seq, cols.qual, cols.seq_off, cols.start, cols.end, cols.contig, cols.genome, cols.read_id, cols.flags, 0u};
    const uint64_t n_batches = (n_reads + FQ_BATCH - 1) / FQ_BATCH;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_batches + 3) / 4, (uint64_t)e->n_cu * 8 * e->fastq_mult);
    const uint32_t hdr_lds = 4 * FQ_BATCH * e->fq_hpitch;
    if (hdr_lds > 48 * 1024)
      HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_fastq_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hdr_lds));
    hipLaunchKernelGGL(k_fastq_write, dim3(grid), dim3(256), hdr_lds, e->stream, e->fq_tpl_dev.as<FqTemplate>(), tb, rd, n_reads, paired ? 1u : 0u,
                       e->fq_lit_bytes, e->fq_hpitch, e->fq_off.as<uint64_t>(), dst);
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "fastq launch failed: %s", hipGetErrorString(s));
    return SIMMR_OK;
  }
  const bool seeds2 = paired && e->prof.kind != SIMMR_K_PERFECT_SHORT;
  PlanArrays pl = plan_arrays(e, seeds2);
  const uint32_t* u_genome = (paired && !e->plan_multi) ? nullptr : e->u_genome.as<uint32_t>();
  unsigned long long* counters = e->d_counters.as<unsigned long long>();
  HIP_TRY(e, next_emit_events(e));
  HIP_TRY(e, hipEventRecord(e->ev_c, e->stream));
  {
    const uint64_t blocks = (n_units + PHILOX_UNITS - 1) / PHILOX_UNITS;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)e->n_cu * e->philox_wgs_per_cu);
    bool exc = false;
    if (paired) exc = e->plan_any_exc;
    else for (const auto& g : e->genomes) exc = exc || (g.staged && g.has_exc);
    const bool cached = paired && !e->plan_multi && e->plan_genome < e->genomes.size() &&
                        e->genomes[e->plan_genome].contigs.size() <= PHILOX_CBASE;
    bool escq = 33u + e->prof.philox_qmax1 <= 127u;
#if defined(SIMMR_NO_ESCQ)
    escq = false;
#endif
    const bool copy_only = e->prof.kind == SIMMR_K_PERFECT_SHORT;  // (perfect-short: bases of the plan, every quality 60)
    // The whole-line form: paired plans whose reads fit its segments, into a buffer its 16-byte chunks are aligned in,
    // with header slots that leave room for two workgroups per CU; everything else takes the item form.
    const uint32_t tl_pitch = tl_slot_pitch(e->fq_maxhdr);
    const uint32_t tl_lds = TL_GROUP * tl_pitch;
    if (e->text_form == 2 && paired && e->plan_short_ok && ((uintptr_t)dst & 15u) == 0 && tl_lds <= 40u * 1024u) {
      auto tk = text_lines_kernel(exc, cached, escq, copy_only);
      const int slot = (exc ? 1 : 0) | (cached ? 2 : 0) | (escq ? 4 : 0) | (copy_only ? 8 : 0);
      if (tl_lds > e->text_lines_lds_set[slot]) {  // (static + dynamic LDS may pass the default limit with long headers)
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(tk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl_lds));
        e->text_lines_lds_set[slot] = tl_lds;
      }
      if (e->tl_debug) {  // (SIMMR_TL_DEBUG, measurement aid: how many workgroups of this launch share a CU)
        int per_cu = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tk, 256, tl_lds);
        fprintf(stderr, "k_emit_text_lines: dynamic LDS %u bytes (slot pitch %u), %d workgroups per CU, grid %u\n", tl_lds, tl_pitch, per_cu, grid);
      }
      const uint32_t t8 = tl_pitch / 8u, t9 = (t8 + 1u) / 2u;
      hipLaunchKernelGGL(tk, dim3(grid), dim3(256), tl_lds, e->stream, e->prof, e->d_genomes.as<GenomeDev>(), e->plan_genome, n_units, pl,
                         e->u_contig.as<uint32_t>(), u_genome, e->u_seed.as<uint64_t>(), dst, 33u, e->plan_first, e->fq_read_id_base,
                         counters, e->fq_hlen.as<uint8_t>(), e->fq_tpl_dev.as<FqTemplate>(), tb, e->fq_lit_bytes, tl_pitch, t9,
                         65536u / t9 + 1u, (const uint64_t*)e->fq_off64.as<uint64_t>());
      HIP_TRY(e, hipEventRecord(e->ev_d, e->stream));
#if defined(TL_DIAG)
      {  // measurement build: time per section, mean per wave, in microseconds (s_memtime: 100 MHz)
        unsigned long long d[16];
        (void)hipStreamSynchronize(e->stream);
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(tl_diag), sizeof d);
        const char* nm[10] = {"prologue", "wait_slots", "format", "wait_format", "segment", "items", "tasks", "extent_fetch", "flush", "tail"};
        fprintf(stderr, "tl_diag (us per wave, %llu waves):", d[10]);
        for (int k = 0; k < 10; k++) fprintf(stderr, " %s=%.1f", nm[k], 0.01 * (double)d[k] / (double)std::max(1ull, d[10]));
        fprintf(stderr, "\n");
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(tl_diag), z, sizeof z);
      }
#endif
      hipError_t s2 = hipGetLastError();
      if (s2 != hipSuccess) return e->fail(SIMMR_ENODEV, "fastq launch failed: %s", hipGetErrorString(s2));
      return SIMMR_OK;
    }
    auto kern = philox_text_kernel(exc, cached, escq, copy_only);
    // windows per run: the power of two that covers the longest run ('\n' + header + '\n'), at most 32 (512 bytes)
    uint32_t wshift = 0;
    while ((16u << wshift) < e->fq_maxhdr + 2u) wshift++;
    const uint32_t slots_lds = std::max<uint32_t>(PHILOX_MAP_ITEMS, FQ_GROUP * e->fq_hpitch);  // header slots; the item map lives there too
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), slots_lds, e->stream, e->prof, paired ? 1u : 0u, e->d_genomes.as<GenomeDev>(),
                       e->plan_genome, n_units, pl, e->u_off.as<uint64_t>(), e->u_contig.as<uint32_t>(), u_genome,
                       e->u_seed.as<uint64_t>(), dst, dst, 33u, e->plan_first, e->fq_read_id_base, OutCols{}, counters,
                       e->fq_hlen.as<uint8_t>(), e->fq_tpl_dev.as<FqTemplate>(), tb, e->fq_lit_bytes, e->fq_hpitch, wshift,
                       (const uint64_t*)e->fq_off64.as<uint64_t>());
  }
  HIP_TRY(e, hipEventRecord(e->ev_d, e->stream));
  hipError_t s = hipGetLastError();
  if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "fastq launch failed: %s", hipGetErrorString(s));
  return SIMMR_OK;
}

// ---- counters / timing ---------------------------------------------------------------
int simmr_counters(simmr_engine* e, uint64_t* dst_device, uint64_t* dst_host) {
  if (!e) return SIMMR_EINVAL;
  HIP_TRY(e, hipSetDevice(e->device));
  // (the counter-mode emit kernel's workgroups add to SIMMR_CNT_SHARDS partial rows behind the counters: fold them in)
  hipLaunchKernelGGL(k_counters_fold, dim3(1), dim3(64), 0, e->stream, e->d_counters.as<unsigned long long>());
  {
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) return e->fail(SIMMR_ENODEV, "counter fold launch failed: %s", hipGetErrorString(s));
  }
  if (dst_device)
    HIP_TRY(e, hipMemcpyAsync(dst_device, e->d_counters.p, 8 * SIMMR_N_COUNTERS, hipMemcpyDeviceToDevice, e->stream));
  if (dst_host) {
    HIP_TRY(e, hipMemcpyAsync(dst_host, e->d_counters.p, 8 * SIMMR_N_COUNTERS, hipMemcpyDeviceToHost, e->stream));
    return sync_check(e, "counter readback");
  }
  return SIMMR_OK;
}

int simmr_counters_reset(simmr_engine* e) {
  if (!e) return SIMMR_EINVAL;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemsetAsync(e->d_counters.p, 0, 8 * SIMMR_N_COUNTERS * (1 + SIMMR_CNT_SHARDS), e->stream));
  return SIMMR_OK;
}

int simmr_last_emit_kernel_ms(simmr_engine* e, float* ms) {
  if (!e || !ms) return SIMMR_EINVAL;
  int rc = sync_check(e, "emit");
  if (rc) return rc;
  HIP_TRY(e, hipEventElapsedTime(&e->last_emit_ms, e->ev_c, e->ev_d));
  *ms = e->last_emit_ms;
  return SIMMR_OK;
}

int simmr_emit_kernel_ms_mean(simmr_engine* e, uint32_t last_n, float* ms) {
  if (!e || !ms || last_n == 0) return SIMMR_EINVAL;
  int rc = sync_check(e, "emit");
  if (rc) return rc;
  const uint64_t have = std::min<uint64_t>(e->n_emits, (uint64_t)simmr_engine::EMIT_RING);
  const uint64_t n = std::min<uint64_t>(last_n, have);
  if (n == 0) return e->fail(SIMMR_ESTATE, "no emit yet");
  // (an emit that failed between taking its slot and recording its end event leaves a pair that cannot be read: such a
  // slot is skipped, not an error of this call)
  double sum = 0.0;
  uint64_t got = 0;
  for (uint64_t k = 0; k < n; k++) {
    const int i = (int)((e->n_emits - 1 - k) % simmr_engine::EMIT_RING);
    float t = 0.f;
    if (!e->ring_c[i] || !e->ring_d[i] || hipEventElapsedTime(&t, e->ring_c[i], e->ring_d[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
    sum += t;
    got++;
  }
  if (got == 0) return e->fail(SIMMR_ESTATE, "no completed emit among the last %llu", (unsigned long long)n);
  *ms = (float)(sum / (double)got);
  return SIMMR_OK;
}

int simmr_last_fastq_plan_ms(simmr_engine* e, float* ms) {
  if (!e || !ms) return SIMMR_EINVAL;
  *ms = e->last_fastq_plan_ms;
  return SIMMR_OK;
}

int simmr_last_plan_ms(simmr_engine* e, float* ms) {
  if (!e || !ms) return SIMMR_EINVAL;
  *ms = e->last_plan_ms;
  return SIMMR_OK;
}

}  // extern "C"
