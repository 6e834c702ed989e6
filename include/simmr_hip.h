/*
 * simmr_hip.h — C ABI of the MI355X-native simmr hot path (libsimmr_hip.so).
 *
 * This is the drop-in boundary for the per-read sampling / mutation path of
 * genomicsoup/simmr.  The reference has no FFI: the seam is the call made once
 * per run from simmr/src/main.rs:180-186 into
 *     simulate::simulate_pe_reads      (simmr/src/simulate.rs:110-150)
 *     simulate::simulate_long_reads    (simmr/src/simulate.rs:323-406)
 * parameterised by the two trait objects
 *     ErrorProfile      (simmr/src/error_profiles/base.rs:6-32)
 *     AbundanceProfile  (simmr/src/abundance_profiles/base.rs:10-69).
 * Every entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no exceptions / unwinding across the boundary;
 *   - every function returns 0 (SIMMR_OK) or a negative errno-style code and
 *     simmr_last_error() gives the message (reference: Result<_, String>,
 *     simulate.rs:165-170,205-210);
 *   - one engine == one GPU == one host thread (not thread-safe), matching
 *     the single-threaded reference;
 *   - output buffers are caller-owned DEVICE pointers (hipMalloc / torch);
 *     the plan step reports the sizes needed (the reference returns owned
 *     Vec<SimulatedRead>, simulate.rs:119).
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with SIMMR_ENODEV.
 */
#ifndef SIMMR_HIP_H
#define SIMMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMMR_ABI_VERSION 1

/* status codes (negative errno values) */
#define SIMMR_OK 0
#define SIMMR_EINVAL (-22)  /* bad argument                                   */
#define SIMMR_ENOMEM (-12)  /* device/host allocation failed                  */
#define SIMMR_ENODEV (-19)  /* no usable gfx950 device / HIP runtime error    */
#define SIMMR_ERANGE (-34)  /* output capacity too small / value out of range */
#define SIMMR_ESTATE (-1)   /* call order violated (emit before plan, ...)    */
#define SIMMR_ENOTSUP (-95) /* valid request this library leaves to the host
                               (simmr_fastq_plan: see there)                  */
#define SIMMR_EGENOME (-61) /* genome unusable (reference: Err(String) at
                               simulate.rs:220-225 / infinite loop at :370)   */

/* ErrorProfile implementations, reference cli.rs:62-70 + error_profiles/mod.rs */
enum simmr_profile_kind {
  SIMMR_PERFECT_SHORT = 0, /* error_profiles/perfect_short.rs */
  SIMMR_MINIMAL_SHORT = 1, /* error_profiles/minimal_short.rs */
  SIMMR_PERFECT_LONG = 2,  /* error_profiles/perfect_long.rs  */
  SIMMR_MINIMAL_LONG = 3,  /* error_profiles/minimal_long.rs  */
  SIMMR_CUSTOM = 4         /* error_profiles/custom_short.rs  */
};

/* Which generator feeds the per-base draws.
 * REFERENCE: the reference's own streams — rand 0.8.5 StdRng (ChaCha12 keyed
 *   by PCG32 seed expansion) consumed exactly as simulate.rs / the profiles
 *   consume them.  Output is bit-identical to the reference for everything the
 *   reference itself makes deterministic under --seed.
 * PHILOX: counter-based Philox4x32-10 (Random123 constants) keyed by the read's Phred seed (key = its low and high
 *   words).  Specification, version 3 — stated in DESIGN.md section 4, restated independently in oracle/philox.c, which
 *   the kernels are compared with bit for bit:
 *     a base draws its Phred score q and its substitution s (0 = none, 1..3 = "ACGT"[(code + s) & 3]) TOGETHER from
 *     their joint law over the 1024 outcomes o = q | s << 8;
 *     level 1: 24 bits per base.  The 16 bases of group g = b >> 4 share the 384 bits of the three calls with counters
 *       (3g + {0, 1, 2}, 0, 0x73696D6D, 0x72000003) ('simm', 'r', version 3); base b takes bits [24 (b & 15), +24) = F;
 *       column F >> 14 of a 1024-column alias table (integer Vose construction over the 2^24 cells), fraction
 *       F & 0x3fff against the column's threshold in 16384ths;
 *     level 2, only for the E of 2^24 cells (E = 118 at mean Phred 30) that the integer split leaves over: counter
 *       (b >> 2, 1, 0x73696D6D, 0x72000003), word b & 3, a 1024-column alias table over the residual law with 22-bit
 *       thresholds.
 *   In rocRAND's terms (the library north_star names; tests/test_oracle_kat.py runs rocRAND's own engine class against
 *   the specification): a read's draws are the stream of rocrand_state_philox4x32_10 after
 *   rocrand_init(seed = the read's Phred seed, subsequence = 0x7200000373696D6D, offset = 4 * (c0 | c1 << 32), &state)
 *   — level 1 reads it from offset 4 * 3g (twelve consecutive outputs per group), level 2 at offset
 *   4 * ((b >> 2) | 1 << 32).  The kernels compute the same words with a hand-written round (two v_mad_u64_u32 and two
 *   v_bitop3_b32) rather than through the library's state object.
 *   Every profile with per-base draws: minimal-short, minimal-long, perfect-long as above, and the long-read path of a
 *   custom model (SIMMR_CUSTOM with an is_long model), whose base-by-base draws are those of the k-mer splice
 *   (simulate_errors, custom_short.rs:455-516; its qualities use a handful of words per read and stay the reference's):
 *     (specification of the splice's draws, version 2 — version 1 took two words per position and was never released)
 *     the alternate of the k-mer visited at position i is drawn from ONE word, X = word i & 3 of the block with
 *     key = the read's seed and counter (i >> 2, 2, 0x73696D6D, 0x72000003), in two levels over the reference's law
 *     P(alternate j) = w_j / sum(w):
 *       level 1, X >> 8 < T24 -> the k-mer stays what it is, with T24 = 2^24 - 2^e, 2^e the smallest power of two
 *         (1 <= e <= 24) of 2^24ths that holds 1 - P(self); a k-mer with an N has no "self": e = 24, T24 = 0;
 *       level 2 otherwise: Z = (X - (T24 << 8)) << (24 - e) is a full word again, m = Z * n (64 bits), column
 *         c = m >> 32 of an n-column alias table (Vose, f64, sums in list order) with thresholds in 2^24ths against
 *         (m & 0xffffffff) >> 8, over the residual law r_j = (p_j - [j is self] (T24 / 2^24) p_j / p_s) / (1 - T24 / 2^24).
 *     (oracle/custom.c: ctr_splice_tables / orc_custom_simulate_errors_philox; the paired-end path of a custom model
 *     has no base-by-base draws and refuses the mode.)
 *   Positions, lengths and seeds still come from the reference's streams.  Statistical tolerance only (BASELINE.json
 *   north_star): the law is the reference's (minimal_short.rs:83-140), the bits are not.
 * PHILOX_FULL: PHILOX, and the draws of the PLAN from Philox counters as well — what north_star describes ("random
 *   start-position draw, length draw, per-base draws: counter-based"), and what makes a shard's plan a function of its
 *   pair indices alone (no stream to walk or to seek in: the plan path costs a quarter).  Specification, version 1:
 *     every generator the reference seeds on the way to a read — StdRng::seed_from_u64(s) for the lengths
 *     (minimal_short.rs:33-67), the start position and the two mate-2 seeds (simulate.rs:227-270), a long read's own
 *     generator — is the word stream W(s): word w = word w & 3 of the Philox4x32-10 block with key = s and counter
 *     (w >> 2, 3, 0x73696D6D, 0x72000003), consumed in the reference's order by the reference's algorithms (gen_range with
 *     its rejection zone, the ziggurat, Gamma, Option<u64>);
 *     the genome's outer stream (simulate.rs:172-186, one generator walked pair by pair) becomes one block per pair: pair p
 *     of the genome's run (p < 2^56) takes the block with key = the run's seed and counter (p & 0xffffffff, 4 | (p >> 32) << 8,
 *     0x73696D6D, 0x72000003) = (w0, w1, w2, w3): contig = ((w0 | w1 << 32) * num_seqs) >> 64, pe_seed = w2 | w3 << 32
 *     (as in the reference, every genome of a run sees the same seed);
 *     the per-base draws are PHILOX's, keyed by the seeds this plan makes.
 *   Minimal-short, and minimal-long / perfect-long with SIMMR_LEN_PER_READ (the one constant length of a seeded reference
 *   run, simulate.rs:358, is a property of its stream); not the custom profiles.  Restated in oracle/ (rand08.c: the
 *   generator's second word source; simulate.c: orc_pe_outer_ctr), compared bit for bit; law tests in
 *   tests/test_gpu_parity.py::test_philox_full_*. */
enum simmr_rng_mode { SIMMR_RNG_REFERENCE = 0, SIMMR_RNG_PHILOX = 1, SIMMR_RNG_PHILOX_FULL = 2 };

/* Long-read length policy (Appendix A Q5 of SURVEY.md).
 * REFERENCE: with a seed, get_random_read_length(seed) (simulate.rs:358) is
 *   the same value for every read of the run; reproduced exactly.
 * PER_READ: every read draws its own Gamma length, contig and read seed from
 *   a private StdRng keyed by mix(seed, read index).  This is what the
 *   reference does without --seed (fresh entropy per call), made reproducible.
 *   Selected automatically when has_seed == 0. */
enum simmr_length_mode { SIMMR_LEN_REFERENCE = 0, SIMMR_LEN_PER_READ = 1 };

/* Where a long read starts on its sequence (Appendix A Q6 of SURVEY.md).
 * REFERENCE: simulate.rs:484 draws read_start in [0, read_length) — not in
 *   [0, size - read_length) — so every long read starts within the first
 *   read_length bases of its sequence; reproduced exactly.
 * UNIFORM: read_start = gen_range(0..size - read_length) from the same
 *   StdRng(read_seed), read_end = read_start + read_length (no re-draw is needed):
 *   what simulate_long_read evidently means to do.  An extension; off by default. */
enum simmr_long_start_mode { SIMMR_START_REFERENCE = 0, SIMMR_START_UNIFORM = 1 };

/* Flattened ErrorProfile (trait: error_profiles/base.rs:6-32; construction:
 * cli.rs:229-301).  Plain data, copied by the callee. */
typedef struct simmr_error_profile {
  uint32_t kind;        /* enum simmr_profile_kind                           */
  uint32_t rng_mode;    /* enum simmr_rng_mode                               */
  uint32_t length_mode; /* enum simmr_length_mode (long reads only)          */
  uint16_t read_length; /* --read-length   (cli.rs:126)                      */
  uint16_t insert_size; /* --insert-size   (cli.rs:143)                      */
  uint8_t mean_phred;   /* --mean-phred-score (cli.rs:152)                   */
  uint8_t long_start_mode; /* enum simmr_long_start_mode (long reads only) */
  uint8_t reserved0[2];
  double read_length_std; /* minimal-short: 15.0 (cli.rs:240)               */
  double insert_size_std; /* minimal-short: 75.0 (cli.rs:239)               */
  float gamma_shape;      /* (mean/std)^2, minimal_long.rs:68                */
  float gamma_scale;      /* std^2/mean,   minimal_long.rs:69                */
  const void* custom_model;    /* bincode ErrorModelParams (shared/encoding.rs:102-117) */
  uint64_t custom_model_bytes;
} simmr_error_profile;

/* Contiguous range of global unit indices (pairs for PE, reads for long) that
 * this engine / GPU generates.  Replaces nothing in the reference (it has no
 * sharding); ids stay identical to the single-process run. */
typedef struct simmr_range {
  uint64_t first;
  uint64_t count;
} simmr_range;

/* Result of a plan step: what the emit step will write. */
typedef struct simmr_plan_info {
  uint64_t n_units;      /* pairs (PE) or reads (long) planned in this shard  */
  uint64_t n_reads;      /* 2*n_units for PE, n_units for long                */
  uint64_t total_bases;  /* bytes needed in seq[] and in qual[]               */
  uint64_t seed_used;    /* the seed (given, or drawn from OS entropy)        */
  uint64_t outer_slots;  /* u64 draws consumed from the outer StdRng stream (0 with SIMMR_RNG_PHILOX_FULL: there is none) */
  uint32_t const_read_length; /* long/REFERENCE: the run-wide length, else 0  */
  uint32_t slot_bytes;   /* 0: compact streams; 16: SIMMR_SLOT16 (simmr_engine_set_read_slots) */
} simmr_plan_info;

/* flags[] bits */
#define SIMMR_FLAG_REVCOMP 0x01u     /* ReadMetadata.is_reverse_complement    */
#define SIMMR_FLAG_QSEED_SUBST 0x02u /* mate-2 Phred seed drew None (simulate.rs:266):
                                        the reference uses OS entropy there; we
                                        substitute simmr_entropy_substitute() */
#define SIMMR_FLAG_MSEED_SUBST 0x04u /* same for the mutation seed (simulate.rs:270) */
#define SIMMR_FLAG_REDRAWN 0x08u     /* mate-2 window was re-drawn (simulate.rs:241-247) */

/* SoA replacement of Vec<SimulatedRead> (simulate.rs:27-75).  All pointers are
 * DEVICE pointers owned by the caller.  Read r of a PE shard is mate (r & 1)
 * of pair (r >> 1); mates are interleaved exactly as fastq.rs:32-121 writes
 * them.  Any pointer except seq/qual/seq_off may be NULL to skip that column. */
typedef struct simmr_reads_out {
  uint8_t* seq;       /* ASCII bases, total_bases bytes (no slack needed: every emit kernel bounds its last store;
                         tests/test_gpu_shapes.py runs each of them between canaries at exactly this size) */
  uint8_t* qual;      /* Phred + qual_offset, total_bases bytes               */
  uint64_t* seq_off;  /* n_reads + 1 CSR offsets into seq / qual              */
  uint64_t* start;    /* ReadMetadata.start_pos (mate 2: the larger bound)    */
  uint64_t* end;      /* ReadMetadata.end_pos                                 */
  uint32_t* contig;   /* index of the source Seq inside its genome            */
  uint32_t* genome;   /* staged genome index (long reads span genomes)        */
  uint32_t* read_id;  /* SimulatedRead.id (simulate.rs:85-89), shared by mates */
  uint8_t* flags;     /* SIMMR_FLAG_*                                         */
  uint64_t seq_capacity;   /* bytes available in seq and in qual              */
  uint64_t reads_capacity; /* entries available in the per-read columns       */
  uint32_t qual_offset;    /* 0: raw Phred as SingleRead.quality; 33: FASTQ   */
  uint32_t slot_bytes;     /* the layout the caller expects: 0 (or 1) compact, 16 SIMMR_SLOT16; an emit call whose
                              plan was made for the other layout answers SIMMR_EINVAL                        */
} simmr_reads_out;

/* Layout of seq[] / qual[] (the reference's SingleRead is a heap Vec per read, simulate.rs:27-40: neither layout
 * below is "the" reference layout).
 * compact (default): read r's bases are seq[seq_off[r] .. seq_off[r+1]) and its qualities the same bytes of qual[]:
 *   two byte streams without gaps.
 * SIMMR_SLOT16 (simmr_engine_set_read_slots(e, 16) before the plan call; what INTEGRATION.md's call sequence, the Python
 *   host and simmr-hip select): every read owns a slot of ceil(L / 16) * 16 bytes that starts on a 16-byte boundary of
 *   both streams (total_bases counts the slots), so that the counter-mode emit kernel writes nothing but whole aligned
 *   16-byte groups — no byte-granular stores at the reads' ends.  Qualities are left-aligned in the slot.
 *   Bases are left-aligned too, except for a reverse-complemented mate (SIMMR_FLAG_REVCOMP), whose bases are written
 *   back to front and therefore RIGHT-aligned: its 16-base draw groups then land on aligned 16 bytes as well.
 *     seq_off[r]         = first base of read r in seq[]   (for a reverse-complemented mate not a multiple of 16)
 *     seq_off[r] & ~15   = first quality of read r in qual[]
 *     L(r)               = |end[r] - start[r]|   (seq_off[r+1] - seq_off[r] is NOT the length in this layout)
 *     seq_off[n_reads]   = total_bases
 *   Padding bytes are written as 0 in both streams.  The setting is a preference: the kernels that gain from slots write
 *   them — SIMMR_RNG_PHILOX with the minimal-short, minimal-long and perfect-long profiles — and a plan for any other
 *   profile is made for the compact layout.  simmr_plan_info.slot_bytes says which one a plan got; size the buffers
 *   from total_bases of the same info and pass that slot_bytes on in simmr_reads_out.  simmr_fastq_plan /
 *   simmr_fastq_emit read either layout. */
#define SIMMR_SLOT16 16u

/* Device-side counters a run accumulates (reduced across GPUs by the caller
 * with one all-reduce).  Indices into the uint64 array. */
enum simmr_counter {
  SIMMR_CNT_READS = 0,
  SIMMR_CNT_BASES = 1,
  SIMMR_CNT_ACGT_BASES = 2,
  SIMMR_CNT_SUBSTITUTIONS = 3,
  SIMMR_CNT_OUTER_REJECTS = 4,
  SIMMR_CNT_REDRAWN = 5,
  SIMMR_CNT_SEED_SUBST = 6,
  SIMMR_CNT_QUAL_SUM = 7,
  SIMMR_N_COUNTERS = 8
};

typedef struct simmr_engine simmr_engine; /* opaque */

/* ---- lifetime ---------------------------------------------------------- */
int simmr_abi_version(void);
/* Binds to HIP device `device_ordinal`; fails with SIMMR_ENODEV if there is
 * none or it is not gfx950. */
int simmr_engine_create(int device_ordinal, simmr_engine** out);
void simmr_engine_destroy(simmr_engine* e);
/* Last error text of this engine (or of engine creation when e == NULL). */
const char* simmr_last_error(const simmr_engine* e);
/* All work is enqueued on this hipStream_t (default: the null stream). */
int simmr_engine_set_stream(simmr_engine* e, void* hip_stream);
/* Layout of the reads that plans made FROM NOW ON will emit: 0 (or 1) = compact, SIMMR_SLOT16 = 16-byte read slots
 * wherever the plan's emit kernel writes them (see simmr_reads_out; simmr_plan_info.slot_bytes reports what a plan
 * got).  Anything else: SIMMR_EINVAL.  The plan in force keeps the layout it was made with. */
int simmr_engine_set_read_slots(simmr_engine* e, uint32_t slot_bytes);

/* The plan of the next shard beside the emit of this one.  A run that is generated shard by shard (the reference keeps a
 * whole run in RAM, main.rs:180-206; here `for range: plan, emit, drain`) calls plan k + 1 while the emit of shard k is
 * still on the device; with on != 0 the plan calls run on a stream of the engine's own and write a second set of the
 * buffers an emit reads, so the two overlap — the plan kernels are bound by latency, the emit kernels by instruction
 * issue — instead of queueing behind one another.  The caller's stream (simmr_engine_set_stream) is made to wait for the
 * plan before the call returns, so everything the caller enqueues afterwards sees it: no call sequence changes.  Costs a
 * second set of plan columns (25 bytes per pair / long read).  Off by default; switching synchronises the device. */
int simmr_engine_set_plan_overlap(simmr_engine* e, int on);

/* ---- reference staging -------------------------------------------------- */
/* Replaces the in-RAM `Genome { sequence: Vec<Seq> }` (genome.rs:17-41): the
 * normalised ASCII contigs are packed once into HBM as a flat 2-bit array
 * (A0 C1 G2 T3; base i in bits 2(i mod 16) of word i/16 — the code points of
 * shared/src/encoding.rs:146-152) plus a 1-bit exception plane for 'N' / '-'.
 * contig_len[i] = bytes in contig_ascii[i] (Seq.seq.len()),
 * contig_size[i] = Seq.size (differs only under --contiguous, genome.rs:127);
 * NULL means size == len. */
int simmr_stage_genome(simmr_engine* e, uint32_t genome_idx, uint32_t n_contigs,
                       const uint8_t* const* contig_ascii, const uint64_t* contig_len,
                       const uint64_t* contig_size);
/* Genome::from_fasta's sequence handling on the device (genome.rs:93-137): the
 * host splits the file into records (header lines, body ranges); the device
 * applies needletail's normalize(false) (genome.rs:114: whitespace and line ends
 * dropped, acgt -> ACGT, u/U -> T, . ~ -> -, A C G T N - kept, everything else ->
 * N) and packs in the same pass, so no normalised copy is ever built on the host.
 * body[c] / body_len[c]: the raw bytes between record c's header line and the
 * next header (HOST memory).  base_count[c] receives the record's number of bases.
 * contiguous == 0: records with more than min_size bases become the genome's
 *   sequences, in order (the size filter of main.rs:117-162; pass 0 to keep all);
 *   *n_staged = how many.  With none left the slot is left unstaged.
 * contiguous != 0: one sequence, every record followed by an 'N' (genome.rs:
 *   121-137); its Seq.size counts the bases without the separators. */
int simmr_stage_fasta(simmr_engine* e, uint32_t genome_idx, uint32_t n_records,
                      const uint8_t* const* body, const uint64_t* body_len, int contiguous,
                      uint64_t min_size, uint64_t* base_count, uint32_t* n_staged);
/* Synthetic genome generated on the device: word k of the packed array (32
 * bases) is SplitMix64 output k of `splitmix_seed` (BASELINE.md / SURVEY §8d). */
int simmr_stage_synthetic(simmr_engine* e, uint32_t genome_idx, uint32_t n_contigs,
                          const uint64_t* contig_len, uint64_t splitmix_seed);
/* Copies staged bases back as ASCII (debug / tests). dst is a HOST pointer. */
int simmr_unstage_contig(simmr_engine* e, uint32_t genome_idx, uint32_t contig, uint64_t first,
                         uint64_t count, uint8_t* dst_host);
int simmr_genome_info(const simmr_engine* e, uint32_t genome_idx, uint32_t* n_contigs,
                      uint64_t* total_size);

/* ---- paired-end path ----------------------------------------------------- */
/* simulate_pe_reads_from_genome (simulate.rs:165-190) for one genome:
 * `genome_reads` is the reference's num_reads (mates; num_reads/2 pairs,
 * simulate.rs:179).  Plans pairs [shard.first, shard.first+shard.count) of
 * that genome: outer StdRng stream (contig draw + pe_seed), then per pair read
 * length / insert size / window (simulate_pe_read, simulate.rs:205-258).
 * shard.count == UINT64_MAX means "to the end". */
int simmr_pe_plan(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile,
                  uint64_t genome_reads, int has_seed, uint64_t seed, simmr_range shard,
                  simmr_plan_info* info);
/* Seeking in a genome's outer stream, so that N GPUs sharing one run do not each
 * re-walk the stream from slot 0 to their shard (simulate.rs:172-184 draws, per
 * pair, a contig index by rejection sampling and then pe_seed: the slot where
 * pair p starts depends on every earlier rejection).  The loop is a two-state
 * machine over u64 slots (0: about to draw a contig index, 1: about to draw
 * pe_seed); simmr_outer_summarize walks slots [slot_first, slot_first +
 * slot_count) (both multiples of 8) once and returns, for either state at
 * slot_first, how many pairs complete inside the range and the state after it.
 * Ranks summarize disjoint ranges, exchange the 4 numbers (the path's only other
 * collective, 32 bytes per rank) and compose them; simmr_pe_plan_at then starts
 * at a known position: pair `start_unit` begins at slot `start_slot`
 * (start_unit <= shard.first; start_slot = start_unit = 0 is simmr_pe_plan).  With SIMMR_RNG_PHILOX_FULL there is no stream:
 * the two positions are ignored and the call is simmr_pe_plan. */
typedef struct simmr_outer_summary {
  uint64_t units[2];      /* pairs completed in the range, by state at slot_first */
  uint32_t end_state[2];  /* state after the range */
} simmr_outer_summary;
int simmr_outer_summarize(simmr_engine* e, uint32_t genome_idx, uint64_t seed, uint64_t slot_first,
                          uint64_t slot_count, simmr_outer_summary* out);
int simmr_pe_plan_at(simmr_engine* e, uint32_t genome_idx, const simmr_error_profile* profile,
                     uint64_t genome_reads, uint64_t seed, simmr_range shard, uint64_t start_slot,
                     uint64_t start_unit, simmr_plan_info* info);
/* simulate_pe_reads (simulate.rs:110-150) over several genomes in ONE plan.  The
 * reference loops over the genomes and re-creates the outer StdRng with the same
 * seed for each (simulate.rs:137,172), so genomes with the same number of
 * sequences draw the same (contig, pe_seed) list: it is generated once per
 * distinct count and shared.  genome_reads[g] are the per-genome read counts of
 * the abundance profile; shard is a range of the global pair index (genomes
 * concatenated in the given order — the order in which the reference's global id
 * counter, simulate.rs:85-89, numbers the pairs).  simmr_pe_emit then emits the
 * shard with read_id_base = 0 and fills the `genome` column per read.
 * SIMMR_ENOTSUP for custom profiles (plan those one genome at a time). */
int simmr_pe_plan_multi(simmr_engine* e, uint32_t n_genomes, const uint32_t* genome_idx,
                        const uint64_t* genome_reads, const simmr_error_profile* profile, int has_seed,
                        uint64_t seed, simmr_range shard, simmr_plan_info* info);
/* Emits the planned pairs: bases, qualities, mutations, reverse complement,
 * metadata (simulate.rs:260-299).  read_id_base = id of pair 0 of this genome
 * (the reference's global AtomicU32, simulate.rs:85-89). */
int simmr_pe_emit(simmr_engine* e, uint32_t read_id_base, const simmr_reads_out* out);

/* ---- long-read path ------------------------------------------------------ */
/* simulate_long_reads (simulate.rs:323-406) over all genomes at once (ONE
 * StdRng stream spans every genome, simulate.rs:348-351).  genome_reads[g] is
 * the per-genome read count from the abundance profile.  shard is a range of
 * global read indices (generation order across genomes).
 * Profiles: minimal-long, perfect-long, or SIMMR_CUSTOM with a model whose
 * is_long flag is set (custom_short.rs:540-542): then the run-wide length is
 * floor(Normal(read_length_mean, read_length_std)) (custom_short.rs:286-301),
 * qualities come from the per-position PDFs (:332-353) and simmr_long_emit
 * applies simulate_errors, the k-mer splice (:455-516); with
 * SIMMR_LEN_PER_READ every read draws its own length from that Normal law.
 * A custom model needs kmer_size <= 10 (else SIMMR_ENOTSUP); an alternate
 * k-mer that deletes bases makes the reference panic and simmr_long_emit
 * return SIMMR_ERANGE.  custom_model is read during the plan call only; the
 * engine keeps the device tables of the last model it was given (keyed by the
 * model's bytes), so planning many shards with one model builds them once. */
int simmr_long_plan(simmr_engine* e, uint32_t n_genomes, const uint32_t* genome_idx,
                    const uint64_t* genome_reads, const simmr_error_profile* profile, int has_seed,
                    uint64_t seed, simmr_range shard, simmr_plan_info* info);
int simmr_long_emit(simmr_engine* e, uint32_t read_id_base, const simmr_reads_out* out);

/* ---- counters across GPUs ---------------------------------------------------
 * The path has one exchange step: the sum of the run counters over the GPUs of a
 * run.  One engine per process and device; rank 0 makes an id and hands it to the other ranks by
 * whatever channel the host has; every rank then joins.  RCCL is loaded at run time
 * (librccl.so.1), so a single-GPU user needs none.  Replaces nothing in the reference (it is a
 * single process); the counters are what a maintainer would log next to the metadata TSV. */
#define SIMMR_COMM_ID_BYTES 128
int simmr_comm_unique_id(uint8_t* id128);                                   /* ncclGetUniqueId */
int simmr_comm_init(simmr_engine* e, const uint8_t* id128, int rank, int world);  /* ncclCommInitRank on the engine's device */
/* In-place sum over the ranks of n u64 values in device memory (ncclAllReduce on the engine's
 * stream; asynchronous like the rest).  Without a communicator (one GPU) it does nothing. */
int simmr_allreduce_counts(simmr_engine* e, uint64_t* counts_device, uint32_t n);

/* ---- counters / timing ---------------------------------------------------- */
/* Copies the SIMMR_N_COUNTERS running counters to a DEVICE array (for the
 * caller's all-reduce) and/or a HOST array; either may be NULL.  The device
 * copy is enqueued on the engine's stream like everything else (a collective
 * launched from another stream has to be ordered after it by the caller; with
 * the default null stream, as torch.distributed uses it, that is implicit);
 * the host copy is complete on return. */
int simmr_counters(simmr_engine* e, uint64_t* dst_device, uint64_t* dst_host);
int simmr_counters_reset(simmr_engine* e);
/* HIP-event time (ms) of the dominant emit kernel of the last *_emit call,
 * measured on the engine's stream. Synchronises the stream. */
int simmr_last_emit_kernel_ms(simmr_engine* e, float* ms);
/* The mean of the last `last_n` emits' times (at most 64 are kept), after ONE synchronisation of the engine's stream —
 * for a loop that must not wait for every emit (simmr_engine_set_plan_overlap: asking after each emit would serialise
 * the plan of the next shard behind it). */
int simmr_emit_kernel_ms_mean(simmr_engine* e, uint32_t last_n, float* ms);
/* HIP-event time (ms) of the last *_plan call's device work. */
int simmr_last_plan_ms(simmr_engine* e, float* ms);

/* ---- documented substitutions --------------------------------------------- */
/* Where the reference re-seeds from OS entropy in the middle of a seeded run
 * (Option<u64>::None at simulate.rs:266,270) we substitute this pure function
 * of (pe_seed, which) so the run stays reproducible; which = 1 for the Phred
 * seed, 2 for the mutation seed.  Also used to derive per-read seeds in
 * SIMMR_LEN_PER_READ mode (which = 3, x = seed ^ read index mix). */
uint64_t simmr_entropy_substitute(uint64_t x, uint32_t which);

/* ---- FASTQ framing on the device: replaces fastq::write_to_fastq
 * (simmr/src/fastq.rs:14-124) up to the file write.  The record of read r is
 *     header '\n' bases '\n' '+' '\n' qualities '\n'      (fastq.rs:58-66, 93-103)
 * with the header built from `header_format` by the reference's chain of
 * String::replace calls (fastq.rs:34-56, 69-91): {:genome_id:} {:read_id:}
 * {:sequence_id:} {:start_position:} {:end_position:} {:reverse_complement:}
 * (t / f) {:pair:} (1 / 2).  Qualities are copied as they are: emit them with
 * qual_offset = 33 (util::encode_quality_scores, util.rs:46-57).
 *
 * names: the text of {:genome_id:} per genome (Genome.uuid, main.rs:73-75) and of
 * {:sequence_id:} per contig (Seq.id, genome.rs:100-112), for every engine genome
 * slot the reads' `genome` column can hold.
 *
 * simmr_fastq_plan sizes every record (device), scans the sizes and returns the
 * total; simmr_fastq_emit writes the n_reads records back to back into dst
 * (device memory, >= total bytes).  `reads` must carry every column.
 *
 * SIMMR_ENOTSUP (nothing written; use the host writer): a genome or sequence id
 * containing '{' or '}' (the chained replace could then re-expand it), a template
 * whose own text has a '{' somewhere before and a '}' somewhere after a
 * {:genome_id:} or {:sequence_id:} field (the inserted id could complete a
 * placeholder with them), more than 24 template pieces, or a header longer than
 * 255 bytes.  SIMMR_EINVAL: a genome_idx entry that is not a staged slot. */
typedef struct simmr_fastq_names {
  uint32_t n_genomes;
  const uint32_t* genome_idx;      /* engine genome slot of each entry */
  const char* const* genome_id;    /* NUL-terminated */
  const uint32_t* n_contigs;       /* contigs of each entry, as staged */
  const char* const* sequence_id;  /* flattened entry by entry, NUL-terminated */
} simmr_fastq_names;

int simmr_fastq_plan(simmr_engine* e, const char* header_format, const simmr_fastq_names* names,
                     const simmr_reads_out* reads, uint64_t n_reads, int paired, uint64_t* total_bytes);
int simmr_fastq_emit(simmr_engine* e, const simmr_reads_out* reads, uint8_t* dst, uint64_t dst_capacity);

/* The same text without the columns in between: what main.rs:180-206 does as a whole — simulate, then
 * fastq::write_to_fastq (fastq.rs:14-124) — for the shard of the CURRENT plan (simmr_pe_plan / simmr_pe_plan_at /
 * simmr_pe_plan_multi / simmr_long_plan).  A header's length depends on plan columns only (decimal widths of read id,
 * start and end; the lengths of the ids), so
 *   simmr_fastq_plan_direct sizes every record from the plan and returns the total, and
 *   simmr_emit_fastq writes the shard's records back to back into dst (device memory, >= total bytes): the emit kernel
 *     stores bases and qualities (offset 33, util.rs:46-57) at their places in the text and formats the headers and
 *     line ends of its block's records itself.  The run counters advance as simmr_*_emit advances them (for a
 *     perfect-short run on a genome with N / '-' bases SIMMR_CNT_ACGT_BASES is counted here, which simmr_pe_emit
 *     leaves at 0).
 * The bytes are those of simmr_*_emit (qual_offset 33) + simmr_fastq_plan + simmr_fastq_emit with the same arguments,
 * and the same cases are refused with SIMMR_ENOTSUP.  Served by a kernel that writes into the text: the counter mode
 * (SIMMR_RNG_PHILOX) of the minimal / perfect-long profiles, and perfect-short (no draws: the planned bases and a
 * constant quality line).  Profiles whose emit kernel cannot write into text (SIMMR_RNG_REFERENCE, custom models) are
 * served through columns held by the engine: same result, no saving. */
int simmr_fastq_plan_direct(simmr_engine* e, const char* header_format, const simmr_fastq_names* names,
                            uint32_t read_id_base, uint64_t* total_bytes);
int simmr_emit_fastq(simmr_engine* e, uint8_t* dst, uint64_t dst_capacity);
/* HIP-event time (ms) of the last simmr_fastq_plan_direct's device work (size pass + scan). */
int simmr_last_fastq_plan_ms(simmr_engine* e, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SIMMR_HIP_H */
