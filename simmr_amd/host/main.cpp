// main.cpp — `simmr-hip`: the reference's run_main (simmr/src/main.rs:20-268)
// over the C ABI of libsimmr_hip.so.  Same flags, same output files:
// interleaved FASTQ (fastq.rs) and "<output>.tsv" metadata (files.rs:100-134).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <algorithm>
#include <string>
#include <vector>

#include "simmr_host.hpp"

using namespace simmr_host;

static void info(const char* msg) { fprintf(stderr, " INFO simmr-hip: %s\n", msg); }
static void warn(const std::string& msg) { fprintf(stderr, " WARN simmr-hip: %s\n", msg.c_str()); }
static int die(const std::string& msg) { fprintf(stderr, "ERROR simmr-hip: %s\n", msg.c_str()); return 1; }
static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

struct DeviceOut {
  simmr_reads_out o{};
  std::vector<void*> allocs;
  ~DeviceOut() { for (void* p : allocs) (void)hipFree(p); }
  template <class T> bool alloc(T** dst, size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    allocs.push_back(p);
    *dst = (T*)p;
    return true;
  }
  bool init(uint64_t n_reads, uint64_t total_bases) {
    o.seq_capacity = total_bases + 32;
    o.reads_capacity = n_reads;
    o.qual_offset = 33;  // util::encode_quality_scores (util.rs:46-57)
    return alloc(&o.seq, o.seq_capacity) && alloc(&o.qual, o.seq_capacity) && alloc(&o.seq_off, n_reads + 1) &&
           alloc(&o.start, n_reads) && alloc(&o.end, n_reads) && alloc(&o.contig, n_reads) &&
           alloc(&o.genome, n_reads) && alloc(&o.read_id, n_reads) && alloc(&o.flags, n_reads);
  }
  bool to_host(uint64_t n_reads, uint64_t total_bases, bool paired, HostReads* h) {
    h->n_reads = n_reads; h->paired = paired;
    h->seq.resize(total_bases); h->qual.resize(total_bases);
    h->seq_off.resize(n_reads + 1); h->start.resize(n_reads); h->end.resize(n_reads);
    h->contig.resize(n_reads); h->genome.resize(n_reads); h->read_id.resize(n_reads); h->flags.resize(n_reads);
    auto cp = [](void* d, const void* s, size_t n) { return n == 0 || hipMemcpy(d, s, n, hipMemcpyDeviceToHost) == hipSuccess; };
    return cp(h->seq.data(), o.seq, total_bases) && cp(h->qual.data(), o.qual, total_bases) &&
           cp(h->seq_off.data(), o.seq_off, (n_reads + 1) * 8) && cp(h->start.data(), o.start, n_reads * 8) &&
           cp(h->end.data(), o.end, n_reads * 8) && cp(h->contig.data(), o.contig, n_reads * 4) &&
           cp(h->genome.data(), o.genome, n_reads * 4) && cp(h->read_id.data(), o.read_id, n_reads * 4) &&
           cp(h->flags.data(), o.flags, n_reads);
  }
};

// FASTQ text built on the device (simmr_fastq_plan / simmr_fastq_emit), appended to `output`.
// Returns 1 when the library leaves this input to the host writer (SIMMR_ENOTSUP), -1 on error.
static int write_fastq_device(simmr_engine* eng, const std::string& fmt, const std::vector<Genome>& genomes, size_t g0,
                              size_t g1, const simmr_reads_out& reads, uint64_t n_reads, bool paired,
                              const std::string& output, std::string* err) {
  std::vector<uint32_t> idx, ncontigs;
  std::vector<const char*> gids, sids;
  for (size_t gi = g0; gi < g1; gi++) {
    idx.push_back((uint32_t)gi);
    gids.push_back(genomes[gi].uuid.c_str());
    ncontigs.push_back((uint32_t)genomes[gi].sequence.size());
    for (const Seq& s : genomes[gi].sequence) sids.push_back(s.id.c_str());
  }
  const simmr_fastq_names names{(uint32_t)idx.size(), idx.data(), gids.data(), ncontigs.data(), sids.data()};
  uint64_t total = 0;
  int rc = simmr_fastq_plan(eng, fmt.c_str(), &names, &reads, n_reads, paired ? 1 : 0, &total);
  if (rc == SIMMR_ENOTSUP) return 1;
  if (rc != SIMMR_OK) { *err = simmr_last_error(eng); return -1; }
  void* dev = nullptr;
  // no room for the framed text next to the reads (it is about 1.2 x their size): not the reference's failure mode,
  // so no output may be lost over it — the host writer takes over
  if (hipMalloc(&dev, total ? total : 1) != hipSuccess) { (void)hipGetLastError(); return 1; }
  rc = simmr_fastq_emit(eng, &reads, (uint8_t*)dev, total);
  if (rc != SIMMR_OK) { *err = simmr_last_error(eng); (void)hipFree(dev); return -1; }
  FILE* f = fopen(output.c_str(), "ab");
  if (!f) { *err = "cannot open " + output; (void)hipFree(dev); return -1; }
  // drain through two pinned buffers: the copy of chunk i + 1 runs while chunk i is written to the file
  const size_t chunk = 256u << 20;
  void* pin[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  bool ok = hipHostMalloc(&pin[0], chunk, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc(&pin[1], chunk, hipHostMallocDefault) == hipSuccess && hipStreamCreate(&cs) == hipSuccess &&
            hipDeviceSynchronize() == hipSuccess;
  const uint64_t n_chunks = (total + chunk - 1) / chunk;
  auto len_of = [&](uint64_t i) { return (size_t)std::min<uint64_t>(chunk, total - i * chunk); };
  if (ok && n_chunks > 0) ok = hipMemcpyAsync(pin[0], (const char*)dev, len_of(0), hipMemcpyDeviceToHost, cs) == hipSuccess;
  for (uint64_t i = 0; ok && i < n_chunks; i++) {
    ok = hipStreamSynchronize(cs) == hipSuccess;  // chunk i is in pin[i & 1]
    if (ok && i + 1 < n_chunks)
      ok = hipMemcpyAsync(pin[(i + 1) & 1], (const char*)dev + (i + 1) * chunk, len_of(i + 1), hipMemcpyDeviceToHost, cs) == hipSuccess;
    if (ok && fwrite(pin[i & 1], 1, len_of(i), f) != len_of(i)) { *err = "short write to " + output; ok = false; }
  }
  if (!ok && err->empty()) *err = "copy back failed";
  fclose(f);
  if (cs) (void)hipStreamDestroy(cs);
  for (void* q : pin) if (q) (void)hipHostFree(q);
  (void)hipFree(dev);
  return ok ? 0 : -1;
}

// Genome::from_fasta (genome.rs:89-162) with the sequences going straight to the device: the host only finds
// the records; simmr_stage_fasta normalises (needletail normalize(false), genome.rs:114), applies the size filter
// of main.rs:117-162 and packs.  Returns 1 if the genome has no usable sequence (it is left out), -1 on error.
static int load_genome_device(simmr_engine* eng, uint32_t slot, const std::string& path, bool contiguous, uint64_t min_size,
                              Genome* g, std::string* err) {
  FastaRecords recs;
  if (!scan_fasta(path, &recs, err)) return -1;
  const uint32_t n = (uint32_t)recs.ids.size();
  std::vector<const uint8_t*> body(n);
  std::vector<uint64_t> len(n), count(n);
  for (uint32_t c = 0; c < n; c++) { body[c] = (const uint8_t*)recs.data.data() + recs.body[c].first; len[c] = recs.body[c].second; }
  uint32_t n_staged = 0;
  if (simmr_stage_fasta(eng, slot, n, body.data(), len.data(), contiguous ? 1 : 0, contiguous ? 0 : min_size, count.data(),
                        &n_staged) != SIMMR_OK) {
    *err = simmr_last_error(eng);
    return -1;
  }
  g->uuid = uuid_from_u64(generate_id());  // genome.rs:124,140
  g->filepath = path;
  g->contiguous = contiguous;
  g->size = 0;
  for (uint32_t c = 0; c < n; c++) g->size += count[c];
  g->sequence.clear();
  if (contiguous) {  // genome.rs:121-137
    Seq whole;
    whole.id = "whole genome";
    whole.uuid = generate_id();
    whole.size = g->size;
    g->sequence.push_back(std::move(whole));
  } else {
    for (uint32_t c = 0; c < n; c++) {
      if (count[c] <= min_size) {
        warn("(" + path + ") Sequence " + recs.ids[c] + " doesn't meet size requirements, size = " + std::to_string(count[c]) +
             ", min size = " + std::to_string(min_size));
        continue;
      }
      Seq q;
      q.id = recs.ids[c];
      q.uuid = generate_id();
      q.size = count[c];
      g->sequence.push_back(std::move(q));
    }
  }
  g->num_seqs = g->sequence.size();
  if (n_staged == 0) { warn("Removing " + path + " from simulation, it doesn't have usable sequences"); return 1; }
  return 0;
}

static int run_main(int argc, char** argv) {
  CliArgs args;
  std::string err;
  bool help = false;
  if (!parse_cli_args(argc, argv, &args, &err, &help)) { fprintf(stderr, "error: %s\n\n%s", err.c_str(), usage().c_str()); return 2; }
  if (help) { fputs(usage().c_str(), stdout); return 0; }

  std::unique_ptr<ErrorProfile> eprofile = determine_error_profile(args, &err);  // main.rs:27
  if (!eprofile) return die(err);
  // main.rs:30-33
  if (args.error_profile == ErrorProfileKind::CustomShort && eprofile->is_long_read())
    return die("You specified a custom short-read error profile but the provided error profile is for long reads");
  if (args.error_profile == ErrorProfileKind::CustomLong && !eprofile->is_long_read())
    return die("You specified a custom long-read error profile but the provided error profile is for short reads");

  simmr_engine* eng = nullptr;
  if (simmr_engine_create(args.device, &eng) != SIMMR_OK) return die(std::string("cannot create engine: ") + simmr_last_error(nullptr));

  info("Loading genomes");
  std::vector<Genome> genomes;
  const uint64_t device_min_size = eprofile->minimum_genome_size();
  if (!args.host_normalize) {
    // main.rs:38-162 with the sequences normalised, filtered and packed on the device
    std::vector<GenomeRecord> records;
    if (args.genome_file) {
      if (!parse_genome_file(*args.genome_file, &records, &err)) return die("Failed to read genome file: " + err);
      for (const auto& rec : records)
        if (!exists(rec.filepath)) return die("Genome (" + rec.filepath + ") does not exist");
    } else {
      for (const auto& path : args.genome) { GenomeRecord r; r.filepath = path; records.push_back(r); }
    }
    for (const auto& rec : records) {
      Genome g;
      const int rc = load_genome_device(eng, (uint32_t)genomes.size(), rec.filepath, args.contiguous, device_min_size, &g, &err);
      if (rc < 0) return die("Failed to parse " + rec.filepath + ": " + err);
      if (rec.uuid) g.uuid = *rec.uuid;
      if (args.genome_file && args.abundance_profile == AbundanceProfileKind::Custom && !rec.abundance)
        return die("You used a custom abundance profile but didn't provide abundances for genome " + g.filepath);
      g.abundance = rec.abundance;
      if (rc == 0) genomes.push_back(std::move(g));
    }
  } else if (args.genome_file) {  // main.rs:38-100
    std::vector<GenomeRecord> records;
    if (!parse_genome_file(*args.genome_file, &records, &err)) return die("Failed to read genome file: " + err);
    for (const auto& rec : records)
      if (!exists(rec.filepath)) return die("Genome (" + rec.filepath + ") does not exist");
    for (const auto& rec : records) {
      Genome g;
      if (!Genome::from_fasta(rec.filepath, args.contiguous, &g, &err)) return die("Failed to parse " + rec.filepath + ": " + err);
      if (rec.uuid) g.uuid = *rec.uuid;
      if (args.abundance_profile == AbundanceProfileKind::Custom && !rec.abundance)
        return die("You used a custom abundance profile but didn't provide abundances for genome " + g.filepath);
      g.abundance = rec.abundance;
      genomes.push_back(std::move(g));
    }
  } else {  // main.rs:101-117
    for (const auto& path : args.genome) {
      Genome g;
      if (!Genome::from_fasta(path, args.contiguous, &g, &err)) return die("Failed to parse " + path + ": " + err);
      genomes.push_back(std::move(g));
    }
  }
  if (args.abundance_profile == AbundanceProfileKind::Custom && !args.genome_file)
    return die("a custom abundance profile needs a --genome-file with abundances");

  info("Ensuring genomes meet minimum sequence length requirements for simulation");
  if (!args.contiguous && args.host_normalize) {  // main.rs:117-162
    const uint64_t min_size = eprofile->minimum_genome_size();
    std::vector<Genome> kept;
    for (Genome& g : genomes) {
      std::vector<Seq> seqs;
      for (Seq& s : g.sequence) {
        if (s.size <= min_size)
          warn("(" + g.filepath + ") Sequence " + s.id + " doesn't meet size requirements, size = " +
               std::to_string(s.size) + ", min size = " + std::to_string(min_size));
        else
          seqs.push_back(std::move(s));
      }
      g.sequence = std::move(seqs);
      if (g.sequence.empty()) { warn("Removing " + g.filepath + " from simulation, it doesn't have usable sequences"); continue; }
      g.num_seqs = g.sequence.size();
      kept.push_back(std::move(g));
    }
    genomes = std::move(kept);
  }
  if (genomes.empty()) return die("no usable genomes");

  std::optional<std::vector<double>> custom_ab;
  if (args.abundance_profile == AbundanceProfileKind::Custom) {
    custom_ab.emplace();
    for (const Genome& g : genomes) custom_ab->push_back(*g.abundance);
  }
  std::unique_ptr<AbundanceProfile> aprofile = determine_abundance_profile(args, custom_ab);

  // ---- stage the references once (replaces keeping Vec<Seq> in RAM for the loop)
  for (size_t gi = 0; args.host_normalize && gi < genomes.size(); gi++) {
    const Genome& g = genomes[gi];
    std::vector<const uint8_t*> ptrs;
    std::vector<uint64_t> lens, sizes;
    for (const Seq& s : g.sequence) { ptrs.push_back((const uint8_t*)s.seq.data()); lens.push_back(s.seq.size()); sizes.push_back(s.size); }
    if (simmr_stage_genome(eng, (uint32_t)gi, (uint32_t)ptrs.size(), ptrs.data(), lens.data(), sizes.data()) != SIMMR_OK)
      return die(std::string("staging failed: ") + simmr_last_error(eng));
  }

  // abundances (simulate.rs:121-132 / :334-343)
  const bool is_long = eprofile->is_long_read();
  Abundances ab = aprofile->determine_abundances(args.num_reads, genomes.size());
  if (aprofile->is_size_aware()) ab = aprofile->adjust_for_size(genomes, ab, is_long ? 20000 : args.read_length, !is_long);

  // main.rs:191-198: remove previous outputs
  if (exists(args.output)) remove(args.output.c_str());
  const std::string meta_path = args.output + ".tsv";
  if (exists(meta_path)) remove(meta_path.c_str());

  const simmr_error_profile pod = eprofile->pod();
  const int has_seed = args.seed ? 1 : 0;
  const uint64_t seed = args.seed.value_or(0);
  const simmr_range all{0, UINT64_MAX};

  if (!is_long) {
    info("Simulating short reads");
    // all genomes in one device plan (simulate_pe_reads, simulate.rs:110-150); the library leaves
    // custom profiles to the genome-by-genome loop below
    std::vector<uint32_t> all_idx(genomes.size());
    std::vector<uint64_t> all_reads(genomes.size());
    for (size_t gi = 0; gi < genomes.size(); gi++) { all_idx[gi] = (uint32_t)gi; all_reads[gi] = ab[gi].first; }
    simmr_plan_info mpi{};
    const int mrc = simmr_pe_plan_multi(eng, (uint32_t)genomes.size(), all_idx.data(), all_reads.data(), &pod, has_seed, seed, all, &mpi);
    if (mrc != SIMMR_OK && mrc != SIMMR_ENOTSUP) return die(simmr_last_error(eng));
    if (mrc == SIMMR_OK) {
      DeviceOut d;
      if (!d.init(mpi.n_reads, mpi.total_bases)) return die("device allocation failed");
      if (simmr_pe_emit(eng, 0, &d.o) != SIMMR_OK) return die(simmr_last_error(eng));
      int fq = args.host_fastq ? 1 : write_fastq_device(eng, args.read_header_format, genomes, 0, genomes.size(), d.o, mpi.n_reads,
                                                      true, args.output, &err);
      if (fq < 0) fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
      if (fq > 0) {  // host writer (fastq.rs restated in host.cpp), genome by genome
        HostReads h;
        if (!d.to_host(mpi.n_reads, mpi.total_bases, true, &h)) return die("copy back failed");
        uint64_t first_read = 0;
        for (size_t gi = 0; gi < genomes.size(); gi++) {
          const uint64_t n = ab[gi].first / 2 * 2;
          if (!write_to_fastq(genomes[gi].uuid, genomes[gi], h, first_read, n, args.output, args.read_header_format, true, &err))
            fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
          first_read += n;
        }
      }
    }
    uint32_t id_base = 0;  // the global AtomicU32 of simulate.rs:85-89
    for (size_t gi = 0; mrc == SIMMR_ENOTSUP && gi < genomes.size(); gi++) {
      simmr_plan_info pi{};
      if (simmr_pe_plan(eng, (uint32_t)gi, &pod, ab[gi].first, has_seed, seed, all, &pi) != SIMMR_OK)
        return die(simmr_last_error(eng));  // the reference unwrap()s this Err (simulate.rs:137)
      DeviceOut d;
      if (!d.init(pi.n_reads, pi.total_bases)) return die("device allocation failed");
      if (simmr_pe_emit(eng, id_base, &d.o) != SIMMR_OK) return die(simmr_last_error(eng));
      int fq = args.host_fastq ? 1 : write_fastq_device(eng, args.read_header_format, genomes, gi, gi + 1, d.o, pi.n_reads, true,
                                                      args.output, &err);
      if (fq < 0) fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
      if (fq > 0) {  // host writer (fastq.rs restated in host.cpp)
        HostReads h;
        if (!d.to_host(pi.n_reads, pi.total_bases, true, &h)) return die("copy back failed");
        if (!write_to_fastq(genomes[gi].uuid, genomes[gi], h, 0, pi.n_reads, args.output, args.read_header_format, true, &err))
          fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
      }
      id_base += (uint32_t)pi.n_units;
    }
  } else {
    info("Simulating long reads");
    std::vector<uint32_t> idx(genomes.size());
    std::vector<uint64_t> reads(genomes.size());
    for (size_t gi = 0; gi < genomes.size(); gi++) { idx[gi] = (uint32_t)gi; reads[gi] = ab[gi].first; }
    simmr_plan_info pi{};
    if (simmr_long_plan(eng, (uint32_t)genomes.size(), idx.data(), reads.data(), &pod, has_seed, seed, all, &pi) != SIMMR_OK)
      return die(simmr_last_error(eng));
    DeviceOut d;
    if (!d.init(pi.n_reads, pi.total_bases)) return die("device allocation failed");
    if (simmr_long_emit(eng, 0, &d.o) != SIMMR_OK) return die(simmr_last_error(eng));
    int fq = args.host_fastq ? 1 : write_fastq_device(eng, args.read_header_format, genomes, 0, genomes.size(), d.o, pi.n_reads,
                                                    false, args.output, &err);
    if (fq < 0) fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
    if (fq > 0) {
      HostReads h;
      if (!d.to_host(pi.n_reads, pi.total_bases, false, &h)) return die("copy back failed");
      uint64_t first = 0;
      for (size_t gi = 0; gi < genomes.size(); gi++) {
        if (!write_to_fastq(genomes[gi].uuid, genomes[gi], h, first, reads[gi], args.output, args.read_header_format, true, &err))
          fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
        first += reads[gi];
      }
    }
  }
  info(("Writing simulated reads to " + args.output).c_str());

  // main.rs:213-258
  std::vector<MetadataRow> rows;
  for (size_t gi = 0; gi < genomes.size(); gi++) rows.push_back({genomes[gi].uuid, genomes[gi].filepath, ab[gi].first, ab[gi].second});
  info(("Writing simulation metadata to " + meta_path).c_str());
  if (!write_metadata(rows, meta_path, &err)) fprintf(stderr, "ERROR simmr-hip: Failed to write metadata file: %s\n", err.c_str());
  simmr_engine_destroy(eng);
  return 0;
}

int main(int argc, char** argv) { return run_main(argc, argv); }
