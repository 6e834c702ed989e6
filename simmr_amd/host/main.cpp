// main.cpp — `simmr-hip`: the reference's run_main (simmr/src/main.rs:20-268)
// over the C ABI of libsimmr_hip.so.  Same flags, same output files:
// interleaved FASTQ (fastq.rs) and "<output>.tsv" metadata (files.rs:100-134).
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "simmr_host.hpp"

using namespace simmr_host;

static void info(const char* msg) { fprintf(stderr, " INFO simmr-hip: %s\n", msg); }
static void warn(const std::string& msg) { fprintf(stderr, " WARN simmr-hip: %s\n", msg.c_str()); }
static int die(const std::string& msg) { fprintf(stderr, "ERROR simmr-hip: %s\n", msg.c_str()); return 1; }
static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static bool is_regular_file(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

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
  bool init(uint64_t n_reads, uint64_t total_bases, uint32_t slot_bytes) {
    o.slot_bytes = slot_bytes;  // simmr_plan_info.slot_bytes: the layout the plan in force emits
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
    if (!(cp(h->seq.data(), o.seq, total_bases) && cp(h->qual.data(), o.qual, total_bases) &&
          cp(h->seq_off.data(), o.seq_off, (n_reads + 1) * 8) && cp(h->start.data(), o.start, n_reads * 8) &&
          cp(h->end.data(), o.end, n_reads * 8) && cp(h->contig.data(), o.contig, n_reads * 4) &&
          cp(h->genome.data(), o.genome, n_reads * 4) && cp(h->read_id.data(), o.read_id, n_reads * 4) &&
          cp(h->flags.data(), o.flags, n_reads)))
      return false;
    if (o.slot_bytes == SIMMR_SLOT16) {
      // the consumer's side of the slot layout (include/simmr_hip.h): L = |end - start|, bases from seq_off[r], qualities
      // from seq_off[r] & ~15 — closed up in place into the compact form write_to_fastq reads (reads ascend in both)
      uint64_t at = 0;
      for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t L = h->end[r] > h->start[r] ? h->end[r] - h->start[r] : h->start[r] - h->end[r];
        const uint64_t sb = h->seq_off[r], qb = sb & ~15ull;
        memmove(h->seq.data() + at, h->seq.data() + sb, L);
        memmove(h->qual.data() + at, h->qual.data() + qb, L);
        h->seq_off[r] = at;
        at += L;
      }
      h->seq_off[n_reads] = at;
      h->seq.resize(at); h->qual.resize(at);
    }
    return true;
  }
};

// ---- output of one planned range ------------------------------------------------------------------------------------
// A run is generated range by range of its units (pairs / long reads): the reference holds every read of a run in RAM
// before it writes (main.rs:180-206, readme.md:219-220); here a range is what fits the device next to the reference
// (--device-chunk-reads, or a share of the free memory), and the FASTQ text of range k is copied out and written
// while range k + 1 is planned and emitted.  The text of a range comes straight from its plan
// (simmr_fastq_plan_direct / simmr_emit_fastq: no SoA columns in between).
struct TextDrain {
  // two device buffers for the text and two pinned buffers for the copy out; the file is appended to in order
  void* dev[2] = {nullptr, nullptr};
  uint64_t cap[2] = {0, 0}, len[2] = {0, 0};
  void* pin[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  hipEvent_t emitted = nullptr;
  FILE* f = nullptr;
  static constexpr size_t CHUNK = 256u << 20;
  int pending = -1;  // buffer whose text still has to go to the file
  bool own_file = true;
  // `shared`: the file several drains append to, each when it is its turn (run_scope_devices); the caller closes it
  bool open(const std::string& output, std::string* err, FILE* shared = nullptr) {
    if (shared) { f = shared; own_file = false; }
    else f = fopen(output.c_str(), "ab");
    if (!f) { *err = "cannot open " + output; return false; }
    if (hipHostMalloc(&pin[0], CHUNK, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&pin[1], CHUNK, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&emitted, hipEventDisableTiming) != hipSuccess) {
      *err = "pinned buffer / stream allocation failed";
      return false;
    }
    return true;
  }
  // a device buffer of at least `bytes` for the next range; the buffer is free again: its previous text was drained
  // two ranges ago.  nullptr if the device has no room.
  uint8_t* buffer(int b, uint64_t bytes) {
    if (cap[b] < bytes) {
      if (dev[b]) (void)hipFree(dev[b]);
      dev[b] = nullptr; cap[b] = 0;
      if (hipMalloc(&dev[b], bytes + bytes / 16 + 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      cap[b] = bytes + bytes / 16 + 256;
    }
    return (uint8_t*)dev[b];
  }
  // the text in buffer b (written by work queued on the null stream up to now) goes to the file; returns at once after
  // queueing the first copy, the rest happens in flush()
  bool submit(int b, uint64_t bytes, std::string* err) {
    if (!flush(err)) return false;
    len[b] = bytes;
    if (hipEventRecord(emitted, nullptr) != hipSuccess || hipStreamWaitEvent(cs, emitted, 0) != hipSuccess) { *err = "event failed"; return false; }
    if (bytes > 0 && hipMemcpyAsync(pin[0], dev[b], std::min<uint64_t>(CHUNK, bytes), hipMemcpyDeviceToHost, cs) != hipSuccess) { *err = "copy back failed"; return false; }
    pending = b;
    return true;
  }
  // write the pending text: the copy of chunk i + 1 runs while chunk i is written to the file
  bool flush(std::string* err) {
    if (pending < 0) return true;
    const int b = pending;
    pending = -1;
    const uint64_t total = len[b], n_chunks = (total + CHUNK - 1) / CHUNK;
    auto len_of = [&](uint64_t i) { return (size_t)std::min<uint64_t>(CHUNK, total - i * CHUNK); };
    for (uint64_t i = 0; i < n_chunks; i++) {
      if (hipStreamSynchronize(cs) != hipSuccess) { *err = "copy back failed"; return false; }  // chunk i is in pin[i & 1]
      if (i + 1 < n_chunks && hipMemcpyAsync(pin[(i + 1) & 1], (const char*)dev[b] + (i + 1) * CHUNK, len_of(i + 1), hipMemcpyDeviceToHost, cs) != hipSuccess) {
        *err = "copy back failed";
        return false;
      }
      if (fwrite(pin[i & 1], 1, len_of(i), f) != len_of(i)) { *err = "short write"; return false; }
    }
    return true;
  }
  ~TextDrain() {
    if (f && own_file) fclose(f);
    if (cs) (void)hipStreamDestroy(cs);
    if (emitted) (void)hipEventDestroy(emitted);
    for (void* q : pin) if (q) (void)hipHostFree(q);
    for (void* q : dev) if (q) (void)hipFree(q);
  }
};

struct NameTables {  // simmr_fastq_names of genomes [g0, g1)
  std::vector<uint32_t> idx, ncontigs;
  std::vector<const char*> gids, sids;
  simmr_fastq_names names{};
  NameTables(const std::vector<Genome>& genomes, size_t g0, size_t g1) {
    for (size_t gi = g0; gi < g1; gi++) {
      idx.push_back((uint32_t)gi);
      gids.push_back(genomes[gi].uuid.c_str());
      ncontigs.push_back((uint32_t)genomes[gi].sequence.size());
      for (const Seq& s : genomes[gi].sequence) sids.push_back(s.id.c_str());
    }
    names = simmr_fastq_names{(uint32_t)idx.size(), idx.data(), gids.data(), ncontigs.data(), sids.data()};
  }
};

// One scope of the run = one plan function over a range of units: all genomes' pairs in one plan, one genome's pairs
// (custom profiles), or all long reads.  `genome_units[g - g0]` = units of genome g, in generation order.
struct Scope {
  bool paired;
  size_t g0, g1;
  std::vector<uint64_t> genome_units;
  uint32_t id_base;
  // plans units [first, first + count) of the scope; SIMMR_* code
  std::function<int(simmr_engine*, simmr_range, simmr_plan_info*)> plan;
  std::function<int(simmr_engine*, uint32_t, const simmr_reads_out*)> emit;  // columns (host writer only)
};

// Generates the scope range by range and appends its FASTQ to args.output.  0, or 1 after die().
static int run_scope(simmr_engine* eng, const CliArgs& args, const std::vector<Genome>& genomes, const Scope& sc, uint64_t chunk_units) {
  uint64_t total_units = 0;
  for (uint64_t u : sc.genome_units) total_units += u;
  const uint32_t rpu = sc.paired ? 2u : 1u;
  if (chunk_units == 0 || chunk_units > total_units) chunk_units = std::max<uint64_t>(total_units, 1);
  NameTables nt(genomes, sc.g0, sc.g1);
  std::string err;
  bool use_device_text = !args.host_fastq;
  uint64_t text_bytes = 0, n_ranges = 0;
  TextDrain drain;
  if (use_device_text && !drain.open(args.output, &err)) return die(err);
  int buf = 0;
  for (uint64_t first = 0; first < total_units || (first == 0 && total_units == 0); first += chunk_units) {
    const simmr_range rg{first, std::min<uint64_t>(chunk_units, total_units - first)};
    simmr_plan_info pi{};
    if (sc.plan(eng, rg, &pi) != SIMMR_OK) return die(simmr_last_error(eng));  // the reference unwrap()s this Err (simulate.rs:137)
    bool written = false;
    if (use_device_text) {
      uint64_t bytes = 0;
      const int rc = simmr_fastq_plan_direct(eng, args.read_header_format.c_str(), &nt.names, sc.id_base, &bytes);
      if (rc == SIMMR_ENOTSUP) {
        use_device_text = false;  // an id with a brace, a header over 255 bytes: the host writer frames this run
        if (!drain.flush(&err)) return die(err);
        fclose(drain.f); drain.f = nullptr;
      } else if (rc != SIMMR_OK) {
        return die(simmr_last_error(eng));
      } else if (bytes == 0) {
        written = true;  // a scope without a unit (--num-reads < 2, a genome whose share is below one pair): the reference
                         // writes nothing for it and goes on (simulate.rs:179, main.rs:188-206)
      } else {
        uint8_t* dst = drain.buffer(buf, bytes);
        if (!dst) return die("no device memory for " + std::to_string(bytes) + " bytes of FASTQ text: use a smaller --device-chunk-reads");
        if (simmr_emit_fastq(eng, dst, bytes) != SIMMR_OK) return die(simmr_last_error(eng));
        if (!drain.submit(buf, bytes, &err)) { fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str()); }
        buf ^= 1;
        written = true;
        text_bytes += bytes;
        n_ranges++;
      }
    }
    if (!written) {  // columns to the host, framed by the restatement of fastq.rs in host.cpp, genome by genome
      DeviceOut d;
      if (!d.init(pi.n_reads, pi.total_bases, pi.slot_bytes)) return die("device allocation failed");
      if (sc.emit(eng, sc.id_base, &d.o) != SIMMR_OK) return die(simmr_last_error(eng));
      HostReads h;
      if (!d.to_host(pi.n_reads, pi.total_bases, sc.paired, &h)) return die("copy back failed");
      uint64_t g_first = 0;  // first unit of genome gi in the scope
      for (size_t gi = sc.g0; gi < sc.g1; gi++) {
        const uint64_t g_units = sc.genome_units[gi - sc.g0];
        const uint64_t lo = std::max(first, g_first), hi = std::min(first + rg.count, g_first + g_units);
        if (hi > lo &&
            !write_to_fastq(genomes[gi].uuid, genomes[gi], h, (lo - first) * rpu, (hi - lo) * rpu, args.output, args.read_header_format, true, &err))
          fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
        g_first += g_units;
      }
    }
    if (total_units == 0) break;
  }
  if (use_device_text && !drain.flush(&err)) fprintf(stderr, "ERROR simmr-hip: Failed to write reads to the output file: %s\n", err.c_str());
  if (n_ranges > 1)
    info((std::to_string(total_units * rpu) + " reads in " + std::to_string(n_ranges) + " device passes, " + std::to_string(text_bytes) + " bytes of FASTQ").c_str());
  return 0;
}

// ---- the same over several engines (--devices a,b,...): the whole node behind the reference's one call ------------------
// The reference's product is one call that writes one FASTQ (main.rs:180-206).  Here the scope's ranges — the same ranges
// run_scope walks, whose text does not depend on how the run is cut (tests/test_gpu_cli.py) — are dealt to the engines in
// turn: engine d plans and emits ranges d, d + N, d + 2N, ... on a host thread of its own (one engine per device, or
// several on one: an ordinal may repeat), copies each range's text to pinned host memory over ITS device's link as soon as
// it is emitted — the devices' copies run side by side; a drain that waited for its turn would put the whole node behind one
// PCIe link — and appends it to the file when the range before it is on disk.  Ranges are at most MAX_RANGE_TEXT bytes of
// text (a range waits in host memory for its turn).  Ids are
// those of the single-engine run (the library's ids come from the global unit index, simulate.rs:85-89); nothing is
// exchanged between devices, the run counters are not needed for the files.  0, or 1 after die().
static int run_scope_devices(const std::vector<simmr_engine*>& engs, const std::vector<int>& ordinals, const CliArgs& args,
                             const std::vector<Genome>& genomes, const Scope& sc, uint64_t chunk_units, uint64_t text_bytes_per_unit) {
  constexpr uint64_t MAX_RANGE_TEXT = 1ull << 30;  // (pinned host memory per engine; allocating it costs ~0.25 s per GB, once)
  uint64_t total_units = 0;
  for (uint64_t u : sc.genome_units) total_units += u;
  if (total_units == 0) return 0;  // (the reference writes nothing for a scope without a unit)
  const uint32_t rpu = sc.paired ? 2u : 1u;
  const size_t N = engs.size();
  // ranges no larger than a device pass, and at least one per engine
  if (chunk_units == 0 || chunk_units > total_units) chunk_units = total_units;
  chunk_units = std::max<uint64_t>(std::min<uint64_t>(chunk_units, (total_units + N - 1) / N), 1);
  chunk_units = std::max<uint64_t>(std::min<uint64_t>(chunk_units, MAX_RANGE_TEXT / std::max<uint64_t>(text_bytes_per_unit, 1)), 1);
  const uint64_t n_ranges = (total_units + chunk_units - 1) / chunk_units;
  FILE* f = fopen(args.output.c_str(), "ab");
  if (!f) return die("cannot open " + args.output);
  std::mutex m;
  std::condition_variable cv;
  uint64_t turn = 0;        // the range whose text goes to the file next
  bool failed = false;
  std::string first_error;
  uint64_t text_bytes = 0;
  auto fail = [&](const std::string& what) {
    std::lock_guard<std::mutex> lk(m);
    if (!failed) { failed = true; first_error = what; }
    cv.notify_all();
  };
  auto worker = [&](size_t d) {
    simmr_engine* eng = engs[d];
    if (hipSetDevice(ordinals[d]) != hipSuccess) return fail("hipSetDevice failed");
    NameTables nt(genomes, sc.g0, sc.g1);
    // this engine's text buffers: one on the device, one pinned on the host (both grow to the largest range), a copy stream
    struct Bufs {
      void* dev = nullptr; void* host = nullptr; uint64_t dev_cap = 0, host_cap = 0; hipStream_t cs = nullptr;
      ~Bufs() { if (dev) (void)hipFree(dev); if (host) (void)hipHostFree(host); if (cs) (void)hipStreamDestroy(cs); }
    } b;
    if (hipStreamCreateWithFlags(&b.cs, hipStreamNonBlocking) != hipSuccess) return fail("stream allocation failed");
    for (uint64_t k = d; k < n_ranges; k += N) {
      { std::lock_guard<std::mutex> lk(m); if (failed) return; }
      const simmr_range rg{k * chunk_units, std::min<uint64_t>(chunk_units, total_units - k * chunk_units)};
      simmr_plan_info pi{};
      if (sc.plan(eng, rg, &pi) != SIMMR_OK) return fail(simmr_last_error(eng));
      uint64_t bytes = 0;
      const int rc = simmr_fastq_plan_direct(eng, args.read_header_format.c_str(), &nt.names, sc.id_base, &bytes);
      if (rc == SIMMR_ENOTSUP) return fail(std::string("--devices writes the text on the devices, and this run's headers need the host writer (") + simmr_last_error(eng) + "): use --device with --host-fastq");
      if (rc != SIMMR_OK) return fail(simmr_last_error(eng));
      if (bytes > b.dev_cap) {
        if (b.dev) (void)hipFree(b.dev);
        b.dev = nullptr; b.dev_cap = 0;
        if (hipMalloc(&b.dev, bytes + bytes / 16 + 256) != hipSuccess) { (void)hipGetLastError(); return fail("no device memory for " + std::to_string(bytes) + " bytes of FASTQ text: use a smaller --device-chunk-reads"); }
        b.dev_cap = bytes + bytes / 16 + 256;
      }
      if (bytes > b.host_cap) {
        if (b.host) (void)hipHostFree(b.host);
        b.host = nullptr; b.host_cap = 0;
        if (hipHostMalloc(&b.host, bytes + bytes / 16 + 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail("no pinned host memory for " + std::to_string(bytes) + " bytes of FASTQ text"); }
        b.host_cap = bytes + bytes / 16 + 256;
      }
      if (bytes) {
        if (simmr_emit_fastq(eng, (uint8_t*)b.dev, bytes) != SIMMR_OK) return fail(simmr_last_error(eng));
        // over this device's own link, now: the emit ran on the null stream of the device, the copy stream waits for it
        hipEvent_t done = nullptr;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, nullptr) != hipSuccess ||
            hipStreamWaitEvent(b.cs, done, 0) != hipSuccess ||
            hipMemcpyAsync(b.host, b.dev, bytes, hipMemcpyDeviceToHost, b.cs) != hipSuccess || hipStreamSynchronize(b.cs) != hipSuccess) {
          if (done) (void)hipEventDestroy(done);
          return fail("copy back failed");
        }
        (void)hipEventDestroy(done);
      }
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return failed || turn == k; });
        if (failed) return;
      }
      const bool ok = bytes == 0 || fwrite(b.host, 1, bytes, f) == bytes;  // this thread owns the file until it passes the turn on
      {
        std::lock_guard<std::mutex> lk(m);
        if (!ok && !failed) { failed = true; first_error = "Failed to write reads to the output file: short write"; }
        text_bytes += bytes;
        turn = k + 1;
      }
      cv.notify_all();
      if (!ok) return;
    }
  };
  std::vector<std::thread> threads;
  for (size_t d = 0; d < N; d++) threads.emplace_back(worker, d);
  for (auto& t : threads) t.join();
  fclose(f);
  if (failed) return die(first_error);
  info((std::to_string(total_units * rpu) + " reads in " + std::to_string(n_ranges) + " device passes on " + std::to_string(N) +
        " engines, " + std::to_string(text_bytes) + " bytes of FASTQ").c_str());
  return 0;
}

// units per range when --device-chunk-reads is not given: what a third of the free device memory holds, at
// `text_bytes_per_unit` of FASTQ text (two buffers) plus the plan's columns per unit
static uint64_t auto_chunk_units(uint64_t text_bytes_per_unit) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  const uint64_t per_unit = 2 * text_bytes_per_unit + 96;
  return std::max<uint64_t>((uint64_t)free_b / 3 / per_unit, 1024);
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

  // one engine per entry of --devices (an ordinal may repeat: two engines on one device), or the one of --device
  const std::vector<int> ordinals = args.devices.empty() ? std::vector<int>{args.device} : args.devices;
  if (ordinals.size() > 1 && args.host_fastq) return die("--devices writes the text on the devices: it does not combine with --host-fastq");
  std::vector<simmr_engine*> engs;
  for (int ord : ordinals) {
    simmr_engine* en = nullptr;
    if (simmr_engine_create(ord, &en) != SIMMR_OK) return die(std::string("cannot create engine: ") + simmr_last_error(nullptr));
    // 16-byte read slots wherever the emit kernel of a plan writes them (include/simmr_hip.h: simmr_reads_out); the columns
    // only exist on the --host-fastq path (the text path writes no columns), and DeviceOut::to_host reads either layout
    if (simmr_engine_set_read_slots(en, SIMMR_SLOT16) != SIMMR_OK) return die(simmr_last_error(en));
    engs.push_back(en);
  }
  simmr_engine* const eng = engs[0];

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
      // every further engine stages its own copy of the reference (replicated, as across ranks: DESIGN.md section 5)
      for (size_t k = 1; rc == 0 && k < engs.size(); k++) {
        Genome again;
        if (load_genome_device(engs[k], (uint32_t)genomes.size(), rec.filepath, args.contiguous, device_min_size, &again, &err) != 0)
          return die("Failed to stage " + rec.filepath + " on device " + std::to_string(ordinals[k]) + ": " + err);
      }
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
    for (simmr_engine* en : engs)
      if (simmr_stage_genome(en, (uint32_t)gi, (uint32_t)ptrs.size(), ptrs.data(), lens.data(), sizes.data()) != SIMMR_OK)
        return die(std::string("staging failed: ") + simmr_last_error(en));
  }

  // abundances (simulate.rs:121-132 / :334-343)
  const bool is_long = eprofile->is_long_read();
  Abundances ab = aprofile->determine_abundances(args.num_reads, genomes.size());
  if (aprofile->is_size_aware()) ab = aprofile->adjust_for_size(genomes, ab, is_long ? 20000 : args.read_length, !is_long);

  // main.rs:191-198: remove previous outputs
  if (is_regular_file(args.output)) remove(args.output.c_str());  // (a pipe or a device given as the output is written to, not replaced)
  const std::string meta_path = args.output + ".tsv";
  if (exists(meta_path)) remove(meta_path.c_str());

  simmr_error_profile pod = eprofile->pod();
  if (args.rng_philox) {  // (extension) the counter mode, for the profiles that draw per base from a parametric law
    // (a custom model draws base by base only in the k-mer splice of its long-read path: include/simmr_hip.h)
    if (pod.kind == SIMMR_CUSTOM && !is_long)
      return die("--rng philox is not defined for custom-short: its qualities are not drawn base by base and it edits no bases");
    if (pod.kind != SIMMR_PERFECT_SHORT) pod.rng_mode = SIMMR_RNG_PHILOX;  // (perfect-short draws nothing per base)
    if (args.rng_philox_full) {  // the plan from counters too (the library says which profiles it serves)
      if (pod.kind == SIMMR_PERFECT_SHORT || pod.kind == SIMMR_CUSTOM)
        return die("--rng philox-full covers minimal-short, minimal-long and perfect-long");
      pod.rng_mode = SIMMR_RNG_PHILOX_FULL;
    }
  }
  const int has_seed = args.seed ? 1 : 0;
  const uint64_t seed = args.seed.value_or(0);

  const uint64_t chunk_reads = args.device_chunk_reads;
  auto run = [&](const Scope& sc, uint64_t chunk_units) {
    const uint64_t text_per_unit = sc.paired ? 2 * (2 * (uint64_t)args.read_length + 4 + 160) : 2 * 24000 + 260;  // (as the range sizes below)
    return engs.size() > 1 ? run_scope_devices(engs, ordinals, args, genomes, sc, chunk_units, text_per_unit) : run_scope(eng, args, genomes, sc, chunk_units);
  };
  if (!is_long) {
    info("Simulating short reads");
    const uint64_t text_per_pair = 2 * (2 * (uint64_t)args.read_length + 4 + 160);
    const uint64_t chunk_units = chunk_reads ? std::max<uint64_t>(chunk_reads / 2, 1) : auto_chunk_units(text_per_pair);
    // all genomes in one device plan (simulate_pe_reads, simulate.rs:110-150); the library leaves
    // custom profiles to the genome-by-genome loop below
    std::vector<uint32_t> all_idx(genomes.size());
    std::vector<uint64_t> all_reads(genomes.size());
    for (size_t gi = 0; gi < genomes.size(); gi++) { all_idx[gi] = (uint32_t)gi; all_reads[gi] = ab[gi].first; }
    simmr_plan_info probe{};
    const int mrc = simmr_pe_plan_multi(eng, (uint32_t)genomes.size(), all_idx.data(), all_reads.data(), &pod, has_seed, seed,
                                        simmr_range{0, 0}, &probe);  // (an empty range: does the library take this profile in one plan?)
    if (mrc != SIMMR_OK && mrc != SIMMR_ENOTSUP) return die(simmr_last_error(eng));
    // without --seed every plan call would draw its own seed (simulate.rs:174): draw the run's here, so that the ranges
    // of one genome continue one stream
    const uint64_t run_seed = has_seed ? seed : probe.seed_used;
    if (mrc == SIMMR_OK) {
      Scope sc{true, 0, genomes.size(), {}, 0, nullptr, nullptr};
      for (size_t gi = 0; gi < genomes.size(); gi++) sc.genome_units.push_back(ab[gi].first / 2);  // simulate.rs:179
      sc.plan = [&](simmr_engine* en, simmr_range rg, simmr_plan_info* pi) {
        return simmr_pe_plan_multi(en, (uint32_t)genomes.size(), all_idx.data(), all_reads.data(), &pod, 1, run_seed, rg, pi);
      };
      sc.emit = [&](simmr_engine* en, uint32_t idb, const simmr_reads_out* o) { return simmr_pe_emit(en, idb, o); };
      if (int rc = run(sc, chunk_units)) return rc;
    }
    uint32_t id_base = 0;  // the global AtomicU32 of simulate.rs:85-89
    for (size_t gi = 0; mrc == SIMMR_ENOTSUP && gi < genomes.size(); gi++) {
      // the reference draws a fresh entropy seed per genome when there is no --seed (simulate.rs:174)
      uint64_t g_seed = seed;
      if (!has_seed) {
        simmr_plan_info p0{};
        if (simmr_pe_plan(eng, (uint32_t)gi, &pod, ab[gi].first, 0, 0, simmr_range{0, 0}, &p0) != SIMMR_OK) return die(simmr_last_error(eng));
        g_seed = p0.seed_used;
      }
      Scope sc{true, gi, gi + 1, {ab[gi].first / 2}, id_base, nullptr, nullptr};
      sc.plan = [&, gi, g_seed](simmr_engine* en, simmr_range rg, simmr_plan_info* pi) { return simmr_pe_plan(en, (uint32_t)gi, &pod, ab[gi].first, 1, g_seed, rg, pi); };
      sc.emit = [&](simmr_engine* en, uint32_t idb, const simmr_reads_out* o) { return simmr_pe_emit(en, idb, o); };
      if (int rc = run(sc, chunk_units)) return rc;
      id_base += (uint32_t)(ab[gi].first / 2);
    }
  } else {
    info("Simulating long reads");
    std::vector<uint32_t> idx(genomes.size());
    std::vector<uint64_t> reads(genomes.size());
    for (size_t gi = 0; gi < genomes.size(); gi++) { idx[gi] = (uint32_t)gi; reads[gi] = ab[gi].first; }
    uint64_t run_seed = seed;
    if (!has_seed) {
      simmr_plan_info p0{};
      if (simmr_long_plan(eng, (uint32_t)genomes.size(), idx.data(), reads.data(), &pod, 0, 0, simmr_range{0, 0}, &p0) != SIMMR_OK)
        return die(simmr_last_error(eng));
      run_seed = p0.seed_used;
    }
    // a long read is up to 65 535 bases (u16 lengths); the gamma profiles average 20 000 (minimal_long.rs:64-65)
    const uint64_t chunk_units = chunk_reads ? chunk_reads : auto_chunk_units(2 * 24000 + 260);
    Scope sc{false, 0, genomes.size(), reads, 0, nullptr, nullptr};
    sc.plan = [&](simmr_engine* en, simmr_range rg, simmr_plan_info* pi) {
      // (without --seed the library selects per-read lengths by itself; the ranges then share the seed drawn above)
      simmr_error_profile p = pod;
      if (!has_seed) p.length_mode = SIMMR_LEN_PER_READ;
      return simmr_long_plan(en, (uint32_t)genomes.size(), idx.data(), reads.data(), &p, 1, run_seed, rg, pi);
    };
    sc.emit = [&](simmr_engine* en, uint32_t idb, const simmr_reads_out* o) { return simmr_long_emit(en, idb, o); };
    if (int rc = run(sc, chunk_units)) return rc;
  }
  info(("Writing simulated reads to " + args.output).c_str());

  // main.rs:213-258
  std::vector<MetadataRow> rows;
  for (size_t gi = 0; gi < genomes.size(); gi++) rows.push_back({genomes[gi].uuid, genomes[gi].filepath, ab[gi].first, ab[gi].second});
  info(("Writing simulation metadata to " + meta_path).c_str());
  if (!write_metadata(rows, meta_path, &err)) fprintf(stderr, "ERROR simmr-hip: Failed to write metadata file: %s\n", err.c_str());
  for (simmr_engine* en : engs) simmr_engine_destroy(en);
  return 0;
}

int main(int argc, char** argv) { return run_main(argc, argv); }
