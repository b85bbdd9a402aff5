// classifier_stream.cpp - the device-ingest streaming path of the command line (the default for the non-extended CSV): batches of
// whole records go to the GPU as the bytes of the file and come back as the bytes of the CSV (mic_ingest_*); host threads only move
// bytes.  Ingest slots, the loaders' FASTQ stripper, and run_stream's three thread pools (loaders / device / writer).
#include "classifier_internal.hpp"

namespace mic {
using namespace detail;

// ---- device-ingest streaming ----------------------------------------------------------------------------------------
bool Classifier::device_ingest() const {
  return !opt_.extended && getenv("MIC_HOST_INGEST") == nullptr;
}

void Classifier::release_ingest() {
  for (mic_engine* e : engines_) mic_ingest_free(e);
  ingest_raw_.clear();
  ingest_bytes_ = ingest_workers_ = 0;
}

// slot size and number of slots of the streaming path for an input of total_bytes
void Classifier::ingest_geometry(size_t total_bytes, size_t& bytes, size_t& workers) const {
  // one worker (host thread + slot + stream) moves ~30 Mreads/s; eight saturate the link
  // slots: one per host thread and half as many again in the queues between the stages - at most 16 (6 GB of HBM and 2 GB of
  // pinned memory at the default slot size): more slots only deepen the queues
  workers = std::min<size_t>(std::max<size_t>(opt_.threads, 1), 48);
  workers += workers / 2;
  if (workers > 16) workers = 16;
  if (const char* env = getenv("MIC_INGEST_SLOTS")) { long v = atol(env); if (v >= 1 && v <= 96) workers = (size_t)v; }
  bytes = 64u << 20;
  if (const char* env = getenv("MIC_INGEST_MB")) { long v = atol(env); if (v >= 1 && v <= 128) bytes = (size_t)v << 20; }
  if (const char* env = getenv("MIC_INGEST_KB")) { long v = atol(env); if (v >= 4) bytes = (size_t)v << 10; }
  if (const char* env = getenv("MIC_INGEST_WORKERS")) { long v = atol(env); if (v >= 1 && v <= 64) workers = (size_t)v; }
  // small inputs: do not pin more than the input needs
  while (bytes > (1u << 20) && total_bytes / workers < bytes / 2) bytes /= 2;
  if (total_bytes < bytes) workers = 1;
}

void Classifier::ensure_ingest(size_t total_bytes) {
  size_t bytes = 0, workers = 0;
  ingest_geometry(total_bytes, bytes, workers);
  if (!ingest_raw_.empty() && bytes <= ingest_bytes_ && workers <= ingest_workers_) return;
  release_ingest();
  const size_t n_eng = engines_.size();
  std::vector<const char*> nm(names_.size());
  for (size_t t = 0; t < names_.size(); ++t) nm[t] = names_[t].c_str();
  ingest_raw_.resize(n_eng);
  for (size_t d = 0; d < n_eng; ++d) {
    const size_t slots = (workers + n_eng - 1 - d) / n_eng;
    if (!slots) continue;
    ingest_raw_[d].resize(slots);
    check(mic_ingest_alloc(engines_[d], slots, bytes, nm.data(), (uint32_t)names_.size(), 0, ingest_raw_[d].data()), "ingest slots");
  }
  ingest_bytes_ = bytes; ingest_workers_ = workers;
}

// FASTQ: copy the header and the sequence line of every four-line record, drop the '+' and the quality line (nothing
// reads them: CuCLARK_hh.hh:1496-1523 only steps over them).  `phase` = line of the record the input is in (0..3),
// carried across calls; returns the bytes written.
static size_t strip_fastq_scalar(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, unsigned& phase) {   // (size_t)-1: dst is full
  size_t pos = 0, w = 0;
  while (pos < n) {
    if (phase == 0) {   // common case: the record's four lines are all in this piece
      const uint8_t* a = (const uint8_t*)memchr(src + pos, '\n', n - pos);
      const uint8_t* b = a ? (const uint8_t*)memchr(a + 1, '\n', (size_t)(src + n - (a + 1))) : nullptr;
      const uint8_t* c = b ? (const uint8_t*)memchr(b + 1, '\n', (size_t)(src + n - (b + 1))) : nullptr;
      const uint8_t* d = c ? (const uint8_t*)memchr(c + 1, '\n', (size_t)(src + n - (c + 1))) : nullptr;
      if (d) {
        const size_t len = (size_t)(b + 1 - (src + pos));
        if (w + len > dst_cap) return (size_t)-1;
        memcpy(dst + w, src + pos, len);
        w += len;
        pos = (size_t)(d + 1 - src);
        continue;
      }
    }
    const uint8_t* nl = (const uint8_t*)memchr(src + pos, '\n', n - pos);
    const size_t end = nl ? (size_t)(nl - src) + 1 : n;
    if (phase < 2) { if (w + (end - pos) > dst_cap) return (size_t)-1; memcpy(dst + w, src + pos, end - pos); w += end - pos; }
    if (nl) phase = (phase + 1) & 3;
    pos = end;
  }
  return w;
}

// The same with AVX2: the line ends of 64 input bytes are two compares and two move-masks; only two of a record's four line
// ends do anything (the sequence line's end closes a span that is copied, the quality line's end opens the next one), and the
// span - header + sequence, ~165 bytes - is copied 32 bytes at a time.  Byte-identical to the scalar form for every input and
// every split of it into calls (tests/test_cli.py: --strip-fastq); 2-3 x its rate per thread, which is what the loaders of the
// streaming command line spend their time in.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,bmi,bmi2")))
static inline void copy_span_avx2(uint8_t* d_, const uint8_t* s_, size_t len) {
  size_t i = 0;
  for (; i + 32 <= len; i += 32) _mm256_storeu_si256((__m256i*)(d_ + i), _mm256_loadu_si256((const __m256i*)(s_ + i)));
  if (i < len) memcpy(d_ + i, s_ + i, len - i);
}
__attribute__((target("avx2,bmi,bmi2")))
static size_t strip_fastq_avx2(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, unsigned& phase) {
  const __m256i nl = _mm256_set1_epi8('\n');
  size_t w = 0;
  unsigned ph = phase;
  size_t open = ph < 2 ? 0 : (size_t)-1;      // start of the span being kept ((size_t)-1: inside the dropped lines)
#define copy(from, to) ((w + ((to) - (from)) > dst_cap) ? false : (copy_span_avx2(dst + w, src + (from), (to) - (from)), w += (to) - (from), true))
  size_t pos = 0;
  for (; pos + 64 <= n; pos += 64) {
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(src + pos)), nl));
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(src + pos + 32)), nl));
    uint64_t m = ((uint64_t)hi << 32) | lo;
    while (m) {
      const size_t e = pos + (size_t)__builtin_ctzll(m) + 1;      // one past the line end
      m &= m - 1;
      if (ph == 1) { if (!copy(open, e)) return (size_t)-1; open = (size_t)-1; }
      else if (ph == 3) open = e;
      ph = (ph + 1) & 3;
    }
  }
  for (; pos < n; ++pos) {
    if (src[pos] != '\n') continue;
    const size_t e = pos + 1;
    if (ph == 1) { if (!copy(open, e)) return (size_t)-1; open = (size_t)-1; }
    else if (ph == 3) open = e;
    ph = (ph + 1) & 3;
  }
  if (open != (size_t)-1 && open < n) { if (!copy(open, n)) return (size_t)-1; }     // a kept line that continues in the next call
#undef copy
  phase = ph;
  return w;
}
#endif

static size_t strip_fastq(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, unsigned& phase) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && !getenv("MIC_STRIP_SCALAR");
  if (avx2) return strip_fastq_avx2(src, n, dst, dst_cap, phase);
#endif
  return strip_fastq_scalar(src, n, dst, dst_cap, phase);
}

// test hook (cuCLARK --strip-fastq): the whole input through strip_fastq in pieces of `piece` bytes
std::string strip_fastq_text(const std::string& in, size_t piece, bool scalar, int reps) {
  std::string out(in.size() + 64, '\0');
  size_t w = 0;
  for (int rep = 0; rep < reps; ++rep) {      // (reps > 1: timing runs over the same buffers)
    unsigned phase = 0;
    w = 0;
    for (size_t o = 0; o < in.size(); o += piece) {
      const size_t n = std::min(piece, in.size() - o);
      const size_t got = scalar ? strip_fastq_scalar((const uint8_t*)in.data() + o, n, (uint8_t*)out.data() + w, out.size() - w, phase)
                                : strip_fastq((const uint8_t*)in.data() + o, n, (uint8_t*)out.data() + w, out.size() - w, phase);
      if (got == (size_t)-1) throw std::runtime_error("strip_fastq: destination full");
      w += got;
    }
  }
  out.resize(w);
  return out;
}

// test hook (cuCLARK --strip-fastq <file> - <chunk> loaders <threads> [mmap]): what the loaders of run_stream do with a plain FASTQ
// file, without the device: ranges of 32 MiB dealt to `threads`, each range read in chunks of `chunk` bytes (pread into a
// stage buffer, or straight out of a mapping) and stripped into a slot-sized buffer.  Returns GB/s of input.
double strip_fastq_loaders_rate(const std::string& path, size_t chunk, unsigned threads, bool use_mmap) {
  const int fd = open(path.c_str(), O_RDONLY);
  struct stat st;
  if (fd == -1 || fstat(fd, &st) != 0) throw std::runtime_error("cannot open " + path);
  const size_t size = (size_t)st.st_size, RANGE = (size_t)32 << 20;
  const uint8_t* map = nullptr;
  if (use_mmap) {
    map = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) throw std::runtime_error("mmap failed");
  }
  std::atomic<size_t> next{0};
  struct timeval a, b;
  gettimeofday(&a, nullptr);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      std::vector<uint8_t> stage(chunk), dst(RANGE + 64);
      for (;;) {
        const size_t o = next.fetch_add(RANGE);
        if (o >= size) break;
        const size_t len = std::min(RANGE, size - o);
        unsigned phase = 0; size_t w = 0;       // (ranges are not cut at records here: the phase only has to be carried inside one)
        for (size_t c = 0; c < len; c += chunk) {
          const size_t n = std::min(chunk, len - c);
          const uint8_t* src;
          if (map) src = map + o + c;
          else { if (pread(fd, stage.data(), n, (off_t)(o + c)) != (ssize_t)n) return; src = stage.data(); }
          const size_t got = strip_fastq(src, n, dst.data() + w, dst.size() - w, phase);
          if (got == (size_t)-1) return;
          w += got;
        }
      }
    });
  for (auto& th : pool) th.join();
  gettimeofday(&b, nullptr);
  if (map) munmap((void*)map, size);
  close(fd);
  return (double)size / ((b.tv_sec - a.tv_sec) + (b.tv_usec - a.tv_usec) * 1e-6) / 1e9;
}

bool Classifier::run_stream(Feeder& feed, const std::string& results_base, bool paired, size_t total_bytes) {
  const std::string csv = results_base + ".csv";  // CuCLARK_hh.hh:539-540
  // a fresh file, not a truncated one: ext4 writes a truncated-and-rewritten file's blocks out when it is closed
  // (auto_da_alloc), 60 ms for the CSV of 16 M reads
  unlink(csv.c_str());
  const int out_fd = open(csv.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
  if (out_fd == -1) { std::cerr << "Failed to create/open file result: " << csv << std::endl; return true; }
  struct timeval t0, t1;
  gettimeofday(&t0, nullptr);
  // The CSV's blocks are allocated up front (a quarter of the input's size: 42 bytes of CSV per ~165 - 330 bytes of record; cut
  // to size at the end): the one writer thread is the slowest stage of the pipeline once it falls behind (it
  // writes back to back from the first late batch on), and a buffered write into allocated blocks is ~15 % cheaper than one that
  // reserves them page by page - 52-62 -> 46-52 ms for 10 M reads, which is where the run without any write ends
  // (MIC_CSV_DISCARD: 45-49 ms).  A file system that refuses the call is written as before; MIC_CSV_FALLOCATE=0 turns it off.
  uint64_t prealloc = total_bytes < ((size_t)1 << 40) ? std::min<uint64_t>((uint64_t)total_bytes / 4, (uint64_t)16 << 30) : 0;
  if (const char* env = getenv("MIC_CSV_FALLOCATE")) { long v = atol(env); prealloc = v > 0 && total_bytes < ((size_t)1 << 40) ? (uint64_t)total_bytes / 100 * (uint64_t)v : 0; }
  if (prealloc < ((uint64_t)1 << 20) || fallocate(out_fd, 0, 0, (off_t)prealloc) != 0) prealloc = 0;
  ensure_ingest(total_bytes);          // inside the timed region, like the reference's CuClarkDB::malloc (CuCLARK_hh.hh:1600-1606)
  std::atomic<uint64_t> ts_first_loaded{0}, ts_last_loaded{0}, ts_last_dev{0}, ts_alloc{0}, ts_last_write{0}, us_write_max{0};   // MIC_CLI_TIMING: stage ends since t0
  const uint64_t t0_us = (uint64_t)t0.tv_sec * 1000000u + (uint64_t)t0.tv_usec;
  { struct timeval t; gettimeofday(&t, nullptr); ts_alloc = (uint64_t)t.tv_sec * 1000000u + (uint64_t)t.tv_usec; }
  n_objects_ = 0;
  uint64_t out_off = 0;
  {  // header (CuCLARK_hh.hh:1957-1972)
    std::vector<const char*> nm(names_.size());
    for (size_t t = 0; t < names_.size(); ++t) nm[t] = names_[t].c_str();
    char hb[512];
    const int w = mic_csv_header(hb, sizeof(hb), 0, nm.data(), (uint32_t)names_.size());
    if (w > 0 && pwrite(out_fd, hb, (size_t)w, 0) == w) out_off = (uint64_t)w;
  }
  // Three pools around a set of slots (pinned input + pinned CSV + device buffers each):
  //   loaders   file / memory -> the slot's pinned input, FASTQ without its '+' and quality lines
  //   device    mic_ingest_classify (blocking: H2D, kernels, D2H); a batch the device hands back goes through the host path
  //   writers   CSV text -> file at the offset the batch's turn gives it
  // Batches are numbered when their range is assigned; offsets in the CSV are handed out in that order.
  const size_t n_eng = engines_.size(), cap = ingest_bytes_;
  struct SlotRef { size_t eng, slot; uint8_t* raw; };
  std::vector<SlotRef> slots;
  for (size_t d = 0; d < n_eng; ++d)
    for (size_t i = 0; i < ingest_raw_[d].size(); ++i) slots.push_back({d, i, ingest_raw_[d][i]});
  const size_t S = slots.size();
  // threads: opt_.threads in all; a quarter of them drive the device, an eighth write, the rest load
  const size_t T = std::max<size_t>(opt_.threads, 1);
  // one writer: concurrent pwrite()s to one file take turns on the inode lock and come out slower than a single stream
  size_t ND = std::min<size_t>(6, std::max<size_t>(1, T / 4)), NW = 1;
  if (const char* env = getenv("MIC_INGEST_ND")) { long v = atol(env); if (v >= 1 && v <= 32) ND = (size_t)v; }
  if (const char* env = getenv("MIC_INGEST_NW")) { long v = atol(env); if (v >= 1 && v <= 32) NW = (size_t)v; }
  size_t NL = T > ND + NW ? T - ND - NW : 1;
  if (S == 1) { ND = NW = NL = 1; }
  const bool strip_ok = getenv("MIC_KEEP_QUALITY") == nullptr;
  const bool timing = getenv("MIC_CLI_TIMING") != nullptr;

  struct Item {
    size_t id = 0, slot = 0; Range r; size_t n = 0; int flags = 0; bool host = false;      // loader -> device
    const char* text = nullptr; size_t text_n = 0, reads = 0; uint64_t off = 0;             // device -> writer
    std::shared_ptr<std::string> own;                                                      // CSV of a host-path batch
    uint64_t ts[6] = {0, 0, 0, 0, 0, 0};   // MIC_CLI_TRACE: slot taken / loaded / device start / device end / write start / write end (us since start)
  };
  const bool trace = getenv("MIC_CLI_TRACE") != nullptr;
  std::vector<std::string> trace_lines;
  std::mutex mu;                       // queues, turn bookkeeping, error
  std::condition_variable cv_free, cv_loaded, cv_write;
  std::vector<size_t> free_slots;
  for (size_t i = 0; i < S; ++i) free_slots.push_back(S - 1 - i);
  std::deque<Item> loaded, to_write;
  std::map<size_t, Item> waiting;      // finished batches whose turn has not come
  size_t next_id = 0, next_out = 0, loaders_left = NL, device_left = ND;
  bool fed_all = false;
  std::string err;
  std::mutex feed_mu, host_mu;
  std::atomic<size_t> n_fallback{0}, n_batches{0};
  std::atomic<uint64_t> us_load{0}, us_dev{0}, us_write{0}, bytes_in{0}, bytes_h2d{0};
  auto now_us = [] { struct timeval t; gettimeofday(&t, nullptr); return (uint64_t)t.tv_sec * 1000000u + (uint64_t)t.tv_usec; };
  auto fail = [&](const std::string& what) {
    { std::lock_guard<std::mutex> lk(mu); if (err.empty()) err = what; }
    cv_free.notify_all();
  };

  auto loader = [&]() {
    mic_thread_bind_near_device(engines_[0], 1);     // the pinned slots sit on the device's socket
    std::vector<uint8_t> stage;
    for (;;) {
      Item it;
      {
        // the slot is taken BEFORE the batch gets its number: a later batch can then never hold the last free slot while an
        // earlier one, whose turn everybody waits for, has none
        std::lock_guard<std::mutex> lk(feed_mu);
        {
          std::unique_lock<std::mutex> lk2(mu);
          cv_free.wait(lk2, [&] { return !free_slots.empty() || fed_all || !err.empty(); });
          if (fed_all || !err.empty()) break;
          it.slot = free_slots.back(); free_slots.pop_back();
        }
        bool more = false;
        // FASTQ travels without its quality lines: about half the bytes of a range reach the slot
        const bool fq = strip_ok && feed.fastq();
        size_t want = fq ? cap + cap / 2 : cap - cap / 8;
        {
          // the first batches are small so that the device and the writer start early (a full batch takes a loader ~10 ms),
          // the last ones are cut so that the loaders finish together
          const double ramp = std::min(1.0, std::max(0.125, (double)(next_id + 1) / (2.0 * (double)NL)));
          const uint64_t left = feed.remaining();
          size_t w = (size_t)((double)want * ramp);
          if (left / NL < w) w = (size_t)(left / NL);
          want = std::max<size_t>(std::min(w, want), std::min<size_t>((size_t)1 << 20, want));   // (a floor above `want` would not fit the slot)
        }
        try { more = feed.assign(want, cap, it.r); } catch (const std::exception& ex) { fail(ex.what()); }
        if (!more) {
          { std::lock_guard<std::mutex> lk2(mu); fed_all = true; free_slots.push_back(it.slot); }
          cv_free.notify_all();
          break;
        }
        it.id = next_id++;
        it.flags = (paired ? MIC_INGEST_PAIRED : 0) | (fq ? MIC_INGEST_FASTQ_2LINE : 0);
      }
      if (trace) it.ts[0] = now_us() - t0_us;
      const uint64_t ta = timing ? now_us() : 0;
      try {
        uint8_t* dst = slots[it.slot].raw;
        if (it.flags & MIC_INGEST_FASTQ_2LINE) {
          unsigned phase = 0; size_t w = 0; bool fits = true;
          const uint8_t* mem = it.r.mem;
          const size_t CH = 1u << 18;     // the stage of a pread stays in the core's L2 (256 KiB: 77 GB/s with 12 loaders, 1 MiB: 58; tools/loader_rate.sh)
          for (size_t o = 0; o < it.r.len && fits; o += CH) {
            const size_t n = std::min(CH, it.r.len - o);
            const uint8_t* src = mem ? mem + o : nullptr;
            if (!src) { if (stage.size() < CH) stage.resize(CH); feed.read(it.r, o, stage.data(), n); src = stage.data(); }
            const size_t got = strip_fastq(src, n, dst + w, cap - w, phase);
            if (got == (size_t)-1) { fits = false; break; }
            w += got;
          }
          // a range that ends inside a record (file cut short) or does not fit goes through the host path as it is
          if (!fits || phase != 0) it.host = true;
          it.n = w;
        } else {
          size_t got;
          if (feed.resident()) { got = feed.fill_resident(it.r, engines_[slots[it.slot].eng], slots[it.slot].slot); it.flags |= feed.resident_flags(); }
          else got = feed.fill(it.r, dst, cap);
          if (got == (size_t)-1) it.host = true; else it.n = got;
        }
        if (timing) {
          const uint64_t tn = now_us();
          us_load += tn - ta; const bool res = (it.flags & MIC_INGEST_RESIDENT) != 0;
          bytes_in += res ? it.n : it.r.len + it.r.len2; if (!it.host && !res) bytes_h2d += it.n;
          uint64_t z = 0; ts_first_loaded.compare_exchange_strong(z, tn); ts_last_loaded = tn;
        }
      } catch (const std::exception& ex) { fail(ex.what()); it.host = true; it.n = 0; }
      if (trace) it.ts[1] = now_us() - t0_us;
      { std::lock_guard<std::mutex> lk(mu); loaded.push_back(std::move(it)); }
      cv_loaded.notify_one();
    }
    { std::lock_guard<std::mutex> lk(mu); --loaders_left; }
    cv_loaded.notify_all();
  };

  auto device = [&]() {
    mic_thread_bind_near_device(engines_[0], 1);
    for (;;) {
      Item it;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_loaded.wait(lk, [&] { return !loaded.empty() || loaders_left == 0; });
        if (loaded.empty()) break;
        it = std::move(loaded.front()); loaded.pop_front();
      }
      const uint64_t ta = timing ? now_us() : 0;
      if (trace) it.ts[2] = now_us() - t0_us;
      bool failed;
      { std::lock_guard<std::mutex> lk(mu); failed = !err.empty(); }
      try {
        if (!failed && !it.host) {
          mic_ingest_result res;
          // the slot's engine alone, or - table-sharded - its group of parts_ engines, each probing the batch against its part
          const size_t eng = slots[it.slot].eng;
          if (parts_ == 1) check(mic_ingest_classify(engines_[eng], slots[it.slot].slot, it.n, it.flags, &res), "device ingest");
          else check(mic_ingest_classify_group(engines_.data() + eng / parts_ * parts_, parts_, eng % parts_, slots[it.slot].slot, it.n, it.flags, &res),
                     "device ingest (table-sharded)");
          if (res.status == MIC_INGEST_OK) { it.text = res.csv; it.text_n = (size_t)res.csv_bytes; it.reads = (size_t)res.n_reads; }
          else it.host = true;
        }
        if (!failed && it.host) {   // the host indexer / packer / CSV writer on the ORIGINAL bytes of the range (rare: one at a time)
          std::string bytes;
          feed.text(it.r, bytes);
          it.own = std::make_shared<std::string>();
          std::lock_guard<std::mutex> lk(host_mu);
          ++n_fallback;
          sink_ = it.own.get();
          try { it.reads = process_segment((const uint8_t*)bytes.data(), bytes.size(), paired, nullptr); } catch (...) { sink_ = nullptr; throw; }
          sink_ = nullptr;
          it.text = it.own->data(); it.text_n = it.own->size();
        }
      } catch (const std::exception& ex) { fail(ex.what()); it.text_n = 0; it.reads = 0; }
      it.r.keep.reset();
      if (trace) it.ts[3] = now_us() - t0_us;
      if (timing) { const uint64_t tn = now_us(); us_dev += tn - ta; ts_last_dev = tn; }
      ++n_batches;
      {
        std::lock_guard<std::mutex> lk(mu);
        waiting.emplace(it.id, std::move(it));
        for (auto f = waiting.find(next_out); f != waiting.end(); f = waiting.find(next_out)) {   // whose turn has come
          f->second.off = out_off; out_off += f->second.text_n; n_objects_ += f->second.reads; ++next_out;
          to_write.push_back(std::move(f->second));
          waiting.erase(f);
        }
      }
      cv_write.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); --device_left; }
    cv_write.notify_all();
  };

  const bool discard_csv = getenv("MIC_CSV_DISCARD") != nullptr;
  auto writer = [&]() {
    mic_thread_bind_near_device(engines_[0], 1);
    for (;;) {
      Item it;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_write.wait(lk, [&] { return !to_write.empty() || device_left == 0; });
        if (to_write.empty()) break;
        it = std::move(to_write.front()); to_write.pop_front();
      }
      const uint64_t ta = timing ? now_us() : 0;
      if (trace) it.ts[4] = now_us() - t0_us;
      size_t done = discard_csv ? it.text_n : 0;      // MIC_CSV_DISCARD=1: a measuring run without the writes (what the other stages can do)
      while (done < it.text_n) {
        const ssize_t n = pwrite(out_fd, it.text + done, it.text_n - done, (off_t)(it.off + done));
        if (n <= 0) { fail("Failed to write the results file."); break; }
        done += (size_t)n;
      }
      if (timing) { const uint64_t tn = now_us(); us_write += tn - ta; ts_last_write = tn; if (tn - ta > us_write_max) us_write_max = tn - ta; }
      if (trace) {
        it.ts[5] = now_us() - t0_us;
        char ln[200];
        snprintf(ln, sizeof(ln), "[trace] batch %zu slot %zu bytes %zu: taken %llu loaded %llu dev %llu-%llu write %llu-%llu us", it.id, it.slot, it.n,
                 (unsigned long long)it.ts[0], (unsigned long long)it.ts[1], (unsigned long long)it.ts[2], (unsigned long long)it.ts[3],
                 (unsigned long long)it.ts[4], (unsigned long long)it.ts[5]);
        std::lock_guard<std::mutex> lk(mu);
        trace_lines.push_back(ln);
      }
      { std::lock_guard<std::mutex> lk(mu); free_slots.push_back(it.slot); }
      cv_free.notify_one();
    }
    mic_thread_bind_near_device(engines_[0], 0);
  };

  std::vector<std::thread> th;
  for (size_t i = 0; i < NL; ++i) th.emplace_back(loader);
  for (size_t i = 0; i < ND; ++i) th.emplace_back(device);
  for (size_t i = 1; i < NW; ++i) th.emplace_back(writer);
  writer();
  const uint64_t tj0 = now_us();
  for (auto& t : th) t.join();
  const uint64_t tj1 = now_us();
  if (prealloc && ftruncate(out_fd, (off_t)out_off) != 0 && err.empty()) err = "Failed to write the results file.";
  close(out_fd);
  const uint64_t tj2 = now_us();
  release_batches();
  const uint64_t tj3 = now_us();
  for (const std::string& ln : trace_lines) std::cerr << ln << "\n";
  if (timing) std::cerr << "[timing] teardown: join " << (tj1 - tj0) / 1e3 << " ms, close " << (tj2 - tj1) / 1e3 << " ms, batch buffers " << (tj3 - tj2) / 1e3 << " ms" << std::endl;
  if (feed.gave_up()) { unlink(csv.c_str()); return false; }
  if (!err.empty()) die(err);
  gettimeofday(&t1, nullptr);
  // (the time it took to inflate a compressed input up front belongs to the assignment time)
  const double diff = (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) / 1000000.0 + prelude_s_;
  if (timing) std::cerr << "[timing] device ingest: " << n_batches << " batches of <= " << (cap >> 10) << " KB on " << S
                        << " slot(s), " << n_fallback << " through the host path; threads: " << NL << " load, " << ND << " device, " << NW
                        << " write; thread-seconds: load " << us_load / 1e6 << ", device " << us_dev / 1e6 << ", write " << us_write / 1e6
                        << "; input " << bytes_in / 1e6 << " MB, over the link " << bytes_h2d / 1e6 << " MB; ms since start: slots ready "
                        << (ts_alloc - t0_us) / 1e3 << ", first batch loaded " << (ts_first_loaded - t0_us) / 1e3 << ", last loaded "
                        << (ts_last_loaded - t0_us) / 1e3 << ", last off the device " << (ts_last_dev - t0_us) / 1e3 << ", last write done " << (ts_last_write - t0_us) / 1e3
                        << " (longest " << us_write_max / 1e3 << "), end " << diff * 1e3 << std::endl;
  if (timing && parts_ > 1) {
    // MIC_GROUP_TIMING=1: HIP events on every engine's stream around the packed-read fan-out, the query kernel and the row exchange of
    // every batch (mic_ingest_group_stats), summed over the slots' owners
    double tot[MIC_GROUP_STATS_FIELDS] = {0};
    for (mic_engine* e : engines_) {
      double v[MIC_GROUP_STATS_FIELDS];
      if (mic_ingest_group_stats(e, v, MIC_GROUP_STATS_FIELDS) > 0) for (size_t i = 0; i < MIC_GROUP_STATS_FIELDS; ++i) tot[i] += v[i];
    }
    if (tot[0] > 0)
      std::cerr << "[timing] table-sharded batches: " << (uint64_t)tot[0] << " timed, " << (uint64_t)tot[1] << " reads, " << parts_ << " part(s); packed-read fan-out "
                << tot[2] / 1e6 << " MB, " << tot[3] << " ms summed over the helpers (slowest helper of each batch: " << tot[4] << " ms); query kernels "
                << tot[5] << " ms summed over the engines (slowest engine of each batch: " << tot[6] << " ms); row exchange " << tot[7] / 1e6 << " MB, "
                << tot[8] << " ms summed over the engines (slowest engine of each batch: " << tot[9] << " ms)" << std::endl;
  }
  std::cout << " - Assignment time: " << diff << " s. Speed: ";  // CuCLARK_hh.hh:1938-1944
  std::cout << (size_t)(((double)n_objects_) / (diff) * 60.0) << " objects/min. (" << n_objects_ << " objects)." << std::endl;
  std::cout << " - Results stored in " << csv << std::endl;
  return true;
}

}  // namespace mic
