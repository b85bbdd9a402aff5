// classifier_feeders.hpp - the inputs of the command line (see classifier_internal.hpp): segment sources and the feeders of the
// device-ingest streaming path.  Header-only; included by classifier.cpp (run / run_paired choose among them).
#ifndef MIC_CLASSIFIER_FEEDERS_HPP
#define MIC_CLASSIFIER_FEEDERS_HPP
#include "classifier_internal.hpp"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace mic {
namespace detail {

// ---- segment sources ------------------------------------------------------------------------------------------------

// plain file: zero-copy views of the mapping, cut at record starts
class MmapSource : public Classifier::SegmentSource {
 public:
  MmapSource(const std::string& path, size_t seg) : seg_(seg) {
    fd_ = open(path.c_str(), O_RDONLY);
    struct stat st;
    if (fd_ == -1 || fstat(fd_, &st) != 0 || st.st_size == 0) return;
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return;
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    map_ = (const uint8_t*)m; nb_ = (size_t)st.st_size;
  }
  ~MmapSource() override { if (map_) munmap((void*)map_, nb_); if (fd_ != -1) close(fd_); }
  bool ok() const { return map_ != nullptr; }
  bool next(Classifier::Segment& s) override {
    if (!map_ || pos_ >= nb_) return false;
    size_t end = nb_;
    if (nb_ - pos_ > seg_ + seg_ / 4) {
      end = mic_find_record_start(map_, nb_, pos_ + seg_);
      if (end <= pos_) end = nb_;
    }
    s.p = map_ + pos_; s.n = end - pos_; s.own.clear();
    // fault the segment in here (this runs on the side thread, ahead of the indexer's 32 threads taking the faults)
    {
      const uintptr_t a = (uintptr_t)(map_ + pos_) & ~(uintptr_t)4095, b = (uintptr_t)(map_ + end);
      bool done = false;
#ifdef MADV_POPULATE_READ
      done = madvise((void*)a, (size_t)(b - a), MADV_POPULATE_READ) == 0;
#endif
      if (!done) {
        unsigned sum = 0;
        for (uintptr_t q = a; q < b; q += 4096) sum += *(volatile const uint8_t*)q;
        (void)sum;
      }
    }
    pos_ = end;
    return true;
  }
 private:
  int fd_ = -1; const uint8_t* map_ = nullptr; size_t nb_ = 0, pos_ = 0, seg_;
};

// Decompressed bytes of a gzip (or plain) file, produced on a background thread so that inflating overlaps whatever
// the consumer does with the bytes (record splitting, the paired-end merge, the other file of a pair).  Block-gzip
// files (BGZF: every member carries its compressed size in a 'BC' extra field, as bgzip / samtools write them) are
// inflated block-parallel by a few threads; ordinary gzip is one zlib stream (~0.45 GB/s), plain files pass through.
// The reference leaves this to `gunzip` in classify_metagenome.sh:116-142.
class InflateStream {
 public:
  explicit InflateStream(const std::string& path, unsigned threads = 0) {
    const unsigned hw = pgz::usable_cpus();
    threads_ = threads ? threads : std::max(1u, std::min(8u, hw / 2));
    if (const char* env = getenv("MIC_INFLATE_THREADS")) { long v = atol(env); if (v >= 1 && v <= 64) threads_ = (unsigned)v; }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return;
    unsigned char h[18];
    size_t n = fread(h, 1, sizeof(h), f);
    bgzf_ = n == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' &&
            h[13] == 'C' && h[14] == 2 && h[15] == 0;
    if (bgzf_) { rewind(f); raw_ = f; }
    else {
      fclose(f);
      gz_ = gzopen(path.c_str(), "rb");
      if (!gz_) return;
      gzbuffer(gz_, 1 << 20);
    }
    ok_ = true;
    path_ = path;
    producer_ = std::thread([this] { bgzf_ ? produce_bgzf() : (threads_ > 1 && !getenv("MIC_SERIAL_GZIP") ? produce_gz_parallel() : produce_gz()); });
  }
  ~InflateStream() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_space_.notify_all();
    if (producer_.joinable()) producer_.join();
    if (gz_) gzclose(gz_);
    if (raw_) fclose(raw_);
  }
  InflateStream(const InflateStream&) = delete;
  InflateStream& operator=(const InflateStream&) = delete;
  bool ok() const { return ok_; }
  bool block_gzip() const { return bgzf_; }
  // like gzread: up to n bytes, 0 at the end of the data, -1 on a corrupt file
  long read(void* dst, size_t n) {
    size_t got = 0;
    char* d = (char*)dst;
    while (got < n) {
      if (pos_ == cur_.size()) {
        std::unique_lock<std::mutex> g(m_);
        cv_data_.wait(g, [&] { return !q_.empty() || done_; });
        if (q_.empty()) { if (failed_) return -1; break; }
        cur_.swap(q_.front()); q_.pop_front(); pos_ = 0;
        g.unlock();
        cv_space_.notify_one();
        continue;
      }
      const size_t take = std::min(n - got, cur_.size() - pos_);
      memcpy(d + got, cur_.data() + pos_, take);
      got += take; pos_ += take;
    }
    return (long)got;
  }

 private:
  bool push(std::vector<char>& chunk) {           // false: the consumer went away
    std::unique_lock<std::mutex> g(m_);
    cv_space_.wait(g, [&] { return q_.size() < 4 || stop_; });
    if (stop_) return false;
    q_.emplace_back(); q_.back().swap(chunk);
    g.unlock();
    cv_data_.notify_one();
    return true;
  }
  void finish(bool failed) {
    { std::lock_guard<std::mutex> g(m_); done_ = true; failed_ = failed; }
    cv_data_.notify_all();
  }
  void produce_gz() {
    for (;;) {
      std::vector<char> chunk(8u << 20);
      int n = gzread(gz_, chunk.data(), (unsigned)chunk.size());
      if (n <= 0) { finish(n < 0); return; }
      chunk.resize((size_t)n);
      if (!push(chunk)) return;
    }
  }
  // ordinary gzip, inflated by threads_ threads at once (pgz.hpp); anything it cannot map falls back to the zlib stream
  void produce_gz_parallel() {
    int fd = open(path_.c_str(), O_RDONLY);
    struct stat st;
    if (fd == -1 || fstat(fd, &st) != 0 || st.st_size < 18) { if (fd != -1) close(fd); produce_gz(); return; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { produce_gz(); return; }
    if (((const uint8_t*)m)[0] != 0x1f || ((const uint8_t*)m)[1] != 0x8b) {       // a plain file: zlib passes it through
      munmap(m, (size_t)st.st_size);
      produce_gz();
      return;
    }
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    bool stopped = false;
    auto sink = pgz::piece_sink([&](pgz::Bytes&& b) {
      std::vector<char> chunk((const char*)b.p, (const char*)b.p + b.n);
      if (!push(chunk)) { stopped = true; return false; }
      return true;
    });
    const int rc = pgz::inflate_all((const uint8_t*)m, (size_t)st.st_size, threads_, (size_t)512 << 10, sink);
    munmap(m, (size_t)st.st_size);
    if (!stopped) finish(rc != 0);
  }
  void produce_bgzf() {
    struct Blk { size_t off, csize, isize, out; };
    std::vector<unsigned char> in;
    for (;;) {
      in.clear();
      std::vector<Blk> blks;
      size_t out_total = 0;
      while (blks.size() < 512) {                  // <= 32 MB of output per batch
        unsigned char h[18];
        size_t n = fread(h, 1, 18, raw_);
        if (n == 0) break;
        if (n != 18 || h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C') { finish(true); return; }
        const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;
        if (bsize < 26) { finish(true); return; }
        const size_t off = in.size();
        in.resize(off + bsize);
        memcpy(in.data() + off, h, 18);
        if (fread(in.data() + off + 18, 1, bsize - 18, raw_) != bsize - 18) { finish(true); return; }
        const unsigned char* t = in.data() + off + bsize - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        if (isize > 65536) { finish(true); return; }
        blks.push_back({off, bsize, isize, out_total});
        out_total += isize;
      }
      if (blks.empty()) { finish(false); return; }
      std::vector<char> chunk(out_total);
      std::atomic<bool> bad{false};
      auto work = [&](unsigned t0) {
        for (size_t b = t0; b < blks.size(); b += threads_) {
          const Blk& k = blks[b];
          if (k.isize == 0) continue;
          z_stream zs; memset(&zs, 0, sizeof(zs));
          if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
          zs.next_in = in.data() + k.off + 18; zs.avail_in = (uInt)(k.csize - 18 - 8);
          zs.next_out = (Bytef*)chunk.data() + k.out; zs.avail_out = (uInt)k.isize;
          const int rc = inflate(&zs, Z_FINISH);
          if (rc != Z_STREAM_END || zs.avail_out != 0) bad = true;
          inflateEnd(&zs);
          const unsigned char* c = in.data() + k.off + k.csize - 8;
          const uLong want = (uLong)c[0] | ((uLong)c[1] << 8) | ((uLong)c[2] << 16) | ((uLong)c[3] << 24);
          if (crc32(crc32(0L, Z_NULL, 0), (const Bytef*)chunk.data() + k.out, (uInt)k.isize) != want) bad = true;
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < threads_ && t < blks.size(); ++t) pool.emplace_back(work, t);
      work(0);
      for (auto& th : pool) th.join();
      if (bad) { finish(true); return; }
      if (!chunk.empty() && !push(chunk)) return;
    }
  }

  bool ok_ = false, bgzf_ = false;
  unsigned threads_ = 1;
  std::string path_;
  gzFile gz_ = nullptr; FILE* raw_ = nullptr;
  std::thread producer_;
  std::mutex m_; std::condition_variable cv_data_, cv_space_;
  std::deque<std::vector<char>> q_;
  bool done_ = false, failed_ = false, stop_ = false;
  std::vector<char> cur_; size_t pos_ = 0;
};


// gzip (or plain) file through zlib: inflate ~seg bytes, keep the incomplete last record for the next segment
class GzSource : public Classifier::SegmentSource {
 public:
  GzSource(const std::string& path, size_t seg) : in_(path), seg_(seg) {}
  bool ok() const { return in_.ok(); }
  bool next(Classifier::Segment& s) override {
    if (!in_.ok() || (eof_ && carry_.empty())) return false;
    std::string buf;
    buf.swap(carry_);
    size_t want = seg_;
    for (;;) {
      while (!eof_ && buf.size() < want) {
        size_t old = buf.size();
        buf.resize(old + (8u << 20));
        long n = in_.read(&buf[old], 8u << 20);
        buf.resize(old + (n > 0 ? (size_t)n : 0));
        if (n < 0) die("Failed to uncompress input objects.");
        if (n <= 0) eof_ = true;
      }
      if (eof_) break;
      // last record start in the buffer: everything from there on is carried over
      const uint8_t* b = (const uint8_t*)buf.data();
      size_t last = 0, from = buf.size() > (1u << 20) ? buf.size() - (1u << 20) : 1;
      for (;;) {
        size_t p = mic_find_record_start(b, buf.size(), from);
        size_t q = p;
        while (q < buf.size()) { last = q; q = mic_find_record_start(b, buf.size(), q + 1); }
        if (last > 0 || from <= 1) break;
        from = from > (8u << 20) ? from - (8u << 20) : 1;   // records longer than the window: look further back
      }
      if (last > 0) { carry_.assign(buf, last, std::string::npos); buf.resize(last); break; }
      want = buf.size() * 2;   // one record larger than the segment: keep reading
    }
    if (buf.empty()) return false;
    s.own.swap(buf); s.p = (const uint8_t*)s.own.data(); s.n = s.own.size();
    return true;
  }
 private:
  InflateStream in_; std::string carry_; bool eof_ = false; size_t seg_;
};

// ---- compressed input, inflated up front ---------------------------------------------------------------------------------
// The reference's script copies a .gz input, gunzips the copy and classifies the plain file (classify_metagenome.sh:116-142).
// The same here, in memory: the file is inflated by many threads at once (pgz.hpp; block gzip block-parallel) straight into
// an anonymous memory file (memfd), and the plain-file path then runs on that file: its loaders cut, strip and - for a pair of
// files - merge in parallel, which no reader of an inflate stream can.  Only when the inflated text would not fit in half of
// the available memory does the input stay a stream (GzSource / PairedSource over InflateStream).
class InflatedFile {
 public:
  ~InflatedFile() { if (fd_ != -1) close(fd_); }
  int fd() const { return fd_; }
  uint64_t size() const { return size_; }
  std::string path() const { return "/proc/self/fd/" + std::to_string(fd_); }
  // 0: inflated; 1: not attempted (does not fit in memory, or MIC_GZ_STREAM); -1: the file is damaged
  int inflate(const std::string& src, unsigned threads) {
    if (getenv("MIC_GZ_STREAM")) return 1;
    int in = open(src.c_str(), O_RDONLY);
    struct stat st;
    if (in == -1 || fstat(in, &st) != 0 || st.st_size < 18) { if (in != -1) close(in); return 1; }
    {  // room for the text?  (deflate of sequence data: 3 - 6 x; 10 x to be safe)
      uint64_t avail_kb = 0;
      if (FILE* f = fopen("/proc/meminfo", "r")) {
        char line[128];
        while (fgets(line, sizeof(line), f)) if (sscanf(line, "MemAvailable: %llu kB", (unsigned long long*)&avail_kb) == 1) break;
        fclose(f);
      }
      if (avail_kb && (uint64_t)st.st_size * 10 > avail_kb * 1024 / 2) { close(in); return 1; }
    }
    fd_ = memfd_create("mic_inflated", MFD_CLOEXEC);
    if (fd_ == -1) { close(in); return 1; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, in, 0);
    close(in);
    if (m == MAP_FAILED) { close(fd_); fd_ = -1; return 1; }
    const uint8_t* h = (const uint8_t*)m;
    const bool bgzf = h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C';
    int rc;
    if (!bgzf) {
      struct FdSink {
        int fd; uint64_t size = 0; uint8_t* map = nullptr; size_t map_len = 0;
        uint8_t* reserve(size_t n) {
          const uint64_t a = size & ~(uint64_t)4095;
          map_len = (size_t)(size - a) + n;
          if (ftruncate(fd, (off_t)(size + n)) != 0) return nullptr;
          void* p = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a);
          if (p == MAP_FAILED) return nullptr;
          map = (uint8_t*)p;
          return map + (size - a);
        }
        bool commit(size_t n) { munmap(map, map_len); size += n; return true; }
      } sink{fd_};
      rc = pgz::inflate_all(h, (size_t)st.st_size, threads, (size_t)1 << 20, sink);
      size_ = sink.size;
      munmap(m, (size_t)st.st_size);
      if (rc == 1) { close(fd_); fd_ = -1; return 1; }           // out of memory for the text: stream instead
    } else {
      munmap(m, (size_t)st.st_size);
      InflateStream is(src, threads);
      std::vector<char> buf((size_t)16 << 20);
      rc = 0;
      for (;;) {
        const long n = is.read(buf.data(), buf.size());
        if (n < 0) { rc = -1; break; }
        if (n == 0) break;
        size_t w = 0;
        while (w < (size_t)n) { const ssize_t k = write(fd_, buf.data() + w, (size_t)n - w); if (k <= 0) { rc = 1; break; } w += (size_t)k; }
        if (rc) break;
        size_ += (uint64_t)n;
      }
      if (rc == 1) { close(fd_); fd_ = -1; return 1; }
    }
    return rc == 0 ? 0 : -1;
  }
 private:
  int fd_ = -1; uint64_t size_ = 0;
};

inline unsigned inflate_threads(size_t cli_threads, unsigned files) {
  // all the CPUs the process may use (the cgroup's quota, not the host's thread count), at least what -n asks for
  const unsigned hw = pgz::usable_cpus();
  unsigned t = std::max<unsigned>((unsigned)cli_threads, std::min(hw, 64u));
  if (const char* env = getenv("MIC_INFLATE_THREADS")) { long v = atol(env); if (v >= 1 && v <= 256) t = (unsigned)v; }
  return std::max(1u, t / std::max(1u, files));
}

// line reader over zlib (plain files are read transparently)
class GzLines {
 public:
  explicit GzLines(const std::string& path) : in_(path), buf_(1 << 20) {}
  bool ok() const { return in_.ok(); }
  bool line(std::string& out) {   // getLineFromFile semantics: strip one trailing '\n' (file.cc:124-141)
    out.clear();
    for (;;) {
      if (pos_ == len_) {
        if (!in_.ok()) return !out.empty();
        long n = in_.read(buf_.data(), buf_.size());
        if (n < 0) die("Failed to uncompress input objects.");
        if (n <= 0) return !out.empty() || false;
        pos_ = 0; len_ = (size_t)n;
      }
      const char* b = buf_.data() + pos_;
      const char* nl = (const char*)memchr(b, '\n', len_ - pos_);
      if (nl) { out.append(b, (size_t)(nl - b)); pos_ += (size_t)(nl - b) + 1; return true; }
      out.append(b, len_ - pos_); pos_ = len_;
    }
  }
 private:
  InflateStream in_; std::vector<char> buf_; size_t pos_ = 0, len_ = 0;
};

// paired-end FASTQ -> segments of the merged FASTA text ">id\nseq1Nseq2\n" (file.cc:205-268)
class PairedSource : public Classifier::SegmentSource {
 public:
  PairedSource(const std::string& f1, const std::string& f2, size_t seg) : a_(f1), b_(f2), seg_(seg) {}
  bool ok() const { return a_.ok() && b_.ok(); }
  bool next(Classifier::Segment& s) override {
    if (done_) return false;
    std::string out;
    out.reserve(std::min<size_t>(seg_, (size_t)64 << 20) + (1u << 16));
    std::string l1, l2;
    const std::string seps = " /\t@";
    while (out.size() < seg_) {
      if (!(a_.line(l1) && b_.line(l2))) { done_ = true; break; }
      if (first_) {
        first_ = false;
        if (l1.empty() || l2.empty() || l1[0] != l2[0]) die("Error: the files have different format!");
        if (l1[0] != '@') die("Error: paired-end reads must be FASTQ files!");
      }
      if (l1.empty() || l2.empty() || l1[0] != '@' || l2[0] != '@') continue;
      std::vector<std::string> e1 = split_seps(l1, seps), e2 = split_seps(l2, seps);
      if (e1.empty() || e2.empty() || e1[0] != e2[0]) die("Error: read id does not match between files!");
      out += ">"; out += e1[0]; out += "\n";
      if (!(a_.line(l1) && b_.line(l2))) die("Error: Found read without sequence");
      out += l1; out += "N"; out += l2; out += "\n";   // NBN = 1 separator (parameters.hh:41)
      if (a_.line(l1) && b_.line(l2)) { a_.line(l1); b_.line(l2); }
    }
    if (out.empty()) return false;
    s.own.swap(out); s.p = (const uint8_t*)s.own.data(); s.n = s.own.size();
    return true;
  }
 private:
  GzLines a_, b_; size_t seg_; bool done_ = false, first_ = true;
};

class OneBuffer : public Classifier::SegmentSource {
 public:
  OneBuffer(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  bool next(Classifier::Segment& s) override { if (!p_) return false; s.p = p_; s.n = n_; s.own.clear(); p_ = nullptr; return true; }
 private:
  const uint8_t* p_; size_t n_;
};


// ---- feeders of the device-ingest streaming path (Classifier::run_stream) ------------------------------------------

// plain file: ranges are cut at record starts found in small windows read with pread; the bytes of a range go straight
// from the page cache into the slot's pinned buffer (no mapping, no page faults)
class FileFeeder : public Classifier::Feeder {
 public:
  explicit FileFeeder(const std::string& path) {
    fd_ = open(path.c_str(), O_RDONLY);
    struct stat st;
    if (fd_ == -1 || fstat(fd_, &st) != 0 || st.st_size == 0) return;
    size_ = (uint64_t)st.st_size;
    uint8_t c = 0;
    if (pread(fd_, &c, 1, 0) != 1) return;
    first_ = c;
    ok_ = true;
  }
  ~FileFeeder() override { if (fd_ != -1) close(fd_); }
  bool ok() const { return ok_; }
  uint64_t size() const { return size_; }
  uint8_t first_byte() const { return first_; }
  bool fastq() const override { return first_ == '@'; }
  uint64_t remaining() const override { return size_ - pos_; }
  bool assign(size_t want, size_t cap, Classifier::Range& r) override {
    (void)cap;
    if (pos_ >= size_) return false;
    uint64_t end = size_;
    if (size_ - pos_ > want + want / 8) {
      // first record start at or after pos_ + want: look in growing windows
      const bool fasta = first_ == '>';
      uint64_t from = pos_ + want;
      size_t win = 1u << 16;
      for (;;) {
        const uint64_t w0 = from - 1, w1 = std::min<uint64_t>(size_, w0 + win);
        buf_.resize((size_t)(w1 - w0));
        if (pread(fd_, buf_.data(), buf_.size(), (off_t)w0) != (ssize_t)buf_.size()) die("Failed to read the objects file.");
        const size_t p = mic_find_record_start_in(buf_.data(), buf_.size(), fasta ? 1 : 0, 1);
        if (p < buf_.size()) { end = w0 + p; break; }
        if (w1 == size_) { end = size_; break; }
        win *= 4;
      }
    }
    r.off = pos_; r.len = (size_t)(end - pos_); r.mem = nullptr; r.keep.reset();
    pos_ = end;
    return true;
  }
  void read(const Classifier::Range& r, size_t off, uint8_t* dst, size_t len) override {
    size_t got = 0;
    while (got < len) {
      const ssize_t n = pread(fd_, dst + got, len - got, (off_t)(r.off + off + got));
      if (n <= 0) die("Failed to read the objects file.");
      got += (size_t)n;
    }
  }
 private:
  int fd_ = -1; uint64_t size_ = 0, pos_ = 0; uint8_t first_ = 0; bool ok_ = false;
  std::vector<uint8_t> buf_;
};

// segments of whole records in memory (inflated gzip, merged paired-end text): ranges are slices of the segments
class SegmentFeeder : public Classifier::Feeder {
 public:
  explicit SegmentFeeder(Classifier::SegmentSource& src) : src_(src) {}
  bool fastq() const override { return cur_ && cur_->n && cur_->p[0] == '@'; }
  bool assign(size_t want, size_t cap, Classifier::Range& r) override {
    (void)cap;
    if (!cur_ || pos_ >= cur_->n) {
      auto s = std::make_shared<Classifier::Segment>();
      if (!src_.next(*s)) return false;
      if (!s->own.empty()) s->p = (const uint8_t*)s->own.data();
      cur_ = s; pos_ = 0;
    }
    size_t end = cur_->n;
    if (cur_->n - pos_ > want + want / 8) {
      const size_t p = mic_find_record_start_in(cur_->p, cur_->n, cur_->p[0] == '>' ? 1 : 0, pos_ + want);
      if (p > pos_ && p < cur_->n) end = p;
    }
    r.off = pos_; r.len = end - pos_; r.mem = cur_->p + pos_; r.keep = cur_;
    pos_ = end;
    return true;
  }
  void read(const Classifier::Range& r, size_t off, uint8_t* dst, size_t len) override { memcpy(dst, r.mem + off, len); }
 private:
  Classifier::SegmentSource& src_;
  std::shared_ptr<Classifier::Segment> cur_;
  size_t pos_ = 0;
};

// newline count of a buffer; the AVX2 variant is picked at run time
inline size_t count_newlines_plain(const uint8_t* p, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
  return c;
}
__attribute__((target("avx2"))) inline size_t count_newlines_avx2(const uint8_t* p, size_t n) {
  const __m256i nl = _mm256_set1_epi8('\n');
  size_t c = 0, i = 0;
  for (; i + 128 <= n; i += 128) {
    const unsigned m0 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nl));
    const unsigned m1 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 32)), nl));
    const unsigned m2 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 64)), nl));
    const unsigned m3 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 96)), nl));
    c += (size_t)__builtin_popcountll(((unsigned long long)m1 << 32) | m0) + (size_t)__builtin_popcountll(((unsigned long long)m3 << 32) | m2);
  }
  for (; i < n; ++i) c += p[i] == '\n';
  return c;
}
inline size_t count_newlines(const uint8_t* p, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  return avx2 ? count_newlines_avx2(p, n) : count_newlines_plain(p, n);
}

// Two plain FASTQ files of a paired-end run, merged by the loaders in parallel.  The reference merges the pair line by
// line into a temporary FASTA file (file.cc:205-268: ">id\nseq1Nseq2\n") and classifies that file.  Here a first pass
// counts the line ends of both files in 1-MB pieces on all threads, which tells where record r starts in either file;
// each loader then writes the merged text of its batch's records straight into its slot.  Whatever the line arithmetic
// does not cover (line counts that differ or are no multiple of four, a header line without '@', ids that differ)
// makes the feeder give up: the caller then runs the serial reader, which does what the reference does with such
// files, messages included.
class PairedFileFeeder : public Classifier::Feeder {
  static constexpr size_t CH = (size_t)1 << 20;
  struct File {
    int fd = -1; uint64_t size = 0, lines = 0;
    std::vector<uint64_t> cum;          // cum[c] = line ends before byte c * CH
  };
  // lines of a byte range of a file, read in pieces
  struct Lines {
    Lines(int fd, uint64_t off, size_t len, std::vector<uint8_t>& buf) : fd_(fd), off_(off), left_(len), buf_(buf) {
      if (buf_.size() < 2 * CH) buf_.resize(2 * CH);
    }
    bool next(const uint8_t*& p, size_t& n) {     // the next line without its '\n'; false at the end of the range
      for (;;) {
        const uint8_t* nl = have_ > pos_ ? (const uint8_t*)memchr(buf_.data() + pos_, '\n', have_ - pos_) : nullptr;
        if (nl) { p = buf_.data() + pos_; n = (size_t)(nl - p); pos_ += n + 1; return true; }
        if (left_ == 0) {
          if (have_ == pos_) return false;
          p = buf_.data() + pos_; n = have_ - pos_; pos_ = have_;   // last line of a file that does not end with '\n'
          return true;
        }
        // keep the unfinished line, read more
        if (pos_) { memmove(buf_.data(), buf_.data() + pos_, have_ - pos_); have_ -= pos_; pos_ = 0; }
        if (buf_.size() - have_ < CH) buf_.resize(buf_.size() * 2);
        const size_t take = std::min(left_, buf_.size() - have_);
        size_t got = 0;
        while (got < take) {
          const ssize_t r = pread(fd_, buf_.data() + have_ + got, take - got, (off_t)(off_ + got));
          if (r <= 0) die("Failed to read the objects file.");
          got += (size_t)r;
        }
        off_ += take; left_ -= take; have_ += take;
      }
    }
    int fd_; uint64_t off_; size_t left_; std::vector<uint8_t>& buf_; size_t pos_ = 0, have_ = 0;
  };
  struct SlotSink {
    uint8_t* d; size_t cap, w = 0;
    bool room(size_t n) const { return w + n <= cap; }
    void put(const void* p, size_t n) { memcpy(d + w, p, n); w += n; }
    void put(char c) { d[w++] = (uint8_t)c; }
  };
  struct StringSink {
    std::string& s;
    bool room(size_t) const { return true; }
    void put(const void* p, size_t n) { s.append((const char*)p, n); }
    void put(char c) { s.push_back(c); }
  };

 public:
  PairedFileFeeder(const std::string& f1, const std::string& f2, unsigned threads) : threads_(std::max(1u, threads)) {
    const std::string* names[2] = {&f1, &f2};
    for (int i = 0; i < 2; ++i) {
      f_[i].fd = open(names[i]->c_str(), O_RDONLY);
      struct stat st;
      if (f_[i].fd == -1 || fstat(f_[i].fd, &st) != 0 || st.st_size == 0) return;
      f_[i].size = (uint64_t)st.st_size;
      uint8_t c = 0;
      if (pread(f_[i].fd, &c, 1, 0) != 1 || c != '@') return;
    }
    ok_ = true;
  }
  ~PairedFileFeeder() override { for (File& f : f_) if (f.fd != -1) close(f.fd); }
  bool ok() const { return ok_; }
  uint64_t merged_estimate() const { return (f_[0].size + f_[1].size) / 2; }
  bool fastq() const override { return false; }          // what the slots get is the merged FASTA text
  bool gave_up() const override { return gave_up_.load(); }
  uint64_t remaining() const override { return (f_[0].size - pos_[0] + f_[1].size - pos_[1]) / 2; }

  bool assign(size_t want, size_t cap, Classifier::Range& r) override {
    if (!counted_) {
      count_lines();
      counted_ = true;
      if (f_[0].lines != f_[1].lines || f_[0].lines % 4 != 0) { gave_up_ = true; return false; }
      records_ = f_[0].lines / 4;
    }
    if (gave_up_ || next_ >= records_) return false;
    // records up to the one that starts behind pos + want in the first file
    uint64_t r1 = records_;
    if (f_[0].size - pos_[0] > want + want / 8) {
      const size_t c = (size_t)((pos_[0] + want) / CH);
      r1 = std::min<uint64_t>(records_, std::max<uint64_t>(f_[0].cum[c] / 4 + 1, next_ + 1));
    }
    uint64_t e0, e1;
    for (int tries = 0;; ++tries) {
      e0 = line_start(f_[0], 4 * r1); e1 = line_start(f_[1], 4 * r1);
      // merged text: one header and both sequences, at most half of what the two files hold for the records
      const uint64_t est = ((e0 - pos_[0]) + (e1 - pos_[1])) / 2;
      if (est <= cap - cap / 16 || r1 == next_ + 1 || tries == 8) break;
      r1 = next_ + std::max<uint64_t>(1, (uint64_t)((double)(r1 - next_) * (double)(cap - cap / 8) / (double)est));
    }
    r.off = pos_[0]; r.len = (size_t)(e0 - pos_[0]); r.off2 = pos_[1]; r.len2 = (size_t)(e1 - pos_[1]); r.mem = nullptr; r.keep.reset();
    pos_[0] = e0; pos_[1] = e1; next_ = r1;
    return true;
  }
  void read(const Classifier::Range&, size_t, uint8_t*, size_t) override { die("paired-end ranges are read through fill()"); }
  size_t fill(const Classifier::Range& r, uint8_t* dst, size_t cap) override {
    SlotSink s{dst, cap};
    return merge(r, s) ? s.w : (size_t)-1;
  }
  void text(const Classifier::Range& r, std::string& out) override {
    out.clear();
    out.reserve((r.len + r.len2) / 2 + 64);
    StringSink s{out};
    merge(r, s);
  }

 private:
  static bool sep(uint8_t c) { return c == ' ' || c == '/' || c == '\t' || c == '@'; }    // file.cc:224
  static void id_of(const uint8_t* p, size_t n, const uint8_t*& id, size_t& len) {
    size_t a = 0;
    while (a < n && sep(p[a])) ++a;
    size_t b = a;
    while (b < n && !sep(p[b])) ++b;
    id = p + a; len = b - a;
  }
  [[noreturn]] void give_up() { gave_up_ = true; throw std::runtime_error("paired-end input needs the serial reader"); }

  template <typename Sink> bool merge(const Classifier::Range& r, Sink& s) {
    static thread_local std::vector<uint8_t> b0, b1;
    Lines A(f_[0].fd, r.off, r.len, b0), B(f_[1].fd, r.off2, r.len2, b1);
    const uint8_t *p, *q; size_t n, m;
    for (;;) {
      const bool ha = A.next(p, n), hb = B.next(q, m);
      if (!ha && !hb) return true;
      if (!ha || !hb || n == 0 || m == 0 || p[0] != '@' || q[0] != '@') give_up();
      const uint8_t *ia, *ib; size_t la, lb;
      id_of(p, n, ia, la); id_of(q, m, ib, lb);
      if (la == 0 || la != lb || memcmp(ia, ib, la) != 0) give_up();
      if (!s.room(la + 2)) return false;
      s.put('>'); s.put(ia, la); s.put('\n');
      if (!A.next(p, n)) give_up();
      if (!s.room(n + 1)) return false;
      s.put(p, n); s.put('N');
      if (!B.next(q, m)) give_up();
      if (!s.room(m + 1)) return false;
      s.put(q, m); s.put('\n');
      if (!A.next(p, n) || !A.next(p, n) || !B.next(q, m) || !B.next(q, m)) give_up();
    }
  }

  void count_lines() {
    struct timeval ta, tb;
    gettimeofday(&ta, nullptr);
    size_t nch[2];
    for (int i = 0; i < 2; ++i) { nch[i] = (size_t)((f_[i].size + CH - 1) / CH); f_[i].cum.assign(nch[i] + 1, 0); }
    std::atomic<size_t> next{0};
    const size_t total = nch[0] + nch[1];
    auto work = [&] {
      const size_t SUB = (size_t)256 << 10;      // read and count in pieces that stay in the core's cache
      std::vector<uint8_t> buf(SUB);
      for (;;) {
        const size_t j = next.fetch_add(1);
        if (j >= total) return;
        File& f = j < nch[0] ? f_[0] : f_[1];
        const size_t c = j < nch[0] ? j : j - nch[0];
        const uint64_t o = (uint64_t)c * CH;
        const size_t n = (size_t)std::min<uint64_t>(CH, f.size - o);
        size_t got = 0, lines = 0;
        while (got < n) {
          const ssize_t r = pread(f.fd, buf.data(), std::min(SUB, n - got), (off_t)(o + got));
          if (r <= 0) die("Failed to read the objects file.");
          lines += count_newlines(buf.data(), (size_t)r);
          got += (size_t)r;
        }
        f.cum[c + 1] = lines;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < threads_ && t < total; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    for (int i = 0; i < 2; ++i) {
      File& f = f_[i];
      for (size_t c = 0; c < nch[i]; ++c) f.cum[c + 1] += f.cum[c];
      uint8_t last = 0;
      if (pread(f.fd, &last, 1, (off_t)(f.size - 1)) != 1) die("Failed to read the objects file.");
      f.lines = f.cum[nch[i]] + (last != '\n' ? 1 : 0);
    }
    gettimeofday(&tb, nullptr);
    if (getenv("MIC_CLI_TIMING"))
      std::cerr << "[timing] paired-end files: " << f_[0].lines << " + " << f_[1].lines << " lines counted in "
                << ((tb.tv_sec - ta.tv_sec) * 1e3 + (tb.tv_usec - ta.tv_usec) / 1e3) << " ms on " << threads_ << " threads" << std::endl;
  }
  // offset of the first byte of line L (0 <= L <= lines; line `lines` starts at the end of the file)
  uint64_t line_start(File& f, uint64_t L) {
    if (L == 0) return 0;
    if (L >= f.lines) return f.size;
    const size_t i = (size_t)(std::lower_bound(f.cum.begin(), f.cum.end(), L) - f.cum.begin());   // cum[i-1] < L <= cum[i]
    const size_t c = i - 1;
    const uint64_t o = (uint64_t)c * CH;
    const size_t n = (size_t)std::min<uint64_t>(CH, f.size - o);
    scan_.resize(CH);
    size_t got = 0;
    while (got < n) {
      const ssize_t r = pread(f.fd, scan_.data() + got, n - got, (off_t)(o + got));
      if (r <= 0) die("Failed to read the objects file.");
      got += (size_t)r;
    }
    uint64_t k = L - f.cum[c];
    const uint8_t* p = scan_.data();
    const uint8_t* end = p + n;
    while (k) {
      const uint8_t* nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
      if (!nl) die("Failed to read the objects file.");      // the file changed under us
      p = nl + 1; --k;
    }
    return o + (uint64_t)(p - scan_.data());
  }

  File f_[2];
  unsigned threads_;
  bool ok_ = false, counted_ = false;
  std::atomic<bool> gave_up_{false};
  uint64_t pos_[2] = {0, 0}, records_ = 0, next_ = 0;
  std::vector<uint8_t> scan_;
};

// Gzip-compressed FASTQ: the file - or both mates of a pair at once - is inflated ON the first engine's device
// (mic_gz_inflate_device), indexed and checked there, and every batch gets into its ingest slot's device buffer without leaving the
// device: a pair merged the way the reference merges it (mic_pairs_merge_to_slot; file.cc:205-268), a single file's records copied
// (mic_text_to_slot).  The compressed bytes are all that crosses the link.  Whatever the device path does not take (several gzip
// members, block gzip, FASTA, mates whose lines or ids do not pair up, texts of 4 GiB or more) leaves ok() false and the caller
// inflates on the host as before.  Ranges count RECORDS: off = first, len = number.
class DeviceGzFeeder : public Classifier::Feeder {
 public:
  // stripes > 1 (one file only): the member is inflated a stripe at a time on a thread of its own and its records are handed out as they
  // become final (mic_gz_stream_*) - the classifier works on stripe i while stripe i + 1 decodes.  A failure after records were
  // handed out makes assign() throw and gave_up() true: the caller starts over on the CPU inflater.
  DeviceGzFeeder(mic_engine* e, const std::string& f1, const std::string& f2, unsigned stripes = 1) : e_(e), paired_(!f2.empty()) {
    const bool timing = getenv("MIC_CLI_TIMING") != nullptr;
    struct timeval t0, t1, t2;
    gettimeofday(&t0, nullptr);
    const std::string* names[2] = {&f1, &f2};
    int rc[2] = {MIC_E_UNSUPPORTED, paired_ ? MIC_E_UNSUPPORTED : MIC_OK};
    auto map_file = [&](int i) -> bool {
      const int fd = open(names[i]->c_str(), O_RDONLY);
      struct stat st;
      if (fd == -1) return false;
      bool ok = false;
      if (fstat(fd, &st) == 0 && st.st_size > 18) {
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        // The mapping stays until the feeder goes: the runtime pins the pages it uploads from, and unmapping pinned pages makes the
        // driver take the process's queues off the device and put them back - the next kernel then starts 4 ms late (measured).
        if (m != MAP_FAILED) { map_[i] = m; map_n_[i] = (size_t)st.st_size; ok = true; }
      }
      close(fd);
      return ok;
    };
    auto inflate = [&](int i) {
      if (!map_file(i)) return;
      uint32_t crc = 0;
      rc[i] = mic_gz_inflate_device(e_, map_[i], map_n_[i], &text_[i], &n_[i], &crc);
    };
    if (!paired_ && stripes > 1) {
      if (!map_file(0)) return;
      const int st = open_stream(stripes, timing, t0);
      if (st > 0) return;                                    // in stripes: the inflater thread is on its way
      if (st < 0) return;                                    // (began and failed, nothing handed out: the host path takes the file)
      // not a text for stripes: the whole of it is there (FASTA), or the member is taken in one piece (block gzip)
      if (!text_[0]) { uint32_t crc = 0; rc[0] = mic_gz_inflate_device(e_, map_[0], map_n_[0], &text_[0], &n_[0], &crc); } else rc[0] = MIC_OK;
    } else {
      std::thread other;
      if (paired_) other = std::thread([&] { inflate(1); });
      inflate(0);
      if (other.joinable()) other.join();
    }
    gettimeofday(&t1, nullptr);
    if (rc[0] != MIC_OK || rc[1] != MIC_OK) { why_ = "the device inflater does not take this file"; return; }
    uint32_t status = 0;
    const uint64_t* s = nullptr; size_t ns = 0;
    if (paired_) {
      if (mic_pairs_index_device(e_, text_[0], n_[0], text_[1], n_[1], &pairs_, &n_rec_, &status) != MIC_OK || status || !pairs_) {
        why_ = "the mates do not pair up line by line";
        return;
      }
      if (mic_pairs_offsets(pairs_, &s, &ns, &stride_) != MIC_OK || ns < 2) return;
    } else {
      if (mic_text_index_device(e_, text_[0], n_[0], &single_, &n_rec_, &status) != MIC_OK || status || !single_) {
        why_ = "neither FASTA nor FASTQ records of four lines";
        return;
      }
      fasta_ = mic_text_format(single_) == '>';
      if (mic_text_offsets(single_, &s, &ns, &stride_) != MIC_OK || ns < 2) return;
    }
    off_.assign(s, s + ns);
    gettimeofday(&t2, nullptr);
    if (timing)
      std::cerr << "[timing] device inflate: " << (n_[0] + n_[1]) / 1e6 << " MB of text in "
                << ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) / 1e3) << " ms, " << n_rec_ << (paired_ ? " pairs" : " records")
                << " indexed and checked in " << ((t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_usec - t1.tv_usec) / 1e3) << " ms" << std::endl;
    ok_ = true;
  }
  ~DeviceGzFeeder() override {
    if (inflater_.joinable()) inflater_.join();
    if (gs_) mic_gz_stream_close(gs_, 1);                   // (the text is freed below)
    for (Piece& pc : pieces_) if (pc.t) mic_text_free(e_, pc.t);
    for (int i = 0; i < 2; ++i) if (map_[i]) munmap(map_[i], map_n_[i]);
    if (pairs_) mic_pairs_free(e_, pairs_);
    if (single_) mic_text_free(e_, single_);
    for (void* t : text_) if (t) mic_gz_free_text(e_, t);
  }
  bool ok() const { return ok_; }
  const char* why() const { return why_; }
  bool gave_up() const override { return gave_up_.load(); }   // the striped inflate failed: the file is the CPU inflater's (run_stream returns false)
  std::string gave_up_why() { std::lock_guard<std::mutex> lk(mu_); return gave_up_why_; }
  bool striped() const { return striped_; }
  uint64_t text_bytes() const { return striped_ ? isize_ : off_.empty() ? 0 : off_.back(); }
  bool fastq() const override { return false; }          // (nothing for the loaders to strip: the slots are filled on the device)
  bool resident() const override { return true; }
  int resident_flags() const override { return paired_ || fasta_ ? MIC_INGEST_RESIDENT : MIC_INGEST_RESIDENT_FASTQ; }
  uint64_t remaining() const override { return striped_ ? (isize_ > handed_ ? isize_ - handed_ : 0) : off_.back() - off_[cur_]; }

  bool assign(size_t want, size_t cap, Classifier::Range& r) override {
    if (striped_) return assign_striped(want, cap, r);
    if (rec_of(cur_) >= n_rec_) return false;
    const uint64_t limit = std::min<uint64_t>(want, cap - cap / 16);
    // the last boundary whose text still fits (at least one stride: a stride that does not fit is handed to the host path)
    size_t hi = (size_t)(std::upper_bound(off_.begin() + (ptrdiff_t)cur_, off_.end(), off_[cur_] + limit) - off_.begin()) - 1;
    if (hi <= cur_) hi = cur_ + 1;
    while (hi + 1 < off_.size() && rec_of(hi) == rec_of(cur_)) ++hi;
    r.off = rec_of(cur_); r.len = (size_t)(rec_of(hi) - rec_of(cur_)); r.off2 = 0; r.len2 = 0; r.mem = nullptr; r.keep.reset();
    cur_ = hi;
    return r.len != 0;
  }
  void read(const Classifier::Range&, size_t, uint8_t*, size_t) override { die("device-resident ranges are filled on the device"); }
  size_t fill_resident(const Classifier::Range& r, mic_engine* e, size_t slot) override {
    size_t n = 0;
    // (e: the engine the slot belongs to - on another device than the text it reads / copies over peer access)
    const int rc = paired_ ? mic_pairs_merge_to_slot(e, pairs_, r.off, r.off + r.len, slot, &n) : mic_text_to_slot(e, text_of(r), r.off, r.off + r.len, slot, &n);
    return rc == MIC_OK ? n : (size_t)-1;
  }
  size_t fill(const Classifier::Range& r, uint8_t* dst, size_t cap) override {
    size_t n = 0;
    return to_host(r, dst, cap, n) == MIC_OK ? n : (size_t)-1;
  }
  void text(const Classifier::Range& r, std::string& out) override {
    uint64_t bytes;
    if (striped_) {
      std::lock_guard<std::mutex> lk(mu_);
      const Piece& pc = pieces_[(size_t)r.off2];
      const uint64_t a = r.off / pc.stride, b = r.off + r.len >= pc.n_rec ? pc.off.size() - 1 : (r.off + r.len) / pc.stride;
      bytes = pc.off[b] - pc.off[a];
    } else {
      const uint64_t a = r.off / stride_, b = r.off + r.len >= n_rec_ ? off_.size() - 1 : (r.off + r.len) / stride_;
      bytes = off_[b] - off_[a];
    }
    out.resize((size_t)bytes);
    size_t n = 0;
    if (!out.empty()) check(to_host(r, &out[0], out.size(), n), "text of a batch");
    out.resize(n);
  }

 private:
  // one stripe's (or several stripes') whole records: a text handle of its own, records numbered from 0
  struct Piece { mic_text* t = nullptr; uint64_t n_rec = 0; uint32_t stride = 64; std::vector<uint64_t> off; size_t cur = 0; };

  mic_text* text_of(const Classifier::Range& r) {
    if (!striped_) return single_;
    std::lock_guard<std::mutex> lk(mu_);
    return pieces_[(size_t)r.off2].t;
  }
  int to_host(const Classifier::Range& r, void* dst, size_t cap, size_t& n) {
    return paired_ ? mic_pairs_text(e_, pairs_, r.off, r.off + r.len, dst, cap, &n) : mic_text_copy(e_, text_of(r), r.off, r.off + r.len, dst, cap, &n);
  }
  uint64_t rec_of(size_t i) const { return std::min<uint64_t>((uint64_t)i * stride_, n_rec_); }

  // ---- the striped form
  void give_up(const std::string& why) {
    std::lock_guard<std::mutex> lk(mu_);
    gave_up_ = true; gave_up_why_ = why; inflating_ = false;
    cv_.notify_all();
  }
  // the text [carry_, n_final) has become final: its whole records (last: all of it) become a piece
  bool index_upto(size_t n_final, bool last) {
    if (n_final <= carry_) { if (last && pieces_.empty()) { give_up("no records"); return false; } return true; }
    mic_text* t = nullptr; uint64_t n_rec = 0, used = 0; uint32_t status = 0;
    const uint8_t* from = (const uint8_t*)text_[0] + carry_;
    const int rc = last ? mic_text_index_device(e_, from, n_final - carry_, &t, &n_rec, &status)
                        : mic_text_index_front_device(e_, from, n_final - carry_, &t, &n_rec, &used, &status);
    if (rc != MIC_OK || status || (last && (!t || mic_text_format(t) != '@'))) {
      if (t) mic_text_free(e_, t);
      give_up("not FASTQ records of four lines");
      return false;
    }
    if (last) used = n_final - carry_;
    if (t) {
      Piece pc;
      pc.t = t; pc.n_rec = n_rec;
      const uint64_t* s = nullptr; size_t ns = 0;
      if (mic_text_offsets(t, &s, &ns, &pc.stride) != MIC_OK || ns < 2) { mic_text_free(e_, t); give_up("no record offsets"); return false; }
      pc.off.assign(s, s + ns);
      std::lock_guard<std::mutex> lk(mu_);
      pieces_.push_back(std::move(pc));
      n_rec_ += n_rec;
      cv_.notify_all();
    }
    carry_ += used;
    return true;
  }
  // 1: in stripes; 0: not in stripes - text_[0] null (take the member whole) or all of it inflated (n_[0] bytes); -1: the host path's
  int open_stream(unsigned stripes, bool timing, const struct timeval& t0) {
    size_t isize = 0;
    if (mic_gz_stream_open(e_, map_[0], map_n_[0], stripes, &gs_, &text_[0], &isize) != MIC_OK) { gs_ = nullptr; text_[0] = nullptr; return 0; }
    isize_ = isize;
    size_t n_final = 0; int done = 0;
    if (mic_gz_stream_next(gs_, &n_final, &done) != MIC_OK) { why_ = "the device inflater does not take this file"; return -1; }
    char first = 0;
    if (n_final == 0 || mic_gz_copy_text(e_, text_[0], 0, 1, &first) != MIC_OK) { why_ = "the device inflater does not take this file"; return -1; }
    if (first != '@') {
      // FASTA (or something the index will turn down): its records are not cut by counting lines - all stripes now, then as one text
      while (!done) if (mic_gz_stream_next(gs_, &n_final, &done) != MIC_OK) { why_ = "the device inflater does not take this file"; return -1; }
      n_[0] = n_final;
      mic_gz_stream_close(gs_, 1); gs_ = nullptr;
      return 0;
    }
    striped_ = true;
    if (!index_upto(n_final, done != 0)) { striped_ = false; why_ = "not FASTQ records of four lines"; return -1; }
    struct timeval t1;
    gettimeofday(&t1, nullptr);
    if (timing)
      std::cerr << "[timing] device inflate in stripes: the first " << n_final / 1e6 << " MB of " << isize_ / 1e6 << " MB of text after "
                << ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) / 1e3) << " ms, " << n_rec_ << " records indexed" << std::endl;
    inflating_ = done == 0;
    if (inflating_)
      inflater_ = std::thread([this, timing, t0] {
        for (;;) {
          size_t nf = 0; int dn = 0;
          if (mic_gz_stream_next(gs_, &nf, &dn) != MIC_OK) { give_up(mic_last_error()); return; }
          if (!index_upto(nf, dn != 0)) return;
          if (dn) break;
        }
        struct timeval t2;
        gettimeofday(&t2, nullptr);
        if (timing)
          std::cerr << "[timing] device inflate in stripes: all " << isize_ / 1e6 << " MB of text after "
                    << ((t2.tv_sec - t0.tv_sec) * 1e3 + (t2.tv_usec - t0.tv_usec) / 1e3) << " ms" << std::endl;
        std::lock_guard<std::mutex> lk(mu_);
        inflating_ = false;
        cv_.notify_all();
      });
    ok_ = true;
    return 1;
  }
  bool assign_striped(size_t want, size_t cap, Classifier::Range& r) {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (gave_up_) throw std::runtime_error("the device inflater gave the file back: " + gave_up_why_);
      if (piece_cur_ < pieces_.size()) {
        Piece& pc = pieces_[piece_cur_];
        const uint64_t rec_cur = std::min<uint64_t>((uint64_t)pc.cur * pc.stride, pc.n_rec);
        if (rec_cur < pc.n_rec) {
          const uint64_t limit = std::min<uint64_t>(want, cap - cap / 16);
          size_t hi = (size_t)(std::upper_bound(pc.off.begin() + (ptrdiff_t)pc.cur, pc.off.end(), pc.off[pc.cur] + limit) - pc.off.begin()) - 1;
          if (hi <= pc.cur) hi = pc.cur + 1;
          auto rec_at = [&](size_t i) { return std::min<uint64_t>((uint64_t)i * pc.stride, pc.n_rec); };
          while (hi + 1 < pc.off.size() && rec_at(hi) == rec_cur) ++hi;
          r.off = rec_cur; r.len = (size_t)(rec_at(hi) - rec_cur); r.off2 = piece_cur_; r.len2 = 0; r.mem = nullptr; r.keep.reset();
          handed_ += pc.off[std::min(hi, pc.off.size() - 1)] - pc.off[pc.cur];
          pc.cur = hi;
          if (r.len != 0) return true;
        }
        ++piece_cur_;
        continue;
      }
      if (!inflating_) return false;
      cv_.wait(lk);
    }
  }

  mic_engine* e_;
  bool paired_, fasta_ = false;
  void* text_[2] = {nullptr, nullptr};
  size_t n_[2] = {0, 0};
  void* map_[2] = {nullptr, nullptr};
  size_t map_n_[2] = {0, 0};
  mic_pairs* pairs_ = nullptr;
  mic_text* single_ = nullptr;
  uint64_t n_rec_ = 0;
  uint32_t stride_ = 64;
  std::vector<uint64_t> off_;
  size_t cur_ = 0;
  bool ok_ = false;
  const char* why_ = "";
  // striped
  mic_gz_stream* gs_ = nullptr;
  bool striped_ = false;
  std::deque<Piece> pieces_;
  size_t piece_cur_ = 0;
  uint64_t isize_ = 0, handed_ = 0;
  size_t carry_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  bool inflating_ = false;
  std::atomic<bool> gave_up_{false};
  std::string gave_up_why_;
  std::thread inflater_;
};


}  // namespace detail
}  // namespace mic
#endif
