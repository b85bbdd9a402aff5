// classifier.cpp — see classifier.hpp.  Host pipeline per input file:
//   mmap -> index reads (mic_index_reads) -> split into batches -> [OpenMP over batches] pack into the engine's
//   pinned buffers (mic_pack_reads) -> mic_batch_query (async H2D + kernels + D2H on the batch's stream) ->
//   mic_batch_wait -> format CSV lines -> ordered write.
// Multi-device: one engine per GPU with the whole table resident; batches are dealt round-robin (reads are
// independent), results are written in file order.
#include "classifier.hpp"
#include "pgz.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mic {

namespace {

[[noreturn]] void die(const std::string& msg) { throw std::runtime_error(msg); }

void check(int rc, const char* what) {
  if (rc != MIC_OK) die(std::string(what) + ": " + mic_last_error());
}

// file.cc:57-80 (split on ' ', ',', '\n', '\t', '\r', at most max elements)
std::vector<std::string> split_line(const std::string& line, size_t max_el) {
  std::vector<std::string> out;
  size_t t = 0, n = line.size();
  auto sep = [](char c) { return c == ' ' || c == ',' || c == '\n' || c == '\t' || c == '\r'; };
  while (t < n && out.size() < max_el) {
    while (t < n && sep(line[t])) ++t;
    std::string v;
    while (t < n && !sep(line[t])) v.push_back(line[t++]);
    if (!v.empty()) out.push_back(v);
  }
  return out;
}

// file.cc:83-122 with an explicit separator list
std::vector<std::string> split_seps(const std::string& line, const std::string& seps) {
  std::vector<std::string> out;
  size_t t = 0, n = line.size();
  while (t < n) {
    while (t < n && seps.find(line[t]) != std::string::npos) ++t;
    std::string v;
    while (t < n && seps.find(line[t]) == std::string::npos) v.push_back(line[t++]);
    if (!v.empty()) out.push_back(v);
  }
  return out;
}

bool get_line(std::istream& in, std::string& line) { return static_cast<bool>(std::getline(in, line)); }

bool file_exists(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return false;
  fclose(f);
  return true;
}

bool is_gzip(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  unsigned char m[2] = {0, 0};
  size_t n = fread(m, 1, 2, f);
  fclose(f);
  return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

}  // namespace

Classifier::Classifier(const Options& opt) : opt_(opt) {
#ifdef _OPENMP
  omp_set_num_threads((int)opt_.threads);
#else
  opt_.threads = 1;
#endif
  parse_targets();
  if (const char* env = getenv("MIC_SEGMENT_MB")) { long v = atol(env); if (v >= 1) segment_bytes_ = (size_t)v << 20; }
  if (const char* env = getenv("MIC_SEGMENT_KB")) { long v = atol(env); if (v >= 1) segment_bytes_ = (size_t)v << 10; }
  std::cerr << "CuCLARK version 1.1 (MI355X engine mi-clark; CuCLARK (c) 2016 Robin Kobus)" << std::endl;
  std::cerr << "Based on CLARK version 1.1.3 (UCR CS&E. Copyright 2013-2016 Rachid Ounit, rouni001@cs.ucr.edu) " << std::endl;
  if (opt_.min_count_t > 0) std::cerr << "Minimum k-mers occurences in Targets is set to " << opt_.min_count_t << std::endl;
  if (opt_.light) std::cerr << "Using light database in RAM (" << opt_.gap << ")" << std::endl;
  if (opt_.sampling > 2) std::cerr << "Sampling factor is " << opt_.sampling << std::endl;

  const std::string db = db_name();
  if (!(file_exists(db + ".sz") && file_exists(db + ".ky") && file_exists(db + ".lb"))) {
    // first run: build the database from the target genomes (reference: makeSpecificTargetSets,
    // CuCLARK_hh.hh:304-309,691-1329) — here on the GPU (mic_db_build)
    std::cerr << "Starting the creation of the database of targets specific " << opt_.k << "-mers from input files..." << std::endl;
    std::vector<const char*> files; std::vector<uint16_t> labs;
    for (const auto& t : targets_id_) {
      files.push_back(t.first.c_str());
      labs.push_back((uint16_t)(std::find(labels_.begin(), labels_.end(), t.second) - labels_.begin()));
    }
    uint64_t n_kmers = 0;
    // cuCLARK-l samples the targets (every gap-th non-overlapping k-block, CuCLARK_hh.hh:694-895)
    int rc = mic_db_build(files.data(), labs.data(), files.size(), (int)opt_.k, opt_.htsize, 0, opt_.min_count_t,
                          opt_.light ? (uint32_t)opt_.gap : 0u, db.c_str(), 0, (int)opt_.threads, 0, &n_kmers);
    if (rc != MIC_OK) die(std::string("Failed to create the database: ") + mic_db_build_error());
    std::cerr << (opt_.light ? "Creating light database in disk..." : "Creating database in disk...") << std::endl;
    std::cerr << n_kmers << " " << opt_.k << "-mers successfully stored in database." << std::endl;
  }
  {  // the table size is part of the database name: a .sz of another size is not this database (e.g. a build cut short)
    struct stat st;
    if (stat((db + ".sz").c_str(), &st) == 0 && (uint64_t)st.st_size != opt_.htsize)
      die("The database file " + db + ".sz holds " + std::to_string((unsigned long long)st.st_size) + " buckets, expected " +
          std::to_string((unsigned long long)opt_.htsize) + ": remove the database files and build them again.");
  }
  int n_dev = 0;
  check(mic_device_count(&n_dev), "device discovery");
  if (n_dev == 0) die("No HIP device found.");
  size_t use = opt_.devices == 0 ? (size_t)n_dev : std::min(opt_.devices, (size_t)n_dev);
  // MIC_SHARD_ENGINES=<n> forces n engines (round-robin over the devices) in either multi-device mode: both modes can then be
  // tested on one GPU
  bool forced = false;
  if (const char* env = getenv("MIC_SHARD_ENGINES")) { long v = atol(env); if (v >= 1 && v <= 64) { use = (size_t)v; forced = true; } }
  if (!opt_.db_sharded && !forced && opt_.batches < use) use = std::max<size_t>(1, opt_.batches);
  // Multi-device layout: `use` engines = groups_ read groups x parts_ table parts.
  //   default (-d N):   the table replicated, N groups of one engine, batches dealt to the groups (reads are independent)
  //   --db-sharded:     the reference's mode (every device holds a share of the table, CuClarkDB.cu:566-574): parts_ engines hold one
  //                     part of the table each (mic_db_set_part) and answer every batch of their group together; --parts P picks P,
  //                     the default is the smallest P that divides the engines and whose part fits a device - a part's kernel
  //                     costs nearly as much as the whole table's, so engines beyond that divide the READS
  const int nd_used = (int)std::min<size_t>(use, (size_t)n_dev);
  gz_on_device_ = true;
  std::vector<int> pm;
  if (use > 1) {
    // peer access between the devices in use (the reference: CuClarkDB.cu:184-208): the row exchange of the table-sharded mode and
    // the slots filled from a text inflated on another device go over it
    pm.assign((size_t)nd_used * nd_used, 0);
    check(mic_peer_matrix(pm.data(), nd_used), "peer access");
    for (int i = 0; i < nd_used; ++i)
      for (int j = 0; j < nd_used; ++j) if (i != j && !pm[(size_t)i * nd_used + j]) gz_on_device_ = false;
  }
  // What the run allocates on the devices NEXT to the table: the ingest slots (pinned and device buffers of the streaming path, set
  // up on a side thread while the database loads) and the buffers of a device inflate.  Known before the table is cut: the number
  // of parts is chosen with them in the sum, and the builders are told (mic_db_reserve_hbm) so that the staging area they size from
  // the free HBM does not depend on how far the side thread has got.
  const bool want_slots = device_ingest() && !opt_.objects.empty();
  size_t in_bytes = ~(size_t)0 >> 1, slot_bytes = 0, workers = 0;
  struct GzFile { size_t bytes; uint32_t isize; };
  std::vector<GzFile> gz;
  uint64_t gz_hbm = 0;
  if (want_slots) {
    struct stat st;
    if (opt_.objects2.empty() && !is_gzip(opt_.objects) && stat(opt_.objects.c_str(), &st) == 0) in_bytes = (size_t)st.st_size;
    ingest_geometry(in_bytes, slot_bytes, workers);
    // two compressed mates (or one compressed file) on the first engine are inflated on the device (run_paired / run)
    const bool gz_pair = !opt_.objects2.empty() && is_gzip(opt_.objects) && is_gzip(opt_.objects2) && !getenv("MIC_SERIAL_PAIRS");
    const bool gz_single = opt_.objects2.empty() && is_gzip(opt_.objects);
    if (gz_on_device_ && (gz_pair || gz_single) && !getenv("MIC_GZ_HOST")) {
      std::vector<const std::string*> files = {&opt_.objects};
      if (gz_pair) files.push_back(&opt_.objects2);
      for (const std::string* f : files) {
        const int fd = open(f->c_str(), O_RDONLY);
        uint8_t t[4];
        uint8_t h[18] = {0};
        const bool bgzf = fd != -1 && pread(fd, h, 18, 0) == 18 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        if (bgzf && fstat(fd, &st) == 0) {
          // block gzip: the text is the sum over the members' trailers (a header and a trailer read per member, while the database loads)
          uint64_t total = 0; off_t o = 0; bool good = true;
          while (good && o < st.st_size) {
            uint8_t b[18], z[4];
            good = pread(fd, b, 18, o) == 18 && b[0] == 0x1f && b[1] == 0x8b && (b[3] & 4) && b[12] == 'B' && b[13] == 'C';
            const off_t bsize = good ? (off_t)(b[16] | (b[17] << 8)) + 1 : 0;
            good = good && bsize >= 28 && o + bsize <= st.st_size && pread(fd, z, 4, o + bsize - 4) == 4;
            if (good) { total += (uint32_t)z[0] | ((uint32_t)z[1] << 8) | ((uint32_t)z[2] << 16) | ((uint32_t)z[3] << 24); o += bsize; }
          }
          if (good && total < 0xFFFFFF00ull) gz.push_back({(size_t)st.st_size, (uint32_t)total});
        } else if (fd != -1 && fstat(fd, &st) == 0 && st.st_size > 18 && pread(fd, t, 4, st.st_size - 4) == 4)
          gz.push_back({(size_t)st.st_size, (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24)});
        if (fd != -1) close(fd);
      }
    }
    for (const GzFile& g : gz) gz_hbm += mic_gz_reserve_bytes(g.bytes, g.isize);
  }
  // device memory engine `e` of `use` allocates next to its table when the table is cut into P parts
  auto reserve_of = [&](size_t P, size_t e) -> uint64_t {
    if (!want_slots) return 0;
    const size_t G = use / P, per_engine = (workers + use - 1) / use;      // ~6.1 x the slot size per slot (mic_ingest.hip)
    // table-sharded: every engine also holds the partial rows, gathered rows and packed reads of every slot of its group (~6.5 x the slot)
    const size_t group_extra = P > 1 ? ((workers + G - 1) / G) * 13 / 2 : 0;
    return (uint64_t)(per_engine * 7 + group_extra) * slot_bytes + (e == 0 ? gz_hbm : 0);   // (the inflated text lives on the first engine's device)
  };
  parts_ = 1;
  if (opt_.db_sharded) {
    if (opt_.parts && use % opt_.parts != 0) die("--parts " + std::to_string(opt_.parts) + " does not divide the " + std::to_string(use) + " device(s) in use.");
    struct stat sz, ky;
    if (stat((db + ".sz").c_str(), &sz) != 0 || stat((db + ".ky").c_str(), &ky) != 0) die("Failed to open " + db + ".sz");
    // the memory a device must offer: the least free memory over the devices in use (a device may be shared with another job)
    uint64_t free_min = ~(uint64_t)0;
    for (int d = 0; d < nd_used; ++d) {
      uint64_t f = 0, t = 0;
      check(mic_device_memory(d, &f, &t), "device memory");
      free_min = std::min(free_min, f);
    }
    const int kb = mic_key_bytes_rule((uint64_t)sz.st_size, (int)opt_.k);
    const uint64_t n_el = (uint64_t)ky.st_size / (uint64_t)(kb > 0 ? kb : 4);
    const uint64_t images = (uint64_t)sz.st_size + n_el * (uint64_t)(kb + 2);
    // resident bytes and how a part is cut follow the layout: super-k-mer tables (k >= 24) are cut by RESIDENT slot
    // range and every part is built from the whole images; direct and minimizer tables are cut by on-disk bucket range, and a device
    // then holds its engines' share of the images only (mic_db_load_files_multi)
    const char* lay = getenv("MIC_LAYOUT");
    const bool bucket_cut = (lay && (!strcmp(lay, "direct") || !strcmp(lay, "minimizer"))) || (!lay && opt_.k < 24);
    const uint64_t table = lay && !strcmp(lay, "minimizer") ? n_el * 30 : lay && !strcmp(lay, "super2") ? n_el * 24
                         : bucket_cut ? (uint64_t)sz.st_size * 64 + n_el / 8 : n_el * 12;
    const size_t epd = (use + (size_t)nd_used - 1) / (size_t)nd_used;       // engines that share a device (MIC_SHARD_ENGINES on fewer devices)
    auto need = [&](size_t P) -> uint64_t {
      const uint64_t part = table / P;
      const uint64_t images_dev = bucket_cut ? images / P * std::min<uint64_t>(epd, P) + (1u << 20) : images;
      // the device's engines' parts, the staging area of the build in progress (the builders work in passes when it is small: a
      // quarter of a part at least), what the engines allocate next to their tables, the runtime's own
      return images_dev + epd * part + part / 4 + epd * reserve_of(P, 0) + ((uint64_t)1 << 30);
    };
    auto sizes = [&](size_t P) {
      char b[256];
      snprintf(b, sizeof(b), "%.1f GB (images %.1f GB%s + %zu part(s) of %.1f GB + build staging + %.1f GB of ingest buffers)", need(P) / 1e9,
               (bucket_cut ? images / P * std::min<uint64_t>(epd, P) : images) / 1e9, bucket_cut ? " of this device's bucket ranges" : ", whole: a super-k-mer part is a slot range of the resident table",
               epd, table / P / 1e9, epd * reserve_of(P, 0) / 1e9);
      return std::string(b);
    };
    if (opt_.parts) {
      parts_ = opt_.parts;
      if (need(parts_) > free_min)
        std::cerr << "Note: --parts " << parts_ << " needs about " << sizes(parts_) << " per device, " << free_min / 1e9 << " GB are free." << std::endl;
    } else {
      // the smallest number of parts that fits: a part's kernel costs nearly as much as the whole table's, so engines
      // beyond what capacity needs divide the READS
      parts_ = 0;
      for (size_t p = 1; p <= use; ++p)
        if (use % p == 0 && need(p) <= free_min) { parts_ = p; break; }
      if (!parts_)
        die("The database does not fit " + std::to_string(use) + " engine(s) on " + std::to_string(nd_used) + " device(s) with " + std::to_string(free_min / 1000000 / 1000.0) +
            " GB free each: " + std::to_string(use) + " parts need about " + sizes(use) + " per device.  Use more devices (-d), or --parts P to try a cut yourself.");
    }
  }
  groups_ = use / parts_;
  std::cerr << "Loading database [" << db << ".*] (s=" << opt_.sampling << ")..." << std::endl;
  const size_t per_engine_batches = std::max<size_t>(1, (opt_.batches + groups_ - 1) / groups_);
  for (size_t d = 0; d < use; ++d) {
    mic_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = (int)(d % (size_t)n_dev); cfg.k = (int)opt_.k; cfg.num_targets = (uint32_t)(names_.size());
    cfg.num_batches = (uint32_t)per_engine_batches;
    cfg.row_words = opt_.extended ? (uint32_t)std::min<size_t>(names_.size() + 1, 65) : 16;
    mic_engine* e = nullptr;
    check(mic_create(&cfg, &e), "engine creation");
    engines_.push_back(e);
    if (parts_ > 1) check(mic_db_set_part(e, (uint32_t)(d % parts_), (uint32_t)parts_), "table part");
  }
  if (use > 1) {
    std::cerr << "Devices: " << use << " engine(s) on " << nd_used << " device(s)";
    if (opt_.db_sharded) std::cerr << ", table-sharded: " << parts_ << " part(s) x " << groups_ << " read group(s)";
    else std::cerr << ", read-sharded (table replicated)";
    std::cerr << "; peer access:";
    for (int i = 0; i < nd_used; ++i) {
      std::cerr << (i ? " | " : " ");
      for (int j = 0; j < nd_used; ++j) std::cerr << pm[(size_t)i * nd_used + j];
    }
    std::cerr << std::endl;
  }
  std::thread slots;
  if (want_slots) {
    for (size_t e = 0; e < engines_.size(); ++e) mic_db_reserve_hbm(engines_[e], reserve_of(parts_, e));
    std::vector<std::pair<size_t, uint32_t>> gzv;
    for (const GzFile& g : gz) gzv.push_back({g.bytes, g.isize});
    slots = std::thread([this, in_bytes, gzv] {
      try { ensure_ingest(in_bytes); } catch (const std::exception&) { release_ingest(); }
      for (const auto& g : gzv) if (mic_gz_reserve(engines_[0], g.first, g.second) != MIC_OK) break;     // (without it the call allocates for itself)
    });
  }
  std::string load_err;
  {
    // one read of .sz/.ky/.lb for all engines; one thread per device builds its tables (a single engine: the same call)
    const int rc = engines_.size() == 1 ? mic_db_load_files(engines_[0], db.c_str(), 0, opt_.sampling, 0, 0)
                                        : mic_db_load_files_multi(engines_.data(), engines_.size(), db.c_str(), 0, opt_.sampling);
    if (rc != MIC_OK) load_err = std::string("Failed to load the database: ") + mic_last_error();
  }
  if (slots.joinable()) slots.join();
  for (mic_engine* e : engines_) mic_db_reserve_hbm(e, 0);
  if (!load_err.empty()) die(load_err);
  {  // a default layout that was given up for another one is said, with the reason (the rate depends on it: DESIGN.md 5.3)
    std::istringstream rep(mic_db_last_build_report());
    std::string ln;
    while (std::getline(rep, ln))
      if (ln.compare(0, 9, "fallback:") == 0) std::cerr << "Note: resident table " << ln.substr(0, ln.rfind(':')) << std::endl;
  }
  mic_db_info info;
  check(mic_db_get_info(engines_[0], &info), "db info");
  static const char* const layout_name[] = {"?", "direct", "minimizer-keyed", "super-k-mer", "super-k-mer, both strands"};
  std::cerr << "Total DB size in HBM:\t" << info.hbm_bytes / 1000000 / 1000.0 << " GB (" << info.n_elems << " k-mers, "
            << info.n_overflow << " overflow slots, " << layout_name[info.layout >= 1 && info.layout <= 4 ? info.layout : 0]
            << " table" << (parts_ > 1 ? ", part 0 of " + std::to_string(parts_) : std::string()) << ") on " << use << " device(s)\n";
  if (getenv("MIC_CLI_TIMING")) {
    char kn[128] = "";
    if (mic_db_kernel_name(engines_[0], kn, sizeof(kn)) > 0) std::cerr << "[timing] query kernel: " << kn << std::endl;
  }
}

Classifier::~Classifier() {
  release_ingest();
  for (mic_engine* e : engines_) mic_destroy(e);
}

std::string Classifier::db_name() const {
  char buf[4096];
  const size_t n_lab = labels_.size() + labels_c_.size();
  if (opt_.light)
    snprintf(buf, sizeof(buf), "%s/db_central_k%lu_t%lu_s%lu_m%lu_light_%lu.tsk", opt_.folder.c_str(), (unsigned long)opt_.k,
             (unsigned long)n_lab, (unsigned long)opt_.htsize, (unsigned long)opt_.min_count_t, (unsigned long)opt_.gap);
  else
    snprintf(buf, sizeof(buf), "%s/db_central_k%lu_t%lu_s%lu_m%lu.tsk", opt_.folder.c_str(), (unsigned long)opt_.k,
             (unsigned long)n_lab, (unsigned long)opt_.htsize, (unsigned long)opt_.min_count_t);
  return buf;
}

void Classifier::parse_targets() {
  std::ifstream meta(opt_.targets);
  if (!meta) die("Failed to open targets data in file: " + opt_.targets);
  std::string line;
  while (get_line(meta, line)) {
    std::vector<std::string> ele = split_line(line, 3);
    if (ele.empty()) continue;
    if (!file_exists(ele[0])) die("Failed to open file: " + ele[0] + " defined in " + opt_.targets);
    if (ele.size() < 2) die(" Missing label for " + ele[0]);
    targets_id_.push_back({ele[0], ele[1]});
    if (std::find(labels_.begin(), labels_.end(), ele[1]) == labels_.end()) labels_.push_back(ele[1]);
    if (ele.size() > 2 && std::find(labels_c_.begin(), labels_c_.end(), ele[2]) == labels_c_.end()) labels_c_.push_back(ele[2]);
  }
  names_ = labels_;  // label index -> name; "NA" is index 0 of the reference's m_targetsName
  names_.insert(names_.end(), labels_c_.begin(), labels_c_.end());
}

// ---- segment sources ------------------------------------------------------------------------------------------------
namespace {

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

static unsigned inflate_threads(size_t cli_threads, unsigned files) {
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

}  // namespace

// ---- feeders of the device-ingest streaming path (Classifier::run_stream) ------------------------------------------
namespace {

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
static size_t count_newlines_plain(const uint8_t* p, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
  return c;
}
__attribute__((target("avx2"))) static size_t count_newlines_avx2(const uint8_t* p, size_t n) {
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
static size_t count_newlines(const uint8_t* p, size_t n) {
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
  DeviceGzFeeder(mic_engine* e, const std::string& f1, const std::string& f2) : e_(e), paired_(!f2.empty()) {
    const bool timing = getenv("MIC_CLI_TIMING") != nullptr;
    struct timeval t0, t1, t2;
    gettimeofday(&t0, nullptr);
    const std::string* names[2] = {&f1, &f2};
    int rc[2] = {MIC_E_UNSUPPORTED, paired_ ? MIC_E_UNSUPPORTED : MIC_OK};
    auto inflate = [&](int i) {
      const int fd = open(names[i]->c_str(), O_RDONLY);
      struct stat st;
      if (fd == -1) return;
      if (fstat(fd, &st) == 0 && st.st_size > 18) {
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
          uint32_t crc = 0;
          rc[i] = mic_gz_inflate_device(e_, m, (size_t)st.st_size, &text_[i], &n_[i], &crc);
          // The mapping stays until the feeder goes: the runtime pins the pages it uploads from, and unmapping pinned pages makes the
          // driver take the process's queues off the device and put them back - the next kernel then starts 4 ms late (measured).
          map_[i] = m; map_n_[i] = (size_t)st.st_size;
        }
      }
      close(fd);
    };
    std::thread other;
    if (paired_) other = std::thread([&] { inflate(1); });
    inflate(0);
    if (other.joinable()) other.join();
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
    for (int i = 0; i < 2; ++i) if (map_[i]) munmap(map_[i], map_n_[i]);
    if (pairs_) mic_pairs_free(e_, pairs_);
    if (single_) mic_text_free(e_, single_);
    for (void* t : text_) if (t) mic_gz_free_text(e_, t);
  }
  bool ok() const { return ok_; }
  const char* why() const { return why_; }
  uint64_t text_bytes() const { return off_.empty() ? 0 : off_.back(); }
  bool fastq() const override { return false; }          // (nothing for the loaders to strip: the slots are filled on the device)
  bool resident() const override { return true; }
  int resident_flags() const override { return paired_ || fasta_ ? MIC_INGEST_RESIDENT : MIC_INGEST_RESIDENT_FASTQ; }
  uint64_t remaining() const override { return off_.back() - off_[cur_]; }

  bool assign(size_t want, size_t cap, Classifier::Range& r) override {
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
    const int rc = paired_ ? mic_pairs_merge_to_slot(e, pairs_, r.off, r.off + r.len, slot, &n) : mic_text_to_slot(e, single_, r.off, r.off + r.len, slot, &n);
    return rc == MIC_OK ? n : (size_t)-1;
  }
  size_t fill(const Classifier::Range& r, uint8_t* dst, size_t cap) override {
    size_t n = 0;
    return to_host(r, dst, cap, n) == MIC_OK ? n : (size_t)-1;
  }
  void text(const Classifier::Range& r, std::string& out) override {
    const uint64_t a = r.off / stride_, b = r.off + r.len >= n_rec_ ? off_.size() - 1 : (r.off + r.len) / stride_;
    out.resize((size_t)(off_[b] - off_[a]));
    size_t n = 0;
    if (!out.empty()) check(to_host(r, &out[0], out.size(), n), "text of a batch");
    out.resize(n);
  }

 private:
  int to_host(const Classifier::Range& r, void* dst, size_t cap, size_t& n) {
    return paired_ ? mic_pairs_text(e_, pairs_, r.off, r.off + r.len, dst, cap, &n) : mic_text_copy(e_, single_, r.off, r.off + r.len, dst, cap, &n);
  }
  uint64_t rec_of(size_t i) const { return std::min<uint64_t>((uint64_t)i * stride_, n_rec_); }
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
};

}  // namespace

std::string merge_paired(const std::string& file1, const std::string& file2) {
  PairedSource src(file1, file2, ~(size_t)0 >> 1);
  if (!src.ok()) die("Error: Found read without sequence");
  Classifier::Segment s;
  std::string out;
  while (src.next(s)) out += s.own;
  return out;
}

bool merge_paired_parallel(const std::string& file1, const std::string& file2, unsigned threads, size_t batch_bytes, std::string& out) {
  PairedFileFeeder feed(file1, file2, threads);
  if (!feed.ok()) return false;
  out.clear();
  Classifier::Range r;
  std::string piece;
  try {
    while (feed.assign(batch_bytes, batch_bytes * 4 + 4096, r)) { feed.text(r, piece); out += piece; }
  } catch (const std::runtime_error&) {
    if (feed.gave_up()) return false;
    throw;
  }
  return !feed.gave_up();
}

void Classifier::run(const std::string& objects, const std::string& results) {
  auto simple = [&](const std::string& obj, const std::string& res) {
    if (is_gzip(obj) && device_ingest()) {
      // inflate up front (all threads), then the plain-file path on the inflated text
      struct timeval ta, tb;
      gettimeofday(&ta, nullptr);
      if (engines_.size() > 1 && !gz_on_device_ && !getenv("MIC_GZ_HOST"))
        std::cerr << "Note: the compressed input is inflated on the host (no peer access between the devices in use)." << std::endl;
      if (gz_on_device_ && !getenv("MIC_GZ_HOST")) {
        // inflated on the device, its FASTQ records handed to the ingest slots there (MIC_GZ_HOST=1: on the host, below)
        DeviceGzFeeder feed(engines_[0], obj, "");
        if (feed.ok()) {
          gettimeofday(&tb, nullptr);
          prelude_s_ = (tb.tv_sec - ta.tv_sec) + (tb.tv_usec - ta.tv_usec) / 1e6;
          run_stream(feed, res, false, (size_t)feed.text_bytes());
          prelude_s_ = 0;
          mic_gz_release(engines_[0]);
          return;
        }
        if (getenv("MIC_CLI_TIMING")) std::cerr << "[timing] device inflate: not used (" << feed.why() << ")" << std::endl;
        mic_gz_release(engines_[0]);
      }
      InflatedFile inf;
      const int rc = inf.inflate(obj, inflate_threads(opt_.threads, 1));
      if (rc < 0) die("Failed to uncompress input objects.");
      if (rc == 0) {
        gettimeofday(&tb, nullptr);
        prelude_s_ = (tb.tv_sec - ta.tv_sec) + (tb.tv_usec - ta.tv_usec) / 1e6;
        if (getenv("MIC_CLI_TIMING")) std::cerr << "[timing] inflate: " << inf.size() / 1e6 << " MB of text in " << prelude_s_ * 1e3 << " ms" << std::endl;
        if (inf.size() == 0) { prelude_s_ = 0; std::cerr << "Failed to open " << obj << std::endl; return; }
        FileFeeder feed(inf.path());
        if (!feed.ok()) { prelude_s_ = 0; std::cerr << "Failed to open " << obj << std::endl; return; }
        if (feed.first_byte() != '>' && feed.first_byte() != '@') { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
        run_stream(feed, res, false, (size_t)feed.size());
        prelude_s_ = 0;
        return;
      }
    }
    if (is_gzip(obj)) {
      GzSource src(obj, segment_bytes_);
      if (!src.ok()) { std::cerr << "Failed to uncompress input objects." << std::endl; return; }
      if (device_ingest()) { SegmentFeeder feed(src); run_stream(feed, res, false, ~(size_t)0 >> 1); }
      else run_segments(src, res, false);
      return;
    }
    if (device_ingest()) {
      FileFeeder feed(obj);
      if (!feed.ok()) { std::cerr << "Failed to open " << obj << std::endl; return; }
      if (feed.first_byte() != '>' && feed.first_byte() != '@') { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
      run_stream(feed, res, false, (size_t)feed.size());
      return;
    }
    MmapSource src(obj, segment_bytes_);
    if (!src.ok()) { std::cerr << "Failed to open " << obj << std::endl; return; }
    run_segments(src, res, false);
  };
  if (!file_exists(results)) {
    std::cout << "Processing file '" << objects << "' in " << opt_.batches << " batches using " << opt_.threads
              << " CPU thread(s)." << std::endl;
    simple(objects, results);
    return;
  }
  GzLines in(objects);
  std::string line;
  in.line(line);
  std::vector<std::string> ele = split_seps(line, " \t,");
  if ((!line.empty() && (line[0] == '>' || line[0] == '@')) || ele.size() == 2) {
    std::cout << "Processing file'" << objects << "' in " << opt_.batches << " batches using " << opt_.threads
              << " CPU thread(s)." << std::endl;
    simple(objects, results);
    return;
  }
  // list-of-files mode: objects and results name two parallel lists (CuCLARK_hh.hh:413-427)
  std::ifstream o_fd(objects), r_fd(results);
  std::string o_line, r_line;
  std::cout << "Using " << opt_.threads << " CPU thread(s)." << std::endl;
  while (get_line(o_fd, o_line) && get_line(r_fd, r_line)) {
    std::cout << "> Processing file '" << o_line << "' in " << opt_.batches << " batches." << std::endl;
    simple(o_line, r_line);
  }
}

void Classifier::run_paired(const std::string& f1, const std::string& f2, const std::string& results) {
  auto one = [&](const std::string& a, const std::string& b, const std::string& res, bool list_mode) {
    const std::string merged_name = a + "_ConcatenatedByCLARK.fa";   // the reference's temporary file (CuCLARK_hh.hh:445-446)
    if (list_mode) std::cout << "> Processing file: '" << merged_name << "' in " << opt_.batches << " batches." << std::endl;
    else std::cout << "Processing file: '" << merged_name << "' in " << opt_.batches << " batches using " << opt_.threads
                   << " CPU thread(s)." << std::endl;
    if (device_ingest() && is_gzip(a) && is_gzip(b) && engines_.size() > 1 && !gz_on_device_ && !getenv("MIC_GZ_HOST"))
      std::cerr << "Note: the compressed input is inflated on the host (no peer access between the devices in use)." << std::endl;
    if (device_ingest() && is_gzip(a) && is_gzip(b) && gz_on_device_ && !getenv("MIC_SERIAL_PAIRS") && !getenv("MIC_GZ_HOST")) {
      // both mates compressed: inflated, paired up and merged on the device (MIC_GZ_HOST=1: on the host, below)
      struct timeval ta, tb;
      gettimeofday(&ta, nullptr);
      DeviceGzFeeder feed(engines_[0], a, b);
      if (feed.ok()) {
        gettimeofday(&tb, nullptr);
        prelude_s_ = (tb.tv_sec - ta.tv_sec) + (tb.tv_usec - ta.tv_usec) / 1e6;
        const bool done = run_stream(feed, res, true, (size_t)feed.text_bytes());
        prelude_s_ = 0;
        mic_gz_release(engines_[0]);
        if (done) return;
      } else {
        if (getenv("MIC_CLI_TIMING")) std::cerr << "[timing] device inflate: not used (" << feed.why() << ")" << std::endl;
        mic_gz_release(engines_[0]);
      }
    }
    if (device_ingest() && (is_gzip(a) || is_gzip(b)) && !getenv("MIC_SERIAL_PAIRS")) {
      // compressed mates: both inflated up front and at the same time, then merged by the loaders like plain files
      struct timeval ta, tb;
      gettimeofday(&ta, nullptr);
      InflatedFile ia, ib;
      int ra = 1, rb = 1;
      const bool ga = is_gzip(a), gb = is_gzip(b);
      const unsigned th = inflate_threads(opt_.threads, (ga ? 1u : 0u) + (gb ? 1u : 0u));
      std::thread tb_thread;
      if (gb) tb_thread = std::thread([&] { rb = ib.inflate(b, th); });
      if (ga) ra = ia.inflate(a, th);
      if (tb_thread.joinable()) tb_thread.join();
      if ((ga && ra < 0) || (gb && rb < 0)) die("Failed to uncompress input objects.");
      if ((!ga || ra == 0) && (!gb || rb == 0)) {
        gettimeofday(&tb, nullptr);
        prelude_s_ = (tb.tv_sec - ta.tv_sec) + (tb.tv_usec - ta.tv_usec) / 1e6;
        if (getenv("MIC_CLI_TIMING")) std::cerr << "[timing] inflate: " << ((ga ? ia.size() : 0) + (gb ? ib.size() : 0)) / 1e6 << " MB of text in " << prelude_s_ * 1e3
                                                << " ms (" << th << " threads per file)" << std::endl;
        PairedFileFeeder feed(ga ? ia.path() : a, gb ? ib.path() : b, (unsigned)opt_.threads);
        const bool done = feed.ok() && run_stream(feed, res, true, (size_t)feed.merged_estimate());
        prelude_s_ = 0;
        if (done) return;
      }
    }
    if (device_ingest() && !is_gzip(a) && !is_gzip(b) && !getenv("MIC_SERIAL_PAIRS")) {
      // two plain FASTQ files: the loaders merge the pair in parallel; files that need the reference's line-by-line
      // treatment come back here
      PairedFileFeeder feed(a, b, (unsigned)opt_.threads);
      if (feed.ok() && run_stream(feed, res, true, (size_t)feed.merged_estimate())) return;
    }
    PairedSource src(a, b, segment_bytes_);
    if (!src.ok()) { std::cerr << "Failed to open " << merged_name << std::endl; return; }
    if (device_ingest()) { SegmentFeeder feed(src); run_stream(feed, res, true, ~(size_t)0 >> 1); }
    else run_segments(src, res, true);
  };
  bool list_mode = false;
  if (file_exists(results)) {
    GzLines in(f1);
    std::string line;
    in.line(line);
    std::vector<std::string> ele = split_seps(line, " \t,");
    list_mode = !((!line.empty() && (line[0] == '>' || line[0] == '@')) || ele.size() == 2);
  }
  if (!list_mode) { one(f1, f2, results, false); return; }
  std::ifstream o1(f1), o2(f2), r_fd(results);
  std::string a, b, r;
  std::cout << "Using " << opt_.threads << " CPU thread(s)." << std::endl;
  while (get_line(o1, a) && get_line(o2, b) && get_line(r_fd, r)) one(a, b, r, true);
}

void Classifier::run_buffer(const uint8_t* map, size_t nb, const std::string& results_base, bool paired) {
  OneBuffer src(map, nb);
  run_segments(src, results_base, paired);
}

void Classifier::release_batches() {
  for (mic_engine* e : engines_) mic_batches_free(e);
  lent_.clear();
  slot_reads_ = slot_cont_ = 0;
}

// batch slots are allocated once and reused by every segment; they grow when a segment needs more
void Classifier::ensure_batches(size_t max_reads, size_t max_cont) {
  if (!lent_.empty() && max_reads <= slot_reads_ && max_cont <= slot_cont_) return;
  release_batches();
  const size_t n_eng = engines_.size();
  slots_per_engine_ = std::max<size_t>(1, (opt_.batches + groups_ - 1) / groups_);
  slot_reads_ = max_reads + max_reads / 8 + 64;
  slot_cont_ = max_cont + max_cont / 8 + 64;
  row_words_ = opt_.extended ? (uint32_t)std::min<size_t>(names_.size() + 1, 65) : 16;
  lent_.resize(n_eng);
  std::vector<uint32_t> index(slots_per_engine_ + 1);
  for (size_t i = 0; i <= slots_per_engine_; ++i) index[i] = (uint32_t)(i * slot_reads_);   // fixed stride: slot i owns rows [i*S, (i+1)*S)
  for (size_t d = 0; d < n_eng; ++d) {
    Lent& L = lent_[d];
    L.rp.resize(slots_per_engine_); L.ct.resize(slots_per_engine_);
    check(mic_batches_alloc(engines_[d], slots_per_engine_ * slot_reads_, slot_reads_, slot_cont_, index.data(),
                            (opt_.extended || parts_ > 1) ? 1 : 0,
                            &L.results, &L.rows, L.rp.data(), L.ct.data()), "batch allocation");
  }
}

void Classifier::run_segments(SegmentSource& src, const std::string& results_base, bool paired) {
  const std::string csv = results_base + ".csv";  // CuCLARK_hh.hh:539-540
  FILE* fout = fopen(csv.c_str(), "w");
  if (!fout) { std::cerr << "Failed to create/open file result: " << csv << std::endl; return; }
  struct timeval t0, t1;
  gettimeofday(&t0, nullptr);
  n_objects_ = 0;
  {  // header (CuCLARK_hh.hh:1957-1972)
    std::vector<const char*> nm(names_.size());
    size_t cap = 256;
    for (size_t t = 0; t < names_.size(); ++t) { nm[t] = names_[t].c_str(); cap += names_[t].size() + 2; }
    std::vector<char> hb(cap);
    int w = mic_csv_header(hb.data(), hb.size(), opt_.extended ? 1 : 0, nm.data(), (uint32_t)names_.size());
    if (w > 0) fwrite(hb.data(), 1, (size_t)w, fout);
  }
  // double buffering: segment i+1 is produced on a side thread while segment i is classified
  Segment cur, nxt;
  bool have = src.next(cur);
  std::string err;
  while (have) {
    bool have_next = false;
    std::string reader_err;
    std::thread reader([&] {
      try { have_next = src.next(nxt); } catch (const std::exception& ex) { reader_err = ex.what(); }
    });
    try { n_objects_ += process_segment(cur.p, cur.n, paired, fout); } catch (const std::exception& ex) { if (err.empty()) err = ex.what(); }
    reader.join();
    if (err.empty() && !reader_err.empty()) err = reader_err;
    if (!err.empty()) break;
    std::swap(cur, nxt);
    if (!cur.own.empty()) cur.p = (const uint8_t*)cur.own.data();
    nxt = Segment();
    have = have_next;
  }
  fclose(fout);
  release_batches();
  if (!err.empty()) die(err);
  gettimeofday(&t1, nullptr);
  const double diff = (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) / 1000000.0;
  std::cout << " - Assignment time: " << diff << " s. Speed: ";  // CuCLARK_hh.hh:1938-1944
  std::cout << (size_t)(((double)n_objects_) / (diff) * 60.0) << " objects/min. (" << n_objects_ << " objects)." << std::endl;
  std::cout << " - Results stored in " << csv << std::endl;
}

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

size_t Classifier::process_segment(const uint8_t* map, size_t nb, bool paired, FILE* fout) {
  struct timeval t0;
  gettimeofday(&t0, nullptr);
  const bool timing = getenv("MIC_CLI_TIMING") != nullptr;
  double last = 0;
  auto lap = [&](const char* what) {
    if (!timing) return;
    struct timeval t; gettimeofday(&t, nullptr);
    double now = (t.tv_sec - t0.tv_sec) + (t.tv_usec - t0.tv_usec) / 1e6;
    std::cerr << "[timing] " << what << ": " << (now - last) << " s" << std::endl;
    last = now;
  };
  // ---- index (CuCLARK_hh.hh:1339-1534)
  if (nb == 0 || (map[0] != '>' && map[0] != '@')) { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
  size_t cap = std::max<size_t>(1024, nb / 96);
  // index arrays live across segments: resizing a fresh vector zero-fills ~200 MB per 512 MB segment
  std::vector<uint64_t>&name_s = ix_[0], &name_e = ix_[1], &seq_s = ix_[2], &seq_e = ix_[3], &length = ix_[4];
  if (name_s.size() > cap) cap = name_s.size();
  long n_reads;
  for (;;) {
    if (name_s.size() < cap) { name_s.resize(cap); name_e.resize(cap); seq_s.resize(cap); seq_e.resize(cap); length.resize(cap); }
    n_reads = mic_index_reads_parallel(map, nb, (int)opt_.threads, cap, name_s.data(), name_e.data(), seq_s.data(), seq_e.data(),
                                       length.data());
    if (n_reads < 0) { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
    if ((size_t)n_reads <= cap) break;
    cap = (size_t)n_reads;
  }
  const size_t N = (size_t)n_reads;
  lap("index reads");
  const int k = (int)opt_.k;
  const size_t nb_total = std::max<size_t>(1, std::min(opt_.batches, std::max<size_t>(N, 1)));
  const size_t per = (N + nb_total - 1) / nb_total;
  std::vector<size_t> cut(nb_total + 1);
  for (size_t b = 0; b <= nb_total; ++b) cut[b] = std::min(N, b * per);
  size_t max_reads = 0, max_cont = 0;
  {
    std::vector<size_t> bound(nb_total);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long b = 0; b < (long)nb_total; ++b)
      bound[b] = mic_pack_bound(seq_s.data() + cut[b], seq_e.data() + cut[b], cut[b + 1] - cut[b], k);
    for (size_t b = 0; b < nb_total; ++b) {
      max_reads = std::max(max_reads, cut[b + 1] - cut[b]);
      max_cont = std::max(max_cont, bound[b]);
    }
  }
  ensure_batches(max_reads, max_cont);
  lap("batch slots");

  // ---- batches: pack -> query -> wait -> format; ordered write
  std::vector<std::string> out(nb_total);
  std::vector<char> ready(nb_total, 0);
  std::mutex wmu;
  size_t next_write = 0;
  std::string err;
  const uint32_t T = (uint32_t)names_.size();
  std::vector<const char*> nm(names_.size());
  for (size_t t = 0; t < names_.size(); ++t) nm[t] = names_[t].c_str();
  const uint32_t row_words = row_words_;
  const size_t line_cap = 512 + (opt_.extended ? (size_t)T * 12 : 0);

  double t_pack = 0, t_query = 0, t_format = 0, t_write = 0;   // thread-seconds, MIC_CLI_TIMING only
  auto now_s = [] { struct timeval t; gettimeofday(&t, nullptr); return t.tv_sec + t.tv_usec / 1e6; };
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) reduction(+ : t_pack, t_query, t_format, t_write)
#endif
  for (long bi = 0; bi < (long)nb_total; ++bi) {
    double ts = timing ? now_s() : 0;
    auto tick = [&](double& acc) { if (timing) { const double n = now_s(); acc += n - ts; ts = n; } };
    // batch b belongs to read group b % groups_: one engine (table replicated), or the parts_ engines that hold the table's parts
    const bool sharded = parts_ > 1;
    const size_t b = (size_t)bi, grp = b % groups_, d = grp * parts_, lb = b / groups_;
    mic_engine* const* group = engines_.data() + d;
    Lent& L = lent_[d];
    const size_t r0 = cut[b], cnt = cut[b + 1] - cut[b];
    try {
      size_t m = mic_pack_reads(map, seq_s.data() + r0, seq_e.data() + r0, length.data() + r0, cnt, k, L.rp[lb], L.ct[lb], slot_cont_);
      if (m == (size_t)-1) die("ERROR: Batch overflow. Please increase the number of batches (-b <numberofbatches>).");
      tick(t_pack);
      if (!sharded) {
        check(mic_batch_ready(engines_[d], lb, cnt, m), "readyBatch");
        check(mic_batch_query(engines_[d], lb, opt_.extended ? 1 : 0, 0), "queryBatch");
        check(mic_batch_wait(engines_[d], lb), "waitForBatch");
      } else {
        // every engine of the group probes the same reads against its part of the table - one upload into the first engine, the packed
        // reads fanned out device to device (mic_batch_query_group; the reference uploads the host arrays to every device,
        // CuClarkDB.cu:886-890) - and the rows are summed read-range owned into the first engine's host arrays (mic_batch_merge_shards)
        check(mic_batch_ready(group[0], lb, cnt, m), "readyBatch");
        check(mic_batch_query_group(group, parts_, lb, 1), "queryBatch");
        check(mic_batch_merge_shards(group, parts_, lb), "merge of the table shards");
      }
      tick(t_query);
      std::string& s = out[b];
      s.reserve(cnt * (opt_.extended ? 64 + 3 * (size_t)T : 72));
      std::vector<char> line(line_cap);
      std::vector<uint32_t> dense;
      const uint32_t* res = L.results + lb * slot_reads_ * MIC_RESULT_WORDS;
      const uint32_t* rows = L.rows ? L.rows + lb * slot_reads_ * row_words : nullptr;
      for (size_t i = 0; i < cnt; ++i) {
        const size_t r = r0 + i;
        const uint32_t* row = rows ? rows + i * row_words : nullptr;
        const uint32_t* dn = nullptr;
        const uint32_t* rr = res + i * MIC_RESULT_WORDS;
        uint32_t fixed[MIC_RESULT_WORDS];
        if (row && row[0] == MIC_ROW_INVALID) {
          dense.resize(T);
          if (!sharded) {
            check(mic_batch_dense_counts(engines_[d], lb, i, dense.data()), "dense counts");
          } else {
            // more targets than a sparse row holds: dense counts of every shard, summed; best / second-best under the
            // reference's order (count descending, target ascending)
            std::vector<uint32_t> part(T);
            std::fill(dense.begin(), dense.end(), 0u);
            for (size_t g = 0; g < parts_; ++g) {
              check(mic_batch_dense_counts(group[g], lb, i, part.data()), "dense counts");
              for (uint32_t t2 = 0; t2 < T; ++t2) dense[t2] += part[t2];
            }
            uint32_t sum = 0, best = 0, ib = 0, sb = 0, is = 0, hit = 0;
            for (uint32_t t2 = 0; t2 < T; ++t2) {
              const uint32_t sc = dense[t2];
              if (!sc) continue;
              ++hit; sum += sc;
              if (sc > best) { sb = best; is = ib; best = sc; ib = t2 + 1; }
              else if (sc > sb) { sb = sc; is = t2 + 1; }
            }
            fixed[0] = sum; fixed[1] = ib; fixed[2] = best; fixed[3] = is; fixed[4] = sb; fixed[5] = hit; fixed[6] = rr[6]; fixed[7] = 0;
            rr = fixed;
          }
          dn = dense.data();
        }
        int w = mic_csv_line(line.data(), line.size(), map + name_s[r], (size_t)(name_e[r] - name_s[r]), length[r], paired ? 1 : 0,
                             k, rr, nm.data(), T, opt_.extended ? 1 : 0, row, dn);
        if (w < 0) die("CSV line too long");
        s.append(line.data(), (size_t)w);
      }
    } catch (const std::exception& ex) {
      std::lock_guard<std::mutex> lk(wmu);
      if (err.empty()) err = ex.what();
    }
    tick(t_format);
    std::lock_guard<std::mutex> lk(wmu);
    ready[b] = 1;
    while (next_write < nb_total && ready[next_write]) {
      if (sink_) sink_->append(out[next_write]);
      else fwrite(out[next_write].data(), 1, out[next_write].size(), fout);
      std::string().swap(out[next_write]);
      ++next_write;
    }
    tick(t_write);
  }
  lap("pack + query + format + write");
  if (timing)
    std::cerr << "[timing]   thread-seconds: pack " << t_pack << ", copy+query+wait " << t_query << ", format " << t_format
              << ", ordered write " << t_write << std::endl;
  if (!err.empty()) die(err);
  return N;
}

}  // namespace mic
