// classifier.cpp - see classifier.hpp.  This file: options, engines (devices, parts of the table, what the run allocates next to
// it, the database load) and the dispatch of an input to its source / feeder.  The streaming path is classifier_stream.cpp, the batch
// path (the reference's flow: index, pack, queryBatch, CSV lines) classifier_batch.cpp, the inputs classifier_feeders.hpp.
#include "classifier_feeders.hpp"

namespace mic {
using namespace detail;

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
        // MIC_GZ_STRIPES=n: the member in n stripes of its deflate blocks, a stripe indexed and classified while the later ones decode
        // (mic_gz_stream_*; round 6).  Built, measured, not the default: the decode holds every wavefront slot of the chip for two rounds of
        // ~10 ms (14 KB of LDS a wavefront: 11 a CU), the first stripe's text is there when nearly all of them are, and the windows kernel
        // of a stripe (128 KB of LDS a block) waits for a CU the decodes have left - 4 M reads: 57-60 ms either way.
        unsigned stripes = 1;
        if (const char* env = getenv("MIC_GZ_STRIPES")) { const long v = atol(env); if (v >= 1 && v <= 64) stripes = (unsigned)v; }
        bool again = false;
        {
          DeviceGzFeeder feed(engines_[0], obj, "", stripes);
          if (feed.ok()) {
            gettimeofday(&tb, nullptr);
            prelude_s_ = (tb.tv_sec - ta.tv_sec) + (tb.tv_usec - ta.tv_usec) / 1e6;
            // a later stripe that does not stitch, holds a second member or fails its checks: the feeder gives up, run_stream removes what it
            // wrote, and everything starts again on the CPU inflater - which takes such files, or reports them the reference's way
            again = !run_stream(feed, res, false, (size_t)feed.text_bytes());
            if (again) std::cerr << "Note: the device inflater gave the file back after its first stripes (" << feed.gave_up_why() << "); inflating on the host." << std::endl;
            prelude_s_ = 0;
            if (!again) { mic_gz_release(engines_[0]); return; }
          } else if (getenv("MIC_CLI_TIMING")) std::cerr << "[timing] device inflate: not used (" << feed.why() << ")" << std::endl;
        }
        mic_gz_release(engines_[0]);
        if (again) gettimeofday(&ta, nullptr);
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

}  // namespace mic
