// classifier.cpp — see classifier.hpp.  Host pipeline per input file:
//   mmap -> index reads (mic_index_reads) -> split into batches -> [OpenMP over batches] pack into the engine's
//   pinned buffers (mic_pack_reads) -> mic_batch_query (async H2D + kernels + D2H on the batch's stream) ->
//   mic_batch_wait -> format CSV lines -> ordered write.
// Multi-device: one engine per GPU with the whole table resident; batches are dealt round-robin (reads are
// independent), results are written in file order.
#include "classifier.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>

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

// whole file in memory, inflating gzip input (what classify_metagenome.sh --gzipped does with cp + gunzip,
// classify_metagenome.sh:116-142, without the temporary copy)
bool slurp(const std::string& p, std::string& out) {
  out.clear();
  gzFile g = gzopen(p.c_str(), "rb");   // reads plain files transparently too
  if (!g) return false;
  gzbuffer(g, 1 << 20);
  std::vector<char> buf(8u << 20);
  int n;
  while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
  gzclose(g);
  return n == 0;
}

}  // namespace

std::string merge_paired(const std::string& file1, const std::string& file2) {
  std::string b1, b2;
  if (!slurp(file1, b1) || !slurp(file2, b2)) die("Error: Found read without sequence");
  std::istringstream f1(b1), f2(b2);
  std::string l1, l2, out;
  out.reserve(b1.size() / 2 + b2.size() / 2 + 64);
  if (!get_line(f1, l1) || !get_line(f2, l2)) die("Error: Found read without sequence");
  if (l1.empty() || l2.empty() || l1[0] != l2[0]) die("Error: the files have different format!");
  if (l1[0] != '@') die("Error: paired-end reads must be FASTQ files!");
  const std::string seps = " /\t@";
  f1.clear(); f1.seekg(0); f2.clear(); f2.seekg(0);
  while (get_line(f1, l1) && get_line(f2, l2)) {
    if (l1.empty() || l2.empty() || l1[0] != '@' || l2[0] != '@') continue;
    std::vector<std::string> e1 = split_seps(l1, seps), e2 = split_seps(l2, seps);
    if (e1.empty() || e2.empty() || e1[0] != e2[0]) die("Error: read id does not match between files!");
    out += ">" + e1[0] + "\n";
    if (!(get_line(f1, l1) && get_line(f2, l2))) die("Error: Found read without sequence");
    out += l1 + "N" + l2 + "\n";  // NBN = 1 separator (parameters.hh:41)
    if (get_line(f1, l1) && get_line(f2, l2)) { get_line(f1, l1); get_line(f2, l2); }
  }
  return out;
}

Classifier::Classifier(const Options& opt) : opt_(opt) {
#ifdef _OPENMP
  omp_set_num_threads((int)opt_.threads);
#else
  opt_.threads = 1;
#endif
  parse_targets();
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
    int rc = mic_db_build(files.data(), labs.data(), files.size(), (int)opt_.k, opt_.htsize, 0, opt_.min_count_t, db.c_str(), 0,
                          (int)opt_.threads, 0, &n_kmers);
    if (rc != MIC_OK) die(std::string("Failed to create the database: ") + mic_db_build_error());
    std::cerr << "Creating database in disk..." << std::endl;
    std::cerr << n_kmers << " " << opt_.k << "-mers successfully stored in database." << std::endl;
  }
  int n_dev = 0;
  check(mic_device_count(&n_dev), "device discovery");
  if (n_dev == 0) die("No HIP device found.");
  size_t use = opt_.devices == 0 ? (size_t)n_dev : std::min(opt_.devices, (size_t)n_dev);
  if (opt_.batches < use) use = std::max<size_t>(1, opt_.batches);
  std::cerr << "Loading database [" << db << ".*] (s=" << opt_.sampling << ")..." << std::endl;
  const size_t per_engine_batches = (opt_.batches + use - 1) / use;
  for (size_t d = 0; d < use; ++d) {
    mic_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = (int)d; cfg.k = (int)opt_.k; cfg.num_targets = (uint32_t)(names_.size());
    cfg.num_batches = (uint32_t)per_engine_batches;
    cfg.row_words = opt_.extended ? (uint32_t)std::min<size_t>(names_.size() + 1, 65) : 16;
    mic_engine* e = nullptr;
    check(mic_create(&cfg, &e), "engine creation");
    engines_.push_back(e);
    int rc = mic_db_load_files(e, db.c_str(), 0, opt_.sampling, 0, 0);
    if (rc != MIC_OK) die(std::string("Failed to load the database: ") + mic_last_error());
  }
  mic_db_info info;
  check(mic_db_get_info(engines_[0], &info), "db info");
  std::cerr << "Total DB size in HBM:\t" << info.hbm_bytes / 1000000 / 1000.0 << " GB (" << info.n_elems << " k-mers, "
            << info.n_overflow << " overflow slots) on " << use << " device(s)\n";
}

Classifier::~Classifier() {
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

void Classifier::run(const std::string& objects, const std::string& results) {
  auto simple = [&](const std::string& obj, const std::string& res) {
    if (is_gzip(obj)) {
      std::string data;
      if (!slurp(obj, data) || data.empty()) { std::cerr << "Failed to uncompress input objects." << std::endl; return; }
      run_buffer((const uint8_t*)data.data(), data.size(), res, false);
      return;
    }
    int fd = open(obj.c_str(), O_RDONLY);
    struct stat st;
    if (fd == -1 || fstat(fd, &st) != 0 || st.st_size == 0) {
      if (fd != -1) close(fd);
      std::cerr << "Failed to open " << obj << std::endl;
      return;
    }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); std::cerr << "Failed to mmapping the file." << std::endl; return; }
    madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
    run_buffer((const uint8_t*)map, (size_t)st.st_size, res, false);
    munmap(map, (size_t)st.st_size);
    close(fd);
  };
  if (!file_exists(results)) {
    std::cout << "Processing file '" << objects << "' in " << opt_.batches << " batches using " << opt_.threads
              << " CPU thread(s)." << std::endl;
    simple(objects, results);
    return;
  }
  std::ifstream in(objects);
  std::string line;
  get_line(in, line);
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
    const std::string merged_name = a + "_ConcatenatedByCLARK.fa";
    std::string merged = merge_paired(a, b);
    if (list_mode) std::cout << "> Processing file: '" << merged_name << "' in " << opt_.batches << " batches." << std::endl;
    else std::cout << "Processing file: '" << merged_name << "' in " << opt_.batches << " batches using " << opt_.threads
                   << " CPU thread(s)." << std::endl;
    if (merged.empty()) { std::cerr << "Failed to open " << merged_name << std::endl; return; }
    run_buffer((const uint8_t*)merged.data(), merged.size(), res, true);
  };
  bool list_mode = false;
  if (file_exists(results)) {
    std::ifstream in(f1);
    std::string line;
    get_line(in, line);
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
  const std::string csv = results_base + ".csv";  // CuCLARK_hh.hh:539-540
  FILE* fout = fopen(csv.c_str(), "w");
  if (!fout) { std::cerr << "Failed to create/open file result: " << csv << std::endl; return; }
  struct timeval t0, t1;
  gettimeofday(&t0, nullptr);
  const bool timing = getenv("MIC_CLI_TIMING") != nullptr;
  auto lap = [&](const char* what) {
    if (!timing) return;
    struct timeval t; gettimeofday(&t, nullptr);
    static double last = 0;
    double now = (t.tv_sec - t0.tv_sec) + (t.tv_usec - t0.tv_usec) / 1e6;
    std::cerr << "[timing] " << what << ": " << (now - last) << " s (t=" << now << ")" << std::endl;
    last = now;
  };

  // ---- index (CuCLARK_hh.hh:1339-1534)
  if (map[0] != '>' && map[0] != '@') { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
  size_t cap = std::max<size_t>(1024, nb / 96);
  std::vector<uint64_t> name_s, name_e, seq_s, seq_e, length;
  long n_reads;
  for (;;) {
    name_s.resize(cap); name_e.resize(cap); seq_s.resize(cap); seq_e.resize(cap); length.resize(cap);
    n_reads = mic_index_reads_parallel(map, nb, (int)opt_.threads, cap, name_s.data(), name_e.data(), seq_s.data(), seq_e.data(),
                                       length.data());
    if (n_reads < 0) { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
    if ((size_t)n_reads <= cap) break;
    cap = (size_t)n_reads;
  }
  n_objects_ = (size_t)n_reads;
  lap("index reads");
  const size_t N = n_objects_;
  const int k = (int)opt_.k;
  const size_t n_eng = engines_.size();
  const size_t nb_total = std::max<size_t>(1, std::min(opt_.batches, std::max<size_t>(N, 1)));
  const size_t per = (N + nb_total - 1) / nb_total;
  std::vector<size_t> cut(nb_total + 1);
  for (size_t b = 0; b <= nb_total; ++b) cut[b] = std::min(N, b * per);

  // ---- per-engine batch tables
  struct Lent { uint32_t* results = nullptr; uint32_t* rows = nullptr; std::vector<uint32_t*> rp; std::vector<uint16_t*> ct;
                std::vector<uint32_t> index; };
  std::vector<Lent> lent(n_eng);
  size_t max_reads = 0, max_cont = 0;
  for (size_t b = 0; b < nb_total; ++b) {
    max_reads = std::max(max_reads, cut[b + 1] - cut[b]);
    max_cont = std::max(max_cont, mic_pack_bound(seq_s.data() + cut[b], seq_e.data() + cut[b], cut[b + 1] - cut[b], k));
  }
  const size_t local_batches = (nb_total + n_eng - 1) / n_eng;
  for (size_t d = 0; d < n_eng; ++d) {
    Lent& L = lent[d];
    L.index.assign(local_batches + 1, 0);
    for (size_t lb = 0; lb < local_batches; ++lb) {
      size_t b = lb * n_eng + d;
      size_t cnt = b < nb_total ? cut[b + 1] - cut[b] : 0;
      L.index[lb + 1] = L.index[lb] + (uint32_t)cnt;
    }
    // the engine owns ceil(batches/devices) batch slots; a file with fewer reads than batches leaves some empty
    L.rp.resize(local_batches); L.ct.resize(local_batches);
    std::vector<uint32_t> full_index(((opt_.batches + n_eng - 1) / n_eng) + 1, L.index.back());
    for (size_t i = 0; i < L.index.size(); ++i) full_index[i] = L.index[i];
    std::vector<uint32_t*> rp(full_index.size() - 1); std::vector<uint16_t*> ct(full_index.size() - 1);
    check(mic_batches_alloc(engines_[d], L.index.back(), max_reads, max_cont, full_index.data(), opt_.extended ? 1 : 0,
                            &L.results, &L.rows, rp.data(), ct.data()), "batch allocation");
    for (size_t lb = 0; lb < local_batches; ++lb) { L.rp[lb] = rp[lb]; L.ct[lb] = ct[lb]; }
  }

  lap("allocate batches");
  // ---- header
  {
    std::vector<const char*> nm(names_.size());
    for (size_t t = 0; t < names_.size(); ++t) nm[t] = names_[t].c_str();
    std::vector<char> hb(256 + names_.size() * 64);
    for (size_t t = 0; t < names_.size(); ++t) hb.resize(hb.size() + names_[t].size());
    int w = mic_csv_header(hb.data(), hb.size(), opt_.extended ? 1 : 0, nm.data(), (uint32_t)names_.size());
    if (w > 0) fwrite(hb.data(), 1, (size_t)w, fout);
  }

  // ---- batches: pack -> query -> wait -> format; ordered write
  std::vector<std::string> out(nb_total);
  std::vector<char> ready(nb_total, 0);
  std::mutex wmu;
  size_t next_write = 0;
  std::string err;
  const uint32_t T = (uint32_t)names_.size();
  std::vector<const char*> nm(names_.size());
  for (size_t t = 0; t < names_.size(); ++t) nm[t] = names_[t].c_str();
  const uint32_t row_words = opt_.extended ? (uint32_t)std::min<size_t>(names_.size() + 1, 65) : 16;
  const size_t line_cap = 512 + (opt_.extended ? (size_t)T * 12 : 0);

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (long bi = 0; bi < (long)nb_total; ++bi) {
    const size_t b = (size_t)bi, d = b % n_eng, lb = b / n_eng;
    Lent& L = lent[d];
    const size_t r0 = cut[b], cnt = cut[b + 1] - cut[b];
    try {
      size_t m = mic_pack_reads(map, seq_s.data() + r0, seq_e.data() + r0, length.data() + r0, cnt, k, L.rp[lb], L.ct[lb], max_cont);
      if (m == (size_t)-1) die("ERROR: Batch overflow. Please increase the number of batches (-b <numberofbatches>).");
      check(mic_batch_ready(engines_[d], lb, cnt, m), "readyBatch");
      check(mic_batch_query(engines_[d], lb, opt_.extended ? 1 : 0, 0), "queryBatch");
      check(mic_batch_wait(engines_[d], lb), "waitForBatch");
      std::string& s = out[b];
      s.reserve(cnt * (opt_.extended ? 64 + 3 * (size_t)T : 72));
      std::vector<char> line(line_cap);
      std::vector<uint32_t> dense;
      const uint32_t* res = L.results + (size_t)L.index[lb] * MIC_RESULT_WORDS;
      const uint32_t* rows = L.rows ? L.rows + (size_t)L.index[lb] * row_words : nullptr;
      for (size_t i = 0; i < cnt; ++i) {
        const size_t r = r0 + i;
        const uint32_t* row = rows ? rows + i * row_words : nullptr;
        const uint32_t* dn = nullptr;
        if (row && row[0] == MIC_ROW_INVALID) {
          dense.resize(T);
          check(mic_batch_dense_counts(engines_[d], lb, i, dense.data()), "dense counts");
          dn = dense.data();
        }
        int w = mic_csv_line(line.data(), line.size(), map + name_s[r], (size_t)(name_e[r] - name_s[r]), length[r], paired ? 1 : 0,
                             k, res + i * MIC_RESULT_WORDS, nm.data(), T, opt_.extended ? 1 : 0, row, dn);
        if (w < 0) die("CSV line too long");
        s.append(line.data(), (size_t)w);
      }
    } catch (const std::exception& ex) {
      std::lock_guard<std::mutex> lk(wmu);
      if (err.empty()) err = ex.what();
    }
    std::lock_guard<std::mutex> lk(wmu);
    ready[b] = 1;
    while (next_write < nb_total && ready[next_write]) {
      fwrite(out[next_write].data(), 1, out[next_write].size(), fout);
      std::string().swap(out[next_write]);
      ++next_write;
    }
  }
  lap("pack + query + format + write");
  fclose(fout);
  for (mic_engine* e : engines_) mic_batches_free(e);
  lap("free batches");
  if (!err.empty()) die(err);

  gettimeofday(&t1, nullptr);
  const double diff = (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) / 1000000.0;
  std::cout << " - Assignment time: " << diff << " s. Speed: ";  // CuCLARK_hh.hh:1938-1944
  std::cout << (size_t)(((double)n_objects_) / (diff) * 60.0) << " objects/min. (" << n_objects_ << " objects)." << std::endl;
  std::cout << " - Results stored in " << csv << std::endl;
}

}  // namespace mic
