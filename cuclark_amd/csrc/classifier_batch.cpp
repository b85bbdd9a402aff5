// classifier_batch.cpp - the batch path of the command line, the reference's own flow per input segment: index the reads
// (CuCLARK_hh.hh:1339-1534), pack them into the engine's lent buffers (:1616-1716), mic_batch_query / wait (queryBatch / waitForBatch),
// format the CSV lines (:1951-2139), write in order.  Used for --extended, for batches the device path hands back, and with
// MIC_HOST_INGEST=1.
#include "classifier_internal.hpp"

namespace mic {
using namespace detail;

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
    if (w > 0 && fwrite(hb.data(), 1, (size_t)w, fout) != (size_t)w) { fclose(fout); die("cannot write " + csv + " (disk full?)"); }
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
  // a full disk shows here at the latest (the command line leaves through _exit: nothing later would flush or report it)
  const bool write_failed = ferror(fout) != 0;
  if (fclose(fout) != 0 || write_failed) { if (err.empty()) err = "cannot write " + csv + " (disk full?)"; }
  release_batches();
  if (!err.empty()) die(err);
  gettimeofday(&t1, nullptr);
  const double diff = (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) / 1000000.0;
  std::cout << " - Assignment time: " << diff << " s. Speed: ";  // CuCLARK_hh.hh:1938-1944
  std::cout << (size_t)(((double)n_objects_) / (diff) * 60.0) << " objects/min. (" << n_objects_ << " objects)." << std::endl;
  std::cout << " - Results stored in " << csv << std::endl;
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
    n_reads = mic_index_reads_parallel(map, nb, sink_ ? 1 : (int)opt_.threads, cap, name_s.data(), name_e.data(), seq_s.data(), seq_e.data(),
                                       length.data());
    if (n_reads < 0) { std::cerr << "Failed to recognize the format of the file." << std::endl; exit(-1); }
    if ((size_t)n_reads <= cap) break;
    cap = (size_t)n_reads;
  }
  const size_t N = (size_t)n_reads;
  lap("index reads");
  const int k = (int)opt_.k;
  // A batch the streaming path hands back (sink_ set) is small and comes while that path's loader / device / writer threads are
  // running: no OpenMP team for it - a team's idle threads spin at the barriers and between the regions (libgomp's default wait
  // policy), and next to the stream's threads that spinning ran a 64-read batch into 0.2-0.5 s of cgroup throttling
  // (tools/cli_small_slots_probe.py: 55 such batches 14.2 s, 0.6 s with OMP_WAIT_POLICY=PASSIVE).
  const int team = sink_ ? 1 : (int)std::max<size_t>(1, opt_.threads);
  const size_t nb_total = std::max<size_t>(1, std::min(opt_.batches, std::max<size_t>(N, 1)));
  const size_t per = (N + nb_total - 1) / nb_total;
  std::vector<size_t> cut(nb_total + 1);
  for (size_t b = 0; b <= nb_total; ++b) cut[b] = std::min(N, b * per);
  size_t max_reads = 0, max_cont = 0;
  {
    std::vector<size_t> bound(nb_total);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(team)
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
#pragma omp parallel for schedule(dynamic) reduction(+ : t_pack, t_query, t_format, t_write) num_threads(team)
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
      else if (fwrite(out[next_write].data(), 1, out[next_write].size(), fout) != out[next_write].size() && err.empty()) err = "short write to the result file (disk full?)";
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
