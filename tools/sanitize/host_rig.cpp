// host_rig.cpp - drives the library's HOST code (compiled host-only under ASan + UBSan against hip_mock.cpp) through the call
// sequences of tools/fuzz_parity.py on a machine without a GPU: engines created and destroyed, tables loaded from arrays (whole,
// bucket-range shards, slot-range parts), batches allocated, filled, queried, merged over shards, the ingest slots and the
// table-sharded group ingest, the device inflate with its error paths.  Kernels do not run (the "device" answers with zeros), so
// only status codes and the sanitizers' verdict count: every call may fail, none may touch memory it does not own.
//     host_rig <seconds> <seed>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <random>
#include <string>
#include <vector>

#include "mi_clark.h"

static std::mt19937_64 rng;
static uint64_t rnd(uint64_t lo, uint64_t hi) { return lo + rng() % (hi - lo + 1); }       // inclusive
#include <map>
static unsigned long n_calls = 0, n_fail = 0;
static std::map<std::string, std::pair<unsigned long, unsigned long>> by_call;     // entry point -> calls, of them failed
static bool tally(const char* text, int rc) {
  std::string name(text);
  name = name.substr(0, name.find('('));
  auto& c = by_call[name];
  ++c.first; ++n_calls;
  if (rc != MIC_OK) { ++c.second; ++n_fail; }
  return rc == MIC_OK;
}
#define CALL(x) tally(#x, (x))

struct Db { uint64_t htsize; int key_bytes; std::vector<uint8_t> sizes; std::vector<uint8_t> keys; std::vector<uint16_t> labels; uint64_t n; };

static Db random_db(int k, uint32_t T) {
  static const uint64_t hts[] = {2, 97, 1009, 4096, 65537, 99991};
  Db d;
  d.htsize = hts[rnd(0, 5)];
  d.key_bytes = mic_key_bytes_rule(d.htsize, k);
  d.sizes.assign(d.htsize, 0);
  const uint64_t want = rnd(50, 60000);
  d.n = 0;
  for (uint64_t i = 0; i < want; ++i) { uint8_t& s = d.sizes[rng() % d.htsize]; if (s < 200) { ++s; ++d.n; } }
  d.keys.resize(d.n * (size_t)d.key_bytes);
  for (auto& b : d.keys) b = (uint8_t)rng();
  // keys ascending inside a bucket, as the files hold them
  size_t at = 0;
  for (uint64_t b = 0; b < d.htsize; ++b) {
    for (unsigned i = 0; i < d.sizes[b]; ++i) {
      uint64_t v = (uint64_t)(i + 1) * 1000 + rng() % 999;
      memcpy(&d.keys[(at + i) * (size_t)d.key_bytes], &v, (size_t)d.key_bytes);
    }
    at += d.sizes[b];
  }
  d.labels.resize(d.n);
  for (auto& l : d.labels) l = (uint16_t)(rng() % T);
  return d;
}

static std::string random_fasta(unsigned n_reads, unsigned L) {
  std::string s;
  for (unsigned r = 0; r < n_reads; ++r) {
    s += ">read_" + std::to_string(r) + " x\n";
    const unsigned len = (unsigned)rnd(1, L);
    for (unsigned i = 0; i < len; ++i) s += (rng() % 97 == 0) ? 'N' : "ACGT"[rng() & 3];
    s += '\n';
  }
  return s;
}

static unsigned n_dev = 1;      // MOCK_HIP_DEVICES: the engines of a sharded configuration are spread over that many (mock) devices

static void one_round() {
  static const int ks[] = {8, 12, 16, 20, 21, 24, 25, 27, 31, 32};
  static const uint32_t Ts[] = {1, 2, 7, 40, 64, 65, 300, 4096};
  const int k = ks[rnd(0, 9)];
  const uint32_t T = Ts[rnd(0, 7)];
  const Db db = random_db(k, T);
  const std::string fa = random_fasta((unsigned)rnd(50, 400), (unsigned)rnd((uint64_t)k, 400));
  // index + pack on the host (mic_host.cpp)
  const size_t cap = 1024;
  std::vector<uint64_t> ns(cap), ne(cap), ss(cap), se(cap), ln(cap);
  const long n_reads = mic_index_reads((const uint8_t*)fa.data(), fa.size(), cap, ns.data(), ne.data(), ss.data(), se.data(), ln.data());
  if (n_reads <= 0) return;
  const size_t bound = mic_pack_bound(ss.data(), se.data(), (size_t)n_reads, k);
  std::vector<uint32_t> rp((size_t)n_reads + 1);
  std::vector<uint16_t> cont(bound + 64);
  const size_t n_cont = mic_pack_reads((const uint8_t*)fa.data(), ss.data(), se.data(), ln.data(), (size_t)n_reads, k, rp.data(), cont.data(), cont.size());
  if (n_cont == (size_t)-1) return;

  const uint32_t layout = (uint32_t)rnd(1, 4);
  // how the table is spread over engines: one, bucket-range shards, slot-range parts
  const unsigned mode = (unsigned)rnd(0, 2);
  const unsigned n_eng = mode == 0 ? 1 : (unsigned)rnd(2, 5);
  std::vector<mic_engine*> eng(n_eng, nullptr);
  for (unsigned i = 0; i < n_eng; ++i) {
    mic_config cfg; cfg.device = (int)(i % n_dev); cfg.k = k; cfg.num_targets = T; cfg.num_batches = 1; cfg.row_words = 16; cfg.layout = layout;
    CALL(mic_create(&cfg, &eng[i]));
    if (!eng[i]) { for (unsigned j = 0; j < i; ++j) mic_destroy(eng[j]); return; }
    if (mode == 2) CALL(mic_db_set_part(eng[i], i, n_eng));
    uint64_t a = 0, b = 0;
    if (mode == 1) { a = db.htsize * i / n_eng; b = db.htsize * (i + 1) / n_eng; if (b <= a) b = a + 1; if (b > db.htsize) { a = 0; b = 0; } }
    CALL(mic_db_load_host(eng[i], db.sizes.data(), db.htsize, db.keys.data(), db.key_bytes, db.labels.data(), 1, a, b));
    mic_db_info info;
    CALL(mic_db_get_info(eng[i], &info));
    char name[256];
    { const int len = mic_db_kernel_name(eng[i], name, sizeof(name)); tally("mic_db_kernel_name(", len < 0 ? len : MIC_OK); }     // (returns the name's length)
  }
  // the batch API on every engine (or: one upload into the first engine and the group query), then the merge over shards
  const bool group_query = n_eng > 1 && (rng() & 1);
  for (unsigned i = 0; i < n_eng; ++i) {
    const uint32_t ib[2] = {0, (uint32_t)n_reads};
    uint32_t *res = nullptr, *rows = nullptr, *brp[1] = {nullptr};
    uint16_t* bct[1] = {nullptr};
    if (!CALL(mic_batches_alloc(eng[i], (size_t)n_reads, (size_t)n_reads, n_cont ? n_cont : 1, ib, 1, &res, &rows, brp, bct))) continue;
    memcpy(brp[0], rp.data(), rp.size() * 4);
    memcpy(bct[0], cont.data(), n_cont * 2);
    if (group_query) {
      if (i == 0) CALL(mic_batch_ready(eng[0], 0, (size_t)n_reads, n_cont));
      if (i + 1 == n_eng) CALL(mic_batch_query_group(eng.data(), n_eng, 0, 1));
      continue;
    }
    CALL(mic_batch_ready(eng[i], 0, (size_t)n_reads, n_cont));
    CALL(mic_batch_query(eng[i], 0, 1, 0));
    if (rng() & 1) { int done = 0; CALL(mic_batch_check(eng[i], 0, &done)); }
    if (n_eng == 1 || (rng() & 1)) CALL(mic_batch_wait(eng[i], 0));
    if (rng() % 4 == 0) { std::vector<uint32_t> counts(T); CALL(mic_batch_dense_counts(eng[i], 0, (size_t)rnd(0, (uint64_t)n_reads - 1), counts.data())); }
  }
  if (n_eng > 1) CALL(mic_batch_merge_shards(eng.data(), n_eng, 0));
  // the ingest slots: one engine, or the group of parts from a random owner
  {
    std::vector<std::string> names(T);
    std::vector<const char*> np(T);
    for (uint32_t t = 0; t < T; ++t) { names[t] = "t" + std::to_string(t); np[t] = names[t].c_str(); }
    const size_t owner = (size_t)rnd(0, n_eng - 1);
    const size_t n_slots = (size_t)rnd(1, 3);
    std::vector<uint8_t*> raw(n_slots, nullptr);
    if (CALL(mic_ingest_alloc(eng[owner], n_slots, (size_t)1 << 20, np.data(), T, (int)(rng() & 1), raw.data()))) {
      for (size_t s = 0; s < n_slots; ++s) {
        const size_t nb = fa.size() < ((size_t)1 << 20) ? fa.size() : 0;
        if (!nb || !raw[s]) continue;
        memcpy(raw[s], fa.data(), nb);
        mic_ingest_result out;
        memset(&out, 0, sizeof(out));
        if (mode == 2) CALL(mic_ingest_classify_group(eng.data(), n_eng, owner, s, nb, 0, &out));
        else CALL(mic_ingest_classify(eng[owner], s, nb, 0, &out));
        if (rng() % 3 == 0) {
          std::vector<uint32_t> frp((size_t)n_reads + 8); std::vector<uint16_t> fct(cont.size() + 64);
          uint64_t a = 0, b = 0;
          CALL(mic_ingest_fetch_packed(eng[owner], s, frp.data(), frp.size(), fct.data(), fct.size(), &a, &b));
        }
      }
      CALL(mic_ingest_free(eng[owner]));
    }
  }
  // the device inflate: a real member, a damaged one, not gzip at all (its error paths leave through `goto done`)
  if (rng() % 3 == 0) {
    std::vector<uint8_t> gz(compressBound(fa.size()) + 64);
    z_stream z; memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 1, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY) == Z_OK) {
      z.next_in = (Bytef*)fa.data(); z.avail_in = (uInt)fa.size(); z.next_out = gz.data(); z.avail_out = (uInt)gz.size();
      deflate(&z, Z_FINISH);
      gz.resize(z.total_out);
      deflateEnd(&z);
      const unsigned what = (unsigned)rnd(0, 2);
      if (what == 1 && gz.size() > 40) gz[rnd(20, gz.size() - 10)] ^= 0x5a;
      if (what == 2) gz[0] = 'x';
      void* d_text = nullptr; size_t n_text = 0; uint32_t crc = 0;
      if (rng() & 1) CALL(mic_gz_reserve(eng[0], gz.size(), (uint32_t)fa.size()));
      if (CALL(mic_gz_inflate_device(eng[0], gz.data(), gz.size(), &d_text, &n_text, &crc))) {
        std::vector<uint8_t> back(n_text ? n_text : 1);
        if (n_text) CALL(mic_gz_copy_text(eng[0], d_text, 0, n_text, back.data()));
        CALL(mic_gz_free_text(eng[0], d_text));
      }
      CALL(mic_gz_release(eng[0]));
    }
  }
  // the device-resident entry points ("device" pointers are host memory on the mock) on engine 0
  if (rng() % 2 == 0) {
    const size_t nr = (size_t)n_reads;
    std::vector<uint32_t> d_rp(rp), d_res(nr * MIC_RESULT_WORDS), d_rows(nr * 16), d_rows2(nr * 16), d_out(nr * 16), d_ids(nr), d_counts(nr * (size_t)T < (1u << 22) ? nr * (size_t)T : 1);
    std::vector<uint16_t> d_ct(cont);
    for (size_t i = 0; i < nr; ++i) d_ids[i] = (uint32_t)i;
    CALL(mic_query_device(eng[0], d_rp.data(), d_ct.data(), nr, d_res.data(), d_rows.data(), nullptr));
    size_t n_res = 0;
    CALL(mic_resolve_flagged_device(eng[0], d_rp.data(), d_ct.data(), d_res.data(), d_rows.data(), nullptr, &n_res));
    float ms = 0;
    CALL(mic_last_query_ms(eng[0], &ms));
    CALL(mic_merge_rows_device(eng[0], d_rows.data(), d_rows2.data(), d_out.data(), nr, nullptr));
    CALL(mic_result_from_rows_device(eng[0], d_out.data(), d_res.data(), nr, nullptr));
    if (d_counts.size() > 1) {
      CALL(mic_count_dense_device(eng[0], d_rp.data(), d_ct.data(), d_ids.data(), nr, d_counts.data(), nullptr));
      CALL(mic_result_from_dense_device(eng[0], d_counts.data(), d_ids.data(), nr, d_res.data(), d_rows.data(), nullptr));
    }
    uint64_t st4[4];
    CALL(mic_probe_stats_device(eng[0], d_rp.data(), d_ct.data(), nr, st4));
    CALL(mic_sync(eng[0]));
  }
  // texts resident on the "device": a FASTQ text indexed, cut into slots, copied back; two mates paired up and merged
  if (rng() % 2 == 0) {
    std::string fq1, fq2;
    const unsigned nrec = (unsigned)rnd(1, 300);
    for (unsigned r = 0; r < nrec; ++r) {
      const unsigned len = (unsigned)rnd(1, 200);
      std::string a, b2;
      for (unsigned i = 0; i < len; ++i) { a += "ACGT"[rng() & 3]; b2 += "ACGT"[rng() & 3]; }
      fq1 += "@p" + std::to_string(r) + "/1\n" + a + "\n+\n" + std::string(len, 'I') + "\n";
      fq2 += "@p" + std::to_string(r) + "/2\n" + b2 + "\n+\n" + std::string(len, 'I') + "\n";
    }
    std::vector<std::string> names(T);
    std::vector<const char*> np(T);
    for (uint32_t t = 0; t < T; ++t) { names[t] = "t" + std::to_string(t); np[t] = names[t].c_str(); }
    uint8_t* raw[1] = {nullptr};
    if (CALL(mic_ingest_alloc(eng[0], 1, (size_t)1 << 20, np.data(), T, 0, raw))) {
      mic_text* tx = nullptr; uint64_t nrx = 0; uint32_t stx = 0;
      if (CALL(mic_text_index_device(eng[0], fq1.data(), fq1.size(), &tx, &nrx, &stx)) && tx) {
        const uint64_t* smp = nullptr; size_t nsmp = 0; uint32_t stride = 0;
        CALL(mic_text_offsets(tx, &smp, &nsmp, &stride));
        (void)mic_text_format(tx);
        size_t nb = 0;
        CALL(mic_text_to_slot(eng[0], tx, 0, nrx, 0, &nb));
        std::vector<uint8_t> back(fq1.size() + 64);
        CALL(mic_text_copy(eng[0], tx, 0, nrx, back.data(), back.size(), &nb));
        CALL(mic_text_free(eng[0], tx));
      }
      mic_pairs* pr = nullptr; uint64_t npr = 0; uint32_t stp = 0;
      if (CALL(mic_pairs_index_device(eng[0], fq1.data(), fq1.size(), fq2.data(), fq2.size(), &pr, &npr, &stp)) && pr) {
        const uint64_t* smp = nullptr; size_t nsmp = 0; uint32_t stride = 0;
        CALL(mic_pairs_offsets(pr, &smp, &nsmp, &stride));
        size_t nb = 0;
        CALL(mic_pairs_merge_to_slot(eng[0], pr, 0, npr, 0, &nb));
        std::vector<uint8_t> back(fq1.size() + fq2.size() + 64);
        CALL(mic_pairs_text(eng[0], pr, 0, npr, back.data(), back.size(), &nb));
        CALL(mic_pairs_free(eng[0], pr));
      }
      CALL(mic_ingest_free(eng[0]));
    }
  }
  // the same table from FILES: one engine, and several engines from one read of the files (mic_db_load_files_multi)
  if (rng() % 4 == 0) {
    char prefix[256];
    snprintf(prefix, sizeof(prefix), "%s/host_rig_%d_db", getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp", (int)getpid());
    bool ok = true;
    { FILE* f = fopen((std::string(prefix) + ".sz").c_str(), "wb"); ok = ok && f && fwrite(db.sizes.data(), 1, db.sizes.size(), f) == db.sizes.size(); if (f) fclose(f); }
    { FILE* f = fopen((std::string(prefix) + ".ky").c_str(), "wb"); ok = ok && f && fwrite(db.keys.data(), 1, db.keys.size(), f) == db.keys.size(); if (f) fclose(f); }
    { FILE* f = fopen((std::string(prefix) + ".lb").c_str(), "wb"); ok = ok && f && fwrite(db.labels.data(), 2, db.labels.size(), f) == db.labels.size(); if (f) fclose(f); }
    if (ok) {
      const unsigned n2 = (unsigned)rnd(1, 4);
      std::vector<mic_engine*> e2(n2, nullptr);
      bool made = true;
      for (unsigned i = 0; i < n2; ++i) {
        mic_config cfg; cfg.device = (int)(i % n_dev); cfg.k = k; cfg.num_targets = T; cfg.num_batches = 1; cfg.row_words = 16; cfg.layout = layout;
        made = made && CALL(mic_create(&cfg, &e2[i])) && e2[i];
        if (made && n2 > 1) CALL(mic_db_set_part(e2[i], i, n2));
      }
      if (made) {
        if (n2 == 1) CALL(mic_db_load_files(e2[0], prefix, rng() & 1 ? db.key_bytes : 0, 1, 0, 0));
        else CALL(mic_db_load_files_multi(e2.data(), n2, prefix, db.key_bytes, 1));
        CALL(mic_db_unload(e2[0]));
      }
      for (unsigned i = 0; i < n2; ++i) if (e2[i]) CALL(mic_destroy(e2[i]));
    }
    for (const char* ext : {".sz", ".ky", ".lb"}) remove((std::string(prefix) + ext).c_str());
  }
  // free in a random order; sometimes the batches explicitly first
  for (unsigned i = 0; i < n_eng; ++i) if (rng() & 1) CALL(mic_batches_free(eng[i]));
  for (unsigned i = n_eng; i-- > 0;) { const unsigned j = (unsigned)rnd(0, i); std::swap(eng[i], eng[j]); CALL(mic_destroy(eng[i])); }
}

int main(int argc, char** argv) {
  const double budget = argc > 1 ? atof(argv[1]) : 10.0;
  const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
  rng.seed(seed);
  { int c = 1; if (mic_device_count(&c) == MIC_OK && c > 0) n_dev = (unsigned)c; }
  { std::vector<int> mat((size_t)n_dev * n_dev); tally("mic_peer_matrix(", mic_peer_matrix(mat.data(), (int)n_dev)); }
  const time_t t_end = time(nullptr) + (time_t)budget;
  unsigned long rounds = 0;
  while (time(nullptr) < t_end) { one_round(); ++rounds; }
  printf("host rig ok: %lu rounds, %lu calls (%lu returned an error: the mock's device answers with zeros), %u device(s), seed %llu\n", rounds, n_calls, n_fail,
         n_dev, (unsigned long long)seed);
  for (const auto& c : by_call) printf("  %-28s %8lu calls, %8lu returned an error\n", c.first.c_str(), c.second.first, c.second.second);
  fflush(stdout);
  return 0;
}
