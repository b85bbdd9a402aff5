// mic_engine.hip — C-ABI implementation (include/mi_clark.h): engine lifetime, DB load, batch API,
// device-resident entry points.  Host code over the HIP runtime; kernels live in mic_kernels.hip /
// mic_build.hip.  Mirrors the public surface of CuClarkDB<HKMERr> (CuClarkDB.cuh:98-150).
#include "mi_clark.h"
#include "mic_internal.h"

#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <sched.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <unistd.h>
#include <utility>
#include <vector>

int mic_bind_thread_near_device(int device, int on);   // below: host memory near the device
bool mic_peer_enable(int from, int to);                 // below: several devices in one process
void mic_peer_enable_engines(mic_engine* const* engines, size_t n);

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace
thread_local uint64_t mic_build_reserved_hbm = 0;

// ---- streams and events: pooled for the life of the process, never destroyed -------------------------------------------------------
// The one native fault this library's soaks ever produced (three sightings in ~100 hours of fuzzing since round 3, "heap corruption,
// cause unknown") was caught in round 6 under the guard-page allocator with its native stack: the HIP runtime's completion-signal
// handler (a thread of libhsa-runtime64 calling into libamdhip64: `lock sub [queue + 0x98]`, `xchg [queue + 0x378]`) running for the
// last command of a stream AFTER hipDeviceSynchronize had returned on the thread that then destroyed that stream - the handler
// decrements a counter in the freed queue object: a write into whatever the heap put there next.  Twice in two soaks of ~750
// engine lifetimes each, both inside mic_destroy.  Nothing the caller can order (the wait had returned) - so the objects whose
// destruction races are not destroyed: an engine's streams and events go back to a pool (drained first), the next engine on that
// device takes them.  The pool is bounded by the most that were ever live at once.  Creating a stream costs ~2 ms (a hardware queue):
// a process that builds engines repeatedly also saves that.
namespace {
struct StreamPool {
  std::mutex mu;
  std::vector<std::pair<int, hipStream_t>> free_streams, all_streams;          // (device, stream)
  std::vector<std::pair<int, hipEvent_t>> free_events, all_events;             // (device * 2 + timing, event)
};
StreamPool& pool() { static StreamPool* p = new StreamPool; return *p; }       // (never destructed: handles outlive every static)
}  // namespace

hipError_t mic_stream_get(hipStream_t* s) {
  int dev = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he != hipSuccess) return he;
  StreamPool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (size_t i = 0; i < P.free_streams.size(); ++i)
      if (P.free_streams[i].first == dev) { *s = P.free_streams[i].second; P.free_streams.erase(P.free_streams.begin() + (ptrdiff_t)i); return hipSuccess; }
  }
  he = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  if (he == hipSuccess) { std::lock_guard<std::mutex> lk(P.mu); P.all_streams.emplace_back(dev, *s); }
  return he;
}

void mic_stream_put(hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  StreamPool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  for (const auto& a : P.all_streams) if (a.second == s) { P.free_streams.push_back(a); return; }
}

hipError_t mic_event_get(hipEvent_t* ev, bool timing) {
  int dev = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he != hipSuccess) return he;
  const int key = dev * 2 + (timing ? 1 : 0);
  StreamPool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (size_t i = 0; i < P.free_events.size(); ++i)
      if (P.free_events[i].first == key) { *ev = P.free_events[i].second; P.free_events.erase(P.free_events.begin() + (ptrdiff_t)i); return hipSuccess; }
  }
  he = timing ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (he == hipSuccess) { std::lock_guard<std::mutex> lk(P.mu); P.all_events.emplace_back(key, *ev); }
  return he;
}

extern "C" int mic_debug_stream_pool(uint32_t out[4]) {
  if (!out) return MIC_E_INVALID;
  StreamPool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  out[0] = (uint32_t)P.all_streams.size(); out[1] = (uint32_t)P.free_streams.size();
  out[2] = (uint32_t)P.all_events.size(); out[3] = (uint32_t)P.free_events.size();
  return MIC_OK;
}

void mic_event_put(hipEvent_t ev) {
  if (!ev) return;
  (void)hipEventSynchronize(ev);             // (recorded work done - or never recorded: returns at once)
  StreamPool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  for (const auto& a : P.all_events) if (a.second == ev) { P.free_events.push_back(a); return; }
}
namespace {
std::mutex g_report_mu;
std::string g_report;     // stage times of the last table build that FINISHED in this process, one "name: seconds" per line
thread_local std::string t_report;   // ... of the build running on this thread

#define HIPTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

struct Batch {
  uint32_t* h_rp = nullptr; uint16_t* h_cont = nullptr;
  uint32_t* d_rp = nullptr; uint16_t* d_cont = nullptr;
  uint32_t* d_results = nullptr; uint32_t* d_rows = nullptr;
  uint32_t* d_flagged = nullptr; uint32_t* h_flagged = nullptr;
  uint32_t* d_peer = nullptr; uint32_t* d_acc = nullptr;   // table-sharded merge: another engine's rows, the running sum
  uint32_t* d_crowd = nullptr;   // work area of the crowded runs' follow-up (mic_internal.h: mic_crowd_dims)
  size_t first_read = 0, n_reads = 0, n_cont = 0, max_reads = 0, max_cont = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr, ev_up = nullptr, ev_k = nullptr;   // batch finished; its upload finished; its kernels finished
  bool scheduled = false, resolved = true, extended = false;
};

}  // namespace

struct mic_engine {
  mic_config cfg;
  int device = 0, n_cu = 256;
  hipStream_t stream = nullptr;
  // ALL uploads of the batch / ingest paths go through one stream and all downloads through another (events tie a
  // batch's kernels in between): with the copies of every batch on that batch's own stream the two directions share
  // copy engines and the link moves 41 instead of 54 GB/s in this traffic shape (tools/host_link_probe.hip)
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  // table
  bool db_loaded = false;
  uint4* slots = nullptr;
  uint4* side = nullptr;       // super-k-mer tables: the k-mers of crowded minimizers (or null)
  uint8_t* d_sizes = nullptr;  // the shard's on-disk bucket sizes (statistics)
  MicTable table;
  int slot_class = 32;
  mic_db_info info;
  // batches
  std::vector<Batch> batches;
  void* h_block = nullptr; void* d_block = nullptr;   // one pinned and one device allocation carved into all buffers
  uint32_t* h_results = nullptr; uint32_t* h_rows = nullptr;
  size_t num_reads_total = 0;
  std::mutex submit_mu;
  // device-API state
  uint32_t* d_flagged = nullptr; uint32_t flagged_cap = 0;
  uint32_t* d_crowd = nullptr; size_t crowd_reads = 0;     // work area of the crowded runs' follow-up, sized for crowd_reads reads
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  size_t last_n_reads = 0;
  void* ingest = nullptr;      // device-side ingest state (mic_ingest.hip)
  uint64_t reserve_hbm = 0;         // mic_db_reserve_hbm: device memory the caller is allocating while the table builds
  uint32_t part = 0, n_parts = 0;   // mic_db_set_part: this engine answers for part `part` of `n_parts` of the database
};

namespace {

const uint32_t kFlaggedCap = 1u << 16;

int set_device(const mic_engine* e) {
  HIPTRY(hipSetDevice(e->device));
  return MIC_OK;
}

void fill_table(mic_engine* e, const MicBuildOut& b, uint64_t htsize, uint64_t s0, uint64_t s1, int key_bytes,
                uint32_t sampling, int layout, int m) {
  e->slots = b.slots;
  const bool parted = (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2) && (b.part_lo != 0 || b.part_hi != b.n_main);
  // a slot-range part is addressed by GLOBAL slot indices: the table pointer is the allocation minus the slots in front of it
  e->table.slots = parted ? b.slots - (ptrdiff_t)b.part_lo * 8 : b.slots;
  e->table.slot_lo = parted ? (uint32_t)b.part_lo : 0; e->table.slot_cnt = parted ? (uint32_t)(b.part_hi - b.part_lo) : 0;
  e->table.parted = parted ? 1 : 0;
  e->side = b.side;
  e->table.side = b.side; e->table.side_mask = b.side ? (uint32_t)(b.side_cells - 1) : 0;
  e->table.n_main = b.n_main;
  e->table.shard_start = s0;
  e->table.shard_end = s1;
  e->table.div = mic_make_div(htsize);
  e->table.k = e->cfg.k;
  e->table.layout = layout == MIC_LAYOUT_MINIMIZER ? 1 : (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2) ? 2 : 0;
  e->table.fwd = layout == MIC_LAYOUT_SUPER2 ? 1 : 0;
  e->table.m = m;
  e->table.sharded = (s0 != 0 || s1 != htsize) ? 1 : 0;
  e->table.sizes = e->d_sizes;
  mic_db_info& i = e->info;
  i.htsize = htsize; i.shard_start = s0; i.shard_end = s1;
  i.n_elems = b.n_elems; i.n_elems_file = b.n_elems_file;
  const uint64_t n_res = parted ? b.part_hi - b.part_lo : b.n_main;    // main slots resident here
  i.n_slots = n_res + b.n_overflow; i.n_overflow = b.n_overflow;
  i.part = e->n_parts > 1 ? e->part : 0; i.n_parts = e->n_parts > 1 ? e->n_parts : 0;
  i.part_slot_lo = parted ? b.part_lo : 0; i.part_slot_hi = parted ? b.part_hi : 0; i.n_slots_whole = b.n_main;
  i.side_kmers = b.side ? b.side_kmers : 0; i.side_bytes = b.side ? b.side_cells * 16 : 0;
  i.hbm_bytes = (b.side ? b.side_cells * 16 : 0) + (b.alloc_slots ? b.alloc_slots : n_res + b.n_overflow + 1) * (uint64_t)(layout != MIC_LAYOUT_DIRECT ? MIC_MSLOT_BYTES : MIC_SLOT_BYTES);
  i.key_bytes = key_bytes; i.slot_class = layout != MIC_LAYOUT_DIRECT ? 128 : e->slot_class; i.max_bucket = b.max_bucket;
  i.sampling = sampling; i.layout = layout; i.minimizer_len = layout != MIC_LAYOUT_DIRECT ? m : 0;
  i.max_chain = layout != MIC_LAYOUT_DIRECT ? b.max_chain : 0; i.reserved = (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2) ? b.walk_ppm : 0;
  i.n_entries = b.n_entries ? b.n_entries : b.n_elems;
  e->db_loaded = true;
  (void)mic_kernels_warm(e->stream);          // (the query kernels' code on the device before the first batch asks for it)
}

int check_shard(uint64_t htsize, uint64_t& s0, uint64_t& s1) {
  if (htsize < 2 || htsize > 0xFFFFFFF0ULL) return fail(MIC_E_INVALID, "unsupported table size %llu", (unsigned long long)htsize);
  if (s1 == 0) s1 = htsize;
  if (s0 >= s1 || s1 > htsize) return fail(MIC_E_INVALID, "bad shard [%llu,%llu) of %llu", (unsigned long long)s0,
                                           (unsigned long long)s1, (unsigned long long)htsize);
  return MIC_OK;
}

// stream a byte range of a file into device memory through two pinned staging buffers; every chunk is read by several
// threads (pread on disjoint slices): one fread stream moves ~6 GB/s out of the page cache, the link takes ~50.
// Several destinations (mic_db_load_files_multi: one per device): every chunk is read ONCE and uploaded to each of them.
// Destinations may want different byte ranges of the file (mic_db_load_files_multi: a device whose engines answer for a bucket range
// holds that range of the images only): [lo, hi) in file offsets, dst = where byte lo goes.  Every chunk is read once and goes to the
// destinations whose range it meets.
struct UploadDst { int device; void* dst; hipStream_t stream; uint64_t lo, hi; };
// The two pinned staging buffers of a load: allocated once for its three files (pinning 2 x 256 MB costs ~0.1 s a time)
struct UploadStage {
  static constexpr size_t CH = 256u << 20;
  void* buf[2] = {nullptr, nullptr};
  bool get(int near_device) {
    if (buf[0] && buf[1]) return true;
    mic_bind_thread_near_device(near_device, 1);
    // (HIP has one context per process: pinned host memory is reachable from every device, whatever device was current)
    bool ok = true;
    for (void*& b : buf) if (!b && ok) { ok = hipHostMalloc(&b, CH, hipHostMallocDefault) == hipSuccess; if (!ok) b = nullptr; }
    // both or none: a stage shared by the three files of a load must never hand out one buffer and a null
    if (!ok) for (void*& b : buf) if (b) { hipHostFree(b); b = nullptr; }
    mic_bind_thread_near_device(near_device, 0);
    return ok;
  }
  ~UploadStage() { for (void* b : buf) if (b) hipHostFree(b); }
};
int upload_file_range_multi(FILE* f, const std::vector<UploadDst>& dsts, const char* what, UploadStage* shared = nullptr) {
  const size_t CH = UploadStage::CH;
  const int fd = fileno(f);
  static const int n_readers = [] { const char* e = getenv("MIC_LOAD_THREADS"); int v = e ? atoi(e) : 0; return v > 0 ? (v > 64 ? 64 : v) : 8; }();
  UploadStage own;
  UploadStage* st = shared ? shared : &own;
  void* stage[2] = {nullptr, nullptr};
  const size_t nd = dsts.size();
  std::vector<hipEvent_t> ev(2 * nd, nullptr);
  int rc = MIC_OK;
  if (nd == 0) return MIC_OK;
  uint64_t off = ~0ull, end = 0;
  for (const UploadDst& d : dsts) if (d.hi > d.lo) { off = d.lo < off ? d.lo : off; end = d.hi > end ? d.hi : end; }
  if (end <= off) return MIC_OK;
  const uint64_t bytes = end - off;
  int dev_now = 0;
  hipGetDevice(&dev_now);
  if (!st->get(dsts[0].device)) rc = fail(MIC_E_NOMEM, "pinned staging alloc failed");
  stage[0] = st->buf[0]; stage[1] = st->buf[1];
  mic_bind_thread_near_device(dsts[0].device, 1);
  for (size_t i = 0; i < 2 * nd && rc == MIC_OK; ++i) {
    if (hipSetDevice(dsts[i % nd].device) != hipSuccess || mic_event_get(&ev[i], false) != hipSuccess)
      rc = fail(MIC_E_HIP, "event create failed");
  }
  uint64_t done = 0; int cur = 0;
  std::vector<char> used(2 * nd, 0);
  while (rc == MIC_OK && done < bytes) {
    size_t n = (size_t)((bytes - done) < CH ? (bytes - done) : CH);
    const uint64_t c0 = off + done, c1 = c0 + n;
    bool wanted = false;
    for (const UploadDst& d : dsts) if (d.lo < c1 && d.hi > c0) wanted = true;
    if (!wanted) { done += n; continue; }                // (a gap between the destinations' ranges)
    for (size_t d = 0; d < nd; ++d)
      if (used[(size_t)cur * nd + d]) {
        if (hipEventSynchronize(ev[(size_t)cur * nd + d]) != hipSuccess) { rc = fail(MIC_E_HIP, "event sync failed"); break; }
        used[(size_t)cur * nd + d] = 0;
      }
    if (rc != MIC_OK) break;
    {
      std::atomic<bool> short_read(false);
      const size_t per = ((n + (size_t)n_readers - 1) / (size_t)n_readers + 4095) & ~(size_t)4095;
      std::vector<std::thread> th;
      for (int t = 0; t < n_readers; ++t) {
        const size_t lo = (size_t)t * per;
        if (lo >= n) break;
        const size_t len = n - lo < per ? n - lo : per;
        th.emplace_back([&, lo, len] {
          size_t got = 0;
          while (got < len) {
            const ssize_t r = pread(fd, (char*)stage[cur] + lo + got, len - got, (off_t)(c0 + lo + got));
            if (r <= 0) { short_read = true; return; }
            got += (size_t)r;
          }
        });
      }
      for (auto& t : th) t.join();
      if (short_read) { rc = fail(MIC_E_IO, "%s is shorter than the bucket sizes imply", what); break; }
    }
    for (size_t d = 0; d < nd && rc == MIC_OK; ++d) {
      const uint64_t a = dsts[d].lo > c0 ? dsts[d].lo : c0, b = dsts[d].hi < c1 ? dsts[d].hi : c1;
      if (a >= b) continue;
      if (hipSetDevice(dsts[d].device) != hipSuccess ||
          hipMemcpyAsync((char*)dsts[d].dst + (a - dsts[d].lo), (char*)stage[cur] + (a - c0), (size_t)(b - a), hipMemcpyHostToDevice, dsts[d].stream) != hipSuccess ||
          hipEventRecord(ev[(size_t)cur * nd + d], dsts[d].stream) != hipSuccess) rc = fail(MIC_E_HIP, "H2D copy failed");
      used[(size_t)cur * nd + d] = 1;
    }
    cur ^= 1; done += n;
  }
  for (size_t d = 0; d < nd; ++d) { hipSetDevice(dsts[d].device); hipStreamSynchronize(dsts[d].stream); }
  for (hipEvent_t e : ev) if (e) mic_event_put(e);
  mic_bind_thread_near_device(dsts[0].device, 0);
  hipSetDevice(dev_now);
  return rc;
}

int upload_file_range(FILE* f, uint64_t off, uint64_t bytes, void* dst, hipStream_t s, const char* what, UploadStage* shared = nullptr) {
  int dev_now = 0;
  hipGetDevice(&dev_now);
  return upload_file_range_multi(f, {UploadDst{dev_now, dst, s, off, off + bytes}}, what, shared);
}

// The .sz image (one byte per bucket, 1.6 GB for the reference's table size) read by several threads at once into one block, and
// sums over bucket ranges by several threads: one fread stream + one summing loop were 0.68 s of the command line's start
bool read_file_parallel(FILE* f, uint8_t* dst, uint64_t bytes) {
  const int fd = fileno(f);
  const int n_thr = bytes < ((uint64_t)8 << 20) ? 1 : 8;
  std::atomic<bool> bad(false);
  std::vector<std::thread> th;
  const uint64_t per = ((bytes + n_thr - 1) / n_thr + 4095) & ~(uint64_t)4095;
  for (int t = 0; t < n_thr; ++t) {
    const uint64_t lo = (uint64_t)t * per;
    if (lo >= bytes) break;
    const uint64_t len = bytes - lo < per ? bytes - lo : per;
    th.emplace_back([&, lo, len] {
      uint64_t got = 0;
      while (got < len) {
        const ssize_t r = pread(fd, dst + lo + got, (size_t)(len - got), (off_t)(lo + got));
        if (r <= 0) { bad = true; return; }
        got += (uint64_t)r;
      }
    });
  }
  for (auto& t : th) t.join();
  return !bad;
}
// elements and non-empty buckets of sizes[lo, hi)
void sum_sizes_parallel(const uint8_t* sizes, uint64_t lo, uint64_t hi, uint64_t* elems, uint64_t* nonzero) {
  const uint64_t n = hi - lo;
  const int n_thr = n < ((uint64_t)8 << 20) ? 1 : 8;
  std::vector<uint64_t> e((size_t)n_thr, 0), z((size_t)n_thr, 0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_thr; ++t)
    th.emplace_back([&, t] {
      const uint64_t a = lo + n * (uint64_t)t / n_thr, b = lo + n * (uint64_t)(t + 1) / n_thr;
      uint64_t se = 0, sz = 0;
      for (uint64_t i = a; i < b; ++i) { se += sizes[i]; sz += sizes[i] > 0; }
      e[(size_t)t] = se; z[(size_t)t] = sz;
    });
  for (auto& t : th) t.join();
  *elems = 0; *nonzero = 0;
  for (int t = 0; t < n_thr; ++t) { *elems += e[(size_t)t]; *nonzero += z[(size_t)t]; }
}

// layout: explicit request, else the environment (MIC_LAYOUT=direct|minimizer|super|super2), else by k
void decide_layout(const mic_engine* e, int& layout, bool& by_default, int& m, int& m0) {
  layout = (int)e->cfg.layout;
  by_default = false;   // nobody asked for this layout: a table that does not fit may fall back to the other one
  if (layout == MIC_LAYOUT_AUTO) {
    const char* env = getenv("MIC_LAYOUT");
    if (env && !strcmp(env, "direct")) layout = MIC_LAYOUT_DIRECT;
    else if (env && !strcmp(env, "minimizer")) layout = MIC_LAYOUT_MINIMIZER;
    else if (env && !strcmp(env, "super")) layout = MIC_LAYOUT_SUPER;
    else if (env && !strcmp(env, "super2")) layout = MIC_LAYOUT_SUPER2;
    else { layout = e->cfg.k >= 24 ? MIC_LAYOUT_SUPER : MIC_LAYOUT_DIRECT; by_default = true; }  // measured: DESIGN.md §3
  }
  m = 20;   // measured best for k = 31: minimizers long enough to be nearly unique in the table
  if (const char* env = getenv("MIC_MINIMIZER_LEN")) m = atoi(env);
  if (m > e->cfg.k - 4) m = e->cfg.k - 4;   // window w = k-m+1 >= 5
  if (m > 31) m = 31;
  if (layout == MIC_LAYOUT_MINIMIZER && (m < 8 || e->cfg.k - m + 1 > 64)) layout = MIC_LAYOUT_DIRECT;
  m0 = m;
  if (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2) {          // the super-k-mer entries hold windows of at most 16 m-mers
    if (m < e->cfg.k - 15) m = e->cfg.k - 15;
    if (m < 8 || m > 31 || e->cfg.k - m + 1 < 2) layout = MIC_LAYOUT_DIRECT;
  }
  // parts of a table-sharded run: every engine must arrive at the same layout, so nothing depends on what fits where
  if (e->n_parts > 1) by_default = false;
}

// a default layout that was given up for another one: said in the build report ("fallback: ..." lines; the CLI prints them)
void note_fallback(const char* from, const char* to, const char* why) {
  char line[400];
  snprintf(line, sizeof(line), "fallback: %s -> %s (%s)", from, to, why);
  for (char* c = line; *c; ++c) if (*c == ':' && c > line + 8) *c = ';';      // one "name: seconds" pair per line
  mic_build_report_add(line, 0.0);
}

// mic_db_set_part: the super-k-mer layouts split their RESIDENT table by slot range inside the build (the images stay whole);
// the other layouts answer for the part's share of the on-disk buckets, the reference's split (CuClarkDB.cu:566-574)
int apply_part(const mic_engine* e, uint64_t htsize, uint64_t& s0, uint64_t& s1) {
  if (e->n_parts <= 1) return MIC_OK;
  if (s0 != 0 || s1 != htsize) return fail(MIC_E_INVALID, "mic_db_set_part and an explicit bucket range exclude each other");
  int layout, m, m0; bool by_default;
  decide_layout(e, layout, by_default, m, m0);
  if (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2) return MIC_OK;
  if (htsize < e->n_parts) return fail(MIC_E_INVALID, "more parts (%u) than buckets (%llu)", e->n_parts, (unsigned long long)htsize);
  s0 = (uint64_t)((unsigned __int128)htsize * e->part / e->n_parts);
  s1 = (uint64_t)((unsigned __int128)htsize * (e->part + 1) / e->n_parts);
  return MIC_OK;
}

int build_from_device(mic_engine* e, const uint8_t* d_sizes_shard, uint64_t htsize, uint64_t s0, uint64_t s1,
                      const void* d_keys_shard, int key_bytes, const uint16_t* d_labels_shard, uint32_t sampling,
                      uint64_t rank_base) {
  e->slot_class = key_bytes == 8 ? 64 : 32;
  MicBuildOut b;
  memset(&b, 0, sizeof(b));
  char err[256] = "";
  t_report.clear();                              // the stages of THIS build (mic_build_report_add appends to the calling thread's copy)
  mic_build_reserved_hbm = e->reserve_hbm;       // the builders size their staging areas from the free HBM minus this
  struct timespec t_b0; clock_gettime(CLOCK_MONOTONIC, &t_b0);
  int layout, m, m0; bool by_default;
  decide_layout(e, layout, by_default, m, m0);
  // slot-range part (mic_db_set_part) of a super-k-mer table; the other layouts were given their bucket range by the caller
  const bool slot_part = e->n_parts > 1 && (layout == MIC_LAYOUT_SUPER || layout == MIC_LAYOUT_SUPER2);
  const uint32_t part = slot_part ? e->part : 0, n_parts = slot_part ? e->n_parts : 0;
  if (e->d_sizes) { hipFree(e->d_sizes); e->d_sizes = nullptr; }
  if (hipMalloc(&e->d_sizes, s1 - s0) == hipSuccess)
    hipMemcpyAsync(e->d_sizes, d_sizes_shard, s1 - s0, hipMemcpyDeviceToDevice, e->stream);
  int rc = 0;
  // By default: super-k-mer slots; if their build does not fit in the free HBM, the minimizer-keyed slots (no staging
  // area), then the direct layout (64 B per bucket).  An explicitly requested layout fails with the sizes in the message.
  if (layout == MIC_LAYOUT_SUPER2) {
    // both strands stored: the fastest kernel, twice the entries; falls back to the one-strand table when it does not fit
    rc = mic_build_stable(d_sizes_shard, s1 - s0, s0, htsize, d_keys_shard, key_bytes, d_labels_shard, sampling, rank_base,
                          e->cfg.k, m, e->stream, &b, err, sizeof(err), by_default ? 1 : 0, 1, part, n_parts);
    if (rc == -3 && !slot_part && (by_default || getenv("MIC_SUPER2_MAY_FALL_BACK"))) { layout = MIC_LAYOUT_SUPER; memset(&b, 0, sizeof(b)); note_fallback("two-strand super-k-mer table", "one-strand super-k-mer table", err); }
    else if (rc == -5 && by_default) { layout = MIC_LAYOUT_MINIMIZER; memset(&b, 0, sizeof(b)); m = m0; note_fallback("two-strand super-k-mer table", "minimizer table", err); }
  }
  if (layout == MIC_LAYOUT_SUPER) {
    rc = mic_build_stable(d_sizes_shard, s1 - s0, s0, htsize, d_keys_shard, key_bytes, d_labels_shard, sampling, rank_base,
                          e->cfg.k, m, e->stream, &b, err, sizeof(err), by_default ? 1 : 0, 0, part, n_parts);
    // -3: does not fit; -5: crowded minimizers (tandem repeats): the minimizer layout's trees answer those faster
    if ((rc == -3 || rc == -5) && by_default) { layout = MIC_LAYOUT_MINIMIZER; memset(&b, 0, sizeof(b)); m = m0; note_fallback("super-k-mer table", "minimizer table", err); }
  }
  if (layout == MIC_LAYOUT_MINIMIZER) {
    rc = mic_build_mtable(d_sizes_shard, s1 - s0, s0, htsize, d_keys_shard, key_bytes, d_labels_shard, sampling, rank_base,
                          e->cfg.k, m, e->stream, &b, err, sizeof(err));
    // -3: even the densest minimizer table exceeds the free HBM; the direct layout is ~40 % smaller (64 B per bucket)
    if (rc == -3 && by_default) { layout = MIC_LAYOUT_DIRECT; memset(&b, 0, sizeof(b)); note_fallback("minimizer table", "direct table", err); }
  }
  if (layout == MIC_LAYOUT_DIRECT)
    rc = mic_build_table(d_sizes_shard, s1 - s0, d_keys_shard, key_bytes, d_labels_shard, sampling, rank_base,
                         e->slot_class, e->stream, &b, err, sizeof(err));
  if (rc != 0) return fail(rc, "table build: %s", err);
  fill_table(e, b, htsize, s0, s1, key_bytes, sampling, layout, m);
  { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    mic_build_report_add("table build, total", (t.tv_sec - t_b0.tv_sec) + (t.tv_nsec - t_b0.tv_nsec) / 1e9); }
  // published in one piece: builds on other threads (mic_db_load_files_multi: one per device) neither wipe nor interleave it
  { std::lock_guard<std::mutex> lk(g_report_mu); g_report = t_report; }
  return MIC_OK;
}

void free_batches(mic_engine* e) {
  for (Batch& b : e->batches) {
    if (b.done) mic_event_put(b.done);
    if (b.ev_up) mic_event_put(b.ev_up);
    if (b.ev_k) mic_event_put(b.ev_k);
    if (b.stream) mic_stream_put(b.stream);
    if (b.d_peer) hipFree(b.d_peer);
    if (b.d_acc) hipFree(b.d_acc);
  }
  e->batches.clear();
  if (e->h_block) { hipHostFree(e->h_block); e->h_block = nullptr; }
  if (e->d_block) { hipFree(e->d_block); e->d_block = nullptr; }
  e->h_results = nullptr; e->h_rows = nullptr;
  e->num_reads_total = 0;
}

// dense path for a list of read ids held on the device (nullptr = reads 0..n_ids-1); patches results (and rows)
int run_dense(mic_engine* e, const uint32_t* d_rp, const uint16_t* d_cont, const uint32_t* d_ids, size_t n_ids,
              uint32_t* d_results, uint32_t* d_rows, hipStream_t s) {
  if (!n_ids) return MIC_OK;
  const uint32_t T = e->cfg.num_targets ? e->cfg.num_targets : 1;
  uint32_t* d_iota = nullptr;
  if (!d_ids) {
    std::vector<uint32_t> iota(n_ids);
    for (size_t i = 0; i < n_ids; ++i) iota[i] = (uint32_t)i;
    HIPTRY(hipMalloc(&d_iota, n_ids * 4));
    hipError_t he = hipMemcpy(d_iota, iota.data(), n_ids * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) { hipFree(d_iota); return fail(MIC_E_HIP, "dense path: %s", hipGetErrorString(he)); }
    d_ids = d_iota;
  }
  // process in slabs of at most ~1 GiB of counters
  size_t slab = (size_t)((1ull << 30) / ((uint64_t)T * 4));
  if (slab == 0) slab = 1;
  if (slab > n_ids) slab = n_ids;
  uint32_t* d_counts = nullptr;
  hipError_t me = hipMalloc(&d_counts, slab * (size_t)T * 4);
  if (me != hipSuccess) { if (d_iota) hipFree(d_iota); return fail(MIC_E_NOMEM, "dense path: %s", hipGetErrorString(me)); }
  int rc = MIC_OK;
  for (size_t off = 0; off < n_ids && rc == MIC_OK; off += slab) {
    size_t n = n_ids - off < slab ? n_ids - off : slab;
    hipError_t he = mic_launch_dense_count(e->table, e->slot_class, d_rp, d_cont, d_ids + off, n, T, d_counts, s);
    if (he == hipSuccess) he = mic_launch_dense_finish(d_counts, d_ids + off, n, T, d_results, d_rows, e->cfg.row_words, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) rc = fail(MIC_E_HIP, "dense path: %s", hipGetErrorString(he));
  }
  hipFree(d_counts);
  if (d_iota) hipFree(d_iota);
  return rc;
}

}  // namespace

// ---- host memory near the device ----------------------------------------------------------------------------------------
// On a two-socket host the device hangs off one socket; pinned buffers on the other socket are reached through the
// inter-socket link and the host link then runs at ~39 instead of ~57 GB/s (measured).  Pinned
// allocations are therefore made - and the CLI's loader threads run - on the CPUs of the device's NUMA node.
namespace {
int device_numa_node(int device) {
  static std::mutex mu;
  static int cache[64]; static bool have[64];
  std::lock_guard<std::mutex> lk(mu);
  if (device >= 0 && device < 64 && have[device]) return cache[device];
  int node = -1;
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) == hipSuccess) {
    for (char* c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  }
  if (device >= 0 && device < 64) { cache[device] = node; have[device] = true; }
  return node;
}

bool node_cpu_set(int node, cpu_set_t* set) {
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  CPU_ZERO(set);
  int a, b; char c; bool any = false;
  while (fscanf(f, "%d", &a) == 1) {
    b = a;
    if (fscanf(f, "%c", &c) == 1 && c == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &c) != 1) c = 0; }
    for (int i = a; i <= b && i < CPU_SETSIZE; ++i) { CPU_SET(i, set); any = true; }
    if (c != ',') break;
  }
  fclose(f);
  return any;
}

thread_local cpu_set_t t_saved_mask;
thread_local int t_bound_depth = 0;
}  // namespace

// binds the calling thread to the CPUs of the device's NUMA node (on != 0) or gives it its previous mask back (on == 0);
// nests; a no-op (returns 0) when the node is unknown or MIC_NO_NUMA is set.  Returns 1 if the mask was changed.
int mic_bind_thread_near_device(int device, int on) {
  static const bool off = getenv("MIC_NO_NUMA") != nullptr;
  if (off) return 0;
  if (on) {
    if (t_bound_depth++ > 0) return 0;
    const int node = device_numa_node(device);
    cpu_set_t want, cur, both;
    if (node < 0 || !node_cpu_set(node, &want) || sched_getaffinity(0, sizeof(cur), &cur) != 0) { --t_bound_depth; return 0; }
    CPU_AND(&both, &want, &cur);
    if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, &cur)) { --t_bound_depth; return 0; }
    t_saved_mask = cur;
    if (sched_setaffinity(0, sizeof(both), &both) != 0) { --t_bound_depth; return 0; }
    return 1;
  }
  if (t_bound_depth > 0 && --t_bound_depth == 0) sched_setaffinity(0, sizeof(t_saved_mask), &t_saved_mask);
  return 0;
}

// ---- what mic_ingest.hip needs from the engine ------------------------------------------------------------------------
int mic_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int mic_engine_table(mic_engine* e, MicTable* t, int* slot_class, int* n_cu, int* device, int* k, uint32_t* n_targets) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  *t = e->table;
  if (!e->db_loaded) t->slots = nullptr;
  *slot_class = e->slot_class; *n_cu = e->n_cu; *device = e->device; *k = e->cfg.k; *n_targets = e->cfg.num_targets;
  return MIC_OK;
}

void** mic_engine_ingest_slot(mic_engine* e) { return &e->ingest; }
void mic_engine_copy_streams(mic_engine* e, hipStream_t* up, hipStream_t* down) { *up = e->up_stream; *down = e->down_stream; }

void mic_build_report_add(const char* what, double seconds) {
  char line[256];
  snprintf(line, sizeof(line), "%s: %.4f\n", what, seconds);
  t_report += line;
}

// ---- several devices in one process -----------------------------------------------------------------------------------
namespace {
std::mutex g_peer_mu;
signed char g_peer[64][64];          // [from][to]: 0 unknown, 1 direct access enabled, -1 none (copies are staged by the runtime)
}
// enables device `from`'s access to device `to`'s memory (once; "already enabled" is success); true when direct access is on
bool mic_peer_enable(int from, int to) {
  if (from == to) return true;
  if (from < 0 || to < 0 || from >= 64 || to >= 64) return false;
  std::lock_guard<std::mutex> lk(g_peer_mu);
  if (g_peer[from][to]) return g_peer[from][to] > 0;
  int can = 0, cur = 0;
  hipGetDevice(&cur);
  bool ok = false;
  if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can && hipSetDevice(from) == hipSuccess) {
    const hipError_t he = hipDeviceEnablePeerAccess(to, 0);
    ok = he == hipSuccess || he == hipErrorPeerAccessAlreadyEnabled;
    (void)hipGetLastError();
  }
  hipSetDevice(cur);
  g_peer[from][to] = ok ? 1 : -1;
  return ok;
}
void mic_peer_enable_engines(mic_engine* const* engines, size_t n) {
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j)
      if (engines[i] && engines[j] && engines[i]->device != engines[j]->device) mic_peer_enable(engines[i]->device, engines[j]->device);
}

extern "C" {

const char* mic_last_error(void) { return g_err; }

int mic_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(MIC_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return MIC_OK;
}

int mic_create(const mic_config* cfg, mic_engine** out) {
  if (!cfg || !out) return fail(MIC_E_INVALID, "null argument");
  if (cfg->k < 2 || cfg->k > 32) return fail(MIC_E_INVALID, "The k-mer length should be in [2,32].");
  if (cfg->num_targets > 65535) return fail(MIC_E_INVALID, "too many targets (%u > 65535)", cfg->num_targets);
  if (cfg->layout > MIC_LAYOUT_SUPER2) return fail(MIC_E_INVALID, "unknown table layout %u", cfg->layout);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(MIC_E_NODEVICE, "no HIP device available: the MI355X engine cannot run (there is no CPU fallback)");
  int dev = cfg->device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= n) return fail(MIC_E_INVALID, "device %d out of range (%d devices)", dev, n);
  hipDeviceProp_t prop;
  HIPTRY(hipSetDevice(dev));
  HIPTRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MIC_E_NODEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", dev, prop.gcnArchName);
  mic_engine* e = new (std::nothrow) mic_engine();
  if (!e) return fail(MIC_E_NOMEM, "out of host memory");
  e->cfg = *cfg;
  if (e->cfg.num_batches == 0) e->cfg.num_batches = 1;
  if (e->cfg.row_words == 0) e->cfg.row_words = 16;
  if (e->cfg.row_words < 2) e->cfg.row_words = 2;
  e->device = dev;
  e->n_cu = prop.multiProcessorCount;
  memset(&e->info, 0, sizeof(e->info));
  memset(&e->table, 0, sizeof(e->table));
  hipError_t he = mic_stream_get(&e->stream);
  if (he == hipSuccess) he = mic_stream_get(&e->up_stream);
  if (he == hipSuccess) he = mic_stream_get(&e->down_stream);
  if (he == hipSuccess) he = mic_event_get(&e->ev0, true);
  if (he == hipSuccess) he = mic_event_get(&e->ev1, true);
  if (he == hipSuccess) he = hipMalloc(&e->d_flagged, (size_t)(kFlaggedCap + 1) * 4);
  if (he != hipSuccess) { mic_destroy(e); return fail(MIC_E_HIP, "engine setup: %s", hipGetErrorString(he)); }
  e->flagged_cap = kFlaggedCap;
  *out = e;
  return MIC_OK;
}

int mic_destroy(mic_engine* e) {
  if (!e) return MIC_OK;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  mic_ingest_free(e);
  mic_gz_release(e);
  free_batches(e);
  if (e->slots) hipFree(e->slots);
  if (e->side) hipFree(e->side);
  if (e->d_sizes) hipFree(e->d_sizes);
  if (e->d_flagged) hipFree(e->d_flagged);
  if (e->d_crowd) hipFree(e->d_crowd);
  if (e->ev0) mic_event_put(e->ev0);
  if (e->ev1) mic_event_put(e->ev1);
  if (e->stream) mic_stream_put(e->stream);
  if (e->up_stream) mic_stream_put(e->up_stream);
  if (e->down_stream) mic_stream_put(e->down_stream);
  delete e;
  return MIC_OK;
}

int mic_db_unload(mic_engine* e) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  int rc = set_device(e);
  if (rc) return rc;
  hipDeviceSynchronize();
  if (e->slots) hipFree(e->slots);
  if (e->side) hipFree(e->side);
  if (e->d_sizes) hipFree(e->d_sizes);
  e->slots = nullptr; e->side = nullptr; e->d_sizes = nullptr; e->db_loaded = false;
  memset(&e->info, 0, sizeof(e->info));
  return MIC_OK;
}

int mic_db_get_info(const mic_engine* e, mic_db_info* info) {
  if (!e || !info) return fail(MIC_E_INVALID, "null argument");
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  *info = e->info;
  return MIC_OK;
}

const char* mic_db_last_build_report(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_report_mu);
  copy = g_report;
  return copy.c_str();
}

int mic_db_reserve_hbm(mic_engine* e, uint64_t bytes) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  e->reserve_hbm = bytes;
  return MIC_OK;
}

int mic_db_set_part(mic_engine* e, uint32_t part, uint32_t n_parts) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  if (n_parts > 1 && part >= n_parts) return fail(MIC_E_INVALID, "part %u of %u", part, n_parts);
  if (n_parts > 65536) return fail(MIC_E_INVALID, "too many parts");
  if (e->db_loaded) return fail(MIC_E_STATE, "set the part before the database is loaded");
  e->part = n_parts > 1 ? part : 0; e->n_parts = n_parts > 1 ? n_parts : 0;
  return MIC_OK;
}

int mic_db_load_device(mic_engine* e, const uint8_t* d_sizes, uint64_t htsize, const void* d_keys, int key_bytes,
                       const uint16_t* d_labels, uint32_t sampling, uint64_t s0, uint64_t s1) {
  if (!e || !d_sizes || !d_keys || !d_labels) return fail(MIC_E_INVALID, "null argument");
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MIC_E_INVALID, "key_bytes must be 2, 4 or 8");
  int rc = set_device(e);
  if (rc) return rc;
  if ((rc = check_shard(htsize, s0, s1))) return rc;
  if ((rc = apply_part(e, htsize, s0, s1))) return rc;
  if (e->db_loaded) mic_db_unload(e);
  uint64_t base_elems = 0, base_rank = 0;
  if (s0 > 0 && mic_reduce_sizes(d_sizes, s0, &base_elems, &base_rank, e->stream) != 0)
    return fail(MIC_E_HIP, "size reduction failed");
  return build_from_device(e, d_sizes + s0, htsize, s0, s1, (const char*)d_keys + base_elems * key_bytes, key_bytes,
                           d_labels + base_elems, sampling, base_rank);
}

int mic_db_load_host(mic_engine* e, const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                     const uint16_t* labels, uint32_t sampling, uint64_t s0, uint64_t s1) {
  if (!e || !sizes || !keys || !labels) return fail(MIC_E_INVALID, "null argument");
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MIC_E_INVALID, "key_bytes must be 2, 4 or 8");
  int rc = set_device(e);
  if (rc) return rc;
  if ((rc = check_shard(htsize, s0, s1))) return rc;
  if ((rc = apply_part(e, htsize, s0, s1))) return rc;
  if (e->db_loaded) mic_db_unload(e);
  uint64_t base_elems = 0, base_rank = 0, n_el = 0;
  for (uint64_t i = 0; i < s0; ++i) { base_elems += sizes[i]; base_rank += sizes[i] > 0; }
  for (uint64_t i = s0; i < s1; ++i) n_el += sizes[i];
  uint8_t* d_sz = nullptr; void* d_ky = nullptr; uint16_t* d_lb = nullptr;
  hipError_t he = hipMalloc(&d_sz, s1 - s0);
  if (he == hipSuccess) he = hipMalloc(&d_ky, n_el * key_bytes + 16);
  if (he == hipSuccess) he = hipMalloc(&d_lb, n_el * 2 + 16);
  if (he == hipSuccess) he = hipMemcpy(d_sz, sizes + s0, s1 - s0, hipMemcpyHostToDevice);
  if (he == hipSuccess && n_el) he = hipMemcpy(d_ky, (const char*)keys + base_elems * key_bytes, n_el * key_bytes, hipMemcpyHostToDevice);
  if (he == hipSuccess && n_el) he = hipMemcpy(d_lb, labels + base_elems, n_el * 2, hipMemcpyHostToDevice);
  if (he != hipSuccess) rc = fail(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "upload of DB images: %s", hipGetErrorString(he));
  else rc = build_from_device(e, d_sz, htsize, s0, s1, d_ky, key_bytes, d_lb, sampling, base_rank);
  if (d_sz) hipFree(d_sz);
  if (d_ky) hipFree(d_ky);
  if (d_lb) hipFree(d_lb);
  return rc;
}

int mic_db_load_files(mic_engine* e, const char* prefix, int key_bytes, uint32_t sampling, uint64_t s0, uint64_t s1) {
  if (!e || !prefix) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  std::string p(prefix);
  FILE* fs = fopen((p + ".sz").c_str(), "rb");
  FILE* fk = fopen((p + ".ky").c_str(), "rb");
  FILE* fl = fopen((p + ".lb").c_str(), "rb");
  uint8_t* h_sz = nullptr; uint8_t* d_sz = nullptr; void* d_ky = nullptr; uint16_t* d_lb = nullptr;
  UploadStage stage;
  const bool timing = getenv("MIC_LOAD_TIMING") != nullptr;
  struct timespec t_prev; clock_gettime(CLOCK_MONOTONIC, &t_prev);
  auto lap = [&](const char* what) {
    if (!timing) return;
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[load] %s: %.3f s\n", what, (t.tv_sec - t_prev.tv_sec) + (t.tv_nsec - t_prev.tv_nsec) / 1e9);
    t_prev = t;
  };
  do {
    // reference: "Failed to open <file>" and read() returns false (CuClarkDB.cu:490-495)
    if (!fs) { rc = fail(MIC_E_IO, "Failed to open %s.sz", prefix); break; }
    if (!fk) { rc = fail(MIC_E_IO, "Failed to open %s.ky", prefix); break; }
    if (!fl) { rc = fail(MIC_E_IO, "Failed to open %s.lb", prefix); break; }
    fseeko(fs, 0, SEEK_END);
    uint64_t htsize = (uint64_t)ftello(fs);
    fseeko(fs, 0, SEEK_SET);
    if ((rc = check_shard(htsize, s0, s1))) break;
    if ((rc = apply_part(e, htsize, s0, s1))) break;
    if (key_bytes == 0) key_bytes = mic_key_bytes_rule(htsize, e->cfg.k);
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) { rc = fail(MIC_E_INVALID, "key_bytes must be 2, 4 or 8"); break; }
    uint64_t base_elems = 0, base_rank = 0, n_el = 0, nz = 0;
    if (e->db_loaded) mic_db_unload(e);
    if (s0 == 0) {
      // the usual case (the whole table, or a first shard): the bucket sizes go straight to the device through the pinned staging
      // buffers (several readers, as for .ky and .lb) and are summed THERE - reading the 1.6 GB into pageable host memory and
      // summing them with one thread each were 0.68 s of the command line's start (0.44 s with eight threads each)
      hipError_t he = hipMalloc(&d_sz, s1);
      if (he != hipSuccess) { rc = fail(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "DB image allocation: %s", hipGetErrorString(he)); break; }
      if ((rc = upload_file_range(fs, 0, s1, d_sz, e->stream, "the .sz file", &stage))) break;
      if (mic_reduce_sizes(d_sz, s1, &n_el, &nz, e->stream) != 0) { rc = fail(MIC_E_HIP, "size reduction failed"); break; }
      lap("read + upload .sz, bucket sizes summed on the device");
      he = hipMalloc(&d_ky, n_el * key_bytes + 16);
      if (he == hipSuccess) he = hipMalloc(&d_lb, n_el * 2 + 16);
      if (he != hipSuccess) { rc = fail(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "DB image allocation: %s", hipGetErrorString(he)); break; }
    } else {
      h_sz = (uint8_t*)malloc(htsize);
      if (!h_sz) { rc = fail(MIC_E_NOMEM, "out of host memory for bucket sizes"); break; }
      if (!read_file_parallel(fs, h_sz, htsize)) { rc = fail(MIC_E_IO, "short read on %s.sz", prefix); break; }
      lap("read .sz");
      sum_sizes_parallel(h_sz, 0, s0, &base_elems, &base_rank);
      sum_sizes_parallel(h_sz, s0, s1, &n_el, &nz);
      lap("sum bucket sizes");
      hipError_t he = hipMalloc(&d_sz, s1 - s0);
      if (he == hipSuccess) he = hipMalloc(&d_ky, n_el * key_bytes + 16);
      if (he == hipSuccess) he = hipMalloc(&d_lb, n_el * 2 + 16);
      if (he == hipSuccess) he = hipMemcpy(d_sz, h_sz + s0, s1 - s0, hipMemcpyHostToDevice);
      if (he != hipSuccess) { rc = fail(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "DB image allocation: %s", hipGetErrorString(he)); break; }
    }
    lap("allocate + upload .sz");
    if ((rc = upload_file_range(fk, base_elems * key_bytes, n_el * key_bytes, d_ky, e->stream, "the .ky file", &stage))) break;
    lap("upload .ky");
    if ((rc = upload_file_range(fl, base_elems * 2, n_el * 2, d_lb, e->stream, "the .lb file", &stage))) break;
    lap("upload .lb");
    rc = build_from_device(e, d_sz, htsize, s0, s1, d_ky, key_bytes, d_lb, sampling, base_rank);
    lap("table build");
  } while (0);
  if (fs) fclose(fs);
  if (fk) fclose(fk);
  if (fl) fclose(fl);
  free(h_sz);
  if (d_sz) hipFree(d_sz);
  if (d_ky) hipFree(d_ky);
  if (d_lb) hipFree(d_lb);
  return rc;
}

// ---- batch API ---------------------------------------------------------------------------------------
int mic_batches_alloc(mic_engine* e, size_t num_reads_total, size_t max_reads, size_t max_containers,
                      const uint32_t* index_batches, int extended, uint32_t** results, uint32_t** rows,
                      uint32_t** reads_pointer, uint16_t** containers) {
  if (!e || !index_batches || !results || !reads_pointer || !containers) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  free_batches(e);
  const size_t nb = e->cfg.num_batches;
  const uint32_t rw = e->cfg.row_words;
  e->num_reads_total = num_reads_total;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // sizes of the per-batch pieces
  const size_t sz_rp = up((max_reads + 2) * 4), sz_ct = up((max_containers + 64) * 2), sz_fl = up((size_t)(kFlaggedCap + 1) * 4);
  const size_t sz_res = up((max_reads + 1) * MIC_RESULT_WORDS * 4), sz_rows = extended ? up((max_reads + 1) * (size_t)rw * 4) : 0;
  const size_t sz_crowd = up(mic_crowd_dims(max_reads).words * 4);
  const size_t h_res = up((num_reads_total + 1) * MIC_RESULT_WORDS * 4), h_rows = extended ? up((num_reads_total + 1) * (size_t)rw * 4) : 0;
  const size_t h_total = h_res + h_rows + nb * (sz_rp + sz_ct + sz_fl);
  const size_t d_total = nb * (sz_rp + sz_ct + sz_res + sz_rows + sz_fl + sz_crowd);
  {
    mic_bind_thread_near_device(e->device, 1);      // pinned memory on the device's socket
    hipError_t he = hipHostMalloc(&e->h_block, h_total, hipHostMallocDefault);
    if (he == hipSuccess) memset(e->h_block, 0, h_total);
    mic_bind_thread_near_device(e->device, 0);
    HIPTRY(he);
  }
  HIPTRY(hipMalloc(&e->d_block, d_total));
  char* hp = (char*)e->h_block; char* dp = (char*)e->d_block;
  e->h_results = (uint32_t*)hp; hp += h_res;
  if (extended) { e->h_rows = (uint32_t*)hp; hp += h_rows; }
  e->batches.resize(nb);
  for (size_t b = 0; b < nb; ++b) {
    Batch& B = e->batches[b];
    B.first_read = index_batches[b];
    B.max_reads = max_reads; B.max_cont = max_containers;
    B.extended = extended != 0;
    B.h_rp = (uint32_t*)hp; hp += sz_rp;
    B.h_cont = (uint16_t*)hp; hp += sz_ct;
    B.h_flagged = (uint32_t*)hp; hp += sz_fl;
    B.d_rp = (uint32_t*)dp; dp += sz_rp;
    B.d_cont = (uint16_t*)dp; dp += sz_ct;
    B.d_results = (uint32_t*)dp; dp += sz_res;
    if (extended) { B.d_rows = (uint32_t*)dp; dp += sz_rows; }
    B.d_flagged = (uint32_t*)dp; dp += sz_fl;
    B.d_crowd = (uint32_t*)dp; dp += sz_crowd;
    HIPTRY(mic_stream_get(&B.stream));
    HIPTRY(mic_event_get(&B.done, false));
    HIPTRY(mic_event_get(&B.ev_up, false));
    HIPTRY(mic_event_get(&B.ev_k, false));
    reads_pointer[b] = B.h_rp;
    containers[b] = B.h_cont;
  }
  *results = e->h_results;
  if (rows) *rows = e->h_rows;
  return MIC_OK;
}

int mic_batch_ready(mic_engine* e, size_t batch, size_t n_reads, size_t n_containers) {
  if (!e || batch >= e->batches.size()) return fail(MIC_E_INVALID, "bad batch id");
  Batch& B = e->batches[batch];
  if (n_reads > B.max_reads || n_containers > B.max_cont)
    return fail(MIC_E_INVALID, "batch %zu exceeds its allocation (%zu reads, %zu containers)", batch, n_reads, n_containers);
  B.n_reads = n_reads; B.n_cont = n_containers;
  return MIC_OK;
}

// queryBatch of one engine.  src == nullptr: the batch's reads come from the engine's own pinned buffers (H2D on the upload stream);
// otherwise they are ALREADY on device src_device in another engine's batch buffers (uploaded there, `src->ev_up` says when) and come
// over by peer copy on this batch's stream - one upload serves every engine of a table-sharded group.
static int batch_query_impl(mic_engine* e, size_t batch, int extended, const Batch* src, int src_device) {
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  std::lock_guard<std::mutex> lock(e->submit_mu);
  int rc = set_device(e);
  if (rc) return rc;
  Batch& B = e->batches[batch];
  if (extended && !B.d_rows) return fail(MIC_E_STATE, "batches were not allocated for extended results");
  hipStream_t s = B.stream, up = e->up_stream, down = e->down_stream;
  if (!src) {
    HIPTRY(hipMemcpyAsync(B.d_rp, B.h_rp, (B.n_reads + 1) * 4, hipMemcpyHostToDevice, up));
    HIPTRY(hipMemcpyAsync(B.d_cont, B.h_cont, (B.n_cont + 16) * 2, hipMemcpyHostToDevice, up));
    HIPTRY(hipEventRecord(B.ev_up, up));
    HIPTRY(hipStreamWaitEvent(s, B.ev_up, 0));
  } else {
    HIPTRY(hipStreamWaitEvent(s, src->ev_up, 0));
    HIPTRY(hipMemcpyPeerAsync(B.d_rp, e->device, src->d_rp, src_device, (B.n_reads + 1) * 4, s));
    HIPTRY(hipMemcpyPeerAsync(B.d_cont, e->device, src->d_cont, src_device, (B.n_cont + 16) * 2, s));
  }
  HIPTRY(hipMemsetAsync(B.d_flagged, 0, 4, s));
  MicQueryArgs a;
  a.t = e->table; a.reads_ptr = B.d_rp; a.cont = B.d_cont; a.n_reads = (uint32_t)B.n_reads;
  a.row_words = e->cfg.row_words; a.results = B.d_results; a.rows = extended ? B.d_rows : nullptr;
  a.flagged = B.d_flagged; a.flagged_cap = kFlaggedCap;
  mic_crowd_attach(a, B.d_crowd, B.max_reads);
  HIPTRY(mic_launch_query(a, e->slot_class, e->n_cu, s));
  HIPTRY(hipEventRecord(B.ev_k, s));
  HIPTRY(hipStreamWaitEvent(down, B.ev_k, 0));
  HIPTRY(hipMemcpyAsync(e->h_results + B.first_read * MIC_RESULT_WORDS, B.d_results, B.n_reads * MIC_RESULT_WORDS * 4,
                        hipMemcpyDeviceToHost, down));
  if (extended)
    HIPTRY(hipMemcpyAsync(e->h_rows + B.first_read * (size_t)e->cfg.row_words, B.d_rows,
                          B.n_reads * (size_t)e->cfg.row_words * 4, hipMemcpyDeviceToHost, down));
  HIPTRY(hipMemcpyAsync(B.h_flagged, B.d_flagged, 4, hipMemcpyDeviceToHost, down));   // count only; ids stay on the device
  HIPTRY(hipEventRecord(B.done, down));
  B.scheduled = true; B.resolved = false; B.extended = extended != 0;
  return MIC_OK;
}

int mic_batch_query(mic_engine* e, size_t batch, int extended, int followup) {
  (void)followup;
  if (!e || batch >= e->batches.size()) return fail(MIC_E_INVALID, "bad batch id");
  return batch_query_impl(e, batch, extended, nullptr, 0);
}

int mic_batch_query_group(mic_engine* const* engines, size_t n_engines, size_t batch, int extended) {
  if (!engines || n_engines == 0 || !engines[0] || batch >= engines[0]->batches.size()) return fail(MIC_E_INVALID, "bad argument");
  const Batch& B0 = engines[0]->batches[batch];
  for (size_t i = 1; i < n_engines; ++i) {
    mic_engine* e = engines[i];
    if (!e || batch >= e->batches.size()) return fail(MIC_E_INVALID, "bad engine or batch id");
    Batch& B = e->batches[batch];
    if (B0.n_reads > B.max_reads || B0.n_cont > B.max_cont) return fail(MIC_E_INVALID, "engine %zu: batch %zu is smaller than engine 0's", i, batch);
    B.n_reads = B0.n_reads; B.n_cont = B0.n_cont;
  }
  mic_peer_enable_engines(engines, n_engines);
  int rc = batch_query_impl(engines[0], batch, extended, nullptr, 0);
  for (size_t i = 1; i < n_engines && rc == MIC_OK; ++i) rc = batch_query_impl(engines[i], batch, extended, &B0, engines[0]->device);
  return rc;
}

int mic_batch_wait(mic_engine* e, size_t batch) {
  if (!e || batch >= e->batches.size()) return fail(MIC_E_INVALID, "bad batch id");
  Batch& B = e->batches[batch];
  if (!B.scheduled) return fail(MIC_E_STATE, "batch %zu was not queried", batch);
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(hipEventSynchronize(B.done));
  if (!B.resolved) {
    std::lock_guard<std::mutex> lock(e->submit_mu);
    if (!B.resolved) {
      uint32_t nf = B.h_flagged[0];
      if (nf) {
        // reads whose row overflowed: exact dense recount (ids beyond the list capacity: whole batch)
        if (nf > kFlaggedCap) {
          rc = run_dense(e, B.d_rp, B.d_cont, nullptr, B.n_reads, B.d_results, B.extended ? B.d_rows : nullptr, B.stream);
        } else {
          rc = run_dense(e, B.d_rp, B.d_cont, B.d_flagged + 1, nf, B.d_results, B.extended ? B.d_rows : nullptr, B.stream);
        }
        if (rc) return rc;
        HIPTRY(hipMemcpy(e->h_results + B.first_read * MIC_RESULT_WORDS, B.d_results, B.n_reads * MIC_RESULT_WORDS * 4,
                         hipMemcpyDeviceToHost));
        if (B.extended)
          HIPTRY(hipMemcpy(e->h_rows + B.first_read * (size_t)e->cfg.row_words, B.d_rows,
                           B.n_reads * (size_t)e->cfg.row_words * 4, hipMemcpyDeviceToHost));
      }
      B.resolved = true;
    }
  }
  return MIC_OK;
}

int mic_batch_dense_counts(mic_engine* e, size_t batch, size_t read_in_batch, uint32_t* counts) {
  if (!e || batch >= e->batches.size() || !counts) return fail(MIC_E_INVALID, "bad argument");
  Batch& B = e->batches[batch];
  if (!B.scheduled || read_in_batch >= B.n_reads) return fail(MIC_E_INVALID, "bad read index");
  int rc = set_device(e);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(e->submit_mu);
  const uint32_t T = e->cfg.num_targets ? e->cfg.num_targets : 1;
  uint32_t* d_counts = nullptr; uint32_t* d_id = nullptr;
  uint32_t id = (uint32_t)read_in_batch;
  HIPTRY(hipEventSynchronize(B.done));
  HIPTRY(hipMalloc(&d_counts, (size_t)T * 4));
  hipError_t he = hipMalloc(&d_id, 4);
  if (he == hipSuccess) he = hipMemcpyAsync(d_id, &id, 4, hipMemcpyHostToDevice, B.stream);
  if (he == hipSuccess) he = mic_launch_dense_count(e->table, e->slot_class, B.d_rp, B.d_cont, d_id, 1, T, d_counts, B.stream);
  if (he == hipSuccess) he = hipMemcpyAsync(counts, d_counts, (size_t)T * 4, hipMemcpyDeviceToHost, B.stream);
  if (he == hipSuccess) he = hipStreamSynchronize(B.stream);
  hipFree(d_counts);
  if (d_id) hipFree(d_id);
  if (he != hipSuccess) return fail(MIC_E_HIP, "dense counts: %s", hipGetErrorString(he));
  return MIC_OK;
}

// Table-sharded batches (the reference's multi-GPU mode, CuClarkDB.cu:934-1001: queryBatch on every device, the partial rows
// copied to device 0 with cudaMemcpyPeer one device after another and summed there by mergeKernel, then resultKernel).
// Here the sum is READ-RANGE OWNED: engine j fetches the rows of reads [n j / N, n (j + 1) / N) from the other N - 1 engines
// (peer copies, all engines at once on their own streams), sums and finishes them, and writes results and rows of its range
// straight into engines[0]'s host arrays.
int mic_batch_merge_shards(mic_engine* const* engines, size_t n_engines, size_t batch) {
  if (!engines || n_engines == 0 || !engines[0]) return fail(MIC_E_INVALID, "bad argument");
  mic_engine* dst = engines[0];
  if (batch >= dst->batches.size()) return fail(MIC_E_INVALID, "bad batch id");
  Batch& D = dst->batches[batch];
  if (!D.d_rows || !dst->h_rows) return fail(MIC_E_STATE, "batches must be allocated and queried with extended = 1");
  const uint32_t rw = dst->cfg.row_words;
  for (size_t i = 0; i < n_engines; ++i) {
    mic_engine* e = engines[i];
    if (!e || batch >= e->batches.size()) return fail(MIC_E_INVALID, "bad engine or batch id");
    const Batch& B = e->batches[batch];
    if (!B.d_rows || e->cfg.row_words != rw || B.n_reads != D.n_reads || !B.extended)
      return fail(MIC_E_STATE, "engine %zu: batch %zu does not match (rows, row width or read count)", i, batch);
    int rc = mic_batch_wait(e, batch);          // kernels done, flagged reads resolved inside their shard
    if (rc) return rc;
  }
  mic_peer_enable_engines(engines, n_engines);
  const size_t n = D.n_reads;
  int rc = MIC_OK;
  for (size_t j = 0; j < n_engines && rc == MIC_OK; ++j) {
    mic_engine* e = engines[j];
    Batch& J = e->batches[batch];
    const size_t lo = n * j / n_engines, hi = n * (j + 1) / n_engines, len = hi - lo;
    if (!len) continue;
    if ((rc = set_device(e))) break;
    std::lock_guard<std::mutex> lock(e->submit_mu);
    if (n_engines > 1 && !J.d_peer) {
      HIPTRY(hipMalloc(&J.d_peer, (J.max_reads + 1) * (size_t)rw * 4));
      HIPTRY(hipMalloc(&J.d_acc, (J.max_reads + 1) * (size_t)rw * 4));
    }
    uint32_t* buf[2] = {J.d_rows + lo * rw, J.d_acc};   // the running sum ping-pongs between the range of the batch's own rows and a spare
    int c = 0; size_t got = 0;
    for (size_t p = 0; p < n_engines; ++p) {
      if (p == j) continue;
      const Batch& P = engines[p]->batches[batch];
      uint32_t* in = J.d_peer + got * len * rw;
      HIPTRY(hipMemcpyPeerAsync(in, e->device, P.d_rows + lo * rw, engines[p]->device, len * (size_t)rw * 4, J.stream));
      HIPTRY(mic_launch_merge_rows(buf[c], in, buf[c ^ 1], rw, len, nullptr, J.stream));
      c ^= 1; ++got;
    }
    uint32_t* cur = buf[c];
    HIPTRY(mic_launch_result_from_rows(cur, rw, J.d_results + lo * MIC_RESULT_WORDS, len, J.stream));
    HIPTRY(hipMemcpyAsync(dst->h_results + (D.first_read + lo) * MIC_RESULT_WORDS, J.d_results + lo * MIC_RESULT_WORDS,
                          len * MIC_RESULT_WORDS * 4, hipMemcpyDeviceToHost, J.stream));
    HIPTRY(hipMemcpyAsync(dst->h_rows + (D.first_read + lo) * (size_t)rw, cur, len * (size_t)rw * 4, hipMemcpyDeviceToHost, J.stream));
  }
  for (size_t j = 0; j < n_engines; ++j) {
    if (set_device(engines[j]) != MIC_OK) continue;
    hipError_t he = hipStreamSynchronize(engines[j]->batches[batch].stream);
    if (he != hipSuccess && rc == MIC_OK) rc = fail(MIC_E_HIP, "merge of the table shards: %s", hipGetErrorString(he));
  }
  return rc;
}

int mic_peer_matrix(int* matrix, int n_devices) {
  if (!matrix || n_devices < 1 || n_devices > 64) return fail(MIC_E_INVALID, "bad argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n_devices > n) return fail(MIC_E_INVALID, "%d devices asked for, %d present", n_devices, n);
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < n_devices; ++j) matrix[i * n_devices + j] = mic_peer_enable(i, j) ? 1 : 0;
  return MIC_OK;
}

int mic_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!free_bytes || !total_bytes) return fail(MIC_E_INVALID, "null argument");
  int cur = 0;
  hipGetDevice(&cur);
  size_t f = 0, t = 0;
  hipError_t he = hipSetDevice(device);
  if (he == hipSuccess) he = hipMemGetInfo(&f, &t);
  hipSetDevice(cur);
  if (he != hipSuccess) return fail(MIC_E_HIP, "hipMemGetInfo: %s", hipGetErrorString(he));
  *free_bytes = f; *total_bytes = t;
  return MIC_OK;
}

int mic_db_kernel_name(const mic_engine* e, char* buf, size_t cap) {
  if (!e || !buf || cap < 2) return fail(MIC_E_INVALID, "bad argument");
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  return mic_query_kernel_name(e->table, e->slot_class, buf, cap);
}

// The database into several engines from one read of the files: every chunk is read once into pinned memory and uploaded to the
// devices that need it; then one thread per device builds the tables of its engines (reference: one read() loop over the files
// filling every device's parts, CuClarkDB.cu:604-808).  What a device needs follows the layout: engines that answer for a BUCKET
// range (direct and minimizer layouts cut by mic_db_set_part - the reference's cut, CuClarkDB.cu:566-574) need that range of the
// images only, so a database larger than one device's memory loads as long as its share does; a super-k-mer part is a slot range of
// the RESIDENT table, every k-mer of the images may land in it, so those devices get the whole images - and give them back as soon
// as their last engine is built.
int mic_db_load_files_multi(mic_engine* const* engines, size_t n_engines, const char* prefix, int key_bytes, uint32_t sampling) {
  if (!engines || !n_engines || !prefix) return fail(MIC_E_INVALID, "null argument");
  for (size_t i = 0; i < n_engines; ++i) {
    if (!engines[i]) return fail(MIC_E_INVALID, "null engine");
    if (engines[i]->cfg.k != engines[0]->cfg.k) return fail(MIC_E_INVALID, "the engines differ in k");
  }
  std::string p(prefix);
  FILE* fs = fopen((p + ".sz").c_str(), "rb");
  FILE* fk = fopen((p + ".ky").c_str(), "rb");
  FILE* fl = fopen((p + ".lb").c_str(), "rb");
  struct Eng { uint64_t s0 = 0, s1 = 0, base_elems = 0, base_rank = 0; };       // bucket range, elements and non-empty buckets in front of it
  struct Dev {
    int device; std::vector<size_t> eng; uint8_t* d_sz = nullptr; void* d_ky = nullptr; uint16_t* d_lb = nullptr; hipStream_t stream = nullptr;
    uint64_t lo = ~0ull, hi = 0, el_lo = 0, el_hi = 0;     // buckets [lo, hi) of the images live here = elements [el_lo, el_hi)
    double build_s = 0;
  };
  std::vector<Dev> devs;
  std::vector<Eng> er(n_engines);
  uint8_t* h_sz = nullptr;
  int rc = MIC_OK;
  const bool timing = getenv("MIC_LOAD_TIMING") != nullptr;
  struct timespec t_prev; clock_gettime(CLOCK_MONOTONIC, &t_prev);
  auto lap = [&](const char* what) {
    if (!timing) return;
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[load x%zu] %s: %.3f s\n", n_engines, what, (t.tv_sec - t_prev.tv_sec) + (t.tv_nsec - t_prev.tv_nsec) / 1e9);
    t_prev = t;
  };
  do {
    if (!fs) { rc = fail(MIC_E_IO, "Failed to open %s.sz", prefix); break; }
    if (!fk) { rc = fail(MIC_E_IO, "Failed to open %s.ky", prefix); break; }
    if (!fl) { rc = fail(MIC_E_IO, "Failed to open %s.lb", prefix); break; }
    fseeko(fs, 0, SEEK_END);
    const uint64_t htsize = (uint64_t)ftello(fs);
    fseeko(fs, 0, SEEK_SET);
    uint64_t z0 = 0, z1 = 0;
    if ((rc = check_shard(htsize, z0, z1))) break;
    if (key_bytes == 0) key_bytes = mic_key_bytes_rule(htsize, engines[0]->cfg.k);
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) { rc = fail(MIC_E_INVALID, "key_bytes must be 2, 4 or 8"); break; }
    h_sz = (uint8_t*)malloc(htsize);
    if (!h_sz) { rc = fail(MIC_E_NOMEM, "out of host memory for bucket sizes"); break; }
    if (!read_file_parallel(fs, h_sz, htsize)) { rc = fail(MIC_E_IO, "short read on %s.sz", prefix); break; }
    // every engine's bucket range (the whole table unless mic_db_set_part cut a bucket-range layout); elements and non-empty
    // buckets in front of each cut from ONE pass over the sizes
    std::vector<uint64_t> cuts;
    for (size_t i = 0; i < n_engines && !rc; ++i) {
      er[i].s0 = 0; er[i].s1 = htsize;
      rc = apply_part(engines[i], htsize, er[i].s0, er[i].s1);
      cuts.push_back(er[i].s0); cuts.push_back(er[i].s1);
    }
    if (rc) break;
    cuts.push_back(0); cuts.push_back(htsize);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<uint64_t> el_at(cuts.size(), 0), rank_at(cuts.size(), 0);
    {
      uint64_t el = 0, rk = 0;
      for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        el_at[c] = el; rank_at[c] = rk;
        uint64_t e1 = 0, r1 = 0;
        sum_sizes_parallel(h_sz, cuts[c], cuts[c + 1], &e1, &r1);
        el += e1; rk += r1;
      }
      el_at[cuts.size() - 1] = el; rank_at[cuts.size() - 1] = rk;
    }
    auto at = [&](uint64_t bucket) { return (size_t)(std::lower_bound(cuts.begin(), cuts.end(), bucket) - cuts.begin()); };
    lap("read .sz, sum bucket sizes");
    for (size_t i = 0; i < n_engines; ++i) {
      er[i].base_elems = el_at[at(er[i].s0)]; er[i].base_rank = rank_at[at(er[i].s0)];
      size_t d = 0;
      while (d < devs.size() && devs[d].device != engines[i]->device) ++d;
      if (d == devs.size()) { devs.emplace_back(); devs[d].device = engines[i]->device; devs[d].stream = engines[i]->stream; }
      devs[d].eng.push_back(i);
      devs[d].lo = std::min(devs[d].lo, er[i].s0); devs[d].hi = std::max(devs[d].hi, er[i].s1);
    }
    for (Dev& dv : devs) {
      dv.el_lo = el_at[at(dv.lo)]; dv.el_hi = el_at[at(dv.hi)];
      if (hipSetDevice(dv.device) != hipSuccess) { rc = fail(MIC_E_HIP, "hipSetDevice failed"); break; }
      for (size_t i : dv.eng) if (engines[i]->db_loaded) mic_db_unload(engines[i]);
      const uint64_t n_el = dv.el_hi - dv.el_lo;
      hipError_t he = hipMalloc(&dv.d_sz, dv.hi - dv.lo);
      if (he == hipSuccess) he = hipMalloc(&dv.d_ky, n_el * key_bytes + 16);
      if (he == hipSuccess) he = hipMalloc(&dv.d_lb, n_el * 2 + 16);
      if (he != hipSuccess) {
        size_t fr = 0, tot = 0;
        hipMemGetInfo(&fr, &tot);
        rc = fail(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "DB image allocation on device %d (%.1f GB of images for buckets [%llu, %llu), %.1f GB free of %.1f): %s",
                  dv.device, ((dv.hi - dv.lo) + n_el * (double)(key_bytes + 2)) / 1e9, (unsigned long long)dv.lo, (unsigned long long)dv.hi, fr / 1e9, tot / 1e9, hipGetErrorString(he));
        break;
      }
    }
    if (rc) break;
    std::vector<UploadDst> dz, dk, dl;
    uint64_t up_bytes = 0;
    for (Dev& dv : devs) {
      dz.push_back({dv.device, dv.d_sz, dv.stream, dv.lo, dv.hi});
      dk.push_back({dv.device, dv.d_ky, dv.stream, dv.el_lo * (uint64_t)key_bytes, dv.el_hi * (uint64_t)key_bytes});
      dl.push_back({dv.device, dv.d_lb, dv.stream, dv.el_lo * 2, dv.el_hi * 2});
      up_bytes += (dv.hi - dv.lo) + (dv.el_hi - dv.el_lo) * (uint64_t)(key_bytes + 2);
    }
    UploadStage stage;
    if ((rc = upload_file_range_multi(fs, dz, "the .sz file", &stage))) break;
    if ((rc = upload_file_range_multi(fk, dk, "the .ky file", &stage))) break;
    if ((rc = upload_file_range_multi(fl, dl, "the .lb file", &stage))) break;
    if (timing) fprintf(stderr, "[load x%zu] images on %zu device(s): %.2f GB uploaded from one read of the files\n", n_engines, devs.size(), up_bytes / 1e9);
    lap("images uploaded to the devices");
    // one thread per device; the engines of a device one after another (a build sizes its staging area from the free HBM)
    std::vector<int> rcs(devs.size(), MIC_OK);
    std::vector<std::string> msgs(devs.size());
    std::vector<std::thread> th;
    for (size_t d = 0; d < devs.size(); ++d)
      th.emplace_back([&, d] {
        Dev& dv = devs[d];
        struct timespec ta; clock_gettime(CLOCK_MONOTONIC, &ta);
        for (size_t i : dv.eng) {
          mic_engine* e = engines[i];
          int r = set_device(e);
          if (!r) r = build_from_device(e, dv.d_sz + (er[i].s0 - dv.lo), htsize, er[i].s0, er[i].s1,
                                        (const char*)dv.d_ky + (er[i].base_elems - dv.el_lo) * key_bytes, key_bytes,
                                        dv.d_lb + (er[i].base_elems - dv.el_lo), sampling, er[i].base_rank);
          if (r) { rcs[d] = r; msgs[d] = g_err; break; }
        }
        // the images go as soon as this device's last engine is built (the other devices may still be building)
        hipSetDevice(dv.device);
        if (dv.d_sz) { hipFree(dv.d_sz); dv.d_sz = nullptr; }
        if (dv.d_ky) { hipFree(dv.d_ky); dv.d_ky = nullptr; }
        if (dv.d_lb) { hipFree(dv.d_lb); dv.d_lb = nullptr; }
        struct timespec tb; clock_gettime(CLOCK_MONOTONIC, &tb);
        dv.build_s = (tb.tv_sec - ta.tv_sec) + (tb.tv_nsec - ta.tv_nsec) / 1e9;
      });
    for (auto& t : th) t.join();
    for (size_t d = 0; d < devs.size(); ++d) if (rcs[d]) { rc = fail(rcs[d], "%s", msgs[d].c_str()); break; }
    if (timing)
      for (Dev& dv : devs) fprintf(stderr, "[load x%zu] device %d: %zu table(s) built in %.3f s\n", n_engines, dv.device, dv.eng.size(), dv.build_s);
    lap("tables built");
  } while (0);
  if (fs) fclose(fs);
  if (fk) fclose(fk);
  if (fl) fclose(fl);
  free(h_sz);
  for (Dev& dv : devs) {
    hipSetDevice(dv.device);
    if (dv.d_sz) hipFree(dv.d_sz);
    if (dv.d_ky) hipFree(dv.d_ky);
    if (dv.d_lb) hipFree(dv.d_lb);
  }
  return rc;
}

int mic_batch_check(mic_engine* e, size_t batch, int* done) {
  if (!e || batch >= e->batches.size() || !done) return fail(MIC_E_INVALID, "bad argument");
  Batch& B = e->batches[batch];
  if (!B.scheduled) { *done = 0; return MIC_OK; }
  hipError_t he = hipEventQuery(B.done);
  if (he == hipSuccess) *done = 1;
  else if (he == hipErrorNotReady) *done = 0;
  else return fail(MIC_E_HIP, "hipEventQuery: %s", hipGetErrorString(he));
  return MIC_OK;
}

int mic_thread_bind_near_device(mic_engine* e, int on) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  mic_bind_thread_near_device(e->device, on);
  return MIC_OK;
}

int mic_sync(mic_engine* e) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(hipDeviceSynchronize());
  return MIC_OK;
}

int mic_batches_free(mic_engine* e) {
  if (!e) return fail(MIC_E_INVALID, "null engine");
  int rc = set_device(e);
  if (rc) return rc;
  hipDeviceSynchronize();
  free_batches(e);
  return MIC_OK;
}

// ---- device-resident entry points ---------------------------------------------------------------------
int mic_query_device(mic_engine* e, const uint32_t* d_rp, const uint16_t* d_cont, size_t n_reads, uint32_t* d_results,
                     uint32_t* d_rows, void* stream) {
  if (!e || !d_rp || !d_cont || !d_results) return fail(MIC_E_INVALID, "null argument");
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  if (n_reads > 0xFFFFFFF0ull) return fail(MIC_E_INVALID, "too many reads in one call");
  if (((uintptr_t)d_cont & 3) || ((uintptr_t)d_rp & 3)) return fail(MIC_E_INVALID, "d_containers and d_reads_pointer must be 4-byte aligned");
  int rc = set_device(e);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : e->stream;
  HIPTRY(hipMemsetAsync(e->d_flagged, 0, 4, s));
  MicQueryArgs a;
  a.t = e->table; a.reads_ptr = d_rp; a.cont = d_cont; a.n_reads = (uint32_t)n_reads;
  a.row_words = e->cfg.row_words; a.results = d_results; a.rows = d_rows;
  a.flagged = e->d_flagged; a.flagged_cap = e->flagged_cap;
  if (e->table.side && n_reads > e->crowd_reads) {      // (first call, or a larger one: the engine's launches are one at a time)
    HIPTRY(hipStreamSynchronize(s));
    if (e->d_crowd) { hipFree(e->d_crowd); e->d_crowd = nullptr; e->crowd_reads = 0; }
    HIPTRY(hipMalloc(&e->d_crowd, mic_crowd_dims(n_reads).words * 4));
    e->crowd_reads = n_reads;
  }
  mic_crowd_attach(a, e->table.side ? e->d_crowd : nullptr, e->crowd_reads);
  HIPTRY(hipEventRecord(e->ev0, s));
  HIPTRY(mic_launch_query(a, e->slot_class, e->n_cu, s));
  HIPTRY(hipEventRecord(e->ev1, s));
  e->timed = true;
  e->last_n_reads = n_reads;
  return MIC_OK;
}

int mic_last_query_ms(mic_engine* e, float* ms) {
  if (!e || !ms) return fail(MIC_E_INVALID, "null argument");
  if (!e->timed) return fail(MIC_E_STATE, "no query was launched");
  HIPTRY(hipEventSynchronize(e->ev1));
  HIPTRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
  return MIC_OK;
}

int mic_last_crowd_stats(mic_engine* e, uint32_t out[4]) {
  if (!e || !out) return fail(MIC_E_INVALID, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!e->timed || !e->d_crowd || !e->table.side) return MIC_OK;
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(hipEventSynchronize(e->ev1));
  HIPTRY(hipMemcpy(out, e->d_crowd, 16, hipMemcpyDeviceToHost));
  return MIC_OK;
}

int mic_debug_fetch_crowd(mic_engine* e, uint32_t* out, size_t words, uint32_t caps[3]) {
  if (!e || !out || !caps) return fail(MIC_E_INVALID, "null argument");
  if (!e->d_crowd || !e->timed) return fail(MIC_E_STATE, "no work area");
  int rc = set_device(e);
  if (rc) return rc;
  const MicCrowdDims d = mic_crowd_dims(e->crowd_reads);
  caps[0] = d.pend_cap; caps[1] = d.item_cap; caps[2] = d.pool_cap;
  HIPTRY(hipEventSynchronize(e->ev1));
  HIPTRY(hipMemcpy(out, e->d_crowd, (words < d.words ? words : d.words) * 4, hipMemcpyDeviceToHost));
  return MIC_OK;
}

int mic_resolve_flagged_device(mic_engine* e, const uint32_t* d_rp, const uint16_t* d_cont, uint32_t* d_results,
                               uint32_t* d_rows, void* stream, size_t* n_resolved) {
  if (!e || !d_rp || !d_cont || !d_results) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : e->stream;
  uint32_t nf = 0;
  HIPTRY(hipMemcpyAsync(&nf, e->d_flagged, 4, hipMemcpyDeviceToHost, s));
  HIPTRY(hipStreamSynchronize(s));
  if (n_resolved) *n_resolved = nf;
  if (!nf) return MIC_OK;
  if (nf > e->flagged_cap) {
    if (n_resolved) *n_resolved = e->last_n_reads;
    return run_dense(e, d_rp, d_cont, nullptr, e->last_n_reads, d_results, d_rows, s);
  }
  return run_dense(e, d_rp, d_cont, e->d_flagged + 1, nf, d_results, d_rows, s);
}

int mic_merge_rows_device(mic_engine* e, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, void* stream) {
  if (!e || !a || !b || !out) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(mic_launch_merge_rows(a, b, out, e->cfg.row_words, n, nullptr, stream ? (hipStream_t)stream : e->stream));
  return MIC_OK;
}

int mic_result_from_rows_device(mic_engine* e, const uint32_t* rows, uint32_t* results, size_t n, void* stream) {
  if (!e || !rows || !results) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(mic_launch_result_from_rows(rows, e->cfg.row_words, results, n, stream ? (hipStream_t)stream : e->stream));
  return MIC_OK;
}

int mic_result_from_dense_device(mic_engine* e, const uint32_t* d_counts, const uint32_t* d_ids, size_t n_ids, uint32_t* d_results,
                                 uint32_t* d_rows, void* stream) {
  if (!e || !d_counts || !d_results) return fail(MIC_E_INVALID, "null argument");
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(mic_launch_dense_finish(d_counts, d_ids, n_ids, e->cfg.num_targets ? e->cfg.num_targets : 1, d_results, d_rows,
                                 e->cfg.row_words, stream ? (hipStream_t)stream : e->stream));
  return MIC_OK;
}

int mic_probe_stats_device(mic_engine* e, const uint32_t* d_rp, const uint16_t* d_cont, size_t n_reads, uint64_t out[4]) {
  if (!e || !d_rp || !d_cont || !out) return fail(MIC_E_INVALID, "null argument");
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  int rc = set_device(e);
  if (rc) return rc;
  unsigned long long* d = nullptr;
  unsigned long long h[4] = {0, 0, 0, 0};
  HIPTRY(hipMalloc(&d, 32));
  hipError_t he = hipMemsetAsync(d, 0, 32, e->stream);
  if (he == hipSuccess) he = mic_launch_probe_stats(e->table, e->slot_class, d_rp, d_cont, n_reads, d, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, e->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  hipFree(d);
  if (he != hipSuccess) return fail(MIC_E_HIP, "probe stats: %s", hipGetErrorString(he));
  for (int i = 0; i < 4; ++i) out[i] = h[i];
  return MIC_OK;
}

int mic_count_dense_device(mic_engine* e, const uint32_t* d_rp, const uint16_t* d_cont, const uint32_t* d_ids,
                           size_t n_ids, uint32_t* d_counts, void* stream) {
  if (!e || !d_rp || !d_cont || !d_counts) return fail(MIC_E_INVALID, "null argument");
  if (!e->db_loaded) return fail(MIC_E_STATE, "no database loaded");
  int rc = set_device(e);
  if (rc) return rc;
  HIPTRY(mic_launch_dense_count(e->table, e->slot_class, d_rp, d_cont, d_ids, n_ids, e->cfg.num_targets ? e->cfg.num_targets : 1,
                                d_counts, stream ? (hipStream_t)stream : e->stream));
  return MIC_OK;
}

}  // extern "C"
