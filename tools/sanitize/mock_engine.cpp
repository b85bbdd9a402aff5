// mock_engine.cpp - a stand-in for libmi_clark.so's DEVICE entry points, for sanitizer builds of the host code only
// (tests/test_sanitizers.py: AddressSanitizer + UBSan, ThreadSanitizer; there is no GPU sanitizer on this pool).
// exe-side code under test, unchanged: classifier*.cpp / classifier_feeders.hpp (FileFeeder, SegmentFeeder, PairedFileFeeder, DeviceGzFeeder, PairedSource, GzSource,
// InflateStream, strip_fastq, run_stream's loader / device / writer threads), cli_main.cpp, and mic_host.cpp (indexer, packer,
// CSV).  What is mocked: the engine.  mic_ingest_classify here "classifies" a slot on the CPU: it walks the records of the
// slot's bytes and writes one CSV line "<name>,<length>" per record - a pure function of the input, so the test can check
// that every record of the input arrives exactly once and in order through the threaded pipeline.
// NEVER linked into the product.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <zlib.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "mi_clark.h"

struct mic_engine {
  std::vector<std::vector<uint8_t>> raw;
  std::vector<std::string> csv;
  size_t max_bytes = 0;
  mic_config cfg;
};

static thread_local char g_err[256] = "mock engine";

extern "C" {

const char* mic_last_error(void) { return g_err; }
int mic_device_count(int* count) { *count = 1; return MIC_OK; }
int mic_create(const mic_config* cfg, mic_engine** out) { mic_engine* e = new mic_engine(); e->cfg = *cfg; *out = e; return MIC_OK; }
int mic_destroy(mic_engine* e) { delete e; return MIC_OK; }
int mic_db_load_files(mic_engine*, const char*, int, uint32_t, uint64_t, uint64_t) { return MIC_OK; }
int mic_db_load_files_multi(mic_engine* const*, size_t, const char*, int, uint32_t) { return MIC_OK; }
int mic_db_set_part(mic_engine*, uint32_t, uint32_t) { return MIC_OK; }
int mic_peer_matrix(int* m, int n) { for (int i = 0; i < n * n; ++i) m[i] = 1; return MIC_OK; }
int mic_device_memory(int, uint64_t* f, uint64_t* t) { *f = *t = (uint64_t)1 << 40; return MIC_OK; }
int mic_db_kernel_name(const mic_engine*, char* buf, size_t cap) { return snprintf(buf, cap, "mock"); }
int mic_db_reserve_hbm(mic_engine*, uint64_t) { return MIC_OK; }
const char* mic_db_last_build_report(void) { return ""; }
int mic_db_get_info(const mic_engine*, mic_db_info* info) { memset(info, 0, sizeof(*info)); info->layout = MIC_LAYOUT_SUPER; return MIC_OK; }
int mic_thread_bind_near_device(mic_engine*, int) { return MIC_OK; }
int mic_db_build(const char* const*, const uint16_t*, size_t, int, uint64_t, int, uint32_t, uint32_t, const char*, int, int, uint32_t, uint64_t*) { return MIC_E_NODEVICE; }
const char* mic_db_build_error(void) { return "mock engine: no database builder"; }

int mic_ingest_alloc(mic_engine* e, size_t n_slots, size_t max_bytes, const char* const*, uint32_t, int, uint8_t** raw) {
  e->raw.assign(n_slots, std::vector<uint8_t>());
  e->csv.assign(n_slots, std::string());
  e->max_bytes = max_bytes;
  for (size_t i = 0; i < n_slots; ++i) { e->raw[i].resize(max_bytes); raw[i] = e->raw[i].data(); }   // exact size: an overrun is ASan's to find
  return MIC_OK;
}
int mic_ingest_free(mic_engine* e) { if (e) { e->raw.clear(); e->csv.clear(); } return MIC_OK; }

int mic_ingest_classify(mic_engine* e, size_t slot, size_t n_bytes, int flags, mic_ingest_result* out) {
  memset(out, 0, sizeof(*out));
  if (slot >= e->raw.size() || n_bytes > e->max_bytes) { snprintf(g_err, sizeof(g_err), "mock: bad slot or size"); return MIC_E_INVALID; }
  const uint8_t* p = e->raw[slot].data();
  std::string& csv = e->csv[slot];
  csv.clear();
  uint64_t n_reads = 0;
  size_t i = 0;
  const bool fastq = n_bytes && p[0] == '@', two = (flags & MIC_INGEST_FASTQ_2LINE) != 0, paired = (flags & MIC_INGEST_PAIRED) != 0;
  auto line_end = [&](size_t a) { const void* q = memchr(p + a, '\n', n_bytes - a); return q ? (size_t)((const uint8_t*)q - p) : n_bytes; };
  while (i < n_bytes) {
    const size_t he = line_end(i);
    size_t ne = i + 1;
    while (ne < he && p[ne] != ' ' && p[ne] != '\t') ++ne;
    size_t nl = ne - (i + 1); if (nl >= 40) nl = 39;
    uint64_t len = 0;
    size_t pos = he < n_bytes ? he + 1 : n_bytes;
    if (fastq) {
      const size_t se = pos < n_bytes ? line_end(pos) : n_bytes;
      len = se - pos;
      pos = se < n_bytes ? se + 1 : n_bytes;
      if (!two) for (int l = 0; l < 2 && pos < n_bytes; ++l) { const size_t x = line_end(pos); pos = x < n_bytes ? x + 1 : n_bytes; }
    } else {
      while (pos < n_bytes && p[pos] != '>') { const size_t x = line_end(pos); len += x - pos; pos = x < n_bytes ? x + 1 : n_bytes; }
    }
    csv.append((const char*)p + i + 1, nl);
    char num[32];
    snprintf(num, sizeof(num), ",%llu\n", (unsigned long long)(paired ? len - 1 : len));
    csv += num;
    ++n_reads;
    i = pos;
  }
  out->n_reads = n_reads; out->csv_bytes = csv.size(); out->csv = csv.data(); out->results = nullptr; out->status = MIC_INGEST_OK;
  return MIC_OK;
}

// table-sharded: the slot's owner "classifies"; the other engines of the group have nothing to add in the mock
int mic_ingest_classify_group(mic_engine* const* group, size_t n_group, size_t owner, size_t slot, size_t n_bytes, int flags, mic_ingest_result* out) {
  if (!group || owner >= n_group) return MIC_E_INVALID;
  return mic_ingest_classify(group[owner], slot, n_bytes, flags, out);
}

int mic_ingest_group_stats(mic_engine*, double* out, size_t cap) { for (size_t i = 0; i < cap && i < MIC_GROUP_STATS_FIELDS; ++i) out[i] = 0; return MIC_GROUP_STATS_FIELDS; }

// the batch API is only reached when a slot is handed back (MIC_INGEST_FALLBACK): the mock never does
int mic_batches_alloc(mic_engine*, size_t, size_t, size_t, const uint32_t*, int, uint32_t**, uint32_t**, uint32_t**, uint16_t**) { return MIC_E_NODEVICE; }
int mic_batch_ready(mic_engine*, size_t, size_t, size_t) { return MIC_E_NODEVICE; }
int mic_batch_query(mic_engine*, size_t, int, int) { return MIC_E_NODEVICE; }
int mic_batch_wait(mic_engine*, size_t) { return MIC_E_NODEVICE; }
int mic_batch_dense_counts(mic_engine*, size_t, size_t, uint32_t*) { return MIC_E_NODEVICE; }
int mic_batch_merge_shards(mic_engine* const*, size_t, size_t) { return MIC_E_NODEVICE; }
int mic_batch_query_group(mic_engine* const*, size_t, size_t, int) { return MIC_E_NODEVICE; }
int mic_batches_free(mic_engine*) { return MIC_OK; }
// ---- compressed input "on the device": the same entry points on the CPU, so that the command line's DeviceGzFeeder (batch
// arithmetic over the sampled offsets, slots filled in place, the hand-back path) runs under the sanitizers too.  "Device" memory
// is malloc memory of the exact size; a slot's device buffer is its raw buffer.  One gzip member or a block-gzip file, as the real
// entry point takes them; anything else: MIC_E_UNSUPPORTED.
int mic_gz_inflate_device(mic_engine*, const void* gz, size_t gz_bytes, void** d_text, size_t* n_text, uint32_t* crc32_expected) {
  const uint8_t* p = (const uint8_t*)gz;
  *d_text = nullptr; *n_text = 0;
  if (gz_bytes < 18 || p[0] != 0x1f || p[1] != 0x8b) return MIC_E_UNSUPPORTED;
  const bool bgzf = (p[3] & 4) && p[12] == 'B' && p[13] == 'C';
  std::string out;
  z_stream z; memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, 31) != Z_OK) return MIC_E_HIP;
  z.next_in = (Bytef*)p; z.avail_in = (uInt)gz_bytes;
  std::vector<uint8_t> buf(1 << 16);
  int members = 0;
  for (;;) {
    z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
    const int r = inflate(&z, Z_NO_FLUSH);
    out.append((const char*)buf.data(), buf.size() - z.avail_out);
    if (r == Z_STREAM_END) {
      ++members;
      if (z.avail_in == 0) break;
      if (!bgzf) { inflateEnd(&z); return MIC_E_UNSUPPORTED; }       // several ordinary members: the host's affair
      if (inflateReset(&z) != Z_OK) { inflateEnd(&z); return MIC_E_INVALID; }
    } else if (r != Z_OK) { inflateEnd(&z); return MIC_E_INVALID; }
  }
  inflateEnd(&z);
  if (crc32_expected) *crc32_expected = 0;
  uint8_t* t = (uint8_t*)malloc(out.size() ? out.size() : 1);
  memcpy(t, out.data(), out.size());
  *d_text = t; *n_text = out.size();
  return MIC_OK;
}
// the member in stripes: everything is inflated at open, the text is handed out a part at a time.  Several ordinary members: the
// first part goes out, the next call gives the file back (what the device does when a later stripe meets the end of the first member)
struct mic_gz_stream { uint8_t* text = nullptr; size_t n = 0, given = 0; uint32_t stripes = 1, at = 0; bool give_back = false; };
int mic_gz_stream_open(mic_engine* e, const void* gz, size_t gz_bytes, uint32_t stripes, mic_gz_stream** out, void** d_text, size_t* n_text) {
  const uint8_t* p = (const uint8_t*)gz;
  *out = nullptr; *d_text = nullptr; *n_text = 0;
  if (gz_bytes >= 18 && (p[3] & 4) && p[12] == 'B' && p[13] == 'C') return MIC_E_UNSUPPORTED;
  void* t = nullptr; size_t n = 0; uint32_t crc = 0;
  bool give_back = false;
  int rc = mic_gz_inflate_device(e, gz, gz_bytes, &t, &n, &crc);
  if (rc == MIC_E_UNSUPPORTED && gz_bytes >= 18 && p[0] == 0x1f && p[1] == 0x8b) {
    // (several members: the first member's text for the stripes in front of the give-back)
    z_stream z; memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, 31) != Z_OK) return MIC_E_HIP;
    std::string o; std::vector<uint8_t> buf(1 << 16);
    z.next_in = (Bytef*)p; z.avail_in = (uInt)gz_bytes;
    for (;;) {
      z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
      const int r = inflate(&z, Z_NO_FLUSH);
      o.append((const char*)buf.data(), buf.size() - z.avail_out);
      if (r == Z_STREAM_END) break;
      if (r != Z_OK) { inflateEnd(&z); return MIC_E_INVALID; }
    }
    inflateEnd(&z);
    t = malloc(o.size() ? o.size() : 1); memcpy(t, o.data(), o.size()); n = o.size();
    give_back = true; rc = MIC_OK;
  }
  if (rc != MIC_OK) return rc;
  mic_gz_stream* h = new mic_gz_stream;
  h->text = (uint8_t*)t; h->n = n; h->stripes = stripes ? stripes : 1; h->give_back = give_back;
  *out = h; *d_text = t; *n_text = n;
  return MIC_OK;
}
int mic_gz_stream_next(mic_gz_stream* h, size_t* n_final, int* done) {
  *n_final = 0; *done = 0;
  if (h->give_back && h->at >= 1) return MIC_E_UNSUPPORTED;
  ++h->at;
  h->given = h->at >= h->stripes && !h->give_back ? h->n : h->n / (h->stripes + 1) * h->at;
  *n_final = h->given; *done = h->given == h->n && !h->give_back;
  return MIC_OK;
}
int mic_gz_stream_close(mic_gz_stream* h, int keep_text) {
  if (!h) return MIC_OK;
  if (!keep_text) free(h->text);
  delete h;
  return MIC_OK;
}
int mic_gz_free_text(mic_engine*, void* t) { free(t); return MIC_OK; }
int mic_gz_copy_text(mic_engine*, const void* t, size_t off, size_t n, void* dst) { memcpy(dst, (const uint8_t*)t + off, n); return MIC_OK; }
int mic_gz_reserve(mic_engine*, size_t, uint32_t) { return MIC_OK; }
uint64_t mic_gz_reserve_bytes(size_t, uint32_t) { return 0; }
int mic_gz_release(mic_engine*) { return MIC_OK; }
}  // extern "C"

namespace {
// line starts of a text; ls[n_lines] = one behind the last line's end (+ 1 for a last line without its line end: as on the device)
std::vector<size_t> line_starts(const uint8_t* t, size_t n) {
  std::vector<size_t> ls(1, 0);
  for (size_t i = 0; i < n; ++i) if (t[i] == '\n') ls.push_back(i + 1);
  if (n && t[n - 1] != '\n') ls.push_back(n + 1);
  return ls;
}
bool sep(uint8_t c) { return c == ' ' || c == '/' || c == '\t' || c == '@'; }
void id_of(const uint8_t* p, size_t n, size_t& a, size_t& len) { a = 0; while (a < n && sep(p[a])) ++a; size_t b = a; while (b < n && !sep(p[b])) ++b; len = b - a; }
}  // namespace

struct mic_pairs {
  const uint8_t* t[2]; std::vector<size_t> ls[2];
  std::vector<uint64_t> off, samples;
  uint64_t n_rec = 0; uint32_t stride = 64;
  std::string merged(uint64_t r0, uint64_t r1) const {
    std::string o;
    for (uint64_t r = r0; r < r1; ++r) {
      const uint8_t* h = t[0] + ls[0][4 * r]; const size_t hn = ls[0][4 * r + 1] - 1 - ls[0][4 * r];
      size_t a, len; id_of(h, hn, a, len);
      o += '>'; o.append((const char*)h + a, len); o += '\n';
      o.append((const char*)t[0] + ls[0][4 * r + 1], ls[0][4 * r + 2] - 1 - ls[0][4 * r + 1]); o += 'N';
      o.append((const char*)t[1] + ls[1][4 * r + 1], ls[1][4 * r + 2] - 1 - ls[1][4 * r + 1]); o += '\n';
    }
    return o;
  }
};
struct mic_text {
  const uint8_t* t; size_t n; bool fasta = false;
  std::vector<uint64_t> rec, samples;       // rec[r] = offset of record r, rec[n_rec] = n
  uint64_t n_rec = 0; uint32_t stride = 64;
};

extern "C" {
int mic_pairs_index_device(mic_engine*, const void* t1, size_t n1, const void* t2, size_t n2, mic_pairs** out, uint64_t* n_records, uint32_t* status) {
  *out = nullptr; *n_records = 0; *status = 0;
  if (!n1 || !n2) { *status = MIC_PAIRS_BIG; return MIC_OK; }
  mic_pairs* p = new mic_pairs;
  p->t[0] = (const uint8_t*)t1; p->t[1] = (const uint8_t*)t2;
  p->ls[0] = line_starts(p->t[0], n1); p->ls[1] = line_starts(p->t[1], n2);
  const size_t l0 = p->ls[0].size() - 1, l1 = p->ls[1].size() - 1;
  if (l0 != l1 || l0 % 4 || !l0) { *status = MIC_PAIRS_LINES; delete p; return MIC_OK; }
  p->n_rec = l0 / 4;
  p->off.assign(p->n_rec + 1, 0);
  for (uint64_t r = 0; r < p->n_rec; ++r) {
    const uint8_t* h[2]; size_t hn[2], a[2], len[2];
    for (int i = 0; i < 2; ++i) { h[i] = p->t[i] + p->ls[i][4 * r]; hn[i] = p->ls[i][4 * r + 1] - 1 - p->ls[i][4 * r]; }
    if (!hn[0] || !hn[1] || h[0][0] != '@' || h[1][0] != '@') { *status |= MIC_PAIRS_HEADER; continue; }
    id_of(h[0], hn[0], a[0], len[0]); id_of(h[1], hn[1], a[1], len[1]);
    if (!len[0] || len[0] != len[1] || memcmp(h[0] + a[0], h[1] + a[1], len[0])) *status |= MIC_PAIRS_ID;
    p->off[r + 1] = p->off[r] + len[0] + 2 + (p->ls[0][4 * r + 2] - 1 - p->ls[0][4 * r + 1]) + 1 + (p->ls[1][4 * r + 2] - 1 - p->ls[1][4 * r + 1]) + 1;
  }
  if (*status) { delete p; return MIC_OK; }
  for (uint64_t i = 0; i < p->n_rec / p->stride + 2; ++i) p->samples.push_back(p->off[std::min<uint64_t>(i * p->stride, p->n_rec)]);
  *out = p; *n_records = p->n_rec;
  return MIC_OK;
}
int mic_pairs_offsets(const mic_pairs* p, const uint64_t** s, size_t* n, uint32_t* stride) { *s = p->samples.data(); *n = p->samples.size(); *stride = p->stride; return MIC_OK; }
static bool whole_strides(uint64_t n_rec, uint32_t stride, uint64_t r0, uint64_t r1) { return r0 < r1 && r1 <= n_rec && r0 % stride == 0 && (r1 % stride == 0 || r1 == n_rec); }
int mic_pairs_merge_to_slot(mic_engine* e, mic_pairs* p, uint64_t r0, uint64_t r1, size_t slot, size_t* n_bytes) {
  if (!whole_strides(p->n_rec, p->stride, r0, r1) || slot >= e->raw.size()) return MIC_E_INVALID;
  const std::string m = p->merged(r0, r1);
  if (m.size() > e->max_bytes) return MIC_E_INVALID;
  memcpy(e->raw[slot].data(), m.data(), m.size());
  *n_bytes = m.size();
  return MIC_OK;
}
int mic_pairs_text(mic_engine*, mic_pairs* p, uint64_t r0, uint64_t r1, void* dst, size_t cap, size_t* n_bytes) {
  if (!whole_strides(p->n_rec, p->stride, r0, r1)) return MIC_E_INVALID;
  const std::string m = p->merged(r0, r1);
  if (m.size() > cap) return MIC_E_INVALID;
  memcpy(dst, m.data(), m.size()); *n_bytes = m.size();
  return MIC_OK;
}
int mic_pairs_free(mic_engine*, mic_pairs* p) { delete p; return MIC_OK; }

int mic_text_index_device(mic_engine*, const void* t, size_t n, mic_text** out, uint64_t* n_records, uint32_t* status) {
  *out = nullptr; *n_records = 0; *status = 0;
  if (!n) { *status = MIC_PAIRS_BIG; return MIC_OK; }
  mic_text* p = new mic_text;
  p->t = (const uint8_t*)t; p->n = n;
  const std::vector<size_t> ls = line_starts(p->t, n);
  const size_t nl = ls.size() - 1;
  if (p->t[0] == '>') {
    p->fasta = true;
    for (size_t L = 0; L < nl; ++L) if (ls[L] < n && p->t[ls[L]] == '>') p->rec.push_back(ls[L]);
  } else if (p->t[0] == '@') {
    if (nl % 4 || !nl) { *status = MIC_PAIRS_LINES; delete p; return MIC_OK; }
    for (size_t r = 0; r < nl / 4; ++r) p->rec.push_back(ls[4 * r]);
  } else { *status = MIC_PAIRS_HEADER; delete p; return MIC_OK; }
  p->n_rec = p->rec.size();
  p->rec.push_back(n);
  for (uint64_t i = 0; i < p->n_rec / p->stride + 2; ++i) p->samples.push_back(p->rec[std::min<uint64_t>(i * p->stride, p->n_rec)]);
  *out = p; *n_records = p->n_rec;
  return MIC_OK;
}
int mic_text_index_front_device(mic_engine*, const void* t, size_t n, mic_text** out, uint64_t* n_records, uint64_t* n_used, uint32_t* status) {
  *out = nullptr; *n_records = 0; *n_used = 0; *status = 0;
  if (!n) return MIC_OK;
  const uint8_t* q = (const uint8_t*)t;
  if (q[0] != '@') { *status = MIC_PAIRS_HEADER; return MIC_OK; }
  std::vector<size_t> ls(1, 0);
  for (size_t i = 0; i < n; ++i) if (q[i] == '\n') ls.push_back(i + 1);
  const size_t whole = (ls.size() - 1) / 4;
  if (!whole) return MIC_OK;
  mic_text* p = new mic_text;
  p->t = q; p->n = ls[4 * whole];
  for (size_t r = 0; r < whole; ++r) p->rec.push_back(ls[4 * r]);
  p->n_rec = whole;
  p->rec.push_back(p->n);
  for (uint64_t i = 0; i < p->n_rec / p->stride + 2; ++i) p->samples.push_back(p->rec[std::min<uint64_t>(i * p->stride, p->n_rec)]);
  *out = p; *n_records = whole; *n_used = p->n;
  return MIC_OK;
}
int mic_text_format(const mic_text* p) { return p ? (p->fasta ? '>' : '@') : 0; }
int mic_text_offsets(const mic_text* p, const uint64_t** s, size_t* n, uint32_t* stride) { *s = p->samples.data(); *n = p->samples.size(); *stride = p->stride; return MIC_OK; }
int mic_text_to_slot(mic_engine* e, mic_text* p, uint64_t r0, uint64_t r1, size_t slot, size_t* n_bytes) {
  if (!whole_strides(p->n_rec, p->stride, r0, r1) || slot >= e->raw.size()) return MIC_E_INVALID;
  const size_t m = (size_t)(p->rec[r1] - p->rec[r0]);
  if (m > e->max_bytes) return MIC_E_INVALID;
  memcpy(e->raw[slot].data(), p->t + p->rec[r0], m);
  *n_bytes = m;
  return MIC_OK;
}
int mic_text_copy(mic_engine*, mic_text* p, uint64_t r0, uint64_t r1, void* dst, size_t cap, size_t* n_bytes) {
  if (!whole_strides(p->n_rec, p->stride, r0, r1)) return MIC_E_INVALID;
  const size_t m = (size_t)(p->rec[r1] - p->rec[r0]);
  if (m > cap) return MIC_E_INVALID;
  memcpy(dst, p->t + p->rec[r0], m); *n_bytes = m;
  return MIC_OK;
}
int mic_text_free(mic_engine*, mic_text* p) { delete p; return MIC_OK; }
}
