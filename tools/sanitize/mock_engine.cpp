// mock_engine.cpp - a stand-in for libmi_clark.so's DEVICE entry points, for sanitizer builds of the host code only
// (tests/test_sanitizers.py: AddressSanitizer + UBSan, ThreadSanitizer; there is no GPU sanitizer on this pool).
// exe-side code under test, unchanged: classifier.cpp (FileFeeder, SegmentFeeder, PairedFileFeeder, PairedSource, GzSource,
// InflateStream, strip_fastq, run_stream's loader / device / writer threads), cli_main.cpp, and mic_host.cpp (indexer, packer,
// CSV).  What is mocked: the engine.  mic_ingest_classify here "classifies" a slot on the CPU: it walks the records of the
// slot's bytes and writes one CSV line "<name>,<length>" per record - a pure function of the input, so the test can check
// that every record of the input arrives exactly once and in order through the threaded pipeline.
// NEVER linked into the product.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

// the batch API is only reached when a slot is handed back (MIC_INGEST_FALLBACK): the mock never does
int mic_batches_alloc(mic_engine*, size_t, size_t, size_t, const uint32_t*, int, uint32_t**, uint32_t**, uint32_t**, uint16_t**) { return MIC_E_NODEVICE; }
int mic_batch_ready(mic_engine*, size_t, size_t, size_t) { return MIC_E_NODEVICE; }
int mic_batch_query(mic_engine*, size_t, int, int) { return MIC_E_NODEVICE; }
int mic_batch_wait(mic_engine*, size_t) { return MIC_E_NODEVICE; }
int mic_batch_dense_counts(mic_engine*, size_t, size_t, uint32_t*) { return MIC_E_NODEVICE; }
int mic_batch_merge_shards(mic_engine* const*, size_t, size_t) { return MIC_E_NODEVICE; }
int mic_batches_free(mic_engine*) { return MIC_OK; }
// compressed mates on the device: the mock has none, the command line inflates on the host
int mic_gz_inflate_device(mic_engine*, const void*, size_t, void**, size_t*, uint32_t*) { return MIC_E_UNSUPPORTED; }
int mic_gz_free_text(mic_engine*, void*) { return MIC_OK; }
int mic_gz_reserve(mic_engine*, size_t, uint32_t) { return MIC_E_NODEVICE; }
uint64_t mic_gz_reserve_bytes(size_t, uint32_t) { return 0; }
int mic_gz_release(mic_engine*) { return MIC_OK; }
int mic_pairs_index_device(mic_engine*, const void*, size_t, const void*, size_t, mic_pairs**, uint64_t*, uint32_t*) { return MIC_E_NODEVICE; }
int mic_pairs_offsets(const mic_pairs*, const uint64_t**, size_t*, uint32_t*) { return MIC_E_NODEVICE; }
int mic_pairs_merge_to_slot(mic_engine*, mic_pairs*, uint64_t, uint64_t, size_t, size_t*) { return MIC_E_NODEVICE; }
int mic_pairs_text(mic_engine*, mic_pairs*, uint64_t, uint64_t, void*, size_t, size_t*) { return MIC_E_NODEVICE; }
int mic_pairs_free(mic_engine*, mic_pairs*) { return MIC_OK; }
int mic_text_index_device(mic_engine*, const void*, size_t, mic_text**, uint64_t*, uint32_t*) { return MIC_E_NODEVICE; }
int mic_text_offsets(const mic_text*, const uint64_t**, size_t*, uint32_t*) { return MIC_E_NODEVICE; }
int mic_text_to_slot(mic_engine*, mic_text*, uint64_t, uint64_t, size_t, size_t*) { return MIC_E_NODEVICE; }
int mic_text_copy(mic_engine*, mic_text*, uint64_t, uint64_t, void*, size_t, size_t*) { return MIC_E_NODEVICE; }
int mic_text_free(mic_engine*, mic_text*) { return MIC_OK; }
int mic_text_format(const mic_text*) { return 0; }
}
