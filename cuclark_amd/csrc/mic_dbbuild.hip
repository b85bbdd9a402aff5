// mic_dbbuild.hip — GPU builder of the target-specific k-mer database (.sz/.ky/.lb), SURVEY.md §8f row N2.
//
// Replaces the reference's CPU build (CuCLARK_hh.hh:691-1329 makeSpecificTargetSets -> EHashtable::addElement
// HashTableStorage_hh.hh:483-523 -> SortAllHashTable hashTable_hh.hh:203-216 -> RemoveCommon
// HashTableStorage_hh.hh:241-292 -> Write hashTable_hh.hh:590-663), which needs ~146 GB of host RAM for the bacteria
// database (README.md:93), by sort-based passes in HBM.  Semantics restated (no --tsk, no centromere labels):
//   * every k-mer occurrence of every target sequence is canonicalised and credited to the target's label;
//   * a k-mer is kept iff ALL its occurrences carry the same label (multiplicity 1) and its occurrence count,
//     which saturates at 254 (lElement::AddToCount, dataType.hh:318-319), is > minCount (-t);
//   * bucket = c mod HTSIZE, key = c div HTSIZE (truncated to the key width), keys ascending inside a bucket,
//     bucket size must stay below 256 (hashTable_hh.hh:616-624).
// Target parsing reuses the read indexer + packer (same rules: header lines skipped, line breaks transparent, any
// non-ACGTU byte ends a run; CuCLARK_hh.hh:1135-1190).  The table is produced in `parts` passes over disjoint bucket
// ranges so that any database fits: each pass holds only the k-mers whose bucket falls in its range.
#include "mi_clark.h"
#include "mic_internal.h"
#include "mic_device.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include <hipcub/hipcub.hpp>

namespace {

thread_local char g_berr[512] = "";

struct PartDesc { uint32_t first; uint32_t len; uint32_t p0; };   // first data container, nucleotides, first emitted position

// one thread per k-mer position of the batch; positions are located by binary search in the per-part prefix sums
__global__ void emit_kmers_kernel(const uint16_t* __restrict__ cont, const PartDesc* __restrict__ parts,
                                  const unsigned long long* __restrict__ pos_prefix, uint32_t n_parts,
                                  unsigned long long n_pos, int k, uint32_t stride, MicDiv div, uint64_t rem_lo, uint64_t rem_hi,
                                  uint16_t label, unsigned long long* __restrict__ out_k, uint16_t* __restrict__ out_l,
                                  unsigned long long* __restrict__ cursor, unsigned long long cap) {
  unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool emit = false; uint64_t c = 0;
  if (t < n_pos) {
    uint32_t lo = 0, hi = n_parts;           // last part with pos_prefix[p] <= t
    while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (pos_prefix[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t pos = parts[lo].p0 + (uint32_t)(t - pos_prefix[lo]) * stride;   // stride 1: every k-mer; k*gap: light
    const uint32_t c0 = parts[lo].first + pos / 8;
    uint64_t h64 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) h64 = (h64 << 16) | cont[c0 + i];
    const uint32_t lo16 = cont[c0 + 4];
    const int s = 2 * (pos & 7);
    const uint64_t x = s ? ((h64 << s) | ((uint64_t)lo16 >> (16 - s))) : h64;
    c = canonical(x >> (64 - 2 * k), k);
    const uint64_t q = mic_div(c, div);
    const uint64_t rem = c - q * div.d;
    emit = rem >= rem_lo && rem < rem_hi;
  }
  // wave-aggregated append
  const unsigned long long m = __ballot(emit);
  if (m) {
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (emit) {
      const unsigned long long dst = base + __popcll(m & ((1ULL << lane) - 1));
      if (dst < cap) { out_k[dst] = c; out_l[dst] = label; }
    }
  }
}

// heads of equal-k-mer runs decide: kept iff one label over the whole run and min(len,254) > min_count
__global__ void decide_kernel(const unsigned long long* __restrict__ k, const uint16_t* __restrict__ l,
                              unsigned long long n, uint32_t min_count, uint8_t* __restrict__ keep) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t kp = 0;
  if (i == 0 || k[i] != k[i - 1]) {
    const unsigned long long key = k[i]; const uint16_t lab = l[i];
    unsigned long long j = i + 1; bool same = true;
    while (j < n && k[j] == key) { same = same && l[j] == lab; ++j; }
    const unsigned long long len = j - i;
    const uint32_t count = len > 254 ? 254u : (uint32_t)len;   // Count starts at 1 and stops growing at 254
    kp = (same && count > min_count) ? 1 : 0;
  }
  keep[i] = kp;
}

__global__ void rem_kernel(const unsigned long long* __restrict__ c, unsigned long long n, MicDiv div, uint64_t rem_lo,
                           uint32_t* __restrict__ rem, uint32_t* __restrict__ idx) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t q = mic_div(c[i], div);
  rem[i] = (uint32_t)(c[i] - q * div.d - rem_lo);
  idx[i] = (uint32_t)i;
}

template <typename KEY>
__global__ void finish_kernel(const unsigned long long* __restrict__ c, const uint16_t* __restrict__ lab,
                              const uint32_t* __restrict__ rem_sorted, const uint32_t* __restrict__ idx_sorted,
                              unsigned long long n, MicDiv div, KEY* __restrict__ keys, uint16_t* __restrict__ labels,
                              uint32_t* __restrict__ sizes, uint32_t* __restrict__ too_big) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = idx_sorted[i];
  keys[i] = (KEY)mic_div(c[j], div);
  labels[i] = lab[j];
  const uint32_t old = atomicAdd(&sizes[rem_sorted[i]], 1u);
  if (old + 1 >= 256) atomicMax(too_big, old + 1);
}

__global__ void sizes_u8_kernel(const uint32_t* __restrict__ s32, uint8_t* __restrict__ s8, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) s8[i] = (uint8_t)s32[i];
}

struct PackedTarget {
  std::vector<uint16_t> cont;
  std::vector<PartDesc> parts;
  std::vector<unsigned long long> prefix;  // n_parts + 1
  uint16_t label;
};

int bfail(int code, const char* fmt, const char* a = "", const char* b = "") {
  snprintf(g_berr, sizeof(g_berr), fmt, a, b);
  return code;
}

// index + pack one target file with the read machinery; every record becomes parts of >= k nucleotides
int pack_target(const char* path, int k, int threads, uint32_t light_gap, PackedTarget& out) {
  int fd = open(path, O_RDONLY);
  struct stat st;
  if (fd == -1 || fstat(fd, &st) != 0) { if (fd != -1) close(fd); return bfail(MIC_E_IO, "Failed to open %s", path); }
  if (st.st_size == 0) { close(fd); out.prefix.assign(1, 0); return MIC_OK; }
  const uint8_t* map = (const uint8_t*)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (map == MAP_FAILED) { close(fd); return bfail(MIC_E_IO, "Failed to map %s", path); }
  const size_t nb = (size_t)st.st_size;
  int rc = MIC_OK;
  std::vector<uint64_t> run_off, run_len;   // per part: offset in its ACGTU run, length of the run
  if (map[0] != '>' && map[0] != '@') {
    rc = bfail(MIC_E_INVALID, "%s: targets must be FASTA or FASTQ (k-mer spectrum targets are not supported)", path);
  } else {
    size_t cap = nb / 64 + 1024;
    std::vector<uint64_t> ns, ne, ss, se, ln;
    long n;
    for (;;) {
      ns.resize(cap); ne.resize(cap); ss.resize(cap); se.resize(cap); ln.resize(cap);
      n = mic_index_reads_parallel(map, nb, threads, cap, ns.data(), ne.data(), ss.data(), se.data(), ln.data());
      if (n < 0 || (size_t)n <= cap) break;
      cap = (size_t)n;
    }
    if (n < 0) rc = bfail(MIC_E_INVALID, "%s: unrecognised format", path);
    else {
      size_t bound = mic_pack_bound(ss.data(), se.data(), (size_t)n, k);
      if (bound > 0xFFFFFFF0ull) rc = bfail(MIC_E_INVALID, "%s: target too large for one batch", path);
      else {
        std::vector<uint32_t> rp((size_t)n + 1);
        out.cont.resize(bound + 8);
        size_t m = mic_pack_reads_runs(map, ss.data(), se.data(), ln.data(), (size_t)n, k, rp.data(), out.cont.data(), bound,
                                       &run_off, &run_len);
        if (m == (size_t)-1) rc = bfail(MIC_E_INVALID, "%s: packing failed", path);
        else {
          out.cont.resize(m + 8);
          for (size_t p = 0; p < m;) {  // walk the parts
            uint32_t len = out.cont[p];
            out.parts.push_back({(uint32_t)(p + 1), len, 0u});
            p += 1 + (len + 7) / 8;
          }
        }
      }
    }
  }
  munmap((void*)map, nb);
  close(fd);
  out.prefix.assign(out.parts.size() + 1, 0);
  if (!light_gap) {
    for (size_t i = 0; i < out.parts.size(); ++i)
      out.prefix[i + 1] = out.prefix[i] + (out.parts[i].len >= (uint32_t)k ? out.parts[i].len - k + 1 : 0);
    return rc;
  }
  // Light database (CuCLARK_hh.hh:705-735, 780-797): a run is cut into consecutive blocks of k nucleotides, the blocks
  // of the whole FILE are numbered in order (`iter`), and block number i is used iff i % gap == 0.  A run of L
  // nucleotides completes L / k blocks; runs shorter than k complete none and were not packed.  Sub-parts of a long
  // run overlap by k-1, so every block lies in exactly one of them.
  if (rc == MIC_OK && run_off.size() != out.parts.size()) rc = bfail(MIC_E_INVALID, "%s: part bookkeeping mismatch", path);
  if (rc != MIC_OK) return rc;
  uint64_t done = 0, cur_blocks = 0;   // blocks completed before the current run; blocks of the current run
  for (size_t i = 0; i < out.parts.size(); ++i) {
    const uint64_t off = run_off[i], L = run_len[i], plen = out.parts[i].len, nblocks = L / (uint64_t)k;
    if (off == 0) { done += cur_blocks; cur_blocks = nblocks; }
    uint64_t cnt = 0, b = 0;
    if (plen >= (uint64_t)k && nblocks) {
      const uint64_t b_lo = (off + (uint64_t)k - 1) / (uint64_t)k;
      uint64_t b_hi = (off + plen - (uint64_t)k) / (uint64_t)k;
      if (b_hi > nblocks - 1) b_hi = nblocks - 1;
      b = b_lo + (light_gap - (done + b_lo) % light_gap) % light_gap;
      if (b <= b_hi) cnt = (b_hi - b) / light_gap + 1;
    }
    out.parts[i].p0 = cnt ? (uint32_t)(b * (uint64_t)k - off) : 0u;
    out.prefix[i + 1] = out.prefix[i] + cnt;
  }
  return rc;
}

#define BHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    snprintf(g_berr, sizeof(g_berr), "%s: %s", #x, hipGetErrorString(e_)); rc = e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP; goto done; } } while (0)

template <typename T>
int append_file(FILE* f, const T* d_ptr, size_t n, std::vector<char>& host) {
  if (!n) return MIC_OK;
  host.resize(n * sizeof(T));
  if (hipMemcpy(host.data(), d_ptr, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return MIC_E_HIP;
  return fwrite(host.data(), sizeof(T), n, f) == n ? MIC_OK : MIC_E_IO;
}

}  // namespace

extern "C" {

const char* mic_db_build_error(void) { return g_berr; }

int mic_db_build(const char* const* target_files, const uint16_t* target_labels, size_t n_files, int k, uint64_t htsize,
                 int key_bytes, uint32_t min_count, uint32_t light_gap, const char* out_prefix, int device, int threads,
                 uint32_t parts, uint64_t* n_kmers_out) {
  if (!target_files || !target_labels || !out_prefix || k < 2 || k > 32 || htsize < 2 || htsize > 0xFFFFFFF0ull)
    return bfail(MIC_E_INVALID, "bad argument");
  if (key_bytes == 0) key_bytes = mic_key_bytes_rule(htsize, k);
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return bfail(MIC_E_INVALID, "bad key width");
  if (threads < 1) threads = 1;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return bfail(MIC_E_NODEVICE, "cannot select the device");

  // ---- targets -> packed 2-bit parts in host memory (2 bits per nucleotide)
  std::vector<PackedTarget> tg(n_files);
  unsigned long long total_pos = 0, max_pos = 0; size_t max_cont = 0, max_parts = 0;
  for (size_t f = 0; f < n_files; ++f) {
    tg[f].label = target_labels[f];
    int rc0 = pack_target(target_files[f], k, threads, light_gap, tg[f]);
    if (rc0 != MIC_OK) return rc0;
    total_pos += tg[f].prefix.back();
    if (tg[f].prefix.back() > max_pos) max_pos = tg[f].prefix.back();
    if (tg[f].cont.size() > max_cont) max_cont = tg[f].cont.size();
    if (tg[f].parts.size() > max_parts) max_parts = tg[f].parts.size();
  }

  int rc = MIC_OK;
  const MicDiv div = mic_make_div(htsize);
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  // per element: k-mers and labels double-buffered for the radix sort, flags, kept copies, (rem, idx) double-buffered
  const unsigned long long per_elem = 2 * 8 + 2 * 2 + 1 + 8 + 2 + 4 * 4 + 8 + 2;
  unsigned long long cap = (unsigned long long)(free_b * 0.80) / per_elem;
  if (cap > 0x7FFF0000ull) cap = 0x7FFF0000ull;
  if (parts == 0) {
    parts = (uint32_t)((total_pos + total_pos / 8) / (cap ? cap : 1)) + 1;   // buckets are uniform: 12 % slack
  }
  const unsigned long long want = total_pos / parts + total_pos / parts / 8 + (1u << 16);
  if (want < cap) cap = want;
  const uint64_t buckets_per_part = (htsize + parts - 1) / parts;

  unsigned long long *d_k[2] = {nullptr, nullptr}, *d_kept = nullptr, *d_cursor = nullptr, *d_prefix = nullptr, *d_nsel = nullptr;
  uint16_t *d_l[2] = {nullptr, nullptr}, *d_lkept = nullptr, *d_cont = nullptr, *d_lout = nullptr;
  uint8_t *d_keep = nullptr, *d_s8 = nullptr;
  uint32_t *d_rem[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr}, *d_sizes = nullptr, *d_big = nullptr;
  PartDesc* d_parts = nullptr;
  void *d_temp = nullptr, *d_kout = nullptr;
  size_t temp_bytes = 0, tb = 0;
  FILE *fs = nullptr, *fk = nullptr, *fl = nullptr;
  std::vector<char> host;
  unsigned long long n_total = 0;
  const std::string pfx(out_prefix);

  BHIP(hipMalloc(&d_k[0], cap * 8)); BHIP(hipMalloc(&d_k[1], cap * 8));
  BHIP(hipMalloc(&d_l[0], cap * 2)); BHIP(hipMalloc(&d_l[1], cap * 2));
  BHIP(hipMalloc(&d_keep, cap)); BHIP(hipMalloc(&d_kept, cap * 8)); BHIP(hipMalloc(&d_lkept, cap * 2));
  BHIP(hipMalloc(&d_rem[0], cap * 4)); BHIP(hipMalloc(&d_rem[1], cap * 4));
  BHIP(hipMalloc(&d_idx[0], cap * 4)); BHIP(hipMalloc(&d_idx[1], cap * 4));
  BHIP(hipMalloc(&d_kout, cap * 8)); BHIP(hipMalloc(&d_lout, cap * 2));
  BHIP(hipMalloc(&d_sizes, buckets_per_part * 4)); BHIP(hipMalloc(&d_s8, buckets_per_part));
  BHIP(hipMalloc(&d_cursor, 8)); BHIP(hipMalloc(&d_nsel, 8)); BHIP(hipMalloc(&d_big, 4));
  BHIP(hipMalloc(&d_cont, (max_cont + 16) * 2)); BHIP(hipMalloc(&d_parts, (max_parts + 1) * sizeof(PartDesc)));
  BHIP(hipMalloc(&d_prefix, (max_parts + 2) * 8));
  // temp storage: the largest of the three library calls
  hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_k[0], d_k[1], d_l[0], d_l[1], (int)cap, 0, 2 * k);
  temp_bytes = tb;
  hipcub::DeviceSelect::Flagged(nullptr, tb, d_k[0], d_keep, d_kept, d_nsel, (int)cap);
  if (tb > temp_bytes) temp_bytes = tb;
  hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_rem[0], d_rem[1], d_idx[0], d_idx[1], (int)cap, 0, 32);
  if (tb > temp_bytes) temp_bytes = tb;
  BHIP(hipMalloc(&d_temp, temp_bytes + 256));

  // the files are written under temporary names and renamed once all three are complete: a build that is interrupted
  // between passes must not leave a short .sz behind that a later run would take for a (smaller) table
  fs = fopen((pfx + ".sz.tmp").c_str(), "wb"); fk = fopen((pfx + ".ky.tmp").c_str(), "wb"); fl = fopen((pfx + ".lb.tmp").c_str(), "wb");
  if (!fs || !fk || !fl) { rc = bfail(MIC_E_IO, "cannot create %s.{sz,ky,lb}", out_prefix); goto done; }

  for (uint32_t p = 0; p < parts && rc == MIC_OK; ++p) {
    const uint64_t rem_lo = (uint64_t)p * buckets_per_part;
    const uint64_t rem_hi = rem_lo + buckets_per_part < htsize ? rem_lo + buckets_per_part : htsize;
    if (rem_lo >= htsize) break;
    const uint64_t nbk = rem_hi - rem_lo;
    BHIP(hipMemset(d_cursor, 0, 8));
    // 1. emit the canonical k-mers of every target whose bucket lies in this pass
    for (size_t f = 0; f < n_files; ++f) {
      const PackedTarget& t = tg[f];
      const unsigned long long npos = t.prefix.back();
      if (!npos) continue;
      BHIP(hipMemcpy(d_cont, t.cont.data(), t.cont.size() * 2, hipMemcpyHostToDevice));
      BHIP(hipMemcpy(d_parts, t.parts.data(), t.parts.size() * sizeof(PartDesc), hipMemcpyHostToDevice));
      BHIP(hipMemcpy(d_prefix, t.prefix.data(), t.prefix.size() * 8, hipMemcpyHostToDevice));
      emit_kmers_kernel<<<(unsigned)((npos + 255) / 256), 256>>>(d_cont, d_parts, d_prefix, (uint32_t)t.parts.size(), npos, k, light_gap ? (uint32_t)k * light_gap : 1u, div,
                                                                 rem_lo, rem_hi, t.label, d_k[0], d_l[0], d_cursor, cap);
      BHIP(hipGetLastError());
    }
    unsigned long long n = 0;
    BHIP(hipMemcpy(&n, d_cursor, 8, hipMemcpyDeviceToHost));
    if (n > cap) { rc = bfail(MIC_E_NOMEM, "a pass of the database build overflowed its buffer; use more passes"); break; }
    unsigned long long nsel = 0;
    if (n) {
      // 2. sort by k-mer (stable: occurrences of a k-mer stay in target order), decide, compact in order
      tb = temp_bytes;
      BHIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_k[0], d_k[1], d_l[0], d_l[1], (int)n, 0, 2 * k));
      decide_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d_k[1], d_l[1], n, min_count, d_keep);
      BHIP(hipGetLastError());
      tb = temp_bytes;
      BHIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_k[1], d_keep, d_kept, d_nsel, (int)n));
      tb = temp_bytes;
      BHIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_l[1], d_keep, d_lkept, d_nsel, (int)n));
      BHIP(hipMemcpy(&nsel, d_nsel, 8, hipMemcpyDeviceToHost));
      nsel &= 0xFFFFFFFFull;  // num_selected is written as an int
    }
    BHIP(hipMemset(d_sizes, 0, nbk * 4));
    BHIP(hipMemset(d_big, 0, 4));
    if (nsel) {
      // 3. stable sort by bucket: inside a bucket the k-mers stay ascending, hence the quotients too
      rem_kernel<<<(unsigned)((nsel + 255) / 256), 256>>>(d_kept, nsel, div, rem_lo, d_rem[0], d_idx[0]);
      BHIP(hipGetLastError());
      tb = temp_bytes;
      BHIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_rem[0], d_rem[1], d_idx[0], d_idx[1], (int)nsel, 0, 32));
      const unsigned blocks = (unsigned)((nsel + 255) / 256);
      if (key_bytes == 2) finish_kernel<uint16_t><<<blocks, 256>>>(d_kept, d_lkept, d_rem[1], d_idx[1], nsel, div, (uint16_t*)d_kout, d_lout, d_sizes, d_big);
      else if (key_bytes == 4) finish_kernel<uint32_t><<<blocks, 256>>>(d_kept, d_lkept, d_rem[1], d_idx[1], nsel, div, (uint32_t*)d_kout, d_lout, d_sizes, d_big);
      else finish_kernel<uint64_t><<<blocks, 256>>>(d_kept, d_lkept, d_rem[1], d_idx[1], nsel, div, (uint64_t*)d_kout, d_lout, d_sizes, d_big);
      BHIP(hipGetLastError());
      uint32_t big = 0;
      BHIP(hipMemcpy(&big, d_big, 4, hipMemcpyDeviceToHost));
      if (big) {
        rc = bfail(MIC_E_INVALID, "This table can not be stored on disk: Some bucket list size exceeds 255. Choose a smaller k-mer "
                                  "length and/or a larger hash table.");
        break;
      }
    }
    sizes_u8_kernel<<<(unsigned)((nbk + 255) / 256), 256>>>(d_sizes, d_s8, nbk);
    BHIP(hipGetLastError());
    // 4. append this bucket range to the three files
    if ((rc = append_file(fs, d_s8, (size_t)nbk, host)) != MIC_OK) { bfail(rc, "write to %s.sz failed", out_prefix); break; }
    if (key_bytes == 2) rc = append_file(fk, (const uint16_t*)d_kout, (size_t)nsel, host);
    else if (key_bytes == 4) rc = append_file(fk, (const uint32_t*)d_kout, (size_t)nsel, host);
    else rc = append_file(fk, (const uint64_t*)d_kout, (size_t)nsel, host);
    if (rc == MIC_OK) rc = append_file(fl, d_lout, (size_t)nsel, host);
    if (rc != MIC_OK) { bfail(rc, "write to %s.{ky,lb} failed", out_prefix); break; }
    n_total += nsel;
  }
  if (n_kmers_out) *n_kmers_out = n_total;
done:
  if (fs && fclose(fs) != 0 && rc == MIC_OK) rc = bfail(MIC_E_IO, "write to %s.sz failed", out_prefix);
  if (fk && fclose(fk) != 0 && rc == MIC_OK) rc = bfail(MIC_E_IO, "write to %s.ky failed", out_prefix);
  if (fl && fclose(fl) != 0 && rc == MIC_OK) rc = bfail(MIC_E_IO, "write to %s.lb failed", out_prefix);
  if (rc == MIC_OK) {
    for (const char* ext : {".ky", ".lb", ".sz"})      // .sz last: the classifier looks for it first
      if (rename((pfx + ext + ".tmp").c_str(), (pfx + ext).c_str()) != 0 && rc == MIC_OK) rc = bfail(MIC_E_IO, "cannot rename %s%s.tmp", out_prefix, ext);
  }
  if (rc != MIC_OK)
    for (const char* ext : {".sz", ".ky", ".lb"}) { remove((pfx + ext + ".tmp").c_str()); remove((pfx + ext).c_str()); }
  void* frees[] = {d_k[0], d_k[1], d_l[0], d_l[1], d_keep, d_kept, d_lkept, d_rem[0], d_rem[1], d_idx[0], d_idx[1], d_kout, d_lout,
                   d_sizes, d_s8, d_cursor, d_nsel, d_big, d_cont, d_parts, d_prefix, d_temp};
  for (void* q : frees) if (q) hipFree(q);
  return rc;
}

}  // extern "C"
