// mic_ingest.hip — device-side ingest: raw FASTA / FASTQ bytes in, result-CSV text out (include/mi_clark.h, mic_ingest_*).
//
// Moves onto the GPU what the reference does on the host around queryBatch, for well-formed input:
//   read indexing            CuCLARK_hh.hh:1339-1534   -> line_count / line_start / record kernels
//   2-bit packing, N-split   CuCLARK_hh.hh:1616-1716   -> pack_kernel  (same container format, same part rules)
//   result CSV lines         CuCLARK_hh.hh:1951-2139   -> csv_len / csv_fmt kernels ("%g" exact, mic_fmt.h)
// The host only moves bytes: file -> pinned buffer -> (H2D) ... (D2H) -> pinned buffer -> file.
//
// A batch is a run of whole records.  Anything the kernels do not reproduce exactly is DETECTED and reported as
// MIC_INGEST_FALLBACK so that the caller sends that batch through the host indexer / packer instead (mic_index_reads,
// mic_pack_reads, mic_csv_line): an empty read name (the reference's name scan then crosses the line end), a FASTA
// record without a sequence line, a truncated FASTQ record, a sequence longer than MIC_MAX_PART bytes, more lines or
// reads than the slot was sized for, a read that needs the dense fallback of the query kernel.
#include "mi_clark.h"
#include "mic_internal.h"
#include "mic_fmt.h"

#include <hipcub/hipcub.hpp>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <string>
#include <thread>
#include <vector>

struct mic_engine;
// engine internals this file needs (mic_engine.hip)
int mic_engine_table(mic_engine* e, MicTable* t, int* slot_class, int* n_cu, int* device, int* k, uint32_t* n_targets);
int mic_set_error(int code, const char* fmt, ...);
int mic_bind_thread_near_device(int device, int on);
void mic_engine_copy_streams(mic_engine* e, hipStream_t* up, hipStream_t* down);

namespace {

enum { H_NLINES = 0, H_NREADS = 1, H_STATUS = 2, H_CSV_BYTES = 3, H_FLAGGED = 4, H_NEWLINES = 5, H_CONT = 6, H_WORDS = 8 };

#define ING_TILE 4096          // bytes per block of the line kernels (256 threads x 16 B)

__device__ __forceinline__ uint32_t newline_mask(uint32_t x) {       // 0x80 in every byte of x that is '\n'
  const uint32_t y = x ^ 0x0A0A0A0Au;
  return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// newline flags of the 16 bytes at byte offset `pos` (16-byte aligned): bit i = byte i is '\n' and lies below nb
__device__ __forceinline__ uint32_t tile_flags(const uint8_t* __restrict__ raw, uint32_t pos, uint32_t nb) {
  if (pos >= nb) return 0;
  const uint4 v = *(const uint4*)(raw + pos);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t f = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t m = newline_mask(w[i]);     // bits 7, 15, 23, 31
    f |= (((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u)) << (4 * i);
  }
  const uint32_t left = nb - pos;
  if (left < 16) f &= (1u << left) - 1u;
  return f;
}

__global__ void __launch_bounds__(256) line_count_kernel(const uint8_t* __restrict__ raw, uint32_t nb, uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t s_w[4];
  const uint32_t pos = blockIdx.x * ING_TILE + threadIdx.x * 16;
  uint32_t c = __popc(tile_flags(raw, pos, nb));
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// line_start[L] = offset of the first byte of line L (line 0 starts at 0; a '\n' at p starts the next line at p + 1)
__global__ void __launch_bounds__(256) line_start_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ tile_off,
                                                         uint32_t* __restrict__ line_start, uint32_t cap) {
  __shared__ uint32_t s_w[4];
  const uint32_t pos = blockIdx.x * ING_TILE + threadIdx.x * 16;
  uint32_t f = tile_flags(raw, pos, nb);
  const uint32_t c = __popc(f);
  uint32_t inc = c;                               // inclusive scan inside the wave
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) s_w[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t base = tile_off[blockIdx.x];
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_w[w];
  uint32_t rank = base + inc - c;                 // newlines before this thread's bytes
  if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
  while (f) {
    const int b = __ffs((int)f) - 1;
    f &= f - 1;
    ++rank;
    if (rank < cap) line_start[rank] = pos + (uint32_t)b + 1u;
  }
}

// one thread: number of lines (an unterminated last line counts and gets a virtual line end at nb), FASTQ record count
__global__ void lines_finish_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ tile_off, uint32_t n_tiles,
                                    uint32_t* __restrict__ line_start, uint32_t cap, int fasta, uint32_t lpr, uint32_t max_reads,
                                    uint32_t* __restrict__ hdr) {
  const uint32_t nl = tile_off[n_tiles];
  uint32_t n_lines = nl;
  if (nb && raw[nb - 1] != '\n') { ++n_lines; if (n_lines < cap) line_start[n_lines] = nb + 1; }
  uint32_t status = 0;
  if (n_lines + 1 >= cap) status |= MIC_INGEST_TOO_MANY;
  hdr[H_NEWLINES] = nl;
  hdr[H_NLINES] = n_lines;
  if (!fasta) {
    if (n_lines % lpr) status |= MIC_INGEST_TRUNCATED;
    const uint32_t nr = n_lines / lpr;
    if (nr > max_reads) status |= MIC_INGEST_TOO_MANY;
    hdr[H_NREADS] = nr;
  }
  if (status) atomicOr(&hdr[H_STATUS], status);
}

// FASTA: flag[L] = line L starts a record ('>' in its first column, CuCLARK_hh.hh:1369-1389)
__global__ void __launch_bounds__(256) fasta_flag_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ line_start,
                                                         const uint32_t* __restrict__ hdr, uint32_t cap, uint32_t n_scan,
                                                         uint32_t* __restrict__ flag) {
  const uint32_t L = blockIdx.x * 256 + threadIdx.x;
  if (L >= n_scan) return;
  const uint32_t n_lines = hdr[H_NLINES];
  uint32_t f = 0;
  if (L < n_lines && n_lines + 1 < cap) { const uint32_t p = line_start[L]; f = p < nb && raw[p] == '>'; }
  flag[L] = f;
}

__global__ void __launch_bounds__(256) fasta_scatter_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rec_of_line,
                                                            uint32_t cap, uint32_t max_reads, uint32_t* __restrict__ hdr_line, uint32_t* __restrict__ hdr) {
  const uint32_t L = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n_lines = hdr[H_NLINES];
  if (n_lines + 1 >= cap) return;
  if (L < n_lines && flag[L]) { const uint32_t r = rec_of_line[L]; if (r < max_reads) hdr_line[r] = L; }
  if (L == n_lines) {                             // rec_of_line[n_lines] = number of records (flags past n_lines are 0)
    const uint32_t nr = rec_of_line[L];
    if (nr > max_reads) atomicOr(&hdr[H_STATUS], (uint32_t)MIC_INGEST_TOO_MANY);
    else hdr_line[nr] = n_lines;
    hdr[H_NREADS] = nr;
  }
}

struct RecArrays {
  uint32_t* name_s; uint32_t* seq_s; uint32_t* seq_e; uint32_t* length; uint32_t* bound;   // bound: containers reserved (scan input)
  uint8_t* name_len;                                                                       // min(name length, 40)
};

__device__ __forceinline__ bool name_sep_dev(uint8_t c) { return c == ' ' || c == '\t' || c == '\n'; }   // CuCLARK_hh.hh:300

// One thread per record: where its name and sequence are, its Length column, how many containers to reserve.
// FASTQ: record r = lines 4r .. 4r+3 (CuCLARK_hh.hh:1496-1523); FASTA: header line hdr_line[r], sequence lines up to the
// next header (CuCLARK_hh.hh:1369-1389), Length = bytes of the sequence lines without their line ends.
template <bool FASTA>
__global__ void __launch_bounds__(256) record_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ line_start,
                                                     const uint32_t* __restrict__ hdr_line, uint32_t n_reads, uint32_t lpr, int k, RecArrays a,
                                                     uint32_t* __restrict__ hdr) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r > n_reads) return;
  if (r == n_reads) { a.bound[r] = 0; return; }
  uint32_t hl, hn;
  if (FASTA) { hl = hdr_line[r]; hn = hdr_line[r + 1]; } else { hl = lpr * r; hn = lpr * r + 2; }
  const uint32_t hs = line_start[hl], ss = line_start[hl + 1], se = line_start[hn] - 1;
  uint32_t status = 0, len;
  if (FASTA) {
    const uint32_t c = hn - hl - 1;               // sequence lines
    if (c == 0) status |= MIC_INGEST_ODD_RECORD;  // the reference counts a phantom byte for such a record: host path
    len = (se + 1 - ss) - c;
  } else {
    len = se - ss;
  }
  // name: from the byte after the marker to the first separator strictly after it (CuCLARK_hh.hh:1369-1372)
  const uint32_t ns = hs + 1;
  if (ns + 1 >= ss) status |= MIC_INGEST_ODD_RECORD;   // header line holds the marker only: the reference's scan leaves the line
  uint32_t j = ns + 1;
  while (j < ss && j - ns < 40 && !name_sep_dev(raw[j])) ++j;
  const uint32_t nbytes = se >= ss ? se - ss : 0;
  if (nbytes > MIC_MAX_PART) status |= MIC_INGEST_LONG_READ;
  a.name_s[r] = ns; a.name_len[r] = (uint8_t)(j - ns);
  a.seq_s[r] = ss; a.seq_e[r] = ss + nbytes; a.length[r] = len;
  // containers reserved for the read: every part of L >= k nt takes 1 + ceil(L/8) <= 2 + L/8, parts are separated by at
  // least one byte; + slack for a trailing run that is dropped after its first containers were written, + the terminator
  a.bound[r] = (len < (uint32_t)k || status) ? 0u : nbytes / 8 + 2 * (nbytes / (uint32_t)(k + 1) + 1) + 8;
  if (status) atomicOr(&hdr[H_STATUS], status);
  (void)nb;
}

// ---- the packer (CuCLARK_hh.hh:1616-1716): one wavefront per read ---------------------------------------------------
// A part is a maximal run of ACGTU (either case); '\n' is transparent, any other byte ends the run; runs shorter than k
// are dropped.  Stored per part: one length slot + ceil(len/8) containers, 8 nt per u16, first nt in the top bits,
// A=3 C=2 G=1 T/U=0, the last container left-aligned.  Reads are laid out at the reserved offsets rp[r] and end with a 0
// length slot when they do not fill their reservation (the query kernel stops there, include/mi_clark.h).
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ raw, const uint32_t* __restrict__ seq_s,
                                                   const uint32_t* __restrict__ seq_e, const uint32_t* __restrict__ rp,
                                                   uint16_t* __restrict__ cont, uint32_t n_reads, int k) {
  __shared__ uint8_t s_codes[4][80];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* codes = s_codes[wv];
  const uint32_t n_waves = gridDim.x * 4;
  for (uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv); r < n_reads; r += n_waves) {
    const uint32_t o0 = rp[r], o1 = rp[r + 1];
    if (o1 == o0) continue;                       // shorter than k: nothing stored (CuCLARK_hh.hh:1633)
    uint16_t* out = cont + o0;
    const uint32_t s = seq_s[r], e = seq_e[r];
    uint32_t hdr = 0, run = 0;                    // header slot of the open part (relative), its nucleotides so far
    auto close_run = [&]() {
      if (run >= (uint32_t)k) {
        const uint32_t rem = run & 7u;
        if (rem && lane == 0) {
          uint32_t v = 0;
          for (uint32_t i = 0; i < rem; ++i) v = (v << 2) | codes[i];
          out[hdr + 1 + run / 8] = (uint16_t)(v << (2 * (8 - rem)));
        }
        if (lane == 0) out[hdr] = (uint16_t)run;
        hdr += 1 + (run + 7) / 8;
      }
      run = 0;
    };
    for (uint32_t base = s; base < e; base += 64) {
      const uint32_t p = base + lane;
      uint32_t cls = 3, code = 0;                 // 0 nucleotide, 1 line end, 2 other byte, 3 past the end
      if (p < e) {
        const uint32_t b = raw[p], u = b & 0xDFu;
        if (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U') { cls = 0; code = (0x4Bu >> (2 * ((u >> 1) & 3u))) & 3u; }
        else cls = b == '\n' ? 1 : 2;
      }
      const uint64_t m_nt = __ballot(cls == 0), m_ot = __ballot(cls == 2);
      int lo = 0;
      for (;;) {
        const uint64_t from = lo >= 64 ? 0ull : ~0ull << lo;
        const uint64_t mo = m_ot & from;
        const int hi = mo ? __builtin_ctzll(mo) : 64;
        const uint64_t upto = hi >= 64 ? ~0ull : ((1ull << hi) - 1);
        const uint64_t seg = m_nt & from & upto;
        const uint32_t n = __popcll(seg);
        if (n) {
          const uint32_t off = run & 7u;
          __builtin_amdgcn_wave_barrier();
          if ((seg >> lane) & 1) codes[off + __popcll(seg & ((1ull << lane) - 1))] = (uint8_t)code;
          __builtin_amdgcn_wave_barrier();
          const uint32_t total = off + n, nfull = total >> 3, rem = total & 7u;
          uint32_t keep = 0;
          if ((uint32_t)lane < nfull) {
            const uint8_t* c = codes + 8 * lane;
            const uint32_t v = (c[0] << 14) | (c[1] << 12) | (c[2] << 10) | (c[3] << 8) | (c[4] << 6) | (c[5] << 4) | (c[6] << 2) | c[7];
            out[hdr + 1 + (run - off) / 8 + lane] = (uint16_t)v;
          }
          if ((uint32_t)lane < rem) keep = codes[8 * nfull + lane];
          __builtin_amdgcn_wave_barrier();
          if ((uint32_t)lane < rem) codes[lane] = (uint8_t)keep;
          __builtin_amdgcn_wave_barrier();
          run += n;
        }
        if (hi >= 64) break;
        close_run();
        lo = hi + 1;
      }
    }
    close_run();
    if (lane == 0 && hdr < o1 - o0) out[hdr] = 0;
  }
}

// ---- CSV (CuCLARK_hh.hh:1951-2139, non-extended): "name,len,gamma,1st,score1,2nd,score2,confidence\n" ----------------
struct CsvArgs {
  const uint8_t* raw; const uint32_t* name_s; const uint8_t* name_len; const uint32_t* length; const uint32_t* results;
  const char* tnames; const uint32_t* tname_off;   // target names back to back; name t = [off[t], off[t+1])
  uint32_t n_targets, n_reads; int k, paired;
};

__device__ __forceinline__ uint32_t digits_u32(uint32_t v) {
  return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : v < 1000000 ? 6 : v < 10000000 ? 7 : v < 100000000 ? 8 : v < 1000000000 ? 9 : 10;
}

__device__ __forceinline__ uint32_t tname_len(const CsvArgs& a, uint32_t idx1) {     // idx1 = target + 1, 0 or out of range = "NA"
  return (idx1 == 0 || idx1 > a.n_targets) ? 2u : a.tname_off[idx1] - a.tname_off[idx1 - 1];
}

__global__ void __launch_bounds__(256) csv_len_kernel(CsvArgs a, uint32_t* __restrict__ line_len, uint32_t* __restrict__ hdr) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r > a.n_reads) return;
  if (r == a.n_reads) { line_len[r] = 0; return; }
  const uint32_t* res = a.results + (size_t)r * 8;
  const uint4 lo = *(const uint4*)res;
  const uint32_t total = lo.x, ib = lo.y, best = lo.z, is = lo.w, sbest = res[4];
  const uint32_t len = a.length[r], norm = a.paired ? len - 1u : len;
  uint32_t nl = a.name_len[r]; if (nl >= 40) nl = 39;                 // OBJECTNAMEMAX (CuCLARK_hh.hh:2114-2117)
  char tmp[16];
  const int g = mic_fmt_gamma(total, norm, a.k, tmp);
  const int c = mic_fmt_conf(best, sbest, tmp);
  if (g < 0) atomicOr(&hdr[H_STATUS], (uint32_t)MIC_INGEST_ODD_RECORD);
  uint32_t ll = nl + 1 + digits_u32(norm) + 1 + (uint32_t)(g < 0 ? 1 : g) + 1 + tname_len(a, ib) + 1 + digits_u32(best) + 1 + tname_len(a, is) + 1 +
                digits_u32(sbest) + 1 + (uint32_t)c + 1;
  // a line this long (target names of hundreds of characters) goes through the host path: with every line below the bound the
  // 32-bit sum of a slot's line lengths cannot wrap (mic_ingest_alloc limits the slot size accordingly)
  if (ll > 768u) { atomicOr(&hdr[H_STATUS], (uint32_t)MIC_INGEST_TOO_MANY); ll = 0; }
  line_len[r] = ll;
}

#define CSV_LDS 24576
__global__ void __launch_bounds__(256) csv_fmt_kernel(CsvArgs a, const uint32_t* __restrict__ line_off, char* __restrict__ out) {
  __shared__ char s_buf[CSV_LDS];
  const uint32_t r0 = blockIdx.x * 256, r = r0 + threadIdx.x;
  const uint32_t r1 = r0 + 256 < a.n_reads ? r0 + 256 : a.n_reads;
  const uint32_t b0 = line_off[r0], b1 = line_off[r1];
  const bool staged = b1 - b0 <= CSV_LDS;
  if (r < a.n_reads) {
    const uint32_t* res = a.results + (size_t)r * 8;
    const uint4 lo = *(const uint4*)res;
    const uint32_t total = lo.x, ib = lo.y, best = lo.z, is = lo.w, sbest = res[4];
    const uint32_t len = a.length[r], norm = a.paired ? len - 1u : len;
    uint32_t nl = a.name_len[r]; if (nl >= 40) nl = 39;
    char* p = staged ? s_buf + (line_off[r] - b0) : out + line_off[r];
    const uint8_t* nm = a.raw + a.name_s[r];
    for (uint32_t i = 0; i < nl; ++i) *p++ = (char)nm[i];
    char tmp[16];
    *p++ = ','; { const int n = mic_fmt_u32(norm, tmp); for (int i = 0; i < n; ++i) *p++ = tmp[i]; }
    *p++ = ','; { int n = mic_fmt_gamma(total, norm, a.k, tmp); if (n < 0) { tmp[0] = '?'; n = 1; } for (int i = 0; i < n; ++i) *p++ = tmp[i]; }
    for (int f = 0; f < 2; ++f) {
      const uint32_t idx1 = f ? is : ib, sc = f ? sbest : best;
      *p++ = ',';
      if (idx1 == 0 || idx1 > a.n_targets) { *p++ = 'N'; *p++ = 'A'; }
      else { const char* t = a.tnames + a.tname_off[idx1 - 1]; const uint32_t tl = a.tname_off[idx1] - a.tname_off[idx1 - 1]; for (uint32_t i = 0; i < tl; ++i) *p++ = t[i]; }
      *p++ = ','; { const int n = mic_fmt_u32(sc, tmp); for (int i = 0; i < n; ++i) *p++ = tmp[i]; }
    }
    *p++ = ','; { const int n = mic_fmt_conf(best, sbest, tmp); for (int i = 0; i < n; ++i) *p++ = tmp[i]; }
    *p++ = '\n';
  }
  if (staged) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < b1 - b0; i += 256) out[b0 + i] = s_buf[i];
  }
}

__global__ void csv_finish_kernel(const uint32_t* __restrict__ line_off, uint32_t n_reads, const uint32_t* __restrict__ flagged,
                                  const uint32_t* __restrict__ rp, uint32_t* __restrict__ hdr) {
  hdr[H_CSV_BYTES] = line_off[n_reads];
  hdr[H_FLAGGED] = flagged[0];
  hdr[H_CONT] = rp[n_reads];
  if (flagged[0]) atomicOr(&hdr[H_STATUS], (uint32_t)MIC_INGEST_DENSE);
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct Slot {
  // pinned host
  uint8_t* h_raw = nullptr; char* h_csv = nullptr; uint32_t* h_hdr = nullptr; uint32_t* h_results = nullptr;
  // device
  uint8_t* d_raw = nullptr; uint32_t* d_tile = nullptr; uint32_t* d_tile_off = nullptr; uint32_t* d_line_start = nullptr;
  uint32_t* d_flag = nullptr; uint32_t* d_rec_of_line = nullptr; uint32_t* d_hdr_line = nullptr;
  RecArrays rec{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* d_rp = nullptr; uint16_t* d_cont = nullptr; uint32_t* d_results = nullptr; uint32_t* d_flagged = nullptr;
  uint32_t* d_line_len = nullptr; uint32_t* d_line_off = nullptr; char* d_csv = nullptr; uint32_t* d_hdr = nullptr;
  void* d_tmp = nullptr; size_t tmp_bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr, ev_up = nullptr, ev_k = nullptr;
  uint32_t n_reads = 0, cont_used = 0;
  std::vector<void*> dev_allocs, host_allocs;
};

struct Ingest {
  mic_engine* eng = nullptr;
  size_t max_bytes = 0, max_lines = 0, max_reads = 0, max_tiles = 0, cont_cap = 0, csv_cap = 0;
  char* d_tnames = nullptr; uint32_t* d_tname_off = nullptr; uint32_t n_targets = 0;
  int want_results = 0;
  std::vector<Slot> slots;
};

const uint32_t kFlaggedCapI = 1024;

// The wait of a slot's host thread for its stream (three times per batch) does not spin: the command line runs one such thread per
// slot next to its loaders, and a job that is allowed 16 CPUs (cgroup quota) has none to burn in busy-waits - with
// hipEventSynchronize's spinning one and the same box gave 108-195 Mreads/s from run to run, with this 189-194.
// Polling with short sleeps, not hipEventBlockingSync: the interrupt-driven wait stalled once for minutes under rocprofv3 --pmc.
// MIC_INGEST_SPIN=1 restores the busy-wait (lowest latency on an idle host).
static hipError_t wait_event(hipEvent_t ev) {
  static const bool spin = getenv("MIC_INGEST_SPIN") != nullptr;
  if (spin) return hipEventSynchronize(ev);
  // (hipErrorNotReady is recorded as the thread's last error like any other: an error of a launch before the wait is taken
  // out first and returned, and the "not ready" answers are taken out behind the wait - hipGetLastError() keeps its meaning
  // for the launches that follow)
  hipError_t pending = hipGetLastError();
  if (pending != hipSuccess) return pending;
  struct timespec ts = {0, 30000};                     // 30 us: a batch takes ~1 ms on the device
  hipError_t e;
  for (int i = 0;; ++i) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
    if (i >= 64) nanosleep(&ts, nullptr);              // a batch that is nearly done: a few microseconds of queries first
  }
  (void)hipGetLastError();
  return e;
}

#define ITRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return mic_set_error(e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// A slot is ONE device allocation and ONE pinned allocation carved into its arrays (a few hundred hipMalloc /
// hipHostMalloc calls cost more than the first batches take): pass 1 adds the sizes up, pass 2 hands the pieces out.
struct Arena {
  char* base = nullptr; size_t off = 0;
  template <typename T> void take(T** p, size_t n) {
    off = (off + 255) & ~(size_t)255;
    if (base) *p = (T*)(base + off);
    off += n * sizeof(T) + 64;
  }
};

void carve_slot(Ingest* g, Slot& s, Arena& dv, Arena& hs, size_t tmp) {
  hs.take(&s.h_raw, g->max_bytes + 64);
  hs.take(&s.h_csv, g->csv_cap);
  hs.take(&s.h_hdr, (size_t)H_WORDS);
  if (g->want_results) hs.take(&s.h_results, g->max_reads * 8);
  dv.take(&s.d_raw, g->max_bytes + 64);
  dv.take(&s.d_tile, g->max_tiles + 1);
  dv.take(&s.d_tile_off, g->max_tiles + 1);
  dv.take(&s.d_line_start, g->max_lines + 2);
  dv.take(&s.d_flag, g->max_lines + 2);
  dv.take(&s.d_rec_of_line, g->max_lines + 2);
  dv.take(&s.d_hdr_line, g->max_reads + 2);
  dv.take(&s.rec.name_s, g->max_reads + 1);
  dv.take(&s.rec.seq_s, g->max_reads + 1);
  dv.take(&s.rec.seq_e, g->max_reads + 1);
  dv.take(&s.rec.length, g->max_reads + 1);
  dv.take(&s.rec.bound, g->max_reads + 1);
  dv.take(&s.rec.name_len, g->max_reads + 1);
  dv.take(&s.d_rp, g->max_reads + 2);
  dv.take(&s.d_cont, g->cont_cap + 192);
  dv.take(&s.d_results, (g->max_reads + 1) * 8);
  dv.take(&s.d_flagged, (size_t)kFlaggedCapI + 1);
  dv.take(&s.d_line_len, g->max_reads + 1);
  dv.take(&s.d_line_off, g->max_reads + 1);
  dv.take(&s.d_csv, g->csv_cap);
  dv.take(&s.d_hdr, (size_t)H_WORDS);
  dv.take((char**)&s.d_tmp, tmp);
  s.tmp_bytes = tmp;
}

int alloc_slot(Ingest* g, Slot& s, size_t tmp, int device) {
  if (hipSetDevice(device) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  Arena dv, hs;
  carve_slot(g, s, dv, hs, tmp);
  void* d = nullptr; void* h = nullptr;
  hipError_t e = hipMalloc(&d, dv.off + 256);
  if (e != hipSuccess) return mic_set_error(MIC_E_NOMEM, "ingest slot: %zu bytes of device memory: %s", dv.off, hipGetErrorString(e));
  s.dev_allocs.push_back(d);
  mic_bind_thread_near_device(device, 1);           // pinned memory on the device's socket: the link runs ~1.4x faster
  e = hipHostMalloc(&h, hs.off + 256, hipHostMallocDefault);
  if (e == hipSuccess) memset(h, 0, hs.off + 256);
  mic_bind_thread_near_device(device, 0);
  if (e != hipSuccess) return mic_set_error(MIC_E_NOMEM, "ingest slot: %zu bytes of pinned memory: %s", hs.off, hipGetErrorString(e));
  s.host_allocs.push_back(h);
  Arena dv2, hs2;
  dv2.base = (char*)d; hs2.base = (char*)h;
  carve_slot(g, s, dv2, hs2, tmp);
  if ((e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming)) != hipSuccess ||
      // the query kernel's read-ahead looks past the last read of a batch: no stale length slots there
      (e = hipMemsetAsync(s.d_cont, 0, (g->cont_cap + 192) * 2, s.stream)) != hipSuccess ||
      (e = hipStreamSynchronize(s.stream)) != hipSuccess)
    return mic_set_error(MIC_E_HIP, "ingest slot setup: %s", hipGetErrorString(e));
  return MIC_OK;
}

void free_ingest(Ingest* g) {
  if (!g) return;
  for (Slot& s : g->slots) {
    if (s.stream) hipStreamSynchronize(s.stream);
    for (void* p : s.dev_allocs) hipFree(p);
    for (void* p : s.host_allocs) hipHostFree(p);
    if (s.ev) hipEventDestroy(s.ev);
    if (s.ev_up) hipEventDestroy(s.ev_up);
    if (s.ev_k) hipEventDestroy(s.ev_k);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  if (g->d_tnames) hipFree(g->d_tnames);
  if (g->d_tname_off) hipFree(g->d_tname_off);
  delete g;
}

double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec / 1e9; }

}  // namespace

// the engine keeps one Ingest* (opaque to it): mic_engine.hip
void** mic_engine_ingest_slot(mic_engine* e);

extern "C" {

int mic_ingest_alloc(mic_engine* e, size_t n_slots, size_t max_bytes, const char* const* target_names, uint32_t n_targets,
                     int want_results, uint8_t** raw) {
  if (!e || !raw || n_slots == 0 || n_slots > 64) return mic_set_error(MIC_E_INVALID, "bad argument");
  // at most 128 MiB: with the per-line bound of csv_len_kernel (768 bytes) the 32-bit scan of the CSV line lengths cannot wrap
  if (max_bytes < 4096 || max_bytes > ((size_t)128 << 20)) return mic_set_error(MIC_E_INVALID, "ingest slots hold 4 KiB .. 128 MiB of input");
  MicTable t; int sc, ncu, dev, k; uint32_t nt;
  int rc = mic_engine_table(e, &t, &sc, &ncu, &dev, &k, &nt);
  if (rc) return rc;
  ITRY(hipSetDevice(dev));
  mic_ingest_free(e);
  Ingest* g = new Ingest();
  g->eng = e;
  g->max_bytes = (max_bytes + ING_TILE - 1) / ING_TILE * ING_TILE;
  g->max_tiles = g->max_bytes / ING_TILE;
  g->max_lines = g->max_bytes / 16 + 64;          // fewer than 16 bytes per line on average: host path
  g->max_reads = g->max_bytes / 32 + 16;          // fewer than 32 bytes per record on average: host path
  // sum of the per-read reservations of record_kernel: nbytes / 8 + 2 (nbytes / (k + 1) + 1) + 8 containers per read
  g->cont_cap = g->max_bytes / 8 + 2 * (g->max_bytes / (size_t)(k + 1)) + 10 * g->max_reads + 64;
  g->csv_cap = g->max_bytes;
  g->want_results = want_results;
  g->n_targets = n_targets;
  *mic_engine_ingest_slot(e) = g;
  {  // target names on the device
    std::vector<uint32_t> off(n_targets + 1, 0);
    std::vector<char> names;
    for (uint32_t i = 0; i < n_targets; ++i) {
      const char* s = target_names && target_names[i] ? target_names[i] : "";
      names.insert(names.end(), s, s + strlen(s));
      off[i + 1] = (uint32_t)names.size();
    }
    ITRY(hipMalloc(&g->d_tnames, names.size() + 16));
    ITRY(hipMalloc(&g->d_tname_off, off.size() * 4));
    if (!names.empty()) ITRY(hipMemcpy(g->d_tnames, names.data(), names.size(), hipMemcpyHostToDevice));
    ITRY(hipMemcpy(g->d_tname_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
  }
  g->slots.resize(n_slots);
  size_t tmp1 = 0, tmp2 = 0, tmp3 = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp1, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(g->max_tiles + 1));
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(g->max_lines + 1));
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp3, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(g->max_reads + 1));
  const size_t tmp = std::max(tmp1, std::max(tmp2, tmp3)) + 256;
  {  // slots are set up concurrently: pinning host pages is the slow part
    std::vector<int> rcs(n_slots, MIC_OK);
    std::vector<std::string> msgs(n_slots);
    std::vector<std::thread> th;
    for (size_t i = 0; i < n_slots; ++i)
      th.emplace_back([&, i] { rcs[i] = alloc_slot(g, g->slots[i], tmp, dev); if (rcs[i]) msgs[i] = mic_last_error(); });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < n_slots; ++i) {
      if (rcs[i]) return mic_set_error(rcs[i], "%s", msgs[i].c_str());
      raw[i] = g->slots[i].h_raw;
    }
  }
  return MIC_OK;
}

int mic_ingest_free(mic_engine* e) {
  if (!e) return mic_set_error(MIC_E_INVALID, "null engine");
  void** slot = mic_engine_ingest_slot(e);
  if (*slot) { free_ingest((Ingest*)*slot); *slot = nullptr; }
  return MIC_OK;
}

int mic_ingest_classify(mic_engine* e, size_t slot_id, size_t n_bytes, int flags, mic_ingest_result* out) {
  const int paired = flags & MIC_INGEST_PAIRED;
  const uint32_t lpr = (flags & MIC_INGEST_FASTQ_2LINE) ? 2u : 4u;      // lines per FASTQ record
  if (!e || !out) return mic_set_error(MIC_E_INVALID, "null argument");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(e);
  if (!g || slot_id >= g->slots.size()) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  if (n_bytes == 0 || n_bytes > g->max_bytes) return mic_set_error(MIC_E_INVALID, "batch of %zu bytes does not fit the slot (%zu)", n_bytes, g->max_bytes);
  MicTable t; int sc, ncu, dev, k; uint32_t nt;
  int rc = mic_engine_table(e, &t, &sc, &ncu, &dev, &k, &nt);
  if (rc) return rc;
  if (!t.slots) return mic_set_error(MIC_E_STATE, "no database loaded");
  ITRY(hipSetDevice(dev));
  Slot& s = g->slots[slot_id];
  memset(out, 0, sizeof(*out));
  const uint8_t first = s.h_raw[0];
  if (first != '>' && first != '@') { out->status = MIC_INGEST_FALLBACK | MIC_INGEST_ODD_RECORD; return MIC_OK; }
  const int fasta = first == '>';
  static const bool timing = getenv("MIC_INGEST_TIMING") != nullptr;
  const double t0 = timing ? now_s() : 0;
  hipStream_t st = s.stream;
  const uint32_t nb = (uint32_t)n_bytes;
  const uint32_t n_tiles = (nb + ING_TILE - 1) / ING_TILE;
  // ---- phase 1: bytes -> lines -> number of records.  The upload runs on the engine's upload stream, the CSV comes
  // back on its download stream (mic_engine.hip: one stream per direction keeps both directions of the link busy)
  hipStream_t up, down;
  mic_engine_copy_streams(e, &up, &down);
  ITRY(hipMemcpyAsync(s.d_raw, s.h_raw, n_bytes, hipMemcpyHostToDevice, up));
  ITRY(hipEventRecord(s.ev_up, up));
  ITRY(hipStreamWaitEvent(st, s.ev_up, 0));
  ITRY(hipMemsetAsync(s.d_hdr, 0, H_WORDS * 4, st));
  line_count_kernel<<<n_tiles, 256, 0, st>>>(s.d_raw, nb, s.d_tile);
  ITRY(hipMemsetAsync(s.d_tile + n_tiles, 0, 4, st));
  size_t tb = s.tmp_bytes;
  ITRY(hipcub::DeviceScan::ExclusiveSum(s.d_tmp, tb, s.d_tile, s.d_tile_off, (int)(n_tiles + 1), st));
  line_start_kernel<<<n_tiles, 256, 0, st>>>(s.d_raw, nb, s.d_tile_off, s.d_line_start, (uint32_t)g->max_lines);
  lines_finish_kernel<<<1, 1, 0, st>>>(s.d_raw, nb, s.d_tile_off, n_tiles, s.d_line_start, (uint32_t)g->max_lines, fasta, lpr, (uint32_t)g->max_reads, s.d_hdr);
  if (fasta) {
    // a line is at least one byte long, so there are at most nb of them: flags and their scan cover lines 0 .. n_scan - 1
    // (n_lines < n_scan whenever the batch is within the slot's line capacity)
    const uint32_t n_scan = (uint32_t)std::min<size_t>(g->max_lines, (size_t)nb + 2);
    const uint32_t gl = (n_scan + 255) / 256;
    fasta_flag_kernel<<<gl, 256, 0, st>>>(s.d_raw, nb, s.d_line_start, s.d_hdr, (uint32_t)g->max_lines, n_scan, s.d_flag);
    tb = s.tmp_bytes;
    ITRY(hipcub::DeviceScan::ExclusiveSum(s.d_tmp, tb, s.d_flag, s.d_rec_of_line, (int)n_scan, st));
    fasta_scatter_kernel<<<gl, 256, 0, st>>>(s.d_flag, s.d_rec_of_line, (uint32_t)g->max_lines, (uint32_t)g->max_reads, s.d_hdr_line, s.d_hdr);
  }
  ITRY(hipMemcpyAsync(s.h_hdr, s.d_hdr, H_WORDS * 4, hipMemcpyDeviceToHost, st));
  ITRY(hipEventRecord(s.ev, st));
  ITRY(wait_event(s.ev));
  const double t1 = timing ? now_s() : 0;
  uint32_t n_reads = s.h_hdr[H_NREADS];
  out->n_lines = s.h_hdr[H_NLINES];
  if (s.h_hdr[H_STATUS] || n_reads == 0) { out->status = MIC_INGEST_FALLBACK | s.h_hdr[H_STATUS]; return MIC_OK; }
  // ---- phase 2: records -> packed reads -> query -> CSV line lengths
  const uint32_t gr = (n_reads + 1 + 255) / 256;
  if (fasta) record_kernel<true><<<gr, 256, 0, st>>>(s.d_raw, nb, s.d_line_start, s.d_hdr_line, n_reads, lpr, k, s.rec, s.d_hdr);
  else record_kernel<false><<<gr, 256, 0, st>>>(s.d_raw, nb, s.d_line_start, nullptr, n_reads, lpr, k, s.rec, s.d_hdr);
  tb = s.tmp_bytes;
  ITRY(hipcub::DeviceScan::ExclusiveSum(s.d_tmp, tb, s.rec.bound, s.d_rp, (int)(n_reads + 1), st));
  {
    unsigned blocks = (n_reads + 3) / 4, cap = (unsigned)ncu * 64u;
    if (blocks > cap) blocks = cap;
    pack_kernel<<<blocks, 256, 0, st>>>(s.d_raw, s.rec.seq_s, s.rec.seq_e, s.d_rp, s.d_cont, n_reads, k);
  }
  ITRY(hipMemsetAsync(s.d_flagged, 0, 4, st));
  MicQueryArgs qa;
  qa.t = t; qa.reads_ptr = s.d_rp; qa.cont = s.d_cont; qa.n_reads = n_reads; qa.row_words = 0; qa.results = s.d_results;
  qa.rows = nullptr; qa.flagged = s.d_flagged; qa.flagged_cap = kFlaggedCapI;
  ITRY(mic_launch_query(qa, sc, ncu, st));
  CsvArgs ca;
  ca.raw = s.d_raw; ca.name_s = s.rec.name_s; ca.name_len = s.rec.name_len; ca.length = s.rec.length; ca.results = s.d_results;
  ca.tnames = g->d_tnames; ca.tname_off = g->d_tname_off; ca.n_targets = g->n_targets; ca.n_reads = n_reads; ca.k = k; ca.paired = paired ? 1 : 0;
  csv_len_kernel<<<gr, 256, 0, st>>>(ca, s.d_line_len, s.d_hdr);
  tb = s.tmp_bytes;
  ITRY(hipcub::DeviceScan::ExclusiveSum(s.d_tmp, tb, s.d_line_len, s.d_line_off, (int)(n_reads + 1), st));
  csv_finish_kernel<<<1, 1, 0, st>>>(s.d_line_off, n_reads, s.d_flagged, s.d_rp, s.d_hdr);
  ITRY(hipMemcpyAsync(s.h_hdr, s.d_hdr, H_WORDS * 4, hipMemcpyDeviceToHost, st));
  ITRY(hipEventRecord(s.ev, st));
  ITRY(wait_event(s.ev));
  const double t2 = timing ? now_s() : 0;
  s.n_reads = n_reads; s.cont_used = s.h_hdr[H_CONT];
  const uint32_t csv_bytes = s.h_hdr[H_CSV_BYTES];
  uint32_t status = s.h_hdr[H_STATUS];
  if (csv_bytes > g->csv_cap || s.h_hdr[H_CONT] > g->cont_cap) status |= MIC_INGEST_TOO_MANY;
  if (status) { out->status = MIC_INGEST_FALLBACK | status; return MIC_OK; }
  // ---- phase 3: CSV text -> host
  csv_fmt_kernel<<<(n_reads + 255) / 256, 256, 0, st>>>(ca, s.d_line_off, s.d_csv);
  ITRY(hipGetLastError());
  ITRY(hipEventRecord(s.ev_k, st));
  ITRY(hipStreamWaitEvent(down, s.ev_k, 0));
  ITRY(hipMemcpyAsync(s.h_csv, s.d_csv, csv_bytes, hipMemcpyDeviceToHost, down));
  if (g->want_results) ITRY(hipMemcpyAsync(s.h_results, s.d_results, (size_t)n_reads * 32, hipMemcpyDeviceToHost, down));
  ITRY(hipEventRecord(s.ev, down));
  ITRY(wait_event(s.ev));
  out->n_reads = n_reads; out->csv_bytes = csv_bytes; out->csv = s.h_csv; out->results = g->want_results ? s.h_results : nullptr;
  out->status = MIC_INGEST_OK;
  if (timing) {
    const double t3 = now_s();
    fprintf(stderr, "[ingest] slot %zu: %u bytes, %u reads: lines %.0f us, pack+query+lengths %.0f us, csv %.0f us\n", slot_id, nb, n_reads,
            (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6);
  }
  return MIC_OK;
}

int mic_ingest_fetch_packed(mic_engine* e, size_t slot_id, uint32_t* reads_pointer, size_t rp_cap, uint16_t* containers, size_t cont_cap,
                            uint64_t* n_reads, uint64_t* n_containers) {
  if (!e) return mic_set_error(MIC_E_INVALID, "null engine");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(e);
  if (!g || slot_id >= g->slots.size()) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  Slot& s = g->slots[slot_id];
  if (n_reads) *n_reads = s.n_reads;
  if (n_containers) *n_containers = s.cont_used;
  if (reads_pointer) {
    if (rp_cap < (size_t)s.n_reads + 1) return mic_set_error(MIC_E_INVALID, "reads_pointer capacity too small");
    ITRY(hipMemcpy(reads_pointer, s.d_rp, ((size_t)s.n_reads + 1) * 4, hipMemcpyDeviceToHost));
  }
  if (containers) {
    if (cont_cap < s.cont_used) return mic_set_error(MIC_E_INVALID, "containers capacity too small");
    ITRY(hipMemcpy(containers, s.d_cont, (size_t)s.cont_used * 2, hipMemcpyDeviceToHost));
  }
  return MIC_OK;
}

// host build of the device formatter (tests pin it against the C library's "%g")
int mic_format_ratio_g(uint32_t num, uint32_t den, char* out16) {
  if (!out16 || den == 0 || num == 0 || num > den) return MIC_E_INVALID;
  const int n = mic_fmt_g_unit((double)num / (double)den, out16);
  out16[n] = 0;
  return n;
}

}  // extern "C"
