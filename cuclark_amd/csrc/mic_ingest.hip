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

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct mic_engine;
// engine internals this file needs (mic_engine.hip)
int mic_engine_table(mic_engine* e, MicTable* t, int* slot_class, int* n_cu, int* device, int* k, uint32_t* n_targets);
int mic_set_error(int code, const char* fmt, ...);
int mic_bind_thread_near_device(int device, int on);
void mic_engine_copy_streams(mic_engine* e, hipStream_t* up, hipStream_t* down);
void mic_peer_enable_engines(mic_engine* const* engines, size_t n);
bool mic_peer_enable(int from, int to);

namespace {

enum { H_NLINES = 0, H_NREADS = 1, H_STATUS = 2, H_CSV_BYTES = 3, H_FLAGGED = 4, H_NEWLINES = 5, H_CONT = 6, H_WORDS = 8 };

#define ING_TILE 4096          // bytes per block of the line kernels (256 threads x 16 B)

__device__ __forceinline__ uint32_t newline_mask(uint32_t x) {       // 0x80 in every byte of x that is '\n'
  const uint32_t y = x ^ 0x0A0A0A0Au;
  return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// newline flags of the 16 bytes at byte offset `pos` (16-byte aligned): bit i = byte i is '\n' and lies below nb
// (from: bytes below it are not the text's - the front of an aligned-down buffer, mic_text_index_front_device)
__device__ __forceinline__ uint32_t tile_flags(const uint8_t* __restrict__ raw, uint32_t pos, uint32_t nb, uint32_t from = 0) {
  if (pos >= nb || pos + 16 <= from) return 0;
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
  if (pos < from) f &= ~((1u << (from - pos)) - 1u);
  return f;
}

__global__ void __launch_bounds__(256) line_count_kernel(const uint8_t* __restrict__ raw, uint32_t nb, uint32_t* __restrict__ tile_cnt, uint32_t from = 0) {
  __shared__ uint32_t s_w[4];
  const uint32_t pos = blockIdx.x * ING_TILE + threadIdx.x * 16;
  uint32_t c = __popc(tile_flags(raw, pos, nb, from));
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// line_start[L] = offset of the first byte of line L (line 0 starts at 0; a '\n' at p starts the next line at p + 1)
__global__ void __launch_bounds__(256) line_start_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ tile_off,
                                                         uint32_t* __restrict__ line_start, uint32_t cap, uint32_t from = 0) {
  __shared__ uint32_t s_w[4];
  const uint32_t pos = blockIdx.x * ING_TILE + threadIdx.x * 16;
  uint32_t f = tile_flags(raw, pos, nb, from);
  const uint32_t c = __popc(f);
  uint32_t inc = c;                               // inclusive scan inside the wave
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) s_w[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t base = tile_off[blockIdx.x];
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_w[w];
  uint32_t rank = base + inc - c;                 // newlines before this thread's bytes
  if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = from;
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
// ---- a pair of FASTQ texts resident on the device (the inflated mates of -P a.fq.gz b.fq.gz): line index, checks, merge --------------
// The reference merges a pair line by line into ">id\nseq1Nseq2\n" (file.cc:205-268); the command line's loaders do that on the
// host for plain files.  Text that was inflated ON the device (mic_gz.hip) is merged here instead, record ranges straight into an
// ingest slot's device buffer, so that the text never crosses the link.  Whatever the line arithmetic does not cover (line
// counts that differ or are no multiple of four, a header without '@', ids that differ or are empty) is reported, and the
// caller goes back to the host reader, which treats such files the way the reference does, messages included.
enum { PS_LINES = MIC_PAIRS_LINES, PS_HEADER = MIC_PAIRS_HEADER, PS_ID = MIC_PAIRS_ID, PS_BIG = MIC_PAIRS_BIG };

__device__ __forceinline__ bool pair_sep(uint8_t c) { return c == ' ' || c == '/' || c == '\t' || c == '@'; }     // file.cc:224
// the id inside a header line [p, p + n): behind leading separators, up to the next one
__device__ __forceinline__ void pair_id(const uint8_t* __restrict__ p, uint32_t n, uint32_t& a, uint32_t& len) {
  a = 0;
  while (a < n && pair_sep(p[a])) ++a;
  uint32_t b = a;
  while (b < n && !pair_sep(p[b])) ++b;
  len = b - a;
}

struct PairText { const uint8_t* t; const uint32_t* ls; uint32_t nb; };
// line L of a text without its '\n' (an unterminated last line has its virtual line end at nb: line_start = nb + 1)
__device__ __forceinline__ void pair_line(const PairText& x, uint64_t L, const uint8_t*& p, uint32_t& n) {
  const uint32_t a = x.ls[L], b = x.ls[L + 1];
  p = x.t + a; n = b - 1u - a;
}

// one thread per record: checks, and the length of the merged record
__global__ void __launch_bounds__(256) pair_len_kernel(PairText A, PairText B, uint64_t n_rec, unsigned long long* __restrict__ mlen,
                                                       uint32_t* __restrict__ status) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > n_rec) return;
  if (r == n_rec) { mlen[r] = 0; return; }
  const uint8_t *p, *q; uint32_t n, m;
  pair_line(A, 4 * r, p, n); pair_line(B, 4 * r, q, m);
  uint32_t bad = 0, ia = 0, la = 0, ib = 0, lb = 0;
  if (n == 0 || m == 0 || p[0] != '@' || q[0] != '@') bad |= PS_HEADER;
  else {
    pair_id(p, n, ia, la); pair_id(q, m, ib, lb);
    if (la == 0 || la != lb) bad |= PS_ID;
    else for (uint32_t i = 0; i < la; ++i) if (p[ia + i] != q[ib + i]) { bad |= PS_ID; break; }
  }
  const uint32_t s1 = A.ls[4 * r + 2] - 1u - A.ls[4 * r + 1], s2 = B.ls[4 * r + 2] - 1u - B.ls[4 * r + 1];
  mlen[r] = (unsigned long long)la + 2u + s1 + 1u + s2 + 1u;
  if (bad) atomicOr(status, bad);
}

__global__ void __launch_bounds__(256) pair_sample_kernel(const unsigned long long* __restrict__ off, uint64_t n_rec, uint32_t stride,
                                                          uint64_t n_samples, unsigned long long* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_samples) return;
  const uint64_t r = i * stride;
  out[i] = off[r < n_rec ? r : n_rec];
}

// a single FASTQ text: out[i] = byte offset of record min(i * stride, n_rec) (line 4 r; behind the last record: the end of the text)
__global__ void __launch_bounds__(256) text_sample_kernel(const uint32_t* __restrict__ ls, uint32_t nb, uint64_t n_rec, uint32_t stride,
                                                          uint64_t n_samples, unsigned long long* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_samples) return;
  const uint64_t r = i * stride < n_rec ? i * stride : n_rec;
  const uint32_t o = ls[4 * r];
  out[i] = o < nb ? o : nb;
}

// FASTA: flag[L] = line L starts a record ('>' in its first column, CuCLARK_hh.hh:1369-1389); flag[n_lines] = 0 for the scan's total
__global__ void __launch_bounds__(256) text_fasta_flag_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ ls,
                                                              uint64_t n_lines, uint32_t* __restrict__ flag) {
  const uint64_t L = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (L > n_lines) return;
  uint32_t f = 0;
  if (L < n_lines) { const uint32_t p = ls[L]; f = p < nb && raw[p] == '>'; }
  flag[L] = f;
}
// out[r / stride] = byte offset of record r for every stride-th record; the entries behind the last one: the end of the text
__global__ void __launch_bounds__(256) text_fasta_sample_kernel(const uint32_t* __restrict__ ls, const uint32_t* __restrict__ flag,
                                                                const uint32_t* __restrict__ rec_of_line, uint64_t n_lines, uint32_t stride,
                                                                unsigned long long* __restrict__ out) {
  const uint64_t L = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (L >= n_lines || !flag[L]) return;
  const uint32_t r = rec_of_line[L];
  if (r % stride == 0) out[r / stride] = ls[L];
}

// (the device code of this file is loaded when its first kernel is launched: mic_ingest_alloc does that, not the first batch)
__global__ void ingest_warm_kernel(uint32_t* p) { if (p && threadIdx.x == 1000) *p = 0; }

// what the host wants to know about a text once its line ends are counted, gathered for ONE small copy (a copy into pageable host
// memory costs about a millisecond whatever its size): out[0] = line ends, out[1] = last byte, out[2] = first byte
__global__ void text_facts_kernel(const uint8_t* __restrict__ raw, uint32_t nb, const uint32_t* __restrict__ tile_off, uint32_t n_tiles,
                                  uint32_t* __restrict__ out, uint32_t from = 0) {
  out[0] = tile_off[n_tiles]; out[1] = raw[nb - 1]; out[2] = raw[from]; out[3] = 0;
}

// one wavefront per record: ">id\n" seq1 "N" seq2 "\n" at off[r] - off[r0]
__global__ void __launch_bounds__(256) pair_merge_kernel(PairText A, PairText B, uint64_t r0, uint64_t r1, const unsigned long long* __restrict__ off,
                                                         uint8_t* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const uint64_t r = r0 + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= r1) return;
  uint8_t* o = dst + (off[r] - off[r0]);
  const uint8_t *p, *q; uint32_t n, m;
  pair_line(A, 4 * r, p, n);
  uint32_t ia, la;
  pair_id(p, n, ia, la);
  if (lane == 0) o[0] = '>';
  for (uint32_t i = lane; i < la; i += 64) o[1 + i] = p[ia + i];
  if (lane == 0) o[1 + la] = '\n';
  o += la + 2;
  pair_line(A, 4 * r + 1, p, n);
  for (uint32_t i = lane; i < n; i += 64) o[i] = p[i];
  if (lane == 0) o[n] = 'N';
  o += n + 1;
  pair_line(B, 4 * r + 1, q, m);
  for (uint32_t i = lane; i < m; i += 64) o[i] = q[i];
  if (lane == 0) o[m] = '\n';
}

struct Slot {
  // pinned host
  uint8_t* h_raw = nullptr; char* h_csv = nullptr; uint32_t* h_hdr = nullptr; uint32_t* h_results = nullptr;
  // device
  uint8_t* d_raw = nullptr; uint32_t* d_tile = nullptr; uint32_t* d_tile_off = nullptr; uint32_t* d_line_start = nullptr;
  uint32_t* d_flag = nullptr; uint32_t* d_rec_of_line = nullptr; uint32_t* d_hdr_line = nullptr;
  RecArrays rec{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* d_rp = nullptr; uint16_t* d_cont = nullptr; uint32_t* d_results = nullptr; uint32_t* d_flagged = nullptr;
  uint32_t* d_crowd = nullptr;   // work area of the crowded runs' follow-up (mic_internal.h: mic_crowd_dims)
  uint32_t* d_line_len = nullptr; uint32_t* d_line_off = nullptr; char* d_csv = nullptr; uint32_t* d_hdr = nullptr;
  void* d_tmp = nullptr; size_t tmp_bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr, ev_up = nullptr, ev_k = nullptr;
  uint32_t n_reads = 0, cont_used = 0;
  std::vector<void*> dev_allocs, host_allocs;
  // table-sharded batches (mic_ingest_classify_group): what this slot keeps on every engine of its group, the owner included
  struct Peer {
    mic_engine* eng = nullptr; int device = 0;
    void* block = nullptr;
    uint32_t* d_rp = nullptr; uint16_t* d_cont = nullptr;   // the packed reads (helpers: a peer copy; owner: the slot's own arrays)
    uint32_t* d_rows = nullptr;                             // this engine's partial rows of ALL reads of the batch
    uint32_t* d_gather = nullptr; uint32_t* d_acc = nullptr; uint32_t* d_acc2 = nullptr;  // rows of this engine's read range from the others; the running sum's two buffers
    uint32_t* d_res = nullptr;                              // helpers: results of all reads from the query kernel, then of the range
    uint32_t* d_flagged = nullptr; uint32_t* d_crowd = nullptr;
    hipStream_t stream = nullptr;                           // helpers: a stream on their device; owner: the slot's stream
    hipEvent_t ev_q = nullptr, ev_done = nullptr;           // this engine's rows are written; its range is finished and delivered
    // MIC_GROUP_TIMING: packed reads asked for / arrived (= kernel start) / kernel done / all other engines' rows ready / range delivered
    hipEvent_t tv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const void* table_id = nullptr;                         // the engine's table when the buffers were set up (a reloaded engine is another peer)
  };
  std::vector<Peer> peers;
  hipEvent_t ev_pack = nullptr;
  size_t group_owner = 0;
  uint64_t fan_bytes = 0, x_bytes = 0;                      // MIC_GROUP_TIMING: bytes of the last batch's fan-out and row exchange
  bool timed = false;
};

// MIC_GROUP_TIMING=1: what the table-sharded batches of an engine's slots cost, summed over its batches (mic_ingest_group_stats)
struct GroupStats {
  std::mutex mu;
  uint64_t batches = 0, reads = 0, fan_bytes = 0, x_bytes = 0;
  double fan_ms_sum = 0, fan_ms_max = 0, kernel_ms_sum = 0, kernel_ms_max = 0, x_ms_sum = 0, x_ms_max = 0, span_ms = 0;
};

struct Ingest {
  mic_engine* eng = nullptr; int device = 0;
  size_t max_bytes = 0, max_lines = 0, max_reads = 0, max_tiles = 0, cont_cap = 0, csv_cap = 0;
  // reads the work area of the crowded runs' follow-up is sized for (mic_internal.h: mic_crowd_dims), 0: none - the engine's table has no
  // side table (a slot holds max_bytes / 32 reads at most and ~max_bytes / 160 in practice: a quarter of the bound; what does not fit
  // takes the dense path).  Sized by the bound it was 100 MB per slot and per helper engine of a group - 13 GB of allocations in
  // front of the first batches of an 8-part run.
  size_t crowd_reads = 0;
  char* d_tnames = nullptr; uint32_t* d_tname_off = nullptr; uint32_t n_targets = 0;
  int want_results = 0;
  std::vector<Slot> slots;
  GroupStats gstats;
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
  if (g->crowd_reads) dv.take(&s.d_crowd, mic_crowd_dims(g->max_reads / 4 + 64).words);
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
  if ((e = mic_stream_get(&s.stream)) != hipSuccess ||
      (e = mic_event_get(&s.ev, false)) != hipSuccess ||
      (e = mic_event_get(&s.ev_up, false)) != hipSuccess ||
      (e = mic_event_get(&s.ev_k, false)) != hipSuccess ||
      // the query kernel's read-ahead looks past the last read of a batch: no stale length slots there
      (e = hipMemsetAsync(s.d_cont, 0, (g->cont_cap + 192) * 2, s.stream)) != hipSuccess ||
      (e = hipStreamSynchronize(s.stream)) != hipSuccess)
    return mic_set_error(MIC_E_HIP, "ingest slot setup: %s", hipGetErrorString(e));
  return MIC_OK;
}

void free_peers(Slot& s);
void free_ingest(Ingest* g) {
  if (!g) return;
  for (Slot& s : g->slots) {
    if (s.stream) hipStreamSynchronize(s.stream);
    free_peers(s);
    if (s.ev_pack) mic_event_put(s.ev_pack);
    hipSetDevice(g->device);
    for (void* p : s.dev_allocs) hipFree(p);
    for (void* p : s.host_allocs) hipHostFree(p);
    if (s.ev) mic_event_put(s.ev);
    if (s.ev_up) mic_event_put(s.ev_up);
    if (s.ev_k) mic_event_put(s.ev_k);
    if (s.stream) mic_stream_put(s.stream);
  }
  if (g->d_tnames) hipFree(g->d_tnames);
  if (g->d_tname_off) hipFree(g->d_tname_off);
  delete g;
}

double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec / 1e9; }

// ---- table-sharded batches: the slot's buffers on the engines of its group, and the exchange ------------------------------
const uint32_t kGroupRowWords = 16;      // 15 (target, count) pairs per partial row = the reference's MAXHITS rows (parameters.hh:44)

// reads whose summed row did not fit (result word 6: MIC_FLAG_ROW_OVERFLOW) are counted into the slot's flagged counter: the
// batch is then handed back with MIC_INGEST_DENSE like a batch with a read of more than 64 targets
__global__ void __launch_bounds__(256) group_overflow_kernel(const uint32_t* __restrict__ results, uint32_t n, uint32_t* __restrict__ flagged) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && (results[(size_t)i * 8 + 6] & MIC_FLAG_ROW_OVERFLOW_)) atomicAdd(&flagged[0], 1u);
}

void free_peers(Slot& s) {
  for (Slot::Peer& p : s.peers) {
    hipSetDevice(p.device);
    if (p.stream && p.stream != s.stream) { hipStreamSynchronize(p.stream); mic_stream_put(p.stream); }
    if (p.ev_q) mic_event_put(p.ev_q);
    if (p.ev_done) mic_event_put(p.ev_done);
    for (hipEvent_t e : p.tv) if (e) mic_event_put(e);
    if (p.block) hipFree(p.block);
  }
  s.peers.clear();
}

static bool group_timing() { static const bool on = getenv("MIC_GROUP_TIMING") != nullptr; return on; }

// The engines of a group must answer for ONE database cut into disjoint pieces that cover it: the same k and target count, and
// either slot-range parts (mic_db_set_part on a super-k-mer table) or bucket ranges (the other layouts, explicit shards) that
// tile the whole table.  Engines that each hold the whole table, or overlapping pieces, would count every hit several times.
int check_group(mic_engine* const* group, size_t P) {
  struct Piece { uint64_t lo, hi; };
  std::vector<Piece> pieces(P);
  int k0 = 0, layout0 = 0, parted0 = 0; uint32_t nt0 = 0; uint64_t whole = 0;
  for (size_t p = 0; p < P; ++p) {
    MicTable t; int sc, ncu, dev, k; uint32_t nt;
    int rc = mic_engine_table(group[p], &t, &sc, &ncu, &dev, &k, &nt);
    if (rc) return rc;
    if (!t.slots) return mic_set_error(MIC_E_STATE, "no database loaded on engine %zu of the group", p);
    if (p == 0) { k0 = k; nt0 = nt; layout0 = t.layout; parted0 = t.parted; whole = t.parted ? t.n_main : t.div.d; }
    if (k != k0 || nt != nt0) return mic_set_error(MIC_E_INVALID, "engine %zu of the group differs in k or in the number of targets (%d / %u against %d / %u)", p, k, nt, k0, nt0);
    if (t.layout != layout0 || t.parted != parted0) return mic_set_error(MIC_E_INVALID, "engine %zu of the group holds another table layout than engine 0", p);
    if (t.parted) {
      if (t.n_main != whole) return mic_set_error(MIC_E_INVALID, "engine %zu of the group holds a part of another table (%llu slots against %llu)", p, (unsigned long long)t.n_main, (unsigned long long)whole);
      pieces[p] = {t.slot_lo, (uint64_t)t.slot_lo + t.slot_cnt};
    } else {
      if (t.div.d != whole) return mic_set_error(MIC_E_INVALID, "engine %zu of the group holds a table of another size", p);
      pieces[p] = {t.shard_start, t.shard_end};
    }
  }
  std::sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.lo < b.lo || (a.lo == b.lo && a.hi < b.hi); });
  uint64_t at = 0;
  for (size_t p = 0; p < P; ++p) {
    if (pieces[p].lo == pieces[p].hi) continue;            // (an empty part answers nothing)
    if (pieces[p].lo != at)
      return mic_set_error(MIC_E_INVALID, "the engines of the group do not hold disjoint pieces that cover the table: piece [%llu, %llu) follows %llu%s",
                           (unsigned long long)pieces[p].lo, (unsigned long long)pieces[p].hi, (unsigned long long)at,
                           P > 1 && pieces[p].lo == 0 && pieces[p].hi == whole ? " (whole tables: use mic_db_set_part)" : "");
    at = pieces[p].hi;
  }
  if (at != whole) return mic_set_error(MIC_E_INVALID, "the engines of the group cover %llu of %llu %s of the table", (unsigned long long)at,
                                        (unsigned long long)whole, parted0 ? "slots" : "buckets");
  return MIC_OK;
}

int setup_peers(Ingest* g, Slot& s, mic_engine* const* group, size_t P, size_t owner) {
  free_peers(s);
  int rc = check_group(group, P);
  if (rc) return rc;
  s.peers.resize(P);
  s.group_owner = owner;
  s.timed = group_timing();
  const size_t len_max = g->max_reads / P + 2, rw = kGroupRowWords;
  for (size_t p = 0; p < P; ++p) {
    Slot::Peer& q = s.peers[p];
    MicTable t; int sc, ncu, dev, k; uint32_t nt;
    rc = mic_engine_table(group[p], &t, &sc, &ncu, &dev, &k, &nt);
    if (rc) return rc;
    q.eng = group[p]; q.device = dev; q.table_id = t.slots;
    ITRY(hipSetDevice(dev));
    const bool own = p == owner;
    for (int pass = 0; pass < 2; ++pass) {          // pass 0 adds the sizes up, pass 1 hands the pieces out
      Arena a;
      a.base = pass ? (char*)q.block : nullptr;
      a.take(&q.d_rows, (g->max_reads + 1) * rw);
      a.take(&q.d_gather, (P - 1) * len_max * rw);
      a.take(&q.d_acc, len_max * rw);
      a.take(&q.d_acc2, len_max * rw);
      if (!own) {
        a.take(&q.d_rp, g->max_reads + 2);
        a.take(&q.d_cont, g->cont_cap + 192);
        a.take(&q.d_res, (g->max_reads + 1) * 8);
        a.take(&q.d_flagged, (size_t)kFlaggedCapI + 1);
        if (t.side) a.take(&q.d_crowd, mic_crowd_dims(g->max_reads / 4 + 64).words);      // (a helper whose PART has a side table, whatever the owner's has)
      }
      if (!pass) {
        hipError_t e = hipMalloc(&q.block, a.off + 256);
        if (e != hipSuccess) return mic_set_error(MIC_E_NOMEM, "table-sharded ingest slot: %zu bytes on device %d: %s", a.off, dev, hipGetErrorString(e));
      }
    }
    ITRY(mic_event_get(&q.ev_q, false));
    ITRY(mic_event_get(&q.ev_done, false));
    if (s.timed) for (hipEvent_t& e : q.tv) ITRY(mic_event_get(&e, true));
    if (own) { q.d_rp = s.d_rp; q.d_cont = s.d_cont; q.d_res = s.d_results; q.d_flagged = s.d_flagged; q.d_crowd = s.d_crowd; q.stream = s.stream; }
    else {
      ITRY(mic_stream_get(&q.stream));
      ITRY(hipMemsetAsync(q.d_cont, 0, (g->cont_cap + 192) * 2, q.stream));     // (the query kernel's read-ahead looks past the last read)
      ITRY(hipStreamSynchronize(q.stream));
    }
  }
  ITRY(hipSetDevice(s.peers[owner].device));
  if (!s.ev_pack) ITRY(mic_event_get(&s.ev_pack, false));
  mic_peer_enable_engines(group, P);
  return MIC_OK;
}

// Every engine of the group probes the batch's packed reads (on the owner's device after pack_kernel) against its part; the rows
// are summed read-range owned; the owner's d_results hold best / second-best of all reads when its stream has passed the waits
// queued here.  Nothing blocks the host.
int group_query_issue(mic_engine* const* group, size_t P, size_t owner, Ingest* g, Slot& s, uint32_t n, uint32_t nb, int k) {
  const size_t rw = kGroupRowWords;
  Slot::Peer& O = s.peers[owner];
  // containers the packer can have written for nb bytes in n reads (record_kernel's reservations) + what the kernel reads ahead
  const size_t cont_n = std::min<size_t>(g->cont_cap + 192, (size_t)nb / 8 + 2 * ((size_t)nb / (size_t)(k + 1)) + 10 * (size_t)n + 128);
  const bool timed = s.timed;
  s.fan_bytes = s.x_bytes = 0;
  ITRY(hipSetDevice(O.device));
  ITRY(hipEventRecord(s.ev_pack, s.stream));
  for (size_t p = 0; p < P; ++p) {
    Slot::Peer& q = s.peers[p];
    MicTable t; int sc, ncu, dev, kk; uint32_t nt;
    int rc = mic_engine_table(group[p], &t, &sc, &ncu, &dev, &kk, &nt);
    if (rc) return rc;
    if (!t.slots) return mic_set_error(MIC_E_STATE, "no database loaded on engine %zu of the group", p);
    ITRY(hipSetDevice(dev));
    if (p != owner) {
      ITRY(hipStreamWaitEvent(q.stream, s.ev_pack, 0));
      if (timed) ITRY(hipEventRecord(q.tv[0], q.stream));
      ITRY(hipMemcpyPeerAsync(q.d_rp, dev, O.d_rp, O.device, ((size_t)n + 2) * 4, q.stream));
      ITRY(hipMemcpyPeerAsync(q.d_cont, dev, O.d_cont, O.device, cont_n * 2, q.stream));
      s.fan_bytes += ((size_t)n + 2) * 4 + cont_n * 2;
    } else if (timed) ITRY(hipEventRecord(q.tv[0], q.stream));
    ITRY(hipMemsetAsync(q.d_flagged, 0, 4, q.stream));
    if (timed) ITRY(hipEventRecord(q.tv[1], q.stream));
    MicQueryArgs qa;
    qa.t = t; qa.reads_ptr = q.d_rp; qa.cont = q.d_cont; qa.n_reads = n; qa.row_words = (uint32_t)rw; qa.results = q.d_res;
    qa.rows = q.d_rows; qa.flagged = q.d_flagged; qa.flagged_cap = kFlaggedCapI;
    mic_crowd_attach(qa, q.d_crowd, g->max_reads / 4 + 64);
    ITRY(mic_launch_query(qa, sc, ncu, q.stream));
    ITRY(hipEventRecord(q.ev_q, q.stream));
    if (timed) ITRY(hipEventRecord(q.tv[2], q.stream));
  }
  for (size_t j = 0; j < P; ++j) {
    Slot::Peer& q = s.peers[j];
    const size_t lo = (size_t)n * j / P, hi = (size_t)n * (j + 1) / P, len = hi - lo;
    ITRY(hipSetDevice(q.device));
    if (timed) {
      // measuring runs: the stream first waits for EVERY engine's rows, so that what tv[3] .. tv[4] brackets is the copies and the
      // merges alone and not the other engines' kernels (the product form below lets a copy start as soon as its source is written)
      for (size_t p = 0; p < P; ++p) if (p != j) ITRY(hipStreamWaitEvent(q.stream, s.peers[p].ev_q, 0));
      ITRY(hipEventRecord(q.tv[3], q.stream));
    }
    if (len) {
      // the running sum: this engine's own rows of the range, then the two spare buffers in turn (the partial rows stay as the
      // kernel wrote them: mic_ingest_fetch_group_rows)
      const uint32_t* cur = q.d_rows + lo * rw;
      uint32_t* spare[2] = {q.d_acc, q.d_acc2};
      int c = 0; size_t got = 0;
      for (size_t p = 0; p < P; ++p) {
        if (p == j) continue;
        uint32_t* in = q.d_gather + got * len * rw;
        if (!timed) ITRY(hipStreamWaitEvent(q.stream, s.peers[p].ev_q, 0));
        ITRY(hipMemcpyPeerAsync(in, q.device, s.peers[p].d_rows + lo * rw, s.peers[p].device, len * rw * 4, q.stream));
        ITRY(mic_launch_merge_rows(cur, in, spare[c], (uint32_t)rw, len, nullptr, q.stream));
        cur = spare[c]; c ^= 1; ++got;
        s.x_bytes += len * rw * 4;
      }
      if (j == owner) ITRY(mic_launch_result_from_rows(cur, (uint32_t)rw, O.d_res + lo * 8, len, q.stream));
      else {
        ITRY(mic_launch_result_from_rows(cur, (uint32_t)rw, q.d_res, len, q.stream));
        ITRY(hipMemcpyPeerAsync(O.d_res + lo * 8, O.device, q.d_res, q.device, len * 32, q.stream));
        s.x_bytes += len * 32;
      }
    }
    if (timed) ITRY(hipEventRecord(q.tv[4], q.stream));
    ITRY(hipEventRecord(q.ev_done, q.stream));
  }
  ITRY(hipSetDevice(O.device));
  for (size_t j = 0; j < P; ++j) if (j != owner) ITRY(hipStreamWaitEvent(s.stream, s.peers[j].ev_done, 0));
  group_overflow_kernel<<<(n + 255) / 256, 256, 0, s.stream>>>(O.d_res, n, s.d_flagged);
  ITRY(hipGetLastError());
  return MIC_OK;
}

int group_query(mic_engine* const* group, size_t P, size_t owner, Ingest* g, Slot& s, uint32_t n, uint32_t nb, int k) {
  // the slot's buffers on the engines of its group are kept from batch to batch while the group is the same: the same engines in the
  // same order, each on the device and with the table it had (an engine destroyed and another created at its address is not the same)
  bool same = s.peers.size() == P && s.group_owner == owner && s.timed == group_timing();
  for (size_t p = 0; same && p < P; ++p) {
    MicTable t; int sc, ncu, dev, kk; uint32_t nt;
    same = s.peers[p].eng == group[p] && mic_engine_table(group[p], &t, &sc, &ncu, &dev, &kk, &nt) == MIC_OK &&
           dev == s.peers[p].device && (const void*)t.slots == s.peers[p].table_id;
  }
  if (!same) { int rc = setup_peers(g, s, group, P, owner); if (rc) { free_peers(s); return rc; } }
  const int rc = group_query_issue(group, P, owner, g, s, n, nb, k);
  if (rc) {
    // an error in the middle leaves work queued on the helpers' streams that reads and writes the slot's buffers: wait for it
    // before the caller reuses or frees them (the error text is kept)
    const std::string msg = mic_last_error();
    for (Slot::Peer& q : s.peers) if (q.stream) { hipSetDevice(q.device); hipStreamSynchronize(q.stream); }
    (void)hipGetLastError();
    hipSetDevice(s.peers[owner].device);
    return mic_set_error(rc, "%s", msg.c_str());
  }
  return MIC_OK;
}

// MIC_GROUP_TIMING: the slot's last table-sharded batch has finished (the owner's stream waited for every engine): its events' times
void group_harvest(Ingest* g, Slot& s, uint32_t n) {
  if (!s.timed || s.peers.empty()) return;
  double fan_sum = 0, fan_max = 0, k_sum = 0, k_max = 0, x_sum = 0, x_max = 0;
  for (size_t p = 0; p < s.peers.size(); ++p) {
    Slot::Peer& q = s.peers[p];
    hipSetDevice(q.device);
    float fan = 0, kq = 0, x = 0;
    if (hipEventElapsedTime(&fan, q.tv[0], q.tv[1]) != hipSuccess || hipEventElapsedTime(&kq, q.tv[1], q.tv[2]) != hipSuccess ||
        hipEventElapsedTime(&x, q.tv[3], q.tv[4]) != hipSuccess) { (void)hipGetLastError(); return; }
    if (p != s.group_owner) { fan_sum += fan; fan_max = std::max(fan_max, (double)fan); }
    k_sum += kq; k_max = std::max(k_max, (double)kq);
    x_sum += x; x_max = std::max(x_max, (double)x);
  }
  hipSetDevice(g->device);
  std::lock_guard<std::mutex> lk(g->gstats.mu);
  GroupStats& G = g->gstats;
  ++G.batches; G.reads += n; G.fan_bytes += s.fan_bytes; G.x_bytes += s.x_bytes;
  G.fan_ms_sum += fan_sum; G.fan_ms_max += fan_max; G.kernel_ms_sum += k_sum; G.kernel_ms_max += k_max; G.x_ms_sum += x_sum; G.x_ms_max += x_max;
}

}  // namespace

struct mic_text {
  const uint8_t* t = nullptr; uint32_t nb = 0;
  void* d_scratch = nullptr; void* d_block = nullptr;
  uint32_t* d_ls = nullptr;
  uint64_t n_rec = 0;
  uint32_t stride = 64;
  std::vector<unsigned long long> samples;      // byte offset of record i * stride, then the size of the text
  int device = 0;
  bool fasta = false;
};

struct mic_pairs {
  PairText t[2];
  void* d_scratch = nullptr; void* d_block = nullptr;     // the two device allocations everything below is carved from
  uint32_t* d_ls[2] = {nullptr, nullptr};
  unsigned long long* d_off = nullptr;          // off[r] = bytes of merged text in front of record r (n_rec + 1 entries)
  uint64_t n_rec = 0;
  uint32_t stride = 64;
  std::vector<unsigned long long> samples;      // off[i * stride] for the host's batch arithmetic, then off[n_rec]
  int device = 0;
};

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
  g->eng = e; g->device = dev;
  g->max_bytes = (max_bytes + ING_TILE - 1) / ING_TILE * ING_TILE;
  g->max_tiles = g->max_bytes / ING_TILE;
  g->max_lines = g->max_bytes / 16 + 64;          // fewer than 16 bytes per line on average: host path
  g->max_reads = g->max_bytes / 32 + 16;          // fewer than 32 bytes per record on average: host path
  // sum of the per-read reservations of record_kernel: nbytes / 8 + 2 (nbytes / (k + 1) + 1) + 8 containers per read
  g->cont_cap = g->max_bytes / 8 + 2 * (g->max_bytes / (size_t)(k + 1)) + 10 * g->max_reads + 64;
  g->csv_cap = g->max_bytes;
  g->crowd_reads = (t.slots && !t.side) ? 0 : g->max_reads / 4 + 64;      // (no table yet: one may come with a side table)
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
    ingest_warm_kernel<<<1, 64>>>((uint32_t*)g->d_tnames);       // (loads this file's device code now, not with the first batch)
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
  return mic_ingest_classify_group(&e, 1, 0, slot_id, n_bytes, flags, out);
}

int mic_ingest_classify_group(mic_engine* const* group, size_t n_group, size_t owner, size_t slot_id, size_t n_bytes, int flags,
                              mic_ingest_result* out) {
  if (!group || n_group == 0 || owner >= n_group) return mic_set_error(MIC_E_INVALID, "bad group");
  for (size_t p = 0; p < n_group; ++p) if (!group[p]) return mic_set_error(MIC_E_INVALID, "null engine in the group");
  mic_engine* e = group[owner];
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
  const bool resident = (flags & MIC_INGEST_RESIDENT) != 0;      // the text is in the slot's device buffer already (mic_pairs_merge_to_slot: merged pairs)
  const uint8_t first = resident ? (uint8_t)((flags & MIC_INGEST_RESIDENT_FASTQ) == MIC_INGEST_RESIDENT_FASTQ ? '@' : '>') : s.h_raw[0];
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
  if (!resident) {
    ITRY(hipMemcpyAsync(s.d_raw, s.h_raw, n_bytes, hipMemcpyHostToDevice, up));
    ITRY(hipEventRecord(s.ev_up, up));
    ITRY(hipStreamWaitEvent(st, s.ev_up, 0));
  }
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
  if (n_group == 1) {
    ITRY(hipMemsetAsync(s.d_flagged, 0, 4, st));
    MicQueryArgs qa;
    qa.t = t; qa.reads_ptr = s.d_rp; qa.cont = s.d_cont; qa.n_reads = n_reads; qa.row_words = 0; qa.results = s.d_results;
    qa.rows = nullptr; qa.flagged = s.d_flagged; qa.flagged_cap = kFlaggedCapI;
    mic_crowd_attach(qa, s.d_crowd, g->max_reads / 4 + 64);
    ITRY(mic_launch_query(qa, sc, ncu, st));
  } else {
    // table-sharded: all engines of the group probe the batch against their parts, the rows are summed read-range owned
    if ((rc = group_query(group, n_group, owner, g, s, n_reads, nb, k))) return rc;
    ITRY(hipSetDevice(dev));
  }
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
  if (n_group > 1) group_harvest(g, s, n_reads);
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

// ---- paired-end texts on the device -------------------------------------------------------------------------------------
#define PTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    rc = mic_set_error(e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

int mic_pairs_free(mic_engine* e, mic_pairs* p) {
  (void)e;
  if (!p) return MIC_OK;
  hipSetDevice(p->device);
  if (p->d_scratch) hipFree(p->d_scratch);
  if (p->d_block) hipFree(p->d_block);
  delete p;
  return MIC_OK;
}

// Two device allocations and no hipFree on the way (each costs about a millisecond and a device-wide wait): a scratch block for the
// line counts of both texts, then - once the numbers of lines are known - one block for everything that depends on them.
int mic_pairs_index_device(mic_engine* e, const void* d_text1, size_t n1, const void* d_text2, size_t n2, mic_pairs** out,
                           uint64_t* n_records, uint32_t* status) {
  if (!e || !d_text1 || !d_text2 || !out || !n_records || !status) return mic_set_error(MIC_E_INVALID, "null argument");
  *out = nullptr; *n_records = 0; *status = 0;
  if (n1 == 0 || n2 == 0 || n1 >= 0xFFFFFF00ull || n2 >= 0xFFFFFF00ull) { *status = PS_BIG; return MIC_OK; }   // 32-bit line starts
  MicTable t; int sc, ncu, dev, k; uint32_t nt;
  int rc = mic_engine_table(e, &t, &sc, &ncu, &dev, &k, &nt);
  if (rc) return rc;
  if (hipSetDevice(dev) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  mic_pairs* p = new mic_pairs;
  p->device = dev;
  const uint8_t* raw[2] = {(const uint8_t*)d_text1, (const uint8_t*)d_text2};
  const uint32_t nb[2] = {(uint32_t)n1, (uint32_t)n2};
  const uint32_t n_tiles[2] = {(nb[0] + ING_TILE - 1) / ING_TILE, (nb[1] + ING_TILE - 1) / ING_TILE};
  uint64_t n_lines[2] = {0, 0};
  uint32_t nl[2] = {0, 0}; uint8_t last[2] = {0, 0};
  hipStream_t st = nullptr;
  uint32_t* d_tile[2]; uint32_t* d_tile_off[2]; void* d_tmp = nullptr; uint32_t* d_facts = nullptr;
  uint32_t facts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t tmp_bytes = 0, tmp2 = 0;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  static const bool timing = getenv("MIC_GZ_TIMING") != nullptr;
  double tl = timing ? now_s() : 0;
  auto lap = [&](const char* what) { if (!timing) return; const double t = now_s(); fprintf(stderr, "[pairs] %s: %.3f ms\n", what, (t - tl) * 1e3); tl = t; };
  { hipStream_t up_, down_; mic_engine_copy_streams(e, &up_, &down_); st = up_; }      // (a stream of its own would cost 2 ms to create)
  {
    size_t t0 = 0, t1 = 0;
    PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, t0, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(n_tiles[0] + 1), st));
    PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(n_tiles[1] + 1), st));
    tmp_bytes = std::max(t0, t1);
    const size_t a0 = up(((size_t)n_tiles[0] + 1) * 4), a1 = up(((size_t)n_tiles[1] + 1) * 4);
    PTRY(hipMalloc(&p->d_scratch, 2 * a0 + 2 * a1 + 256 + up(tmp_bytes + 16)));
    char* q = (char*)p->d_scratch;
    d_tile[0] = (uint32_t*)q; q += a0; d_tile_off[0] = (uint32_t*)q; q += a0;
    d_tile[1] = (uint32_t*)q; q += a1; d_tile_off[1] = (uint32_t*)q; q += a1;
    d_facts = (uint32_t*)q; q += 256;
    d_tmp = q;
  }
  lap("scratch allocated");
  for (int i = 0; i < 2; ++i) {
    line_count_kernel<<<n_tiles[i], 256, 0, st>>>(raw[i], nb[i], d_tile[i]);
    PTRY(hipMemsetAsync(d_tile[i] + n_tiles[i], 0, 4, st));
    size_t tb = tmp_bytes;
    PTRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_tile[i], d_tile_off[i], (int)(n_tiles[i] + 1), st));
    text_facts_kernel<<<1, 1, 0, st>>>(raw[i], nb[i], d_tile_off[i], n_tiles[i], d_facts + 4 * i);
  }
  PTRY(hipGetLastError());
  PTRY(hipMemcpyAsync(facts, d_facts, 32, hipMemcpyDeviceToHost, st));
  PTRY(hipStreamSynchronize(st));
  for (int i = 0; i < 2; ++i) { nl[i] = facts[4 * i]; last[i] = (uint8_t)facts[4 * i + 1]; }
  lap("lines counted");
  for (int i = 0; i < 2; ++i) n_lines[i] = (uint64_t)nl[i] + (last[i] != '\n' ? 1 : 0);
  if (n_lines[0] != n_lines[1] || n_lines[0] % 4 != 0 || n_lines[0] == 0) { *status = PS_LINES; goto done; }
  {
    const uint64_t n_rec = n_lines[0] / 4;
    p->n_rec = n_rec;
    const uint64_t n_samples = n_rec / p->stride + 2;
    PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)(n_rec + 1), st));
    const size_t b_ls = up((n_lines[0] + 2) * 4), b_off = up((n_rec + 1) * 8), b_smp = up(n_samples * 8);
    PTRY(hipMalloc(&p->d_block, 2 * b_ls + 2 * b_off + b_smp + 256 + up(tmp2 + 16)));
    char* q = (char*)p->d_block;
    p->d_ls[0] = (uint32_t*)q; q += b_ls; p->d_ls[1] = (uint32_t*)q; q += b_ls;
    p->d_off = (unsigned long long*)q; q += b_off;
    unsigned long long* d_mlen = (unsigned long long*)q; q += b_off;
    unsigned long long* d_samples = (unsigned long long*)q; q += b_smp;
    uint32_t* d_status = (uint32_t*)q; q += 256;
    void* d_tmp2 = q;
    lap("index block allocated");
    for (int i = 0; i < 2; ++i) {
      line_start_kernel<<<n_tiles[i], 256, 0, st>>>(raw[i], nb[i], d_tile_off[i], p->d_ls[i], (uint32_t)std::min<uint64_t>(n_lines[i] + 2, 0xFFFFFFFFull));
      PTRY(hipGetLastError());
      if (last[i] != '\n') {                   // the virtual line end of an unterminated last line (lines_finish_kernel's convention)
        nl[i] = nb[i] + 1;                     // (nl[] is done with; the source of an asynchronous copy has to outlive it)
        PTRY(hipMemcpyAsync(p->d_ls[i] + n_lines[i], &nl[i], 4, hipMemcpyHostToDevice, st));
      }
      p->t[i].t = raw[i]; p->t[i].ls = p->d_ls[i]; p->t[i].nb = nb[i];
    }
    PTRY(hipMemsetAsync(d_status, 0, 4, st));
    pair_len_kernel<<<(unsigned)((n_rec + 1 + 255) / 256), 256, 0, st>>>(p->t[0], p->t[1], n_rec, d_mlen, d_status);
    PTRY(hipGetLastError());
    PTRY(hipcub::DeviceScan::ExclusiveSum(d_tmp2, tmp2, d_mlen, p->d_off, (int)(n_rec + 1), st));
    p->samples.resize(n_samples);
    pair_sample_kernel<<<(unsigned)((n_samples + 255) / 256), 256, 0, st>>>(p->d_off, n_rec, p->stride, n_samples, d_samples);
    PTRY(hipGetLastError());
    PTRY(hipMemcpyAsync(p->samples.data(), d_samples, n_samples * 8, hipMemcpyDeviceToHost, st));
    PTRY(hipMemcpyAsync(status, d_status, 4, hipMemcpyDeviceToHost, st));
    PTRY(hipStreamSynchronize(st));
    lap("line starts, pair checks, offsets, samples");
  }
done:
  if (st) hipStreamSynchronize(st);
  if (rc != MIC_OK || *status) { mic_pairs_free(e, p); return rc; }
  *out = p; *n_records = p->n_rec;
  return MIC_OK;
}

// bytes of merged text in front of record r, for r a multiple of the stride or r == n_records
static bool pairs_offset(const mic_pairs* p, uint64_t r, unsigned long long& off) {
  if (r == p->n_rec) { off = p->samples.back() ; return true; }
  if (r > p->n_rec || r % p->stride) return false;
  off = p->samples[r / p->stride];
  return true;
}

int mic_pairs_offsets(const mic_pairs* p, const uint64_t** samples, size_t* n_samples, uint32_t* stride) {
  if (!p || !samples || !n_samples || !stride) return mic_set_error(MIC_E_INVALID, "null argument");
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
  *samples = (const uint64_t*)p->samples.data(); *n_samples = p->samples.size(); *stride = p->stride;
  return MIC_OK;
}

int mic_pairs_merge_to_slot(mic_engine* e, mic_pairs* p, uint64_t r0, uint64_t r1, size_t slot_id, size_t* n_bytes) {
  if (!e || !p || !n_bytes) return mic_set_error(MIC_E_INVALID, "null argument");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(e);
  if (!g || slot_id >= g->slots.size()) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  unsigned long long o0, o1;
  if (r0 >= r1 || !pairs_offset(p, r0, o0) || !pairs_offset(p, r1, o1))
    return mic_set_error(MIC_E_INVALID, "records [%llu, %llu): not a range of whole strides", (unsigned long long)r0, (unsigned long long)r1);
  if (o1 - o0 > g->max_bytes) return mic_set_error(MIC_E_INVALID, "merged text of %llu bytes does not fit the slot (%zu)", o1 - o0, g->max_bytes);
  // the kernel runs on the SLOT's device and stream; when the texts live on another device (several engines, one inflated input)
  // it reads them through peer access
  if (g->device != p->device && !mic_peer_enable(g->device, p->device))
    return mic_set_error(MIC_E_UNSUPPORTED, "device %d has no peer access to device %d, where the inflated text lives", g->device, p->device);
  ITRY(hipSetDevice(g->device));
  Slot& s = g->slots[slot_id];
  pair_merge_kernel<<<(unsigned)((r1 - r0 + 3) / 4), 256, 0, s.stream>>>(p->t[0], p->t[1], r0, r1, p->d_off, s.d_raw);
  ITRY(hipGetLastError());
  *n_bytes = (size_t)(o1 - o0);
  return MIC_OK;
}

int mic_pairs_text(mic_engine* e, mic_pairs* p, uint64_t r0, uint64_t r1, void* host_dst, size_t cap, size_t* n_bytes) {
  if (!e || !p || !host_dst || !n_bytes) return mic_set_error(MIC_E_INVALID, "null argument");
  unsigned long long o0, o1;
  if (r0 >= r1 || !pairs_offset(p, r0, o0) || !pairs_offset(p, r1, o1))
    return mic_set_error(MIC_E_INVALID, "records [%llu, %llu): not a range of whole strides", (unsigned long long)r0, (unsigned long long)r1);
  *n_bytes = (size_t)(o1 - o0);
  if (*n_bytes > cap) return mic_set_error(MIC_E_INVALID, "merged text of %zu bytes does not fit the buffer (%zu)", *n_bytes, cap);
  ITRY(hipSetDevice(p->device));
  uint8_t* d = nullptr;
  ITRY(hipMalloc(&d, *n_bytes + 16));
  pair_merge_kernel<<<(unsigned)((r1 - r0 + 3) / 4), 256, 0, 0>>>(p->t[0], p->t[1], r0, r1, p->d_off, d);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipMemcpy(host_dst, d, *n_bytes, hipMemcpyDeviceToHost);
  hipFree(d);
  ITRY(he);
  return MIC_OK;
}

// ---- one FASTQ text on the device (the inflated file of -O reads.fq.gz): where its records start ---------------------------
int mic_text_free(mic_engine* e, mic_text* p) {
  (void)e;
  if (!p) return MIC_OK;
  hipSetDevice(p->device);
  if (p->d_scratch) hipFree(p->d_scratch);
  if (p->d_block) hipFree(p->d_block);
  delete p;
  return MIC_OK;
}

// partial: the text is the front of a longer one (a stripe of a member still being inflated, mic_gz_stream_next) - FASTQ only, the
// whole records of it are indexed (lines that end in '\n', four to a record), *n_used = where the first record that is not whole
// starts; no whole record yet: MIC_OK, *out = nullptr, *n_used = 0
static int text_index(mic_engine* e, const void* d_text, size_t n, bool partial, mic_text** out, uint64_t* n_records, uint64_t* n_used, uint32_t* status);

int mic_text_index_device(mic_engine* e, const void* d_text, size_t n, mic_text** out, uint64_t* n_records, uint32_t* status) {
  uint64_t used = 0;
  return text_index(e, d_text, n, false, out, n_records, &used, status);
}

int mic_text_index_front_device(mic_engine* e, const void* d_text, size_t n, mic_text** out, uint64_t* n_records, uint64_t* n_used, uint32_t* status) {
  if (!n_used) return mic_set_error(MIC_E_INVALID, "null argument");
  return text_index(e, d_text, n, true, out, n_records, n_used, status);
}

static int text_index(mic_engine* e, const void* d_text, size_t n, bool partial, mic_text** out, uint64_t* n_records, uint64_t* n_used, uint32_t* status) {
  if (!e || !d_text || !out || !n_records || !status) return mic_set_error(MIC_E_INVALID, "null argument");
  *out = nullptr; *n_records = 0; *status = 0; *n_used = 0;
  if (n == 0 || n >= 0xFFFFFF00ull) { *status = PS_BIG; return MIC_OK; }
  MicTable t; int sc, ncu, dev, k; uint32_t nt;
  int rc = mic_engine_table(e, &t, &sc, &ncu, &dev, &k, &nt);
  if (rc) return rc;
  if (hipSetDevice(dev) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  mic_text* p = new mic_text;
  // the kernels read 16 aligned bytes a thread: the text counts from the 4-KiB boundary in front of it, `from` bytes are not its own
  const uint32_t from = (uint32_t)((uintptr_t)d_text & 4095u);
  if (n + from >= 0xFFFFFF00ull) { delete p; *status = PS_BIG; return MIC_OK; }
  p->device = dev; p->t = (const uint8_t*)d_text - from; p->nb = (uint32_t)(n + from);
  const uint8_t* raw = p->t;
  const uint32_t nb = p->nb, n_tiles = (nb + ING_TILE - 1) / ING_TILE;
  uint32_t nl = 0; uint8_t first = 0, last = 0;
  uint64_t n_lines = 0;
  hipStream_t st = nullptr;
  uint32_t* d_tile = nullptr; uint32_t* d_tile_off = nullptr; void* d_tmp = nullptr; uint32_t* d_facts = nullptr;
  uint32_t facts[4] = {0, 0, 0, 0};
  size_t tmp_bytes = 0;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  { hipStream_t up_, down_; mic_engine_copy_streams(e, &up_, &down_); st = up_; }      // (a stream of its own would cost 2 ms to create)
  PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(n_tiles + 1), st));
  {
    const size_t a0 = up(((size_t)n_tiles + 1) * 4);
    PTRY(hipMalloc(&p->d_scratch, 2 * a0 + 256 + up(tmp_bytes + 16)));
    char* q = (char*)p->d_scratch;
    d_tile = (uint32_t*)q; q += a0; d_tile_off = (uint32_t*)q; q += a0; d_facts = (uint32_t*)q; q += 256; d_tmp = q;
  }
  line_count_kernel<<<n_tiles, 256, 0, st>>>(raw, nb, d_tile, from);
  PTRY(hipMemsetAsync(d_tile + n_tiles, 0, 4, st));
  {
    size_t tb = tmp_bytes;
    PTRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_tile, d_tile_off, (int)(n_tiles + 1), st));
  }
  text_facts_kernel<<<1, 1, 0, st>>>(raw, nb, d_tile_off, n_tiles, d_facts, from);
  PTRY(hipGetLastError());
  PTRY(hipMemcpyAsync(facts, d_facts, 16, hipMemcpyDeviceToHost, st));
  PTRY(hipStreamSynchronize(st));
  nl = facts[0]; last = (uint8_t)facts[1]; first = (uint8_t)facts[2];
  n_lines = (uint64_t)nl + (last != '\n' ? 1 : 0);
  if (partial) {
    if (first != '@') { *status = PS_HEADER; goto done; }
    n_lines = (uint64_t)nl / 4 * 4;              // whole records of whole lines; what follows is the next call's
    last = '\n';
    if (n_lines == 0) { mic_text_free(e, p); return MIC_OK; }
  }
  if (first == '>') {
    // FASTA: a record is a '>' line and what follows it up to the next one (sequences over several lines)
    p->fasta = true;
    const size_t b_ls = up((n_lines + 2) * 4), b_fl = up((n_lines + 1) * 4);
    size_t tmp2 = 0;
    PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(n_lines + 1), st));
    if (n_lines >= 0x7FFFFFF0ull) { *status = PS_BIG; goto done; }
    // (the samples are sized by the lines - there are no more records than lines - and cut down once the count is known)
    const size_t b_smp = up((n_lines / p->stride + 2) * 8);
    PTRY(hipMalloc(&p->d_block, b_ls + 2 * b_fl + b_smp + up(tmp2 + 16)));
    p->d_ls = (uint32_t*)p->d_block;
    uint32_t* d_flag = (uint32_t*)((char*)p->d_block + b_ls);
    uint32_t* d_rec = (uint32_t*)((char*)p->d_block + b_ls + b_fl);
    unsigned long long* d_samples = (unsigned long long*)((char*)p->d_block + b_ls + 2 * b_fl);
    void* d_tmp2 = (char*)p->d_block + b_ls + 2 * b_fl + b_smp;
    line_start_kernel<<<n_tiles, 256, 0, st>>>(raw, nb, d_tile_off, p->d_ls, (uint32_t)(n_lines + 2), from);
    PTRY(hipGetLastError());
    if (last != '\n') { nl = nb + 1; PTRY(hipMemcpyAsync(p->d_ls + n_lines, &nl, 4, hipMemcpyHostToDevice, st)); }
    const unsigned gl = (unsigned)((n_lines + 1 + 255) / 256);
    text_fasta_flag_kernel<<<gl, 256, 0, st>>>(raw, nb, p->d_ls, n_lines, d_flag);
    PTRY(hipcub::DeviceScan::ExclusiveSum(d_tmp2, tmp2, d_flag, d_rec, (int)(n_lines + 1), st));
    uint32_t n_rec32 = 0;
    PTRY(hipMemcpyAsync(&n_rec32, d_rec + n_lines, 4, hipMemcpyDeviceToHost, st));
    text_fasta_sample_kernel<<<gl, 256, 0, st>>>(p->d_ls, d_flag, d_rec, n_lines, p->stride, d_samples);
    PTRY(hipGetLastError());
    PTRY(hipStreamSynchronize(st));
    p->n_rec = n_rec32;
    if (n_rec32 == 0) { *status = PS_HEADER; goto done; }
    const uint64_t n_samples = p->n_rec / p->stride + 2;
    p->samples.assign(n_samples, (unsigned long long)nb);
    const uint64_t have = (p->n_rec + p->stride - 1) / p->stride;          // samples the kernel wrote: records 0, stride, ...
    PTRY(hipMemcpy(p->samples.data(), d_samples, have * 8, hipMemcpyDeviceToHost));
    goto done;
  }
  if (first != '@') { *status = PS_HEADER; goto done; }                       // (anything else: the host path)
  if (n_lines % 4 != 0 || n_lines == 0) { *status = PS_LINES; goto done; }
  {
    const uint64_t n_rec = n_lines / 4;
    p->n_rec = n_rec;
    const uint64_t n_samples = n_rec / p->stride + 2;
    const size_t b_ls = up((n_lines + 2) * 4), b_smp = up(n_samples * 8);
    PTRY(hipMalloc(&p->d_block, b_ls + b_smp));
    p->d_ls = (uint32_t*)p->d_block;
    unsigned long long* d_samples = (unsigned long long*)((char*)p->d_block + b_ls);
    line_start_kernel<<<n_tiles, 256, 0, st>>>(raw, nb, d_tile_off, p->d_ls, (uint32_t)std::min<uint64_t>(n_lines + 2, 0xFFFFFFFFull), from);
    PTRY(hipGetLastError());
    if (last != '\n') { nl = nb + 1; PTRY(hipMemcpyAsync(p->d_ls + n_lines, &nl, 4, hipMemcpyHostToDevice, st)); }
    p->samples.resize(n_samples);
    text_sample_kernel<<<(unsigned)((n_samples + 255) / 256), 256, 0, st>>>(p->d_ls, nb, n_rec, p->stride, n_samples, d_samples);
    PTRY(hipGetLastError());
    PTRY(hipMemcpyAsync(p->samples.data(), d_samples, n_samples * 8, hipMemcpyDeviceToHost, st));
    PTRY(hipStreamSynchronize(st));
    *n_used = p->samples.back() - from;
  }
done:
  if (st) hipStreamSynchronize(st);
  if (rc != MIC_OK || *status) { mic_text_free(e, p); return rc; }
  *out = p; *n_records = p->n_rec;
  return MIC_OK;
}

static bool text_offset(const mic_text* p, uint64_t r, unsigned long long& off) {
  if (r == p->n_rec) { off = p->samples.back(); return true; }
  if (r > p->n_rec || r % p->stride) return false;
  off = p->samples[r / p->stride];
  return true;
}

int mic_text_format(const mic_text* p) { return p ? (p->fasta ? '>' : '@') : 0; }

int mic_text_offsets(const mic_text* p, const uint64_t** samples, size_t* n_samples, uint32_t* stride) {
  if (!p || !samples || !n_samples || !stride) return mic_set_error(MIC_E_INVALID, "null argument");
  *samples = (const uint64_t*)p->samples.data(); *n_samples = p->samples.size(); *stride = p->stride;
  return MIC_OK;
}

int mic_text_to_slot(mic_engine* e, mic_text* p, uint64_t r0, uint64_t r1, size_t slot_id, size_t* n_bytes) {
  if (!e || !p || !n_bytes) return mic_set_error(MIC_E_INVALID, "null argument");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(e);
  if (!g || slot_id >= g->slots.size()) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  unsigned long long o0, o1;
  if (r0 >= r1 || !text_offset(p, r0, o0) || !text_offset(p, r1, o1))
    return mic_set_error(MIC_E_INVALID, "records [%llu, %llu): not a range of whole strides", (unsigned long long)r0, (unsigned long long)r1);
  if (o1 - o0 > g->max_bytes) return mic_set_error(MIC_E_INVALID, "text of %llu bytes does not fit the slot (%zu)", o1 - o0, g->max_bytes);
  ITRY(hipSetDevice(g->device));
  Slot& s = g->slots[slot_id];
  if (g->device != p->device) {
    mic_peer_enable(g->device, p->device);      // (without peer access the runtime stages the copy through host memory)
    ITRY(hipMemcpyPeerAsync(s.d_raw, g->device, p->t + o0, p->device, (size_t)(o1 - o0), s.stream));
  } else ITRY(hipMemcpyAsync(s.d_raw, p->t + o0, (size_t)(o1 - o0), hipMemcpyDeviceToDevice, s.stream));
  *n_bytes = (size_t)(o1 - o0);
  return MIC_OK;
}

int mic_text_copy(mic_engine* e, mic_text* p, uint64_t r0, uint64_t r1, void* host_dst, size_t cap, size_t* n_bytes) {
  if (!e || !p || !host_dst || !n_bytes) return mic_set_error(MIC_E_INVALID, "null argument");
  unsigned long long o0, o1;
  if (r0 >= r1 || !text_offset(p, r0, o0) || !text_offset(p, r1, o1))
    return mic_set_error(MIC_E_INVALID, "records [%llu, %llu): not a range of whole strides", (unsigned long long)r0, (unsigned long long)r1);
  *n_bytes = (size_t)(o1 - o0);
  if (*n_bytes > cap) return mic_set_error(MIC_E_INVALID, "text of %zu bytes does not fit the buffer (%zu)", *n_bytes, cap);
  ITRY(hipSetDevice(p->device));
  ITRY(hipMemcpy(host_dst, p->t + o0, *n_bytes, hipMemcpyDeviceToHost));
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

int mic_ingest_fetch_group_rows(mic_engine* owner, size_t slot_id, size_t part, uint32_t* rows, size_t cap_words, uint64_t* n_reads,
                                uint32_t* row_words) {
  if (!owner) return mic_set_error(MIC_E_INVALID, "null engine");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(owner);
  if (!g || slot_id >= g->slots.size()) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  Slot& s = g->slots[slot_id];
  if (part >= s.peers.size()) return mic_set_error(MIC_E_STATE, "the slot's last batch was not table-sharded over %zu parts", part + 1);
  if (n_reads) *n_reads = s.n_reads;
  if (row_words) *row_words = kGroupRowWords;
  if (rows) {
    const size_t words = (size_t)s.n_reads * kGroupRowWords;
    if (cap_words < words) return mic_set_error(MIC_E_INVALID, "rows capacity too small");
    ITRY(hipSetDevice(s.peers[part].device));
    ITRY(hipMemcpy(rows, s.peers[part].d_rows, words * 4, hipMemcpyDeviceToHost));
    ITRY(hipSetDevice(g->device));
  }
  return MIC_OK;
}

int mic_ingest_group_stats(mic_engine* owner, double* out, size_t cap) {
  if (!owner || !out || cap < MIC_GROUP_STATS_FIELDS) return mic_set_error(MIC_E_INVALID, "bad argument");
  Ingest* g = (Ingest*)*mic_engine_ingest_slot(owner);
  if (!g) return mic_set_error(MIC_E_STATE, "ingest slots are not allocated");
  std::lock_guard<std::mutex> lk(g->gstats.mu);
  const GroupStats& G = g->gstats;
  const double v[MIC_GROUP_STATS_FIELDS] = {(double)G.batches, (double)G.reads, (double)G.fan_bytes, G.fan_ms_sum, G.fan_ms_max, G.kernel_ms_sum,
                                            G.kernel_ms_max, (double)G.x_bytes, G.x_ms_sum, G.x_ms_max};
  for (size_t i = 0; i < MIC_GROUP_STATS_FIELDS; ++i) out[i] = v[i];
  return (int)MIC_GROUP_STATS_FIELDS;
}

// host build of the device formatter (tests pin it against the C library's "%g")
int mic_format_ratio_g(uint32_t num, uint32_t den, char* out16) {
  if (!out16 || den == 0 || num == 0 || num > den) return MIC_E_INVALID;
  const int n = mic_fmt_g_unit((double)num / (double)den, out16);
  out16[n] = 0;
  return n;
}

}  // extern "C"
