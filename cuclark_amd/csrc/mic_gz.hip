// mic_gz.hip - one gzip member inflated ON THE DEVICE: the two-stage scheme of csrc/pgz.hpp (pugz / rapidgzip)
// with thousands of decode units instead of sixteen threads.  BASELINE config 5 names gzip FASTQ; the reference's scripts gunzip
// to a temporary file first (classify_metagenome.sh:116-142).  A job that is allowed 16 CPUs inflates 3 GB/s of text with all of
// them (pgz.hpp); the device decodes one deflate block per wavefront, ~7 000 of them at once.
//
//   1  gz_find_kernel     one wavefront per 16 KiB of compressed data: the first bit offset at which a block with dynamic codes
//                         starts - 64 offsets per step through the cheap tests (BFINAL / BTYPE bits, HLIT / HDIST in range, the
//                         code-length code exactly complete), the survivors one by one through the whole header (every code
//                         complete, an end-of-block code) and 300 symbols of trial decode;
//   2  gz_decode_kernel   one wavefront per unit = from one found start to the first block boundary at or behind the next found
//                         start: first a counting pass (symbols out, where it ended), the host stitches the units into a chain
//                         (a unit whose start lies inside the unit in front of it was a false find and is dropped; a gap or an
//                         error gives the file back to the caller's CPU inflater), then the same decode writing 16-bit symbols:
//                         a byte, or a MARKER 0x8000 | i for a back-reference to position i of the 32 KiB in front of the unit;
//   3  gz_window_kernel   in chain order the last 32 KiB of every unit are resolved against the window handed on (one block,
//                         the window in LDS) and every unit's incoming window is kept;
//   4  gz_resolve_kernel  all units at once: markers replaced, symbols narrowed to bytes at the unit's offset of the text.
// The member's length is checked here (ISIZE); its CRC-32 is returned for the caller, who checks it on the copy it takes
// (classifier.cpp, pgz::crc32_fast).  Stored and fixed-code blocks are decoded; several members, a preset dictionary or anything
// that does not stitch: MIC_E_UNSUPPORTED, and the caller inflates on the CPU as before - a wrong speculation cannot pass.
#include "mi_clark.h"
#include "mic_internal.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

// engine services (mic_engine.hip)
int mic_engine_table(mic_engine* e, MicTable* t, int* slot_class, int* n_cu, int* device, int* k, uint32_t* n_targets);
int mic_set_error(int code, const char* fmt, ...);
void mic_engine_copy_streams(mic_engine* e, hipStream_t* up, hipStream_t* down);

namespace {

constexpr uint32_t GZ_CHUNK = 16384;      // compressed bytes per finder chunk
constexpr int GZ_TRIAL = 300;             // symbols of trial decode behind a candidate header

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- bits, LSB first.  The data is followed by 16 readable zero bytes; positions behind the data read zeros and set `over`.
struct Bits {
  const uint8_t* p; uint64_t n;           // n bytes of data (p[n .. n+15] readable)
  uint64_t pos;                           // next byte to load
  uint64_t buf; int cnt; bool over;
  __device__ void init(const uint8_t* d, uint64_t len, uint64_t bitpos) {
    p = d; n = len; pos = bitpos >> 3; buf = 0; cnt = 0; over = false;
    refill();
    const int skip = (int)(bitpos & 7);
    buf >>= skip; cnt -= skip;
  }
  __device__ void refill() {
    uint64_t v = 0;
    if (pos + 8 <= n + 16) {
      const uint64_t q = pos <= n + 8 ? pos : n + 8;
      // unaligned eight bytes: two aligned pairs of dwords would need the alignment of p; bytes are cached, this is not the hot part
      const uint8_t* s = p + q;
      v = (uint64_t)s[0] | ((uint64_t)s[1] << 8) | ((uint64_t)s[2] << 16) | ((uint64_t)s[3] << 24) | ((uint64_t)s[4] << 32) |
          ((uint64_t)s[5] << 40) | ((uint64_t)s[6] << 48) | ((uint64_t)s[7] << 56);
      if (pos > n + 8) v = 0;
    }
    if (pos >= n + 8) over = true;
    buf |= v << cnt;
    const int add = (63 - cnt) >> 3;
    pos += (uint64_t)add; cnt += add * 8;
  }
  __device__ uint32_t peek(int k) { if (cnt < k) refill(); return (uint32_t)(buf & ((1ull << k) - 1)); }
  __device__ void drop(int k) { buf >>= k; cnt -= k; }
  __device__ uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }
  __device__ void skip(int k) { if (cnt < k) refill(); drop(k); }
  __device__ uint64_t bitpos() const { return pos * 8 - (uint64_t)cnt; }
  __device__ void align() { drop(cnt & 7); }
};

// ---- canonical Huffman code in the count / symbol form (puff.c): a few hundred bytes of LDS per code
struct Huff {
  uint16_t count[16];
  uint16_t symbol[288];
  // 0 = complete, 1 = incomplete, -1 = over-subscribed
  __device__ int build(const uint8_t* len, int n) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int i = 0; i < n; ++i) ++count[len[i] & 15];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return -1; }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i) if (len[i] & 15) { const uint16_t o = offs[len[i] & 15]++; if (o < 288) symbol[o] = (uint16_t)i; }
    return left > 0 ? 1 : 0;
  }
  __device__ int decode(Bits& b) const {
    const uint32_t v = b.peek(15);
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
      code |= (int)((v >> (l - 1)) & 1u);
      const int c = count[l];
      if (code - c < first) { b.drop(l); const int at = index + (code - first); return at < 288 ? symbol[at] : -1; }
      index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
  }
};

struct Scratch {            // per wavefront, LDS
  Huff lit, dist;
  uint8_t len[320 + 8];
  uint8_t cl[20];
  uint8_t pre[128];
};

__device__ uint32_t kraft(const uint8_t* len, int n) {
  uint32_t sum = 0;
  for (int i = 0; i < n; ++i) if (len[i]) sum += 1u << (15 - (len[i] & 15));
  return sum;
}

// header of a block with dynamic codes (behind BFINAL / BTYPE) -> the two codes; strict: what a real encoder emits (pgz.hpp)
__device__ bool read_dynamic(Bits& b, Scratch& s, bool strict) {
  const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  for (int i = 0; i < 19; ++i) s.cl[i] = 0;
  for (int i = 0; i < hclen; ++i) s.cl[c_cl_order[i]] = (uint8_t)b.get(3);
  const uint32_t ks = kraft(s.cl, 19);
  if (ks > (1u << 15)) return false;
  if (ks < (1u << 15)) {
    if (strict) return false;
    int used = 0; for (int i = 0; i < 19; ++i) used += s.cl[i] != 0;
    if (used != 1) return false;
  }
  for (int i = 0; i < 128; ++i) s.pre[i] = 0;
  {
    uint32_t code = 0;
    for (int l = 1; l <= 7; ++l) {
      for (int sym = 0; sym < 19; ++sym) {
        if (s.cl[sym] != l) continue;
        uint32_t r = 0;
        for (int q = 0; q < l; ++q) r |= ((code >> q) & 1u) << (l - 1 - q);
        for (uint32_t v = r; v < 128; v += 1u << l) s.pre[v] = (uint8_t)((l << 5) | sym);
        ++code;
      }
      code <<= 1;
    }
  }
  int n = 0;
  const int total = hlit + hdist;
  while (n < total) {
    const uint8_t e = s.pre[b.peek(7)];
    if (!e) return false;
    b.drop(e >> 5);
    const int sy = e & 31;
    if (sy < 16) { s.len[n++] = (uint8_t)sy; continue; }
    int rep, val = 0;
    if (sy == 16) { if (n == 0) return false; val = s.len[n - 1]; rep = 3 + (int)b.get(2); }
    else if (sy == 17) rep = 3 + (int)b.get(3);
    else rep = 11 + (int)b.get(7);
    if (n + rep > total) return false;
    while (rep--) s.len[n++] = (uint8_t)val;
  }
  if (b.over || s.len[256] == 0) return false;
  if (kraft(s.len, hlit) != (1u << 15)) return false;
  const uint32_t kd = kraft(s.len + hlit, hdist);
  if (kd > (1u << 15)) return false;
  if (kd < (1u << 15)) {
    int used = 0; for (int i = 0; i < hdist; ++i) used += s.len[hlit + i] != 0;
    if (used > 1) return false;
  }
  s.lit.build(s.len, hlit);
  s.dist.build(s.len + hlit, hdist);
  return true;
}

__device__ void fixed_codes(Scratch& s) {
  for (int i = 0; i < 144; ++i) s.len[i] = 8;
  for (int i = 144; i < 256; ++i) s.len[i] = 9;
  for (int i = 256; i < 280; ++i) s.len[i] = 7;
  for (int i = 280; i < 288; ++i) s.len[i] = 8;
  s.lit.build(s.len, 288);
  for (int i = 0; i < 30; ++i) s.len[i] = 5;
  s.dist.build(s.len, 30);
}

// ---- 1: block finder ----------------------------------------------------------------------------------------------------
__device__ bool candidate(const uint8_t* d, uint64_t n, uint64_t bit) {      // the cheap tests, every lane its own offset
  const uint64_t byte = bit >> 3;
  if (byte + 12 >= n) return false;
  // 3 + 14 + 19 x 3 = 74 bits from `bit` on
  const uint8_t* s = d + byte;
  uint64_t lo = 0, hi = 0;
  for (int i = 0; i < 8; ++i) lo |= (uint64_t)s[i] << (8 * i);
  for (int i = 0; i < 4; ++i) hi |= (uint64_t)s[8 + i] << (8 * i);
  const int sh = (int)(bit & 7);
  uint64_t a = (lo >> sh) | (sh ? hi << (64 - sh) : 0);                        // bits 0..63 from `bit`
  const uint64_t b2 = hi >> sh;                                                 // bits 64.. from `bit`
  if ((a & 7u) != 4u) return false;                                             // BFINAL = 0, BTYPE = 2
  const uint32_t hlit = (uint32_t)(a >> 3) & 31u, hdist = (uint32_t)(a >> 8) & 31u, hclen = ((uint32_t)(a >> 13) & 15u) + 4u;
  if (hlit > 29u || hdist > 29u) return false;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < hclen; ++i) {
    const uint32_t at = 17u + 3u * i;                                           // bit offset of this 3-bit length
    const uint32_t l = at + 3u <= 64u ? (uint32_t)(a >> at) & 7u
                                     : (at >= 64u ? (uint32_t)(b2 >> (at - 64u)) & 7u : (uint32_t)((a >> at) | (b2 << (64u - at))) & 7u);
    if (l) sum += 1u << (15 - l);
  }
  return sum == (1u << 15);
}

__global__ void __launch_bounds__(64) gz_find_kernel(const uint8_t* __restrict__ d, uint64_t n, uint64_t first_bit, uint32_t n_chunks,
                                                    unsigned long long* __restrict__ start) {
  __shared__ Scratch sc;
  const uint32_t c = blockIdx.x;
  const int lane = threadIdx.x;
  if (c >= n_chunks) return;
  if (c == 0) { if (lane == 0) start[0] = first_bit; return; }
  const uint64_t from = (uint64_t)c * GZ_CHUNK * 8, to = (uint64_t)(c + 1) * GZ_CHUNK * 8 < n * 8 ? (uint64_t)(c + 1) * GZ_CHUNK * 8 : n * 8;
  unsigned long long found = ~0ull;
  for (uint64_t base = from > first_bit ? from : first_bit + 1; base < to && found == ~0ull; base += 64) {
    const uint64_t bit = base + (uint64_t)lane;
    const bool cand = bit < to && candidate(d, n, bit);
    unsigned long long m = __ballot(cand);
    while (m && found == ~0ull) {
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      const uint64_t at = base + (uint64_t)l;
      int good = 0;
      if (lane == 0) {
        Bits b; b.init(d, n, at + 3);
        good = read_dynamic(b, sc, true) ? 1 : 0;
        for (int i = 0; i < GZ_TRIAL && good; ++i) {
          if (b.over) { good = 0; break; }
          int s = sc.lit.decode(b);
          if (s < 0) { good = 0; break; }
          if (s < 256) continue;
          if (s == 256) break;
          s -= 257;
          if (s >= 29) { good = 0; break; }
          b.skip(c_len_extra[s]);
          const int ds = sc.dist.decode(b);
          if (ds < 0 || ds >= 30) { good = 0; break; }
          b.skip(c_dist_extra[ds]);
        }
      }
      good = __shfl(good, 0);
      if (good) found = at;
    }
  }
  if (lane == 0) start[c] = found;
}

// ---- 2: decode units ------------------------------------------------------------------------------------------------------
struct GzUnit {
  unsigned long long start_bit, stop_bit;      // decode from start_bit to the first block boundary at or behind stop_bit
  unsigned long long out_off;                  // WRITE pass: where its symbols go (in symbols)
  unsigned long long end_bit;                  // result: where it stopped
  unsigned long long n_sym;                    // result: symbols produced
  uint32_t status;                             // result: 0 ok, 1 ended with the member's last block, >= 2 error
  uint32_t pad;
};
enum { GZ_OK = 0, GZ_FINAL = 1, GZ_ERR_CODE = 2, GZ_ERR_OVER = 3, GZ_ERR_STORED = 4, GZ_ERR_DIST = 5, GZ_ERR_TYPE = 6, GZ_ERR_ROOM = 7 };

template <bool WRITE>
__global__ void __launch_bounds__(64) gz_decode_kernel(const uint8_t* __restrict__ d, uint64_t n, GzUnit* __restrict__ units, uint32_t n_units,
                                                      uint16_t* __restrict__ sym, int first_is_file_start) {
  __shared__ Scratch sc;
  const uint32_t u = blockIdx.x;
  if (u >= n_units || threadIdx.x != 0) return;
  GzUnit& U = units[u];
  const bool known = first_is_file_start && u == 0;       // the member's first unit has no history in front of it: no markers possible
  uint16_t* out = WRITE ? sym + U.out_off : nullptr;
  const uint64_t room = WRITE ? U.n_sym : ~0ull;          // the counting pass said how many
  uint64_t w = 0;
  uint32_t status = GZ_OK;
  Bits b; b.init(d, n, U.start_bit);
  for (;;) {
    if (b.over) { status = GZ_ERR_OVER; break; }
    const uint32_t last = b.get(1), type = b.get(2);
    if (type == 0) {
      b.align();
      const uint32_t len = b.get(16), nlen = b.get(16);
      if ((len ^ 0xFFFFu) != nlen) { status = GZ_ERR_STORED; break; }
      if (w + len > room) { status = GZ_ERR_ROOM; break; }
      for (uint32_t i = 0; i < len; ++i) { const uint32_t c = b.get(8); if (WRITE) out[w] = (uint16_t)c; ++w; }
      if (b.over) { status = GZ_ERR_OVER; break; }
    } else if (type == 1 || type == 2) {
      if (type == 1) fixed_codes(sc);
      else if (!read_dynamic(b, sc, false)) { status = GZ_ERR_CODE; break; }
      for (;;) {
        if (b.over) { status = GZ_ERR_OVER; break; }
        int s = sc.lit.decode(b);
        if (s < 0) { status = GZ_ERR_CODE; break; }
        if (s < 256) {
          if (w >= room || w > (1ull << 28)) { status = GZ_ERR_ROOM; break; }
          if (WRITE) out[w] = (uint16_t)s;
          ++w;
          continue;
        }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) { status = GZ_ERR_CODE; break; }
        const uint32_t len = c_len_base[s] + b.get(c_len_extra[s]);
        const int ds = sc.dist.decode(b);
        if (ds < 0 || ds >= 30) { status = GZ_ERR_CODE; break; }
        const uint32_t dist = c_dist_base[ds] + b.get(c_dist_extra[ds]);
        if (w + len > room || w > (1ull << 28)) { status = GZ_ERR_ROOM; break; }      // (a unit of more than 2^28 symbols: one wavefront would work for seconds - the CPU path)
        if (dist > w && (known || dist - w > 32768u)) { status = GZ_ERR_DIST; break; }
        if (WRITE) {
          for (uint32_t i = 0; i < len; ++i) {
            // a reference that reaches in front of the unit: position 32768 - (dist - w) + i of the unknown window, as a marker
            const uint64_t at = w + i;
            out[at] = at >= dist ? out[at - dist] : (uint16_t)(0x8000u | (uint32_t)(32768u - (dist - at)));
          }
        }
        w += len;
      }
      if (status != GZ_OK) break;
    } else { status = GZ_ERR_TYPE; break; }
    if (last) { status = GZ_FINAL; break; }
    if (b.bitpos() >= U.stop_bit) break;
  }
  U.end_bit = b.bitpos();
  if (!WRITE) U.n_sym = w;
  else if (status <= GZ_FINAL && w != room) status = GZ_ERR_ROOM;
  U.status = status;
}

// ---- 3: windows in chain order --------------------------------------------------------------------------------------------
// win_in[u] = the 32 KiB in front of unit u (bytes); the member starts with an empty (zero) window that nothing refers to
__global__ void __launch_bounds__(1024) gz_window_kernel(const GzUnit* __restrict__ units, uint32_t n_units, const uint16_t* __restrict__ sym,
                                                        uint8_t* __restrict__ win_in) {
  __shared__ uint8_t w0[32768];
  __shared__ uint8_t w1[32768];
  uint8_t* cur = w0; uint8_t* nxt = w1;
  for (int i = threadIdx.x; i < 32768; i += 1024) cur[i] = 0;
  __syncthreads();
  for (uint32_t u = 0; u < n_units; ++u) {
    uint8_t* keep = win_in + (size_t)u * 32768;
    for (int i = threadIdx.x * 4; i < 32768; i += 4096) *(uint32_t*)(keep + i) = *(const uint32_t*)(cur + i);
    const uint64_t m = units[u].n_sym;
    const uint16_t* s = sym + units[u].out_off;
    for (int t = threadIdx.x; t < 32768; t += 1024) {
      // byte t of the new window = byte (t + m) of (old window ++ resolved unit)
      const uint64_t q = (uint64_t)t + m;
      uint8_t v;
      if (q < 32768) v = cur[q];
      else { const uint16_t x = s[q - 32768]; v = x < 0x8000u ? (uint8_t)x : cur[x & 0x7FFFu]; }
      nxt[t] = v;
    }
    __syncthreads();
    uint8_t* t_ = cur; cur = nxt; nxt = t_;
  }
}

// ---- 4: markers out, bytes out --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gz_resolve_kernel(const GzUnit* __restrict__ units, uint32_t n_units, const uint16_t* __restrict__ sym,
                                                        const uint8_t* __restrict__ win_in, uint8_t* __restrict__ text, uint32_t blocks_per_unit) {
  const uint32_t u = blockIdx.x / blocks_per_unit, part = blockIdx.x % blocks_per_unit;
  if (u >= n_units) return;
  const uint64_t m = units[u].n_sym, off = units[u].out_off;
  const uint16_t* s = sym + off;
  const uint8_t* w = win_in + (size_t)u * 32768;
  for (uint64_t i = (uint64_t)part * 256 + threadIdx.x; i < m; i += (uint64_t)blocks_per_unit * 256) {
    const uint16_t x = s[i];
    text[off + i] = x < 0x8000u ? (uint8_t)x : w[x & 0x7FFFu];
  }
}

size_t gzip_header(const uint8_t* p, size_t n) {      // offset of the deflate data, 0 = not a gzip member this code takes
  if (n < 18 + 2 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
  const int flg = p[3];
  if (flg & 0xE0) return 0;
  size_t q = 10;
  if (flg & 4) { if (q + 2 > n) return 0; const size_t xlen = p[q] | (p[q + 1] << 8); q += 2 + xlen; }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  return q + 8 < n ? q : 0;
}

#define GZTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = mic_set_error(e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "gzip on the device: %s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

}  // namespace

extern "C" int mic_gz_inflate_device(mic_engine* e, const void* gz, size_t gz_bytes, void** d_text, size_t* n_text, uint32_t* crc32_expected) {
  if (!e || !gz || !d_text || !n_text) return mic_set_error(MIC_E_INVALID, "null argument");
  *d_text = nullptr; *n_text = 0;
  MicTable t_; int sc_, ncu_, dev_, k_; uint32_t nt_;
  int rc = mic_engine_table(e, &t_, &sc_, &ncu_, &dev_, &k_, &nt_);
  if (rc) return rc;
  if (hipSetDevice(dev_) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  const uint8_t* p = (const uint8_t*)gz;
  const size_t hdr = gzip_header(p, gz_bytes);
  if (!hdr) return mic_set_error(MIC_E_UNSUPPORTED, "not a plain gzip member");
  const size_t n = gz_bytes - 8;                                   // deflate data + nothing else expected in front of the trailer
  const uint32_t crc = (uint32_t)p[n] | ((uint32_t)p[n + 1] << 8) | ((uint32_t)p[n + 2] << 16) | ((uint32_t)p[n + 3] << 24);
  const uint32_t isize = (uint32_t)p[n + 4] | ((uint32_t)p[n + 5] << 8) | ((uint32_t)p[n + 6] << 16) | ((uint32_t)p[n + 7] << 24);
  if (crc32_expected) *crc32_expected = crc;
  uint8_t* d_in = nullptr; unsigned long long* d_start = nullptr; GzUnit* d_units = nullptr; uint16_t* d_sym = nullptr;
  uint8_t* d_win = nullptr; uint8_t* d_out = nullptr;
  const uint32_t n_chunks = (uint32_t)((n + GZ_CHUNK - 1) / GZ_CHUNK);
  std::vector<unsigned long long> h_start(n_chunks);
  std::vector<GzUnit> units, chain;
  uint64_t total = 0;
  hipStream_t s = nullptr;
  {
    hipStream_t up, down;
    mic_engine_copy_streams(e, &up, &down);
    s = up;
  }
  const bool timing = getenv("MIC_GZ_TIMING") != nullptr;
  struct timespec tq0; clock_gettime(CLOCK_MONOTONIC, &tq0);
  auto lap = [&](const char* what) {
    if (!timing) return;
    hipStreamSynchronize(s);
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "[gz] %s: %.3f ms\n", what, (t1.tv_sec - tq0.tv_sec) * 1e3 + (t1.tv_nsec - tq0.tv_nsec) / 1e6);
    tq0 = t1;
  };
  GZTRY(hipMalloc(&d_in, n + 16));
  GZTRY(hipMemsetAsync(d_in + n, 0, 16, s));
  GZTRY(hipMemcpyAsync(d_in, p, n, hipMemcpyHostToDevice, s));
  lap("upload");
  GZTRY(hipMalloc(&d_start, (size_t)n_chunks * 8));
  gz_find_kernel<<<n_chunks, 64, 0, s>>>(d_in, n, (uint64_t)hdr * 8, n_chunks, d_start);
  GZTRY(hipGetLastError());
  GZTRY(hipMemcpyAsync(h_start.data(), d_start, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
  GZTRY(hipStreamSynchronize(s));
  lap("find blocks");
  for (uint32_t c = 0; c < n_chunks; ++c) {
    if (h_start[c] == ~0ull) continue;
    GzUnit u; memset(&u, 0, sizeof(u));
    u.start_bit = h_start[c];
    units.push_back(u);
  }
  for (size_t i = 0; i < units.size(); ++i) units[i].stop_bit = i + 1 < units.size() ? units[i + 1].start_bit : ~0ull;
  GZTRY(hipMalloc(&d_units, units.size() * sizeof(GzUnit)));
  GZTRY(hipMemcpyAsync(d_units, units.data(), units.size() * sizeof(GzUnit), hipMemcpyHostToDevice, s));
  gz_decode_kernel<false><<<(unsigned)units.size(), 64, 0, s>>>(d_in, n, d_units, (uint32_t)units.size(), nullptr, 1);
  GZTRY(hipGetLastError());
  GZTRY(hipMemcpyAsync(units.data(), d_units, units.size() * sizeof(GzUnit), hipMemcpyDeviceToHost, s));
  GZTRY(hipStreamSynchronize(s));
  lap("decode (counting)");
  // the chain: a unit is taken iff the chain so far ends exactly on its start; one that starts inside the chain was a false find
  {
    unsigned long long pos = units[0].start_bit;
    bool final = false;
    for (size_t i = 0; i < units.size() && !final; ++i) {
      const GzUnit& u = units[i];
      if (u.start_bit < pos) continue;
      if (u.start_bit > pos || u.status > GZ_FINAL) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: the units do not stitch (unit %zu, status %u)", i, u.status); goto done; }
      GzUnit v = u;
      v.out_off = total; total += u.n_sym;
      chain.push_back(v);
      pos = u.end_bit;
      final = u.status == GZ_FINAL;
    }
    // behind the last block: the trailer, right there (bits up to the next byte are padding)
    if (!final || (pos + 7) / 8 != n) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: more than one member, or data behind the last block"); goto done; }
    if ((uint32_t)total != isize) { rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done; }
  }
  GZTRY(hipMemcpyAsync(d_units, chain.data(), chain.size() * sizeof(GzUnit), hipMemcpyHostToDevice, s));
  GZTRY(hipMalloc(&d_sym, (total + 8) * 2));
  GZTRY(hipMalloc(&d_win, chain.size() * (size_t)32768));
  GZTRY(hipMalloc(&d_out, total + 64));
  gz_decode_kernel<true><<<(unsigned)chain.size(), 64, 0, s>>>(d_in, n, d_units, (uint32_t)chain.size(), d_sym, 1);
  GZTRY(hipGetLastError());
  lap("decode (writing)");
  gz_window_kernel<<<1, 1024, 0, s>>>(d_units, (uint32_t)chain.size(), d_sym, d_win);
  GZTRY(hipGetLastError());
  lap("windows");
  gz_resolve_kernel<<<(unsigned)chain.size() * 8u, 256, 0, s>>>(d_units, (uint32_t)chain.size(), d_sym, d_win, d_out, 8);
  GZTRY(hipGetLastError());
  {
    std::vector<GzUnit> back(chain.size());
    GZTRY(hipMemcpyAsync(back.data(), d_units, chain.size() * sizeof(GzUnit), hipMemcpyDeviceToHost, s));
    GZTRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < back.size(); ++i)
      if (back[i].status > GZ_FINAL || back[i].end_bit != chain[i].end_bit) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: the writing pass differs (unit %zu)", i); goto done; }
  }
  lap("resolve");
  if (timing) fprintf(stderr, "[gz] %zu bytes -> %llu bytes, %u chunks, %zu units found, %zu in the chain\n", gz_bytes, (unsigned long long)total, n_chunks, units.size(), chain.size());
  *d_text = d_out; d_out = nullptr; *n_text = total;
done:
  if (d_in) hipFree(d_in);
  if (d_start) hipFree(d_start);
  if (d_units) hipFree(d_units);
  if (d_sym) hipFree(d_sym);
  if (d_win) hipFree(d_win);
  if (d_out) hipFree(d_out);
  return rc;
}

extern "C" int mic_gz_copy_text(mic_engine* e, const void* d_text, size_t offset, size_t n, void* host_dst) {
  if (!e || !d_text || !host_dst) return mic_set_error(MIC_E_INVALID, "null argument");
  if (hipMemcpy(host_dst, (const uint8_t*)d_text + offset, n, hipMemcpyDeviceToHost) != hipSuccess) return mic_set_error(MIC_E_HIP, "copy of inflated text failed");
  return MIC_OK;
}

extern "C" int mic_gz_free_text(mic_engine* e, void* d_text) {
  (void)e;
  if (d_text) hipFree(d_text);
  return MIC_OK;
}
