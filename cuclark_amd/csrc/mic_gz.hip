// mic_gz.hip - gzip inflated ON THE DEVICE: the two-stage scheme of csrc/pgz.hpp (pugz / rapidgzip) with one decode
// unit per deflate block - thousands of wavefronts - instead of sixteen threads.  BASELINE config 5 names gzip FASTQ; the
// reference's scripts gunzip to a temporary file first (classify_metagenome.sh:116-142).  The command line's path for compressed
// FASTQ on one engine: the text stays on the device, where the pair merge and the ingest kernels read it (mic_ingest.hip:
// mic_pairs_*, mic_text_*).  316 MB of FASTQ out of 59 MB of gzip in 12 ms (pgz.hpp on the 16 allowed CPUs: 3 GB/s); a
// block-gzip file (BGZF) of the same text, a wavefront per member, in 14-15 ms.  tests/test_gz_device.py, tools/gz_device_timing.py.
//
//   1  gz_find_kernel     one wavefront per 8 KiB of compressed data (staged in LDS): the first bit offset at which a block with
//                         dynamic codes starts - a sieve: BTYPE and the two counts on every offset (a byte a lane), the
//                         code-length code exactly complete on the fifth that pass (collected, 64 at a time), the whole header
//                         of the ~65 a chunk that pass that (64 side by side: every code complete, an end-of-block code), and
//                         the one-lane path with its codes and 300 symbols of trial decode for what is left;
//   2  gz_decode_kernel   one wavefront per unit = from one found start to the first block boundary at or behind the next found
//                         start, into a region of the symbol buffer sized by the unit's compressed span: 16-bit symbols - a
//                         byte, or a MARKER 0x8000 | i for a back-reference to position i of the 32 KiB in front of the unit.
//                         The host stitches the units into a chain (a unit whose start lies inside the unit in front of it was a
//                         false find and is dropped; a gap or an error gives the file back to the caller's CPU inflater); a
//                         unit that outgrew its region was counted exactly and is decoded again into one that fits.
//                         THE WINDOW DECODE: the 64 lanes decode speculatively what starts at each of the next 64 bit offsets
//                         (literal / length code, extra bits, distance code, extra bits: wide tables, three LDS round trips a
//                         window), the scalar side follows the chain of the offsets that really start a symbol - v_readlane,
//                         copy inside an LDS ring of the last 2 Ki symbols, two symbols a step where they do not depend on each
//                         other.  Codes longer than the tables' and matches out of the ordinary take the serial decode (all
//                         lanes redundantly on state the compiler is told is uniform), one symbol;
//   3  gz_compose_kernel / gz_chain_kernel   the 32 KiB in front of every unit: a unit's step is a map of window positions, maps
//                         compose - per group of ~sqrt(units) units in LDS, all groups at once, then one step per group;
//   4  gz_resolve2_kernel all units at once: markers replaced through the unit's map and its group's window, symbols narrowed to
//                         bytes at the unit's offset of the text.
//   5  gz_crc_kernel      the CRC-32 of the text in 4-KiB pieces, combined on the host: length (ISIZE) and CRC-32 are checked against
//                         the member's trailer, as gunzip checks them.
// Stored and fixed-code blocks are decoded; several members, a preset dictionary or anything that does not stitch:
// MIC_E_UNSUPPORTED, and the caller inflates on the CPU as before - a wrong speculation cannot pass.
#include "mi_clark.h"
#include "mic_internal.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <mutex>
#include <vector>

// engine services (mic_engine.hip)
int mic_engine_table(mic_engine* e, MicTable* t, int* slot_class, int* n_cu, int* device, int* k, uint32_t* n_targets);
int mic_set_error(int code, const char* fmt, ...);
void mic_engine_copy_streams(mic_engine* e, hipStream_t* up, hipStream_t* down);

namespace {

constexpr uint32_t GZ_CHUNK = 8192;       // compressed bytes per finder chunk: below the size of a block (gzip: 16 Ki symbols, 10-30 KB), so that
                                          // the first start of every chunk is every start and a unit is one block - the longest unit is the decode time
constexpr int GZ_TRIAL = 300;             // symbols of trial decode behind a candidate header

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- bits, LSB first: the data is followed by 16 readable zero bytes; positions behind the data read zeros and set `over`
// (LBits below for the finder, WBits for the decode).

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
  template <class B> __device__ int decode(B& b) const {
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
template <class B> __device__ bool read_dynamic(B& b, Scratch& s, bool strict) {
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

// ---- 1: block finder ----------------------------------------------------------------------------------------------------
// The finder's input comes out of LDS: the chunk's 8 KiB and 1 KiB behind them (a dynamic header is 562 bytes at most), staged once.
constexpr uint32_t GZ_FIND_STAGE = GZ_CHUNK + 1024;
constexpr uint32_t GZ_PEND = 1024;          // offsets that passed the first test and wait for the second (a step adds 512 at most to fewer than 64)
struct LBits {                                 // Bits on the staged bytes [base, base + GZ_FIND_STAGE) of the data; behind them: the data itself
  const uint32_t* st; const uint8_t* p; uint64_t base, n, pos, buf; int cnt; bool over;
  __device__ void init(const uint32_t* stage, uint64_t stage_base, const uint8_t* d, uint64_t len, uint64_t bitpos) {
    st = stage; base = stage_base; p = d; n = len; pos = bitpos >> 3; buf = 0; cnt = 0; over = false;
    refill();
    const int skip = (int)(bitpos & 7);
    buf >>= skip; cnt -= skip;
  }
  __device__ void refill() {
    const int add = (63 - cnt) >> 3;
    uint64_t v = 0;
    if (pos >= base && pos + 12 <= base + GZ_FIND_STAGE) {
      const uint32_t o = (uint32_t)(pos - base), w = o >> 2, sh = (o & 3u) * 8u;
      const uint32_t a0 = st[w], a1 = st[w + 1], a2 = st[w + 2];
      const uint64_t lo = (uint64_t)a0 | ((uint64_t)a1 << 32);
      v = sh ? (lo >> sh) | ((uint64_t)a2 << (64u - sh)) : lo;
    } else if (pos + 8 <= n + 16) {          // (a trial decode that runs on behind the stage)
      const uint8_t* s = p + (pos <= n + 8 ? pos : n + 8);
      v = (uint64_t)s[0] | ((uint64_t)s[1] << 8) | ((uint64_t)s[2] << 16) | ((uint64_t)s[3] << 24) | ((uint64_t)s[4] << 32) |
          ((uint64_t)s[5] << 40) | ((uint64_t)s[6] << 48) | ((uint64_t)s[7] << 56);
      if (pos > n + 8) v = 0;
    }
    if (pos >= n + 8) over = true;
    buf |= v << cnt;
    pos += (uint64_t)add; cnt += add * 8;
  }
  __device__ uint32_t peek(int k) { if (cnt < k) refill(); return (uint32_t)(buf & ((1ull << k) - 1)); }
  __device__ void drop(int k) { buf >>= k; cnt -= k; }
  __device__ uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }
  __device__ void skip(int k) { if (cnt < k) refill(); drop(k); }
};

// The cheap tests, every lane its own offset, on the 74 bits of a dynamic header's fixed part (rel = the offset - 8 x the stage's
// first byte).  The first test (in the kernel's loop): BTYPE = 2 (BFINAL either way: the member's last block is a unit like any other) and the two
// counts - 17 bits, a fifth of all offsets pass.  kraft_test: the code-length code
// exactly complete - its up to 19 lengths of 3 bits as two words of ten and nine fields, the ones behind HCLEN masked off, and
// (0x80 >> l) & 0x7f = 2^(7 - l) for a length l, 0 for none: the sum has to be 128.  32-bit arithmetic, no loop over HCLEN.
__device__ __forceinline__ bool kraft_test(const uint32_t* st, uint32_t rel) {
  const uint32_t w = rel >> 5, sh = rel & 31u;
  const uint32_t a0 = st[w], a1 = st[w + 1], a2 = st[w + 2], a3 = st[w + 3];
  const uint32_t s0 = __builtin_amdgcn_alignbit(a1, a0, sh), s1 = __builtin_amdgcn_alignbit(a2, a1, sh), s2 = __builtin_amdgcn_alignbit(a3, a2, sh);
  const uint32_t hclen = ((s0 >> 13) & 15u) + 4u;
  uint32_t f0 = ((s0 >> 17) | (s1 << 15)) & 0x3FFFFFFFu;                        // lengths 0..9: bits 17..46
  uint32_t f1 = ((s1 >> 15) | (s2 << 17)) & 0x07FFFFFFu;                        // lengths 10..18: bits 47..73
  if (hclen <= 10u) { f0 &= (1u << (3u * hclen)) - 1u; f1 = 0u; }
  else f1 &= (1u << (3u * (hclen - 10u))) - 1u;
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) sum += (0x80u >> ((f0 >> (3 * i)) & 7u)) & 0x7Fu;
#pragma unroll
  for (int i = 0; i < 9; ++i) sum += (0x80u >> ((f1 >> (3 * i)) & 7u)) & 0x7Fu;
  return sum == 128u;
}

// What read_dynamic(strict) decides, without keeping the codes: one lane, its own candidate - 64 candidates at a time.  The
// code-length code's table is the lane's column of an LDS array (pre[v * 64]), the 19 lengths one 57-bit number, and the two codes'
// Kraft sums, the number of distance codes and the end-of-block code's length are kept up as the lengths are read.
__device__ bool header_holds(const uint32_t* st, uint64_t st_base, const uint8_t* d, uint64_t n, uint64_t at, uint8_t* pre) {
  LBits b; b.init(st, st_base, d, n, at + 3);
  const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint64_t cl = 0;
  for (int i = 0; i < hclen; ++i) cl |= (uint64_t)b.get(3) << (3 * (int)c_cl_order[i]);
  {
    uint32_t ks = 0;
    for (int sy = 0; sy < 19; ++sy) { const uint32_t l = (uint32_t)(cl >> (3 * sy)) & 7u; if (l) ks += 1u << (15 - l); }
    if (ks != (1u << 15)) return false;                    // (complete: every 7-bit index below gets its entry)
  }
  {
    uint32_t code = 0;
    for (uint32_t l = 1; l <= 7; ++l) {
      for (int sy = 0; sy < 19; ++sy) {
        if (((uint32_t)(cl >> (3 * sy)) & 7u) != l) continue;
        const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
        for (uint32_t v = r; v < 128; v += 1u << l) pre[v * 64] = (uint8_t)((l << 5) | (uint32_t)sy);
        ++code;
      }
      code <<= 1;
    }
  }
  const int total = hlit + hdist;
  int at_n = 0, prev = 0;
  uint32_t kl = 0, kd = 0, used = 0, len256 = 0;
  while (at_n < total) {
    const uint32_t e = pre[b.peek(7) * 64];
    b.drop((int)(e >> 5));
    const int sy = (int)(e & 31u);
    int rep = 1, val = 0;
    if (sy < 16) val = sy;
    else if (sy == 16) { if (at_n == 0) return false; val = prev; rep = 3 + (int)b.get(2); }
    else if (sy == 17) rep = 3 + (int)b.get(3);
    else rep = 11 + (int)b.get(7);
    if (at_n + rep > total) return false;
    if (val) {
      const int a = at_n < hlit ? (rep < hlit - at_n ? rep : hlit - at_n) : 0;      // how many of them are literal / length codes
      kl += (uint32_t)a << (15 - val);
      kd += (uint32_t)(rep - a) << (15 - val);
      used += (uint32_t)(rep - a);
    }
    if (at_n <= 256 && 256 < at_n + rep) len256 = (uint32_t)val;
    at_n += rep; prev = val;
    if (kl > (1u << 15) || kd > (1u << 15)) return false;          // (over-subscribed already: what the sums below would say at the end)
  }
  if (b.over || len256 == 0) return false;
  if (kl != (1u << 15)) return false;
  if (kd > (1u << 15)) return false;
  if (kd < (1u << 15) && used > 1) return false;
  return true;
}

// The finder is arithmetic: 65 536 offsets a chunk.  Every offset takes first_test (a dozen instructions); the fifth that pass are
// COLLECTED and take kraft_test 64 at a time (all lanes busy - tested in place, three lanes in four would idle through the
// nineteen lengths); the ~65 a chunk that pass that are collected again and go through their whole header side by side
// (header_holds), and only those whose header holds - one in thousands is not a block - take the one-lane path with its codes and
// the trial decode.  In the order of the offsets throughout: the first start of the chunk is the answer.
__global__ void __launch_bounds__(64) gz_find_kernel(const uint8_t* __restrict__ d, uint64_t n, uint64_t first_bit, uint32_t n_chunks,
                                                    unsigned long long* __restrict__ start) {
  // (LDS is what limits the wavefronts a CU holds here, and the finder's time goes with their number: 19.5 KB a chunk, eight a CU.
  // The lanes' code-length tables and the one lane's codes are never live together; offsets inside a chunk fit 16 bits.)
  __shared__ union { Scratch sc; uint8_t pre_l[128 * 64]; } un;
  Scratch& sc = un.sc;
  uint8_t* pre_l = un.pre_l;
  __shared__ uint16_t cand[192];
  __shared__ uint16_t pend[GZ_PEND];
  __shared__ uint32_t stg[GZ_FIND_STAGE / 4 + 4];
  const uint32_t c = blockIdx.x;
  const int lane = threadIdx.x;
  if (c >= n_chunks) return;
  if (c == 0) { if (lane == 0) start[0] = first_bit; return; }
  const uint64_t from = (uint64_t)c * GZ_CHUNK * 8, to = (uint64_t)(c + 1) * GZ_CHUNK * 8 < n * 8 ? (uint64_t)(c + 1) * GZ_CHUNK * 8 : n * 8;
  const uint64_t st_base = (uint64_t)c * GZ_CHUNK;                 // (d is 256-byte aligned, a chunk a multiple of 4 bytes)
  for (uint32_t i = lane; i < GZ_FIND_STAGE / 4 + 4; i += 64) {
    const uint64_t byte = st_base + 4ull * i;
    stg[i] = byte + 4 <= n + 16 ? *(const uint32_t*)(d + byte) : 0u;
  }
  __builtin_amdgcn_wave_barrier();
  unsigned long long found = ~0ull;
  uint32_t n_c = 0;
  // the first `cnt` collected candidates, in the order of their offsets
  auto batch = [&](uint32_t cnt) {
    __builtin_amdgcn_wave_barrier();
    const uint64_t mine = from + (uint64_t)cand[(uint32_t)lane < cnt ? lane : 0];
    const bool holds = (uint32_t)lane < cnt && header_holds(stg, st_base, d, n, mine, pre_l + lane);
    unsigned long long m = __ballot(holds);
    while (m && found == ~0ull) {
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      const uint64_t at = from + (uint64_t)cand[l];
      int good = 0;
      if (lane == 0) {
        LBits b; b.init(stg, st_base, d, n, at + 3);
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
    __builtin_amdgcn_wave_barrier();
    // the rest of the list moves up
    const uint32_t left = n_c - cnt;
    uint32_t keep0 = 0, keep1 = 0;
    if ((uint32_t)lane < left) keep0 = cand[cnt + (uint32_t)lane];
    if ((uint32_t)lane + 64u < left) keep1 = cand[cnt + (uint32_t)lane + 64u];
    __builtin_amdgcn_wave_barrier();
    if ((uint32_t)lane < left) cand[lane] = keep0;
    if ((uint32_t)lane + 64u < left) cand[lane + 64] = keep1;
    n_c = left;
    __builtin_amdgcn_wave_barrier();
  };
  uint32_t n_p = 0, head = 0;                            // pend is a ring: entries head .. head + n_p - 1 (mod GZ_PEND)
  const unsigned long long below = (1ull << lane) - 1ull;
  // the first `cnt` offsets that passed first_test through kraft_test; what passes joins the candidates
  auto sift = [&](uint32_t cnt) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t rel = pend[(head + ((uint32_t)lane < cnt ? (uint32_t)lane : 0u)) & (GZ_PEND - 1u)];
    const bool is = (uint32_t)lane < cnt && kraft_test(stg, rel);
    const unsigned long long m = __ballot(is);
    if (is) cand[n_c + (uint32_t)__builtin_popcountll(m & below)] = (uint16_t)rel;
    n_c += (uint32_t)__builtin_popcountll(m);
    head += cnt; n_p -= cnt;
    __builtin_amdgcn_wave_barrier();
    if (n_c >= 64u) batch(64u);
  };
  // first_test, a BYTE a lane: the 32 bits from its byte on hold the 17 bits of all eight offsets inside it; 64 bytes a step.  What
  // passes is appended in the order of the offsets: lane by lane (a count over the lanes below per bit of the byte), bit by bit.
  const uint64_t lo_bit = from > first_bit ? from : first_bit + 1;
  const uint32_t lo_rel = (uint32_t)(lo_bit - from);
  const uint32_t bytes_here = (uint32_t)((to - from) >> 3);
  for (uint32_t bb = 0; bb < bytes_here && found == ~0ull; bb += 64) {
    const uint32_t B = bb + (uint32_t)lane;
    const bool live = B < bytes_here && st_base + B + 12 < n;
    const uint32_t v = __builtin_amdgcn_alignbit(stg[(B >> 2) + 1], stg[B >> 2], (B & 3u) * 8u);
    uint32_t mask8 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t t = v >> k;
      if ((t & 6u) == 4u && ((t >> 3) & 31u) <= 29u && ((t >> 8) & 31u) <= 29u && B * 8u + (uint32_t)k >= lo_rel) mask8 |= 1u << k;
    }
    if (!live) mask8 = 0;
    uint32_t at = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long m = __ballot((mask8 >> k) & 1u);
      at += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      tot += (uint32_t)__builtin_popcountll(m);
    }
    at += head + n_p;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((mask8 >> k) & 1u) { pend[at & (GZ_PEND - 1u)] = (uint16_t)(B * 8u + (uint32_t)k); ++at; }
    n_p += tot;
    while (n_p >= 64u && found == ~0ull) sift(64u);
  }
  while (n_p && found == ~0ull) sift(n_p < 64u ? n_p : 64u);
  while (n_c && found == ~0ull) batch(n_c < 64u ? n_c : 64u);
  if (lane == 0) start[c] = found;
}

// ---- 2: decode units ------------------------------------------------------------------------------------------------------
struct GzUnit {
  unsigned long long start_bit, stop_bit;      // decode from start_bit to the first block boundary at or behind stop_bit
  unsigned long long sym_off, sym_cap;         // its region of the symbol buffer (in symbols)
  unsigned long long out_off;                  // where its bytes go in the text (set once the chain is known)
  unsigned long long end_bit;                  // result: where it stopped
  unsigned long long n_sym;                    // result: symbols produced (also when they did not fit: status GZ_ERR_ROOM)
  uint32_t status;                             // result: 0 ok, 1 ended with the member's last block, >= 2 error
  uint32_t pad;
};
enum { GZ_OK = 0, GZ_FINAL = 1, GZ_ERR_CODE = 2, GZ_ERR_OVER = 3, GZ_ERR_STORED = 4, GZ_ERR_DIST = 5, GZ_ERR_TYPE = 6, GZ_ERR_ROOM = 7 };

// The decode of one unit is a serial thing; the wavefront runs it REDUNDANTLY in all 64 lanes (same addresses, same values, no
// divergence) so that the lanes are there for what is parallel: staging the compressed bytes into LDS, filling the decode tables,
// copying a match.  Bits come out of a 512-byte LDS window of the input; a symbol is one LDS lookup in a table of the codes of at
// most 10 (distances: 8) bits - longer codes take the count / symbol walk.
constexpr int GZ_FAST_LIT = 10, GZ_FAST_DIST = 8, GZ_STAGE = 512, GZ_RING = 2048;

// Everything the decode decides on is the same in all lanes; the compiler cannot know that of a value that came out of memory, and
// treats every branch on it as divergent (exec masks, both sides).  UNI() says it: the value goes through v_readfirstlane into a
// scalar register, what is computed from it is scalar arithmetic, and the branches are scalar branches.
__device__ __forceinline__ uint32_t UNI(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t UNI64(uint64_t x) { return ((uint64_t)UNI((uint32_t)(x >> 32)) << 32) | UNI((uint32_t)x); }

struct Fast {
  Scratch sc;
  uint16_t lit_fast[1 << GZ_FAST_LIT];        // len << 12 | symbol, 0 = longer code
  uint16_t dist_fast[1 << GZ_FAST_DIST];
  uint32_t lit_wide[1 << GZ_FAST_LIT];        // the window decode's tables (fill_wide): a code's length, its extra bits and its base in one entry
  uint32_t dist_wide[1 << GZ_FAST_DIST];
  uint32_t stage[GZ_STAGE / 4 + 4];
  uint32_t len_tab[32], dist_tab[32];         // base | extra bits << 16: the constant tables, here because a lookup with a vector index
                                              // in constant memory is a global load (hundreds of cycles on the decode's critical path)
  uint16_t ring[GZ_RING];                     // the unit's last GZ_RING symbols: literals land here, matches are copied inside it, and
                                              // whole groups of 64 go out to the symbol buffer in one store - a match that reads the
                                              // unit's own output from global memory waits for every store in flight first
};

struct WBits {                                 // the Bits of a whole wavefront: refills out of the LDS stage
  const uint8_t* p; uint64_t n; uint64_t pos; uint64_t buf; int cnt; bool over;
  uint64_t base;                               // the stage holds bytes [base, base + GZ_STAGE) of the data (base a multiple of 4)
  uint32_t* stage; int lane;
  __device__ void restage(uint64_t at) {
    base = at & ~3ull;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < GZ_STAGE / 4 + 2; i += 64) {
      const uint64_t byte = base + 4ull * (uint64_t)i;
      stage[i] = byte + 4 <= n + 16 ? *(const uint32_t*)(p + byte) : 0u;       // p is 256-byte aligned, base a multiple of 4
    }
    __builtin_amdgcn_wave_barrier();
  }
  __device__ void init(const uint8_t* d, uint64_t len, uint64_t bitpos, uint32_t* st, int ln) {
    p = d; n = len; pos = bitpos >> 3; buf = 0; cnt = 0; over = false; stage = st; lane = ln;
    restage(pos);
    refill();
    const int skip = (int)(bitpos & 7);
    buf >>= skip; cnt -= skip;
  }
  __device__ void refill() {
    if (pos >= n + 8) { over = true; const int add = (63 - cnt) >> 3; pos += (uint64_t)add; cnt += add * 8; return; }
    if (pos < base || pos + 8 > base + GZ_STAGE) restage(pos);
    const uint32_t o = (uint32_t)(pos - base);
    const uint32_t w = o >> 2, sh = (o & 3u) * 8u;
    const uint32_t s0 = UNI(stage[w]), s1 = UNI(stage[w + 1]), s2 = UNI(stage[w + 2]);
    const uint64_t lo = (uint64_t)s0 | ((uint64_t)s1 << 32);
    const uint64_t v = sh ? (lo >> sh) | ((uint64_t)s2 << (64 - sh)) : lo;
    buf |= v << cnt;
    const int add = (63 - cnt) >> 3;
    pos += (uint64_t)add; cnt += add * 8;
  }
  __device__ uint32_t peek(int k) { if (cnt < k) refill(); return (uint32_t)(buf & ((1ull << k) - 1)); }
  __device__ void drop(int k) { buf >>= k; cnt -= k; }
  __device__ uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }
  __device__ uint64_t bitpos() const { return pos * 8 - (uint64_t)cnt; }
  __device__ void align() { drop(cnt & 7); }
  __device__ void seek(uint64_t bit) {           // continue at this bit (the window decode hands back to the serial one)
    pos = bit >> 3; buf = 0; cnt = 0;
    refill();
    const int skip = (int)(bit & 7);
    buf >>= skip; cnt -= skip;
  }
};

// WBits::restage for the window decode, NOT inlined: inlined, its loads make the compiler wait for every outstanding memory
// operation (the counter is one for loads and stores) at the top of every window - also behind a flush, whose store then costs
// a window its whole latency.
__device__ __attribute__((noinline)) void restage_apart(const uint8_t* p, uint64_t n, uint64_t base, uint32_t* stage, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < GZ_STAGE / 4 + 2; i += 64) {
    const uint64_t byte = base + 4ull * (uint64_t)i;
    stage[i] = byte + 4 <= n + 16 ? *(const uint32_t*)(p + byte) : 0u;
  }
  __builtin_amdgcn_wave_barrier();
}

// the same Huffman walk on the wavefront's bits
__device__ int slow_decode(const Huff& h, WBits& b, int limit) {
  const uint32_t v = b.peek(15);
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; ++l) {
    code |= (int)((v >> (l - 1)) & 1u);
    const int c = (int)UNI(h.count[l]);
    if (code - c < first) { b.drop(l); const int at = index + (code - first); return at < limit ? (int)UNI(h.symbol[at]) : -1; }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}

// direct table of the codes of at most FASTBITS bits out of a built Huff (count / symbol): all lanes fill
template <int FASTBITS>
__device__ void fill_fast(const Huff& h, const uint8_t* len, uint16_t* fast, int lane) {
  for (int i = lane; i < (1 << FASTBITS); i += 64) fast[i] = 0;
  __builtin_amdgcn_wave_barrier();
  // first code and first sorted index of every length
  int first_code[16], first_idx[16];
  { int code = 0, idx = 0; for (int l = 1; l <= 15; ++l) { first_code[l] = code; first_idx[l] = idx; code = (code + h.count[l]) << 1; idx += h.count[l]; } }
  int n_codes = 0;
  for (int l = 1; l <= 15; ++l) n_codes += h.count[l];
  for (int idx = lane; idx < n_codes; idx += 64) {
    const int sym = h.symbol[idx];
    const int l = len[sym] & 15;
    if (l == 0 || l > FASTBITS) continue;
    const uint32_t code = (uint32_t)(first_code[l] + (idx - first_idx[l]));
    const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
    const uint16_t e = (uint16_t)((l << 12) | sym);
    for (uint32_t v = r; v < (1u << FASTBITS); v += 1u << l) fast[v] = e;
  }
  __builtin_amdgcn_wave_barrier();
}

// The first level of the literal decode lives in ONE vector register: lane i holds the entry of the 6-bit index i, and a lookup is
// v_readlane with a scalar lane number - no LDS round trip on the critical path of the symbols that make up a FASTQ file
// (nucleotides: codes of 2-3 bits, two of them in six).  Entry: bits used << 16 | second literal << 8 | first literal | count << 24
// (count 1 or 2); 0 = the code is longer than six bits or no literal: the LDS tables decide.
__device__ uint32_t first_level(const Fast& f, int lane) {
  const uint32_t e1 = f.lit_fast[lane];
  if (!e1 || (e1 & 0xFFFu) >= 256u || (e1 >> 12) > 6u) return 0u;
  const uint32_t l1 = e1 >> 12;
  const uint32_t e2 = f.lit_fast[(uint32_t)lane >> l1];
  const uint32_t l2 = e2 >> 12;
  if (e2 && (e2 & 0xFFFu) < 256u && l1 + l2 <= 6u) return (2u << 24) | ((l1 + l2) << 16) | ((e2 & 0xFFu) << 8) | (e1 & 0xFFu);
  return (1u << 24) | (l1 << 16) | (e1 & 0xFFu);
}

// The tables of the window decode (gz_decode_body): what the serial path finds in two dependent lookups - the code, then the
// base and the extra bits of a length or a distance - in one 32-bit entry.
//   lit_wide:  bits of the code | extra bits << 4 | kind << 8 (0 literal, 1 length, 2 end of block) | literal or length base << 16
//   dist_wide: bits of the code | extra bits << 4 | distance base << 16            0 = a longer code, or no valid symbol: the serial path
__device__ void fill_wide(Fast& f, int lane) {
  for (int i = lane; i < (1 << GZ_FAST_LIT); i += 64) {
    const uint32_t e = f.lit_fast[i], l = e >> 12, sy = e & 0xFFFu;
    uint32_t x = 0;
    if (e) {
      if (sy < 256u) x = l | (sy << 16);
      else if (sy == 256u) x = l | (2u << 8);
      else if (sy - 257u < 29u) { const uint32_t lt = f.len_tab[sy - 257u]; x = l | ((lt >> 16) << 4) | (1u << 8) | ((lt & 0xFFFFu) << 16); }
    }
    f.lit_wide[i] = x;
  }
  for (int i = lane; i < (1 << GZ_FAST_DIST); i += 64) {
    const uint32_t e = f.dist_fast[i], l = e >> 12, sy = e & 0xFFFu;
    uint32_t x = 0;
    if (e && sy < 30u) { const uint32_t dt = f.dist_tab[sy]; x = l | ((dt >> 16) << 4) | ((dt & 0xFFFFu) << 16); }
    f.dist_wide[i] = x;
  }
  __builtin_amdgcn_wave_barrier();
}

// read_dynamic on the wavefront's bits (every lane the same), then the fast tables
__device__ bool read_dynamic_w(WBits& b, Fast& f, int lane) {
  Scratch& s = f.sc;
  const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t cl[19];
  for (int i = 0; i < 19; ++i) cl[i] = 0;
  for (int i = 0; i < hclen; ++i) cl[c_cl_order[i]] = (uint8_t)b.get(3);
  const uint32_t ks = kraft(cl, 19);
  if (ks > (1u << 15)) return false;
  if (ks < (1u << 15)) { int used = 0; for (int i = 0; i < 19; ++i) used += cl[i] != 0; if (used != 1) return false; }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    for (int i = 0; i < 128; ++i) s.pre[i] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 7; ++l) {
      for (int sym = 0; sym < 19; ++sym) {
        if (cl[sym] != l) continue;
        const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
        for (uint32_t v = r; v < 128; v += 1u << l) s.pre[v] = (uint8_t)((l << 5) | sym);
        ++code;
      }
      code <<= 1;
    }
  }
  __builtin_amdgcn_wave_barrier();
  int n = 0;
  const int total = hlit + hdist;
  int prev = 0;
  while (n < total) {
    const uint32_t e = UNI(s.pre[b.peek(7)]);
    if (!e) return false;
    b.drop((int)(e >> 5));
    const int sy = (int)(e & 31);
    if (sy < 16) { if (lane == 0) s.len[n] = (uint8_t)sy; prev = sy; ++n; continue; }
    int rep, val = 0;
    if (sy == 16) { if (n == 0) return false; val = prev; rep = 3 + (int)b.get(2); }
    else if (sy == 17) rep = 3 + (int)b.get(3);
    else rep = 11 + (int)b.get(7);
    if (n + rep > total) return false;
    if (lane == 0) for (int i = 0; i < rep; ++i) s.len[n + i] = (uint8_t)val;
    n += rep; prev = val;
  }
  __builtin_amdgcn_wave_barrier();
  if (b.over || UNI(s.len[256]) == 0) return false;
  {
    // the two Kraft sums and the number of distance codes: 316 LDS reads, spread over the lanes and summed
    uint32_t kl = 0, kd = 0, used = 0;
    for (int i = lane; i < total; i += 64) {
      const uint32_t l = s.len[i];
      if (!l) continue;
      if (i < hlit) kl += 1u << (15 - l); else { kd += 1u << (15 - l); ++used; }
    }
    for (int o = 32; o > 0; o >>= 1) { kl += __shfl_xor(kl, o); kd += __shfl_xor(kd, o); used += __shfl_xor(used, o); }
    kl = UNI(kl); kd = UNI(kd); used = UNI(used);
    if (kl != (1u << 15)) return false;
    if (kd > (1u << 15)) return false;
    if (kd < (1u << 15) && used > 1) return false;
  }
  if (lane == 0) { s.lit.build(s.len, hlit); s.dist.build(s.len + hlit, hdist); }
  __builtin_amdgcn_wave_barrier();
  fill_fast<GZ_FAST_LIT>(s.lit, s.len, f.lit_fast, lane);
  fill_fast<GZ_FAST_DIST>(s.dist, s.len + hlit, f.dist_fast, lane);
  fill_wide(f, lane);
  return true;
}

__device__ void fixed_codes_w(Fast& f, int lane) {
  Scratch& s = f.sc;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    for (int i = 0; i < 144; ++i) s.len[i] = 8;
    for (int i = 144; i < 256; ++i) s.len[i] = 9;
    for (int i = 256; i < 280; ++i) s.len[i] = 7;
    for (int i = 280; i < 288; ++i) s.len[i] = 8;
    for (int i = 0; i < 30; ++i) s.len[288 + i] = 5;
    s.lit.build(s.len, 288);
    s.dist.build(s.len + 288, 30);
  }
  __builtin_amdgcn_wave_barrier();
  fill_fast<GZ_FAST_LIT>(s.lit, s.len, f.lit_fast, lane);
  fill_fast<GZ_FAST_DIST>(s.dist, s.len + 288, f.dist_fast, lane);
  fill_wide(f, lane);
}

// WRITE = false: the counting form (units that did not fit their region are counted exactly and decoded again)
template <bool WRITE>
__device__ __forceinline__ void gz_decode_body(Fast& f, const uint8_t* __restrict__ d, uint64_t n, GzUnit* __restrict__ units, uint32_t n_units,
                                               uint16_t* __restrict__ sym, const uint32_t* __restrict__ which, int all_known, uint32_t u_base = 0) {
  const uint32_t u = which ? which[blockIdx.x] : blockIdx.x + u_base;         // (u_base: a stripe of the units, GzJob::next)
  const int lane = threadIdx.x;
  if (u >= n_units) return;
  GzUnit& U = units[u];
  const bool known = u == 0 || all_known;                 // a member's first unit has nothing in front of it: no markers possible
  const uint64_t cyc0 = __builtin_readcyclecounter();
  uint16_t* out = sym + UNI64(U.sym_off);
  const uint64_t room64 = WRITE ? UNI64(U.sym_cap) : 0;
  const uint32_t room = room64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)room64;
  const uint64_t stop_bit = UNI64(U.stop_bit);
  uint32_t w = 0, wf = 0;                                 // symbols produced / symbols that have left the ring for the symbol buffer
                                                          // (32 bits - a unit of more than 2^28 symbols is given back, below: the scalar unit
                                                          // compares 32-bit numbers, 64-bit ones go through vector compares)
  uint32_t status = GZ_OK;
  constexpr uint32_t RM = GZ_RING - 1;
  // symbols [wf, upto) out of the ring, 64 per store (only what fits the unit's region: the rest is counted, and decoded again)
  auto flush = [&](uint32_t upto) {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t g = wf; g < upto; g += 64) {
      const uint32_t at = g + (uint32_t)lane;
      if (at < upto && at < room) out[at] = f.ring[at & RM];
    }
    wf = upto;
    __builtin_amdgcn_wave_barrier();
  };
  if (lane < 29) f.len_tab[lane] = (uint32_t)c_len_base[lane] | ((uint32_t)c_len_extra[lane] << 16);
  if (lane < 30) f.dist_tab[lane] = (uint32_t)c_dist_base[lane] | ((uint32_t)c_dist_extra[lane] << 16);
  __builtin_amdgcn_wave_barrier();
  WBits b; b.init(d, n, UNI64(U.start_bit), f.stage, lane);
  for (;;) {
    if (b.over) { status = GZ_ERR_OVER; break; }
    const uint32_t last = b.get(1), type = b.get(2);
    if (type == 0) {
      b.align();
      const uint32_t len = b.get(16), nlen = b.get(16);
      if ((len ^ 0xFFFFu) != nlen) { status = GZ_ERR_STORED; break; }
      // stored bytes: straight out of the data (the bit buffer is byte-aligned here), all lanes; the ring keeps their tail
      const uint64_t from = b.bitpos() >> 3;
      if (from + len > n) { status = GZ_ERR_OVER; break; }
      flush(w);
      for (uint32_t i = lane; i < len; i += 64) {
        const uint16_t c = d[from + i];
        if (w + i < room) out[w + i] = c;
        if (len - i <= GZ_RING) f.ring[(w + i) & RM] = c;
      }
      w += len; wf = w;
      if (w > (1u << 28)) { status = GZ_ERR_ROOM; break; }
      __threadfence();                                     // (rare; the far path above counts on whole groups otherwise)
      __builtin_amdgcn_wave_barrier();
      b.pos = from + len; b.buf = 0; b.cnt = 0;
    } else if (type == 1 || type == 2) {
      if (type == 1) fixed_codes_w(f, lane);
      else if (!read_dynamic_w(b, f, lane)) { status = GZ_ERR_CODE; break; }
      const uint32_t t6 = first_level(f, lane);
      bool bad = false;
      uint32_t go = 1u;              // the symbol loop has ONE exit, and it is uniform by construction: with an exit the compiler cannot prove
                                     // uniform the whole loop runs under an exec mask and every value it carries lives in vector registers
                                     // (eleven v_readfirstlane and seven moves back per symbol)
      // (said before the loop as well: a value that enters the loop from the header's code, which the compiler cannot prove uniform,
      // makes the loop's own copy of it a vector register whatever the loop does with it)
      b.buf = UNI64(b.buf); b.pos = UNI64(b.pos); b.cnt = (int)UNI((uint32_t)b.cnt); b.base = UNI64(b.base);
      w = UNI(w); wf = UNI(wf); b.over = UNI((uint32_t)b.over) != 0u;
      // A match, one element per lane: element i comes from w - dist + i, or - where that is inside the match itself - from
      // w - dist + (i mod dist).  In front of the unit: position 32768 - (dist - (w + i)) of the unknown window, as a marker; inside
      // the ring (the last GZ_RING symbols, minus what this match overwrites): an LDS copy; further back: the symbol buffer, once
      // everything the ring still holds has been written out and has arrived.
      auto match = [&](uint32_t len, uint32_t dist) -> bool {
        if (dist > w && (known || dist - w > 32768u)) { status = GZ_ERR_DIST; return false; }
        const bool far = dist > (uint32_t)GZ_RING - 258u;                     // some source symbol may have left the ring
        if (far) {
          // Its sources lie GZ_RING - 516 symbols or more behind w: they left the ring in groups of 64 at least
          // (GZ_RING - 516 - 385) / 64 = 17 stores ago.  Memory operations of a wavefront complete in the order of their issue for
          // the counter, so "at most 8 still in flight" means those stores have arrived - no flush, no full wait.
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
        if (len <= 64u && dist >= len && dist <= w && !far) {
          // the common match in one step: at most 64 symbols, its source whole inside the ring and in front of the match (three
          // scalar compares instead of the general loop's per-lane cases: self-overlap, markers, sources that left the ring)
          if ((uint32_t)lane < len) f.ring[(w + (uint32_t)lane) & RM] = f.ring[(w - dist + (uint32_t)lane) & RM];
        } else
        for (uint32_t i = lane; i < len; i += 64) {
          uint32_t j = i;
          if (j >= dist) j = dist == 1 ? 0u : j % dist;      // (len > dist: a run; dist 1 is the common one)
          const int32_t rel = (int32_t)(w - dist + j);          // (w <= 2^28, dist <= 32 Ki)
          uint16_t x;
          if (rel < 0) x = (uint16_t)(0x8000u | (uint32_t)(32768 + rel));
          else if (far && (uint32_t)rel < wf) x = (uint32_t)rel < room ? __atomic_load_n(&out[rel], __ATOMIC_RELAXED) : (uint16_t)0;   // (past the region: the unit is decoded again anyway)
          else x = f.ring[(uint32_t)rel & RM];
          f.ring[(w + i) & RM] = x;
        }
        __builtin_amdgcn_wave_barrier();
        w += len;
        if (w - wf >= 64) flush(w & ~63u);
        return true;
      };
      const uint64_t nbits = n * 8ull;
      for (;;) {
        if (!UNI(go)) break;
        // (the loop-carried state, said uniform once per symbol: the compiler keeps it in scalar registers from here to the back edge)
        b.buf = UNI64(b.buf); b.pos = UNI64(b.pos); b.cnt = (int)UNI((uint32_t)b.cnt); b.base = UNI64(b.base);
        w = UNI(w); wf = UNI(wf); b.over = UNI((uint32_t)b.over) != 0u;
        if (b.over) { status = GZ_ERR_OVER; bad = true; go = 0u; continue; }
        // THE WINDOW DECODE.  The serial decode spends ~200 cycles on a literal and ~1 150 on a match (a wavefront on its own issues an
        // instruction every nine cycles or so - counted - and a match is four dependent LDS round trips).  Here the 64 lanes decode
        // SPECULATIVELY what starts at each of the next 64 bit offsets - lane i: the literal / length code at offset i with its extra
        // bits and, behind them, a distance code with its extra bits, out of the wide tables: three LDS round trips for the whole
        // window instead of four per match - and the scalar side then only follows the chain of the offsets that really start a
        // symbol: v_readlane of lane p, act, p += bits used.
        // An offset whose code is longer than the tables' (or invalid, or the end of the data near) is left to the serial path below,
        // one symbol, after which the next window starts.
        {
          uint64_t P = b.bitpos();
          uint32_t stop = 0u;                                // 1: this offset takes the serial path, 2: end of block, 3: error
          if (P + 256u <= nbits) {
            for (;;) {
              P = UNI64(P); w = UNI(w); wf = UNI(wf); b.base = UNI64(b.base);
              if (P + 256u > nbits) break;
              if (w > (1u << 28)) { status = GZ_ERR_ROOM; stop = 3u; break; }
              const uint64_t byte = P >> 3;
              if (byte < b.base || byte + 24u > b.base + GZ_STAGE) { b.base = byte & ~3ull; restage_apart(b.p, b.n, b.base, b.stage, lane); }
              const uint32_t bo = (uint32_t)(P - b.base * 8ull) + (uint32_t)lane;
              const uint32_t wi = bo >> 5, sh = bo & 31u;
              const uint32_t a0 = f.stage[wi], a1 = f.stage[wi + 1], a2 = f.stage[wi + 2];
              const uint32_t vlo = __builtin_amdgcn_alignbit(a1, a0, sh), vhi = __builtin_amdgcn_alignbit(a2, a1, sh);   // 64 bits from this lane's offset on
              const uint32_t e = f.lit_wide[vlo & ((1u << GZ_FAST_LIT) - 1u)];
              const uint32_t l = e & 15u, eb = (e >> 4) & 15u, kind = (e >> 8) & 3u;
              const uint32_t c1 = l + eb;                                                        // (<= 15)
              const uint32_t val = (e >> 16) + ((vlo >> l) & ((1u << eb) - 1u));                // the literal, or the match's length
              const uint32_t v2 = __builtin_amdgcn_alignbit(vhi, vlo, c1);
              const uint32_t e2 = f.dist_wide[v2 & ((1u << GZ_FAST_DIST) - 1u)];
              const uint32_t l2 = e2 & 15u, eb2 = (e2 >> 4) & 15u;                               // (l2 + eb2 <= 21)
              const uint32_t dist_l = (e2 >> 16) + ((v2 >> l2) & ((1u << eb2) - 1u));           // the distance, were this a match
              const bool deep = w >= (uint32_t)GZ_RING;          // (then every distance of a common match lies inside the unit's own output)
              // R0: bits used | flags << 8 | symbols << 16;  R1: the distance - of a literal: its value.  Flag 4 = HOT: a literal, or
              // THE COMMON MATCH - at most 64 symbols, its source in front of it and inside the ring; decided here, by every lane
              // for its offset.  Flag 1 = literal, 2 = end of block.  (Selects, not branches: the lanes differ.)
              const bool is_lit = kind == 0u, is_len = kind == 1u;
              const uint32_t cons = l == 0u ? 0u : (is_len ? (l2 ? c1 + l2 + eb2 : 0u) : c1);
              const bool common = is_len && deep && val <= 64u && dist_l >= val && dist_l <= (uint32_t)GZ_RING - 258u;
              const uint32_t fl = cons == 0u ? 0u : (is_lit ? 5u : (kind == 2u ? 2u : (common ? 4u : 0u)));
              const uint32_t n_out = is_lit ? 1u : val, R1 = is_lit ? val : dist_l;
              const uint32_t R0 = cons | (fl << 8) | (n_out << 16);
              // ... and the symbol BEHIND this offset's, were it one: lane i looks at lane i + bits used.  Where both are hot and the
              // second does not read what the first writes (distance >= both lengths), the chain takes the two in one step: N0 / N1 are
              // the second's R0 / R1, or 0.
              const uint32_t nx = (uint32_t)lane + cons;
              const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nx & 63u) << 2), (int)R0);
              const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nx & 63u) << 2), (int)R1);
              const bool two = (fl & 4u) && nx < 64u && (b0 & 0x400u) && ((b0 & 0x100u) || b1 >= n_out + (b0 >> 16));
              const uint32_t N0 = two ? b0 : 0u, N1 = two ? b1 : 0u;
              uint32_t p = 0u;
              for (;;) {
                // Literals and common matches: ONE straight body in a loop of its own - a literal is a copy of one symbol whose value
                // comes out of R1 instead of the ring.  (One loop over all kinds came out of the compiler as eight branches a symbol,
                // and a taken branch costs a wavefront on its own ~20 cycles; so does every `if` in here.)  The loop leaves for
                // anything else, and for the flush of a whole group of 64.
                while (p < 64u && w - wf < 64u) {
                  const uint32_t r0 = __builtin_amdgcn_readlane(R0, (int)p);
                  if (!(r0 & 0x400u)) break;
                  const uint32_t r1 = __builtin_amdgcn_readlane(R1, (int)p);
                  const uint32_t q0 = __builtin_amdgcn_readlane(N0, (int)p), q1 = __builtin_amdgcn_readlane(N1, (int)p);
                  const uint32_t cnt = r0 >> 16, cnt2 = q0 >> 16, w2 = w + cnt;
                  const uint16_t x = f.ring[(w - r1 + (uint32_t)lane) & RM];
                  const uint16_t x2 = f.ring[(w2 - q1 + (uint32_t)lane) & RM];
                  const uint16_t y = (r0 & 0x100u) ? (uint16_t)r1 : x;
                  const uint16_t y2 = (q0 & 0x100u) ? (uint16_t)q1 : x2;
                  if ((uint32_t)lane < cnt) f.ring[(w + (uint32_t)lane) & RM] = y;
                  if ((uint32_t)lane < cnt2) f.ring[(w2 + (uint32_t)lane) & RM] = y2;
                  w = w2 + cnt2;
                  p += (r0 & 0xFFu) + (q0 & 0xFFu);
                }
                if (w - wf >= 64u) { flush(w & ~63u); continue; }
                if (p >= 64u) break;
                // everything else, one symbol
                const uint32_t r0 = __builtin_amdgcn_readlane(R0, (int)p);
                const uint32_t used = r0 & 0xFFu;
                if (r0 & 0x400u) continue;
                if (used == 0u) { stop = 1u; break; }
                if (r0 & 0x200u) { p += used; stop = 2u; break; }
                const uint32_t r1 = __builtin_amdgcn_readlane(R1, (int)p);
                if (!match(r0 >> 16, r1)) { stop = 3u; break; }
                p += used;
              }
              P += p;
              if (stop) break;
            }
            b.seek(P);
          }
          if (stop == 2u) { go = 0u; continue; }
          if (stop == 3u) { bad = true; go = 0u; continue; }
        }
        // one symbol by the serial decode
        const uint32_t v = b.peek(15);
        {
          const uint32_t q = __builtin_amdgcn_readlane(t6, (int)(v & 63u));
          if (q) {                                        // one or two literals out of the register-resident first level
            const uint32_t cnt2 = q >> 24;
            b.drop((int)((q >> 16) & 0xFFu));
            if (lane == 0) { f.ring[w & RM] = (uint16_t)(q & 0xFFu); if (cnt2 == 2) f.ring[(w + 1) & RM] = (uint16_t)((q >> 8) & 0xFFu); }
            w += cnt2;
            if ((w & 63u) < cnt2) {
              if (w > (1u << 28)) { status = GZ_ERR_ROOM; bad = true; go = 0u; continue; }
              if (w - wf >= 64) flush(w & ~63u);
            }
            continue;
          }
        }
        const uint32_t e = UNI(f.lit_fast[v & ((1u << GZ_FAST_LIT) - 1)]);
        int s;
        if (e) { b.drop((int)(e >> 12)); s = (int)(e & 0xFFF); }
        else { s = slow_decode(f.sc.lit, b, 288); if (s < 0) { status = GZ_ERR_CODE; bad = true; go = 0u; continue; } }
        if (s < 256) {
          if (lane == 0) f.ring[w & RM] = (uint16_t)s;
          ++w;
          if ((w & 63u) == 0) {
            if (w > (1u << 28)) { status = GZ_ERR_ROOM; bad = true; go = 0u; continue; }       // (one wavefront would work for seconds: the CPU path)
            if (w - wf >= 64) flush(w);
          }
          continue;
        }
        if (s == 256) { go = 0u; continue; }
        s -= 257;
        if (s >= 29) { status = GZ_ERR_CODE; bad = true; go = 0u; continue; }
        const uint32_t lt = UNI(f.len_tab[s]);
        const uint32_t len = (lt & 0xFFFFu) + b.get((int)(lt >> 16));
        const uint32_t v2 = b.peek(15);
        const uint32_t e2 = UNI(f.dist_fast[v2 & ((1u << GZ_FAST_DIST) - 1)]);
        int ds;
        if (e2) { b.drop((int)(e2 >> 12)); ds = (int)(e2 & 0xFFF); }
        else { ds = slow_decode(f.sc.dist, b, 30); }
        if (ds < 0 || ds >= 30) { status = GZ_ERR_CODE; bad = true; go = 0u; continue; }
        const uint32_t dt = UNI(f.dist_tab[ds]);
        const uint32_t dist = (dt & 0xFFFFu) + b.get((int)(dt >> 16));
        if (!match(len, dist)) { bad = true; go = 0u; continue; }
      }
      if (bad) break;
    } else { status = GZ_ERR_TYPE; break; }
    if (last) { status = GZ_FINAL; break; }
    if (b.bitpos() >= stop_bit) break;
  }
  flush(w);
  if (lane == 0) { U.end_bit = b.bitpos(); U.n_sym = w; U.status = status; U.pad = (uint32_t)(__builtin_readcyclecounter() - cyc0); }
}

// One member: a wavefront per deflate block, all resident, and the largest block is the kernel's time - wavefronts that share a SIMD
// slow each other down, so the registers the compiler likes to take (158: three wavefronts per SIMD) are left to it.
template <bool WRITE>
__global__ void __launch_bounds__(64) gz_decode_kernel(const uint8_t* __restrict__ d, uint64_t n, GzUnit* __restrict__ units, uint32_t n_units,
                                                      uint16_t* __restrict__ sym, const uint32_t* __restrict__ which, int all_known, uint32_t u_base) {
  __shared__ Fast f;
  gz_decode_body<WRITE>(f, d, n, units, n_units, sym, which, all_known, u_base);
}
// Block gzip: thousands of small members, more wavefronts than the chip holds - 128 registers, four wavefronts per SIMD (measured on
// 2 x 4 842 members: 22-27 ms against 28-32)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
gz_decode_members_kernel(const uint8_t* __restrict__ d, uint64_t n, GzUnit* __restrict__ units, uint32_t n_units, uint16_t* __restrict__ sym) {
  __shared__ Fast f;
  gz_decode_body<true>(f, d, n, units, n_units, sym, nullptr, 1);
}

// ---- 3: windows in chain order --------------------------------------------------------------------------------------------
// The 32 KiB in front of unit u: the last 32 Ki bytes of (the window in front of unit u - 1 ++ unit u - 1 resolved against it) - a
// chain through all units.  One unit's step is a MAP of window positions: entry t of the new window is a byte, or position i of the
// old one (0x8000 | i).  Maps compose, so the chain is cut into groups of ~sqrt(units) units:
//   gz_compose_kernel   a block per group, all groups at once: the units' maps composed in chain order, from the identity on, in LDS
//                       (two maps of 64 KiB); pmap[u] = the map from the window in front of the GROUP to the window in front of unit u,
//                       qmap[g] = the same over the whole group;
//   gz_chain_kernel     one block: the windows in front of the groups (wg[g]), one qmap applied per step - sqrt(units) steps where
//                       the chain over single units took one step per unit (10.8 ms of the 53 for 2 404 units; now 0.5);
//   gz_resolve2_kernel  looks a marker up in two steps: pmap[u], and - where that entry is a position again - wg[group of u].
__global__ void __launch_bounds__(1024) gz_compose_kernel(const GzUnit* __restrict__ units, uint32_t n_units, uint32_t group, const uint16_t* __restrict__ sym,
                                                         uint16_t* __restrict__ pmap, uint16_t* __restrict__ qmap) {
  extern __shared__ uint16_t gz_lds_maps[];             // 2 x 32 Ki entries
  uint16_t* cur = gz_lds_maps; uint16_t* nxt = gz_lds_maps + 32768;
  const int tid = threadIdx.x;
  const uint32_t u0 = blockIdx.x * group, u1 = u0 + group < n_units ? u0 + group : n_units;
  if (u0 >= n_units) return;
  for (int i = tid; i < 32768; i += 1024) cur[i] = (uint16_t)(0x8000u | (uint32_t)i);
  // the unit's symbols are loaded ONE UNIT AHEAD (its descriptor two ahead): a step then waits for no global load
  uint16_t x[32];
  unsigned long long m = units[u0].n_sym, so = units[u0].sym_off;
  unsigned long long m1 = u0 + 1 < u1 ? units[u0 + 1].n_sym : 0, so1 = u0 + 1 < u1 ? units[u0 + 1].sym_off : 0;
  auto fetch = [&](unsigned long long mm, unsigned long long off) {
    const uint16_t* s = sym + off;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const unsigned long long q = (unsigned long long)(tid + 1024 * j) + mm;   // byte (t + m) of (old window ++ resolved unit) = byte t of the new one
      x[j] = q >= 32768 ? s[q - 32768] : (uint16_t)0xFFFF;
    }
  };
  fetch(m, so);
  __syncthreads();
  for (uint32_t u = u0; u < u1; ++u) {
    uint16_t* keep = pmap + (size_t)u * 32768;
    for (int i = tid * 8; i < 32768; i += 8192) *(uint4*)(keep + i) = *(const uint4*)(cur + i);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int t = tid + 1024 * j;
      const unsigned long long q = (unsigned long long)t + m;
      nxt[t] = q < 32768 ? cur[q] : (x[j] < 0x8000u ? x[j] : cur[x[j] & 0x7FFFu]);
    }
    m = m1; so = so1;
    if (u + 1 < u1) fetch(m, so);
    if (u + 2 < u1) { m1 = units[u + 2].n_sym; so1 = units[u + 2].sym_off; }
    __syncthreads();
    uint16_t* t_ = cur; cur = nxt; nxt = t_;
  }
  uint16_t* q = qmap + (size_t)blockIdx.x * 32768;
  for (int i = tid * 8; i < 32768; i += 8192) *(uint4*)(q + i) = *(const uint4*)(cur + i);
}

// wg[g] = the 32 KiB in front of group g (bytes); the member starts with an empty (zero) window that nothing refers to, a later stripe
// of it (GzJob::next) with the last `have` bytes of the text so far (init = where they start).
__global__ void __launch_bounds__(1024) gz_chain_kernel(const uint16_t* __restrict__ qmap, uint32_t n_groups, uint8_t* __restrict__ wg,
                                                       const uint8_t* __restrict__ init, uint32_t have) {
  __shared__ uint8_t w0[32768];
  __shared__ uint8_t w1[32768];
  uint8_t* cur = w0; uint8_t* nxt = w1;
  const int tid = threadIdx.x;
  for (int i = tid; i < 32768; i += 1024) cur[i] = (uint32_t)i >= 32768u - have ? init[(uint32_t)i - (32768u - have)] : (uint8_t)0;
  uint16_t x[32];
  auto fetch = [&](uint32_t g) {
    const uint16_t* q = qmap + (size_t)g * 32768;
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = q[tid + 1024 * j];
  };
  if (n_groups) fetch(0);
  __syncthreads();
  for (uint32_t g = 0; g < n_groups; ++g) {
    uint8_t* keep = wg + (size_t)g * 32768;
    for (int i = tid * 8; i < 32768; i += 8192) *(uint2*)(keep + i) = *(const uint2*)(cur + i);
#pragma unroll
    for (int j = 0; j < 32; ++j) nxt[tid + 1024 * j] = x[j] < 0x8000u ? (uint8_t)x[j] : cur[x[j] & 0x7FFFu];
    if (g + 1 < n_groups) fetch(g + 1);
    __syncthreads();
    uint8_t* t_ = cur; cur = nxt; nxt = t_;
  }
}

// (four symbols a thread, one aligned 32-bit store: a byte a lane was 64 bytes a store instruction - 1.3 TB/s where the text is 1.25 GB;
// the symbols in front of the unit's first aligned word and behind its last one go out byte by byte)
template <class FIX>
__device__ __forceinline__ void gz_bytes_out(uint8_t* __restrict__ text, uint64_t off, uint64_t m, uint32_t part, uint32_t blocks_per_unit,
                                             const uint16_t* __restrict__ sy, FIX&& fix) {
  auto one = [&](uint64_t i) -> uint32_t { return fix((uint32_t)sy[i]); };
  const uint64_t lead = (4u - (uint32_t)(off & 3u)) & 3u;
  const uint64_t lead_n = lead < m ? lead : m;
  if (part == 0 && threadIdx.x < lead_n) text[off + threadIdx.x] = (uint8_t)one(threadIdx.x);
  const uint64_t words = (m - lead_n) >> 2;
  uint32_t* __restrict__ tw = (uint32_t*)(text + off + lead_n);
  for (uint64_t q = (uint64_t)part * 256 + threadIdx.x; q < words; q += (uint64_t)blocks_per_unit * 256) {
    const uint64_t i = lead_n + 4 * q;
    uint2 v;
    __builtin_memcpy(&v, sy + i, 8);              // four symbols in one load (2-byte aligned: the compiler knows what the target takes)
    const uint32_t b0 = fix(v.x & 0xFFFFu), b1 = fix(v.x >> 16), b2 = fix(v.y & 0xFFFFu), b3 = fix(v.y >> 16);
    tw[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  }
  const uint64_t tail = lead_n + 4 * words;
  if (part == 0 && threadIdx.x < 4 && tail + threadIdx.x < m) text[off + tail + threadIdx.x] = (uint8_t)one(tail + threadIdx.x);
}

__global__ void __launch_bounds__(256) gz_resolve2_kernel(const GzUnit* __restrict__ units, uint32_t n_units, uint32_t group, const uint16_t* __restrict__ sym,
                                                         const uint16_t* __restrict__ pmap, const uint8_t* __restrict__ wg, uint8_t* __restrict__ text,
                                                         uint32_t blocks_per_unit) {
  const uint32_t u = blockIdx.x / blocks_per_unit, part = blockIdx.x % blocks_per_unit;
  if (u >= n_units) return;
  const uint64_t m = units[u].n_sym, off = units[u].out_off;
  const uint16_t* s = sym + units[u].sym_off;
  const uint16_t* pm = pmap + (size_t)u * 32768;
  const uint8_t* w = wg + (size_t)(u / group) * 32768;
  gz_bytes_out(text, off, m, part, blocks_per_unit, s, [&](uint32_t x) -> uint32_t {
    if (x >= 0x8000u) { x = pm[x & 0x7FFFu]; if (x >= 0x8000u) x = w[x & 0x7FFFu]; }
    return x & 0xFFu;
  });
}

// ---- 4: markers out, bytes out --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gz_resolve_kernel(const GzUnit* __restrict__ units, uint32_t n_units, const uint16_t* __restrict__ sym,
                                                        const uint8_t* __restrict__ win_in, uint8_t* __restrict__ text, uint32_t blocks_per_unit) {
  const uint32_t u = blockIdx.x / blocks_per_unit, part = blockIdx.x % blocks_per_unit;
  if (u >= n_units) return;
  const uint64_t m = units[u].n_sym, off = units[u].out_off;
  const uint16_t* s = sym + units[u].sym_off;
  const uint8_t* w = win_in + (size_t)u * 32768;
  gz_bytes_out(text, off, m, part, blocks_per_unit, s, [&](uint32_t x) -> uint32_t {
    return x < 0x8000u ? x & 0xFFu : (uint32_t)w[x & 0x7FFFu];
  });
}

size_t gzip_header(const uint8_t* p, size_t n) {      // offset of the deflate data, 0 = not a gzip member this code takes
  if (n < 18 + 2 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
  const int flg = p[3];
  if (flg & 0xE0) return 0;
  size_t q = 10;
  if (flg & 4) {
    if (q + 2 > n) return 0;
    const size_t xlen = p[q] | (p[q + 1] << 8);
    // block gzip (BGZF: a 'BC' subfield with the member's size) is thousands of members: the host inflates those block-parallel
    for (size_t x = q + 2; x + 4 <= q + 2 + xlen && x + 4 <= n;) {
      if (p[x] == 'B' && p[x + 1] == 'C') return 0;
      x += 4 + (p[x + 2] | ((size_t)p[x + 3] << 8));
    }
    q += 2 + xlen;
  }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  return q + 8 < n ? q : 0;
}

// ---- CRC-32 of the text (the member's trailer holds it; gunzip checks it, and so does this path) ------------------------------------
// Every thread the CRC of its own 4-KiB piece (table in LDS, a byte a step); the host strings the pieces together: the CRC of A || B
// is the CRC of A carried over len(B) zero bytes - a 32 x 32 matrix over GF(2), the same for every whole piece - xor the CRC of B.
constexpr uint32_t GZ_CRC_PIECE = 4096;

__global__ void __launch_bounds__(256) gz_crc_kernel(const uint8_t* __restrict__ d, uint64_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t tab[256];
  uint32_t c = threadIdx.x;
  for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  tab[threadIdx.x] = c;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x, o = i * GZ_CRC_PIECE;
  if (o >= n) return;
  const uint32_t len = (uint32_t)(n - o < GZ_CRC_PIECE ? n - o : GZ_CRC_PIECE);
  uint32_t crc = 0xFFFFFFFFu;
  const uint8_t* p = d + o;                   // (the text buffer is 256-byte aligned: whole 16-byte words while they last)
  uint32_t j = 0;
  for (; j + 16 <= len; j += 16) {
    const uint4 v = *(const uint4*)(p + j);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = w[q];
#pragma unroll
      for (int b = 0; b < 4; ++b) { crc = tab[(crc ^ x) & 0xFFu] ^ (crc >> 8); x >>= 8; }
    }
  }
  for (; j < len; ++j) crc = tab[(crc ^ p[j]) & 0xFFu] ^ (crc >> 8);
  out[i] = crc ^ 0xFFFFFFFFu;
}

struct Gf2 { uint32_t row[32]; };           // row[i] = image of bit i
uint32_t gf2_times(const Gf2& m, uint32_t v) { uint32_t s = 0; for (int i = 0; v; v >>= 1, ++i) if (v & 1u) s ^= m.row[i]; return s; }
Gf2 gf2_mul(const Gf2& a, const Gf2& b) { Gf2 c; for (int i = 0; i < 32; ++i) c.row[i] = gf2_times(a, b.row[i]); return c; }   // a after b
Gf2 crc_zero_bytes(uint64_t n_bytes) {      // the operator "n_bytes zero bytes follow" on a (reflected) CRC-32 register
  Gf2 one;                                   // one zero bit
  one.row[0] = 0xEDB88320u;
  for (int i = 1; i < 32; ++i) one.row[i] = 1u << (i - 1);
  Gf2 p = gf2_mul(one, one); p = gf2_mul(p, p); p = gf2_mul(p, p);      // eight bits
  Gf2 r;
  for (int i = 0; i < 32; ++i) r.row[i] = 1u << i;                      // identity
  for (; n_bytes; n_bytes >>= 1) { if (n_bytes & 1) r = gf2_mul(p, r); p = gf2_mul(p, p); }
  return r;
}
uint32_t crc_of_pieces(const std::vector<uint32_t>& piece, uint64_t n) {
  if (piece.empty()) return 0;
  const Gf2 whole = crc_zero_bytes(GZ_CRC_PIECE);
  static uint32_t tab[4][256];                 // the whole-piece operator byte by byte (77 k pieces in a 316-MB text: 32 conditional xors each took 10 ms)
  static std::once_flag once;
  std::call_once(once, [&] { for (int b = 0; b < 4; ++b) for (uint32_t v = 0; v < 256; ++v) tab[b][v] = gf2_times(whole, v << (8 * b)); });
  uint32_t crc = piece[0];
  for (size_t i = 1; i < piece.size(); ++i) {
    const uint64_t len = i + 1 < piece.size() ? GZ_CRC_PIECE : n - (uint64_t)i * GZ_CRC_PIECE;
    crc = (len == GZ_CRC_PIECE ? tab[0][crc & 255u] ^ tab[1][(crc >> 8) & 255u] ^ tab[2][(crc >> 16) & 255u] ^ tab[3][crc >> 24]
                               : gf2_times(crc_zero_bytes(len), crc)) ^ piece[i];
  }
  return crc;
}

// ---- block gzip (BGZF: what bgzip / samtools write): thousands of small members, each with its compressed size in a 'BC' subfield of
// its header and its CRC-32 and length in its trailer.  The members are independent - a wavefront each, no back-reference can leave
// a member, no speculation, no windows - and their places in the text are known before anything is decoded.
struct BgzfMember { uint64_t data_off, data_end; uint32_t isize, crc; uint64_t text_off; };

bool bgzf_members(const uint8_t* p, size_t n, std::vector<BgzfMember>& out, uint64_t& total) {
  size_t o = 0;
  total = 0;
  while (o < n) {
    if (o + 18 > n || p[o] != 0x1f || p[o + 1] != 0x8b || p[o + 2] != 8 || !(p[o + 3] & 4) || (p[o + 3] & 0xFA)) return false;
    const size_t xlen = p[o + 10] | ((size_t)p[o + 11] << 8);
    if (o + 12 + xlen > n) return false;
    size_t bsize = 0;
    for (size_t x = o + 12; x + 4 <= o + 12 + xlen;) {
      const size_t sl = p[x + 2] | ((size_t)p[x + 3] << 8);
      if (p[x] == 'B' && p[x + 1] == 'C' && sl == 2 && x + 6 <= o + 12 + xlen) bsize = (p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
      x += 4 + sl;
    }
    if (bsize < 12 + xlen + 8 + 2 || o + bsize > n) return false;
    BgzfMember m;
    m.data_off = o + 12 + xlen; m.data_end = o + bsize - 8;
    const uint8_t* t = p + o + bsize - 8;
    m.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    m.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    if (m.isize > 65536u) return false;
    m.text_off = total; total += m.isize;
    out.push_back(m);
    o += bsize;
  }
  return !out.empty();
}

// The device code of this file is loaded when its first kernel is launched (the runtime defers it, file by file): 4 ms that belong
// in front of the first call, with the reservation
__global__ void gz_warm_kernel(uint32_t* p) { if (p && threadIdx.x == 1000) *p = 0; }

// CRC-32 of every member's text: a wavefront per member, lane j its 4-KiB piece j (a member holds 64 KiB of text at most)
__global__ void __launch_bounds__(64) gz_crc_members_kernel(const uint8_t* __restrict__ text, const unsigned long long* __restrict__ off,
                                                            const uint32_t* __restrict__ size, uint32_t n_members, uint32_t* __restrict__ out) {
  __shared__ uint32_t tab[256];
  for (uint32_t v = threadIdx.x; v < 256; v += 64) {
    uint32_t c = v;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    tab[v] = c;
  }
  __syncthreads();
  const uint32_t m = blockIdx.x, j = threadIdx.x;
  if (m >= n_members || j >= 16) return;
  const uint32_t sz = size[m], from = j * GZ_CRC_PIECE;
  if (from >= sz) return;
  const uint32_t len = sz - from < GZ_CRC_PIECE ? sz - from : GZ_CRC_PIECE;
  const uint8_t* p = text + off[m] + from;
  uint32_t crc = 0xFFFFFFFFu;
#pragma unroll 8
  for (uint32_t i = 0; i < len; ++i) crc = tab[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  out[m * 16 + j] = crc ^ 0xFFFFFFFFu;
}

// Buffers set up ahead of a call (mic_gz_reserve): one scratch block for everything the call needs on the way and the text buffer.
// A fresh gigabyte of device memory takes the driver tens of milliseconds (longer right behind a table build, whose freed pages it
// still wipes) - as long as the decode; the command line reserves while its database loads, as it does for its ingest slots.
struct GzReserve {
  mic_engine* eng = nullptr; size_t gz_bytes = 0;
  char* scratch = nullptr; size_t scratch_bytes = 0;
  uint8_t* text = nullptr; size_t text_bytes = 0;
  bool taken = false;
};
std::mutex g_res_mu;
std::vector<GzReserve> g_res;

// A call runs on one of the engine's two copy streams, taken in turn: the two mates of a pair are inflated by two host threads at
// once and want a stream each, and creating one costs two milliseconds (a hardware queue) - more than the upload of a file.  (Two
// calls that meet on one stream take turns on it; nothing else depends on which one a call gets.)
std::atomic<unsigned> g_call_no{0};
hipStream_t call_stream(mic_engine* e) {
  hipStream_t up, down;
  mic_engine_copy_streams(e, &up, &down);
  return (g_call_no.fetch_add(1) & 1u) ? down : up;
}

// Host buffers that large copies from the device land in are kept for the next call instead of freed: the runtime pins the pages of
// such a destination for the copy, an allocation of this size is a mapping of its own, and unmapping pages that were pinned a
// moment ago makes the driver take the process's queues off the device and put them back - the next kernel, whoever launches it,
// starts 4 ms late (measured: the first kernel behind a call).
std::mutex g_hostbuf_mu;
std::vector<std::vector<uint32_t>> g_hostbufs;
std::vector<uint32_t> take_hostbuf(size_t n) {
  std::vector<uint32_t> v;
  {
    std::lock_guard<std::mutex> lk(g_hostbuf_mu);
    for (size_t i = 0; i < g_hostbufs.size(); ++i)
      if (g_hostbufs[i].capacity() >= n) { v.swap(g_hostbufs[i]); g_hostbufs.erase(g_hostbufs.begin() + (ptrdiff_t)i); break; }
  }
  v.resize(n);
  return v;
}
void keep_hostbuf(std::vector<uint32_t>& v) {
  if (v.capacity() == 0) return;
  std::lock_guard<std::mutex> lk(g_hostbuf_mu);
  if (g_hostbufs.size() < 8) { g_hostbufs.emplace_back(); g_hostbufs.back().swap(v); }
}

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
unsigned long long sym_bound_of(size_t n, uint32_t n_chunks) { return 8ull * (n + n_chunks) + 16384ull * n_chunks; }

#define GZTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = mic_set_error(e_ == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "gzip on the device: %s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

}  // namespace

namespace {
int inflate_bgzf(mic_engine* e, const uint8_t* p, size_t gz_bytes, void** d_text, size_t* n_text) {
  std::vector<BgzfMember> mem;
  uint64_t total = 0;
  if (!bgzf_members(p, gz_bytes, mem, total)) return mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: not a whole block-gzip file");
  int rc = MIC_OK;
  hipStream_t s = nullptr;
  uint8_t* d_in = nullptr; GzUnit* d_units = nullptr; uint16_t* d_sym = nullptr; uint8_t* d_out = nullptr;
  unsigned long long* d_off = nullptr; uint32_t* d_size = nullptr; uint32_t* d_crc = nullptr;
  std::vector<GzUnit> units;
  std::vector<uint32_t> unit_member;
  std::vector<unsigned long long> h_off;
  std::vector<uint32_t> h_size, h_crc;
  char* arena = nullptr; size_t arena_left = 0;
  std::vector<void*> owned;
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    for (GzReserve& r : g_res)
      if (r.eng == e && r.gz_bytes == gz_bytes && !r.taken) {
        r.taken = true; arena = r.scratch; arena_left = r.scratch_bytes;
        if (r.text && (size_t)total + 64 <= r.text_bytes) { d_out = r.text; r.text = nullptr; }
        break;
      }
  }
  auto dev_alloc = [&](void** ptr, size_t bytes) -> hipError_t {
    const size_t b = up256(bytes);
    if (b <= arena_left) { *ptr = arena; arena += b; arena_left -= b; return hipSuccess; }
    const hipError_t he = hipMalloc(ptr, bytes);
    if (he == hipSuccess) owned.push_back(*ptr);
    return he;
  };
  const bool timing = getenv("MIC_GZ_TIMING") != nullptr;
  struct timespec tq0; clock_gettime(CLOCK_MONOTONIC, &tq0);
  auto lap = [&](const char* what) {
    if (!timing) return;
    if (s) hipStreamSynchronize(s);
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "[gz] %s: %.3f ms\n", what, (t1.tv_sec - tq0.tv_sec) * 1e3 + (t1.tv_nsec - tq0.tv_nsec) / 1e6);
    tq0 = t1;
  };
  for (size_t i = 0; i < mem.size(); ++i) {
    if (mem[i].isize == 0) { if (mem[i].crc != 0) { rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done; } continue; }   // (the end-of-file marker)
    GzUnit u; memset(&u, 0, sizeof(u));
    u.start_bit = mem[i].data_off * 8; u.stop_bit = ~0ull;
    u.sym_off = mem[i].text_off; u.sym_cap = mem[i].isize; u.out_off = mem[i].text_off;
    units.push_back(u); unit_member.push_back((uint32_t)i);
    h_off.push_back(mem[i].text_off); h_size.push_back(mem[i].isize);
  }
  s = call_stream(e);
  GZTRY(dev_alloc((void**)&d_in, gz_bytes + 16));
  GZTRY(hipMemsetAsync(d_in + gz_bytes, 0, 16, s));
  GZTRY(hipMemcpyAsync(d_in, p, gz_bytes, hipMemcpyHostToDevice, s));
  if (!d_out) GZTRY(hipMalloc(&d_out, (size_t)total + 64));
  if (!units.empty()) {
    const uint32_t nu = (uint32_t)units.size();
    GZTRY(dev_alloc((void**)&d_units, (size_t)nu * sizeof(GzUnit)));
    GZTRY(dev_alloc((void**)&d_sym, ((size_t)total + 8) * 2));
    GZTRY(dev_alloc((void**)&d_off, (size_t)nu * 8));
    GZTRY(dev_alloc((void**)&d_size, (size_t)nu * 4));
    GZTRY(dev_alloc((void**)&d_crc, (size_t)nu * 16 * 4));
    GZTRY(hipMemcpyAsync(d_units, units.data(), (size_t)nu * sizeof(GzUnit), hipMemcpyHostToDevice, s));
    GZTRY(hipMemcpyAsync(d_off, h_off.data(), (size_t)nu * 8, hipMemcpyHostToDevice, s));
    GZTRY(hipMemcpyAsync(d_size, h_size.data(), (size_t)nu * 4, hipMemcpyHostToDevice, s));
    lap("upload (block gzip)");
    gz_decode_members_kernel<<<nu, 64, 0, s>>>(d_in, gz_bytes, d_units, nu, d_sym);
    GZTRY(hipGetLastError());
    GZTRY(hipMemcpyAsync(units.data(), d_units, (size_t)nu * sizeof(GzUnit), hipMemcpyDeviceToHost, s));
    lap("decode (a wavefront per member)");
    // no member can hold a marker (nothing in front of it is known to be nothing): the resolve pass only narrows symbols to bytes
    gz_resolve_kernel<<<nu * 2u, 256, 0, s>>>(d_units, nu, d_sym, (const uint8_t*)d_sym, d_out, 2);
    GZTRY(hipGetLastError());
    gz_crc_members_kernel<<<nu, 64, 0, s>>>(d_out, d_off, d_size, nu, d_crc);
    GZTRY(hipGetLastError());
    h_crc = take_hostbuf((size_t)nu * 16);
    GZTRY(hipMemcpyAsync(h_crc.data(), d_crc, (size_t)nu * 16 * 4, hipMemcpyDeviceToHost, s));
    GZTRY(hipStreamSynchronize(s));
    lap("bytes + CRC-32 of the members");
    {
      const Gf2 whole = crc_zero_bytes(GZ_CRC_PIECE);
      uint32_t wt[4][256];                                      // (the whole-piece operator byte by byte, as in crc_of_pieces)
      for (int b = 0; b < 4; ++b) for (uint32_t v = 0; v < 256; ++v) wt[b][v] = gf2_times(whole, v << (8 * b));
      uint32_t tail_len = 0; Gf2 tail = whole;                 // (the members of a file are of one size, but for the last)
      for (uint32_t i = 0; i < nu; ++i) {
        const GzUnit& u = units[i];
        const BgzfMember& m = mem[unit_member[i]];
        if (u.status != GZ_FINAL || u.n_sym != m.isize || (u.end_bit + 7) / 8 != m.data_end) {
          rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done;
        }
        const uint32_t np = (m.isize + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE;
        uint32_t crc = h_crc[(size_t)i * 16];
        for (uint32_t j = 1; j < np; ++j) {
          const uint32_t len = j + 1 < np ? GZ_CRC_PIECE : m.isize - j * GZ_CRC_PIECE;
          if (len == GZ_CRC_PIECE) crc = wt[0][crc & 255u] ^ wt[1][(crc >> 8) & 255u] ^ wt[2][(crc >> 16) & 255u] ^ wt[3][crc >> 24];
          else { if (len != tail_len) { tail = crc_zero_bytes(len); tail_len = len; } crc = gf2_times(tail, crc); }
          crc ^= h_crc[(size_t)i * 16 + j];
        }
        if (crc != m.crc) { rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done; }     // (gunzip: "crc error")
      }
    }
    lap("members checked");
  } else GZTRY(hipStreamSynchronize(s));
  if (timing) fprintf(stderr, "[gz] block gzip: %zu bytes -> %llu bytes, %zu members\n", gz_bytes, (unsigned long long)total, mem.size());
  *d_text = d_out; d_out = nullptr; *n_text = (size_t)total;
done:
  if (s) { hipStreamSynchronize(s); s = nullptr; }
  for (void* q : owned) hipFree(q);
  if (d_out) hipFree(d_out);
  keep_hostbuf(h_crc);
  return rc;
}
}  // namespace

// ---- one plain member, in stripes ---------------------------------------------------------------------------------------------------
// The units of a member (one per found block) are taken through decode -> stitch -> windows -> resolve in STRIPES of consecutive units:
// when a stripe is done its text is final - the window in front of its first unit is the last 32 KiB of the text so far - and the
// caller can index and classify it while the next stripe decodes (mic_gz_stream_*, round 6).  One stripe of all
// units is the whole-member call (mic_gz_inflate_device): the same code.  The decode of a unit is a serial thing of ~6 ms whatever
// the number of units, up to the ~3 000 wavefronts the chip holds: a stripe is about a thousand units - its wavefronts then do not
// share a SIMD, and what is left of the chip runs the query kernels of the stripe before.
struct GzJob {
  mic_engine* eng = nullptr;
  const uint8_t* p = nullptr; size_t gz_bytes = 0, n = 0, hdr = 0;
  uint32_t crc = 0, isize = 0;
  hipStream_t s = nullptr;
  uint8_t* d_in = nullptr; unsigned long long* d_start = nullptr; GzUnit* d_units = nullptr; uint16_t* d_sym = nullptr;
  uint16_t* d_pmap = nullptr; uint16_t* d_qmap = nullptr; uint8_t* d_wg = nullptr; uint8_t* d_out = nullptr; GzUnit* d_chain = nullptr;
  uint32_t* d_crc = nullptr;
  uint32_t n_chunks = 0;
  std::vector<GzUnit> units;
  unsigned long long sym_total = 0, sym_bound = 0, sym_used = 0;      // regions handed out at the start / the bound / symbols the buffer holds now
  uint64_t total = 0;                                                 // bytes of text that are final
  unsigned long long pos = 0; bool final = false;                     // the chain so far: where it ends, whether it met the last block
  size_t next_unit = 0, per_stripe = 0, chain_len = 0;
  size_t pmap_units = 0, wg_groups = 0;
  size_t crc_pieces_done = 0;
  std::vector<uint32_t> piece;
  // stripes: every stripe's decode is launched at the start, on two streams in turn - the decode of a unit is a serial thing of ~10 ms
  // and only all units in flight at once fill the chip; a stripe's stitching, windows and resolve follow on `s` when its event fires
  hipStream_t sd[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_stripe;
  hipEvent_t ev_units = nullptr;
  bool striped = false;
  char* arena = nullptr; size_t arena_left = 0;
  std::vector<void*> owned;
  bool timing = false;
  struct timespec tq0;

  hipError_t dev_alloc(void** ptr, size_t bytes) {
    const size_t b = up256(bytes);
    if (b <= arena_left) { *ptr = arena; arena += b; arena_left -= b; return hipSuccess; }
    const hipError_t he = hipMalloc(ptr, bytes);
    if (he == hipSuccess) owned.push_back(*ptr);
    return he;
  }
  void dev_free(void* ptr) {                      // (memory of the reservation stays where it is)
    for (size_t i = 0; i < owned.size(); ++i) if (owned[i] == ptr) { hipFree(ptr); owned.erase(owned.begin() + (ptrdiff_t)i); return; }
  }
  void lap(const char* what) {
    if (!timing) return;
    if (s) hipStreamSynchronize(s);
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "[gz] %s: %.3f ms\n", what, (t1.tv_sec - tq0.tv_sec) * 1e3 + (t1.tv_nsec - tq0.tv_nsec) / 1e6);
    tq0 = t1;
  }
  bool done() const { return next_unit >= units.size() || final; }

  // upload, find the blocks, set the units up.  stripes = 0: one stripe (everything at once)
  int open(mic_engine* e, const void* gz, size_t bytes, uint32_t stripes) {
    int rc = MIC_OK;
    eng = e; p = (const uint8_t*)gz; gz_bytes = bytes;
    hdr = gzip_header(p, gz_bytes);
    if (!hdr) return mic_set_error(MIC_E_UNSUPPORTED, "not a plain gzip member");
    n = gz_bytes - 8;                                   // deflate data + nothing else expected in front of the trailer
    crc = (uint32_t)p[n] | ((uint32_t)p[n + 1] << 8) | ((uint32_t)p[n + 2] << 16) | ((uint32_t)p[n + 3] << 24);
    isize = (uint32_t)p[n + 4] | ((uint32_t)p[n + 5] << 8) | ((uint32_t)p[n + 6] << 16) | ((uint32_t)p[n + 7] << 24);
    n_chunks = (uint32_t)((n + GZ_CHUNK - 1) / GZ_CHUNK);
    std::vector<unsigned long long> h_start(n_chunks);
    // device memory: out of the reservation made for a file of this size, if there is one (mic_gz_reserve), else allocated here
    {
      std::lock_guard<std::mutex> lk(g_res_mu);
      for (GzReserve& r : g_res)
        if (r.eng == e && r.gz_bytes == gz_bytes && !r.taken) {
          r.taken = true; arena = r.scratch; arena_left = r.scratch_bytes;
          if (r.text && (size_t)isize + 64 <= r.text_bytes) { d_out = r.text; r.text = nullptr; }      // (the text is this call's from here on)
          break;
        }
    }
    timing = getenv("MIC_GZ_TIMING") != nullptr;
    clock_gettime(CLOCK_MONOTONIC, &tq0);
    s = call_stream(e);
    GZTRY(dev_alloc((void**)&d_in, n + 16));
    GZTRY(dev_alloc((void**)&d_start, (size_t)n_chunks * 8));
    GZTRY(hipMemsetAsync(d_in + n, 0, 16, s));
    // (tried in round 6: the file up in 16 / 32-MiB pieces on the other copy stream, the finder following piece by piece - 16.1 ms from
    // pageable memory, 15.0 ms from a pinned copy of the file, against 4.3-5.0 + 8.7 one after the other: the finder's 29 039 chunk
    // wavefronts pack the chip for 14 rounds when launched at once; a launch per piece is a round of its own with its own tail)
    GZTRY(hipMemcpyAsync(d_in, p, n, hipMemcpyHostToDevice, s));
    lap("upload");
    gz_find_kernel<<<n_chunks, 64, 0, s>>>(d_in, n, (uint64_t)hdr * 8, n_chunks, d_start);
    GZTRY(hipGetLastError());
    GZTRY(hipMemcpyAsync(h_start.data(), d_start, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
    // The big buffers are allocated while the upload and the finder run (a fresh gigabyte takes the driver 30-40 ms: as much as the
    // decode): the symbol buffer by its bound - every unit gets 8 x its compressed span + 16 Ki symbols, there are at most n_chunks
    // units - and the text by the member's ISIZE, which the decode has to arrive at anyway
    sym_bound = sym_bound_of(n, n_chunks);
    GZTRY(dev_alloc((void**)&d_sym, (sym_bound + 8) * 2));
    if (!d_out && (unsigned long long)isize <= 1100ull * n) GZTRY(hipMalloc(&d_out, (size_t)isize + 64));
    GZTRY(hipStreamSynchronize(s));
    lap("find blocks (+ symbol and text buffers allocated)");
    for (uint32_t c = 0; c < n_chunks; ++c) {
      if (h_start[c] == ~0ull) continue;
      GzUnit u; memset(&u, 0, sizeof(u));
      u.start_bit = h_start[c];
      units.push_back(u);
    }
    if (units.empty()) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: no block found"); goto done; }
    {
      // every unit gets a region of the symbol buffer sized by its compressed span (8 x + 16 Ki symbols: FASTQ inflates 3-6 x);
      // a unit that needs more is counted exactly by the same pass and decoded again into a region of its own
      unsigned long long so = 0;
      for (size_t i = 0; i < units.size(); ++i) {
        units[i].stop_bit = i + 1 < units.size() ? units[i + 1].start_bit : ~0ull;
        const unsigned long long span = ((i + 1 < units.size() ? units[i + 1].start_bit : (unsigned long long)n * 8) - units[i].start_bit) / 8 + 1;
        units[i].sym_off = so; units[i].sym_cap = span * 8 + 16384;
        so += units[i].sym_cap;
      }
      sym_total = so;
    }
    if (sym_total > sym_bound) { rc = mic_set_error(MIC_E_HIP, "gzip on the device: symbol regions beyond their bound"); goto done; }
    sym_used = sym_total + 8;
    per_stripe = units.size();
    if (stripes > 1) per_stripe = std::max<size_t>((units.size() + stripes - 1) / stripes, 64);
    if (const char* env = getenv("MIC_GZ_STRIPE_UNITS")) { const long v = atol(env); if (v >= 16) per_stripe = (size_t)v; }
    GZTRY(dev_alloc((void**)&d_units, units.size() * sizeof(GzUnit)));
    GZTRY(hipMemcpyAsync(d_units, units.data(), units.size() * sizeof(GzUnit), hipMemcpyHostToDevice, s));
    GZTRY(dev_alloc((void**)&d_chain, per_stripe * sizeof(GzUnit)));
    {
      uint32_t group = 1;
      while ((size_t)group * group < per_stripe) ++group;                        // units per group of the window chain: ~sqrt(units)
      pmap_units = per_stripe; wg_groups = (per_stripe + group - 1) / group;
      GZTRY(dev_alloc((void**)&d_pmap, pmap_units * (size_t)65536));
      GZTRY(dev_alloc((void**)&d_qmap, wg_groups * (size_t)65536));
      GZTRY(dev_alloc((void**)&d_wg, wg_groups * (size_t)32768));
    }
    pos = units[0].start_bit;
    if (stripes > 1 && per_stripe < units.size()) {
      striped = true;
      const size_t n_str = (units.size() + per_stripe - 1) / per_stripe;
      GZTRY(mic_event_get(&ev_units, false));
      GZTRY(hipEventRecord(ev_units, s));
      for (int i = 0; i < 2; ++i) { GZTRY(mic_stream_get(&sd[i])); GZTRY(hipStreamWaitEvent(sd[i], ev_units, 0)); }
      ev_stripe.assign(n_str, nullptr);
      for (size_t i = 0; i < n_str; ++i) {
        const size_t a = i * per_stripe, b = std::min(units.size(), a + per_stripe);
        GZTRY(mic_event_get(&ev_stripe[i], false));
        gz_decode_kernel<true><<<(unsigned)(b - a), 64, 0, sd[i & 1]>>>(d_in, n, d_units, (uint32_t)units.size(), d_sym, nullptr, 0, (uint32_t)a);
        GZTRY(hipGetLastError());
        GZTRY(hipEventRecord(ev_stripe[i], sd[i & 1]));
      }
    }
    {
      static std::once_flag lds_once; static hipError_t lds_rc = hipSuccess;
      std::call_once(lds_once, [] { lds_rc = hipFuncSetAttribute((const void*)gz_compose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536); });
      GZTRY(lds_rc);
    }
  done:
    return rc;
  }

  // the next stripe: on MIC_OK the first `total` bytes of the text are final (the stream has been waited for)
  int next() {
    int rc = MIC_OK;
    if (done()) return MIC_OK;
    const size_t a = next_unit, b = std::min(units.size(), a + per_stripe);
    const bool last_stripe = b == units.size();
    const uint64_t text_from = total;
    std::vector<GzUnit> chain;
    std::vector<uint32_t> redo;
    unsigned long long extra = 0;
    next_unit = b;
    if (striped) GZTRY(hipStreamWaitEvent(s, ev_stripe[a / per_stripe], 0));
    else {
      gz_decode_kernel<true><<<(unsigned)(b - a), 64, 0, s>>>(d_in, n, d_units, (uint32_t)units.size(), d_sym, nullptr, 0, (uint32_t)a);
      GZTRY(hipGetLastError());
    }
    GZTRY(hipMemcpyAsync(units.data() + a, d_units + a, (b - a) * sizeof(GzUnit), hipMemcpyDeviceToHost, s));
    GZTRY(hipStreamSynchronize(s));
    lap("decode");
    if (timing) {
      unsigned long long mx = 0, sum = 0, mxspan = 0, mxc = 0, sumc = 0;
      for (size_t i = a; i < b; ++i) { const GzUnit& u = units[i]; if (u.n_sym > mx) mx = u.n_sym; sum += u.n_sym; if (u.end_bit - u.start_bit > mxspan) mxspan = u.end_bit - u.start_bit; if (u.pad > mxc) mxc = u.pad; sumc += u.pad; }
      fprintf(stderr, "[gz] cycles per unit: mean %llu, max %llu; cycles per output symbol %.1f\n", sumc / (b - a), mxc, (double)sumc / (double)(sum ? sum : 1));
      fprintf(stderr, "[gz] units: %zu, symbols per unit: mean %llu, max %llu; longest span %llu bytes\n", b - a, sum / (b - a), mx, mxspan / 8);
    }
    // the chain: a unit is taken iff the chain so far ends exactly on its start; one that starts inside the chain was a false find
    for (size_t i = a; i < b && !final; ++i) {
      GzUnit& u = units[i];
      if (u.start_bit < pos) continue;
      if (u.start_bit > pos || u.status > GZ_FINAL) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: the units do not stitch (unit %zu, status %u)", i, u.status); goto done; }
      if (u.n_sym > u.sym_cap) { redo.push_back((uint32_t)i); u.sym_off = sym_used + extra; u.sym_cap = u.n_sym; extra += u.n_sym; }
      u.out_off = total; total += u.n_sym;
      // more text than the trailer's 32-bit length: a text of 4 GiB or more (the length wraps), or damage - the CPU inflater's case either way
      if (d_out && total > (uint64_t)isize) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: more text than the trailer's length"); goto done; }
      chain.push_back(u);
      pos = u.end_bit;
      final = u.status == GZ_FINAL;
    }
    if (final || last_stripe) {
      // behind the last block: the trailer, right there (bits up to the next byte are padding)
      if (!final || (pos + 7) / 8 != n) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: more than one member, or data behind the last block"); goto done; }
      if ((uint32_t)total != isize) { rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done; }
    }
    if (!redo.empty()) {
      // (a second symbol buffer behind the first would need the first one's copy: the units that overflowed get room behind the
      // symbols there are, addressed through the same base pointer - offsets are relative to d_sym, so the buffer grows as ONE)
      uint16_t* bigger = nullptr;
      GZTRY(dev_alloc((void**)&bigger, (sym_used + extra + 8) * 2));
      GZTRY(hipMemcpyAsync(bigger, d_sym, sym_used * 2, hipMemcpyDeviceToDevice, s));
      GZTRY(hipStreamSynchronize(s));
      dev_free(d_sym); d_sym = bigger; sym_used += extra;
      uint32_t* d_which = nullptr;
      GZTRY(dev_alloc((void**)&d_which, redo.size() * 4));
      hipError_t e1 = hipMemcpyAsync(d_which, redo.data(), redo.size() * 4, hipMemcpyHostToDevice, s);
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(d_units + a, units.data() + a, (b - a) * sizeof(GzUnit), hipMemcpyHostToDevice, s);
      if (e1 == hipSuccess) { gz_decode_kernel<true><<<(unsigned)redo.size(), 64, 0, s>>>(d_in, n, d_units, (uint32_t)units.size(), d_sym, d_which, 0, 0); e1 = hipGetLastError(); }
      std::vector<GzUnit> back(b - a);
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(back.data(), d_units + a, (b - a) * sizeof(GzUnit), hipMemcpyDeviceToHost, s);
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
      dev_free(d_which);
      GZTRY(e1);
      for (uint32_t i : redo)
        if (back[i - a].status != units[i].status || back[i - a].end_bit != units[i].end_bit || back[i - a].n_sym != units[i].n_sym) {
          rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: the second decode of unit %u differs", i); goto done;
        }
      for (GzUnit& c : chain) for (uint32_t i : redo) if (c.start_bit == units[i].start_bit) { c.sym_off = units[i].sym_off; c.sym_cap = units[i].sym_cap; }
      lap("decode again (units that overflowed their region)");
    }
    if (!d_out) {                    // (a trailer that promises more than 1100 x the file: the text is allocated once its length is known)
      if (!done()) { rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: no text buffer for a member in stripes"); goto done; }
      GZTRY(hipMalloc(&d_out, total + 64));
    }
    if (!chain.empty()) {
      uint32_t group = 1;
      while ((size_t)group * group < chain.size()) ++group;
      const uint32_t n_groups = (uint32_t)((chain.size() + group - 1) / group);
      if (chain.size() > pmap_units || n_groups > wg_groups) { rc = mic_set_error(MIC_E_HIP, "gzip on the device: a stripe beyond its window maps"); goto done; }
      GZTRY(hipMemcpyAsync(d_chain, chain.data(), chain.size() * sizeof(GzUnit), hipMemcpyHostToDevice, s));
      gz_compose_kernel<<<n_groups, 1024, 2 * 65536, s>>>(d_chain, (uint32_t)chain.size(), group, d_sym, d_pmap, d_qmap);
      GZTRY(hipGetLastError());
      // the window in front of the stripe: the end of the text so far (nothing in front of the member's first byte)
      const uint32_t have = text_from >= 32768 ? 32768u : (uint32_t)text_from;
      gz_chain_kernel<<<1, 1024, 0, s>>>(d_qmap, n_groups, d_wg, d_out + text_from - have, have);
      GZTRY(hipGetLastError());
      lap("windows");
      gz_resolve2_kernel<<<(unsigned)chain.size() * 8u, 256, 0, s>>>(d_chain, (uint32_t)chain.size(), group, d_sym, d_pmap, d_wg, d_out, 8);
      GZTRY(hipGetLastError());
      chain_len += chain.size();
      lap("resolve");
    }
    {
      // CRC-32 of the whole 4-KiB pieces that became final (the last stripe: and of the rest)
      const size_t n_pieces = done() ? (size_t)((total + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE) : (size_t)(total / GZ_CRC_PIECE);
      if (!d_crc) { GZTRY(dev_alloc((void**)&d_crc, ((size_t)isize / GZ_CRC_PIECE + 2) * 4)); piece = take_hostbuf((size_t)isize / GZ_CRC_PIECE + 2); }
      if (n_pieces > crc_pieces_done) {
        const size_t from = crc_pieces_done, cnt = n_pieces - from;
        const uint64_t upto = done() ? total : (uint64_t)n_pieces * GZ_CRC_PIECE;
        gz_crc_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, s>>>(d_out + (uint64_t)from * GZ_CRC_PIECE, upto - (uint64_t)from * GZ_CRC_PIECE, d_crc + from);
        GZTRY(hipGetLastError());
        GZTRY(hipMemcpyAsync(piece.data() + from, d_crc + from, cnt * 4, hipMemcpyDeviceToHost, s));
        crc_pieces_done = n_pieces;
      }
      GZTRY(hipStreamSynchronize(s));
      lap("CRC-32 of the pieces");
      if (done()) {
        piece.resize(crc_pieces_done);
        if (crc_of_pieces(piece, total) != crc) { rc = mic_set_error(MIC_E_INVALID, "Failed to uncompress input objects."); goto done; }   // (gunzip: "crc error")
        if (timing) fprintf(stderr, "[gz] %zu bytes -> %llu bytes, %u chunks, %zu units found, %zu in the chain\n", gz_bytes, (unsigned long long)total, n_chunks, units.size(), chain_len);
      }
    }
  done:
    return rc;
  }

  // everything but the text; the text too unless the caller took it (take_text)
  uint8_t* take_text() { uint8_t* t = d_out; d_out = nullptr; return t; }
  void close() {
    for (hipStream_t& q : sd) if (q) { mic_stream_put(q); q = nullptr; }        // (drained first: decodes of stripes never asked for)
    for (hipEvent_t& q : ev_stripe) if (q) { mic_event_put(q); q = nullptr; }
    if (ev_units) { mic_event_put(ev_units); ev_units = nullptr; }
    if (s) { hipStreamSynchronize(s); s = nullptr; }
    for (void* q : owned) hipFree(q);
    owned.clear();
    if (d_out) { hipFree(d_out); d_out = nullptr; }
    keep_hostbuf(piece);
    lap("buffers freed");
  }
};

extern "C" int mic_gz_inflate_device(mic_engine* e, const void* gz, size_t gz_bytes, void** d_text, size_t* n_text, uint32_t* crc32_expected) {
  if (!e || !gz || !d_text || !n_text) return mic_set_error(MIC_E_INVALID, "null argument");
  *d_text = nullptr; *n_text = 0;
  MicTable t_; int sc_, ncu_, dev_, k_; uint32_t nt_;
  int rc = mic_engine_table(e, &t_, &sc_, &ncu_, &dev_, &k_, &nt_);
  if (rc) return rc;
  if (hipSetDevice(dev_) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  const uint8_t* p = (const uint8_t*)gz;
  if (gz_bytes >= 18 && p[0] == 0x1f && p[1] == 0x8b && (p[3] & 4) && p[12] == 'B' && p[13] == 'C') {      // block gzip
    if (crc32_expected) *crc32_expected = 0;                                                               // (checked per member, inside)
    return inflate_bgzf(e, p, gz_bytes, d_text, n_text);
  }
  GzJob job;
  rc = job.open(e, gz, gz_bytes, 0);
  if (crc32_expected) *crc32_expected = job.crc;
  while (rc == MIC_OK && !job.done()) rc = job.next();
  if (rc == MIC_OK) { *n_text = (size_t)job.total; *d_text = job.take_text(); }
  job.close();
  return rc;
}

// The same in stripes: open uploads, finds the blocks and hands the (whole) text buffer out; every next() makes a further piece of it
// final.  A failure in a later stripe (MIC_E_UNSUPPORTED: units that do not stitch, a second member) comes after text was handed
// out: the caller that has used it starts over on its CPU inflater.
struct mic_gz_stream { GzJob job; int device = 0; bool text_taken = false; };

extern "C" int mic_gz_stream_open(mic_engine* e, const void* gz, size_t gz_bytes, uint32_t stripes, mic_gz_stream** out, void** d_text, size_t* n_text) {
  if (!e || !gz || !out || !d_text || !n_text) return mic_set_error(MIC_E_INVALID, "null argument");
  *out = nullptr; *d_text = nullptr; *n_text = 0;
  MicTable t_; int sc_, ncu_, dev_, k_; uint32_t nt_;
  int rc = mic_engine_table(e, &t_, &sc_, &ncu_, &dev_, &k_, &nt_);
  if (rc) return rc;
  if (hipSetDevice(dev_) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  const uint8_t* p = (const uint8_t*)gz;
  if (gz_bytes >= 18 && p[0] == 0x1f && p[1] == 0x8b && (p[3] & 4) && p[12] == 'B' && p[13] == 'C')
    return mic_set_error(MIC_E_UNSUPPORTED, "block gzip is inflated in one piece (mic_gz_inflate_device)");
  mic_gz_stream* h = new mic_gz_stream;
  h->device = dev_;
  rc = h->job.open(e, gz, gz_bytes, stripes ? stripes : 1);
  if (rc == MIC_OK && !h->job.d_out) rc = mic_set_error(MIC_E_UNSUPPORTED, "gzip on the device: a trailer that promises more than 1100 x the file");
  if (rc != MIC_OK) { h->job.close(); delete h; return rc; }
  *out = h; *d_text = h->job.d_out; *n_text = (size_t)h->job.isize;
  return MIC_OK;
}

extern "C" int mic_gz_stream_next(mic_gz_stream* h, size_t* n_final, int* done) {
  if (!h || !n_final || !done) return mic_set_error(MIC_E_INVALID, "null argument");
  if (hipSetDevice(h->device) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  const int rc = h->job.next();
  *n_final = rc == MIC_OK ? (size_t)h->job.total : 0;
  *done = h->job.done() ? 1 : 0;
  return rc;
}

extern "C" int mic_gz_stream_close(mic_gz_stream* h, int keep_text) {
  if (!h) return MIC_OK;
  (void)hipSetDevice(h->device);
  if (keep_text) (void)h->job.take_text();       // (the caller's from here on: mic_gz_free_text)
  h->job.close();
  delete h;
  return MIC_OK;
}

extern "C" int mic_gz_reserve(mic_engine* e, size_t gz_bytes, uint32_t isize) {
  if (!e || gz_bytes < 18) return mic_set_error(MIC_E_INVALID, "bad argument");
  MicTable t_; int sc_, ncu_, dev_, k_; uint32_t nt_;
  int rc = mic_engine_table(e, &t_, &sc_, &ncu_, &dev_, &k_, &nt_);
  if (rc) return rc;
  if (hipSetDevice(dev_) != hipSuccess) return mic_set_error(MIC_E_HIP, "hipSetDevice failed");
  const size_t n = gz_bytes - 8;
  const uint32_t n_chunks = (uint32_t)((n + GZ_CHUNK - 1) / GZ_CHUNK);
  GzReserve r;
  r.eng = e; r.gz_bytes = gz_bytes;
  // input, block starts, units and chain, symbols by their bound, window maps (64 KiB a unit) for a third of the chunks (a block of gzip's is three
  // chunks and more; a file of smaller blocks gets the rest of its windows from hipMalloc)
  r.scratch_bytes = up256(n + 16) + up256((size_t)n_chunks * 8) + 2 * up256((size_t)n_chunks * sizeof(GzUnit)) +
                    up256((sym_bound_of(n, n_chunks) + 8) * 2) + up256(((size_t)n_chunks / 3 + 64) * 65536) + 2 * up256(((size_t)n_chunks / 24 + 64) * 65536) + up256(((size_t)isize / GZ_CRC_PIECE + 2) * 4) + 4096;
  hipError_t he = hipMalloc(&r.scratch, r.scratch_bytes);
  if (he == hipSuccess && (unsigned long long)isize <= 1100ull * n) {
    r.text_bytes = (size_t)isize + 64;
    he = hipMalloc(&r.text, r.text_bytes);
  }
  if (he != hipSuccess) {
    if (r.scratch) hipFree(r.scratch);
    (void)hipGetLastError();
    return mic_set_error(he == hipErrorOutOfMemory ? MIC_E_NOMEM : MIC_E_HIP, "gzip on the device: reservation: %s", hipGetErrorString(he));
  }
  gz_warm_kernel<<<1, 64>>>((uint32_t*)r.scratch);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_res_mu);
  g_res.push_back(r);
  return MIC_OK;
}

extern "C" uint64_t mic_gz_reserve_bytes(size_t gz_bytes, uint32_t isize) {
  if (gz_bytes < 18) return 0;
  const size_t n = gz_bytes - 8;
  const uint32_t n_chunks = (uint32_t)((n + GZ_CHUNK - 1) / GZ_CHUNK);
  return (uint64_t)(n + 16 + (size_t)n_chunks * (8 + 2 * sizeof(GzUnit)) + (sym_bound_of(n, n_chunks) + 8) * 2 + ((size_t)n_chunks / 3 + 64) * 32768 + ((size_t)isize / GZ_CRC_PIECE + 2) * 4 + 8192) +
         (uint64_t)isize + 64;
}

extern "C" int mic_gz_release(mic_engine* e) {
  std::vector<GzReserve> mine;
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    for (size_t i = 0; i < g_res.size();)
      if (g_res[i].eng == e) { mine.push_back(g_res[i]); g_res.erase(g_res.begin() + (ptrdiff_t)i); } else ++i;
  }
  for (GzReserve& r : mine) { if (r.scratch) hipFree(r.scratch); if (r.text) hipFree(r.text); }
  return MIC_OK;
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
