// pgz.hpp - parallel inflate of ORDINARY gzip files (one deflate stream per member, no block index), host threads only.
//
// The reference leaves compressed input to `gunzip` in classify_metagenome.sh:116-142 (copy, gunzip, classify the plain
// file).  One zlib stream inflates ~0.5 GB/s; the query kernel consumes the text of 10 M reads in milliseconds, so for gzip
// input the inflate IS the run time.  A deflate stream cannot be entered in the middle in general - a block may refer to
// the 32 KiB before it - but it can be entered SPECULATIVELY (the two-stage scheme of pugz / rapidgzip, restated here):
//   1. the compressed bytes are cut into chunks; for every chunk but the first a thread looks for the next deflate block
//      with dynamic Huffman codes by trying every bit offset: header fields in range, the code-length code complete, the
//      literal/length and distance codes complete and with an end-of-block symbol (false positives do not survive this);
//   2. every chunk is decoded from its block start to the block start of the next chunk into 16-bit symbols: a byte, or -
//      for a back-reference that reaches in front of the chunk - a MARKER naming the position in the unknown 32 KiB window;
//   3. in file order, the last 32 KiB of every chunk are resolved against the window handed on by the chunk before (cheap,
//      sequential), which gives every chunk its window;
//   4. in parallel again, every chunk replaces its markers, narrows to bytes and takes its CRC-32.
// A chunk whose speculation fails (no block found, a decode error, not ending on the next chunk's block start) is decoded
// again by the thread that stitches, from where the chunk before really ended: always correct, only slower.  Every member's
// CRC-32 and length are checked against its trailer (crc32_combine over the chunks), so a damaged file ends with an error
// exactly as with zlib, and a wrong speculation cannot pass silently.
#ifndef MIC_PGZ_HPP
#define MIC_PGZ_HPP

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pgz {

// ---- bits, LSB first -----------------------------------------------------------------------------------------------------
struct Bits {
  const uint8_t* p; size_t n; size_t pos = 0;     // pos = next byte to load
  uint64_t buf = 0; int cnt = 0;                  // cnt valid bits in buf
  bool over = false;                              // read past the end of the data
  Bits(const uint8_t* d, size_t len, uint64_t bitpos) : p(d), n(len) {
    pos = (size_t)(bitpos >> 3);
    const int skip = (int)(bitpos & 7);
    refill();
    buf >>= skip; cnt -= skip;
  }
  inline void refill() {
    if (pos + 8 <= n) {                           // eight bytes at once; the bytes that do not fit are read again next time
      uint64_t v;
      memcpy(&v, p + pos, 8);
      buf |= v << cnt;
      const int add = (63 - cnt) >> 3;
      pos += (size_t)add; cnt += add * 8;
      return;
    }
    while (cnt <= 56) {                           // the last bytes of the data (zeros behind them; `over` once well past the end)
      if (pos < n) buf |= (uint64_t)p[pos] << cnt; else if (pos >= n + 8) { over = true; }
      ++pos; cnt += 8;
    }
  }
  inline uint32_t peek(int k) { if (cnt < k) refill(); return (uint32_t)(buf & ((1ull << k) - 1)); }
  inline void drop(int k) { buf >>= k; cnt -= k; }
  inline uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }
  inline void skip(int k) { if (cnt < k) refill(); drop(k); }
  uint64_t bitpos() const { return (uint64_t)pos * 8 - (uint64_t)cnt; }
  void align() { drop(cnt & 7); }
};

// ---- canonical Huffman code: 11-bit direct table, longer codes by the count / symbol arrays --------------------------------
struct Huff {
  static const int FAST = 11;
  uint16_t fast[1 << FAST];      // len << 12 | symbol, 0 = longer than FAST bits (or no such code)
  uint16_t count[16], symbol[320];
  int max_len = 0;
  // returns 0 = complete code, 1 = incomplete, -1 = over-subscribed
  int build(const uint8_t* len, int n) {
    memset(count, 0, sizeof(count));
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    int left = 1; max_len = 0;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return -1; if (count[l]) max_len = l; }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    memset(fast, 0, sizeof(fast));
    // canonical codes, bit-reversed into the table
    uint32_t code = 0; int idx = 0;
    for (int l = 1; l <= 15; ++l) {
      for (int c = 0; c < count[l]; ++c, ++idx, ++code) {
        if (l > FAST) continue;
        uint32_t r = 0;
        for (int b = 0; b < l; ++b) r |= ((code >> b) & 1u) << (l - 1 - b);
        const uint16_t e = (uint16_t)((l << 12) | symbol[idx]);
        for (uint32_t v = r; v < (1u << FAST); v += 1u << l) fast[v] = e;
      }
      code <<= 1;
    }
    return left > 0 ? 1 : 0;
  }
  inline int decode(Bits& b) const {
    const uint32_t v = b.peek(15);
    const uint16_t e = fast[v & ((1u << FAST) - 1)];
    if (e) { b.drop(e >> 12); return e & 0xFFF; }
    // longer code: walk the lengths (puff.c's way), bit by bit from FAST + 1
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= max_len; ++l) {
      code |= (int)((v >> (l - 1)) & 1u);
      const int c = count[l];
      if (code - c < first) { b.drop(l); return symbol[index + (code - first)]; }
      index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
  }
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Reads the header of a dynamic block (after BFINAL / BTYPE) into the two codes.  strict: everything a real encoder emits
// must hold (used by the block finder); returns false on anything invalid.
// Kraft sum of a set of code lengths (in units of 2^-15): 1 << 15 for a complete code
static inline uint32_t kraft(const uint8_t* len, int n) {
  uint32_t sum = 0;
  for (int i = 0; i < n; ++i) if (len[i]) sum += 1u << (15 - len[i]);
  return sum;
}

// Reads the header of a dynamic block (after BFINAL / BTYPE) into the two codes.  strict: what a real encoder emits must
// hold - every code complete (used by the block finder, where the cheap tests come first: of the bit offsets that are not
// block starts almost none survives the code-length code's Kraft sum, so the finder costs tens of nanoseconds per offset).
static inline bool read_dynamic(Bits& b, Huff& lit, Huff& dist, bool strict) {
  const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t cl[19]; memset(cl, 0, sizeof(cl));
  for (int i = 0; i < hclen; ++i) cl[kClOrder[i]] = (uint8_t)b.get(3);
  const uint32_t ks = kraft(cl, 19);
  if (ks > (1u << 15)) return false;
  if (ks < (1u << 15)) {                                       // incomplete: legal only as one single code (zlib's rule), never from an encoder
    if (strict) return false;
    int used = 0; for (int i = 0; i < 19; ++i) used += cl[i] != 0;
    if (used != 1) return false;
  }
  // the code-length code has at most 7 bits: a 128-entry table
  uint8_t pre[128];
  {
    memset(pre, 0, sizeof(pre));
    uint32_t code = 0;
    for (int l = 1; l <= 7; ++l) {
      for (int sym = 0; sym < 19; ++sym) {
        if (cl[sym] != l) continue;
        uint32_t r = 0;
        for (int q = 0; q < l; ++q) r |= ((code >> q) & 1u) << (l - 1 - q);
        for (uint32_t v = r; v < 128; v += 1u << l) pre[v] = (uint8_t)((l << 5) | sym);
        ++code;
      }
      code <<= 1;
    }
  }
  uint8_t len[320]; int n = 0;
  while (n < hlit + hdist) {
    const uint8_t e = pre[b.peek(7)];
    if (!e) return false;
    b.drop(e >> 5);
    const int s = e & 31;
    if (s < 16) { len[n++] = (uint8_t)s; continue; }
    int rep, val = 0;
    if (s == 16) { if (n == 0) return false; val = len[n - 1]; rep = 3 + (int)b.get(2); }
    else if (s == 17) rep = 3 + (int)b.get(3);
    else rep = 11 + (int)b.get(7);
    if (n + rep > hlit + hdist) return false;
    while (rep--) len[n++] = (uint8_t)val;
  }
  if (b.over || len[256] == 0) return false;                   // no end-of-block code
  if (kraft(len, hlit) != (1u << 15)) return false;            // zlib rejects over-subscribed and incomplete literal/length codes
  const uint32_t kd = kraft(len + hlit, hdist);
  if (kd > (1u << 15)) return false;
  if (kd < (1u << 15)) {                                       // incomplete distance code: a single code (or none) is legal
    int used = 0; for (int i = 0; i < hdist; ++i) used += len[hlit + i] != 0;
    if (used > 1) return false;
  }
  lit.build(len, hlit);
  dist.build(len + hlit, hdist);
  return true;
}

static inline void fixed_codes(Huff& lit, Huff& dist) {
  uint8_t len[288];
  for (int i = 0; i < 144; ++i) len[i] = 8;
  for (int i = 144; i < 256; ++i) len[i] = 9;
  for (int i = 256; i < 280; ++i) len[i] = 7;
  for (int i = 280; i < 288; ++i) len[i] = 8;
  lit.build(len, 288);
  uint8_t dl[30]; for (int i = 0; i < 30; ++i) dl[i] = 5;
  dist.build(dl, 30);
}

// ---- CRC-32 (the gzip polynomial) by carry-less multiplication ------------------------------------------------------
// zlib's table-driven crc32 runs at ~1 GB/s per core - a sixth of the time of a round once the decode itself is spread over
// the threads.  Folding four 128-bit lanes with PCLMULQDQ (Gopal et al., "Fast CRC computation for generic polynomials using
// PCLMULQDQ"; constants for the reflected polynomial 0x1DB710641) runs at memory speed.  Used when the CPU has it, checked
// against zlib's crc32 on every start-up; otherwise zlib's.
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_clmul(uint32_t crc, const uint8_t* buf, size_t len) {
  // len >= 64 and a multiple of 16; crc in its pre-/post-conditioned (~) form, as zlib's works inside
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
  const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
  const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
  x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
  x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
  x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = k1k2;
  buf += 64; len -= 64;
  while (len >= 64) {
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    buf += 64; len -= 64;
  }
  // four lanes -> one
  x0 = k3k4;
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {
    x2 = _mm_loadu_si128((const __m128i*)buf);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16; len -= 16;
  }
  // 128 -> 64 bits
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = k5k0;
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  // Barrett reduction to 32 bits
  x0 = poly;
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

static inline bool clmul_crc_ok() {
  static const bool ok = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t t[1024 + 48];
    for (size_t i = 0; i < sizeof(t); ++i) t[i] = (uint8_t)(i * 131u + (i >> 3) * 7u + 5u);
    for (size_t len : {(size_t)64, (size_t)80, (size_t)128, (size_t)1024, (size_t)1072}) {
      const uint32_t want = (uint32_t)crc32(0x12345678u, t, (uInt)len);
      if (~crc32_clmul(~0x12345678u, t, len) != want) return false;
    }
    return true;
  }();
  return ok;
}

// crc32() of zlib, any length
static inline uint32_t crc32_fast(uint32_t crc, const uint8_t* p, size_t n) {
  if (n >= 256 && clmul_crc_ok()) {
    const size_t body = n & ~(size_t)15;
    crc = ~crc32_clmul(~crc, p, body);
    p += body; n -= body;
  }
  while (n) { const size_t k = std::min<size_t>(n, (size_t)1 << 30); crc = (uint32_t)crc32(crc, p, (uInt)k); p += k; n -= k; }
  return crc;
}

// ---- gzip member header / trailer -------------------------------------------------------------------------------------
// returns the byte offset of the deflate data, or 0 if `p + off` is not a gzip header that fits
static inline size_t gzip_header(const uint8_t* p, size_t n, size_t off) {
  if (off + 18 > n || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8) return 0;
  const int flg = p[off + 3];
  size_t q = off + 10;
  if (flg & 4) { if (q + 2 > n) return 0; const size_t xlen = p[q] | (p[q + 1] << 8); q += 2 + xlen; }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  return q < n ? q : 0;
}

struct MemberEnd { uint64_t out_off; uint32_t crc, isize; };      // a member ended after out_off bytes of the chunk's output

// ---- one chunk ------------------------------------------------------------------------------------------------------
struct Chunk {
  uint64_t start_bit = 0;         // where its decode starts: a block start (or, for the file's first chunk, the member header)
  bool found = false;             // phase 1 found a plausible block start
  bool ok = false;                // phase 2 decoded it without error
  uint64_t end_bit = 0;           // the block start at which it stopped (first block boundary at or after its stop mark)
  bool at_eof = false;            // ... or the end of the file
  // decoded symbols: bytes, or markers 0x8000 | position in the unknown 32 KiB window.  A plain malloc'd array that keeps its
  // capacity from round to round: fresh multi-megabyte vectors per chunk per round serialise the threads in mmap / page faults
  uint16_t* sym = nullptr; size_t n_sym = 0, cap = 0;
  std::vector<MemberEnd> ends;
  std::vector<uint32_t> piece_crc; std::vector<uint64_t> piece_len;   // CRC-32 / length of the pieces between member ends
  uint64_t out_off = 0;           // where its bytes go in the round's output
  Chunk() {}
  Chunk(const Chunk&) = delete;
  Chunk& operator=(const Chunk&) = delete;
  ~Chunk() { free(sym); }
  void reset() { start_bit = end_bit = 0; found = ok = at_eof = false; n_sym = 0; ends.clear(); piece_crc.clear(); piece_len.clear(); out_off = 0; }
  bool room(size_t w, size_t need) {
    if (w + need <= cap) return true;
    // 2-MiB aligned and advised as huge pages: a chunk's symbols are tens of megabytes of fresh memory in the first round, and
    // taking them in 4-KiB page faults costs as much as decoding them
    const size_t nc = (std::max(cap * 2, w + need + (1u << 16)) * sizeof(uint16_t) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* p = nullptr;
    if (posix_memalign(&p, (size_t)2 << 20, nc) != 0 || !p) return false;
    madvise(p, nc, MADV_HUGEPAGE);
    if (w) memcpy(p, sym, w * sizeof(uint16_t));
    free(sym);
    sym = (uint16_t*)p; cap = nc / sizeof(uint16_t);
    return true;
  }
};

// Decodes blocks from bit position `from` until the first block boundary at or after bit `stop_at` (the expected case: exactly
// at it), or the end of the file.  file_start: `from` is the gzip header of the file's first member.  window: the 32 KiB in
// front of `from` when known (the first chunk of a round), else nullptr: references in front of the chunk become markers.
// Returns false on any error (a false block start, damaged data, more than max_out symbols).
static inline bool decode_blocks(const uint8_t* data, size_t n, uint64_t from, uint64_t stop_at, const uint8_t* window, bool file_start,
                                 Chunk& c, size_t max_out) {
  size_t hist0 = 0;                 // output index before which no history exists (a member started there)
  bool before_ok = !file_start;     // references in front of the chunk are possible (window or markers)
  Bits b(data, n, from);
  if (file_start) {
    const size_t d = gzip_header(data, n, (size_t)(from >> 3));
    if (!d) return false;
    b = Bits(data, n, (uint64_t)d * 8);
  }
  Huff lit, dist;
  size_t w = c.n_sym;               // write position
  bool first = true;
  for (;;) {
    if (!first && b.bitpos() >= stop_at) { c.end_bit = b.bitpos(); c.n_sym = w; return true; }
    first = false;
    const uint32_t bfinal = b.get(1), btype = b.get(2);
    if (b.over) return false;
    if (btype == 0) {
      b.align();
      const uint32_t len = b.get(16), nlen = b.get(16);
      if ((len ^ 0xFFFFu) != nlen) return false;
      const size_t bytepos = (size_t)(b.bitpos() >> 3);
      if (bytepos + len > n) return false;
      if (!c.room(w, len)) return false;
      for (uint32_t i = 0; i < len; ++i) c.sym[w + i] = data[bytepos + i];
      w += len;
      b = Bits(data, n, (uint64_t)(bytepos + len) * 8);
    } else if (btype == 3) {
      return false;
    } else {
      if (btype == 1) fixed_codes(lit, dist);
      else if (!read_dynamic(b, lit, dist, false)) return false;
      for (;;) {
        if (b.over) return false;
        if (w + 260 > c.cap && !c.room(w, 260)) return false;
        int s = lit.decode(b);
        if (s < 0) return false;
        if (s < 256) { c.sym[w++] = (uint16_t)s; continue; }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) return false;
        const uint32_t len = kLenBase[s] + b.get(kLenExtra[s]);
        const int ds = dist.decode(b);
        if (ds < 0 || ds >= 30) return false;
        const size_t d = (size_t)kDistBase[ds] + b.get(kDistExtra[ds]);
        uint16_t* o = c.sym;
        if (d <= w - hist0) {
          if (d >= len) memcpy(o + w, o + w - d, (size_t)len * 2);                     // no overlap
          else if (d == 1) { const uint16_t v = o[w - 1]; for (uint32_t i = 0; i < len; ++i) o[w + i] = v; }   // a run
          else for (uint32_t i = 0; i < len; ++i) o[w + i] = o[w + i - d];
        } else {
          if (hist0 > 0 || !before_ok) return false;                  // in front of the member's first byte
          if (d - w > 32768) return false;                            // farther back than any window reaches
          for (uint32_t i = 0; i < len; ++i) {
            const size_t at = w + i;                                  // output index being written
            if (at >= d) o[at] = o[at - d];
            else {
              const size_t wi = 32768 - (d - at);                     // index into the 32 KiB in front of the chunk
              o[at] = window ? (uint16_t)window[wi] : (uint16_t)(0x8000u | wi);
            }
          }
        }
        w += len;
        if (w > max_out) return false;
      }
    }
    if (w > max_out) return false;
    if (bfinal) {
      // member trailer, then another member or the end of the file
      b.align();
      size_t bytepos = (size_t)(b.bitpos() >> 3);
      if (bytepos + 8 > n) return false;
      MemberEnd e;
      e.out_off = w;
      e.crc = (uint32_t)data[bytepos] | ((uint32_t)data[bytepos + 1] << 8) | ((uint32_t)data[bytepos + 2] << 16) | ((uint32_t)data[bytepos + 3] << 24);
      e.isize = (uint32_t)data[bytepos + 4] | ((uint32_t)data[bytepos + 5] << 8) | ((uint32_t)data[bytepos + 6] << 16) | ((uint32_t)data[bytepos + 7] << 24);
      c.ends.push_back(e);
      bytepos += 8;
      while (bytepos < n && data[bytepos] == 0) ++bytepos;            // zero padding after a member (gzip tolerates it)
      if (bytepos >= n) { c.end_bit = (uint64_t)n * 8; c.at_eof = true; c.n_sym = w; return true; }
      const size_t d = gzip_header(data, n, bytepos);
      if (!d) return false;                                           // trailing garbage
      hist0 = w; window = nullptr;                                    // the new member's history starts here
      b = Bits(data, n, (uint64_t)d * 8);
    }
  }
}

// Phase 1: the first block with dynamic codes that starts at or after bit `from` (searching up to `limit_bit`).
static inline bool find_block(const uint8_t* data, size_t n, uint64_t from, uint64_t limit_bit, uint64_t& at) {
  Huff lit, dist;
  for (uint64_t bit = from; bit < limit_bit; ++bit) {
    const size_t byte = (size_t)(bit >> 3);
    if (byte + 8 >= n) return false;
    // BFINAL = 0, BTYPE = 2 (LSB first: the three bits read 0, 0, 1): the cheap test first
    const uint32_t three = (((uint32_t)data[byte] | ((uint32_t)data[byte + 1] << 8)) >> (bit & 7)) & 7u;
    if (three != 4u) continue;
    Bits b(data, n, bit + 3);
    if (!read_dynamic(b, lit, dist, true)) continue;
    bool good = true;                                                  // a real block decodes: try a few hundred symbols
    for (int i = 0; i < 300 && good; ++i) {
      if (b.over) { good = false; break; }
      int s = lit.decode(b);
      if (s < 0) { good = false; break; }
      if (s < 256) continue;
      if (s == 256) break;
      s -= 257;
      if (s >= 29) { good = false; break; }
      b.skip(kLenExtra[s]);
      const int ds = dist.decode(b);
      if (ds < 0 || ds >= 30) { good = false; break; }
      b.skip(kDistExtra[ds]);
    }
    if (!good) continue;
    at = bit;
    return true;
  }
  return false;
}

// CPUs this process may really use: the hardware's threads, cut to the cgroup's CPU quota (cpu.max) and the affinity mask.
// More runnable threads than the quota allows get the whole process throttled (CFS): a pool sized by the hardware count
// alone ran 2.5 x slower in a 16-CPU container on a 256-thread host.
static inline unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32]; unsigned long long period = 0;
    if (fscanf(f, "%31s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const unsigned long long quota = strtoull(q, nullptr, 10);
      const unsigned c = (unsigned)((quota + period - 1) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {          // cgroup v1
    long long quota = -1, period = 100000;
    if (fscanf(g, "%lld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    if (quota > 0 && period > 0) { const unsigned c = (unsigned)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
  }
  return n;
}

// A pool that lives as long as one file: the phases of a round are short, thread creation per phase would show.
class Pool {
 public:
  explicit Pool(unsigned threads) {
    // (pinning one worker per physical core - the scheduler may put two on the hardware threads of one core under a CPU-time
    // quota - was measured on the GPU box: 5-15 % SLOWER; the workers float)
    for (unsigned t = 1; t < threads; ++t) th_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // f() is run by every thread of the pool and by the caller; returns when all are done
  void run(const std::function<void()>& f) {
    { std::lock_guard<std::mutex> g(m_); job_ = &f; left_ = th_.size(); ++gen_; }
    cv_.notify_all();
    f();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return left_ == 0; });
    job_ = nullptr;
  }
 private:
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void()>* f;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = job_;
      }
      (*f)();
      { std::lock_guard<std::mutex> g(m_); --left_; }
      done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_; std::condition_variable cv_, done_;
  const std::function<void()>* job_ = nullptr; size_t left_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};

// ---- the whole file -------------------------------------------------------------------------------------------------
// One round's output: a plain array (no value-initialisation of tens of megabytes) handed to the consumer by move.
struct Bytes {
  uint8_t* p = nullptr; size_t n = 0;
  Bytes() {}
  Bytes(Bytes&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  Bytes& operator=(Bytes&& o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  Bytes(const Bytes&) = delete;
  Bytes& operator=(const Bytes&) = delete;
  ~Bytes() { free(p); }
};

// sink that hands every round's bytes to f(Bytes&&) -> bool
template <typename F>
struct PieceSink {
  F f; Bytes cur;
  explicit PieceSink(F fn) : f(fn) {}
  uint8_t* reserve(size_t n) { cur = Bytes(); cur.p = (uint8_t*)malloc(n ? n : 1); cur.n = n; return cur.p; }
  bool commit(size_t) { return f(std::move(cur)); }
};
template <typename F> static inline PieceSink<F> piece_sink(F f) { return PieceSink<F>(f); }

// Inflates a gzip file held in memory (all members), `threads` at a time.  The output of every round goes where the sink
// says: uint8_t* sink.reserve(size_t n) hands out room for the round's n bytes (nullptr: stop), the chunks of the round are
// written into it in parallel, bool sink.commit(size_t n) takes it over (false: stop).  Returns 0 on success, -1 on a damaged
// file (as zlib would report), 1 if the sink stopped.  chunk_bytes: compressed bytes per chunk.
template <typename Sink>
int inflate_all(const uint8_t* data, size_t n, unsigned threads, size_t chunk_bytes, Sink&& sink) {
  if (threads < 1) threads = 1;
  if (chunk_bytes < 4096) chunk_bytes = 4096;
  if (!gzip_header(data, n, 0)) return -1;
  const size_t round_chunks = (size_t)threads * 2;
  const size_t max_out = chunk_bytes * 1100 + (1u << 20);     // deflate expands at most 1032 : 1
  uint64_t cur_bit = 0;            // where the next round starts: a block start (the file's header for the first round)
  bool file_start = true;
  std::vector<uint8_t> window(32768, 0);
  uint32_t member_crc = (uint32_t)crc32(0L, Z_NULL, 0);
  uint64_t member_len = 0;
  // chunk slots live as long as the file: their symbol arrays keep their capacity from round to round
  std::vector<Chunk> ch(round_chunks), redo(round_chunks + 2);
  Pool pool(threads);
  const bool debug = getenv("PGZ_DEBUG") != nullptr;
  for (;;) {
    // chunk starts of this round (compressed byte offsets), then their block starts
    const size_t base = (size_t)(cur_bit >> 3);
    size_t n_ch = 1;
    ch[0].reset(); ch[0].start_bit = cur_bit; ch[0].found = true;
    for (size_t i = 1; i < round_chunks; ++i) {
      const size_t off = base + i * chunk_bytes;
      if (off + 64 >= n) break;
      ch[i].reset(); ch[i].start_bit = (uint64_t)off * 8;
      ++n_ch;
    }
    const uint64_t round_end = (uint64_t)std::min(n, base + n_ch * chunk_bytes) * 8;
    {
      std::atomic<size_t> next{1};
      pool.run([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n_ch) return;
          // (the search range comes from the nominal cuts: a neighbour's start_bit is being rewritten by its own finder)
          const uint64_t from = (uint64_t)(base + i * chunk_bytes) * 8;
          const uint64_t limit = i + 1 < n_ch ? (uint64_t)(base + (i + 1) * chunk_bytes) * 8 : round_end;
          uint64_t at = 0;
          if (find_block(data, n, from, limit, at)) { ch[i].start_bit = at; ch[i].found = true; }
        }
      });
    }
    {
      std::atomic<size_t> next{0};
      pool.run([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n_ch) return;
          Chunk& c = ch[i];
          if (!c.found) continue;
          uint64_t stop = round_end;
          for (size_t j = i + 1; j < n_ch; ++j) if (ch[j].found) { stop = ch[j].start_bit; break; }
          if (!c.room(0, chunk_bytes * 6)) { c.ok = false; continue; }
          c.ok = i == 0 ? decode_blocks(data, n, c.start_bit, stop, file_start ? nullptr : window.data(), file_start, c, (size_t)-1 >> 2)
                        : decode_blocks(data, n, c.start_bit, stop, nullptr, false, c, max_out);
        }
      });
    }
    if (!ch[0].ok) return -1;          // the round's first chunk is not speculative: its failure is the file's
    // phase 3: stitch in file order; a chunk is accepted iff the decoded prefix ends exactly where it starts
    std::vector<Chunk*> order;
    size_t n_redo = 0;
    order.push_back(&ch[0]);
    uint64_t pos = ch[0].end_bit;
    bool eof = ch[0].at_eof;
    auto bridge = [&](uint64_t stop) -> bool {       // sequential decode of what the speculation did not cover
      if (n_redo == redo.size()) return false;
      Chunk& r = redo[n_redo++];
      r.reset();
      r.start_bit = pos; r.found = true;
      r.ok = decode_blocks(data, n, pos, stop, nullptr, false, r, (size_t)-1 >> 2);
      if (!r.ok) return false;
      order.push_back(&r); pos = r.end_bit; eof = r.at_eof;
      return true;
    };
    for (size_t j = 1; j < n_ch && !eof; ) {
      Chunk& c = ch[j];
      if (!c.found || !c.ok || c.start_bit < pos) { ++j; continue; }           // unusable, or swallowed by an earlier chunk
      if (c.start_bit == pos) { order.push_back(&c); pos = c.end_bit; eof = c.at_eof; ++j; continue; }
      if (!bridge(c.start_bit)) return -1;                                      // a gap in front of it (then look at c again)
    }
    if (!eof && pos < round_end && !bridge(round_end)) return -1;
    if (debug) {
      size_t nf = 0, nok = 0;
      for (size_t i = 0; i < n_ch; ++i) { nf += ch[i].found; nok += ch[i].ok; }
      fprintf(stderr, "[pgz] round: %zu chunks, %zu found, %zu decoded, %zu accepted, %zu bridged\n", n_ch, nf, nok, order.size() - n_redo, n_redo);
    }
    // windows in order: the last 32 KiB of every accepted chunk, resolved (cheap, sequential); and where its bytes go
    std::vector<std::vector<uint8_t>> win(order.size());
    uint64_t total = 0;
    for (size_t i = 0; i < order.size(); ++i) {
      Chunk& c = *order[i];
      c.out_off = total; total += c.n_sym;
      win[i] = window;
      const size_t m = c.n_sym, take = std::min<size_t>(m, 32768);
      std::vector<uint8_t> nw(32768);
      if (take < 32768) memcpy(nw.data(), window.data() + take, 32768 - take);
      for (size_t q = 0; q < take; ++q) {
        const uint16_t s = c.sym[m - take + q];
        nw[32768 - take + q] = s < 256 ? (uint8_t)s : window[s & 0x7FFF];
      }
      window.swap(nw);
    }
    uint8_t* out_p = total ? sink.reserve((size_t)total) : nullptr;
    if (total && !out_p) return 1;
    // phase 4 in parallel: markers -> bytes into the round's output, CRC-32 of the pieces between member ends
    {
      std::atomic<size_t> next{0};
      pool.run([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= order.size()) return;
          Chunk& c = *order[i];
          const uint8_t* w = win[i].data();
          const size_t m = c.n_sym;
          const uint16_t* s = c.sym;
          uint8_t* o = out_p + c.out_off;
          for (size_t q0 = 0; q0 < m; q0 += 4096) {           // markers are rare behind a chunk's first 32 KiB: whole blocks narrow at once
            const size_t q1 = std::min(m, q0 + 4096);
            uint16_t any = 0;
            for (size_t q = q0; q < q1; ++q) any |= s[q];
            if (any < 256) for (size_t q = q0; q < q1; ++q) o[q] = (uint8_t)s[q];
            else for (size_t q = q0; q < q1; ++q) o[q] = s[q] < 256 ? (uint8_t)s[q] : w[s[q] & 0x7FFF];
          }
          uint64_t a = 0;
          for (size_t e = 0; e <= c.ends.size(); ++e) {
            const uint64_t z = e < c.ends.size() ? c.ends[e].out_off : m;
            const uint32_t crc = crc32_fast((uint32_t)crc32(0L, Z_NULL, 0), o + a, (size_t)(z - a));
            c.piece_crc.push_back(crc);
            c.piece_len.push_back(z - a);
            a = z;
          }
        }
      });
    }
    // member checks (CRC-32 and length against the trailers), then the bytes go out
    for (size_t i = 0; i < order.size(); ++i) {
      Chunk& c = *order[i];
      for (size_t e = 0; e < c.piece_crc.size(); ++e) {
        if (c.piece_len[e]) { member_crc = (uint32_t)crc32_combine(member_crc, c.piece_crc[e], (z_off_t)c.piece_len[e]); member_len += c.piece_len[e]; }
        if (e < c.ends.size()) {
          if (member_crc != c.ends[e].crc || (uint32_t)member_len != c.ends[e].isize) return -1;
          member_crc = (uint32_t)crc32(0L, Z_NULL, 0); member_len = 0;
        }
      }
    }
    if (total && !sink.commit((size_t)total)) return 1;
    if (eof) return 0;
    if (pos >= (uint64_t)n * 8) return -1;        // the data ended inside a member
    cur_bit = pos;
    file_start = false;
  }
}

}  // namespace pgz
#endif
