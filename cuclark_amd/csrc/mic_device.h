// mic_device.h — device-side helpers shared by the kernels (mic_kernels.hip) and the table builders (mic_build.hip).
#ifndef MIC_DEVICE_H
#define MIC_DEVICE_H

#include "mic_internal.h"

__device__ __forceinline__ uint32_t bperm(int src_lane, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define QP_BCAST0 0x00
#define QP_BCAST1 0x55
#define QP_BCAST3 0xFF
#define QP_XOR1 0xB1
#define QP_XOR2 0x4E

// reverse complement of a k-mer value: reverse all 64 bits, swap the two bits of every pair back, complement
// (written on the 32-bit halves so that each pair swap is shift, shift, v_bfi: the query kernel is VALU-issue bound)
__device__ __forceinline__ uint64_t revcomp_bits(uint64_t x, int k) {
  uint32_t rh = __builtin_bitreverse32((uint32_t)x), rl = __builtin_bitreverse32((uint32_t)(x >> 32));
  rh = ((rh >> 1) & 0x55555555u) | ((rh << 1) & ~0x55555555u);
  rl = ((rl >> 1) & 0x55555555u) | ((rl << 1) & ~0x55555555u);
  return (((uint64_t)~rh << 32) | (uint32_t)~rl) >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t canonical(uint64_t kmer, int k) {
  uint64_t rc = revcomp_bits(kmer, k);
  return kmer < rc ? kmer : rc;
}

// ---- minimizer-keyed table (M-table) ---------------------------------------------------------------------------
// order key of an m-mer (m <= 31): 32-bit mix of its canonical value.  The slot of a k-mer is a function
// of the MINIMUM order key over its w = k-m+1 m-mers only, so a k-mer and its reverse complement (same canonical
// m-mers) agree, and ties between different m-mers are harmless.
#ifndef MIC_HASH_LITE
#define MIC_HASH_LITE 2   /* bit 0: order key without its xorshift (+0.4 %, kept off: real genomes are not random); bit 1: slot = plain multiplicative hash (+0.6 %) */
#endif
#ifndef MIC_CHEAP_HASH
#define MIC_CHEAP_HASH 1
#endif
__device__ __forceinline__ uint32_t mmer_order_key_canon(uint64_t u) {   // u = canonical m-mer value
#if MIC_CHEAP_HASH
  // 32-bit multiplies issue at half the rate of plain integer ops (tools/valu_rate_bench.hip): the high word (8 bits
  // for m = 20) goes through one full-rate 24-bit multiply-add; bits 56+ of a 29..31-mer only reach the key through
  // the fold.  The order only has to look random and be the same function in the table build and the query.
  const uint32_t hi = (uint32_t)(u >> 32);
  uint32_t h = (uint32_t)u * 0x9E3779B1u ^ (__umul24(hi ^ (hi >> 24), 0xEBCA77u) + 0x27D4EB2Fu);
#else
  uint32_t h = (uint32_t)u * 0x9E3779B1u ^ ((uint32_t)(u >> 32) * 0x85EBCA77u + 0x27D4EB2Fu);
#endif
#if MIC_HASH_LITE & 1
  h *= 0x2C1B3C6Du;                    // the second multiply alone carries the low bits upwards
#else
  h ^= h >> 15; h *= 0x2C1B3C6Du;      // one multiply-xorshift round: the order only has to look random
#endif
  return h;
}

__device__ __forceinline__ uint32_t mmer_order_key(uint64_t x, int m) { return mmer_order_key_canon(canonical(x, m)); }

__device__ __forceinline__ uint32_t mslot_of_key(uint32_t min_key, uint32_t n_slots) {
#if MIC_CHEAP_HASH
  // the minimum of w hashes is small but its low bits are uniform: one odd multiply carries them into the high bits
  return __umulhi(min_key * 0xC2B2AE3Du, n_slots);
#else
  // the minimum of w hashes is small: remix before taking the high bits
  uint32_t z = min_key * 0xC2B2AE3Du;
  z ^= z >> 15; z *= 0x165667B1u;
  return __umulhi(z, n_slots);
#endif
}

// sequential form (table build, dense fallback, statistics)
__device__ __forceinline__ uint32_t mslot_of_kmer(uint64_t kmer, int k, int m, uint32_t n_slots) {
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  uint32_t best = 0xFFFFFFFFu;
  for (int j = 0; j + m <= k; ++j) {
    uint32_t h = mmer_order_key((kmer >> (2 * (k - m - j))) & mask, m);
    best = h < best ? h : best;
  }
  return mslot_of_key(best, n_slots);
}

// ---- super-k-mer table ("S-table", layout 3) --------------------------------------------------------------------
// The k-mers of the database that share a minimizer OCCURRENCE overlap by k-1 nucleotides: they are stored as one
// entry, the super-k-mer S of k+w-1 nucleotides with the minimizer at the fixed position w-1, plus a w-bit presence
// mask (bit j: the k-mer whose minimizer sits at its position j, i.e. S[w-1-j, w-1-j+k), is in the database) and one
// label.  ~6.5 k-mers per 16-byte entry instead of 10 bytes each, so a 128-byte slot (6 entries) practically never
// overflows and a lookup is ONE slot.  Everything is exact: the k-mer is compared in full against the entry.
//   orientation: the k-mer is taken in the strand where the minimizer reads as its canonical m-mer value x;
//   order of m-mers: the top 27 bits of mmer_order_key_canon(x); the low 5 bits of the query's sliding-minimum key carry
//   the strand and the position, so positions that tie on the 27 bits are resolved arbitrarily by the query - the
//   build stores a k-mer under EVERY tied position (and under both strands of a palindromic minimizer);
//   slot = hash of the full x (not of the 32-bit order key: distinct minimizers no longer collide in key space).
// 128-byte slot, 32 words: [0..5] sort keys = low 32 bits of x, ascending, unused = ~0 | [6..23] S as 3 words per
// entry, nucleotide 0 in the top bits of word 0 | [24..29] presence mask << 16 | label | [30] entries | HAS_NEXT << 8
// | [31] index of the continuation slot (same format) when more than 6 entries hash here.
#define MIC_S_CAP 6
#define MIC_S_NEXT 0x100u

// Which m-mer of a k-mer is "its minimizer" (round 4): MOD-SAMPLING (Groot Koerkamp & Pibiri, "The mod-minimizer", WABI 2024).
// Take t = m - j w with the largest j that keeps t >= 7 (t = m when there is none: the plain random minimizer).  Among the
// W = k - t + 1 = (j + 1) w t-mers of the k-mer find the one with the smallest order, at position i; the sampled m-mer is the one
// at position p = i mod w.  Consecutive k-mers of a read keep the same sampled m-mer for longer than under the plain minimizer:
// density 0.120 instead of 0.154 for k = 31, m = 20 (t = 8, W = 24), i.e. ~15 instead of ~19.5 runs per 150-bp read, 21 % fewer
// entries in the table, and the query computes orders of 16-bit values (one 24-bit multiply-add) instead of 40-bit ones.
// p is a function of the k-mer alone, so everything else - entries, presence masks, slots, exactness - is as before.
//   one-strand table: the order is taken of the CANONICAL t-mer (min of the t-mer and its reverse complement); because
//   t = m (mod w) the positions of rc(K) mirror those of K (p -> w-1-p), and the strand is that of the sampled m-mer;
//   ties (the same t-mer twice in the window, or 27-bit collisions) are resolved by the position bits of the query's key;
//   the build stores a k-mer under EVERY tied position.
#ifndef MIC_S_MOD
#define MIC_S_MOD 1      /* 0: t = m, the plain minimizer of rounds 1-3 through the same code (ablation) */
#endif
__host__ __device__ __forceinline__ int s_tlen(int k, int m) {
  int t = m;
#if MIC_S_MOD
  const int w = k - m + 1;
  while (t - w >= 7) t -= w;
#endif
  return t;
}
// order of a t-mer value (the top 27 bits count): t <= 12 is one full-rate 24-bit multiply-add
__device__ __forceinline__ uint32_t s_torder(uint64_t tv) {
  uint32_t h = __umul24((uint32_t)tv & 0xFFFFFFu, 0x9E3779u) + 0x27D4EB2Fu;
  const uint32_t hi = (uint32_t)(tv >> 24);
  if (hi) h += hi * 0x85EBCA77u + (uint32_t)(tv >> 56) * 0xC2B2AE3Du;
  return h;
}
__device__ __forceinline__ uint32_t s_torder24(uint32_t tv) { return __umul24(tv, 0x9E3779u) + 0x27D4EB2Fu; }   // tv < 2^24
// reverse complement of a t-mer of at most 16 nucleotides
__device__ __forceinline__ uint32_t revcomp_bits32(uint32_t x, int t) {
  uint32_t r = __builtin_bitreverse32(x) >> (32 - 2 * t);
  r = ((r >> 1) & 0x55555555u) | ((r << 1) & ~0x55555555u);
  return ~r & (t >= 16 ? 0xFFFFFFFFu : (1u << (2 * t)) - 1u);
}
// sequential form: f(p) for every position p (0 .. w-1) the query may sample for the k-mer K as it reads (every tie)
template <typename F>
__device__ __forceinline__ void s_sampled(uint64_t K, int k, int m, bool canon, F&& f) {
  const int w = k - m + 1, t = s_tlen(k, m), W = k - t + 1;
  const uint64_t tmask = t >= 32 ? ~0ULL : (1ULL << (2 * t)) - 1;
  // one pass: the smallest order so far and the set of positions p = i mod w that reach it (w <= 16: a bit each)
  uint32_t hmin = 0xFFFFFFFFu, set = 0;
  int p = 0;
  for (int i = 0; i < W; ++i) {
    uint32_t h;
    if (t <= 12) {                       // 24-bit values: 32-bit arithmetic throughout (the build evaluates 2 x 24 of these per k-mer)
      uint32_t tv = (uint32_t)(K >> (2 * (k - t - i))) & (uint32_t)tmask;
      if (canon) { const uint32_t tr = revcomp_bits32(tv, t); tv = tr < tv ? tr : tv; }
      h = s_torder24(tv) >> 5;
    } else {
      uint64_t tv = (K >> (2 * (k - t - i))) & tmask;
      if (canon) { const uint64_t tr = revcomp_bits(tv, t); tv = tr < tv ? tr : tv; }
      h = s_torder(tv) >> 5;
    }
    if (h < hmin) { hmin = h; set = 0; }
    if (h == hmin) set |= 1u << p;
    if (++p == w) p = 0;
  }
  while (set) {
    const int q = __builtin_ctz(set);
    set &= set - 1;
    f(q);
  }
}

// slot of a minimizer value x: a 32-bit hash of x, scaled to the table - slot = floor(h * n_slots / 2^32), monotone in h (the sorted
// build of mic_build.hip orders its records by h before the number of slots is known)
__device__ __forceinline__ uint32_t sslot_hash(uint64_t x) {
  const uint32_t hi = (uint32_t)(x >> 32);
  uint32_t h = (uint32_t)x * 0x85EBCA77u + __umul24(hi ^ (hi >> 24), 0xC2B2AFu);
#if !(MIC_HASH_LITE & 2)
  h ^= h >> 15; h *= 0x165667B1u;
#endif
  return h;
}
__device__ __forceinline__ uint32_t sslot_of_x(uint64_t x, uint32_t n_slots) { return __umulhi(sslot_hash(x), n_slots); }

// the same function of x given as (low word, bits above it): 32-bit operations only (query_kernel_r)
__device__ __forceinline__ uint32_t sslot_of_x32(uint32_t lo, uint32_t hi, uint32_t n_slots) {
  uint32_t h = lo * 0x85EBCA77u + __umul24(hi ^ (hi >> 24), 0xC2B2AFu);
#if !(MIC_HASH_LITE & 2)
  h ^= h >> 15; h *= 0x165667B1u;
#endif
  return __umulhi(h, n_slots);
}

// k-mer at nucleotide offset a (0..15) of a super-k-mer stored as three words
__device__ __forceinline__ uint64_t s_extract(uint32_t w0, uint32_t w1, uint32_t w2, int a, int k) {
  const int s = 2 * a;
  const uint64_t hi = ((uint64_t)w0 << 32) | w1;
  const uint64_t x = (hi << s) | ((uint64_t)w2 >> (32 - s));     // s = 0: the 64-bit shift by 32 yields 0
  return x >> (64 - 2 * k);
}

// every (oriented k-mer, minimizer position, minimizer value) under which the query may look canonical k-mer c up
template <typename F>
__device__ __forceinline__ void s_candidates(uint64_t c, int k, int m, F&& f) {
  const int w = k - m + 1;
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  const uint64_t rc = revcomp_bits(c, k);
  uint32_t P = 0;
  s_sampled(c, k, m, true, [&](int p) { P |= 1u << p; });
  for (uint32_t left = P; left; left &= left - 1) {
    const int p = __builtin_ctz(left);
    const uint64_t mf = (c >> (2 * (k - m - p))) & mask, mr = revcomp_bits(mf, m);
    if (mf <= mr) f(c, p, mf);
    // A k-mer that is its own reverse complement (even k) has mirrored positions, and (rc, w-1-p) IS (c, q) with q = w-1-p: when q
    // is sampled too, q's own first form stores it.  Stored twice, the k-mer would sit in two entries of its minimizer and the
    // per-run popcount would count it in both (found by the fuzzer's (GC)n microsatellites at k = 32, round 6).
    if (mf >= mr && !(rc == c && ((P >> (w - 1 - p)) & 1u))) f(rc, w - 1 - p, mr);
  }
}

// Two-strand form of the same table ("S2", MIC_LAYOUT_SUPER2): BOTH orientations of every database k-mer are stored,
// each under the sampled m-mer of its own t-mers AS THEY READ (no canonical form).  A query k-mer is then looked up exactly as
// it stands in the read: no reverse complement and no strand bookkeeping in the query kernel, at the price of twice the
// entries.  A k-mer that is its own reverse complement is stored once.
template <typename F>
__device__ __forceinline__ void s_candidates_fwd(uint64_t c, int k, int m, F&& f) {
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  const uint64_t rc = revcomp_bits(c, k);
  for (int strand = 0; strand < 2; ++strand) {
    if (strand && rc == c) break;
    const uint64_t K = strand ? rc : c;
    s_sampled(K, k, m, false, [&](int p) { f(K, p, (K >> (2 * (k - m - p))) & mask); });
  }
}

// Side table of the crowded minimizers' k-mers (mic_build.hip: s_crowd_move_kernel): cells {k-mer lo, hi, label + 1, 0}, linear probing
// The cell of a k-mer: the TOP bits of a two-round multiply-xorshift of K.  (Until round 5: bits 32 .. of K * phi under the mask -
// the middle of the product, which the top nucleotides of K barely reach: the k-mers of one microsatellite with different left
// flanks - exactly what a crowded minimizer collects - fell into a handful of cells and were walked one after the other.)
__device__ __forceinline__ uint32_t s_side_hash(uint64_t K, uint32_t mask) {
  uint64_t x = K * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  return (uint32_t)(x >> 32) >> __builtin_clz(mask | 1u);      // mask = 2^b - 1: the top b bits
}
__device__ inline uint32_t s_side_probe(const uint4* __restrict__ side, uint32_t mask, uint64_t K) {
  uint32_t h = s_side_hash(K, mask);
  for (;;) {
    const uint4 c = side[h];
    if (c.z == 0) return 0;
    if (c.x == (uint32_t)K && c.y == (uint32_t)(K >> 32)) return c.z;
    h = (h + 1) & mask;
  }
}

// The lookup the QUERY KERNELS perform for the k-mer K that reads at nucleotide `tpos` of its read part, done sequentially
// (dense fallback, statistics).  The kernels' sliding minimum runs over keys order27 << 5 | position & 31 of the t-mers, so
// t-mers that tie on the 27 bits are resolved by the position (mod 32; the window holds at most 26 t-mers and chunks start at
// multiples of 128, so the key is a function of tpos + i).  The table stores a k-mer under EVERY tied position, so any choice
// finds it; but in a table-sharded run the tied positions may hash to slots of DIFFERENT parts, and the parts' counts only add
// up if every path - per-run kernel, per-k-mer kernel, this one - makes the same choice.  One-strand table: the k-mer is looked
// up in the strand in which the sampled m-mer is the smaller of itself and its reverse complement (forward when they are equal).
//   parted: only the main slots [slot_lo, slot_lo + slot_cnt) are resident (slots = allocation - slot_lo slots); *mine tells
//   whether the chosen slot is.
__device__ inline uint32_t s_probe_read(const uint4* __restrict__ slots, uint32_t n_slots, bool parted, uint32_t slot_lo,
                                        uint32_t slot_cnt, uint64_t K, uint32_t tpos, int k, int m, bool fwd, bool* mine,
                                        const uint4* __restrict__ side = nullptr, uint32_t side_mask = 0) {
  const int w = k - m + 1, t = s_tlen(k, m), W = k - t + 1;
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  const uint64_t tmask = t >= 32 ? ~0ULL : (1ULL << (2 * t)) - 1;
  uint32_t best = 0xFFFFFFFFu; int bi = 0;
  for (int i = 0; i < W; ++i) {
    uint64_t tv = (K >> (2 * (k - t - i))) & tmask;
    if (!fwd) { const uint64_t tr = revcomp_bits(tv, t); tv = tr < tv ? tr : tv; }
    const uint32_t key = (s_torder(tv) & ~31u) | ((tpos + (uint32_t)i) & 31u);
    if (key < best) { best = key; bi = i; }
  }
  const int p = bi % w;
  const uint64_t mf = (K >> (2 * (k - m - p))) & mask;
  const uint64_t mr = fwd ? 0 : revcomp_bits(mf, m);
  const bool rev = !fwd && mr < mf;
  const uint64_t x = rev ? mr : mf;
  const uint64_t Kq = rev ? revcomp_bits(K, k) : K;
  const int j = rev ? w - 1 - p : p;
  uint64_t slot = sslot_of_x(x, n_slots);
  if (parted && (uint32_t)slot - slot_lo >= slot_cnt) { *mine = false; return 0; }
  *mine = true;
  for (;;) {
    const uint32_t* q = (const uint32_t*)(slots + slot * 8);
    for (int e = 0; e < MIC_S_CAP; ++e) {
      if (q[e] != (uint32_t)x) continue;
      const uint32_t pl = q[24 + e];
      // a marker (presence mask 0) with this very minimizer: its k-mers live in the side table
      if ((pl >> 16) == 0 && side && s_extract(q[6 + 3 * e], q[7 + 3 * e], q[8 + 3 * e], w - 1, m) == x) return s_side_probe(side, side_mask, Kq);
      if (!((pl >> (16 + j)) & 1)) continue;
      if (s_extract(q[6 + 3 * e], q[7 + 3 * e], q[8 + 3 * e], w - 1 - j, k) == Kq) return (pl & 0xFFFFu) + 1;
    }
    if (!(q[30] & MIC_S_NEXT)) return 0;
    slot = q[31];
  }
}

// M-slot words (uint4 q[8]): q[0..5] = 12 keys (u64, ascending, unused = ~0); q[6], q[7].xy = 12 labels (u16) in a
// LEAF, or q[6].x = index of the first child slot in a DIRECTORY; q[7].z = meta (bits 0..7 = entries / children).
#define MIC_M_N(meta) ((meta) & 0xFFu)
#define MIC_M_DIR 0x100u      /* keys are separators: the smallest key below each (contiguous) child slot */

#endif
