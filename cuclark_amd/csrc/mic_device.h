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
  h ^= h >> 15; h *= 0x2C1B3C6Du;      // one multiply-xorshift round: the order only has to look random
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

// M-slot words (uint4 q[8]): q[0..5] = 12 keys (u64, ascending, unused = ~0); q[6], q[7].xy = 12 labels (u16) in a
// LEAF, or q[6].x = index of the first child slot in a DIRECTORY; q[7].z = meta (bits 0..7 = entries / children).
#define MIC_M_N(meta) ((meta) & 0xFFu)
#define MIC_M_DIR 0x100u      /* keys are separators: the smallest key below each (contiguous) child slot */

#endif
