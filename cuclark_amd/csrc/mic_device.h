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
__device__ __forceinline__ uint64_t revcomp_bits(uint64_t x, int k) {
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
  return (~r) >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t canonical(uint64_t kmer, int k) {
  uint64_t rc = revcomp_bits(kmer, k);
  return kmer < rc ? kmer : rc;
}

// ---- minimizer-keyed table (M-table) ---------------------------------------------------------------------------
// order key of an m-mer: 32-bit mix of its canonical value.  The slot of a k-mer is a function of the MINIMUM order
// key over its w = k-m+1 m-mers only, so k-mer and reverse complement (same canonical m-mers) agree, and ties between
// different m-mers are harmless.
__device__ __forceinline__ uint32_t mmer_order_key(uint64_t mmer, int m) {
  uint64_t u = canonical(mmer, m);
  u *= 0x9E3779B97F4A7C15ULL; u ^= u >> 32; u *= 0xD6E8FEB86659FD93ULL; u ^= u >> 32;
  return (uint32_t)u;
}

__device__ __forceinline__ uint64_t mslot_of_key(uint32_t min_key, uint64_t n_slots) {
  uint64_t z = (uint64_t)min_key * 0xD1B54A32D192ED03ULL + 0x9E3779B97F4A7C15ULL;
  z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 32;
  return __umul64hi(z, n_slots);
}

// sequential form (table build, dense fallback, statistics)
__device__ __forceinline__ uint64_t mslot_of_kmer(uint64_t kmer, int k, int m, uint64_t n_slots) {
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  uint32_t best = 0xFFFFFFFFu;
  for (int j = 0; j + m <= k; ++j) {
    uint32_t h = mmer_order_key((kmer >> (2 * (k - m - j))) & mask, m);
    best = h < best ? h : best;
  }
  return mslot_of_key(best, n_slots);
}

#endif
