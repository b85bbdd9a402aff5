// radix_sort_probe2.hip - which forms of hipcub::DeviceRadixSort::SortPairs keep pairs together and sort, at which sizes (ROCm 7.2)?
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef unsigned long long u64;
__host__ __device__ inline u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 32; return x; }
__global__ void gen64(u64* k, u64* v, size_t n) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { const u64 key = mix(i * 2654435761ull + 77); k[i] = key; v[i] = mix(key ^ 0x1234); } }
__global__ void chk64(const u64* k, const u64* v, size_t n, int b0, int b1, unsigned long long* bad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 m = b1 - b0 >= 64 ? ~0ull : ((1ull << (b1 - b0)) - 1);
  if (v[i] != mix(k[i] ^ 0x1234)) atomicAdd(&bad[0], 1ull);
  if (i + 1 < n && ((k[i] >> b0) & m) > ((k[i + 1] >> b0) & m)) atomicAdd(&bad[1], 1ull);
}
__global__ void gen16(unsigned short* k, uint4* v, size_t n) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { const u64 x = mix(i * 2654435761ull + 77); k[i] = (unsigned short)(x >> 48); v[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)mix(x), (uint32_t)(mix(x) >> 32)); } }
__global__ void chk16(const unsigned short* k, const uint4* v, size_t n, unsigned long long* bad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 x = ((u64)v[i].y << 32) | v[i].x;
  if (k[i] != (unsigned short)(x >> 48) || v[i].z != (uint32_t)mix(x) || v[i].w != (uint32_t)(mix(x) >> 32)) atomicAdd(&bad[0], 1ull);
  if (i + 1 < n && k[i] > k[i + 1]) atomicAdd(&bad[1], 1ull);
}
int main() {
  const size_t sizes[] = {3000, 82117, 1u << 20, 1u << 24, 1u << 27};
  const int ranges[][2] = {{0, 16}, {0, 32}, {16, 32}, {48, 64}};
  for (size_t n : sizes) {
    for (auto& r : ranges) {
      u64 *k0, *k1, *v0, *v1; unsigned long long* bad; void* tmp = nullptr; size_t tb = 0;
      CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8)); CK(hipMalloc(&bad, 16));
      gen64<<<(unsigned)((n + 255) / 256), 256>>>(k0, v0, n);
      CK(hipMemset(bad, 0, 16));
      CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const u64*)k0, k1, (const u64*)v0, v1, (int)n, r[0], r[1]));
      CK(hipMalloc(&tmp, tb ? tb : 16));
      CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const u64*)k0, k1, (const u64*)v0, v1, (int)n, r[0], r[1]));
      chk64<<<(unsigned)((n + 255) / 256), 256>>>(k1, v1, n, r[0], r[1], bad);
      unsigned long long h[2]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
      printf("u64 / u64  n = %zu bits [%d, %d): torn pairs %llu, out of order %llu\n", n, r[0], r[1], h[0], h[1]);
      hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(bad); hipFree(tmp);
    }
    {
      unsigned short *k0, *k1; uint4 *v0, *v1; unsigned long long* bad; void* tmp = nullptr; size_t tb = 0;
      CK(hipMalloc(&k0, n * 2)); CK(hipMalloc(&k1, n * 2)); CK(hipMalloc(&v0, n * 16)); CK(hipMalloc(&v1, n * 16)); CK(hipMalloc(&bad, 16));
      gen16<<<(unsigned)((n + 255) / 256), 256>>>(k0, v0, n);
      CK(hipMemset(bad, 0, 16));
      CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const unsigned short*)k0, k1, (const uint4*)v0, v1, (int)n));
      CK(hipMalloc(&tmp, tb ? tb : 16));
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
      CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const unsigned short*)k0, k1, (const uint4*)v0, v1, (int)n));
      hipEventRecord(b); hipEventSynchronize(b); float ms = 0; hipEventElapsedTime(&ms, a, b);
      chk16<<<(unsigned)((n + 255) / 256), 256>>>(k1, v1, n, bad);
      unsigned long long h[2]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
      printf("u16 / uint4 n = %zu whole key: torn pairs %llu, out of order %llu, %.3f ms, temp %zu\n", n, h[0], h[1], ms, tb);
      hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(bad); hipFree(tmp);
    }
  }
  return 0;
}
