// radix_sort_probe.hip - hipcub::DeviceRadixSort::SortPairs on 64-bit keys / 64-bit values over a partial bit range, as the sorted
// build of the super-k-mer table calls it (mic_build.hip): are the pairs still pairs afterwards?  value = f(key), checked after the sort.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/radix_sort_probe tools/radix_sort_probe.hip && /tmp/radix_sort_probe
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef unsigned long long u64;
__host__ __device__ inline u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 32; return x; }
__global__ void gen(u64* k, u64* v, size_t n, u64 seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const u64 key = mix(i * 2654435761ull + seed); k[i] = key; v[i] = mix(key ^ 0x1234); }
}
__global__ void check(const u64* k, const u64* v, size_t n, int b0, unsigned long long* bad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (v[i] != mix(k[i] ^ 0x1234)) atomicAdd(&bad[0], 1ull);                               // the pair was torn
  if (i + 1 < n && (k[i] >> b0) > (k[i + 1] >> b0)) atomicAdd(&bad[1], 1ull);            // not sorted by the bits asked for
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t sizes[] = {3000, 82117, 1u << 20, 1u << 27, (size_t)1 << 30};
  const int ranges[][2] = {{48, 64}, {0, 64}, {32, 64}};
  for (size_t n : sizes) for (auto& r : ranges) for (int use_stream = 0; use_stream < 2; ++use_stream) {
    u64 *k0, *k1, *v0, *v1; unsigned long long* bad; void* tmp = nullptr; size_t tb = 0;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8)); CK(hipMalloc(&bad, 16));
    hipStream_t st = use_stream ? s : 0;
    gen<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(k0, v0, n, 77);
    CK(hipMemsetAsync(bad, 0, 16, st));
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const u64*)k0, k1, (const u64*)v0, v1, (int)n, r[0], r[1], st));
    CK(hipMalloc(&tmp, tb ? tb : 16));
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const u64*)k0, k1, (const u64*)v0, v1, (int)n, r[0], r[1], st));
    check<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(k1, v1, n, r[0], bad);
    unsigned long long h[2];
    CK(hipMemcpyAsync(h, bad, 16, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("n = %zu bits [%d, %d) %s: temp %zu bytes, torn pairs %llu, out of order %llu\n", n, r[0], r[1], use_stream ? "non-blocking stream" : "null stream", tb, h[0], h[1]);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(bad); hipFree(tmp);
  }
  return 0;
}
