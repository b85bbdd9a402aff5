// Random-gather microbenchmark for MI355X (gfx950).
//
// Purpose: calibrate the HBM random-access roofline that bounds the k-mer probe
// (SURVEY.md §8d "calibrate with a streaming + a 64 B-random-gather microbenchmark").
// It answers: how many independent random W-byte accesses per second does one
// MI355X sustain into a table of S bytes, for W in {4,8,16,32,64,128} and S from
// L2-resident to 100+ GB?  The answer decides the in-HBM DB layout.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gpurun_out/gather_bench
// Run:   ./gather_bench [max_table_GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

__global__ void fill_kernel(uint32_t* t, size_t n_words) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n_words; i += stride) t[i] = (uint32_t)mix64(i);
}

// Each lane does ITERS rounds of UNROLL independent random accesses of W bytes.
template <int W, int UNROLL>
__global__ void __launch_bounds__(256) gather_kernel(const uint8_t* __restrict__ table, uint64_t n_slots,
                                                     int iters, uint32_t* __restrict__ sink, uint64_t seed) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint64_t idx[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      uint64_t h = mix64(seed + tid * 0x9E3779B97F4A7C15ULL + (uint64_t)(it * UNROLL + u) * 0xD1B54A32D192ED03ULL);
      idx[u] = (uint64_t)(((unsigned __int128)h * n_slots) >> 64);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint8_t* p = table + idx[u] * (uint64_t)W;
      if constexpr (W == 4) { acc ^= *(const uint32_t*)p; }
      else if constexpr (W == 8) { uint2 v = *(const uint2*)p; acc ^= v.x ^ v.y; }
      else {
#pragma unroll
        for (int j = 0; j < W / 16; ++j) { uint4 v = ((const uint4*)p)[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;  // keep loads live
}

// 4 lanes cooperate on one 64-byte slot (each lane loads 16 B): one wave-instruction covers 16 slots.
template <int UNROLL>
__global__ void __launch_bounds__(256) gather_coop64_kernel(const uint8_t* __restrict__ table, uint64_t n_slots,
                                                            int iters, uint32_t* __restrict__ sink, uint64_t seed) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t grp = tid >> 2; uint32_t sub = tid & 3;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      uint64_t h = mix64(seed + grp * 0x9E3779B97F4A7C15ULL + (uint64_t)(it * UNROLL + u) * 0xD1B54A32D192ED03ULL);
      uint64_t idx = (uint64_t)(((unsigned __int128)h * n_slots) >> 64);
      uint4 v = ((const uint4*)(table + idx * 64))[sub];
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// Dependent two-level probe: 8-byte descriptor gather, then a 16-byte gather at an address derived from it
// (models bucket-descriptor -> keys in the on-disk CuCLARK layout).
template <int UNROLL>
__global__ void __launch_bounds__(256) gather_dep_kernel(const uint8_t* __restrict__ table, uint64_t n_desc,
                                                         uint64_t n_slots16, int iters, uint32_t* __restrict__ sink, uint64_t seed) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint8_t* second = table + n_desc * 8;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint2 d[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      uint64_t h = mix64(seed + tid * 0x9E3779B97F4A7C15ULL + (uint64_t)(it * UNROLL + u) * 0xD1B54A32D192ED03ULL);
      uint64_t idx = (uint64_t)(((unsigned __int128)h * n_desc) >> 64);
      d[u] = *(const uint2*)(table + idx * 8);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      uint64_t h2 = mix64(((uint64_t)d[u].y << 32) | d[u].x);
      uint64_t idx2 = (uint64_t)(((unsigned __int128)h2 * n_slots16) >> 64);
      uint4 v = *(const uint4*)(second + idx2 * 16);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void stream_kernel(const uint4* __restrict__ t, size_t n, uint32_t* sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += stride) { uint4 v = t[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}

int main(int argc, char** argv) {
  double max_gib = argc > 1 ? atof(argv[1]) : 96.0;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
  printf("# device %s CUs=%d clock=%d MHz  mem free=%.1f GiB total=%.1f GiB\n", prop.name, prop.multiProcessorCount,
         prop.clockRate / 1000, free_b / 1073741824.0, total_b / 1073741824.0);
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  const uint64_t lanes = (uint64_t)blocks * threads;

  std::vector<double> sizes_gib = {0.0039, 0.0625, 0.5, 4.0, 40.0, max_gib};
  printf("table_GiB,kind,W,unroll,Gaccess_per_s,useful_GBps\n");
  for (double gib : sizes_gib) {
    if (gib > max_gib) continue;
    size_t bytes = (size_t)(gib * 1073741824.0) & ~(size_t)4095;
    uint8_t* table;
    if (hipMalloc(&table, bytes) != hipSuccess) { printf("# hipMalloc(%.1f GiB) failed\n", gib); continue; }
    fill_kernel<<<blocks, threads>>>((uint32_t*)table, bytes / 4);
    CK(hipDeviceSynchronize());
    int iters = 16;
    auto report = [&](const char* kind, int W, int unroll, double ms, double acc_per_lane) {
      double acc = (double)lanes * acc_per_lane;
      printf("%.4f,%s,%d,%d,%.2f,%.1f\n", gib, kind, W, unroll, acc / ms / 1e6, acc * W / ms / 1e6);
      fflush(stdout);
    };
    {
      double ms = time_ms([&] { stream_kernel<<<blocks, threads>>>((const uint4*)table, bytes / 16, sink); }, 3);
      printf("%.4f,stream,16,1,%.2f,%.1f\n", gib, bytes / 16.0 / ms / 1e6, bytes / ms / 1e6);
    }
#define RUN(W, U) { uint64_t n = bytes / W; \
      double ms = time_ms([&] { gather_kernel<W, U><<<blocks, threads>>>(table, n, iters, sink, 1234); }, 3); \
      report("gather", W, U, ms, (double)iters * U); }
    RUN(4, 4) RUN(8, 4) RUN(16, 4) RUN(32, 4) RUN(64, 1) RUN(64, 2) RUN(64, 4) RUN(128, 2)
    RUN(4, 8) RUN(16, 8)
#undef RUN
    {
      uint64_t n = bytes / 64;
      double ms = time_ms([&] { gather_coop64_kernel<4><<<blocks, threads>>>(table, n, iters, sink, 99); }, 3);
      double acc = (double)lanes / 4 * iters * 4;
      printf("%.4f,coop64,64,4,%.2f,%.1f\n", gib, acc / ms / 1e6, acc * 64 / ms / 1e6);
      ms = time_ms([&] { gather_coop64_kernel<8><<<blocks, threads>>>(table, n, iters, sink, 99); }, 3);
      acc = (double)lanes / 4 * iters * 8;
      printf("%.4f,coop64,64,8,%.2f,%.1f\n", gib, acc / ms / 1e6, acc * 64 / ms / 1e6);
    }
    {
      uint64_t n_desc = bytes / 5 / 8, n16 = (bytes - n_desc * 8) / 16;
      double ms = time_ms([&] { gather_dep_kernel<4><<<blocks, threads>>>(table, n_desc, n16, iters, sink, 7); }, 3);
      double acc = (double)lanes * iters * 4;
      printf("%.4f,dep8+16,24,4,%.2f,%.1f\n", gib, acc / ms / 1e6, acc * 24 / ms / 1e6);
    }
    CK(hipFree(table));
  }
  CK(hipFree(sink));
  return 0;
}
