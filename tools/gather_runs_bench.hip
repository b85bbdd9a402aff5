// Microbenchmark for locality-preserving tables: RUN consecutive probes share one slot of SLOT bytes
// (as consecutive k-mers share a minimizer).  SLOT/16 lanes cooperate on a probe; probes are laid out over
// the wave exactly like the query kernel's sub-passes.  Reports probes/s and distinct-slot requests/s.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_runs_bench.hip -o build/tools/gather_runs_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void fill_kernel(uint32_t* t, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) t[i] = (uint32_t)mix64(i); }

// each wave handles "reads" of 128 probes: 128*G lanes-worth of loads = 128*G/64 instructions
template <int SLOT, int RUN, int NT>
__global__ void __launch_bounds__(256) runs_kernel(const uint8_t* __restrict__ table, uint64_t n_slots, int reads_per_wave,
                                                   uint32_t* __restrict__ sink, uint64_t seed) {
  constexpr int G = SLOT / 16;            // lanes per probe
  constexpr int PPI = 64 / G;             // probes per wave-instruction
  constexpr int NI = 128 / PPI;           // instructions per read
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  for (int r = 0; r < reads_per_wave; ++r) {
    const uint64_t read_id = wave * reads_per_wave + r;
    const uint32_t phase = (uint32_t)mix64(read_id ^ seed) % RUN;   // runs are not aligned to the read start
    uint4 v[NI > 16 ? 16 : NI];
#pragma unroll
    for (int i = 0; i < (NI > 16 ? 16 : NI); ++i) {
      const int probe = i * PPI + lane / G;                          // consecutive probes in consecutive groups
      const uint64_t run_id = read_id * 256 + (probe + phase) / RUN;
      const uint64_t slot = (uint64_t)(((unsigned __int128)mix64(run_id * 0x9E3779B97F4A7C15ULL + seed) * n_slots) >> 64);
      const uint4* p = (const uint4*)(table + slot * SLOT) + (lane % G);
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      if (NT) { u4 t = __builtin_nontemporal_load((const u4*)p); v[i] = make_uint4(t.x, t.y, t.z, t.w); } else v[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < (NI > 16 ? 16 : NI); ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int SLOT, int RUN, int NT>
static void run(const uint8_t* table, size_t bytes, uint32_t* sink, int blocks) {
  constexpr int G = SLOT / 16, PPI = 64 / G, NI = 128 / PPI, NIc = NI > 16 ? 16 : NI;
  const int rpw = 8;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  uint64_t n_slots = bytes / SLOT;
  runs_kernel<SLOT, RUN, NT><<<blocks, 256>>>(table, n_slots, rpw, sink, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 3; ++i) runs_kernel<SLOT, RUN, NT><<<blocks, 256>>>(table, n_slots, rpw, sink, 2 + i);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  double waves = (double)blocks * 4, probes = waves * rpw * (NIc * PPI);
  double slots = probes / RUN;
  printf("%d,%d,%d,%.2f,%.2f,%.1f\n", SLOT, RUN, NT, probes / ms / 1e6, slots / ms / 1e6, slots * SLOT / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 100.0;
  size_t bytes = (size_t)(gib * 1073741824.0) & ~(size_t)4095;
  uint8_t* table; CK(hipMalloc(&table, bytes));
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  fill_kernel<<<2048, 256>>>((uint32_t*)table, bytes / 4); CK(hipDeviceSynchronize());
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int blocks = prop.multiProcessorCount * 64;
  printf("# table %.1f GiB; columns: slot_bytes,run,nontemporal,Gprobes_per_s,Gslots_per_s,slot_GBps\n", gib);
  run<64, 1, 0>(table, bytes, sink, blocks);   run<64, 1, 1>(table, bytes, sink, blocks);
  run<64, 2, 0>(table, bytes, sink, blocks);   run<64, 4, 0>(table, bytes, sink, blocks);
  run<64, 8, 0>(table, bytes, sink, blocks);   run<64, 16, 0>(table, bytes, sink, blocks);
  run<128, 1, 0>(table, bytes, sink, blocks);  run<128, 1, 1>(table, bytes, sink, blocks);
  run<128, 4, 0>(table, bytes, sink, blocks);  run<128, 8, 0>(table, bytes, sink, blocks);
  run<128, 16, 0>(table, bytes, sink, blocks);
  run<256, 1, 0>(table, bytes, sink, blocks);  run<256, 4, 0>(table, bytes, sink, blocks);
  run<256, 8, 0>(table, bytes, sink, blocks);  run<256, 16, 0>(table, bytes, sink, blocks);
  run<512, 1, 0>(table, bytes, sink, blocks);  run<512, 8, 0>(table, bytes, sink, blocks);
  return 0;
}
