// memcpy2d_probe.hip - does the runtime write exactly what a small strided device-to-host copy asks for?
// The super-k-mer table build samples its offsets array with ONE hipMemcpy2D of 8-byte rows at a pitch of 8 * G bytes into a
// std::vector of n_slots / G + 1 words (mic_build.hip: "slot ranges whose candidates fit the staging area"); for the small tables of
// the fuzzer that vector is a few dozen bytes of pageable heap.  This probe repeats that call shape (and the small asynchronous
// copies into pageable memory the builders make) thousands of times into malloc'ed buffers with canary words on both sides.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/memcpy2d_probe tools/memcpy2d_probe.hip && /tmp/memcpy2d_probe [iterations]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill(unsigned long long* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0x1000000000000000ull + i;
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  const size_t N = (size_t)1 << 24;      // 128 MB of offsets
  unsigned long long* d = nullptr;
  CK(hipMalloc(&d, N * 8));
  fill<<<(unsigned)((N + 255) / 256), 256>>>(d, N);
  CK(hipDeviceSynchronize());
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const unsigned long long CAN = 0xC0FFEE11DEADBEEFull;
  uint64_t rng = 88172645463325252ull;
  auto rnd = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  long bad_canary = 0, bad_value = 0;
  for (long it = 0; it < iters; ++it) {
    // ---- the builder's shape: n rows of 8 bytes, source pitch 8 * G
    uint64_t G = (uint64_t)64 << (rnd() % 11);                 // 64 .. 65536
    const uint64_t max_rows = N / G;
    uint64_t n = 1 + rnd() % (it % 4 == 0 ? 2000 : 40);
    if (n > max_rows) n = max_rows;
    const size_t pad = 2 + rnd() % 3;
    unsigned long long* buf = (unsigned long long*)malloc((n + 2 * pad) * 8);
    for (size_t i = 0; i < n + 2 * pad; ++i) buf[i] = CAN;
    CK(hipMemcpy2D(buf + pad, 8, d, 8 * G, 8, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < pad; ++i) if (buf[i] != CAN || buf[pad + n + i] != CAN) { ++bad_canary; fprintf(stderr, "CANARY hit: 2D copy of %llu rows, G = %llu, word %zu\n", (unsigned long long)n, (unsigned long long)G, i); break; }
    for (size_t i = 0; i < n; ++i) if (buf[pad + i] != 0x1000000000000000ull + i * G) { ++bad_value; break; }
    free(buf);
    // ---- small asynchronous copies into pageable heap memory, then a stream sync (mic_build.hip: h_scal, h_kept, h_max ...)
    const size_t words = 1 + rnd() % 4;
    unsigned long long* sm = (unsigned long long*)malloc((words + 4) * 8);
    for (size_t i = 0; i < words + 4; ++i) sm[i] = CAN;
    const size_t off = rnd() % (N - 8);
    CK(hipMemcpyAsync(sm + 2, d + off, words * 8, hipMemcpyDeviceToHost, s));
    // a 4-byte copy at an odd word offset, as for the u32 maxima
    uint32_t* w32 = (uint32_t*)malloc(5 * 4);
    for (int i = 0; i < 5; ++i) w32[i] = 0xABCD1234u;
    CK(hipMemcpyAsync(w32 + 2, (const char*)(d + off) + 4, 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (sm[0] != CAN || sm[1] != CAN || sm[2 + words] != CAN || sm[3 + words] != CAN) { ++bad_canary; fprintf(stderr, "CANARY hit: async copy of %zu words\n", words); }
    for (size_t i = 0; i < words; ++i) if (sm[2 + i] != 0x1000000000000000ull + off + i) { ++bad_value; break; }
    if (w32[0] != 0xABCD1234u || w32[1] != 0xABCD1234u || w32[3] != 0xABCD1234u || w32[4] != 0xABCD1234u) { ++bad_canary; fprintf(stderr, "CANARY hit: 4-byte async copy\n"); }
    if (w32[2] != 0x10000000u) ++bad_value;
    free(sm); free(w32);
  }
  printf("memcpy2d probe: %ld iterations, %ld canary hits, %ld wrong values\n", iters, bad_canary, bad_value);
  hipFree(d);
  return bad_canary || bad_value ? 1 : 0;
}
