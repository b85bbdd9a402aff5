// malloc_time.hip — how long hipMalloc / hipFree of a very large buffer take: tools/malloc_time [GB] [repeats]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec / 1e9; }
int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 173; const int rep = argc > 2 ? atoi(argv[2]) : 2;
  double t0 = now(); hipFree(0); printf("context: %.3f s\n", now() - t0);
  for (int i = 0; i < rep; ++i) {
    void* p = nullptr; t0 = now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
    printf("hipMalloc %.0f GB: %.3f s (%s)\n", gb, now() - t0, hipGetErrorString(e));
    t0 = now(); hipMemset(p, 0, 1 << 20); hipDeviceSynchronize(); printf("  first touch: %.3f s\n", now() - t0);
    t0 = now(); hipFree(p); printf("hipFree: %.3f s\n", now() - t0);
  }
  return 0;
}
