// host_link_probe.hip — what the host side of the GPU box can feed: cores, pinned H2D / D2H bandwidth (one and both
// directions), multi-threaded memcpy into pinned memory, pread of a page-cached file into pinned memory, pwrite.
// Sizes the ingest pipeline.  Build: hipcc --offload-arch=gfx950 -O2 -o build/tools/host_link_probe $0 -lpthread
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <algorithm>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static void par(int n, F&& f) {
  std::vector<std::thread> th;
  for (int t = 1; t < n; ++t) th.emplace_back([&f, t] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

__global__ void spin_kernel(unsigned long long cycles, unsigned* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (sink && threadIdx.x == 1024) *sink = 1;
}

// the batch pipeline's shape: NB batches on NB streams, each H2D (in) -> kernel (kus microseconds) -> D2H (out)
static void pipeline_shape(void* h_in, void* d_in, void* h_out, void* d_out, size_t in_bytes, size_t out_bytes, int NB, int kus, bool d2h) {
  std::vector<hipStream_t> st(NB);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t pi = in_bytes / NB & ~(size_t)255, po = out_bytes / NB & ~(size_t)255;
  double best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    const double t = now();
    for (int b = 0; b < NB; ++b) {
      CK(hipMemcpyAsync((char*)d_in + b * pi, (char*)h_in + b * pi, pi, hipMemcpyHostToDevice, st[b]));
      if (kus) spin_kernel<<<256 * 8, 256, 0, st[b]>>>((unsigned long long)kus * 100, nullptr);   // s_memtime-ish counter: 100 MHz
      if (d2h) CK(hipMemcpyAsync((char*)h_out + b * po, (char*)d_out + b * po, po, hipMemcpyDeviceToHost, st[b]));
    }
    for (auto& s : st) CK(hipStreamSynchronize(s));
    best = std::min(best, now() - t);
  }
  printf("pipeline shape: %2d batches, H2D %.0f MB%s%s: %.2f ms -> H2D %.1f GB/s\n", NB, in_bytes / 1e6, kus ? ", kernel" : "", d2h ? ", D2H" : "",
         best * 1e3, pi * NB / best / 1e9);
  for (auto& s : st) hipStreamDestroy(s);
}

// the same with ONE upload stream and ONE download stream (events tie a batch's kernel between them)
static void pipeline_shape2(void* h_in, void* d_in, void* h_out, void* d_out, size_t in_bytes, size_t out_bytes, int NB, int kus) {
  std::vector<hipStream_t> st(NB);
  std::vector<hipEvent_t> e1(NB), e2(NB);
  hipStream_t up, down;
  CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
  for (int b = 0; b < NB; ++b) { CK(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&e1[b], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2[b], hipEventDisableTiming)); }
  const size_t pi = in_bytes / NB & ~(size_t)255, po = out_bytes / NB & ~(size_t)255;
  double best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    const double t = now();
    for (int b = 0; b < NB; ++b) {
      CK(hipMemcpyAsync((char*)d_in + b * pi, (char*)h_in + b * pi, pi, hipMemcpyHostToDevice, up));
      CK(hipEventRecord(e1[b], up));
      CK(hipStreamWaitEvent(st[b], e1[b], 0));
      spin_kernel<<<256 * 8, 256, 0, st[b]>>>((unsigned long long)kus * 100, nullptr);
      CK(hipEventRecord(e2[b], st[b]));
      CK(hipStreamWaitEvent(down, e2[b], 0));
      CK(hipMemcpyAsync((char*)h_out + b * po, (char*)d_out + b * po, po, hipMemcpyDeviceToHost, down));
    }
    CK(hipStreamSynchronize(down));
    best = std::min(best, now() - t);
  }
  printf("pipeline shape, one upload + one download stream: %2d batches: %.2f ms -> H2D %.1f GB/s\n", NB, best * 1e3, pi * NB / best / 1e9);
}

int main(int argc, char** argv) {
  const size_t GB = (size_t)(argc > 1 ? atof(argv[1]) * (1 << 30) : (size_t)2 << 30);
  cpu_set_t cs; CPU_ZERO(&cs); sched_getaffinity(0, sizeof(cs), &cs);
  printf("cores: online %ld, affinity %d\n", sysconf(_SC_NPROCESSORS_ONLN), CPU_COUNT(&cs));
  { FILE* f = fopen("/proc/meminfo", "r"); char l[256]; for (int i = 0; i < 3 && fgets(l, sizeof l, f); ++i) fputs(l, stdout); fclose(f); }
  { FILE* f = popen("lscpu | grep -E 'Model name|Socket|NUMA node\\(s\\)|Thread' ; cat /sys/fs/cgroup/cpu.max 2>/dev/null", "r"); char l[256]; while (f && fgets(l, sizeof l, f)) fputs(l, stdout); if (f) pclose(f); }
  void *h0, *h1, *d0, *d1;
  CK(hipHostMalloc(&h0, GB, hipHostMallocDefault)); CK(hipHostMalloc(&h1, GB, hipHostMallocDefault));
  CK(hipMalloc(&d0, GB)); CK(hipMalloc(&d1, GB));
  memset(h0, 1, GB); memset(h1, 2, GB);
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; ++rep) {
    double t = now(); CK(hipMemcpyAsync(d0, h0, GB, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
    printf("H2D pinned %.1f GB: %.1f GB/s\n", GB / 1e9, GB / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(h1, d1, GB, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
    printf("D2H pinned: %.1f GB/s\n", GB / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(d0, h0, GB, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h1, d1, GB, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    printf("H2D + D2H concurrently: %.1f GB/s each way\n", GB / (now() - t) / 1e9);
  }
  // chunked H2D (8 MB pieces on two streams): what a batch pipeline sees
  for (size_t ch : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
    double t = now(); int i = 0;
    for (size_t o = 0; o < GB; o += ch, ++i) CK(hipMemcpyAsync((char*)d0 + o, (char*)h0 + o, ch, hipMemcpyHostToDevice, (i & 1) ? s1 : s0));
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    printf("H2D in %zu MB pieces on 2 streams: %.1f GB/s\n", ch >> 20, GB / (now() - t) / 1e9);
  }
  for (int NB : {1, 4, 16}) {
    pipeline_shape(h0, d0, h1, d1, 760u << 20, 320u << 20, NB, 0, false);
    pipeline_shape(h0, d0, h1, d1, 760u << 20, 320u << 20, NB, 0, true);
    pipeline_shape(h0, d0, h1, d1, 760u << 20, 320u << 20, NB, 7200 / NB, true);
    if (NB > 1) pipeline_shape2(h0, d0, h1, d1, 760u << 20, 320u << 20, NB, 7200 / NB);
  }
  // host memcpy pageable -> pinned with T threads
  char* src = (char*)malloc(GB); memset(src, 3, GB);
  for (int T : {1, 2, 4, 8, 16, 32, 64}) {
    double t = now();
    par(T, [&](int i) { size_t per = GB / T; memcpy((char*)h0 + i * per, src + i * per, per); });
    printf("memcpy pageable->pinned, %2d threads: %.1f GB/s\n", T, GB / (now() - t) / 1e9);
  }
  // page-cached file -> pinned via pread, T threads; and mmap + memcpy
  const char* path = "/tmp/host_link_probe.bin";
  { int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644); double t = now(); size_t w = 0; while (w < GB) { ssize_t r = write(fd, src + w, GB - w > ((size_t)64 << 20) ? ((size_t)64 << 20) : GB - w); if (r <= 0) break; w += r; } close(fd);
    printf("write() 1 thread to %s: %.1f GB/s\n", path, GB / (now() - t) / 1e9); }
  for (int T : {1, 2, 4, 8, 16, 32}) {
    int fd = open(path, O_RDONLY);
    double t = now();
    par(T, [&](int i) { size_t per = GB / T, got = 0; while (got < per) { ssize_t r = pread(fd, (char*)h0 + i * per + got, per - got, i * per + got); if (r <= 0) break; got += r; } });
    printf("pread page cache->pinned, %2d threads: %.1f GB/s\n", T, GB / (now() - t) / 1e9);
    close(fd);
  }
  for (int T : {1, 4, 8, 16, 32}) {
    int fd = open(path, O_WRONLY);
    double t = now();
    par(T, [&](int i) { size_t per = GB / T, got = 0; while (got < per) { ssize_t r = pwrite(fd, (char*)h1 + i * per + got, per - got, i * per + got); if (r <= 0) break; got += r; } });
    printf("pwrite pinned->page cache (existing pages), %2d threads: %.1f GB/s\n", T, GB / (now() - t) / 1e9);
    close(fd);
  }
  unlink(path);
  // kernel launch + event sync round trip
  { hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); double t = now();
    for (int i = 0; i < 200; ++i) { CK(hipMemcpyAsync(d0, h0, 4096, hipMemcpyHostToDevice, s0)); CK(hipEventRecord(ev, s0)); CK(hipEventSynchronize(ev)); }
    printf("4 KB H2D + event sync round trip: %.1f us\n", (now() - t) / 200 * 1e6); }
  return 0;
}
