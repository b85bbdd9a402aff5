// hip_mock.cpp - a stand-in for the HIP runtime on a machine WITHOUT a GPU, for one purpose: the library's HOST code
// (mic_engine.hip, mic_build.hip, mic_ingest.hip, mic_gz.hip, mic_synth.hip: arenas, sizing arithmetic, batch and slot
// bookkeeping, everything that hands host memory to asynchronous copies) compiled host-only (hipcc --offload-host-only) under
// AddressSanitizer + UBSan and run on the CPU (tools/sanitize/host_rig.cpp, tests/test_sanitizers.py).  VERDICT r3 item 4c.
//
//   * device memory is host memory (calloc): the sanitizer sees every byte the host code copies in and out of it;
//   * kernels do not run (hipLaunchKernel is a no-op that keeps the stream's order): what comes back from the "device" is zeros,
//     so the rig drives the paths whose control flow does not hang on a kernel's answer, and checks status codes only;
//   * ASYNCHRONOUS COPIES ARE DEFERRED until something waits for them (stream / event / device synchronisation, a synchronous
//     copy, a free), in stream order, with stream-to-stream waits honoured - as late as the real runtime may run them.  A host
//     buffer that dies, or a stack variable whose frame returns, while a copy into or out of it is still queued is then a
//     use-after-free / use-after-return the sanitizer reports with both stacks - the class of bug the fuzzer's rare native
//     fault points at.
// Test infrastructure: nothing of the product links this file.
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <vector>

namespace {

struct Event;
struct Op {
  enum Kind { COPY, COPY2D, SET, LAUNCH, RECORD, WAIT } kind;
  void* dst; const void* src; size_t n; int value;
  size_t dpitch, spitch, width, height;
  Event* ev;
};
struct Stream { std::deque<Op> q; int device; };
struct Event { Stream* on = nullptr; bool pending = false; };

std::recursive_mutex g_mu;
Stream g_null{{}, 0};
std::set<Stream*> g_streams;
int g_n_devices = 1;
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;

Stream* S(hipStream_t s) { return s ? (Stream*)s : &g_null; }

void run(Stream* st, Event* until);
void run_op(const Op& o) {
  switch (o.kind) {
    case Op::COPY: if (o.n) memmove(o.dst, o.src, o.n); break;
    case Op::COPY2D: for (size_t r = 0; r < o.height; ++r) memmove((char*)o.dst + r * o.dpitch, (const char*)o.src + r * o.spitch, o.width); break;
    case Op::SET: if (o.n) memset(o.dst, o.value, o.n); break;
    case Op::LAUNCH: break;
    case Op::RECORD: o.ev->pending = false; break;
    case Op::WAIT: if (o.ev->pending && o.ev->on) run(o.ev->on, o.ev); break;
  }
}
// everything queued on st, or everything up to and including the record of `until`
void run(Stream* st, Event* until) {
  while (!st->q.empty()) {
    const Op o = st->q.front();
    st->q.pop_front();
    run_op(o);
    if (until && o.kind == Op::RECORD && o.ev == until) return;
  }
}
void run_all() {
  run(&g_null, nullptr);
  for (Stream* s : g_streams) run(s, nullptr);
}

}  // namespace

extern "C" {

// ---- what the compiler's kernel stubs and the fat-binary registration call
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static thread_local struct { dim3 g, b; size_t shm; hipStream_t s; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t s) { t_cfg.g = g; t_cfg.b = b; t_cfg.shm = shm; t_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* s) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *s = t_cfg.s; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t s) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Op o{}; o.kind = Op::LAUNCH; S(s)->q.push_back(o);
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// ---- devices
hipError_t hipGetDeviceCount(int* n) { if (const char* e = getenv("MOCK_HIP_DEVICES")) g_n_devices = atoi(e) > 0 ? atoi(e) : 1; *n = g_n_devices; return hipSuccess; }
hipError_t hipSetDevice(int d) { t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "mock MI355X");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
  p->totalGlobalMem = (size_t)8 << 30;
  p->multiProcessorCount = 256;
  p->warpSize = 64;
  p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 65536;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->clockRate = 2400000;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : a == hipDeviceAttributeWarpSize ? 64 : 0;
  return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d) { snprintf(buf, (size_t)len, "0000:%02x:00.0", d); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = (size_t)6 << 30; *tot = (size_t)8 << 30; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "mock error"; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }

// ---- memory: the "device" is the host
hipError_t hipMalloc(void** p, size_t n) {
  *p = calloc(n ? n : 1, 1);
  if (!*p) { t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  return hipSuccess;
}
hipError_t hipFree(void* p) { std::lock_guard<std::recursive_mutex> lk(g_mu); run_all(); free(p); return hipSuccess; }   // (hipFree waits for the device)
// Pinned host memory: the mock does NOT assume that hipHostFree waits for the device - a copy still queued into or out of the
// block when it is freed is reported and the process aborts (the library must have waited for its own copies by then).
static std::map<void*, size_t>& g_pinned = *new std::map<void*, size_t>();
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = calloc(n ? n : 1, 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  g_pinned[*p] = n ? n : 1;
  return hipSuccess;
}
static bool touches(const Op& o, const char* lo, const char* hi) {
  if (o.kind != Op::COPY && o.kind != Op::SET) return false;
  const char* d = (const char*)o.dst; const char* s = (const char*)o.src;
  return (d && d < hi && d + o.n > lo) || (o.kind == Op::COPY && s && s < hi && s + o.n > lo);
}
hipError_t hipHostFree(void* p) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  auto it = g_pinned.find(p);
  if (p && it != g_pinned.end()) {
    const char* lo = (const char*)p; const char* hi = lo + it->second;
    bool queued = false;
    for (const Op& o : g_null.q) queued |= touches(o, lo, hi);
    for (Stream* st : g_streams) for (const Op& o : st->q) queued |= touches(o, lo, hi);
    if (queued) { fprintf(stderr, "hip_mock: hipHostFree(%p) with a copy into or out of the block still queued\n", p); abort(); }
    g_pinned.erase(it);
  }
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  run_all();
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { std::lock_guard<std::recursive_mutex> lk(g_mu); run_all(); if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Op o{}; o.kind = Op::COPY; o.dst = d; o.src = s; o.n = n; S(st)->q.push_back(o);
  return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyDefault, st); }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  run_all();
  for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Op o{}; o.kind = Op::SET; o.dst = d; o.value = v; o.n = n; S(st)->q.push_back(o);
  return hipSuccess;
}

// ---- streams and events
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Stream* st = new Stream{{}, t_device};
  g_streams.insert(st);
  *s = (hipStream_t)st;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  // the library pools its streams and events for the life of the process (mic_engine.hip: mic_stream_put - the runtime's completion
  // handler races their destruction): a destroy is a regression
  if (!getenv("MOCK_HIP_ALLOW_DESTROY")) { fprintf(stderr, "hip_mock: hipStreamDestroy called - streams are pooled, never destroyed\n"); abort(); }
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Stream* st = (Stream*)s;
  run(st, nullptr);
  g_streams.erase(st);
  delete st;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { std::lock_guard<std::recursive_mutex> lk(g_mu); run(S(s), nullptr); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { std::lock_guard<std::recursive_mutex> lk(g_mu); run_all(); return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t s) { return s ? ((Stream*)s)->device : t_device; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new Event(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
static std::vector<Event*>& g_graveyard = *new std::vector<Event*>();       // (a destroyed event may still be waited for by a queued operation: the real
                                                                            // runtime keeps it alive too; never freed, and reachable for the leak check)
hipError_t hipEventDestroy(hipEvent_t e) {
  if (!getenv("MOCK_HIP_ALLOW_DESTROY")) { fprintf(stderr, "hip_mock: hipEventDestroy called - events are pooled, never destroyed\n"); abort(); }
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  g_graveyard.push_back((Event*)e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Event* ev = (Event*)e;
  if (ev->pending && ev->on) run(ev->on, ev);          // a second record of an event that has not fired: the first one first
  ev->on = S(s); ev->pending = true;
  Op o{}; o.kind = Op::RECORD; o.ev = ev; S(s)->q.push_back(o);
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Event* ev = (Event*)e;
  if (ev->pending && ev->on) run(ev->on, ev);
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) { return hipEventSynchronize(e); }     // (a poll would spin on a mock that runs nothing by itself)
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  Op o{}; o.kind = Op::WAIT; o.ev = (Event*)e; S(s)->q.push_back(o);
  return hipSuccess;
}

}  // extern "C"
