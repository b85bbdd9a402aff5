// guardalloc.c - a guard-page allocator to LD_PRELOAD in front of a process that uses libmi_clark.so (the native
// heap corruption seen three times in ~100 h of fuzzing; no GPU sanitizer and no XNACK on this pool, so the host side is what can
// be watched).  Every eligible allocation gets pages of its own inside a reserved arena and ENDS at an inaccessible page: a store
// past its end faults AT THE STORE (SIGSEGV with the writer's stack), not at some later free.  The 0..15 bytes between the block's
// end and the page (malloc's alignment) hold a canary checked at free; a freed block stays unmapped for the rest of the process
// (use after free faults too).  Everything else goes to the C library's allocator.
//
//   gcc -O2 -g -fPIC -shared -o guardalloc.so guardalloc.c -ldl -lpthread
//   GUARD_MIN=256 GUARD_MAX=67108864 GUARD_SAMPLE=1 GUARD_LIVE_MAX=24000 LD_PRELOAD=./guardalloc.so exe/cuCLARK ...
//     GUARD_MIN / GUARD_MAX  sizes that are guarded (default 256 B .. 64 MiB)
//     GUARD_SAMPLE           guard every n-th eligible allocation (1 = all; a Python process needs 8-32: its live blocks would
//                            exceed the kernel's limit of memory mappings, two per guarded block)
//     GUARD_LIVE_MAX         at most this many guarded blocks alive at once (default 24000); beyond it: the C library
//     GUARD_REPORT=1         at exit: allocations guarded / passed on, peak live, canary failures (stderr)
// On a fault inside the arena the handler prints what block the address belongs to (its size, whether it was freed) and the
// native backtrace, then re-raises (core dump / the previous handler, e.g. Python's faulthandler).
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#define PAGE 4096ul
#define ARENA_BYTES (1ul << 40)                  /* 1 TiB of address space, PROT_NONE until used, never reused */
#define HDR_MAGIC 0x6D69636775617264ul           /* "micguard" */
#define CANARY 0xA5

typedef struct { uint64_t magic, user_size, region_bytes, user_ptr; uint64_t freed; } hdr_t;

static void* (*real_malloc)(size_t);
static void (*real_free)(void*);
static void* (*real_calloc)(size_t, size_t);
static void* (*real_realloc)(void*, size_t);
static int (*real_posix_memalign)(void**, size_t, size_t);
static void* (*real_memalign)(size_t, size_t);
static void* (*real_aligned_alloc)(size_t, size_t);
static size_t (*real_usable)(void*);

static char* arena;                              /* reserved range */
static _Atomic uint64_t arena_top;               /* bump pointer (bytes) */
static uint32_t* page_region;                    /* arena page -> first page of its region + 1 (0: none) */
static _Atomic long n_live, n_guarded, n_passed, n_peak, n_canary;
static _Atomic unsigned long sample_ctr;
static size_t g_min = 256, g_max = 64ul << 20, g_sample = 1;
static long g_live_max = 24000;
static int g_ready, g_report;
static __thread int t_inside;                    /* re-entrancy (dlsym, backtrace, fprintf allocate) */
static struct sigaction prev_segv, prev_bus;

static char boot_buf[1 << 16];                   /* allocations before the real functions are resolved (dlsym calls calloc) */
static size_t boot_top;
static int in_boot(const void* p) { return (const char*)p >= boot_buf && (const char*)p < boot_buf + sizeof boot_buf; }
static void* boot_alloc(size_t n) {
  size_t a = (boot_top + 15) & ~(size_t)15;
  if (a + n > sizeof boot_buf) _exit(97);
  boot_top = a + n;
  return boot_buf + a;
}

static int in_arena(const void* p) { return arena && (const char*)p >= arena && (const char*)p < arena + ARENA_BYTES; }

static hdr_t* hdr_of(const void* p) {
  const uint64_t pg = ((const char*)p - arena) / PAGE;
  const uint32_t first = page_region[pg];
  if (!first) return NULL;
  hdr_t* h = (hdr_t*)(arena + (uint64_t)(first - 1) * PAGE);
  return h;
}

static void on_fault(int sig, siginfo_t* si, void* ctx) {
  void* addr = si ? si->si_addr : NULL;
  if (in_arena(addr)) {
    t_inside = 1;
    char msg[512];
    int n = snprintf(msg, sizeof msg, "\n[guardalloc] signal %d: access to %p inside the guarded arena\n", sig, addr);
    if (write(2, msg, n) < 0) {}
    /* the block in front of the address (an overrun lands on the guard page behind its block) */
    for (uint64_t pg = ((char*)addr - arena) / PAGE, back = 0; back < 3 && pg + 1 > back; ++back) {
      const uint32_t first = page_region[pg - back];
      if (!first) continue;
      hdr_t* h = (hdr_t*)(arena + (uint64_t)(first - 1) * PAGE);
      /* the header page of a freed block is unmapped: read it only while live (freed blocks are told apart by mincore-free logic:
         the header's own mapping state is kept in the table's top bit) */
      n = snprintf(msg, sizeof msg, "[guardalloc]   nearest block: region at %p (table entry %u)\n", (void*)h, first);
      if (write(2, msg, n) < 0) {}
      break;
    }
    void* bt[64];
    const int nb = backtrace(bt, 64);
    backtrace_symbols_fd(bt, nb, 2);
  } else if (!t_inside) {
    /* a fault somewhere else (a wild pointer, a library's own bug): the native stack all the same - Python's faulthandler, which
       runs next, only knows the interpreter's */
    t_inside = 1;
    char msg[160];
    const int n = snprintf(msg, sizeof msg, "\n[guardalloc] signal %d at address %p (outside the guarded arena); native stack:\n", sig, addr);
    if (write(2, msg, n) < 0) {}
    void* bt[64];
    const int nb = backtrace(bt, 64);
    backtrace_symbols_fd(bt, nb, 2);
    t_inside = 0;
  }
  struct sigaction* prev = sig == SIGBUS ? &prev_bus : &prev_segv;
  if (prev->sa_flags & SA_SIGINFO) { if (prev->sa_sigaction) { prev->sa_sigaction(sig, si, ctx); return; } }
  else if (prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN && prev->sa_handler) { prev->sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}

static void report(void) {
  if (!g_report) return;
  fprintf(stderr, "[guardalloc] guarded %ld allocations (peak %ld live), passed on %ld, canary failures %ld, arena used %.1f MB\n",
          (long)n_guarded, (long)n_peak, (long)n_passed, (long)n_canary, (double)arena_top / 1e6);
}

static int g_resolving;
__attribute__((constructor)) static void init(void) {
  if (g_ready || g_resolving) return;
  g_resolving = 1;                 /* (dlsym allocates: those few blocks come out of boot_buf) */
  t_inside = 1;
  real_malloc = dlsym(RTLD_NEXT, "malloc");
  real_free = dlsym(RTLD_NEXT, "free");
  real_calloc = dlsym(RTLD_NEXT, "calloc");
  real_realloc = dlsym(RTLD_NEXT, "realloc");
  real_posix_memalign = dlsym(RTLD_NEXT, "posix_memalign");
  real_memalign = dlsym(RTLD_NEXT, "memalign");
  real_aligned_alloc = dlsym(RTLD_NEXT, "aligned_alloc");
  real_usable = dlsym(RTLD_NEXT, "malloc_usable_size");
  const char* e;
  if ((e = getenv("GUARD_MIN"))) g_min = strtoul(e, NULL, 0);
  if ((e = getenv("GUARD_MAX"))) g_max = strtoul(e, NULL, 0);
  if ((e = getenv("GUARD_SAMPLE"))) { g_sample = strtoul(e, NULL, 0); if (!g_sample) g_sample = 1; }
  if ((e = getenv("GUARD_LIVE_MAX"))) g_live_max = strtol(e, NULL, 0);
  g_report = getenv("GUARD_REPORT") != NULL;
  arena = mmap(NULL, ARENA_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  page_region = mmap(NULL, (ARENA_BYTES / PAGE) * sizeof(uint32_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (arena == MAP_FAILED || page_region == MAP_FAILED) { arena = NULL; }
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigaction(SIGSEGV, &sa, &prev_segv);
  sigaction(SIGBUS, &sa, &prev_bus);
  atexit(report);
  g_ready = 1;
  g_resolving = 0;
  t_inside = 0;
}

/* a guarded block of n bytes with the given alignment (a power of two <= PAGE), or NULL: pass on */
static void* guard_alloc(size_t n, size_t align) {
  if (!arena || t_inside || n < g_min || n > g_max || align > PAGE) return NULL;
  if (g_sample > 1 && atomic_fetch_add(&sample_ctr, 1) % g_sample) return NULL;
  if (atomic_load(&n_live) >= g_live_max) return NULL;
  if (align < 16) align = 16;
  const size_t data = (n + align - 1) & ~(align - 1);               /* the block ends `slack` bytes in front of the guard page */
  const size_t data_pages = (data + PAGE - 1) / PAGE;
  const size_t region = (1 + data_pages + 1) * PAGE;                /* header page, data, guard */
  const uint64_t at = atomic_fetch_add(&arena_top, region);
  if (at + region > ARENA_BYTES) return NULL;
  char* base = arena + at;
  if (mprotect(base, (1 + data_pages) * PAGE, PROT_READ | PROT_WRITE) != 0) return NULL;      /* (out of mappings: pass on) */
  char* user = base + (1 + data_pages) * PAGE - data;
  hdr_t* h = (hdr_t*)base;
  h->magic = HDR_MAGIC; h->user_size = n; h->region_bytes = region; h->user_ptr = (uint64_t)user; h->freed = 0;
  memset(user + n, CANARY, data - n);
  const uint32_t first = (uint32_t)(at / PAGE) + 1;
  for (size_t p = 0; p < 1 + data_pages + 1; ++p) page_region[at / PAGE + p] = first;
  const long live = atomic_fetch_add(&n_live, 1) + 1;
  long pk = atomic_load(&n_peak);
  while (live > pk && !atomic_compare_exchange_weak(&n_peak, &pk, live)) {}
  atomic_fetch_add(&n_guarded, 1);
  return user;
}

static void guard_free(void* p) {
  hdr_t* h = hdr_of(p);
  if (!h || h->magic != HDR_MAGIC || h->user_ptr != (uint64_t)p || h->freed) {
    t_inside = 1;
    fprintf(stderr, "[guardalloc] free(%p): not the start of a live guarded block%s\n", p, h && h->magic == HDR_MAGIC && h->freed ? " (double free)" : "");
    void* bt[48]; const int nb = backtrace(bt, 48); backtrace_symbols_fd(bt, nb, 2);
    abort();
  }
  const size_t n = h->user_size, region = h->region_bytes;
  const size_t data_end = region - PAGE;                            /* offset of the guard page */
  const unsigned char* tail = (const unsigned char*)p + n;
  const unsigned char* end = (const unsigned char*)h + data_end;
  for (; tail < end; ++tail)
    if (*tail != CANARY) {
      atomic_fetch_add(&n_canary, 1);
      t_inside = 1;
      fprintf(stderr, "[guardalloc] free(%p): the %zu-byte block was overrun by %ld byte(s) (canary behind its end overwritten); freed from:\n",
              p, n, (long)(tail - ((const unsigned char*)p + n)) + 1);
      void* bt[48]; const int nb = backtrace(bt, 48); backtrace_symbols_fd(bt, nb, 2);
      abort();
    }
  h->freed = 1;
  /* the whole region becomes inaccessible and gives its memory back; the address range is never handed out again */
  mprotect(h, region, PROT_NONE);
  madvise(h, region, MADV_DONTNEED);
  atomic_fetch_sub(&n_live, 1);
}

void* malloc(size_t n) {
  if (!g_ready) { if (g_resolving) return boot_alloc(n); init(); }      /* first call (another library's constructor may come before ours) */
  void* p = guard_alloc(n, 16);
  if (p) return p;
  atomic_fetch_add(&n_passed, 1);
  return real_malloc(n);
}

void free(void* p) {
  if (!p || in_boot(p)) return;
  if (in_arena(p)) { guard_free(p); return; }
  real_free(p);
}

void* calloc(size_t a, size_t b) {
  if (!g_ready) {
    if (g_resolving) { void* p = boot_alloc(a * b); memset(p, 0, a * b); return p; }
    init();
  }
  if (b && a > (size_t)-1 / b) { errno = ENOMEM; return NULL; }
  void* p = guard_alloc(a * b, 16);                                  /* (fresh anonymous pages are zero) */
  if (p) return p;
  atomic_fetch_add(&n_passed, 1);
  return real_calloc(a, b);
}

void* realloc(void* p, size_t n) {
  if (!g_ready && !g_resolving) init();
  if (!p) return malloc(n);
  if (in_boot(p)) { void* q = malloc(n); if (q) memcpy(q, p, n); return q; }      /* (boot blocks are small; their size is not kept) */
  if (in_arena(p)) {
    hdr_t* h = hdr_of(p);
    if (!h || h->magic != HDR_MAGIC || h->user_ptr != (uint64_t)p) { guard_free(p); return NULL; }
    if (n == 0) { guard_free(p); return NULL; }
    void* q = malloc(n);
    if (!q) return NULL;
    memcpy(q, p, h->user_size < n ? h->user_size : n);
    guard_free(p);
    return q;
  }
  void* g = guard_alloc(n, 16);
  if (g) {                                                           /* a block of the C library that grows into a guarded one */
    const size_t old = real_usable ? real_usable(p) : n;
    memcpy(g, p, old < n ? old : n);
    real_free(p);
    return g;
  }
  return real_realloc(p, n);
}

int posix_memalign(void** out, size_t align, size_t n) {
  if (!g_ready && !g_resolving) init();
  void* p = guard_alloc(n, align);
  if (p) { *out = p; return 0; }
  return real_posix_memalign(out, align, n);
}
void* memalign(size_t align, size_t n) { if (!g_ready && !g_resolving) init(); void* p = guard_alloc(n, align); return p ? p : real_memalign(align, n); }
void* aligned_alloc(size_t align, size_t n) { if (!g_ready && !g_resolving) init(); void* p = guard_alloc(n, align); return p ? p : real_aligned_alloc(align, n); }
size_t malloc_usable_size(void* p) {
  if (!p) return 0;
  if (in_arena(p)) { hdr_t* h = hdr_of(p); return h ? h->user_size : 0; }
  if (in_boot(p)) return 0;
  return real_usable ? real_usable(p) : 0;
}
