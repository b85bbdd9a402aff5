// mic_build.hip — builds the resident slot table in HBM from the on-disk arrays (.sz/.ky/.lb images).
//
// Replaces CuClarkDB::read's host-side u8 -> u32 prefix sums and part copies (CuClarkDB.cu:594-648,
// 678-782) and swapDbParts' upload (:813-858): the raw images are uploaded once and the table is laid
// out on the GPU.  One thread per bucket; three light passes over the bucket sizes:
//   A  per-tile sums of (elements, non-empty buckets)           -> host scan -> tile bases
//   B  per-tile overflow-slot demand / kept elements / max size -> host scan -> overflow bases
//   C  slot construction (reachability filter, chain layout)
#include "mic_internal.h"
#include "mic_device.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <math.h>

#include <hipcub/hipcub.hpp>

#define TILE 256

namespace {

struct TileA { unsigned long long elems; unsigned long long nonzero; };
struct TileB { unsigned long long ovf_slots; unsigned long long kept; unsigned int max_bucket; unsigned int pad; };

// exclusive scan of (a,b) pairs over a 256-thread block; returns this thread's exclusive prefix and the
// block totals through tot_a/tot_b.
__device__ inline void block_scan2(uint32_t a, uint32_t b, uint32_t& ex_a, uint32_t& ex_b, uint32_t& tot_a,
                                   uint32_t& tot_b) {
  __shared__ uint32_t sa[TILE], sb[TILE];
  const int t = threadIdx.x;
  sa[t] = a; sb[t] = b;
  __syncthreads();
  for (int off = 1; off < TILE; off <<= 1) {
    uint32_t va = 0, vb = 0;
    if (t >= off) { va = sa[t - off]; vb = sb[t - off]; }
    __syncthreads();
    sa[t] += va; sb[t] += vb;
    __syncthreads();
  }
  ex_a = sa[t] - a; ex_b = sb[t] - b;
  tot_a = sa[TILE - 1]; tot_b = sb[TILE - 1];
  __syncthreads();
}

__global__ void __launch_bounds__(TILE) tile_a_kernel(const uint8_t* __restrict__ sizes, uint64_t n, TileA* out) {
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t sz = i < n ? sizes[i] : 0;
  uint32_t ea, eb, ta, tb;
  block_scan2(sz, sz > 0, ea, eb, ta, tb);
  if (threadIdx.x == 0) { out[blockIdx.x].elems = ta; out[blockIdx.x].nonzero = tb; }
}

__device__ inline bool kept_bucket(uint32_t sz, uint64_t rank_incl, uint32_t sampling) {
  // CuClarkDB.cu:508-519: choice = (all || nbNonZeroBuckets % mod == 0) ? keep : skip, counted over non-empty buckets
  return sz > 0 && (sampling <= 1 || (rank_incl % sampling) == 0);
}

template <int CAP>
__global__ void __launch_bounds__(TILE) tile_b_kernel(const uint8_t* __restrict__ sizes, uint64_t n,
                                                      const TileA* __restrict__ base, uint32_t sampling,
                                                      uint64_t rank_base, TileB* out) {
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t sz = i < n ? sizes[i] : 0;
  uint32_t ea, eb, ta, tb;
  block_scan2(sz, sz > 0, ea, eb, ta, tb);
  uint64_t rank_incl = rank_base + base[blockIdx.x].nonzero + eb + (sz > 0);
  bool keep = kept_bucket(sz, rank_incl, sampling);
  uint32_t ovf = (keep && sz > CAP) ? (sz - CAP + CAP - 1) / CAP : 0;
  uint32_t kept = keep ? sz : 0;
  uint32_t e1, e2, t1, t2;
  block_scan2(ovf, kept, e1, e2, t1, t2);
  __shared__ uint32_t smax;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  if (kept) atomicMax(&smax, kept);
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x].ovf_slots = t1; out[blockIdx.x].kept = t2; out[blockIdx.x].max_bucket = smax; out[blockIdx.x].pad = 0;
  }
}

template <typename RAW>
__device__ inline uint64_t raw_key(const void* keys, uint64_t i) { return (uint64_t)((const RAW*)keys)[i]; }

// Emits the chain of one bucket.  Reachable entries are visited twice (count, then emit).
template <typename RAW, bool KEY64>
__device__ inline uint32_t build_bucket(const void* __restrict__ keys, const uint16_t* __restrict__ labels,
                                        uint64_t off, uint32_t n_raw, uint4* __restrict__ slots, uint64_t main_idx,
                                        uint64_t ovf_idx) {
  constexpr int CAP = KEY64 ? MIC_CAP64 : MIC_CAP32;
  // pass 1: count entries the reference scan can reach (CuClarkDB.cu:1291-1307)
  uint32_t m = 0;
  uint64_t last = 0;
  if (n_raw) {
    last = raw_key<RAW>(keys, off + n_raw - 1);
    uint64_t run = 0; bool first = true;
    for (uint32_t i = 0; i < n_raw; ++i) {
      uint64_t kv = raw_key<RAW>(keys, off + i);
      if ((first || kv > run) && kv <= last) ++m;
      if (first || kv > run) { run = kv; first = false; }
    }
  }
  // pass 2: emit slots
  uint32_t emitted = 0, i = 0;
  uint64_t run = 0; bool first = true;
  uint64_t slot = main_idx;
  uint64_t next = ovf_idx;
  do {
    uint64_t kk[CAP]; uint32_t ll[CAP];
#pragma unroll
    for (int e = 0; e < CAP; ++e) { kk[e] = ~0ULL; ll[e] = 0; }
    uint32_t remaining = m - emitted;
    int pos = 0;
    while (pos < CAP && i < n_raw) {
      uint64_t kv = raw_key<RAW>(keys, off + i);
      bool reach = (first || kv > run) && kv <= last;
      if (first || kv > run) { run = kv; first = false; }
      if (reach) {
        uint32_t lb = labels[off + i];
#pragma unroll
        for (int e = 0; e < CAP; ++e) if (pos == e) { kk[e] = kv; ll[e] = lb; }
        ++pos; ++emitted;
      }
      ++i;
    }
    uint32_t nmeta = remaining > 255 ? 255 : remaining;
    uint32_t w0 = nmeta | (uint32_t)((next & 0xFFFFFF) << 8);
    uint32_t w1 = nmeta | (uint32_t)(((next >> 24) & 0xFFFFFF) << 8);
    uint4 q[4];
    if constexpr (KEY64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = make_uint4((uint32_t)kk[j], (uint32_t)(kk[j] >> 32), ll[j], nmeta);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        q[j] = make_uint4((uint32_t)kk[2 * j], (uint32_t)kk[2 * j + 1], ll[2 * j] | (ll[2 * j + 1] << 16), nmeta);
    }
    q[0].w = w0; q[1].w = w1;
#pragma unroll
    for (int j = 0; j < 4; ++j) slots[slot * 4 + j] = q[j];
    slot = next; ++next;
  } while (emitted < m);
  return m;
}

template <typename RAW, bool KEY64>
__global__ void __launch_bounds__(TILE) tile_c_kernel(const uint8_t* __restrict__ sizes, uint64_t n,
                                                      const void* __restrict__ keys,
                                                      const uint16_t* __restrict__ labels,
                                                      const TileA* __restrict__ base_a,
                                                      const TileB* __restrict__ base_b, uint32_t sampling,
                                                      uint64_t rank_base, uint4* __restrict__ slots, uint64_t n_main,
                                                      unsigned long long* __restrict__ kept_elems) {
  constexpr int CAP = KEY64 ? MIC_CAP64 : MIC_CAP32;
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t sz = i < n ? sizes[i] : 0;
  uint32_t ea, eb, ta, tb;
  block_scan2(sz, sz > 0, ea, eb, ta, tb);
  uint64_t rank_incl = rank_base + base_a[blockIdx.x].nonzero + eb + (sz > 0);
  bool keep = kept_bucket(sz, rank_incl, sampling);
  uint32_t ovf = (keep && sz > CAP) ? (sz - CAP + CAP - 1) / CAP : 0;
  uint32_t e1, e2, t1, t2;
  block_scan2(ovf, 0, e1, e2, t1, t2);
  if (i >= n) return;
  uint64_t off = base_a[blockIdx.x].elems + ea;
  uint64_t ovf_idx = n_main + base_b[blockIdx.x].ovf_slots + e1;
  uint32_t m = build_bucket<RAW, KEY64>(keys, labels, off, keep ? sz : 0, slots, i, ovf_idx);
  if (m) atomicAdd(kept_elems, (unsigned long long)m);
}

__global__ void reduce_sizes_kernel(const uint8_t* __restrict__ sizes, uint64_t n, unsigned long long* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long tot = 0, nz = 0;
  for (; i < n; i += stride) { uint32_t s = sizes[i]; tot += s; nz += s > 0; }
  for (int off = 32; off > 0; off >>= 1) { tot += __shfl_down(tot, off); nz += __shfl_down(nz, off); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], tot); atomicAdd(&out[1], nz); }
}

// an allocation that fails for lack of memory is reported as -3 (MIC_E_NOMEM): the engine then falls back to a smaller
// layout when the layout was chosen by default (mic_engine.hip: build_from_device)
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    snprintf(err, err_cap, "%s failed: %s", #x, hipGetErrorString(e_)); (void)hipGetLastError(); \
    rc = e_ == hipErrorOutOfMemory ? -3 : -4; goto done; } } while (0)

}  // namespace

int mic_reduce_sizes(const uint8_t* d_sizes, uint64_t n, uint64_t* total, uint64_t* nonzero, hipStream_t s) {
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 16) != hipSuccess) return -3;
  unsigned long long h[2] = {0, 0};
  hipError_t e = hipMemsetAsync(d, 0, 16, s);
  if (e == hipSuccess && n) {
    reduce_sizes_kernel<<<2048, 256, 0, s>>>(d_sizes, n, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d);
  if (e != hipSuccess) return -4;
  *total = h[0]; *nonzero = h[1];
  return 0;
}

template <int CAP>
static void launch_b(const uint8_t* sizes, uint64_t n, const TileA* a, uint32_t sampling, uint64_t rank_base, TileB* b,
                     unsigned n_tiles, hipStream_t s) {
  tile_b_kernel<CAP><<<n_tiles, TILE, 0, s>>>(sizes, n, a, sampling, rank_base, b);
}

int mic_build_table(const uint8_t* d_sizes, uint64_t n_buckets, const void* d_keys, int key_bytes,
                    const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int slot_class, hipStream_t s,
                    MicBuildOut* out, char* err, size_t err_cap) {
  int rc = 0;
  const uint64_t n_tiles64 = (n_buckets + TILE - 1) / TILE;
  const unsigned n_tiles = (unsigned)n_tiles64;
  TileA* d_a = nullptr; TileB* d_b = nullptr; unsigned long long* d_kept = nullptr; uint4* slots = nullptr;
  std::vector<TileA> h_a(n_tiles);
  std::vector<TileB> h_b(n_tiles);
  uint64_t tot_elems = 0, tot_nz = 0, tot_ovf = 0, tot_kept = 0; uint32_t maxb = 0;
  unsigned long long h_kept = 0;
  if (n_buckets == 0 || n_tiles64 > 0x7fffffffULL) { snprintf(err, err_cap, "bad bucket count"); return -1; }
  HIPCK(hipMalloc(&d_a, sizeof(TileA) * n_tiles));
  HIPCK(hipMalloc(&d_b, sizeof(TileB) * n_tiles));
  HIPCK(hipMalloc(&d_kept, 8));
  HIPCK(hipMemsetAsync(d_kept, 0, 8, s));
  // pass A
  tile_a_kernel<<<n_tiles, TILE, 0, s>>>(d_sizes, n_buckets, d_a);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(h_a.data(), d_a, sizeof(TileA) * n_tiles, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  for (unsigned t = 0; t < n_tiles; ++t) {
    uint64_t e = h_a[t].elems, z = h_a[t].nonzero;
    h_a[t].elems = tot_elems; h_a[t].nonzero = tot_nz;
    tot_elems += e; tot_nz += z;
  }
  HIPCK(hipMemcpyAsync(d_a, h_a.data(), sizeof(TileA) * n_tiles, hipMemcpyHostToDevice, s));
  // pass B
  if (slot_class == 64) launch_b<MIC_CAP64>(d_sizes, n_buckets, d_a, sampling, rank_base, d_b, n_tiles, s);
  else launch_b<MIC_CAP32>(d_sizes, n_buckets, d_a, sampling, rank_base, d_b, n_tiles, s);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(h_b.data(), d_b, sizeof(TileB) * n_tiles, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  for (unsigned t = 0; t < n_tiles; ++t) {
    uint64_t o = h_b[t].ovf_slots;
    h_b[t].ovf_slots = tot_ovf; tot_ovf += o; tot_kept += h_b[t].kept;
    if (h_b[t].max_bucket > maxb) maxb = h_b[t].max_bucket;
  }
  HIPCK(hipMemcpyAsync(d_b, h_b.data(), sizeof(TileB) * n_tiles, hipMemcpyHostToDevice, s));
  {
    hipError_t e_ = hipMalloc(&slots, (size_t)(n_buckets + tot_ovf + 1) * MIC_SLOT_BYTES);
    if (e_ != hipSuccess) {
      snprintf(err, err_cap, "hipMalloc of %.2f GB for the slot table failed: %s",
               (double)(n_buckets + tot_ovf + 1) * MIC_SLOT_BYTES / 1e9, hipGetErrorString(e_));
      rc = -3; goto done;
    }
  }
  // pass C
#define LAUNCH_C(RAW, K64) tile_c_kernel<RAW, K64><<<n_tiles, TILE, 0, s>>>(d_sizes, n_buckets, d_keys, d_labels, d_a, \
    d_b, sampling, rank_base, slots, n_buckets, d_kept)
  if (slot_class == 64) {
    if (key_bytes == 8) LAUNCH_C(uint64_t, true);
    else if (key_bytes == 4) LAUNCH_C(uint32_t, true);
    else LAUNCH_C(uint16_t, true);
  } else {
    if (key_bytes == 4) LAUNCH_C(uint32_t, false);
    else if (key_bytes == 2) LAUNCH_C(uint16_t, false);
    else { snprintf(err, err_cap, "slot class 32 cannot hold 8-byte keys"); rc = -1; goto done; }
  }
#undef LAUNCH_C
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(&h_kept, d_kept, 8, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  out->slots = slots; slots = nullptr;
  out->n_main = n_buckets; out->n_overflow = tot_ovf; out->n_elems = h_kept; out->n_elems_file = tot_elems;
  out->max_bucket = maxb;
  (void)tot_kept;
done:
  // (an error return: copies queued on s may still name this frame's host variables - they must have landed before it goes)
  if (rc) hipStreamSynchronize(s);
  if (d_a) hipFree(d_a);
  if (d_b) hipFree(d_b);
  if (d_kept) hipFree(d_kept);
  if (slots) hipFree(slots);
  return rc;
}


// =====================================================================================================================
// Minimizer-keyed table (layout 1).  Passes:
//   A   (shared) per-tile element / non-empty sums            -> raw offsets, sampling ranks
//   M1  per bucket: reachable entries -> canonical k-mer c = key*H + bucket -> slot(c) -> count[slot]++
//   M2  per-tile sums of overflow-slot demand -> host scan -> per slot: headers of its whole chain (keys = ~0)
//   M3  per bucket again: scatter (c, label) to chain position atomicAdd(cursor[slot])
//   M4  per slot: sort the chain by c (shell sort over the virtual array)
// =====================================================================================================================
namespace {

struct MSlot {
  unsigned long long keys[MIC_MCAP];
  unsigned short labels[MIC_MCAP];
  unsigned int meta;
  unsigned int next;
};
static_assert(sizeof(MSlot) == MIC_MSLOT_BYTES, "M-slot must be 128 bytes");

struct MBuildArgs {
  const uint8_t* sizes; uint64_t n_buckets; uint64_t bucket0; uint64_t htsize;
  const void* keys; const uint16_t* labels;
  const TileA* base_a; uint32_t sampling; uint64_t rank_base;
  int k, m; uint64_t n_mslots;
  int fwd;     // super-k-mer table: both strands under forward-strand minimizers (mic_device.h: s_candidates_fwd)
};

// visit the reachable entries of this thread's bucket: f(c, label)
template <typename RAW, typename F>
__device__ inline void for_reachable(const MBuildArgs& a, F&& f) {
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t sz = i < a.n_buckets ? a.sizes[i] : 0;
  uint32_t ea, eb, ta, tb;
  block_scan2(sz, sz > 0, ea, eb, ta, tb);
  if (i >= a.n_buckets || sz == 0) return;
  uint64_t rank_incl = a.rank_base + a.base_a[blockIdx.x].nonzero + eb + 1;
  if (!kept_bucket(sz, rank_incl, a.sampling)) return;
  const uint64_t off = a.base_a[blockIdx.x].elems + ea;
  const uint64_t last = raw_key<RAW>(a.keys, off + sz - 1);
  uint64_t run = 0; bool first = true;
  for (uint32_t e = 0; e < sz; ++e) {
    uint64_t kv = raw_key<RAW>(a.keys, off + e);
    bool reach = (first || kv > run) && kv <= last;   // CuClarkDB.cu:1291-1307
    if (first || kv > run) { run = kv; first = false; }
    if (reach) f(kv * a.htsize + (a.bucket0 + i), a.labels[off + e]);
  }
}

template <typename RAW>
__global__ void __launch_bounds__(TILE) m_count_kernel(MBuildArgs a, uint32_t* __restrict__ cnt,
                                                       unsigned long long* __restrict__ kept) {
  unsigned long long mine = 0;
  for_reachable<RAW>(a, [&](uint64_t c, uint16_t) {
    atomicAdd(&cnt[mslot_of_kmer(c, a.k, a.m, (uint32_t)a.n_mslots)], 1u);
    ++mine;
  });
  if (mine) atomicAdd(kept, mine);
}

// A bucket of n entries is a fan-out-12 tree rooted in its main slot.  n <= 12: the root is a LEAF (entries inline).
// Otherwise level 0 = ceil(n/12) leaves, level l = ceil(level(l-1)/12) directories, up to a single root; all levels
// except the root live in the overflow area, highest level first, leaves last.  A directory's keys are the smallest
// key below each of its (contiguous) children; lookups descend one slot per level: 1 + ceil(log12(n/12)) slots.
struct MTree { uint32_t height; uint32_t cnt[8]; uint32_t off[8]; uint32_t total; };   // cnt/off per level (0 = leaves)

__device__ inline MTree m_tree(uint32_t n) {
  MTree t; t.height = 0; t.total = 0;
  for (int i = 0; i < 8; ++i) { t.cnt[i] = 0; t.off[i] = 0; }
  if (n <= MIC_MCAP) return t;
  uint32_t c = (n + MIC_MCAP - 1) / MIC_MCAP; int l = 0;
  while (c > 1) { t.cnt[l++] = c; c = (c + MIC_MCAP - 1) / MIC_MCAP; }
  t.height = (uint32_t)l;  // root sits at level `height` in the main slot
  uint32_t o = 0;
  for (int i = l - 1; i >= 0; --i) { t.off[i] = o; o += t.cnt[i]; }
  t.total = o;
  return t;
}

__device__ inline uint32_t chain_ovf(uint32_t n) { return m_tree(n).total; }

__global__ void __launch_bounds__(TILE) m_ovf_tile_kernel(const uint32_t* __restrict__ cnt, uint64_t n,
                                                          unsigned long long* __restrict__ tile_sum,
                                                          uint32_t* __restrict__ max_cnt) {
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t c = i < n ? cnt[i] : 0;
  uint32_t ea, eb, ta, tb;
  block_scan2(chain_ovf(c), 0, ea, eb, ta, tb);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = ta;
  if (c) atomicMax(max_cnt, c);
}

__device__ inline void m_write_empty(MSlot* sl, uint32_t meta, uint32_t child_base) {
  uint4* q = (uint4*)sl;
#pragma unroll
  for (int w = 0; w < 6; ++w) q[w] = make_uint4(~0u, ~0u, ~0u, ~0u);
  q[6] = make_uint4(child_base, 0, 0, 0);
  q[7] = make_uint4(0, 0, meta, 0);
}

// Headers of every slot of every bucket; keys = ~0.  ovf_first[s] = first overflow slot of bucket s.
__global__ void __launch_bounds__(TILE) m_header_kernel(const uint32_t* __restrict__ cnt, uint64_t n,
                                                        const unsigned long long* __restrict__ tile_base,
                                                        MSlot* __restrict__ slots, uint32_t* __restrict__ ovf_first) {
  uint64_t i = (uint64_t)blockIdx.x * TILE + threadIdx.x;
  uint32_t c = i < n ? cnt[i] : 0;
  const MTree t = m_tree(c);
  uint32_t ea, eb, ta, tb;
  block_scan2(t.total, 0, ea, eb, ta, tb);
  if (i >= n) return;
  const uint64_t base = n + tile_base[blockIdx.x] + ea;
  ovf_first[i] = (uint32_t)base;
  if (t.height == 0) { m_write_empty(&slots[i], c, 0); return; }
  // root (level `height`) has cnt[height-1] children starting at off[height-1]
  m_write_empty(&slots[i], t.cnt[t.height - 1] | MIC_M_DIR, (uint32_t)(base + t.off[t.height - 1]));
  for (uint32_t l = t.height - 1; l >= 1; --l) {          // inner directory levels
    for (uint32_t j = 0; j < t.cnt[l]; ++j) {
      const uint32_t kids = (t.cnt[l - 1] - j * MIC_MCAP) > MIC_MCAP ? MIC_MCAP : (t.cnt[l - 1] - j * MIC_MCAP);
      m_write_empty(&slots[base + t.off[l] + j], kids | MIC_M_DIR, (uint32_t)(base + t.off[l - 1] + j * MIC_MCAP));
    }
  }
  for (uint32_t j = 0; j < t.cnt[0]; ++j) {               // leaves
    const uint32_t here = (c - j * MIC_MCAP) > MIC_MCAP ? MIC_MCAP : (c - j * MIC_MCAP);
    m_write_empty(&slots[base + t.off[0] + j], here, 0);
  }
}

// element e of bucket s (n entries): in the main slot if n <= 12, else in leaf e/12
__device__ inline MSlot* m_elem_slot(MSlot* slots, uint64_t s, uint32_t n, uint32_t first_ovf, uint32_t e) {
  if (n <= MIC_MCAP) return &slots[s];
  return &slots[(uint64_t)first_ovf + (chain_ovf(n) - (n + MIC_MCAP - 1) / MIC_MCAP) + e / MIC_MCAP];
}

template <typename RAW>
__global__ void __launch_bounds__(TILE) m_scatter_kernel(MBuildArgs a, const uint32_t* __restrict__ cnt,
                                                         const uint32_t* __restrict__ ovf_first,
                                                         uint32_t* __restrict__ cursor, MSlot* __restrict__ slots) {
  for_reachable<RAW>(a, [&](uint64_t c, uint16_t lb) {
    uint32_t s = mslot_of_kmer(c, a.k, a.m, (uint32_t)a.n_mslots);
    uint32_t pos = atomicAdd(&cursor[s], 1u);
    MSlot* sl = m_elem_slot(slots, s, cnt[s], ovf_first[s], pos);
    sl->keys[pos % MIC_MCAP] = c;
    sl->labels[pos % MIC_MCAP] = lb;
  });
}

// ---- sorting the entries of every bucket, then the directory separators of the tree buckets ------------------------
// serial form: shell sort over the bucket's virtual array (entry e lives in leaf e / 12)
__device__ void m_sort_bucket_serial(MSlot* __restrict__ slots, uint64_t s, uint32_t c, uint32_t fo) {
  const uint32_t gaps[] = {701, 301, 132, 57, 23, 10, 4, 1};
  for (int g = 0; g < 8; ++g) {
    const uint32_t gap = gaps[g];
    if (gap >= c) continue;
    for (uint32_t i = gap; i < c; ++i) {
      MSlot* si = m_elem_slot(slots, s, c, fo, i);
      unsigned long long kv = si->keys[i % MIC_MCAP]; unsigned short lv = si->labels[i % MIC_MCAP];
      uint32_t j = i;
      while (j >= gap) {
        MSlot* sj = m_elem_slot(slots, s, c, fo, j - gap);
        unsigned long long kj = sj->keys[(j - gap) % MIC_MCAP];
        if (kj <= kv) break;
        MSlot* sd = m_elem_slot(slots, s, c, fo, j);
        sd->keys[j % MIC_MCAP] = kj; sd->labels[j % MIC_MCAP] = sj->labels[(j - gap) % MIC_MCAP];
        j -= gap;
      }
      MSlot* sd = m_elem_slot(slots, s, c, fo, j);
      sd->keys[j % MIC_MCAP] = kv; sd->labels[j % MIC_MCAP] = lv;
    }
  }
  const MTree t = m_tree(c);
  const uint64_t leaf0 = (uint64_t)fo + t.off[0];
  uint32_t span = 1;  // leaves below one child of a level-l directory = 12^(l-1)
  for (uint32_t l = 1; l <= t.height; ++l) {
    const uint32_t nodes = l == t.height ? 1 : t.cnt[l];
    for (uint32_t j = 0; j < nodes; ++j) {
      MSlot* dir = l == t.height ? &slots[s] : &slots[(uint64_t)fo + t.off[l] + j];
      for (uint32_t e = 0; e < MIC_MCAP; ++e) {
        const uint64_t child = (uint64_t)j * MIC_MCAP + e;          // index within level l-1
        if (child >= t.cnt[l - 1]) break;
        dir->keys[e] = slots[leaf0 + child * span].keys[0];
      }
    }
    span *= MIC_MCAP;
  }
}

// One lane per main slot for the buckets that fit it (sorting network in registers); the tree buckets of the wave's 64
// slots are then sorted one after the other by the whole wave: entries to LDS, rank of every entry by counting the
// smaller ones (LDS broadcast reads), scatter to the final position, separators from the sorted copy.  Buckets beyond
// MSORT_CAP entries fall back to the serial form on one lane.  (The serial form for every tree bucket took 4.2 s of
// the 5.7 s table build of the headline table: divergent waves waiting for their slowest lane.)
#define MSORT_CAP 256
__global__ void __launch_bounds__(256) m_sort_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ovf_first,
                                                     uint64_t n, MSlot* __restrict__ slots) {
  __shared__ unsigned long long s_key[4][MSORT_CAP];
  __shared__ unsigned long long s_sorted[4][MSORT_CAP];
  __shared__ unsigned short s_lab[4][MSORT_CAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = s < n ? cnt[s] : 0;
  const uint32_t fo_lane = (s < n && c > MIC_MCAP) ? ovf_first[s] : 0;
  if (c >= 2 && c <= MIC_MCAP) {
    // The common case, a bucket that fits its main slot: the 128 bytes are read once into registers, sorted by a
    // fixed 12-input network (39 compare-exchanges, every index static; unused keys are ~0 and sink to the end) and
    // written back once.  The in-memory shell sort below walked the slot with dependent loads and stores.
    uint4* q = (uint4*)&slots[s];
    uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6], w7 = q[7];
    unsigned long long k0 = w0.x | ((unsigned long long)w0.y << 32), k1 = w0.z | ((unsigned long long)w0.w << 32),
                       k2 = w1.x | ((unsigned long long)w1.y << 32), k3 = w1.z | ((unsigned long long)w1.w << 32),
                       k4 = w2.x | ((unsigned long long)w2.y << 32), k5 = w2.z | ((unsigned long long)w2.w << 32),
                       k6 = w3.x | ((unsigned long long)w3.y << 32), k7 = w3.z | ((unsigned long long)w3.w << 32),
                       k8 = w4.x | ((unsigned long long)w4.y << 32), k9 = w4.z | ((unsigned long long)w4.w << 32),
                       k10 = w5.x | ((unsigned long long)w5.y << 32), k11 = w5.z | ((unsigned long long)w5.w << 32);
    uint32_t l0 = w6.x & 0xFFFF, l1 = w6.x >> 16, l2 = w6.y & 0xFFFF, l3 = w6.y >> 16, l4 = w6.z & 0xFFFF, l5 = w6.z >> 16,
             l6 = w6.w & 0xFFFF, l7 = w6.w >> 16, l8 = w7.x & 0xFFFF, l9 = w7.x >> 16, l10 = w7.y & 0xFFFF, l11 = w7.y >> 16;
#define CE(a, b) { const bool sw_ = k##a > k##b; const unsigned long long tk_ = sw_ ? k##b : k##a; k##b = sw_ ? k##a : k##b; k##a = tk_; \
                   const uint32_t tl_ = sw_ ? l##b : l##a; l##b = sw_ ? l##a : l##b; l##a = tl_; }
    CE(0, 8) CE(1, 7) CE(2, 6) CE(3, 11) CE(4, 10) CE(5, 9)
    CE(0, 1) CE(2, 5) CE(3, 4) CE(6, 9) CE(7, 8) CE(10, 11)
    CE(0, 2) CE(1, 6) CE(5, 10) CE(9, 11)
    CE(0, 3) CE(1, 2) CE(4, 6) CE(5, 7) CE(8, 11) CE(9, 10)
    CE(1, 4) CE(3, 5) CE(6, 8) CE(7, 10)
    CE(1, 3) CE(2, 5) CE(6, 9) CE(8, 10)
    CE(2, 3) CE(4, 5) CE(6, 7) CE(8, 9)
    CE(4, 6) CE(5, 7)
    CE(3, 4) CE(5, 6) CE(7, 8)
#undef CE
#define LO(v) (uint32_t)(v)
#define HI(v) (uint32_t)((v) >> 32)
    q[0] = make_uint4(LO(k0), HI(k0), LO(k1), HI(k1)); q[1] = make_uint4(LO(k2), HI(k2), LO(k3), HI(k3));
    q[2] = make_uint4(LO(k4), HI(k4), LO(k5), HI(k5)); q[3] = make_uint4(LO(k6), HI(k6), LO(k7), HI(k7));
    q[4] = make_uint4(LO(k8), HI(k8), LO(k9), HI(k9)); q[5] = make_uint4(LO(k10), HI(k10), LO(k11), HI(k11));
    q[6] = make_uint4(l0 | (l1 << 16), l2 | (l3 << 16), l4 | (l5 << 16), l6 | (l7 << 16));
    q[7] = make_uint4(l8 | (l9 << 16), l10 | (l11 << 16), w7.z, w7.w);
#undef LO
#undef HI
  }
  unsigned long long* key = s_key[wv]; unsigned long long* sorted = s_sorted[wv]; unsigned short* lab = s_lab[wv];
  for (unsigned long long todo = __ballot(c > MIC_MCAP); todo; todo &= todo - 1) {
    const int b = __builtin_ctzll(todo);
    const uint32_t cb = __builtin_amdgcn_readlane(c, b), fo = __builtin_amdgcn_readlane(fo_lane, b);
    const uint64_t sb = s - lane + b;
    if (cb > MSORT_CAP) {
      if (lane == 0) m_sort_bucket_serial(slots, sb, cb, fo);
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = lane; e < cb; e += 64) {
      const MSlot* sl = m_elem_slot(slots, sb, cb, fo, e);
      key[e] = sl->keys[e % MIC_MCAP]; lab[e] = sl->labels[e % MIC_MCAP];
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt/lgkmcnt 0: the LDS writes above are visible to the wave
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = lane; e < cb; e += 64) {
      const unsigned long long ke = key[e];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cb; ++j) { const unsigned long long kj = key[j]; rank += (kj < ke || (kj == ke && j < e)) ? 1u : 0u; }
      MSlot* sd = m_elem_slot(slots, sb, cb, fo, rank);
      sd->keys[rank % MIC_MCAP] = ke; sd->labels[rank % MIC_MCAP] = lab[e];
      sorted[rank] = ke;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const MTree t = m_tree(cb);
    uint32_t span = 1;
    for (uint32_t l = 1; l <= t.height; ++l) {
      for (uint32_t child = lane; child < t.cnt[l - 1]; child += 64) {       // child = index within level l-1
        MSlot* dir = l == t.height ? &slots[sb] : &slots[(uint64_t)fo + t.off[l] + child / MIC_MCAP];
        dir->keys[child % MIC_MCAP] = sorted[(uint64_t)child * span * MIC_MCAP];   // first key of the child's first leaf
      }
      span *= MIC_MCAP;
    }
  }
}


// =====================================================================================================================
// Super-k-mer table (layout 3, mic_device.h): candidates -> per-slot staging -> merge into entries -> slots.
//   S1  per bucket: reachable k-mers -> candidates (oriented k-mer, minimizer position, x) -> count[slot(x)]++
//   S2  exclusive scan of the counts (hipcub) -> staging offsets; S3 scatter of (k-mer, position | label << 8)
//   S4  per slot, one thread: sort the staged candidates by (low 32 bits of x, x, position), merge the ones whose
//       nucleotides agree where they overlap (same x, same label) into entries; first run counts the entries (-> chain
//       slots, second scan), second run writes the slots.
// =====================================================================================================================
typedef unsigned __int128 u128;

template <typename RAW>
__global__ void __launch_bounds__(TILE) s_count_kernel(MBuildArgs a, uint32_t* __restrict__ cnt,
                                                       unsigned long long* __restrict__ kept) {
  unsigned long long mine = 0;
  for_reachable<RAW>(a, [&](uint64_t c, uint16_t) {
    auto add = [&](uint64_t, int, uint64_t x) { atomicAdd(&cnt[sslot_of_x(x, (uint32_t)a.n_mslots)], 1u); };
    if (a.fwd) s_candidates_fwd(c, a.k, a.m, add); else s_candidates(c, a.k, a.m, add);
    ++mine;
  });
  if (mine) atomicAdd(kept, mine);
}

template <typename RAW>
// Only the candidates of the slots [slot_lo, slot_hi) are staged (at their offset minus `base`): a table whose staging
// area does not fit next to it is built in several passes over slot ranges.
__global__ void __launch_bounds__(TILE) s_scatter_kernel(MBuildArgs a, const unsigned long long* __restrict__ off,
                                                         uint32_t* __restrict__ cursor, unsigned long long* __restrict__ cand_k,
                                                         uint32_t* __restrict__ cand_m, uint32_t slot_lo, uint32_t slot_hi,
                                                         unsigned long long base) {
  for_reachable<RAW>(a, [&](uint64_t c, uint16_t lb) {
    auto put = [&](uint64_t K, int j, uint64_t x) {
      const uint32_t s = sslot_of_x(x, (uint32_t)a.n_mslots);
      if (s < slot_lo || s >= slot_hi) return;
      const unsigned long long pos = off[s] - base + atomicAdd(&cursor[s], 1u);
      cand_k[pos] = K; cand_m[pos] = (uint32_t)j | ((uint32_t)lb << 8);
    };
    if (a.fwd) s_candidates_fwd(c, a.k, a.m, put); else s_candidates(c, a.k, a.m, put);
  });
}

// ---- the sorted build (round 5): candidates once, ordered by their slot hash, then counted and staged with LOCAL traffic ------------
// The classic form above walks the database three times with one thread per on-disk BUCKET (buckets hold 0 .. 255 k-mers: a
// wavefront runs as long as its fullest bucket), evaluates the sampling scheme of every k-mer each time (~400 vector operations),
// and ends every k-mer in a random atomic or a random write somewhere in gigabytes of counters / staging: count (twice: the table
// is sized from a first count), scatter.  Here:
//   s_expand_kernel   a tile's k-mers are spread over the block's threads through LDS (one k-mer per thread whatever the bucket
//                     sizes); every k-mer's candidates are computed ONCE: record = (h = slot hash of the minimizer, oriented k-mer,
//                     position | label), written at the k-mer's own index (coalesced); a HyperLogLog sketch of the minimizer values
//                     (LDS registers, merged at the end) sizes the table - it replaces the first counting pass of the classic form too;
//   radix sort        (hipcub, chunks of <= 2^30 records) by the top 16 bits of h: slot = floor(h n / 2^32) is monotone in h, so
//                     a chunk's records now sweep the slots in order, ~1/65536 of the table at a time;
//   s_rec_count / s_rec_place   the counting and the staging of the classic form over the RECORDS: a streaming read, atomics and
//                     writes that stay inside a window of counters / staging the caches hold.
// The staging arrays that come out are the classic form's (the same candidates per slot, in another order: s_merge_kernel sorts
// them anyway).  One-strand tables that fit the memory below take this road; everything else the classic one.
// a record = a 64-bit sort key ((minimizer position | label << 8) << 32 | slot hash; ~0: no record) and the oriented k-mer.
// The hash sits in the LOW word and the sort takes bits [16, 32): rocPRIM 4.2 (ROCm 7.2) tears pairs apart and leaves them unsorted
// when a 64-bit key is sorted by a bit range that starts at bit 32 or above and there are fewer than ~2^24 pairs
// (tools/radix_sort_probe.hip, profiles/r05_radix_sort_probe.log); ranges inside the low word sort correctly at every size.
typedef unsigned long long SKey;
typedef unsigned long long SVal;
#define S_NOREC (~0ull)
#define S_HLL_BITS 12
#define S_XCAP 2048                                   // k-mers of a tile staged per round

template <typename RAW, bool EMIT>
__global__ void __launch_bounds__(TILE) s_expand_kernel(MBuildArgs a, uint32_t tile_lo, uint32_t tile_hi, unsigned long long elem0,
                                                        SKey* __restrict__ out_h, SVal* __restrict__ out_v,
                                                        SKey* __restrict__ ex_h, SVal* __restrict__ ex_v, uint32_t ex_cap,
                                                        unsigned long long* __restrict__ counters,      // [0] kept k-mers, [1] extra records, [2] extras dropped
                                                        uint32_t* __restrict__ hll) {
  __shared__ unsigned long long s_c[S_XCAP];
  __shared__ unsigned short s_lb[S_XCAP];
  __shared__ uint32_t s_hll[1 << S_HLL_BITS];
  const int tid = threadIdx.x;
  for (int i = tid; i < (1 << S_HLL_BITS); i += TILE) s_hll[i] = 0;
  unsigned long long kept_mine = 0;
  for (uint32_t tile = tile_lo + blockIdx.x; tile < tile_hi; tile += gridDim.x) {
    const uint64_t i = (uint64_t)tile * TILE + tid;
    const uint32_t sz = i < a.n_buckets ? a.sizes[i] : 0;
    uint32_t ea, eb, ta, tb;
    block_scan2(sz, sz > 0, ea, eb, ta, tb);
    const bool keep = sz > 0 && kept_bucket(sz, a.rank_base + a.base_a[tile].nonzero + eb + 1, a.sampling);
    const uint64_t off = a.base_a[tile].elems + ea;
    const uint64_t last = keep ? raw_key<RAW>(a.keys, off + sz - 1) : 0;
    uint32_t e = 0; uint64_t run = 0; bool first = true;
    for (uint32_t base = 0; base < ta; base += S_XCAP) {
      // the bucket's thread puts its k-mers of this round into LDS: the canonical k-mer, or ~0 for what the reference's scan cannot
      // reach (CuClarkDB.cu:1291-1307) and for buckets the sampling drops
      while (e < sz && ea + e < base + S_XCAP) {
        unsigned long long cv = ~0ull;
        if (keep) {
          const uint64_t kv = raw_key<RAW>(a.keys, off + e);
          const bool reach = (first || kv > run) && kv <= last;
          if (first || kv > run) { run = kv; first = false; }
          if (reach) cv = kv * a.htsize + (a.bucket0 + i);
        }
        s_c[ea + e - base] = cv;
        s_lb[ea + e - base] = keep ? a.labels[off + e] : (unsigned short)0;
        ++e;
      }
      __syncthreads();
      const uint32_t n_here = ta - base < S_XCAP ? ta - base : S_XCAP;
      for (uint32_t j = tid; j < n_here; j += TILE) {
        const unsigned long long c = s_c[j];
        SKey h0 = S_NOREC; SVal r0 = 0;
        if (c != ~0ull) {
          ++kept_mine;
          const uint32_t lb = s_lb[j];
          int n_c = 0;
          auto put = [&](uint64_t K, int p, uint64_t x) {
            // the sketch's own hash of x: two multiply-xorshift rounds (register index = its top bits, rank = the bits below: from ONE
            // product both are functions of the same carries, and structured sets - the m-mers of tandem repeats - skew the estimate)
            unsigned long long g = x * 0x9E3779B97F4A7C15ull;
            g ^= g >> 32; g *= 0xD6E8FEB86659FD93ull; g ^= g >> 32;
            const uint32_t rank = (uint32_t)__builtin_clzll((g << S_HLL_BITS) | (1ull << (S_HLL_BITS - 1))) + 1u;
            atomicMax(&s_hll[(uint32_t)(g >> (64 - S_HLL_BITS))], rank);
            if (!EMIT) return;
            const uint32_t h = sslot_hash(x);
            const uint32_t cm = (uint32_t)p | (lb << 8);
            const SKey key = ((SKey)cm << 32) | h;
            if (n_c++ == 0) { h0 = key; r0 = K; }
            else {
              // a k-mer stored under several positions (tied t-mers: low complexity; a palindromic m-mer): the extras' own list
              const unsigned long long at = atomicAdd(&counters[1], 1ull);
              if (at < ex_cap) { ex_h[at] = key; ex_v[at] = K; }
              else atomicAdd(&counters[2], 1ull);
            }
          };
          if (a.fwd) s_candidates_fwd(c, a.k, a.m, put); else s_candidates(c, a.k, a.m, put);
        }
        if (EMIT) {
          const unsigned long long at = a.base_a[tile].elems + base + j - elem0;
          out_h[at] = h0; out_v[at] = r0;
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = tid; i < (1 << S_HLL_BITS); i += TILE) if (s_hll[i]) atomicMax(&hll[i], s_hll[i]);
  for (int o = 32; o > 0; o >>= 1) kept_mine += __shfl_down(kept_mine, o);
  if ((tid & 63) == 0 && kept_mine) atomicAdd(&counters[0], kept_mine);
}

// counts / stages the records of the slots [slot_lo, slot_hi) (see s_count_kernel / s_scatter_kernel); grid-stride: a launch
// holds fewer than 2^32 threads, a database more records
__global__ void __launch_bounds__(256) s_rec_count_kernel(const SKey* __restrict__ h, uint64_t n, uint32_t n_slots,
                                                          uint32_t slot_lo, uint32_t slot_hi, uint32_t* __restrict__ cnt) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const SKey key = h[i];
    if (key == S_NOREC) continue;
    const uint32_t s = __umulhi((uint32_t)key, n_slots);
    if (s >= slot_lo && s < slot_hi) atomicAdd(&cnt[s], 1u);
  }
}
__global__ void __launch_bounds__(256) s_rec_place_kernel(const SKey* __restrict__ h, const SVal* __restrict__ v, uint64_t n, uint32_t n_slots,
                                                          const unsigned long long* __restrict__ off, uint32_t* __restrict__ cursor,
                                                          unsigned long long* __restrict__ cand_k, uint32_t* __restrict__ cand_m,
                                                          uint32_t slot_lo, uint32_t slot_hi, unsigned long long base) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const SKey key = h[i];
    if (key == S_NOREC) continue;
    const uint32_t s = __umulhi((uint32_t)key, n_slots);
    if (s < slot_lo || s >= slot_hi) continue;
    const unsigned long long pos = off[s] - base + atomicAdd(&cursor[s], 1u);
    cand_k[pos] = v[i]; cand_m[pos] = (uint32_t)(key >> 32);
  }
}

static unsigned rec_grid(uint64_t n) { const uint64_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : b > (1u << 20) ? (1u << 20) : b); }

// distinct values seen by a HyperLogLog sketch of 2^S_HLL_BITS registers (Flajolet et al. 2007, with the small-range correction)
static double hll_estimate(const uint32_t* reg) {
  const int mreg = 1 << S_HLL_BITS;
  double sum = 0; int zeros = 0;
  for (int j = 0; j < mreg; ++j) { sum += ldexp(1.0, -(int)reg[j]); zeros += reg[j] == 0; }
  const double alpha = 0.7213 / (1.0 + 1.079 / mreg);
  double E = alpha * mreg * (double)mreg / sum;
  if (E <= 2.5 * mreg && zeros) E = mreg * log((double)mreg / zeros);
  return E;
}

__global__ void s_nonzero_kernel(const uint32_t* __restrict__ cnt, uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long mine = 0;     // grid-stride: one atomic per wave of a small grid, not one per 64 slots
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mine += cnt[i] != 0;
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

__global__ void s_sum_kernel(const uint32_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mine += v[i];
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

__global__ void s_max_kernel(const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mine = v[i] > mine ? v[i] : mine;
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(mine, off); mine = o > mine ? o : mine; }
  if ((threadIdx.x & 63) == 0 && mine) atomicMax(out, mine);
}

// How far lookups have to walk: sum over the slots of (candidates staged for the slot) x (continuation slots behind it).
// Divided by the number of candidates this is the mean number of EXTRA slots a stored k-mer sits behind - the cost of
// crowded minimizers (tandem repeats, low complexity: one minimizer value in thousands of contexts).
__global__ void s_walk_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ n_ent, uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t e = n_ent[i];
    if (e > MIC_S_CAP) mine += (unsigned long long)cnt[i] * ((e - 1) / MIC_S_CAP);
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

struct SOpen { u128 S, known; uint32_t pmask, label; };

__device__ inline uint64_t s_x_of(unsigned long long K, uint32_t j, int k, int m) {
  return (K >> (2 * (k - m - (int)j))) & ((1ULL << (2 * m)) - 1);
}
// order of the staged candidates: (low 32 bits of x, high bits of x, minimizer position)
__device__ inline bool s_less(unsigned long long Ka, uint32_t ma, unsigned long long Kb, uint32_t mb, int k, int m) {
  const uint64_t xa = s_x_of(Ka, ma & 0xFF, k, m), xb = s_x_of(Kb, mb & 0xFF, k, m);
  if ((uint32_t)xa != (uint32_t)xb) return (uint32_t)xa < (uint32_t)xb;
  if (xa != xb) return xa < xb;
  return (ma & 0xFF) < (mb & 0xFF);
}

struct SWriter {   // where the entries of one slot go: the main slot, then its contiguous continuation slots
  uint32_t* slots; uint64_t main, chain; uint32_t n_total, n_out;
  uint64_t cur, n_slots; uint32_t n_here, pool_cap; uint32_t* pool; bool full;     // one-pass form
  // The MAIN slot's 32 words are put together in REGISTERS (round 6; every index below is a constant after unrolling) and written by
  // their thread as eight 16-byte stores.  Until round 5 they were a row of the block's LDS (written out coalesced by the whole
  // block): 16.9 of the block's 41 KB - the kernel runs a long chain of dependent instructions per wavefront (28 000 of them), and
  // LDS capacity held it at 1.5 wavefronts per SIMD; without the rows twice as many are resident.
  uint32_t row[32];
  // (spelled out, no loops: the array is promoted to registers only if every index is a constant when the promotion pass first runs)
  __device__ __forceinline__ void row_init() {
    row[0] = row[1] = row[2] = row[3] = row[4] = row[5] = 0xFFFFFFFFu;
    row[6] = row[7] = row[8] = row[9] = row[10] = row[11] = row[12] = row[13] = row[14] = row[15] = row[16] = row[17] = row[18] = 0;
    row[19] = row[20] = row[21] = row[22] = row[23] = row[24] = row[25] = row[26] = row[27] = row[28] = row[29] = row[30] = row[31] = 0;
  }
  // (selects, not branches: a chain of `if (e == 0) row[0] = .. else if (e == 1) row[1] = ..` is folded back into row[e] = .. - a dynamic index,
  // and the array goes to scratch memory)
#define S_ROW_PUT(ee) { const bool h_ = e == ee; row[ee] = h_ ? key : row[ee]; row[6 + 3 * ee] = h_ ? a : row[6 + 3 * ee]; row[7 + 3 * ee] = h_ ? b : row[7 + 3 * ee]; \
                        row[8 + 3 * ee] = h_ ? c : row[8 + 3 * ee]; row[24 + ee] = h_ ? pl : row[24 + ee]; }
  __device__ __forceinline__ void put_main(uint32_t e, uint32_t key, uint32_t a, uint32_t b, uint32_t c, uint32_t pl) {
    static_assert(MIC_S_CAP == 6, "one S_ROW_PUT per entry of a slot");
    S_ROW_PUT(0) S_ROW_PUT(1) S_ROW_PUT(2) S_ROW_PUT(3) S_ROW_PUT(4) S_ROW_PUT(5)
  }
#undef S_ROW_PUT
  // entry e of slot `slot` (the main slot: registers; a continuation slot: the table)
  __device__ __forceinline__ void put(uint64_t slot, uint32_t e, uint32_t key, uint32_t a, uint32_t b, uint32_t c, uint32_t pl) {
    if (slot == main) put_main(e, key, a, b, c, pl);
    else { uint32_t* q = slots + slot * 32; q[e] = key; q[6 + 3 * e] = a; q[7 + 3 * e] = b; q[8 + 3 * e] = c; q[24 + e] = pl; }
  }
  __device__ __forceinline__ void set_hdr(uint64_t slot, uint32_t w30, uint32_t w31) {
    if (slot == main) { row[30] = w30; row[31] = w31; } else { uint32_t* q = slots + slot * 32; q[30] = w30; q[31] = w31; }
  }
};

// One-pass form: the number of entries of a bucket is not known when its first entry is written, so continuation slots
// are taken one at a time from a pool behind the main slots (an atomic counter) and linked through word 31; a bucket's
// chain is then not contiguous, which no reader assumes.  A full pool is reported (pool[1]) and the caller builds the
// table the two-pass way.
__device__ inline void s_slot_init(uint32_t* q) {
  for (int e = 0; e < 6; ++e) q[e] = 0xFFFFFFFFu;
  for (int e = 6; e < 32; ++e) q[e] = 0;
}
// MODE 0: count the entries, 1: write them (entry counts and chain bases known), 2: write them in one pass (pool)
template <int MODE>
__device__ __forceinline__ void s_emit(const SOpen& o, uint64_t x, int L, SWriter& wr) {
  const u128 v = o.S << (96 - 2 * L);
  const uint32_t v0 = (uint32_t)(v >> 64), v1 = (uint32_t)(v >> 32), v2 = (uint32_t)v, pl = (o.pmask << 16) | o.label;
  if (MODE == 2) {
    if (wr.n_here == MIC_S_CAP && !wr.full) {
      const uint32_t idx = atomicAdd(wr.pool, 1u);
      if (idx >= wr.pool_cap) { wr.full = true; atomicMax(wr.pool + 1, 1u); }
      else {
        const uint64_t nxt = wr.n_slots + idx;
        wr.set_hdr(wr.cur, MIC_S_CAP | MIC_S_NEXT, (uint32_t)nxt);
        s_slot_init(wr.slots + nxt * 32);
        wr.cur = nxt; wr.n_here = 0;
      }
    }
    if (!wr.full) wr.put(wr.cur, wr.n_here++, (uint32_t)x, v0, v1, v2, pl);
  }
  if (MODE == 1) {
    const uint32_t idx = wr.n_out, e = idx % MIC_S_CAP;
    if (idx < MIC_S_CAP) wr.put_main(e, (uint32_t)x, v0, v1, v2, pl);
    else { uint32_t* q = wr.slots + (wr.chain + (idx - MIC_S_CAP) / MIC_S_CAP) * 32; q[e] = (uint32_t)x; q[6 + 3 * e] = v0; q[7 + 3 * e] = v1; q[8 + 3 * e] = v2; q[24 + e] = pl; }
  }
  ++wr.n_out;
}

#define S_MAXOPEN 4
// One thread per slot: sort the slot's staged candidates, merge them into entries, put the slot together.  The block works
// out of LDS (round 4): the candidates of its S_TPB consecutive slots are one contiguous piece of the staging area - loaded
// cooperatively (coalesced), sorted and merged by their threads at LDS latency instead of one dependent global access per
// comparison.  The main slot is put together in its thread's registers and written as eight 16-byte stores (round 6: SWriter).
// A block whose slots hold more candidates than the LDS piece (crowded minimizers) works on the staging area directly, as before;
// continuation slots are written directly (rare).
#define S_TPB 128
#define S_LDS_CAND 2048
template <int MODE>
__global__ void __launch_bounds__(S_TPB) s_merge_kernel(const unsigned long long* __restrict__ off, const uint32_t* __restrict__ cnt,
                                                        uint64_t n_slots, unsigned long long* __restrict__ cand_k,
                                                        uint32_t* __restrict__ cand_m, int k, int m,
                                                        uint32_t* __restrict__ n_ent, const unsigned long long* __restrict__ chain_off,
                                                        uint32_t* __restrict__ slots, uint32_t* __restrict__ max_ent,
                                                        uint64_t slot_lo, uint64_t slot_hi, unsigned long long base, int sort_now,
                                                        uint32_t* __restrict__ pool, uint32_t pool_cap) {
  constexpr bool WRITE = MODE == 1;
  __shared__ unsigned long long s_k[S_LDS_CAND];
  __shared__ uint32_t s_m[S_LDS_CAND];
  const uint64_t s_first = slot_lo + (uint64_t)blockIdx.x * S_TPB;
  const uint64_t s_end = s_first + S_TPB < slot_hi ? s_first + S_TPB : slot_hi;     // (the grid covers [slot_lo, slot_hi): s_first < slot_hi)
  const uint64_t s = s_first + threadIdx.x;
  const bool live = s < slot_hi;
  const unsigned long long c0 = off[s_first] - base, c1 = off[s_end - 1] + cnt[s_end - 1] - base;
  const bool in_lds = c1 - c0 <= S_LDS_CAND;
  // Round 5: a candidate staged in LDS is re-packed by the thread that loads it (the loads are spread evenly over the block) into ITS
  // SORT KEY - minimizer value as the slot orders it (low 32 bits, then the bits above), minimizer position, label: 60 bits - and the
  // 2 (k - m) bits of the k-mer around the minimizer: the same 12 bytes, but the shell sort below compares two 64-bit words where
  // it extracted two minimizers out of two k-mers per comparison (~55 % of the kernel's instructions), and the merge loop reads
  // value, position and label off the key.  (m <= 20: 40 bits of minimizer; other m and blocks that do not fit LDS keep the old form.)
  const bool packed = in_lds && m <= 20;
  if (in_lds) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)(c1 - c0); i += S_TPB) {
      const unsigned long long Kv = cand_k[c0 + i]; const uint32_t Mv = cand_m[c0 + i];
      if (packed) {
        const uint32_t j = Mv & 0xFF, lb = Mv >> 8;
        const int r = 2 * (k - m - (int)j);
        const unsigned long long x = (Kv >> r) & ((1ULL << (2 * m)) - 1);
        const unsigned long long left = r + 2 * m >= 64 ? 0ULL : Kv >> (r + 2 * m);        // (k = 32, position 0: nothing to the left)
        const unsigned long long F = (left << r) | (Kv & ((1ULL << r) - 1));
        s_k[i] = ((unsigned long long)(uint32_t)x << 28) | ((x >> 32) << 20) | ((unsigned long long)j << 16) | lb;
        s_m[i] = (uint32_t)F;
      } else { s_k[i] = Kv; s_m[i] = Mv; }
    }
    __syncthreads();
  }
  auto unpack = [&](unsigned long long key, uint32_t F, unsigned long long& kv, uint32_t& j, uint32_t& lb, uint64_t& x) {
    x = (key >> 28) | (((key >> 20) & 0xFF) << 32); j = (uint32_t)(key >> 16) & 15u; lb = (uint32_t)key & 0xFFFFu;
    const int r = 2 * (k - m - (int)j);
    const unsigned long long left = (unsigned long long)F >> r;
    kv = (r + 2 * m >= 64 ? 0ULL : left << (r + 2 * m)) | (x << r) | ((unsigned long long)F & ((1ULL << r) - 1));
  };
  const uint32_t n = live ? cnt[s] : 0;
  const unsigned long long my = live ? off[s] - base : c0;
  const int w = k - m + 1, L = k + w - 1;
  SWriter wr; wr.slots = slots; wr.main = s; wr.chain = 0; wr.n_total = 0; wr.n_out = 0;
  wr.cur = s; wr.n_slots = n_slots; wr.n_here = 0; wr.pool_cap = pool_cap; wr.pool = pool; wr.full = false;
  wr.row_init();
  if (WRITE && live) {
    wr.n_total = n_ent[s];
    const uint32_t n_chain = wr.n_total > MIC_S_CAP ? (wr.n_total - 1) / MIC_S_CAP : 0;
    wr.chain = n_slots + chain_off[s];
    for (uint32_t c = 0; c <= n_chain; ++c) {               // headers and empty entries of every slot of this bucket
      const uint32_t here = wr.n_total - c * MIC_S_CAP > MIC_S_CAP ? MIC_S_CAP : wr.n_total - c * MIC_S_CAP;
      const uint32_t w30 = (wr.n_total ? here : 0) | (c < n_chain ? MIC_S_NEXT : 0), w31 = c < n_chain ? (uint32_t)(wr.chain + c) : 0;
      if (c == 0) { wr.row[30] = w30; wr.row[31] = w31; }
      else { uint32_t* q = slots + (wr.chain + c - 1) * 32; s_slot_init(q); q[30] = w30; q[31] = w31; }
    }
  }
  // Sort and merge, instantiated TWICE - on the block's LDS piece and on the staging area in HBM (blocks that do not fit) - so that the
  // LDS road is ds_read / ds_write: through ONE pointer that may be either, every access was a FLAT instruction (1 100 per wavefront,
  // each through the texture path with its aperture check: several times an LDS round trip in a chain of dependent accesses) - round 6.
  auto sort_and_merge = [&](auto* K, auto* M) __attribute__((always_inline)) {
  if (sort_now) {
    // shell sort of the staged candidates (in place; a second run over the same staging area finds them sorted)
    constexpr uint32_t gaps[8] = {701, 301, 132, 57, 23, 10, 4, 1};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const uint32_t gap = gaps[g];
      if (gap >= n) continue;
      for (uint32_t i = gap; i < n; ++i) {
        const unsigned long long kv = K[i]; const uint32_t mv = M[i];
        uint32_t j = i;
        while (j >= gap && (packed ? kv < K[j - gap] : s_less(kv, mv, K[j - gap], M[j - gap], k, m))) { K[j] = K[j - gap]; M[j] = M[j - gap]; j -= gap; }
        K[j] = kv; M[j] = mv;
      }
    }
  }
  // the open contexts live in registers: every index below is a constant after unrolling (a dynamically indexed array goes to
  // scratch memory - 208 bytes per thread and a memory round trip per access before round 4)
  SOpen open[S_MAXOPEN]; int n_open = 0; uint64_t cur_x = 0;
#pragma unroll
  for (int o = 0; o < S_MAXOPEN; ++o) { open[o].S = 0; open[o].known = 0; open[o].pmask = 0; open[o].label = 0; }
  const u128 kmask = (((u128)1) << (2 * k)) - 1;
  for (uint32_t i = 0; i < n; ++i) {
    unsigned long long kv = K[i]; const uint32_t mv = M[i];
    uint32_t j, lb; uint64_t x;
    if (packed) unpack(kv, mv, kv, j, lb, x);
    else { j = mv & 0xFF; lb = mv >> 8; x = s_x_of(kv, j, k, m); }
    if (n_open && x != cur_x) {
#pragma unroll
      for (int o = 0; o < S_MAXOPEN; ++o) if (o < n_open) s_emit<MODE>(open[o], cur_x, L, wr);
      n_open = 0;
    }
    cur_x = x;
    const u128 SK = (u128)kv << (2 * j), MK = kmask << (2 * j);
    bool done = false;
#pragma unroll
    for (int o = 0; o < S_MAXOPEN; ++o) {
      if (o < n_open && !done && open[o].label == lb && !((open[o].pmask >> j) & 1) && ((open[o].S ^ SK) & open[o].known & MK) == 0) {
        open[o].S |= SK; open[o].known |= MK; open[o].pmask |= 1u << j; done = true;
      }
    }
    if (!done) {
      if (n_open == S_MAXOPEN) {                              // keep the most recent contexts open
        s_emit<MODE>(open[0], cur_x, L, wr);
#pragma unroll
        for (int o = 1; o < S_MAXOPEN; ++o) open[o - 1] = open[o];
        --n_open;
      }
#pragma unroll
      for (int o = 0; o < S_MAXOPEN; ++o)
        if (o == n_open) { open[o].S = SK; open[o].known = MK; open[o].pmask = 1u << j; open[o].label = lb; }
      ++n_open;
    }
  }
#pragma unroll
  for (int o = 0; o < S_MAXOPEN; ++o) if (o < n_open) s_emit<MODE>(open[o], cur_x, L, wr);
  };
  if (in_lds) sort_and_merge(&s_k[my - c0], &s_m[my - c0]);
  else sort_and_merge(cand_k + my, cand_m + my);
  if (MODE != 1 && live) { n_ent[s] = wr.n_out; if (wr.n_out > MIC_S_CAP) atomicMax(max_ent, wr.n_out); }
  if (MODE == 2 && live && !wr.full) wr.set_hdr(wr.cur, wr.n_here, 0u);
  __syncthreads();
  // the counting pass of the two-pass form leaves the candidates sorted for the writing pass (one range: it does not sort again)
  if (MODE == 0 && sort_now && in_lds)
    for (uint32_t i = threadIdx.x; i < (uint32_t)(c1 - c0); i += S_TPB) {
      unsigned long long kv = s_k[i]; uint32_t mv = s_m[i];
      if (packed) { uint32_t j, lb; uint64_t x; unpack(kv, mv, kv, j, lb, x); mv = j | (lb << 8); }
      cand_k[c0 + i] = kv; cand_m[c0 + i] = mv;
    }
  // the main slot: 128 bytes from its thread's registers
  if (MODE != 0 && live) {
    uint4* out = (uint4*)(slots + s * 32);
#define S_ROW_OUT(i) out[i] = make_uint4(wr.row[4 * i], wr.row[4 * i + 1], wr.row[4 * i + 2], wr.row[4 * i + 3]);
    S_ROW_OUT(0) S_ROW_OUT(1) S_ROW_OUT(2) S_ROW_OUT(3) S_ROW_OUT(4) S_ROW_OUT(5) S_ROW_OUT(6) S_ROW_OUT(7)
#undef S_ROW_OUT
  }
}

__global__ void s_chain_demand_kernel(const uint32_t* __restrict__ n_ent, uint64_t n, uint32_t* __restrict__ demand) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t e = n_ent[i]; demand[i] = e > MIC_S_CAP ? (e - 1) / MIC_S_CAP : 0; }
}

// ---- crowded minimizers: a side table keyed by the whole k-mer ------------------------------------------------------------
// A minimizer value that sits in thousands of contexts (the m-mers of a microsatellite: every flank of every occurrence is a
// context of its own) gives a chain of thousands of entries under ONE sort key, which a lookup can only walk
// slot by slot.  After the table is built, the groups of more than MIC_S_CROWD entries with the same minimizer value are
// taken OUT of their chains: their k-mers go into an open-addressing hash table keyed by the oriented k-mer itself (16-byte
// cells {k-mer lo, hi, label + 1, 0}, at most half full), and ONE marker entry (presence mask 0) stays behind.  A query run
// that meets the marker looks its k-mers up one by one in the side table (query_kernel_r<.., SIDE>: a wave-uniform rare
// path); every other run of the table is as fast as before.  Exact: a k-mer is in exactly one of the two places.
#ifndef MIC_S_CROWD
#define MIC_S_CROWD 12
#endif

struct SChain {      // entries of a slot's chain, in order
  const uint32_t* slots; uint64_t slot; uint32_t e, n_here; bool valid;
  __device__ SChain(const uint32_t* sl, uint64_t s) : slots(sl), slot(s), e(0) { n_here = sl[s * 32 + 30] & 0xFFu; valid = true; settle(); }
  __device__ void settle() {
    while (valid && e >= n_here) {
      const uint32_t* q = slots + slot * 32;
      if (q[30] & MIC_S_NEXT) { slot = q[31]; n_here = slots[slot * 32 + 30] & 0xFFu; e = 0; } else valid = false;
    }
  }
  __device__ void next() { ++e; settle(); }
  __device__ const uint32_t* q() const { return slots + slot * 32; }
};

__device__ inline uint64_t s_entry_x(const uint32_t* q, uint32_t e, int k, int m) {
  const u128 S = ((u128)q[6 + 3 * e] << 64) | ((u128)q[7 + 3 * e] << 32) | q[8 + 3 * e];
  const int ctx = k - m;
  return (uint64_t)(S >> (96 - 2 * (ctx + m))) & ((1ULL << (2 * m)) - 1);
}

// out[0] += k-mers of crowded groups, out[1] += crowded groups
__global__ void __launch_bounds__(256) s_crowd_count_kernel(const uint32_t* __restrict__ slots, uint64_t slot_lo, uint64_t slot_hi, int k, int m,
                                                            unsigned long long* __restrict__ out) {
  const uint64_t s = slot_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slot_hi) return;
  if ((slots[s * 32 + 30] & 0xFFu) < MIC_S_CAP || !(slots[s * 32 + 30] & MIC_S_NEXT)) return;   // a chain of one slot holds no crowded group
  SChain c(slots, s);
  while (c.valid) {
    const uint64_t x = s_entry_x(c.q(), c.e, k, m);
    uint32_t n = 0, kmers = 0;
    while (c.valid && s_entry_x(c.q(), c.e, k, m) == x) { ++n; kmers += __popc(c.q()[24 + c.e] >> 16); c.next(); }
    if (n > MIC_S_CROWD) { atomicAdd(&out[0], (unsigned long long)kmers); atomicAdd(&out[1], 1ull); }
  }
}

__device__ inline void s_side_insert(uint4* __restrict__ side, uint32_t mask, uint64_t K, uint32_t label1) {
  uint32_t h = s_side_hash(K, mask);
  for (;;) {
    uint32_t* cell = (uint32_t*)(side + h);
    if (atomicCAS(&cell[2], 0u, label1) == 0u) { cell[0] = (uint32_t)K; cell[1] = (uint32_t)(K >> 32); return; }
    h = (h + 1) & mask;
  }
}

// moves the crowded groups of every chain into the side table and compacts the chain in place (one thread per main slot)
__global__ void __launch_bounds__(256) s_crowd_move_kernel(uint32_t* __restrict__ slots, uint64_t slot_lo, uint64_t slot_hi, int k, int m,
                                                           uint4* __restrict__ side, uint32_t side_mask, uint32_t* __restrict__ n_ent,
                                                           uint32_t* __restrict__ max_ent) {
  const uint64_t s = slot_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slot_hi) return;
  if ((slots[s * 32 + 30] & 0xFFu) < MIC_S_CAP || !(slots[s * 32 + 30] & MIC_S_NEXT)) return;
  {  // anything to do here?
    SChain c(slots, s);
    bool any = false;
    while (c.valid && !any) {
      const uint64_t x = s_entry_x(c.q(), c.e, k, m);
      uint32_t n = 0;
      while (c.valid && s_entry_x(c.q(), c.e, k, m) == x) { ++n; c.next(); }
      any = n > MIC_S_CROWD;
    }
    if (!any) return;
  }
  const int w = k - m + 1, ctx = k - m;
  SChain rd(slots, s);
  uint64_t w_slot = s; uint32_t w_e = 0, total = 0;      // write position: never ahead of the read position
  auto put = [&](uint32_t key, uint32_t S0, uint32_t S1, uint32_t S2, uint32_t pl) {
    if (w_e == MIC_S_CAP) { w_slot = slots[w_slot * 32 + 31]; w_e = 0; }       // the chain's own next slot (it had more entries before)
    uint32_t* q = slots + w_slot * 32;
    q[w_e] = key; q[6 + 3 * w_e] = S0; q[7 + 3 * w_e] = S1; q[8 + 3 * w_e] = S2; q[24 + w_e] = pl;
    ++w_e; ++total;
  };
  while (rd.valid) {
    const uint64_t x = s_entry_x(rd.q(), rd.e, k, m);
    uint32_t n = 0;
    { SChain la = rd; while (la.valid && s_entry_x(la.q(), la.e, k, m) == x) { ++n; la.next(); } }
    if (n > MIC_S_CROWD) {
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t* q = rd.q();
        const uint32_t S0 = q[6 + 3 * rd.e], S1 = q[7 + 3 * rd.e], S2 = q[8 + 3 * rd.e], pl = q[24 + rd.e];
        for (int j = 0; j < w; ++j)
          if ((pl >> (16 + j)) & 1u) s_side_insert(side, side_mask, s_extract(S0, S1, S2, w - 1 - j, k), (pl & 0xFFFFu) + 1u);
        rd.next();
      }
      const u128 v = (u128)x << (96 - 2 * (ctx + m));                       // the marker: the minimizer alone, presence mask 0
      put((uint32_t)x, (uint32_t)(v >> 64), (uint32_t)(v >> 32), (uint32_t)v, 0xFFFFu);
    } else {
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t* q = rd.q();
        const uint32_t key = q[rd.e], S0 = q[6 + 3 * rd.e], S1 = q[7 + 3 * rd.e], S2 = q[8 + 3 * rd.e], pl = q[24 + rd.e];
        rd.next();                                                          // (read before the write below may overwrite it)
        put(key, S0, S1, S2, pl);
      }
    }
  }
  // close the chain behind the last entry written: counts, unused keys, the continuation flag
  {
    uint64_t sl = s; uint32_t left = total;
    for (;;) {
      uint32_t* q = slots + sl * 32;
      const uint32_t here = left > MIC_S_CAP ? MIC_S_CAP : left;
      left -= here;
      for (uint32_t e = here; e < MIC_S_CAP; ++e) { q[e] = 0xFFFFFFFFu; q[6 + 3 * e] = q[7 + 3 * e] = q[8 + 3 * e] = 0; q[24 + e] = 0; }
      const uint32_t nxt = q[31];
      if (left) { q[30] = here | MIC_S_NEXT; sl = nxt; } else { q[30] = here; q[31] = 0; break; }
    }
  }
  n_ent[s] = total;
  atomicMax(max_ent, total);
}

}  // namespace

int mic_build_mtable(const uint8_t* d_sizes, uint64_t n_buckets, uint64_t bucket0, uint64_t htsize, const void* d_keys,
                     int key_bytes, const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int k, int m,
                     hipStream_t s, MicBuildOut* out, char* err, size_t err_cap) {
  int rc = 0;
  const unsigned n_tiles = (unsigned)((n_buckets + TILE - 1) / TILE);
  TileA* d_a = nullptr; uint32_t* d_cnt = nullptr; uint32_t* d_cur = nullptr; unsigned long long* d_tile = nullptr;
  unsigned long long* d_kept = nullptr; uint32_t* d_max = nullptr; MSlot* slots = nullptr; uint32_t* d_of = nullptr;
  std::vector<TileA> h_a(n_tiles);
  std::vector<unsigned long long> h_tile;
  uint64_t tot_elems = 0, tot_nz = 0, n_mslots = 0, tot_ovf = 0; unsigned m_tiles = 0;
  unsigned long long h_kept = 0; uint32_t h_max = 0;
  MBuildArgs a;
  const bool timing = getenv("MIC_LOAD_TIMING") != nullptr;
  struct timespec t_prev; clock_gettime(CLOCK_MONOTONIC, &t_prev);
  auto lap = [&](const char* what) {     // stage times: always into the build report (mic_db_last_build_report), on stderr with MIC_LOAD_TIMING
    hipStreamSynchronize(s);
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    const double dt = (t.tv_sec - t_prev.tv_sec) + (t.tv_nsec - t_prev.tv_nsec) / 1e9;
    mic_build_report_add(what, dt);
    if (timing) fprintf(stderr, "[load]   %s: %.3f s\n", what, dt);
    t_prev = t;
  };
  HIPCK(hipMalloc(&d_a, sizeof(TileA) * n_tiles));
  HIPCK(hipMalloc(&d_kept, 8));
  HIPCK(hipMalloc(&d_max, 4));
  HIPCK(hipMemsetAsync(d_kept, 0, 8, s));
  HIPCK(hipMemsetAsync(d_max, 0, 4, s));
  tile_a_kernel<<<n_tiles, TILE, 0, s>>>(d_sizes, n_buckets, d_a);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(h_a.data(), d_a, sizeof(TileA) * n_tiles, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  for (unsigned t = 0; t < n_tiles; ++t) {
    uint64_t e = h_a[t].elems, z = h_a[t].nonzero;
    h_a[t].elems = tot_elems; h_a[t].nonzero = tot_nz; tot_elems += e; tot_nz += z;
  }
  HIPCK(hipMemcpyAsync(d_a, h_a.data(), sizeof(TileA) * n_tiles, hipMemcpyHostToDevice, s));
  // Average entries per 12-entry main slot.  6 (half-full slots) is the measured optimum when memory allows: the
  // clustering of k-mers by minimizer makes the load lumpy and overflowing slots cost a second round.  When the table
  // would not fit in the free HBM the load is raised step by step (each trial is one counting pass, exact size).
  a.sizes = d_sizes; a.n_buckets = n_buckets; a.bucket0 = bucket0; a.htsize = htsize; a.keys = d_keys; a.labels = d_labels;
  a.base_a = d_a; a.sampling = sampling; a.rank_base = rank_base; a.k = k; a.m = m; a.fwd = 0;
#define BY_RAW(KERN, ...) do { if (key_bytes == 8) KERN<uint64_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); \
    else if (key_bytes == 4) KERN<uint32_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); \
    else KERN<uint16_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); } while (0)
  {
    unsigned long long loads[5] = {6, 7, 8, 9, 10};   // beyond ~9 the overflow trees outgrow what the main slots save
    int n_loads = 5;
    if (const char* env = getenv("MIC_MSLOT_LOAD")) { long v = atol(env); if (v >= 1 && v <= 12) { loads[0] = (unsigned long long)v; n_loads = 1; } }
    bool fits = false;
    double least_need = 0, avail_b = 0;
    for (int li = 0; li < n_loads && !fits; ++li) {
      const unsigned long long load = loads[li];
      n_mslots = tot_elems / (sampling > 1 ? load * sampling : load) + 64;
      if (n_mslots > 0xFFFFFF00ull) { snprintf(err, err_cap, "too many M-slots"); rc = -1; goto done; }
      m_tiles = (unsigned)((n_mslots + TILE - 1) / TILE);
      h_tile.resize(m_tiles);
      if (d_cnt) { hipFree(d_cnt); d_cnt = nullptr; }
      if (d_tile) { hipFree(d_tile); d_tile = nullptr; }
      if (hipMalloc(&d_cnt, n_mslots * 4) != hipSuccess || hipMalloc(&d_tile, (size_t)m_tiles * 8) != hipSuccess) { (void)hipGetLastError(); continue; }
      HIPCK(hipMemsetAsync(d_cnt, 0, n_mslots * 4, s));
      HIPCK(hipMemsetAsync(d_kept, 0, 8, s));
      HIPCK(hipMemsetAsync(d_max, 0, 4, s));
      a.n_mslots = n_mslots;
      BY_RAW(m_count_kernel, a, d_cnt, d_kept);
      HIPCK(hipGetLastError());
      m_ovf_tile_kernel<<<m_tiles, TILE, 0, s>>>(d_cnt, n_mslots, d_tile, d_max);
      HIPCK(hipGetLastError());
      HIPCK(hipMemcpyAsync(h_tile.data(), d_tile, (size_t)m_tiles * 8, hipMemcpyDeviceToHost, s));
      HIPCK(hipMemcpyAsync(&h_kept, d_kept, 8, hipMemcpyDeviceToHost, s));
      HIPCK(hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, s));
      HIPCK(hipStreamSynchronize(s));
      tot_ovf = 0;
      for (unsigned t = 0; t < m_tiles; ++t) { unsigned long long v = h_tile[t]; h_tile[t] = tot_ovf; tot_ovf += v; }
      if (n_mslots + tot_ovf > 0xFFFFFF00ull) continue;
      size_t free_b = 0, total_b = 0;
      HIPCK(hipMemGetInfo(&free_b, &total_b));
      // room for the slots = free HBM minus the two per-slot cursors of the scatter and the query batches afterwards
      double avail = (double)free_b - (double)n_mslots * 8 - 1.5e9 - (double)mic_build_reserved_hbm;
      if (const char* env = getenv("MIC_HBM_LIMIT_GB")) {   // test hook: pretend only this much is available for the slots
        const double lim = atof(env) * 1e9;
        if (lim > 0 && avail > lim) avail = lim;
      }
      const double need = (double)(n_mslots + tot_ovf + 1) * sizeof(MSlot);
      if (least_need == 0 || need < least_need) least_need = need;
      avail_b = avail;
      fits = need <= avail;
    }
    if (!fits) {
      snprintf(err, err_cap, "the minimizer table needs at least %.3f GB, %.3f GB of HBM are available for it", least_need / 1e9,
               avail_b / 1e9);
      rc = -3; goto done;
    }
  }
  lap("bucket sums + slot counts (sizing)");
  HIPCK(hipMemcpyAsync(d_tile, h_tile.data(), (size_t)m_tiles * 8, hipMemcpyHostToDevice, s));
  {
    hipError_t e_ = hipMalloc(&slots, (size_t)(n_mslots + tot_ovf + 1) * sizeof(MSlot));
    if (e_ != hipSuccess) {
      snprintf(err, err_cap, "hipMalloc of %.2f GB for the minimizer table failed: %s",
               (double)(n_mslots + tot_ovf + 1) * sizeof(MSlot) / 1e9, hipGetErrorString(e_));
      rc = -3; goto done;
    }
  }
  lap("hipMalloc of the slots");
  HIPCK(hipMalloc(&d_of, n_mslots * 4));
  m_header_kernel<<<m_tiles, TILE, 0, s>>>(d_cnt, n_mslots, d_tile, slots, d_of);
  HIPCK(hipGetLastError());
  lap("headers");
  HIPCK(hipMalloc(&d_cur, n_mslots * 4));
  HIPCK(hipMemsetAsync(d_cur, 0, n_mslots * 4, s));
  BY_RAW(m_scatter_kernel, a, d_cnt, d_of, d_cur, slots);
  HIPCK(hipGetLastError());
  lap("scatter");
  m_sort_kernel<<<(unsigned)((n_mslots + 255) / 256), 256, 0, s>>>(d_cnt, d_of, n_mslots, slots);
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(s));
  lap("sort + separators");
#undef BY_RAW
  out->slots = (uint4*)slots; slots = nullptr;
  out->n_main = n_mslots; out->n_overflow = tot_ovf; out->n_elems = h_kept; out->n_elems_file = tot_elems;
  out->max_bucket = 0; out->max_chain = h_max;
done:
  // (an error return: copies queued on s may still name this frame's host variables - they must have landed before it goes)
  if (rc) hipStreamSynchronize(s);
  if (d_a) hipFree(d_a);
  if (d_cnt) hipFree(d_cnt);
  if (d_cur) hipFree(d_cur);
  if (d_of) hipFree(d_of);
  if (d_tile) hipFree(d_tile);
  if (d_kept) hipFree(d_kept);
  if (d_max) hipFree(d_max);
  if (slots) hipFree(slots);
  return rc;
}

struct CastU64 { __host__ __device__ unsigned long long operator()(const uint32_t& v) const { return (unsigned long long)v; } };

static hipError_t scan_u32_to_u64(const uint32_t* in, unsigned long long* out, uint64_t n, hipStream_t s) {
  hipcub::TransformInputIterator<unsigned long long, CastU64, const uint32_t*> it(in, CastU64());
  size_t tb = 0; void* tmp = nullptr;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, out, (int)n, s);
  if (e != hipSuccess) return e;
  if ((e = hipMalloc(&tmp, tb ? tb : 16)) != hipSuccess) return e;
  e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, it, out, (int)n, s);
  hipError_t e2 = hipStreamSynchronize(s);
  hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

int mic_build_stable(const uint8_t* d_sizes, uint64_t n_buckets, uint64_t bucket0, uint64_t htsize, const void* d_keys,
                     int key_bytes, const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int k, int m,
                     hipStream_t s, MicBuildOut* out, char* err, size_t err_cap, int allow_fallback, int both_strands,
                     uint32_t part, uint32_t n_parts) {
  int rc = 0;
  const unsigned n_tiles = (unsigned)((n_buckets + TILE - 1) / TILE);
  TileA* d_a = nullptr; uint32_t* d_cnt = nullptr; uint32_t* d_cur = nullptr; unsigned long long* d_off = nullptr;
  // Slot-range part (table-sharded runs): the table is sized from the WHOLE database - every engine of a
  // sharded run computes the same n_slots - but only the main slots [part_lo, part_hi) and their continuation slots are
  // built and kept.  Slot indices stay global ("virtual": the slot pointer handed out is the allocation minus part_lo
  // slots), so the query kernels need nothing but the range test.
  uint64_t part_lo = 0, part_hi = 0, n_part = 0; unsigned long long off_lo = 0, off_hi = 0;
  uint4* side = nullptr; uint64_t side_cells = 0, side_kmers = 0;
  unsigned long long* d_coff = nullptr; unsigned long long* d_scal = nullptr; uint32_t* d_max = nullptr;
  unsigned long long* d_ck = nullptr; uint32_t* d_cm = nullptr; uint32_t* slots = nullptr; uint32_t* d_dem = nullptr;
  std::vector<TileA> h_a(n_tiles);
  uint64_t tot_elems = 0, tot_nz = 0, n_slots = 0, n_cand = 0, n_chain = 0;
  unsigned long long h_scal[2] = {0, 0}, h_entries = 0; uint32_t h_max = 0; double avail_b = 0;
  uint32_t* d_nent = nullptr; uint64_t stage_cap = 0; size_t n_ranges = 0; bool one_pass = false; uint64_t alloc_slots = 0;
  std::vector<uint64_t> range_lo; std::vector<unsigned long long> range_base;
  // the sorted build (see s_expand_kernel): sketch, counters, the records of all chunks (sorted), one chunk unsorted, the extras' list
  uint32_t* d_hll = nullptr; unsigned long long* d_ctr = nullptr; SKey* d_sh = nullptr; SVal* d_sv = nullptr;
  SKey* d_th = nullptr; SVal* d_tv = nullptr; SKey* d_xh = nullptr; SVal* d_xv = nullptr; void* d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0; uint32_t ex_cap = 0; uint64_t n_extra = 0; bool sorted = false, staged = false;
  auto drop_records = [&] {
    if (d_sh) { hipFree(d_sh); d_sh = nullptr; } if (d_sv) { hipFree(d_sv); d_sv = nullptr; }
    if (d_xh) { hipFree(d_xh); d_xh = nullptr; } if (d_xv) { hipFree(d_xv); d_xv = nullptr; }
  };
  MBuildArgs a;
  const bool timing = getenv("MIC_LOAD_TIMING") != nullptr;
  struct timespec t_prev; clock_gettime(CLOCK_MONOTONIC, &t_prev);
  auto lap = [&](const char* what) {     // stage times: always into the build report (mic_db_last_build_report), on stderr with MIC_LOAD_TIMING
    hipStreamSynchronize(s);
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    const double dt = (t.tv_sec - t_prev.tv_sec) + (t.tv_nsec - t_prev.tv_nsec) / 1e9;
    mic_build_report_add(what, dt);
    if (timing) fprintf(stderr, "[load]   %s: %.3f s\n", what, dt);
    t_prev = t;
  };
#define BY_RAW(KERN, ...) do { if (key_bytes == 8) KERN<uint64_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); \
    else if (key_bytes == 4) KERN<uint32_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); \
    else KERN<uint16_t><<<n_tiles, TILE, 0, s>>>(__VA_ARGS__); } while (0)
  HIPCK(hipMalloc(&d_a, sizeof(TileA) * n_tiles));
  HIPCK(hipMalloc(&d_scal, 16));
  HIPCK(hipMalloc(&d_max, 4));
  tile_a_kernel<<<n_tiles, TILE, 0, s>>>(d_sizes, n_buckets, d_a);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(h_a.data(), d_a, sizeof(TileA) * n_tiles, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  for (unsigned t = 0; t < n_tiles; ++t) {
    uint64_t e = h_a[t].elems, z = h_a[t].nonzero;
    h_a[t].elems = tot_elems; h_a[t].nonzero = tot_nz; tot_elems += e; tot_nz += z;
  }
  HIPCK(hipMemcpyAsync(d_a, h_a.data(), sizeof(TileA) * n_tiles, hipMemcpyHostToDevice, s));
  a.sizes = d_sizes; a.n_buckets = n_buckets; a.bucket0 = bucket0; a.htsize = htsize; a.keys = d_keys; a.labels = d_labels;
  a.base_a = d_a; a.sampling = sampling; a.rank_base = rank_base; a.k = k; a.m = m; a.fwd = both_strands ? 1 : 0;
  // Sizing: the number of entries is about the number of distinct minimizer values D among the stored k-mers' candidates; the
  // table gets D / 1.5 slots (6 entries each: P(overflow) ~ 2e-4 for Poisson(1.5)); MIC_SSLOT_LOAD overrides the 1.5.  D comes from
  // a HyperLogLog sketch (4096 registers: +-1.6 %) filled by s_expand_kernel - on the sorted road in the same pass that writes the
  // records, on the classic road in a pass of its own (it replaces the classic form's first counting pass; until round 5 D was
  // read off the occupancy of 1.4 G provisional counters).  The sketch is a function of the SET of minimizer values: every engine
  // of a table-sharded run, whichever road it takes, sizes the table identically.
  {
    const uint64_t CHUNK = [] { const char* e = getenv("MIC_S_SORT_CHUNK"); const long long v = e ? atoll(e) : 0; return v >= 1024 ? (uint64_t)v : (uint64_t)1 << 28; }();
    bool want_sorted = !both_strands && tot_elems > 0 && !getenv("MIC_S_CLASSIC");
    std::vector<uint32_t> cut;                 // chunk c = tiles [cut[c], cut[c + 1]): at most CHUNK k-mers (or one tile)
    uint64_t chunk_max = 0;
    {
      uint32_t t0 = 0;
      cut.push_back(0);
      for (unsigned tl = 0; tl < n_tiles; ++tl) {
        const uint64_t end = tl + 1 < n_tiles ? h_a[tl + 1].elems : tot_elems;
        if (end - h_a[t0].elems > CHUNK && tl > t0) { cut.push_back(tl); t0 = tl; }
        const uint64_t here = end - h_a[t0].elems;
        if (here > chunk_max) chunk_max = here;
      }
      cut.push_back(n_tiles);
      if (chunk_max > 0x7FFFFF00ull) want_sorted = false;       // (one tile of more k-mers than a sort call takes: no such table)
    }
    ex_cap = (uint32_t)std::min<uint64_t>(tot_elems / 16 + (1u << 20), 0x7FFFFF00ull);
    if (const char* env = getenv("MIC_S_EXTRA_CAP")) { const long v = atol(env); if (v > 0) ex_cap = (uint32_t)v; }   // test hook: a list that runs over
    if (want_sorted) {
      // scratch of the sort: the largest any chunk asks for (the library picks its algorithm - and its scratch layout - by the number
      // of items: the need is not promised to grow with it)
      sort_tmp_bytes = 0;
      for (size_t c = 0; c + 1 < cut.size(); ++c) {
        const uint64_t e0 = h_a[cut[c]].elems, e1 = cut[c + 1] < n_tiles ? h_a[cut[c + 1]].elems : tot_elems;
        if (e1 == e0) continue;
        size_t tb = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const SKey*)nullptr, (SKey*)nullptr, (const SVal*)nullptr, (SVal*)nullptr, (int)(e1 - e0), 16, 32, s);
        if (tb > sort_tmp_bytes) sort_tmp_bytes = tb;
      }
      size_t free_b = 0, total_b = 0;
      HIPCK(hipMemGetInfo(&free_b, &total_b));
      double avail = (double)free_b - 1.5e9 - (double)mic_build_reserved_hbm;
      if (const char* env = getenv("MIC_HBM_LIMIT_GB")) { const double lim = atof(env) * 1e9; if (lim > 0 && avail > lim) avail = lim; }
      // records of every k-mer + one chunk unsorted + the sort's scratch + the extras' list, and - while the records are still
      // there - the staging area of this engine's slots and the per-slot arrays (the table itself comes after the records went)
      const double need = (double)tot_elems * 16 + (double)chunk_max * 16 + (double)sort_tmp_bytes + (double)ex_cap * 16 +
                          (double)tot_elems * 12 / (n_parts > 1 ? n_parts : 1) + (double)tot_elems * 4 + 128e6;
      if (need > avail) want_sorted = false;
      if (timing) fprintf(stderr, "[load]   sorted build: %.2f GB needed, %.2f GB available -> %s\n", need / 1e9, avail / 1e9, want_sorted ? "taken" : "classic build");
    }
    HIPCK(hipMalloc(&d_hll, sizeof(uint32_t) << S_HLL_BITS));
    HIPCK(hipMalloc(&d_ctr, 3 * 8));
    HIPCK(hipMemsetAsync(d_hll, 0, sizeof(uint32_t) << S_HLL_BITS, s));
    HIPCK(hipMemsetAsync(d_ctr, 0, 3 * 8, s));
    if (want_sorted) {
      hipError_t e_ = hipMalloc(&d_sh, (tot_elems + 1) * sizeof(SKey));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_sv, (tot_elems + 1) * sizeof(SVal));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_th, (chunk_max + 1) * sizeof(SKey));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_tv, (chunk_max + 1) * sizeof(SVal));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_xh, (size_t)(ex_cap + 1) * sizeof(SKey));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_xv, (size_t)(ex_cap + 1) * sizeof(SVal));
      if (e_ == hipSuccess) e_ = hipMalloc(&d_sort_tmp, sort_tmp_bytes ? sort_tmp_bytes : 16);
      if (e_ != hipSuccess) { (void)hipGetLastError(); want_sorted = false; }
    }
#define EXPAND(EMIT, T0, T1, E0) do { const unsigned g_ = (unsigned)std::min<uint64_t>((uint64_t)(T1) - (T0), 2048); \
    if (key_bytes == 8) s_expand_kernel<uint64_t, EMIT><<<g_, TILE, 0, s>>>(a, T0, T1, E0, d_th, d_tv, d_xh, d_xv, ex_cap, d_ctr, d_hll); \
    else if (key_bytes == 4) s_expand_kernel<uint32_t, EMIT><<<g_, TILE, 0, s>>>(a, T0, T1, E0, d_th, d_tv, d_xh, d_xv, ex_cap, d_ctr, d_hll); \
    else s_expand_kernel<uint16_t, EMIT><<<g_, TILE, 0, s>>>(a, T0, T1, E0, d_th, d_tv, d_xh, d_xv, ex_cap, d_ctr, d_hll); } while (0)
    a.n_mslots = 0;
    if (want_sorted) {
      bool sort_failed = false;
      for (size_t c = 0; c + 1 < cut.size(); ++c) {
        const uint32_t t0 = cut[c], t1 = cut[c + 1];
        const uint64_t e0 = h_a[t0].elems, e1 = t1 < n_tiles ? h_a[t1].elems : tot_elems;
        if (e1 == e0) continue;
        EXPAND(true, t0, t1, e0);
        HIPCK(hipGetLastError());
        size_t tb = sort_tmp_bytes;
        if (getenv("MIC_S_NOSORT")) {          // (debugging: the records as they come, unsorted)
          HIPCK(hipMemcpyAsync(d_sh + e0, d_th, (e1 - e0) * sizeof(SKey), hipMemcpyDeviceToDevice, s));
          HIPCK(hipMemcpyAsync(d_sv + e0, d_tv, (e1 - e0) * sizeof(SVal), hipMemcpyDeviceToDevice, s));
        } else
        if (hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tb, (const SKey*)d_th, d_sh + e0, (const SVal*)d_tv, d_sv + e0, (int)(e1 - e0), 16, 32, s) != hipSuccess) {
          (void)hipGetLastError();
          sort_failed = true;            // (a sort that refuses its scratch or its launch: the classic road builds the same table)
          break;
        }
      }
      unsigned long long h_ctr[3] = {0, 0, 0};
      HIPCK(hipMemcpyAsync(h_ctr, d_ctr, 24, hipMemcpyDeviceToHost, s));
      HIPCK(hipStreamSynchronize(s));
      if (sort_failed) {
        want_sorted = false;
        if (timing) fprintf(stderr, "[load]   sorted build given up: the radix sort of a chunk failed\n");
      } else if (h_ctr[2]) {   // more tied candidates than the extras' list holds (a database of low complexity): the classic road
        want_sorted = false;
        if (timing) fprintf(stderr, "[load]   sorted build given up: %llu extra candidates beyond the list of %u\n", h_ctr[2], ex_cap);
      } else { sorted = true; n_extra = h_ctr[1]; h_scal[0] = h_ctr[0]; }
      if (timing) fprintf(stderr, "[load]   sorted build: %llu k-mers kept of %llu, %llu extra candidates\n", h_ctr[0], (unsigned long long)tot_elems, h_ctr[1]);
      hipFree(d_th); d_th = nullptr; hipFree(d_tv); d_tv = nullptr; hipFree(d_sort_tmp); d_sort_tmp = nullptr;
      if (sorted) lap("candidates of every k-mer once + sketch of the minimizers + radix sort by slot hash (sorted build)");
    }
    if (!sorted) {
      if (d_sh) { hipFree(d_sh); d_sh = nullptr; } if (d_sv) { hipFree(d_sv); d_sv = nullptr; }
      if (d_xh) { hipFree(d_xh); d_xh = nullptr; } if (d_xv) { hipFree(d_xv); d_xv = nullptr; }
      if (d_th) { hipFree(d_th); d_th = nullptr; } if (d_tv) { hipFree(d_tv); d_tv = nullptr; }
      if (d_sort_tmp) { hipFree(d_sort_tmp); d_sort_tmp = nullptr; }
      HIPCK(hipMemsetAsync(d_hll, 0, sizeof(uint32_t) << S_HLL_BITS, s));
      HIPCK(hipMemsetAsync(d_ctr, 0, 3 * 8, s));
      EXPAND(false, 0u, n_tiles, 0ull);
      HIPCK(hipGetLastError());
    }
#undef EXPAND
    std::vector<uint32_t> h_hll((size_t)1 << S_HLL_BITS);
    HIPCK(hipMemcpyAsync(h_hll.data(), d_hll, sizeof(uint32_t) << S_HLL_BITS, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    const double D = hll_estimate(h_hll.data());
    double load = 1.5;
    if (const char* env = getenv("MIC_SSLOT_LOAD")) { double v = atof(env); if (v > 0.05 && v < 6) load = v; }
    n_slots = (uint64_t)(D / load) + 64;
    // the scans below take an int item count; a table this large (> 2^31 slots = 275 GB) does not fit one GPU anyway
    if (n_slots > 0x7FFFFF00ull) { snprintf(err, err_cap, "the super-k-mer table would need %llu slots", (unsigned long long)n_slots); rc = -3; goto done; }
    HIPCK(hipMalloc(&d_cnt, n_slots * 4));
    HIPCK(hipMemsetAsync(d_cnt, 0, n_slots * 4, s));
    a.n_mslots = n_slots;
    if (sorted) {
      // (a slot-range part counts - and later stages - its own slots only)
      const uint32_t c_lo = n_parts > 1 ? (uint32_t)((unsigned __int128)n_slots * part / n_parts) : 0u;
      const uint32_t c_hi = n_parts > 1 ? (uint32_t)((unsigned __int128)n_slots * (part + 1) / n_parts) : (uint32_t)n_slots;
      s_rec_count_kernel<<<rec_grid(tot_elems), 256, 0, s>>>(d_sh, tot_elems, (uint32_t)n_slots, c_lo, c_hi, d_cnt);
      if (n_extra) s_rec_count_kernel<<<rec_grid(n_extra), 256, 0, s>>>(d_xh, n_extra, (uint32_t)n_slots, c_lo, c_hi, d_cnt);
      HIPCK(hipGetLastError());
      HIPCK(hipStreamSynchronize(s));
    } else {
      HIPCK(hipMemsetAsync(d_scal, 0, 16, s));
      BY_RAW(s_count_kernel, a, d_cnt, d_scal);
      HIPCK(hipGetLastError());
      HIPCK(hipMemcpyAsync(h_scal, d_scal, 8, hipMemcpyDeviceToHost, s));
      HIPCK(hipStreamSynchronize(s));
    }
  }
  lap("bucket sums + slot counts (sizing)");
  HIPCK(hipMalloc(&d_off, (n_slots + 1) * 8));
  HIPCK(scan_u32_to_u64(d_cnt, d_off, n_slots, s));
  {
    unsigned long long last_off = 0; uint32_t last_cnt = 0;
    HIPCK(hipMemcpy(&last_off, d_off + n_slots - 1, 8, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(&last_cnt, d_cnt + n_slots - 1, 4, hipMemcpyDeviceToHost));
    n_cand = last_off + last_cnt;
  }
  if (n_parts > 1) {
    part_lo = (uint64_t)((unsigned __int128)n_slots * part / n_parts);
    part_hi = (uint64_t)((unsigned __int128)n_slots * (part + 1) / n_parts);
    HIPCK(hipMemcpy(&off_lo, d_off + part_lo, 8, hipMemcpyDeviceToHost));
    if (part_hi < n_slots) HIPCK(hipMemcpy(&off_hi, d_off + part_hi, 8, hipMemcpyDeviceToHost)); else off_hi = n_cand;
  } else { part_lo = 0; part_hi = n_slots; off_lo = 0; off_hi = n_cand; }
  n_part = part_hi - part_lo;
  {
    // What must fit next to what is resident already: the table (128 B per slot, ~1 % continuation slots), 32 B of per-slot
    // arrays, and a staging area of 12 B per candidate.  When all candidates do not fit at once the staging area takes what
    // is left and the table is built in several passes over slot ranges (each pass scatters only its range's candidates).
    size_t free_b = 0, total_b = 0;
    HIPCK(hipMemGetInfo(&free_b, &total_b));
    avail_b = (double)free_b - 1.5e9 - (double)mic_build_reserved_hbm;
    // sorted build: the records are resident now and are freed after the staging is filled, before the table is allocated
    const double rec_bytes = sorted ? (double)tot_elems * 16 + (double)ex_cap * 16 : 0.0;
    if (const char* env = getenv("MIC_HBM_LIMIT_GB")) {   // test hook: pretend only this much is available
      const double lim = atof(env) * 1e9;
      if (lim > 0 && avail_b + rec_bytes > lim) avail_b = lim - rec_bytes;
    }
    const double fixed = (double)n_part * (128 * 1.02) + (double)n_slots * 24 + 64e6;    // table + chain offsets / cursors / entry counts (counts and offsets exist already)
    double budget = avail_b + rec_bytes - fixed;
    if (sorted && ((double)(off_hi - off_lo) * 12 > avail_b - (double)n_slots * 24 - 64e6 || (double)(off_hi - off_lo) * 12 > budget)) {
      // the staging area does not fit next to the records (or would need several passes): the classic road from here on - the
      // counts and offsets are the same on both
      if (timing) fprintf(stderr, "[load]   sorted build given up: no room for the staging area next to the records\n");
      hipFree(d_sh); d_sh = nullptr; hipFree(d_sv); d_sv = nullptr; hipFree(d_xh); d_xh = nullptr; hipFree(d_xv); d_xv = nullptr;
      sorted = false;
      HIPCK(hipMemGetInfo(&free_b, &total_b));
      avail_b = (double)free_b - 1.5e9 - (double)mic_build_reserved_hbm;
      if (const char* env = getenv("MIC_HBM_LIMIT_GB")) { const double lim = atof(env) * 1e9; if (lim > 0 && avail_b > lim) avail_b = lim; }
      budget = avail_b - fixed;
    }
    if (const char* env = getenv("MIC_S_STAGING_LIMIT_MB")) {           // test hook: a small staging area forces several passes
      const double lim = atof(env) * 1e6;
      if (lim > 0 && budget > lim) budget = lim;
    }
    const uint64_t n_cand_part = off_hi - off_lo;
    if (budget < (double)n_cand_part * 12 / 16 || budget < 4096) {
      snprintf(err, err_cap, "the super-k-mer table and its build need at least %.3f GB, %.3f GB of HBM are available",
               (fixed + (double)n_cand_part * 12 / 16) / 1e9, avail_b / 1e9);
      rc = -3; goto done;
    }
    stage_cap = (double)n_cand_part * 12 <= budget ? n_cand_part : (uint64_t)(budget / 12);
  }
  {
    // slot ranges whose candidates fit the staging area: boundaries at multiples of G slots (2^16 for a large table), from a
    // strided copy of the offsets
    uint64_t G = 1u << 16;
    while (G > 64 && n_slots / G < 64) G >>= 1;
    const uint64_t n_samples = n_slots / G + 1;
    std::vector<unsigned long long> samp(n_samples);
    HIPCK(hipMemcpy2D(samp.data(), 8, d_off, 8 * G, 8, n_samples, hipMemcpyDeviceToHost));
    range_lo.clear(); range_base.clear();
    uint64_t lo = part_lo; unsigned long long lo_off = off_lo;
    while (lo < part_hi) {
      // largest multiple of G below the end of the part (or that end) whose offset stays within the staging capacity
      uint64_t hi = part_hi; unsigned long long hi_off = off_hi;
      if (off_hi - lo_off > stage_cap) {
        uint64_t g = lo / G + 1;
        while (g < n_samples && g * G < part_hi && samp[g] - lo_off <= stage_cap) ++g;
        hi = (g - 1) * G;
        if (hi <= lo) { snprintf(err, err_cap, "a block of %llu slots holds more candidates than the staging area", (unsigned long long)G); rc = -3; goto done; }
        hi_off = samp[g - 1];
      }
      range_lo.push_back(lo); range_base.push_back(lo_off);
      lo = hi; lo_off = hi_off;
    }
    range_lo.push_back(part_hi); range_base.push_back(off_hi);
    n_ranges = range_lo.size() - 1;
    uint64_t biggest = 0;
    for (size_t r = 0; r < n_ranges; ++r) biggest = std::max<uint64_t>(biggest, range_base[r + 1] - range_base[r]);
    hipError_t e1 = hipMalloc(&d_ck, (biggest + 1) * 8), e2 = hipMalloc(&d_cm, (biggest + 1) * 4);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void)hipGetLastError();
      snprintf(err, err_cap, "the super-k-mer build needs %.2f GB of staging", (double)biggest * 12 / 1e9); rc = -3; goto done;
    }
    if (timing) fprintf(stderr, "[load]   %zu pass(es) over slot ranges, staging %.2f GB for %.2f G candidates (%llu)\n", n_ranges, (double)biggest * 12 / 1e9, n_cand / 1e9, (unsigned long long)n_cand);
  }
  HIPCK(hipMalloc(&d_cur, n_slots * 4));      // scatter cursors
  HIPCK(hipMalloc(&d_nent, n_slots * 4));     // entries per slot
  HIPCK(hipMemsetAsync(d_cur, 0, n_slots * 4, s));
  if (n_part != n_slots) HIPCK(hipMemsetAsync(d_nent, 0, n_slots * 4, s));   // slots of other parts: no entries, no chains
  HIPCK(hipMemsetAsync(d_max, 0, 4, s));
  // One pass when it works: the table is allocated with a pool of continuation slots (2 % of the main slots; Poisson(1.5)
  // entries per slot needs 0.1 %) and every range is scattered, sorted, merged and WRITTEN in one go - no counting merge, and
  // for a table built in several ranges no second scatter + sort either.  A pool that runs out (a database of crowded
  // minimizers, which the rule below hands to the minimizer layout anyway) or an allocation that fails falls back to the
  // two-pass form: count the entries, size the chains exactly, write.  MIC_S_TWO_PASS=1 forces that form.
  if (sorted) {
    // sorted build: the staging area is filled from the records (a streaming read; the writes stay inside the window of the
    // staging area the records' order sweeps), then the records go - before the table is allocated
    if (n_ranges != 1) {       // (a staging area cut into ranges - MIC_S_STAGING_LIMIT_MB - is the classic road's)
      hipFree(d_sh); d_sh = nullptr; hipFree(d_sv); d_sv = nullptr; hipFree(d_xh); d_xh = nullptr; hipFree(d_xv); d_xv = nullptr;
      sorted = false;
    }
  }
  if (sorted) {
    s_rec_place_kernel<<<rec_grid(tot_elems), 256, 0, s>>>(d_sh, d_sv, tot_elems, (uint32_t)n_slots, d_off, d_cur, d_ck, d_cm,
                                                                     (uint32_t)range_lo[0], (uint32_t)range_lo[1], range_base[0]);
    if (n_extra) s_rec_place_kernel<<<rec_grid(n_extra), 256, 0, s>>>(d_xh, d_xv, n_extra, (uint32_t)n_slots, d_off, d_cur, d_ck, d_cm,
                                                                                 (uint32_t)range_lo[0], (uint32_t)range_lo[1], range_base[0]);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s));
    staged = true;
    lap("records counted per slot, offsets, staged (sorted build)");
    // The records (16 B per k-mer) go only AFTER the table is allocated, when there is room for both: memory that was just freed
    // is handed out again only after the driver has wiped it (~40 GB/s: an allocation of the table's size right behind the free
    // waited 2.8 s for it), memory that was never used comes at once.  Without the room they go first.
  }
  if (staged && getenv("MIC_S_TWO_PASS")) drop_records();
  if (!getenv("MIC_S_TWO_PASS")) {
    uint64_t pool_cap = n_part / 50 + 65536;
    if (const char* env = getenv("MIC_S_POOL_SLOTS")) { long v = atol(env); if (v > 0) pool_cap = (uint64_t)v; }   // test hook: a pool that runs out
    if (part_hi + pool_cap > 0xFFFFFF00ull) pool_cap = 0xFFFFFF00ull > part_hi ? 0xFFFFFF00ull - part_hi : 0;
    uint32_t* d_pool = nullptr;
    hipError_t e_ = pool_cap ? hipMalloc(&slots, (size_t)(n_part + pool_cap + 1) * 128) : hipErrorOutOfMemory;
    if (e_ != hipSuccess && d_sh) {          // no room next to the records: they go first (and the allocation waits for the wipe)
      (void)hipGetLastError();
      drop_records();
      e_ = pool_cap ? hipMalloc(&slots, (size_t)(n_part + pool_cap + 1) * 128) : hipErrorOutOfMemory;
    }
    drop_records();
    if (e_ == hipSuccess) e_ = hipMalloc(&d_pool, 8);
    if (e_ == hipSuccess) e_ = hipMemsetAsync(d_pool, 0, 8, s);
    lap("offsets scan + staging / cursor / table allocations (hipMalloc waits for the driver's wipe of freed pages)");
    if (e_ == hipSuccess) {
      for (size_t r = 0; r < n_ranges && e_ == hipSuccess; ++r) {
        const uint64_t lo = range_lo[r], hi = range_lo[r + 1];
        if (!staged) BY_RAW(s_scatter_kernel, a, d_off, d_cur, d_ck, d_cm, (uint32_t)lo, (uint32_t)hi, range_base[r]);
        s_merge_kernel<2><<<(unsigned)((hi - lo + S_TPB - 1) / S_TPB), S_TPB, 0, s>>>(d_off, d_cnt, part_hi, d_ck, d_cm, k, m, d_nent, nullptr, slots - part_lo * 32,
                                                                           d_max, lo, hi, range_base[r], 1, d_pool, (uint32_t)pool_cap);
        e_ = hipGetLastError();
      }
      uint32_t h_pool[2] = {0, 0};
      if (e_ == hipSuccess) e_ = hipMemcpyAsync(h_pool, d_pool, 8, hipMemcpyDeviceToHost, s);
      if (e_ == hipSuccess) e_ = hipStreamSynchronize(s);
      if (e_ == hipSuccess && !h_pool[1]) { one_pass = true; n_chain = h_pool[0]; alloc_slots = n_part + pool_cap + 1; }
    }
    if (d_pool) hipFree(d_pool);
    if (!one_pass) {
      (void)hipGetLastError();
      if (slots) { hipFree(slots); slots = nullptr; }
      HIPCK(hipMemsetAsync(d_cur, 0, n_slots * 4, s));
      HIPCK(hipMemsetAsync(d_max, 0, 4, s));
    }
    lap(one_pass ? "scatter of the candidates + sort + merge + write (one pass)" : "one-pass build abandoned (continuation pool full or no memory)");
  }
  // phase A (two-pass form): entries per slot (scatter + sort + merge without writing), range by range
  for (size_t r = 0; r < n_ranges && !one_pass; ++r) {
    const uint64_t lo = range_lo[r], hi = range_lo[r + 1];
    if (!staged) BY_RAW(s_scatter_kernel, a, d_off, d_cur, d_ck, d_cm, (uint32_t)lo, (uint32_t)hi, range_base[r]);
    HIPCK(hipGetLastError());
    s_merge_kernel<0><<<(unsigned)((hi - lo + S_TPB - 1) / S_TPB), S_TPB, 0, s>>>(d_off, d_cnt, part_hi, d_ck, d_cm, k, m, d_nent, nullptr,
                                                                       nullptr, d_max, lo, hi, range_base[r], 1, nullptr, 0);
    HIPCK(hipGetLastError());
  }
  if (!one_pass) lap("scatter of the candidates + sort + merge (count)");
  HIPCK(hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  HIPCK(hipMemsetAsync(d_scal + 1, 0, 8, s));
  s_sum_kernel<<<4096, 256, 0, s>>>(d_nent, n_slots, d_scal + 1);       // entries = super-k-mers stored (statistics)
  HIPCK(hipMemcpyAsync(&h_entries, d_scal + 1, 8, hipMemcpyDeviceToHost, s));
  {
    unsigned long long h_walk = 0;
    HIPCK(hipMemsetAsync(d_scal, 0, 8, s));
    s_walk_kernel<<<4096, 256, 0, s>>>(d_cnt, d_nent, n_slots, d_scal);
    HIPCK(hipMemcpyAsync(&h_walk, d_scal, 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    const double walk = off_hi > off_lo ? (double)h_walk / (double)(off_hi - off_lo) : 0.0;
    out->walk_ppm = walk * 1e6 > 4e9 ? 4000000000u : (uint32_t)(walk * 1e6);
    // A table whose stored k-mers sit, on average, behind more than this many continuation slots is answered faster by the
    // minimizer layout (fan-out-12 trees: logarithmic in the bucket size): measured with tandem repeats in the targets,
    // profiles/r02_nonideal_databases.json, DESIGN.md 5.3.  Only when the layout was not asked for explicitly.
    // Measured (2 G k-mers, 4 M reads): 0.0046 random genomes and 0.0060 with 10 % homologous segments (super-k-mer layout
    // 1 300 Mreads/s, minimizer layout 930); 0.0148 with 2 % tandem repeats (790 vs 945) and 0.0192 with 5 % (550 vs 925).
    // (Until round 3 a table whose walk exceeded 0.010 was abandoned for the minimizer layout when nobody had asked for this one;
    // the crowded minimizers now leave their chains for a side table below, and the walk of what remains is ~0.)
    if (const char* env = getenv("MIC_S_WALK_LIMIT")) {
      const double limit = atof(env);
      if (allow_fallback && limit > 0 && walk > limit && getenv("MIC_S_NO_SIDE")) {
        snprintf(err, err_cap, "crowded minimizers: a stored k-mer sits behind %.4f continuation slots on average (limit %.4f)", walk, limit);
        rc = -5; goto done;
      }
    }
    // ... and a database whose k-mers do not overlap (cuCLARK-l's sampled blocks: one k-mer per entry) gains nothing from
    // super-k-mers in MEMORY: the minimizer layout holds it in a quarter of the bytes.  Since the per-run kernel works on
    // t-mers (round 3) the super-k-mer table answers faster even so (1 620 vs 1 480 Mreads/s on cuCLARK-l's blocks), so the
    // minimizer layout is only taken when the table would be a large one (MIC_S_SMALL_TABLE_GB, default 16: cuCLARK-l's
    // databases are made for 4-GB cards and stay far below)
    double min_per_entry = 1.3, small_gb = 16.0;
    if (const char* env = getenv("MIC_S_MIN_KMERS_PER_ENTRY")) min_per_entry = atof(env);
    if (const char* env = getenv("MIC_S_SMALL_TABLE_GB")) small_gb = atof(env);
    const double stored = (double)h_scal[0] * (both_strands ? 2.0 : 1.0);
    if (allow_fallback && h_entries && stored / (double)h_entries < min_per_entry && (double)n_slots * 128.0 / 1e9 > small_gb) {
      snprintf(err, err_cap, "%.2f k-mers per super-k-mer entry (limit %.2f) in a table of %.1f GB: no adjacency to exploit", stored / (double)h_entries,
               min_per_entry, (double)n_slots * 128.0 / 1e9);
      rc = -5; goto done;
    }
  }
  if (!one_pass) {
    // chain slots: demand per slot into a scratch u32 array (the candidate offsets stay), scanned to 64-bit bases
    HIPCK(hipMalloc(&d_dem, n_slots * 4));
    s_chain_demand_kernel<<<(unsigned)((n_slots + 255) / 256), 256, 0, s>>>(d_nent, n_slots, d_dem);
    HIPCK(hipMalloc(&d_coff, (n_slots + 1) * 8));
    hipError_t e = scan_u32_to_u64(d_dem, d_coff, n_slots, s);
    unsigned long long last_off = 0; uint32_t last_dem = 0;
    if (e == hipSuccess) e = hipMemcpy(&last_off, d_coff + n_slots - 1, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&last_dem, d_dem + n_slots - 1, 4, hipMemcpyDeviceToHost);
    hipFree(d_dem); d_dem = nullptr;
    HIPCK(e);
    n_chain = last_off + last_dem;
  }
  if (part_hi + n_chain > 0xFFFFFF00ull) { snprintf(err, err_cap, "too many S-slots"); rc = -1; goto done; }
  if (!one_pass) {
    hipError_t e_ = hipMalloc(&slots, (size_t)(n_part + n_chain + 1) * 128);
    if (e_ != hipSuccess) {
      (void)hipGetLastError();
      snprintf(err, err_cap, "hipMalloc of %.2f GB for the super-k-mer table failed: %s",
               (double)(n_part + n_chain + 1) * 128 / 1e9, hipGetErrorString(e_));
      rc = -3; goto done;
    }
  }
  // phase B: write the slots.  One range: its candidates are still staged and sorted.  Several: scatter and sort each again.
  if (n_ranges > 1 && !one_pass) HIPCK(hipMemsetAsync(d_cur, 0, n_slots * 4, s));
  for (size_t r = 0; r < n_ranges && !one_pass; ++r) {
    const uint64_t lo = range_lo[r], hi = range_lo[r + 1];
    if (n_ranges > 1) {
      BY_RAW(s_scatter_kernel, a, d_off, d_cur, d_ck, d_cm, (uint32_t)lo, (uint32_t)hi, range_base[r]);
      HIPCK(hipGetLastError());
    }
    s_merge_kernel<1><<<(unsigned)((hi - lo + S_TPB - 1) / S_TPB), S_TPB, 0, s>>>(d_off, d_cnt, part_hi, d_ck, d_cm, k, m, d_nent, d_coff,
                                                                       slots - part_lo * 32, d_max, lo, hi, range_base[r], n_ranges > 1 ? 1 : 0, nullptr, 0);
    HIPCK(hipGetLastError());
  }
  HIPCK(hipMemsetAsync(slots + (size_t)(n_part + n_chain) * 32, 0xFF, 128, s));   // the spare slot after the table: empty
  HIPCK(hipStreamSynchronize(s));
  if (!one_pass) lap("merge (write)");
  // Crowded minimizers (one minimizer value in more than MIC_S_CROWD contexts): out of the chains, into a side table keyed by
  // the k-mer (see s_crowd_move_kernel).  Only the one-pass / two-pass builds that left the table in `slots` can be
  // post-processed here, i.e. always; a side table that cannot be allocated leaves the chains as they are (slower, exact).
  if (h_max > MIC_S_CROWD && !getenv("MIC_S_NO_SIDE")) {
    unsigned long long h_crowd[2] = {0, 0};
    HIPCK(hipMemsetAsync(d_scal, 0, 16, s));
    s_crowd_count_kernel<<<(unsigned)((n_part + 255) / 256), 256, 0, s>>>(slots - part_lo * 32, part_lo, part_hi, k, m, d_scal);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(h_crowd, d_scal, 16, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    if (h_crowd[1] > 0) {
      // open addressing with linear probing, probed by all 64 lanes of a wavefront in lockstep: the wavefront pays the LONGEST probe
      // sequence of its lanes, so the table is kept sparse (at 1/2 full, the round-3 setting, the unluckiest of 64 lanes walks 8
      // cells and more).  MIC_S_SIDE_SPARSE overrides the factor (cells >= factor x k-mers).
      // Measured on the 5 %-tandem-repeat database (tools/nonideal_bench.py, 4 M reads, one-strand / two-strand table): factor 2:
      // 1 188 / 1 321 Mreads/s, 4: 1 318 / 1 510, 8: 1 420 / 1 578 (random genomes: 1 787 / 1 917).  16 by default, less when the side
      // table would outgrow an eighth of the main table.
      uint64_t cells = 1024, sparse = 16;
      if (const char* env = getenv("MIC_S_SIDE_SPARSE")) { const long v = atol(env); if (v >= 2 && v <= 64) sparse = (uint64_t)v; }
      while (sparse > 2 && sparse * h_crowd[0] * 16 > n_part * 128 / 8) sparse >>= 1;
      while (cells < sparse * h_crowd[0]) cells <<= 1;
      if (cells <= 0x80000000ull && hipMalloc(&side, cells * 16) == hipSuccess) {
        HIPCK(hipMemsetAsync(side, 0, cells * 16, s));
        HIPCK(hipMemsetAsync(d_max, 0, 4, s));
        s_crowd_move_kernel<<<(unsigned)((n_part + 255) / 256), 256, 0, s>>>(slots - part_lo * 32, part_lo, part_hi, k, m, side, (uint32_t)(cells - 1),
                                                                           d_nent, d_max);
        HIPCK(hipGetLastError());
        HIPCK(hipMemsetAsync(d_max, 0, 4, s));
        s_max_kernel<<<4096, 256, 0, s>>>(d_nent, n_slots, d_max);
        HIPCK(hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        side_cells = cells; side_kmers = h_crowd[0];
        // the statistics of what stays in the chains
        unsigned long long h_walk = 0;
        HIPCK(hipMemsetAsync(d_scal, 0, 16, s));
        s_sum_kernel<<<4096, 256, 0, s>>>(d_nent, n_slots, d_scal + 1);
        s_walk_kernel<<<4096, 256, 0, s>>>(d_cnt, d_nent, n_slots, d_scal);
        HIPCK(hipMemcpyAsync(&h_entries, d_scal + 1, 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipMemcpyAsync(&h_walk, d_scal, 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        { const double walk = off_hi > off_lo ? (double)h_walk / (double)(off_hi - off_lo) : 0.0; out->walk_ppm = (uint32_t)(walk * 1e6); }
        lap("crowded minimizers moved to the side table");
      } else {
        (void)hipGetLastError();
        side = nullptr;
      }
    }
  }
#undef BY_RAW
  out->slots = (uint4*)slots; slots = nullptr;
  out->part_lo = part_lo; out->part_hi = part_hi;
  out->side = side; side = nullptr; out->side_cells = side_cells; out->side_kmers = side_kmers;
  out->n_main = n_slots; out->n_overflow = n_chain; out->n_elems = h_scal[0]; out->n_elems_file = tot_elems;
  out->max_bucket = 0; out->max_chain = h_max; out->n_entries = h_entries; out->alloc_slots = alloc_slots;
done:
  // (an error return: copies queued on s may still name this frame's host variables - they must have landed before it goes)
  if (rc) hipStreamSynchronize(s);
  if (d_a) hipFree(d_a);
  if (d_cnt) hipFree(d_cnt);
  if (d_cur) hipFree(d_cur);
  if (d_off) hipFree(d_off);
  if (d_coff) hipFree(d_coff);
  if (d_scal) hipFree(d_scal);
  if (d_max) hipFree(d_max);
  if (d_ck) hipFree(d_ck);
  if (d_cm) hipFree(d_cm);
  if (d_dem) hipFree(d_dem);
  if (d_nent) hipFree(d_nent);
  if (d_hll) hipFree(d_hll);
  if (d_ctr) hipFree(d_ctr);
  if (d_sh) hipFree(d_sh);
  if (d_sv) hipFree(d_sv);
  if (d_th) hipFree(d_th);
  if (d_tv) hipFree(d_tv);
  if (d_xh) hipFree(d_xh);
  if (d_xv) hipFree(d_xv);
  if (d_sort_tmp) hipFree(d_sort_tmp);
  if (slots) hipFree(slots);
  if (side) hipFree(side);
  return rc;
}
