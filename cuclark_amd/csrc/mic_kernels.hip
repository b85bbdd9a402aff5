// mic_kernels.hip — the fused k-mer query kernel and its companions, hand-written for gfx950 (wave64).
//
// Replaces queryKernel + queryElement + resultKernel (CuClarkDB.cu:1045-1314, 1421-1471) and
// mergeKernel (:1321-1415).  Design:
//   * one wavefront per read; k-mer t of a 128-k-mer chunk is assembled by lane t%64 in pass t/64 from
//     the packed containers (10 dwords per chunk held one per lane, fetched with ds_bpermute);
//   * canonical k-mer, exact division by the runtime HTSIZE (magic multiply), shard filter;
//   * the probe is transposed so that a QUAD of 4 lanes loads one 64-byte slot (one request per probe);
//     8 such loads per lane are in flight before the first compare;
//   * hits are tallied with ballot/popcount on wave-uniform labels into a register-resident sparse row
//     (one (target,count) entry per lane, 64 entries), best/second are a scalar top-2 over that row;
//   * reads with more than 64 distinct targets, or whose row does not fit the caller's row pitch, are
//     listed for the dense fallback kernels below, which are exact for any read.
#include "mic_internal.h"
#include "mic_device.h"

#include <stdlib.h>

#ifndef MIC_EXP
#define MIC_EXP 0
#endif
#ifndef MIC_FUNNEL64
#define MIC_FUNNEL64 1
#endif
#ifndef MIC_X
#define MIC_X 0          /* measuring builds (tools/ablate_r6.sh): 1 no marker test, 2 no hand-over of crowded runs, 4 no spill of their reads' rows - 7: what the crowded path costs every read */
#endif

namespace {

// Slot loads have no reuse (one random 64-byte request per probe): nontemporal loads keep them from displacing
// the streamed read data in L2 and measured +4 % request rate (profiles/r01_gather_runs_microbench.csv).
__device__ __forceinline__ uint4 load_slot_quarter(const uint4* p) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 t = __builtin_nontemporal_load((const u4*)p);
  return make_uint4(t.x, t.y, t.z, t.w);
}

// k-mer value of nucleotides [nt, nt+k) of a stream whose 32-bit big-endian-in-nt dwords are d0,d1,d2
// starting at dword nt/16.
__device__ __forceinline__ uint64_t kmer_from_dwords(uint32_t d0, uint32_t d1, uint32_t d2, int nt_in_dword, int k) {
  const int s = 2 * nt_in_dword;  // 0..30
  uint64_t a = ((uint64_t)d0 << 32) | d1;
#if MIC_FUNNEL64
  uint64_t x = (a << s) | ((uint64_t)d2 >> (32 - s));      // the 64-bit shift by 32 (s = 0) yields 0: no select
#else
  uint64_t x = s ? ((a << s) | (uint64_t)(d2 >> (32 - s))) : a;
#endif
  return x >> (64 - 2 * k);
}


// ---- software prefetch of the next read --------------------------------------------------------------------------
// Per read the wave needs reads_ptr[r..r+1] -> part header -> first window: three dependent loads before any probe.
// They are issued one (window) and two (pointers) reads ahead, so the current read's probes overlap them.
// The window of the first part is loaded without knowing the part length: callers keep >= 32 readable containers
// after the last one of a batch (include/mi_clark.h), and containers past the part are masked off at use.
struct ReadAhead { uint32_t pp, pe, hdr, w; };   // w = containers (2*lane, 2*lane+1) of the first part, packed

__device__ __forceinline__ void ahead_ptr(const MicQueryArgs& a, uint32_t r, ReadAhead& x) {
  if (r < a.n_reads) { x.pp = a.reads_ptr[r]; x.pe = a.reads_ptr[r + 1]; } else { x.pp = 0; x.pe = 0; }
}

__device__ __forceinline__ void ahead_window(const uint16_t* __restrict__ cont, uint32_t r, uint32_t n_reads, int lane,
                                             ReadAhead& x) {
  x.hdr = 0; x.w = 0;
  if (r < n_reads) {
    const uint32_t pp = __builtin_amdgcn_readfirstlane(x.pp);
    x.hdr = cont[pp];
    if (lane < 10) x.w = ((uint32_t)cont[pp + 1 + 2 * lane] << 16) | cont[pp + 2 + 2 * lane];
  }
}

// window dword of chunk `base` of the part starting at container `first` (cend = one past its last container)
__device__ __forceinline__ uint32_t window_word(const uint16_t* __restrict__ cont, uint32_t first, uint32_t cend,
                                                uint32_t base, int lane, bool use_ahead, const ReadAhead& x) {
  uint32_t w = 0;
  const uint32_t ci = first + base / 8 + 2 * lane;
  if (use_ahead) w = x.w;
  else if (lane < 10) w = ((ci < cend ? (uint32_t)cont[ci] : 0u) << 16) | (ci + 1 < cend ? (uint32_t)cont[ci + 1] : 0u);
  if (lane >= 10 || ci >= cend) w &= 0x0000FFFFu;
  if (lane >= 10 || ci + 1 >= cend) w &= 0xFFFF0000u;
  return w;
}

__device__ __forceinline__ uint32_t window_word_w(const uint16_t* __restrict__ cont, uint32_t first, uint32_t cend,
                                                  uint32_t base, int lane, bool use_ahead, uint32_t ahead_w) {
  uint32_t w = 0;
  const uint32_t ci = first + base / 8 + 2 * lane;
  if (use_ahead) w = ahead_w;
  else if (lane < 10) w = ((ci < cend ? (uint32_t)cont[ci] : 0u) << 16) | (ci + 1 < cend ? (uint32_t)cont[ci + 1] : 0u);
  if (lane >= 10 || ci >= cend) w &= 0x0000FFFFu;
  if (lane >= 10 || ci + 1 >= cend) w &= 0xFFFF0000u;
  return w;
}

struct Probe {  // what a pass lane knows about its k-mer
  uint32_t slot;  // slot index relative to the shard, 0xFFFFFFFF = nothing to probe
  uint32_t qlo, qhi;
};

template <bool KEY64>
__device__ __forceinline__ Probe make_probe(const MicTable& t, uint64_t kmer, bool active) {
  uint64_t c = canonical(kmer, t.k);
  uint64_t q = mic_div(c, t.div);
  uint64_t rem = c - q * t.div.d;
  Probe p;
  bool ok = active && rem >= t.shard_start && rem < t.shard_end;  // CuClarkDB.cu:1272-1274
  // the reference compares the full-width quotient with the stored key (CuClarkDB.cu:1291-1298): a
  // quotient that does not fit the 32-bit slot class can never be equal to one
  if (!KEY64) ok = ok && (q >> 32) == 0;
  p.slot = ok ? (uint32_t)(rem - t.shard_start) : 0xFFFFFFFFu;
  p.qlo = (uint32_t)q; p.qhi = (uint32_t)(q >> 32);
  return p;
}

// Compare one loaded quarter against the quotient.  Returns label+1 in the hitting lane, else 0.
template <bool KEY64>
__device__ __forceinline__ uint32_t match_quarter(const uint4& q, uint32_t qlo, uint32_t qhi, int j) {
  const uint32_t n = q.w & 0xFF;
  if constexpr (KEY64) {
    bool hit = (uint32_t)j < n && q.x == qlo && q.y == qhi;
    return hit ? (q.z & 0xFFFF) + 1 : 0;
  } else {
    bool ha = (uint32_t)(2 * j) < n && q.x == qlo;
    bool hb = (uint32_t)(2 * j + 1) < n && q.y == qlo;
    uint32_t lab = ha ? (q.z & 0xFFFF) : (q.z >> 16);
    return (ha | hb) ? lab + 1 : 0;
  }
}

// Follow the overflow chain for quads whose bucket holds more than one slot (rare).  `res` is the
// quad-reduced result so far (label+1 or 0, identical in the 4 lanes).
template <bool KEY64>
__device__ __forceinline__ uint32_t chase_chain(const uint4* __restrict__ slots, uint4 q, uint32_t qlo, uint32_t qhi,
                                                int j, uint32_t res, bool probing) {
  constexpr uint32_t CAP = KEY64 ? MIC_CAP64 : MIC_CAP32;
  for (;;) {
    const uint32_t n = q.w & 0xFF;
    // largest key of this slot lives in lane 3 of the quad (x,y for 64-bit keys; y for 32-bit keys)
    uint32_t lk_lo = quad_perm<QP_BCAST3>(KEY64 ? q.x : q.y);
    uint32_t lk_hi = KEY64 ? quad_perm<QP_BCAST3>(q.y) : 0;
    uint64_t lastk = ((uint64_t)lk_hi << 32) | lk_lo;
    uint64_t qq = ((uint64_t)(KEY64 ? qhi : 0) << 32) | qlo;
    uint32_t w0 = quad_perm<QP_BCAST0>(q.w), w1 = quad_perm<QP_BCAST1>(q.w);
    uint64_t next = (uint64_t)(w0 >> 8) | ((uint64_t)(w1 >> 8) << 24);
    bool more = probing && n > CAP && res == 0 && qq > lastk;
    if (__ballot(more) == 0) break;
    q = make_uint4(0, 0, 0, 0);
    if (more) q = load_slot_quarter(slots + next * 4 + j);
    uint32_t m = more ? match_quarter<KEY64>(q, qlo, qhi, j) : 0;
    m |= quad_perm<QP_XOR1>(m);
    m |= quad_perm<QP_XOR2>(m);
    res |= m;
    probing = more;
  }
  return res;
}

// Register-resident sparse row of one read: lane i holds entry i.
struct RowAcc {
  uint32_t label1;  // label+1, 0 = empty
  uint32_t count;
};

__device__ __forceinline__ void row_add(RowAcc& acc, uint32_t& n_ent, uint32_t& overflow, uint32_t l1, uint32_t cnt,
                                        int lane) {
  uint64_t f = __ballot(acc.label1 == l1);
  if (f) {
    if (acc.label1 == l1) acc.count += cnt;
  } else if (n_ent < 64) {
    if ((uint32_t)lane == n_ent) { acc.label1 = l1; acc.count = cnt; }
    ++n_ent;
  } else {
    overflow = 1;
  }
}

// Tally the two result registers of a chunk (each lane: label+1 or 0) into the row.
__device__ __forceinline__ void tally2(uint32_t r0, uint32_t r1, RowAcc& acc, uint32_t& n_ent, uint32_t& overflow,
                                       uint32_t& total, int lane) {
  uint64_t m0 = __ballot(r0 != 0), m1 = __ballot(r1 != 0);
  total += __popcll(m0) + __popcll(m1);
  while (m0 | m1) {
    uint32_t l1;
    if (m0) l1 = __builtin_amdgcn_readlane(r0, __builtin_ctzll(m0));
    else l1 = __builtin_amdgcn_readlane(r1, __builtin_ctzll(m1));
    uint64_t e0 = __ballot(r0 == l1), e1 = __ballot(r1 == l1);
    m0 &= ~e0; m1 &= ~e1;
    row_add(acc, n_ent, overflow, l1, __popcll(e0) + __popcll(e1), lane);
  }
}

// the same over three arrays (the third holds the hits found below the first tree level)
__device__ __forceinline__ void tally3(uint32_t r0, uint32_t r1, uint32_t r2, RowAcc& acc, uint32_t& n_ent, uint32_t& overflow,
                                       uint32_t& total, int lane) {
  uint64_t m0 = __ballot(r0 != 0), m1 = __ballot(r1 != 0), m2 = __ballot(r2 != 0);
  total += __popcll(m0) + __popcll(m1) + __popcll(m2);
  while (m0 | m1 | m2) {
    uint32_t l1;
    if (m0) l1 = __builtin_amdgcn_readlane(r0, __builtin_ctzll(m0));
    else if (m1) l1 = __builtin_amdgcn_readlane(r1, __builtin_ctzll(m1));
    else l1 = __builtin_amdgcn_readlane(r2, __builtin_ctzll(m2));
    uint64_t e0 = __ballot(r0 == l1), e1 = __ballot(r1 == l1), e2 = __ballot(r2 == l1);
    m0 &= ~e0; m1 &= ~e1; m2 &= ~e2;
    row_add(acc, n_ent, overflow, l1, __popcll(e0) + __popcll(e1) + __popcll(e2), lane);
  }
}

// Scalar top-2 over the row under the order (count desc, target asc) — equivalent to resultKernel's
// ascending scan with strict '>' (CuClarkDB.cu:1440-1459), see DESIGN.md §4.
template <typename ARGS>   // MicQueryArgs, or the cold fields re-read from the kernarg segment (query_kernel_m)
__device__ __forceinline__ void finish_read(const RowAcc& acc, uint32_t n_ent, uint32_t total, uint32_t overflow,
                                            uint32_t r, const ARGS& a, int lane) {
  // (count, label + 1) pairs compared as 32-bit scalars: a 64-bit key has no scalar compare and went through the vector unit
  uint32_t bc = 0, bl = 0, sc = 0, sl = 0;       // best and second: count, label + 1 (0 = none)
  // A row of at most one entry (most reads): the row is written by lane 0, and lane 0 HOLDS entry 0 (or zeros) - best label and
  // count go into the result row straight from its registers, no v_readlane into scalars and v_mov back (round 6: -4 VALU per read)
  uint32_t vbl = acc.label1, vbc = acc.count;
  if (n_ent > 1) {
    // (the first entry is the best so far without a comparison - its count is at least 1 - and the comparisons are scalar work)
    bl = __builtin_amdgcn_readlane(acc.label1, 0); bc = __builtin_amdgcn_readlane(acc.count, 0);
    for (uint32_t i = 1; i < n_ent; ++i) {
      const uint32_t l1 = __builtin_amdgcn_readlane(acc.label1, i);
      const uint32_t c = __builtin_amdgcn_readlane(acc.count, i);
      const bool over_best = c > bc || (c == bc && l1 < bl);
      const bool over_second = c > sc || (c == sc && l1 < sl);
      if (over_best) { sc = bc; sl = bl; bc = c; bl = l1; }
      else if (over_second) { sc = c; sl = l1; }
    }
    vbl = bl; vbc = bc;
  }
  uint32_t flags = 0;
  if (a.rows) {
    uint32_t* row = a.rows + (size_t)r * a.row_words;
    bool fits = n_ent <= a.row_words - 1 && !overflow;
    uint32_t rank = 0;
    for (uint32_t i = 0; i < n_ent; ++i) {
      uint32_t li = __builtin_amdgcn_readlane(acc.label1, i);
      rank += li < acc.label1;
    }
    bool big = (uint32_t)lane < n_ent && acc.count > 0xFFFF;
    if (__ballot(big)) fits = false;
    if (fits) {
      if ((uint32_t)lane < n_ent) row[1 + rank] = (acc.count << 16) | (acc.label1 - 1);
      if (lane == 0) row[0] = n_ent;
    } else {
      if (lane == 0) row[0] = MIC_ROW_INVALID;
      flags |= MIC_FLAG_ROW_OVERFLOW_;
    }
  }
  if (overflow) flags |= MIC_FLAG_ROW_OVERFLOW_;
  if (lane == 0) {
    uint4 lo, hi;
    lo.x = total;
    lo.y = vbl;   // label+1
    lo.z = vbc;
    lo.w = sl;
    hi.x = sc;
    hi.y = n_ent; hi.z = flags; hi.w = 0;
    uint4* out = (uint4*)(a.results + (size_t)r * 8);
    out[0] = lo; out[1] = hi;
  }
  // wave-uniform branch: the wait for the atomic's return value must not sit on the common path
#if MIC_EXP & 2
  if (__builtin_expect(__builtin_amdgcn_readfirstlane(flags) != 0, 0)) if (a.flagged) {
#else
  if (__builtin_amdgcn_readfirstlane(flags) && a.flagged) {
#endif
    if (lane == 0) {
      uint32_t pos = atomicAdd(&a.flagged[0], 1u);
      if (pos < a.flagged_cap) a.flagged[1 + pos] = r;
    }
  }
}

template <bool KEY64>
__global__ void __launch_bounds__(256) query_kernel(const MicQueryArgs a) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 3;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const uint32_t n_waves = gridDim.x * 4;
  const MicTable& t = a.t;
  const int k = t.k;
  const uint4* __restrict__ slots = t.slots;
  const uint16_t* __restrict__ cont = a.cont;

  for (uint32_t r = wave0; r < a.n_reads; r += n_waves) {
    ReadAhead cur;   // header and first window are fetched together (one latency instead of two)
    ahead_ptr(a, r, cur);
    ahead_window(cont, r, a.n_reads, lane, cur);
    uint32_t pp = __builtin_amdgcn_readfirstlane(cur.pp);
    const uint32_t pe = __builtin_amdgcn_readfirstlane(cur.pe);
    RowAcc acc; acc.label1 = 0; acc.count = 0;
    uint32_t n_ent = 0, overflow = 0, total = 0;
    bool first_part = true;

    while (pp < pe) {  // parts of the read (CuClarkDB.cu:1090-1097)
      const uint32_t plen = __builtin_amdgcn_readfirstlane(first_part ? cur.hdr : (uint32_t)cont[pp]);
      const bool ahead_ok = first_part;
      first_part = false;
      if (plen == 0) break;
      const uint32_t first = pp + 1;
      pp = first + (plen + 7) / 8;
      if (plen < (uint32_t)k) continue;
      const uint32_t nk = plen - k + 1;
      const uint32_t cend = pp;
      for (uint32_t base = 0; base < nk; base += 128) {
        // 10 dwords (160 nt) cover the 128 + k - 1 nucleotides of this chunk; lane i holds dword i
        const uint32_t w = window_word(cont, first, cend, base, lane, ahead_ok && base == 0, cur);
        Probe pr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = 4 * h + (lane >> 4);
          uint32_t d0 = bperm(idx, w), d1 = bperm(idx + 1, w), d2 = bperm(idx + 2, w);
          uint64_t kmer = kmer_from_dwords(d0, d1, d2, lane & 15, k);
          pr[h] = make_probe<KEY64>(t, kmer, base + 64 * h + lane < nk);
        }
        // transpose: sub-pass s serves k-mers 16*(s&3) .. +15 of pass s>>2, one quad per k-mer
        uint4 q[8]; uint32_t qlo[8], qhi[8]; uint32_t sl[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int src = 16 * (s & 3) + (lane >> 2);
          sl[s] = bperm(src, pr[s >> 2].slot);
          qlo[s] = bperm(src, pr[s >> 2].qlo);
          qhi[s] = KEY64 ? bperm(src, pr[s >> 2].qhi) : 0;
          q[s] = make_uint4(0, 0, 0, 0);
          if (sl[s] != 0xFFFFFFFFu) q[s] = load_slot_quarter(slots + (uint64_t)sl[s] * 4 + j);
        }
        uint32_t res0 = 0, res1 = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          uint32_t m = match_quarter<KEY64>(q[s], qlo[s], qhi[s], j);
          m |= quad_perm<QP_XOR1>(m);
          m |= quad_perm<QP_XOR2>(m);
          constexpr uint32_t CAP = KEY64 ? MIC_CAP64 : MIC_CAP32;
          if (__ballot((q[s].w & 0xFF) > CAP))
            m = chase_chain<KEY64>(slots, q[s], qlo[s], qhi[s], j, m, sl[s] != 0xFFFFFFFFu);
          // lane j of the quad keeps the result of sub-pass with (s&3)==j
          if (s < 4) res0 = (j == (s & 3)) ? m : res0;
          else res1 = (j == (s & 3)) ? m : res1;
        }
        tally2(res0, res1, acc, n_ent, overflow, total, lane);
      }
    }
    finish_read(acc, n_ent, total, overflow, r, a, lane);
  }
}


// =====================================================================================================================
// query_kernel_m — minimizer-keyed table (layout 1).  Same work mapping, k-mer assembly, tally and result code as
// query_kernel; the probe differs: consecutive k-mers that share a minimizer share a slot, so per 128-k-mer chunk the
// wave (1) computes every k-mer's slot from a sliding minimum of 32-bit m-mer order keys, (2) finds the runs of equal
// slots with ballot/popcount, (3) loads each distinct 128-byte slot ONCE (8 lanes x 16 B, up to 8 slots per
// wave-instruction) and stages it in the wave's private LDS region, (4) lets every k-mer compare its canonical value
// against the 12 staged keys, (5) repeats for the lanes whose chain continues.  HBM sees one 128-byte request per
// distinct slot (~17 per 150-bp read) instead of one 64-byte request per k-mer (120).
// =====================================================================================================================
#define MIC_RMAX 32        // slots staged per round
#define MIC_MSTRIDE 8      // uint4 per staged slot in LDS (linear: LDS-DMA writes lane L at base + 16*L)
// Staged slots lie 128 bytes apart inside the 8 slots of one LDS-DMA instruction - all on two banks for a given word.  Each
// group of 8 therefore starts MIC_R_SKEW uint4 further: reads of the same word of different slots spread over four times as
// many banks (per-run kernel: 16-way -> 4-way conflicts, +4.5 %).
#ifndef MIC_R_SKEW
#define MIC_R_SKEW 2
#endif
#ifndef MIC_R_PIPE
#define MIC_R_PIPE 2     /* query_kernel_r software-pipelined across reads: 0 no instantiation, 1 the two-strand table's, 2 the one-strand table's too */
#endif
__device__ __forceinline__ uint32_t staged_at(uint32_t i) { return i * MIC_MSTRIDE + (i >> 3) * MIC_R_SKEW; }   // uint4 offset of staged slot i
// the minimizer kernel reads its staged keys as 64-bit words and loses 5 % with the skew (11.5 against 11.0 ms): none there
#ifndef MIC_M_SKEW
#define MIC_M_SKEW 0
#endif
__device__ __forceinline__ uint32_t staged_at_m(uint32_t i) { return i * MIC_MSTRIDE + (i >> 3) * MIC_M_SKEW; }

__device__ __forceinline__ void sliding_min3(uint32_t& a0, uint32_t& a1, uint32_t& a2, int w, int lane) {
  // a_h(l) holds the order key of position 64h+l; afterwards a_h(l) = min over positions [64h+l, 64h+l+w)
  auto step = [&](int s) {
    const int src = (lane + s) & 63;
    const bool wrap = lane + s >= 64;
    uint32_t x0 = bperm(src, a0), x1 = bperm(src, a1), x2 = bperm(src, a2);
    uint32_t n0 = wrap ? x1 : x0, n1 = wrap ? x2 : x1, n2 = wrap ? 0xFFFFFFFFu : x2;
    a0 = n0 < a0 ? n0 : a0; a1 = n1 < a1 ? n1 : a1; a2 = n2 < a2 ? n2 : a2;
  };
  int cover = 1;
  while (2 * cover <= w) { step(cover); cover *= 2; }
  if (cover < w) step(w - cover);
}

// Two-array form: the m-mers past position 127 are not a third array but arrive as `tail` (see the kernel).
// (Fetching the last two window starts - shifts 4 and 8 for w = 12 - in one LDS round trip instead of a fourth step:
// measured, no gain.)
__device__ __forceinline__ void sliding_min2(uint32_t& a0, uint32_t& a1, int w, int lane) {
  auto step = [&](int s) {
    const int src = (lane + s) & 63;
    const bool wrap = lane + s >= 64;
    uint32_t x0 = bperm(src, a0), x1 = bperm(src, a1);
    uint32_t n0 = wrap ? x1 : x0, n1 = wrap ? 0xFFFFFFFFu : x1;
    a0 = n0 < a0 ? n0 : a0; a1 = n1 < a1 ? n1 : a1;
  };
  int cover = 1;
  while (2 * cover <= w) { step(cover); cover *= 2; }
  if (cover < w) step(w - cover);
}

__device__ __forceinline__ uint32_t row_prefix_min(uint32_t t) {
  uint32_t x;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x111, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x112, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x114, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x118, 0xF, 0xF, false); t = x < t ? x : t;
  return t;
}

__device__ __forceinline__ uint32_t row_suffix_min(uint32_t t) {
  uint32_t x;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x101, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x102, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x104, 0xF, 0xF, false); t = x < t ? x : t;
  x = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)t, 0x108, 0xF, 0xF, false); t = x < t ? x : t;
  return t;
}

// Sliding minimum over windows of 16 <= W <= 32 positions, positions 64 h + lane in a_h: afterwards a0, a1 = min over
// [position, position + W).  Rows of 16 lanes are scanned from both sides by DPP (full-rate v_min_u32_dpp, no LDS):
// P = prefix minimum, S = suffix minimum inside the row; a window of 16 that starts at x is S(x) and P(x + 15) (the next
// row, or the same one when x starts a row), and a window of W is two windows of 16, at x and at x + W - 16.  Two rounds of
// ds_bpermute instead of the five of the doubling form (sliding_min3), which serves W < 16.
template <int BANKS>
__device__ __forceinline__ uint32_t with_row_in_front(uint32_t P) {     // min(P, lane 15 of the row in front) in the lanes of BANKS, rows 1-3
  const uint32_t f = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)P, 0x142, 0xE, BANKS, false);      // row_bcast:15
  return f < P ? f : P;
}
__device__ __forceinline__ void sliding_min_rows(uint32_t& a0, uint32_t& a1, uint32_t a2, int W, int lane, int lane_inv) {
  const uint32_t P0 = row_prefix_min(a0), P1 = row_prefix_min(a1), P2 = row_prefix_min(a2);
  if (W > 16 && ((W - 16) & 3) == 0) {
    // One round of ds_bpermute: the window [x, x + W) is the rest of x's row, S(x), the prefix P(e) of the row of its last
    // position e = x + W - 1, and - when that row is the next but one, which is when e sits in lanes 0 .. W-18 of its row -
    // the whole row in between.  That row's minimum is lane 15 of its P: DPP row_bcast:15 hands it to the next row, the bank
    // mask cuts it to lanes 0 .. W-17 (lane W-17 belongs to a window that starts its row: the row in front is part of it
    // anyway); the first row of arrays 1 and 2 gets lane 63 of the array in front through a scalar.
    const uint32_t S0 = row_suffix_min(a0), S1 = row_suffix_min(a1);
    const int s = W - 16, bm = (1 << (s >> 2)) - 1;
    uint32_t Q0, Q1, Q2;
    // (the bank mask is an immediate of the instruction)
    switch (bm) {
      case 1: Q0 = with_row_in_front<1>(P0); Q1 = with_row_in_front<1>(P1); Q2 = with_row_in_front<1>(P2); break;
      case 3: Q0 = with_row_in_front<3>(P0); Q1 = with_row_in_front<3>(P1); Q2 = with_row_in_front<3>(P2); break;
      case 7: Q0 = with_row_in_front<7>(P0); Q1 = with_row_in_front<7>(P1); Q2 = with_row_in_front<7>(P2); break;
      default: Q0 = with_row_in_front<15>(P0); Q1 = with_row_in_front<15>(P1); Q2 = with_row_in_front<15>(P2); break;
    }
    const uint32_t f0 = __builtin_amdgcn_readlane(P0, 63), f1 = __builtin_amdgcn_readlane(P1, 63);
    const bool head = lane < s;
    const uint32_t h1 = f0 < Q1 ? f0 : Q1, h2 = f1 < Q2 ? f1 : Q2;
    Q1 = head ? h1 : Q1; Q2 = head ? h2 : Q2;
    const int at = (lane_inv << 2) + ((W - 1) << 2);
    const uint32_t z0 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)Q0), z1 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)Q1),
                   z2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)Q2);
    const bool wz = lane >= 65 - W;                               // lane + W - 1 >= 64
    const uint32_t v0 = wz ? z1 : z0, v1 = wz ? z2 : z1;
    a0 = v0 < S0 ? v0 : S0; a1 = v1 < S1 ? v1 : S1;
    return;
  }
  const uint32_t S0 = row_suffix_min(a0), S1 = row_suffix_min(a1), S2 = row_suffix_min(a2);
  const int at = lane_inv << 2;                                   // ds_bpermute wraps the lane number by itself
  const uint32_t x0 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + 60, (int)P0), x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + 60, (int)P1),
                 x2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + 60, (int)P2);
  const bool wr = lane >= 49;
  const uint32_t n0 = wr ? x1 : x0, n1 = wr ? x2 : x1;
  uint32_t M0 = n0 < S0 ? n0 : S0, M1 = n1 < S1 ? n1 : S1;
  if (W > 16) {
    const uint32_t M2 = x2 < S2 ? x2 : S2;                        // exact for lanes < 49: lanes < W - 16 are read
    const int sh = (W - 16) << 2;
    const uint32_t y0 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + sh, (int)M0), y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + sh, (int)M1),
                   y2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at + sh, (int)M2);
    const bool wy = lane >= 80 - W;                               // lane + W - 16 >= 64
    const uint32_t v0 = wy ? y1 : y0, v1 = wy ? y2 : y1;
    M0 = v0 < M0 ? v0 : M0; M1 = v1 < M1 ? v1 : M1;
  }
  a0 = M0; a1 = M1;
}

// Front half of the super-k-mer kernels: where in the chunk the SAMPLED m-mer (mic_device.h: mod-sampling) of the k-mers at
// chunk positions lane and 64 + lane sits.  wd: window dword `lane` of the chunk (16 nucleotides); `past`: windows of counted
// k-mers reach t-mers behind position 127 (a third pass of keys).  CANON: one-strand table, canonical t-mers.
// ln: the lane number as an opaque per-chunk copy (what is derived from it - compare masks - is recomputed per chunk: kept across
// the kernel they would live in scalar register pairs the per-run kernel does not have); lane_inv: the lane number the compiler
// may hoist from (vector registers are plentiful: shift counts and permute addresses stay resident, 8 VALU per chunk fewer).
template <bool CANON>
__device__ __forceinline__ void sampled_positions(uint32_t wd, int ln, int lane_inv, int k, int m, bool past, uint32_t& qa0, uint32_t& qa1) {
  const int w = k - m + 1, t = s_tlen(k, m), W = k - t + 1;
  const uint32_t pb = (uint32_t)lane_inv & 31u;                     // chunks start at multiples of 128: position & 31 = lane & 31 in every pass
  uint32_t a0, a1, a2 = 0xFFFFFFFFu;
  if (t <= 16) {
    // the t-mer at position P = 64 h + lane: 16 nucleotides from P on are one funnel shift of window dwords (P - 1) / 16 and
    // the next (the shift stays below 32 this way; P = 0 reads lane 63 and shifts it out)
    const int s1 = lane_inv - 1;
    const int ad = (s1 >> 4) << 2;
    const uint32_t tsh = 30u - 2u * (uint32_t)(s1 & 15);
    auto tkey = [&](int h) {
      const uint32_t W0 = (uint32_t)__builtin_amdgcn_ds_bpermute(ad + 16 * h, (int)wd), W1 = (uint32_t)__builtin_amdgcn_ds_bpermute(ad + 16 * h + 4, (int)wd);
      uint32_t tv = __builtin_amdgcn_alignbit(W0, W1, tsh) >> (32 - 2 * t);
      if (CANON) { const uint32_t tr = revcomp_bits32(tv, t); tv = tr < tv ? tr : tv; }
      return ((t <= 12 ? s_torder24(tv) : s_torder(tv)) & ~31u) | pb;
    };
    a0 = tkey(0); a1 = tkey(1);
    if (past) a2 = tkey(2);
  } else {
    auto tkey = [&](int h) {
      const int idx = 4 * h + (lane_inv >> 4);
      const uint32_t d0 = bperm(idx, wd), d1 = bperm(idx + 1, wd), d2 = bperm(idx + 2, wd);
      uint64_t tv = kmer_from_dwords(d0, d1, d2, lane_inv & 15, t);
      if (CANON) { const uint64_t tr = revcomp_bits(tv, t); tv = tr < tv ? tr : tv; }
      return (s_torder(tv) & ~31u) | pb;
    };
    a0 = tkey(0); a1 = tkey(1);
    if (past) a2 = tkey(2);
  }
  if (W >= 16) sliding_min_rows(a0, a1, a2, W, ln, lane_inv);
  else sliding_min3(a0, a1, a2, W, ln);
  // position of the minimal t-mer inside the k-mer, then of the sampled m-mer: i mod w
  uint32_t d0 = (a0 - pb) & 31u, d1 = (a1 - pb) & 31u;
  if (W == 2 * w) { const uint32_t e0 = d0 - (uint32_t)w, e1 = d1 - (uint32_t)w; d0 = e0 < d0 ? e0 : d0; d1 = e1 < d1 ? e1 : d1; }
  else if (W != w) {
    const uint32_t rcp = (1024u + (uint32_t)w - 1u) / (uint32_t)w;      // d < 32, w <= 16: (d * rcp) >> 10 = d / w exactly
    d0 -= (uint32_t)w * ((d0 * rcp) >> 10); d1 -= (uint32_t)w * ((d1 * rcp) >> 10);
  }
  qa0 = (uint32_t)lane_inv + d0; qa1 = 64u + (uint32_t)lane_inv + d1;
}

#ifndef MIC_SPEC
#define MIC_SPEC 1
#endif
#ifndef MIC_M_WPB
#define MIC_M_WPB 4        // waves per block of query_kernel_m (measured: 1: 778, 2: 757, 4: 824-834 Mreads/s; tools/wpb_sweep.sh)
#endif
// KK / MM: k and the minimizer length as compile-time constants (0 = take them from the table): the launcher picks the
// instantiation for cuCLARK's k = 31 and cuCLARK-l's k = 27 with m = 20; shift counts, masks and the window loop fold.
template <int KK, int MM>
__global__ void __launch_bounds__(64 * MIC_M_WPB, 32 / MIC_M_WPB) query_kernel_m(const MicQueryArgs a) {
  __shared__ uint4 s_stage[MIC_M_WPB][MIC_RMAX * MIC_MSTRIDE + (MIC_RMAX / 8 - 1) * MIC_M_SKEW];
  __shared__ uint32_t s_run[MIC_M_WPB][MIC_RMAX];
  __shared__ uint32_t s_ahead[MIC_M_WPB][2][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: LDS bases stay in SGPRs
  uint4* stage = s_stage[wv];
  uint32_t* runslot = s_run[wv];
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * MIC_M_WPB + wv);
  const uint32_t n_waves = gridDim.x * MIC_M_WPB;
  const MicTable& t = a.t;
  const int k = KK ? KK : t.k, m = MM ? MM : t.m, w = k - m + 1;
  const uint4* __restrict__ slots = t.slots;
  const uint16_t* __restrict__ cont = a.cont;
  // number of set bits of a 64-bit lane mask below this lane (v_mbcnt_lo/hi: no mask register to keep alive)
  auto below = [](uint64_t mask) { return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };

  // Read-ahead through LDS (global_load_lds_dword): no VGPR lives across a read and no load result is touched near its
  // issue, so nothing waits for it.  One DMA instruction per read fetches {header + first window of read j+2 (lanes
  // 0..11, aligned dwords), reads_ptr of read j+3 (lanes 12, 13)}; it is taken one read later, before the result stores
  // of finish_read (stores count in vmcnt too - waiting for the entry after them would wait for their acknowledgement).
  uint32_t* ahead0 = s_ahead[wv][0];
  uint32_t* ahead1 = s_ahead[wv][1];
  auto ahead_issue = [&](uint32_t* entry, uint32_t pp_w, uint32_t r_ptr) {
    // lanes 0..11: aligned dwords of the window; lanes 12, 13: reads_ptr[rr], reads_ptr[rr + 1]; the rest repeat lane 0.
    // One select between two wave-uniform bases, then base + 4 * lane (the pointer base is pre-biased by -48).
    const uint64_t abase = ((uint64_t)(cont + pp_w)) & ~3ULL;
    const uint32_t rr = r_ptr < a.n_reads ? r_ptr : a.n_reads - 1;
    const uint64_t pbase = (uint64_t)(a.reads_ptr + rr) - 48;
    uint32_t lv = (uint32_t)lane;
    asm volatile("" : "+v"(lv));      // recomputed per read (3 VALU) instead of a 64-bit offset kept live in VGPRs
    const uint32_t li = lv < 14 ? lv : 0u;
    const uint64_t addr = (li < 12 ? abase : pbase) + 4 * li;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr,
                                     (__attribute__((address_space(3))) void*)entry, 4, 0, 0);
  };
  auto ahead_take = [&](const uint32_t* entry, uint32_t pp_w, uint32_t& hdr, uint32_t& wword, uint32_t& npp, uint32_t& npe) {
    const uint32_t raw = entry[lane];
    npp = __builtin_amdgcn_readlane(raw, 12); npe = __builtin_amdgcn_readlane(raw, 13);
    const bool odd = (((uint64_t)(cont + pp_w)) >> 1) & 1;      // is container pp_w the high half of its dword?
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x101, 0xF, 0xF, true);   // row_shl:1 = lane+1
    const uint32_t first = __builtin_amdgcn_readfirstlane(raw);
    hdr = odd ? first >> 16 : first & 0xFFFFu;
    // window word of lane L = (container pp+1+2L) << 16 | container pp+2+2L
    wword = odd ? ((up << 16) | (up >> 16)) : ((raw & 0xFFFF0000u) | (up & 0xFFFFu));
  };
  uint32_t cur_pp, cur_pe, cur_hdr, cur_w;   // read r: pointers, first part header, first window
  uint32_t n_pp, n_pe;                       // pointers of read r + n_waves
  uint32_t ahead_sel;
  {
    const uint32_t r0 = wave0 < a.n_reads ? wave0 : a.n_reads - 1;
    cur_pp = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0]); cur_pe = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0 + 1]);
    ahead_issue(ahead0, cur_pp, wave0 + n_waves);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead0, cur_pp, cur_hdr, cur_w, n_pp, n_pe);
    ahead_issue(ahead1, n_pp, wave0 + 2 * n_waves);
    ahead_sel = 1;
  }
  for (uint32_t r = wave0; r < a.n_reads; r += n_waves) {
    uint32_t pp = cur_pp;
    const uint32_t pe = cur_pe;
    RowAcc acc; acc.label1 = 0; acc.count = 0;
    uint32_t n_ent = 0, overflow = 0, total = 0;
    bool first_part = true;

    while (pp < pe) {
      const uint32_t plen = __builtin_amdgcn_readfirstlane(first_part ? cur_hdr : (uint32_t)cont[pp]);
      const bool ahead_ok = first_part;
      first_part = false;
      if (plen == 0) break;
      const uint32_t first = pp + 1;
      pp = first + (plen + 7) / 8;
      if (plen < (uint32_t)k) continue;
      const uint32_t nk = plen - k + 1;
      const uint32_t cend = pp;
      for (uint32_t base = 0; base < nk; base += 128) {
        const uint32_t wd = window_word_w(cont, first, cend, base, lane, ahead_ok && base == 0, cur_w);
        // k-mers of the two passes and the order keys of the m-mers at positions base+64h+lane, h = 0..2
        uint64_t c[2]; bool act[2]; uint32_t hk0, hk1, hk2;
        // The windows of the chunk's last k-mers reach m-mers at positions base+128 .. base+128+w-2.  m-mer 128+j is the
        // LAST m-mer of k-mer 64 + (65-w+j) of pass 1, so for w <= 16 its order key comes out of that lane's k-mer and
        // reverse complement (no third assembly pass) and the minimum over positions 128 .. 64+lane+w-1 is a prefix
        // minimum inside the last DPP row.  A chunk of at most 129-w k-mers (every 100/125-bp read) needs neither.
        const bool past = nk - base > (uint32_t)(129 - w);
        const bool tail_path = w <= 16;
        uint32_t tail = 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = 4 * h + (lane >> 4);
          uint32_t d0 = bperm(idx, wd), d1 = bperm(idx + 1, wd), d2 = bperm(idx + 2, wd);
          uint64_t kmer = kmer_from_dwords(d0, d1, d2, lane & 15, k);
          const uint64_t rck = revcomp_bits(kmer, k);
          c[h] = kmer < rck ? kmer : rck;
          act[h] = base + 64 * h + lane < nk;
          if (t.sharded) {   // table-sharded mode only: divisor and bounds are re-read from the kernarg segment (see finish)
            uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
            MicDiv dv; dv.d = kc->t.div.d; dv.magic = kc->t.div.magic; dv.shift = kc->t.div.shift; dv.add = kc->t.div.add;
            const uint64_t s_lo = kc->t.shard_start, s_hi = kc->t.shard_end;
            uint64_t q = mic_div(c[h], dv);
            uint64_t rem = c[h] - q * dv.d;
            act[h] = act[h] && rem >= s_lo && rem < s_hi;
          }
          // m-mer at this position = first m nt of the k-mer; its reverse complement = last m nt of rc(k-mer)
          const uint64_t mf = kmer >> (2 * (k - m)), mr = rck & ((1ULL << (2 * m)) - 1);
          uint32_t key = mmer_order_key_canon(mf < mr ? mf : mr);
          if (h == 0) hk0 = key; else hk1 = key;
          if (h == 1 && tail_path && past) {
            // last m-mer of the k-mer = its last m nt; reverse complement = first m nt of rc(k-mer)
            const uint64_t tf = kmer & ((1ULL << (2 * m)) - 1), tr = rck >> (2 * (k - m));
            const uint32_t tk = mmer_order_key_canon(tf < tr ? tf : tr);
            tail = row_prefix_min(lane >= 65 - w ? tk : 0xFFFFFFFFu);
          }
        }
        if (tail_path) {
          sliding_min2(hk0, hk1, w, lane);
          hk1 = tail < hk1 ? tail : hk1;
        } else {
          hk2 = 0xFFFFFFFFu;
          if (past) {
            const int idx = 8 + (lane >> 4);
            uint32_t d0 = bperm(idx, wd), d1 = bperm(idx + 1, wd), d2 = bperm(idx + 2, wd);
            uint64_t mm = kmer_from_dwords(d0, d1, d2, lane & 15, m);
            hk2 = lane < w - 1 ? mmer_order_key(mm, m) : 0xFFFFFFFFu;
          }
          sliding_min3(hk0, hk1, hk2, w, lane);
        }
        uint32_t sl0 = act[0] ? mslot_of_key(hk0, (uint32_t)t.n_main) : 0xFFFFFFFFu;
        uint32_t sl1 = act[1] ? mslot_of_key(hk1, (uint32_t)t.n_main) : 0xFFFFFFFFu;
        uint32_t res0 = 0, res1 = 0, res2 = 0;   // label + 1 of the hit: passes 0, 1 and the compacted deeper levels

        // one level of the table on both passes: runs of equal slots, LDS-DMA of the distinct slots, lockstep search
        auto level = [&](uint32_t s0_, uint32_t s1_, uint64_t k0_, uint64_t k1_, uint32_t& o0_, uint32_t& o1_, uint32_t& y0_, uint32_t& y1_) {
          // runs of equal slots over the 128 positions
          uint32_t p0 = bperm((lane + 63) & 63, s0_), p1 = bperm((lane + 63) & 63, s1_);
          uint32_t last0 = bperm(63, s0_);
          if (lane == 0) { p0 = 0xFFFFFFFFu; p1 = last0; }
          const bool f0 = s0_ != 0xFFFFFFFFu && s0_ != p0, f1 = s1_ != 0xFFFFFFFFu && s1_ != p1;
          const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
          const uint32_t R0 = __popcll(b0), R = R0 + __popcll(b1);
          const uint32_t rid0 = below(b0) + (f0 ? 1u : 0u) - 1, rid1 = R0 + below(b1) + (f1 ? 1u : 0u) - 1;
          y0_ = 0xFFFFFFFFu; y1_ = 0xFFFFFFFFu;   // slots to probe at the next level
          for (uint32_t rbase = 0; rbase < R; rbase += MIC_RMAX) {
            __builtin_amdgcn_wave_barrier();
            if (f0 && rid0 - rbase < MIC_RMAX) runslot[rid0 - rbase] = s0_;
            if (f1 && rid1 - rbase < MIC_RMAX) runslot[rid1 - rbase] = s1_;
            __builtin_amdgcn_wave_barrier();
            const uint32_t nrun = R - rbase < MIC_RMAX ? R - rbase : MIC_RMAX;
            // every distinct slot goes HBM -> LDS directly (global_load_lds_dwordx4: lane L lands at base + 16*L, no
            // VGPRs): all staging loads of the round are in flight together and are awaited once.  (Staging through
            // registers put each load in its own basic block: load, wait, ds_write, next load - up to four serialized
            // HBM latencies per round; rotating the quarters to dodge LDS bank conflicts measured slower.)
            uint32_t sidx[MIC_RMAX / 8];
#pragma unroll
            for (int i = 0; i < MIC_RMAX / 8; ++i) sidx[i] = runslot[8 * i + (lane >> 3)];
#pragma unroll
            for (int i = 0; i < MIC_RMAX / 8; ++i) {
              if (8u * i >= nrun) break;                 // wave-uniform: no address arithmetic for unused groups
              if (8u * i + (lane >> 3) < nrun)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slots + (uint64_t)sidx[i] * 8 + (lane & 7)),
                                                 (__attribute__((address_space(3))) void*)(stage + (64 + MIC_M_SKEW) * i), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // Both k-mers of the lane search their staged slot in lockstep: five LDS round trips per round
            // (binary, two reads per step) instead of five per k-mer one
            // after the other.  Lanes without a search read slot 0 of the area and discard.
            {
              const bool v0 = s0_ != 0xFFFFFFFFu && rid0 - rbase < MIC_RMAX, v1 = s1_ != 0xFFFFFFFFu && rid1 - rbase < MIC_RMAX;
              const uint4* sp0 = stage + (v0 ? staged_at_m(rid0 - rbase) : 0);
              const uint4* sp1 = stage + (v1 ? staged_at_m(rid1 - rbase) : 0);
              const unsigned long long* k0 = (const unsigned long long*)sp0;
              const unsigned long long* k1 = (const unsigned long long*)sp1;
              const uint64_t c0 = k0_, c1 = k1_;
              const uint32_t mz0 = sp0[7].z, mz1 = sp1[7].z;
              uint32_t pos0 = 0, pos1 = 0; bool eq0 = false, eq1 = false;
              { const uint64_t x = k0[7], y = k1[7];
                if (x <= c0) pos0 = 8; if (y <= c1) pos1 = 8; eq0 = x == c0; eq1 = y == c1; }
              { const uint64_t x = k0[pos0 + 3], y = k1[pos1 + 3];                                  // index <= 11
                if (x <= c0) pos0 += 4; if (y <= c1) pos1 += 4; eq0 = eq0 || x == c0; eq1 = eq1 || y == c1; }
              { const uint32_t i = pos0 + 1, j = pos1 + 1;
                const uint64_t x = k0[i < 11 ? i : 11], y = k1[j < 11 ? j : 11];
                if (i < 12 && x <= c0) pos0 += 2; if (j < 12 && y <= c1) pos1 += 2; eq0 = eq0 || x == c0; eq1 = eq1 || y == c1; }
              { const uint32_t i = pos0, j = pos1;
                const uint64_t x = k0[i < 11 ? i : 11], y = k1[j < 11 ? j : 11];
                if (i < 12 && x <= c0) pos0 += 1; if (j < 12 && y <= c1) pos1 += 1; eq0 = eq0 || x == c0; eq1 = eq1 || y == c1; }
              const bool leaf0 = !(mz0 & MIC_M_DIR), leaf1 = !(mz1 & MIC_M_DIR);
              // third trip: the label of key pos-1 (u16 at byte 96 + 2(pos-1)) or the child base (u32 at byte 96)
              const uint32_t p0 = pos0 ? pos0 - 1 : 0, p1 = pos1 ? pos1 - 1 : 0;
              const uint32_t w0 = ((const uint32_t*)sp0)[24 + (leaf0 ? p0 >> 1 : 0)], w1 = ((const uint32_t*)sp1)[24 + (leaf1 ? p1 >> 1 : 0)];
              if (v0) {
                o0_ = 0; y0_ = 0xFFFFFFFFu;
                if (pos0) { if (leaf0) { if (eq0) o0_ = ((p0 & 1) ? w0 >> 16 : w0 & 0xFFFFu) + 1; } else y0_ = w0 + p0; }
              }
              if (v1) {
                o1_ = 0; y1_ = 0xFFFFFFFFu;
                if (pos1) { if (leaf1) { if (eq1) o1_ = ((p1 & 1) ? w1 >> 16 : w1 & 0xFFFFu) + 1; } else y1_ = w1 + p1; }
              }
            }
          }
        };
        if (__ballot(sl0 != 0xFFFFFFFFu) | __ballot(sl1 != 0xFFFFFFFFu)) {
          uint32_t nx0, nx1;
          level(sl0, sl1, c[0], c[1], res0, res1, nx0, nx1);
          sl0 = nx0; sl1 = nx1;
          // ---- deeper levels of the bucket trees.  Nearly every chunk has k-mers in buckets of more than 12 entries (two
          // super-k-mers sharing a slot are enough), but only ~40 % of its k-mers: instead of a second full round over
          // two sparse 64-lane passes (240 VALU per read, measured by ablation), the k-mers that descend are packed into
          // ONE array in k-mer order (equal child slots stay adjacent) and levels 2, 3, ... run on that.  The stage area
          // is free between levels and serves as the scratch for the packing.
          const uint64_t mm0 = __ballot(sl0 != 0xFFFFFFFFu), mm1 = __ballot(sl1 != 0xFFFFFFFFu);
          const uint32_t n0 = __popcll(mm0), n2 = n0 + __popcll(mm1);
          if (n2 > 64) {               // more than one array holds (rare): level by level on both passes
            while (__ballot(sl0 != 0xFFFFFFFFu) | __ballot(sl1 != 0xFFFFFFFFu)) {
              uint32_t y0, y1;
              level(sl0, sl1, c[0], c[1], res0, res1, y0, y1);   // a k-mer that descends had no hit yet: res0/res1 are free
              sl0 = y0; sl1 = y1;
            }
          } else if (n2) {
            unsigned long long* ck = (unsigned long long*)stage;          // 64 keys
            uint32_t* cs = (uint32_t*)(stage + 32);                        // 64 slots, behind the keys
            __builtin_amdgcn_wave_barrier();
            if ((mm0 >> lane) & 1) { const uint32_t d = below(mm0); ck[d] = c[0]; cs[d] = sl0; }
            if ((mm1 >> lane) & 1) { const uint32_t d = n0 + below(mm1); ck[d] = c[1]; cs[d] = sl1; }
            __builtin_amdgcn_wave_barrier();
            const uint64_t c2 = ck[lane];
            uint32_t s2 = (uint32_t)lane < n2 ? cs[lane] : 0xFFFFFFFFu;
            __builtin_amdgcn_wave_barrier();
            while (__ballot(s2 != 0xFFFFFFFFu)) {
              const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)s2, 0x138, 0xF, 0xF, false);   // wave_shr:1
              const bool f2 = s2 != 0xFFFFFFFFu && s2 != pv;
              const uint64_t b2 = __ballot(f2);
              const uint32_t R2 = __popcll(b2), rid2 = below(b2) + (f2 ? 1u : 0u) - 1;
              uint32_t y2 = 0xFFFFFFFFu;
              for (uint32_t rbase = 0; rbase < R2; rbase += MIC_RMAX) {
                __builtin_amdgcn_wave_barrier();
                if (f2 && rid2 - rbase < MIC_RMAX) runslot[rid2 - rbase] = s2;
                __builtin_amdgcn_wave_barrier();
                const uint32_t nrun = R2 - rbase < MIC_RMAX ? R2 - rbase : MIC_RMAX;
                uint32_t sidx[MIC_RMAX / 8];
#pragma unroll
                for (int i = 0; i < MIC_RMAX / 8; ++i) sidx[i] = runslot[8 * i + (lane >> 3)];
#pragma unroll
                for (int i = 0; i < MIC_RMAX / 8; ++i) {
                  if (8u * i >= nrun) break;
                  if (8u * i + (lane >> 3) < nrun)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slots + (uint64_t)sidx[i] * 8 + (lane & 7)),
                                                     (__attribute__((address_space(3))) void*)(stage + (64 + MIC_M_SKEW) * i), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const bool v2 = s2 != 0xFFFFFFFFu && rid2 - rbase < MIC_RMAX;
                const uint4* sp = stage + (v2 ? staged_at_m(rid2 - rbase) : 0);
                const unsigned long long* kk = (const unsigned long long*)sp;
                const uint32_t mz = sp[7].z;
                uint32_t pos = 0; bool eq;
                { const uint64_t x = kk[7]; if (x <= c2) pos = 8; eq = x == c2; }
                { const uint64_t x = kk[pos + 3]; if (x <= c2) pos += 4; eq = eq || x == c2; }
                { const uint32_t i = pos + 1; const uint64_t x = kk[i < 11 ? i : 11]; if (i < 12 && x <= c2) pos += 2; eq = eq || x == c2; }
                { const uint32_t i = pos; const uint64_t x = kk[i < 11 ? i : 11]; if (i < 12 && x <= c2) pos += 1; eq = eq || x == c2; }
                const bool leaf = !(mz & MIC_M_DIR);
                const uint32_t pp2 = pos ? pos - 1 : 0;
                const uint32_t wv2 = ((const uint32_t*)sp)[24 + (leaf ? pp2 >> 1 : 0)];
                if (v2 && pos) { if (leaf) { if (eq) res2 = ((pp2 & 1) ? wv2 >> 16 : wv2 & 0xFFFFu) + 1; } else y2 = wv2 + pp2; }
              }
              s2 = y2;
            }
          }
        }
        tally3(res0, res1, res2, acc, n_ent, overflow, total, lane);
      }
    }
    // next read's header/window and the pointers of the one after it: take before the stores below, issue after
    uint32_t t_hdr, t_w, t_pp, t_pe;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead_sel ? ahead1 : ahead0, n_pp, t_hdr, t_w, t_pp, t_pe);
    __builtin_amdgcn_wave_barrier();
    {
      // The output pointers are needed once per read: they are re-read from the kernarg segment here (scalar loads
      // that hit the scalar cache) instead of living in SGPRs for the whole kernel - the kernel was spilling 35 SGPRs
      // into VGPR lanes, ~58 v_readlane/v_writelane per read.  The asm keeps the loads from being hoisted.
      uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
      struct { uint32_t* results; uint32_t* rows; uint32_t* flagged; uint32_t row_words, flagged_cap; } fa;
      fa.results = kc->results; fa.rows = kc->rows; fa.flagged = kc->flagged; fa.row_words = kc->row_words; fa.flagged_cap = kc->flagged_cap;
      finish_read(acc, n_ent, total, overflow, r, fa, lane);
    }
    ahead_issue(ahead_sel ? ahead0 : ahead1, t_pp, r + 3 * n_waves);
    ahead_sel ^= 1;
    cur_pp = n_pp; cur_pe = n_pe; cur_hdr = t_hdr; cur_w = t_w; n_pp = t_pp; n_pe = t_pe;
  }
}

// =====================================================================================================================
// query_kernel_s - super-k-mer table (layout 2 internally, MIC_LAYOUT_SUPER; format and rules in mic_device.h).  Same
// work mapping, k-mer assembly, run detection, LDS-DMA staging, tally and result code as query_kernel_m.  Differences:
// the sliding-minimum keys carry the strand and position of their m-mer in the low 5 bits, so every k-mer knows WHERE its
// minimizer sits: it orients itself by the minimizer's strand, takes the slot from the full minimizer value, finds the
// entries with that value by a 3-step search over the slot's six sort keys and compares itself with the super-k-mer at
// its alignment.  A slot holds ~1.5 entries on average: continuation slots are rare and there is no second level.
// =====================================================================================================================
// FWD: the table holds both strands of every k-mer under forward-strand minimizers (MIC_LAYOUT_SUPER2, mic_device.h:
// s_candidates_fwd): a k-mer is looked up as it stands in the read - no reverse complement, no canonical m-mer, no strand.
template <int KK, int MM, bool SHARDED, bool FWD>
__global__ void __launch_bounds__(64 * MIC_M_WPB, 32 / MIC_M_WPB) query_kernel_s(const MicQueryArgs a) {
  __shared__ uint4 s_stage[MIC_M_WPB][MIC_RMAX * MIC_MSTRIDE + (MIC_RMAX / 8 - 1) * MIC_R_SKEW];
  __shared__ uint32_t s_run[MIC_M_WPB][MIC_RMAX];
  __shared__ uint32_t s_ahead[MIC_M_WPB][2][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: LDS bases stay in SGPRs
  uint4* stage = s_stage[wv];
  uint32_t* runslot = s_run[wv];
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * MIC_M_WPB + wv);
  const uint32_t n_waves = gridDim.x * MIC_M_WPB;
  const MicTable& t = a.t;
  const int k = KK ? KK : t.k, m = MM ? MM : t.m, w = k - m + 1;
  const uint4* __restrict__ slots = t.slots;
  const uint16_t* __restrict__ cont = a.cont;
  // number of set bits of a 64-bit lane mask below this lane (v_mbcnt_lo/hi: no mask register to keep alive)
  auto below = [](uint64_t mask) { return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };

  // Read-ahead through LDS (global_load_lds_dword): no VGPR lives across a read and no load result is touched near its
  // issue, so nothing waits for it.  One DMA instruction per read fetches {header + first window of read j+2 (lanes
  // 0..11, aligned dwords), reads_ptr of read j+3 (lanes 12, 13)}; it is taken one read later, before the result stores
  // of finish_read (stores count in vmcnt too - waiting for the entry after them would wait for their acknowledgement).
  uint32_t* ahead0 = s_ahead[wv][0];
  uint32_t* ahead1 = s_ahead[wv][1];
  auto ahead_issue = [&](uint32_t* entry, uint32_t pp_w, uint32_t r_ptr) {
    // lanes 0..11: aligned dwords of the window; lanes 12, 13: reads_ptr[rr], reads_ptr[rr + 1]; the rest repeat lane 0.
    // One select between two wave-uniform bases, then base + 4 * lane (the pointer base is pre-biased by -48).
    const uint64_t abase = ((uint64_t)(cont + pp_w)) & ~3ULL;
    const uint32_t rr = r_ptr < a.n_reads ? r_ptr : a.n_reads - 1;
    const uint64_t pbase = (uint64_t)(a.reads_ptr + rr) - 48;
    uint32_t lv = (uint32_t)lane;
    asm volatile("" : "+v"(lv));      // recomputed per read (3 VALU) instead of a 64-bit offset kept live in VGPRs
    const uint32_t li = lv < 14 ? lv : 0u;
    const uint64_t addr = (li < 12 ? abase : pbase) + 4 * li;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr,
                                     (__attribute__((address_space(3))) void*)entry, 4, 0, 0);
  };
  auto ahead_take = [&](const uint32_t* entry, uint32_t pp_w, uint32_t& hdr, uint32_t& npp, uint32_t& npe) {
    const uint32_t raw = entry[lane];
    npp = __builtin_amdgcn_readlane(raw, 12); npe = __builtin_amdgcn_readlane(raw, 13);
    const bool odd = (((uint64_t)(cont + pp_w)) >> 1) & 1;      // is container pp_w the high half of its dword?
    const uint32_t first = __builtin_amdgcn_readfirstlane(raw);
    hdr = odd ? first >> 16 : first & 0xFFFFu;
  };
  // the first window of the read comes out of the same LDS entry when its first chunk starts (the entry is not written
  // again before the end of that read): no VGPR carries it across finish_read
  auto ahead_word = [&](const uint32_t* entry, uint32_t pp_w) {
    const uint32_t raw = entry[lane];
    const bool odd = (((uint64_t)(cont + pp_w)) >> 1) & 1;
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x101, 0xF, 0xF, true);   // row_shl:1 = lane+1
    // window word of lane L = (container pp+1+2L) << 16 | container pp+2+2L
    return odd ? ((up << 16) | (up >> 16)) : ((raw & 0xFFFF0000u) | (up & 0xFFFFu));
  };
  uint32_t cur_pp, cur_pe, cur_hdr;   // read r: pointers, first part header
  uint32_t n_pp, n_pe;                       // pointers of read r + n_waves
  uint32_t ahead_sel;
  {
    const uint32_t r0 = wave0 < a.n_reads ? wave0 : a.n_reads - 1;
    cur_pp = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0]); cur_pe = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0 + 1]);
    ahead_issue(ahead0, cur_pp, wave0 + n_waves);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead0, cur_pp, cur_hdr, n_pp, n_pe);
    ahead_issue(ahead1, n_pp, wave0 + 2 * n_waves);
    ahead_sel = 1;
  }
  for (uint32_t r = wave0; r < a.n_reads; r += n_waves) {
    uint32_t pp = cur_pp;
    const uint32_t pe = cur_pe;
    RowAcc acc; acc.label1 = 0; acc.count = 0;
    uint32_t n_ent = 0, overflow = 0, total = 0;
    bool first_part = true;

    while (pp < pe) {
      // the header of a later part - for nearly every read the 0 that ends it - is among the 24 containers of the read-ahead
      // entry more often than not: an LDS read instead of a global load the whole wave waits for
      uint32_t plen;
      {
        const uint32_t rel = pp - cur_pp + (uint32_t)((((uint64_t)(cont + cur_pp)) >> 1) & 1);    // u16 offset inside the entry
        if (first_part) plen = cur_hdr;
        else if (rel < 24u) {
          const uint32_t v = (ahead_sel ? ahead0 : ahead1)[rel >> 1];
          plen = __builtin_amdgcn_readfirstlane((rel & 1u) ? v >> 16 : v & 0xFFFFu);
        } else plen = __builtin_amdgcn_readfirstlane((uint32_t)cont[pp]);
      }
      const bool ahead_ok = first_part;
      first_part = false;
      if (plen == 0) break;
      const uint32_t first = pp + 1;
      pp = first + (plen + 7) / 8;
      if (plen < (uint32_t)k) continue;
      const uint32_t nk = plen - k + 1;
      const uint32_t cend = pp;
      for (uint32_t base = 0; base < nk; base += 128) {
        const bool use_ahead = ahead_ok && base == 0;
        const uint32_t wd = window_word_w(cont, first, cend, base, lane, use_ahead,
                                          use_ahead ? ahead_word(ahead_sel ? ahead0 : ahead1, cur_pp) : 0u);
        // k-mers of the two passes
        uint64_t km[2], rk[2]; bool act[2];
        int ln = lane;
        asm volatile("" : "+v"(ln));   // lane-derived shift counts and positions are recomputed per chunk, not kept in VGPRs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = 4 * h + (lane >> 4);
          uint32_t d0 = bperm(idx, wd), d1 = bperm(idx + 1, wd), d2 = bperm(idx + 2, wd);
          uint64_t kmer = kmer_from_dwords(d0, d1, d2, ln & 15, k);
          // the reverse complement is needed for the orientation (one-strand table) and for the bucket filter of the
          // table-sharded mode (the buckets are those of the canonical k-mer)
          const uint64_t rck = (!FWD || SHARDED) ? revcomp_bits(kmer, k) : 0;
          km[h] = kmer; rk[h] = rck;
          act[h] = base + 64 * h + lane < nk;
          if (SHARDED) {   // table-sharded mode only (its own instantiation: the unsharded kernel carries none of this): divisor and bounds are re-read from the kernarg segment (see finish)
            uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
            if (!kc->t.parted) {   // (a slot-range part is filtered below, by the slot)
              MicDiv dv; dv.d = kc->t.div.d; dv.magic = kc->t.div.magic; dv.shift = kc->t.div.shift; dv.add = kc->t.div.add;
              const uint64_t s_lo = kc->t.shard_start, s_hi = kc->t.shard_end;
              const uint64_t cc = kmer < rck ? kmer : rck;
              uint64_t q = mic_div(cc, dv);
              uint64_t rem = cc - q * dv.d;
              act[h] = act[h] && rem >= s_lo && rem < s_hi;
            }
          }
        }
        // where the sampled m-mer of every k-mer sits (mod-sampling, mic_device.h; shared with query_kernel_r)
        const uint32_t n_act = nk - base < 128u ? nk - base : 128u;
        const bool past = n_act + (uint32_t)(k - s_tlen(k, m)) >= 129u;
        uint32_t qa[2];
        sampled_positions<!FWD>(wd, ln, ln, k, m, past, qa[0], qa[1]);
        // every k-mer now knows its m-mer: position -> strand (the smaller of the m-mer and its reverse complement), oriented
        // k-mer, nucleotide offset in the entry, minimizer value x -> slot and sort key
        uint64_t ko[2]; uint32_t ao[2], tk32[2], sl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t j = qa[h] - (uint32_t)(64 * h + ln);                 // 0 .. w - 1
          const uint64_t mm = (1ULL << (2 * m)) - 1;
          const uint64_t xf = (km[h] >> (2 * ((uint32_t)(w - 1) - j))) & mm;
          const uint64_t xr = FWD ? 0 : (rk[h] >> (2 * j)) & mm;               // its reverse complement = m-mer w-1-j of rc(k-mer)
          const bool rev = !FWD && xr < xf;
          ko[h] = rev ? rk[h] : km[h];
          // minimizer position in the oriented k-mer: jo = rev ? w-1-j : j; offset of the k-mer in the super-k-mer:
          // w-1-jo; and because k-m = w-1 that offset is also the number of nucleotides to the right of the minimizer
          ao[h] = rev ? j : (uint32_t)(w - 1) - j;
          const uint64_t x = rev ? xr : xf;
          tk32[h] = (uint32_t)x;
          sl[h] = act[h] ? sslot_of_x(x, (uint32_t)t.n_main) : 0xFFFFFFFFu;
          if (SHARDED) {   // slot-range part: the k-mer is this engine's iff its slot is resident here (slots are global indices)
            uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
            if (kc->t.parted && sl[h] - kc->t.slot_lo >= kc->t.slot_cnt) sl[h] = 0xFFFFFFFFu;
          }
        }
        uint32_t sl0 = sl[0], sl1 = sl[1];
        uint32_t res0 = 0, res1 = 0;   // label + 1 of the hit

        // one level of the table on both passes: runs of equal slots, LDS-DMA of the distinct slots, lockstep search
        auto level = [&](uint32_t s0_, uint32_t s1_, uint32_t& o0_, uint32_t& o1_, uint32_t& y0_, uint32_t& y1_) {
          // runs of equal slots over the 128 positions
          uint32_t p0 = bperm((lane + 63) & 63, s0_), p1 = bperm((lane + 63) & 63, s1_);
          uint32_t last0 = bperm(63, s0_);
          if (lane == 0) { p0 = 0xFFFFFFFFu; p1 = last0; }
          const bool f0 = s0_ != 0xFFFFFFFFu && s0_ != p0, f1 = s1_ != 0xFFFFFFFFu && s1_ != p1;
          const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
          const uint32_t R0 = __popcll(b0), R = R0 + __popcll(b1);
          const uint32_t rid0 = below(b0) + (f0 ? 1u : 0u) - 1, rid1 = R0 + below(b1) + (f1 ? 1u : 0u) - 1;
          y0_ = 0xFFFFFFFFu; y1_ = 0xFFFFFFFFu;   // slots to probe at the next level
          for (uint32_t rbase = 0; rbase < R; rbase += MIC_RMAX) {
            __builtin_amdgcn_wave_barrier();
            if (f0 && rid0 - rbase < MIC_RMAX) runslot[rid0 - rbase] = s0_;
            if (f1 && rid1 - rbase < MIC_RMAX) runslot[rid1 - rbase] = s1_;
            __builtin_amdgcn_wave_barrier();
            const uint32_t nrun = R - rbase < MIC_RMAX ? R - rbase : MIC_RMAX;
            // every distinct slot goes HBM -> LDS directly (global_load_lds_dwordx4: lane L lands at base + 16*L, no
            // VGPRs): all staging loads of the round are in flight together and are awaited once.  (Staging through
            // registers put each load in its own basic block: load, wait, ds_write, next load - up to four serialized
            // HBM latencies per round; rotating the quarters to dodge LDS bank conflicts measured slower.)
            uint32_t sidx[MIC_RMAX / 8];
#pragma unroll
            for (int i = 0; i < MIC_RMAX / 8; ++i) sidx[i] = runslot[8 * i + (lane >> 3)];
#pragma unroll
            for (int i = 0; i < MIC_RMAX / 8; ++i) {
              if (8u * i >= nrun) break;                 // wave-uniform: no address arithmetic for unused groups
              if (8u * i + (lane >> 3) < nrun)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slots + (uint64_t)sidx[i] * 8 + (lane & 7)),
                                                 (__attribute__((address_space(3))) void*)(stage + (64 + MIC_R_SKEW) * i), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // Both k-mers of the lane look in their staged slot in lockstep: lower bound of the sort key among the six
            // (three reads), then the entry: key, super-k-mer, mask | label.  Lanes without a search read slot 0 and discard.
            {
              const bool v0 = s0_ != 0xFFFFFFFFu && rid0 - rbase < MIC_RMAX, v1 = s1_ != 0xFFFFFFFFu && rid1 - rbase < MIC_RMAX;
              const uint32_t* q0 = (const uint32_t*)(stage + (v0 ? staged_at(rid0 - rbase) : 0));
              const uint32_t* q1 = (const uint32_t*)(stage + (v1 ? staged_at(rid1 - rbase) : 0));
              const uint32_t t0 = tk32[0], t1 = tk32[1];
              uint32_t e0 = q0[3] < t0 ? 4u : 0u, e1 = q1[3] < t1 ? 4u : 0u;
              e0 += q0[e0 + 1] < t0 ? 2u : 0u; e1 += q1[e1 + 1] < t1 ? 2u : 0u;                    // index <= 5
              e0 += q0[e0 < 5 ? e0 : 5] < t0 ? 1u : 0u; e1 += q1[e1 < 5 ? e1 : 5] < t1 ? 1u : 0u;
              const uint32_t mz0 = q0[30], mz1 = q1[30];      // entries | NEXT
              bool more0 = v0 && e0 < 6, more1 = v1 && e1 < 6;
              uint32_t hit0 = 0, hit1 = 0;
              e0 = e0 < 5 ? e0 : 5; e1 = e1 < 5 ? e1 : 5;
              for (;;) {
                const uint32_t g0 = q0[e0], g1 = q1[e1];
                const uint32_t a0 = q0[6 + 3 * e0], b0 = q0[7 + 3 * e0], c0 = q0[8 + 3 * e0], p0 = q0[24 + e0];
                const uint32_t a1 = q1[6 + 3 * e1], b1 = q1[7 + 3 * e1], c1 = q1[8 + 3 * e1], p1 = q1[24 + e1];
                const bool same0 = more0 && g0 == t0, same1 = more1 && g1 == t1;
                const bool m0 = same0 && ((p0 >> (16 + (w - 1) - ao[0])) & 1) && s_extract(a0, b0, c0, (int)ao[0], k) == ko[0];
                const bool m1 = same1 && ((p1 >> (16 + (w - 1) - ao[1])) & 1) && s_extract(a1, b1, c1, (int)ao[1], k) == ko[1];
                if (m0) hit0 = (p0 & 0xFFFFu) + 1;
                if (m1) hit1 = (p1 & 0xFFFFu) + 1;
                {
                  // a marker entry (presence mask 0) of this very minimizer: the minimizer is crowded, its k-mers are in the
                  // side table (mic_build.hip: s_crowd_move_kernel).  Rare: a wave-uniform branch around the probes.
                  const bool cr0 = same0 && (p0 >> 16) == 0, cr1 = same1 && (p1 >> 16) == 0;
                  if (__ballot(cr0) | __ballot(cr1)) {
                    uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(kp));
                    const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
                    const uint4* side = kc->t.side; const uint32_t smask = kc->t.side_mask;
                    const uint64_t mm = (1ULL << (2 * m)) - 1;
                    if (side && cr0 && s_extract(a0, b0, c0, w - 1, m) == ((ko[0] >> (2 * ao[0])) & mm)) hit0 = s_side_probe(side, smask, ko[0]);
                    if (side && cr1 && s_extract(a1, b1, c1, w - 1, m) == ((ko[1] >> (2 * ao[1])) & mm)) hit1 = s_side_probe(side, smask, ko[1]);
                  }
                }
                more0 = same0 && !m0 && e0 < 5; more1 = same1 && !m1 && e1 < 5;     // another entry of the same minimizer?
                e0 += more0 ? 1u : 0u; e1 += more1 ? 1u : 0u;
                if (!(__ballot(more0) | __ballot(more1))) break;
              }
              if (v0) { o0_ = hit0; y0_ = 0xFFFFFFFFu; }
              if (v1) { o1_ = hit1; y1_ = 0xFFFFFFFFu; }
              // continuation slot (rare, wave-uniform test): only if the entries there can carry this key - they are sorted
              // across the chain, so the last key of this slot must not be above it
              const bool n0 = v0 && !hit0 && (mz0 & MIC_S_NEXT), n1 = v1 && !hit1 && (mz1 & MIC_S_NEXT);
              if (__ballot(n0) | __ballot(n1)) {
                if (n0 && q0[5] <= t0) y0_ = q0[31];
                if (n1 && q1[5] <= t1) y1_ = q1[31];
              }
            }
          }
        };
        while (__ballot(sl0 != 0xFFFFFFFFu) | __ballot(sl1 != 0xFFFFFFFFu)) {   // second and later rounds: continuation slots (rare)
          uint32_t nx0, nx1;
          level(sl0, sl1, res0, res1, nx0, nx1);
          sl0 = nx0; sl1 = nx1;
        }
        tally2(res0, res1, acc, n_ent, overflow, total, lane);
      }
    }
    // next read's header/window and the pointers of the one after it: take before the stores below, issue after
    uint32_t t_hdr, t_pp, t_pe;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead_sel ? ahead1 : ahead0, n_pp, t_hdr, t_pp, t_pe);
    __builtin_amdgcn_wave_barrier();
    {
      // The output pointers are needed once per read: they are re-read from the kernarg segment here (scalar loads
      // that hit the scalar cache) instead of living in SGPRs for the whole kernel - the kernel was spilling 35 SGPRs
      // into VGPR lanes, ~58 v_readlane/v_writelane per read.  The asm keeps the loads from being hoisted.
      uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
      struct { uint32_t* results; uint32_t* rows; uint32_t* flagged; uint32_t row_words, flagged_cap; } fa;
      fa.results = kc->results; fa.rows = kc->rows; fa.flagged = kc->flagged; fa.row_words = kc->row_words; fa.flagged_cap = kc->flagged_cap;
      finish_read(acc, n_ent, total, overflow, r, fa, lane);
    }
    ahead_issue(ahead_sel ? ahead0 : ahead1, t_pp, r + 3 * n_waves);
    ahead_sel ^= 1;
    cur_pp = n_pp; cur_pe = n_pe; cur_hdr = t_hdr; n_pp = t_pp; n_pe = t_pe;
  }
}


// =====================================================================================================================
// query_kernel_r - the two-strand super-k-mer table (MIC_LAYOUT_SUPER2) probed per RUN, not per k-mer.  Front half as in
// query_kernel_s<.., FWD>: window, m-mer order keys, sliding minimum - every k-mer knows the position of its minimizer.
// Consecutive k-mers with the minimizer at the same read position form a run; they are the k-mers of ONE super-k-mer of
// the read, and the table stores super-k-mers.  So the back half works on runs (<= 32 per round, one lane each, ~19 per
// 150-bp read) instead of on 2 x 64 k-mers:
//   * the run's lane cuts the k+w-1 nucleotides around the minimizer out of the window (4 ds_bpermute + 3 alignbit), in
//     the entry's own alignment (minimizer at nucleotide w-1); the minimizer value x, the slot and the sort key come out
//     of that region - one slot hash per run instead of one per k-mer;
//   * entry against region: XOR, then the number of equal nucleotides to the left (count trailing zeros) and to the right
//     (count leading zeros) of the minimizer.  The k-mer with its minimizer at position j lies inside the equal stretch
//     iff w-1-right <= j <= left: the run's k-mers are a range of j, the entry's presence mask has a bit per j, and the
//     hits of the run against the entry are one popcount;
//   * entries of one minimizer value are adjacent in the slot (sorted by key); the lane walks them while k-mers of its run
//     are unaccounted for, then the continuation slot (rare, wave-uniform loops as before);
//   * the tally adds (label, count) pairs: per distinct label the counts are summed with one ballot per count bit.
// Exactness: a database k-mer is stored once per minimizer position (mic_device.h: s_candidates_fwd), so no k-mer is
// counted twice; nucleotides outside the read part take part in the comparison only beyond the run's own range of j.
// =====================================================================================================================
__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

static_assert(MIC_RMAX <= 32, "tally_counts sums the first two rows of lanes");
#define MIC_R_CROWDED (-16)     /* `remaining` of a run that met the marker of a crowded minimizer: negative (a count never is), an inline constant */
__device__ __forceinline__ void tally_counts(uint32_t lab1, uint32_t cnt, RowAcc& acc, uint32_t& n_ent, uint32_t& overflow,
                                             uint32_t& total, int lane) {
  uint64_t mm = wballot(lab1 != 0);
  while (mm) {
    const uint32_t l1 = __builtin_amdgcn_readlane(lab1, __builtin_ctzll(mm));
    const bool mine = lab1 == l1;
    mm &= ~wballot(mine);
    const uint32_t c = mine ? cnt : 0u;
    uint32_t sum = 0;
    // the runs sit in lanes 0 .. 31 = the first two rows of 16 lanes: an inclusive row scan by DPP (four full-rate adds) leaves the
    // rows' sums in lanes 15 and 31.  (One ballot per count bit - 4-5 and-compare pairs and three scalar operations each - was
    // what the product ran until round 4: 1.0-1.5 % slower, 4.99 against 5.04-5.08 ms.)
    {
      uint32_t v = c;
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
      sum = __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31);
    }
    total += sum;
    row_add(acc, n_ent, overflow, l1, sum, lane);
  }
}

template <int KK, int MM, bool FWD, bool PART>
__global__ void __launch_bounds__(64 * MIC_M_WPB, 32 / MIC_M_WPB) query_kernel_r(const MicQueryArgs a) {
  // staged slots: 8 per LDS-DMA instruction, 128 bytes apart (the DMA's own layout: lane L lands at base + 16 L); each
  // group of 8 starts MIC_R_SKEW uint4 further so that the run lanes' reads of the same word of their slots spread over
  // four times as many banks.  The list of slots to load sits in the same area: it is consumed before the DMA lands.
  __shared__ uint4 s_stage[MIC_M_WPB][MIC_RMAX * MIC_MSTRIDE + (MIC_RMAX / 8 - 1) * MIC_R_SKEW];
  __shared__ uint16_t s_rec[MIC_M_WPB][132];              // runs of a chunk: minimizer position | first k-mer << 8
  __shared__ uint32_t s_ahead[MIC_M_WPB][2][64];
  __shared__ uint32_t s_part[PART ? MIC_M_WPB : 1][2];     // slot-range part: first resident slot, number of resident slots
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4* stage = s_stage[wv];
  uint16_t* rec = s_rec[wv];
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * MIC_M_WPB + wv);
  const uint32_t n_waves = gridDim.x * MIC_M_WPB;
  const MicTable& t = a.t;
  if (PART) { if (lane == 0) { s_part[wv][0] = t.slot_lo; s_part[wv][1] = t.slot_cnt; } __builtin_amdgcn_wave_barrier(); }
  const int k = KK ? KK : t.k, m = MM ? MM : t.m, ctx = k - m;
  const uint4* __restrict__ slots = t.slots;
  const uint16_t* __restrict__ cont = a.cont;
  auto below = [](uint64_t mask) { return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };

  // read-ahead through LDS: see query_kernel_s
  uint32_t* ahead0 = s_ahead[wv][0];
  uint32_t* ahead1 = s_ahead[wv][1];
  auto ahead_issue = [&](uint32_t* entry, uint32_t pp_w, uint32_t r_ptr) {
    const uint64_t abase = ((uint64_t)(cont + pp_w)) & ~3ULL;
    const uint32_t rr = r_ptr < a.n_reads ? r_ptr : a.n_reads - 1;
    const uint64_t pbase = (uint64_t)(a.reads_ptr + rr) - 48;
    // two LDS-DMA instructions under their lanes' masks, scalar base + 4 * lane each (lane L lands at entry + 4 L whatever
    // the mask): 12 window dwords, 2 pointers - 14 loads instead of 64, and none of the selects of the one-instruction form
    // (the pointers are loaded by lanes 0, 1 into entry + 12: with one destination the compiler merges the two loads again)
    int ln = lane;
    asm volatile("" : "+v"(ln));           // (the two lane masks recomputed per read instead of two scalar register pairs kept across the kernel)
    if (ln < 12)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const uint32_t*)abase + lane),
                                       (__attribute__((address_space(3))) void*)entry, 4, 0, 0);
    // (the scalar base opaque: reassociated into (reads_ptr + lane) + rr the lane part is hoisted out of the loop as a 64-bit
    // vector address - two registers the kernel does not have; this way it is scalar base + the 32-bit lane offset of the load above)
    uint64_t pb = pbase + 48;
    asm volatile("" : "+s"(pb));
    if (ln < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const uint32_t*)pb + lane),
                                       (__attribute__((address_space(3))) void*)(entry + 12), 4, 0, 0);
  };
  // (the pointers through the scalar cache instead - one s_load_dwordx2 two reads ahead, no second DMA, no v_readlane - measured in
  // round 6: -3 VALU, +6 SALU per read, 1 % slower on the two-strand table, 0.5 % faster on the one-strand one: not kept)
  auto ahead_take = [&](const uint32_t* entry, uint32_t pp_w, uint32_t& hdr, uint32_t& npp, uint32_t& npe) {
    const uint32_t raw = entry[lane];
    npp = __builtin_amdgcn_readlane(raw, 12); npe = __builtin_amdgcn_readlane(raw, 13);
    const bool odd = (((uint64_t)(cont + pp_w)) >> 1) & 1;
    const uint32_t first = __builtin_amdgcn_readfirstlane(raw);
    hdr = odd ? first >> 16 : first & 0xFFFFu;
  };
  auto ahead_word = [&](const uint32_t* entry, uint32_t pp_w) {
    const uint32_t raw = entry[lane];
    const bool odd = (((uint64_t)(cont + pp_w)) >> 1) & 1;
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x101, 0xF, 0xF, true);
    // odd: (up << 16) | (up >> 16); even: (raw & 0xFFFF0000) | (up & 0xFFFF) - one byte permute with a scalar selector instead of
    // a scalar branch around either form
    return __builtin_amdgcn_perm(up, raw, odd ? 0x05040706u : 0x03020504u);
  };
  uint32_t cur_pp, cur_pe, cur_hdr, n_pp, n_pe, ahead_sel;
  {
    const uint32_t r0 = wave0 < a.n_reads ? wave0 : a.n_reads - 1;
    cur_pp = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0]); cur_pe = __builtin_amdgcn_readfirstlane(a.reads_ptr[r0 + 1]);
    ahead_issue(ahead0, cur_pp, wave0 + n_waves);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead0, cur_pp, cur_hdr, n_pp, n_pe);
    ahead_issue(ahead1, n_pp, wave0 + 2 * n_waves);
    ahead_sel = 1;
  }
  // Which instantiations run the software-pipelined road of the loop below: those with k and m as constants - the others are at
  // the register budget as they are (pipelined they spill into scratch memory and lose 10-17 %) and take every read through the
  // plain road.  MIC_R_PIPE: 0 none, 1 the two-strand table's only, 2 (default) the one-strand table's too.
  constexpr bool PIPE = KK != 0 && (MIC_R_PIPE >= 2 || (MIC_R_PIPE == 1 && FWD));
  // ---- the loop over the wave's reads, SOFTWARE-PIPELINED across reads (round 5) --------------------------------------------------
  // With 8 wavefronts per SIMD and ~2-3 us between the issue of a read's slot loads and their arrival, the vector unit stood idle
  // ~15 % of the time: every wavefront waited for ITS slots with nothing else to do (a closed queue of 8 customers around one
  // server: utilisation 0.85-0.89 at that think time).  Now a read's slot loads are issued and the wavefront goes on to the FRONT
  // HALF of its next read (window, sampled positions, runs, regions, slot hashes: ~60 % of a read's vector work, no memory access
  // of its own); only then it waits, compares and tallies the earlier read.  What lives across: the run lanes' region words,
  // range of positions, slot (7 VGPRs) and three scalars.  ONE stage area still: the next read's slot list is written after
  // the earlier read's slots have been consumed.  Reads that are not one round of one chunk of one part (long reads, several
  // parts, more than 32 runs) drain the pipeline and take the plain road: front / setup / issue / consume one after the other.
  // Measured (headline, 10 M x 150 bp, 119 GB table): 4.63 -> 4.37-4.45 ms by HIP events, 241 VALU + 172 SALU + 45 branches per read
  // against 240 + 167 + 40: the vector unit 89 % busy instead of 85 %.  One-strand table: 4.97 -> 4.83 ms.
  //
  // CROWDED MINIMIZERS (round 6).  A run whose minimizer is a crowded one (mic_build.hip: s_crowd_move_kernel) meets a marker
  // entry; its k-mers live in the side table, keyed by the k-mer.  Until round 5 a table with a side table selected an
  // instantiation of this kernel that carried the per-k-mer side lookups as a rare path - at the register budget, so without
  // the pipelined road, for EVERY read of every real database.  Now the kernel only hands such a run over: its region words and
  // range of positions (everything the follow-up needs: the run's k-mers are substrings of the region) go to a work list in HBM
  // (`crowd_emit`), the read's row so far is spilled instead of finished (`finish`), and crowd_finish_kernel - launched behind
  // this kernel on the same stream - probes the side table one lane per k-mer, adds the hits to the row and finishes the read.
  // The common path pays one compare per entry and one scalar test per read.
  struct Round {
    uint32_t G0, G1, G2, cur;
    int jmax, jmin, remaining;
  };
  // front half of a chunk: window word, sampled positions, run records in LDS; returns the number of runs
  auto front = [&](const uint32_t first, const uint32_t cend, const uint32_t base, const uint32_t nk, const bool use_ahead, uint32_t& wd_out) __attribute__((always_inline)) -> uint32_t {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const uint32_t wd = use_ahead ? ahead_word(ahead_sel ? ahead0 : ahead1, cur_pp)
                                  : window_word_w(cont, first, cend, base, ln, false, 0u);
    wd_out = wd;
    const uint32_t n_act = nk - base < 128u ? nk - base : 128u;
    const bool past = n_act + (uint32_t)(k - s_tlen(k, m)) >= 129u;
    uint32_t qa0, qa1;
    sampled_positions<!FWD>(wd, ln, lane, k, m, past, qa0, qa1);
    qa0 = (uint32_t)lane < n_act ? qa0 : 0xFFu;
    qa1 = 64u + (uint32_t)lane < n_act ? qa1 : 0xFFu;
    const uint32_t last0 = __builtin_amdgcn_readlane(qa0, 63);
    uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)qa0, 0x138, 0xF, 0xF, false);
    uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp((int)last0, (int)qa1, 0x138, 0xF, 0xF, false);
    const bool f0 = qa0 != p0, f1 = qa1 != p1;
    const uint64_t b0 = wballot(f0), b1 = wballot(f1);
    const uint32_t R0 = __popcll(b0), R = R0 + __popcll(b1) - (n_act < 128u ? 1u : 0u);
    __builtin_amdgcn_wave_barrier();
    if (f0) rec[below(b0)] = (uint16_t)(qa0 | ((uint32_t)lane << 8));
    if (f1) rec[R0 + below(b1)] = (uint16_t)(qa1 | ((64u + (uint32_t)lane) << 8));
    if (__builtin_expect(n_act == 128u, 0)) {
      asm volatile("" ::: "memory");
      if (ln == 0) rec[R] = (uint16_t)(128u << 8);
    }
    __builtin_amdgcn_wave_barrier();
    return R;
  };
  // one round of runs (<= MIC_RMAX, one lane each): region, minimizer, slot
  const uint32_t xsh = (64u - 2u * (uint32_t)k) & 31u;                              // k > 16: 0 .. 30
  auto region_x = [&](uint32_t A0, uint32_t A1, uint32_t& lo, uint32_t& hi) {
    if (k > 16) {
      lo = __builtin_amdgcn_alignbit(A0, A1, xsh) & (m >= 16 ? 0xFFFFFFFFu : (1u << ((2 * m) & 31)) - 1u);
      hi = m > 16 ? (A0 >> xsh) & ((1u << ((2 * m - 32) & 31)) - 1u) : 0u;
    } else {
      const uint64_t x = ((((uint64_t)A0 << 32) | A1) << (2 * ctx)) >> (64 - 2 * m);
      lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    }
  };
  auto setup = [&](const uint32_t wd, const uint32_t rbase, const uint32_t R, Round& L, uint32_t& nrun_out) __attribute__((always_inline)) {
    const uint32_t nrun = R - rbase < MIC_RMAX ? R - rbase : MIC_RMAX;
    nrun_out = nrun;
    const bool vr = (uint32_t)lane < nrun;
    const uint32_t ri = vr ? rbase + (uint32_t)lane : 0u;
    const uint32_t rc0 = rec[ri], rc1 = rec[ri + 1];
    const int qa = (int)(rc0 & 255u), i0 = (int)(rc0 >> 8);
    const int n = (int)(rc1 >> 8) - i0;
    const int s1 = qa - ctx - 1;
    const int D = s1 >> 4;
    const uint32_t tsh = 30u - 2u * (uint32_t)(s1 & 15);
    const int a0 = D << 2;
    const uint32_t W0 = (uint32_t)__builtin_amdgcn_ds_bpermute(a0, (int)wd), W1 = (uint32_t)__builtin_amdgcn_ds_bpermute(a0 + 4, (int)wd),
                   W2 = (uint32_t)__builtin_amdgcn_ds_bpermute(a0 + 8, (int)wd), W3 = (uint32_t)__builtin_amdgcn_ds_bpermute(a0 + 12, (int)wd);
    uint32_t G0 = __builtin_amdgcn_alignbit(W0, W1, tsh), G1 = __builtin_amdgcn_alignbit(W1, W2, tsh), G2 = __builtin_amdgcn_alignbit(W2, W3, tsh);
    uint32_t key, xhi;
    region_x(G0, G1, key, xhi);
    bool rev = false;
    if (!FWD) {
      const uint32_t sh = 96u - 2u * (uint32_t)(k + ctx);
      const uint32_t r0 = __builtin_bitreverse32(G2), r1 = __builtin_bitreverse32(G1), r2 = __builtin_bitreverse32(G0);
      uint32_t q0 = sh ? __builtin_amdgcn_alignbit(r0, r1, 32u - sh) : r0;
      uint32_t q1 = sh ? __builtin_amdgcn_alignbit(r1, r2, 32u - sh) : r1;
      uint32_t q2 = r2 << sh;
      q0 = ~(((q0 >> 1) & 0x55555555u) | ((q0 << 1) & 0xAAAAAAAAu));
      q1 = ~(((q1 >> 1) & 0x55555555u) | ((q1 << 1) & 0xAAAAAAAAu));
      q2 = ~(((q2 >> 1) & 0x55555555u) | ((q2 << 1) & 0xAAAAAAAAu));
      uint32_t kr, hr;
      region_x(q0, q1, kr, hr);
      rev = hr < xhi || (hr == xhi && kr < key);
      G0 = rev ? q0 : G0; G1 = rev ? q1 : G1; G2 = rev ? q2 : G2;
      key = rev ? kr : key; xhi = rev ? hr : xhi;
    }
    const int jmaxf = qa - i0, jminf = jmaxf - n + 1;
    const int jmax = rev ? ctx - jminf : jmaxf, jmin = rev ? ctx - jmaxf : jminf;
    uint32_t cur = vr ? sslot_of_x32(key, xhi, (uint32_t)t.n_main) : 0xFFFFFFFFu;
    bool mine = vr;
    if (PART) {
      mine = vr && cur - s_part[wv][0] < s_part[wv][1];
      cur = mine ? cur : 0xFFFFFFFFu;
    }
    L.G0 = G0; L.G1 = G1; L.G2 = G2; L.cur = cur;
    L.jmax = jmax; L.jmin = jmin; L.remaining = mine ? n : 0;
  };
  // the slots of a round from HBM into the stage area (LDS-DMA; the list of slots sits in the stage area itself: it is consumed
  // before the DMA lands)
  auto issue = [&](const uint32_t cur, const uint32_t nrun) __attribute__((always_inline)) {
    uint32_t sidx[MIC_RMAX / 8];
    int ln = lane;
    asm volatile("" : "+v"(ln));           // (a lane mask recomputed here instead of a scalar register pair kept across the kernel)
    __builtin_amdgcn_wave_barrier();
    if (ln < MIC_RMAX) ((uint32_t*)stage)[lane] = cur;
    __builtin_amdgcn_wave_barrier();
    // (the DMA destinations from an opaque copy of the area's address: as loop invariants the three group offsets are three scalar
    // registers kept across the kernel - one s_add into m0 each does the same work as the s_mov they replace)
    // (an LDS byte offset, taken and used as one: a generic pointer rebuilt from its low word would read as NULL for the wave whose
    // area starts at offset 0 of the block's LDS)
    uint32_t so = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)stage;
    asm volatile("" : "+s"(so));
#pragma unroll
    for (int i = 0; i < MIC_RMAX / 8; ++i) sidx[i] = ((const uint32_t*)stage)[8 * i + (lane >> 3)];
#pragma unroll
    for (int i = 0; i < MIC_RMAX / 8; ++i) {
      if (8u * i >= nrun) break;
      if (sidx[i] != 0xFFFFFFFFu)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slots + (uint64_t)sidx[i] * 8 + (lane & 7)),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(so + 16u * (64 + MIC_R_SKEW) * (uint32_t)i), 16, 0, 0);
    }
  };
  // wait for the slots, entries against regions, tally; continuation slots; crowded runs are handed to the follow-up kernel
  // (cg: the read's last group of crowded runs in the work list - first item | (runs - 1) << 27 -, MIC_CG_NONE, or MIC_CG_DENSE)
  auto consume = [&](Round& L, RowAcc& acc, uint32_t& n_ent, uint32_t& overflow, uint32_t& total, uint32_t& cg) __attribute__((always_inline)) {
    const uint32_t G0 = L.G0, G1 = L.G1, G2 = L.G2;
    uint32_t key, xhi_;
    region_x(G0, G1, key, xhi_);            // (the sort key again from the region: cheaper than a register across the front half)
    const int jmax = L.jmax, jmin = L.jmin;
    uint32_t cur = L.cur;
    int remaining = L.remaining;
    bool again = false;
    do {
      if (again) issue(cur, MIC_RMAX);      // (continuation slots: rare; every group of eight under its lanes' mask)
      again = true;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const bool vl = cur != 0xFFFFFFFFu;
      const uint32_t* q = (const uint32_t*)(stage + (vl ? staged_at((uint32_t)lane) : 0));
      // all six sort keys in one LDS round trip (the slot is 16-byte aligned); rank = number of keys below ours
      const uint4 ka = *(const uint4*)q;
      const uint2 kb = *(const uint2*)(q + 4);
      uint32_t e = (ka.x < key) + (ka.y < key) + (ka.z < key) + (ka.w < key) + (kb.x < key) + (kb.y < key);
      const uint32_t mz = q[30];
      bool more = vl && e < 6;
      e = e < 5 ? e : 5;
      for (;;) {
        uint32_t e3 = e + (e << 1);
        asm volatile("" : "+v"(e3));     // (the compiler would turn 3 e into a 64-bit multiply-add of the LDS address: quarter rate)
        uint32_t g = q[e], S0 = q[6 + e3], S1 = q[7 + e3], S2 = q[8 + e3], pl = q[24 + e];
        asm volatile("" : "+v"(S2));     // read with the others: the compiler would sink it into the branch below, one more LDS round trip
        const bool same = more && g == key;
        const uint32_t d0 = G0 ^ S0, d1 = G1 ^ S1, d2 = G2 ^ S2;
        // the whole minimizer, not only its low 32 bits: its nucleotides are the low 32 - 2 ctx bits of word 0 and the top
        // 2 (ctx + m) - 32 bits of word 1 (32-bit operations: the 64-bit form shifts and compares at half rate)
        const bool mineq = k > 16
                               ? ((d0 & ((1u << ((32 - 2 * ctx) & 31)) - 1u)) | (d1 >> xsh)) == 0
                               : (((((uint64_t)d0 << 32) | d1) << (2 * ctx)) >> (64 - 2 * m)) == 0;
        const uint32_t dl = d0 >> (32 - 2 * ctx);                                      // left context, nucleotide ctx-1 in the low bits
        const uint32_t dr = k > 16 ? __builtin_amdgcn_alignbit(d1, d2, xsh)
                                        : (uint32_t)((((((uint64_t)d0 << 32) | d1)) << ((2 * k) & 63)) >> 32);   // right context, its first nucleotide on top
        const int left = __builtin_ctz(dl | (1u << (2 * ctx))) >> 1;                   // equal nucleotides next to the minimizer
        const int right = __builtin_clz(dr | (1u << (31 - 2 * ctx))) >> 1;
        const int hi = left < jmax ? left : jmax, lo = ctx - right > jmin ? ctx - right : jmin;
        const uint32_t range = ((2u << (hi & 31)) - 1u) & (~0u << (lo & 31));         // empty when hi < lo
        const uint32_t hits = (same && mineq) ? (uint32_t)__popc((pl >> 16) & range) : 0u;
        // the marker of a crowded minimizer (presence mask 0): nothing of this minimizer is in the chains, the lane's walk ends
        // (kept in `remaining`, as a value no count reaches: a lane mask of its own would cost the loop a scalar register pair)
#if !(MIC_X & 1)
        remaining = (same && mineq && (pl >> 16) == 0) ? MIC_R_CROWDED : remaining;
#endif
        // The run's hits are tallied ONCE per round: a lane keeps (label, count) of its run; a second entry with ANOTHER
        // label (the same minimizer in two targets' genomes, both contexts matching parts of the run) is tallied on the
        // spot - a wave-uniform branch that is virtually never taken.
        const uint32_t lab_new = (pl & 0xFFFFu) + 1u;
        tally_counts(hits ? lab_new : 0u, hits, acc, n_ent, overflow, total, lane);
        remaining -= (int)hits;
        more = same && remaining > 0 && e < 5;                        // another entry of the same minimizer?
        e += more ? 1u : 0u;
        if (!wballot(more)) break;
      }
      // continuation slot (rare): entries are sorted across the chain, so only if this slot's last key is not above ours
      const bool nx = vl && remaining > 0 && (mz & MIC_S_NEXT);
      cur = 0xFFFFFFFFu;
      if (wballot(nx)) { if (nx && q[5] <= key) cur = q[31]; }
    } while (wballot(cur != 0xFFFFFFFFu));
    if (!(MIC_X & 2) && __builtin_expect(wballot(remaining <= MIC_R_CROWDED) != 0, 0)) {
      // Rare (a database with microsatellites, a read that overlaps one): the crowded runs of this round become items of the
      // follow-up's work list - the region as the table orients it and the run's range of minimizer positions; the k-mer with
      // its minimizer at position j is the region's nucleotides [ctx - j, ctx - j + k).  One reservation per round.
      // Everything here is kept in VECTOR registers on purpose (the kernarg pointer made opaque, so that the loads from it are
      // vector loads): the entry loop's temporaries are dead at this point, while the scalar file is full - what this block
      // would take of it, the common path would spill and reload per read.
      const bool crowded = remaining <= MIC_R_CROWDED;
      const uint32_t n_c = (uint32_t)__popcll(wballot(crowded));
      uint64_t kpv = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+v"(kpv));
      const MicQueryArgs* kv = (const MicQueryArgs*)kpv;
      uint32_t* cw = kv->crowd;
      uint32_t* itb = kv->crowd_items;
      const uint32_t icap = kv->crowd_item_cap;
      uint32_t ib = 0xFFFFFFFFu;
      if (cw != nullptr && cg != MIC_CG_DENSE) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&cw[1], n_c);
        ib = __builtin_amdgcn_readfirstlane(v);
      }
      asm volatile("" : "+v"(ib));
      const bool room = ib <= icap && n_c <= icap - ib;
      if (crowded && room) {
        uint32_t* it = itb + 8 * ((size_t)ib + below(wballot(crowded)));
        it[0] = G0; it[1] = G1; it[2] = G2; it[3] = (uint32_t)jmin | ((uint32_t)jmax << 8);
        it[4] = cg;                        // (the group in front of this one: read from the group's first item)
      }
      // no work area, or no room in it: the read is recounted by the dense path
      cg = __builtin_amdgcn_readfirstlane(room ? (ib | ((n_c - 1u) << 27)) : MIC_CG_DENSE);
    }
  };
  // end of a read: best / second and the result row - or, for a read with crowded runs, its row so far into the work area
  // (ONE scalar test on the common path; everything about the work area behind it, in vector registers as in the block above)
  auto finish = [&](const RowAcc& acc, uint32_t n_ent, uint32_t total, uint32_t overflow, uint32_t r, const uint32_t cg) __attribute__((always_inline)) {
    uint32_t spilled = 0;
    if (!(MIC_X & 4) && __builtin_expect(cg != MIC_CG_NONE, 0)) {
      uint64_t kpv = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+v"(kpv));
      const MicQueryArgs* kv = (const MicQueryArgs*)kpv;
      uint32_t* cw = kv->crowd;
      uint32_t* pool = kv->crowd_pool;
      const uint32_t pcap = kv->crowd_pend_cap, wcap = kv->crowd_pool_cap;
      uint32_t p = 0xFFFFFFFFu, off = 0;
      if (cg != MIC_CG_DENSE) {
        if (lane == 0) { p = atomicAdd(&cw[0], 1u); off = atomicAdd(&cw[2], 2u * n_ent); }
        p = __builtin_amdgcn_readfirstlane(p); off = __builtin_amdgcn_readfirstlane(off);
      }
      asm volatile("" : "+v"(p), "+v"(off));
      const bool room = off <= wcap && 2u * n_ent <= wcap - off;
      if (p < pcap) {
        uint32_t* pd = cw + MIC_CROWD_HDR + 8 * (size_t)p;
        if (lane == 0) { pd[0] = r; pd[1] = total; pd[2] = room ? (n_ent | (overflow << 8)) : 0xFFFFFFFFu; pd[3] = off; pd[4] = cg; }
        if (room && (uint32_t)lane < n_ent) { uint32_t* pe_ = pool + (size_t)off + 2 * lane; pe_[0] = acc.label1; pe_[1] = acc.count; }
      }
      spilled = __builtin_amdgcn_readfirstlane((p < pcap && room) ? 1u : 0u);
      if (!spilled) {                      // no room (or none wanted): flagged - the dense path recounts the read, side table and all
        if (lane == 0 && cw != nullptr && cg != MIC_CG_DENSE) atomicAdd(&cw[3], 1u);
        overflow = 1;
      }
    }
    if (!spilled) {
      uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      const __attribute__((address_space(4))) MicQueryArgs* kc = (const __attribute__((address_space(4))) MicQueryArgs*)kp;
      struct { uint32_t* results; uint32_t* rows; uint32_t* flagged; uint32_t row_words, flagged_cap; } fa;
      fa.results = kc->results; fa.rows = kc->rows; fa.flagged = kc->flagged; fa.row_words = kc->row_words; fa.flagged_cap = kc->flagged_cap;
      finish_read(acc, n_ent, total, overflow, r, fa, lane);
    }
  };

  // One step = one read taken up (N) and the read before it finished (P: its slots are on their way).  The loop below calls it
  // with the two sets of registers in alternating roles: no copy of the run lanes' state from "next" to "pending".
  uint32_t r = wave0; bool p_valid = false;
  auto step = [&](Round& P, Round& N) __attribute__((always_inline)) -> bool {
    const bool have = r < a.n_reads;
    uint32_t pp = cur_pp;
    const uint32_t pe = cur_pe;
    // A read that is ONE part of at most 128 k-mers in the packer's format (its part ends where the read ends, or a 0 follows -
    // the device packer's reservations, the generator's pitch; the header behind the part sits in the read-ahead entry for reads
    // of up to ~170 nucleotides): no part loop, no chunk loop, the window out of the read-ahead entry.  The scalar unit (one
    // per CU) is what the kernel is short of, and the loops' bookkeeping is scalar work a 150-bp read does not need.
    bool shape = false, simple = false;
    uint32_t n_nrun = 0, plen0 = 0, pp1 = 0;
    if (have) {
      plen0 = cur_hdr; pp1 = pp + 1 + (plen0 + 7) / 8;
      shape = plen0 - (uint32_t)k < 128u && pp1 == pe;       // (k <= plen0 < k + 128 by the unsigned wrap; pp1 == pe implies pp < pe)
      if (!shape && plen0 - (uint32_t)k < 128u && pp1 < pe) {
        const uint32_t rel = pp1 - cur_pp + (uint32_t)((((uint64_t)(cont + cur_pp)) >> 1) & 1);
        if (rel < 24u) {
          const uint32_t v = (ahead_sel ? ahead0 : ahead1)[rel >> 1];
          shape = __builtin_amdgcn_readfirstlane((rel & 1u) ? v >> 16 : v & 0xFFFFu) == 0;
        }
      }
      if (PIPE && shape) {                                    // ... and at most one round of runs: the pipelined road
        uint32_t n_wd;
        const uint32_t R = front(pp + 1, pp1, 0u, plen0 - (uint32_t)k + 1u, true, n_wd);
        if (R <= MIC_RMAX) { setup(n_wd, 0u, R, N, n_nrun); simple = true; }
      }
    }
    // the earlier read: its slots have had the front half above to arrive
    if (PIPE && p_valid) {
      RowAcc acc; acc.label1 = 0; acc.count = 0;
      uint32_t n_ent = 0, overflow = 0, total = 0, cg = MIC_CG_NONE;
      consume(P, acc, n_ent, overflow, total, cg);
      finish(acc, n_ent, total, overflow, r - n_waves, cg);  // (a pending read is the one before this one)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the read-ahead entry taken below)
    }
    if (!have) return false;
    if (!simple) {
      RowAcc acc; acc.label1 = 0; acc.count = 0;
      uint32_t n_ent = 0, overflow = 0, total = 0, cg = MIC_CG_NONE;
      auto chunk = [&](const uint32_t first, const uint32_t cend, const uint32_t base, const uint32_t nk, const bool use_ahead) __attribute__((always_inline)) {
        uint32_t wd;
        const uint32_t R = front(first, cend, base, nk, use_ahead, wd);
        for (uint32_t rbase = 0; rbase < R; rbase += MIC_RMAX) {
          Round L; uint32_t nrun;
          setup(wd, rbase, R, L, nrun);
          issue(L.cur, nrun);
          consume(L, acc, n_ent, overflow, total, cg);
        }
      };
      // (the straight line only where it is the common road: next to the pipelined road it would be a third copy of the back half)
      if (!PIPE && shape) chunk(pp + 1, pp1, 0u, plen0 - (uint32_t)k + 1u, true);
      else {
        bool first_part = true;
        while (pp < pe) {
          // the header of a later part - for nearly every read the 0 that ends it - is among the 24 containers of the read-ahead
          // entry more often than not: an LDS read instead of a global load the whole wave waits for
          uint32_t plen;
          {
            const uint32_t rel = pp - cur_pp + (uint32_t)((((uint64_t)(cont + cur_pp)) >> 1) & 1);    // u16 offset inside the entry
            if (first_part) plen = cur_hdr;
            else if (rel < 24u) {
              const uint32_t v = (ahead_sel ? ahead0 : ahead1)[rel >> 1];
              plen = __builtin_amdgcn_readfirstlane((rel & 1u) ? v >> 16 : v & 0xFFFFu);
            } else plen = __builtin_amdgcn_readfirstlane((uint32_t)cont[pp]);
          }
          const bool ahead_ok = first_part;
          first_part = false;
          if (plen == 0) break;
          const uint32_t first = pp + 1;
          pp = first + (plen + 7) / 8;
          if (plen < (uint32_t)k) continue;
          const uint32_t nk = plen - k + 1;
          const uint32_t cend = pp;
          for (uint32_t base = 0; base < nk; base += 128) chunk(first, cend, base, nk, ahead_ok && base == 0);
        }
      }
      finish(acc, n_ent, total, overflow, r, cg);
    }
    uint32_t t_hdr, t_pp, t_pe;
    __builtin_amdgcn_wave_barrier();
    ahead_take(ahead_sel ? ahead1 : ahead0, n_pp, t_hdr, t_pp, t_pe);
    __builtin_amdgcn_wave_barrier();
    ahead_issue(ahead_sel ? ahead0 : ahead1, t_pp, r + 3 * n_waves);
    ahead_sel ^= 1;
    cur_pp = n_pp; cur_pe = n_pe; cur_hdr = t_hdr; n_pp = t_pp; n_pe = t_pe;
    p_valid = simple;
    if (simple) issue(N.cur, n_nrun);
    r += n_waves;
    return true;
  };
  Round Ra, Rb;
  while (step(Ra, Rb) && step(Rb, Ra)) {}
}

// ---- crowded runs: the follow-up of query_kernel_r ---------------------------------------------------------------------------------
// One wavefront per read that met a crowded minimizer (the pending list of the work area): its row so far comes back into
// registers (lane i = entry i), its crowded runs are probed in the side table four at a time - 16 lanes per run, one lane per
// k-mer: the k-mer with its minimizer at position j is the region's nucleotides [ctx - j, ctx - j + k), oriented as the table
// stores it - the hits are tallied into the row and the read is finished exactly as query_kernel_r finishes the others
// (best / second: CuClarkDB.cu:1440-1459).  Every k-mer occurrence is counted once: the run's k-mers are in the side table
// or nowhere (s_crowd_move_kernel takes ALL entries of a crowded minimizer out of the chains).
__global__ void __launch_bounds__(256) crowd_finish_kernel(const MicQueryArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), n_waves = gridDim.x * 4;
  const uint32_t* __restrict__ cw = a.crowd;
  const uint32_t pcap = a.crowd_pend_cap;
  uint32_t np = cw[0];
  np = np < pcap ? np : pcap;
  const uint32_t* __restrict__ pend = cw + MIC_CROWD_HDR;
  const uint32_t* __restrict__ items = a.crowd_items;
  const uint32_t* __restrict__ pool = a.crowd_pool;
  const uint4* __restrict__ side = a.t.side;
  const uint32_t smask = a.t.side_mask;
  const int k = a.t.k, ctx = k - a.t.m;
  for (uint32_t p = wave0; p < np; p += n_waves) {
    const uint4 h = *(const uint4*)(pend + 8 * (size_t)p);
    const uint32_t r = __builtin_amdgcn_readfirstlane(h.x), meta = __builtin_amdgcn_readfirstlane(h.z), off = __builtin_amdgcn_readfirstlane(h.w);
    uint32_t total = __builtin_amdgcn_readfirstlane(h.y);
    uint32_t g = __builtin_amdgcn_readfirstlane(pend[8 * (size_t)p + 4]);
    if (meta == 0xFFFFFFFFu) continue;                       // (no room for its row: the main kernel sent the read to the dense path)
    uint32_t n_ent = meta & 0xFFu, overflow = (meta >> 8) & 1u;
    RowAcc acc; acc.label1 = 0; acc.count = 0;
    if ((uint32_t)lane < n_ent) { const uint2 e = *(const uint2*)(pool + off + 2 * lane); acc.label1 = e.x; acc.count = e.y; }
    while (g != MIC_CG_NONE) {
      const uint32_t ib = g & 0x7FFFFFFu, n_c = (g >> 27) + 1u;
      g = __builtin_amdgcn_readfirstlane(items[8 * (size_t)ib + 4]);
      for (uint32_t b = 0; b < n_c; b += 4) {
        const uint32_t idx = b + ((uint32_t)lane >> 4);
        const bool valid = idx < n_c;
        uint4 it = make_uint4(0, 0, 0, 0);
        if (valid) it = *(const uint4*)(items + 8 * (size_t)(ib + idx));
        const int jmin = (int)(it.w & 0xFFu), jmax = (int)((it.w >> 8) & 0xFFu);
        int j = jmin + (lane & 15);
        bool go = valid && j <= jmax;
        j = go ? j : jmin;
        const uint64_t K = s_extract(it.x, it.y, it.z, ctx - j, k);
        uint32_t hsh = s_side_hash(K, smask), got = 0;
        while (wballot(go)) {
          uint4 c = make_uint4(0, 0, 0, 0);
          if (go) c = load_slot_quarter(side + hsh);
          if (go) {
            if (c.z == 0) go = false;
            else if (c.x == (uint32_t)K && c.y == (uint32_t)(K >> 32)) { got = c.z; go = false; }
            else hsh = (hsh + 1) & smask;
          }
        }
        tally2(got, 0u, acc, n_ent, overflow, total, lane);
      }
    }
    finish_read(acc, n_ent, total, overflow, r, a, lane);
  }
}


// ---- merge / result on sparse rows ----------------------------------------------------------------
// mergeKernel (CuClarkDB.cu:1321-1415): one thread per read, two-pointer merge by ascending target.
__global__ void merge_rows_kernel(const uint32_t* __restrict__ ra, const uint32_t* __restrict__ rb,
                                  uint32_t* __restrict__ out, uint32_t row_words, size_t n,
                                  uint32_t* __restrict__ results) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* a = ra + i * row_words; const uint32_t* b = rb + i * row_words; uint32_t* o = out + i * row_words;
  uint32_t na = a[0], nb = b[0], ia = 0, ib = 0, no = 0;
  bool ok = na != MIC_ROW_INVALID && nb != MIC_ROW_INVALID;
  if (!ok) { na = 0; nb = 0; }
  while (ia < na || ib < nb) {
    uint32_t va = ia < na ? a[1 + ia] : 0xFFFFFFFFu, vb = ib < nb ? b[1 + ib] : 0xFFFFFFFFu;
    uint32_t ta = ia < na ? (va & 0xFFFF) : 0x10000u, tb = ib < nb ? (vb & 0xFFFF) : 0x10000u;
    uint32_t tgt, cnt;
    if (ta < tb) { tgt = ta; cnt = va >> 16; ++ia; }
    else if (tb < ta) { tgt = tb; cnt = vb >> 16; ++ib; }
    else { tgt = ta; cnt = (va >> 16) + (vb >> 16); ++ia; ++ib; }
    if (no + 1 < row_words && cnt <= 0xFFFF) o[1 + no] = (cnt << 16) | tgt; else ok = false;
    ++no;
  }
  o[0] = ok ? no : MIC_ROW_INVALID;
  (void)results;
}

// resultKernel (CuClarkDB.cu:1421-1471) on a sparse row.
__global__ void result_rows_kernel(const uint32_t* __restrict__ rows, uint32_t row_words,
                                   uint32_t* __restrict__ results, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* row = rows + i * row_words;
  uint32_t cnt = row[0], sum = 0, best = 0, ib = 0, sb = 0, is = 0, flags = 0;
  if (cnt == MIC_ROW_INVALID) { cnt = 0; flags = MIC_FLAG_ROW_OVERFLOW_; }
  for (uint32_t e = 0; e < cnt; ++e) {
    uint32_t v = row[1 + e], tgt = v & 0xFFFF, sc = v >> 16;
    if (sc > best) { sb = best; is = ib; best = sc; ib = tgt + 1; }
    else if (sc > sb) { sb = sc; is = tgt + 1; }
    sum += sc;
  }
  uint4* out = (uint4*)(results + i * 8);
  out[0] = make_uint4(sum, ib, best, is);
  out[1] = make_uint4(sb, cnt, flags, 0);
}

// ---- dense fallback: exact for any read (any length, any number of targets) ------------------------
// One 256-thread block per listed read; every thread walks k-mer positions t, t+256, ... of each part,
// probes the slot chain sequentially and atomically increments counts[id][label].
template <bool KEY64>
__device__ inline uint32_t probe_scalar(const MicTable& t, uint64_t kmer) {
  constexpr uint32_t CAP = KEY64 ? MIC_CAP64 : MIC_CAP32;
  uint64_t c = canonical(kmer, t.k);
  uint64_t q = mic_div(c, t.div);
  uint64_t rem = c - q * t.div.d;
  if (rem < t.shard_start || rem >= t.shard_end) return 0;
  if (!KEY64 && (q >> 32) != 0) return 0;
  uint64_t slot = rem - t.shard_start;
  for (;;) {
    uint4 qq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qq[j] = t.slots[slot * 4 + j];
    uint32_t n = qq[0].w & 0xFF;
    uint64_t lastk = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t m = match_quarter<KEY64>(qq[j], (uint32_t)q, (uint32_t)(q >> 32), j);
      if (m) return m;
    }
    lastk = KEY64 ? (((uint64_t)qq[3].y << 32) | qq[3].x) : (uint64_t)qq[3].y;
    if (n <= CAP || q <= lastk) return 0;
    slot = (uint64_t)(qq[0].w >> 8) | ((uint64_t)(qq[1].w >> 8) << 24);
  }
}

// Sequential probe of the minimizer-keyed table (dense fallback, statistics).
__device__ inline uint32_t probe_scalar_m(const MicTable& t, uint64_t kmer) {
  uint64_t c = canonical(kmer, t.k);
  if (t.sharded) {
    uint64_t q = mic_div(c, t.div);
    uint64_t rem = c - q * t.div.d;
    if (rem < t.shard_start || rem >= t.shard_end) return 0;
  }
  uint64_t slot = mslot_of_kmer(kmer, t.k, t.m, (uint32_t)t.n_main);
  for (;;) {
    const uint4* q = t.slots + slot * 8;
    const uint4 meta = q[7];
    const uint32_t n = MIC_M_N(meta.z);
    uint32_t cnt = 0;  // keys <= c
    bool eq = false;
    for (uint32_t e = 0; e < n; ++e) {
      uint4 kv = q[e >> 1];
      uint64_t key = (e & 1) ? (((uint64_t)kv.w << 32) | kv.z) : (((uint64_t)kv.y << 32) | kv.x);
      if (key <= c) { ++cnt; eq = key == c; }
    }
    if (!(meta.z & MIC_M_DIR)) {
      if (!eq) return 0;
      const uint32_t e = cnt - 1;
      uint4 lw = q[6 + (e >> 3)];
      uint32_t wi = (e & 7) >> 1;
      uint32_t word = wi == 0 ? lw.x : wi == 1 ? lw.y : wi == 2 ? lw.z : lw.w;
      return ((e & 1) ? (word >> 16) : (word & 0xFFFF)) + 1;
    }
    if (cnt == 0) return 0;                     // smaller than every key below this directory
    slot = (uint64_t)q[6].x + (cnt - 1);        // descend into the child whose range holds c
  }
}

// kmer = the k-mer as it reads, tpos = its first nucleotide's position in its read part (the super-k-mer tables resolve
// tied minimizers by position, s_probe_read); *mine (optional) = this engine answers for the k-mer
template <bool KEY64>
__device__ inline uint32_t probe_any(const MicTable& t, uint64_t kmer, uint32_t tpos, bool* mine = nullptr) {
  if (mine) *mine = true;
  if (t.layout == 2) {
    if (t.sharded) {
      const uint64_t c = canonical(kmer, t.k);
      const uint64_t q = mic_div(c, t.div), rem = c - q * t.div.d;
      if (rem < t.shard_start || rem >= t.shard_end) { if (mine) *mine = false; return 0; }
    }
    bool in_part = true;
    const uint32_t r = s_probe_read(t.slots, (uint32_t)t.n_main, t.parted != 0, t.slot_lo, t.slot_cnt, kmer, tpos, t.k, t.m, t.fwd != 0, &in_part,
                                    t.side, t.side_mask);
    if (mine && !in_part) *mine = false;
    return r;
  }
  if (mine && (t.sharded || t.layout == 0)) {
    const uint64_t c = canonical(kmer, t.k);
    const uint64_t q = mic_div(c, t.div), rem = c - q * t.div.d;
    *mine = rem >= t.shard_start && rem < t.shard_end;
  }
  return t.layout ? probe_scalar_m(t, kmer) : probe_scalar<KEY64>(t, kmer);
}

template <bool KEY64>
__global__ void __launch_bounds__(256) dense_count_kernel(const MicTable t, const uint32_t* __restrict__ reads_ptr,
                                                          const uint16_t* __restrict__ cont,
                                                          const uint32_t* __restrict__ ids, uint32_t n_targets,
                                                          uint32_t* __restrict__ counts) {
  const uint32_t r = ids ? ids[blockIdx.x] : blockIdx.x;
  uint32_t* row = counts + (size_t)blockIdx.x * n_targets;
  uint32_t pp = reads_ptr[r];
  const uint32_t pe = reads_ptr[r + 1];
  const int k = t.k;
  while (pp < pe) {
    const uint32_t plen = cont[pp];
    if (plen == 0) break;
    const uint32_t first = pp + 1;
    pp = first + (plen + 7) / 8;
    if (plen < (uint32_t)k) continue;
    const uint32_t nk = plen - k + 1;
    for (uint32_t pos = threadIdx.x; pos < nk; pos += blockDim.x) {
      // nucleotides [pos, pos+k): containers pos/8 .. (pos+k-1)/8, at most 5
      uint32_t c0 = first + pos / 8;
      uint64_t hi = 0; uint32_t lo = 0;  // 80-bit window hi(64) : lo(16)
#pragma unroll
      for (int i = 0; i < 4; ++i) hi = (hi << 16) | (c0 + i < pp ? cont[c0 + i] : 0);
      lo = c0 + 4 < pp ? cont[c0 + 4] : 0;
      int s = 2 * (pos & 7);
      uint64_t x = s ? ((hi << s) | ((uint64_t)lo >> (16 - s))) : hi;
      uint64_t kmer = x >> (64 - 2 * k);
      uint32_t m = probe_any<KEY64>(t, kmer, pos);
      if (m && m - 1 < n_targets) atomicAdd(&row[m - 1], 1u);
    }
  }
}

// One block per listed read: scan its dense counts (ascending target) -> result row (+ sparse row if it fits).
__global__ void __launch_bounds__(256) dense_finish_kernel(const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ ids, uint32_t n_targets,
                                                           uint32_t* __restrict__ results, uint32_t* __restrict__ rows,
                                                           uint32_t row_words) {
  const uint32_t r = ids ? ids[blockIdx.x] : blockIdx.x;
  const uint32_t* row = counts + (size_t)blockIdx.x * n_targets;
  __shared__ unsigned long long s_best[256], s_second[256];
  __shared__ uint32_t s_sum[256], s_n[256];
  unsigned long long best = 0, second = 0; uint32_t sum = 0, nz = 0;
  for (uint32_t tg = threadIdx.x; tg < n_targets; tg += blockDim.x) {
    uint32_t c = row[tg];
    if (!c) continue;
    unsigned long long key = ((unsigned long long)c << 16) | (0xFFFFu - tg);
    if (key > best) { second = best; best = key; } else if (key > second) second = key;
    sum += c; ++nz;
  }
  s_best[threadIdx.x] = best; s_second[threadIdx.x] = second; s_sum[threadIdx.x] = sum; s_n[threadIdx.x] = nz;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = 0; second = 0; sum = 0; nz = 0;
    for (int i = 0; i < 256; ++i) {
      unsigned long long c2[2] = {s_best[i], s_second[i]};
      for (int u = 0; u < 2; ++u) {
        unsigned long long key = c2[u];
        if (key > best) { second = best; best = key; } else if (key > second) second = key;
      }
      sum += s_sum[i]; nz += s_n[i];
    }
    uint32_t flags = MIC_FLAG_DENSE_PATH_;
    if (rows) {
      uint32_t* orow = rows + (size_t)r * row_words;
      bool fits = nz <= row_words - 1;
      if (fits) {
        uint32_t no = 0;
        for (uint32_t tg = 0; tg < n_targets && fits; ++tg) {
          uint32_t c = row[tg];
          if (!c) continue;
          if (c > 0xFFFF) { fits = false; break; }
          orow[1 + no++] = (c << 16) | tg;
        }
        if (fits) orow[0] = nz;
      }
      if (!fits) { orow[0] = MIC_ROW_INVALID; flags |= MIC_FLAG_ROW_OVERFLOW_; }
    }
    uint4* out = (uint4*)(results + (size_t)r * 8);
    out[0] = make_uint4(sum, best ? 0x10000u - (uint32_t)(best & 0xFFFF) : 0, (uint32_t)(best >> 16),
                        second ? 0x10000u - (uint32_t)(second & 0xFFFF) : 0);
    out[1] = make_uint4((uint32_t)(second >> 16), nz, flags, 0);
  }
}

// Diagnostics for the roofline bookkeeping (not on the timed path): one thread per read walks its k-mers and
// accumulates {k-mers, k-mers probed in this shard, hits, sum of the probed buckets' lengths}.
template <bool KEY64>
__global__ void __launch_bounds__(256) probe_stats_kernel(const MicTable t, const uint32_t* __restrict__ reads_ptr,
                                                          const uint16_t* __restrict__ cont, uint32_t n_reads,
                                                          unsigned long long* __restrict__ out) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nk = 0, np = 0, nh = 0, nb = 0;
  if (r < n_reads) {
    uint32_t pp = reads_ptr[r];
    const uint32_t pe = reads_ptr[r + 1];
    const int k = t.k;
    const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    while (pp < pe) {
      const uint32_t plen = cont[pp];
      if (plen == 0) break;
      const uint32_t first = pp + 1;
      pp = first + (plen + 7) / 8;
      uint64_t kmer = 0;
      for (uint32_t i = 0; i < plen; ++i) {
        uint32_t nt = (cont[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
        kmer = ((kmer << 2) | nt) & cutoff;
        if (i + 1 < (uint32_t)k) continue;
        ++nk;
        uint64_t c = canonical(kmer, k);
        uint64_t q = mic_div(c, t.div);
        uint64_t rem = c - q * t.div.d;
        if (rem < t.shard_start || rem >= t.shard_end) continue;
        bool mine = true;
        const bool hit = probe_any<KEY64>(t, kmer, i + 1 - (uint32_t)k, &mine) != 0;
        if (!mine) continue;     // a slot-range part: the k-mer's slot is resident on another engine
        ++np;
        nb += t.layout ? (t.sizes ? t.sizes[rem - t.shard_start] : 0)
                                               : (t.slots[(rem - t.shard_start) * 4].w & 0xFF);
        nh += hit;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    nk += __shfl_down(nk, off); np += __shfl_down(np, off); nh += __shfl_down(nh, off); nb += __shfl_down(nb, off);
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], nk); atomicAdd(&out[1], np); atomicAdd(&out[2], nh); atomicAdd(&out[3], nb); }
}

}  // namespace

hipError_t mic_launch_probe_stats(const MicTable& t, int slot_class, const uint32_t* reads_ptr, const uint16_t* cont,
                                  size_t n_reads, unsigned long long* d_out, hipStream_t s) {
  if (!n_reads) return hipSuccess;
  unsigned blocks = (unsigned)((n_reads + 255) / 256);
  if (slot_class == 64) probe_stats_kernel<true><<<blocks, 256, 0, s>>>(t, reads_ptr, cont, (uint32_t)n_reads, d_out);
  else probe_stats_kernel<false><<<blocks, 256, 0, s>>>(t, reads_ptr, cont, (uint32_t)n_reads, d_out);
  return hipGetLastError();
}

static bool per_kmer_env() { static const bool v = getenv("MIC_S_PER_KMER") != nullptr; return v; }

// The runtime loads this file's device code (the query kernels: megabytes of instantiations) when its first kernel is launched;
// the table load does that, so that the first batch does not (4 ms, measured on the ingest path's file)
__global__ void kernels_warm_kernel(uint32_t* p) { if (p && threadIdx.x == 1000) *p = 0; }
hipError_t mic_kernels_warm(hipStream_t s) {
  kernels_warm_kernel<<<1, 64, 0, s>>>(nullptr);
  return hipGetLastError();
}

hipError_t mic_launch_query(const MicQueryArgs& a_in, int slot_class, int n_cu, hipStream_t s) {
  if (a_in.n_reads == 0) return hipSuccess;
  MicQueryArgs a = a_in;
  // test hook: a work area of the crowded runs' follow-up that holds this many pending reads / runs / row words at most, so that
  // a small parity case overflows it (it can only shrink the capacities: what does not fit takes the dense path)
  if (a.crowd) if (const char* e = getenv("MIC_CROWD_CAP")) {
    const long v = atol(e);
    if (v >= 0) {
      if ((uint32_t)v < a.crowd_pend_cap) a.crowd_pend_cap = (uint32_t)v;
      if ((uint32_t)v < a.crowd_item_cap) a.crowd_item_cap = (uint32_t)v;
      if ((uint32_t)v < a.crowd_pool_cap) a.crowd_pool_cap = (uint32_t)v;
    }
  }
  unsigned blocks = (a.n_reads + 3) / 4;
  // Resident blocks per CU are 8; the grid is much larger so that each wave strides over only ~20 reads: reads differ in
  // cost (rounds, parts) and the hardware's block scheduler then evens the waves out.  Measured on the headline
  // workload: 8 blocks/CU 697 Mreads/s, 32: 760, 128: 811, 512: 835, 1024: 821 (direct layout: flat); round 4, per-run kernel:
  // 96: 5.08 ms, 128: 5.07, 192: 5.04, 256: 5.02, 320: 5.05, 512: 5.07, 1024: 5.18, 2048: 5.24.
  static int per_cu = [] { const char* e = getenv("MIC_BLOCKS_PER_CU"); int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
  unsigned cap = (unsigned)n_cu * (unsigned)per_cu;
  if (blocks > cap) blocks = cap;
  {
    // Small launches (a CLI batch): a wave pays two dependent loads before its first read, so do not shrink below ~8
    // reads per wave unless that would leave resident block slots empty.
    const unsigned fill = (unsigned)n_cu * 8u;                       // one generation of resident blocks
    static const bool grid_dbg = getenv("MIC_GRID_DEBUG") != nullptr;   // measuring runs (tools/small_launch_probe.py): reads per wave from the environment, per call
    unsigned rpw = 8;
    if (grid_dbg) if (const char* e = getenv("MIC_READS_PER_WAVE")) { const int v = atoi(e); if (v > 0) rpw = (unsigned)v; }
    const unsigned by_work = (unsigned)((a.n_reads + 4 * rpw - 1) / (4 * rpw));      // 8 reads per wave
    unsigned want = by_work > fill ? by_work : fill;
    if (blocks > want) blocks = want;
  }
  // test hook: a grid of this many blocks, so that a small parity case runs hundreds of reads of every shape through each wavefront
  // (the software-pipelined loop's transitions between its pipelined and its plain road)
  // (it can only shrink the grid: a stray value in a user's environment costs speed, never a launch dimension out of range)
  if (const char* e = getenv("MIC_QUERY_BLOCKS")) { const int v = atoi(e); if (v > 0 && (unsigned)v < blocks) blocks = (unsigned)v; }
  if (a.t.layout == 2) {
    const unsigned g = (blocks * 4 + MIC_M_WPB - 1) / MIC_M_WPB, b = 64 * MIC_M_WPB;
    static const bool generic = getenv("MIC_S_GENERIC") != nullptr;
    // pt: slot-range part (one compare per run in query_kernel_r; the per-k-mer kernel filters in its `sharded` instantiation)
    const bool pt = a.t.parted != 0, fw = a.t.fwd != 0;
    const bool sh = a.t.sharded != 0 || (pt && (per_kmer_env() || (!fw && !(2 * a.t.k - a.t.m > 32 && 2 * a.t.k - a.t.m <= 48))));
    // instantiations: k and m as constants for cuCLARK's 31, cuCLARK-l's 27 and k = 32 with m = 20; the table-sharded filter
    // and the two-strand table each in their own (the common kernel carries neither's scalars)
    // the two-strand table is probed per run (query_kernel_r); MIC_S_PER_KMER=1 keeps the per-k-mer kernel for comparison
    const bool per_kmer = per_kmer_env();
    // the one-strand table's per-run kernel reverse-complements a region of 2k - m nucleotides inside three words: its
    // realignment is a single funnel shift when the region is longer than 32 nucleotides (always for cuCLARK's k and m)
    const bool run_ok = 2 * a.t.k - a.t.m > 32 && 2 * a.t.k - a.t.m <= 48;
    static const unsigned extra_lds = [] { const char* e = getenv("MIC_EXTRA_LDS"); return e ? (unsigned)atoi(e) : 0u; }();   // occupancy experiments
    // crowded minimizers in a side table: the per-run kernel hands their runs to crowd_finish_kernel through the work area
    // (without one such reads take the dense path: exact, slow); the per-k-mer kernel looks them up itself
    const bool crowd = a.t.side != nullptr && a.crowd != nullptr;
    bool ran_r = false;
    if (crowd) { const hipError_t ce = hipMemsetAsync(a.crowd, 0, MIC_CROWD_HDR * 4, s); if (ce != hipSuccess) return ce; }
#define LAUNCH_R(KK_, MM_, FW_) do { \
      if (pt) query_kernel_r<KK_, MM_, FW_, true><<<g, b, extra_lds, s>>>(a); else query_kernel_r<KK_, MM_, FW_, false><<<g, b, extra_lds, s>>>(a); \
      ran_r = true; } while (0)
#define LAUNCH_S(KK_, MM_) do { \
      if (fw && !sh && !per_kmer) LAUNCH_R(KK_, MM_, true); \
      else if (!fw && !sh && !per_kmer && run_ok) LAUNCH_R(KK_, MM_, false); \
      else if (fw) { if (sh) query_kernel_s<KK_, MM_, true, true><<<g, b, 0, s>>>(a); else query_kernel_s<KK_, MM_, false, true><<<g, b, 0, s>>>(a); } \
      else { if (sh) query_kernel_s<KK_, MM_, true, false><<<g, b, 0, s>>>(a); else query_kernel_s<KK_, MM_, false, false><<<g, b, 0, s>>>(a); } } while (0)
    if (!generic && a.t.k == 31 && a.t.m == 20) LAUNCH_S(31, 20);
    else if (!generic && a.t.k == 27 && a.t.m == 20) LAUNCH_S(27, 20);
    else if (!generic && a.t.k == 32 && a.t.m == 20) LAUNCH_S(32, 20);
    else LAUNCH_S(0, 0);
#undef LAUNCH_S
#undef LAUNCH_R
    if (ran_r && crowd) {
      // the number of pending reads is on the device only: a grid for the most there can be, at most two generations of resident
      // blocks; a launch with nothing to do costs a few microseconds behind the main kernel
      const unsigned most = a.n_reads < a.crowd_pend_cap ? a.n_reads : a.crowd_pend_cap;
      unsigned cg = (most + 3) / 4, ccap = (unsigned)n_cu * 16u;
      if (cg > ccap) cg = ccap;
      if (cg) crowd_finish_kernel<<<cg, 256, 0, s>>>(a);
    }
  }
  else if (a.t.layout) {
    {
      const unsigned g = (blocks * 4 + MIC_M_WPB - 1) / MIC_M_WPB, b = 64 * MIC_M_WPB;
      if (MIC_SPEC && a.t.k == 31 && a.t.m == 20) query_kernel_m<31, 20><<<g, b, 0, s>>>(a);        // cuCLARK
      else if (MIC_SPEC && a.t.k == 27 && a.t.m == 20) query_kernel_m<27, 20><<<g, b, 0, s>>>(a);   // cuCLARK-l
      else query_kernel_m<0, 0><<<g, b, 0, s>>>(a);
    }
  }
  else if (slot_class == 64) query_kernel<true><<<blocks, 256, 0, s>>>(a);
  else query_kernel<false><<<blocks, 256, 0, s>>>(a);
  return hipGetLastError();
}

// The instantiation mic_launch_query launches for table t, spelled the way a kernel trace spells it (same decisions, same order).
int mic_query_kernel_name(const MicTable& t, int slot_class, char* buf, size_t cap) {
  if (t.layout == 2) {
    static const bool generic = getenv("MIC_S_GENERIC") != nullptr;
    const bool pt = t.parted != 0, fw = t.fwd != 0, per_kmer = per_kmer_env();
    const bool run_ok = 2 * t.k - t.m > 32 && 2 * t.k - t.m <= 48;
    const bool sh = t.sharded != 0 || (pt && (per_kmer || (!fw && !run_ok)));
    const bool spec = !generic && t.m == 20 && (t.k == 31 || t.k == 27 || t.k == 32);
    const int kk = spec ? t.k : 0, mm = spec ? t.m : 0;
    const char* b[2] = {"false", "true"};
    if (!sh && !per_kmer && (fw || run_ok))
      return snprintf(buf, cap, "query_kernel_r<%d, %d, %s, %s>", kk, mm, b[fw], b[pt]);
    return snprintf(buf, cap, "query_kernel_s<%d, %d, %s, %s>", kk, mm, b[sh], b[fw]);
  }
  if (t.layout) {
    const bool spec = MIC_SPEC && t.m == 20 && (t.k == 31 || t.k == 27);
    return snprintf(buf, cap, "query_kernel_m<%d, %d>", spec ? t.k : 0, spec ? t.m : 0);
  }
  return snprintf(buf, cap, "query_kernel<%s>", slot_class == 64 ? "true" : "false");
}

hipError_t mic_launch_merge_rows(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t row_words, size_t n,
                                 uint32_t* flags_results, hipStream_t s) {
  if (!n) return hipSuccess;
  merge_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, b, out, row_words, n, flags_results);
  return hipGetLastError();
}

hipError_t mic_launch_result_from_rows(const uint32_t* rows, uint32_t row_words, uint32_t* results, size_t n,
                                       hipStream_t s) {
  if (!n) return hipSuccess;
  result_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(rows, row_words, results, n);
  return hipGetLastError();
}

hipError_t mic_launch_dense_count(const MicTable& t, int slot_class, const uint32_t* reads_ptr, const uint16_t* cont,
                                  const uint32_t* ids, size_t n_ids, uint32_t n_targets, uint32_t* counts,
                                  hipStream_t s) {
  if (!n_ids) return hipSuccess;
  hipError_t e = hipMemsetAsync(counts, 0, n_ids * (size_t)n_targets * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  if (slot_class == 64) dense_count_kernel<true><<<(unsigned)n_ids, 256, 0, s>>>(t, reads_ptr, cont, ids, n_targets, counts);
  else dense_count_kernel<false><<<(unsigned)n_ids, 256, 0, s>>>(t, reads_ptr, cont, ids, n_targets, counts);
  return hipGetLastError();
}

hipError_t mic_launch_dense_finish(const uint32_t* counts, const uint32_t* ids, size_t n_ids, uint32_t n_targets,
                                   uint32_t* results, uint32_t* rows, uint32_t row_words, hipStream_t s) {
  if (!n_ids) return hipSuccess;
  dense_finish_kernel<<<(unsigned)n_ids, 256, 0, s>>>(counts, ids, n_targets, results, rows, row_words);
  return hipGetLastError();
}
