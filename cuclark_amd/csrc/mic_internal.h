// mic_internal.h — shared between the HIP kernels and the engine (not part of the public ABI).
#ifndef MIC_INTERNAL_H
#define MIC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// ---- exact 64-bit division by a runtime-constant divisor (HTSIZE) -------------------------------
// Granlund–Montgomery "round-up" magic, computed once on the host (mic_make_div):
//   magic == 0 : divisor is a power of two, q = n >> shift
//   add == 0   : q = mulhi(magic, n) >> shift
//   add == 1   : q = (((n - t) >> 1) + t) >> shift,  t = mulhi(magic, n)
// Replaces the reference's compile-time-constant division `c / HTSIZE` (CuClarkDB.cu:1268-1269):
// HTSIZE is a runtime value here (= size of .sz), so one binary serves cuCLARK and cuCLARK-l.
struct MicDiv {
  uint64_t d;
  uint64_t magic;
  uint32_t shift;
  uint32_t add;
};

MicDiv mic_make_div(uint64_t d);

#ifdef __cplusplus
#include <vector>
// mic_host.cpp: mic_pack_reads plus, per part, the offset of its first nucleotide in its ACGTU run and the run's length
size_t mic_pack_reads_runs(const uint8_t* map, const uint64_t* seq_s, const uint64_t* seq_e, const uint64_t* length,
                           size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap,
                           std::vector<uint64_t>* run_off, std::vector<uint64_t>* run_len);
#endif

static inline __host__ __device__ uint64_t mic_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

static inline __host__ __device__ uint64_t mic_div(uint64_t n, const MicDiv& dv) {
  if (dv.magic == 0) return n >> dv.shift;
  uint64_t t = mic_mulhi64(dv.magic, n);
  if (dv.add) return (((n - t) >> 1) + t) >> dv.shift;
  return t >> dv.shift;
}

// ---- resident table --------------------------------------------------------------------------
// One 64-byte slot per bucket = 4 quarters of 16 bytes; a quad of 4 lanes loads one slot with a single
// wave-instruction (global_load_dwordx4), i.e. ONE random 64-byte request per probe.
//   slot class 32 (quotients < 2^32; u16/u32 keys on disk): quarter j = { key[2j], key[2j+1],
//        label[2j] | label[2j+1] << 16, meta }        -> 8 entries inline
//   slot class 64 (u64 keys on disk): quarter j = { key_lo, key_hi, label, meta } -> 4 entries inline
//   meta: bits 0..7 = n, entries stored from this slot onwards along the chain (saturating at 255);
//         bits 8..31 of quarters 0 and 1 = low / high 24 bits of the absolute index of the next slot
//         of the chain (used only when n > capacity).
// Entries are the bucket's keys that the reference's linear scan can reach (strict prefix maxima
// that are <= the bucket's last key, CuClarkDB.cu:1291-1307), in ascending order.
#define MIC_SLOT_BYTES 64
#define MIC_FLAG_ROW_OVERFLOW_ 1u  /* == MIC_FLAG_ROW_OVERFLOW */
#define MIC_FLAG_DENSE_PATH_ 2u    /* == MIC_FLAG_DENSE_PATH   */
#define MIC_ROW_INVALID 0xFFFFFFFFu     /* row[0] of a sparse row that did not fit */
#define MIC_CAP32 8
#define MIC_CAP64 4

// ---- minimizer-keyed table ("M-table", layout 1) ---------------------------------------------------------------
// A k-mer's slot is chosen by the smallest 32-bit order key among its w = k-m+1 canonical m-mers (mic_device.h), so
// the ~ (w+1)/2 consecutive k-mers of a read that share a minimizer share ONE slot: the query kernel loads each
// distinct slot of a 128-k-mer chunk once from HBM into LDS (8 lanes x 16 B) and every k-mer compares against the
// staged entries.  Slots hold full canonical k-mers, so membership and labels are exact.
//   128-byte slot: keys u64[12] (ascending along the chain, unused = ~0) | labels u16[12] | meta u32 | next u32
//   meta: bits 0..7 = entries in this slot, bit 8 = chain continues at slot `next` (overflow slots are contiguous)
#define MIC_MCAP 12
#define MIC_MSLOT_BYTES 128
#ifndef MIC_LAYOUT_MINIMIZER
#define MIC_LAYOUT_AUTO 0
#define MIC_LAYOUT_DIRECT 1
#define MIC_LAYOUT_MINIMIZER 2
#endif
#ifndef MIC_LAYOUT_SUPER
#define MIC_LAYOUT_SUPER 3
#endif
#ifndef MIC_LAYOUT_SUPER2
#define MIC_LAYOUT_SUPER2 4
#endif

struct MicTable {
  const uint4* slots;    // layout 0: 4 x uint4 per slot; layout 1: 8 x uint4 per slot
  uint64_t n_main;       // layout 0: shard_end - shard_start; layout 1: number of main M-slots
  uint64_t shard_start;  // first bucket of the shard
  uint64_t shard_end;
  MicDiv div;            // division by htsize
  int k;
  int layout;            // 0 = direct slots, 1 = minimizer-keyed slots, 2 = super-k-mer slots
  int m;                 // minimizer length (layout 1)
  int sharded;           // 1 if [shard_start, shard_end) is a strict subset of the table
  int fwd;               // layout 2: both strands stored under forward-strand minimizers (mic_device.h: s_candidates_fwd)
  // layout 2, slot-range part (table-sharded runs): only the main slots [slot_lo, slot_lo + slot_cnt) of the n_main the
  // minimizer hash spreads over are resident; `slots` is then the allocation MINUS slot_lo slots, so global slot indices
  // (and the continuation indices stored in word 31) address it directly.  slot_cnt == 0: the whole table is here.
  uint32_t slot_lo, slot_cnt;
  // layout 2: the k-mers of crowded minimizers, keyed by the oriented k-mer: cells {lo, hi, label + 1, 0}, label + 1 == 0 is
  // empty, linear probing, side_mask + 1 cells (a power of two); null when the table has none
  const uint4* side; uint32_t side_mask;
  int parted;            // 1: slot-range part as above (slot_cnt may be 0: an empty part answers nothing)
  const uint8_t* sizes;  // kept copy of the shard's on-disk bucket sizes (statistics only)
};

struct MicQueryArgs {
  MicTable t;
  const uint32_t* reads_ptr;
  const uint16_t* cont;
  uint32_t n_reads;
  uint32_t row_words;   // 0 => no rows
  uint32_t* results;    // n_reads * 8
  uint32_t* rows;       // n_reads * row_words or nullptr
  uint32_t* flagged;    // [0] = count, [1..] = read ids needing the dense path (or nullptr)
  uint32_t flagged_cap;
  // layout 2 with a side table: the work area through which query_kernel_r hands the runs of crowded minimizers (and the rows
  // of their reads) to crowd_finish_kernel - mic_crowd_dims / mic_crowd_attach below - or nullptr: such reads take the dense path
  uint32_t* crowd;
  uint32_t* crowd_items; uint32_t* crowd_pool;      // = crowd + MIC_CROWD_HDR + 8 pend_cap, ... + 8 item_cap (the kernels do no address arithmetic with the capacities)
  uint32_t crowd_pend_cap, crowd_item_cap, crowd_pool_cap;
};

// ---- work area of the crowded runs' follow-up (mic_kernels.hip: query_kernel_r -> crowd_finish_kernel) ----------------------
// One per query launch in flight, any 16-byte aligned device memory; mic_launch_query zeroes the header.  Words:
//   [0] pending reads  [1] items  [2] pool words  [3] reads sent to the dense path for lack of room  [4..7] unused
//   pending read p (8 words): read, hits so far, entries | overflow << 8 (0xFFFFFFFF: skipped), pool offset of its row, last group
//   item i (8 words): the run's region (3 words, oriented as the table stores it), jmin | jmax << 8, the group in front (first item of a group)
//   pool: the rows so far, (label + 1, count) per entry
// A group = the crowded runs of one round of one read, adjacent items: first item | (runs - 1) << 27.
#define MIC_CROWD_HDR 8
#define MIC_CG_NONE 0xFFFFFFFFu
#define MIC_CG_DENSE 0xFFFFFFFEu
struct MicCrowdDims { uint32_t pend_cap, item_cap, pool_cap; size_t words; };
static inline MicCrowdDims mic_crowd_dims(size_t max_reads) {
  // a quarter of the reads pending with four crowded runs and eight row words each (48 bytes per read of capacity); what does
  // not fit takes the dense path (exact, slow: real data has a few per cent of such reads, a run or two each)
  MicCrowdDims d;
  const size_t cap27 = (1u << 27) - 64;
  size_t pend = max_reads / 4 + 1024, item = max_reads + 4096, pool = max_reads * 2 + 8192;
  d.pend_cap = (uint32_t)(pend < cap27 ? pend : cap27);
  d.item_cap = (uint32_t)(item < cap27 ? item : cap27);
  d.pool_cap = (uint32_t)(pool < 0xFFFFFF00u ? pool : 0xFFFFFF00u);
  d.words = MIC_CROWD_HDR + 8 * ((size_t)d.pend_cap + d.item_cap) + d.pool_cap;
  return d;
}
static inline void mic_crowd_attach(MicQueryArgs& a, uint32_t* area, size_t max_reads) {
  const MicCrowdDims d = mic_crowd_dims(max_reads);
  a.crowd = area; a.crowd_pend_cap = area ? d.pend_cap : 0; a.crowd_item_cap = area ? d.item_cap : 0; a.crowd_pool_cap = area ? d.pool_cap : 0;
  a.crowd_items = area ? area + MIC_CROWD_HDR + 8 * (size_t)d.pend_cap : nullptr;
  a.crowd_pool = area ? a.crowd_items + 8 * (size_t)d.item_cap : nullptr;
}

// stage times of the table build in progress (mic_engine.hip; read back with mic_db_last_build_report)
void mic_build_report_add(const char* what, double seconds);
// device memory the engine's caller is allocating concurrently with the build (mic_db_reserve_hbm): not available to it
extern thread_local uint64_t mic_build_reserved_hbm;

// launchers (mic_kernels.hip)
hipError_t mic_launch_query(const MicQueryArgs& a, int slot_class, int n_cu, hipStream_t s);
// name of the instantiation mic_launch_query picks for this table, as a kernel trace shows it; returns the length
int mic_query_kernel_name(const MicTable& t, int slot_class, char* buf, size_t cap);
hipError_t mic_kernels_warm(hipStream_t s);      // loads the query kernels' device code (a no-op kernel of their file)
hipError_t mic_launch_merge_rows(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t row_words, size_t n,
                                 uint32_t* flags_results, hipStream_t s);
hipError_t mic_launch_result_from_rows(const uint32_t* rows, uint32_t row_words, uint32_t* results, size_t n,
                                       hipStream_t s);
hipError_t mic_launch_dense_count(const MicTable& t, int slot_class, const uint32_t* reads_ptr, const uint16_t* cont,
                                  const uint32_t* ids, size_t n_ids, uint32_t n_targets, uint32_t* counts,
                                  hipStream_t s);
hipError_t mic_launch_dense_finish(const uint32_t* counts, const uint32_t* ids, size_t n_ids, uint32_t n_targets,
                                   uint32_t* results, uint32_t* rows, uint32_t row_words, hipStream_t s);

hipError_t mic_launch_probe_stats(const MicTable& t, int slot_class, const uint32_t* reads_ptr, const uint16_t* cont,
                                  size_t n_reads, unsigned long long* d_out, hipStream_t s);

// table build (mic_build.hip)
struct MicBuildOut {
  uint4* slots;
  uint64_t n_main, n_overflow, n_elems, n_elems_file;
  uint32_t max_bucket;
  uint32_t max_chain;   // layout 1: entries in the fullest slot chain
  uint32_t walk_ppm;    // super-k-mer table: mean continuation slots in front of a stored k-mer, x 1e6
  uint64_t n_entries;   // super-k-mer table: entries (super-k-mers) stored; other layouts: 0 (= one entry per k-mer)
  uint4* side; uint64_t side_cells, side_kmers;   // super-k-mer table: the k-mers of crowded minimizers (mic_build.hip: s_crowd_move_kernel), or null
  uint64_t part_lo, part_hi; // super-k-mer table built as a slot-range part: the main slots [part_lo, part_hi) of n_main are resident (else 0, n_main)
  uint64_t alloc_slots; // slots allocated when that is more than n_main + n_overflow + 1 (one-pass super-k-mer build: the unused part of its continuation pool), else 0
};
// d_sizes/d_keys/d_labels point at the first bucket / first element of the shard.
// rank_base = number of non-empty buckets before the shard (sampling is defined on the whole table).
int mic_build_table(const uint8_t* d_sizes, uint64_t n_buckets, const void* d_keys, int key_bytes,
                    const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int slot_class, hipStream_t s,
                    MicBuildOut* out, char* err, size_t err_cap);
// Minimizer-keyed table from the same inputs.  bucket0 = index of the shard's first bucket in the whole table.
int mic_build_mtable(const uint8_t* d_sizes, uint64_t n_buckets, uint64_t bucket0, uint64_t htsize, const void* d_keys,
                     int key_bytes, const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int k, int m,
                     hipStream_t s, MicBuildOut* out, char* err, size_t err_cap);
// Super-k-mer table (layout 3, mic_device.h) from the same inputs.
int mic_build_stable(const uint8_t* d_sizes, uint64_t n_buckets, uint64_t bucket0, uint64_t htsize, const void* d_keys,
                     int key_bytes, const uint16_t* d_labels, uint32_t sampling, uint64_t rank_base, int k, int m,
                     hipStream_t s, MicBuildOut* out, char* err, size_t err_cap, int allow_fallback, int both_strands,
                     uint32_t part, uint32_t n_parts);
// (part of n_parts > 1: only the main slots [n_slots * part / n_parts, n_slots * (part + 1) / n_parts) of the table sized for the
// whole input are built; out->slots is the allocation, slot s of the table lives at out->slots + (s - out->part_lo) * 8)
// (allow_fallback: return -5 instead of building a table whose minimizers are crowded: see s_walk_kernel)
// sums over d_sizes[0..n): total elements and non-empty buckets
int mic_reduce_sizes(const uint8_t* d_sizes, uint64_t n, uint64_t* total, uint64_t* nonzero, hipStream_t s);

// Streams and events come from a process-wide pool and go back to it; none is ever destroyed (mic_engine.hip: why).
hipError_t mic_stream_get(hipStream_t* s);                 // a non-blocking stream of the current device
void mic_stream_put(hipStream_t s);                        // drained first; nullptr is fine
hipError_t mic_event_get(hipEvent_t* ev, bool timing);     // an event of the current device
void mic_event_put(hipEvent_t ev);

#endif
