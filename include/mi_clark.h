/*
 * mi_clark.h — C ABI of libmi_clark.so: the MI355X (gfx950) k-mer query engine that replaces
 * CuCLARK's CuClarkDB (device DB + query) behind the same caller contract.
 *
 * Plain C types only (pointers, sizes, fixed-width ints): bindable from C, C++, ctypes, cgo, JNI.
 * Every entry point cites the reference interface it replaces, as file:line under
 * Funatiq/cuclark src/.  All functions return MIC_OK (0) or a negative MIC_E_* code and never
 * exit() or print; mic_last_error() returns the message of the last failure on this thread
 * (reference: CUERR/CUMEMERR print + exit(1), CuClarkDB.cu:45-63).
 *
 * Data contract (unchanged from the reference unless stated):
 *   reads in   : readsPointer u32[nReads+1] + readsInContainers u16[]      CuCLARK_hh.hh:1616-1716
 *                (per read: parts; per part: 1 length slot + ceil(len/8) containers, 8 nt per u16,
 *                 first nt in the top bits, A=3 C=2 G=1 T/U=0).  A length slot of 0 ends a read
 *                 early (lets callers over-allocate per read); the reference never emits one.
 *   results out: MIC_RESULT_WORDS u32 per read:
 *                {sum, idxBest, best, idxSecond, second, nTargetsHit, flags, 0}
 *                words 0..4 are resultKernel's {sumN,indexBest,best,index_sBest,s_best}
 *                (CuClarkDB.cu:1421-1471) widened from u16 to u32; indices are target+1, 0 = "NA".
 *   sparse rows: optional, row_words u32 per read: word0 = n, words 1..n = (count<<16 | target) in
 *                ascending target order (the reference's [n,(t,c)*] u16 rows, CuClarkDB.cu:1178-1243,
 *                one pair per word).  A row that does not fit (n > row_words-1 or a count > 65535) has
 *                word0 = MIC_ROW_INVALID and its read carries MIC_FLAG_ROW_OVERFLOW; words 0..4 of its
 *                result are still exact (dense path), per-target counts come from mic_count_dense_device.
 */
#ifndef MI_CLARK_H
#define MI_CLARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIC_OK 0
#define MIC_E_INVALID (-1)  /* bad argument                                   */
#define MIC_E_IO (-2)       /* file missing / short (reference: read() returns false, CuClarkDB.cu:490-495) */
#define MIC_E_NOMEM (-3)    /* host or device allocation failed              */
#define MIC_E_HIP (-4)      /* HIP runtime error                              */
#define MIC_E_STATE (-5)    /* call out of order (e.g. query before DB load)  */
#define MIC_E_NODEVICE (-6) /* no usable gfx950 device / kernels not loadable */
#define MIC_E_UNSUPPORTED (-7) /* mic_gz_*: an input this path does not take; the caller uses its other path   */

#define MIC_RESULT_WORDS 8
#define MIC_FLAG_ROW_OVERFLOW 1u /* sparse row did not fit; dense path was used */
#define MIC_FLAG_DENSE_PATH 2u   /* read was (re)processed by the dense fallback */

#define MIC_ROW_INVALID 0xFFFFFFFFu
#define MIC_MAX_PART 65528u /* parts longer than this are packed as overlapping sub-parts */

typedef struct mic_engine mic_engine;

typedef struct mic_config {
  int32_t device;       /* HIP device ordinal; -1 = current device                     */
  int32_t k;            /* k-mer length, 2..32                       main.cc:110-114  */
  uint32_t num_targets; /* number of labels T (<= 65535)             dataType.hh:48   */
  uint32_t num_batches; /* batches of the batch API (>=1)            main.cc:220-226  */
  uint32_t row_words;   /* u32 words per sparse row (0 = default 16 => 15 pairs = MAXHITS, parameters.hh:44) */
  uint32_t layout;      /* resident table layout: MIC_LAYOUT_AUTO / _DIRECT / _MINIMIZER / _SUPER / _SUPER2 */
} mic_config;

#define MIC_LAYOUT_AUTO 0      /* super-k-mer table for k >= 24 (minimizer, then direct if it does not fit), else direct; env MIC_LAYOUT=direct|minimizer|super|super2 overrides */
#define MIC_LAYOUT_DIRECT 1    /* one 64-byte slot per on-disk bucket (one HBM request per k-mer) */
#define MIC_LAYOUT_MINIMIZER 2 /* 128-byte slots keyed by the k-mer's minimizer (one HBM request per ~7 k-mers) */
#define MIC_LAYOUT_SUPER 3     /* 128-byte slots of super-k-mers: the k-mers sharing a minimizer occurrence are one entry; one slot per lookup */
#define MIC_LAYOUT_SUPER2 4    /* the same with BOTH strands of every k-mer stored: the query kernel needs no reverse complement (fastest; twice the table) */

typedef struct mic_db_info {
  uint64_t htsize;         /* buckets in the whole table (= size of .sz)               */
  uint64_t shard_start;    /* first bucket resident on this engine                     */
  uint64_t shard_end;      /* one past the last resident bucket                        */
  uint64_t n_elems;        /* elements resident (after sampling / reachability filter) */
  uint64_t n_elems_file;   /* elements the shard holds on disk                         */
  uint64_t n_slots;        /* 64-byte slots in HBM (main + overflow chain)             */
  uint64_t n_overflow;     /* overflow slots                                            */
  uint64_t hbm_bytes;      /* bytes of HBM held by the table                            */
  int32_t key_bytes;       /* width of the keys on disk: 2, 4 or 8                      */
  int32_t slot_class;      /* 32: 8 entries/slot (u32 quotients); 64: 4 entries/slot; 128: minimizer / super-k-mer table */
  uint32_t max_bucket;     /* largest kept bucket                                       */
  uint32_t sampling;
  int32_t layout;          /* MIC_LAYOUT_DIRECT, _MINIMIZER, _SUPER or _SUPER2          */
  int32_t minimizer_len;   /* m (layouts MINIMIZER, SUPER), else 0                      */
  uint32_t max_chain;      /* entries in the fullest slot chain (MINIMIZER, SUPER)      */
  uint32_t reserved;       /* SUPER: mean number of continuation slots in front of a stored k-mer, x 1e6 (crowded minimizers) */
  uint64_t n_entries;      /* entries stored: super-k-mers (SUPER: several k-mers each), else = n_elems */
  uint32_t part, n_parts;  /* mic_db_set_part (0, 0 when the engine holds the whole database)     */
  uint64_t part_slot_lo;   /* SUPER / SUPER2 with a part: the resident main slots [lo, hi) ...     */
  uint64_t part_slot_hi;
  uint64_t n_slots_whole;  /* ... of this many the whole table has (identical on every part)       */
  uint64_t side_kmers;     /* SUPER / SUPER2: k-mers of crowded minimizers (one minimizer value in more than 12 contexts: microsatellites),
                              kept in a side table keyed by the whole k-mer instead of in slot chains */
  uint64_t side_bytes;     /* HBM of that side table (part of hbm_bytes)                            */
} mic_db_info;

/* ---- engine lifetime: CuClarkDB ctor/dtor (CuClarkDB.cu:85-253) -----------------------------
 * mic_destroy waits for the device and frees the engine's memory; its HIP streams and events go back to a pool of the process
 * for the next engine on that device and are never destroyed (the runtime's completion handler races a stream's destruction:
 * DESIGN.md 7). */
int mic_create(const mic_config* cfg, mic_engine** out);
int mic_destroy(mic_engine* e);
/* test hook: {streams ever created, streams in the pool now, events ever created, events in the pool now} of this process */
int mic_debug_stream_pool(uint32_t out[4]);
const char* mic_last_error(void);
/* Number of usable devices (reference: cudaGetDeviceCount loop, CuClarkDB.cu:104-181). */
int mic_device_count(int* count);

/* ---- database: CuClarkDB::read (CuClarkDB.cu:461-808) + swapDbParts (:813-858) --------------
 * Loads <prefix>.sz/.ky/.lb and builds the resident table.  key_bytes 0 => rule of main.cc:274-316.
 * sampling <=1 keeps all buckets, else every sampling-th non-empty bucket (CuClarkDB.cu:497-524).
 * [shard_start, shard_end) selects the bucket range this engine answers for (the reference's
 * m_partPointer ranges, CuClarkDB.cu:566-574, 1272-1274); shard_end 0 => htsize. */
int mic_db_load_files(mic_engine* e, const char* prefix, int key_bytes, uint32_t sampling, uint64_t shard_start,
                      uint64_t shard_end);
/* Same from host memory images of the three files (whole table; the shard is cut out here). */
int mic_db_load_host(mic_engine* e, const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                     const uint16_t* labels, uint32_t sampling, uint64_t shard_start, uint64_t shard_end);
/* Same from device memory images (d_keys/d_labels hold the WHOLE table's elements). */
int mic_db_load_device(mic_engine* e, const uint8_t* d_sizes, uint64_t htsize, const void* d_keys, int key_bytes,
                       const uint16_t* d_labels, uint32_t sampling, uint64_t shard_start, uint64_t shard_end);
/* Table-sharded runs (the reference's multi-device mode: m_partPointer ranges, CuClarkDB.cu:566-574, every device
 * queried with all reads :886-890, partial rows summed :934-1001).  Call BEFORE mic_db_load_* (whole table: shard_start
 * = shard_end = 0): the engine then answers for part `part` of `n_parts` of the database, and the per-read rows of the
 * n_parts engines sum to the whole table's (mic_merge_rows_device / mic_batch_merge_shards).  HOW the database is cut
 * follows the resident layout.  The super-k-mer layouts cut their resident table by SLOT range: a slot is chosen by
 * the minimizer, so all k-mers of a run of a read (one super-k-mer) belong to one part - a part loads and searches
 * 1/n_parts of the slots and the per-run kernel keeps its form (one compare per run).  The bucket cut of the on-disk
 * hash (c mod HTSIZE) would scatter the k-mers of one super-k-mer over all parts.  The other layouts keep the
 * reference's cut: part p answers for the buckets [HTSIZE p / n, HTSIZE (p + 1) / n).  Each part is built from the
 * whole .sz/.ky/.lb images (all parts size the table identically); n_parts <= 1 clears the setting.
 * shard_start / shard_end of mic_db_load_* remain the explicit bucket-range form (not combinable with a part). */
int mic_db_set_part(mic_engine* e, uint32_t part, uint32_t n_parts);
/* Device memory the caller is about to allocate on this engine's device WHILE the next mic_db_load_* runs (e.g. the ingest
 * slots, set up on a side thread): the table builders size their staging areas from the free HBM minus this, so the layout
 * and the number of build passes do not depend on which allocation comes first.  0 clears it. */
int mic_db_reserve_hbm(mic_engine* e, uint64_t bytes);
/* ---- several devices in one process (the command line's -d N; the reference: CuClarkDB ctor + read, CuClarkDB.cu:104-208, 461-808) ----
 * mic_peer_matrix       the reference checks and enables peer access for every pair of its devices before it merges their
 *                       partial rows with cudaMemcpyPeer (CuClarkDB.cu:184-208, 954-974).  Here: matrix[i * n + j] = 1 when device
 *                       i reaches device j's memory directly (hipDeviceCanAccessPeer; enabled once per ordered pair, "already
 *                       enabled" is fine), 0 when copies between the two are staged through host memory by the runtime; the
 *                       diagonal is 1.  n_devices <= mic_device_count.  The table-sharded exchange calls it for its engines itself.
 * mic_db_load_files_multi   the database of mic_db_load_files into n engines from ONE read of <prefix>.sz/.ky/.lb: every chunk
 *                       of the files is read once into pinned memory and uploaded to every device that hosts an engine; the
 *                       tables - each engine's part when mic_db_set_part was called, the whole table otherwise - are then built
 *                       by one thread per device at the same time.  Engines that share a device share its copy of the images. */
int mic_peer_matrix(int* matrix, int n_devices);
/* Free and total memory of a device in bytes (hipMemGetInfo): how many parts a table needs is decided from it. */
int mic_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int mic_db_load_files_multi(mic_engine* const* engines, size_t n_engines, const char* prefix, int key_bytes, uint32_t sampling);
/* Name of the query-kernel instantiation mic_query_device / the batch API / the ingest path launch for this engine's table (what
 * rocprofv3's kernel trace shows), e.g. "query_kernel_r<31, 20, false, true, false>"; returns the length, < 0 without a table. */
int mic_db_kernel_name(const mic_engine* e, char* buf, size_t cap);
int mic_db_get_info(const mic_engine* e, mic_db_info* info);
int mic_db_unload(mic_engine* e);
/* Stage times of the last table build of this process, one "<stage>: <seconds>" per line (what MIC_LOAD_TIMING=1 prints on
 * stderr); valid until the calling thread asks again.  The reference prints its load time, CuCLARK_hh.hh:610-625. */
const char* mic_db_last_build_report(void);

/* ---- batch API: the calls CuCLARK_hh.hh makes on CuClarkDB ----------------------------------
 * mic_batches_alloc   = CuClarkDB::malloc        (CuClarkDB.cu:317-419;  caller CuCLARK_hh.hh:1600-1606)
 *   The engine allocates pinned host buffers and lends them out; the caller fills
 *   reads_pointer[b] / containers[b] for batch b and reads results after mic_batch_wait.
 *   index_batches[num_batches+1] = index of the first read of each batch; results/rows are indexed
 *   by global read index.  rows is NULL unless extended != 0.
 * mic_batch_ready     = CuClarkDB::readyBatch    (CuClarkDB.cu:864-873;  caller :1735)
 * mic_batch_query     = CuClarkDB::queryBatch    (CuClarkDB.cu:878-1033; caller :1743,:1772)
 *   H2D, fused query kernel, result, D2H, event — asynchronous.  `followup` (swap cycles) is accepted
 *   and ignored: the whole table is resident, there are no cycles.
 * mic_batch_wait      = CuClarkDB::waitForBatch  (CuClarkDB.cu:440-445;  caller :1997,:2006)
 * mic_batch_check     = CuClarkDB::checkBatch    (CuClarkDB.cu:450-456)  *done = 1 when finished
 * mic_sync            = CuClarkDB::sync          (CuClarkDB.cu:424-435)
 * mic_batches_free    = CuClarkDB::freeBatchMemory (CuClarkDB.cu:279-312)
 * Threading: distinct batches may be filled concurrently; mic_batch_query is internally serialised;
 * mic_batch_wait may be called from another thread (reference: CuCLARK_hh.hh:1738-1760). */
int mic_batches_alloc(mic_engine* e, size_t num_reads_total, size_t max_reads, size_t max_containers,
                      const uint32_t* index_batches, int extended, uint32_t** results, uint32_t** rows,
                      uint32_t** reads_pointer /*[num_batches]*/, uint16_t** containers /*[num_batches]*/);
int mic_batch_ready(mic_engine* e, size_t batch, size_t n_reads, size_t n_containers);
int mic_batch_query(mic_engine* e, size_t batch, int extended, int followup);
int mic_batch_wait(mic_engine* e, size_t batch);
/* Dense per-target counts (u32[num_targets]) of read `read_in_batch` of a finished batch whose sparse row did
 * not fit (row[0] == MIC_ROW_INVALID); the reference has no equivalent — it truncates at MAXHITS and corrupts
 * the row (CuClarkDB.cu:1200-1211).  Synchronous; valid until the batch is queried again or freed. */
int mic_batch_dense_counts(mic_engine* e, size_t batch, size_t read_in_batch, uint32_t* counts);
int mic_batch_check(mic_engine* e, size_t batch, int* done);
/* Table-sharded batches: the reference's multi-device mode (CuClarkDB.cu:934-1001 - queryBatch on every device, the
 * partial result rows copied to device 0 with cudaMemcpyPeer, summed by mergeKernel, finished by resultKernel).
 * Every engine holds one bucket range of the same database (mic_db_load_*: shard_start, shard_end), the same batch
 * geometry (mic_batches_alloc with extended = 1) and was given the same packed reads for `batch`
 * (mic_batch_ready + mic_batch_query(extended = 1) on each).  The call waits for all of them, sums the sparse rows
 * into engines[0] and leaves best / second-best and the merged rows in engines[0]'s host arrays.  A read whose merged
 * row does not fit (MIC_FLAG_ROW_OVERFLOW in its results word 6, row[0] == MIC_ROW_INVALID) is completed by the
 * caller: the sum over the engines of mic_batch_dense_counts.  Synchronous.  Engines may share a device. */
int mic_batch_merge_shards(mic_engine* const* engines, size_t n_engines, size_t batch);
/* queryBatch on every engine of a table-sharded group from ONE upload (the reference copies the batch's host arrays to every device,
 * CuClarkDB.cu:886-890): the reads are engines[0]'s (filled into ITS lent buffers, mic_batch_ready on engines[0]); engines[0]
 * uploads them, the other engines take the packed reads from engines[0]'s device buffers by peer copy on their batch's stream and
 * run their kernel; all asynchronous, as mic_batch_query.  Follow with mic_batch_merge_shards.  The other engines' host buffers
 * for this batch are not read. */
int mic_batch_query_group(mic_engine* const* engines, size_t n_engines, size_t batch, int extended);
/* (How the rows are summed: READ-RANGE OWNED, all engines at once - engine j fetches the rows of the j-th 1/n of the batch's
 * reads from the other n - 1 engines (n - 1 peer copies of 1/n of the rows each, hipMemcpyPeerAsync over xGMI with peer access
 * enabled), sums them, finishes best / second-best for its range and writes results and rows of the range straight into
 * engines[0]'s host arrays.  The reference funnels whole row arrays into device 0 one after another, CuClarkDB.cu:954-974.
 * Peer copies, not RCCL: one process drives all devices here, the copy engines move the rows while the compute units run the
 * next batch's kernels, and there is nothing to rendez-vous; the one-process-per-GPU form of the same exchange is bench.py's
 * all_to_all over RCCL, cuclark_amd/multi.py.) */
int mic_sync(mic_engine* e);
/* Host threads that fill the engine's pinned buffers should run on the socket the device hangs off: on != 0 binds the
 * calling thread to the CPUs of the device's NUMA node, on == 0 restores its previous mask (no-op when the node is
 * unknown or MIC_NO_NUMA is set).  The engine's own pinned allocations are made there already. */
int mic_thread_bind_near_device(mic_engine* e, int on);
int mic_batches_free(mic_engine* e);

/* ---- device-resident entry points (kernels only; inputs/outputs already in HBM) --------------
 * mic_query_device: queryKernel + resultKernel fused (CuClarkDB.cu:1045-1243,1421-1471).
 *   d_results: n_reads*MIC_RESULT_WORDS u32.  d_rows: n_reads*row_words u32 or NULL.
 *   Reads whose register row (64 distinct targets) or sparse row overflowed are flagged and listed in an
 *   engine-owned device list; mic_resolve_flagged_device completes them exactly with the dense path.
 *   d_containers must be 4-byte aligned and readable for 32 elements past the last container (the kernel prefetches
 *   windows as aligned dwords); MIC_E_INVALID otherwise.
 *   stream: a hipStream_t (NULL = the engine's stream).  Asynchronous. */
int mic_query_device(mic_engine* e, const uint32_t* d_reads_pointer, const uint16_t* d_containers, size_t n_reads,
                     uint32_t* d_results, uint32_t* d_rows, void* stream);
/* Completes the reads flagged by the LAST mic_query_device call (same buffers, same stream): waits for the
 * stream, runs the dense kernels for the listed reads and patches d_results / d_rows.  *n_resolved
 * (optional) receives how many reads took the dense path.  Synchronous. */
int mic_resolve_flagged_device(mic_engine* e, const uint32_t* d_reads_pointer, const uint16_t* d_containers,
                               uint32_t* d_results, uint32_t* d_rows, void* stream, size_t* n_resolved);
/* mergeKernel (CuClarkDB.cu:1321-1415): out = a (+) b per read (sum by target); rows as above. */
int mic_merge_rows_device(mic_engine* e, const uint32_t* d_rows_a, const uint32_t* d_rows_b, uint32_t* d_rows_out,
                          size_t n_reads, void* stream);
/* resultKernel (CuClarkDB.cu:1421-1471) on sparse rows. */
int mic_result_from_rows_device(mic_engine* e, const uint32_t* d_rows, uint32_t* d_results, size_t n_reads,
                                void* stream);
/* Dense per-read per-target counts (u32 [n_reads*num_targets]) for the listed reads: the exact
 * fallback for reads whose rows overflow, and the --extended source of truth. */
int mic_count_dense_device(mic_engine* e, const uint32_t* d_reads_pointer, const uint16_t* d_containers,
                           const uint32_t* d_read_ids, size_t n_ids, uint32_t* d_counts, void* stream);
/* resultKernel on DENSE counts (u32 [n_ids * num_targets], e.g. mic_count_dense_device's, or their sum over the shards of
 * a table-sharded run): writes result row d_ids[i] (or i when d_ids is NULL) of d_results, and the sparse row when d_rows is
 * given and it fits.  This is how a read with more targets than a sparse row holds is completed exactly across shards. */
int mic_result_from_dense_device(mic_engine* e, const uint32_t* d_counts, const uint32_t* d_ids, size_t n_ids,
                                 uint32_t* d_results, uint32_t* d_rows, void* stream);
/* Bookkeeping for the roofline figure (not timed): out = {k-mers in the reads, k-mers whose bucket lies in this
 * engine's shard, hits, sum of the lengths of the probed buckets}.  Synchronous. */
int mic_probe_stats_device(mic_engine* e, const uint32_t* d_reads_pointer, const uint16_t* d_containers, size_t n_reads,
                           uint64_t out[4]);
/* Duration in ms of the last mic_query_device launch on this engine, measured with HIP events on the
 * stream it ran on (blocks until it finished). */
int mic_last_query_ms(mic_engine* e, float* ms);
/* Bookkeeping of the last mic_query_device launch on a super-k-mer table with a side table of crowded minimizers
 * (microsatellites; see mic_db_info.side_kmers): out = {reads that met a crowded minimizer and were finished by the
 * follow-up kernel, crowded runs handed to it, words of spilled rows, reads sent to the dense path for lack of room in
 * the work area}.  All zero for any other table.  Synchronous. */
int mic_last_crowd_stats(mic_engine* e, uint32_t out[4]);
/* test hook: the first `words` 32-bit words of the work area of the last mic_query_device launch (mic_internal.h: header, pending reads,
 * items, pool) and its capacities {pending reads, items, pool words}; MIC_E_STATE when the engine has none.  Synchronous. */
int mic_debug_fetch_crowd(mic_engine* e, uint32_t* out, size_t words, uint32_t caps[3]);

/* ---- device-side ingest: raw FASTA / FASTQ bytes in, result-CSV text out ----------------------------------------
 * The reference does the read indexing (CuCLARK_hh.hh:1339-1534), the 2-bit packing with its N-splitting
 * (CuCLARK_hh.hh:1616-1716) and the CSV lines (printExtendedResultsSynced, CuCLARK_hh.hh:1951-2139) on host threads
 * around queryBatch; at MI355X kernel rates those loops are the bottleneck by two orders of magnitude.  Here a batch
 * of WHOLE records is handed over as the bytes of the file and comes back as the bytes of the CSV (no header line,
 * non-extended format): line index, record index, packer, query kernel and CSV formatter all run on the device,
 * the host only moves bytes.  Output is byte-identical to mic_index_reads + mic_pack_reads + mic_batch_query +
 * mic_csv_line on the same bytes.
 *
 * mic_ingest_alloc    engine-owned pinned input buffers (raw[i], max_bytes each, 4 KiB .. 128 MiB) lent to the caller, as
 *                     CuClarkDB::malloc lends its batch buffers (CuClarkDB.cu:355-360); target_names as in mic_csv_line.
 * mic_ingest_classify blocking; slot-private stream: call it from one host thread per slot to keep the device busy.
 *                     The slot's first byte must be '>' (FASTA, also the merged paired-end text of file.cc:205-268 with
 *                     MIC_INGEST_PAIRED in flags) or '@' (FASTQ; with MIC_INGEST_FASTQ_2LINE two lines per record).
 *                     out->status == MIC_INGEST_OK: out->csv / csv_bytes / n_reads are valid until the slot's next call.
 *                     out->status & MIC_INGEST_FALLBACK: the batch holds something the device path does not
 *                     reproduce (the other bits say what); nothing was produced and the caller runs the host path
 *                     (mic_index_reads ... mic_csv_line) on these bytes.
 * mic_ingest_fetch_packed  test hook: the packed reads of the slot's last batch as the query kernel saw them. */
#define MIC_INGEST_PAIRED 1        /* flags: objects are merged pairs: Length column minus the separator (CuCLARK_hh.hh:2119)        */
#define MIC_INGEST_FASTQ_2LINE 2   /* flags: FASTQ records come as header + sequence line only (the caller dropped the '+' and
                                      quality lines, which nothing reads: halves the bytes that cross the host link)           */
#define MIC_INGEST_RESIDENT 4       /* flags: the slot's DEVICE buffer already holds the n_bytes of FASTA text - merged pairs
                                      (mic_pairs_merge_to_slot, with MIC_INGEST_PAIRED) or records of a FASTA text
                                      (mic_text_to_slot): nothing is uploaded                                                    */
#define MIC_INGEST_RESIDENT_FASTQ 12 /* flags: as MIC_INGEST_RESIDENT, and the text is four-line FASTQ (mic_text_to_slot), not merged pairs */
#define MIC_INGEST_OK 0u
#define MIC_INGEST_FALLBACK 1u     /* run the host path on this batch                                        */
#define MIC_INGEST_ODD_RECORD 2u   /* empty read name, FASTA record without a sequence line, unknown format   */
#define MIC_INGEST_TRUNCATED 4u    /* FASTQ line count not a multiple of four                                 */
#define MIC_INGEST_LONG_READ 8u    /* a sequence of more than MIC_MAX_PART bytes                              */
#define MIC_INGEST_TOO_MANY 16u    /* more lines / reads / containers / CSV bytes than the slot was sized for */
#define MIC_INGEST_DENSE 32u       /* a read needs the dense fallback (more than 64 targets hit)              */

typedef struct mic_ingest_result {
  uint64_t n_reads;
  uint64_t csv_bytes;
  const char* csv;           /* pinned host memory of the slot                                    */
  const uint32_t* results;   /* MIC_RESULT_WORDS per read when want_results was set, else NULL    */
  uint32_t status;
  uint32_t n_lines;
} mic_ingest_result;

int mic_ingest_alloc(mic_engine* e, size_t n_slots, size_t max_bytes, const char* const* target_names, uint32_t n_targets,
                     int want_results, uint8_t** raw /*[n_slots]*/);
int mic_ingest_classify(mic_engine* e, size_t slot, size_t n_bytes, int flags, mic_ingest_result* out);
/* The same for a table-sharded run (cuCLARK --db-sharded): group[p] holds part p of n_group parts of the database
 * (mic_db_set_part(p, n_group)); the slot belongs to group[owner] (mic_ingest_alloc on that engine), which indexes and packs the
 * batch and formats its CSV.  In between, every engine of the group probes ALL reads of the batch against its part (the packed
 * reads travel to the other engines by peer copy: 44 B per 150-bp read), and the per-read sparse rows are summed READ-RANGE
 * OWNED: engine j fetches the rows of the j-th 1/n_group of the reads from the other engines, sums, finishes best / second-best
 * and sends the 32-byte results of its range to the owner - all engines at once, nothing funnels through one device (the
 * reference: queryBatch on every device, cudaMemcpyPeer of whole row arrays into device 0, CuClarkDB.cu:886-1001).  A batch with
 * a read whose summed row does not fit (more than 15 targets) comes back as MIC_INGEST_FALLBACK | MIC_INGEST_DENSE like any other
 * batch the device path hands back.  n_group == 1 is mic_ingest_classify.  Blocking; one host thread per slot. */
int mic_ingest_classify_group(mic_engine* const* group, size_t n_group, size_t owner, size_t slot, size_t n_bytes, int flags,
                              mic_ingest_result* out);
int mic_ingest_fetch_packed(mic_engine* e, size_t slot, uint32_t* reads_pointer, size_t rp_cap, uint16_t* containers,
                            size_t cont_cap, uint64_t* n_reads, uint64_t* n_containers);
/* mic_ingest_fetch_group_rows  test hook: the partial sparse rows (row_words u32 per read) engine `part` of the slot's group
 * computed for the slot's last table-sharded batch, as its kernel wrote them. */
int mic_ingest_fetch_group_rows(mic_engine* owner, size_t slot, size_t part, uint32_t* rows, size_t cap_words, uint64_t* n_reads,
                                uint32_t* row_words);
/* mic_ingest_group_stats  what the table-sharded batches of this engine's slots cost, when the process runs with
 * MIC_GROUP_TIMING=1 (HIP events on every engine's stream; a measuring mode: the exchange then starts when ALL engines' rows are
 * written, so that its bracket holds copies and merges only).  out[0 .. MIC_GROUP_STATS_FIELDS): batches, reads, bytes of the
 * packed-read fan-out (owner -> the other engines), its ms summed over the helpers and over the batches, the same with the
 * slowest helper of each batch only, query-kernel ms summed over engines and batches, slowest engine of each batch only, bytes of
 * the row exchange (rows of a read range from the other engines + the results to the owner), its ms summed, slowest engine only.
 * Returns the number of fields (> 0: the one entry point that does not return MIC_OK on success) or a negative code; zeros
 * without MIC_GROUP_TIMING.  (The reference times nothing per device; its exchange is
 * CuClarkDB.cu:954-974.) */
#define MIC_GROUP_STATS_FIELDS 10
int mic_ingest_group_stats(mic_engine* owner, double* out, size_t cap);
int mic_ingest_free(mic_engine* e);
/* ---- compressed input: one gzip member inflated on the device ------------------------------------------------
 * Replaces the `gunzip` the reference's scripts run in front of the classifier (classify_metagenome.sh:116-142) for the
 * common cases: ONE member without a preset dictionary (what `gzip` writes), or a whole block-gzip file (BGZF: members with a
 * 'BC' size subfield, what bgzip / samtools write).  `gz` = the whole .gz file in host memory.  On MIC_OK *d_text is a
 * device buffer of *n_text bytes of text (mic_gz_free_text releases it, mic_gz_copy_text copies a piece to the host);
 * length (ISIZE) and CRC-32 of the member - of every member of a block-gzip file - have been checked against the trailer, as
 * gunzip checks them; *crc32_expected is the one member's CRC-32 (0 for block gzip).
 * MIC_E_UNSUPPORTED: several ordinary members, data behind the last block, or blocks that could not be found speculatively - the
 * caller inflates on the CPU (csrc/pgz.hpp / zlib) as before; MIC_E_INVALID: the data is damaged (zlib would fail too). */
int mic_gz_inflate_device(mic_engine* e, const void* gz, size_t gz_bytes, void** d_text, size_t* n_text, uint32_t* crc32_expected);
int mic_gz_copy_text(mic_engine* e, const void* d_text, size_t offset, size_t n, void* host_dst);
int mic_gz_free_text(mic_engine* e, void* d_text);
/* mic_gz_reserve: the device buffers of one mic_gz_inflate_device call for a file of gz_bytes bytes whose trailer says isize
 * (the last four bytes of the file), set up ahead of it - a fresh gigabyte of device memory takes the driver as long as the decode
 * does; the command line reserves while its database loads, as it does with its ingest slots (CuClarkDB::malloc's place in the
 * reference: CuCLARK_hh.hh:1600-1606).  The next call for a file of exactly this size uses them; mic_gz_release (also part of
 * mic_destroy) frees what the engine's reservations still hold.  mic_gz_reserve_bytes: what a reservation takes, for
 * mic_db_reserve_hbm. */
int mic_gz_reserve(mic_engine* e, size_t gz_bytes, uint32_t isize);
uint64_t mic_gz_reserve_bytes(size_t gz_bytes, uint32_t isize);
int mic_gz_release(mic_engine* e);
/* One plain member in STRIPES (round 6; BASELINE config 5, "gzip FASTQ ingest overlapped"): the units of the member - one per deflate
 * block - go through decode, stitching, windows and resolve a stripe of consecutive units at a time, and when a stripe is done its
 * text is final: the caller indexes and classifies it (mic_text_index_front_device, mic_text_to_slot) while the next stripe decodes.
 * mic_gz_stream_open   uploads the file, finds its blocks, hands out the text buffer of *n_text bytes (the trailer's ISIZE), nothing
 *                      of it final yet; `stripes` = how many pieces to cut the units into (>= 1).  Block gzip: MIC_E_UNSUPPORTED
 *                      (mic_gz_inflate_device takes it whole).
 * mic_gz_stream_next   the next stripe, synchronously: on MIC_OK the first *n_final bytes of the text are final; *done = 1 with the last
 *                      stripe, after length and CRC-32 were checked against the trailer.  MIC_E_UNSUPPORTED / MIC_E_INVALID as for
 *                      mic_gz_inflate_device - but possibly after text was handed out: a caller that used it starts over on its CPU
 *                      inflater (exe/cuCLARK does, classifier.cpp).
 * mic_gz_stream_close  frees everything; keep_text != 0: but the text, the caller's from then on (mic_gz_free_text). */
typedef struct mic_gz_stream mic_gz_stream;
int mic_gz_stream_open(mic_engine* e, const void* gz, size_t gz_bytes, uint32_t stripes, mic_gz_stream** out, void** d_text, size_t* n_text);
int mic_gz_stream_next(mic_gz_stream* s, size_t* n_final, int* done);
int mic_gz_stream_close(mic_gz_stream* s, int keep_text);

/* ---- paired-end FASTQ texts that are on the device already (inflated there): the reference's merge, on the device ----
 * file.cc:205-268 (mergePairedFiles) writes ">id\nseq1Nseq2\n" per pair of records, id = the header's first field between
 * the separators ' ', '/', '\t', '@' (file.cc:224), both ids equal.  The command line's loaders do that on the host for
 * files (classifier.cpp: PairedFileFeeder); for two texts in device memory (mic_gz_inflate_device of both mates):
 * mic_pairs_index_device  line index of both texts, every pair of records checked, the offsets of the merged records.
 *                         *status != 0 (MIC_PAIRS_*): these files are not what the line arithmetic covers - line counts that
 *                         differ or are no multiple of four, a header line without '@', ids that differ or are empty, a text
 *                         of 4 GiB or more; nothing is returned and the caller runs its host reader, which treats such
 *                         input the way the reference does.  The texts must stay allocated while the handle lives
 *                         (16 readable bytes behind each: mic_gz_inflate_device's buffers have them).
 * mic_pairs_offsets       host array: bytes of merged text in front of record i * stride (last entry: in front of record
 *                         n_records = the whole text), from which the caller cuts batches that fit its slots.
 * mic_pairs_merge_to_slot the merged text of records [r0, r1) written into an ingest slot's device buffer on the slot's
 *                         stream (r0 and r1 multiples of the stride, or n_records); then mic_ingest_classify(slot, *n_bytes,
 *                         MIC_INGEST_PAIRED | MIC_INGEST_RESIDENT).
 * mic_pairs_text          the same text copied to the host (for a batch the device path hands back).                  */
#define MIC_PAIRS_LINES 1u     /* line counts differ, are zero or no multiple of four */
#define MIC_PAIRS_HEADER 2u    /* a record whose first line is empty or has no '@' in front */
#define MIC_PAIRS_ID 4u        /* ids of a pair differ, or are empty */
#define MIC_PAIRS_BIG 8u       /* a text of 4 GiB or more (32-bit line starts), or an empty one */
typedef struct mic_pairs mic_pairs;
int mic_pairs_index_device(mic_engine* e, const void* d_text1, size_t n1, const void* d_text2, size_t n2, mic_pairs** out,
                           uint64_t* n_records, uint32_t* status);
int mic_pairs_offsets(const mic_pairs* p, const uint64_t** samples, size_t* n_samples, uint32_t* stride);
int mic_pairs_merge_to_slot(mic_engine* e, mic_pairs* p, uint64_t r0, uint64_t r1, size_t slot, size_t* n_bytes);
int mic_pairs_text(mic_engine* e, mic_pairs* p, uint64_t r0, uint64_t r1, void* host_dst, size_t cap, size_t* n_bytes);
int mic_pairs_free(mic_engine* e, mic_pairs* p);

/* ---- one FASTQ text that is on the device already (the inflated file of -O reads.fq.gz) ------------------------------------
 * mic_text_index_device  where the records start: FASTQ (the text begins with '@' and has a multiple of four lines) or FASTA (it
 *                        begins with '>'; a record is a '>' line and the lines up to the next one); else *status
 *                        (MIC_PAIRS_HEADER: something else, MIC_PAIRS_LINES, MIC_PAIRS_BIG) and the caller's host reader.
 * mic_text_format        '@' or '>': what the index found.
 * mic_text_offsets       host array: byte offset of record i * stride (last entry: the size of the text).
 * mic_text_to_slot       records [r0, r1) copied device to device into an ingest slot's buffer on the slot's stream; then
 *                        mic_ingest_classify(slot, *n_bytes, MIC_INGEST_RESIDENT_FASTQ, or MIC_INGEST_RESIDENT for FASTA): records the device path does not
 *                        reproduce come back as MIC_INGEST_FALLBACK as always, and mic_text_copy gives their bytes to the host. */
typedef struct mic_text mic_text;
int mic_text_index_device(mic_engine* e, const void* d_text, size_t n, mic_text** out, uint64_t* n_records, uint32_t* status);
/* the FRONT of a FASTQ text that is still growing (mic_gz_stream_next): its whole records - lines that end in a newline, four to a
 * record - are indexed, *n_used = where the first record that is not whole begins (the next call's text starts there).  No whole
 * record yet: MIC_OK, *out = NULL, *n_used = 0.  *status = MIC_PAIRS_HEADER when the text does not begin with '@'.  The rest of the
 * text, once it is all there: mic_text_index_device from where the last front ended (d_text need not be aligned for either). */
int mic_text_index_front_device(mic_engine* e, const void* d_text, size_t n, mic_text** out, uint64_t* n_records, uint64_t* n_used, uint32_t* status);
int mic_text_format(const mic_text* p);
int mic_text_offsets(const mic_text* p, const uint64_t** samples, size_t* n_samples, uint32_t* stride);
int mic_text_to_slot(mic_engine* e, mic_text* p, uint64_t r0, uint64_t r1, size_t slot, size_t* n_bytes);
int mic_text_copy(mic_engine* e, mic_text* p, uint64_t r0, uint64_t r1, void* host_dst, size_t cap, size_t* n_bytes);
int mic_text_free(mic_engine* e, mic_text* p);

/* "%g" of (double)num / den for 0 < num <= den, by the integer-only formatter the device CSV kernel uses
 * (csrc/mic_fmt.h); writes at most 14 characters and a terminator, returns the length. */
int mic_format_ratio_g(uint32_t num, uint32_t den, char* out16);

/* ---- host-side pieces of the path (pure CPU, no device needed) -------------------------------- */
/* Key width rule, main.cc:274-316. */
int mic_key_bytes_rule(uint64_t htsize, int k);
/* Read indexer, CuCLARK_hh.hh:1339-1534 (one batch): fills caller arrays of capacity cap; returns the
 * number of reads (may exceed cap: call again with a larger cap), or MIC_E_INVALID if the first byte
 * is neither '>' nor '@'. */
long mic_index_reads(const uint8_t* map, size_t nb, size_t cap, uint64_t* name_s, uint64_t* name_e, uint64_t* seq_s,
                     uint64_t* seq_e, uint64_t* length);
/* Same result, indexed by n_threads OpenMP threads over byte ranges of the file (record boundaries inside a range are
 * found like the reference finds its batch starts, CuCLARK_hh.hh:1409-1471). */
long mic_index_reads_parallel(const uint8_t* map, size_t nb, int n_threads, size_t cap, uint64_t* name_s, uint64_t* name_e,
                              uint64_t* seq_s, uint64_t* seq_e, uint64_t* length);
/* Position of the first record ('>' at a line start; '@' line followed by a letters-only line and a '+' line) at or
 * after byte `from`, or nb: lets callers cut a large input into segments of whole records. */
size_t mic_find_record_start(const uint8_t* map, size_t nb, size_t from);
/* The same inside a window win[0, nb) of an input of known format (fasta != 0: FASTA, else FASTQ), from >= 1: position
 * of the first record start at or after `from`, or nb when the window holds none that can be verified. */
size_t mic_find_record_start_in(const uint8_t* win, size_t nb, int fasta, size_t from);
/* Upper bound of containers mic_pack_reads can emit for these reads. */
size_t mic_pack_bound(const uint64_t* seq_s, const uint64_t* seq_e, size_t n_reads, int k);
/* Read packer, CuCLARK_hh.hh:1616-1716.  Returns containers written or (size_t)-1 if cap is too small. */
size_t mic_pack_reads(const uint8_t* map, const uint64_t* seq_s, const uint64_t* seq_e, const uint64_t* length,
                      size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap);
/* CSV, CuCLARK_hh.hh:1951-2139.  Return bytes written or -1 if cap is too small. */
int mic_csv_header(char* buf, size_t cap, int extended, const char* const* target_names, uint32_t n_targets);
int mic_csv_line(char* buf, size_t cap, const uint8_t* name, size_t name_len, uint64_t length, int paired, int k,
                 const uint32_t* result /*MIC_RESULT_WORDS*/, const char* const* target_names, uint32_t n_targets,
                 int extended, const uint32_t* row /*sparse row or NULL*/, const uint32_t* dense /*or NULL*/);

/* ---- database construction on the GPU ------------------------------------------------------
 * Builds <out_prefix>.sz/.ky/.lb from target FASTA/FASTQ files exactly as the reference's first run does
 * (makeSpecificTargetSets + RemoveCommon + Write, CuCLARK_hh.hh:691-1329, HashTableStorage_hh.hh:241-292,483-523,
 * hashTable_hh.hh:590-663): a canonical k-mer is stored iff all its occurrences belong to one label and its
 * occurrence count (saturating at 254) exceeds min_count.  target_labels[i] = label index of target_files[i]
 * (first-appearance order of the labels in the targets file).  key_bytes 0 => rule of main.cc:274-316.
 * light_gap 0 => every k-mer (cuCLARK); g > 0 => the light database of cuCLARK-l (CuCLARK_hh.hh:694-895): each maximal
 * ACGT run is cut into consecutive non-overlapping blocks of k nucleotides, numbered through the whole file, and
 * block i is used iff i % g == 0.
 * parts 0 => as many passes over disjoint bucket ranges as the free HBM requires.  Synchronous. */
int mic_db_build(const char* const* target_files, const uint16_t* target_labels, size_t n_files, int k, uint64_t htsize,
                 int key_bytes, uint32_t min_count, uint32_t light_gap, const char* out_prefix, int device, int threads,
                 uint32_t parts, uint64_t* n_kmers_out);
const char* mic_db_build_error(void);

/* ---- synthetic workload generation in HBM (bench.py / tests) ------------------- */
typedef struct mic_synth_spec {
  uint64_t seed;
  uint64_t htsize;        /* buckets                                                        */
  uint64_t genome_nt;     /* total nucleotides of the procedural genomes (~ elements)       */
  uint32_t n_targets;     /* labels; genome g (of equal length) carries label g % n_targets */
  uint32_t n_genomes;
  int32_t k;
  int32_t key_bytes;      /* 4 or 8                                                         */
  uint32_t keep_ppm;      /* 0: every k-mer of the genomes is in the database.  Otherwise a FRAGMENTED database, as the removal
                             of k-mers common to several targets leaves one (HashTableStorage_hh.hh:241-292): the k-mer start
                             positions of a genome fall into segments of geometric length (mean run_len), and a segment's k-mers
                             are kept with probability keep_ppm / 1e6                                                         */
  uint32_t run_len;       /* mean segment length in k-mer positions (0: 8)                                                    */
  uint32_t repeat_ppm;    /* 0: uniformly random genomes.  Otherwise this fraction (x 1e-6) of every genome is TANDEM REPEATS: tracts of
                             256 .. 1279 nucleotides, units of 4 .. 50; units of up to 6 nucleotides come from a pool shared by all
                             genomes (microsatellites).  The database holds what the reference's builder would keep of them: nothing that
                             lies wholly inside a shared unit's tract (common to many targets), a private unit's k-mers once          */
  uint32_t mosaic_ppm;    /* this fraction (x 1e-6) of the 2048-nucleotide segments of every genome carries MOSAIC labels: the label of
                             a k-mer there changes every 1, 2, 4 or 8 positions (close relatives after the removal of shared k-mers):
                             reads from there hit many targets with small equal counts - ties, rows of more than 15 / 64 targets       */
} mic_synth_spec;
/* Builds the on-disk-format arrays of a synthetic database in device memory the caller owns:
 *   d_sizes u8[htsize], d_keys key_bytes*[capacity], d_labels u16[capacity]; *n_elems out. */
int mic_synth_db_device(const mic_synth_spec* spec, uint8_t* d_sizes, void* d_keys, uint16_t* d_labels,
                        uint64_t capacity, uint64_t* n_elems, void* stream);
/* Builds n_reads packed reads of read_len nt in device memory (reads_pointer u32[n_reads+1],
 * containers u16[n_reads*mic_synth_read_pitch(read_len,k)]): a fraction `random_frac` are uniform random, the
 * rest are sampled from the genomes (either strand) with per-base substitution rate `sub_rate` and
 * N rate `n_rate`.  d_truth (optional, u32[n_reads*2]) receives {label+1 or 0, expected hits lower bound}. */
uint32_t mic_synth_read_pitch(uint32_t read_len, int k); /* containers reserved per read by the generator */
int mic_synth_reads_device(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len,
                           double random_frac, double sub_rate, double n_rate, uint32_t* d_reads_pointer,
                           uint16_t* d_containers, size_t containers_cap, uint32_t* d_truth, void* stream);
/* paired != 0: every object is read 1 + 'N' + read 2 of a pair (the merged form of file.cc:205-268), 2 * read_len + 1
 * characters, containers at mic_synth_read_pitch(2 * read_len + 1, k). */
int mic_synth_reads_device2(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len, int paired,
                            double random_frac, double sub_rate, double n_rate, uint32_t* d_reads_pointer,
                            uint16_t* d_containers, size_t containers_cap, uint32_t* d_truth, void* stream);
/* The same reads as text, one fixed-size record per read: FASTQ "@r<9 digits>\n" SEQ "\n+\n" QUAL "\n"
 * (mic_synth_text_record_bytes = 2 L + 16) or FASTA ">r<9 digits>\n" SEQ "\n" (L + 13), an N wherever the packed form ends a
 * part.  mate < 0: exactly the reads of mic_synth_reads_device (same seeds); mate 0 / 1: the two reads of a pair drawn
 * from both ends of a stretch of 2 L nucleotides (read 1 and the reverse strand's read 2), same genome, same label. */
size_t mic_synth_text_record_bytes(uint32_t read_len, int fasta);
int mic_synth_reads_text_device(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len,
                                double random_frac, double sub_rate, double n_rate, int fasta, int mate, uint8_t* d_text,
                                size_t text_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
