/*
 * clark_oracle.h — CPU restatement of CuCLARK's k-mer query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product (cuclark_amd/, libmi_clark.so) never links, loads or calls it.
 *
 * Parity status: PINNED.  The per-k-mer answers of orc_db_find() and the .sz/.ky/.lb layout
 * are checked against the reference's own CPU hash table (hTable::find / EHashtable::Write,
 * compiled from /root/reference/src by oracle/Makefile into oracle/_ref/) and against the
 * golden vectors committed in tests/golden/ that were produced by that binary
 * (tests/golden/make_golden.py).  The reference repository has no tests or golden vectors of
 * its own; the scoring / CSV rules are restated from the cited lines.
 *
 * Every function cites the reference file:line (relative to /root/reference/src) it follows.
 */
#ifndef CLARK_ORACLE_H
#define CLARK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codec --------------------------------------------------------------------------- */

/* 2-bit code of a nucleotide byte, -1 if the byte ends a part, -10 for '\n'.
 * A/a=3 C/c=2 G/g=1 T/t/U/u=0.  CuCLARK_hh.hh:263-295 (m_rTable), kmersConversion.cc:55-62. */
int orc_nt_code(uint8_t c);

/* Reverse complement of a k-mer value (2 bits/nt, first nt in the most significant pair).
 * CuClarkDB.cu:1256-1263, hashTable_hh.hh:478-486, kmersConversion.cc:39-47. */
uint64_t orc_revcomp(uint64_t kmer, int k);

/* min(kmer, revcomp).  CuClarkDB.cu:1266, hashTable_hh.hh:489. */
uint64_t orc_canonical(uint64_t kmer, int k);

/* k-mer value of the first k bytes of s (must all be ACGTU); kmersConversion.cc:49-68. */
uint64_t orc_kmer_from_ascii(const uint8_t* s, int k);

/* Width in bytes (2/4/8) of the stored key for (htsize,k): main.cc:274-316. */
int orc_key_bytes_rule(uint64_t htsize, int k);

/* ---- database ------------------------------------------------------------------------ */

typedef struct orc_db {
  uint64_t htsize;       /* number of buckets == size of .sz                      */
  int key_bytes;         /* 2, 4 or 8                                             */
  uint64_t n_elems;      /* elements kept after sampling                          */
  uint64_t* bucket_off;  /* [htsize+1] exclusive prefix sum of kept bucket sizes   */
  void* keys;            /* [n_elems] quotients, key_bytes wide                    */
  uint16_t* labels;      /* [n_elems]                                             */
  int borrowed;          /* keys/labels belong to the caller (orc_db_wrap_arrays) */
} orc_db;

/* Load <prefix>.sz/.ky/.lb.  htsize==0 => take it from the size of .sz.  sampling<=1 keeps every
 * bucket; otherwise only every sampling-th NON-EMPTY bucket is kept (the others become empty).
 * Layout: hashTable_hh.hh:590-663 (write); loader + sampling: CuClarkDB.cu:461-808 (esp. :497-524,
 * :594-648).  Returns NULL on failure. */
orc_db* orc_db_load(const char* prefix, uint64_t htsize, int key_bytes, uint32_t sampling);

/* Same, from in-memory copies of the three files. */
orc_db* orc_db_from_arrays(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                           const uint16_t* labels, uint32_t sampling);

/* Zero-copy variant for sampling <= 1: keys/labels are borrowed (must outlive the db); only the prefix sums
 * are built.  Used by bench.py's cpu_baseline leg for the 36 GB table. */
orc_db* orc_db_wrap_arrays(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                           const uint16_t* labels);

void orc_db_free(orc_db* db);

/* Probe one (forward-strand) k-mer.  Only buckets in [part_start, part_end) answer (DB shard
 * filter, CuClarkDB.cu:1272-1274); pass 0, htsize for the whole table.  Returns 1 and *label on a
 * hit.  CuClarkDB.cu:1249-1314 == hashTable_hh.hh:475-513. */
int orc_db_find(const orc_db* db, uint64_t kmer_fwd, int k, uint64_t part_start, uint64_t part_end,
                uint16_t* label);

/* Mean length of the probed bucket and hit count over a list of forward k-mers (bench
 * bookkeeping for the algorithmic-bytes figure of SURVEY.md §8d). */
void orc_probe_stats(const orc_db* db, const uint64_t* kmers, size_t n, int k, double* mean_bucket_len,
                     uint64_t* hits);

/* ---- A1: read packing (CuCLARK_hh.hh:1616-1716) --------------------------------------- */

/* Pack n_reads reads of `map` into the reference's container format.
 *   spos/epos/length: per read, first sequence byte, one-past-last sequence byte, and Length
 *   (non-newline bytes) exactly as the indexer produced them (CuCLARK_hh.hh:1339-1534).
 *   reads_pointer[n_reads+1], containers[cap] are outputs.
 * Returns the number of containers written, or (size_t)-1 if cap is too small. */
size_t orc_pack_batch(const uint8_t* map, const uint64_t* spos, const uint64_t* epos, const uint64_t* length,
                      size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap);

/* ---- A2-A4: query (CuClarkDB.cu:1045-1174) -------------------------------------------- */

/* Dense per-read per-target hit counts from a packed batch: counts[n_reads * n_targets] (u32,
 * zeroed by the callee).  Labels >= n_targets are reported through the return value (number of
 * such hits; the reference would write out of bounds). */
uint64_t orc_query_batch(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                         size_t n_reads, uint64_t part_start, uint64_t part_end, uint32_t n_targets,
                         uint32_t* counts);

/* {sum, idxBest, best, idxSecond, second} for every read of a packed batch (results[n_reads*5]), reads spread
 * over `threads` OpenMP threads (0 = runtime default).  The CPU baseline timed by bench.py. */
/* The same results, organised for memory throughput (three probe sweeps with software prefetch per read, sparse tally):
 * what bench.py times as the CPU baseline.  tests/test_oracle.py checks it against orc_classify_batch. */
uint64_t orc_classify_batch_fast(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                                 size_t n_reads, uint32_t n_targets, uint32_t* results, int threads);
/* A private copy of the table for that baseline: huge pages requested, written by all threads in static ranges so that
 * its pages are spread over the NUMA nodes (sampling is not applied: every bucket is kept). */
orc_db* orc_db_copy_spread(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                           int threads);
/* One replica of the table per NUMA node, threads pinned to the node whose replica they probe (no miss crosses the socket
 * link).  threads <= 0: all.  The results are orc_classify_batch's. */
#define ORC_MAX_NODES 8
typedef struct orc_numa_db {
  int n, threads;
  orc_db* db[ORC_MAX_NODES];
  unsigned long cpus[ORC_MAX_NODES][1024 / (8 * sizeof(unsigned long))];     /* cpu_set_t of each node (glibc: 1024 bits) */
} orc_numa_db;
orc_numa_db* orc_numa_db_create(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                                int threads);
void orc_numa_db_free(orc_numa_db* nd);
uint64_t orc_classify_batch_numa(const orc_numa_db* nd, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                                 size_t n_reads, uint32_t n_targets, uint32_t* results);
uint64_t orc_classify_batch(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                            size_t n_reads, uint32_t n_targets, uint32_t* results, int threads);

/* The same counts straight from sequence bytes, using the part rule of SURVEY appendix item 5
 * (maximal ACGTU runs, '\n' transparent, anything else ends a part; parts shorter than k and
 * reads whose Length < k contribute nothing).  `length` is the indexer's Length for the read. */
uint64_t orc_count_read_ascii(const orc_db* db, int k, const uint8_t* seq, size_t n_bytes, uint64_t length,
                              uint64_t part_start, uint64_t part_end, uint32_t n_targets, uint32_t* counts);

/* ---- A5-A7: sparse rows, merge, result ------------------------------------------------- */

/* Sparse row [n, t0,c0, t1,c1, ...] in ascending target order (CuClarkDB.cu:1178-1243).
 * Returns n (number of distinct targets); writes at most max_pairs pairs. */
uint32_t orc_sparse_row(const uint32_t* counts, uint32_t n_targets, uint16_t* row, uint32_t max_pairs);

/* Sum of two sparse rows by target (CuClarkDB.cu:1321-1415). */
void orc_merge_rows(const uint16_t* a, const uint16_t* b, uint16_t* out);

/* {sum, idxBest, best, idxSecond, second} from a sparse row (CuClarkDB.cu:1421-1471); indices are
 * target+1, 0 = "NA".  Values are kept in 32 bits (the reference's u16 wrap is not reproduced). */
void orc_result_from_row(const uint16_t* row, uint32_t out[5]);

/* The same from dense counts, iterating targets in ascending order. */
void orc_result_from_counts(const uint32_t* counts, uint32_t n_targets, uint32_t out[5]);

/* ---- H1: FASTA/FASTQ indexing (CuCLARK_hh.hh:1339-1534, one batch) -------------------- */

typedef struct orc_index {
  size_t n_reads;
  uint64_t* name_s; /* first byte of the name (after '>' / '@')                     */
  uint64_t* name_e; /* one past the name (first of ' ', '\t', '\n' or end of file)  */
  uint64_t* seq_s;  /* first sequence byte                                          */
  uint64_t* seq_e;  /* one past the last sequence byte                              */
  uint64_t* length; /* Length column before the paired-end correction               */
} orc_index;

/* Returns NULL if the first byte is neither '>' nor '@'. */
orc_index* orc_index_reads(const uint8_t* map, size_t nb);
void orc_index_free(orc_index* ix);

/* ---- H2: CSV (CuCLARK_hh.hh:1951-2139) ------------------------------------------------- */

/* Writes the header line.  target_names[0..n_targets) are the label names in label order (the
 * reference's m_targetsName[1..]); only used when extended != 0. */
int orc_csv_header(char* buf, size_t cap, int extended, const char* const* target_names, uint32_t n_targets);

/* One result line.  name/name_len: raw object name (truncated here to 39 bytes, :2114-2117).
 * length: indexer Length; paired != 0 subtracts NBN=1 (:2119).  res: {sum,idxBest,best,idxSecond,
 * second}.  counts (dense, may be NULL) is only used when extended != 0.  Returns bytes written
 * (excluding the NUL) or -1 if cap is too small. */
int orc_csv_line(char* buf, size_t cap, const uint8_t* name, size_t name_len, uint64_t length, int paired, int k,
                 const uint32_t res[5], const char* const* target_names, uint32_t n_targets, int extended,
                 const uint32_t* counts);

/* ---- whole-file convenience: index + count + result + CSV ------------------------------ */

/* Classifies every read of the FASTA/FASTQ image `map` against db and writes the CSV text into a
 * malloc'ed buffer (*csv, *csv_len).  results (may be NULL) receives n_reads*5 u32.  Returns the
 * number of reads or -1. */
long orc_classify_file(const orc_db* db, int k, const uint8_t* map, size_t nb, const char* const* target_names,
                       uint32_t n_targets, int paired, int extended, char** csv, size_t* csv_len,
                       uint32_t** results);

/* ---- the product's table partition for table-sharded runs (part_rule.c; NOT a rule of the reference) ----------
 * Slot, out of n_slots, of the minimizer of the k-mer that reads at nucleotide tpos of its read part; and the dense counts of
 * a packed batch restricted to the k-mer occurrences whose slot lies in part `part` of `n_parts` of the slots.  The sum of
 * the parts' counts is orc_query_batch's. */
uint32_t orc_part_slot_of_kmer(uint64_t kmer, uint32_t tpos, int k, int m, int both_strands, uint32_t n_slots);
uint64_t orc_query_batch_slot_part(const orc_db* db, int k, int m, int both_strands, uint32_t n_slots, uint32_t part,
                                   uint32_t n_parts, const uint32_t* reads_pointer, const uint16_t* containers, size_t n_reads,
                                   uint32_t n_targets, uint32_t* counts);

void orc_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
