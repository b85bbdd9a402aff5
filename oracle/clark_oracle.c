/*
 * clark_oracle.c — CPU restatement of CuCLARK's k-mer query hot path (see clark_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into or loaded by the product.
 * Parity: pinned against the reference's own CPU table (oracle/_ref, tests/golden).
 * Citations are file:line under /root/reference/src.
 */
#define _GNU_SOURCE
#include "clark_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static orc_db* orc_db_copy_pinned(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                                  int threads, const cpu_set_t* pin);

/* ------------------------------------------------------------------ codec */

/* CuCLARK_hh.hh:263-295: m_rTable is -1 everywhere except ACGTU (both cases), '>' = -2, '\n' = -10.
 * Only ">= 0" (extends a part) and "== '\n'" (transparent) matter to the packer (:1641-1689). */
int orc_nt_code(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 3;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 1;
    case 'T': case 't': case 'U': case 'u': return 0;
    case '\n': return -10;
    case '>': return -2;
    default: return -1;
  }
}

/* CuClarkDB.cu:1256-1263: reverse the 32 2-bit groups of the word, complement, drop the low
 * 64-2k bits. */
uint64_t orc_revcomp(uint64_t x, int k) {
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  x = (x >> 32) | (x << 32);
  return (~x) >> (64 - 2 * k);
}

uint64_t orc_canonical(uint64_t kmer, int k) {
  uint64_t r = orc_revcomp(kmer, k);
  return kmer < r ? kmer : r; /* CuClarkDB.cu:1266 */
}

uint64_t orc_kmer_from_ascii(const uint8_t* s, int k) {
  uint64_t v = 0;
  for (int i = 0; i < k; ++i) v = (v << 2) | (uint64_t)orc_nt_code(s[i]); /* kmersConversion.cc:49-68 */
  return v;
}

/* main.cc:274-316: t_b = floor(log(HTSIZE)/log(4)) computed in double; k<=t_b+8 -> 2 bytes,
 * k<=t_b+16 -> 4 bytes, else 8. */
int orc_key_bytes_rule(uint64_t htsize, int k) {
  size_t t_b = (size_t)(log((double)htsize) / log(4.0));
  if ((size_t)k <= t_b + 8) return 2;
  if ((size_t)k <= t_b + 16) return 4;
  return 8;
}

/* --------------------------------------------------------------- database */

static uint64_t key_at(const orc_db* db, uint64_t i) {
  switch (db->key_bytes) {
    case 2: return ((const uint16_t*)db->keys)[i];
    case 4: return ((const uint32_t*)db->keys)[i];
    default: return ((const uint64_t*)db->keys)[i];
  }
}

/* CuClarkDB.cu:497-524 chooses buckets (choice[i]==2 keeps), :594-648 builds the exclusive prefix
 * sum over kept sizes, :678-782 copies keys/labels of kept buckets in file order. */
orc_db* orc_db_from_arrays(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                           const uint16_t* labels, uint32_t sampling) {
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return NULL;
  orc_db* db = (orc_db*)calloc(1, sizeof(orc_db));
  if (!db) return NULL;
  db->htsize = htsize;
  db->key_bytes = key_bytes;
  db->bucket_off = (uint64_t*)malloc((htsize + 1) * sizeof(uint64_t));
  if (!db->bucket_off) { free(db); return NULL; }
  const int all = sampling <= 1;
  uint64_t kept = 0, nonzero = 0;
  for (uint64_t i = 0; i < htsize; ++i) {
    db->bucket_off[i] = kept;
    if (sizes[i] > 0) {
      ++nonzero;
      if (all || (nonzero % sampling) == 0) kept += sizes[i];
    }
  }
  db->bucket_off[htsize] = kept;
  db->n_elems = kept;
  db->keys = malloc(kept * (size_t)key_bytes + 8);
  db->labels = (uint16_t*)malloc(kept * sizeof(uint16_t) + 8);
  if (!db->keys || !db->labels) { orc_db_free(db); return NULL; }
  uint64_t src = 0;
  nonzero = 0;
  for (uint64_t i = 0; i < htsize; ++i) {
    if (sizes[i] == 0) continue;
    ++nonzero;
    if (all || (nonzero % sampling) == 0) {
      uint64_t dst = db->bucket_off[i];
      memcpy((char*)db->keys + dst * key_bytes, (const char*)keys + src * key_bytes, (size_t)sizes[i] * key_bytes);
      memcpy(db->labels + dst, labels + src, (size_t)sizes[i] * sizeof(uint16_t));
    }
    src += sizes[i];
  }
  return db;
}

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc((size_t)sz + 8);
  if (p && sz > 0 && fread(p, 1, (size_t)sz, f) != (size_t)sz) { free(p); p = NULL; }
  fclose(f);
  if (p) *n = (size_t)sz;
  return p;
}

orc_db* orc_db_load(const char* prefix, uint64_t htsize, int key_bytes, uint32_t sampling) {
  size_t len = strlen(prefix);
  char* path = (char*)malloc(len + 8);
  size_t n_sz = 0, n_ky = 0, n_lb = 0;
  sprintf(path, "%s.sz", prefix); /* hashTable_hh.hh:593-598 */
  uint8_t* sz = (uint8_t*)slurp(path, &n_sz);
  sprintf(path, "%s.ky", prefix);
  void* ky = slurp(path, &n_ky);
  sprintf(path, "%s.lb", prefix);
  uint16_t* lb = (uint16_t*)slurp(path, &n_lb);
  free(path);
  orc_db* db = NULL;
  if (sz && ky && lb) {
    if (htsize == 0) htsize = n_sz;
    uint64_t total = 0;
    for (uint64_t i = 0; i < htsize && i < n_sz; ++i) total += sz[i];
    if (htsize <= n_sz && total * (uint64_t)key_bytes <= n_ky && total * 2 <= n_lb)
      db = orc_db_from_arrays(sz, htsize, ky, key_bytes, lb, sampling);
  }
  free(sz); free(ky); free(lb);
  return db;
}

orc_db* orc_db_wrap_arrays(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes,
                           const uint16_t* labels) {
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return NULL;
  orc_db* db = (orc_db*)calloc(1, sizeof(orc_db));
  if (!db) return NULL;
  db->htsize = htsize; db->key_bytes = key_bytes; db->borrowed = 1;
  db->bucket_off = (uint64_t*)malloc((htsize + 1) * sizeof(uint64_t));
  if (!db->bucket_off) { free(db); return NULL; }
  uint64_t run = 0;
  for (uint64_t i = 0; i < htsize; ++i) { db->bucket_off[i] = run; run += sizes[i]; }
  db->bucket_off[htsize] = run;
  db->n_elems = run;
  db->keys = (void*)keys; db->labels = (uint16_t*)labels;
  return db;
}

static void big_free(void* p, size_t bytes);
void orc_db_free(orc_db* db) {
  if (!db) return;
  if (db->borrowed == 2) {            /* orc_db_copy_spread */
    big_free(db->bucket_off, (db->htsize + 1) * sizeof(uint64_t));
    big_free(db->keys, db->n_elems * (size_t)db->key_bytes + 64);
    big_free(db->labels, db->n_elems * sizeof(uint16_t) + 64);
    free(db);
    return;
  }
  free(db->bucket_off);
  if (!db->borrowed) { free(db->keys); free(db->labels); }
  free(db);
}

/* CuClarkDB.cu:1249-1314.  quotient is compared at full width against the stored key
 * (:1291,:1296-1298); the scan walks keys in storage order while key <= quotient, which is only
 * guaranteed to stay inside the bucket because of the last-key pre-check (:1291). */
int orc_db_find(const orc_db* db, uint64_t kmer_fwd, int k, uint64_t part_start, uint64_t part_end,
                uint16_t* label) {
  uint64_t c = orc_canonical(kmer_fwd, k);
  uint64_t quotient = c / db->htsize;
  uint64_t remainder = c - quotient * db->htsize;
  if (remainder < part_start || remainder >= part_end) return 0; /* :1272-1274 */
  uint64_t b = db->bucket_off[remainder], e = db->bucket_off[remainder + 1];
  if (e - b == 0) return 0;
  uint64_t i = b;
  uint64_t key = key_at(db, i);
  if (key > quotient || key_at(db, e - 1) < quotient) return 0;
  while (key <= quotient) {
    if (key == quotient) { *label = db->labels[i]; return 1; }
    key = key_at(db, ++i);
  }
  return 0;
}

void orc_probe_stats(const orc_db* db, const uint64_t* kmers, size_t n, int k, double* mean_bucket_len,
                     uint64_t* hits) {
  uint64_t tot = 0, h = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t c = orc_canonical(kmers[i], k);
    uint64_t r = c % db->htsize;
    tot += db->bucket_off[r + 1] - db->bucket_off[r];
    uint16_t l;
    h += (uint64_t)orc_db_find(db, kmers[i], k, 0, db->htsize, &l);
  }
  if (mean_bucket_len) *mean_bucket_len = n ? (double)tot / (double)n : 0.0;
  if (hits) *hits = h;
}

/* ------------------------------------------------------------ A1: packing */

/* Parts longer than this are emitted as consecutive sub-parts that overlap by k-1 nucleotides, so
 * the multiset of k-mers is unchanged.  (The reference keeps the part length in a u16 container,
 * CuCLARK_hh.hh:1646,1669, and silently wraps for longer parts — SURVEY appendix item 8, not
 * reproduced.)  The product packer uses the same constant. */
#define ORC_MAX_PART 65528u

typedef struct {
  uint16_t* out; size_t cap; size_t n; int overflow;
} cont_sink;

static void sink_put(cont_sink* s, uint16_t v) {
  if (s->n < s->cap) s->out[s->n] = v; else s->overflow = 1;
  s->n++;
}

/* Emit one maximal run of codes[0..len) (len >= k) as one or more parts:
 * [len][ceil(len/8) containers, 8 nt each, first nt in the top bits, last container left-aligned]
 * (CuCLARK_hh.hh:1661-1671,1683,1695). */
static void emit_run(cont_sink* s, const uint8_t* codes, size_t len, int k) {
  size_t start = 0;
  for (;;) {
    size_t plen = len - start;
    if (plen > ORC_MAX_PART) plen = ORC_MAX_PART;
    sink_put(s, (uint16_t)plen);
    uint16_t c = 0; unsigned cur = 0;
    for (size_t i = 0; i < plen; ++i) {
      c = (uint16_t)((c << 2) | codes[start + i]);
      if (++cur == 8) { sink_put(s, c); c = 0; cur = 0; }
    }
    if (cur) sink_put(s, (uint16_t)(c << (2 * (8 - cur))));
    if (start + plen >= len) break;
    start += plen - (size_t)(k - 1);
  }
}

size_t orc_pack_batch(const uint8_t* map, const uint64_t* spos, const uint64_t* epos, const uint64_t* length,
                      size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap) {
  cont_sink s = {containers, cap, 0, 0};
  uint8_t* codes = NULL; size_t codes_cap = 0;
  for (size_t r = 0; r < n_reads; ++r) {
    reads_pointer[r] = (uint32_t)s.n;
    if (length[r] < (uint64_t)k) continue; /* :1633 */
    size_t nb = (size_t)(epos[r] - spos[r]);
    if (nb > codes_cap) { codes_cap = nb * 2 + 64; codes = (uint8_t*)realloc(codes, codes_cap); }
    size_t run = 0;
    for (size_t i = 0; i <= nb; ++i) {
      int code = i < nb ? orc_nt_code(map[spos[r] + i]) : -1;
      if (code >= 0) { codes[run++] = (uint8_t)code; continue; } /* :1641-1672 */
      if (i < nb && map[spos[r] + i] == '\n') continue;           /* :1674-1678 */
      if (run >= (size_t)k) emit_run(&s, codes, run, k);          /* :1647-1651,:1679-1704 */
      run = 0;
    }
  }
  reads_pointer[n_reads] = (uint32_t)s.n;
  free(codes);
  return s.overflow ? (size_t)-1 : s.n;
}

/* ---------------------------------------------------------- A2-A4: query */

static uint64_t tally(const orc_db* db, uint64_t kmer, int k, uint64_t ps, uint64_t pe, uint32_t n_targets,
                      uint32_t* counts) {
  uint16_t label;
  if (orc_db_find(db, kmer, k, ps, pe, &label)) {
    if (label < n_targets) counts[label]++; /* CuClarkDB.cu:1156-1166 */
    else return 1;
  }
  return 0;
}

uint64_t orc_query_batch(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                         size_t n_reads, uint64_t part_start, uint64_t part_end, uint32_t n_targets,
                         uint32_t* counts) {
  const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); /* CuClarkDB.cu:1078 */
  uint64_t bad = 0;
  memset(counts, 0, n_reads * (size_t)n_targets * sizeof(uint32_t));
  for (size_t r = 0; r < n_reads; ++r) {
    uint32_t* row = counts + r * (size_t)n_targets;
    uint32_t p = reads_pointer[r], end = reads_pointer[r + 1];
    while (p < end) { /* CuClarkDB.cu:1090-1097 */
      uint32_t plen = containers[p];
      if (plen == 0) break; /* terminator used by padded batches of the product; never in reference data */
      uint32_t first = p + 1;
      p = first + (plen - 1) / 8 + 1;
      uint64_t kmer = 0;
      for (uint32_t i = 0; i < plen; ++i) {
        uint32_t nt = (containers[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
        kmer = ((kmer << 2) | nt) & cutoff; /* value of nts (i-k, i] : CuClarkDB.cu:1100-1135 */
        if (i + 1 >= (uint32_t)k) bad += tally(db, kmer, k, part_start, part_end, n_targets, row);
      }
    }
  }
  return bad;
}

uint64_t orc_classify_batch(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                            size_t n_reads, uint32_t n_targets, uint32_t* results, int threads) {
  const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t bad = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel reduction(+ : bad)
  {
    uint32_t* counts = (uint32_t*)calloc(n_targets ? n_targets : 1, sizeof(uint32_t));
#pragma omp for schedule(dynamic, 256)
    for (long r = 0; r < (long)n_reads; ++r) {
      uint32_t p = reads_pointer[r], end = reads_pointer[r + 1];
      while (p < end) {
        uint32_t plen = containers[p];
        if (plen == 0) break;
        uint32_t first = p + 1;
        p = first + (plen - 1) / 8 + 1;
        uint64_t kmer = 0;
        for (uint32_t i = 0; i < plen; ++i) {
          uint32_t nt = (containers[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
          kmer = ((kmer << 2) | nt) & cutoff;
          if (i + 1 >= (uint32_t)k) bad += tally(db, kmer, k, 0, db->htsize, n_targets, counts);
        }
      }
      orc_result_from_counts(counts, n_targets, results + 5 * (size_t)r);
      memset(counts, 0, (size_t)n_targets * sizeof(uint32_t));
    }
    free(counts);
  }
  return bad;
}

/* ---- throughput form of orc_classify_batch (bench.py's cpu_baseline) --------------------------------------------
 * Same rules, same results; what changes is how a core keeps memory busy:
 *   - the k-mers of a read are collected first, then the probes run as three sweeps with software prefetch between them
 *     (bucket bounds -> first keys of the bucket -> scan), so a core has dozens of cache misses in flight instead of two
 *     dependent ones per k-mer;
 *   - hits are tallied in a per-thread table of which only the touched entries are read back and cleared (ascending target
 *     order, so the tie rule of CuClarkDB.cu:1440-1459 sees the same sequence), not all T per read. */
#define ORC_FAST_MAX 1024
static uint64_t classify_fast(const orc_db* db0, const orc_numa_db* nd, int k, const uint32_t* reads_pointer,
                              const uint16_t* containers, size_t n_reads, uint32_t n_targets, uint32_t* results, int threads);

uint64_t orc_classify_batch_fast(const orc_db* db, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                                 size_t n_reads, uint32_t n_targets, uint32_t* results, int threads) {
  return classify_fast(db, NULL, k, reads_pointer, containers, n_reads, n_targets, results, threads);
}

uint64_t orc_classify_batch_numa(const orc_numa_db* nd, int k, const uint32_t* reads_pointer, const uint16_t* containers,
                                 size_t n_reads, uint32_t n_targets, uint32_t* results) {
  return classify_fast(nd->db[0], nd, k, reads_pointer, containers, n_reads, n_targets, results, nd->threads);
}

static uint64_t classify_fast(const orc_db* db0, const orc_numa_db* nd, int k, const uint32_t* reads_pointer,
                              const uint16_t* containers, size_t n_reads, uint32_t n_targets, uint32_t* results, int threads) {
  const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t H = db0->htsize;
  const int kb = db0->key_bytes;
  uint64_t bad = 0;
  int G = 32;                                     /* k-mers per probe group */
  { const char* env = getenv("ORC_FAST_GROUP"); if (env && atoi(env) > 0) G = atoi(env); }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel reduction(+ : bad)
  {
    const orc_db* db = db0;
    cpu_set_t old_mask;
    if (nd) {       /* thread t works on node t % nodes with that node's replica */
#ifdef _OPENMP
      const int node = omp_get_thread_num() % nd->n;
#else
      const int node = 0;
#endif
      sched_getaffinity(0, sizeof(old_mask), &old_mask);
      sched_setaffinity(0, sizeof(cpu_set_t), (const cpu_set_t*)&nd->cpus[node]);
      db = nd->db[node];
    }
    uint32_t* counts = (uint32_t*)calloc(n_targets ? n_targets : 1, sizeof(uint32_t));
    uint32_t* touched = (uint32_t*)malloc((n_targets ? n_targets : 1) * sizeof(uint32_t));
    uint64_t quot[ORC_FAST_MAX], rem[ORC_FAST_MAX], b0[ORC_FAST_MAX], b1[ORC_FAST_MAX];
#pragma omp for schedule(dynamic, 64)
    for (long r = 0; r < (long)n_reads; ++r) {
      uint32_t p = reads_pointer[r], end = reads_pointer[r + 1];
      uint32_t n_touched = 0;
      size_t nk = 0;
      int more = 1;
      while (more) {
        /* sweep 0: k-mers of the read (all its parts), ORC_FAST_MAX at a time */
        more = 0;
        while (p < end) {
          uint32_t plen = containers[p];
          if (plen == 0) { p = end; break; }
          uint32_t first = p + 1;
          uint64_t kmer = 0;
          uint32_t i0 = 0;
          if (plen >= (uint32_t)k && nk + (plen - k + 1) > ORC_FAST_MAX) {      /* very long part: probe it the plain way */
            for (uint32_t i = 0; i < plen; ++i) {
              uint32_t nt = (containers[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
              kmer = ((kmer << 2) | nt) & cutoff;
              if (i + 1 >= (uint32_t)k) {
                uint16_t label;
                if (orc_db_find(db, kmer, k, 0, H, &label)) {
                  if (label < n_targets) { if (counts[label]++ == 0) touched[n_touched++] = label; } else ++bad;
                }
              }
            }
            p = first + (plen - 1) / 8 + 1;
            continue;
          }
          for (uint32_t i = i0; i < plen; ++i) {
            uint32_t nt = (containers[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
            kmer = ((kmer << 2) | nt) & cutoff;
            if (i + 1 >= (uint32_t)k) {
              uint64_t c = orc_canonical(kmer, k);
              quot[nk] = c / H; rem[nk] = c - quot[nk] * H;
              ++nk;
            }
          }
          p = first + (plen - 1) / 8 + 1;
        }
        /* the three probe stages run one group of G k-mers apart (a core tracks a few dozen misses; prefetches beyond
         * that are dropped): prefetch of the bucket bounds for group g0, bounds + key prefetch for g0 - G, scan for g0 - 2G */
        for (size_t g0 = 0; g0 < nk + 2 * (size_t)G; g0 += (size_t)G) {
        for (size_t i = g0; i < nk && i < g0 + (size_t)G; ++i) __builtin_prefetch(&db->bucket_off[rem[i]]);
        if (g0 < (size_t)G) continue;
        /* stage 2 (group g0 - G): bucket bounds, prefetch of the first keys (and of the last one, for the pre-check) */
        for (size_t i = g0 - (size_t)G; i < nk && i < g0; ++i) {
          b0[i] = db->bucket_off[rem[i]]; b1[i] = db->bucket_off[rem[i] + 1];
          if (b1[i] > b0[i]) {
            __builtin_prefetch((const char*)db->keys + b0[i] * kb);
            __builtin_prefetch((const char*)db->keys + (b1[i] - 1) * kb);
            __builtin_prefetch(db->labels + b0[i]);
          }
        }
        /* stage 3 (group g0 - 2G): the scan of CuClarkDB.cu:1291-1307 */
        if (g0 < 2 * (size_t)G) continue;
        for (size_t i = g0 - 2 * (size_t)G; i < nk && i < g0 - (size_t)G; ++i) {
          uint64_t b = b0[i], e = b1[i];
          if (e == b) continue;
          uint64_t key = key_at(db, b);
          if (key > quot[i] || key_at(db, e - 1) < quot[i]) continue;
          uint64_t j = b;
          while (key <= quot[i]) {
            if (key == quot[i]) {
              uint16_t label = db->labels[j];
              if (label < n_targets) { if (counts[label]++ == 0) touched[n_touched++] = label; } else ++bad;
              break;
            }
            key = key_at(db, ++j);
          }
        }
        }
        nk = 0;
      }
      /* result over the touched targets in ascending order (CuClarkDB.cu:1440-1459), then clear them */
      for (uint32_t a = 1; a < n_touched; ++a) {      /* insertion sort: a handful of entries */
        uint32_t v = touched[a], j = a;
        while (j > 0 && touched[j - 1] > v) { touched[j] = touched[j - 1]; --j; }
        touched[j] = v;
      }
      uint32_t* out = results + 5 * (size_t)r;
      memset(out, 0, 5 * sizeof(uint32_t));
      for (uint32_t a = 0; a < n_touched; ++a) {
        const uint32_t t = touched[a], sc = counts[t];
        if (sc > out[2]) { out[4] = out[2]; out[3] = out[1]; out[2] = sc; out[1] = t + 1; }
        else if (sc > out[4]) { out[4] = sc; out[3] = t + 1; }
        out[0] += sc;
        counts[t] = 0;
      }
    }
    free(counts); free(touched);
    if (nd) sched_setaffinity(0, sizeof(old_mask), &old_mask);
  }
  return bad;
}

/* A private copy of the table spread over the machine: every array is allocated with huge pages requested and written by
 * all threads in static ranges (first touch), so its pages sit on every NUMA node and a probe does not pay a TLB miss
 * per access.  bench.py hands the oracle the arrays it downloaded from the GPU; one thread touching 40+ GB puts them
 * all on that thread's node. */
#include <sys/mman.h>
static void* big_alloc(size_t bytes) {
  bytes = (bytes + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1);
  void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return NULL;
#ifdef MADV_HUGEPAGE
  madvise(p, bytes, MADV_HUGEPAGE);
#endif
  return p;
}
static void big_free(void* p, size_t bytes) {
  if (p) munmap(p, (bytes + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1));
}

orc_db* orc_db_copy_spread(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                           int threads) {
  return orc_db_copy_pinned(sizes, htsize, keys, key_bytes, labels, threads, NULL);
}

/* pin != NULL: every copying thread binds itself to that CPU set first (one replica per NUMA node) */
static orc_db* orc_db_copy_pinned(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                           int threads, const cpu_set_t* pin) {
  if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return NULL;
  orc_db* db = (orc_db*)calloc(1, sizeof(orc_db));
  if (!db) return NULL;
  db->htsize = htsize; db->key_bytes = key_bytes; db->borrowed = 2;      /* 2: arrays come from big_alloc */
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  const int nt = omp_get_max_threads();
#else
  const int nt = 1; (void)threads;
#endif
  uint64_t* part = (uint64_t*)calloc((size_t)nt + 1, sizeof(uint64_t));
  db->bucket_off = (uint64_t*)big_alloc((htsize + 1) * sizeof(uint64_t));
  if (!db->bucket_off || !part) { free(part); free(db); return NULL; }
  const uint64_t per = (htsize + nt - 1) / nt;
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    if (pin) sched_setaffinity(0, sizeof(cpu_set_t), pin);
    const uint64_t lo = (uint64_t)t * per < htsize ? (uint64_t)t * per : htsize, hi = lo + per < htsize ? lo + per : htsize;
    uint64_t sum = 0;
    for (uint64_t i = lo; i < hi; ++i) sum += sizes[i];
    part[t + 1] = sum;
#pragma omp barrier
#pragma omp single
    { for (int i = 0; i < nt; ++i) part[i + 1] += part[i]; }
    uint64_t run = part[t];
    for (uint64_t i = lo; i < hi; ++i) { db->bucket_off[i] = run; run += sizes[i]; }
  }
  const uint64_t n = part[nt];
  db->bucket_off[htsize] = n;
  db->n_elems = n;
  db->keys = big_alloc(n * (size_t)key_bytes + 64);
  db->labels = (uint16_t*)big_alloc(n * sizeof(uint16_t) + 64);
  if (!db->keys || !db->labels) { free(part); orc_db_free(db); return NULL; }
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    if (pin) sched_setaffinity(0, sizeof(cpu_set_t), pin);
    const uint64_t eper = (n + nt - 1) / nt;
    const uint64_t lo = (uint64_t)t * eper < n ? (uint64_t)t * eper : n, hi = lo + eper < n ? lo + eper : n;
    memcpy((char*)db->keys + lo * key_bytes, (const char*)keys + lo * key_bytes, (size_t)(hi - lo) * key_bytes);
    memcpy(db->labels + lo, labels + lo, (size_t)(hi - lo) * sizeof(uint16_t));
  }
  free(part);
  return db;
}

/* ---- one table replica per NUMA node ----------------------------------------------------------------------------------
 * A probe is two or three random cache misses; on a two-socket host half of them cross the socket link when the table is
 * spread over both nodes.  With one replica per node (288 GB-class hosts have the RAM) every miss is local: thread t is
 * pinned to node t % nodes for the copy (first touch) and for the classification. */
static int node_cpus(int node, cpu_set_t* set) {
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  CPU_ZERO(set);
  int a, b, any = 0; char c;
  while (fscanf(f, "%d", &a) == 1) {
    b = a; c = 0;
    if (fscanf(f, "%c", &c) == 1 && c == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &c) != 1) c = 0; }
    for (int i = a; i <= b && i < CPU_SETSIZE; ++i) { CPU_SET(i, set); any = 1; }
    if (c != ',') break;
  }
  fclose(f);
  return any;
}

orc_numa_db* orc_numa_db_create(const uint8_t* sizes, uint64_t htsize, const void* keys, int key_bytes, const uint16_t* labels,
                                int threads) {
  orc_numa_db* nd = (orc_numa_db*)calloc(1, sizeof(orc_numa_db));
  if (!nd) return NULL;
  cpu_set_t allowed, set;
  sched_getaffinity(0, sizeof(allowed), &allowed);
  for (int node = 0; node < ORC_MAX_NODES; ++node) {
    if (!node_cpus(node, &set)) break;
    CPU_AND(&set, &set, &allowed);
    if (CPU_COUNT(&set) == 0) continue;
    memcpy(&nd->cpus[nd->n], &set, sizeof(set));
    nd->n++;
  }
  if (nd->n == 0) { memcpy(&nd->cpus[0], &allowed, sizeof(allowed)); nd->n = 1; }
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
  nd->threads = threads;
  for (int r = 0; r < nd->n; ++r) {
    /* the replica is written by threads pinned to node r: first touch puts its pages there */
    cpu_set_t old; sched_getaffinity(0, sizeof(old), &old);
    sched_setaffinity(0, sizeof(cpu_set_t), (cpu_set_t*)&nd->cpus[r]);
    int per = threads / nd->n > 0 ? threads / nd->n : 1;
    nd->db[r] = orc_db_copy_pinned(sizes, htsize, keys, key_bytes, labels, per, (const cpu_set_t*)&nd->cpus[r]);
    sched_setaffinity(0, sizeof(old), &old);
    if (!nd->db[r]) { orc_numa_db_free(nd); return NULL; }
  }
  return nd;
}

void orc_numa_db_free(orc_numa_db* nd) {
  if (!nd) return;
  for (int r = 0; r < nd->n; ++r) orc_db_free(nd->db[r]);
  free(nd);
}

uint64_t orc_count_read_ascii(const orc_db* db, int k, const uint8_t* seq, size_t n_bytes, uint64_t length,
                              uint64_t part_start, uint64_t part_end, uint32_t n_targets, uint32_t* counts) {
  const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t bad = 0, kmer = 0;
  size_t run = 0;
  if (length < (uint64_t)k) return 0;
  for (size_t i = 0; i < n_bytes; ++i) {
    int code = orc_nt_code(seq[i]);
    if (code >= 0) {
      kmer = ((kmer << 2) | (uint64_t)code) & cutoff;
      if (++run >= (size_t)k) bad += tally(db, kmer, k, part_start, part_end, n_targets, counts);
    } else if (seq[i] != '\n') {
      run = 0; kmer = 0;
    }
  }
  return bad;
}

/* ------------------------------------------------- A5-A7: rows and result */

uint32_t orc_sparse_row(const uint32_t* counts, uint32_t n_targets, uint16_t* row, uint32_t max_pairs) {
  uint32_t n = 0;
  for (uint32_t t = 0; t < n_targets; ++t) {
    if (counts[t] == 0) continue;
    if (n < max_pairs) { row[1 + 2 * n] = (uint16_t)t; row[2 + 2 * n] = (uint16_t)counts[t]; }
    ++n;
  }
  row[0] = (uint16_t)n;
  return n;
}

/* Two-pointer merge by ascending target, summing on equal targets (CuClarkDB.cu:1321-1415). */
void orc_merge_rows(const uint16_t* a, const uint16_t* b, uint16_t* out) {
  uint32_t na = a[0], nb = b[0], ia = 0, ib = 0, n = 0;
  while (ia < na || ib < nb) {
    uint16_t t, c;
    if (ib >= nb || (ia < na && a[1 + 2 * ia] < b[1 + 2 * ib])) { t = a[1 + 2 * ia]; c = a[2 + 2 * ia]; ++ia; }
    else if (ia >= na || b[1 + 2 * ib] < a[1 + 2 * ia]) { t = b[1 + 2 * ib]; c = b[2 + 2 * ib]; ++ib; }
    else { t = a[1 + 2 * ia]; c = (uint16_t)(a[2 + 2 * ia] + b[2 + 2 * ib]); ++ia; ++ib; }
    out[1 + 2 * n] = t; out[2 + 2 * n] = c; ++n;
  }
  out[0] = (uint16_t)n;
}

static void result_step(uint32_t target, uint32_t score, uint32_t out[5]) {
  /* CuClarkDB.cu:1440-1459: strict '>' for best, 'else if >' for second; indices are target+1 */
  if (score > out[2]) { out[4] = out[2]; out[3] = out[1]; out[2] = score; out[1] = target + 1; }
  else if (score > out[4]) { out[4] = score; out[3] = target + 1; }
  out[0] += score;
}

void orc_result_from_row(const uint16_t* row, uint32_t out[5]) {
  memset(out, 0, 5 * sizeof(uint32_t));
  for (uint32_t i = 0; i < row[0]; ++i) result_step(row[1 + 2 * i], row[2 + 2 * i], out);
}

void orc_result_from_counts(const uint32_t* counts, uint32_t n_targets, uint32_t out[5]) {
  memset(out, 0, 5 * sizeof(uint32_t));
  for (uint32_t t = 0; t < n_targets; ++t)
    if (counts[t]) result_step(t, counts[t], out);
}

/* ------------------------------------------------------------ H1: indexing */

static int is_sep(uint8_t c) { return c == ' ' || c == '\t' || c == '\n'; } /* CuCLARK_hh.hh:300 */

typedef struct { uint64_t* p; size_t n, cap; } u64vec;
static void push(u64vec* v, uint64_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (uint64_t*)realloc(v->p, v->cap * 8); }
  v->p[v->n++] = x;
}

orc_index* orc_index_reads(const uint8_t* map, size_t nb) {
  if (nb == 0 || (map[0] != '>' && map[0] != '@')) return NULL;
  u64vec ns = {0}, ne = {0}, ss = {0}, se = {0}, ln = {0};
  size_t i = 1;
  if (map[0] == '>') { /* CuCLARK_hh.hh:1340-1396 with a single batch */
    for (;;) {
      push(&ns, i);
      while (i < nb && !(i + 1 < nb ? is_sep(map[i + 1]) : 1)) ++i; /* :1372 (pre-increment scan) */
      ++i;
      if (i > nb) i = nb;
      push(&ne, i);
      while (i < nb && map[i++] != '\n') {}
      uint64_t s = i, e = i, lines = 0;
      while (i < nb && map[i] != '>') {
        while (i < nb && map[i] != '\n') ++i;
        ++lines;
        e = i++;
      }
      push(&ss, s); push(&se, e);
      push(&ln, (uint64_t)((int64_t)(e - s + 1) - (int64_t)lines)); /* :1385-1389 */
      if (i >= nb) break;
      ++i; /* skip '>' */
    }
  } else { /* FASTQ, :1474-1529 */
    for (;;) {
      push(&ns, i);
      while (i < nb && !(i + 1 < nb ? is_sep(map[i + 1]) : 1)) ++i;
      ++i;
      if (i > nb) i = nb;
      push(&ne, i);
      while (i < nb && map[i++] != '\n') {}
      uint64_t s = i;
      while (i < nb && map[i] != '\n') ++i;
      uint64_t e = i++;
      push(&ss, s); push(&se, e < nb ? e : nb);
      push(&ln, (e < nb ? e : nb) - s);
      while (i < nb && map[i++] != '\n') {}
      while (i < nb && map[i++] != '\n') {}
      if (++i >= nb) break;
    }
  }
  orc_index* ix = (orc_index*)calloc(1, sizeof(orc_index));
  ix->n_reads = ns.n; ix->name_s = ns.p; ix->name_e = ne.p; ix->seq_s = ss.p; ix->seq_e = se.p; ix->length = ln.p;
  return ix;
}

void orc_index_free(orc_index* ix) {
  if (!ix) return;
  free(ix->name_s); free(ix->name_e); free(ix->seq_s); free(ix->seq_e); free(ix->length); free(ix);
}

/* ------------------------------------------------------------------ H2: CSV */

int orc_csv_header(char* buf, size_t cap, int extended, const char* const* target_names, uint32_t n_targets) {
  size_t n = 0;
#define APPEND(...) do { int w_ = snprintf(buf + n, n < cap ? cap - n : 0, __VA_ARGS__); if (w_ < 0) return -1; n += (size_t)w_; } while (0)
  APPEND("Object_ID"); /* CuCLARK_hh.hh:1957-1972 */
  if (extended) for (uint32_t t = 0; t < n_targets; ++t) APPEND(",%s", target_names[t]);
  APPEND(",Length,Gamma,1st_assignment,score1,2nd_assignment,score2,confidence\n");
  return n < cap ? (int)n : -1;
}

int orc_csv_line(char* buf, size_t cap, const uint8_t* name, size_t name_len, uint64_t length, int paired, int k,
                 const uint32_t res[5], const char* const* target_names, uint32_t n_targets, int extended,
                 const uint32_t* counts) {
  size_t n = 0;
  char obj[40];
  if (name_len >= 40) name_len = 39; /* OBJECTNAMEMAX, :2114-2117 */
  memcpy(obj, name, name_len); obj[name_len] = 0;
  uint32_t norm = paired ? (uint32_t)length - 1u : (uint32_t)length; /* ITYPE arithmetic, :2119 */
  uint32_t total = res[0], ib = res[1], best = res[2], is = res[3], sbest = res[4];
  double gamma = (double)total / (((double)norm - (double)k) + 1.0); /* :2127 */
  double delta = (double)(best + sbest);
  delta = (delta < 0.001) ? 0 : ((double)best) / delta; /* :2128-2129 */
  const char* n1 = ib == 0 ? "NA" : target_names[ib - 1]; /* m_targetsName[0]="NA", :1879-1883 */
  const char* n2 = is == 0 ? "NA" : target_names[is - 1];
  APPEND("%s", obj);
  if (extended) for (uint32_t t = 0; t < n_targets; ++t) APPEND(",%u", counts ? counts[t] : 0u); /* :2014-2031 */
  APPEND(",%u,%g,%s,%u,%s,%u,%g\n", norm, gamma, n1, best, n2, sbest, delta); /* :2132 */
#undef APPEND
  return n < cap ? (int)n : -1;
}

/* ------------------------------------------------------- whole-file helper */

long orc_classify_file(const orc_db* db, int k, const uint8_t* map, size_t nb, const char* const* target_names,
                       uint32_t n_targets, int paired, int extended, char** csv, size_t* csv_len,
                       uint32_t** results) {
  orc_index* ix = orc_index_reads(map, nb);
  if (!ix) return -1;
  size_t cap = 1 << 16, n = 0;
  char* out = (char*)malloc(cap);
  size_t line_cap = 256 + (extended ? (size_t)n_targets * 48 : 0);
  uint32_t* counts = (uint32_t*)calloc(n_targets ? n_targets : 1, sizeof(uint32_t));
  uint32_t* res_all = results ? (uint32_t*)malloc(ix->n_reads * 5 * sizeof(uint32_t) + 8) : NULL;
  while (cap < line_cap + 64) cap *= 2;
  out = (char*)realloc(out, cap);
  int w = orc_csv_header(out, cap, extended, target_names, n_targets);
  n = (size_t)w;
  for (size_t r = 0; r < ix->n_reads; ++r) {
    memset(counts, 0, n_targets * sizeof(uint32_t));
    orc_count_read_ascii(db, k, map + ix->seq_s[r], (size_t)(ix->seq_e[r] - ix->seq_s[r]), ix->length[r], 0,
                         db->htsize, n_targets, counts);
    uint32_t res[5];
    orc_result_from_counts(counts, n_targets, res);
    if (res_all) memcpy(res_all + 5 * r, res, sizeof(res));
    if (n + line_cap > cap) { while (n + line_cap > cap) cap *= 2; out = (char*)realloc(out, cap); }
    w = orc_csv_line(out + n, cap - n, map + ix->name_s[r], (size_t)(ix->name_e[r] - ix->name_s[r]), ix->length[r],
                     paired, k, res, target_names, n_targets, extended, counts);
    n += (size_t)w;
  }
  long nr = (long)ix->n_reads;
  free(counts);
  orc_index_free(ix);
  *csv = out; *csv_len = n;
  if (results) *results = res_all;
  return nr;
}

void orc_free(void* p) { free(p); }
