/*
 * part_rule.c - CPU restatement of the PRODUCT's table partition for table-sharded runs, so that the tests can check every
 * part on its own and not only the sum of the parts.
 *
 * TEST INFRASTRUCTURE ONLY (see clark_oracle.h).  The reference partitions its table by on-disk bucket range
 * (CuClarkDB.cu:566-574, filter :1272-1274); per-target counts are additive over ANY partition of the k-mer occurrences
 * (mergeKernel sums, CuClarkDB.cu:1385-1388), so the result of the whole run does not depend on how the table is cut.  The
 * product's super-k-mer layouts cut the RESIDENT table by slot range: a k-mer occurrence belongs to the part that holds the
 * slot of its sampled m-mer.  That rule is restated here from its description (cuclark_amd/csrc/mic_device.h: s_tlen, s_torder,
 * s_probe_read, sslot_of_x), independently of the HIP code:
 *   - mod-sampling: t = m - j w (w = k - m + 1) with the largest j that keeps t >= 7, t = m when there is none; the k-mer that
 *     reads at nucleotide tpos of its read part has W = k - t + 1 t-mers; t-mer i gets the key order27(u) << 5 | (tpos + i) & 31,
 *     where u = the t-mer as it reads (two-strand table) or the smaller of the t-mer and its reverse complement (one-strand);
 *   - the t-mer with the smallest key, at position i, samples the m-mer at position p = i mod w; its value (one-strand table:
 *     the smaller of the m-mer and its reverse complement) is hashed to a slot in [0, n_slots);
 *   - part p of n answers for the slots [n_slots p / n, n_slots (p + 1) / n).
 * Whether the k-mer is in the database, and with which label, is orc_db_find's business as for every other count.
 */
#include "clark_oracle.h"

#include <string.h>

static uint32_t umul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)); }

static int tmer_len(int k, int m) {
  const int w = k - m + 1;
  int t = m;
  while (t - w >= 7) t -= w;
  return t;
}

/* 32-bit order key of a t-mer value; the top 27 bits order the t-mers */
static uint32_t tmer_order_key(uint64_t tv) {
  uint32_t h = umul24((uint32_t)tv, 0x9E3779u) + 0x27D4EB2Fu;
  const uint32_t hi = (uint32_t)(tv >> 24);
  if (hi) h += hi * 0x85EBCA77u + (uint32_t)(tv >> 56) * 0xC2B2AE3Du;
  return h;
}

static uint32_t slot_of_minimizer(uint64_t x, uint32_t n_slots) {
  const uint32_t hi = (uint32_t)(x >> 32);
  const uint32_t h = (uint32_t)x * 0x85EBCA77u + umul24(hi ^ (hi >> 24), 0xC2B2AFu);
  return (uint32_t)(((uint64_t)h * (uint64_t)n_slots) >> 32);
}

uint32_t orc_part_slot_of_kmer(uint64_t kmer, uint32_t tpos, int k, int m, int both_strands, uint32_t n_slots) {
  const int w = k - m + 1, t = tmer_len(k, m), W = k - t + 1;
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  const uint64_t tmask = t >= 32 ? ~0ULL : (1ULL << (2 * t)) - 1;
  uint32_t best = 0xFFFFFFFFu;
  int bi = 0;
  for (int i = 0; i < W; ++i) {
    uint64_t tv = (kmer >> (2 * (k - t - i))) & tmask;
    if (!both_strands) {
      const uint64_t tr = orc_revcomp(tv, t);
      if (tr < tv) tv = tr;
    }
    const uint32_t key = (tmer_order_key(tv) & ~31u) | ((tpos + (uint32_t)i) & 31u);
    if (key < best) { best = key; bi = i; }
  }
  uint64_t x = (kmer >> (2 * (k - m - bi % w))) & mask;
  if (!both_strands) {
    const uint64_t xr = orc_revcomp(x, m);
    if (xr < x) x = xr;
  }
  return slot_of_minimizer(x, n_slots);
}

uint64_t orc_query_batch_slot_part(const orc_db* db, int k, int m, int both_strands, uint32_t n_slots, uint32_t part,
                                   uint32_t n_parts, const uint32_t* reads_pointer, const uint16_t* containers, size_t n_reads,
                                   uint32_t n_targets, uint32_t* counts) {
  const uint64_t cutoff = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t lo = (uint64_t)n_slots * part / n_parts, hi = (uint64_t)n_slots * (part + 1) / n_parts;
  uint64_t bad = 0;
  memset(counts, 0, n_reads * (size_t)n_targets * sizeof(uint32_t));
  for (size_t r = 0; r < n_reads; ++r) {
    uint32_t* row = counts + r * (size_t)n_targets;
    uint32_t p = reads_pointer[r];
    const uint32_t end = reads_pointer[r + 1];
    while (p < end) {
      const uint32_t plen = containers[p];
      if (plen == 0) break;
      const uint32_t first = p + 1;
      p = first + (plen - 1) / 8 + 1;
      uint64_t kmer = 0;
      for (uint32_t i = 0; i < plen; ++i) {
        const uint32_t nt = (containers[first + i / 8] >> (14 - 2 * (i % 8))) & 3u;
        kmer = ((kmer << 2) | nt) & cutoff;
        if (i + 1 < (uint32_t)k) continue;
        const uint32_t slot = orc_part_slot_of_kmer(kmer, i + 1 - (uint32_t)k, k, m, both_strands, n_slots);
        if (slot < lo || slot >= hi) continue;
        uint16_t label;
        if (orc_db_find(db, kmer, k, 0, db->htsize, &label)) {
          if (label < n_targets) ++row[label]; else ++bad;
        }
      }
    }
  }
  return bad;
}
