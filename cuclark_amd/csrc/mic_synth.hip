// mic_synth.hip — synthetic workloads generated directly in HBM.
//
// Procedural genomes: n_genomes sequences of equal length; nucleotide p of genome g is a pure function
// of (seed, g, p), so the database builder and the read sampler agree without storing 5.7e9 nucleotides.
//   DB:    every k-mer of every genome -> canonical -> (rem, quot) -> counting sort by bucket -> per-bucket
//          sort by (key, label) -> the on-disk arrays (.sz u8 / .ky / .lb images) in device memory.
//          This is the same transformation the reference's DB build applies on the CPU
//          (hashTable_hh.hh:221-269 insert, :203-216 sort, :590-663 write) minus the
//          discriminative-k-mer filter, which is meaningless for random genomes.
//   reads: sampled from the genomes (either strand, substitutions, N) or uniform random; emitted in the
//          packed container format of CuCLARK_hh.hh:1616-1716.
#include "mi_clark.h"
#include "mic_internal.h"

#include <stdio.h>
#include <vector>

namespace {

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// 32 nucleotides (first one in the top bits) of genome g starting at 32*b
__host__ __device__ inline uint64_t genome_word(uint64_t seed, uint64_t g, uint64_t b) {
  return mix64(mix64(seed ^ (g * 0x9E3779B97F4A7C15ULL)) + b * 0xD1B54A32D192ED03ULL);
}

// ---- genomes that are not uniformly random (mic_synth_spec.repeat_ppm / mosaic_ppm) ------------------------------------------
// Still pure functions of (seed, genome, position).  A genome is cut into segments of 2048 nucleotides:
//   * TANDEM REPEATS: a segment holds a tract with probability repeat_ppm x 2048 / 768 / 1e6 - 256 .. 1279 nucleotides somewhere in
//     its first 1792, a unit of 4 .. 50 nucleotides repeated.  Units of up to 6 nucleotides come from a pool of 64 per length SHARED
//     by all genomes (microsatellites: the same unit at hundreds of loci - what makes a minimizer crowded); longer units are the
//     locus's own.  In the DATABASE a k-mer that lies wholly inside a tract of a shared unit is absent (common to many targets:
//     HashTableStorage_hh.hh:241-292 removes it; what stays are the k-mers across the tract's ends, each locus's own), and inside a
//     tract of a private unit only its first occurrence is stored (the reference's table holds a k-mer once);
//   * MOSAIC segments (probability mosaic_ppm / 1e6): the sequence is the genome's, but the LABEL of the k-mer starting at position
//     p is a function of (segment, p / run), run 1, 2, 4 or 8 - a stretch where ownership changes every few k-mers, as between
//     close relatives after the removal of what they share: reads from there hit many targets with small equal counts (ties,
//     rows of more than 15 / 64 targets).
struct SynthMods { uint32_t tract_ppm, mosaic_ppm, n_targets; };
struct SegInfo { bool tract, shared, mosaic; uint32_t t_start, t_len, u, mrun; uint64_t useed, mseed; };

__host__ __device__ inline SegInfo seg_info(uint64_t seed, uint64_t g, uint64_t s, const SynthMods& md) {
  SegInfo si;
  const uint64_t h = mix64(mix64(seed ^ 0x7A11D3B5ull) + g * 0x9E3779B97F4A7C15ULL + s * 0xD6E8FEB86659FD93ULL);
  si.tract = md.tract_ppm && (uint32_t)(h % 1000000ull) < md.tract_ppm;
  si.t_start = (uint32_t)((h >> 20) & 511u);
  si.t_len = 256u + (uint32_t)((h >> 29) & 1023u);
  si.u = 4u + (uint32_t)((h >> 39) % 47u);      // (4 .. 50: units of 2 or 3 nucleotides are 16 / 64 in all - at the headline's scale the
                                                   // same end-of-tract k-mer would sit in hundreds of genomes and overflow a bucket of 255)
  si.shared = si.u <= 6u;
  si.useed = si.shared ? mix64(seed ^ (0xC0FFEEull + si.u * 0x100ull + ((h >> 48) & 63u))) : mix64(h ^ 0x51A7ull);
  const uint64_t h2 = mix64(h + 0x2545F4914F6CDD1Dull);
  si.mosaic = md.mosaic_ppm && (uint32_t)(h2 % 1000000ull) < md.mosaic_ppm;
  { const uint32_t v = (uint32_t)(h2 >> 20) & 31u; si.mrun = v == 0 ? 1u : v <= 10 ? 2u : v <= 20 ? 4u : 8u; }    // (1: one segment in 32)
  si.mseed = mix64(h2 ^ 0x77ull);
  return si;
}

__host__ __device__ inline uint32_t unit_nt(uint64_t useed, uint32_t j) {      // nucleotide j of a tract's unit
  return (uint32_t)(mix64(useed + (j >> 5)) >> (2 * (31 - (j & 31)))) & 3u;
}

__host__ __device__ inline uint32_t genome_nt_plain(uint64_t seed, uint64_t g, uint64_t p) {
  return (uint32_t)(genome_word(seed, g, p >> 5) >> (2 * (31 - (p & 31)))) & 3u;
}

__host__ __device__ inline uint32_t genome_nt(uint64_t seed, uint64_t g, uint64_t p, const SynthMods& md) {
  if (md.tract_ppm) {
    const SegInfo si = seg_info(seed, g, p >> 11, md);
    const uint32_t o = (uint32_t)(p & 2047u);
    if (si.tract && o - si.t_start < si.t_len) return unit_nt(si.useed, (o - si.t_start) % si.u);
  }
  return genome_nt_plain(seed, g, p);
}

// the k-mer starting at p; *stored = false when the database does not hold it (see above)
__device__ inline uint64_t genome_kmer(uint64_t seed, uint64_t g, uint64_t p, int k, const SynthMods& md, bool* stored) {
  *stored = true;
  if (md.tract_ppm) {
    const uint64_t s0 = p >> 11, s1 = (p + (uint64_t)k - 1) >> 11;
    const SegInfo a = seg_info(seed, g, s0, md);
    const bool t1 = s1 != s0 && seg_info(seed, g, s1, md).tract;
    if (a.tract || t1) {
      const uint32_t o = (uint32_t)(p & 2047u);
      if (a.tract && o >= a.t_start && o + (uint32_t)k <= a.t_start + a.t_len) {      // wholly inside the tract
        if (a.shared || o - a.t_start >= a.u) *stored = false;
      }
      uint64_t x = 0;
      for (int i = 0; i < k; ++i) x = (x << 2) | genome_nt(seed, g, p + (uint64_t)i, md);
      return x;
    }
  }
  uint64_t w0 = genome_word(seed, g, p >> 5), w1 = genome_word(seed, g, (p >> 5) + 1);
  int s = 2 * (int)(p & 31);
  uint64_t x = s ? ((w0 << s) | (w1 >> (64 - s))) : w0;
  return x >> (64 - 2 * k);
}

// label of the k-mer starting at p of genome g
__host__ __device__ inline uint32_t genome_label(uint64_t seed, uint64_t g, uint64_t p, const SynthMods& md) {
  if (md.mosaic_ppm) {
    const SegInfo si = seg_info(seed, g, p >> 11, md);
    if (si.mosaic) return (uint32_t)(mix64(si.mseed + (p & 2047u) / si.mrun) % md.n_targets);
  }
  return (uint32_t)(g % md.n_targets);
}

__host__ inline SynthMods make_mods(const mic_synth_spec* spec) {
  SynthMods md;
  const uint64_t q = (uint64_t)spec->repeat_ppm * 2048ull / 768ull;
  md.tract_ppm = (uint32_t)(q > 1000000ull ? 1000000ull : q);
  md.mosaic_ppm = spec->mosaic_ppm > 1000000u ? 1000000u : spec->mosaic_ppm;
  md.n_targets = spec->n_targets;
  return md;
}

__device__ inline uint64_t revcomp_bits(uint64_t x, int k) {
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
  return (~r) >> (64 - 2 * k);
}

struct SynthDev {
  uint64_t seed, genome_len, kmers_per_genome, n_kmers;
  uint32_t n_genomes, n_targets;
  int k;
  MicDiv div;
  uint32_t keep_ppm, run_len;      // fragmented database: see kept_position
  SynthMods md;
};

// A database of DISCRIMINATIVE k-mers holds stretches of a genome's k-mers and lacks others (CLARK removes every k-mer two targets
// share, HashTableStorage_hh.hh:241-292; what overlaps a shared region goes in runs).  keep_ppm != 0: the k-mer start positions of a
// genome fall into segments whose ends are drawn with probability 1 / run_len per position (geometric lengths, mean run_len), and a
// segment is kept with probability keep_ppm / 1e6; a function of (seed, genome, position), like the genomes themselves.
__host__ __device__ inline bool kept_position(uint64_t seed, uint64_t g, uint64_t p, uint32_t keep_ppm, uint32_t run_len) {
  if (!keep_ppm) return true;
  const uint64_t gs = mix64(seed ^ 0x5EED5EEDull) + g * 0xA24BAED4963EE407ULL;
  uint64_t q = p;
  for (uint32_t back = 0; back < 16 * run_len && q > 0; ++back, --q)
    if (mix64(gs + q * 0x9FB21C651E98DF25ULL) % run_len == 0) break;             // q starts a segment
  return mix64(gs ^ (q * 0xD6E8FEB86659FD93ULL + 1)) % 1000000ull < keep_ppm;
}

__device__ inline bool kmer_at(const SynthDev& sp, uint64_t idx, uint64_t& rem, uint64_t& quot, uint32_t& label) {
  uint64_t g = idx / sp.kmers_per_genome, p = idx - g * sp.kmers_per_genome;
  if (!kept_position(sp.seed, g, p, sp.keep_ppm, sp.run_len)) return false;
  bool stored;
  uint64_t km = genome_kmer(sp.seed, g, p, sp.k, sp.md, &stored);
  if (!stored) return false;
  uint64_t rc = revcomp_bits(km, sp.k);
  uint64_t c = km < rc ? km : rc;
  quot = mic_div(c, sp.div);
  rem = c - quot * sp.div.d;
  label = genome_label(sp.seed, g, p, sp.md);
  return true;
}

__global__ void count_kernel(SynthDev sp, uint32_t* __restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < sp.n_kmers; i += stride) {
    uint64_t rem, quot; uint32_t label;
    if (!kmer_at(sp, i, rem, quot, label)) continue;
    atomicAdd(&cnt[rem], 1u);
  }
}

#define STILE 1024
__global__ void __launch_bounds__(256) sizes_tile_kernel(const uint32_t* __restrict__ cnt, uint64_t n,
                                                         uint8_t* __restrict__ sizes,
                                                         unsigned long long* __restrict__ tile_sum,
                                                         uint32_t* __restrict__ max_cnt) {
  __shared__ uint32_t s[256];
  uint64_t base = (uint64_t)blockIdx.x * STILE + threadIdx.x * 4;
  uint32_t sum = 0, mx = 0;
  for (int j = 0; j < 4; ++j) {
    uint64_t i = base + j;
    if (i < n) { uint32_t c = cnt[i]; mx = c > mx ? c : mx; sizes[i] = (uint8_t)(c > 255 ? 255 : c); sum += c > 255 ? 255 : c; }
  }
  s[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0];
  if (mx > 255) atomicMax(max_cnt, mx);
}

__global__ void __launch_bounds__(256) offsets_kernel(const uint8_t* __restrict__ sizes, uint64_t n,
                                                      const unsigned long long* __restrict__ tile_base,
                                                      unsigned long long* __restrict__ offsets) {
  __shared__ uint32_t s[256];
  uint64_t base = (uint64_t)blockIdx.x * STILE + threadIdx.x * 4;
  uint32_t v[4], sum = 0;
  for (int j = 0; j < 4; ++j) { uint64_t i = base + j; v[j] = i < n ? sizes[i] : 0; sum += v[j]; }
  s[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    uint32_t a = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  unsigned long long run = tile_base[blockIdx.x] + s[threadIdx.x] - sum;
  for (int j = 0; j < 4; ++j) { uint64_t i = base + j; if (i < n) offsets[i] = run; run += v[j]; }
}

template <typename KEY>
__global__ void scatter_kernel(SynthDev sp, const unsigned long long* __restrict__ offsets, uint32_t* __restrict__ cursor,
                               KEY* __restrict__ keys, uint16_t* __restrict__ labels) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < sp.n_kmers; i += stride) {
    uint64_t rem, quot; uint32_t label;
    if (!kmer_at(sp, i, rem, quot, label)) continue;
    uint32_t pos = atomicAdd(&cursor[rem], 1u);
    if (pos < 255) { uint64_t d = offsets[rem] + pos; keys[d] = (KEY)quot; labels[d] = (uint16_t)label; }
  }
}

template <typename KEY>
__global__ void bucket_sort_kernel(const uint8_t* __restrict__ sizes, uint64_t n, const unsigned long long* __restrict__ offsets,
                                   KEY* __restrict__ keys, uint16_t* __restrict__ labels) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  uint32_t m = sizes[b];
  if (m < 2) return;
  KEY* kk = keys + offsets[b]; uint16_t* ll = labels + offsets[b];
  for (uint32_t i = 1; i < m; ++i) {  // insertion sort by (key, label)
    KEY kv = kk[i]; uint16_t lv = ll[i];
    uint32_t j = i;
    while (j > 0 && (kk[j - 1] > kv || (kk[j - 1] == kv && ll[j - 1] > lv))) { kk[j] = kk[j - 1]; ll[j] = ll[j - 1]; --j; }
    kk[j] = kv; ll[j] = lv;
  }
}

// ---- reads ------------------------------------------------------------------------------------------
struct ReadGen {
  uint64_t seed, read_seed, genome_len;
  uint32_t n_genomes, n_targets, read_len, pitch;
  int k;
  uint32_t random_thr, sub_thr, n_thr;  // thresholds on a 32-bit uniform
  uint32_t keep_ppm, run_len;           // fragmented database (kept_position): the expected hits count kept windows only
  SynthMods md;
};

// One read (or the two reads of a pair), nucleotide by nucleotide; shared by the packed and the text generators so that
// both describe the same reads.  mate < 0: a single read of L nt at p0 on strand `rev`.  mate 0 / 1: the two reads of a
// pair, drawn from both ends of the stretch [p0, p0 + 2L) read on strand `rev`: read 1 = its first L nt, read 2 = the
// first L nt of its other strand; random "genome-free" reads get an independent random second read.
struct ReadDraw {
  uint64_t h0, g, p0; bool rnd, rev; uint32_t span;
};

__device__ inline ReadDraw draw_read(const ReadGen& rg, size_t r, bool paired) {
  ReadDraw d;
  d.h0 = mix64(rg.read_seed * 0x9E3779B97F4A7C15ULL + r);
  d.rnd = (uint32_t)d.h0 < rg.random_thr;
  const uint64_t h1 = mix64(d.h0 + 1), h2 = mix64(d.h0 + 2);
  d.g = h1 % rg.n_genomes;
  d.span = paired ? 2 * rg.read_len : rg.read_len;
  d.p0 = h2 % (rg.genome_len - d.span + 1);
  d.rev = (d.h0 >> 40) & 1;
  return d;
}

// nucleotide i of the read (mate as above): 0..3, or -1 for an N; *clean = the nucleotide is the genome's own
__device__ inline int draw_nt(const ReadGen& rg, const ReadDraw& d, int mate, uint32_t i, bool* clean) {
  const uint64_t hb = mix64((mate > 0 ? d.h0 + 0x51ED27ULL : d.h0) ^ (0xABCD0000ULL + i));
  uint32_t nt; bool sub = false;
  if (d.rnd) nt = (uint32_t)(hb & 3);
  else {
    const uint64_t fwd = d.p0 + i, bwd = d.p0 + d.span - 1 - i;
    const bool rc = mate > 0 ? !d.rev : d.rev;
    nt = genome_nt(rg.seed, d.g, (mate > 0) != d.rev ? bwd : fwd, rg.md);
    if (rc) nt = 3u - nt;
    sub = (uint32_t)(hb >> 32) < rg.sub_thr;
    if (sub) nt = (nt + 1 + (uint32_t)((hb >> 8) % 3)) & 3u;
  }
  const bool isn = (uint32_t)(mix64(hb) >> 32) < rg.n_thr;
  *clean = !d.rnd && !sub && !isn;
  return isn ? -1 : (int)nt;
}

// Packed reads (CuCLARK_hh.hh:1616-1716).  paired: the object is read 1 + 'N' + read 2 (file.cc:205-268), 2L + 1 characters.
__global__ void reads_kernel(ReadGen rg, size_t n_reads, int paired, uint32_t* __restrict__ rp, uint16_t* __restrict__ cont,
                             uint32_t* __restrict__ truth) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_reads) return;
  if (r == n_reads) { rp[r] = (uint32_t)(n_reads * rg.pitch); return; }
  rp[r] = (uint32_t)(r * rg.pitch);
  uint16_t* out = cont + r * rg.pitch;
  const ReadDraw d = draw_read(rg, r, paired != 0);
  const uint32_t L = rg.read_len, total = paired ? 2 * L + 1 : L;
  uint32_t w = 0;            // write cursor in containers
  uint32_t hdr = 0;          // index of the current part's length slot
  uint32_t run = 0;          // nt in the current part
  uint32_t clean = 0;        // consecutive unmodified nt (for the expected-hit count)
  uint32_t expect = 0;
  uint16_t cur = 0; uint32_t ncur = 0;
  bool open = false;
  for (uint32_t i = 0; i <= total; ++i) {
    int code = -1;
    if (i < total) {
      bool cl = false;
      if (!paired) code = draw_nt(rg, d, -1, i, &cl);
      else if (i < L) code = draw_nt(rg, d, 0, i, &cl);
      else if (i > L) code = draw_nt(rg, d, 1, i - L - 1, &cl);
      clean = (cl && code >= 0) ? clean + 1 : 0;
      if (code >= 0 && clean >= (uint32_t)rg.k) {
        if (!rg.keep_ppm) ++expect;
        else if (!paired) {      // the k-mer that ends here starts at genome position p0 + i - k + 1 (forward) or its mirror
          const uint64_t a = d.p0 + (d.rev ? (uint64_t)(d.span - 1 - i) : (uint64_t)(i + 1 - (uint32_t)rg.k));
          if (kept_position(rg.seed, d.g, a, rg.keep_ppm, rg.run_len)) ++expect;
        }
      }
    }
    if (code >= 0) {
      if (!open) { hdr = w++; open = true; run = 0; cur = 0; ncur = 0; }
      cur = (uint16_t)((cur << 2) | (uint16_t)code); ++ncur; ++run;
      if (ncur == 8) { out[w++] = cur; cur = 0; ncur = 0; }
    } else if (open) {
      if (ncur) out[w++] = (uint16_t)(cur << (2 * (8 - ncur)));
      if (run >= (uint32_t)rg.k) out[hdr] = (uint16_t)run; else w = hdr;  // parts shorter than k are dropped
      open = false;
    }
  }
  if (w < rg.pitch) out[w] = 0;  // terminator
  if (truth) {
    // (no claim about a read that touches a segment with a tandem repeat or with mosaic labels: truth[1] = 0)
    if (!d.rnd && (rg.md.tract_ppm || rg.md.mosaic_ppm)) {
      for (uint64_t sg = d.p0 >> 11; sg <= (d.p0 + d.span - 1) >> 11; ++sg) {
        const SegInfo si = seg_info(rg.seed, d.g, sg, rg.md);
        if (si.tract || si.mosaic) expect = 0;
      }
    }
    truth[2 * r] = d.rnd ? 0 : (uint32_t)(d.g % rg.n_targets) + 1; truth[2 * r + 1] = expect;
  }
}

// The same reads as FASTQ / FASTA text, one record of fixed size per read: "@r<9 digits>\n" SEQ "\n+\n" QUAL "\n"
// (2 L + 16 bytes) or ">r<9 digits>\n" SEQ "\n" (L + 13); mate as in draw_nt.
__global__ void reads_text_kernel(ReadGen rg, size_t n_reads, int fasta, int mate, uint8_t* __restrict__ text) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint32_t L = rg.read_len;
  const size_t rec = fasta ? (size_t)L + 13 : 2 * (size_t)L + 16;
  uint8_t* out = text + r * rec;
  const ReadDraw d = draw_read(rg, r, mate >= 0);
  size_t w = 0;
  out[w++] = fasta ? '>' : '@'; out[w++] = 'r';
  { uint32_t v = (uint32_t)(r % 1000000000u); for (int i = 8; i >= 0; --i) { out[w + i] = (uint8_t)('0' + v % 10); v /= 10; } w += 9; }
  out[w++] = '\n';
  for (uint32_t i = 0; i < L; ++i) {
    bool cl;
    const int code = draw_nt(rg, d, mate, i, &cl);
    out[w++] = code < 0 ? 'N' : "TGCA"[code];
  }
  out[w++] = '\n';
  if (!fasta) {
    out[w++] = '+'; out[w++] = '\n';
    for (uint32_t i = 0; i < L; ++i) out[w++] = 'I';
    out[w++] = '\n';
  }
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "mic_synth: %s: %s\n", #x, hipGetErrorString(e_)); rc = MIC_E_HIP; goto done; } } while (0)

}  // namespace

extern "C" {

// containers per read used by mic_synth_reads_device: data + 2 per possible part + terminator + room for
// a short part that is written and then dropped
uint32_t mic_synth_read_pitch(uint32_t read_len, int k) {
  return (read_len + 7) / 8 + 2 * (read_len / (uint32_t)(k + 1) + 1) + 1 + 6;
}

int mic_synth_db_device(const mic_synth_spec* spec, uint8_t* d_sizes, void* d_keys, uint16_t* d_labels,
                        uint64_t capacity, uint64_t* n_elems, void* stream) {
  if (!spec || !d_sizes || !d_keys || !d_labels || !n_elems) return MIC_E_INVALID;
  if (spec->k < 2 || spec->k > 32 || spec->n_genomes == 0 || spec->n_targets == 0 || spec->htsize < 2) return MIC_E_INVALID;
  if (spec->key_bytes != 4 && spec->key_bytes != 8) return MIC_E_INVALID;
  {  // 4-byte keys must hold every quotient (main.cc:274-316 picks the width so that they do)
    unsigned __int128 max_c = spec->k == 32 ? ~(unsigned __int128)0 >> 64 : (((unsigned __int128)1 << (2 * spec->k)) - 1);
    if (spec->key_bytes == 4 && (uint64_t)(max_c / spec->htsize) >> 32) return MIC_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  int rc = MIC_OK;
  SynthDev sp;
  sp.seed = spec->seed; sp.genome_len = spec->genome_nt / spec->n_genomes;
  if (sp.genome_len < (uint64_t)spec->k + 1) return MIC_E_INVALID;
  sp.kmers_per_genome = sp.genome_len - spec->k + 1;
  sp.n_kmers = sp.kmers_per_genome * spec->n_genomes;
  sp.n_genomes = spec->n_genomes; sp.n_targets = spec->n_targets; sp.k = spec->k;
  sp.keep_ppm = spec->keep_ppm; sp.run_len = spec->run_len ? spec->run_len : 8;
  sp.md = make_mods(spec);
  sp.div = mic_make_div(spec->htsize);
  const uint64_t H = spec->htsize;
  const unsigned n_tiles = (unsigned)((H + STILE - 1) / STILE);
  uint32_t* d_cnt = nullptr; unsigned long long* d_tile = nullptr; unsigned long long* d_off = nullptr; uint32_t* d_max = nullptr;
  std::vector<unsigned long long> h_tile(n_tiles);
  unsigned long long total = 0; uint32_t h_max = 0;
  HIPCK(hipMalloc(&d_cnt, H * 4));
  HIPCK(hipMalloc(&d_tile, (size_t)n_tiles * 8));
  HIPCK(hipMalloc(&d_off, (H + 1) * 8));
  HIPCK(hipMalloc(&d_max, 4));
  HIPCK(hipMemsetAsync(d_cnt, 0, H * 4, s));
  HIPCK(hipMemsetAsync(d_max, 0, 4, s));
  count_kernel<<<256 * 16, 256, 0, s>>>(sp, d_cnt);
  HIPCK(hipGetLastError());
  sizes_tile_kernel<<<n_tiles, 256, 0, s>>>(d_cnt, H, d_sizes, d_tile, d_max);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(h_tile.data(), d_tile, (size_t)n_tiles * 8, hipMemcpyDeviceToHost, s));
  HIPCK(hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, s));
  HIPCK(hipStreamSynchronize(s));
  if (h_max > 255) { fprintf(stderr, "mic_synth: a bucket would hold %u > 255 elements; enlarge htsize\n", h_max); rc = MIC_E_INVALID; goto done; }
  for (unsigned t = 0; t < n_tiles; ++t) { unsigned long long v = h_tile[t]; h_tile[t] = total; total += v; }
  *n_elems = total;
  if (total > capacity) { rc = MIC_E_NOMEM; goto done; }
  HIPCK(hipMemcpyAsync(d_tile, h_tile.data(), (size_t)n_tiles * 8, hipMemcpyHostToDevice, s));
  offsets_kernel<<<n_tiles, 256, 0, s>>>(d_sizes, H, d_tile, d_off);
  HIPCK(hipGetLastError());
  HIPCK(hipMemsetAsync(d_cnt, 0, H * 4, s));
  if (spec->key_bytes == 4) scatter_kernel<uint32_t><<<256 * 16, 256, 0, s>>>(sp, d_off, d_cnt, (uint32_t*)d_keys, d_labels);
  else scatter_kernel<uint64_t><<<256 * 16, 256, 0, s>>>(sp, d_off, d_cnt, (uint64_t*)d_keys, d_labels);
  HIPCK(hipGetLastError());
  {
    unsigned blocks = (unsigned)((H + 255) / 256);
    if (spec->key_bytes == 4) bucket_sort_kernel<uint32_t><<<blocks, 256, 0, s>>>(d_sizes, H, d_off, (uint32_t*)d_keys, d_labels);
    else bucket_sort_kernel<uint64_t><<<blocks, 256, 0, s>>>(d_sizes, H, d_off, (uint64_t*)d_keys, d_labels);
    HIPCK(hipGetLastError());
  }
  HIPCK(hipStreamSynchronize(s));
done:
  // (an error return: copies queued on s may still name this frame's host variables - they must have landed before it goes)
  if (rc) hipStreamSynchronize(s);
  if (d_cnt) hipFree(d_cnt);
  if (d_tile) hipFree(d_tile);
  if (d_off) hipFree(d_off);
  if (d_max) hipFree(d_max);
  return rc;
}

int mic_synth_reads_device(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len,
                           double random_frac, double sub_rate, double n_rate, uint32_t* d_rp, uint16_t* d_cont,
                           size_t containers_cap, uint32_t* d_truth, void* stream) {
  return mic_synth_reads_device2(spec, read_seed, n_reads, read_len, 0, random_frac, sub_rate, n_rate, d_rp, d_cont, containers_cap, d_truth, stream);
}

int mic_synth_reads_device2(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len, int paired,
                            double random_frac, double sub_rate, double n_rate, uint32_t* d_rp, uint16_t* d_cont,
                            size_t containers_cap, uint32_t* d_truth, void* stream) {
  if (!spec || !d_rp || !d_cont || read_len == 0) return MIC_E_INVALID;
  ReadGen rg;
  rg.seed = spec->seed; rg.read_seed = read_seed;
  rg.genome_len = spec->genome_nt / spec->n_genomes;
  if (rg.genome_len < (paired ? 2ull : 1ull) * read_len) return MIC_E_INVALID;
  rg.n_genomes = spec->n_genomes; rg.n_targets = spec->n_targets; rg.read_len = read_len; rg.k = spec->k;
  rg.keep_ppm = spec->keep_ppm; rg.run_len = spec->run_len ? spec->run_len : 8;
  rg.md = make_mods(spec);
  rg.pitch = mic_synth_read_pitch(paired ? 2 * read_len + 1 : read_len, spec->k);
  if ((uint64_t)n_reads * rg.pitch > containers_cap || (uint64_t)n_reads * rg.pitch > 0xFFFFFFF0ull) return MIC_E_NOMEM;
  auto thr = [](double p) { double v = p * 4294967296.0; return v <= 0 ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v); };
  rg.random_thr = thr(random_frac); rg.sub_thr = thr(sub_rate); rg.n_thr = thr(n_rate);
  reads_kernel<<<(unsigned)((n_reads + 1 + 255) / 256), 256, 0, (hipStream_t)stream>>>(rg, n_reads, paired, d_rp, d_cont, d_truth);
  return hipGetLastError() == hipSuccess ? MIC_OK : MIC_E_HIP;
}

size_t mic_synth_text_record_bytes(uint32_t read_len, int fasta) { return fasta ? (size_t)read_len + 13 : 2 * (size_t)read_len + 16; }

int mic_synth_reads_text_device(const mic_synth_spec* spec, uint64_t read_seed, size_t n_reads, uint32_t read_len,
                                double random_frac, double sub_rate, double n_rate, int fasta, int mate, uint8_t* d_text,
                                size_t text_cap, void* stream) {
  if (!spec || !d_text || read_len == 0) return MIC_E_INVALID;
  ReadGen rg;
  rg.seed = spec->seed; rg.read_seed = read_seed;
  rg.genome_len = spec->genome_nt / spec->n_genomes;
  if (rg.genome_len < 2ull * read_len) return MIC_E_INVALID;
  rg.n_genomes = spec->n_genomes; rg.n_targets = spec->n_targets; rg.read_len = read_len; rg.k = spec->k;
  rg.keep_ppm = spec->keep_ppm; rg.run_len = spec->run_len ? spec->run_len : 8;
  rg.md = make_mods(spec);
  rg.pitch = 0;
  if ((uint64_t)n_reads * mic_synth_text_record_bytes(read_len, fasta) > text_cap) return MIC_E_NOMEM;
  auto thr = [](double p) { double v = p * 4294967296.0; return v <= 0 ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v); };
  rg.random_thr = thr(random_frac); rg.sub_thr = thr(sub_rate); rg.n_thr = thr(n_rate);
  reads_text_kernel<<<(unsigned)((n_reads + 255) / 256), 256, 0, (hipStream_t)stream>>>(rg, n_reads, fasta, mate, d_text);
  return hipGetLastError() == hipSuccess ? MIC_OK : MIC_E_HIP;
}

}  // extern "C"
