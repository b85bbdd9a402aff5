// mic_host.cpp — host-side pieces of the path: key-width rule, read indexer, read packer, CSV lines,
// and the division magic.  Pure C++ (no device code); exported through the C ABI of include/mi_clark.h
// and used by the classifier (classifier.cpp) and the CLI.
#include "mi_clark.h"
#include "mic_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <memory>
#include <mutex>
#include <vector>

// ---- division magic ---------------------------------------------------------------------------------
MicDiv mic_make_div(uint64_t d) {
  MicDiv r;
  r.d = d; r.magic = 0; r.shift = 0; r.add = 0;
  if (d == 0) return r;
  const uint32_t fl = 63 - (uint32_t)__builtin_clzll(d);
  if ((d & (d - 1)) == 0) { r.shift = fl; return r; }  // power of two
  const unsigned __int128 num = (unsigned __int128)1 << (64 + fl);
  uint64_t m = (uint64_t)(num / d);
  uint64_t rem = (uint64_t)(num % d);
  const uint64_t e = d - rem;
  if (e < ((uint64_t)1 << fl)) {
    r.shift = fl;
  } else {
    m += m;
    const uint64_t twice = rem + rem;
    if (twice >= d || twice < rem) m += 1;
    r.shift = fl; r.add = 1;
  }
  r.magic = m + 1;
  return r;
}

// run f(0..n-1) on n std::threads (the library does not use OpenMP: its callers may bring their own runtime)
template <typename F>
static void run_threads(int n, F&& f) {
  std::vector<std::thread> th;
  th.reserve(n > 1 ? n - 1 : 0);
  for (int t = 1; t < n; ++t) th.emplace_back([&f, t] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

extern "C" {

// main.cc:274-316.  t_b is computed in double exactly as the reference does.
int mic_key_bytes_rule(uint64_t htsize, int k) {
  if (htsize < 2 || k < 2 || k > 32) return MIC_E_INVALID;
  const size_t t_b = (size_t)(log((double)htsize) / log(4.0));
  if ((size_t)k <= t_b + 8) return 2;
  if ((size_t)k <= t_b + 16) return 4;
  return 8;
}

// ---- read indexer (CuCLARK_hh.hh:1339-1534 for a single batch) -----------------------------------------
static inline bool name_sep(uint8_t c) { return c == ' ' || c == '\t' || c == '\n'; }  // CuCLARK_hh.hh:300

struct Rec { uint64_t ns, ne, ss, se, len; };

// Parse the record whose marker ('>' or '@') is at map[i-1]; returns the position of the next record's marker
// (FASTA) / of the byte after the quality line (FASTQ), or a value >= nb at the end of the file.
static inline size_t find_nl(const uint8_t* map, size_t nb, size_t i) {   // first '\n' at or after i, or nb
  if (i >= nb) return nb;
  const void* q = memchr(map + i, '\n', nb - i);
  return q ? (size_t)((const uint8_t*)q - map) : nb;
}

static inline size_t parse_record(const uint8_t* map, size_t nb, bool fasta, size_t i, Rec& r) {
  r.ns = i;  // name: from the byte after the marker up to the first separator found strictly after it
  while (i + 1 < nb && !name_sep(map[i + 1])) ++i;
  ++i;
  if (i > nb) i = nb;
  r.ne = i;
  i = find_nl(map, nb, i);  // rest of the header line
  if (i < nb) ++i;
  size_t s = i, e = i;
  if (fasta) {
    size_t lines = 0;
    while (i < nb && map[i] != '>') {  // sequence lines until a line that starts the next record
      i = find_nl(map, nb, i);
      ++lines;
      e = i++;
    }
    r.len = (e - s + 1) - lines;  // non-newline bytes (CuCLARK_hh.hh:1385-1389)
  } else {
    i = find_nl(map, nb, i);
    e = i;
    ++i;
    r.len = e - s;
    i = find_nl(map, nb, i);  // '+' line
    if (i < nb) ++i;
    i = find_nl(map, nb, i);  // quality line
    if (i < nb) ++i;
  }
  r.ss = s; r.se = e;
  return i;
}

static inline void store_rec(const Rec& r, size_t n, size_t cap, uint64_t* name_s, uint64_t* name_e, uint64_t* seq_s,
                             uint64_t* seq_e, uint64_t* length) {
  if (n < cap) { name_s[n] = r.ns; name_e[n] = r.ne; seq_s[n] = r.ss; seq_e[n] = r.se; length[n] = r.len; }
}

long mic_index_reads(const uint8_t* map, size_t nb, size_t cap, uint64_t* name_s, uint64_t* name_e, uint64_t* seq_s,
                     uint64_t* seq_e, uint64_t* length) {
  if (!map || nb == 0 || (map[0] != '>' && map[0] != '@')) return MIC_E_INVALID;
  const bool fasta = map[0] == '>';
  size_t n = 0, i = 1;
  for (;;) {
    Rec r;
    i = parse_record(map, nb, fasta, i, r);
    store_rec(r, n, cap, name_s, name_e, seq_s, seq_e, length);
    ++n;
    if (fasta) { if (i >= nb) break; ++i; }   // skip '>'
    else { if (++i >= nb) break; }             // skip '@'
  }
  return (long)n;
}

// First record marker at or after `from` (from > 0).  FASTA: a '>' at the start of a line.  FASTQ: the reference's
// batch-start heuristic (CuCLARK_hh.hh:1409-1471): a line that starts with '@', whose next line holds only letters
// and whose line after that starts with '+'.  Returns nb if there is none.
static size_t find_record_start(const uint8_t* map, size_t nb, bool fasta, size_t from) {
  size_t p = from;
  while (p < nb && map[p - 1] != '\n') ++p;  // first line start >= from
  while (p < nb) {
    if (fasta) {
      if (map[p] == '>') return p;
    } else if (map[p] == '@') {
      size_t q = p;
      while (q < nb && map[q] != '\n') ++q;
      size_t s = q + 1, e = s;
      bool letters = s < nb;
      while (e < nb && map[e] != '\n') { uint8_t c = map[e]; letters = letters && ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')); ++e; }
      if (letters && e > s && e + 1 < nb && map[e + 1] == '+') return p;
    }
    while (p < nb && map[p] != '\n') ++p;
    ++p;
  }
  return nb;
}

// exported form: position of the first record marker at or after `from`, or nb
size_t mic_find_record_start(const uint8_t* map, size_t nb, size_t from) {
  if (!map || nb == 0 || (map[0] != '>' && map[0] != '@')) return nb;
  if (from == 0) return 0;
  return find_record_start(map, nb, map[0] == '>', from);
}

// the same on a window of the input whose format is known (win[from - 1] must be inside the window): lets a caller that
// reads a file in pieces cut it at record starts without mapping it.  A FASTQ candidate whose next two lines do not end
// inside the window is not accepted (returns nb): read a larger window.
size_t mic_find_record_start_in(const uint8_t* win, size_t nb, int fasta, size_t from) {
  if (!win || from == 0 || from > nb) return nb;
  return find_record_start(win, nb, fasta != 0, from);
}

// Parallel form of mic_index_reads: the file is cut into `n_threads` byte ranges, each range is indexed from the first
// record that starts in it.  Same output as the serial function for well-formed FASTA/FASTQ.
long mic_index_reads_parallel(const uint8_t* map, size_t nb, int n_threads, size_t cap, uint64_t* name_s, uint64_t* name_e,
                              uint64_t* seq_s, uint64_t* seq_e, uint64_t* length) {
  if (!map || nb == 0 || (map[0] != '>' && map[0] != '@')) return MIC_E_INVALID;
  if (n_threads < 1) n_threads = 1;
  if ((size_t)n_threads > nb / 65536 + 1) n_threads = (int)(nb / 65536 + 1);
  if (n_threads == 1) return mic_index_reads(map, nb, cap, name_s, name_e, seq_s, seq_e, length);
  const bool fasta = map[0] == '>';
  std::vector<size_t> start(n_threads + 1, nb);
  start[0] = 0;
  // Per-range record buffers are kept between calls (one pool, taken by whoever gets the lock): a fresh 10 MB vector
  // per thread per call is returned to the kernel on free and faulted in again, which cost more than the parsing.
  static std::mutex pool_mu;
  static std::vector<std::vector<Rec>> pool;
  std::vector<std::vector<Rec>> local;
  std::unique_lock<std::mutex> pool_lock(pool_mu, std::try_to_lock);
  std::vector<std::vector<Rec>>& recs = pool_lock.owns_lock() ? pool : local;
  if (recs.size() < (size_t)n_threads) recs.resize(n_threads);
  for (auto& v : recs) v.clear();
  run_threads(n_threads, [&](int t) { if (t > 0) start[t] = find_record_start(map, nb, fasta, (size_t)t * (nb / n_threads)); });
  run_threads(n_threads, [&](int t) {
    size_t lo = start[t], hi = nb;
    for (int u = t + 1; u <= n_threads; ++u) if (start[u] > lo) { hi = start[u]; break; }
    bool dup = false;                                  // empty range (a record longer than the range)
    for (int u = 0; u < t; ++u) dup = dup || start[u] == lo;
    if (dup || lo >= nb) return;
    std::vector<Rec>& out = recs[t];
    out.reserve((hi - lo) / 64 + 16);
    size_t i = lo + 1;
    for (;;) {
      Rec r;
      i = parse_record(map, nb, fasta, i, r);
      out.push_back(r);
      if (fasta) { if (i >= nb || i >= hi) break; ++i; }
      else { if (i + 1 >= nb || i >= hi) break; ++i; }
    }
  });
  size_t n = 0;
  std::vector<size_t> base(n_threads + 1, 0);
  for (int t = 0; t < n_threads; ++t) { base[t] = n; n += recs[t].size(); }
  if (n <= cap)
    run_threads(n_threads, [&](int t) {
      for (size_t j = 0; j < recs[t].size(); ++j) store_rec(recs[t][j], base[t] + j, cap, name_s, name_e, seq_s, seq_e, length);
    });
  return (long)n;
}

// ---- read packer (CuCLARK_hh.hh:1616-1716) --------------------------------------------------------------
static inline int nt_code(uint8_t c) {
  switch (c) {  // m_rTable, CuCLARK_hh.hh:291-295
    case 'A': case 'a': return 3;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 1;
    case 'T': case 't': case 'U': case 'u': return 0;
    default: return -1;
  }
}

}  // extern "C"
namespace {
struct CodeTable {
  enum { kNewline = 0x40, kOther = 0x80 };
  uint8_t code[256];   // 0..3 = nucleotide code (m_rTable), kNewline, kOther
};
const CodeTable& code_table() {
  static const CodeTable t = [] {
    CodeTable x;
    for (int c = 0; c < 256; ++c) { const int v = nt_code((uint8_t)c); x.code[c] = v >= 0 ? (uint8_t)v : (uint8_t)CodeTable::kOther; }
    x.code[(unsigned char)'\n'] = CodeTable::kNewline;
    return x;
  }();
  return t;
}
}  // namespace
extern "C" {

size_t mic_pack_bound(const uint64_t* seq_s, const uint64_t* seq_e, size_t n_reads, int k) {
  // a part of L nt takes 1 + ceil(L/8) containers; parts need >= k nt and are separated by >= 1 byte;
  // sub-part splitting adds one header and k-1 nt per MIC_MAX_PART nt.
  size_t total = 0;
  for (size_t r = 0; r < n_reads; ++r) {
    size_t nb = (size_t)(seq_e[r] - seq_s[r]);
    size_t parts = nb / (size_t)(k + 1) + 1 + nb / (MIC_MAX_PART - 64);
    total += nb / 8 + 2 * parts + (nb / (MIC_MAX_PART - 64)) * 8 + 2;
  }
  return total + 16;
}

namespace {
struct Sink {
  uint16_t* out; size_t cap; size_t n; bool overflow;
  // writes past the capacity are dropped; whether the output fits is decided at the end (a run shorter than k is written
  // and then rolled back: it may pass the capacity without the final output doing so)
  inline void put(uint16_t v) { if (n < cap) out[n] = v; ++n; }
};

// one maximal ACGTU run -> one or more parts [len][containers...]
void emit_run(Sink& s, const uint8_t* codes, size_t len, int k, std::vector<uint64_t>* run_off, std::vector<uint64_t>* run_len) {
  size_t start = 0;
  for (;;) {
    size_t plen = len - start;
    if (plen > MIC_MAX_PART) plen = MIC_MAX_PART;
    if (run_off) { run_off->push_back(start); run_len->push_back(len); }
    s.put((uint16_t)plen);
    size_t i = 0;
    for (; i + 8 <= plen; i += 8) {
      const uint8_t* c = codes + start + i;
      s.put((uint16_t)((c[0] << 14) | (c[1] << 12) | (c[2] << 10) | (c[3] << 8) | (c[4] << 6) | (c[5] << 4) | (c[6] << 2) | c[7]));
    }
    if (i < plen) {
      uint16_t v = 0; unsigned cnt = 0;
      for (; i < plen; ++i, ++cnt) v = (uint16_t)((v << 2) | codes[start + i]);
      s.put((uint16_t)(v << (2 * (8 - cnt))));
    }
    if (start + plen >= len) break;
    start += plen - (size_t)(k - 1);
  }
}
}  // namespace

size_t mic_pack_reads(const uint8_t* map, const uint64_t* seq_s, const uint64_t* seq_e, const uint64_t* length,
                      size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap) {
  return mic_pack_reads_runs(map, seq_s, seq_e, length, n_reads, k, reads_pointer, containers, cap, nullptr, nullptr);
}

}  // extern "C"

// internal: the packer, optionally recording for every part the offset of its first nucleotide inside its maximal
// ACGTU run and the length of that run (the light database builder needs the run structure)
size_t mic_pack_reads_runs(const uint8_t* map, const uint64_t* seq_s, const uint64_t* seq_e, const uint64_t* length,
                           size_t n_reads, int k, uint32_t* reads_pointer, uint16_t* containers, size_t cap,
                           std::vector<uint64_t>* run_off, std::vector<uint64_t>* run_len) {
  Sink s{containers, cap, 0, false};
  std::vector<uint8_t> codes;
  const CodeTable& tab = code_table();
  for (size_t r = 0; r < n_reads; ++r) {
    reads_pointer[r] = (uint32_t)s.n;
    if (length[r] < (uint64_t)k) continue;  // reads without a k-mer store nothing (CuCLARK_hh.hh:1633)
    const uint8_t* p = map + seq_s[r];
    const size_t nb = (size_t)(seq_e[r] - seq_s[r]);
    if (nb <= MIC_MAX_PART) {
      // Streaming path (no part can need splitting): containers are written as the bytes are read, eight nucleotides
      // per step when the next eight bytes are all nucleotides; a run that ends below k nucleotides is rolled back.
      size_t hdr = s.n, run = 0;   // header position of the open part, its nucleotides so far
      uint32_t acc = 0;            // the (run & 7) nucleotides not yet stored, 2 bits each
      s.put(0);
      auto close_part = [&]() {
        if (run >= (size_t)k) {
          if (run & 7) s.put((uint16_t)(acc << (2 * (8 - (run & 7)))));
          if (hdr < s.cap) s.out[hdr] = (uint16_t)run;
          if (run_off) { run_off->push_back(0); run_len->push_back(run); }
          hdr = s.n;
          s.put(0);
        } else {
          s.n = hdr + 1;            // drop what was written for this run, keep the header slot
        }
        run = 0; acc = 0;
      };
      size_t i = 0;
      while (i < nb) {
        if (i + 8 <= nb) {
          const uint8_t c0 = tab.code[p[i]], c1 = tab.code[p[i + 1]], c2 = tab.code[p[i + 2]], c3 = tab.code[p[i + 3]],
                        c4 = tab.code[p[i + 4]], c5 = tab.code[p[i + 5]], c6 = tab.code[p[i + 6]], c7 = tab.code[p[i + 7]];
          if (!((c0 | c1 | c2 | c3 | c4 | c5 | c6 | c7) & 0xFC)) {
            const uint32_t v = (uint32_t)((c0 << 14) | (c1 << 12) | (c2 << 10) | (c3 << 8) | (c4 << 6) | (c5 << 4) | (c6 << 2) | c7);
            const unsigned have = (unsigned)(run & 7);               // nucleotides waiting in acc
            s.put((uint16_t)((acc << (2 * (8 - have))) | (v >> (2 * have))));
            acc = v & ((1u << (2 * have)) - 1);
            run += 8; i += 8;
            continue;
          }
        }
        const uint8_t c = tab.code[p[i]];
        ++i;
        if (c < 4) {
          acc = (acc << 2) | c;
          if ((++run & 7) == 0) { s.put((uint16_t)acc); acc = 0; }
          continue;
        }
        if (c == CodeTable::kNewline) continue;  // line breaks are transparent (CuCLARK_hh.hh:1674-1678)
        close_part();                            // any other byte ends the part
      }
      close_part();
      s.n -= 1;                                  // the header slot opened for a part that never came
      continue;
    }
    if (codes.size() < nb) codes.resize(nb * 2 + 64);
    size_t run = 0;
    for (size_t i = 0; i < nb; ++i) {
      const int c = nt_code(p[i]);
      if (c >= 0) { codes[run++] = (uint8_t)c; continue; }
      if (p[i] == '\n') continue;  // line breaks are transparent (CuCLARK_hh.hh:1674-1678)
      if (run >= (size_t)k) emit_run(s, codes.data(), run, k, run_off, run_len);  // any other byte ends the part
      run = 0;
    }
    if (run >= (size_t)k) emit_run(s, codes.data(), run, k, run_off, run_len);
  }
  reads_pointer[n_reads] = (uint32_t)s.n;
  // rolled-back writes beyond cap were dropped, kept ones are all below s.n: the output is whole iff s.n <= cap ... and
  // nothing that was kept had been dropped, which holds because a kept part is never rolled back below its own start
  return (s.n > s.cap || s.overflow) ? (size_t)-1 : s.n;
}

extern "C" {

// ---- CSV (CuCLARK_hh.hh:1951-2139) ---------------------------------------------------------------------
}  // extern "C"
namespace {
inline int put_u32(char* p, uint32_t v) {
  char t[10]; int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int i = 0; i < n; ++i) p[i] = t[n - 1 - i];
  return n;
}
// "%g" of value = num / den for integers 0 <= num < 256, 1 <= den < 256, cached; anything else goes through snprintf
struct RatioCache { uint8_t len[256][256]; char text[256][256][14]; };
inline int put_ratio(char* p, uint32_t num, double den, double value, char* tmp) {
  thread_local std::unique_ptr<RatioCache> cache;
  const int di = (den >= 1.0 && den < 256.0 && den == (double)(int)den) ? (int)den : 0;
  if (di && num < 256) {
    if (!cache) { cache.reset(new RatioCache); memset(cache->len, 0, sizeof(cache->len)); }
    uint8_t& l = cache->len[num][di];
    if (!l) {
      int w = snprintf(tmp, 64, "%g", value);
      if (w > 0 && w < 14) { memcpy(cache->text[num][di], tmp, (size_t)w); l = (uint8_t)w; }
      else { memcpy(p, tmp, (size_t)w); return w; }
    }
    memcpy(p, cache->text[num][di], l);
    return l;
  }
  const int w = snprintf(tmp, 64, "%g", value);
  memcpy(p, tmp, (size_t)w);
  return w;
}
}  // namespace
extern "C" {
int mic_csv_header(char* buf, size_t cap, int extended, const char* const* target_names, uint32_t n_targets) {
  size_t n = 0;
  auto app = [&](const char* s) { size_t l = strlen(s); if (n + l < cap) memcpy(buf + n, s, l); n += l; };
  app("Object_ID");
  if (extended)
    for (uint32_t t = 0; t < n_targets; ++t) { app(","); app(target_names[t]); }
  app(",Length,Gamma,1st_assignment,score1,2nd_assignment,score2,confidence\n");
  if (n >= cap) return -1;
  buf[n] = 0;
  return (int)n;
}

int mic_csv_line(char* buf, size_t cap, const uint8_t* name, size_t name_len, uint64_t length, int paired, int k,
                 const uint32_t* res, const char* const* target_names, uint32_t n_targets, int extended,
                 const uint32_t* row, const uint32_t* dense) {
  size_t n = 0;
  if (name_len >= 40) name_len = 39;  // OBJECTNAMEMAX (parameters.hh:46, CuCLARK_hh.hh:2114-2117)
  if (n + name_len < cap) memcpy(buf, name, name_len);
  n += name_len;
  if (extended) {  // dense per-target counts rebuilt from the sparse row (CuCLARK_hh.hh:2014-2031)
    uint32_t next = 0, nrow = row ? row[0] : 0, e = 0;
    for (uint32_t t = 0; t < n_targets; ++t) {
      uint32_t c = 0;
      if (dense) c = dense[t];
      else {
        while (e < nrow && (row[1 + e] & 0xFFFF) < t) ++e;
        if (e < nrow && (row[1 + e] & 0xFFFF) == t) c = row[1 + e] >> 16;
      }
      (void)next;
      int w = snprintf(buf + (n < cap ? n : cap), n < cap ? cap - n : 0, ",%u", c);
      n += (size_t)w;
    }
  }
  const uint32_t norm = paired ? (uint32_t)length - 1u : (uint32_t)length;  // ITYPE arithmetic, NBN=1
  const uint32_t total = res[0], ib = res[1], best = res[2], is = res[3], sbest = res[4];
  const double gamma = (double)total / (((double)norm - (double)k) + 1.0);
  double delta = (double)(best + sbest);
  delta = (delta < 0.001) ? 0 : ((double)best) / delta;
  const char* n1 = (ib == 0 || ib > n_targets) ? "NA" : target_names[ib - 1];
  const char* n2 = (is == 0 || is > n_targets) ? "NA" : target_names[is - 1];
  // The two %g fields are ratios of small integers: their printf images are cached per thread (filled by snprintf
  // itself, so the text is identical); the rest is assembled by hand.
  const double den_g = ((double)norm - (double)k) + 1.0;
  const uint32_t den_d = best + sbest;
  char tmp[64];
  if (n + 128 + strlen(n1) + strlen(n2) >= cap) return -1;
  char* p = buf + n;
  *p++ = ','; p += put_u32(p, norm);
  *p++ = ','; p += put_ratio(p, total, den_g, gamma, tmp);
  *p++ = ','; { size_t l = strlen(n1); memcpy(p, n1, l); p += l; }
  *p++ = ','; p += put_u32(p, best);
  *p++ = ','; { size_t l = strlen(n2); memcpy(p, n2, l); p += l; }
  *p++ = ','; p += put_u32(p, sbest);
  *p++ = ','; p += put_ratio(p, den_d ? best : 0u, den_d ? (double)den_d : 1.0, delta, tmp);   // no hits: delta = 0
  *p++ = '\n';
  *p = 0;
  n = (size_t)(p - buf);
  return n < cap ? (int)n : -1;
}

}  // extern "C"
