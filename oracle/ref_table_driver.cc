// ref_table_driver.cc — thin command-line driver around the REFERENCE's own CPU hash table.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the reference's
// headers where they lie (/root/reference/src, passed with -I by oracle/Makefile) and calls
//   EHashtable<HKMERr,lElement>::addElement / SortAllHashTable / RemoveCommon / Write   (DB build,
//       the same call sequence CuCLARK_hh.hh:896-1112 uses without --tsk)
//   EHashtable<HKMERr,lElement>::Read / queryElement(uint64_t, ILBL&)                    (CPU probe,
//       HashTableStorage_hh.hh:128-131 -> hashTable_hh.hh:475-513)
// so that the oracle restatement (clark_oracle.c) and the HIP path can be pinned to what the
// reference itself writes to disk and answers per k-mer.  HTSIZE is a compile-time macro of the
// reference (parameters.hh / parameters_light_hh); the Makefile builds one binary per variant:
//   oracle/_ref/ref_table_light  (HTSIZE 57777779,   -include parameters_light_hh)
//   oracle/_ref/ref_table_full   (HTSIZE 1610612741, needs ~26 GB RAM: fixture generation only)
//
// Usage:
//   ref_table_X info
//   ref_table_X build <k> <key_bytes> <out_prefix> <targets.tsv> [min_count] [light_gap]
//         targets.tsv lines: <fasta path>\t<label>     (labels in first-appearance order)
//   ref_table_X query <k> <key_bytes> <prefix> <kmers.txt> [sampling] [mmap]
//         kmers.txt: one forward-strand k-mer value (decimal u64) per line
//         prints "<kmer> <found 0/1> <label>" per line
//   ref_table_X merge <file1.fq> <file2.fq> <out.fa>
//         the reference's mergePairedFiles (file.cc:205-268) on the two files; its exits (perror + exit(1)) are the process's
//   ref_table_X codec <k> <kmers_ascii.txt>
//         per line of k nucleotides: "<getKmers value> <getReverse of it>" (kmersConversion.cc:39-68: the codec and the
//         reverse complement every other piece of the reference builds on)
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <map>
#include <fstream>
#include <iostream>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "HashTableStorage_hh.hh"
#include "file.hh"

// defined in the reference's kmersConversion.cc (not declared in its header)
void getKmers(const std::string& c, uint64_t& _km_f, uint8_t k);
void getReverse(uint64_t& _km_r, uint8_t k);

static int code_of(unsigned char c) {
  switch (c) {
    case 'A': case 'a': return 3;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 1;
    case 'T': case 't': case 'U': case 'u': return 0;
    default: return -1;
  }
}

template <typename KEY>
static int do_build(int k, const char* prefix, const char* targets_tsv, size_t min_count, size_t gap) {
  std::vector<std::pair<std::string, std::string> > targets;
  std::vector<std::string> labels, labels_c;
  {
    std::ifstream in(targets_tsv);
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      size_t tab = line.find('\t');
      if (tab == std::string::npos) { fprintf(stderr, "bad targets line: %s\n", line.c_str()); return 2; }
      std::string f = line.substr(0, tab), l = line.substr(tab + 1);
      targets.push_back(std::make_pair(f, l));
      if (std::find(labels.begin(), labels.end(), l) == labels.end()) labels.push_back(l);
    }
  }
  EHashtable<KEY, lElement> table(k, labels, labels_c);
  if (!table.iskmerLengthValid()) return 3;
  const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  size_t nt = 0;
  for (size_t t = 0; t < targets.size(); ++t) {
    std::ifstream in(targets[t].first.c_str(), std::ios::binary);
    if (!in) { fprintf(stderr, "Failed to open %s\n", targets[t].first.c_str()); return 4; }
    std::string line;
    uint64_t kmer = 0; int run = 0;
    uint64_t iter = 0;   // light: completed k-blocks of this file (CuCLARK_hh.hh:709)
    while (std::getline(in, line)) {
      if (!line.empty() && line[0] == '>') { kmer = 0; run = 0; continue; }
      for (size_t i = 0; i < line.size(); ++i) {
        int c = code_of((unsigned char)line[i]);
        if (c < 0) { kmer = 0; run = 0; ++nt; continue; }
        ++nt;
        kmer = ((kmer << 2) | (uint64_t)c) & mask;
        if (gap == 0) {                       // every k-mer (CuCLARK_hh.hh:920-950)
          if (++run >= k) table.addElement(kmer, targets[t].second, (size_t)1);
        } else if (++run == k) {              // light: non-overlapping blocks, every gap-th (CuCLARK_hh.hh:721-731)
          if (iter % gap == 0) table.addElement(kmer, targets[t].second, (size_t)1);
          ++iter; kmer = 0; run = 0;
        }
      }
    }
  }
  fprintf(stderr, "%zu nt read; %zu k-mers in mother table\n", nt, table.Size());
  table.SortAllHashTable(2);
  table.RemoveCommon(labels_c, min_count);
  uint64_t n = table.Write(prefix, 2);
  printf("%llu\n", (unsigned long long)n);
  return 0;
}

template <typename KEY>
static int do_query(int k, const char* prefix, const char* kmers_txt, size_t sampling, bool use_mmap) {
  EHashtable<KEY, lElement> table(k);
  size_t fsize = 0;
  if (!table.Read(prefix, fsize, 1, sampling, use_mmap)) { fprintf(stderr, "Read failed\n"); return 5; }
  std::ifstream in(kmers_txt);
  unsigned long long v;
  while (in >> v) {
    ILBL label = 0;
    bool found = table.queryElement((uint64_t)v, label);
    printf("%llu %d %u\n", v, found ? 1 : 0, found ? (unsigned)label : 0u);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "info") {
    printf("HTSIZE %llu LHTSIZE %llu MAXHITS %d NBN %d\n", (unsigned long long)HTSIZE, (unsigned long long)LHTSIZE,
           (int)MAXHITS, (int)NBN);
    return 0;
  }
  if (argc >= 6 && std::string(argv[1]) == "build") {
    int k = atoi(argv[2]), kb = atoi(argv[3]);
    size_t minc = argc > 6 ? (size_t)atol(argv[6]) : 0;
    size_t gap = argc > 7 ? (size_t)atol(argv[7]) : 0;
    if (kb == 2) return do_build<T16>(k, argv[4], argv[5], minc, gap);
    if (kb == 4) return do_build<T32>(k, argv[4], argv[5], minc, gap);
    if (kb == 8) return do_build<T64>(k, argv[4], argv[5], minc, gap);
  }
  if (argc >= 6 && std::string(argv[1]) == "query") {
    int k = atoi(argv[2]), kb = atoi(argv[3]);
    size_t s = argc > 6 ? (size_t)atol(argv[6]) : 1;
    bool mm = argc > 7 && atoi(argv[7]) != 0;
    if (kb == 2) return do_query<T16>(k, argv[4], argv[5], s, mm);
    if (kb == 4) return do_query<T32>(k, argv[4], argv[5], s, mm);
    if (kb == 8) return do_query<T64>(k, argv[4], argv[5], s, mm);
  }
  if (argc >= 5 && std::string(argv[1]) == "merge") {
    mergePairedFiles(argv[2], argv[3], argv[4]);
    return 0;
  }
  if (argc >= 4 && std::string(argv[1]) == "codec") {
    const int k = atoi(argv[2]);
    std::ifstream in(argv[3]);
    std::string line;
    while (std::getline(in, line)) {
      if ((int)line.size() < k) continue;
      uint64_t f = 0;
      getKmers(line, f, (uint8_t)k);
      uint64_t r = f;
      getReverse(r, (uint8_t)k);
      printf("%llu %llu\n", (unsigned long long)f, (unsigned long long)r);
    }
    return 0;
  }
  fprintf(stderr, "usage: %s info | merge <f1> <f2> <out> | codec <k> <kmers_ascii.txt> | build <k> <key_bytes> <out_prefix> <targets.tsv> [min_count] [light_gap] | "
                  "query <k> <key_bytes> <prefix> <kmers.txt> [sampling] [mmap]\n", argv[0]);
  return 1;
}
