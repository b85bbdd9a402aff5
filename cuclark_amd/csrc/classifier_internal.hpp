// classifier_internal.hpp - what the translation units of the command line's host class share (not installed, not part of the C ABI):
// error helpers and the reference's line splitters (file.cc:57-122).
//   classifier.cpp          options, engines (the constructor: devices, parts, reserves, database load), run / run_paired dispatch
//   classifier_feeders.hpp  the inputs: segment sources (mmap, inflate streams, the serial pair reader) and the feeders of the streaming
//                           path (plain file, segments, two plain mates merged by the loaders, compressed input on the device)
//   classifier_stream.cpp   the device-ingest streaming path: ingest slots, FASTQ stripping, run_stream's three thread pools
//   classifier_batch.cpp    the batch path: index, pack, mic_batch_*, CSV lines (the reference's flow, CuCLARK_hh.hh:1339-2139)
#ifndef MIC_CLASSIFIER_INTERNAL_HPP
#define MIC_CLASSIFIER_INTERNAL_HPP
#include "classifier.hpp"

#include "pgz.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mic {
namespace detail {

[[noreturn]] inline void die(const std::string& msg) { throw std::runtime_error(msg); }

inline void check(int rc, const char* what) {
  if (rc != MIC_OK) die(std::string(what) + ": " + mic_last_error());
}

// file.cc:57-80 (split on ' ', ',', '\n', '\t', '\r', at most max elements)
inline std::vector<std::string> split_line(const std::string& line, size_t max_el) {
  std::vector<std::string> out;
  size_t t = 0, n = line.size();
  auto sep = [](char c) { return c == ' ' || c == ',' || c == '\n' || c == '\t' || c == '\r'; };
  while (t < n && out.size() < max_el) {
    while (t < n && sep(line[t])) ++t;
    std::string v;
    while (t < n && !sep(line[t])) v.push_back(line[t++]);
    if (!v.empty()) out.push_back(v);
  }
  return out;
}

// file.cc:83-122 with an explicit separator list
inline std::vector<std::string> split_seps(const std::string& line, const std::string& seps) {
  std::vector<std::string> out;
  size_t t = 0, n = line.size();
  while (t < n) {
    while (t < n && seps.find(line[t]) != std::string::npos) ++t;
    std::string v;
    while (t < n && seps.find(line[t]) == std::string::npos) v.push_back(line[t++]);
    if (!v.empty()) out.push_back(v);
  }
  return out;
}

inline bool get_line(std::istream& in, std::string& line) { return static_cast<bool>(std::getline(in, line)); }

inline bool file_exists(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return false;
  fclose(f);
  return true;
}

inline bool is_gzip(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  unsigned char m[2] = {0, 0};
  size_t n = fread(m, 1, 2, f);
  fclose(f);
  return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}


}  // namespace detail
}  // namespace mic
#endif
