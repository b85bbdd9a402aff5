// text_util.hpp — line reader and tokenizer shared by the target-definition tools.
// Behaviour follows the reference's helpers (file.cc:89-122 getElementsFromLine with a separator set: any run of
// separator characters splits, empty tokens are dropped; file.cc:124-140 getLineFromFile: one trailing '\n' removed).
#ifndef MIC_TEXT_UTIL_HPP
#define MIC_TEXT_UTIL_HPP

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace textutil {

inline bool read_line(FILE* f, std::string& out) {
  char* buf = nullptr;
  size_t cap = 0;
  const ssize_t n = getline(&buf, &cap, f);
  if (n < 0) { free(buf); return false; }
  out.assign(buf, (size_t)n);
  free(buf);
  if (!out.empty() && out.back() == '\n') out.pop_back();
  return true;
}

inline std::vector<std::string> split(const std::string& line, const char* seps) {
  std::vector<std::string> out;
  std::string cur;
  auto is_sep = [&](char c) { for (const char* s = seps; *s; ++s) if (*s == c) return true; return false; };
  for (char c : line) {
    if (is_sep(c)) { if (!cur.empty()) { out.push_back(cur); cur.clear(); } }
    else cur.push_back(c);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

}  // namespace textutil
#endif
