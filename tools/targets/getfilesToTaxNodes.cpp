// getfilesToTaxNodes — lineage (species, genus, family, order, class, phylum) of every sequence file's taxonomy ID.
// Drop-in for the reference tool of the same name (src/getfilesToTaxNodes.cc:47-154), used by make_metadata.sh:
//   getfilesToTaxNodes <nodes.dmp> <db>.fileToAccssnTaxID   > <db>.fileToTaxIDs
// Output per input line "<file> <accession> <taxid>": "<file>\t<taxid>" followed by six tab-separated columns, each the
// taxonomy ID of the ancestor (or the node itself) at that rank, or UNKNOWN.  Ranks named "... group"/"... subgroup"
// do not count; the walk stops below the root (a node whose parent is 1 is not reported, as in the reference) and at an
// ID missing from nodes.dmp (where the reference would not terminate).
#include <cstdint>
#include <iostream>
#include <map>

#include "text_util.hpp"

namespace {
const int kRanks = 6;
struct Node { uint32_t parent = 0; uint8_t rank = 255; };
}

int main(int argc, char** argv) {
  if (argc != 3) {
    std::cerr << "Usage: " << argv[0] << " <./nodes.dmp> <./file_taxid>" << std::endl;
    return 255;
  }
  FILE* fn = fopen(argv[1], "r");
  if (!fn) { std::cerr << "Failed to open " << argv[1] << std::endl; return 255; }
  FILE* ft = fopen(argv[2], "r");
  if (!ft) { std::cerr << "Failed to open " << argv[2] << std::endl; return 255; }

  const std::map<std::string, uint8_t> rank_of = {{"species", 0}, {"genus", 1}, {"family", 2},
                                                  {"order", 3},   {"class", 4}, {"phylum", 5}};
  std::vector<Node> nodes;
  std::string line;
  std::cerr << "Loading nodes of taxonomy tree... ";
  while (textutil::read_line(fn, line)) {        // "id | parent | rank name | ..."
    const std::vector<std::string> w = textutil::split(line, " |\t");
    if (w.size() < 3) continue;
    const long id = atol(w[0].c_str()), parent = atol(w[1].c_str());
    if (id < 0 || parent < 0) continue;
    if ((size_t)id >= nodes.size()) nodes.resize((size_t)id + 1 + nodes.size() / 2);
    nodes[(size_t)id].parent = (uint32_t)parent;
    auto r = rank_of.find(w[2]);
    if (r != rank_of.end() && (w.size() == 3 || w[3].find("group") == std::string::npos)) nodes[(size_t)id].rank = r->second;
  }
  fclose(fn);
  std::cerr << "done." << std::endl;

  std::cerr << "Retrieving lineage for each sequence... ";
  while (textutil::read_line(ft, line)) {
    const std::vector<std::string> w = textutil::split(line, " |\t");
    if (w.size() < 3) continue;
    const long id = atol(w[2].c_str());
    std::cout << w[0] << "\t" << id;
    long at_rank[kRanks];
    for (int r = 0; r < kRanks; ++r) at_rank[r] = -1;
    if (id > 0) {
      size_t it = (size_t)id;
      while (it != 1 && it < nodes.size() && it != 0 && nodes[it].parent != 1) {
        const uint8_t r = nodes[it].rank;
        if (r < kRanks && at_rank[r] < 0) at_rank[r] = (long)it;   // the lowest node of each rank
        if (nodes[it].parent == it) break;
        it = nodes[it].parent;
      }
    }
    for (int r = 0; r < kRanks; ++r) {
      if (at_rank[r] >= 0) std::cout << "\t" << at_rank[r];
      else std::cout << "\tUNKNOWN";
    }
    std::cout << std::endl;
  }
  fclose(ft);
  std::cerr << "done." << std::endl;
  return 0;
}
