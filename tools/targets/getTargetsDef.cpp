// getTargetsDef — picks one taxonomy rank out of <db>.fileToTaxIDs and prints the targets definition.
// Drop-in for the reference tool of the same name (src/getTargetsDef.cc:38-96), used by set_targets.sh:
//   getTargetsDef <db>.fileToTaxIDs [rank 0..5]   >> targets.txt        (0 species ... 5 phylum; default 1 as in the reference)
// Prints "<file>\t<taxid at the rank>" for every line whose second column is not -1 and whose rank column is not UNKNOWN.
// Files without a taxonomy ID (second column -1) are listed in ./files_excluded.txt; their number is the exit code.
#include <fstream>
#include <iostream>

#include "text_util.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "Usage: " << argv[0]
              << " <FilestoTaxIDs>, option: <Rank: 0,1,2,3,4,5>, 0 for species, 1 for genus, ..., 5 for phylum. Default is species."
              << std::endl;
    return 1;
  }
  FILE* f = fopen(argv[1], "r");
  if (!f) { std::cerr << "Failed to open " << argv[1] << std::endl; return 1; }
  int rank = 1;
  if (argc > 2) {
    rank = atoi(argv[2]);
    if (rank > 5) {
      std::cerr << "Failed to recognize the rank. Please type a number between 0 and 5, according to the following:" << std::endl;
      std::cerr << "0: species, 1: genus, 2: family, 3: order, 4:class, and 5: phylum." << std::endl;
      return 1;
    }
  }
  std::ofstream excluded("files_excluded.txt", std::ios::binary);
  size_t n_excluded = 0;
  std::string line;
  while (textutil::read_line(f, line)) {
    const std::vector<std::string> w = textutil::split(line, "\t, ");
    if (w.size() < 2) continue;
    if (w[1] != "-1") {
      const size_t col = (size_t)(2 + rank);
      if (col < w.size() && w[col] != "UNKNOWN") std::cout << w[0] << "\t" << w[col] << std::endl;
    } else {
      if (++n_excluded == 1) excluded << "The following files have been excluded from the targets definition" << std::endl;
      excluded << w[0] << std::endl;
    }
  }
  fclose(f);
  excluded.close();
  return (int)n_excluded;
}
