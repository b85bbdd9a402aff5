// getAccssnTaxID — maps every sequence file to the accession of its first record and that accession's taxonomy ID.
// Drop-in for the reference tool of the same name (src/getAccssnTaxID.cc:47-190), used by make_metadata.sh:
//   getAccssnTaxID <file of filenames> <nucl_accession2taxid> <merged.dmp>   > <db>.fileToAccssnTaxID
// Output, one line per readable FASTA file in input order: "<file>\t<accession>\t<taxid>" (taxid -1 = not found);
// a file that cannot be opened prints "<file>\tUNKNOWN"; a file whose first line is not a FASTA header is skipped.
// The accession is the header's first word (split on space, tab, ':') cut at '|', '.', '>': the last-but-one piece
// when there are several (">gi|1|ref|NC_1.2|" and ">NC_1.2" both give NC_1), the only piece otherwise.
// Taxonomy IDs retired by NCBI are replaced through merged.dmp (old -> new).
#include <cstdint>
#include <iostream>
#include <map>
#include <unordered_map>

#include "text_util.hpp"

int main(int argc, char** argv) {
  if (argc != 4) {
    std::cerr << "Usage: " << argv[0] << " <./file of filenames> <./nucl_accession2taxid> <./merged.dmp>" << std::endl;
    return 255;
  }
  FILE* merged = fopen(argv[3], "r");
  if (!merged) { std::cerr << "Failed to open " << argv[3] << std::endl; return 255; }
  FILE* acc2tax = fopen(argv[2], "r");
  if (!acc2tax) { std::cerr << "Failed to open " << argv[2] << std::endl; return 255; }
  FILE* list = fopen(argv[1], "r");
  if (!list) { std::cerr << "Failed to open " << argv[1] << std::endl; return 1; }

  struct Seq { std::string file, accession; };
  std::vector<Seq> seqs;
  std::unordered_map<std::string, size_t> slot_of;   // accession -> index into taxid[]
  std::vector<long> taxid;

  std::cerr << "Loading accession number of all files... ";
  std::string file, line;
  while (textutil::read_line(list, file)) {
    FILE* f = fopen(file.c_str(), "r");
    if (!f) {
      std::cerr << "Failed to open sequence file: " << file << std::endl;
      std::cout << file << "\tUNKNOWN" << std::endl;
      continue;
    }
    if (textutil::read_line(f, line)) {
      const std::vector<std::string> words = textutil::split(line, " \t:");
      if (!line.empty() && line[0] == '>' && !words.empty()) {
        const std::vector<std::string> pieces = textutil::split(words[0], "|.>");
        if (!pieces.empty()) {
          const std::string acc = pieces[pieces.size() > 1 ? pieces.size() - 2 : 0];
          if (slot_of.emplace(acc, taxid.size()).second) taxid.push_back(-1);
          seqs.push_back({file, acc});
        }
      }
    }
    fclose(f);
  }
  fclose(list);
  std::cerr << "done (" << slot_of.size() << ")" << std::endl;

  std::cerr << "Loading merged Tax ID... ";
  std::map<long, long> renamed;
  while (textutil::read_line(merged, line)) {
    const std::vector<std::string> w = textutil::split(line, "|.> \t");
    if (w.size() >= 2) renamed.emplace(atol(w[0].c_str()), atol(w[1].c_str()));   // first mapping of an ID wins
  }
  fclose(merged);
  std::cerr << "done" << std::endl;

  std::cerr << "Retrieving taxonomy ID for each file... ";
  size_t found = 0;
  while (found < taxid.size() && textutil::read_line(acc2tax, line)) {   // columns: accession, accession.version, taxid, gi
    const std::vector<std::string> w = textutil::split(line, " \t");
    if (w.size() < 3) continue;
    auto it = slot_of.find(w[0]);
    if (it == slot_of.end()) continue;
    ++found;
    long id = atol(w[2].c_str());
    auto r = renamed.find(id);
    if (r != renamed.end()) id = r->second;
    taxid[it->second] = id;
  }
  fclose(acc2tax);

  size_t mapped = 0, unknown = 0;
  for (const Seq& s : seqs) {
    const long id = taxid[slot_of[s.accession]];
    std::cout << s.file << "\t" << s.accession << "\t" << id << std::endl;
    if (id == -1) ++unknown; else ++mapped;
  }
  std::cerr << "done (" << mapped << " files were successfully mapped";
  if (unknown) std::cerr << ", and " << unknown << " unidentified";
  std::cerr << ")." << std::endl;
  return 0;
}
