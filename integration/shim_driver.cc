// shim_driver.cc — exercises integration/CuClarkDB.cuh with the call sequence CuCLARK_hh.hh makes on the class
// (CuCLARK_hh.hh:608 ctor, :621 read, :514-515 swapDbParts+sync, :1600-1606 malloc, :1735 readyBatch, :1743 queryBatch,
// :1997 waitForBatch, :335 freeBatchMemory).  Reads a packed batch (reads_pointer u32, containers u16) from two binary
// files, prints "sum idxBest best idxSecond second" per read.
//   shim_driver <db prefix> <k> <num targets> <reads_pointer.bin> <containers.bin> [numDevices [extended [rowSize]]]
// rowSize: u16 words per sparse row the caller allocates (the reference: 2 * MAXHITS + 2, CuCLARK_hh.hh:1587-1590)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include "CuClarkDB.cuh"

typedef uint32_t T32;

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const uint8_t k = (uint8_t)atoi(argv[2]);
  const size_t numTargets = (size_t)atol(argv[3]);
  std::ifstream f1(argv[4], std::ios::binary), f2(argv[5], std::ios::binary);
  std::vector<char> b1((std::istreambuf_iterator<char>(f1)), std::istreambuf_iterator<char>());
  std::vector<char> b2((std::istreambuf_iterator<char>(f2)), std::istreambuf_iterator<char>());
  const size_t numReads = b1.size() / 4 - 1, numCont = b2.size() / 2;
  const size_t numBatches = 1;
  const size_t numDevices = argc > 6 ? (size_t)atol(argv[6]) : 1;
  const bool extended = argc > 7 && atoi(argv[7]) != 0;
  CuClarkDB<T32> db(numDevices, k, numBatches, numTargets);
  size_t fileSize = 0, dbParts = 0;
  if (!db.read(argv[1], fileSize, dbParts, 1, false)) return 3;
  db.swapDbParts();
  db.sync();
  std::vector<ITYPE> indexBatches(numBatches + 1);
  indexBatches[0] = 0; indexBatches[1] = (ITYPE)numReads;
  RESULTS *full = nullptr, *fin = nullptr;
  std::vector<uint32_t*> readsPointer;
  std::vector<CONTAINER*> readsInContainers;
  const size_t rowSize = argc > 8 ? (size_t)atol(argv[8]) : 2 * MAXHITS + 2, finalRowSize = 5;
  db.malloc(numReads, numReads, numCont, indexBatches, full, rowSize, fin, finalRowSize, extended, readsPointer, readsInContainers);
  memcpy(readsPointer[0], b1.data(), b1.size());
  memcpy(readsInContainers[0], b2.data(), b2.size());
  db.readyBatch(0, numReads, numCont);
  db.queryBatch(0, extended);
  db.waitForBatch(0);
  if (!db.checkBatch(0)) return 4;
  {  // getFinalResult (CuClarkDB.cuh:148): the same rows as the lent buffer holds
    std::vector<RESULTS> copy(numReads * finalRowSize);
    if (!db.getFinalResult(0, copy.data()) || memcmp(copy.data(), fin, copy.size() * sizeof(RESULTS)) != 0) return 5;
  }
  for (size_t t = 0; t < numReads; ++t)
    printf("%u %u %u %u %u\n", fin[t * finalRowSize], fin[t * finalRowSize + 1], fin[t * finalRowSize + 2],
           fin[t * finalRowSize + 3], fin[t * finalRowSize + 4]);
  if (extended)      // the sparse rows as CuCLARK_hh.hh:2014-2031 reads them: n, then (target, count) pairs
    for (size_t t = 0; t < numReads; ++t) {
      printf("row %u", full[t * rowSize]);
      for (unsigned i = 0; i < full[t * rowSize]; ++i) printf(" %u:%u", full[t * rowSize + 2 * i + 1], full[t * rowSize + 2 * i + 2]);
      printf("\n");
    }
  db.freeBatchMemory();
  return 0;
}
