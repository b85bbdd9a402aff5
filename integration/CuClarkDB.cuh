// integration/CuClarkDB.cuh — drop-in replacement for the reference's src/CuClarkDB.cuh + CuClarkDB.cu.
//
// A CuCLARK maintainer who wants to keep the reference's host code (main.cc, CuCLARK_hh.hh) copies this file over
// src/CuClarkDB.cuh, removes CuClarkDB.cu from the build, compiles with g++ (no nvcc, no CUDA headers) and links
// -lmi_clark.  Every public method of class CuClarkDB<HKMERr> (CuClarkDB.cuh:98-150 of the reference) forwards to one
// call of the C ABI in include/mi_clark.h; results are converted to the u16 RESULTS arrays CuCLARK_hh.hh reads.
// This file is the text shown in INTEGRATION.md §2; tests/test_integration_shim.py compiles it against the
// reference's dataType.hh and runs it on the GPU.
#ifndef CUCLARKDB_
#define CUCLARKDB_
#include <vector>
#include <cstdlib>
#include <iostream>
#include "dataType.hh"      // ILBL, RESULTS, CONTAINER, ITYPE, HTSIZE
#include "mi_clark.h"

template <typename HKMERr> class CuClarkDB {
  mic_engine* e_ = nullptr; size_t nb_; uint8_t k_; bool ext_ = false;
  uint32_t *res32_ = nullptr, *rows32_ = nullptr; RESULTS *final_ = nullptr, *full_ = nullptr;
  std::vector<ITYPE> index_; std::vector<size_t> nreads_; size_t rowSize_ = 0, finalRowSize_ = 5;
  static void ck(int rc) { if (rc) { std::cerr << mic_last_error() << std::endl; exit(1); } }   // CUERR behaviour
 public:
  CuClarkDB(size_t numDevices, uint8_t k, size_t numBatches, size_t numTargets) : nb_(numBatches), k_(k) {
    mic_config c = {0, (int32_t)k, (uint32_t)numTargets, (uint32_t)numBatches, 0, 0};      // CuClarkDB.cu:85-253
    ck(mic_create(&c, &e_)); nreads_.resize(numBatches);
  }
  ~CuClarkDB() { mic_destroy(e_); free(final_); free(full_); }
  bool read(const char* prefix, size_t& fileSize, size_t& dbParts, const ITYPE& mod = 1, const bool& = false) {
    int rc = mic_db_load_files(e_, prefix, sizeof(HKMERr), mod, 0, 0);                      // CuClarkDB.cu:461-808
    if (rc == MIC_E_IO) { std::cerr << mic_last_error() << std::endl; return false; }       // "Failed to open ..."
    ck(rc); mic_db_info i; mic_db_get_info(e_, &i); fileSize = i.hbm_bytes; dbParts = 1; return true;
  }
  bool swapDbParts() { return false; }            // whole table resident: no cycles        // CuClarkDB.cu:813-858
  bool sync() { ck(mic_sync(e_)); return true; }
  size_t malloc(size_t numReads, size_t maxReads, size_t maxCont, std::vector<ITYPE>& indexBatches, RESULTS*& full,
                size_t rowSize, RESULTS*& fin, size_t finRowSize, bool isExtended,
                std::vector<uint32_t*>& rp, std::vector<CONTAINER*>& ct) {                  // CuClarkDB.cu:317-419
    index_ = indexBatches; ext_ = isExtended; rowSize_ = rowSize; finalRowSize_ = finRowSize;
    rp.resize(nb_); ct.resize(nb_);
    ck(mic_batches_alloc(e_, numReads, maxReads, maxCont, indexBatches.data(), isExtended, &res32_, &rows32_,
                         rp.data(), (uint16_t**)ct.data()));
    fin = final_ = (RESULTS*)calloc(numReads * finRowSize, sizeof(RESULTS));
    full = full_ = isExtended ? (RESULTS*)calloc(numReads * rowSize, sizeof(RESULTS)) : nullptr;
    return numReads;
  }
  bool readyBatch(size_t b, size_t nReads, size_t nCont) { nreads_[b] = nReads; ck(mic_batch_ready(e_, b, nReads, nCont)); return true; }
  bool queryBatch(size_t b, bool isExtended, bool isFollowup = false) { ck(mic_batch_query(e_, b, isExtended, isFollowup)); return true; }
  bool waitForBatch(size_t b) {                                                             // CuClarkDB.cu:440-445
    ck(mic_batch_wait(e_, b));
    for (size_t r = index_[b]; r < index_[b] + nreads_[b]; ++r) {      // u32 -> RESULTS, the layout CuCLARK_hh.hh reads
      for (int w = 0; w < 5; ++w) final_[r * finalRowSize_ + w] = (RESULTS)res32_[r * MIC_RESULT_WORDS + w];
      if (ext_) {
        const uint32_t* row = rows32_ + r * 16; uint32_t n = row[0] == MIC_ROW_INVALID ? 0 : row[0];
        full_[r * rowSize_] = (RESULTS)n;
        for (uint32_t i = 0; i < n && 2 * i + 2 < rowSize_; ++i) { full_[r * rowSize_ + 2 * i + 1] = row[1 + i] & 0xFFFF;
                                                                   full_[r * rowSize_ + 2 * i + 2] = row[1 + i] >> 16; }
      }
    }
    return true;
  }
  bool checkBatch(size_t b) { int d = 0; ck(mic_batch_check(e_, b, &d)); return d; }
  void freeBatchMemory() { mic_batches_free(e_); free(final_); free(full_); final_ = full_ = nullptr; }
};

#endif
