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
#include <cstring>
#include <cstdio>
#include <string>
#include <iostream>
#include "dataType.hh"      // ILBL, RESULTS, CONTAINER, ITYPE, HTSIZE
#include "mi_clark.h"

template <typename HKMERr> class CuClarkDB {
  // numDevices engines, each holding one part of the table (the reference's multi-device mode, CuClarkDB.cu:104-208, 566-574,
  // 886-974): every engine gets the batch's reads, the sparse rows are summed read-range owned (mic_batch_merge_shards)
  std::vector<mic_engine*> e_; size_t nb_; uint8_t k_; bool ext_ = false; uint32_t rw_ = 16, nt_ = 0;   // rw_: u32 words per sparse row
  uint32_t *res32_ = nullptr, *rows32_ = nullptr; RESULTS *final_ = nullptr, *full_ = nullptr;
  std::vector<ITYPE> index_; std::vector<size_t> nreads_, ncont_; size_t rowSize_ = 0, finalRowSize_ = 5;
  std::vector<char> waited_;      // batch b's results are merged and converted (until its next queryBatch)
  std::vector<std::vector<uint32_t*> > rp_; std::vector<std::vector<uint16_t*> > ct_;   // [engine][batch]
  static void ck(int rc) { if (rc) { std::cerr << mic_last_error() << std::endl; exit(1); } }   // CUERR behaviour
 public:
  CuClarkDB(size_t numDevices, uint8_t k, size_t numBatches, size_t numTargets) : nb_(numBatches), k_(k), nt_((uint32_t)numTargets) {
    int have = 0; ck(mic_device_count(&have));                                              // CuClarkDB.cu:104-181
    if (numDevices == 0 || numDevices > (size_t)have) numDevices = have > 0 ? (size_t)have : 1;
    if (const char* env = getenv("MIC_SHARD_ENGINES")) { long v = atol(env); if (v >= 1 && v <= 64) numDevices = (size_t)v; }
    for (size_t d = 0; d < numDevices; ++d) {
      mic_config c = {(int32_t)(d % (size_t)(have > 0 ? have : 1)), (int32_t)k, (uint32_t)numTargets, (uint32_t)numBatches, rw_, 0};
      mic_engine* e = nullptr; ck(mic_create(&c, &e)); e_.push_back(e);
    }
    nreads_.resize(numBatches); ncont_.resize(numBatches);
  }
  ~CuClarkDB() { for (size_t d = 0; d < e_.size(); ++d) mic_destroy(e_[d]); free(final_); free(full_); }
  bool read(const char* prefix, size_t& fileSize, size_t& dbParts, const ITYPE& mod = 1, const bool& = false) {
    fileSize = 0;
    uint64_t H = (uint64_t)HTSIZE;                                // the table size is the size of the .sz file (= HTSIZE in a CuCLARK build)
    if (FILE* f = fopen((std::string(prefix) + ".sz").c_str(), "rb")) { fseek(f, 0, SEEK_END); H = (uint64_t)ftell(f); fclose(f); }
    (void)H;
    // every device answers for one part of the table (the reference's m_partPointer ranges, CuClarkDB.cu:566-574; here a slot range
    // of the resident table, mic_db_set_part); the files are read once for all devices, the parts built at the same time (:461-808)
    for (size_t d = 0; d < e_.size() && e_.size() > 1; ++d) ck(mic_db_set_part(e_[d], (uint32_t)d, (uint32_t)e_.size()));
    int rc = e_.size() > 1 ? mic_db_load_files_multi(e_.data(), e_.size(), prefix, sizeof(HKMERr), mod)
                           : mic_db_load_files(e_[0], prefix, sizeof(HKMERr), mod, 0, 0);
    if (rc == MIC_E_IO) { std::cerr << mic_last_error() << std::endl; return false; }       // "Failed to open ..."
    ck(rc);
    for (size_t d = 0; d < e_.size(); ++d) { mic_db_info i; mic_db_get_info(e_[d], &i); fileSize += i.hbm_bytes; }
    dbParts = 1; return true;
  }
  bool swapDbParts() { return false; }            // whole table resident: no cycles        // CuClarkDB.cu:813-858
  bool sync() { for (size_t d = 0; d < e_.size(); ++d) ck(mic_sync(e_[d])); return true; }
  size_t malloc(size_t numReads, size_t maxReads, size_t maxCont, std::vector<ITYPE>& indexBatches, RESULTS*& full,
                size_t rowSize, RESULTS*& fin, size_t finRowSize, bool isExtended,
                std::vector<uint32_t*>& rp, std::vector<CONTAINER*>& ct) {                  // CuClarkDB.cu:317-419
    index_ = indexBatches; ext_ = isExtended; rowSize_ = rowSize; finalRowSize_ = finRowSize;
    const bool rows = isExtended || e_.size() > 1;                // shards exchange their sparse rows
    rp_.assign(e_.size(), std::vector<uint32_t*>(nb_)); ct_.assign(e_.size(), std::vector<uint16_t*>(nb_));
    for (size_t d = 0; d < e_.size(); ++d) {
      uint32_t *res = nullptr, *rw = nullptr;
      ck(mic_batches_alloc(e_[d], numReads, maxReads, maxCont, indexBatches.data(), rows, &res, &rw, rp_[d].data(), ct_[d].data()));
      if (d == 0) { res32_ = res; rows32_ = rw; }
    }
    rp = rp_[0]; ct.resize(nb_); for (size_t b = 0; b < nb_; ++b) ct[b] = (CONTAINER*)ct_[0][b];
    fin = final_ = (RESULTS*)calloc(numReads * finRowSize, sizeof(RESULTS));
    full = full_ = isExtended ? (RESULTS*)calloc(numReads * rowSize, sizeof(RESULTS)) : nullptr;
    return numReads;
  }
  bool readyBatch(size_t b, size_t nReads, size_t nCont) {
    nreads_[b] = nReads; ncont_[b] = nCont;
    ck(mic_batch_ready(e_[0], b, nReads, nCont));                 // the reads live in the first engine's lent buffers
    return true;
  }
  bool queryBatch(size_t b, bool isExtended, bool isFollowup = false) {                     // CuClarkDB.cu:878-1033
    if (waited_.size() <= b) waited_.resize(b + 1, 0);
    waited_[b] = 0;
    // every device sees all reads (:886-890): ONE upload into the first engine, the packed reads fanned out device to device
    if (e_.size() > 1) ck(mic_batch_query_group(e_.data(), e_.size(), b, 1));
    else ck(mic_batch_query(e_[0], b, isExtended, isFollowup));
    return true;
  }
  // A read whose sparse row did not fit (more targets than a row holds - MAXHITS in the reference, whose kernel then prints "Too
  // many different tagets hit by a sequence. Results will be corrupted." and leaves a row count beyond the row, CuClarkDB.cu:1200-1211)
  // is completed EXACTLY from dense counts summed over the devices: sum / best / second-best under the reference's order
  // (:1440-1459), and the row gets as many (target, count) pairs, ascending by target, as rowSize holds, with n = the pairs stored.
  void complete(size_t b, size_t r, uint32_t T) {
    std::vector<uint32_t> dense(T, 0), part(T);
    for (size_t d = 0; d < e_.size(); ++d) {
      ck(mic_batch_dense_counts(e_[d], b, r - index_[b], part.data()));
      for (uint32_t t = 0; t < T; ++t) dense[t] += part[t];
    }
    uint32_t sum = 0, best = 0, ib = 0, sb = 0, is = 0, n = 0;
    for (uint32_t t = 0; t < T; ++t) {
      const uint32_t c = dense[t];
      if (!c) continue;
      sum += c;
      if (c > best) { sb = best; is = ib; best = c; ib = t + 1; } else if (c > sb) { sb = c; is = t + 1; }
      if (ext_ && 2 * n + 2 < rowSize_) { full_[r * rowSize_ + 2 * n + 1] = (RESULTS)t; full_[r * rowSize_ + 2 * n + 2] = (RESULTS)c; ++n; }
    }
    const uint32_t five[5] = {sum, ib, best, is, sb};
    for (int w = 0; w < 5; ++w) final_[r * finalRowSize_ + w] = (RESULTS)five[w];
    if (ext_) full_[r * rowSize_] = (RESULTS)n;
  }
  bool waitForBatch(size_t b) {                                                             // CuClarkDB.cu:440-445
    if (waited_.size() > b && waited_[b]) return true;            // (the reference's wait is an event wait: calling it again is harmless)
    if (waited_.size() <= b) waited_.resize(b + 1, 0);
    waited_[b] = 1;
    if (e_.size() > 1) ck(mic_batch_merge_shards(e_.data(), e_.size(), b));                 // peer copies + mergeKernel + resultKernel, :954-1024
    else ck(mic_batch_wait(e_[0], b));
    for (size_t r = index_[b]; r < index_[b] + nreads_[b]; ++r) {      // u32 -> RESULTS, the layout CuCLARK_hh.hh reads
      const uint32_t* row = rows32_ ? rows32_ + r * rw_ : nullptr;
      const bool over = (res32_[r * MIC_RESULT_WORDS + 6] & MIC_FLAG_ROW_OVERFLOW) || (row && row[0] == MIC_ROW_INVALID) ||
                        (ext_ && row && 2 * (size_t)row[0] + 2 > rowSize_);
      if (over) { complete(b, r, nt_); continue; }
      for (int w = 0; w < 5; ++w) final_[r * finalRowSize_ + w] = (RESULTS)res32_[r * MIC_RESULT_WORDS + w];
      if (ext_) {
        const uint32_t n = row[0];
        full_[r * rowSize_] = (RESULTS)n;
        for (uint32_t i = 0; i < n; ++i) { full_[r * rowSize_ + 2 * i + 1] = row[1 + i] & 0xFFFF;
                                           full_[r * rowSize_ + 2 * i + 2] = row[1 + i] >> 16; }
      }
    }
    return true;
  }
  // CuClarkDB.cuh:148 - declared by the reference, never defined or called there (results reach the caller through the buffers
  // malloc lent out): here, for completeness, a copy of the batch's final rows once the batch is done
  bool getFinalResult(size_t b, RESULTS* finalResult) {
    if (!finalResult || !waitForBatch(b)) return false;
    memcpy(finalResult, final_ + index_[b] * finalRowSize_, nreads_[b] * finalRowSize_ * sizeof(RESULTS));
    return true;
  }
  bool checkBatch(size_t b) { int d = 1; for (size_t i = 0; i < e_.size() && d; ++i) ck(mic_batch_check(e_[i], b, &d)); return d; }
  void freeBatchMemory() { for (size_t d = 0; d < e_.size(); ++d) mic_batches_free(e_[d]); free(final_); free(full_); final_ = full_ = nullptr; }
};

#endif
