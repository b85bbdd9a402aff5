// classifier.hpp — C++ host above the C ABI: the equivalent of the reference's CuCLARK<HKMERr> host class
// (CuCLARK_hh.hh:51-195) for the classification path.  One class, runtime key width and table size.
#ifndef MIC_CLASSIFIER_HPP
#define MIC_CLASSIFIER_HPP

#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "mi_clark.h"

namespace mic {

struct Options {
  size_t k = 31;              // -k
  uint32_t min_count_t = 0;   // -t
  size_t threads = 1;         // -n
  size_t batches = 1;         // -b
  size_t devices = 0;         // -d (0 = all)
  uint32_t sampling = 1;      // -s
  uint64_t gap = 0;           // -g (light only)
  bool tsk = false;           // --tsk
  bool extended = false;      // --extended
  bool light = false;         // cuCLARK-l
  bool db_sharded = false;    // --db-sharded: the devices hold parts of the table (the reference's -d mode), see Classifier's ctor
  size_t parts = 0;           // --parts P: parts the table is cut into under --db-sharded (0 = the smallest number whose part fits a device)
  uint64_t htsize = 1610612741ull;  // parameters.hh:39 / parameters_light_hh:40; --htsize overrides
  std::string targets, folder, objects, objects2, results;
};

class Classifier {
 public:
  // CuCLARK ctor (CuCLARK_hh.hh:221-310): parses the targets file, locates the database, creates one engine per
  // device and loads the database into each.  Throws std::runtime_error with the reference's message on failure.
  explicit Classifier(const Options& opt);
  ~Classifier();

  // CuCLARK::run single-end (CuCLARK_hh.hh:383-428) and paired-end (:433-506), incl. list-of-files mode.
  void run(const std::string& objects, const std::string& results);
  void run_paired(const std::string& f1, const std::string& f2, const std::string& results);

  // CuCLARK::runSimple (:512-574) on an in-memory FASTA/FASTQ image.
  void run_buffer(const uint8_t* map, size_t nb, const std::string& results_base, bool paired);
  // The same over a stream of segments (whole records each): the next segment is read / inflated / merged on a side
  // thread while the current one is indexed, packed, queried and written.  Memory is bounded by two segments.
  struct Segment { const uint8_t* p = nullptr; size_t n = 0; std::string own; };
  class SegmentSource { public: virtual ~SegmentSource() {} virtual bool next(Segment& s) = 0; };
  void run_segments(SegmentSource& src, const std::string& results_base, bool paired);

  // Device-ingest streaming (the default for the non-extended CSV): batches of whole records go to the GPU as the bytes
  // of the file and come back as the bytes of the CSV (mic_ingest_*); host threads only move bytes.  A batch the device
  // path hands back (MIC_INGEST_FALLBACK) or that does not fit a slot goes through process_segment instead.
  struct Range {
    uint64_t off = 0; size_t len = 0; const uint8_t* mem = nullptr; std::shared_ptr<Segment> keep;
    uint64_t off2 = 0; size_t len2 = 0;      // paired-end files: the same records in the second file
  };
  class Feeder {            // assign() is called under a lock and hands out consecutive ranges of whole records
   public:
    virtual ~Feeder() {}
    virtual bool assign(size_t want, size_t cap, Range& r) = 0;
    virtual void read(const Range& r, size_t off, uint8_t* dst, size_t len) = 0;   // bytes [off, off + len) of the range; any thread
    virtual bool fastq() const = 0;                          // the input's records are four-line FASTQ records
    virtual uint64_t remaining() const { return ~(uint64_t)0 >> 1; }   // bytes not yet handed out, if known
    // the text of the range as the classifier sees it, into a slot (returns its size, (size_t)-1 when it does not fit) or
    // into a string; for most feeders that is the bytes of the range, for a pair of files it is the merged text
    virtual size_t fill(const Range& r, uint8_t* dst, size_t cap) { if (r.len > cap) return (size_t)-1; read(r, 0, dst, r.len); return r.len; }
    virtual void text(const Range& r, std::string& out) { out.resize(r.len); if (r.len) read(r, 0, (uint8_t*)&out[0], r.len); }
    // a feeder whose text is on the device already writes a range into an ingest slot's DEVICE buffer (returns its size, (size_t)-1
    // when the range has to go through the host path); the batch is then classified with MIC_INGEST_RESIDENT
    virtual bool resident() const { return false; }
    virtual int resident_flags() const { return MIC_INGEST_RESIDENT; }
    virtual size_t fill_resident(const Range&, mic_engine*, size_t /*slot*/) { return (size_t)-1; }
    virtual bool gave_up() const { return false; }           // the input is not what the feeder can cut: run the serial reader instead
  };
  bool run_stream(Feeder& f, const std::string& results_base, bool paired, size_t total_bytes);   // false: feeder gave up, nothing written

  std::string db_name() const;  // getdbName, CuCLARK_hh.hh:580-591
  const std::vector<std::string>& target_names() const { return names_; }

 private:
  void parse_targets();  // getTargetsData, CuCLARK_hh.hh:1795-1906
  size_t process_segment(const uint8_t* map, size_t nb, bool paired, FILE* fout);   // returns the number of objects
  bool device_ingest() const;                 // is the streaming path usable for this run?
  void ingest_geometry(size_t total_bytes, size_t& slot_bytes, size_t& slots) const;
  void ensure_ingest(size_t total_bytes);     // slots sized for the input at hand
  void release_ingest();
  std::vector<std::vector<uint8_t*>> ingest_raw_;   // [engine][slot]: pinned input buffers lent by the engines
  size_t ingest_bytes_ = 0, ingest_workers_ = 0;
  std::string* sink_ = nullptr;               // process_segment appends here instead of writing to the file
  void ensure_batches(size_t max_reads, size_t max_cont);
  void release_batches();
  struct Lent { uint32_t* results = nullptr; uint32_t* rows = nullptr; std::vector<uint32_t*> rp; std::vector<uint16_t*> ct; };
  std::vector<Lent> lent_;
  std::vector<uint64_t> ix_[5];   // name_s, name_e, seq_s, seq_e, length of the current segment (reused)
  size_t slot_reads_ = 0, slot_cont_ = 0, slots_per_engine_ = 0;
  uint32_t row_words_ = 16;
  size_t segment_bytes_ = 512u << 20;
  Options opt_;
  std::vector<std::pair<std::string, std::string>> targets_id_;
  std::vector<std::string> labels_, labels_c_, names_;
  std::vector<mic_engine*> engines_;          // groups_ x parts_ engines: engine g * parts_ + p holds part p of the table for read group g
  size_t parts_ = 1, groups_ = 1;
  bool gz_on_device_ = true;                  // compressed input is inflated on the first engine's device (needs peer access from the others)
  std::atomic<size_t> n_objects_{0};
  double prelude_s_ = 0;                      // seconds spent inflating a compressed input before the streaming path started
};

// file.cc:205-268: merged FASTA text of two FASTQ mates ("seq1" + 'N' + "seq2").
std::string merge_paired(const std::string& file1, const std::string& file2);
// The same text from the loaders' parallel merger (PairedFileFeeder: line counts, then batches of `batch_bytes` merged
// independently); false when the files are not what it can cut (the caller then runs merge_paired, which ends the way
// the reference ends on such files).
bool merge_paired_parallel(const std::string& file1, const std::string& file2, unsigned threads, size_t batch_bytes, std::string& out);
// test hook: FASTQ text with the '+' and quality lines dropped, as the loaders hand it to the device (cuCLARK --strip-fastq)
std::string strip_fastq_text(const std::string& in, size_t piece, bool scalar, int reps = 1);
double strip_fastq_loaders_rate(const std::string& path, size_t chunk, unsigned threads, bool use_mmap);

}  // namespace mic
#endif
