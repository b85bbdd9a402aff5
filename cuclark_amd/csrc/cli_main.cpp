// cli_main.cpp — command line of exe/cuCLARK and exe/cuCLARK-l (one binary; the light variant is selected by the
// program name or --light).  Flag surface, defaults, messages and exit codes follow the reference's main.cc:74-320,
// so classify_metagenome.sh can exec this binary unchanged (classify_metagenome.sh:155-159).
// Additions that do not change defaults: --light, --htsize <n> (table size = size of the .sz file; the reference
// fixes it at compile time, parameters.hh:39 / parameters_light_hh:40).
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>

#include <sys/time.h>

#include <fstream>
#include <iostream>
#include <iterator>
#include <string>

#include "classifier.hpp"

#define MAXK 32
#define SFACTORMAX 30
#define VERSION "1.1"
static const uint64_t HTSIZE_FULL = 1610612741ull, HTSIZE_LIGHT = 57777779ull;

static bool valid_file(const char* f) {
  FILE* fd = fopen(f, "r");
  if (!fd) return false;
  fclose(fd);
  return true;
}

static void print_usage(const char* prog) {
  std::cout << "\n" << prog << " -- MI355X-native k-mer read classifier with CuCLARK's command line\n\n";
  std::cout << prog << " -k <kmerSize> -t <minFreqTarget> -T <fileTargets> -D <directoryDB/> -O <fileObjects> -R <fileResults> "
               "-n <numberofthreads> -b <numberofbatches> -d <numberofdevices> ...\n\n";
  std::cout << "Definitions of parameters (cf. README of CuCLARK):\n";
  std::cout << "-k <kmerSize>,       k-mer length, integer in [2,32] (default 31; the light variant always uses 27)\n";
  std::cout << "-t <minFreqTarget>,  minimum k-mer frequency in targets (part of the database name)\n";
  std::cout << "-T <fileTargets>,    targets definition: one line per reference file, '<file> <label>'\n";
  std::cout << "-D <directoryDB/>,   directory of the database files db_central_k*_t*_s*_m*.tsk.{sz,ky,lb}\n";
  std::cout << "-O <fileObjects>,    FASTA/FASTQ file of objects (or a list of files when -R names an existing list)\n";
  std::cout << "-P <file1> <file2>,  paired-end FASTQ files\n";
  std::cout << "-R <fileResults>,    results file name ('.csv' is appended)\n";
  std::cout << "-n <numberofthreads> host threads (raises the number of batches if needed)\n";
  std::cout << "-b <numberofbatches> batches the objects are split into (>= threads)\n";
  std::cout << "-d <numberofdevices> GPUs to use (default: all)\n";
  std::cout << "-g <iteration>,      gap for the light database name (>= 4)\n";
  std::cout << "-s <factor>,         sampling factor in [2," << SFACTORMAX << "]\n";
  std::cout << "--db-sharded,        several GPUs: the table is cut into parts held by different GPUs (the mode of CuCLARK's -d); default: the table\n"
               "                     is replicated on every GPU and the batches are dealt to the GPUs\n";
  std::cout << "--parts <P>,         with --db-sharded: number of parts (divides -d; the GPUs form d/P groups that share the reads);\n"
               "                     default: the smallest number of parts that fit a GPU's memory\n";
  std::cout << "--tsk, --extended, --light, --htsize <n>, --help, --version\n\n";
}

int main(int argc, char** argv) {
  const char* slash = strrchr(argv[0], '/');
  const std::string prog = slash ? slash + 1 : argv[0];
  if (argc == 2) {
    std::string val(argv[1]);
    if (val == "--help" || val == "--HELP") { print_usage(argv[0]); return 0; }
    if (val == "--version" || val == "--VERSION") {
      std::cout << "Version: " << VERSION << " (mi-clark MI355X engine; CuCLARK Copyright 2016-2017 Robin Kobus, rkobus@students.uni-mainz.de)" << std::endl;
      std::cout << "Based on CLARK version 1.1.3 (UCR CS&E. Copyright 2013-2016 Rachid Ounit, rouni001@cs.ucr.edu) " << std::endl;
      return 0;
    }
  }
  // cuCLARK --merge-pairs <file1> <file2> <out.fa> [serial|parallel [threads [batch_bytes]]]: the paired-end merge alone
  // (file.cc:205-268: the reference writes <file1>_ConcatenatedByCLARK.fa and classifies that), no device involved.
  // "parallel" prints "gave up" and exits 3 when the loaders' merger hands the files to the serial reader.
  if (argc >= 5 && std::string(argv[1]) == "--merge-pairs") {
    try {
      std::string text;
      const bool par = argc > 5 && std::string(argv[5]) == "parallel";
      if (par) {
        const unsigned th = argc > 6 ? (unsigned)atoi(argv[6]) : 4u;
        const size_t bb = argc > 7 ? (size_t)strtoull(argv[7], nullptr, 10) : (size_t)1 << 20;
        if (!mic::merge_paired_parallel(argv[2], argv[3], th ? th : 1u, bb ? bb : 1, text)) { std::cerr << "gave up" << std::endl; return 3; }
      } else text = mic::merge_paired(argv[2], argv[3]);
      FILE* f = fopen(argv[4], "wb");
      if (!f || fwrite(text.data(), 1, text.size(), f) != text.size() || fclose(f) != 0) { std::cerr << "Failed to write " << argv[4] << std::endl; return 1; }
      return 0;
    } catch (const std::exception& ex) {
      std::cerr << ex.what() << std::endl;
      return 1;
    }
  }
  // cuCLARK --strip-fastq <in.fq> <out> [piece_bytes [scalar|bench]]: the loaders' FASTQ stripper alone (header + sequence line
  // of every record), fed in pieces; "bench" prints both forms' rates instead of writing
  if (argc >= 4 && std::string(argv[1]) == "--strip-fastq") {
    try {
      const size_t piece = argc > 4 ? (size_t)strtoull(argv[4], nullptr, 10) : (size_t)1 << 20;
      const std::string mode = argc > 5 ? argv[5] : "";
      std::string in;
      if (mode != "loaders") {
        std::ifstream f(argv[2], std::ios::binary);
        in.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      }
      if (mode == "loaders") {
        const unsigned th = argc > 6 ? (unsigned)atoi(argv[6]) : 8u;
        const bool mm = argc > 7 && std::string(argv[7]) == "mmap";
        for (int rep = 0; rep < 3; ++rep)
          std::cout << (mm ? "mmap" : "pread") << " chunk " << piece << " threads " << th << ": " << mic::strip_fastq_loaders_rate(argv[2], piece, th ? th : 1, mm) << " GB/s" << std::endl;
        return 0;
      }
      if (mode == "bench") {
        for (int sc = 1; sc >= 0; --sc) {
          struct timeval a, b; gettimeofday(&a, nullptr);
          size_t tot = 0;
          tot = 5 * mic::strip_fastq_text(in, piece ? piece : 1, sc != 0, 5).size();
          gettimeofday(&b, nullptr);
          const double sec = (b.tv_sec - a.tv_sec) + (b.tv_usec - a.tv_usec) * 1e-6;
          std::cout << (sc ? "scalar" : "vector") << ": " << 5.0 * in.size() / sec / 1e9 << " GB/s in (" << tot / 5 << " bytes out)" << std::endl;
        }
        return 0;
      }
      const std::string out = mic::strip_fastq_text(in, piece ? piece : 1, mode == "scalar");
      FILE* o = fopen(argv[3], "wb");
      if (!o || fwrite(out.data(), 1, out.size(), o) != out.size() || fclose(o) != 0) { std::cerr << "Failed to write " << argv[3] << std::endl; return 1; }
      return 0;
    } catch (const std::exception& ex) {
      std::cerr << ex.what() << std::endl;
      return 1;
    }
  }
  if (argc < 6) {
    std::cerr << "To run " << argv[0] << ", at least four  parameters are necessary:\n";
    std::cerr << "filename of the targets definition, directory of database, filename for objects, filename for results." << std::endl;
    print_usage(argv[0]);
    return -1;
  }
  mic::Options o;
  size_t k = 31, cpu = 1, batches = 1, devices = 0, gap = 0;
  uint32_t minT = 0, sfactor = 1;
  bool ext = false, tsk = false;
  bool light = prog.size() >= 2 && prog.compare(prog.size() - 2, 2, "-l") == 0;
  uint64_t htsize_override = 0;
  bool db_sharded = false;
  size_t parts = 0;
  int i_targets = -1, i_objects = -1, i_objects2 = -1, i_folder = -1, i_results = -1;

  for (int i = 1; i < argc; i++) {
    std::string val(argv[i]);
    auto need = [&](const char* msg) { if (++i >= argc) { std::cerr << msg << std::endl; exit(1); } };
    if (val == "-k") {
      need("Please specify the k-mer length!");
      k = (size_t)atoi(argv[i]);
      if (k <= 1 || k > MAXK) { std::cerr << "The k-mer length should be in [2," << MAXK << "]." << std::endl; exit(1); }
      continue;
    }
    if (val == "-t") {
      need("Please specify the minimum frequency (targets)!");
      minT = (uint32_t)atoi(argv[i]);
      if (minT >= 65536) { std::cerr << "The min k-mer frequency should be in [0,65535]." << std::endl; exit(1); }
      continue;
    }
    if (val == "-n") {
      need("Please specify the number of threads!");
      int c = atoi(argv[i]);
      if (c < 1) { std::cerr << "The number of threads should be higher than 0." << std::endl; exit(1); }
      cpu = (size_t)c;
      if (batches < cpu) batches = cpu;
      continue;
    }
    if (val == "--tsk") { tsk = true; continue; }
    if (val == "--extended") { ext = true; continue; }
    if (val == "--db-sharded") { db_sharded = true; continue; }
    if (val == "--parts") {
      need("Please specify the number of parts of the table!");
      int p = atoi(argv[i]);
      if (p < 1 || p > 64) { std::cerr << "The number of parts should be in [1,64]." << std::endl; exit(1); }
      parts = (size_t)p; db_sharded = true;
      continue;
    }
    if (val == "--light") { light = true; continue; }
    if (val == "--htsize") {
      need("Please specify the table size!");
      htsize_override = strtoull(argv[i], nullptr, 10);
      if (htsize_override < 2) { std::cerr << "The table size should be >= 2." << std::endl; exit(1); }
      continue;
    }
    if (val == "-T") {
      need("Please specify the targets!");
      i_targets = i;
      if (!valid_file(argv[i])) { std::cerr << "Failed to find/read the file of the targets definition: " << argv[i] << std::endl; exit(1); }
      continue;
    }
    if (val == "-O") {
      need("Please specify the objects!");
      i_objects = i;
      if (!valid_file(argv[i])) { std::cerr << "Failed to find/read the filename of objects: " << argv[i] << std::endl; exit(1); }
      continue;
    }
    if (val == "-P") {
      if (i + 2 >= argc) { std::cerr << "Please specify the paired-end reads!" << std::endl; exit(1); }
      i++;
      i_objects = i; i_objects2 = i + 1;
      if (!valid_file(argv[i++])) { std::cerr << "Failed to find/read " << argv[i - 1] << std::endl; exit(1); }
      if (!valid_file(argv[i])) { std::cerr << "Failed to find/read " << argv[i] << std::endl; exit(1); }
      continue;
    }
    if (val == "-D") {
      need("Please specify the database directory!");
      i_folder = i;
      if (!valid_file(argv[i])) { std::cerr << "Failed to find/read the directory:  " << argv[i] << std::endl; exit(1); }
      continue;
    }
    if (val == "-R") { need("Please specify where to store results!"); i_results = i; continue; }
    if (val == "-g") {
      need("Please specify a gap value!");
      gap = (size_t)atoi(argv[i]);
      if (gap < 4) { std::cerr << "The gap value should be >= 4." << std::endl; exit(1); }
      continue;
    }
    if (val == "-s") {
      need("Please specify a sampling factor value!");
      int s = atoi(argv[i]);
      if (s < 2 || s > SFACTORMAX) { std::cerr << "The sampling factor value should be in the interval [2," << SFACTORMAX << "]." << std::endl; exit(1); }
      sfactor = (uint32_t)s;
      continue;
    }
    if (val == "-b") {
      need("Please specify the number of batches!");
      int b = atoi(argv[i]);
      if (b < 1 || (size_t)b < cpu) { std::cerr << "The number of batches should be higher than the number of threads." << std::endl; exit(1); }
      batches = (size_t)b;
      continue;
    }
    if (val == "-d") {
      need("Please specify the number of devices to use!");
      int d = atoi(argv[i]);
      if (d < 1) { std::cerr << "The number of devices should be higher than 0." << std::endl; exit(1); }
      devices = (size_t)d;
      continue;
    }
    std::cerr << "Failed to recognize option: " << val << std::endl;
    exit(1);
  }
  if (light) {  // CuCLARK-l (main.cc:241-249): k is forced to 27, gap defaults to 4, no sampling
    if (gap == 0) gap = 4;
    k = 27;
    sfactor = 1;
  } else {
    gap = 0;
  }
  if (i_targets < 0 || i_folder < 0 || i_objects < 0 || i_results < 0) {
    std::cerr << "Failed to run " << argv[0] << ": at least four  parameters are necessary";
    std::cerr << ": file of targets, directory of database, file of objects, file for results." << std::endl;
    print_usage(argv[0]);
    exit(1);
  }
  o.k = k; o.min_count_t = minT; o.threads = cpu; o.batches = batches; o.devices = devices; o.sampling = sfactor; o.gap = gap;
  o.tsk = tsk; o.extended = ext; o.light = light; o.db_sharded = db_sharded; o.parts = parts;
  if (tsk) std::cerr << "Note: --tsk is accepted for compatibility; the per-target .ht text files (k-mer, count) are not written." << std::endl;
  o.htsize = htsize_override ? htsize_override : (light ? HTSIZE_LIGHT : HTSIZE_FULL);
  o.targets = argv[i_targets];
  o.folder = argv[i_folder];
  if (o.folder.empty() || o.folder.back() != '/') o.folder.push_back('/');
  o.objects = argv[i_objects];
  if (i_objects2 > 0) o.objects2 = argv[i_objects2];
  o.results = argv[i_results];
  mic::Classifier* classifier = nullptr;
  try {
    classifier = new mic::Classifier(o);
    if (i_objects2 > 0) classifier->run_paired(o.objects, o.objects2, o.results);
    else classifier->run(o.objects, o.results);
  } catch (const std::exception& ex) {
    std::cerr << ex.what() << std::endl;
    delete classifier;
    return 1;
  }
  // The results are written and closed.  Taking the engines apart in order - tens of gigabytes of table, staging and pinned slots
  // freed one allocation at a time - took the headline run a second (4.46 s of process for 3.45 s of work); the process ends here and
  // the driver takes everything back at once.  MIC_CLI_ORDERLY_EXIT=1 keeps the orderly teardown (the sanitizer builds run with it).
  std::cout.flush(); std::cerr.flush();
  fflush(nullptr);
  // (under a profiler or a coverage / trace tool - rocprofv3 preloads its library and flushes its output from an exit handler - the
  // orderly road is taken without being asked: a fast exit would lose the tool's output)
  auto tooled = [] {
    for (const char* v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "LLVM_PROFILE_FILE"}) if (getenv(v)) return true;
    const char* pre = getenv("LD_PRELOAD");
    return pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "guardalloc"));      // (sanitizer runs ask with the variable)
  };
  if (!getenv("MIC_CLI_ORDERLY_EXIT") && !tooled()) _exit(0);
  delete classifier;
  return 0;
}
