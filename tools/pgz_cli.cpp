// pgz_cli.cpp - command-line driver of the parallel gzip inflater (cuclark_amd/csrc/pgz.hpp) for its tests and timings:
//   pgz_cli <in.gz> <out | -> [threads [chunk_bytes]]      exit 0 ok, 1 damaged input (what zlib would report), 2 usage / io
#include <chrono>
#include <string>

#include "pgz.hpp"

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.gz out|- [threads [chunk_bytes]]\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> gz;
  fseek(f, 0, SEEK_END);
  gz.resize((size_t)ftell(f));
  rewind(f);
  if (fread(gz.data(), 1, gz.size(), f) != gz.size()) return 2;
  fclose(f);
  const unsigned th = argc > 3 ? (unsigned)atoi(argv[3]) : 8;
  const size_t cb = argc > 4 ? (size_t)atol(argv[4]) : ((size_t)1 << 20);
  FILE* o = std::string(argv[2]) == "-" ? nullptr : fopen(argv[2], "wb");
  size_t total = 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto sink = pgz::piece_sink([&](pgz::Bytes&& b) {
    total += b.n;
    return !o || fwrite(b.p, 1, b.n, o) == b.n;
  });
  const int rc = pgz::inflate_all(gz.data(), gz.size(), th, cb, sink);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (o) fclose(o);
  fprintf(stderr, "rc=%d %zu bytes in %.3f s = %.1f MB/s on %u threads\n", rc, total, dt, total / dt / 1e6, th);
  return rc == 0 ? 0 : 1;
}
