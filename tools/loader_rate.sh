#!/bin/bash
# Host-side microbenchmark of the streaming command line's loaders (no device work): a FASTQ file of 150-bp reads in the page
# cache, read in ranges by N threads and stripped of its '+' / quality lines (exe/cuCLARK --strip-fastq ... loaders).
#   tools/loader_rate.sh [GB]
R=${GRAFT_REPO_ROOT:-$(pwd)}
GB=${1:-3}
F=/tmp/loader_rate_$$.fq
python3 - "$F" "$GB" <<'PY'
import sys, numpy as np
path, gb = sys.argv[1], float(sys.argv[2])
rng = np.random.default_rng(1)
n = 200000
seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
buf = bytearray()
for i in range(n):
    buf += b"@read_%09d/1\n" % i + seqs[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n"
with open(path, "wb") as f:
    for _ in range(int(gb * 1e9 / len(buf)) + 1):
        f.write(buf)
PY
cat $F > /dev/null
for th in 8 12 16; do
  $R/exe/cuCLARK --strip-fastq $F - 1048576 loaders $th | tail -1
  MIC_STRIP_SCALAR=1 $R/exe/cuCLARK --strip-fastq $F - 1048576 loaders $th | tail -1 | sed 's/^/scalar /'
  $R/exe/cuCLARK --strip-fastq $F - 1048576 loaders $th mmap | tail -1
  $R/exe/cuCLARK --strip-fastq $F - 262144 loaders $th | tail -1
done
rm -f $F
