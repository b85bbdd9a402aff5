#!/bin/bash
# The oracle's C code (oracle/clark_oracle.c, part_rule.c) under AddressSanitizer + UBSan, driven through its ctypes binding by the
# fuzzer's generator, with no product library in the process (tools/sanitize/oracle_rig.py).  CPU only.
#   tools/sanitize/oracle_rig.sh [seconds] [seed]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/tools/sanitize/_build
mkdir -p "$OUT"
gcc -O1 -g -std=gnu99 -fPIC -shared -fopenmp -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -Wall -Wextra -o "$OUT/liboracle_asan.so" "$R/oracle/clark_oracle.c" "$R/oracle/part_rule.c" -lm
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" PYTHONMALLOC=malloc ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ORACLE_LIB_PATH="$OUT/liboracle_asan.so" \
  python3 "$R/tools/sanitize/oracle_rig.py" "${1:-60}" "${2:-1}"
