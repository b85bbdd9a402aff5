#!/bin/bash
# the product under a guard-page allocator (tools/sanitize/guardalloc.c; CPU side only - no GPU sanitizer on this pool).
#   1. exe/cuCLARK alone (no Python in the process): tools/cli_soak.sh with GUARD=1 - EVERY host allocation of 256 B .. 64 MiB guarded;
#   2. the split-process fuzzer's PRODUCT side (Python + ctypes + the hardened library; the oracle is a child without the preload):
#      one allocation in GUARD_SAMPLE guarded (a Python process has more live blocks than the kernel allows mappings), faulthandler on.
# A store past a guarded block faults at the store: the report (native backtrace, Python stack) lands in gpurun_out/guard_soak/.
#   tools/guard_soak.sh [seconds for the command line] [seconds for the fuzzer] [seed]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S1=${1:-600}; S2=${2:-600}; SEED=${3:-1}
OUT=$R/gpurun_out/guard_soak; mkdir -p $OUT
ulimit -c unlimited 2>/dev/null
gcc -O2 -g -fPIC -shared -o $OUT/guardalloc.so $R/tools/sanitize/guardalloc.c -ldl -lpthread || exit 2
echo "guard soak: command line, $S1 s" | tee -a $OUT/log.txt
( cd $OUT && GUARD=1 $R/tools/cli_soak.sh $S1 $SEED $OUT/cli_work ) 2>&1 | tee -a $OUT/log.txt | grep -v "^\.\.\. round" ; rc1=${PIPESTATUS[0]}
echo "guard soak: fuzzer product side, $S2 s, one allocation in ${GUARD_SAMPLE:-16} guarded" | tee -a $OUT/log.txt
( cd $OUT && LD_PRELOAD=$OUT/guardalloc.so GUARD_SAMPLE=${GUARD_SAMPLE:-16} GUARD_REPORT=1 GUARD_LIVE_MAX=26000 MIC_LIB_PATH=$R/cuclark_amd/lib/libmi_clark_hard.so \
    python3 -X faulthandler $R/tools/fuzz_parity.py $S2 $SEED --split ) > $OUT/fuzz.out 2> $OUT/fuzz.err; rc2=$?
tail -3 $OUT/fuzz.out | tee -a $OUT/log.txt
grep "\[guardalloc\]" $OUT/fuzz.err | tail -5 | tee -a $OUT/log.txt
ls core* $OUT/core* 2>/dev/null | head -3
echo "guard soak: command line rc=$rc1, fuzzer rc=$rc2" | tee -a $OUT/log.txt
rm -rf $OUT/cli_work $OUT/guardalloc.so
[ $rc1 -eq 0 ] && [ $rc2 -eq 0 ]
