#!/bin/bash
# C++-ONLY soak of the product: exe/cuCLARK on the HARDENED build of the library (libstdc++ assertions, fortified
# libc, stack protectors; glibc's heap checks on) over random inputs - FASTA, FASTQ, gzip, block gzip, pairs, compressed pairs -
# through its modes: one engine, -d N read-sharded, --db-sharded with 2..4 parts, 2-D, small ingest slots, thread counts, the host
# ingest path.  Every CSV is compared with `cmp` against the one-engine run of the same input.  No Python in the process under test:
# the inputs are written by tools/cli_soak_gen.py in a process of its own.  Stops at the first difference or abnormal exit.
#   tools/cli_soak.sh [seconds] [seed] [workdir]
# GUARD=1: every run of the command under tools/sanitize/guardalloc.c (LD_PRELOAD: every host allocation of 256 B .. 64 MiB on pages of
# its own, ending at an inaccessible page - a store past a block's end faults at the store, with the writer's stack in last.err) instead
# of glibc's heap checks; the orderly exit (MIC_CLI_ORDERLY_EXIT) so that every block is also freed through the canary check.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SECS=${1:-120}; SEED=${2:-1}; W=${3:-/tmp/cli_soak_$$}
mkdir -p "$W/hard" "$W/DB"
ln -sf "$R/cuclark_amd/lib/libmi_clark_hard.so" "$W/hard/libmi_clark.so"
export LD_LIBRARY_PATH="$W/hard:$LD_LIBRARY_PATH"
EXE="$R/exe/cuCLARK"
GUARD_ENV=()
if [ -n "$GUARD" ]; then
  gcc -O2 -g -fPIC -shared -o "$W/guardalloc.so" "$R/tools/sanitize/guardalloc.c" -ldl -lpthread || exit 2
  GUARD_ENV=(LD_PRELOAD="$W/guardalloc.so" GUARD_REPORT=1 MIC_CLI_ORDERLY_EXIT=1 GUARD_LIVE_MAX=${GUARD_LIVE_MAX:-28000})
else
  export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
fi
guarded=0
python3 "$R/tools/cli_soak_gen.py" db "$W" "$SEED" || exit 2
runs=0; rounds=0; t_end=$(( $(date +%s) + SECS ))
run() {   # run <csv base> <env assignments...> -- <args...>
  local out=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${GUARD_ENV[@]}" "${envs[@]}" "$EXE" -k 31 --htsize 999983 -T "$W/targets.txt" -D "$W/DB" -R "$out" "$@" > "$W/last.out" 2> "$W/last.err"
  local rc=$?
  runs=$((runs + 1))
  if [ -n "$GUARD" ]; then
    g=$(grep -o "guarded [0-9]* allocations" "$W/last.err" | grep -o "[0-9]*" | head -1); guarded=$((guarded + ${g:-0}))
    if grep -q "\[guardalloc\] \(signal\|free\)" "$W/last.err"; then
      echo "cli soak: guardalloc report (round $rounds, seed $((SEED + rounds))): ${envs[*]} $EXE ... $*"; grep -A30 "\[guardalloc\]" "$W/last.err" | head -60
      cp "$W/last.err" "$R/gpurun_out/cli_soak_guard_report.err" 2>/dev/null
      exit 1
    fi
  fi
  if [ $rc -ne 0 ]; then
    echo "cli soak: exit code $rc (round $rounds, seed $((SEED + rounds))): ${envs[*]} $EXE ... $*"; tail -5 "$W/last.err"
    cp "$W/last.err" "$R/gpurun_out/cli_soak_failure.err" 2>/dev/null
    exit 1
  fi
}
same() {  # same <csv base> <reference csv base> <what>
  if ! cmp -s "$1.csv" "$2.csv"; then
    echo "cli soak: CSV differs from the one-engine run (round $rounds, seed $((SEED + rounds))): $3"
    mkdir -p "$R/gpurun_out/cli_soak_diff" && cp "$1.csv" "$2.csv" "$W"/in.* "$W"/p?.fq* "$R/gpurun_out/cli_soak_diff/" 2>/dev/null
    exit 1
  fi
}
while [ "$(date +%s)" -lt "$t_end" ]; do
  python3 "$R/tools/cli_soak_gen.py" reads "$W" $((SEED + rounds)) || exit 2
  RND=$(( (SEED + rounds) * 7919 ))
  NT=$(( 1 + RND % 12 )); KB=$(( 16 << (RND % 8) )); P=$(( 2 + RND % 3 )); E=$(( 2 + (RND / 3) % 3 ))
  for inp in "-O $W/in.fa" "-O $W/in.fq" "-P $W/p1.fq $W/p2.fq" "-O $W/in.fq.gz" "-P $W/p1.fq.gz $W/p2.fq.gz" "-O $W/in.fq.bgz" "-P $W/p1.fq.bgz $W/p2.fq.bgz"; do
    run "$W/ref" -- $inp -n 4
    run "$W/v1" MIC_INGEST_KB=$KB -- $inp -n $NT;                                        same "$W/v1" "$W/ref" "slots of $KB KB, -n $NT: $inp"
    run "$W/v2" MIC_SHARD_ENGINES=$E -- $inp -n $NT -d $E;                                same "$W/v2" "$W/ref" "-d $E: $inp"
    run "$W/v3" MIC_SHARD_ENGINES=$P MIC_INGEST_KB=$KB -- $inp -n 6 --db-sharded --parts $P;   same "$W/v3" "$W/ref" "--db-sharded --parts $P: $inp"
    run "$W/v4" MIC_SHARD_ENGINES=4 -- $inp -n 5 --db-sharded --parts 2;                  same "$W/v4" "$W/ref" "--db-sharded --parts 2 on 4 engines: $inp"
    run "$W/v5" MIC_HOST_INGEST=1 -- $inp -n 3 -b 3;                                      same "$W/v5" "$W/ref" "host ingest: $inp"
    if [ $(( (RND / 7) % 3 )) -eq 0 ]; then
      run "$W/v6" MIC_GZ_HOST=1 MIC_LAYOUT=super2 -- $inp -n $NT;                         same "$W/v6" "$W/ref" "two-strand table, host inflate: $inp"
    fi
  done
  rounds=$((rounds + 1))
  echo "... round $rounds done, $runs runs, $(( t_end - $(date +%s) )) s left"
done
echo "cli soak ok: $rounds rounds of 7 inputs, $runs runs of exe/cuCLARK on the hardened library${GUARD:+ under guard pages ($guarded allocations guarded)}, every CSV equal to the one-engine run's (seeds $SEED..$((SEED + rounds - 1)))"
rm -rf "$W"
