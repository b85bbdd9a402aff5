#!/bin/bash
# Where the command line's wall time goes at the headline size: bench.py writes the 36 GB-scale database and the 10 M-read FASTQ
# (MIC_BENCH_KEEP), then exe/cuCLARK runs alone - nothing else on the GPU - with its timing lines and a timestamp per line.
#   tools/cli_wall_timeline.sh [out_dir]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/r5}
mkdir -p "$OUT"
MIC_BENCH_KEEP=1 MIC_BENCH_NO_FASTA=1 python3 "$R/bench.py" --steps 2 --warmup 1 --no-multi-engine --no-parts-proxy --no-default-layout --e2e-reps 1 --no-cpu --no-pipeline \
  > "$OUT/cli_wall_bench.json" 2> "$OUT/cli_wall_bench.err"
D=$(grep -o "files kept in .*" "$OUT/cli_wall_bench.err" | awk '{print $4}')
echo "files in $D"
sleep 45     # (the driver wipes what the bench freed: ~40 GB/s)
for i in 1 2; do
  echo "== run $i"
  s=$(date +%s.%N)
  MIC_LOAD_TIMING=1 MIC_CLI_TIMING=1 "$R/exe/cuCLARK" -k 31 --htsize 1610612741 -T "$D/targets.txt" -D "$D/DB" -O "$D/reads_1.fq" -R "$D/out_t$i" -n 12 2>&1 | \
    while IFS= read -r line; do printf "%8.3f %s\n" "$(echo "$(date +%s.%N) - $s" | bc -l 2>/dev/null || python3 -c "import time;print(time.time()-$s)")" "$line"; done
  e=$(date +%s.%N)
  python3 -c "print('   process wall %.2f s' % ($e - $s))"
  sleep 60
done
rm -rf "$D"
