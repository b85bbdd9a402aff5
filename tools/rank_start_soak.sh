#!/bin/bash
# Looks for the rank-start hang: `bench.py --gpus 2 --backend gloo --workload tiny` N times on one GPU, the
# ranks' watchdog at 60 s (MIC_BENCH_WATCHDOG: every thread's Python stack, then exit).  Prints what it saw; the stacks of a hang go
# to gpurun_out/rank_start_hang_stacks.log.
#   tools/rank_start_soak.sh [runs] [mode]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-30}; MODE=${2:-read}
mkdir -p "$R/gpurun_out"
ok=0; hung=0; failed=0
for i in $(seq 1 "$N"); do
  t0=$(date +%s%N)
  MIC_BENCH_WATCHDOG=60 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29600 + i % 7)) "$R/bench.py" --gpus 2 --steps 2 --warmup 1 --workload tiny --mode "$MODE" --backend gloo --no-db-leg \
    > /tmp/rank_soak.out 2> /tmp/rank_soak.err
  rc=$?
  t1=$(date +%s%N)
  if [ $rc -eq 0 ] && grep -q '"n_gpus": 2' /tmp/rank_soak.out; then ok=$((ok + 1));
  elif grep -q "Timeout (" /tmp/rank_soak.err || [ $rc -eq 124 ]; then
    hung=$((hung + 1)); { echo "--- run $i rc=$rc"; cat /tmp/rank_soak.err; } >> "$R/gpurun_out/rank_start_hang_stacks.log"
  else failed=$((failed + 1)); { echo "--- run $i rc=$rc (failure, not a hang)"; tail -40 /tmp/rank_soak.err; } >> "$R/gpurun_out/rank_start_hang_stacks.log"; fi
  echo "run $i: rc=$rc $(( (t1 - t0) / 1000000 )) ms"
done
echo "rank start soak: $N runs, $ok ok, $hung hung (watchdog), $failed failed otherwise"
