#!/bin/bash
# Headline bench under each given environment setting (no rebuild): tools/env_sweep.sh "A=1" "A=2 B=3" ...
for e in "$@"; do
  env $e python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['table']; print('$e', d['value'], d['ms_per_step'], 'GB', t['hbm_GB'], 'chain', t['overflow_slots'], 'max', t['largest_minimizer_bucket'], 'm', t['minimizer_len'], 'ok', d['known_answer']['label_and_count_ok'])"
done
