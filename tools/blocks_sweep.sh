#!/bin/bash
# Headline kernel time against the grid size (blocks per CU; the waves stride over the reads): tools/blocks_sweep.sh [values...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "${@:-64 128 256 512}"; do
  MIC_BLOCKS_PER_CU=$v python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 20 --warmup 3 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks per CU $v:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
