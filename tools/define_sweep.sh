#!/bin/bash
# Rebuild mic_kernels.o on the GPU box with each given -D option and run the headline bench: tools/define_sweep.sh "-DA=1" "-DA=2" ...
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
for d in "$@"; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -I../../include -I. $d -c mic_kernels.hip -o obj/mic_kernels.o 2>&1 | grep -E "error" -A3
  make all 2>&1 | grep -E "error" -A3
  for i in 1 2; do
    python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'])"
  done
done
