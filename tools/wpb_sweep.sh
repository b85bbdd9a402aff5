#!/bin/bash
# Waves per block of query_kernel_m: rebuild mic_kernels.o with -DMIC_M_WPB=<n> on the GPU box and run the bench.
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
for w in ${WPBS:-1 2 4}; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -I../../include -I. -DMIC_M_WPB=$w -c mic_kernels.hip -o obj/mic_kernels.o 2>&1 | grep -E "error" -A3
  make all 2>&1 | grep -E "error" -A3
  for b in 512 1024; do
    MIC_BLOCKS_PER_CU=$b python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pipeline --no-e2e --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wpb', $w, 'blocks/cu', $b, d['value'], d['ms_per_step'])"
  done
done
