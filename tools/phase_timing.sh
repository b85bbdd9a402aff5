#!/bin/bash
# Per-phase cycle shares of the default query kernel: rebuild mic_kernels.o with -DMIC_PHASE_TIMING on the GPU box, run the bench.
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -I../../include -I. -DMIC_PHASE_TIMING $PHASE_EXTRA -c mic_kernels.hip -o obj/mic_kernels.o 2>&1 | grep -E "error" -A3
make all 2>&1 | grep -E "error" -A3
python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 2>&1 | grep -E "phase cycles" | tail -3
