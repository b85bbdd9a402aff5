#!/bin/bash
export MIC_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/cuclark_amd/csrc/obj_var/libmi_clark_var.so   # the product library stays as it is
# Per-phase cycle shares of the default query kernel: rebuild mic_kernels.o with -DMIC_PHASE_TIMING on the GPU box, run the bench.
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
make variant VARIANT_FLAGS="-DMIC_PHASE_TIMING $PHASE_EXTRA" 2>&1 | grep -E "error" -A3   # a measuring build: obj_var/, libmi_clark_var.so (csrc/Makefile)
python $GRAFT_REPO_ROOT/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 2>&1 | grep -E "phase cycles" | tail -3
