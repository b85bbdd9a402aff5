#!/bin/bash
export MIC_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/cuclark_amd/csrc/obj_var/libmi_clark_var.so   # the product library stays as it is
# Waves per block of query_kernel_m: rebuild mic_kernels.o with -DMIC_M_WPB=<n> on the GPU box and run the bench.
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
for w in ${WPBS:-1 2 4}; do
  make variant VARIANT_FLAGS="-DMIC_M_WPB=$w" 2>&1 | grep -E "error" -A3   # a measuring build: obj_var/, libmi_clark_var.so (csrc/Makefile)
  for b in 512 1024; do
    MIC_BLOCKS_PER_CU=$b python $GRAFT_REPO_ROOT/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wpb', $w, 'blocks/cu', $b, d['value'], d['ms_per_step'])"
  done
done
