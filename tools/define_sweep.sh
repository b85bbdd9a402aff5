#!/bin/bash
# Rebuild mic_kernels.hip with each given -D option as a MEASURING build (csrc/Makefile: make variant -> obj_var/,
# csrc/obj_var/libmi_clark_var.so; the product library is untouched) and run the headline bench on it:
#   tools/define_sweep.sh "-DA=1" "-DA=2" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MIC_LIB_PATH=$R/cuclark_amd/csrc/obj_var/libmi_clark_var.so
for d in "$@"; do
  make -C $R/cuclark_amd/csrc variant VARIANT_FLAGS="$d" 2>&1 | grep -E "error" -A3
  for i in 1 2; do
    python $R/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
