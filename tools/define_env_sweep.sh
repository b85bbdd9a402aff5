#!/bin/bash
export MIC_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/cuclark_amd/csrc/obj_var/libmi_clark_var.so   # the product library stays as it is
# Rebuild mic_kernels.o with each -D set and run the headline bench under each environment setting:
#   tools/define_env_sweep.sh "ENV1|ENV2" "-DA=1" "-DA=2" ...
cd $GRAFT_REPO_ROOT/cuclark_amd/csrc
IFS='|' read -ra ENVS <<< "$1"; shift
for d in "$@"; do
  make variant VARIANT_FLAGS="$d" 2>&1 | grep -E "error" -A3   # a measuring build: obj_var/, libmi_clark_var.so (csrc/Makefile)
  for e in "${ENVS[@]}"; do for i in 1 2; do
    env $e python $GRAFT_REPO_ROOT/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', '$e', d['value'], d['ms_per_step'])"
  done; done
done
