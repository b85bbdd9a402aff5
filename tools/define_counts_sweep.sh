#!/bin/bash
# tools/quick_counts.sh on measuring builds of mic_kernels.hip: tools/define_counts_sweep.sh "-DA=1" "-DA=2 -DB=1" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for d in "$@"; do
  i=$((i+1))
  make -C $R/cuclark_amd/csrc variant VARIANT_FLAGS="$d" 2>&1 | grep -E "error" -A3
  echo "== $d"
  MIC_LIB_PATH=$R/cuclark_amd/csrc/obj_var/libmi_clark_var.so bash $R/tools/quick_counts.sh v$i
done
