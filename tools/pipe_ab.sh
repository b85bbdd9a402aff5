#!/bin/bash
# The pipelined loop of query_kernel_r against the plain one on the same box in one call: the product library and a measuring build
#   (cd cuclark_amd/csrc && make variant VARIANT_FLAGS="-DMIC_R_PIPE=0" && mkdir -p ../../build/variants/nopipe && cp obj_var/libmi_clark_var.so ../../build/variants/nopipe/)
# one-strand table (--layout super) and the headline.  tools/pipe_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py $2 $3 $4 --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['known_answer']['label_and_count_ok'])"; }
run product_super --layout super
MIC_LIB_PATH=$R/build/variants/nopipe/libmi_clark_var.so run nopipe_super --layout super --allow-variant-lib
run product_super --layout super
run product_headline
