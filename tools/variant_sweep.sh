#!/bin/bash
# Headline kernel time of every measuring build under build/variants/*/libmi_clark_var.so (made on the build box: compiler options
# for mic_kernels.hip) next to the product library's:  tools/variant_sweep.sh [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
STEPS=${1:-30}
run() {
  python3 $R/bench.py $2 --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps $STEPS --warmup 3 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['known_answer']['label_and_count_ok'])"
}
run product ""
for so in $R/build/variants/*/libmi_clark_var.so; do
  MIC_LIB_PATH=$so run $(basename $(dirname $so)) --allow-variant-lib
done
run product ""
