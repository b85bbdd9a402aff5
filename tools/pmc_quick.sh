#!/bin/bash
# A quick counter comparison of the headline kernel: tools/pmc_quick.sh <tag> [ENV=VALUE ...]; prints per-read means
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pmcq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$TAG", {k: round(sum(v)/len(v)/1e7,2) for k,v in sorted(agg.items())})
PY
