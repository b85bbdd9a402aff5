#!/bin/bash
export MIC_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/cuclark_amd/csrc/obj_var/libmi_clark_var.so   # the product library stays as it is
# Per-phase instruction counts of query_kernel_m: rebuild with -DMIC_ABLATE=<n> on the GPU box and read SQ_INSTS_* per read.
R=$GRAFT_REPO_ROOT
cd $R/cuclark_amd/csrc
for v in ${@:-0 1 2 3 6}; do
  make variant VARIANT_FLAGS="-DMIC_ABLATE=$v" 2>&1 | grep -E "error" -A3   # a measuring build: obj_var/, libmi_clark_var.so (csrc/Makefile)
  OUT=$R/gpurun_out/ablate_$v; mkdir -p $OUT
  ( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/err.txt )
  python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); v = d["value"]
except Exception as e:
    v = None
print("ablate $v:", v, {c.replace("SQ_INSTS_", ""): round(sum(x) / len(x) / 1e7, 1) for c, x in sorted(agg.items())})
PY
done
