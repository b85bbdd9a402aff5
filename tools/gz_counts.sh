#!/bin/bash
# One PMC pass over tools/gz_device_timing.py: instruction counts and busy figures of the gzip kernels (per launch).
#   tools/gz_counts.sh <tag>
TAG=${1:-gz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gzc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/p -- python3 $R/tools/gz_device_timing.py > $OUT/log.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gz_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    v = {c: sum(x) / len(x) for c, x in d.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(k, "| ms %.3f (2.4 GHz) | M instr: VALU %.1f SALU %.1f LDS %.1f branch %.1f | per CU cycle: VALU busy %.1f %% scalar busy %.1f %% | wait-LDS %.0f M" % (
        cyc / 2.4e6, v["SQ_INSTS_VALU"] / 1e6, v["SQ_INSTS_SALU"] / 1e6, v["SQ_INSTS_LDS"] / 1e6, v.get("SQ_INSTS_BRANCH", 0) / 1e6,
        100 * v["SQ_ACTIVE_INST_VALU"] / (cyc * 256 * 4), 100 * v["SQ_ACTIVE_INST_SCA"] / (cyc * 256), v["SQ_WAIT_INST_LDS"] / 1e6))
PY
