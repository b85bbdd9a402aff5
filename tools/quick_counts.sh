#!/bin/bash
# One PMC pass over the headline bench: instructions per read and busy figures of the query kernel.
#   tools/quick_counts.sh <tag> [ENV=VALUE ...]      (set MIC_LIB_PATH for a variant library)
TAG=${1:-q}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qc_$TAG
mkdir -p $OUT
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --allow-variant-lib --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
v = {c: sum(x) / len(x) for c, x in agg.items()}
cyc = v["GRBM_GUI_ACTIVE"] / 8
print("$TAG", name.split("(")[0][-40:], "ms %.3f" % (cyc / 2.4e6), "(at 2.4 GHz) | per read: VALU %.1f SALU %.1f LDS %.1f | VALU busy %.1f %% | LDS active %.1f %% of CU cycles, conflicts %.1f %%, wait-LDS/read %.0f" % (
    v["SQ_INSTS_VALU"] / 1e7, v["SQ_INSTS_SALU"] / 1e7, v["SQ_INSTS_LDS"] / 1e7, 100 * v["SQ_ACTIVE_INST_VALU"] / (cyc * 256),
    100 * v["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 100 * v["SQ_LDS_BANK_CONFLICT"] / (cyc * 256), v["SQ_WAIT_INST_LDS"] / 1e7))
PY
