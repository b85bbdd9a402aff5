#!/bin/bash
# Quick PMC passes for the query kernel of the current default layout: tools/profile_pmc_quick.sh <tag> [extra env]
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/p$i.json 2> $OUT/p$i.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:34s} {sum(v)/len(v):.4g}")
PY
