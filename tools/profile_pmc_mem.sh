#!/bin/bash
# Memory-pipeline PMC passes for the query kernel: tools/profile_pmc_mem.sh <tag>   (one small counter set per pass,
# each under its own timeout: a pass with TA_* counters once sat until gpurun's limit)
TAG=${1:-mem}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCC_BUSY_avr TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/p$i.json 2> $OUT/p$i.err
  grep -iE "error|invalid|not" $OUT/p$i.err | head -2 | cut -c1-160
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:44s} {sum(v)/len(v):.5g}")
PY
