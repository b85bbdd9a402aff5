#!/bin/bash
# Counters of the device inflater's kernels on the bench's FASTQ shape: tools/pmc_gz.sh <tag> [reads]; prints per-output-symbol means of gz_decode_kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-gz}; N=${2:-1000000}
OUT=$R/gpurun_out/pmcgz_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/tools/gz_device_timing.py $N > $OUT/$name.log 2> $OUT/$name.err
done
python3 - <<PY
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"gz_\w+", r["Kernel_Name"]); k=m.group(0) if m else ""
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
n_sym = $N * 316.0
for k, d in agg.items():
    print("$TAG", k, {c: round(sum(v)/len(v)/n_sym, 3) for c, v in sorted(d.items())}, "(per output byte of the text)")
PY
