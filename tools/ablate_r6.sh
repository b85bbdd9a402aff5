#!/bin/bash
# Round-6 ablations of query_kernel_r (MIC_X bits, mic_kernels.hip), each a variant library built ON the GPU box (obj_var does not travel),
# one PMC pass per variant:   tools/ablate_r6.sh "0 7 rc0=-DMIC_T_RCWIN=0" [bench args, e.g. --layout super2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=$1; shift
mkdir -p $R/gpurun_out/abl
for item in $VARS; do
  # an item is a MIC_X value, or name=flags (e.g. rc0=-DMIC_T_RCWIN=0)
  if [[ "$item" == *=* ]]; then x=${item%%=*}; FL=${item#*=}; else x=$item; FL="-DMIC_X=$item"; fi
  rm -rf $R/gpurun_out/abl/p_$x
  make -C $R/cuclark_amd/csrc variant VARIANT_FLAGS="$FL" > $R/gpurun_out/abl/build_$x.log 2>&1 || { echo "build $x failed"; tail -5 $R/gpurun_out/abl/build_$x.log; continue; }
  cp $R/cuclark_amd/csrc/obj_var/libmi_clark_var.so $R/gpurun_out/abl/lib_$x.so
  ( cd /tmp && export TMPDIR=/tmp MIC_LIB_PATH=$R/gpurun_out/abl/lib_$x.so
    echo "variant $x: counters" >> $R/gpurun_out/abl/progress.txt
    timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/abl/p_$x -- python3 $R/bench.py --allow-variant-lib --no-parts-proxy --no-two-strand --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 "$@" > $R/gpurun_out/abl/bench_$x.json 2> $R/gpurun_out/abl/err_$x.txt
    echo "variant $x: timing" >> $R/gpurun_out/abl/progress.txt
    timeout -k 10 200 python3 $R/bench.py --allow-variant-lib --no-parts-proxy --no-two-strand --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 3 "$@" > $R/gpurun_out/abl/time_$x.json 2>> $R/gpurun_out/abl/err_$x.txt )
  python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/abl/p_$x/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {c: sum(q) / len(q) for c, q in agg.items()}
try:
    t = json.load(open("$R/gpurun_out/abl/time_$x.json"))
    ms = t["roofline"]["kernel_ms"]; nr = t["config"]["reads_per_gpu"]
except Exception as e:
    ms, nr = float("nan"), 1e7
print("MIC_X=$x  kernel %.3f ms | per read: VALU %.1f SALU %.1f branches %.1f LDS %.1f" % (ms, v.get("SQ_INSTS_VALU", 0) / nr, v.get("SQ_INSTS_SALU", 0) / nr, v.get("SQ_INSTS_BRANCH", 0) / nr, v.get("SQ_INSTS_LDS", 0) / nr))
PY
rm -f $R/gpurun_out/abl/lib_$x.so
done
