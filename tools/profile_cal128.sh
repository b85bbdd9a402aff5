#!/bin/bash
# Only the 128-byte-slot calibration of tools/profile_bench.sh, into an existing gpurun_out/prof_<tag>/ directory.
TAG=${1:-r01s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/build/tools
cd /tmp && export TMPDIR=/tmp
[ -x $R/build/tools/gather_runs_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/build/tools/gather_runs_bench $R/tools/gather_runs_bench.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal128_fetch -- $R/build/tools/gather_runs_bench 16 > $OUT/cal128_gather.csv 2>> $OUT/cal.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/cal128_rdreq -- $R/build/tools/gather_runs_bench 16 > /dev/null 2>> $OUT/cal.err
python3 - <<PY
import csv, glob, collections
for name in ("cal128_fetch", "cal128_rdreq"):
    for f in glob.glob("$OUT/" + name + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "runs_kernel" in k[0]: print(k, v[-1])
PY
