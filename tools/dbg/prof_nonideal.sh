#!/bin/bash
# kernel trace of the non-ideal cases: per-kernel time of query_kernel_r and crowd_finish_kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ni -o ni -- python3 tools/nonideal_bench.py --cases ${1:-repeats5} --layouts ${2:-auto,super2} > gpurun_out/nonideal_prof.json 2> gpurun_out/nonideal_prof.err
f=$(find gpurun_out/prof_ni -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-220
