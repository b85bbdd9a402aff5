#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run, as the
# MI355X guide prescribes).  Outputs under gpurun_out/prof_<tag>/; summaries are copied to profiles/ by hand.
TAG=${1:-r01}
WL=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --workload $WL --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 3 > $OUT/kt_bench.json 2> $OUT/kt.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --workload $WL --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
[ -x $R/build/tools/gather_bench ] || { mkdir -p $R/build/tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/build/tools/gather_bench $R/tools/gather_bench.hip; }
# calibration of FETCH_SIZE on a known byte count in the same access shape (4 lanes x 16 B per 64-B slot)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- $R/build/tools/gather_bench 8 > $OUT/cal_gather.csv 2> $OUT/cal.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/cal_rdreq -- $R/build/tools/gather_bench 8 > /dev/null 2>> $OUT/cal.err
# the same for the 128-byte slots of the minimizer / super-k-mer layouts (8 lanes x 16 B per random 128-B slot): the guide's
# gfx950 rule says a 128-B request is tallied at 64 B; runs_kernel<128, 1, 0> loads a known number of slots
[ -x $R/build/tools/gather_runs_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/build/tools/gather_runs_bench $R/tools/gather_runs_bench.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal128_fetch -- $R/build/tools/gather_runs_bench 16 > $OUT/cal128_gather.csv 2>> $OUT/cal.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/cal128_rdreq -- $R/build/tools/gather_runs_bench 16 > /dev/null 2>> $OUT/cal.err
find $OUT -name "*.csv" | head -50
du -sh $OUT
