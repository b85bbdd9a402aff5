#!/bin/bash
# Round-6 profiles at one commit (each rocprofv3 run a process of its own, counters in passes of their own):
#   r06a  the headline = the command line's table (one strand, query_kernel_r<31, 20, false, false>): kernel trace + stats, PMC passes, calibrations
#   r06s  the two-strand table (query_kernel_r<31, 20, true, false>): kernel trace + stats, PMC passes
#   r06cli  exe/cuCLARK on the headline files under the kernel trace: build, ingest and query kernels of the product run
# Summaries: python tools/summarize_profile.py r06a / r06s.     tools/profile_round6.sh [s|d|cli ...]   (s = r06a, d = r06s)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
PMCS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA")
prof() {   # prof <tag> <bench args...>
  TAG=$1; shift
  OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 20 --warmup 3 "$@" > $OUT/kt_bench.json 2> $OUT/kt.err
    echo "$TAG kernel trace done"
    for pmc in "${PMCS[@]}"; do
      name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
      rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 "$@" > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
      echo "$TAG pmc $name done"
    done )
}
for stage in "${@:-s d cli}"; do
case "$stage" in
s) prof r06a
   mkdir -p $R/build/tools
   [ -x $R/build/tools/gather_runs_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/build/tools/gather_runs_bench $R/tools/gather_runs_bench.hip
   ( cd /tmp && export TMPDIR=/tmp
     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06a/cal128_fetch -- $R/build/tools/gather_runs_bench 16 > $R/gpurun_out/prof_r06a/cal128_gather.csv 2>> $R/gpurun_out/prof_r06a/cal.err
     rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06a/cal128_rdreq -- $R/build/tools/gather_runs_bench 16 > /dev/null 2>> $R/gpurun_out/prof_r06a/cal.err ) ;;
d) prof r06s --layout super2 ;;
cli)
  OUT=$R/gpurun_out/prof_r06cli; mkdir -p $OUT
  MIC_BENCH_KEEP=1 MIC_BENCH_NO_FASTA=1 python3 $R/bench.py --steps 2 --warmup 1 --no-multi-engine --no-parts-proxy --no-default-layout --e2e-reps 1 --no-cpu --no-pipeline --time-budget 100000 > $OUT/bench.json 2> $OUT/bench.err
  D=$(grep -o "files kept in .*" $OUT/bench.err | awk '{print $4}')
  sleep 30
  ( cd /tmp && export TMPDIR=/tmp
    MIC_CLI_ORDERLY_EXIT=1 MIC_LOAD_TIMING=1 MIC_CLI_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $R/exe/cuCLARK -k 31 --htsize 1610612741 -T $D/targets.txt -D $D/DB -O $D/reads_1.fq -R $D/out_p -n 12 > $OUT/kt.out 2> $OUT/kt.err )
  rm -rf $D ;;
esac
done
find $R/gpurun_out/prof_r05* -name "*_kernel_stats.csv" | head
du -sh $R/gpurun_out/prof_r05*
