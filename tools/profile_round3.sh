#!/bin/bash
# Round-3 profiles, all at one commit: tools/profile_round3.sh <stage>
#   a  headline (two-strand table, query_kernel_r<31, 20, true, false>): kernel trace + stats, all PMC passes, calibrations
#   b  the command line's default table (one-strand, query_kernel_r<31, 20, false, false>) and the table-sharded instantiation
#      (part 0 of 8, query_kernel_r<31, 20, true, true>): kernel trace + the counters that matter
#   c  the command line under the profiler (light table): ingest kernels (line index, record, pack, CSV) and build kernels
# Summaries: python tools/summarize_profile.py <tag>; the CLI traces are copied as they are.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
case "$1" in
a) tools/profile_bench.sh r03s full ;;
b)
  for cfg in "r03d MIC_LAYOUT=super" "r03p MIC_SINGLE_PART=8"; do
    set -- $cfg; TAG=$1; ENVV=$2
    OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
    EXTRA=""; [ "$ENVV" = "MIC_SINGLE_PART=8" ] && EXTRA="--single-part 8" || export $ENVV
    ( cd /tmp && export TMPDIR=/tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 3 $EXTRA > $OUT/kt_bench.json 2> $OUT/kt.err
      for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
        name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
        rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --no-parts-proxy --no-default-layout --no-cpu --no-pipeline --no-e2e --steps 3 --warmup 1 $EXTRA > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
      done )
    unset MIC_LAYOUT
    # the 128-byte-slot calibration of FETCH_SIZE is the headline run's (stage a); summarize_profile.py falls back to scale 2
  done ;;
c)
  D=/tmp/prof_cli; rm -rf $D; mkdir -p $D
  python tools/make_synth_files.py $D --light --reads 16000000 --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
  OUT=$R/gpurun_out/prof_r03cli; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out -n 12 > $OUT/kt.out 2> $OUT/kt.err
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
      name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
      rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out -n 12 > $OUT/pmc_$name.out 2> $OUT/pmc_$name.err
    done )
  ls -la $D/reads.fq | awk '{print $5}' > $OUT/input_bytes.txt
  rm -rf $D ;;
esac
find $R/gpurun_out -name "*_kernel_stats.csv" -newer $R/bench.py | head
du -sh $R/gpurun_out/prof_r03* 2>/dev/null
