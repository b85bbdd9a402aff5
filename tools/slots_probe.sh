#!/bin/bash
# the command line's query kernel summed over a 10 M-read run, by the size of its ingest slots (MIC_INGEST_MB), with the
# run's assignment time and process wall beside it.  tools/slots_probe.sh [MB ...]      (on a GPU box; ~4 minutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/slots_probe; mkdir -p $OUT
MIC_BENCH_KEEP=1 MIC_BENCH_NO_FASTA=1 timeout -k 10 400 python3 $R/bench.py --steps 2 --warmup 1 --no-multi-engine --no-parts-proxy --no-default-layout --e2e-reps 1 --no-cpu --no-pipeline --time-budget 100000 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
D=$(grep -o "files kept in .*" $OUT/bench.err | awk '{print $4}')
CMD="$R/exe/cuCLARK -k 31 --htsize 1610612741 -T $D/targets.txt -D $D/DB -O $D/reads_1.fq -R $D/out_p -n 12"
for MB in ${@:-64 128}; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    MIC_INGEST_MB=$MB MIC_CLI_TIMING=1 timeout -k 10 120 $CMD > $OUT/run_${MB}_$rep.out 2> $OUT/run_${MB}_$rep.err
    t1=$(date +%s.%N)
    echo "MB=$MB rep=$rep: $(grep -o 'Assignment time: [0-9.]* s' $OUT/run_${MB}_$rep.out) wall $(python3 -c "print(round($t1 - $t0, 2))") s $(grep -o 'device ingest: [0-9]* batches of <= [0-9]* KB on [0-9]* slot' $OUT/run_${MB}_$rep.err)"
  done
  ( cd /tmp && export TMPDIR=/tmp MIC_INGEST_MB=$MB MIC_CLI_ORDERLY_EXIT=1
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$MB -- $CMD > $OUT/kt_$MB.out 2> $OUT/kt_$MB.err )
  f=$(find $OUT/kt_$MB -name "*kernel_stats.csv" | head -1)
  echo "MB=$MB query kernel: $(grep query_kernel_r $f | awk -F, '{n=NF; print "calls " $(n-6) ", total " $(n-5)/1e6 " ms, average " $(n-4)/1e3 " us"}')"
done
rm -rf $D
