#!/bin/bash
# Where a run on ONE gzip FASTQ file spends its assignment time: exe/cuCLARK-l with its timing lines on, plain vs gzip.
#   tools/gz_phases.sh [reads]
N=${1:-4000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/gzp; rm -rf $D; mkdir -p $D
python3 $R/tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
gzip -1 -k $D/reads.fq
ls -la $D/reads.fq $D/reads.fq.gz | awk '{print $5, $9}'
for f in reads.fq reads.fq.gz reads.fq.gz; do
  echo "== $f (gzip: the whole member first, MIC_GZ_STRIPES=1)"
  MIC_GZ_STRIPES=1 MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/$f -R $D/out_$f -n 12 2> $D/err.txt | grep -E "Assignment"
  grep -E "^\[timing\]" $D/err.txt | grep -v "thread-seconds" | cut -c1-260
done
for S in "" "" 8 3 2; do
  echo "== reads.fq.gz in stripes (MIC_GZ_STRIPES=${S:-default})"
  MIC_GZ_STRIPES=$S MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq.gz -R $D/out_s$S -n 12 2> $D/err.txt | grep -E "Assignment"
  grep -E "^\[timing\]|Note" $D/err.txt | grep -v "thread-seconds" | cut -c1-260
  cmp $D/out_reads.fq.csv $D/out_s$S.csv && echo "CSV identical to the plain file's"
done
echo "== reads.fq.gz, the inflater's phases (MIC_GZ_TIMING: a wait behind every phase)"
MIC_GZ_STRIPES=1 MIC_GZ_TIMING=1 MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq.gz -R $D/out_t -n 12 2>&1 | grep -E "^\[gz\]|Assignment" | cut -c1-200
cmp $D/out_reads.fq.csv $D/out_reads.fq.gz.csv && echo "CSVs identical"
rm -rf $D
