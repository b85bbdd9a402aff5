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
  echo "== $f"
  MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/$f -R $D/out_$f -n 12 2> $D/err.txt | grep -E "Assignment"
  grep -E "^\[timing\]" $D/err.txt | grep -v "thread-seconds" | cut -c1-260
done
cmp $D/out_reads.fq.csv $D/out_reads.fq.gz.csv && echo "CSVs identical"
rm -rf $D
