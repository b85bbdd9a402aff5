#!/bin/bash
# every [timing] line of a run on one gzip FASTQ file and on the plain file: tools/dbg/gz_ingest_timing.sh [reads]
N=${1:-4000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/gzp; rm -rf $D; mkdir -p $D
python3 $R/tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
gzip -1 -k $D/reads.fq
for f in reads.fq reads.fq.gz reads.fq.gz; do
  echo "== $f"
  MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/$f -R $D/out_$f -n 12 2>&1 | grep -E "^\[timing\]|Assignment" | cut -c1-600
  ls -la $D/out_$f.csv | awk '{print "csv bytes", $5}'
done
rm -rf $D
