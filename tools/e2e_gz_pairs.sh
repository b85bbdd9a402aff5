#!/bin/bash
# The command line on a pair of gzip -1 FASTQ files (1 M pairs, light table) with its timelines: MIC_CLI_TIMING, MIC_GZ_TIMING.
#   tools/e2e_gz_pairs.sh [pairs] [threads]
N=${1:-1000000}; TH=${2:-12}
D=/tmp/e2egz; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 --paired > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
gzip -1 -c $D/reads_1.fq > $D/r1.fq.gz; gzip -1 -c $D/reads_2.fq > $D/r2.fq.gz
for rep in 1 2 3; do
  MIC_CLI_TIMING=1 MIC_GZ_TIMING=${GZT:-0} ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/r1.fq.gz $D/r2.fq.gz -R $D/out -n $TH 2>&1 | grep -E "gz\]|device inflate|device ingest|Assignment|pairs" | cut -c1-400
done
