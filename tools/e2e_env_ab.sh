#!/bin/bash
# A/B of environment settings of the streaming command line on a synthetic FASTQ (light table): tools/e2e_env_ab.sh <reads> <threads> "ENV=V ENV2=V" ...
# GZPAIRS=1: the input is a pair of gzip -1 files (-P a.fq.gz b.fq.gz) instead of one plain file.
set -e
N=${1:-10000000}; TH=${2:-12}; shift 2
D=/tmp/e2e; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 ${GZPAIRS:+--paired} > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
IN="-O $D/reads.fq"
if [ -n "$GZPAIRS" ]; then gzip -1 -c $D/reads_1.fq > $D/r1.fq.gz; gzip -1 -c $D/reads_2.fq > $D/r2.fq.gz; IN="-P $D/r1.fq.gz $D/r2.fq.gz"; fi
for cfg in "$@"; do
  for rep in 1 2 3; do ( env $cfg MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ $IN -R $D/out -n $TH ) 2>&1 | grep -E "device ingest:|Assignment" | sed -e "s/.*threads: /[$cfg] /" -e 's/input .*ms since start//' -e 's/Speed.*//' | tr '\n' ' '; echo; done
done
