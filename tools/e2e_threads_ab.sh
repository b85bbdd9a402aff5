#!/bin/bash
# The streaming command line at several thread counts: tools/e2e_threads_ab.sh <reads> "<-n> [ENV=V ...]" ...
set -e
N=${1:-10000000}; shift
D=/tmp/e2e; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
for cfg in "$@"; do
  set -- $cfg; TH=$1; shift
  for rep in 1 2 3; do ( env "$@" MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out -n $TH ) 2>&1 | grep -E "device ingest:|Assignment" | sed -e "s/.*threads: /[-n $cfg] /" -e 's/input .*ms since start//' -e 's/Speed.*//' | tr '\n' ' '; echo; done
done
