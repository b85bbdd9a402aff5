#!/bin/bash
# Wall time of exe/cuCLARK on a 36 GB-scale database read from files (page cache): tools/time_db_load.sh [kmers] [reads]
set -e
KM=${1:-5700000000}; RD=${2:-4000000}
D=/tmp/bigdb; rm -rf $D; mkdir -p $D
( time python tools/make_synth_files.py $D --reads $RD --kmers $KM --targets 4096 ) 2>&1 | tail -4
du -sh $D/DB
for run in 1 2; do
  echo "== run $run"
  ( time MIC_LOAD_TIMING=1 ./exe/cuCLARK -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out -n 32 -b 32 ) 2>&1 | grep -E "real|Assignment|load\]" 
done
rm -rf $D
