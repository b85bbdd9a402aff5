#!/bin/bash
# Paired-end gzip input through exe/cuCLARK-l -P a.fq.gz b.fq.gz (BASELINE config 5's named input), with the stage times:
#   tools/gz_pairs_timing.sh [pairs] ;  THREADS="8 16 32" to sweep MIC_INFLATE_THREADS
set -e
N=${1:-1000000}
D=/tmp/gzp; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 --paired > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
for i in 1 2; do gzip -1 -k $D/reads_$i.fq; done
ls -la $D/reads_1.fq $D/reads_1.fq.gz | awk '{print $5, $9}'
( time zcat $D/reads_1.fq.gz > /dev/null ) 2>&1 | grep real
g++ -O2 -std=c++17 -Icuclark_amd/csrc -o $D/pgz_cli tools/pgz_cli.cpp -lz -lpthread
for t in 1 8 16 32 64; do $D/pgz_cli $D/reads_1.fq.gz - $t 1048576; done
echo "== plain"
MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq $D/reads_2.fq -R $D/plain -n 12 2>&1 | grep -E "Assignment|paired-end files"
for t in ${THREADS:-0}; do
  echo "== gzip, MIC_INFLATE_THREADS=$t"
  if [ "$t" = 0 ]; then unset MIC_INFLATE_THREADS; else export MIC_INFLATE_THREADS=$t; fi
  for rep in 1 2; do MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq.gz $D/reads_2.fq.gz -R $D/gz -n 12 2>&1 | grep -E "Assignment|inflate|paired-end files|device ingest" | sed 's/thread-seconds.*ms since start/.. ms since start/'; done
done
cmp $D/plain.csv $D/gz.csv && echo "CSVs identical"
rm -rf $D
