#!/bin/bash
# End-to-end timing of exe/cuCLARK-l on a synthetic FASTQ: tools/e2e_cli_timing.sh [reads] [threads...]
# Each thread count runs the device-ingest path (default) and, with HOST=1, the host indexer/packer path as well.
set -e
N=${1:-16000000}; shift || true
TH=${@:-32}
D=/tmp/e2e; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
tail -1 $D/make.log
ls -la $D/reads.fq
for n in $TH; do
  echo "== device ingest, -n $n"
  ( time MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out_$n -n $n ) 2>&1 | grep -E "timing|real|objects" | head -${LINES_MAX:-12}
  if [ -n "$HOST" ]; then
    echo "== host ingest, -n $n -b $n"
    ( time MIC_HOST_INGEST=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/outh_$n -n $n -b $n ) 2>&1 | grep -E "real|objects" | head -4
    cmp $D/out_$n.csv $D/outh_$n.csv && echo "CSV identical ($(wc -c < $D/out_$n.csv) bytes)"
  fi
done
