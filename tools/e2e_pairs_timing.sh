#!/bin/bash
# End-to-end timing of exe/cuCLARK-l -P on two synthetic FASTQ files: tools/e2e_pairs_timing.sh [pairs] [threads...]
set -e
N=${1:-10000000}; shift || true
TH=${@:-12}
D=/tmp/e2ep; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 --paired > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
tail -1 $D/make.log
for n in $TH; do for rep in 1 2; do
  echo "== pairs merged by the loaders, -n $n"
  ( time MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq $D/reads_2.fq -R $D/out_$n -n $n ) 2>&1 | grep -E "timing|real|objects" | head -12
done; done
if [ -n "$SERIAL" ]; then
  echo "== serial reader"
  ( time MIC_SERIAL_PAIRS=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq $D/reads_2.fq -R $D/ser -n 12 ) 2>&1 | grep -E "real|objects" | head -4
  cmp $D/out_${TH##* }.csv $D/ser.csv && echo "CSV identical"
fi
