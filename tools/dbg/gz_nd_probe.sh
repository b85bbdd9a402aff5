#!/bin/bash
# device threads for a resident (device-inflated) input: tools/dbg/gz_nd_probe.sh
N=${1:-4000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/gzp; rm -rf $D; mkdir -p $D
python3 $R/tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
gzip -1 -k $D/reads.fq
for nd in 3 10 3 10 3 10 8 8; do
  echo "== ND=$nd: $(MIC_INGEST_ND=$nd MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq.gz -R $D/out_$nd -n 12 2>&1 | grep -E "device ingest|Assignment|device inflate" | sed 's/.*thread-seconds: //; s/.*of text in/inflate/; s/, 4000000 records.*//; s/input.*last off the device/last off the device/; s/, last write.*//' | tr '\n' ' ' | cut -c1-200)"
done
rm -rf $D
