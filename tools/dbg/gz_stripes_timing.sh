#!/bin/bash
# the inflater's laps in stripes: tools/dbg/gz_stripes_timing.sh [reads]
N=${1:-4000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/gzp; rm -rf $D; mkdir -p $D
python3 $R/tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
gzip -1 -k $D/reads.fq
for S in "" 2; do
echo "== MIC_GZ_STRIPES=${S:-default}"
MIC_GZ_STRIPES=$S MIC_GZ_TIMING=1 MIC_CLI_TIMING=1 $R/exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq.gz -R $D/out_t -n 12 2>&1 | grep -E "^\[gz\]|Assignment|stripes" | grep -v "cycles per\|units:" | cut -c1-200
done
rm -rf $D
