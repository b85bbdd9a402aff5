#!/bin/bash
# Per-batch timeline of the streaming command line (MIC_CLI_TRACE=1): tools/e2e_trace.sh [reads] [threads] [fasta]
set -e
N=${1:-10000000}; TH=${2:-12}
D=/tmp/e2e; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
for rep in 1 2; do MIC_CLI_TIMING=1 MIC_CLI_TRACE=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/out -n $TH 2> $D/trace_$rep.log > /dev/null; done
grep -E "device ingest:" $D/trace_2.log | sed -e 's/.*threads: //' -e 's/input .*ms since start//'
python3 - <<PY
import re
rows=[]
for l in open("$D/trace_2.log"):
    m=re.match(r"\[trace\] batch (\d+) slot (\d+) bytes (\d+): taken (\d+) loaded (\d+) dev (\d+)-(\d+) write (\d+)-(\d+)", l)
    if m: rows.append(tuple(int(x) for x in m.groups()))
rows.sort()
n=len(rows)
def avg(f): return sum(f(r) for r in rows)/n/1e3
print(f"{n} batches; ms per batch: load {avg(lambda r:r[4]-r[3]):.2f}, wait for a device thread {avg(lambda r:r[5]-r[4]):.2f}, device {avg(lambda r:r[6]-r[5]):.2f}, wait for the turn / writer {avg(lambda r:r[7]-r[6]):.2f}, write {avg(lambda r:r[8]-r[7]):.2f}; slot held {avg(lambda r:r[8]-r[3]):.2f}")
for r in rows:
    print("batch %3d slot %2d %5.1f MB taken %6.2f loaded %6.2f dev %6.2f-%6.2f write %6.2f-%6.2f" % (r[0], r[1], r[2]/1e6, r[3]/1e3, r[4]/1e3, r[5]/1e3, r[6]/1e3, r[7]/1e3, r[8]/1e3))
PY
