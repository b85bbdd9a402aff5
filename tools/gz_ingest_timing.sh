#!/bin/bash
# Assignment time of exe/cuCLARK-l on the same reads as plain FASTQ, gzip and block-gzip (BGZF): tools/gz_ingest_timing.sh [reads]
set -e
N=${1:-4000000}
D=/tmp/gzt; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
( time gzip -1 -k $D/reads.fq ) 2>&1 | grep real
python - <<PY
import struct, zlib
data = open("$D/reads.fq", "rb").read()
with open("$D/reads.fq.bgz", "wb") as fo:
    for off in list(range(0, len(data), 65280)) + [len(data)]:
        raw = data[off:off + 65280] if off < len(data) else b""
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        fo.write(b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1))
        fo.write(body + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)))
PY
ls -la $D/reads.fq $D/reads.fq.gz $D/reads.fq.bgz | awk '{print $5, $9}'
for f in reads.fq reads.fq.gz reads.fq.bgz; do
  for t in ${THREADS:-8}; do
    echo "== $f MIC_INFLATE_THREADS=$t"
    MIC_INFLATE_THREADS=$t ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/$f -R $D/out_$f -n 32 -b 32 2>/dev/null | grep -E "Assignment"
  done
done
cmp $D/out_reads.fq.csv $D/out_reads.fq.gz.csv && cmp $D/out_reads.fq.csv $D/out_reads.fq.bgz.csv && echo "CSVs identical"
rm -rf $D
