#!/bin/bash
# -P a.fq.gz b.fq.gz with the mates inflated and merged on the device against the host inflater (MIC_GZ_HOST=1):
#   tools/gz_pairs_device_timing.sh [pairs]
set -e
N=${1:-1000000}
D=/tmp/gzpd; rm -rf $D; mkdir -p $D
python tools/make_synth_files.py $D --light --reads $N --kmers 60000000 --paired > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
for i in 1 2; do gzip -1 -k $D/reads_$i.fq; done
ls -la $D/reads_1.fq $D/reads_1.fq.gz | awk '{print $5, $9}'
echo "== host inflate"
for rep in 1 2; do MIC_GZ_HOST=1 MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq.gz $D/reads_2.fq.gz -R $D/host -n 12 2>&1 | grep -E "Assignment|inflate|device ingest" | sed 's/thread-seconds.*ms since start/.. ms since start/'; done
echo "== device inflate"
if [ -n "$GZ_TIMING" ]; then export MIC_GZ_TIMING=1; fi
for rep in 1 2 3; do MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq.gz $D/reads_2.fq.gz -R $D/dev -n 12 2>&1 | grep -E "Assignment|inflate|device ingest|^\[gz\]|^\[pairs\]" | sed 's/thread-seconds.*ms since start/.. ms since start/'; done
cmp $D/host.csv $D/dev.csv && echo "CSVs identical"
echo "== one file: host inflate, device inflate"
MIC_GZ_HOST=1 MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads_1.fq.gz -R $D/host1 -n 12 2>&1 | grep -E "Assignment|inflate" | sed 's/thread-seconds.*ms since start/.. ms since start/'
for rep in 1 2; do MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads_1.fq.gz -R $D/dev1 -n 12 2>&1 | grep -E "Assignment|inflate" | sed 's/thread-seconds.*ms since start/.. ms since start/'; done
cmp $D/host1.csv $D/dev1.csv && echo "CSVs identical"

echo "== block gzip (BGZF, 0xFF00-byte members, level 1): host, device"
python3 - $D <<'PY'
import struct, sys, zlib
d = sys.argv[1]
for i in (1, 2):
    data = open(f"{d}/reads_{i}.fq", "rb").read()
    with open(f"{d}/reads_{i}.fq.bgz.gz", "wb") as f:
        for o in range(0, len(data), 0xFF00):
            blk = data[o:o + 0xFF00]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
PY
MIC_GZ_HOST=1 MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq.bgz.gz $D/reads_2.fq.bgz.gz -R $D/hostb -n 12 2>&1 | grep -E "Assignment|inflate" | sed 's/thread-seconds.*ms since start/.. ms since start/'
for rep in 1 2 3; do MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -P $D/reads_1.fq.bgz.gz $D/reads_2.fq.bgz.gz -R $D/devb -n 12 2>&1 | grep -E "Assignment|inflate|^\[gz\]|^\[pairs\]" | sed 's/thread-seconds.*ms since start/.. ms since start/'; done
cmp $D/hostb.csv $D/devb.csv && cmp $D/hostb.csv $D/host.csv && echo "CSVs identical"
rm -rf $D
