R=$(pwd)
F=/tmp/lr.fq
python3 - "$F" 3 <<'PY'
import sys, numpy as np
path, gb = sys.argv[1], float(sys.argv[2])
rng = np.random.default_rng(1)
n = 200000
seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
buf = bytearray()
for i in range(n):
    buf += b"@read_%09d/1\n" % i + seqs[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n"
with open(path, "wb") as f:
    for _ in range(int(gb * 1e9 / len(buf)) + 1):
        f.write(buf)
PY
cat $F > /dev/null
ls /sys/devices/system/node/ | grep node
for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-200
nproc
for th in 8 12; do
  echo -n "free th=$th: "; $R/exe/cuCLARK --strip-fastq $F - 262144 loaders $th | tail -1
  for n in /sys/devices/system/node/node*; do
    cl=$(cat $n/cpulist)
    echo -n "$(basename $n) th=$th: "; taskset -c $cl $R/exe/cuCLARK --strip-fastq $F - 262144 loaders $th 2>&1 | tail -1
  done
done
rm -f $F
