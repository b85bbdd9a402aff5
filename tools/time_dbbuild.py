#!/usr/bin/env python3
"""Time mic_db_build on synthetic genomes: python tools/time_dbbuild.py [n_genomes] [genome_nt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuclark_amd import host
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gl = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
d = "/tmp/dbbuild_time"
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
files, labels = [], []
t0 = time.time()
for g in range(ng):
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, gl, dtype=np.uint8)]
    lines = seq.reshape(-1, 80)
    with open(f"{d}/g{g}.fa", "wb") as f:
        f.write(b">genome%d\n" % g)
        f.write(b"\n".join(l.tobytes() for l in lines) + b"\n")
    files.append(f"{d}/g{g}.fa"); labels.append(g % 48)
print(f"wrote {ng} genomes x {gl} nt in {time.time()-t0:.1f} s")
for htsize, k in ((1610612741, 31), (57777779, 27)):
    t0 = time.time()
    n = host.build_db(files, labels, k, htsize, f"{d}/db", threads=32)
    dt = time.time() - t0
    print(f"HTSIZE {htsize} k={k}: {n} k-mers from {ng*gl} nt in {dt:.2f} s = {ng*gl/dt/1e6:.1f} Mnt/s")
