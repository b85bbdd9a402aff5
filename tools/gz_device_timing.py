#!/usr/bin/env python3
"""One FASTQ file of the bench's shape, gzip -1, inflated on the device (mic_gz_inflate_device) with its stage times
(MIC_GZ_TIMING=1) next to zlib on one host thread.    python tools/gz_device_timing.py [reads [binned]]"""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(3)
nt = np.frombuffer(b"ACGT", np.uint8)
seqs = nt[rng.integers(0, 4, (n, 150))]
qual = np.full((n, 150), ord("I"), np.uint8)        # the bench's files (mic_synth_reads_text_device): one quality value
if len(sys.argv) > 2 and sys.argv[2] == "binned":    # binned qualities as sequencers of the last decade write them: four values in runs
    runs = rng.integers(0, 4, (n, 150 // 5))
    qual = np.frombuffer(b"F:,#", np.uint8)[np.repeat(runs, 5, axis=1)]
    qual[rng.random((n, 150)) < 0.85] = ord("F")
recs = np.empty((n, 316), np.uint8)
for i, h in enumerate(np.char.mod("@r%09d\n", np.arange(n)).astype("S12")):
    pass
hdr = np.frombuffer(b"".join(b"@r%09d\n" % i for i in range(n)), np.uint8).reshape(n, 12)
recs[:, :12] = hdr
recs[:, 12:162] = seqs
recs[:, 162] = 10
recs[:, 163] = ord("+")
recs[:, 164] = 10
recs[:, 165:315] = qual
recs[:, 315] = 10
data = recs.tobytes()
t0 = time.time()
c = zlib.compressobj(1, zlib.DEFLATED, 31)
gz = c.compress(data) + c.flush()
print(f"{len(data) / 1e6:.0f} MB of FASTQ -> {len(gz) / 1e6:.1f} MB gzip -1 ({time.time() - t0:.1f} s to compress)", flush=True)
t0 = time.time()
assert zlib.decompress(gz, 31) == data
print(f"zlib, one thread: {len(data) / 1e6 / (time.time() - t0):.0f} MB/s", flush=True)
os.environ["MIC_GZ_TIMING"] = "1"
from cuclark_amd import MiClarkDB
with MiClarkDB(31, 4) as e:
    for rep in range(3):
        t0 = time.time()
        text, crc = e.gunzip(gz)
        dt = time.time() - t0
        print(f"device (with the copy back to pageable memory): {dt * 1e3:.1f} ms = {len(data) / 1e9 / dt:.2f} GB/s of text; equal: {text == data}, crc ok: {crc == zlib.crc32(data)}", flush=True)
