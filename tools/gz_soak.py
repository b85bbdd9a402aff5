#!/usr/bin/env python3
"""Random deflate streams through the device inflate against zlib, for a while: the generator of tests/test_gz_device.py
(levels, strategies, memLevels, windows, flushes; DNA / FASTQ / text / noise / runs), as one member and as block gzip.
    python tools/gz_soak.py [seconds] [seed]"""
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gz_device as tg
from cuclark_amd import MiClarkDB
from cuclark_amd.db import MiClarkUnsupported

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_streams = took = n_bgzf = 0
n_bytes = 0
with MiClarkDB(31, 4) as e:
    while time.time() < t_end:
        data, gz, what = tg._random_stream(rng)
        n_streams += 1
        try:
            text, crc = e.gunzip(gz)
            if text != data or crc != zlib.crc32(data):
                print("MISMATCH", seed, n_streams, what, flush=True)
                sys.exit(1)
            took += 1
            n_bytes += len(data)
        except MiClarkUnsupported:
            pass
        if what[1] and n_streams % 3 == 0:
            block = int(rng.choice([300, 4096, 0xFF00, 65536]))
            try:
                members = tg._bgzf(data, block, what[2] if what[2] else 1)
            except struct.error:            # (a member of 64 KiB of noise does not fit the format's 16-bit size)
                members = None
            if members is not None:
                if e.gunzip(members)[0] != data:
                    print("BGZF MISMATCH", seed, n_streams, what, block, flush=True)
                    sys.exit(1)
                n_bgzf += 1
        if n_streams % 500 == 0:
            print(f"... {n_streams} streams, {took} taken, {n_bgzf} as block gzip, {n_bytes / 1e6:.0f} MB of text, {t_end - time.time():.0f} s left", flush=True)
print(f"gz soak ok: {n_streams} streams ({took} taken by the device path, the rest handed back), {n_bgzf} as block gzip, {n_bytes / 1e6:.0f} MB of text, seed {seed}")
