#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random table sizes, k, key widths, label counts, read lengths and N rates, both table
layouts, whole table and bucket-range shards.  python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
import test_gpu_parity as tp
from cuclark_amd import MiClarkDB, host

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
o = gu.oracle()
n_cases = n_reads = 0
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    k = int(rng.choice([8, 12, 16, 20, 21, 24, 25, 27, 31, 32]))
    htsize = int(rng.choice([2, 97, 1009, 4096, 65537, 99991, 1 << 20, 999983, 57777779]))
    key_bytes = host.key_bytes_rule(htsize, k)
    n_elems = int(rng.integers(50, 120000))
    if k < 16:
        n_elems = min(n_elems, (1 << (2 * k)) // 3)
    n_elems = min(n_elems, htsize * 200)
    T = int(rng.choice([1, 2, 7, 40, 64, 65, 300, 4096]))
    sizes, keys, labels, canon = gu.random_db(rng, htsize, n_elems, k, key_bytes, T)
    odb = o.db_from_arrays(sizes, keys, labels)
    L = int(rng.choice([k, k + 1, 40, 100, 150, 151, 250, 400, 1000]))
    data = tp._random_reads(rng, canon, k, int(rng.integers(50, 400)), max(L, k))
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = tp._oracle_results(odb, k, rp, cont, T)
    for layout in (1, 2, 3, 4):
        with MiClarkDB(k, T, layout=layout) as e:
            e.read_arrays(sizes, keys, labels)
            res, rows = e.classify_packed(rp, cont, extended=True)
        if not (res[:, :5] == expect).all():
            bad = np.flatnonzero((res[:, :5] != expect).any(axis=1))
            print(f"MISMATCH seed={seed} layout={layout} k={k} htsize={htsize} n={n_elems} T={T} L={L} reads={bad[:5]}")
            print(res[bad[0], :6], expect[bad[0]])
            sys.exit(1)
        # two shards through the batch API merge
        if htsize >= 4 and rng.random() < 0.5:
            cut = int(rng.integers(1, htsize))
            engines = [MiClarkDB(k, T, layout=layout) for _ in range(2)]
            try:
                n = rp.size - 1
                for e, sh in zip(engines, ((0, cut), (cut, htsize))):
                    e.read_arrays(sizes, keys, labels, shard=sh)
                    b = e.malloc(n, n, max(cont.size, 1), [0, n], True)
                    b["reads_pointer"][0][: n + 1] = rp
                    b["containers"][0][: cont.size] = cont
                    e.readyBatch(0, n, cont.size)
                    e.queryBatch(0, True)
                MiClarkDB.merge_shards(engines, 0)
                r2 = engines[0]._bufs["results"].copy()
            finally:
                for e in engines:
                    e.close()
            fits = (r2[:, 6] & 1) == 0          # rows that fit: must equal the whole-table answer
            if not (r2[fits, :5] == expect[fits]).all():
                print(f"SHARD MISMATCH seed={seed} layout={layout} k={k} htsize={htsize} cut={cut} T={T}")
                sys.exit(1)
    n_cases += 1
    n_reads += rp.size - 1
    seed += 1
    if n_cases % 200 == 0:
        print(f"... {n_cases} configurations, {n_reads} reads, {t_end - time.time():.0f} s left", flush=True)
print(f"fuzz ok: {n_cases} random configurations x 4 layouts, {n_reads} reads, seeds {seed0}..{seed - 1}")
