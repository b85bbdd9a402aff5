#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random table sizes, k, key widths, label counts, read lengths and N rates, all four table
layouts, the whole table, bucket-range shards and parts of the table (mic_db_set_part) merged through the batch API.
    python tools/fuzz_parity.py [seconds] [seed]
tests/test_fuzz_slice.py runs a 60-second seeded slice of it under -m gpu."""
import faulthandler
import os
import sys
import time

faulthandler.enable(all_threads=True)      # a native crash in a soak of hours must at least name the call it happened in

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def fuzz(budget, seed0=1, verbose=True):
    """Runs random configurations for `budget` seconds; returns (configurations, reads); raises AssertionError on a mismatch."""
    import golden_util as gu
    import test_gpu_parity as tp
    from cuclark_amd import MiClarkDB, host
    t_end = time.time() + budget
    o = gu.oracle()
    n_cases = n_reads = 0
    n_group = [0]
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
        n = rp.size - 1
        for layout in (1, 2, 3, 4):
            tag = f"seed={seed} layout={layout} k={k} htsize={htsize} n={n_elems} T={T} L={L}"
            if os.environ.get("MIC_FUZZ_TRACE"):
                print(tag, flush=True)
            with MiClarkDB(k, T, layout=layout) as e:
                e.read_arrays(sizes, keys, labels)
                res, rows = e.classify_packed(rp, cont, extended=True)
            bad = np.flatnonzero((res[:, :5] != expect).any(axis=1))
            assert bad.size == 0, f"MISMATCH {tag} reads={bad[:5]} got={res[bad[0], :6]} want={expect[bad[0]]}"
            # the table cut in two bucket ranges, or in 2..5 parts (mic_db_set_part), through the batch API merge
            mode = rng.random()
            if mode < 0.35 and htsize >= 4:
                cut = int(rng.integers(1, htsize))
                cuts = [dict(shard=(0, cut)), dict(shard=(cut, htsize))]
            elif mode < 0.7 and htsize >= 8:
                np_ = int(rng.integers(2, 6))
                cuts = [dict(part=(p, np_)) for p in range(np_)]
            else:
                continue
            if os.environ.get("MIC_FUZZ_TRACE"):
                print("   cuts", cuts, flush=True)
            engines = [MiClarkDB(k, T, layout=layout) for _ in cuts]
            try:
                for e, c in zip(engines, cuts):
                    if "part" in c:
                        e.set_part(*c["part"])
                        e.read_arrays(sizes, keys, labels)
                    else:
                        e.read_arrays(sizes, keys, labels, shard=c["shard"])
                    b = e.malloc(n, n, max(cont.size, 1), [0, n], True)
                    b["reads_pointer"][0][: n + 1] = rp
                    b["containers"][0][: cont.size] = cont
                    e.readyBatch(0, n, cont.size)
                    e.queryBatch(0, True)
                MiClarkDB.merge_shards(engines, 0)
                r2 = engines[0]._bufs["results"].copy()
                if "part" in cuts[0] and len(data) <= (1 << 20):
                    # the command line's table-sharded path on the same parts: the bytes of the batch through the owner's ingest
                    # slot, every engine probing its part, rows summed read-range owned (mic_ingest_classify_group)
                    owner = int(rng.integers(0, len(engines)))
                    engines[owner].ingest_alloc(1, 1 << 20, [f"t{i}" for i in range(T)], want_results=True)
                    g = MiClarkDB.ingest_classify_group(engines, owner, 0, data)
                    if g["status"] == 0:
                        assert (g["results"][:, :5] == expect).all(), f"GROUP INGEST MISMATCH {tag} cuts={cuts} owner={owner}"
                        n_group[0] += 1
                    else:       # handed back: only for what the device path does not take (a row beyond 15 targets, odd records)
                        assert g["status"] & 1, g
                    engines[owner].ingest_free()
            finally:
                for e in engines:
                    e.close()
            fits = (r2[:, 6] & 1) == 0          # rows that fit: must equal the whole-table answer
            assert (r2[fits, :5] == expect[fits]).all(), f"SHARD MISMATCH {tag} cuts={cuts}"
        n_cases += 1
        n_reads += n
        seed += 1
        if verbose and n_cases % 200 == 0:
            print(f"... {n_cases} configurations, {n_reads} reads, {n_group[0]} table-sharded ingest batches, {t_end - time.time():.0f} s left", flush=True)
    if verbose:
        print(f"table-sharded ingest batches checked: {n_group[0]}", flush=True)
    return n_cases, n_reads


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    try:
        n_cases, n_reads = fuzz(budget, seed0)
    except AssertionError as ex:
        print(ex)
        sys.exit(1)
    print(f"fuzz ok: {n_cases} random configurations x 4 layouts, {n_reads} reads, seeds {seed0}..{seed0 + n_cases - 1}", flush=True)
