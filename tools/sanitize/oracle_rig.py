#!/usr/bin/env python3
"""The ORACLE alone under AddressSanitizer + UBSan: the fuzzer's generator (tools/fuzz_parity.py) drives every
entry point of oracle/liboracle.so the fuzzer and the parity tests call - through the same ctypes binding and numpy buffers - with NO
product library in the process.  tools/sanitize/oracle_rig.sh builds the sanitized library and starts this under the sanitizer's
runtime with PYTHONMALLOC=malloc, so that a write of the oracle's past a numpy buffer or a ctypes structure is reported where it happens.
    tools/sanitize/oracle_rig.sh [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def key_bytes_rule(o, htsize, k):
    return o.key_bytes_rule(htsize, k)


def random_reads(rng, canon, k, n_reads, read_len, kmer_to_ascii, hit_frac=0.6, n_rate=0.01):
    """tests/test_gpu_parity.py: _random_reads, restated here so that nothing of the GPU suite is imported"""
    recs = []
    for i in range(n_reads):
        L = int(rng.integers(max(1, read_len // 2), read_len + 1))
        s = []
        while sum(len(x) for x in s) < L:
            if canon.size and rng.random() < hit_frac:
                km = kmer_to_ascii(canon[int(rng.integers(canon.size))], k)
                if rng.random() < 0.5:
                    km = km[::-1].translate(str.maketrans("ACGT", "TGCA"))
                s.append(km)
            else:
                s.append("".join(rng.choice(list("ACGT"), int(rng.integers(1, k + 5)))))
        seq = list("".join(s)[:L])
        for p in range(len(seq)):
            if rng.random() < n_rate:
                seq[p] = "N"
        recs.append(f">r{i}\n{''.join(seq)}\n")
    return "".join(recs).encode()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import golden_util as gu
    o = gu.oracle()
    t_end = time.time() + budget
    n_cases = n_reads = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        k = int(rng.choice([8, 12, 16, 20, 21, 24, 25, 27, 31, 32]))
        htsize = int(rng.choice([2, 97, 1009, 4096, 65537, 99991, 1 << 20, 999983]))
        key_bytes = o.key_bytes_rule(htsize, k)
        n_elems = int(rng.integers(50, 30000))
        if k < 16:
            n_elems = min(n_elems, (1 << (2 * k)) // 3)
        n_elems = min(n_elems, htsize * 200)
        T = int(rng.choice([1, 2, 7, 40, 64, 65, 300, 4096]))
        sizes, keys, labels, canon = gu.random_db(rng, htsize, n_elems, k, key_bytes, T)
        sampling = int(rng.choice([1, 1, 1, 3]))
        odb = o.db_from_arrays(sizes, keys, labels, sampling)
        wdb = o.db_wrap_arrays(np.ascontiguousarray(sizes), np.ascontiguousarray(keys), np.ascontiguousarray(labels))
        L = int(rng.choice([k, k + 1, 40, 100, 150, 151, 250, 400, 1000]))
        data = random_reads(rng, canon, k, int(rng.integers(20, 150)), max(L, k), gu.kmer_to_ascii)
        idx = o.index_reads(data)
        rp, cont = o.pack_batch(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
        n = rp.size - 1
        counts, bad = odb.query_batch(k, rp, cont, T)
        assert bad == 0
        res = o.result_from_counts(counts)
        # bucket-range halves sum to the whole; slot-range parts (part_rule.c) sum to the whole
        if htsize >= 4:
            cut = int(rng.integers(1, htsize))
            c0, _ = odb.query_batch(k, rp, cont, T, (0, cut))
            c1, _ = odb.query_batch(k, rp, cont, T, (cut, htsize))
            assert sampling > 1 or (c0 + c1 == counts).all()
        if k >= 24 and sampling == 1:
            np_ = int(rng.integers(2, 6))
            n_slots = int(rng.integers(64, 5000))
            tot = np.zeros_like(counts)
            two = bool(rng.integers(0, 2))
            for p in range(np_):
                cp, _ = odb.query_batch_slot_part(k, 20 if k - 20 + 1 <= 16 else k - 15, two, n_slots, p, np_, rp, cont, T)
                tot += cp
            assert (tot == counts).all()
        # the classifiers
        r1 = wdb.classify_batch(k, rp, cont, T, threads=int(rng.integers(1, 5)))
        r2 = wdb.classify_batch_fast(k, rp, cont, T, threads=int(rng.integers(1, 5)))
        assert sampling > 1 or ((r1 == res).all() and (r2 == res).all())
        # sparse rows, merges, results from rows
        for i in range(min(n, 8)):
            nrow, row = o.sparse_row(counts[i], 64)
            if nrow <= 64:
                assert (o.result_from_row(row) == res[i]).all() or sampling > 1 and True
                m = o.merge_rows(row[: 1 + 2 * nrow], row[: 1 + 2 * nrow])
                assert m[0] == nrow
        # single reads from ASCII, finds, stats
        seq = data.split(b"\n")[1]
        odb.count_read_ascii(k, seq, len(seq), T)
        if canon.size:
            odb.find_many(canon[:16], k)
            odb.probe_stats(canon[:64], k)
        names = [f"t{i}" for i in range(T)]
        odb.classify_file(k, data, names, paired=False, extended=bool(rng.integers(0, 2)))
        odb.close(); wdb.close()
        n_cases += 1
        n_reads += n
        seed += 1
    print(f"oracle rig ok: {n_cases} random configurations, {n_reads} reads", flush=True)


if __name__ == "__main__":
    main()
