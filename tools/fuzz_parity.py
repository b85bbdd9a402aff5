#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random table sizes, k, key widths, label counts, read lengths and N rates, all four table
layouts, the whole table, bucket-range shards and parts of the table (mic_db_set_part) merged through the batch API.
    python tools/fuzz_parity.py [seconds] [seed] [--split]
tests/test_fuzz_slice.py runs a seeded slice of it under -m gpu.

--split: the PRODUCT and the ORACLE live in two processes.  This process loads the product library only
(libmi_clark*.so: HIP runtime, kernels, the host packer); a child process (`--oracle-side`) loads liboracle.so only, generates every
configuration and computes what the product must answer; the arrays travel over a pipe.  A native fault - the heap corruption seen
three times in ~100 hours of soaking, DESIGN.md 7 - then names its side by which process dies."""
import faulthandler
import os
import pickle
import struct
import subprocess
import sys
import time

faulthandler.enable(all_threads=True)      # a native crash in a soak of hours must at least name the call it happened in

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def random_reads(rng, canon, k, n_reads, read_len, kmer_to_ascii, hit_frac=0.6, n_rate=0.01):
    """ASCII FASTA with reads stitched from DB k-mers (so they hit) and random sequence (tests/test_gpu_parity.py: _random_reads)."""
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


def microsatellite_db(rng, o, htsize, k, key_bytes, T):
    """A database of tandem repeats between random flanks (round 6): the flanks of one unit at many sites are hundreds of contexts of
    ONE minimizer - crowded minimizers, a side table, reads handed to crowd_finish_kernel; plus plain random sequence.  Every k-mer is
    labelled by the first site it occurs in.  Returns what random_db returns and the sites (the reads are cut from them)."""
    code = {"A": 3, "C": 2, "G": 1, "T": 0}
    units = ["AC", "AG", "AAT", "ACG", "AAAC", "ACAG", "A", "AGC", "GC", "AT", "ACGT", "".join(rng.choice(list("ACGT"), int(rng.integers(2, 9))))]
    sites = []
    for i in range(int(rng.integers(60, 260))):
        u = units[int(rng.integers(0, len(units)))]
        rep_ = (u * 60)[: int(rng.integers(k, 2 * k + 20))]
        fl = "".join(rng.choice(list("ACGT"), 2 * k + 8))
        sites.append(fl[: k + 4] + rep_ + fl[k + 4:])
    for i in range(20):
        sites.append("".join(rng.choice(list("ACGT"), int(rng.integers(k, 6 * k)))))
    mask = (1 << (2 * k)) - 1
    first = {}
    for i, sq in enumerate(sites):
        v = 0
        for p, ch in enumerate(sq):
            v = ((v << 2) | code[ch]) & mask
            if p >= k - 1:
                first.setdefault(o.canonical(v, k), i % T)
    canon = sorted(first, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    keep = []
    for c in canon:
        if sizes[c % htsize] < 255 and (c // htsize) < (1 << (8 * key_bytes)):
            sizes[c % htsize] += 1
            keep.append(c)
    import golden_util as gu
    keys = np.array([c // htsize for c in keep], dtype=np.uint64).astype(gu.KEY_DTYPE[key_bytes])
    labels = np.array([first[c] for c in keep], dtype=np.uint16)
    return sizes.astype(np.uint8), keys, labels, np.array(keep, dtype=np.uint64), sites


def site_reads(rng, sites, n_reads, read_len, n_rate=0.01):
    """reads cut out of the sites (either strand, a few substitutions and N) and random ones"""
    comp = str.maketrans("ACGT", "TGCA")
    recs = []
    for i in range(n_reads):
        if rng.random() < 0.15:
            sq = "".join(rng.choice(list("ACGT"), int(rng.integers(1, read_len + 1))))
        else:
            src = sites[int(rng.integers(0, len(sites)))]
            a = int(rng.integers(0, max(1, len(src) - 20)))
            sq = src[a:a + int(rng.integers(max(1, read_len // 2), read_len + 1))]
            while len(sq) < read_len // 2 and rng.random() < 0.7:      # long reads: several sites end to end
                sq += sites[int(rng.integers(0, len(sites)))]
            sq = list(sq[:read_len])
            for p in range(len(sq)):
                if rng.random() < n_rate:
                    sq[p] = "ACGTN"[int(rng.integers(0, 5))]
            sq = "".join(sq)
            if i % 2:
                sq = sq[::-1].translate(comp)
        recs.append(f">r{i}\n{sq}\n")
    return "".join(recs).encode()


def make_case(seed, pack):
    """ORACLE SIDE: configuration `seed` - the database, the reads, and the answer the oracle gives.  pack(data, k) -> (rp, cont):
    the product's host packer in the one-process form (as before), the oracle's own in the split form."""
    import golden_util as gu
    o = gu.oracle()
    rng = np.random.default_rng(seed)
    k = int(rng.choice([8, 12, 16, 20, 21, 24, 25, 27, 31, 32]))
    htsize = int(rng.choice([2, 97, 1009, 4096, 65537, 99991, 1 << 20, 999983, 57777779]))
    key_bytes = o.key_bytes_rule(htsize, k)
    n_elems = int(rng.integers(50, 120000))
    if k < 16:
        n_elems = min(n_elems, (1 << (2 * k)) // 3)
    n_elems = min(n_elems, htsize * 200)
    T = int(rng.choice([1, 2, 7, 40, 64, 65, 300, 4096]))
    L = int(rng.choice([k, k + 1, 40, 100, 150, 151, 250, 400, 1000]))
    if seed % 5 == 3 and k >= 20 and htsize >= 1009:
        # every fifth configuration: tandem repeats - crowded minimizers in the super-k-mer layouts (same answers from every layout)
        sizes, keys, labels, canon, sites = microsatellite_db(rng, o, htsize, k, key_bytes, T)
        n_elems = int(keys.size)
        odb = o.db_from_arrays(sizes, keys, labels)
        data = site_reads(rng, sites, int(rng.integers(50, 400)), max(L, k))
    else:
        sizes, keys, labels, canon = gu.random_db(rng, htsize, n_elems, k, key_bytes, T)
        odb = o.db_from_arrays(sizes, keys, labels)
        data = random_reads(rng, canon, k, int(rng.integers(50, 400)), max(L, k), gu.kmer_to_ascii)
    rp, cont = pack(data, k)
    counts, bad = odb.query_batch(k, rp, cont, T)
    assert bad == 0
    expect = o.result_from_counts(counts)
    odb.close()
    return dict(seed=seed, k=k, htsize=htsize, key_bytes=key_bytes, n_elems=n_elems, T=T, L=L, sizes=sizes, keys=keys, labels=labels,
                data=data, rp=rp, cont=cont, expect=expect, rng_state=rng.bit_generator.state)


def check_case(c, n_group):
    """PRODUCT SIDE: the case through all four layouts, shards, parts and the table-sharded ingest; AssertionError on a mismatch."""
    from cuclark_amd import MiClarkDB, host
    rng = np.random.default_rng(0)
    rng.bit_generator.state = c["rng_state"]
    k, htsize, T, L, n_elems, seed = c["k"], c["htsize"], c["T"], c["L"], c["n_elems"], c["seed"]
    sizes, keys, labels, data, expect = c["sizes"], c["keys"], c["labels"], c["data"], c["expect"]
    # the product's own indexer and packer on the bytes; they must give the arrays the oracle's answer was computed from
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    assert rp.shape == c["rp"].shape and (rp == c["rp"]).all() and cont.shape == c["cont"].shape and (cont == c["cont"]).all(), \
        f"PACKER MISMATCH seed={seed} k={k}"
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
            one_upload = rng.random() < 0.5      # mic_batch_query_group: the first engine uploads, the others copy device to device
            for i, (e, ct) in enumerate(zip(engines, cuts)):
                if "part" in ct:
                    e.set_part(*ct["part"])
                    e.read_arrays(sizes, keys, labels)
                else:
                    e.read_arrays(sizes, keys, labels, shard=ct["shard"])
                b = e.malloc(n, n, max(cont.size, 1), [0, n], True)
                if one_upload and i:
                    continue
                b["reads_pointer"][0][: n + 1] = rp
                b["containers"][0][: cont.size] = cont
                e.readyBatch(0, n, cont.size)
                if not one_upload:
                    e.queryBatch(0, True)
            if one_upload:
                MiClarkDB.query_group(engines, 0, True)
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
    return n


def oracle_side():
    """child of --split: reads seeds (one per line) from stdin, writes length-prefixed pickles of the cases to stdout; no product here"""
    import golden_util as gu
    o = gu.oracle()
    assert "cuclark_amd" not in sys.modules

    def pack(data, k):
        idx = o.index_reads(data)
        return o.pack_batch(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    out = sys.stdout.buffer
    for line in sys.stdin:
        blob = pickle.dumps(make_case(int(line), pack), protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<Q", len(blob)))
        out.write(blob)
        out.flush()


class OracleProcess:
    """the oracle side of --split as a child process; a few seeds are kept in flight so that it works ahead of the product"""
    AHEAD = 3

    def __init__(self, seed0):
        env = {k: v for k, v in os.environ.items() if k not in ("MIC_LIB_PATH",)}
        if "guardalloc" in env.get("LD_PRELOAD", ""):      # (tools/guard_soak.sh watches the PRODUCT side's heap; the oracle has its own rig)
            del env["LD_PRELOAD"]
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-side"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
        self.next_seed = seed0
        for _ in range(self.AHEAD):
            self._ask()

    def _ask(self):
        self.p.stdin.write(f"{self.next_seed}\n".encode())
        self.p.stdin.flush()
        self.next_seed += 1

    def case(self):
        self._ask()
        head = self.p.stdout.read(8)
        if len(head) != 8:
            rc = self.p.wait()
            raise RuntimeError(f"ORACLE SIDE died (exit {rc}{': signal ' + str(-rc) if rc < 0 else ''}) - the product process is alive")
        (n,) = struct.unpack("<Q", head)
        return pickle.loads(self.p.stdout.read(n))

    def close(self):
        # (the cases still in flight are not wanted any more: the child is ended, not waited for)
        try:
            self.p.kill()
            self.p.wait(timeout=30)
        except Exception:
            pass


def fuzz(budget, seed0=1, verbose=True, split=False):
    """Runs random configurations for `budget` seconds; returns (configurations, reads); raises AssertionError on a mismatch."""
    t_end = time.time() + budget
    n_cases = n_reads = 0
    n_group = [0]
    seed = seed0
    orc = OracleProcess(seed0) if split else None
    if split:
        assert "oracle" not in sys.modules and "oracle.binding" not in sys.modules
    else:
        from cuclark_amd import host

        def pack(data, k):
            idx = host.index_reads(data)
            return host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    try:
        while time.time() < t_end:
            c = orc.case() if split else make_case(seed, pack)
            assert c["seed"] == seed
            n_reads += check_case(c, n_group)
            n_cases += 1
            seed += 1
            if verbose and n_cases % 200 == 0:
                print(f"... {n_cases} configurations, {n_reads} reads, {n_group[0]} table-sharded ingest batches, {t_end - time.time():.0f} s left", flush=True)
    finally:
        if orc:
            orc.close()
    if split:
        assert "oracle.binding" not in sys.modules, "the product side of a split run must not load the oracle"
    if verbose:
        print(f"table-sharded ingest batches checked: {n_group[0]}", flush=True)
    return n_cases, n_reads


if __name__ == "__main__":
    if "--oracle-side" in sys.argv:
        oracle_side()
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    split = "--split" in sys.argv
    budget = float(args[0]) if len(args) > 0 else 120.0
    seed0 = int(args[1]) if len(args) > 1 else 1
    try:
        n_cases, n_reads = fuzz(budget, seed0, split=split)
    except AssertionError as ex:
        print(ex)
        sys.exit(1)
    print(f"fuzz ok: {n_cases} random configurations x 4 layouts, {n_reads} reads, seeds {seed0}..{seed0 + n_cases - 1}"
          + (" (product and oracle in separate processes)" if split else ""), flush=True)
