"""Crowded minimizers: databases in which ONE minimizer value sits in hundreds of contexts - the flanks of
microsatellites.  The super-k-mer layouts move such a minimizer's k-mers out of its slot chain into a side table keyed by the
whole k-mer and leave a marker behind; a query run that meets the marker looks its k-mers up one by one.  Everything must stay
bit-exact against the oracle: whole table, parts of the table, the per-k-mer kernel, the dense path."""
import os

import numpy as np
import pytest

import golden_util as gu
from test_gpu_parity import _canonical_np, _oracle_results

pytestmark = pytest.mark.gpu


def _microsatellite_db(rng, k, htsize, T, n_sites=900, units=("AC", "AG", "AAT", "ACG", "AAAC", "ACAG", "A", "AGC")):
    """n_sites occurrences of short tandem repeats, each between its own random flanks, spread over T targets, plus plain
    random sequence; the database = every k-mer (labelled by the first sequence it occurs in)"""
    seqs, lab = [], []
    for i in range(n_sites):
        u = units[int(rng.integers(0, len(units)))]
        rep = (u * 40)[: int(rng.integers(30, 70))]
        fl = "".join(rng.choice(list("ACGT"), 90))
        seqs.append(fl[:45] + rep + fl[45:])
        lab.append(i % T)
    for i in range(60):
        seqs.append("".join(rng.choice(list("ACGT"), 400)))
        lab.append(i % T)
    code = {"A": 3, "C": 2, "G": 1, "T": 0}
    first = {}
    for s, l in zip(seqs, lab):
        v = 0
        for i, ch in enumerate(s):
            v = ((v << 2) | code[ch]) & ((1 << (2 * k)) - 1)
            if i >= k - 1:
                first.setdefault(int(_canonical_np(np.array([v], np.uint64), k)[0]), l)
    canon = sorted(first, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    keep = []
    for c in canon:
        if sizes[c % htsize] < 255:
            sizes[c % htsize] += 1
            keep.append(c)
    keys = np.array([c // htsize for c in keep], dtype=np.uint64)
    labels = np.array([first[c] for c in keep], dtype=np.uint16)
    return seqs, sizes.astype(np.uint8), keys, labels


def _reads(rng, seqs, n):
    comp = str.maketrans("ACGT", "TGCA")
    out = []
    for i in range(n):
        if rng.random() < 0.15:
            s = "".join(rng.choice(list("ACGT"), 150))
        else:
            src = seqs[int(rng.integers(0, len(seqs)))]
            a = int(rng.integers(0, max(1, len(src) - 120)))
            s = list(src[a:a + int(rng.integers(60, 151))])
            for p in range(len(s)):
                if rng.random() < 0.01:
                    s[p] = "ACGTN"[int(rng.integers(0, 5))]
            s = "".join(s)
            if i % 2:
                s = s[::-1].translate(comp)
        out.append(f">r{i}\n{s}\n")
    return "".join(out).encode()


@pytest.mark.parametrize("layout", ["direct", "minimizer", "super", "super2"])
def test_crowded_minimizers_go_to_the_side_table(layout, monkeypatch):
    from cuclark_amd import MiClarkDB, host
    import torch
    monkeypatch.setenv("MIC_LAYOUT", layout)
    rng = np.random.default_rng(41)
    k, T, htsize = 31, 12, 1 << 18
    seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T)
    data = _reads(rng, seqs, 3000)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert (expect[:, 0] > 0).mean() > 0.5
    with MiClarkDB(k, T, row_words=16) as e:
        e.read_arrays(sizes, keys, labels)
        info = e.info()
        res, rows = e.classify_packed(rp, cont, extended=True)
    assert (res[:, :5] == expect).all()
    if layout not in ("super", "super2"):
        assert info["side_kmers"] == 0
        return
    assert info["side_kmers"] > 500 and info["side_bytes"] >= 32 * info["side_kmers"] and info["max_chain"] <= 40, info
    # the same table with its chains left alone: same answers, far longer chains
    monkeypatch.setenv("MIC_S_NO_SIDE", "1")
    with MiClarkDB(k, T, row_words=16) as e:
        e.read_arrays(sizes, keys, labels)
        info0 = e.info()
        res0 = e.classify_packed(rp, cont)
    monkeypatch.delenv("MIC_S_NO_SIDE")
    assert info0["side_kmers"] == 0 and info0["max_chain"] > 3 * info["max_chain"] and (res0[:, :5] == expect).all()
    assert info0["n_elems"] == info["n_elems"]
    # the per-k-mer kernel, the dense path (rows of 2 entries overflow), and 3 parts of the table merged
    monkeypatch.setenv("MIC_S_PER_KMER", "1")
    with MiClarkDB(k, T) as e:
        e.read_arrays(sizes, keys, labels)
        assert (e.classify_packed(rp, cont)[:, :5] == expect).all()
    monkeypatch.delenv("MIC_S_PER_KMER")
    with MiClarkDB(k, T, row_words=3) as e:
        e.read_arrays(sizes, keys, labels)
        res3, rows3 = e.classify_packed(rp, cont, extended=True)
    assert (res3[:, :5] == expect).all() and (res3[:, 6] & 2).sum() > 10          # dense path taken, same results
    dev = torch.device("cuda:0")
    hits, side_total, acc = np.zeros(n, np.int64), 0, None
    with MiClarkDB(k, T, row_words=16) as m:
        for p in range(3):
            with MiClarkDB(k, T, row_words=16) as e:
                e.set_part(p, 3)
                e.read_arrays(sizes, keys, labels)
                side_total += e.info()["side_kmers"]
                r, rw = e.classify_packed(rp, cont, extended=True)
            hits += r[:, 0]
            cur = torch.from_numpy(rw.astype(np.int64)).to(dev).to(torch.int32).contiguous()
            if acc is None:
                acc = cur
            else:
                out = torch.empty_like(acc)
                torch.cuda.synchronize()
                m.merge_rows_device(acc.data_ptr(), cur.data_ptr(), out.data_ptr(), n)
                m.sync()
                acc = out
        results = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        m.result_from_rows_device(acc.data_ptr(), results.data_ptr(), n)
        m.sync()
    # (a group of 12 or 13 entries may fall on either side of the limit from one build to the next: the order in which equal
    # candidates are merged is not fixed)
    assert (hits == expect[:, 0]).all() and abs(side_total - info["side_kmers"]) <= 0.05 * info["side_kmers"]
    fits = acc[:, 0].cpu().numpy() != -1
    assert (results.cpu().numpy().view(np.uint32)[fits, :5] == expect[fits]).all() and fits.mean() > 0.9


def _long_reads(rng, seqs, n):
    """reads of 300 .. 3000 nt stitched from database sequences (several chunks, many rounds, crowded runs in several of them),
    some with an N in the middle (two parts)"""
    comp = str.maketrans("ACGT", "TGCA")
    out = []
    for i in range(n):
        s = ""
        while len(s) < int(rng.integers(300, 3000)):
            src = seqs[int(rng.integers(0, len(seqs)))]
            a = int(rng.integers(0, max(1, len(src) - 100)))
            s += src[a:a + int(rng.integers(50, 160))]
            if rng.random() < 0.1:
                s += "N"
        if i % 3 == 0:
            s = s[::-1].translate(comp)
        out.append(f">L{i}\n{s}\n")
    return "".join(out).encode()


@pytest.mark.parametrize("layout", ["super", "super2"])
def test_crowded_runs_go_through_the_follow_up_kernel(layout, monkeypatch):
    """query_kernel_r hands the runs of crowded minimizers to crowd_finish_kernel (work list in HBM): short reads through the
    pipelined road, long reads (several groups of crowded runs per read), a grid so small that every wavefront takes hundreds of
    reads, and a work area too small for the batch (what does not fit takes the dense path) - all equal to the oracle."""
    from cuclark_amd import MiClarkDB, host
    import torch
    monkeypatch.setenv("MIC_LAYOUT", layout)
    rng = np.random.default_rng(43)
    k, T, htsize = 31, 12, 1 << 18
    seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T)
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    dev = torch.device("cuda:0")
    # (the work area is sized for a few crowded runs per read of a batch: random reads around the microsatellite ones)
    filler = lambda n: b"".join(b">f%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)) + b"\n" for i in range(n))
    cases = {"short": _reads(rng, seqs, 3000) + filler(30000), "long": _long_reads(rng, seqs, 200) + filler(40000)}
    with MiClarkDB(k, T, row_words=16) as e:
        e.read_arrays(sizes, keys, labels)
        assert e.info()["side_kmers"] > 500
        for name, data in cases.items():
            idx = host.index_reads(data)
            rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
            n = rp.size - 1
            counts, expect = _oracle_results(odb, k, rp, cont, T)
            d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
            d_ct = torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint16)]).view(np.int16)).to(dev)
            for env, val in ((None, None), ("MIC_QUERY_BLOCKS", "2"), ("MIC_CROWD_CAP", "40"), ("MIC_CROWD_CAP", "0")):
                if env:
                    monkeypatch.setenv(env, val)
                d_res = torch.zeros((n, 8), dtype=torch.int32, device=dev)
                e.query_device(d_rp.data_ptr(), d_ct.data_ptr(), n, d_res.data_ptr())
                st = e.last_crowd_stats()
                dense = e.resolve_flagged_device(d_rp.data_ptr(), d_ct.data_ptr(), d_res.data_ptr())
                res = d_res.cpu().numpy().view(np.uint32)
                if env:
                    monkeypatch.delenv(env)
                assert (res[:, :5] == expect).all(), (name, env, val, int((res[:, :5] != expect).any(axis=1).sum()))
                if env == "MIC_CROWD_CAP":
                    assert dense > 0, (name, val, st, dense)
                    if val == "0":
                        assert st["reads"] == 0 or st["reads_to_dense_path"] > 0
                else:
                    # every read with a crowded run is finished by the follow-up, none by the dense path
                    assert st["reads"] > 150 and st["runs"] >= st["reads"] and st["reads_to_dense_path"] == 0 and dense == 0, (name, st, dense)
                    if name == "long":
                        assert st["runs"] > 3 * st["reads"], st


@pytest.mark.parametrize("layout", ["super", "super2"])
def test_crowded_runs_through_the_streaming_ingest_and_a_group_of_parts(layout, monkeypatch):
    """The same hand-over inside the command line's paths: mic_ingest_classify (FASTA bytes in, CSV out: the slot's own work area) and
    mic_ingest_classify_group (three parts of the table on three engines: every helper engine has a work area of its own, a crowded
    run belongs to the part that holds its slot) - results equal to the oracle, CSVs equal to each other."""
    from cuclark_amd import MiClarkDB, host
    monkeypatch.setenv("MIC_LAYOUT", layout)
    rng = np.random.default_rng(47)
    k, T, htsize = 31, 12, 1 << 18
    seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T)
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    data = _reads(rng, seqs, 2500) + b"".join(b">f%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)) + b"\n" for i in range(20000))
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    names = [f"T{i}" for i in range(T)]
    with MiClarkDB(k, T) as whole:
        whole.read_arrays(sizes, keys, labels)
        assert whole.info()["side_kmers"] > 500
        whole.ingest_alloc(1, 8 << 20, names, want_results=True)
        r_w = whole.ingest_classify(0, data)
        whole.ingest_free()
    assert r_w["status"] == 0 and r_w["n_reads"] == n and (r_w["results"][:, :5] == expect).all()
    group = [MiClarkDB(k, T) for _ in range(3)]
    try:
        side = 0
        for p, e in enumerate(group):
            e.set_part(p, 3)
            e.read_arrays(sizes, keys, labels)
            side += e.info()["side_kmers"]
        assert side > 500
        group[1].ingest_alloc(1, 8 << 20, names, want_results=True)
        r = MiClarkDB.ingest_classify_group(group, 1, 0, data)
        assert r["status"] == 0 and r["n_reads"] == n
        assert (r["results"][:, :5] == expect).all() and r["csv"] == r_w["csv"]
    finally:
        for e in group:
            e.close()


@pytest.mark.parametrize("n_sites", [12, 400])
@pytest.mark.parametrize("k", [32, 30, 22])
def test_palindromic_repeats_at_even_k(k, n_sites, monkeypatch):
    """(GC)n, (AT)n, (ACGT)n at even k: the repeat's k-mers AND its minimizers are their own reverse complements.  The one-strand
    table stores a k-mer under (c, p) and (rc(c), w-1-p); for rc(c) = c the two forms coincide pairwise, and until round 6 such a
    k-mer was stored twice, landed in two entries of its minimizer, and a run that met both entries counted it twice (found by
    tools/fuzz_parity.py's microsatellite configurations at seed 7308: one hit too many per read).  Few sites: the minimizers stay
    in their chains; many: they are crowded and go through the side table.  Every substring of a site is a read, so every
    alignment of the runs against the 32-nucleotide position keys occurs."""
    from cuclark_amd import MiClarkDB, host
    rng = np.random.default_rng(7308 + k + n_sites)
    T, htsize = 9, 1 << 14
    seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T, n_sites=n_sites, units=("GC", "AT", "ACGT", "AGCT", "AC"))
    recs = []
    for s_ in seqs[:6]:
        lo = 45 - k - 6
        recs += [f">s\n{s_[a:b]}\n" for a in range(lo, lo + 40) for b in range(a + k, min(len(s_), a + k + 48) + 1, 3)]
    data = ("".join(recs)).encode() + _reads(rng, seqs, 1500)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert (expect[:, 0] > 0).mean() > 0.5
    for layout in ("super", "super2", "minimizer"):
        monkeypatch.setenv("MIC_LAYOUT", layout)
        with MiClarkDB(k, T, row_words=16) as e:
            e.read_arrays(sizes, keys, labels)
            info = e.info()
            res = e.classify_packed(rp, cont)
        bad = np.flatnonzero((res[:, :5] != expect).any(axis=1))
        assert bad.size == 0, (layout, k, bad[:5], res[bad[:3], :5], expect[bad[:3]])
        if layout == "super" and n_sites >= 400:
            assert info["side_kmers"] > 0


def test_engines_come_and_go_on_the_same_streams():
    """mic_destroy gives the engine's streams and events back to a pool of the process and never destroys them (the HIP runtime's
    completion handler races a stream's destruction - the native fault of the fuzzer): engine after engine, with batch
    buffers and ingest slots, the numbers of streams and events ever created stay where the first engines put them."""
    import ctypes as C
    from cuclark_amd import MiClarkDB, host
    rng = np.random.default_rng(3)
    k, T, htsize = 21, 5, 1 << 12
    seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T, n_sites=40)
    data = _reads(rng, seqs, 200)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, first = None, None
    for i in range(40):
        with MiClarkDB(k, T, num_batches=1 + i % 3) as e:
            e.read_arrays(sizes, keys, labels)
            if e.num_batches == 1:
                res = e.classify_packed(rp, cont)
                first = res if first is None else first
                assert (res == first).all()
            else:                                               # (batch buffers: a stream and three events per batch)
                n = rp.size - 1
                e.malloc(n, n, max(cont.size, 1), np.linspace(0, n, e.num_batches + 1).astype(np.uint32))
                e.freeBatchMemory()
            out = (C.c_uint32 * 4)()
            assert e.L.mic_debug_stream_pool(out) == 0
            if i == 5:
                counts = (out[0], out[2])
            if i > 5:
                assert (out[0], out[2]) == counts, (i, list(out), counts)
    out = (C.c_uint32 * 4)()
    assert first is not None and e.L.mic_debug_stream_pool(out) == 0 and out[0] == out[1] and out[2] == out[3]     # all back in the pool
