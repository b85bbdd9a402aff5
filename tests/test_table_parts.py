"""Table-sharded runs on the RESIDENT partition (mic_db_set_part): part p of n of the database on its own
engine, all engines queried with the same reads, per-read rows summed (mergeKernel, CuClarkDB.cu:1321-1415) - equal to
one engine holding everything and to the oracle.  The super-k-mer layouts cut their table by slot range (a run of a read
belongs to one part), the other layouts by on-disk bucket range (CuClarkDB.cu:566-574).  Every comparison is bit-exact."""
import os

import numpy as np
import pytest

import golden_util as gu
from test_gpu_parity import _canonical_np, _kmer_reads, _oracle_results

pytestmark = pytest.mark.gpu
LAYOUT_ID = {"direct": 1, "minimizer": 2, "super": 3, "super2": 4}


@pytest.fixture(params=["direct", "minimizer", "super", "super2"])
def table_layout(request, monkeypatch):
    monkeypatch.setenv("MIC_LAYOUT", request.param)
    return request.param


@pytest.fixture(params=["super", "super2"])
def super_layout(request, monkeypatch):
    monkeypatch.setenv("MIC_LAYOUT", request.param)
    return request.param


def _engine(k, T, **kw):
    from cuclark_amd import MiClarkDB
    return MiClarkDB(k, T, **kw)


def _genome_db(rng, G, k, htsize, T, stride=40000):
    """Every k-mer of a random genome, labelled by position: the database the super-k-mer layouts are made for (long runs)."""
    codes = rng.integers(0, 4, G, dtype=np.uint8)
    v = np.zeros(G - k + 1, np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | codes[j:j + G - k + 1].astype(np.uint64)
    canon, first = np.unique(_canonical_np(v, k), return_index=True)
    lab = ((first // stride) % T).astype(np.uint16)
    order = np.lexsort((canon // np.uint64(htsize), canon % np.uint64(htsize)))
    canon, lab = canon[order], lab[order]
    sizes = np.bincount((canon % np.uint64(htsize)).astype(np.int64), minlength=htsize)
    assert sizes.max() < 256
    return codes, sizes.astype(np.uint8), (canon // np.uint64(htsize)).astype(np.uint64), lab


def _reads_from(rng, codes, n, L, sub=0.01, n_rate=0.002, random_frac=0.2):
    ascii_of = np.frombuffer(b"TGCA", np.uint8)
    recs = []
    for i in range(n):
        ln = int(rng.integers(L // 2, L + 1))
        if rng.random() < random_frac:
            seq = ascii_of[rng.integers(0, 4, ln)].copy()
        else:
            p = int(rng.integers(0, codes.size - ln))
            seq = ascii_of[codes[p:p + ln]].copy()
            if i % 2:
                seq = np.frombuffer(bytes(seq[::-1]).translate(bytes.maketrans(b"ACGT", b"TGCA")), np.uint8).copy()
            mut = rng.random(ln) < sub
            seq[mut] = ascii_of[rng.integers(0, 4, int(mut.sum()))]
        seq[rng.random(ln) < n_rate] = ord("N")
        recs.append(b">r%d\n" % i + seq.tobytes() + b"\n")
    return b"".join(recs)


def _oracle_part(odb, info, k, rp, cont, T):
    """what ONE part must answer, from the oracle: the super-k-mer layouts cut by resident slot range (oracle/part_rule.c restates
    the rule), the other layouts by on-disk bucket range (CuClarkDB.cu:1272-1274)"""
    if info["layout"] in (3, 4):
        counts, bad = odb.query_batch_slot_part(k, info["minimizer_len"], info["layout"] == 4, info["n_slots_whole"], info["part"],
                                                info["n_parts"], rp, cont, T)
    else:
        counts, bad = odb.query_batch(k, rp, cont, T, (info["shard_start"], info["shard_end"]))
    assert bad == 0
    return gu.oracle().result_from_counts(counts)


def _merge_rows(k, T, row_words, rows_list, n):
    """mergeKernel + resultKernel on the device (mic_merge_rows_device / mic_result_from_rows_device)."""
    import torch
    dev = torch.device("cuda:0")
    with _engine(k, T, row_words=row_words) as e:
        acc = torch.from_numpy(rows_list[0].astype(np.int64)).to(dev).to(torch.int32).contiguous()
        for rows in rows_list[1:]:
            nxt = torch.from_numpy(rows.astype(np.int64)).to(dev).to(torch.int32).contiguous()
            out = torch.empty_like(acc)
            torch.cuda.synchronize()
            e.merge_rows_device(acc.data_ptr(), nxt.data_ptr(), out.data_ptr(), n)
            e.sync()
            acc = out
        results = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        e.result_from_rows_device(acc.data_ptr(), results.data_ptr(), n)
        e.sync()
        return acc.cpu().numpy().view(np.uint32), results.cpu().numpy().view(np.uint32)


def _assert_rows_equal(a, b):
    assert (a[:, 0] == b[:, 0]).all()
    for r in range(a.shape[0]):
        m = int(a[r, 0])
        if m != 0xFFFFFFFF:
            assert (a[r, 1:1 + m] == b[r, 1:1 + m]).all(), r


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_parts_sum_to_the_whole_table_and_to_the_oracle(n_parts, table_layout):
    from cuclark_amd import host
    rng = np.random.default_rng(100 + n_parts)
    k, T, htsize = 31, 24, 1 << 18
    codes, sizes, keys, lab = _genome_db(rng, 300_000, k, htsize, T, stride=9000)
    data = _reads_from(rng, codes, 1500, 150)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    odb = gu.oracle().db_from_arrays(sizes, keys, lab)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    with _engine(k, T, row_words=32) as whole:
        whole.read_arrays(sizes, keys, lab)
        res_w, rows_w = whole.classify_packed(rp, cont, extended=True)
        info_w = whole.info()
    assert (res_w[:, :5] == expect).all() and info_w["n_parts"] == 0 and info_w["layout"] == LAYOUT_ID[table_layout]
    parts, infos, hits = [], [], np.zeros(n, np.int64)
    for p in range(n_parts):
        with _engine(k, T, row_words=32) as e:
            e.set_part(p, n_parts)
            e.read_arrays(sizes, keys, lab)
            res, rows = e.classify_packed(rp, cont, extended=True)
            infos.append(e.info())
        assert (res[:, :5] == _oracle_part(odb, infos[-1], k, rp, cont, T)).all(), p      # every part on its own, too
        parts.append(rows)
        hits += res[:, 0]
    assert (hits == expect[:, 0]).all()                 # every k-mer occurrence is counted by exactly one part
    merged, results = _merge_rows(k, T, 32, parts, n)
    _assert_rows_equal(merged, rows_w)
    assert (results[:, :5] == expect).all()
    for p, i in enumerate(infos):
        assert (i["part"], i["n_parts"], i["layout"]) == (p, n_parts, LAYOUT_ID[table_layout])
    if table_layout in ("super", "super2"):
        # resident slot ranges tile the whole table's main slots; every part was built from the whole images
        assert all(i["n_slots_whole"] == info_w["n_slots_whole"] == info_w["n_slots"] - info_w["n_overflow"] for i in infos)
        assert infos[0]["part_slot_lo"] == 0 and infos[-1]["part_slot_hi"] == info_w["n_slots_whole"]
        assert all(a["part_slot_hi"] == b["part_slot_lo"] for a, b in zip(infos[:-1], infos[1:]))
        assert sum(i["n_entries"] for i in infos) == info_w["n_entries"]
        assert all((i["shard_start"], i["shard_end"]) == (0, htsize) for i in infos)
    else:
        assert infos[0]["shard_start"] == 0 and infos[-1]["shard_end"] == htsize
        assert all(a["shard_end"] == b["shard_start"] for a, b in zip(infos[:-1], infos[1:]))
        assert sum(i["n_elems"] for i in infos) == info_w["n_elems"]


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_parts_on_the_golden_reads(n_parts, table_layout):
    """The reference-written golden database and reads: merged parts -> the committed CSV, byte for byte."""
    from cuclark_amd import host
    k = 31
    db = gu.load_golden_db("light_k31_u64")
    names = gu.target_names()
    T = len(names)
    data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    sizes = gu.golden_sizes(db)
    parts = []
    for p in range(n_parts):
        with _engine(k, T) as e:
            e.set_part(p, n_parts)
            e.read_arrays(sizes, db["ky"], db["lb"])
            parts.append(e.classify_packed(rp, cont, extended=True)[1])
    merged, results = _merge_rows(k, T, 16, parts, n)
    text = host.format_csv(data, idx, results, names, k)
    assert text == open(os.path.join(gu.GOLDEN, "expected_k31_fa.csv"), "rb").read()


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_parts_through_the_batch_api(n_parts, table_layout):
    """mic_batch_merge_shards over engines that hold parts (the reference's multi-device flow, CuClarkDB.cu:934-1001)."""
    from cuclark_amd import MiClarkDB, host
    k = 31
    db = gu.load_golden_db("light_k31_u64")
    T = len(gu.target_names())
    data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    sizes = gu.golden_sizes(db)
    with MiClarkDB(k, T) as whole:
        whole.read_arrays(sizes, db["ky"], db["lb"])
        ref_res, ref_rows = whole.classify_packed(rp, cont, extended=True)
    engines = [MiClarkDB(k, T) for _ in range(n_parts)]
    try:
        for p, e in enumerate(engines):
            e.set_part(p, n_parts)
            e.read_arrays(sizes, db["ky"], db["lb"])
            bufs = e.malloc(n, n, max(cont.size, 1), [0, n], True)
            bufs["reads_pointer"][0][: n + 1] = rp
            bufs["containers"][0][: cont.size] = cont
            e.readyBatch(0, n, cont.size)
            e.queryBatch(0, True)
        MiClarkDB.merge_shards(engines, 0)
        res = engines[0]._bufs["results"].copy()
        rows = engines[0]._bufs["rows"].copy()
    finally:
        for e in engines:
            e.close()
    assert (res[:, :6] == ref_res[:, :6]).all()
    _assert_rows_equal(rows, ref_rows)


def test_long_reads_over_parts(super_layout):
    """40-kb reads (hundreds of full 128-k-mer chunks, rounds with more runs than one round stages) against 3 slot-range parts."""
    from cuclark_amd import host
    rng = np.random.default_rng(78)
    k, T, htsize = 31, 16, 1 << 20
    codes, sizes, keys, lab = _genome_db(rng, 1_200_000, k, htsize, T)
    ascii_of = np.frombuffer(b"TGCA", np.uint8)
    recs = []
    for i in range(24):
        p = int(rng.integers(0, codes.size - 40000))
        seq = ascii_of[codes[p:p + 40000]].copy()
        if i % 2:
            seq = np.frombuffer(bytes(seq[::-1]).translate(bytes.maketrans(b"ACGT", b"TGCA")), np.uint8).copy()
        mut = rng.random(seq.size) < 0.005
        seq[mut] = ascii_of[rng.integers(0, 4, int(mut.sum()))]
        recs.append(b">r%d\n" % i + seq.tobytes() + b"\n")
    data = b"".join(recs)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    odb = gu.oracle().db_from_arrays(sizes, keys, lab)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert expect[:, 0].min() > 20000
    parts, hits = [], np.zeros(n, np.int64)
    for p in range(3):
        with _engine(k, T, row_words=32) as e:
            e.set_part(p, 3)
            e.read_arrays(sizes, keys, lab)
            res, rows = e.classify_packed(rp, cont, extended=True)
        assert (res[:, 0] > 3000).all()               # a third of the runs each, give or take
        parts.append(rows)
        hits += res[:, 0]
    assert (hits == expect[:, 0]).all()
    merged, results = _merge_rows(k, T, 32, parts, n)
    assert (results[:, :5] == expect).all()


def test_tied_minimizers_and_dense_rows_over_parts(table_layout, monkeypatch):
    """Low-complexity k-mers (every window position ties on the minimizer order) and palindromic minimizers, where the tied
    positions may hash into DIFFERENT parts: the parts' counts must still add up to the oracle's whichever path a part takes -
    per-run kernel, per-k-mer kernel (MIC_S_PER_KMER) or the dense path (rows that do not fit: row_words = 3)."""
    from cuclark_amd import host
    import torch
    rng = np.random.default_rng(31)
    k, T, htsize = 31, 40, 100003
    o = gu.oracle()
    enc = lambda seq: int("".join(str("TGCA".index(ch)) for ch in seq), 4)
    seqs = ["A" * 31, "C" * 31, "AC" * 15 + "A", "ACG" * 10 + "A", "AAT" * 10 + "T", "ACGT" * 7 + "ACG", "TTGCAA" * 5 + "T"]
    for _ in range(40):
        off = int(rng.integers(0, 12))
        fl = "".join(rng.choice(list("ACGT"), 31))
        seqs.append(fl[:off] + "ACGT" * 5 + fl[off + 20:])
        seqs.append(fl[:off] + "GAATTC" * 3 + "GA" + fl[off + 20:])
    # long low-complexity stretches: every k-mer of them is in the database, under many tied positions
    rep = ["A" * 200, "AC" * 100, "AAT" * 70, "ACGT" * 50, "GAATTC" * 40, "AAAAAAAAAC" * 20]
    kms = {enc(q) for q in seqs}
    for r in rep:
        kms |= {enc(r[i:i + k]) for i in range(len(r) - k + 1)}
    extra = [int(v) for v in rng.integers(0, 1 << 62, 3000, dtype=np.uint64)]
    canon = sorted({o.canonical(v, k) for v in list(kms) + extra}, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    for c in canon:
        sizes[c % htsize] += 1
    keys = np.array([c // htsize for c in canon], dtype=np.uint64)
    labels = np.array([(c >> 3) % T for c in canon], dtype=np.uint16)
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
    reads = []
    for sq in seqs[:30] + rep:
        fl = "".join(rng.choice(list("ACGT"), 40))
        reads.append(fl[:20] + sq + fl[20:])
        reads.append("".join({"A": "T", "C": "G", "G": "C", "T": "A"}[ch] for ch in reversed(reads[-1])))
    reads.append("N".join(gu.kmer_to_ascii(c, k) for c in canon[::40]))            # many targets: register row overflows
    data = "".join(f">r{i}\n{sq}\n" for i, sq in enumerate(reads)).encode()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    dev = torch.device("cuda:0")
    d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
    d_ct = torch.zeros(cont.size + 64, dtype=torch.int16, device=dev)
    d_ct[:cont.size] = torch.from_numpy(cont.view(np.int16)).to(dev)
    for per_kmer in (False, True):
        if per_kmer:
            if table_layout not in ("super", "super2"):
                break
            monkeypatch.setenv("MIC_S_PER_KMER", "1")
        for n_parts in (2, 5):
            for row_words in (32, 3):
                hits, dense = np.zeros(n, np.int64), torch.zeros((n, T), dtype=torch.int32, device=dev)
                parts = []
                for p in range(n_parts):
                    with _engine(k, T, row_words=row_words) as e:
                        e.set_part(p, n_parts)
                        e.read_arrays(sizes.astype(np.uint8), keys, labels)
                        res, rows = e.classify_packed(rp, cont, extended=True)
                        assert (res[:, :5] == _oracle_part(odb, e.info(), k, rp, cont, T)).all(), (per_kmer, n_parts, row_words, p)
                        part = torch.zeros((n, T), dtype=torch.int32, device=dev)
                        e.count_dense_device(d_rp.data_ptr(), d_ct.data_ptr(), 0, n, part.data_ptr())
                        e.sync()
                    hits += res[:, 0]
                    dense += part
                    parts.append(rows)
                assert (hits == expect[:, 0]).all(), (per_kmer, n_parts, row_words)
                assert (dense.cpu().numpy().view(np.uint32) == counts).all()
                merged, results = _merge_rows(k, T, row_words, parts, n)
                fits = merged[:, 0] != 0xFFFFFFFF
                assert (results[fits, :5] == expect[fits]).all()
                assert fits.sum() >= (n - 1 if row_words == 32 else 1)


def test_set_part_contract():
    from cuclark_amd import MicError
    db = gu.load_golden_db("light_k31_u64")
    sizes = gu.golden_sizes(db)
    with _engine(31, 6) as e:
        with pytest.raises(MicError):
            e.set_part(3, 3)
        e.set_part(1, 2)
        with pytest.raises(MicError, match="exclude each other"):
            e.read_arrays(sizes, db["ky"], db["lb"], shard=(0, 1000))
        e.read_arrays(sizes, db["ky"], db["lb"])
        assert e.info()["n_parts"] == 2 and e.info()["part"] == 1
        with pytest.raises(MicError, match="before the database"):
            e.set_part(0, 2)
    with _engine(31, 6) as e:
        e.set_part(0, 1)                            # one part = the whole database
        e.read_arrays(sizes, db["ky"], db["lb"])
        assert e.info()["n_parts"] == 0


def test_parts_at_scale(super_layout):
    """200 M-k-mer table, 2 M reads: 8 slot-range parts built from the same images, all reads against each, rows summed on the
    device == the whole table's rows and results; per part ~1/8 of the hits, of the table and of the entries; the constructive
    known answer holds on the sum."""
    import ctypes as C
    import torch
    from cuclark_amd import _lib, MiClarkDB
    L = _lib.load()
    dev = torch.device("cuda:0")
    layout = LAYOUT_ID[super_layout]
    T, k, htsize, nt = 300, 31, 57777779, 200_000_000
    spec = _lib.MicSynthSpec(seed=21, htsize=htsize, genome_nt=nt, n_targets=T, n_genomes=2 * T, k=k, key_bytes=8)
    cap = nt + 1024
    d_sizes = torch.empty(htsize, dtype=torch.uint8, device=dev)
    d_keys = torch.empty(cap, dtype=torch.int64, device=dev)
    d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
    n_el = C.c_uint64(0)
    torch.cuda.synchronize()
    assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
    n_reads, read_len = 2_000_000, 150
    pitch = L.mic_synth_read_pitch(read_len, k)
    d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev)
    d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
    d_truth = torch.zeros(n_reads * 2, dtype=torch.int32, device=dev)
    assert L.mic_synth_reads_device(C.byref(spec), 5, n_reads, read_len, 0.2, 0.01, 0.002, d_rp.data_ptr(), d_cont.data_ptr(),
                                    d_cont.numel(), d_truth.data_ptr(), None) == 0
    torch.cuda.synchronize()

    def run(part=None, n_parts=0):
        with MiClarkDB(k, T, layout=layout) as e:
            if n_parts:
                e.set_part(part, n_parts)
            e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), 8, d_labels.data_ptr())
            res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
            rows = torch.zeros((n_reads, 16), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, res.data_ptr(), rows.data_ptr())
            e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), res.data_ptr(), rows.data_ptr())
            e.sync()
            return res, rows, e.info(), e.last_query_ms()
    res_w, rows_w, info_w, ms_w = run()
    n_parts = 8
    acc, hits, infos, ms = None, torch.zeros(n_reads, dtype=torch.int64, device=dev), [], []
    with MiClarkDB(k, T) as m:
        for p in range(n_parts):
            res, rows, info, t = run(p, n_parts)
            infos.append(info)
            ms.append(t)
            hits += res[:, 0].to(torch.int64)
            if acc is None:
                acc = rows
            else:
                out = torch.empty_like(acc)
                torch.cuda.synchronize()
                m.merge_rows_device(acc.data_ptr(), rows.data_ptr(), out.data_ptr(), n_reads)
                m.sync()
                acc = out
        res2 = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
        m.result_from_rows_device(acc.data_ptr(), res2.data_ptr(), n_reads)
        m.sync()
    assert torch.equal(hits, res_w[:, 0].to(torch.int64))
    valid = (rows_w[:, 0] != -1) & (acc[:, 0] != -1)
    assert valid.float().mean() > 0.999
    assert torch.equal(acc[valid][:, 0], rows_w[valid][:, 0])
    nmax = int(rows_w[valid][:, 0].max())
    cols = torch.arange(16, device=dev)[None, :]
    used = cols <= rows_w[:, 0:1].clamp(min=0)
    assert torch.equal(torch.where(used, acc, 0)[valid], torch.where(used, rows_w, 0)[valid]) and nmax <= 15
    assert torch.equal(res2[valid][:, :5], res_w[valid][:, :5])
    assert sum(i["n_entries"] for i in infos) == info_w["n_entries"]
    assert max(i["hbm_bytes"] for i in infos) < info_w["hbm_bytes"] / n_parts * 1.25
    share = hits.sum().item() / max(int(res_w[:, 0].sum().item()), 1)
    assert share == 1.0
    truth = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
    r = res2.cpu().numpy().view(np.uint32)
    g = truth[:, 0] > 0
    okk = (truth[g, 1] == 0) | ((r[g, 1] == truth[g, 0]) & (r[g, 2] >= truth[g, 1]))
    assert okk.mean() > 0.999
    print(f"\n[{super_layout}] whole table {ms_w:.3f} ms; parts of 8: {' '.join(f'{t:.3f}' for t in ms)} ms")


@pytest.mark.parametrize("n_parts,owner", [(2, 1), (3, 0), (5, 3)])
def test_streaming_ingest_of_a_group_of_parts_equals_the_oracle_part_by_part(n_parts, owner, super_layout):
    """The command line's table-sharded path (mic_ingest_classify_group: what exe/cuCLARK --db-sharded runs per ingest slot): the
    bytes of a FASTA batch go to the slot's owner, every engine of the group probes the packed reads against ITS part - its partial
    rows must be exactly what oracle/part_rule.c says that part answers for -, the rows are summed read-range owned and the CSV
    the owner formats is the whole table's, byte for byte."""
    from cuclark_amd import MiClarkDB, host
    rng = np.random.default_rng(900 + n_parts)
    k, T, htsize = 31, 12, 1 << 18           # (12 targets: every summed row fits the 15 pairs of a partial row)
    codes, sizes, keys, lab = _genome_db(rng, 250_000, k, htsize, T, stride=7000)
    data = _reads_from(rng, codes, 2000, 150)
    names = [f"T{i}" for i in range(T)]
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    odb = gu.oracle().db_from_arrays(sizes, keys, lab)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    with _engine(k, T) as whole:
        whole.read_arrays(sizes, keys, lab)
        whole.ingest_alloc(1, 1 << 20, names, want_results=True)
        r_w = whole.ingest_classify(0, data)
        whole.ingest_free()
    assert r_w["status"] == 0 and (r_w["results"][:, :5] == expect).all()
    group = [_engine(k, T) for _ in range(n_parts)]
    try:
        for p, e in enumerate(group):
            e.set_part(p, n_parts)
            e.read_arrays(sizes, keys, lab)
        group[owner].ingest_alloc(2, 1 << 20, names, want_results=True)
        for rep in range(2):          # the slot and its buffers on the other engines are reused
            r = MiClarkDB.ingest_classify_group(group, owner, 1, data)
            assert r["status"] == 0 and r["n_reads"] == n
            assert r["csv"] == r_w["csv"] and (r["results"][:, :5] == expect).all()
        total = np.zeros((n, T), np.int64)
        for p, e in enumerate(group):
            info = e.info()
            want, bad = odb.query_batch_slot_part(k, info["minimizer_len"], info["layout"] == 4, info["n_slots_whole"], p, n_parts, rp, cont, T)
            assert bad == 0
            rows = group[owner].ingest_fetch_group_rows(1, p)
            got = np.zeros((n, T), np.int64)
            for i in range(n):
                m = int(rows[i, 0])
                assert m != 0xFFFFFFFF
                ent = rows[i, 1:1 + m]
                assert (np.diff((ent & 0xFFFF).astype(np.int64)) > 0).all()          # ascending targets (CuClarkDB.cu:1178-1243)
                got[i, ent & 0xFFFF] = ent >> 16
            assert (got == want).all(), p
            total += got
        assert (total == counts).all()
        # a batch with a read that hits more targets than a partial row holds is handed back (the caller's host path completes it)
        group[owner].ingest_free()
    finally:
        for e in group:
            e.close()
