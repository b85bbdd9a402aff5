"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden
vectors produced by the reference's own CPU hash table.  Integer work: every comparison is bit-exact."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["direct", "minimizer", "super", "super2"])
def table_layout(request, monkeypatch):
    """Every test runs against the resident layouts: one 64-byte slot per on-disk bucket, the
    minimizer-keyed 128-byte slots, the super-k-mer slots and their two-strand form.  The engine reads MIC_LAYOUT when a
    table is built."""
    monkeypatch.setenv("MIC_LAYOUT", request.param)
    return request.param


def _engine(k, n_targets, **kw):
    from cuclark_amd import MiClarkDB
    return MiClarkDB(k, n_targets, **kw)


def _kmer_reads(kmers, k):
    """One read per k-mer: a single part of k nucleotides."""
    n = len(kmers)
    per = 1 + (k + 7) // 8
    rp = (np.arange(n + 1, dtype=np.uint64) * per).astype(np.uint32)
    cont = np.zeros(n * per, np.uint16)
    cont[0::per] = k
    for j in range((k + 7) // 8):
        # container j holds nts [8j, 8j+8) of the k-mer, first nt in the top bits
        hi = 2 * k - 16 * j  # number of k-mer bits at or below this container's top
        vals = np.array([(int(v) << 16 >> hi) & 0xFFFF if hi >= 16 else (int(v) << (16 - hi)) & 0xFFFF for v in kmers],
                        dtype=np.uint16)
        cont[1 + j::per] = vals
    return rp, cont


def _oracle_results(odb, k, rp, cont, T, part=(0, None)):
    counts, bad = odb.query_batch(k, rp, cont, T, part)
    assert bad == 0
    return counts, gu.oracle().result_from_counts(counts)


# ------------------------------------------------------------------ golden vectors from the reference

@pytest.mark.parametrize("name", ["light_k27_u32", "light_k31_u64", "light_k20_u16", "light_k32_u64", "full_k31_u32"])
def test_golden_queries_from_files(name, db_dir):
    """Per-k-mer answers of the reference's queryElement(), DB loaded from .sz/.ky/.lb files."""
    prefix, meta = gu.materialize_db(name, db_dir)
    q = np.load(os.path.join(gu.GOLDEN, f"queries_{name}.npz"))
    k = meta["k"]
    T = len(gu.target_names())
    with _engine(k, T) as e:
        assert e.read(prefix, key_bytes=0)  # key width from the rule of main.cc:274-316
        info = e.info()
        assert info["htsize"] == meta["htsize"] and info["key_bytes"] == meta["key_bytes"]
        assert info["n_elems"] == meta["ky"].size
        assert info["layout"] == {"direct": 1, "minimizer": 2, "super": 3, "super2": 4}[os.environ["MIC_LAYOUT"]]
        rp, cont = _kmer_reads(q["kmers"], k)
        res = e.classify_packed(rp, cont)
    found = res[:, 0] == 1
    assert (res[:, 0] <= 1).all()
    assert (found == (q["found"] == 1)).all()
    assert (res[found, 1] == q["label"][found].astype(np.uint32) + 1).all()
    assert (res[found, 2] == 1).all() and (res[~found, 1] == 0).all()
    os.remove(prefix + ".sz")  # the full-size .sz is 1.6 GB


def test_golden_sampling(db_dir):
    name = "light_k27_u32"
    prefix, meta = gu.materialize_db(name, db_dir)
    q = np.load(os.path.join(gu.GOLDEN, f"queries_{name}.npz"))
    with _engine(27, 6) as e:
        assert e.read(prefix, sampling=3)
        rp, cont = _kmer_reads(q["kmers"], 27)
        res = e.classify_packed(rp, cont)
    found = res[:, 0] == 1
    assert (found == (q["found_s3"] == 1)).all()
    assert (res[found, 1] == q["label_s3"][found].astype(np.uint32) + 1).all()


def test_missing_db_files_return_false(db_dir):
    with _engine(31, 6) as e:
        assert e.read(os.path.join(db_dir, "does_not_exist")) is False
        with pytest.raises(Exception):
            e.classify_packed(np.zeros(2, np.uint32), np.zeros(4, np.uint16))


@pytest.mark.parametrize("case", [c[0] for c in gu.expected_csv_cases()])
def test_golden_csv(case, db_dir):
    """End to end: index + pack (C++ host) -> HIP query -> CSV, byte-identical to the committed CSV."""
    from cuclark_amd import host
    spec = {c[0]: c for c in gu.expected_csv_cases()}[case]
    _, k, dbname, data, paired, ext = spec
    names = gu.target_names()
    db = gu.load_golden_db(dbname)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    with _engine(k, len(names)) as e:
        e.read_arrays(gu.golden_sizes(db), db["ky"], db["lb"])
        out = e.classify_packed(rp, cont, extended=ext)
    res, rows = out if ext else (out, None)
    text = host.format_csv(data, idx, res, names, k, paired=paired, extended=ext, rows=rows)
    expected = open(os.path.join(gu.GOLDEN, f"expected_{case}.csv"), "rb").read()
    assert text == expected


def test_full_htsize_csv_equals_light(db_dir):
    """Same k-mer set at HTSIZE 1610612741 (u32 keys) and 57777779 (u64 keys): identical classification."""
    from cuclark_amd import host
    data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
    names = gu.target_names()
    db = gu.load_golden_db("full_k31_u32")
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], 31)
    with _engine(31, len(names)) as e:
        e.read_arrays(gu.golden_sizes(db), db["ky"], db["lb"])
        assert e.info()["slot_class"] == (32 if os.environ["MIC_LAYOUT"] == "direct" else 128)
        res = e.classify_packed(rp, cont)
    text = host.format_csv(data, idx, res, names, 31)
    assert text == open(os.path.join(gu.GOLDEN, "expected_k31_fa.csv"), "rb").read()


# ------------------------------------------------------------------ randomized parity vs the oracle

def _random_reads(rng, canon, k, n_reads, read_len, hit_frac=0.6, n_rate=0.01):
    """ASCII FASTA with reads stitched from DB k-mers (so they hit) and random sequence."""
    recs = []
    for i in range(n_reads):
        L = int(rng.integers(max(1, read_len // 2), read_len + 1))
        s = []
        while sum(len(x) for x in s) < L:
            if canon.size and rng.random() < hit_frac:
                km = gu.kmer_to_ascii(canon[int(rng.integers(canon.size))], k)
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


CASES = [
    # htsize, n_elems, k, key_bytes, n_labels, read_len
    (1009, 3000, 31, 8, 7, 150),        # tiny prime table, long chains, u64 keys
    (4096, 9000, 16, 4, 300, 120),      # power-of-two table (shift division), many labels
    (65537, 40000, 31, 8, 50, 150),
    (1 << 20, 200000, 25, 4, 4096, 150),
    (999983, 150000, 12, 2, 12, 100),   # u16 keys
    (57777779, 50000, 27, 4, 64, 150),  # CuCLARK-l table size
    (2, 200, 8, 8, 3, 40),              # two buckets
    (7919, 2000, 2, 2, 5, 30),          # k = 2
    (104729, 30000, 32, 8, 9, 200),     # k = 32
]


@pytest.mark.parametrize("htsize,n_elems,k,key_bytes,n_labels,read_len", CASES)
def test_random_db_parity(htsize, n_elems, k, key_bytes, n_labels, read_len):
    from cuclark_amd import host
    rng = np.random.default_rng(htsize * 31 + k)
    n_elems = min(n_elems, (1 << (2 * k)) // 3 if k < 16 else n_elems)
    sizes, keys, labels, canon = gu.random_db(rng, htsize, n_elems, k, key_bytes, n_labels)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    data = _random_reads(rng, canon, k, 400, read_len)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp, cont, n_labels)
    with _engine(k, n_labels, row_words=16) as e:
        e.read_arrays(sizes, keys, labels)
        res, rows = e.classify_packed(rp, cont, extended=True)
        info = e.info()
    assert info["n_elems"] == keys.size
    assert (res[:, :5] == expect).all()
    # sparse rows equal the oracle's (ascending targets) whenever they fit
    for r in range(counts.shape[0]):
        n, row = o.sparse_row(counts[r], 15)
        if n <= 15:
            assert rows[r, 0] == n
            got = rows[r, 1:1 + n]
            assert ((got & 0xFFFF) == row[1:1 + 2 * n:2]).all() and ((got >> 16) == row[2:2 + 2 * n:2]).all()
        else:
            assert rows[r, 0] == 0xFFFFFFFF and res[r, 6] & 1


def test_shards_sum_to_whole():
    """DB sharded by bucket range (CuClarkDB.cu:566-574,1272): merged shard rows == whole-table rows."""
    from cuclark_amd import host
    import torch
    rng = np.random.default_rng(5)
    htsize, k, T = 30011, 31, 40
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 60000, k, 8, T)
    data = _random_reads(rng, canon, k, 300, 150)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    bounds = [0, 7000, 7001, 20000, htsize]
    shard_rows = []
    with _engine(k, T, row_words=32) as whole:
        whole.read_arrays(sizes, keys, labels)
        res_whole, rows_whole = whole.classify_packed(rp, cont, extended=True)
    for a, b in zip(bounds[:-1], bounds[1:]):
        with _engine(k, T, row_words=32) as e:
            e.read_arrays(sizes, keys, labels, shard=(a, b))
            res, rows = e.classify_packed(rp, cont, extended=True)
            counts, expect = _oracle_results(odb, k, rp, cont, T, part=(a, b))
            assert (res[:, :5] == expect).all()
            shard_rows.append(rows)
    # merge on the GPU with the engine's merge kernel, then result kernel
    with _engine(k, T, row_words=32) as e:
        dev = torch.device("cuda:0")
        acc = torch.from_numpy(shard_rows[0].astype(np.int64)).to(dev).to(torch.int32).contiguous()
        for rows in shard_rows[1:]:
            nxt = torch.from_numpy(rows.astype(np.int64)).to(dev).to(torch.int32).contiguous()
            out = torch.empty_like(acc)
            torch.cuda.synchronize()
            e.merge_rows_device(acc.data_ptr(), nxt.data_ptr(), out.data_ptr(), n)
            e.sync()
            acc = out
        results = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        e.result_from_rows_device(acc.data_ptr(), results.data_ptr(), n)
        e.sync()
        merged = acc.cpu().numpy().view(np.uint32)
        results = results.cpu().numpy().view(np.uint32)
    assert (merged[:, 0] == rows_whole[:, 0]).all()
    for r in range(n):  # words beyond n are unspecified
        assert (merged[r, 1:1 + merged[r, 0]] == rows_whole[r, 1:1 + rows_whole[r, 0]]).all()
    assert (results[:, :5] == res_whole[:, :5]).all()


def test_many_targets_dense_path():
    """Reads hitting more than 64 distinct targets (register row overflow) and more than the row pitch."""
    rng = np.random.default_rng(11)
    htsize, k, T = 200003, 21, 500
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 20000, k, 4, T)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    # read 0: 300 DB k-mers separated by N -> ~250 distinct targets; read 1: 30 k-mers; read 2: nothing
    picks = rng.choice(canon.size, 300, replace=False)
    r0 = "N".join(gu.kmer_to_ascii(canon[i], k) for i in picks)
    r1 = "N".join(gu.kmer_to_ascii(canon[i], k) for i in picks[:30])
    data = f">many\n{r0}\n>some\n{r1}\n>none\n{'ACGT' * 20}\n".encode()
    from cuclark_amd import host
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert (counts[0] > 0).sum() > 64
    with _engine(k, T, row_words=16) as e:
        e.read_arrays(sizes, keys, labels)
        res, rows = e.classify_packed(rp, cont, extended=True)
        plain = e.classify_packed(rp, cont)
    assert (res[:, :5] == expect).all() and (plain[:, :5] == expect).all()
    assert res[0, 6] & 2 and rows[0, 0] == 0xFFFFFFFF      # dense path, row does not fit
    assert res[0, 5] == (counts[0] > 0).sum()
    assert res[1, 6] & 1                                    # fits the register row but not 15 pairs


def test_dense_counts_of_shards_sum_to_the_exact_result():
    """What the table-sharded ranks do for a read whose merged row overflows (cuclark_amd/multi.py: complete_overflowed):
    dense counts per bucket range (mic_count_dense_device), summed, finished by mic_result_from_dense_device."""
    import torch
    rng = np.random.default_rng(12)
    htsize, k, T = 200003, 21, 300
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 20000, k, 4, T)
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    picks = rng.choice(canon.size, 200, replace=False)
    reads = ["N".join(gu.kmer_to_ascii(canon[i], k) for i in picks[a:b]) for a, b in ((0, 200), (0, 20), (50, 90), (10, 12))]
    data = "".join(f">r{i}\n{r}\n" for i, r in enumerate(reads)).encode()
    from cuclark_amd import host
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    dev = torch.device("cuda:0")
    d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
    d_ct = torch.zeros(cont.size + 64, dtype=torch.int16, device=dev)
    d_ct[:cont.size] = torch.from_numpy(cont.view(np.int16)).to(dev)
    ids = torch.tensor([0, 2, 3], dtype=torch.int32, device=dev)
    total = torch.zeros((3, T), dtype=torch.int32, device=dev)
    for s0, s1 in ((0, 70000), (70000, 150000), (150000, htsize)):
        with _engine(k, T, row_words=16) as e:
            e.read_arrays(sizes, keys, labels, shard=(s0, s1))
            part = torch.zeros((3, T), dtype=torch.int32, device=dev)
            e.count_dense_device(d_rp.data_ptr(), d_ct.data_ptr(), ids.data_ptr(), 3, part.data_ptr())
            e.sync()
            total += part
    assert (total.cpu().numpy().view(np.uint32) == counts[[0, 2, 3]]).all()
    res = torch.zeros((4, 8), dtype=torch.int32, device=dev)
    with _engine(k, T, row_words=16) as e:
        e.result_from_dense_device(total.data_ptr(), ids.data_ptr(), 3, res.data_ptr())
        e.sync()
    got = res.cpu().numpy().view(np.uint32)
    assert (got[[0, 2, 3], :5] == expect[[0, 2, 3]]).all() and (got[1] == 0).all()


def test_long_reads_and_part_splitting():
    """A 200 kb sequence (parts longer than 65528 nt are split with k-1 overlap) and a multi-chunk read."""
    from cuclark_amd import host
    rng = np.random.default_rng(3)
    htsize, k, T = 1 << 16, 31, 20
    genome = "".join(rng.choice(list("ACGT"), 200000))
    o = gu.oracle()
    # DB = every 3rd k-mer of the genome
    canon = sorted({o.canonical(int(v), k) for v in
                    [int("".join(str("TGCA".index(c)) for c in genome[i:i + k]), 4) for i in range(0, 199000, 3)]},
                   key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    keep = []
    for c in canon:
        if sizes[c % htsize] < 255:
            sizes[c % htsize] += 1
            keep.append(c)
    keys = np.array([c // htsize for c in keep], dtype=np.uint64)
    labels = rng.integers(0, T, len(keep)).astype(np.uint16)
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
    wrapped = "\n".join(genome[i:i + 70] for i in range(0, len(genome), 70))
    data = f">genome\n{wrapped}\n>mid\n{genome[5000:5700]}\n>withN\n{genome[100:400]}N{genome[9000:9030]}N{genome[400:900]}\n".encode()
    idx = host.index_reads(data)
    assert idx["length"][0] == 200000
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    rp_o, cont_o = o.pack_batch(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    assert (rp == rp_o).all() and (cont == cont_o).all()
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    ascii_counts = odb.count_read_ascii(k, data[int(idx["seq_s"][0]):int(idx["seq_e"][0])], 200000, T)
    assert (ascii_counts == counts[0]).all()
    with _engine(k, T) as e:
        e.read_arrays(sizes.astype(np.uint8), keys, labels)
        res = e.classify_packed(rp, cont)
    assert (res[:, :5] == expect).all()
    assert expect[0, 0] > 60000


@pytest.mark.parametrize("k", [31, 27, 24])
def test_few_wavefronts_many_reads_of_every_shape(k, monkeypatch):
    """The per-run kernel's loop is software-pipelined across the reads of a wavefront: a read that is one part of at most 128 k-mers
    has its slot loads in flight during the next read's front half; every other read (several parts, several chunks, no k-mer)
    drains the pipeline and takes the plain road.  The small cases above give a wavefront one read at most; here MIC_QUERY_BLOCKS (test
    hook) cuts the grid to 2 and 5 blocks, so that each wavefront runs hundreds of reads of mixed shapes back to back - every
    transition between the two roads, the first and the last read of a wavefront - against the oracle.  k = 24: the instantiation
    with k and m from the table (plain loop)."""
    from cuclark_amd import host
    rng = np.random.default_rng(1000 + k)
    htsize, T = 65537, 40
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 60000, k, 8, T)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    # lengths 10 .. 320: no k-mer / one chunk / two and three chunks; N every ~100 nt in a third of the reads (several parts)
    recs = []
    for part in range(6):
        recs.append(_random_reads(rng, canon, k, 500, (20, 150, 157, 158, 320, 150)[part], hit_frac=0.7, n_rate=(0.0, 0.0, 0.01, 0.0, 0.01, 0.0)[part]))
    lines = b"".join(recs).split(b">")[1:]
    order = rng.permutation(len(lines))
    data = b"".join(b">" + lines[i] for i in order)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert (expect[:, 0] > 0).sum() > 1500
    with _engine(k, T) as e:
        e.read_arrays(sizes, keys, labels)
        base = e.classify_packed(rp, cont)
        assert (base[:, :5] == expect).all()
        for blocks in ("2", "5"):
            monkeypatch.setenv("MIC_QUERY_BLOCKS", blocks)
            res = e.classify_packed(rp, cont)
            monkeypatch.delenv("MIC_QUERY_BLOCKS")
            assert (res == base).all(), blocks


def _canonical_np(v, k):
    """canonical k-mer values of a uint64 array (A=3 C=2 G=1 T=0: the complement is the bitwise NOT)"""
    x = ~v
    x = ((x >> np.uint64(2)) & np.uint64(0x3333333333333333)) | ((x & np.uint64(0x3333333333333333)) << np.uint64(2))
    x = ((x >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F)) | ((x & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4))
    x = ((x >> np.uint64(8)) & np.uint64(0x00FF00FF00FF00FF)) | ((x & np.uint64(0x00FF00FF00FF00FF)) << np.uint64(8))
    x = ((x >> np.uint64(16)) & np.uint64(0x0000FFFF0000FFFF)) | ((x & np.uint64(0x0000FFFF0000FFFF)) << np.uint64(16))
    x = (x >> np.uint64(32)) | (x << np.uint64(32))
    rc = x >> np.uint64(64 - 2 * k)
    return np.minimum(v, rc)


def test_thousands_of_chunks_of_long_reads():
    """2.4 M nucleotides of 40-kb reads cut from a 2.5-Mb genome whose k-mers are the database: 19 000 full 128-k-mer chunks,
    among them the few with more than 32 super-k-mers (more runs than one round of staged slots holds) and every strand and
    alignment of a run against its entry; substitutions every ~200 nt cut runs at all positions."""
    from cuclark_amd import host
    rng = np.random.default_rng(77)
    htsize, k, T = 1 << 20, 31, 16
    G = 2_500_000
    codes = rng.integers(0, 4, G, dtype=np.uint8)             # 0..3 = T G C A (the packed code)
    v = np.zeros(G - k + 1, np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | codes[j:j + G - k + 1].astype(np.uint64)
    canon, first = np.unique(_canonical_np(v, k), return_index=True)
    lab = ((first // 40000) % T).astype(np.uint16)
    order = np.lexsort((canon // np.uint64(htsize), canon % np.uint64(htsize)))
    canon, lab = canon[order], lab[order]
    sizes = np.bincount((canon % np.uint64(htsize)).astype(np.int64), minlength=htsize)
    assert sizes.max() < 256
    keys = (canon // np.uint64(htsize)).astype(np.uint64)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, lab)
    ascii_of = np.frombuffer(b"TGCA", np.uint8)
    recs = []
    for i in range(60):
        p = int(rng.integers(0, G - 40000))
        seq = ascii_of[codes[p:p + 40000]].copy()
        if i % 2:
            seq = np.frombuffer(bytes(seq[::-1]).translate(bytes.maketrans(b"ACGT", b"TGCA")), np.uint8).copy()
        mut = rng.random(seq.size) < 0.005
        seq[mut] = ascii_of[rng.integers(0, 4, int(mut.sum()))]
        recs.append(b">r%d\n" % i + seq.tobytes() + b"\n")
    data = b"".join(recs)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp, cont, T)
    assert expect[:, 0].min() > 20000
    with _engine(k, T) as e:
        e.read_arrays(sizes.astype(np.uint8), keys, lab)
        res = e.classify_packed(rp, cont)
    assert (res[:, :5] == expect).all()


def test_malformed_buckets_follow_reference_scan():
    """Unsorted buckets / duplicate keys: the table must answer exactly like the reference's linear scan."""
    rng = np.random.default_rng(9)
    htsize, k, T = 251, 20, 9
    sizes = rng.integers(0, 30, htsize).astype(np.uint8)
    n = int(sizes.sum())
    keys = rng.integers(0, 60, n).astype(np.uint32)  # small range: many duplicates, unsorted
    labels = rng.integers(0, T, n).astype(np.uint16)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    kmers = []
    for r in range(htsize):
        for qv in range(0, 62):
            c = qv * htsize + r
            if o.canonical(c, k) == c:
                kmers.append(c)
    kmers = np.array(kmers[:20000], dtype=np.uint64)
    rp, cont = _kmer_reads(kmers, k)
    f, l = odb.find_many(kmers, k)
    with _engine(k, T) as e:
        e.read_arrays(sizes, keys, labels)
        res = e.classify_packed(rp, cont)
    assert ((res[:, 0] == 1) == (f == 1)).all()
    assert (res[f == 1, 1] == l[f == 1].astype(np.uint32) + 1).all()
    assert f.sum() > 100


def test_batches_concurrent_and_order():
    """Several batches in flight (CuCLARK_hh.hh:1616-1760 usage): results land at the global read index."""
    from cuclark_amd import host, MiClarkDB
    rng = np.random.default_rng(21)
    htsize, k, T = 10007, 25, 30
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 20000, k, 8, T)
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    data = _random_reads(rng, canon, k, 1000, 120)
    idx = host.index_reads(data)
    nb = 4
    cuts = [0, 200, 500, 501, 1000]
    packed = [host.pack_reads(data, idx["seq_s"][a:b], idx["seq_e"][a:b], idx["length"][a:b], k) for a, b in zip(cuts[:-1], cuts[1:])]
    with MiClarkDB(k, T, num_batches=nb) as e:
        e.read_arrays(sizes, keys, labels)
        bufs = e.malloc(1000, 499, max(c.size for _, c in packed), cuts)
        for b, (rp, ct) in enumerate(packed):
            bufs["reads_pointer"][b][:rp.size] = rp
            bufs["containers"][b][:ct.size] = ct
            e.readyBatch(b, rp.size - 1, ct.size)
        for b in range(nb):
            e.queryBatch(b)
        for b in reversed(range(nb)):
            e.waitForBatch(b)
            assert e.checkBatch(b)
        res = bufs["results"].copy()
        e.freeBatchMemory()
    rp_all, ct_all = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    _, expect = _oracle_results(odb, k, rp_all, ct_all, T)
    assert (res[:, :5] == expect).all()


def test_synthetic_generator_matches_oracle():
    """mic_synth_* (bench workload): DB arrays + packed reads made in HBM; GPU results == oracle on them."""
    import ctypes as C
    import torch
    from cuclark_amd import _lib, MiClarkDB
    L = _lib.load()
    dev = torch.device("cuda:0")
    spec = _lib.MicSynthSpec(seed=7, htsize=500009, genome_nt=600000, n_targets=37, n_genomes=60, k=31, key_bytes=8)
    cap = 700000
    d_sizes = torch.zeros(spec.htsize, dtype=torch.uint8, device=dev)
    d_keys = torch.zeros(cap, dtype=torch.int64, device=dev)
    d_labels = torch.zeros(cap, dtype=torch.int16, device=dev)
    n_el = C.c_uint64(0)
    torch.cuda.synchronize()
    assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
    n_el = n_el.value
    assert 590000 < n_el <= 600000
    n_reads, read_len = 5000, 150
    pitch = L.mic_synth_read_pitch(read_len, 31)
    d_rp = torch.zeros(n_reads + 1, dtype=torch.int32, device=dev)
    d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
    d_truth = torch.zeros(n_reads * 2, dtype=torch.int32, device=dev)
    assert L.mic_synth_reads_device(C.byref(spec), 99, n_reads, read_len, 0.2, 0.01, 0.002, d_rp.data_ptr(), d_cont.data_ptr(),
                                    d_cont.numel(), d_truth.data_ptr(), None) == 0
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy()
    keys = d_keys[:n_el].cpu().numpy().view(np.uint64)
    labels = d_labels[:n_el].cpu().numpy().view(np.uint16)
    rp = d_rp.cpu().numpy().view(np.uint32)
    cont = d_cont.cpu().numpy().view(np.uint16)
    truth = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
    assert int(sizes.sum()) == n_el
    o = gu.oracle()
    odb = o.db_from_arrays(sizes, keys, labels)
    _, expect = _oracle_results(odb, 31, rp, cont, 37)
    with MiClarkDB(31, 37) as e:
        e.read_device(d_sizes.data_ptr(), spec.htsize, d_keys.data_ptr(), 8, d_labels.data_ptr())
        d_res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, d_res.data_ptr())
        assert e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), d_res.data_ptr()) == 0
        e.sync()
        assert e.last_query_ms() > 0
        res = d_res.cpu().numpy().view(np.uint32)
    assert (res[:, :5] == expect).all()
    # constructive known answer: genome reads hit their own genome's label for every unmodified window
    g = truth[:, 0] > 0
    assert 0.7 < g.mean() < 0.9
    ok = (truth[g, 1] == 0) | ((res[g, 1] == truth[g, 0]) & (res[g, 2] >= truth[g, 1]))
    assert ok.mean() > 0.999
    assert (res[~g, 0] == 0).mean() > 0.99


def test_skewed_minimizer_bucket():
    """Thousands of k-mers that share a 25-nt core (hence, for about half of them, the minimizer): in the minimizer
    layout they pile into one bucket, which becomes a multi-level directory tree; results must stay exact."""
    import itertools
    rng = np.random.default_rng(17)
    k, T, htsize = 31, 11, 1000003
    o = gu.oracle()
    core = "".join(rng.choice(list("ACGT"), 25))
    kmers = []
    for tail in itertools.product("ACGT", repeat=6):
        s = core + "".join(tail)
        kmers.append(int("".join(str("TGCA".index(ch)) for ch in s), 4))
    extra = [int(rng.integers(0, 1 << 62, dtype=np.uint64)) for _ in range(3000)]
    canon = sorted({o.canonical(v, k) for v in kmers + extra}, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    for c in canon:
        sizes[c % htsize] += 1
    assert sizes.max() < 255
    keys = np.array([c // htsize for c in canon], dtype=np.uint64)
    labels = rng.integers(0, T, len(canon)).astype(np.uint16)
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
    # queries: every stored k-mer in both orientations, plus neighbours that are absent
    q = np.array(kmers + [o.revcomp(v, k) for v in kmers[::3]] + [v ^ 1 for v in kmers[::5]] + extra[:500], dtype=np.uint64)
    rp, cont = _kmer_reads(q, k)
    f, l = odb.find_many(q, k)
    with _engine(k, T) as e:
        e.read_arrays(sizes.astype(np.uint8), keys, labels)
        info = e.info()
        res = e.classify_packed(rp, cont)
    if info["layout"] == 2:
        assert info["max_chain"] > 500           # one bucket holds a large share of the 4096 core k-mers (3-level tree)
    if info["layout"] in (3, 4):
        # random labels keep the k-mers of a minimizer apart - hundreds of entries under one sort key: they leave the chain for
        # the side table, what stays is short
        assert info["side_kmers"] > 1000 and info["max_chain"] <= 40
    assert ((res[:, 0] == 1) == (f == 1)).all()
    assert (res[f == 1, 1] == l[f == 1].astype(np.uint32) + 1).all()
    assert f[:4096].all()


def test_default_layout_follows_the_database(monkeypatch, table_layout):
    """Nobody asks for a layout (MIC_LAYOUT unset, cfg.layout 0): super-k-mer slots - also for unrelated k-mers (one per entry:
    cuCLARK-l's sampled blocks, or one low-complexity core in thousands of contexts with unrelated labels) as long as the table
    is a small one; a LARGE table of unrelated k-mers (here: MIC_S_SMALL_TABLE_GB=0) -> the minimizer layout, a quarter of the
    memory.  Same answers either way.  (Crowded minimizers alone do not change the layout: their k-mers go to
    the side table, tests/test_crowded.py.)"""
    if table_layout != "super":
        pytest.skip("one run is enough")
    monkeypatch.delenv("MIC_LAYOUT")
    import itertools
    rng = np.random.default_rng(29)
    k, T, htsize = 31, 7, 3000017
    o = gu.oracle()
    code = lambda s_: int("".join(str("TGCA".index(ch)) for ch in s_), 4)
    genome = "".join(rng.choice(list("ACGT"), 60000))
    genome_kmers = [code(genome[i:i + k]) for i in range(len(genome) - k + 1)]
    unrelated = [int(v) for v in rng.integers(0, 1 << 62, 40000, dtype=np.uint64)]
    core = "AT" * 12 + "A"                                      # a low-complexity stretch of 25 nt in 4096 contexts
    crowded = [code(core + "".join(t)) for t in itertools.product("ACGT", repeat=6)] + [code(genome[i:i + k]) for i in range(0, 3000)]
    for name, kmers, want in (("genome", genome_kmers, 3), ("unrelated", unrelated, 3), ("crowded", crowded, 3),
                              ("unrelated, large", unrelated, 2), ("genome, large", genome_kmers, 3)):
        if name.endswith("large"):
            monkeypatch.setenv("MIC_S_SMALL_TABLE_GB", "0")
        name = name.split(",")[0]
        canon = sorted({o.canonical(v, k) for v in kmers}, key=lambda c: (c % htsize, c // htsize))
        sizes = np.zeros(htsize, np.int64)
        for c in canon:
            sizes[c % htsize] += 1
        keys = np.array([c // htsize for c in canon], dtype=np.uint64)
        # one target for the genome (adjacent k-mers merge into super-k-mers); unrelated labels keep the other contexts apart
        labels = np.array([3 if name == "genome" else (c * 2654435761 >> 11) % T for c in canon], dtype=np.uint16)
        odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
        q = np.array(kmers[::3] + [o.revcomp(v, k) for v in kmers[1::7]] + [v ^ 2 for v in kmers[::11]], dtype=np.uint64)
        rp, cont = _kmer_reads(q, k)
        f, l = odb.find_many(q, k)
        with _engine(k, T) as e:
            e.read_arrays(sizes.astype(np.uint8), keys, labels)
            info = e.info()
            res = e.classify_packed(rp, cont)
        assert info["layout"] == want, (name, info["layout"], info["reserved"], info["n_elems"], info["n_entries"])
        assert ((res[:, 0] == 1) == (f == 1)).all()
        assert (res[f == 1, 1] == l[f == 1].astype(np.uint32) + 1).all()


def test_low_complexity_and_palindromic_minimizers():
    """k-mers whose m-mers repeat (poly-A, dinucleotide and trinucleotide repeats: every window position ties) or whose
    minimizer is its own reverse complement ((ACGT)n): the minimizer-keyed layouts must find them from either strand of a
    read, and must not find their one-nucleotide neighbours."""
    rng = np.random.default_rng(31)
    k, T, htsize = 31, 5, 100003
    o = gu.oracle()
    enc = lambda seq: int("".join(str("TGCA".index(ch)) for ch in seq), 4)
    seqs = ["A" * 31, "C" * 31, "AC" * 15 + "A", "ACG" * 10 + "A", "AAT" * 10 + "T", "ACGT" * 7 + "ACG", "TTGCAA" * 5 + "T"]
    for _ in range(40):                       # palindromic 20-mers inside random flanks, at every offset
        off = int(rng.integers(0, 12))
        fl = "".join(rng.choice(list("ACGT"), 31))
        seqs.append(fl[:off] + "ACGT" * 5 + fl[off + 20:])
        seqs.append(fl[:off] + "GAATTC" * 3 + "GA" + fl[off + 20:])
    kmers = [enc(q) for q in seqs]
    extra = [int(v) for v in rng.integers(0, 1 << 62, 500, dtype=np.uint64)]
    canon = sorted({o.canonical(v, k) for v in kmers + extra}, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    for c in canon:
        sizes[c % htsize] += 1
    keys = np.array([c // htsize for c in canon], dtype=np.uint64)
    labels = np.array([(c >> 3) % T for c in canon], dtype=np.uint16)
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
    q = np.array(kmers + [o.revcomp(v, k) for v in kmers] + [v ^ 1 for v in kmers] + [v ^ (3 << 40) for v in kmers] + extra[:50],
                 dtype=np.uint64)
    rp, cont = _kmer_reads(q, k)
    f, l = odb.find_many(q, k)
    assert f[:2 * len(kmers)].all()
    with _engine(k, T) as e:
        e.read_arrays(sizes.astype(np.uint8), keys, labels)
        res = e.classify_packed(rp, cont)
    assert ((res[:, 0] == 1) == (f == 1)).all()
    assert (res[f == 1, 1] == l[f == 1].astype(np.uint32) + 1).all()
    # the same k-mers inside longer reads (the sliding window sees them next to unrelated m-mers), both strands
    reads = []
    for sq in seqs[:20]:
        fl = "".join(rng.choice(list("ACGT"), 40))
        reads.append(fl[:20] + sq + fl[20:])
        reads.append("".join({"A": "T", "C": "G", "G": "C", "T": "A"}[ch] for ch in reversed(reads[-1])))
    from cuclark_amd import host
    data = "".join(f">r{i}\n{sq}\n" for i, sq in enumerate(reads)).encode()
    idx = host.index_reads(data)
    rp2, cont2 = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    counts, expect = _oracle_results(odb, k, rp2, cont2, T)
    with _engine(k, T) as e:
        e.read_arrays(sizes.astype(np.uint8), keys, labels)
        res2 = e.classify_packed(rp2, cont2)
    assert (res2[:, :5] == expect).all() and (expect[:, 0] >= 1).all()


def test_super_table_with_crowded_slots(monkeypatch):
    """The super-k-mer table at 5.9 entries per 6-entry slot instead of 1.5 (MIC_SSLOT_LOAD): most slots continue in a
    chain of further slots; unrelated k-mers (one entry each) and k-mers cut from genomes (shared entries); answers equal
    the sparse table's and the oracle's."""
    if os.environ["MIC_LAYOUT"] not in ("super", "super2"):
        pytest.skip("sizing of the super-k-mer table")
    rng = np.random.default_rng(29)
    k, T, htsize = 31, 9, 2000003
    o = gu.oracle()
    genome = rng.integers(0, 4, 60000)
    gk = set()
    v = 0
    for i, nt in enumerate(genome):
        v = ((v << 2) | int(nt)) & ((1 << 62) - 1)
        if i >= k - 1 and (i // 97) % 3:           # runs of present k-mers with gaps (presence masks with holes)
            gk.add(o.canonical(v, k))
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 60000, k, 8, T)
    allk = sorted(set(int(c) for c in canon) | gk, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    for c in allk:
        sizes[c % htsize] += 1
    keys = np.array([c // htsize for c in allk], dtype=np.uint64)
    labels = np.array([(c >> 7) % T for c in allk], dtype=np.uint16)
    odb = o.db_from_arrays(sizes.astype(np.uint8), keys, labels)
    q = np.array(allk[::3] + [o.revcomp(c, k) for c in allk[1::11]] + [c ^ 3 for c in allk[::13]], dtype=np.uint64)
    q = np.concatenate([q, rng.integers(0, 1 << 62, 3000, dtype=np.uint64)])
    rp, cont = _kmer_reads(q, k)
    f, l = odb.find_many(q, k)

    def run():
        with _engine(k, T) as e:
            e.read_arrays(sizes.astype(np.uint8), keys, labels)
            return e.info(), e.classify_packed(rp, cont)
    base_info, base = run()
    monkeypatch.setenv("MIC_SSLOT_LOAD", "5.9")
    info, res = run()
    assert info["layout"] in (3, 4) and info["n_slots"] < base_info["n_slots"] / 2 and info["n_overflow"] > base_info["n_overflow"]
    assert (res == base).all()
    assert ((res[:, 0] == 1) == (f == 1)).all()
    assert (res[f == 1, 1] == l[f == 1].astype(np.uint32) + 1).all()
    # the table is normally written in ONE pass with continuation slots from a pool; a pool that runs out (test hook) or
    # MIC_S_TWO_PASS falls back to counting the entries first: same slots, same chains, same answers
    for name, value in (("MIC_S_POOL_SLOTS", "8"), ("MIC_S_TWO_PASS", "1")):
        monkeypatch.setenv(name, value)
        info2, res2 = run()
        monkeypatch.delenv(name)
        assert info2["n_slots"] == info["n_slots"] and info2["n_overflow"] == info["n_overflow"] and info2["n_entries"] == info["n_entries"]
        assert info2["hbm_bytes"] < info["hbm_bytes"]          # no unused pool behind the table
        assert (res2 == base).all()
    # a staging area that holds a fraction of the candidates: the table is built in several passes over slot ranges and
    # must come out the same (MIC_S_STAGING_LIMIT_MB is the test hook; the headline's two-strand table needs this path)
    monkeypatch.delenv("MIC_SSLOT_LOAD")
    monkeypatch.setenv("MIC_S_STAGING_LIMIT_MB", "0.4")
    info3, res3 = run()
    assert info3["n_slots"] == base_info["n_slots"] and info3["n_entries"] == base_info["n_entries"]
    assert (res3 == base).all()
    monkeypatch.delenv("MIC_S_STAGING_LIMIT_MB")
    # the one-strand table's sorted build (candidates once, radix sort by slot hash; mic_build.hip: s_expand_kernel) against the
    # classic three-walk build, records sorted in many small chunks, and a list of tied candidates that runs over (test hook: the
    # build then takes the classic road): the same table, the same answers
    for name, value in (("MIC_S_CLASSIC", "1"), ("MIC_S_SORT_CHUNK", "4096"), ("MIC_S_EXTRA_CAP", "1")):
        monkeypatch.setenv(name, value)
        info4, res4 = run()
        monkeypatch.delenv(name)
        if info4["layout"] == 3:                 # which road the build took, from its report of stage times
            from cuclark_amd import _lib
            report = _lib.load().mic_db_last_build_report().decode()
            assert ("(sorted build)" in report) == (name == "MIC_S_SORT_CHUNK"), (name, report)
        assert info4["n_slots"] == base_info["n_slots"] and info4["n_entries"] == base_info["n_entries"] and info4["n_overflow"] == base_info["n_overflow"], name
        assert (res4 == base).all(), name


def test_table_adapts_to_the_free_hbm(monkeypatch):
    """A table that does not fit at half-full slots is built denser (exact counting pass per setting); when even the
    densest minimizer table does not fit and nobody asked for that layout, the engine falls back to the direct one.
    MIC_HBM_LIMIT_GB is the test hook that pretends less HBM is available.  Results stay identical."""
    if os.environ["MIC_LAYOUT"] != "minimizer":
        pytest.skip("sizing of the minimizer table")
    rng = np.random.default_rng(23)
    k, T, htsize = 31, 9, 2000003
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 150000, k, 8, T)
    q = np.concatenate([canon[::7], rng.integers(0, 1 << 62, 2000, dtype=np.uint64)])
    rp, cont = _kmer_reads(q, k)

    def run():
        with _engine(k, T) as e:
            e.read_arrays(sizes, keys, labels)
            return e.info(), e.classify_packed(rp, cont)
    base_info, base = run()
    assert base_info["layout"] == 2
    table_gb = base_info["hbm_bytes"] / 1e9
    monkeypatch.setenv("MIC_HBM_LIMIT_GB", repr(table_gb * 0.9))      # random k-mers: 7 per slot is ~11 % smaller than 6
    info, res = run()
    assert info["layout"] == 2 and info["n_slots"] < base_info["n_slots"] and info["hbm_bytes"] <= table_gb * 0.9e9
    assert (res == base).all()
    monkeypatch.setenv("MIC_HBM_LIMIT_GB", repr(table_gb * 0.05))
    with pytest.raises(Exception, match="minimizer table needs at least"):     # the layout was requested explicitly (MIC_LAYOUT)
        run()
    monkeypatch.delenv("MIC_LAYOUT")                                   # by default: fall back to the direct layout
    info, res = run()
    assert info["layout"] == 1
    assert (res[:, :5] == base[:, :5]).all()


def test_properties_at_scale():
    """200 M-k-mer table, 2 M reads (a scale the oracle cannot sweep in a test): the four table layouts give identical
    result rows and sparse rows; two bucket-range shards merged on the device equal the whole table; a second pass is
    bit-identical; the oracle agrees on a 20 000-read sample; the constructive known answer holds."""
    if os.environ["MIC_LAYOUT"] != "minimizer":
        pytest.skip("one run covers both layouts")
    import ctypes as C
    import torch
    from cuclark_amd import _lib, MiClarkDB
    L = _lib.load()
    dev = torch.device("cuda:0")
    T, k, htsize, nt = 300, 31, 57777779, 200_000_000
    spec = _lib.MicSynthSpec(seed=21, htsize=htsize, genome_nt=nt, n_targets=T, n_genomes=2 * T, k=k, key_bytes=8)
    cap = nt + 1024
    d_sizes = torch.empty(htsize, dtype=torch.uint8, device=dev)
    d_keys = torch.empty(cap, dtype=torch.int64, device=dev)
    d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
    n_el = C.c_uint64(0)
    torch.cuda.synchronize()
    assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
    n_el = n_el.value
    n_reads, read_len = 2_000_000, 150
    pitch = L.mic_synth_read_pitch(read_len, k)
    d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev)
    d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
    d_truth = torch.zeros(n_reads * 2, dtype=torch.int32, device=dev)
    assert L.mic_synth_reads_device(C.byref(spec), 5, n_reads, read_len, 0.2, 0.01, 0.002, d_rp.data_ptr(), d_cont.data_ptr(),
                                    d_cont.numel(), d_truth.data_ptr(), None) == 0
    torch.cuda.synchronize()

    def run(layout, shard=None):
        with MiClarkDB(k, T, layout=layout) as e:
            if shard is None:
                e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), 8, d_labels.data_ptr())
            else:
                e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), 8, d_labels.data_ptr(), shard=shard)
            res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
            rows = torch.zeros((n_reads, 16), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, res.data_ptr(), rows.data_ptr())
            e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), res.data_ptr(), rows.data_ptr())
            e.sync()
            first = res.clone()
            e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, res.data_ptr(), rows.data_ptr())
            e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), res.data_ptr(), rows.data_ptr())
            e.sync()
            assert torch.equal(first, res)                      # a second pass is bit-identical
            return res, rows, e
    res_m, rows_m, _ = run(2)
    res_d, rows_d, _ = run(1)
    res_s, rows_s, _ = run(3)
    res_t, rows_t, _ = run(4)                                   # the two-strand table: the layout the headline is quoted on
    assert torch.equal(res_m[:, :6], res_d[:, :6])              # words 0..5: sum, best/second, targets hit
    assert torch.equal(res_s[:, :6], res_d[:, :6])
    assert torch.equal(res_t[:, :6], res_d[:, :6])
    valid = (rows_m[:, 0] != -1) & (rows_d[:, 0] != -1)
    assert valid.float().mean() > 0.999 and torch.equal(rows_m[valid], rows_d[valid])
    assert torch.equal(rows_s[valid], rows_d[valid])
    assert torch.equal(rows_t[valid], rows_d[valid])
    _, rows_ta, _ = run(4, (0, htsize // 3))                    # ... and sharded by on-disk bucket range (per-k-mer kernel)
    _, rows_tb, _ = run(4, (htsize // 3, htsize))
    with MiClarkDB(k, T) as e:
        merged_t = torch.zeros_like(rows_ta)
        torch.cuda.synchronize()
        e.merge_rows_device(rows_ta.data_ptr(), rows_tb.data_ptr(), merged_t.data_ptr(), n_reads)
        e.sync()
    ok_t = (merged_t[:, 0] != -1) & valid
    assert ok_t.float().mean() > 0.999 and torch.equal(merged_t[ok_t], rows_m[ok_t])
    _, rows_sa, _ = run(3, (0, htsize // 3))                    # the super-k-mer table sharded by on-disk bucket range
    _, rows_sb, _ = run(3, (htsize // 3, htsize))
    # two shards merged on the device == the whole table
    half = htsize // 2
    _, rows_a, _ = run(2, (0, half))
    _, rows_b, _ = run(2, (half, htsize))
    with MiClarkDB(k, T) as e:
        e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), 8, d_labels.data_ptr(), shard=(0, 1000))
        merged = torch.zeros_like(rows_a)
        res2 = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.merge_rows_device(rows_a.data_ptr(), rows_b.data_ptr(), merged.data_ptr(), n_reads)
        e.result_from_rows_device(merged.data_ptr(), res2.data_ptr(), n_reads)
        e.sync()
    with MiClarkDB(k, T) as e:
        e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), 8, d_labels.data_ptr(), shard=(0, 1000))
        merged_s = torch.zeros_like(rows_sa)
        torch.cuda.synchronize()
        e.merge_rows_device(rows_sa.data_ptr(), rows_sb.data_ptr(), merged_s.data_ptr(), n_reads)
        e.sync()
    ok_s = (merged_s[:, 0] != -1) & valid
    assert ok_s.float().mean() > 0.999 and torch.equal(merged_s[ok_s], rows_m[ok_s])
    ok_rows = (merged[:, 0] != -1) & valid
    assert ok_rows.float().mean() > 0.999
    assert torch.equal(merged[ok_rows], rows_m[ok_rows]) and torch.equal(res2[ok_rows][:, :5], res_m[ok_rows][:, :5])
    # oracle on a sample
    ns = 20000
    sizes = d_sizes.cpu().numpy()
    keys = d_keys[:n_el].cpu().numpy().view(np.uint64)
    labels = d_labels[:n_el].cpu().numpy().view(np.uint16)
    rp = d_rp[:ns + 1].cpu().numpy().view(np.uint32)
    cont = d_cont[:int(rp[-1]) + 64].cpu().numpy().view(np.uint16)
    odb = gu.oracle().db_from_arrays(sizes, keys, labels)
    _, expect = _oracle_results(odb, k, rp, cont, T)
    assert (res_m[:ns, :5].cpu().numpy().view(np.uint32) == expect).all()
    # constructive known answer over all reads
    truth = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
    r = res_m.cpu().numpy().view(np.uint32)
    g = truth[:, 0] > 0
    okk = (truth[g, 1] == 0) | ((r[g, 1] == truth[g, 0]) & (r[g, 2] >= truth[g, 1]))
    assert okk.mean() > 0.999 and (r[~g, 0] == 0).mean() > 0.99


@pytest.mark.parametrize("n_shards", [2, 3])
def test_batch_merge_shards_equals_whole_table(n_shards):
    """The reference's multi-device mode through the batch API: every engine holds a bucket range, all get the same
    packed reads, mic_batch_merge_shards sums the rows into engine 0 - identical to one engine holding everything.
    (Engines share cuda:0 here; the peer copy is the same call.)"""
    from cuclark_amd import MiClarkDB, host
    name, k = "light_k31_u64", 31
    db = gu.load_golden_db(name)
    names = gu.target_names()
    T = len(names)
    data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    sizes = gu.golden_sizes(db)
    with MiClarkDB(k, T) as whole:
        whole.read_arrays(sizes, db["ky"], db["lb"])
        ref_res, ref_rows = whole.classify_packed(rp, cont, extended=True)
    H = int(db["htsize"])
    cuts = [H * i // n_shards for i in range(n_shards + 1)]
    engines = [MiClarkDB(k, T) for _ in range(n_shards)]
    try:
        for e, lo, hi in zip(engines, cuts[:-1], cuts[1:]):
            e.read_arrays(sizes, db["ky"], db["lb"], shard=(lo, hi))
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
    assert (rows[:, 0] != 0xFFFFFFFF).all() and (ref_rows[:, 0] != 0xFFFFFFFF).all()
    for r in range(n):
        m = int(rows[r, 0])
        assert m == int(ref_rows[r, 0]) and (rows[r, 1:1 + m] == ref_rows[r, 1:1 + m]).all()
