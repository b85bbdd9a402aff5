"""CPU tests of the oracle (test infrastructure): it must reproduce the reference's golden vectors before it is
allowed to judge the HIP path."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

REF_LIGHT = os.path.join(gu.ROOT, "oracle", "_ref", "ref_table_light")


@pytest.mark.parametrize("name", [n for n in gu.golden_db_names() if not n.startswith("lightgap")])
def test_oracle_matches_reference_queries(name):
    """Golden vectors = answers of the reference's EHashtable::queryElement on the DB the reference wrote."""
    if name.startswith("full"):
        # the reference's real table size (1 610 612 741 buckets): 1.6 GB of sizes + 12.9 GB of prefix sums in the oracle
        import psutil
        if psutil.virtual_memory().available < 24e9:
            pytest.skip("needs ~15 GB of free memory for the 1.6e9-bucket table")
    odb, meta = gu.oracle_db_from_golden(name)
    q = np.load(os.path.join(gu.GOLDEN, f"queries_{name}.npz"))
    f, l = odb.find_many(q["kmers"], meta["k"])
    assert (f == q["found"]).all()
    assert (l[f == 1] == q["label"][f == 1]).all()
    assert odb.n_elems == meta["ky"].size
    if "found_s3" in q:
        odb3, _ = gu.oracle_db_from_golden(name, sampling=3)
        f3, l3 = odb3.find_many(q["kmers"], meta["k"])
        assert (f3 == q["found_s3"]).all() and (l3[f3 == 1] == q["label_s3"][f3 == 1]).all()
        assert 0 < f3.sum() < f.sum()


def test_full_and_light_tables_hold_the_same_kmers():
    """Same targets, k=31: HTSIZE 1610612741/u32 keys vs 57777779/u64 keys -> same canonical k-mers and labels."""
    a, b = gu.load_golden_db("full_k31_u32"), gu.load_golden_db("light_k31_u64")

    def kmers(db):
        rem = np.repeat(db["sz_idx"], db["sz_val"].astype(np.int64))
        return dict(zip((db["ky"].astype(object) * db["htsize"] + rem.astype(object)).tolist(), db["lb"].tolist()))
    assert kmers(a) == kmers(b)


@pytest.mark.skipif(not os.access(REF_LIGHT, os.X_OK), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_binary_live(tmp_path):
    """Run the reference's own table (compiled from /root/reference/src) on fresh random k-mers."""
    name = "light_k27_u32"
    prefix, meta = gu.materialize_db(name, str(tmp_path))
    rng = np.random.default_rng(123)
    odb, _ = gu.oracle_db_from_golden(name)
    elems_rem = np.repeat(meta["sz_idx"], meta["sz_val"].astype(np.int64))
    present = (meta["ky"].astype(np.uint64) * np.uint64(meta["htsize"]) + elems_rem)[rng.integers(0, meta["ky"].size, 300)]
    o = gu.oracle()
    present = np.array([o.revcomp(int(v), 27) if i % 2 else int(v) for i, v in enumerate(present)], dtype=np.uint64)
    kmers = np.concatenate([present, rng.integers(0, 1 << 54, 300, dtype=np.uint64)])
    qf = tmp_path / "q.txt"
    qf.write_text("\n".join(str(int(v)) for v in kmers) + "\n")
    out = subprocess.run([REF_LIGHT, "query", "27", "4", prefix, str(qf)], check=True, capture_output=True, text=True).stdout
    ref = np.array([[int(x) for x in line.split()[1:]] for line in out.strip().splitlines()])
    f, l = odb.find_many(kmers, 27)
    assert (f == ref[:, 0]).all() and (l[f == 1] == ref[f == 1, 1]).all()
    assert f[:300].all()


@pytest.mark.parametrize("case", [c[0] for c in gu.expected_csv_cases()])
def test_expected_csv_regenerates(case):
    spec = {c[0]: c for c in gu.expected_csv_cases()}[case]
    _, k, dbname, data, paired, ext = spec
    odb, _ = gu.oracle_db_from_golden(dbname)
    text, _ = odb.classify_file(k, data, gu.target_names(), paired, ext)
    assert text == open(os.path.join(gu.GOLDEN, f"expected_{case}.csv"), "rb").read()


def test_hand_derived_csv_lines():
    """Edge cases of SURVEY.md §8c worked out by hand."""
    lines = open(os.path.join(gu.GOLDEN, "expected_k31_fa.csv")).read().splitlines()
    by = {l.split(",")[0]: l for l in lines}
    assert lines[0] == "Object_ID,Length,Gamma,1st_assignment,score1,2nd_assignment,score2,confidence"
    assert by["short_lt_k"] == "short_lt_k,10,-0,NA,0,NA,0,0"                 # 0/(10-31+1) = -0
    assert by["len_k_minus_1"] == "len_k_minus_1,30,-nan,NA,0,NA,0,0"         # 0/0
    assert by["random_nohit"] == "random_nohit,140,0,NA,0,NA,0,0"
    assert by["exact_k"].split(",")[1:3] == ["31", "1"]
    assert "a_very_long_read_name_that_exceeds_the_" in by and len("a_very_long_read_name_that_exceeds_the_") == 39
    tie = by["two_targets_tie"].split(",")
    assert tie[3:8] == ["T_alpha", "10", "T_beta", "10", "0.5"]               # tie: lower target index is first
    assert by["iupac_R"].split(",")[1] == "101"                               # IUPAC byte counts in Length
    pl = open(os.path.join(gu.GOLDEN, "expected_k31_pairs.csv")).read().splitlines()[1].split(",")
    p1 = open(os.path.join(gu.GOLDEN, "pairs_k31_1.fq")).read().splitlines()[1]
    p2 = open(os.path.join(gu.GOLDEN, "pairs_k31_2.fq")).read().splitlines()[1]
    assert pl[0] == "pair0" and int(pl[1]) == len(p1) + len(p2)               # len(seq1 N seq2) - NBN


def test_pack_query_consistency(orc):
    """Counts from the packed containers == counts straight from the ASCII (part rule)."""
    odb, _ = gu.oracle_db_from_golden("light_k31_u64")
    data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
    ix = orc.index_reads(data)
    rp, cont = orc.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], 31)
    counts, bad = odb.query_batch(31, rp, cont, 6)
    assert bad == 0
    for r in range(len(ix["length"])):
        c = odb.count_read_ascii(31, data[int(ix["seq_s"][r]):int(ix["seq_e"][r])], ix["length"][r], 6)
        assert (c == counts[r]).all()
    shard_sum = sum(odb.query_batch(31, rp, cont, 6, part=(a, b))[0] for a, b in [(0, 1000), (1000, 30000000), (30000000, None)])
    assert (shard_sum == counts).all()


def test_top2_rule_equals_ascending_scan(orc):
    """resultKernel's ascending scan == top-2 under (count desc, target asc)."""
    rng = np.random.default_rng(1)
    for _ in range(2000):
        T = int(rng.integers(1, 12))
        counts = rng.integers(0, 4, T).astype(np.uint32)
        res = orc.result_from_counts(counts)
        order = sorted([(-(int(c)), t) for t, c in enumerate(counts) if c])
        exp = [int(counts.sum()), 0, 0, 0, 0]
        if order:
            exp[1], exp[2] = order[0][1] + 1, -order[0][0]
        if len(order) > 1:
            exp[3], exp[4] = order[1][1] + 1, -order[1][0]
        assert res.tolist() == exp
        n, row = orc.sparse_row(counts, 16)
        assert (orc.result_from_row(row) == res).all()


def test_merge_rows(orc):
    rng = np.random.default_rng(2)
    for _ in range(300):
        a = rng.integers(0, 3, 20).astype(np.uint32)
        b = rng.integers(0, 3, 20).astype(np.uint32)
        _, ra = orc.sparse_row(a, 32)
        _, rb = orc.sparse_row(b, 32)
        n, rs = orc.sparse_row(a + b, 32)
        m = orc.merge_rows(ra, rb)
        assert m[0] == n and (m[1:1 + 2 * n] == rs[1:1 + 2 * n]).all()


def test_key_width_rule(orc):
    assert [orc.key_bytes_rule(1610612741, k) for k in (23, 24, 31, 32)] == [2, 4, 4, 8]
    assert [orc.key_bytes_rule(57777779, k) for k in (20, 21, 27, 28, 29)] == [2, 4, 4, 4, 8]


def test_fast_batch_classifier_equals_the_plain_one(orc):
    """bench.py times orc_classify_batch_fast (prefetch sweeps, sparse tally) on a table copied by orc_db_copy_spread;
    both must give what the plain restatement gives: golden reads, ties, many targets, long parts, padded batches."""
    for k, name in ((31, "light_k31_u64"), (27, "light_k27_u32")):
        db = gu.load_golden_db(name)
        sizes = gu.golden_sizes(db)
        odb = orc.db_from_arrays(sizes, db["ky"], db["lb"])
        spread = orc.db_copy_spread(sizes, np.ascontiguousarray(db["ky"]), np.ascontiguousarray(db["lb"], np.uint16), threads=3)
        data = open(os.path.join(gu.GOLDEN, f"reads_k{k}.fa"), "rb").read()
        ix = orc.index_reads(data)
        rp, ct = orc.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        ref = odb.classify_batch(k, rp, ct, 6)
        assert (odb.classify_batch_fast(k, rp, ct, 6, threads=2) == ref).all()
        assert (spread.classify_batch_fast(k, rp, ct, 6, threads=4) == ref).all()
    rng = np.random.default_rng(5)
    k, T, htsize = 21, 300, 20011
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 6000, k, 4, T)
    odb = orc.db_from_arrays(sizes, keys, labels)
    spread = orc.db_copy_spread(sizes, keys, labels, threads=2)
    picks = rng.choice(canon.size, 1500, replace=False)
    long_part = "".join(gu.kmer_to_ascii(canon[i], k) for i in picks[:200])            # one part of 4200 nt: > 1024 k-mers
    many = "N".join(gu.kmer_to_ascii(canon[i], k) for i in picks[200:1500])             # 1300 parts, ~290 targets
    tie = gu.kmer_to_ascii(canon[picks[0]], k) + "N" + gu.kmer_to_ascii(canon[picks[1]], k)
    data = f">a\n{long_part}\n>b\n{many}\n>c\n{tie}\n>d\nACGT\n>e\n{many}N{long_part}\n".encode()
    ix = orc.index_reads(data)
    rp, ct = orc.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
    ref = odb.classify_batch(k, rp, ct, T)
    assert (odb.classify_batch_fast(k, rp, ct, T, threads=3) == ref).all()
    assert (spread.classify_batch_fast(k, rp, ct, T, threads=1) == ref).all()
    nd = orc.numa_db(sizes, keys, labels, threads=4)           # one replica per NUMA node, threads pinned
    assert (nd.classify_batch(k, rp, ct, T) == ref).all()
    nd.close()


def test_mod_sampling_rule_of_the_partition_restatement():
    """The sampling rule the table partition follows (oracle/part_rule.c restates mic_device.h: s_tlen, s_torder, s_probe_read;
    DESIGN.md 3.6), on the CPU: (1) a k-mer and its reverse complement are answered by the same slot of the one-strand table
    whatever their position in the read - the positions of rc(K) mirror those of K because t = m (mod w); (2) consecutive k-mers
    of a random read keep their slot for 1 / 0.120 positions on average (the plain minimizer: 1 / 0.154), i.e. a 150-bp read
    falls into 15-16 runs instead of 19-20."""
    o = gu.oracle()
    rng = np.random.default_rng(77)
    n_slots = 1 << 30
    for k, m in ((31, 20), (27, 20), (32, 20), (24, 20), (21, 17)):
        seq = rng.integers(0, 4, 20000)
        vals = []
        v = 0
        mask = (1 << (2 * k)) - 1
        for i, c in enumerate(seq):
            v = ((v << 2) | int(c)) & mask
            if i >= k - 1:
                vals.append(v)
        # (1) strand symmetry of the one-strand rule: K at position p and rc(K) at any other position
        w = k - m + 1
        t = m
        while t - w >= 7:
            t -= w

        def tied(K):      # the smallest 27-bit order of the canonical t-mers of K occurs more than once: either position may be taken
            orders = []
            for i in range(k - t + 1):
                tv = (K >> (2 * (k - t - i))) & ((1 << (2 * t)) - 1)
                tv = min(tv, o.revcomp(tv, t))
                orders.append((((tv & 0xFFFFFF) * 0x9E3779 + 0x27D4EB2F + (tv >> 24) * 0x85EBCA77) & 0xFFFFFFFF) >> 5)
            return orders.count(min(orders)) > 1

        n_checked = 0
        for j in range(0, 2000, 7):
            K = vals[j]
            if tied(K):
                continue
            a = o.L.orc_part_slot_of_kmer(K, j, k, m, 0, n_slots)
            b = o.L.orc_part_slot_of_kmer(o.revcomp(K, k), (j * 5 + 3) % 977, k, m, 0, n_slots)
            assert a == b, (k, m, j)
            n_checked += 1
        assert n_checked > 250
        # (2) density of the two-strand rule (t-mers as they read)
        slots = np.array([o.L.orc_part_slot_of_kmer(K, j, k, m, 1, n_slots) for j, K in enumerate(vals[:6000])])
        density = (1 + np.count_nonzero(slots[1:] != slots[:-1])) / slots.size
        plain = 2.0 / (w + 1)
        if t < m:       # mod-sampling applies: clearly below the plain minimizer's density
            assert density < 0.88 * plain, (k, m, t, density, plain)
        else:
            assert abs(density - plain) < 0.15 * plain, (k, m, t, density, plain)
        if (k, m) == (31, 20):
            assert 0.112 < density < 0.128, density
