"""CPU tests of the product's host side (C ABI, no device needed)."""
import os
import re

import numpy as np
import pytest

import golden_util as gu


def test_library_exports_every_declared_symbol(lib):
    from cuclark_amd import _lib
    header = open(os.path.join(gu.ROOT, "include", "mi_clark.h")).read()
    declared = set(re.findall(r"\b(mic_[a-z0-9_]+)\s*\(", header))
    declared -= {"mic_engine", "mic_config", "mic_db_info", "mic_synth_spec"}
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cuclark_amd import MiClarkDB, MicError
    with pytest.raises(MicError) as ei:
        MiClarkDB(31, 4)
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(gu.ROOT, "cuclark_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "oracle" not in text.replace("no CPU fallback", "").lower() or f == "__init__.py", f
    init = open(os.path.join(gu.ROOT, "cuclark_amd", "__init__.py")).read()
    assert "import oracle" not in init and "from oracle" not in init


def test_key_bytes_rule(lib, orc):
    from cuclark_amd import host
    for h in (1610612741, 57777779, 1009, 1 << 20, 999983):
        for k in range(2, 33):
            assert host.key_bytes_rule(h, k) == orc.key_bytes_rule(h, k)


@pytest.mark.parametrize("fn", ["reads_k27.fa", "reads_k27.fq", "reads_k31.fa", "reads_k31.fq", "pairs_k31_1.fq"])
def test_index_and_pack_match_oracle(fn, lib, orc):
    from cuclark_amd import host
    data = open(os.path.join(gu.GOLDEN, fn), "rb").read()
    a, b = host.index_reads(data), orc.index_reads(data)
    for f in a:
        assert (a[f] == b[f]).all(), f
    for k in (12, 27, 31, 32):
        rp1, c1 = host.pack_reads(data, a["seq_s"], a["seq_e"], a["length"], k)
        rp2, c2 = orc.pack_batch(data, b["seq_s"], b["seq_e"], b["length"], k)
        assert (rp1 == rp2).all() and c1.size == c2.size and (c1 == c2).all()


def test_index_edge_cases(lib, orc):
    from cuclark_amd import host
    cases = [b">a\nACGT", b">a\nACGT\n", b">a b c\nAC\nGT\n\n>b\n\n>c\nA\n", b"@r1\nACGT\n+\nIIII", b"@r1 x\nACGT\n+\nIIII\n\n",
             b"@r/1\nAC\r\n+\nII\n@r/2\nGG\n+\n@I\n", b">only_header", b">x\n" + b"ACGT" * 1000 + b"\n"]
    for data in cases:
        a, b = host.index_reads(data), orc.index_reads(data)
        assert a is not None and b is not None
        for f in a:
            assert (a[f] == b[f]).all(), (data[:20], f)
    assert host.index_reads(b"ACGT\n") is None and orc.index_reads(b"ACGT\n") is None


def test_pack_long_part_split(lib, orc):
    """Parts longer than 65528 nt become overlapping sub-parts: same k-mers, product == oracle byte for byte."""
    from cuclark_amd import host
    rng = np.random.default_rng(4)
    seq = "".join(rng.choice(list("ACGT"), 150000))
    data = f">big\n{seq}\n".encode()
    ix = host.index_reads(data)
    for k in (5, 31):
        rp, ct = host.pack_reads(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        rp2, ct2 = orc.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        assert (rp == rp2).all() and (ct == ct2).all()
        # walk the parts: lengths sum to n + (parts-1)*(k-1)
        p, lens = 0, []
        while p < ct.size:
            lens.append(int(ct[p]))
            p += 1 + (lens[-1] + 7) // 8
        assert len(lens) == 3 and sum(lens) == 150000 + 2 * (k - 1) and max(lens) == 65528


def test_pack_random_records_match_oracle(lib, orc):
    """The streaming packer (eight nucleotides per step, roll-back of short runs) against the oracle's byte-by-byte one:
    random line widths (containers straddle line breaks), Ns and other bytes, lower case, U, runs around k, empty
    lines, records around the 65528-nt limit where the packer switches to the splitting path."""
    from cuclark_amd import host
    rng = np.random.default_rng(77)
    recs = []
    for i in range(400):
        n = int(rng.choice([0, 1, 7, 8, 9, 30, 31, 32, 33, 64, 100, 151, 250, 1000]))
        seq = rng.choice(list("ACGTacgtUu"), n)
        for pos in rng.integers(0, max(n, 1), int(rng.integers(0, 4))):
            if n:
                seq[pos] = rng.choice(list("NnRY-*. "))
        body = "".join(seq)
        width = int(rng.choice([1, 7, 8, 13, 60, 70, 80, 10 ** 6]))
        lines = [body[a:a + width] for a in range(0, len(body), width)] or [""]
        if rng.random() < 0.1:
            lines.insert(int(rng.integers(0, len(lines) + 1)), "")
        recs.append(f">r{i} d\n" + "\n".join(lines) + "\n")
    for n in (65527, 65528, 65529, 65600):
        seq = rng.choice(list("ACGT"), n)
        seq[n // 3] = "N"
        body = "".join(seq)
        recs.append(f">long{n}\n" + "\n".join(body[a:a + 70] for a in range(0, n, 70)) + "\n")
        recs.append(f">solid{n}\n" + "".join(rng.choice(list("ACGT"), n)) + "\n")
    data = "".join(recs).encode()
    ix = host.index_reads(data)
    assert ix["seq_s"].size == len(recs)
    for k in (4, 8, 9, 21, 31, 32):
        rp, ct = host.pack_reads(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        rp2, ct2 = orc.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        assert (rp == rp2).all() and ct.size == ct2.size and (ct == ct2).all(), k


@pytest.mark.parametrize("case", [c[0] for c in gu.expected_csv_cases()])
def test_csv_formatting_matches_golden(case, lib):
    """C++ CSV writer fed with the oracle's results reproduces the committed CSV byte for byte."""
    from cuclark_amd import host
    _, k, dbname, data, paired, ext = {c[0]: c for c in gu.expected_csv_cases()}[case]
    odb, _ = gu.oracle_db_from_golden(dbname)
    names = gu.target_names()
    _, results = odb.classify_file(k, data, names, paired, ext)
    idx = host.index_reads(data)
    res8 = np.zeros((results.shape[0], 8), np.uint32)
    res8[:, :5] = results
    dense = None
    if ext:
        dense = {r: odb.count_read_ascii(k, data[int(idx["seq_s"][r]):int(idx["seq_e"][r])], idx["length"][r], len(names))
                 for r in range(results.shape[0])}
    text = host.format_csv(data, idx, res8, names, k, paired=paired, extended=ext, dense=dense)
    assert text == open(os.path.join(gu.GOLDEN, f"expected_{case}.csv"), "rb").read()


def test_parallel_indexer_equals_serial(lib):
    """mic_index_reads_parallel cuts the file into byte ranges; the result must equal the serial indexer's."""
    from cuclark_amd import host
    rng = np.random.default_rng(8)
    # FASTQ with quality lines that start with '@' and '+', FASTA with wrapped lines and long records
    fq = []
    for i in range(20000):
        L = int(rng.integers(30, 200))
        seq = "".join(rng.choice(list("ACGTN"), L))
        qual = "".join(rng.choice(list("@+IJ#5"), L))
        fq.append(f"@read{i} len={L}\n{seq}\n+\n{qual}\n")
    fq = "".join(fq).encode()
    fa = []
    for i in range(3000):
        L = int(rng.integers(1, 3000))
        seq = "".join(rng.choice(list("ACGT"), L))
        fa.append(f">rec{i}\n" + "\n".join(seq[j:j + 60] for j in range(0, L, 60)) + "\n")
    fa = "".join(fa).encode()
    for data in (fq, fa, fq[:-1], b">one\n" + b"ACGT" * 100000 + b"\n"):
        a = host.index_reads(data)
        for t in (2, 3, 8, 37):
            b = host.index_reads(data, threads=t)
            assert len(b["length"]) == len(a["length"])
            for f in a:
                assert (a[f] == b[f]).all(), (t, f)


def test_pack_with_exact_capacity_after_a_rolled_back_run(lib):
    """A run shorter than k is written while it is read and rolled back at its end; when that happens near the end of
    the buffer the writes may pass a capacity that the final output fits exactly (CuCLARK_hh.hh:1647-1651, 1701-1704)."""
    from cuclark_amd import host
    import ctypes as C
    k = 31
    rng = np.random.default_rng(3)
    seq = lambda n: bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
    data = b">a\n" + seq(80) + b"\n>b\n" + seq(45) + b"N" + seq(29) + b"\n"      # the last run (29 nt) is dropped
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    buf = np.frombuffer(data, np.uint8)
    for cap, ok in ((cont.size, True), (cont.size - 1, False)):
        rp2 = np.zeros(rp.size, np.uint32)
        out = np.zeros(cont.size + 8, np.uint16)
        m = lib.mic_pack_reads(buf.ctypes.data, idx["seq_s"].ctypes.data, idx["seq_e"].ctypes.data, idx["length"].ctypes.data,
                               rp.size - 1, k, rp2.ctypes.data, out.ctypes.data, cap)
        if ok:
            assert m == cont.size and (out[:m] == cont).all() and (rp2 == rp).all()
        else:
            assert m == C.c_size_t(-1).value


def test_sketch_hash_on_structured_minimizer_sets():
    """The table is sized from a HyperLogLog sketch of the minimizer values (mic_build.hip: s_expand_kernel / hll_estimate, 4096
    registers).  The sketch's hash - restated here with the same constants - must not be thrown by STRUCTURED sets: the m-mers of
    tandem repeats (periodic bit patterns), runs of consecutive values, values that differ in their top nucleotides only.  Each
    estimate within 6 % of the exact count (the sketch's standard error is 1.04 / sqrt(4096) = 1.6 %)."""
    import numpy as np
    M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

    def estimate(x):
        x = np.unique(np.asarray(x, dtype=np.uint64))
        with np.errstate(over="ignore"):
            g = x * np.uint64(0x9E3779B97F4A7C15)
            g ^= g >> np.uint64(32)
            g = g * np.uint64(0xD6E8FEB86659FD93)
            g ^= g >> np.uint64(32)
        idx = (g >> np.uint64(52)).astype(np.int64)
        rest = ((g << np.uint64(12)) & M64) | np.uint64(1 << 11)
        # count leading zeros of a 64-bit word = 63 - floor(log2)
        clz = 63 - np.floor(np.log2(rest.astype(np.float64) * (1 + 1e-17))).astype(np.int64)
        hi = rest >> np.uint64(32)
        clz = np.where(hi > 0, 31 - np.floor(np.log2(np.maximum(hi, 1).astype(np.float64))).astype(np.int64),
                       63 - np.floor(np.log2(np.maximum(rest & np.uint64(0xFFFFFFFF), 1).astype(np.float64))).astype(np.int64))
        reg = np.zeros(4096, np.int64)
        np.maximum.at(reg, idx, clz + 1)
        alpha = 0.7213 / (1.0 + 1.079 / 4096)
        E = alpha * 4096 * 4096 / np.sum(np.ldexp(1.0, -reg))
        zeros = int((reg == 0).sum())
        if E <= 2.5 * 4096 and zeros:
            E = 4096 * np.log(4096 / zeros)
        return float(E), x.size

    rng = np.random.default_rng(5)
    m = 20
    sets = {}
    sets["random 40-bit values"] = rng.integers(0, 1 << 40, 300_000, dtype=np.uint64)
    sets["consecutive values"] = np.arange(1 << 33, (1 << 33) + 500_000, dtype=np.uint64)
    sets["top nucleotides only"] = (np.arange(200_000, dtype=np.uint64) << np.uint64(22)) | np.uint64(0x2AAAAA)
    # the m-mers of tandem repeats: every rotation of units of 2 .. 50 nucleotides, repeated to m nucleotides, plus the m-mers across
    # a tract's end into a random flank (what a database keeps of a microsatellite)
    reps = []
    for _ in range(20_000):
        u = int(rng.integers(2, 51))
        unit = rng.integers(0, 4, u)
        seq = np.concatenate([np.tile(unit, (2 * m) // u + 2), rng.integers(0, 4, m)])
        for s0 in range(0, min(u, 8) + m, 1):
            if s0 + m <= seq.size:
                v = 0
                for c in seq[s0:s0 + m]:
                    v = (v << 2) | int(c)
                reps.append(v)
    sets["m-mers of tandem repeats"] = np.array(reps, dtype=np.uint64)
    for name, vals in sets.items():
        est, exact = estimate(vals)
        assert abs(est / exact - 1) < 0.06, (name, est, exact)
