"""Device-side ingest (mic_ingest_*): raw FASTA/FASTQ bytes -> CSV text on the GPU must give the bytes the host path
gives (mic_index_reads + mic_pack_reads + batch API + mic_csv_line), which the golden CSVs and the oracle pin.
Covers CuCLARK_hh.hh:1339-1534 (indexing), :1616-1716 (packing, N-splitting), :1951-2139 (CSV)."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as gu


def test_ratio_formatter_equals_libc_printf(lib):
    """The integer-only "%g" (csrc/mic_fmt.h, used by the device CSV kernel) against the C library's."""
    libc = C.CDLL("libc.so.6")
    libc.snprintf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_double]
    a, b = C.create_string_buffer(32), C.create_string_buffer(32)
    rng = np.random.default_rng(7)
    pairs = [(n, d) for d in range(1, 260) for n in range(1, d + 1)]
    pairs += [(127, 128), (639, 640), (1, 512), (1, 65535), (1, 4000000000), (999999, 1000000), (9999995, 10000000),
              (99999, 100000), (1, 3), (2, 3), (1, 1 << 31), (12345, 1 << 20)]
    dens = rng.integers(1, 1 << 32, 40000, dtype=np.uint64)
    nums = (rng.integers(0, 1 << 62, 40000, dtype=np.uint64) % dens) + 1
    pairs += [(int(n), int(d)) for n, d in zip(nums, dens)]
    for n, d in pairs:
        ln = lib.mic_format_ratio_g(n, d, a)
        libc.snprintf(b, 32, b"%g", n / d)
        assert ln == len(a.value) and a.value == b.value, (n, d, a.value, b.value)
    assert lib.mic_format_ratio_g(3, 2, a) < 0 and lib.mic_format_ratio_g(0, 2, a) < 0


def _host_path(e, data, names, k, paired=False):
    from cuclark_amd import host
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    res = e.classify_packed(rp, cont)
    text = host.format_csv(data, idx, res, names, k, paired=paired)
    return text[text.index(b"\n") + 1:], res, rp, cont      # the device path emits no header line


def _same_packed(rp_h, ct_h, rp_d, ct_d):
    """Every read of the device packer holds the host packer's containers, then a 0 length slot or its end."""
    for r in range(rp_h.size - 1):
        n = int(rp_h[r + 1] - rp_h[r])
        o = int(rp_d[r])
        room = int(rp_d[r + 1]) - o
        assert room >= n, (r, room, n)
        assert (ct_d[o:o + n] == ct_h[rp_h[r]:rp_h[r + 1]]).all(), r
        assert room == n or ct_d[o + n] == 0, r


def _engine(k, names, dbname):
    from cuclark_amd import MiClarkDB
    db = gu.load_golden_db(dbname)
    e = MiClarkDB(k, len(names))
    e.read_arrays(gu.golden_sizes(db), db["ky"], db["lb"])
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c[0] for c in gu.expected_csv_cases() if not c[5]])
def test_golden_files_through_device_ingest(case):
    _, k, dbname, data, paired, _ = [c for c in gu.expected_csv_cases() if c[0] == case][0]
    names = gu.target_names()
    with _engine(k, names, dbname) as e:
        e.ingest_alloc(2, 1 << 20, names, want_results=True)
        r = e.ingest_classify(1, data, paired=paired)
        assert r["status"] == 0, r
        expect = open(os.path.join(gu.GOLDEN, f"expected_{case}.csv"), "rb").read()
        assert r["csv"] == expect[expect.index(b"\n") + 1:]
        csv_h, res_h, rp_h, ct_h = _host_path(e, data, names, k, paired)
        assert r["csv"] == csv_h and (r["results"][:, :7] == res_h[:, :7]).all()
        _same_packed(rp_h, ct_h, *e.ingest_fetch_packed(1))
        e.ingest_free()


def _genomes():
    out = []
    for fn, _ in gu.target_files_and_labels():
        seq = b"".join(l.strip() for l in open(fn, "rb") if not l.startswith(b">"))
        out.append(seq)
    return out


def _random_reads(rng, genomes, n, fasta, crlf=False, paired=False):
    """Records with the things the reference's parser and packer treat specially: N and other bytes, lower case, U,
    reads shorter than k, names with blanks / tabs / more than 39 characters, multi-line FASTA, empty sequences."""
    recs = []
    for i in range(n):
        g = genomes[int(rng.integers(len(genomes)))]
        L = int(rng.choice([0, 5, 26, 27, 30, 31, 32, 33, 50, 64, 65, 100, 127, 128, 129, 150, 151, 191, 193, 250, 301, 600, 2500]))
        p = int(rng.integers(0, max(1, len(g) - L)))
        s = bytearray(g[p:p + L]) if rng.random() < 0.8 else bytearray(rng.choice(list(b"ACGT"), L).astype(np.uint8).tobytes())
        for _ in range(int(rng.integers(0, 3)) if len(s) else 0):
            s[int(rng.integers(len(s)))] = int(rng.choice(list(b"NnRY-*.x")))
        if rng.random() < 0.3:
            s = bytearray(bytes(s).lower())
        if rng.random() < 0.2:
            s = bytearray(bytes(s).replace(b"T", b"U"))
        name = [b"r%d" % i, b"read_%d some description" % i, b"q%d\twith tab" % i, b"x" * 45 + b"%d" % i, b"n%d/1" % i,
                b"a", b"_lead%d" % i if paired else b" lead%d" % i][int(rng.integers(7))]
        eol = b"\r\n" if crlf else b"\n"
        if paired:       # the merged text of file.cc:205-268: >id / seq1 N seq2
            g2 = genomes[int(rng.integers(len(genomes)))]
            p2 = int(rng.integers(0, len(g2) - 200))
            recs.append(b">" + name.split(b" ")[0].split(b"/")[0] + b"\n" + bytes(s) + b"N" + g2[p2:p2 + int(rng.integers(0, 160))] + b"\n")
        elif fasta:
            w = int(rng.choice([0, 0, 60, 70, 7]))
            body = bytes(s)
            if w and len(body) > w:
                body = eol.join(body[j:j + w] for j in range(0, len(body), w))
            recs.append(b">" + name + eol + body + eol)
        else:
            recs.append(b"@" + name + eol + bytes(s) + eol + b"+" + eol + b"I" * len(s) + eol)
    return b"".join(recs)


@pytest.mark.gpu
@pytest.mark.parametrize("k,dbname", [(31, "light_k31_u64"), (27, "light_k27_u32"), (20, "light_k20_u16")])
def test_random_records_device_ingest_equals_host_path(k, dbname):
    names = gu.target_names()
    genomes = _genomes()
    rng = np.random.default_rng(100 + k)
    with _engine(k, names, dbname) as e:
        e.ingest_alloc(1, 4 << 20, names, want_results=True)
        for trial in range(12):
            fasta = trial % 2 == 0
            data = _random_reads(rng, genomes, 700, fasta, crlf=trial in (4, 5), paired=trial == 8)
            if trial in (6, 7):
                data = data[:-1]              # no line end after the last record
            r = e.ingest_classify(0, data, paired=trial == 8)
            assert r["status"] == 0, (trial, r["status"])
            csv_h, res_h, rp_h, ct_h = _host_path(e, data, names, k, paired=trial == 8)
            assert r["n_reads"] == res_h.shape[0]
            assert (r["results"][:, :7] == res_h[:, :7]).all(), trial
            _same_packed(rp_h, ct_h, *e.ingest_fetch_packed(0))
            assert r["csv"] == csv_h, trial
        e.ingest_free()


@pytest.mark.gpu
def test_what_the_device_path_does_not_reproduce_is_handed_back():
    from cuclark_amd import _lib
    names = gu.target_names()
    ok = b"@r1\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    with _engine(31, names, "light_k31_u64") as e:
        e.ingest_alloc(1, 1 << 20, names)
        cases = {
            "empty name": (b"@\nACGT\n+\nIIII\n" + ok, _lib.MIC_INGEST_ODD_RECORD),
            "truncated fastq": (ok + b"@r2\nACGT\n+\n", _lib.MIC_INGEST_TRUNCATED),
            "fasta record without sequence": (b">a\nACGT\n>b\n>c\nACGT\n", _lib.MIC_INGEST_ODD_RECORD),
            "header only at the end": (b">a\nACGT\n>b", _lib.MIC_INGEST_ODD_RECORD),
            "long sequence": (b">a\n" + b"ACGT" * 20000 + b"\n", _lib.MIC_INGEST_LONG_READ),
            "short lines": (b">a\n" + b"A\n" * 200000, _lib.MIC_INGEST_TOO_MANY),
            "unknown format": (b"ACGT\n", _lib.MIC_INGEST_ODD_RECORD),
        }
        for what, (data, bit) in cases.items():
            r = e.ingest_classify(0, data)
            assert r["status"] & _lib.MIC_INGEST_FALLBACK and r["status"] & bit, (what, r["status"])
        r = e.ingest_classify(0, ok)          # the slot is usable afterwards
        assert r["status"] == 0 and r["n_reads"] == 1
        e.ingest_free()


@pytest.mark.gpu
def test_many_batches_on_concurrent_slots():
    """Four host threads, one slot each, many batches: what a streaming caller does."""
    import threading
    names = gu.target_names()
    genomes = _genomes()
    rng = np.random.default_rng(5)
    batches = [_random_reads(rng, genomes, 3000, fasta=(i % 3 == 0)) for i in range(24)]
    with _engine(31, names, "light_k31_u64") as e:
        expect = [_host_path(e, b, names, 31)[0] for b in batches]
        e.ingest_alloc(4, 8 << 20, names)
        got = [None] * len(batches)

        def work(slot):
            for i in range(slot, len(batches), 4):
                r = e.ingest_classify(slot, batches[i])
                got[i] = r["csv"] if r["status"] == 0 else r["status"]
        th = [threading.Thread(target=work, args=(s,)) for s in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        for i in range(len(batches)):
            assert got[i] == expect[i], i
        e.ingest_free()
