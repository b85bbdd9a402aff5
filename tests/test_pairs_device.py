"""Paired-end FASTQ texts that are on the device (mic_pairs_*, csrc/mic_ingest.hip): the reference's merge (file.cc:205-268:
">id\\nseq1Nseq2\\n", id = first field of the header between ' ', '/', TAB, '@') against the reference-made golden file, against
a restatement of the host merger's rule in Python, and the CSV of the merged batches against the host-merged text through the same
ingest path; texts that do not pair up line by line must be refused, never merged."""
import gzip
import io
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

SEP = b" /\t@"


def _gz(data, level=1):
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", compresslevel=level, mtime=0) as f:
        f.write(data)
    return buf.getvalue()


def _id(header):
    a = 0
    while a < len(header) and header[a] in SEP:
        a += 1
    b = a
    while b < len(header) and header[b] not in SEP:
        b += 1
    return header[a:b]


def _lines(text):
    ls = text.split(b"\n")
    if ls and ls[-1] == b"":
        ls.pop()                     # (a text that ends with a line end has no empty last line)
    return ls


def _merge(fq1, fq2):
    """classifier.cpp PairedFileFeeder::merge restated: lines split at '\\n' only, four per record."""
    l1, l2 = _lines(fq1), _lines(fq2)
    assert len(l1) == len(l2) and len(l1) % 4 == 0
    out = []
    for i in range(0, len(l1), 4):
        assert l1[i][:1] == b"@" and l2[i][:1] == b"@" and _id(l1[i]) == _id(l2[i]) != b""
        out.append(b">" + _id(l1[i]) + b"\n" + l1[i + 1] + b"N" + l2[i + 1] + b"\n")
    return out


class _Texts:
    def __init__(self, e, fq1, fq2):
        self.e = e
        self.t = [e.gunzip_device(_gz(fq1)), e.gunzip_device(_gz(fq2))]
        assert self.t[0][1] == len(fq1) and self.t[1][1] == len(fq2)

    def index(self):
        return self.e.pairs_index(self.t[0][0], self.t[0][1], self.t[1][0], self.t[1][1])

    def free(self):
        for t in self.t:
            self.e.free_text(t[0])


def _engine(k=31, dbname="light_k31_u64"):
    from cuclark_amd import MiClarkDB
    names = gu.target_names()
    db = gu.load_golden_db(dbname)
    e = MiClarkDB(k, len(names))
    e.read_arrays(gu.golden_sizes(db), db["ky"], db["lb"])
    return e, names


@pytest.mark.parametrize("k,dbname", [(31, "light_k31_u64"), (27, "light_k27_u32")])
def test_golden_pairs_merged_and_classified_on_the_device(k, dbname):
    p1 = open(os.path.join(gu.GOLDEN, f"pairs_k{k}_1.fq"), "rb").read()
    p2 = open(os.path.join(gu.GOLDEN, f"pairs_k{k}_2.fq"), "rb").read()
    e, names = _engine(k, dbname)
    with e:
        tx = _Texts(e, p1, p2)
        h, n_rec, off, stride = tx.index()
        assert h is not None and n_rec == len(_lines(p1)) // 4
        merged = e.pairs_text(h, 0, n_rec)
        assert merged == gu.merge_pairs(p1, p2) == b"".join(_merge(p1, p2))
        ref = os.path.join(gu.GOLDEN, f"pairs_k{k}_merged.fa")      # made by the reference's mergePairedFiles (make_golden.py)
        if os.path.exists(ref):
            assert merged == open(ref, "rb").read()
        assert off[-1] == len(merged) and off[0] == 0
        e.ingest_alloc(1, 1 << 20, names, want_results=True)
        r = e.pairs_classify(h, 0, 0, n_rec)
        assert r["status"] == 0 and r["n_reads"] == n_rec and r["n_bytes"] == len(merged)
        expect = open(os.path.join(gu.GOLDEN, f"expected_k{k}_pairs.csv"), "rb").read()
        assert r["csv"] == expect[expect.index(b"\n") + 1:]
        via_host = e.ingest_classify(0, merged, paired=True)
        assert via_host["csv"] == r["csv"] and (via_host["results"] == r["results"]).all()
        e.pairs_free(h)
        tx.free()
        e.ingest_free()


def _random_pairs(rng, genomes, n, unterminated=False, crlf=False):
    f1, f2 = [], []
    eol = b"\r\n" if crlf else b"\n"
    for i in range(n):
        g = genomes[int(rng.integers(len(genomes)))]
        seqs = []
        for _ in range(2):
            L = int(rng.choice([0, 1, 26, 31, 64, 100, 150, 151, 250, 700]))
            p = int(rng.integers(0, len(g) - L))
            s = bytearray(g[p:p + L])
            if L and rng.random() < 0.2:
                s[int(rng.integers(L))] = ord("N")
            seqs.append(bytes(s))
        kind = int(rng.integers(6))
        base = [b"r%d" % i, b"read_%d" % i, b"x" * 70 + b"%d" % i, b"q%d" % i, b"a%d" % i, b"p.%d" % i][kind]
        h1 = [b"@" + base + b"/1", b"@" + base + b" 1:N:0", b"@" + base + b"\tfirst", b"@@ " + base + b"/1 extra", b"@" + base, b"@/" + base + b"@1"][kind]
        h2 = [b"@" + base + b"/2", b"@" + base + b" 2:N:0 longer description", b"@" + base + b"\t2", b"@" + base + b"/2", b"@" + base, b"@" + base + b" x"][kind]
        f1.append(h1 + eol + seqs[0] + eol + b"+" + eol + b"I" * len(seqs[0]) + eol)
        f2.append(h2 + eol + seqs[1] + eol + b"+" + h2[1:] + eol + b"F" * len(seqs[1]) + eol)
    a, b = b"".join(f1), b"".join(f2)
    if unterminated:
        a, b = a[:-len(eol)], b[:-len(eol)]
    return a, b


def _genomes():
    out = []
    for fn, _ in gu.target_files_and_labels():
        out.append(b"".join(l.strip() for l in open(fn, "rb") if not l.startswith(b">")))
    return out


@pytest.mark.parametrize("variant", ["plain", "unterminated", "crlf", "few"])
def test_random_pairs_in_batches_of_strides(variant):
    rng = np.random.default_rng({"plain": 1, "unterminated": 2, "crlf": 3, "few": 4}[variant])
    n = 37 if variant == "few" else 3000
    fq1, fq2 = _random_pairs(rng, _genomes(), n, unterminated=variant == "unterminated", crlf=variant == "crlf")
    e, names = _engine()
    with e:
        tx = _Texts(e, fq1, fq2)
        h, n_rec, off, stride = tx.index()
        assert h is not None and n_rec == n and off.size == n // stride + 2
        recs = _merge(fq1, fq2)
        want_off = np.cumsum([0] + [len(r) for r in recs])
        for i in range(off.size):
            assert off[i] == want_off[min(i * stride, n)]
        e.ingest_alloc(2, 1 << 20, names, want_results=True)
        cuts = sorted({0, n} | {int(c) * stride for c in rng.integers(0, n // stride + 1, 6)})
        csv = []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            text = b"".join(recs[r0:r1])
            assert e.pairs_text(h, r0, r1) == text, (r0, r1)
            if len(text) > (1 << 20):
                continue
            d = e.pairs_classify(h, 1, r0, r1)
            v = e.ingest_classify(0, text, paired=True)
            assert d["status"] == v["status"], (r0, r1, d["status"], v["status"])
            if d["status"] == 0:
                assert d["csv"] == v["csv"] and (d["results"] == v["results"]).all() and d["n_reads"] == r1 - r0
                csv.append(d["csv"])
        assert csv or variant == "few"
        with pytest.raises(Exception):
            e.pairs_text(h, 1, n)            # not a stride boundary
        e.pairs_free(h)
        tx.free()
        e.ingest_free()


def test_texts_that_do_not_pair_up_are_refused():
    rng = np.random.default_rng(9)
    fq1, fq2 = _random_pairs(rng, _genomes(), 500)
    l2 = fq2.split(b"\n")
    cases = {
        "one record less": (fq1, b"\n".join(l2[4:]), 1),
        "three lines more": (fq1 + b"@x\nACGT\n+\n", fq2 + b"@x\nACGT\n+\n", 1),
        "header without @": (fq1, b"\n".join(l2[:400] + [b">" + l2[400][1:]] + l2[401:]), 2),
        "empty header line": (fq1, b"\n".join(l2[:400] + [b""] + l2[401:]), 2),
        "another id": (fq1, b"\n".join(l2[:800] + [b"@somebody_else/2"] + l2[801:]), 4),
        "an id of separators only": (fq1.replace(fq1.split(b"\n")[0], b"@ /@"), fq2.replace(l2[0], b"@@/"), 4),
    }
    e, _ = _engine()
    with e:
        for name, (a, b, status) in cases.items():
            tx = _Texts(e, a, b)
            h, st, _, _ = tx.index()
            assert h is None and st & status, (name, st)
            tx.free()
        # and the same pair of texts untouched is taken
        tx = _Texts(e, fq1, fq2)
        h, n_rec, _, _ = tx.index()
        assert h is not None and n_rec == 500
        e.pairs_free(h)
        tx.free()


@pytest.mark.parametrize("variant", ["plain", "unterminated", "crlf", "fasta", "fasta_crlf"])
def test_single_fastq_text_on_the_device(variant):
    """mic_text_*: the records of one FASTQ text cut at strides, copied into a slot on the device and classified there, against
    the same bytes handed over from the host; FASTA, a line count that is no multiple of four: refused."""
    import test_ingest as ti
    rng = np.random.default_rng({"plain": 11, "unterminated": 12, "crlf": 13, "fasta": 14, "fasta_crlf": 15}[variant])
    n = 2500
    fasta = variant.startswith("fasta")
    data = ti._random_reads(rng, _genomes(), n, fasta=fasta, crlf=variant.endswith("crlf"))
    if variant == "unterminated":
        data = data + b"@last\nACGTACGTACGTTTGACCA\n+\nIIIIIIIIIIIIIIIIIII"
        n += 1
    e, names = _engine()
    with e:
        d, nb, _ = e.gunzip_device(_gz(data))
        assert nb == len(data)
        h, n_rec, off, stride = e.text_index(d, nb)
        assert h is not None and n_rec == n and off[0] == 0 and off[-1] == len(data) and e.text_format(h) == (">" if fasta else "@")
        lines = data.split(b"\n")
        starts = np.cumsum([0] + [len(l) + 1 for l in lines])
        if fasta:
            rec_start = [int(starts[i]) for i, l in enumerate(lines) if l[:1] == b">"] + [len(data)]     # (multi-line sequences: a record is a '>' line to the next)
        else:
            rec_start = [int(min(starts[4 * r], len(data))) for r in range(n + 1)]
        assert len(rec_start) == n + 1
        for i in range(off.size):
            assert off[i] == rec_start[min(i * stride, n)]
        e.ingest_alloc(2, 2 << 20, names, want_results=True)
        cuts = sorted({0, n} | {int(c) * stride for c in rng.integers(0, n // stride + 1, 5)})
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            text = e.text_copy(h, r0, r1)
            assert text == data[off[r0 // stride]:(off[-1] if r1 == n else off[r1 // stride])]
            dv = e.text_classify(h, 1, r0, r1)
            v = e.ingest_classify(0, text)
            assert dv["status"] == v["status"], (r0, r1, dv["status"], v["status"])
            if dv["status"] == 0:
                assert dv["csv"] == v["csv"] and (dv["results"] == v["results"]).all()
        e.text_free(h)
        e.free_text(d)
        for bad, status in ((b"r1\nACGT\n>r2\nACGT\n", 2), (b"@x\nAC\n+\n" + (data if not fasta else b""), 1), (b"@r\nACGT\n+\n", 1), (b"\n>r\nAC\n", 2)):
            d, nb, _ = e.gunzip_device(_gz(bad))
            h, st, _, _ = e.text_index(d, nb)
            assert h is None and st & status, (bad[:10], st)
            e.free_text(d)
        e.ingest_free()
