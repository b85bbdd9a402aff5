"""Pins taken from the REFERENCE's own host code (compiled in place, oracle/Makefile; vectors made by
tests/golden/make_golden_pairs.py): the paired-end merge mergePairedFiles (file.cc:205-268) and the k-mer codec
getKmers / getReverse (kmersConversion.cc:39-68).  Checked against them: the product's serial pair reader (PairedSource) and
the loaders' parallel merger (PairedFileFeeder) through `exe/cuCLARK --merge-pairs` (no device involved), the oracle's
restatements, the test helper golden_util.merge_pairs, the host packer."""
import json
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

EXE = os.path.join(gu.ROOT, "exe", "cuCLARK")
EDGE = json.load(open(os.path.join(gu.GOLDEN, "pairs_edge.json")))["cases"]
CODEC = json.load(open(os.path.join(gu.GOLDEN, "codec_vectors.json")))["vectors"]


def _product_merge(tmp_path, f1, f2, *mode):
    p1, p2, out = (str(tmp_path / n) for n in ("a.fq", "b.fq", "m.fa"))
    open(p1, "wb").write(f1)
    open(p2, "wb").write(f2)
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([EXE, "--merge-pairs", p1, p2, out, *mode], capture_output=True, text=True, timeout=120)
    return r.returncode, (open(out, "rb").read() if os.path.exists(out) else None), r.stderr


@pytest.fixture(scope="module", autouse=True)
def _built(lib):
    assert os.path.exists(EXE)


@pytest.mark.parametrize("k", [27, 31])
def test_golden_pair_files_merge_like_the_reference(k, tmp_path):
    f1 = open(os.path.join(gu.GOLDEN, f"pairs_k{k}_1.fq"), "rb").read()
    f2 = open(os.path.join(gu.GOLDEN, f"pairs_k{k}_2.fq"), "rb").read()
    want = open(os.path.join(gu.GOLDEN, f"pairs_k{k}_merged.fa"), "rb").read()
    assert gu.merge_pairs(f1, f2) == want                       # the helper the CSV goldens were made with
    rc, got, err = _product_merge(tmp_path, f1, f2)
    assert rc == 0 and got == want, err
    for threads, batch in ((1, 1 << 20), (3, 200), (8, 1)):     # batches of a few records down to one record each
        rc, got, err = _product_merge(tmp_path, f1, f2, "parallel", str(threads), str(batch))
        assert rc == 0 and got == want, (threads, batch, err)


@pytest.mark.parametrize("case", EDGE, ids=[c["name"] for c in EDGE])
def test_pair_edge_cases_end_like_the_reference(case, tmp_path):
    f1, f2 = case["f1"].encode("latin1"), case["f2"].encode("latin1")
    want = None if case["merged"] is None else case["merged"].encode("latin1")
    rc, got, err = _product_merge(tmp_path, f1, f2)
    if case["rc"] == 0:
        assert rc == 0 and got == want, err
        assert gu.merge_pairs(f1, f2) == want
    else:
        # the reference: perror("<message>") + exit(1); the message is the contract, perror's ": <errno text>" is not
        msg = case["stderr"].split(":")[0] + ":" + case["stderr"].split(":")[1] if case["stderr"].startswith("Error:") else case["stderr"]
        assert rc == 1 and got is None and err.strip() == msg.strip(), (err, msg)
    # the parallel merger either produces the same text or hands the files to the serial reader (exit 3), never something else
    rc, got, err = _product_merge(tmp_path, f1, f2, "parallel", "4", "64")
    assert (rc == 0 and case["rc"] == 0 and got == want) or (rc == 3 and got is None), (rc, err)


def test_codec_vectors(orc):
    """getKmers / getReverse of the reference vs the oracle's codec, and vs the host packer's containers."""
    from cuclark_amd import host
    for v in CODEC:
        k, seq = v["k"], v["seq"].encode()
        assert orc.L.orc_kmer_from_ascii(seq, k) == v["fwd"], v
        assert orc.revcomp(v["fwd"], k) == v["rev"] and orc.revcomp(v["rev"], k) == v["fwd"]
        assert orc.canonical(v["fwd"], k) == min(v["fwd"], v["rev"])
    for k in sorted({v["k"] for v in CODEC}):
        vs = [v for v in CODEC if v["k"] == k]
        data = "".join(f">s{i}\n{v['seq']}\n" for i, v in enumerate(vs)).encode()
        idx = host.index_reads(data)
        rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
        for i, v in enumerate(vs):          # one part of k nucleotides: length slot, then ceil(k / 8) containers, first nt on top
            p = int(rp[i])
            assert cont[p] == k
            val = 0
            for c in cont[p + 1:p + 1 + (k + 7) // 8]:
                val = (val << 16) | int(c)
            assert val >> (16 * ((k + 7) // 8) - 2 * k) == v["fwd"], v


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["direct", "minimizer", "super", "super2"])
def test_codec_vectors_through_the_gpu_query(layout, monkeypatch, orc):
    """One k-mer per read: a database of min(getKmers, getReverse) of the reference's vectors answers the k-mer and the ASCII of
    its reverse complement, and nothing else."""
    from cuclark_amd import MiClarkDB, host
    monkeypatch.setenv("MIC_LAYOUT", layout)
    for k in (20, 27, 31, 32):
        vs = [v for v in CODEC if v["k"] == k]
        canon = sorted({min(v["fwd"], v["rev"]) for v in vs[::2]})            # every second vector is in the database
        htsize = 100003
        order = sorted(canon, key=lambda c: (c % htsize, c // htsize))
        sizes = np.zeros(htsize, np.uint8)
        for c in order:
            sizes[c % htsize] += 1
        keys = np.array([c // htsize for c in order], dtype=np.uint64)
        labels = np.array([c % 7 for c in order], dtype=np.uint16)
        comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
        seqs = [v["seq"] for v in vs] + [v["seq"].encode().translate(comp)[::-1].decode() for v in vs]
        data = "".join(f">s{i}\n{s}\n" for i, s in enumerate(seqs)).encode()
        idx = host.index_reads(data)
        rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
        with MiClarkDB(k, 7) as e:
            e.read_arrays(sizes, keys, labels)
            res = e.classify_packed(rp, cont)
        want_in = np.array([min(v["fwd"], v["rev"]) in set(canon) for v in vs] * 2)
        assert ((res[:, 0] == 1) == want_in).all()
        lab = np.array([min(v["fwd"], v["rev"]) % 7 + 1 for v in vs] * 2)
        assert (res[want_in, 1] == lab[want_in]).all()


def _strip_reference(text):
    """header + sequence line of every four-line record, the rest dropped - a trailing incomplete line is kept as it stands when
    it is a header or a sequence line (CuCLARK_hh.hh:1496-1523 steps over the '+' and quality lines and reads nothing of them)"""
    out = []
    for i, line in enumerate(text.split(b"\n")[:-1]):
        if i % 4 < 2:
            out.append(line + b"\n")
    tail = text.split(b"\n")[-1]
    if tail and (text.count(b"\n") % 4) < 2:
        out.append(tail)
    return b"".join(out)


@pytest.mark.parametrize("piece", [1, 5, 31, 32, 33, 63, 64, 65, 127, 4096, 1 << 20])
def test_fastq_stripper_vector_form_equals_scalar_form(tmp_path, piece):
    """The loaders drop the '+' and quality lines while they copy FASTQ into the device's slots; the AVX2 form (64 bytes per step)
    must produce the bytes of the scalar form for every split of the input into calls - lines longer than a step, empty reads,
    CR LF line ends, a file cut inside any of the four lines."""
    rng = np.random.default_rng(piece)
    recs = []
    for i in range(3000):
        L = int(rng.choice([0, 1, 31, 32, 33, 63, 64, 65, 100, 150, 151, 300, 1000]))
        seq = "".join(rng.choice(list("ACGTN"), L))
        eol = "\r\n" if i % 97 == 0 else "\n"
        recs.append(f"@r{i} d{'x' * int(rng.integers(0, 90))}{eol}{seq}{eol}+{eol}{'I' * L}{eol}")
    text = "".join(recs).encode()
    for cut in (len(text), len(text) - 1, len(text) - 160, len(text) // 2 + 7, 2, 0):
        src = tmp_path / "in.fq"
        src.write_bytes(text[:cut])
        outs = []
        for mode in ([], ["scalar"]):
            out = tmp_path / ("o_" + "".join(mode))
            r = subprocess.run([EXE, "--strip-fastq", str(src), str(out), str(piece), *mode], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            outs.append(out.read_bytes())
        assert outs[0] == outs[1], (piece, cut)
        assert outs[0] == _strip_reference(text[:cut]), (piece, cut)
