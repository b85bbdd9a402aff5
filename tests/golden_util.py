"""Helpers shared by the tests: golden fixtures, the oracle binding, small synthetic generators."""
import os
import re
import sys

import numpy as np

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TESTS)
GOLDEN = os.path.join(TESTS, "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEY_DTYPE = {2: np.uint16, 4: np.uint32, 8: np.uint64}
_ORACLE = None


def oracle():
    global _ORACLE
    if _ORACLE is None:
        from oracle.binding import Oracle
        _ORACLE = Oracle()
    return _ORACLE


def golden_db_names():
    return sorted(f[3:-4] for f in os.listdir(GOLDEN) if f.startswith("db_") and f.endswith(".npz"))


def load_golden_db(name):
    z = np.load(os.path.join(GOLDEN, f"db_{name}.npz"))
    return dict(htsize=int(z["htsize"]), k=int(z["k"]), key_bytes=int(z["key_bytes"]), sz_idx=z["sz_idx"],
                sz_val=z["sz_val"], ky=z["ky"], lb=z["lb"])


def golden_sizes(db):
    sz = np.zeros(db["htsize"], np.uint8)
    sz[db["sz_idx"].astype(np.int64)] = db["sz_val"]
    return sz


def materialize_db(name, out_dir):
    """Write <out_dir>/<name>.{sz,ky,lb} exactly as the reference wrote them; returns (prefix, meta)."""
    db = load_golden_db(name)
    prefix = os.path.join(out_dir, name)
    if not os.path.exists(prefix + ".sz"):
        golden_sizes(db).tofile(prefix + ".sz")
        db["ky"].tofile(prefix + ".ky")
        db["lb"].tofile(prefix + ".lb")
    return prefix, db


def target_files_and_labels():
    out = []
    with open(os.path.join(GOLDEN, "targets.tsv")) as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                fn, label = line.split("\t")
                out.append((os.path.join(GOLDEN, fn), label))
    return out


def target_names():
    names = []
    for _, label in target_files_and_labels():
        if label not in names:
            names.append(label)
    return names


def merge_pairs(fq1, fq2):
    """file.cc:205-268: '>' + first token of the header (split on ' ', '/', TAB, '@') + seq1 'N' seq2."""
    out = []
    l1 = fq1.decode().split("\n")
    l2 = fq2.decode().split("\n")
    i = 0
    while i + 1 < len(l1) and i + 1 < len(l2):
        if l1[i][:1] == "@" and l2[i][:1] == "@":
            t1 = [t for t in re.split(r"[ /\t@]", l1[i]) if t][0]
            out.append(f">{t1}\n{l1[i + 1]}N{l2[i + 1]}\n")
            i += 4
        else:
            i += 1
    return "".join(out).encode()


DB_FOR_K = {27: "light_k27_u32", 31: "light_k31_u64"}


def expected_csv_cases():
    """(case name, k, db fixture, input bytes, paired, extended)"""
    cases = []
    for k in (27, 31):
        fa = open(os.path.join(GOLDEN, f"reads_k{k}.fa"), "rb").read()
        fq = open(os.path.join(GOLDEN, f"reads_k{k}.fq"), "rb").read()
        p1 = open(os.path.join(GOLDEN, f"pairs_k{k}_1.fq"), "rb").read()
        p2 = open(os.path.join(GOLDEN, f"pairs_k{k}_2.fq"), "rb").read()
        cases.append((f"k{k}_fa", k, DB_FOR_K[k], fa, False, False))
        cases.append((f"k{k}_fa_ext", k, DB_FOR_K[k], fa, False, True))
        cases.append((f"k{k}_fq", k, DB_FOR_K[k], fq, False, False))
        cases.append((f"k{k}_pairs", k, DB_FOR_K[k], merge_pairs(p1, p2), True, False))
    return cases


def oracle_db_from_golden(name, sampling=1):
    db = load_golden_db(name)
    return oracle().db_from_arrays(golden_sizes(db), db["ky"], db["lb"], sampling), db


def write_expected_csvs():
    names = target_names()
    cache = {}
    for case, k, dbname, data, paired, ext in expected_csv_cases():
        if dbname not in cache:
            cache[dbname] = oracle_db_from_golden(dbname)[0]
        text, _ = cache[dbname].classify_file(k, data, names, paired, ext)
        with open(os.path.join(GOLDEN, f"expected_{case}.csv"), "wb") as f:
            f.write(text)


# ---------------------------------------------------------------- small synthetic databases

def random_db(rng, htsize, n_elems, k, key_bytes, n_labels, max_bucket=255):
    """Random valid database (distinct ascending keys per bucket): returns sizes, keys, labels and the
    canonical k-mers it contains."""
    o = oracle()
    limit = (1 << (2 * k)) if k < 32 else (1 << 64)
    canon = set()
    while len(canon) < n_elems:
        v = int(rng.integers(0, limit, dtype=np.uint64)) if k < 32 else int(rng.integers(0, 2 ** 63, dtype=np.uint64)) * 2 + int(rng.integers(0, 2))
        canon.add(o.canonical(v, k))
    canon = sorted(canon, key=lambda c: (c % htsize, c // htsize))
    sizes = np.zeros(htsize, np.int64)
    keep = []
    for c in canon:
        r = c % htsize
        if sizes[r] < max_bucket:
            sizes[r] += 1
            keep.append(c)
    keys = np.array([c // htsize for c in keep], dtype=np.uint64)
    assert keys.max(initial=0) < (1 << (8 * key_bytes))
    labels = rng.integers(0, n_labels, len(keep)).astype(np.uint16)
    return sizes.astype(np.uint8), keys.astype(KEY_DTYPE[key_bytes]), labels, np.array(keep, dtype=np.uint64)


def kmer_to_ascii(v, k):
    return "".join("TGCA"[(int(v) >> (2 * (k - 1 - i))) & 3] for i in range(k))
