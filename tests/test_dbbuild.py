"""GPU database builder (mic_db_build) against the databases the REFERENCE wrote for the same targets
(tests/golden/db_*.npz were produced by EHashtable::addElement/SortAllHashTable/RemoveCommon/Write)."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _targets():
    files, labels, names = [], [], []
    for fn, label in gu.target_files_and_labels():
        if label not in names:
            names.append(label)
        files.append(fn)
        labels.append(names.index(label))
    return files, labels


@pytest.mark.parametrize("name", ["light_k27_u32", "light_k31_u64", "light_k20_u16", "light_k32_u64", "full_k31_u32"])
@pytest.mark.parametrize("parts", [0, 3])
def test_built_database_is_byte_identical_to_the_reference(name, parts, tmp_path):
    from cuclark_amd import host
    if name.startswith("full") and parts:
        pytest.skip("one pass is enough for the 1.6e9-bucket case")
    db = gu.load_golden_db(name)
    files, labels = _targets()
    prefix = str(tmp_path / "built")
    n = host.build_db(files, labels, db["k"], db["htsize"], prefix, parts=parts)
    assert n == db["ky"].size
    ky = np.fromfile(prefix + ".ky", dtype=gu.KEY_DTYPE[db["key_bytes"]])
    lb = np.fromfile(prefix + ".lb", dtype=np.uint16)
    assert os.path.getsize(prefix + ".sz") == db["htsize"]
    sz = np.fromfile(prefix + ".sz", dtype=np.uint8)
    nz = np.flatnonzero(sz)
    assert (nz == db["sz_idx"].astype(np.int64)).all() and (sz[nz] == db["sz_val"]).all()
    assert (ky == db["ky"]).all() and (lb == db["lb"]).all()
    os.remove(prefix + ".sz")


def test_min_count_and_shared_kmers(tmp_path):
    """-t: a k-mer needs more than min_count occurrences inside its single label; the oracle decides what to expect."""
    from cuclark_amd import host
    rng = np.random.default_rng(5)
    k, htsize = 21, 99991
    unit = "".join(rng.choice(list("ACGT"), 300))
    other = "".join(rng.choice(list("ACGT"), 300))
    t0 = tmp_path / "a.fa"; t1 = tmp_path / "b.fa"; t2 = tmp_path / "c.fq"
    t0.write_text(f">a\n{unit}\n>a2\n{unit[:150]}\n")                 # first 150 nt occur twice in label 0
    t1.write_text(f">b\n{other}\n{unit[200:260]}\n")                  # shares k-mers of unit[200:260] with label 0
    t2.write_text(f"@q\n{other[:100]}\n+\n{'I' * 100}\n")              # FASTQ target, same label as b
    prefix = str(tmp_path / "db")
    o = gu.oracle()

    def kmers(s):
        return [o.canonical(int("".join(str("TGCA".index(c)) for c in s[i:i + k]), 4), k) for i in range(len(s) - k + 1)]
    occ = {}
    for lab, seqs in ((0, [unit, unit[:150]]), (1, [other + unit[200:260], other[:100]])):
        for s in seqs:
            for c in kmers(s):
                occ.setdefault(c, []).append(lab)
    for min_count in (0, 1):
        n = host.build_db([str(t0), str(t1), str(t2)], [0, 1, 1], k, htsize, prefix, min_count=min_count)
        expect = {c: labs[0] for c, labs in occ.items() if len(set(labs)) == 1 and min(len(labs), 254) > min_count}
        assert n == len(expect)
        sz = np.fromfile(prefix + ".sz", dtype=np.uint8)
        ky = np.fromfile(prefix + ".ky", dtype=gu.KEY_DTYPE[host.key_bytes_rule(htsize, k)])
        lb = np.fromfile(prefix + ".lb", dtype=np.uint16)
        rem = np.repeat(np.arange(htsize), sz)
        got = {int(q) * htsize + int(r): int(l) for q, r, l in zip(ky, rem, lb)}
        assert got == expect


@pytest.mark.parametrize("name,gap", [("lightgap4_k27_u32", 4), ("lightgap5_k31_u64", 5), ("lightgap1_k20_u16", 1)])
def test_light_database_is_byte_identical_to_the_reference(name, gap, tmp_path):
    """cuCLARK-l's sampled database: non-overlapping k-blocks, every gap-th one (CuCLARK_hh.hh:694-895); the fixtures
    were written by the reference's EHashtable fed with that enumeration (oracle/ref_table_driver.cc)."""
    from cuclark_amd import host
    db = gu.load_golden_db(name)
    files, labels = _targets()
    prefix = str(tmp_path / "built")
    n = host.build_db(files, labels, db["k"], db["htsize"], prefix, light_gap=gap, key_bytes=db["key_bytes"])
    assert n == db["ky"].size
    ky = np.fromfile(prefix + ".ky", dtype=gu.KEY_DTYPE[db["key_bytes"]])
    lb = np.fromfile(prefix + ".lb", dtype=np.uint16)
    sz = np.fromfile(prefix + ".sz", dtype=np.uint8)
    nz = np.flatnonzero(sz)
    assert (nz == db["sz_idx"].astype(np.int64)).all() and (sz[nz] == db["sz_val"]).all()
    assert (ky == db["ky"]).all() and (lb == db["lb"]).all()


def test_light_database_long_runs_and_block_numbering(tmp_path):
    """Runs longer than one packed part (65528 nt), runs shorter than k, Ns, line breaks and several records: the block
    counter runs through the whole file and only whole blocks count."""
    from cuclark_amd import host
    rng = np.random.default_rng(11)
    k, gap, htsize = 21, 3, 1000003
    o = gu.oracle()

    def rand(n):
        return "".join(rng.choice(list("ACGT"), n))
    recs_a = [rand(150000) + "N" + rand(10) + "NN" + rand(70001), rand(20), rand(65528) + "N" + rand(65529)]
    recs_b = [rand(4000) + "n" + rand(41)]
    paths, labels = [], []
    for i, recs in enumerate((recs_a, recs_b)):
        p = tmp_path / f"t{i}.fa"
        with open(p, "w") as f:
            for j, r in enumerate(recs):
                f.write(f">r{j}\n")
                for a in range(0, len(r), 70):
                    f.write(r[a:a + 70] + "\n")
        paths.append(str(p)); labels.append(i)
    occ = {}
    for lab, recs in enumerate((recs_a, recs_b)):
        it = 0                                       # per file
        for r in recs:
            for run in __import__("re").split("[^ACGTacgt]", r):
                for b in range(len(run) // k):
                    if it % gap == 0:
                        s = run[b * k:(b + 1) * k].upper()
                        c = o.canonical(int("".join(str("TGCA".index(ch)) for ch in s), 4), k)
                        occ.setdefault(c, set()).add(lab)
                    it += 1
    expect = {c: next(iter(l)) for c, l in occ.items() if len(l) == 1}
    prefix = str(tmp_path / "db")
    n = host.build_db(paths, labels, k, htsize, prefix, light_gap=gap)
    assert n == len(expect)
    sz = np.fromfile(prefix + ".sz", dtype=np.uint8)
    ky = np.fromfile(prefix + ".ky", dtype=gu.KEY_DTYPE[host.key_bytes_rule(htsize, k)])
    lb = np.fromfile(prefix + ".lb", dtype=np.uint16)
    rem = np.repeat(np.arange(htsize), sz)
    got = {int(q) * htsize + int(r): int(l) for q, r, l in zip(ky, rem, lb)}
    assert got == expect


def test_cli_builds_a_missing_database(tmp_path):
    """First run without .sz/.ky/.lb: the binary creates the database (as the reference does) and classifies."""
    import subprocess
    tmp = str(tmp_path)
    t = os.path.join(tmp, "targets.txt")
    with open(t, "w") as f:
        for fn, label in gu.target_files_and_labels():
            f.write(f"{fn} {label}\n")
    d = os.path.join(tmp, "DB")
    os.makedirs(d)
    exe = os.path.join(gu.ROOT, "exe", "cuCLARK")
    out = os.path.join(tmp, "res")
    r = subprocess.run([exe, "-k", "27", "--htsize", "57777779", "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k27.fa"),
                        "-R", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Starting the creation of the database of targets specific 27-mers from input files..." in r.stderr
    assert "Creating database in disk..." in r.stderr
    assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_fa.csv"), "rb").read()
    db = gu.load_golden_db("light_k27_u32")
    ky = np.fromfile(os.path.join(d, "db_central_k27_t6_s57777779_m0.tsk.ky"), dtype=np.uint32)
    assert (ky == db["ky"]).all()


def test_cli_light_builds_the_sampled_database(tmp_path):
    """cuCLARK-l without a database builds the light one (gap 4 by default) and classifies with it; the expected CSV
    comes from the oracle on the reference-written light database."""
    import subprocess
    tmp = str(tmp_path)
    t = os.path.join(tmp, "targets.txt")
    with open(t, "w") as f:
        for fn, label in gu.target_files_and_labels():
            f.write(f"{fn} {label}\n")
    d = os.path.join(tmp, "DB")
    os.makedirs(d)
    exe = os.path.join(gu.ROOT, "exe", "cuCLARK-l")
    out = os.path.join(tmp, "res")
    reads = os.path.join(gu.GOLDEN, "reads_k27.fa")
    r = subprocess.run([exe, "-T", t, "-D", d, "-O", reads, "-R", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Creating light database in disk..." in r.stderr
    db = gu.load_golden_db("lightgap4_k27_u32")
    ky = np.fromfile(os.path.join(d, "db_central_k27_t6_s57777779_m0_light_4.tsk.ky"), dtype=np.uint32)
    assert ky.size == db["ky"].size and (ky == db["ky"]).all()
    odb, _ = gu.oracle_db_from_golden("lightgap4_k27_u32")
    text, _ = odb.classify_file(27, open(reads, "rb").read(), gu.target_names(), False, False)
    assert open(out + ".csv", "rb").read() == text
