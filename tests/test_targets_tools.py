"""Target-definition helpers: exe/getAccssnTaxID, exe/getfilesToTaxNodes, exe/getTargetsDef and the
set_targets.sh / make_metadata.sh flow, on a small hand-made taxonomy.  Expectations are written out by hand from the
reference's rules (src/getAccssnTaxID.cc, src/getfilesToTaxNodes.cc, src/getTargetsDef.cc) and, where the reference's own
tools are available (oracle/_ref/ref_get*, compiled from its sources), compared with their output byte for byte."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "exe")
REF = os.path.join(ROOT, "oracle", "_ref")

NODES = [  # id, parent, rank
    (1, 1, "no rank"), (131567, 1, "no rank"), (2, 131567, "superkingdom"), (1224, 2, "phylum"), (1236, 1224, "class"),
    (91347, 1236, "order"), (543, 91347, "family"), (561, 543, "genus"), (562, 561, "species"), (83333, 562, "no rank"),
    (1000, 561, "species group"), (1001, 1000, "species"), (570, 543, "genus"), (573, 570, "species"),
]
ACC2TAX = [("NC_000001", "NC_000001.1", 83333, 11), ("NC_000002", "NC_000002.2", 9999, 12), ("ACC3", "ACC3.1", 1001, 13),
           ("NC_777", "NC_777.1", 562, 14)]
MERGED = [(9999, 573), (12, 562)]
SEQS = [  # file name, first line, rest
    ("a.fa", ">NC_000001.1 Escherichia coli K-12", "ACGTACGT"),
    ("b.fna", ">gi|123|ref|NC_000002.2| Klebsiella pneumoniae", "ACGTAAAA"),
    ("c.fa", ">ACC3", "GGGGACGT"),
    ("d.fa", ">NC_404.1 nobody knows", "TTTTACGT"),
    ("e.fasta", "ACGTACGTAC", "ACGT"),          # no header on the first line: skipped
]


def need_tools():
    missing = [t for t in ("getAccssnTaxID", "getfilesToTaxNodes", "getTargetsDef") if not os.path.exists(os.path.join(EXE, t))]
    if missing:
        pytest.fail(f"exe/{missing[0]} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")


def make_taxonomy(d):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for i, p, r in NODES:
            f.write(f"{i}\t|\t{p}\t|\t{r}\t|\tXX\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t1\t|\t0\t|\t\t|\n")
    with open(os.path.join(d, "merged.dmp"), "w") as f:
        for a, b in MERGED:
            f.write(f"{a}\t|\t{b}\t|\n")
    with open(os.path.join(d, "nucl_accss"), "w") as f:
        f.write("accession\taccession.version\ttaxid\tgi\n")
        for a, av, t, g in ACC2TAX:
            f.write(f"{a}\t{av}\t{t}\t{g}\n")


def make_seqs(d):
    os.makedirs(d, exist_ok=True)
    out = []
    for name, first, rest in SEQS:
        p = os.path.join(d, name)
        with open(p, "w") as f:
            f.write(first + "\n" + rest + "\n")
        out.append(p)
    return out


def run(cmd, cwd, **kw):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=120, **kw)


def test_tools_against_hand_written_expectations(tmp_path):
    need_tools()
    tmp = str(tmp_path)
    tax = os.path.join(tmp, "taxonomy")
    make_taxonomy(tax)
    files = make_seqs(os.path.join(tmp, "seqs"))
    a, b, c, d, _ = files
    missing = os.path.join(tmp, "seqs", "gone.fa")
    lst = os.path.join(tmp, "list")
    with open(lst, "w") as f:
        f.write("\n".join(files[:2] + [missing] + files[2:]) + "\n")

    r = run([os.path.join(EXE, "getAccssnTaxID"), lst, os.path.join(tax, "nucl_accss"), os.path.join(tax, "merged.dmp")], tmp)
    assert r.returncode == 0
    assert r.stdout == (f"{missing}\tUNKNOWN\n"              # printed while the files are read
                        f"{a}\tNC_000001\t83333\n{b}\tNC_000002\t573\n{c}\tACC3\t1001\n{d}\tNC_404\t-1\n")
    assert "3 files were successfully mapped, and 1 unidentified" in r.stderr
    acc = os.path.join(tmp, "acc")
    open(acc, "w").write(r.stdout.split("\n", 1)[1])           # make_metadata.sh feeds the whole file; drop UNKNOWN here

    r = run([os.path.join(EXE, "getfilesToTaxNodes"), os.path.join(tax, "nodes.dmp"), acc], tmp)
    assert r.returncode == 0
    lineage = "543\t91347\t1236\t1224"
    assert r.stdout == (f"{a}\t83333\t562\t561\t{lineage}\n{b}\t573\t573\t570\t{lineage}\n"
                        f"{c}\t1001\t1001\t561\t{lineage}\n"        # 1000 is a 'species group': not a rank
                        f"{d}\t-1" + "\tUNKNOWN" * 6 + "\n")
    tid = os.path.join(tmp, "taxids")
    open(tid, "w").write(r.stdout)

    for rank, col in ((0, ("562", "573", "1001")), (1, ("561", "570", "561")), (5, ("1224",) * 3)):
        r = run([os.path.join(EXE, "getTargetsDef"), tid, str(rank)], tmp)
        assert r.returncode == 1                                # one file excluded
        assert r.stdout == "".join(f"{f}\t{t}\n" for f, t in zip((a, b, c), col))
        assert open(os.path.join(tmp, "files_excluded.txt")).read() == (
            "The following files have been excluded from the targets definition\n" + d + "\n")
    r = run([os.path.join(EXE, "getTargetsDef"), tid], tmp)    # the default rank is 1, as in the reference
    assert r.stdout.split()[1] == "561"
    assert run([os.path.join(EXE, "getTargetsDef"), tid, "6"], tmp).returncode == 1

    # an ID that nodes.dmp does not know (the reference does not terminate on it) and the root itself
    open(acc, "w").write(f"{a}\tX\t7777\n{b}\tY\t1\n{c}\tZ\t131567\n")
    r = run([os.path.join(EXE, "getfilesToTaxNodes"), os.path.join(tax, "nodes.dmp"), acc], tmp)
    assert r.stdout == "".join(f"{f}\t{t}" + "\tUNKNOWN" * 6 + "\n" for f, t in ((a, 7777), (b, 1), (c, 131567)))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_getTargetsDef")), reason="oracle/_ref tools not built")
def test_tools_match_the_reference_binaries(tmp_path):
    need_tools()
    tmp = str(tmp_path)
    tax = os.path.join(tmp, "taxonomy")
    make_taxonomy(tax)
    files = make_seqs(os.path.join(tmp, "seqs"))
    lst = os.path.join(tmp, "list")
    with open(lst, "w") as f:
        f.write("\n".join(files + [os.path.join(tmp, "nope.fa")]) + "\n")
    outs = {}
    for who, d, pre in (("mine", EXE, ""), ("ref", REF, "ref_")):
        w = os.path.join(tmp, who)
        os.makedirs(w)
        r1 = run([os.path.join(d, pre + "getAccssnTaxID"), lst, os.path.join(tax, "nucl_accss"), os.path.join(tax, "merged.dmp")], w)
        acc = os.path.join(w, "acc")
        open(acc, "w").write("".join(l + "\n" for l in r1.stdout.splitlines() if not l.endswith("UNKNOWN")))
        r2 = run([os.path.join(d, pre + "getfilesToTaxNodes"), os.path.join(tax, "nodes.dmp"), acc], w)
        tid = os.path.join(w, "tid")
        open(tid, "w").write(r2.stdout)
        r3 = [run([os.path.join(d, pre + "getTargetsDef"), tid] + ([str(k)] if k is not None else []), w) for k in (None, 0, 3, 5)]
        outs[who] = (r1.stdout, r1.returncode, r2.stdout, r2.returncode, [(x.stdout, x.returncode) for x in r3],
                     open(os.path.join(w, "files_excluded.txt")).read())
    assert outs["mine"] == outs["ref"]


def test_set_targets_flow_for_a_custom_database(tmp_path):
    """set_targets.sh <dir> custom --genus with the taxonomy already in place: no download, same files as the reference's
    scripts leave behind (.DBDirectory, .settings, targets.txt, files_excluded.txt, <dir>/custom_1_canonical/)."""
    need_tools()
    cwd = str(tmp_path)
    db = os.path.join(cwd, "DBD")
    make_taxonomy(os.path.join(db, "taxonomy"))
    open(os.path.join(db, ".taxondata"), "w").close()
    make_seqs(os.path.join(db, "Custom"))
    r = run([os.path.join(ROOT, "set_targets.sh"), db, "custom", "--genus"], cwd)
    assert r.returncode == 0, r.stdout + r.stderr
    cust = os.path.join(db, "Custom")
    got = sorted(open(os.path.join(db, "targets.txt")).read().splitlines())
    assert got == sorted([f"{cust}/a.fa\t561", f"{cust}/b.fna\t570", f"{cust}/c.fa\t561"])
    assert open(os.path.join(cwd, ".DBDirectory")).read() == db + "\n"
    assert open(os.path.join(cwd, ".settings")).read() == f"-T {db}/targets.txt\n-D {db}/custom_1_canonical/\n"
    assert os.path.isdir(os.path.join(db, "custom_1_canonical"))
    assert open(os.path.join(db, "files_excluded.txt")).read().splitlines()[1:] == [f"{cust}/d.fa"]
    assert "Collecting metadata of custom... " in r.stdout and "done." in r.stdout
    # a second run with another rank reuses the metadata and replaces the definition
    r = run([os.path.join(ROOT, "set_targets.sh"), db, "custom"], cwd)
    assert r.returncode == 0
    assert sorted(l.split("\t")[1] for l in open(os.path.join(db, "targets.txt")).read().splitlines()) == ["1001", "562", "573"]
    assert open(os.path.join(cwd, ".settings")).read().endswith(f"-D {db}/custom_0_canonical/\n")
    # unknown option and empty Custom directory
    assert "Failed to recognize this parameter: --kingdom" in run([os.path.join(ROOT, "set_targets.sh"), db, "custom", "--kingdom"], cwd).stdout
    db2 = os.path.join(cwd, "EMPTY")
    make_taxonomy(os.path.join(db2, "taxonomy"))
    open(os.path.join(db2, ".taxondata"), "w").close()
    r = run([os.path.join(ROOT, "set_targets.sh"), db2, "custom"], cwd)
    assert "The database directory 'Custom' is empty." in r.stdout and r.returncode != 0


def test_human_metadata_uses_the_fixed_lineage(tmp_path):
    need_tools()
    cwd = str(tmp_path)
    db = os.path.join(cwd, "DBD")
    make_taxonomy(os.path.join(db, "taxonomy"))
    open(os.path.join(db, ".taxondata"), "w").close()
    os.makedirs(os.path.join(db, "Human"))
    chrs = [os.path.join(db, "Human", f"chr{i}.fa") for i in (1, 2)]
    for c in chrs:
        open(c, "w").write(">chr\nACGT\n")
    open(os.path.join(db, ".human"), "w").write("\n".join(chrs) + "\n")
    r = run([os.path.join(ROOT, "set_targets.sh"), db, "human", "--family"], cwd)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(os.path.join(db, "targets.txt")).read() == "".join(f"{c}\t9604\n" for c in chrs)
    assert open(os.path.join(cwd, ".settings")).read() == f"-T {db}/targets.txt\n-D {db}/human_2_canonical/\n"
