"""The cuCLARK-compatible command line (exe/cuCLARK, exe/cuCLARK-l): argv contract of main.cc:74-320, .csv output,
stdout lines scripts parse (CuCLARK_hh.hh:1938-1944)."""
import os
import re
import subprocess

import pytest

import golden_util as gu

EXE = os.path.join(gu.ROOT, "exe", "cuCLARK")
EXE_L = os.path.join(gu.ROOT, "exe", "cuCLARK-l")


def _run(args, **kw):
    if not os.environ.get("MIC_TEST_TIMING"):
        return subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)
    import time                                   # where the suite's wall time goes: one line per run of the command line
    t0 = time.time()
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)
    env = kw.get("env") or {}
    with open(os.environ["MIC_TEST_TIMING"], "a") as f:
        f.write(f"{time.time() - t0:7.2f} s  {' '.join(os.path.basename(a) for a in args[:1] + args[5:])}  "
                f"{ {k: v for k, v in env.items() if k.startswith('MIC_')} }\n" + "".join("    " + l + "\n" for l in r.stderr.splitlines() if "[timing]" in l or "[load]" in l))
    return r


def _run_many(jobs, workers=4):
    """Independent runs of the command line side by side (each its own process; at most `workers` of them use the GPU at once, next
    to the test process itself - the pool allows six).  jobs: callables; returns their results in order.  What is checked is
    unchanged - the suite's wall time is what this buys (GPUTEST budget)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(lambda j: j(), jobs))


def _targets_file(tmp):
    p = os.path.join(tmp, "targets.txt")
    with open(p, "w") as f:
        for fn, label in gu.target_files_and_labels():
            f.write(f"{fn} {label}\n")
    return p


def _db_dir(tmp, name, light):
    """Lay the golden DB out under the reference's file name (CuCLARK_hh.hh:580-591)."""
    d = os.path.join(tmp, "DB")
    os.makedirs(d, exist_ok=True)
    meta = gu.load_golden_db(name)
    base = f"db_central_k{meta['k']}_t6_s{meta['htsize']}_m0" + ("_light_4" if light else "") + ".tsk"
    prefix, _ = gu.materialize_db(name, d)
    for ext in (".sz", ".ky", ".lb"):
        os.replace(prefix + ext, os.path.join(d, base + ext))
    return d


def test_help_version_and_argument_errors(lib):
    r = _run([EXE, "--version"])
    assert r.returncode == 0 and "Version: 1.1" in r.stdout and "CLARK version 1.1.3" in r.stdout
    r = _run([EXE, "--help"])
    assert r.returncode == 0 and "-T <fileTargets>" in r.stdout
    r = _run([EXE, "-k", "31"])
    assert r.returncode != 0 and "at least four  parameters are necessary" in r.stderr
    r = _run([EXE, "-k", "40", "-T", "x", "-D", "y", "-O", "z", "-R", "w"])
    assert r.returncode == 1 and "The k-mer length should be in [2,32]." in r.stderr
    r = _run([EXE, "-T", "/nonexistent", "-D", "y", "-O", "z", "-R", "w"])
    assert r.returncode == 1 and "Failed to find/read the file of the targets definition" in r.stderr
    r = _run([EXE, "-n", "4", "-b", "2", "-T", "x", "-D", "y", "-O", "z"])
    assert r.returncode == 1 and "number of batches should be higher" in r.stderr
    r = _run([EXE, "--bogus", "1", "2", "3", "4", "5"])
    assert r.returncode == 1 and "Failed to recognize option: --bogus" in r.stderr


@pytest.mark.gpu
def test_light_cli_matches_golden(tmp_path):
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    out = os.path.join(tmp, "res")
    r = _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k27.fa"), "-R", out, "-n", "2", "-b", "5"])
    assert r.returncode == 0, r.stderr
    assert re.search(r"Processing file '.*reads_k27.fa' in 5 batches using 2 CPU thread\(s\)\.", r.stdout)
    assert re.search(r" - Assignment time: [0-9.e+-]+ s\. Speed: \d+ objects/min\. \(131 objects\)\.", r.stdout)
    assert f" - Results stored in {out}.csv" in r.stdout
    assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_fa.csv"), "rb").read()
    # FASTQ, extended, and paired-end through the same binary
    for flag, src, exp in (([], ["-O", os.path.join(gu.GOLDEN, "reads_k27.fq")], "expected_k27_fq.csv"),
                           (["--extended"], ["-O", os.path.join(gu.GOLDEN, "reads_k27.fa")], "expected_k27_fa_ext.csv"),
                           ([], ["-P", os.path.join(gu.GOLDEN, "pairs_k27_1.fq"), os.path.join(gu.GOLDEN, "pairs_k27_2.fq")],
                            "expected_k27_pairs.csv")):
        out2 = os.path.join(tmp, "res_" + exp)
        r = _run([EXE_L, "-T", t, "-D", d, *src, "-R", out2, *flag])
        assert r.returncode == 0, r.stderr
        assert open(out2 + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, exp), "rb").read(), exp
    # every table layout behind the same command line, the two-strand one (per-run kernel) included
    for layout in ("super2", "super", "minimizer", "direct"):
        out3 = os.path.join(tmp, "res_" + layout)
        r = _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k27.fq"), "-R", out3, "-n", "3"], env=dict(os.environ, MIC_LAYOUT=layout))
        assert r.returncode == 0, r.stderr
        assert open(out3 + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_fq.csv"), "rb").read(), layout


@pytest.mark.gpu
def test_gzip_input_and_launcher_script(tmp_path):
    """gzip input is inflated by the binary; classify_metagenome.sh reads .settings like the reference's script."""
    import gzip
    import shutil
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    gz = os.path.join(tmp, "reads.fa.gz")
    with open(os.path.join(gu.GOLDEN, "reads_k27.fa"), "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    open(os.path.join(tmp, ".settings"), "w").write(f"-T {t}\n-D {d}/\n")
    script = os.path.join(gu.ROOT, "classify_metagenome.sh")
    r = _run([script, "-O", gz, "-R", os.path.join(tmp, "viascript"), "--light", "--gzipped", "-n", "2"], cwd=tmp)
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(tmp, "viascript.csv"), "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_fa.csv"), "rb").read()
    # paired gz
    p1, p2 = os.path.join(tmp, "p1.fq.gz"), os.path.join(tmp, "p2.fq.gz")
    for src, dst in ((os.path.join(gu.GOLDEN, "pairs_k27_1.fq"), p1), (os.path.join(gu.GOLDEN, "pairs_k27_2.fq"), p2)):
        with open(src, "rb") as fi, gzip.open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    r = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", os.path.join(tmp, "pairs")])
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(tmp, "pairs.csv"), "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_pairs.csv"), "rb").read()


@pytest.mark.gpu
def test_full_cli_matches_golden_and_missing_targets(tmp_path):
    tmp = str(tmp_path)
    d = _db_dir(tmp, "full_k31_u32", light=False)
    t = _targets_file(tmp)
    out = os.path.join(tmp, "res")
    r = _run([EXE, "-k", "31", "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k31.fa"), "-R", out])
    assert r.returncode == 0, r.stderr
    assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k31_fa.csv"), "rb").read()
    # list-of-files mode: -O and -R name parallel lists (CuCLARK_hh.hh:413-427)
    lo, lr = os.path.join(tmp, "objs.txt"), os.path.join(tmp, "ress.txt")
    open(lo, "w").write(os.path.join(gu.GOLDEN, "reads_k31.fa") + "\n" + os.path.join(gu.GOLDEN, "reads_k31.fq") + "\n")
    open(lr, "w").write(os.path.join(tmp, "l1") + "\n" + os.path.join(tmp, "l2") + "\n")
    r = _run([EXE, "-k", "31", "-T", t, "-D", d, "-O", lo, "-R", lr])
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(tmp, "l1.csv"), "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k31_fa.csv"), "rb").read()
    assert open(os.path.join(tmp, "l2.csv"), "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k31_fq.csv"), "rb").read()
    for ext in (".sz", ".ky", ".lb"):
        for f in os.listdir(d):
            if f.endswith(ext):
                os.remove(os.path.join(d, f))
    # no database and unreadable target genomes -> the reference's message for a missing target file
    bad = os.path.join(tmp, "bad_targets.txt")
    open(bad, "w").write("/nonexistent/genome.fa L1\n")
    r = _run([EXE, "-k", "31", "-T", bad, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k31.fa"), "-R", out])
    assert r.returncode != 0 and "Failed to open file: /nonexistent/genome.fa defined in" in r.stderr


@pytest.mark.gpu
def test_streaming_segments_give_identical_csv(tmp_path):
    """Inputs are processed as a stream of segments (whole records each); tiny segments must not change a byte."""
    import gzip
    import shutil
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    gz = os.path.join(tmp, "reads.fq.gz")
    with open(os.path.join(gu.GOLDEN, "reads_k27.fq"), "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    cases = [(["-O", os.path.join(gu.GOLDEN, "reads_k27.fa")], "expected_k27_fa.csv"),
             (["-O", os.path.join(gu.GOLDEN, "reads_k27.fq")], "expected_k27_fq.csv"),
             (["-O", gz], "expected_k27_fq.csv"),
             (["-P", os.path.join(gu.GOLDEN, "pairs_k27_1.fq"), os.path.join(gu.GOLDEN, "pairs_k27_2.fq")], "expected_k27_pairs.csv")]
    for seg_kb in ("1", "3"):
        for src, exp in cases:
            out = os.path.join(tmp, f"s{seg_kb}_{exp}")
            r = _run([EXE_L, "-T", t, "-D", d, *src, "-R", out, "-n", "3", "-b", "4"], env=dict(os.environ, MIC_SEGMENT_KB=seg_kb))
            assert r.returncode == 0, r.stderr
            assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, exp), "rb").read(), (seg_kb, exp)
            assert "(131 objects)" in r.stdout or "(80 objects)" in r.stdout or "(40 objects)" in r.stdout


def _write_bgzf(src, dst, block=700):
    """Block-gzip (BGZF, what bgzip / samtools write): one gzip member per block with a 'BC' extra field holding its size."""
    import struct
    import zlib
    data = open(src, "rb").read()
    with open(dst, "wb") as fo:
        for off in list(range(0, len(data), block)) + [len(data)]:      # the last, empty block is the EOF marker
            raw = data[off:off + block] if off < len(data) else b""
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(raw) + co.flush()
            bsize = 18 + len(body) + 8
            fo.write(b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            fo.write(body + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)))


@pytest.mark.gpu
def test_block_gzip_and_multi_member_gzip_inputs(tmp_path):
    """Block-gzip input is inflated block-parallel, concatenated gzip members by zlib, both on a side thread; paired files
    are inflated concurrently.  The CSV must not change by a byte; a damaged block is reported, not classified."""
    import gzip
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    fq = os.path.join(gu.GOLDEN, "reads_k27.fq")
    bg = os.path.join(tmp, "reads.fq.bgz")
    _write_bgzf(fq, bg)
    import zlib
    assert zlib.decompressobj(31).decompress(open(bg, "rb").read())[:20] == open(fq, "rb").read()[:20]   # a valid gzip file
    mm = os.path.join(tmp, "reads_mm.fq.gz")
    data = open(fq, "rb").read()
    cut = data.index(b"\n@", len(data) // 2) + 1
    with open(mm, "wb") as fo:
        fo.write(gzip.compress(data[:cut]) + gzip.compress(data[cut:]))
    p1, p2 = os.path.join(tmp, "p1.fq.bgz"), os.path.join(tmp, "p2.fq.gz")
    _write_bgzf(os.path.join(gu.GOLDEN, "pairs_k27_1.fq"), p1, block=333)
    with open(os.path.join(gu.GOLDEN, "pairs_k27_2.fq"), "rb") as fi, gzip.open(p2, "wb") as fo:
        fo.write(fi.read())
    cases = [(["-O", bg], "expected_k27_fq.csv"), (["-O", mm], "expected_k27_fq.csv"), (["-P", p1, p2], "expected_k27_pairs.csv")]
    for threads in ("1", "5"):
        for src, exp in cases:
            out = os.path.join(tmp, f"t{threads}_{exp}")
            r = _run([EXE_L, "-T", t, "-D", d, *src, "-R", out, "-n", "2", "-b", "3"],
                     env=dict(os.environ, MIC_SEGMENT_KB="2", MIC_INFLATE_THREADS=threads))
            assert r.returncode == 0, r.stderr
            assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, exp), "rb").read(), (threads, exp)
    broken = bytearray(open(bg, "rb").read())
    broken[len(broken) // 2] ^= 0x5A
    bad = os.path.join(tmp, "broken.fq.bgz")
    open(bad, "wb").write(bytes(broken))
    r = _run([EXE_L, "-T", t, "-D", d, "-O", bad, "-R", os.path.join(tmp, "broken")])
    assert r.returncode != 0 and "uncompress" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_set_targets_then_classify_end_to_end(tmp_path):
    """The reference's user flow on a fresh directory: set_targets.sh <dir> custom (taxonomy already downloaded), then
    classify_metagenome.sh builds the database on first use and classifies; targets are species taxonomy IDs.  The
    expectation is the oracle's CSV on the reference-written golden database with the labels in targets.txt order."""
    import shutil
    import numpy as np
    import test_targets_tools as tt
    tmp = str(tmp_path)
    db = os.path.join(tmp, "DBD")
    tax = os.path.join(db, "taxonomy")
    os.makedirs(os.path.join(db, "Custom"))
    golden = gu.target_files_and_labels()
    names_golden = gu.target_names()
    species = {lab: 5000 + i for i, lab in enumerate(names_golden)}          # one species ID per golden label
    tt.make_taxonomy(tax)
    with open(os.path.join(tax, "nodes.dmp"), "a") as f:
        for s in species.values():
            f.write(f"{s}\t|\t561\t|\tspecies\t|\tXX\t|\n")
    with open(os.path.join(tax, "nucl_accss"), "a") as f:
        for i, (fn, lab) in enumerate(golden):
            f.write(f"rec{i}a\trec{i}a.1\t{species[lab]}\t{100 + i}\n")
            shutil.copy(fn, os.path.join(db, "Custom", os.path.basename(fn)))
    open(os.path.join(db, ".taxondata"), "w").close()
    r = _run([os.path.join(gu.ROOT, "set_targets.sh"), db, "custom"], cwd=tmp)
    assert r.returncode == 0, r.stdout + r.stderr
    order = []                                                               # taxonomy IDs in targets.txt order
    for line in open(os.path.join(db, "targets.txt")).read().splitlines():
        t = line.split("\t")[1]
        if t not in order:
            order.append(t)
    assert sorted(order) == sorted(str(s) for s in species.values())
    reads = os.path.join(gu.GOLDEN, "reads_k27.fa")
    r = _run([os.path.join(gu.ROOT, "classify_metagenome.sh"), "-O", reads, "-R", os.path.join(tmp, "out"), "-k", "27",
              "--htsize", "57777779"], cwd=tmp)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(db, "custom_0_canonical", f"db_central_k27_t{len(order)}_s57777779_m0.tsk.ky"))
    g = gu.load_golden_db("light_k27_u32")
    perm = np.array([order.index(str(species[lab])) for lab in names_golden], np.uint16)   # golden label -> new index
    odb = gu.oracle().db_from_arrays(gu.golden_sizes(g), g["ky"], perm[g["lb"]], 1)
    text, _ = odb.classify_file(27, open(reads, "rb").read(), order, False, False)
    assert open(os.path.join(tmp, "out.csv"), "rb").read() == text


@pytest.mark.gpu
@pytest.mark.parametrize("engines,parts", [(3, 3), (8, 8), (4, 2)])
def test_db_sharded_cli_matches_golden(tmp_path, engines, parts):
    """--db-sharded [--parts P]: the engines hold parts of the table (mic_db_set_part: the super-k-mer table cut by resident slot
    range, the per-run kernel's PART instantiation) and answer every batch of their read group together, rows summed read-range
    owned - the reference's multi-device mode (CuClarkDB.cu:886-1024).  `engines` engines share the one GPU here
    (MIC_SHARD_ENGINES); 4 engines / 2 parts = the 2-D layout (2 read groups).  CSVs identical to the whole-table ones: the
    streaming device path (plain, paired, FASTQ), the batch path (--extended), with sampling, and with 80 targets (rows that do
    not fit 15 entries: the batch falls back, dense counts summed over the parts)."""
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    env = dict(os.environ, MIC_SHARD_ENGINES=str(engines), MIC_CLI_TIMING="1")
    sh = ["--db-sharded", "--parts", str(parts)]
    for flag, src, exp in (([], ["-O", os.path.join(gu.GOLDEN, "reads_k27.fa")], "expected_k27_fa.csv"),
                           (["-n", "3"], ["-O", os.path.join(gu.GOLDEN, "reads_k27.fq")], "expected_k27_fq.csv"),
                           (["--extended", "-n", "2", "-b", "4"], ["-O", os.path.join(gu.GOLDEN, "reads_k27.fa")], "expected_k27_fa_ext.csv"),
                           (["-b", "3"], ["-P", os.path.join(gu.GOLDEN, "pairs_k27_1.fq"), os.path.join(gu.GOLDEN, "pairs_k27_2.fq")],
                            "expected_k27_pairs.csv")):
        out = os.path.join(tmp, "sh_" + exp)
        r = _run([EXE_L, "-T", t, "-D", d, *src, "-R", out, *sh, *flag], env=env)
        assert r.returncode == 0, r.stderr
        assert f"on {engines} device(s)" in r.stderr and f"table-sharded: {parts} part(s) x {engines // parts} read group(s)" in r.stderr, r.stderr
        assert "peer access: 1" in r.stderr
        # the per-run kernel's PART instantiation (k = 27, m = 20, one strand, slot-range part)
        assert "[timing] query kernel: query_kernel_r<27, 20, false, true>" in r.stderr, r.stderr
        assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, exp), "rb").read(), exp
        if "--extended" not in flag:      # the streaming path, nothing handed back to the host
            assert re.search(r"device ingest: \d+ batches .* 0 through the host path", r.stderr), r.stderr
    if (engines, parts) != (3, 3):
        return
    # sampling: whole table vs parts
    outs = []
    for extra, e in ((["-s", "3"], dict(os.environ)), (["-s", "3", *sh], env)):
        out = os.path.join(tmp, "samp%d" % len(outs))
        r = _run([EXE, "-k", "27", "--htsize", "57777779", "-T", t, "-D", _db_dir(os.path.join(tmp, "full"), "light_k27_u32", light=False),
                  "-O", os.path.join(gu.GOLDEN, "reads_k27.fa"), "-R", out, *extra], env=e)
        assert r.returncode == 0, r.stderr
        outs.append(open(out + ".csv", "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 132
    # the default number of parts: the smallest whose part fits a device - one for this table: the engines split the reads
    out = os.path.join(tmp, "auto")
    r = _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k27.fa"), "-R", out, "--db-sharded"], env=env)
    assert r.returncode == 0 and "table-sharded: 1 part(s) x 3 read group(s)" in r.stderr, r.stderr
    assert open(out + ".csv", "rb").read() == open(os.path.join(gu.GOLDEN, "expected_k27_fa.csv"), "rb").read()
    r = _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(gu.GOLDEN, "reads_k27.fa"), "-R", out, "--db-sharded", "--parts", "2"], env=env)
    assert r.returncode != 0 and "--parts 2 does not divide the 3 device(s)" in r.stderr
    # many targets: one read made of k-mers of 80 different targets
    import numpy as np
    rng = np.random.default_rng(3)
    k, T, htsize = 27, 80, 99991
    o = gu.oracle()
    seqs = ["".join(rng.choice(list("ACGT"), k + 4)) for _ in range(T)]
    canon = {}
    for lab, s_ in enumerate(seqs):
        for i in range(len(s_) - k + 1):
            canon[o.canonical(int("".join(str("TGCA".index(c)) for c in s_[i:i + k]), 4), k)] = lab
    items = sorted(canon.items(), key=lambda kv: (kv[0] % htsize, kv[0] // htsize))
    sizes = np.zeros(htsize, np.uint8)
    for c, _ in items:
        sizes[c % htsize] += 1
    dd = os.path.join(tmp, "DB80")
    os.makedirs(dd)
    base = os.path.join(dd, f"db_central_k{k}_t{T}_s{htsize}_m0.tsk")
    sizes.tofile(base + ".sz")
    from cuclark_amd import host
    np.array([c // htsize for c, _ in items], dtype=gu.KEY_DTYPE[host.key_bytes_rule(htsize, k)]).tofile(base + ".ky")
    np.array([l for _, l in items], np.uint16).tofile(base + ".lb")
    tt = os.path.join(tmp, "t80.txt")
    with open(tt, "w") as f:
        for lab in range(T):
            f.write(f"{os.path.join(gu.GOLDEN, 'targets', 'genome_0.fa')} L{lab:02d}\n")
    reads = os.path.join(tmp, "r80.fa")
    with open(reads, "w") as f:
        f.write(">all\n" + "N".join(seqs) + "\n>few\n" + "N".join(seqs[:3]) + "\n")
    res = []
    for extra, e in (([], dict(os.environ)), (sh, env)):
        for ext in ([], ["--extended"]):
            out = os.path.join(tmp, "t80_%d" % len(res))
            r = _run([EXE, "-k", str(k), "--htsize", str(htsize), "-T", tt, "-D", dd, "-O", reads, "-R", out, *extra, *ext], env=e)
            assert r.returncode == 0, r.stderr
            res.append(open(out + ".csv", "rb").read())
            if extra and not ext:      # the summed row of "all" does not fit: the batch went through the host path, exactly
                assert re.search(r"device ingest: 1 batches .* 1 through the host path", r.stderr), r.stderr
    assert res[0] == res[2] and res[1] == res[3]
    assert res[0].splitlines()[1].startswith(b"all,") and b",L00,5," in res[0].splitlines()[1]


@pytest.fixture(scope="module")
def multi_engine_rig(tmp_path_factory):
    """Inputs of every form the command line takes, and the ONE-engine run's CSV of each (the reference of the multi-engine runs)."""
    import gzip
    import numpy as np
    import test_ingest as ti
    tmp = str(tmp_path_factory.mktemp("multi_engine"))
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    rng = np.random.default_rng(41)
    genomes = ti._genomes()
    fq = os.path.join(tmp, "r.fq")
    fa = os.path.join(tmp, "r.fa")
    open(fq, "wb").write(ti._random_reads(rng, genomes, 5000, fasta=False))
    open(fa, "wb").write(ti._random_reads(rng, genomes, 1500, fasta=True))
    m1, m2 = _pair_files(rng, genomes, 3000)
    p1, p2 = os.path.join(tmp, "m_1.fq"), os.path.join(tmp, "m_2.fq")
    open(p1, "wb").write(m1)
    open(p2, "wb").write(m2)
    gz, bg = os.path.join(tmp, "r.fq.gz"), os.path.join(tmp, "r.fq.bgz")
    open(gz, "wb").write(gzip.compress(open(fq, "rb").read(), 1))
    _write_bgzf(fq, bg, block=60000)
    g1, g2 = p1 + ".gz", p2 + ".gz"
    open(g1, "wb").write(gzip.compress(m1, 1))
    open(g2, "wb").write(gzip.compress(m2, 6))
    lo = os.path.join(tmp, "objs.txt")
    open(lo, "w").write(fa + "\n" + fq + "\n")
    cases = {"fq": ["-O", fq], "fa": ["-O", fa], "ext": ["-O", fa, "--extended", "-b", "7"], "pairs": ["-P", p1, p2], "gz": ["-O", gz],
             "bgzf": ["-O", bg], "gzpairs": ["-P", g1, g2], "list": ["-O", lo]}
    rig = dict(tmp=tmp, d=d, t=t, cases=cases, lr=os.path.join(tmp, "ress.txt"))

    def run(name, tag, env):
        out = os.path.join(tmp, f"{tag}_{name}")
        res = ["-R", out]
        if name == "list":
            lr = rig["lr"] + "." + tag
            open(lr, "w").write(out + "_a\n" + out + "_b\n")
            res = ["-R", lr]
        r = _run([EXE_L, "-T", t, "-D", d, *cases[name], *res, "-n", "5"], env=env)
        assert r.returncode == 0, (name, r.stderr)
        return r, [open(out + sfx + ".csv", "rb").read() for sfx in (("_a", "_b") if name == "list" else ("",))]
    rig["run"] = run
    one = dict(os.environ, MIC_CLI_TIMING="1", MIC_INGEST_KB="24")
    rig["one"] = dict(zip(cases, (x[1] for x in _run_many([lambda name=name: run(name, "one", one) for name in cases]))))
    return rig


@pytest.mark.gpu
@pytest.mark.parametrize("engines", [2, 3])
def test_read_sharded_cli_with_several_engines_on_one_gpu(multi_engine_rig, engines):
    """-d N, the default multi-device mode: the table replicated on every engine, the ingest slots - and with them the batches -
    dealt over the engines.  MIC_SHARD_ENGINES puts N engines on this box's one GPU.  Every input form gives the one-engine
    run's CSV byte for byte (every record once, in order): plain FASTA / FASTQ over many small batches, --extended (batch API),
    two plain mates, gzip and block gzip (inflated on the first engine's device, slots of the other engines filled from there),
    compressed mates, list-of-files."""
    rig = multi_engine_rig
    multi = dict(os.environ, MIC_SHARD_ENGINES=str(engines), MIC_CLI_TIMING="1", MIC_INGEST_KB="24")
    names = list(rig["cases"])
    for name, (r, got) in zip(names, _run_many([lambda name=name: rig["run"](name, f"multi{engines}", multi) for name in names], workers=4)):
        assert f"{engines} engine(s) on 1 device(s), read-sharded (table replicated)" in r.stderr, r.stderr
        assert f"on {engines} device(s)" in r.stderr
        if name in ("gz", "bgzf", "gzpairs"):      # inflated on the device although the slots sit on several engines
            assert re.search(r"device inflate: [\d.]+ MB of text", r.stderr) and "over the link 0 MB" in r.stderr, r.stderr
        if name in ("gz", "bgzf"):
            # 24 KB slots are smaller than the 64 whole FASTQ records between two sampled offsets of the resident text: most batches
            # are handed back to the host indexer / packer / CSV writer while the stream's threads run - that path, exercised
            assert re.search(r", [1-9]\d* through the host path", r.stderr), r.stderr
        if name != "ext":
            m = re.search(r"device ingest: (\d+) batches of <= 24 KB on (\d+) slot", r.stderr)
            assert m and int(m.group(1)) > 2 * engines and int(m.group(2)) >= engines, r.stderr
        assert got == rig["one"][name], name
        assert all(len(x) > 1000 for x in got)


@pytest.mark.gpu
def test_device_ingest_equals_host_ingest_over_many_batches(tmp_path):
    """The CLI's default path hands batches of raw bytes to the GPU (mic_ingest_*); MIC_HOST_INGEST=1 keeps the host
    indexer / packer / CSV writer.  Same bytes out for FASTA, FASTQ, gzip, CRLF files, tiny and large batches, several
    workers, and for input with records the device path hands back (empty names, a 70 kb sequence)."""
    import gzip
    import numpy as np
    import test_ingest as ti
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    rng = np.random.default_rng(11)
    genomes = ti._genomes()
    odd = b">\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>long\n" + genomes[0][:3000] * 24 + b"\n"
    files = {
        "a.fq": ti._random_reads(rng, genomes, 6000, fasta=False),
        "b.fa": ti._random_reads(rng, genomes, 6000, fasta=True),
        "c.fa": ti._random_reads(rng, genomes, 2000, fasta=True, crlf=True),
        "d.fa": ti._random_reads(rng, genomes, 1500, fasta=True) + odd + ti._random_reads(rng, genomes, 1500, fasta=True),
        "e.fq": ti._random_reads(rng, genomes, 3000, fasta=False)[:-1],
    }
    for name, data in files.items():
        open(os.path.join(tmp, name), "wb").write(data)
    with gzip.open(os.path.join(tmp, "a.fq.gz"), "wb") as f:
        f.write(files["a.fq"])
    names = list(files) + ["a.fq.gz"]

    def host_run(name):
        r = _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(tmp, name), "-R", os.path.join(tmp, "host_" + name), "-n", "4"], env=dict(os.environ, MIC_HOST_INGEST="1"))
        assert r.returncode == 0, r.stderr
        return re.search(r"\((\d+) objects\)", r.stdout).group(1)

    def dev_run(name, kb, n):
        env = dict(os.environ, MIC_CLI_TIMING="1")
        if kb != "0":
            env["MIC_INGEST_KB"] = kb
        return _run([EXE_L, "-T", t, "-D", d, "-O", os.path.join(tmp, name), "-R", os.path.join(tmp, f"dev{kb}_{name}"), "-n", n], env=env)
    n_obj = dict(zip(names, _run_many([lambda name=name: host_run(name) for name in names])))
    combos = [(name, kb, n) for name in names for kb, n in (("16", "4"), ("300", "3"), ("0", "1"))]
    for (name, kb, n), r in zip(combos, _run_many([lambda c=c: dev_run(*c) for c in combos])):
        assert r.returncode == 0, r.stderr
        assert f"({n_obj[name]} objects)" in r.stdout
        assert "device ingest:" in r.stderr
        if name == "d.fa":
            assert re.search(r", [1-9]\d* through the host path", r.stderr), r.stderr
        assert open(os.path.join(tmp, f"dev{kb}_{name}.csv"), "rb").read() == open(os.path.join(tmp, "host_" + name + ".csv"), "rb").read(), (name, kb)


def _pair_files(rng, genomes, n, crlf=False):
    """two FASTQ files of n pairs: ids with /1 /2 suffixes, descriptions, reads shorter than k, N and lower case"""
    import numpy as np
    a, b = [], []
    eol = b"\r\n" if crlf else b"\n"
    for i in range(n):
        recs = []
        for _ in range(2):
            g = genomes[int(rng.integers(len(genomes)))]
            L = int(rng.choice([0, 5, 26, 27, 31, 40, 64, 100, 101, 150, 151, 250]))
            p = int(rng.integers(0, len(g) - L))
            s = bytearray(g[p:p + L]) if rng.random() < 0.8 else bytearray(rng.choice(list(b"ACGT"), L).astype(np.uint8).tobytes())
            if len(s) and rng.random() < 0.3:
                s[int(rng.integers(len(s)))] = ord("N")
            if rng.random() < 0.2:
                s = bytearray(bytes(s).lower())
            recs.append(bytes(s))
        style = int(rng.integers(5))
        ids = [(b"p%d/1" % i, b"p%d/2" % i), (b"pair_%d 1:N:0" % i, b"pair_%d 2:N:0" % i), (b"q%d\tfirst" % i, b"q%d\tsecond" % i),
               (b"y" * 44 + b"%d/1" % i, b"y" * 44 + b"%d/2" % i), (b"@z%d" % i, b"z%d" % i)][style]
        plus = b"+" + (ids[0] if rng.random() < 0.2 else b"")
        a.append(b"@" + ids[0] + eol + recs[0] + eol + plus + eol + b"@" * len(recs[0]) + eol)    # quality lines full of '@'
        b.append(b"@" + ids[1] + eol + recs[1] + eol + b"+" + eol + b"+" * len(recs[1]) + eol)
    return b"".join(a), b"".join(b)


@pytest.mark.gpu
def test_paired_files_merged_by_the_loaders_equal_the_serial_reader(tmp_path):
    """-P with two plain FASTQ files: the loaders cut both files at the same record numbers (line counts) and merge their
    batches in parallel; MIC_SERIAL_PAIRS=1 keeps the line-by-line reader of file.cc:205-268.  Same CSV over tiny and
    large batches, CRLF files and a last line without its line end; files the line arithmetic does not cover (a blank line
    in front of a record, unequal record counts, ids that differ) end exactly as the serial reader ends them."""
    import numpy as np
    import test_ingest as ti
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    rng = np.random.default_rng(23)
    genomes = ti._genomes()
    cases = {"plain": _pair_files(rng, genomes, 5000), "crlf": _pair_files(rng, genomes, 800, crlf=True)}
    a, b = _pair_files(rng, genomes, 700)
    cases["no_last_eol"] = (a[:-1], b[:-1])
    a, b = _pair_files(rng, genomes, 900)
    recs = a.split(b"\n")
    cases["blank_line"] = (b"\n".join(recs[:400] + [b""] + recs[400:]), b"\n".join(b.split(b"\n")[:400] + [b""] + b.split(b"\n")[400:]))
    a, b = _pair_files(rng, genomes, 600)
    cases["unequal"] = (a, b"\n".join(b.split(b"\n")[:4 * 450]) + b"\n")
    for name, (f1, f2) in cases.items():
        p1, p2 = os.path.join(tmp, name + "_1.fq"), os.path.join(tmp, name + "_2.fq")
        open(p1, "wb").write(f1)
        open(p2, "wb").write(f2)
        ref = os.path.join(tmp, "serial_" + name)
        r0 = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", ref, "-n", "3"], env=dict(os.environ, MIC_SERIAL_PAIRS="1"))
        assert r0.returncode == 0, r0.stderr
        n_obj = re.search(r"\((\d+) objects\)", r0.stdout).group(1)
        for kb, n in (("16", "4"), ("200", "6"), ("0", "2")):
            out = os.path.join(tmp, f"par{kb}_{name}")
            env = dict(os.environ, MIC_CLI_TIMING="1")
            if kb != "0":
                env["MIC_INGEST_KB"] = kb
            r = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", out, "-n", n], env=env)
            assert r.returncode == 0, r.stderr
            assert f"({n_obj} objects)" in r.stdout and r.stdout.count("Assignment time") == 1
            assert open(out + ".csv", "rb").read() == open(ref + ".csv", "rb").read(), (name, kb)
            if name in ("plain", "crlf", "no_last_eol") and kb == "16":
                assert int(re.search(r"device ingest: (\d+) batches", r.stderr).group(1)) > 8, r.stderr
    # ids that differ: both readers stop with the reference's message
    a, b = _pair_files(rng, genomes, 300)
    p1, p2 = os.path.join(tmp, "bad_1.fq"), os.path.join(tmp, "bad_2.fq")
    open(p1, "wb").write(a)
    open(p2, "wb").write(b.replace(b"@p7/2", b"@p8/2").replace(b"@pair_7 ", b"@pair_8 ").replace(b"@q7\t", b"@q8\t").replace(b"@z7\n", b"@z8\n"))
    if open(p2, "rb").read() != b:
        for env in (dict(os.environ, MIC_SERIAL_PAIRS="1"), dict(os.environ, MIC_INGEST_KB="16")):
            r = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", os.path.join(tmp, "bad"), "-n", "3"], env=env)
            assert r.returncode != 0 and "read id does not match between files" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_compressed_mates_inflated_and_merged_on_the_device_equal_the_host_path(tmp_path):
    """-P a.fq.gz b.fq.gz: both files are inflated on the device (mic_gz_inflate_device), paired up and merged there (mic_pairs_*),
    the batches never cross the link as text; MIC_GZ_HOST=1 keeps the host inflater and the loaders' merge.  Same CSV over tiny and
    large batches, CRLF and a last line without its end; files the device path does not take (two gzip members, mates with unequal
    record counts or ids that differ) go through the host path and end as they did before."""
    import gzip
    import numpy as np
    import test_ingest as ti
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    rng = np.random.default_rng(29)
    genomes = ti._genomes()
    cases = {"plain": _pair_files(rng, genomes, 6000), "crlf": _pair_files(rng, genomes, 700, crlf=True)}
    a, b = _pair_files(rng, genomes, 500)
    last = genomes[0][1000:1100]             # (a last record with an empty quality line would lose a whole line with its line end)
    cases["no_last_eol"] = (a + b"@last/1\n" + last + b"\n+\n" + b"I" * 100, b + b"@last/2\n" + last[::-1] + b"\n+\n" + b"I" * 100)
    a, b = _pair_files(rng, genomes, 600)
    cases["unequal"] = (a, b"\n".join(b.split(b"\n")[:4 * 450]) + b"\n")
    a, b = _pair_files(rng, genomes, 900)
    cases["two_members"] = (a, b)
    for name, (f1, f2) in cases.items():
        p1, p2 = os.path.join(tmp, name + "_1.fq.gz"), os.path.join(tmp, name + "_2.fq.gz")
        open(p1, "wb").write(gzip.compress(f1, 1))
        open(p2, "wb").write(gzip.compress(f2, 6) if name != "two_members" else gzip.compress(f2[:len(f2) // 2]) + gzip.compress(f2[len(f2) // 2:]))
        ref = os.path.join(tmp, "host_" + name)
        r0 = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", ref, "-n", "4"], env=dict(os.environ, MIC_GZ_HOST="1", MIC_CLI_TIMING="1"))
        assert r0.returncode == 0 and "device inflate" not in r0.stderr, r0.stderr
        n_obj = re.search(r"\((\d+) objects\)", r0.stdout).group(1)
        def dev_pair(kb, n, name=name, p1=p1, p2=p2):
            env = dict(os.environ, MIC_CLI_TIMING="1")
            if kb != "0":
                env["MIC_INGEST_KB"] = kb
            return _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", os.path.join(tmp, f"dev{kb}_{name}"), "-n", n], env=env)
        combos = (("16", "4"), ("300", "6"), ("0", "3"))
        for (kb, n), r in zip(combos, _run_many([lambda c=c: dev_pair(*c) for c in combos], workers=3)):
            out = os.path.join(tmp, f"dev{kb}_{name}")
            assert r.returncode == 0, r.stderr
            assert f"({n_obj} objects)" in r.stdout and r.stdout.count("Assignment time") == 1
            assert open(out + ".csv", "rb").read() == open(ref + ".csv", "rb").read(), (name, kb)
            if name in ("plain", "crlf", "no_last_eol"):
                assert re.search(r"device inflate: [\d.]+ MB of text", r.stderr), r.stderr
                assert re.search(r"over the link 0 MB", r.stderr), r.stderr            # no text went up
                if kb == "16" and name == "plain":
                    assert int(re.search(r"device ingest: (\d+) batches", r.stderr).group(1)) > 8, r.stderr
            else:
                assert "device inflate: not used" in r.stderr, r.stderr
    # one compressed file, FASTQ or FASTA (sequences over several lines): the same, records copied instead of merged
    fq = ti._random_reads(rng, genomes, 4000, fasta=False)
    fa = ti._random_reads(rng, genomes, 500, fasta=True)
    for name, data in (("single.fq.gz", fq), ("single.fa.gz", fa)):
        pth = os.path.join(tmp, name)
        open(pth, "wb").write(gzip.compress(data, 1))
        ref = os.path.join(tmp, "host_" + name)
        r0 = _run([EXE_L, "-T", t, "-D", d, "-O", pth, "-R", ref, "-n", "4"], env=dict(os.environ, MIC_GZ_HOST="1", MIC_CLI_TIMING="1"))
        assert r0.returncode == 0 and "device inflate" not in r0.stderr, r0.stderr
        for kb in ("16", "0"):
            out = os.path.join(tmp, f"dev{kb}_{name}")
            env = dict(os.environ, MIC_CLI_TIMING="1")
            if kb != "0":
                env["MIC_INGEST_KB"] = kb
            r = _run([EXE_L, "-T", t, "-D", d, "-O", pth, "-R", out, "-n", "4"], env=env)
            assert r.returncode == 0, r.stderr
            assert open(out + ".csv", "rb").read() == open(ref + ".csv", "rb").read(), (name, kb)
            n_rec = 4000 if name.endswith(".fq.gz") else 500
            assert re.search(r"device inflate: [\d.]+ MB of text .* %d records indexed" % n_rec, r.stderr), r.stderr
            assert re.search(r"over the link 0 MB", r.stderr), r.stderr
    # ids that differ: refused by the device's check, then the host readers stop with the reference's message
    a, b = _pair_files(rng, genomes, 300)
    b2 = b.replace(b"@p7/2", b"@p8/2").replace(b"@pair_7 ", b"@pair_8 ").replace(b"@q7\t", b"@q8\t").replace(b"@z7\n", b"@z8\n")
    if b2 != b:
        p1, p2 = os.path.join(tmp, "bad_1.fq.gz"), os.path.join(tmp, "bad_2.fq.gz")
        open(p1, "wb").write(gzip.compress(a))
        open(p2, "wb").write(gzip.compress(b2))
        r = _run([EXE_L, "-T", t, "-D", d, "-P", p1, p2, "-R", os.path.join(tmp, "bad"), "-n", "3"], env=dict(os.environ, MIC_CLI_TIMING="1"))
        assert r.returncode != 0 and "read id does not match between files" in (r.stderr + r.stdout)
        assert "device inflate: not used" in r.stderr


@pytest.mark.gpu
def test_one_gzip_member_in_stripes_and_a_file_given_back_midway(tmp_path):
    """MIC_GZ_STRIPES (mic_gz_stream_*, DeviceGzFeeder): the member inflated a stripe of deflate blocks at a time on a thread of its
    own, its FASTQ records handed to the ingest slots as they become final.  Same CSV as the plain file.  Two members in one file: the
    first stripes go through, the stripe that meets the end of the first member gives the file back, the run starts over on the CPU
    inflater - same CSV again, and a note on stderr."""
    import gzip
    tmp = str(tmp_path)
    d = _db_dir(tmp, "light_k27_u32", light=True)
    t = _targets_file(tmp)
    one = open(os.path.join(gu.GOLDEN, "reads_k27.fq"), "rb").read()
    data = one * 700                                                # ~13 MB of text: ~50 deflate blocks at level 1
    plain = os.path.join(tmp, "many.fq")
    open(plain, "wb").write(data)
    gz1 = os.path.join(tmp, "many.fq.gz")
    open(gz1, "wb").write(gzip.compress(data, 1))
    cut = data.index(b"\n@fq0/1", len(data) // 2) + 1
    gz2 = os.path.join(tmp, "many_mm.fq.gz")
    open(gz2, "wb").write(gzip.compress(data[:cut], 1) + gzip.compress(data[cut:], 1))
    r = _run([EXE_L, "-T", t, "-D", d, "-O", plain, "-R", os.path.join(tmp, "plain"), "-n", "4"])
    assert r.returncode == 0, r.stderr
    expect = open(os.path.join(tmp, "plain.csv"), "rb").read()
    assert expect.count(b"\n") == 1 + 700 * (open(os.path.join(gu.GOLDEN, "expected_k27_fq.csv"), "rb").read().count(b"\n") - 1)
    env = dict(os.environ, MIC_GZ_STRIPES="3", MIC_GZ_STRIPE_UNITS="16", MIC_CLI_TIMING="1")
    r = _run([EXE_L, "-T", t, "-D", d, "-O", gz1, "-R", os.path.join(tmp, "s1"), "-n", "4"], env=env)
    assert r.returncode == 0, r.stderr
    assert "device inflate in stripes: all" in r.stderr and "gave the file back" not in r.stderr, r.stderr
    assert open(os.path.join(tmp, "s1.csv"), "rb").read() == expect
    r = _run([EXE_L, "-T", t, "-D", d, "-O", gz2, "-R", os.path.join(tmp, "s2"), "-n", "4"], env=env)
    assert r.returncode == 0, r.stderr
    assert "device inflate in stripes: the first" in r.stderr and "gave the file back" in r.stderr, r.stderr
    assert open(os.path.join(tmp, "s2.csv"), "rb").read() == expect
    # damage in a later stripe: reported the reference's way by the inflater the run falls back to, nothing classified
    broken = bytearray(open(gz1, "rb").read())
    broken[len(broken) * 3 // 4] ^= 0x5A
    bad = os.path.join(tmp, "broken.fq.gz")
    open(bad, "wb").write(bytes(broken))
    r = _run([EXE_L, "-T", t, "-D", d, "-O", bad, "-R", os.path.join(tmp, "s3"), "-n", "4"], env=env)
    assert r.returncode != 0 and "uncompress" in (r.stderr + r.stdout)
