"""integration/CuClarkDB.cuh replaces the reference's CuClarkDB class.  The driver (built in the
build container against the reference's own dataType.hh / parameters.hh) makes the calls CuCLARK_hh.hh makes."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

DRIVER = os.path.join(gu.ROOT, "integration", "_bin", "shim_driver")


def test_shim_matches_the_documented_text():
    """The header shipped in integration/ is the code block of INTEGRATION.md."""
    md = open(os.path.join(gu.ROOT, "INTEGRATION.md")).read()
    hdr = open(os.path.join(gu.ROOT, "integration", "CuClarkDB.cuh")).read()
    block = md[md.index("template <typename HKMERr> class CuClarkDB {"):md.index("```\n\nNotes for that route")]
    assert block.strip() in hdr


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference headers")
def test_shim_compiles_against_reference_types(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(gu.ROOT, "integration")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.access(DRIVER, os.X_OK)


@pytest.mark.gpu
@pytest.mark.skipif(not os.access(DRIVER, os.X_OK), reason="integration/_bin/shim_driver not built")
def test_shim_drives_the_engine_like_cuclark(tmp_path):
    from cuclark_amd import host
    prefix, meta = gu.materialize_db("light_k27_u32", str(tmp_path))
    data = open(os.path.join(gu.GOLDEN, "reads_k27.fa"), "rb").read()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], 27)
    rp.tofile(str(tmp_path / "rp.bin"))
    cont.tofile(str(tmp_path / "ct.bin"))
    r = subprocess.run([DRIVER, prefix, "27", "6", str(tmp_path / "rp.bin"), str(tmp_path / "ct.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(x) for x in l.split()] for l in r.stdout.strip().splitlines()], dtype=np.uint32)
    odb, _ = gu.oracle_db_from_golden("light_k27_u32")
    counts, _ = odb.query_batch(27, rp, cont, 6)
    assert (got == gu.oracle().result_from_counts(counts)).all()
    # numDevices > 1: the reference's table-sharded mode through the same class (three engines share the one GPU here),
    # with the sparse rows CuCLARK_hh.hh:2014-2031 reads in --extended mode
    r = subprocess.run([DRIVER, prefix, "27", "6", str(tmp_path / "rp.bin"), str(tmp_path / "ct.bin"), "3", "1"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MIC_SHARD_ENGINES="3"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    n = got.shape[0]
    got3 = np.array([[int(x) for x in l.split()] for l in lines[:n]], dtype=np.uint32)
    assert (got3 == got).all()
    for i, l in enumerate(lines[n:]):
        f = l.split()
        assert f[0] == "row" and int(f[1]) == int(np.count_nonzero(counts[i]))
        assert [tuple(int(x) for x in p.split(":")) for p in f[2:]] == [(t, int(counts[i][t])) for t in np.nonzero(counts[i])[0]]


@pytest.mark.gpu
@pytest.mark.skipif(not os.access(DRIVER, os.X_OK), reason="integration/_bin/shim_driver not built")
def test_shim_completes_a_read_of_80_targets_exactly_on_three_engines(tmp_path):
    """a read whose row does not fit (80 targets, MAXHITS = 15) through the shim on three engines (numDevices = 3:
    parts of the table, ONE upload fanned out by mic_batch_query_group, rows summed by mic_batch_merge_shards).  Sum / best /
    second-best are exact; the extended row holds the first (rowSize - 2) / 2 pairs in ascending target order - all 80 when the caller
    allocates rows of 2 T + 2, and then every count equals the command line's own --extended CSV."""
    import re
    from cuclark_amd import host
    rng = np.random.default_rng(3)
    k, T, htsize = 27, 80, 57777779           # (cuCLARK-l's table size: u32 keys for k = 27, the driver's CuClarkDB<uint32_t>)
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
    tmp = str(tmp_path)
    dd = os.path.join(tmp, "DB80")
    os.makedirs(dd)
    base = os.path.join(dd, f"db_central_k{k}_t{T}_s{htsize}_m0.tsk")
    sizes.tofile(base + ".sz")
    kb = host.key_bytes_rule(htsize, k)
    assert kb == 4          # the driver's CuClarkDB<uint32_t>
    np.array([c // htsize for c, _ in items], dtype=gu.KEY_DTYPE[kb]).tofile(base + ".ky")
    np.array([l for _, l in items], np.uint16).tofile(base + ".lb")
    data = (">all\n" + "N".join(seqs) + "\n>few\n" + "N".join(seqs[:3]) + "\n>none\n" + "ACGT" * 10 + "\n").encode()
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    rp.tofile(os.path.join(tmp, "rp.bin"))
    cont.tofile(os.path.join(tmp, "ct.bin"))
    odb = o.db_from_arrays(sizes, np.array([c // htsize for c, _ in items], dtype=gu.KEY_DTYPE[kb]), np.array([l for _, l in items], np.uint16))
    counts, _ = odb.query_batch(k, rp, cont, T)
    expect = o.result_from_counts(counts)
    assert np.count_nonzero(counts[0]) == T and np.count_nonzero(counts[1]) == 3

    def drive(row_size):
        r = subprocess.run([DRIVER, base, str(k), str(T), os.path.join(tmp, "rp.bin"), os.path.join(tmp, "ct.bin"), "3", "1", *( [str(row_size)] if row_size else [])],
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, MIC_SHARD_ENGINES="3"))
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        res = np.array([[int(x) for x in l.split()] for l in lines[:3]], dtype=np.uint32)
        rows = [[tuple(int(x) for x in p_.split(":")) for p_ in l.split()[2:]] for l in lines[3:]]
        ns = [int(l.split()[1]) for l in lines[3:]]
        return res, rows, ns
    # the reference's row size (2 * MAXHITS + 2 = 32): exact results, the first 15 pairs ascending, n = 15
    res, rows, ns = drive(0)
    assert (res == expect).all(), (res, expect)
    assert ns == [15, 3, 0] and rows[0] == [(t, int(counts[0][t])) for t in range(15)] and rows[1] == [(t, int(counts[1][t])) for t in range(3)]
    # rows of 2 T + 2: every pair, equal to the command line's --extended columns on the same database and reads
    res, rows, ns = drive(2 * T + 2)
    assert (res == expect).all() and ns == [T, 3, 0]
    assert rows[0] == [(t, int(counts[0][t])) for t in range(T)]
    tt = os.path.join(tmp, "t80.txt")
    with open(tt, "w") as f:
        for lab in range(T):
            f.write(f"{os.path.join(gu.GOLDEN, 'targets', 'genome_0.fa')} L{lab:02d}\n")
    reads = os.path.join(tmp, "r80.fa")
    open(reads, "wb").write(data)
    out = os.path.join(tmp, "cli")
    r = subprocess.run([os.path.join(gu.ROOT, "exe", "cuCLARK"), "-k", str(k), "--htsize", str(htsize), "-T", tt, "-D", dd, "-O", reads, "-R", out,
                        "--extended", "--db-sharded", "--parts", "3"], capture_output=True, text=True, timeout=300, env=dict(os.environ, MIC_SHARD_ENGINES="3"))
    assert r.returncode == 0, r.stderr
    csv = open(out + ".csv").read().splitlines()
    assert csv[0].split(",")[1:T + 1] == [f"L{lab:02d}" for lab in range(T)]
    for i in range(3):
        got = [int(x) for x in csv[1 + i].split(",")[1:T + 1]]
        shim = dict(rows[i])
        assert got == [shim.get(t, 0) for t in range(T)] == [int(c) for c in counts[i]], i
