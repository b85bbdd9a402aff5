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
