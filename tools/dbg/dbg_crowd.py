import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import golden_util as gu
from test_crowded import _microsatellite_db, _reads
from test_gpu_parity import _oracle_results
from cuclark_amd import MiClarkDB, host
os.environ["MIC_LAYOUT"] = "super"
rng = np.random.default_rng(43)
k, T, htsize = 31, 12, 1 << 18
seqs, sizes, keys, labels = _microsatellite_db(rng, k, htsize, T)
odb = gu.oracle().db_from_arrays(sizes, keys, labels)
dev = torch.device("cuda:0")
data = _reads(rng, seqs, 4000)
idx = host.index_reads(data)
rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
n = rp.size - 1
counts, expect = _oracle_results(odb, k, rp, cont, T)
with MiClarkDB(k, T, row_words=16) as e:
    e.read_arrays(sizes, keys, labels)
    print(e.info())
    d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
    d_ct = torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint16)]).view(np.int16)).to(dev)
    d_res = torch.full((n, 8), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    e.query_device(d_rp.data_ptr(), d_ct.data_ptr(), n, d_res.data_ptr())
    print("ms", e.last_query_ms(), e.last_crowd_stats())
    res = d_res.cpu().numpy().view(np.uint32)
    unwritten = (res[:, 0] == 0x5A5A5A5A)
    fl = res[:, 6]
    print("unwritten", unwritten.sum(), "flag1", ((fl & 1) != 0).sum(), "n_ent hist", np.bincount(res[~unwritten, 5].clip(0, 70))[:20])
    ok = (res[:, :5] == expect).all(axis=1)
    print("equal before resolve", ok.sum(), "of", n, "; flagged&equal", (ok & ((fl & 1) != 0)).sum())
    flagged_ids = np.nonzero((fl & 1) != 0)[0]
    print("first flagged", flagged_ids[:10], res[flagged_ids[:5]], expect[flagged_ids[:5]])
    lens = np.diff(rp.astype(np.int64))
    print("flagged read container counts", np.bincount(lens[flagged_ids])[:30])
    print("all read container counts", np.bincount(lens)[:30])
    dense = e.resolve_flagged_device(d_rp.data_ptr(), d_ct.data_ptr(), d_res.data_ptr())
    print("dense", dense)
