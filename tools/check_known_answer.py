"""Constructive known answer of the synthetic workload at any read length, and GPU vs oracle on the reads that miss it:
    python tools/check_known_answer.py [read_len] [workload]"""
import os, sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from cuclark_amd import _lib, MiClarkDB
import golden_util as gu
L = _lib.load(); dev = torch.device("cuda:0")
import bench
w = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "light"]
T, k, htsize, nt, KB = w["n_targets"], w["k"], w["htsize"], int(w["genome_nt"]), w["key_bytes"]
spec = _lib.MicSynthSpec(seed=4, htsize=htsize, genome_nt=nt, n_targets=T, n_genomes=w["n_genomes"], k=k, key_bytes=KB)
cap = nt + 1024
d_sizes = torch.empty(htsize, dtype=torch.uint8, device=dev); d_keys = torch.empty(cap, dtype=torch.int32 if KB == 4 else torch.int64, device=dev); d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
n_el = C.c_uint64(0); torch.cuda.synchronize()
assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
n_el = n_el.value
n_reads, read_len = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 250
pitch = L.mic_synth_read_pitch(read_len, k)
d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev); d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev); d_truth = torch.zeros(n_reads * 2, dtype=torch.int32, device=dev)
assert L.mic_synth_reads_device(C.byref(spec), 5, n_reads, read_len, 0.2, 0.01, 0.001, d_rp.data_ptr(), d_cont.data_ptr(), d_cont.numel(), d_truth.data_ptr(), None) == 0
torch.cuda.synchronize()
with MiClarkDB(k, T) as e:
    e.read_device(d_sizes.data_ptr(), htsize, d_keys.data_ptr(), KB, d_labels.data_ptr())
    res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev); torch.cuda.synchronize()
    e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, res.data_ptr())
    print("flagged", e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), res.data_ptr())); e.sync()
r = res.cpu().numpy().view(np.uint32); truth = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
g = truth[:, 0] > 0
ok = (truth[:, 1] == 0) | ((r[:, 1] == truth[:, 0]) & (r[:, 2] >= truth[:, 1]))
bad = np.flatnonzero(g & ~ok)
print("genome reads", int(g.sum()), "failing", bad.size, bad[:10])
if bad.size:
    sizes = d_sizes.cpu().numpy(); keys = d_keys[:n_el].cpu().numpy().view(np.uint32 if KB == 4 else np.uint64); labels = d_labels[:n_el].cpu().numpy().view(np.uint16)
    from oracle.binding import Oracle
    odb = Oracle().db_wrap_arrays(sizes, keys, labels)
    rp = d_rp.cpu().numpy().view(np.uint32); cont = d_cont.cpu().numpy().view(np.uint16)
    for b in bad[:10]:
        lo, hi = int(rp[b]), int(rp[b + 1])
        sub_rp = np.array([0, hi - lo], np.uint32); sub_ct = np.concatenate([cont[lo:hi], np.zeros(64, np.uint16)])
        exp = odb.classify_batch(k, sub_rp, sub_ct, T, threads=1)
        print("read", b, "truth", truth[b], "gpu", r[b, :6], "oracle", exp[0], "first part", int(cont[lo]))
