#!/usr/bin/env python3
"""The query kernel on the generator's reads (fixed pitch, a 0 ends every read) against the same reads in the packer's format
(back to back, no terminator: CuCLARK_hh.hh:1616-1716) - what does the terminator's extra pass of the part loop cost?
    python tools/compact_reads_probe.py [workload]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from cuclark_amd import MiClarkDB, _lib

w = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "full"])
L = _lib.load()
dev = torch.device("cuda", 0)
k, T, n_reads, read_len = w["k"], w["n_targets"], w["n_reads"], w["read_len"]
spec = _lib.MicSynthSpec(seed=4, htsize=w["htsize"], genome_nt=w["genome_nt"], n_targets=T, n_genomes=w["n_genomes"], k=k, key_bytes=w["key_bytes"])
cap = int(w["genome_nt"]) + 1024
d_sizes = torch.empty(w["htsize"], dtype=torch.uint8, device=dev)
d_keys = torch.empty(cap, dtype=torch.int32 if w["key_bytes"] == 4 else torch.int64, device=dev)
d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
n_el = C.c_uint64(0)
assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
eng = MiClarkDB(k, T, device=0, layout=4)
os.environ.setdefault("MIC_SUPER2_MAY_FALL_BACK", "1")
eng.read_device(d_sizes.data_ptr(), w["htsize"], d_keys.data_ptr(), w["key_bytes"], d_labels.data_ptr())
del d_sizes, d_keys, d_labels
torch.cuda.empty_cache()
pitch = L.mic_synth_read_pitch(read_len, k)
d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev)
d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
assert L.mic_synth_reads_device2(C.byref(spec), 5, n_reads, read_len, 0, 0.2, 0.01, 0.001, d_rp.data_ptr(), d_cont.data_ptr(), d_cont.numel(), None, None) == 0
torch.cuda.synchronize()
d_res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)


def kernel_ms(rp, ct, reps=10):
    ms = []
    for i in range(reps + 2):
        eng.query_device(rp.data_ptr(), ct.data_ptr(), n_reads, d_res.data_ptr(), 0, 0)
        eng.resolve_flagged_device(rp.data_ptr(), ct.data_ptr(), d_res.data_ptr(), 0, 0)
        if i >= 2:
            ms.append(eng.last_query_ms())
    return float(np.mean(ms)), d_res.cpu().numpy().copy()


ms_pitch, res_pitch = kernel_ms(d_rp, d_cont)
rp = d_rp.cpu().numpy().view(np.uint32)
cont = d_cont.cpu().numpy().view(np.uint16)
pos = rp[:-1].astype(np.int64).copy()
end = rp[1:].astype(np.int64)
live = np.ones(n_reads, bool)
for _ in range(64):
    live &= pos < end
    plen = np.where(live, cont[np.minimum(pos, cont.size - 1)], 0).astype(np.int64)
    live &= plen > 0
    if not live.any():
        break
    pos = np.where(live, pos + 1 + (plen + 7) // 8, pos)
used = (pos - rp[:-1]).astype(np.int64)
rp_c = np.zeros(n_reads + 1, np.int64)
np.cumsum(used, out=rp_c[1:])
src = np.repeat(rp[:-1].astype(np.int64) - rp_c[:-1], used) + np.arange(int(rp_c[-1]), dtype=np.int64)
cont_c = np.concatenate([cont[src], np.zeros(64, np.uint16)])
d_rp2 = torch.from_numpy(rp_c.astype(np.uint32).view(np.int32)).to(dev)
d_cont2 = torch.from_numpy(cont_c.view(np.int16)).to(dev)
ms_compact, res_compact = kernel_ms(d_rp2, d_cont2)
print(f"pitch layout ({pitch * 2} B per read): {ms_pitch:.3f} ms; packer's format ({2 * rp_c[-1] / n_reads:.1f} B per read): {ms_compact:.3f} ms; "
      f"results equal: {bool((res_pitch[:, :5] == res_compact[:, :5]).all())}")
