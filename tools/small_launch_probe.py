#!/usr/bin/env python3
"""Query-kernel time of SMALL launches (a command-line batch: 50 k - 1 M reads) against the grid rule of mic_launch_query:
reads per wave (MIC_GRID_DEBUG=1 MIC_READS_PER_WAVE=r) on the headline table.   python tools/small_launch_probe.py [--layout auto]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIC_GRID_DEBUG"] = "1"
import numpy as np
import torch
import bench as B
from cuclark_amd import MiClarkDB, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--layout", default="auto")
ap.add_argument("--workload", default="full")
ap.add_argument("--parts", type=int, default=0)
args = ap.parse_args()
L = _lib.load()
w = B.WORKLOADS[args.workload]
dev = torch.device("cuda:0")
k, T = w["k"], w["n_targets"]
spec = _lib.MicSynthSpec(seed=4, htsize=w["htsize"], genome_nt=w["genome_nt"], n_targets=T, n_genomes=w["n_genomes"], k=k, key_bytes=w["key_bytes"])
cap = int(w["genome_nt"]) + 1024
d_sizes = torch.empty(w["htsize"], dtype=torch.uint8, device=dev)
d_keys = torch.empty(cap, dtype=torch.int32 if w["key_bytes"] == 4 else torch.int64, device=dev)
d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
n_el = C.c_uint64(0)
assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
eng = MiClarkDB(k, T, layout={"auto": 0, "super": 3, "super2": 4}[args.layout])
if args.parts:
    eng.set_part(0, args.parts)
eng.read_device(d_sizes.data_ptr(), w["htsize"], d_keys.data_ptr(), w["key_bytes"], d_labels.data_ptr())
del d_keys, d_labels, d_sizes
n_max = 2_000_000
pitch = L.mic_synth_read_pitch(150, k)
d_rp = torch.empty(n_max + 1, dtype=torch.int32, device=dev)
d_cont = torch.zeros(n_max * pitch + 64, dtype=torch.int16, device=dev)
assert L.mic_synth_reads_device2(C.byref(spec), 5, n_max, 150, 0, 0.2, 0.01, 0.001, d_rp.data_ptr(), d_cont.data_ptr(), d_cont.numel(), None, None) == 0
torch.cuda.synchronize()
d_res = torch.zeros((n_max, 8), dtype=torch.int32, device=dev)
out = {}
for n in (30_000, 70_000, 137_000, 300_000, 600_000, 2_000_000):
    row = {}
    # (besides the fixed values: exactly one / two / three generations of the 8 192 resident wavefronts)
    gens = sorted({max(1, -(-n // (8192 * g))) for g in (1, 2, 3)})
    for rpw in sorted(set((4, 8, 12, 16, 24, 32, 48)) | set(gens)):
        os.environ["MIC_READS_PER_WAVE"] = str(rpw)
        ms = []
        for it in range(8):
            eng.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n, d_res.data_ptr())
            ms.append(eng.last_query_ms())
        row[rpw] = round(float(np.median(ms[2:])) * 1e3, 1)
    out[n] = {"us": row, "generations_1_2_3_rpw": gens[::-1], "best_rpw": min(row, key=row.get), "Mreads_s_at_8": round(n / row[8], 1), "Mreads_s_best": round(n / min(row.values()), 1)}
    print(n, json.dumps(out[n]), file=sys.stderr, flush=True)
print(json.dumps({"what": "query kernel time (us) of small launches by reads per wave", "layout": args.layout, "parts": args.parts, "launches": out}))
