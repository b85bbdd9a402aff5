"""Debugging aid: one read (or a substring of it) of fuzz configuration `seed` through the one-strand super-k-mer table, then the
crowd work area of that launch decoded (mi_clark.h: mic_debug_fetch_crowd): pending reads, items, the k-mers each item stands for.
  python tools/dbg/crowd_dump.py SEED READ_INDEX [FROM TO]      (on a GPU box)"""
import os, sys, collections, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import fuzz_parity as fp
import golden_util as gu
o = gu.oracle()
def pack(data, k):
    idx = o.index_reads(data)
    return o.pack_batch(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
seed, ridx = int(sys.argv[1]), int(sys.argv[2])
c = fp.make_case(seed, pack)
from cuclark_amd import MiClarkDB, host, _lib
k, T, ht = c["k"], c["T"], c["htsize"]
m = max(min(20, k - 4, 31), k - 15)       # mic_engine.hip: the minimizer length of the super-k-mer layouts
ctx = k - m
sizes = c["sizes"].astype(np.int64); keys = c["keys"].astype(object); labels = c["labels"]
db = {}; pos = 0
for b in range(ht):
    for i in range(int(sizes[b])):
        db[int(keys[pos]) * ht + b] = int(labels[pos]); pos += 1
seq = c["data"].split(b"\n")[2 * ridx + 1].decode()        # (FASTA configurations: one line per read)
a0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0; b0 = int(sys.argv[4]) if len(sys.argv) > 4 else len(seq)
data = f">p\n{seq[a0:b0]}\n".encode()
print("read", seq[a0:b0])
idx = host.index_reads(data)
rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
dev = torch.device("cuda:0")
L = _lib.load()
def tostr(v, n): return "".join("TGCA"[(v >> (2 * (n - 1 - i))) & 3] for i in range(n))
with MiClarkDB(k, T, layout=3) as e:
    e.read_arrays(c["sizes"], c["keys"], c["labels"])
    d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
    d_ct = torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint16)]).view(np.int16)).to(dev)
    d_res = torch.zeros((1, 8), dtype=torch.int32, device=dev)
    e.query_device(d_rp.data_ptr(), d_ct.data_ptr(), 1, d_res.data_ptr())
    print("result", d_res.cpu().numpy().view(np.uint32), e.last_crowd_stats())
    buf = np.zeros(1 << 16, np.uint32); caps = (C.c_uint32 * 3)()
    assert L.mic_debug_fetch_crowd(e.h, buf.ctypes.data, buf.size, caps) == 0
    pc, ic = caps[0], caps[1]
    print("hdr", buf[:4], "caps", list(caps))
    pend = buf[8:8 + 8 * pc].reshape(-1, 8); items = buf[8 + 8 * pc:]
    npend, nitems = int(buf[0]), int(buf[1])
    for p in range(npend):
        r, total, meta, off, cg = [int(x) for x in pend[p, :5]]
        print("pending read", r, "hits so far", total, "entries", meta & 255, "cg", hex(cg))
    itm = items[: 8 * nitems].reshape(-1, 8)
    for i in range(nitems):
        G0, G1, G2, jj, prev = [int(x) for x in itm[i, :5]]
        jmin, jmax = jj & 255, (jj >> 8) & 255
        G = (G0 << 64) | (G1 << 32) | G2
        region = tostr(G >> (96 - 2 * (k + ctx)), k + ctx)
        out = []
        for j in range(jmin, jmax + 1):
            km = region[ctx - j: ctx - j + k]
            v = 0
            for ch in km: v = (v << 2) | {"A": 3, "C": 2, "G": 1, "T": 0}[ch]
            out.append((j, km, db.get(int(o.canonical(v, k)))))
        print("item", i, "jmin", jmin, "jmax", jmax, "prev", hex(prev), "region", region)
        for t_ in out: print("      ", t_)
