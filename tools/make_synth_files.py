#!/usr/bin/env python3
"""Write a synthetic database (.sz/.ky/.lb under CuCLARK's file name), a targets file and a FASTQ of reads to a
directory, using libmi_clark.so's in-HBM generators — for end-to-end runs of exe/cuCLARK on the GPU box.

    python tools/make_synth_files.py OUTDIR [--light] [--reads N] [--kmers M]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--light", action="store_true")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--kmers", type=int, default=50_000_000)
    ap.add_argument("--targets", type=int, default=200)
    ap.add_argument("--paired", action="store_true", help="also write reads_1.fq / reads_2.fq (the two mates of every pair)")
    a = ap.parse_args()
    from cuclark_amd import _lib
    L = _lib.load()
    os.makedirs(a.out, exist_ok=True)
    dev = torch.device("cuda:0")
    htsize = 57777779 if a.light else 1610612741
    k = 27 if a.light else 31
    T = a.targets
    spec = _lib.MicSynthSpec(seed=11, htsize=htsize, genome_nt=a.kmers, n_targets=T, n_genomes=T * 2, k=k, key_bytes=4)
    cap = a.kmers + 1024
    d_sizes = torch.empty(htsize, dtype=torch.uint8, device=dev)
    d_keys = torch.empty(cap, dtype=torch.int32, device=dev)
    d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
    n_el = C.c_uint64(0)
    torch.cuda.synchronize()
    assert L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None) == 0
    n_el = n_el.value
    dbdir = os.path.join(a.out, "DB")
    os.makedirs(dbdir, exist_ok=True)
    name = f"db_central_k{k}_t{T}_s{htsize}_m0" + ("_light_4" if a.light else "") + ".tsk"
    d_sizes.cpu().numpy().tofile(os.path.join(dbdir, name + ".sz"))
    d_keys[:n_el].cpu().numpy().tofile(os.path.join(dbdir, name + ".ky"))
    d_labels[:n_el].cpu().numpy().tofile(os.path.join(dbdir, name + ".lb"))
    # targets file: one (existing) file per label
    with open(os.path.join(a.out, "targets.txt"), "w") as f:
        dummy = os.path.join(a.out, "genome.fa")
        open(dummy, "w").write(">g\nACGT\n")
        for t in range(T):
            f.write(f"{dummy} TARGET_{t:04d}\n")
    # reads -> FASTQ
    read_len = 150
    pitch = L.mic_synth_read_pitch(read_len, k)
    d_rp = torch.empty(a.reads + 1, dtype=torch.int32, device=dev)
    d_cont = torch.zeros(a.reads * pitch + 64, dtype=torch.int16, device=dev)
    assert L.mic_synth_reads_device(C.byref(spec), 5, a.reads, read_len, 0.2, 0.01, 0.0, d_rp.data_ptr(), d_cont.data_ptr(),
                                    d_cont.numel(), None, None) == 0
    torch.cuda.synchronize()
    cont = d_cont.cpu().numpy().view(np.uint16)[: a.reads * pitch].reshape(a.reads, pitch)
    assert (cont[:, 0] == read_len).all()          # n_rate = 0: one part per read
    nc = (read_len + 7) // 8
    codes = np.zeros((a.reads, nc * 8), np.uint8)
    for j in range(8):
        codes[:, j::8] = (cont[:, 1:1 + nc] >> (14 - 2 * j)) & 3
    seq = np.frombuffer(b"TGCA", np.uint8)[codes[:, :read_len]]
    qual = np.full(read_len, ord("I"), np.uint8)
    with open(os.path.join(a.out, "reads.fq"), "wb") as f:
        CH = 100000
        for s in range(0, a.reads, CH):
            parts = []
            for i in range(s, min(a.reads, s + CH)):
                parts.append(b"@r%d\n" % i + seq[i].tobytes() + b"\n+\n" + qual.tobytes() + b"\n")
            f.write(b"".join(parts))
    if a.paired:
        rec = int(L.mic_synth_text_record_bytes(read_len, 0))
        d_text = torch.empty(a.reads * rec, dtype=torch.uint8, device=dev)
        for mate in (0, 1):
            assert L.mic_synth_reads_text_device(C.byref(spec), 5, a.reads, read_len, 0.2, 0.01, 0.001, 0, mate, d_text.data_ptr(), d_text.numel(), None) == 0
            torch.cuda.synchronize()
            d_text.cpu().numpy().tofile(os.path.join(a.out, f"reads_{mate + 1}.fq"))
    print(f"wrote {dbdir}/{name}.* ({n_el} k-mers), {a.reads} reads, k={k}")


if __name__ == "__main__":
    main()
