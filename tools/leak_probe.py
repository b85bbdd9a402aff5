#!/usr/bin/env python3
"""Create, use and destroy engines in a loop and watch what the process holds on to: file descriptors, threads, resident memory,
free HBM.  The parity fuzzer builds tens of thousands of engines in one process; anything an engine leaks ends there.
    python tools/leak_probe.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def held():
    st = open("/proc/self/status").read()
    g = lambda key: int([l for l in st.splitlines() if l.startswith(key)][0].split()[1])
    import torch
    free, _ = torch.cuda.mem_get_info()
    return dict(fds=len(os.listdir("/proc/self/fd")), threads=g("Threads:"), rss_MB=g("VmRSS:") // 1024, hbm_free_GB=round(free / 1e9, 2),
                maps=sum(1 for _ in open("/proc/self/maps")))


def main(n_iter):
    import golden_util as gu
    import test_gpu_parity as tp
    from cuclark_amd import MiClarkDB, host
    rng = np.random.default_rng(5)
    k, T, htsize = 31, 300, 65537
    kb = host.key_bytes_rule(htsize, k)
    sizes, keys, labels, canon = gu.random_db(rng, htsize, 20000, k, kb, T)
    data = tp._random_reads(rng, canon, k, 120, 150)
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    n = rp.size - 1
    print(0, held(), flush=True)
    for it in range(1, n_iter + 1):
        layout = 1 + it % 4
        with MiClarkDB(k, T, layout=layout) as e:
            if it % 3 == 0:
                e.set_part(it % 2, 2)
            e.read_arrays(sizes, keys, labels)
            if it % 2:
                e.classify_packed(rp, cont, extended=True)
            else:
                b = e.malloc(n, n, max(cont.size, 1), [0, n], True)
                b["reads_pointer"][0][: n + 1] = rp
                b["containers"][0][: cont.size] = cont
                e.readyBatch(0, n, cont.size)
                e.queryBatch(0, True)
                e.waitForBatch(0)
        if it % 1000 == 0:
            print(it, held(), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
