#!/usr/bin/env python3
"""tools/fuzz_parity.py's call sequences from PYTHON (ctypes wrappers, numpy buffers) against the host-only, ASan-instrumented build of
the library on the mock HIP runtime (tools/sanitize/build_host_rig.sh: libmi_clark_mock.so) - no GPU, no oracle, no comparisons (the
mock's device answers with zeros): the sanitizer's runtime is preloaded, so every malloc of the process is under its watch and
a write of the library's past a ctypes structure, an out-parameter or a numpy buffer is reported where it happens.
    LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) PYTHONMALLOC=malloc \
    ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=1 MIC_LIB_PATH=<out>/libmi_clark_mock.so python tools/sanitize/py_rig.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
import test_gpu_parity as tp
from cuclark_amd import MiClarkDB, host
budget = float(sys.argv[1]); seed = int(sys.argv[2])
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    k = int(rng.choice([8, 12, 16, 20, 21, 24, 25, 27, 31, 32]))
    htsize = int(rng.choice([2, 97, 1009, 4096, 65537, 99991]))
    key_bytes = host.key_bytes_rule(htsize, k)
    n_elems = int(rng.integers(50, 60000))
    if k < 16: n_elems = min(n_elems, (1 << (2 * k)) // 3)
    n_elems = min(n_elems, htsize * 200)
    T = int(rng.choice([1, 2, 7, 40, 64, 65, 300, 4096]))
    sizes, keys, labels, canon = gu.random_db(rng, htsize, n_elems, k, key_bytes, T)
    L = int(rng.choice([k, k + 1, 40, 100, 150, 151, 250, 400]))
    data = tp._random_reads(rng, canon, k, int(rng.integers(50, 400)), max(L, k))
    idx = host.index_reads(data)
    rp, cont = host.pack_reads(data, idx["seq_s"], idx["seq_e"], idx["length"], k)
    nr = rp.size - 1
    for layout in (1, 2, 3, 4):
        try:
            with MiClarkDB(k, T, layout=layout) as e:
                e.read_arrays(sizes, keys, labels)
                e.info()
                res, rows = e.classify_packed(rp, cont, extended=True)
        except Exception as ex:
            pass
        np_ = int(rng.integers(2, 6))
        cuts = [dict(part=(p, np_)) for p in range(np_)] if rng.random() < 0.5 else [dict(shard=(0, max(1, htsize // 2))), dict(shard=(max(1, htsize // 2), htsize))]
        engines = [MiClarkDB(k, T, layout=layout) for _ in cuts]
        try:
            for e, c in zip(engines, cuts):
                if "part" in c:
                    e.set_part(*c["part"]); e.read_arrays(sizes, keys, labels)
                else:
                    e.read_arrays(sizes, keys, labels, shard=c["shard"])
                b = e.malloc(nr, nr, max(cont.size, 1), [0, nr], True)
                b["reads_pointer"][0][: nr + 1] = rp
                b["containers"][0][: cont.size] = cont
                e.readyBatch(0, nr, cont.size); e.queryBatch(0, True)
            MiClarkDB.merge_shards(engines, 0)
            r2 = engines[0]._bufs["results"].copy()
            if "part" in cuts[0] and len(data) <= (1 << 20):
                owner = int(rng.integers(0, len(engines)))
                engines[owner].ingest_alloc(1, 1 << 20, [f"t{i}" for i in range(T)], want_results=True)
                g = MiClarkDB.ingest_classify_group(engines, owner, 0, data)
                engines[owner].ingest_free()
        except Exception as ex:
            pass
        finally:
            for e in engines: e.close()
    n += 1; seed += 1
print("py rig ok:", n, "configurations")
