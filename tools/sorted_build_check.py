#!/usr/bin/env python3
"""The sorted build of the one-strand super-k-mer table (mic_build.hip: s_expand_kernel ...) against the classic one (MIC_S_CLASSIC=1):
the same answers on random databases, and the stage times of both on a synthetic database of N k-mers.
    python tools/sorted_build_check.py [n_kmers_for_timing]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(mode):
    import numpy as np
    import golden_util as gu
    from cuclark_amd import MiClarkDB, host
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fp
    import pickle
    out = []
    cases = pickle.load(open(os.path.join(ROOT, "tools", "_sbc_cases.pkl"), "rb"))     # made on the CPU beforehand (fuzz_parity.make_case: slow Python)
    for seed, c in enumerate(cases, 1):
        with MiClarkDB(c["k"], c["T"], layout=3) as e:
            e.read_arrays(c["sizes"], c["keys"], c["labels"])
            res, _ = e.classify_packed(c["rp"], c["cont"], extended=True)
            info = e.info()
        ok = bool((res[:, :5] == c["expect"]).all())
        out.append((seed, c["k"], c["htsize"], int(info["n_elems"]), int(info["n_entries"]), int(info["n_slots_whole"]), ok))
        print(mode, out[-1], flush=True)
    assert all(o[-1] for o in out), "MISMATCH"


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("sorted", "classic", "sorted_chunks", "sorted_nosort"):
        child(sys.argv[1])
        sys.exit(0)
    for mode, env in (("classic", {"MIC_S_CLASSIC": "1"}), ("sorted", {}), ("sorted_chunks", {"MIC_S_SORT_CHUNK": "3000"}), ("sorted_nosort", {"MIC_S_NOSORT": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, MIC_LOAD_TIMING="1", **env), capture_output=True, text=True)
        print(r.stdout[-3000:], flush=True)
        print("\n".join(l for l in r.stderr.splitlines() if "sorted build:" in l or "candidates" in l or "Error" in l or "error" in l)[-3000:])
        print(mode, "exit", r.returncode, flush=True)
    n = sys.argv[1] if len(sys.argv) > 1 else "0"
    if int(n) > 0:
        for mode, env in (("classic", {"MIC_S_CLASSIC": "1"}), ("sorted", {})):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "light27", "--layout", "super", "--steps", "2", "--warmup", "1", "--no-cpu",
                                "--no-pipeline", "--no-e2e", "--no-parts-proxy", "--no-default-layout"], env=dict(os.environ, MIC_LOAD_TIMING="1", **env),
                               capture_output=True, text=True)
            print(mode, "bench exit", r.returncode)
            print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[load]")))
            print(r.stdout[-300:] if r.returncode else "", r.stderr[-1500:] if r.returncode else "")
