#!/usr/bin/env python3
"""Where the time of a run of exe/cuCLARK-l over a small compressed FASTQ goes when the ingest slots are tiny (the GPU suite's
many-batches cases): one run at a time, the slot size swept.  Usage: python tools/cli_small_slots_probe.py [out.log]"""
import gzip, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_cli as tc
import test_ingest as ti

def main():
    log = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    tmp = tempfile.mkdtemp(prefix="mic_probe_")
    d = tc._db_dir(tmp, "light_k27_u32", light=True)
    t = tc._targets_file(tmp)
    rng = np.random.default_rng(41)
    fq = os.path.join(tmp, "r.fq")
    open(fq, "wb").write(ti._random_reads(rng, ti._genomes(), 5000, fasta=False))
    gz = fq + ".gz"
    open(gz, "wb").write(gzip.compress(open(fq, "rb").read(), 1))
    for src in (fq, gz):
        for kb, extra in (("24", {}), ("48", {}), ("96", {}), ("0", {}), ("24", {"OMP_WAIT_POLICY": "PASSIVE"}), ("24", {"OMP_NUM_THREADS": "1"})):
            env = dict(os.environ, MIC_CLI_TIMING="1", **extra)
            if kb != "0":
                env["MIC_INGEST_KB"] = kb
            t0 = time.time()
            r = subprocess.run([tc.EXE_L, "-T", t, "-D", d, "-O", src, "-R", os.path.join(tmp, "o"), "-n", "5"], capture_output=True, text=True, env=env)
            dt = time.time() - t0
            ing = [l for l in r.stderr.splitlines() if "device ingest:" in l]
            slow = [l for l in r.stderr.splitlines() if "batch slots" in l]
            log.write(f"{os.path.basename(src)} KB={kb} {extra}: {dt:.2f} s rc={r.returncode}; {ing[0].strip() if ing else ''}; host-path segments {len(slow)}"
                      + (f", batch-slot laps {slow[1].split(':')[-1].strip()} .. {slow[-1].split(':')[-1].strip()}" if len(slow) > 1 else "") + "\n")
            log.flush()

if __name__ == "__main__":
    main()
