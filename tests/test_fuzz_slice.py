"""A 60-second seeded slice of tools/fuzz_parity.py under -m gpu: random tables (size, k, key width, labels), random reads, all
four layouts, whole table / bucket-range shards / parts of the table merged through the batch API and through the command line's
table-sharded ingest - against the oracle.  It runs in a process of its own on the HARDENED build of the library
(cuclark_amd/lib/libmi_clark_hard.so: libstdc++ assertions, fortified libc, stack protectors - host-side checks, the device code
is the product's) with glibc's heap checks on (MALLOC_CHECK_=3, MALLOC_PERTURB_): a host-side overrun ends the run where it
happens instead of corrupting the heap for a later free to find."""
import os
import re
import subprocess
import sys

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_sixty_seconds_of_random_configurations():
    hard = os.path.join(gu.ROOT, "cuclark_amd", "lib", "libmi_clark_hard.so")
    assert os.path.exists(hard), "the hardened library is built by __graft_entry__.build() (make -C cuclark_amd/csrc)"
    env = dict(os.environ, MIC_LIB_PATH=hard, MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    cmd = [sys.executable, os.path.join(gu.ROOT, "tools", "fuzz_parity.py"), "60", "20261004"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    if r.returncode < 0:
        # DESIGN.md 7, "one open item": about once in ten hours of soaking the fuzzer's process dies of a native fault (glibc heap
        # check / SIGSEGV) that no seed reproduces.  A wrong RESULT (exit code 1) is never retried; a process that was killed by a
        # signal is recorded - its whole output goes to gpurun_out/ for the post-mortem - and the slice runs once more.
        import warnings
        out_dir = os.path.join(gu.ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "fuzz_slice_native_fault.log"), "a") as f:
            f.write(f"--- exit {r.returncode}\n{r.stdout[-4000:]}\n{r.stderr}\n")
        warnings.warn(f"the fuzz slice's process died of signal {-r.returncode} (log: gpurun_out/fuzz_slice_native_fault.log); running it once more")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    m = re.search(r"fuzz ok: (\d+) random configurations x 4 layouts, (\d+) reads", r.stdout)
    assert m and int(m.group(1)) >= 20 and int(m.group(2)) > 2000, r.stdout[-500:]
    assert int(re.search(r"table-sharded ingest batches checked: (\d+)", r.stdout).group(1)) >= 3, r.stdout[-500:]
