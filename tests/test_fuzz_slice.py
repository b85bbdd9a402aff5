"""A 30-second seeded slice of tools/fuzz_parity.py under -m gpu: random tables (size, k, key width, labels), random reads, all
four layouts, whole table / bucket-range shards / parts of the table merged through the batch API and through the command line's
table-sharded ingest - against the oracle.  PRODUCT AND ORACLE RUN IN TWO PROCESSES (--split): this test's child loads the
HARDENED build of the library (cuclark_amd/lib/libmi_clark_hard.so: libstdc++ assertions, fortified libc, stack protectors -
host-side checks, the device code is the product's) with glibc's heap checks on (MALLOC_CHECK_=3, MALLOC_PERTURB_) and never
the oracle; the oracle lives in a grandchild and sends its arrays over a pipe.  A native fault is a FAILURE of this test and names
its side - nothing is run twice.  (The long form: tools/fuzz_parity.py <seconds> <seed> --split; the oracle alone under
AddressSanitizer: tools/sanitize/oracle_rig.sh; the command line alone: tools/cli_soak.sh.)"""
import os
import re
import subprocess
import sys

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_thirty_seconds_of_random_configurations_product_and_oracle_in_separate_processes():
    hard = os.path.join(gu.ROOT, "cuclark_amd", "lib", "libmi_clark_hard.so")
    assert os.path.exists(hard), "the hardened library is built by __graft_entry__.build() (make -C cuclark_amd/csrc)"
    env = dict(os.environ, MIC_LIB_PATH=hard, MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    cmd = [sys.executable, os.path.join(gu.ROOT, "tools", "fuzz_parity.py"), "30", "20261004", "--split"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        out_dir = os.path.join(gu.ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "fuzz_slice_failure.log"), "a") as f:
            f.write(f"--- exit {r.returncode}\n{r.stdout[-4000:]}\n{r.stderr}\n")
    assert r.returncode >= 0, f"the PRODUCT side of the fuzz slice died of signal {-r.returncode}:\n{r.stderr[-3000:]}"
    assert "ORACLE SIDE died" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    m = re.search(r"fuzz ok: (\d+) random configurations x 4 layouts, (\d+) reads", r.stdout)
    assert m and int(m.group(1)) >= 12 and int(m.group(2)) > 1500, r.stdout[-500:]
    assert "separate processes" in r.stdout
    assert int(re.search(r"table-sharded ingest batches checked: (\d+)", r.stdout).group(1)) >= 2, r.stdout[-500:]
