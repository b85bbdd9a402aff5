"""A 60-second seeded slice of tools/fuzz_parity.py under -m gpu: random tables (size, k, key width, labels), random reads, all
four layouts, whole table / bucket-range shards / parts of the table merged through the batch API - against the oracle."""
import os
import sys

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_sixty_seconds_of_random_configurations():
    sys.path.insert(0, os.path.join(gu.ROOT, "tools"))
    import fuzz_parity
    n_cases, n_reads = fuzz_parity.fuzz(60.0, seed0=20261004, verbose=False)
    assert n_cases >= 20 and n_reads > 2000
