import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


from oracle.synth import voronoi_image  # noqa: E402,F401


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# the product's stand-in for the re-sample draw of mpeg7Descriptors.py:101, restated beside the oracle
from oracle.dcd_ref import mix32, hash_resample_idx as resample_rule      # noqa: E402,F401
