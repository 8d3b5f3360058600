import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("MICRONET_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np

    class G:
        q = np.load(os.path.join(GOLDEN, "quantizers.npz"))
        m = np.load(os.path.join(GOLDEN, "modules.npz"))
        mo = np.load(os.path.join(GOLDEN, "models.npz"))
        meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    return G
