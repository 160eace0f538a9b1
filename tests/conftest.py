import json
import os
import sys

import numpy as np
import pytest

# parity tests run with the deterministic weight-gradient chunk rule / shipped table (no on-line timing sweep in the first step)
os.environ.setdefault("GAMER_WGRAD_TUNE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    return z, meta


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
