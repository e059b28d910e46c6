import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    class G:
        def npz(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

        def json(self, name):
            with open(os.path.join(GOLDEN, name + ".json")) as f:
                return json.load(f)
    return G()
