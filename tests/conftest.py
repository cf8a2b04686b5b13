import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no GPU is visible, e.g. a plain `pytest tests/` here."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_meta(z):
    return json.loads(bytes(z["meta_json"]).decode())


@pytest.fixture(scope="session")
def golden_search():
    z = load_golden("search.npz")
    return z, golden_meta(z)


@pytest.fixture(scope="session")
def golden_games():
    z = load_golden("games.npz")
    return z, golden_meta(z)
