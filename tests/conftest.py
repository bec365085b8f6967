import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _seed_everything():
    """every test draws the same inputs on every run (no tolerance check may depend on an unseeded generator)"""
    import numpy as np
    import torch
    np.random.seed(0)
    torch.manual_seed(0)        # seeds the default CPU and (when present) GPU generators
    yield
