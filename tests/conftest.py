import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_tree():
    """a checkout without build products (libraries, kept device assembly) and with hipcc at hand: build once, as __graft_entry__.build() does.
    Nothing is rebuilt when the products exist -- a stale build is for tests/test_waits_cpu.py to report, not to paper over."""
    import glob
    import shutil
    import subprocess
    pkg = os.path.join(ROOT, "amq_amd")
    have = (os.path.exists(os.path.join(pkg, "libamq_hip.so")) and os.path.exists(os.path.join(pkg, "libamq_hip_safe.so"))
            and glob.glob(os.path.join(pkg, "csrc", "asm", "*.s")))
    if not have and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-j", str(min(8, os.cpu_count() or 1)), "all", "safe"],
                       check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield


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
