import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-faster-rcnn_amd")
for p in (ROOT, os.path.join(ROOT, "oracle"), PKG, os.path.join(PKG, "lib")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLD, name + ".npz"))
    return G()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, scope="session")
def _pin_cpu_cython_path():
    """BASELINE.json's parity clause names the reference's CPU/Cython path: the goldens come from cpu_nms (`ovr >= thresh`).
    cfg.USE_GPU_NMS defaults to True like the reference and then selects the CUDA kernel's `>` rule (tests/test_boundary_gpu.py
    covers that mode explicitly), so the suite pins the CPU rule here."""
    from model.config import cfg
    old = cfg.USE_GPU_NMS
    cfg.USE_GPU_NMS = False
    yield
    cfg.USE_GPU_NMS = old
