import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    # The per-frame CNN (DepthFusionNet) runs on MIOpen, which by default does not pick the same convolution algorithm on every call: rebuilt maps differ in the last
    # bits (6-7e-7, 0 of 156 rebuilds bit-identical: tools/miopen_repeat.py), which round 5 answered by widening the module-level repeatability bars to 3e-5.  With the
    # deterministic switch every rebuild is bit-identical (156 of 156), so the suite pins it and the bars are back at 1e-5.
    try:
        import torch
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
