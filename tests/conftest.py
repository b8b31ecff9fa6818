import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A gpu-marked test on a box without a GPU is an error in how the suite
    was invoked, not a skip: the driver selects with -m gpu / -m "not gpu"."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        # the one-GPU-per-rank RCCL test first, before this process has touched a GPU (its ranks are spawned processes)
        first = [it for it in items if "rccl" in it.name]
        if first:
            items[:] = first + [it for it in items if it not in first]
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
