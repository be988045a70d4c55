import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# stated fp32 tolerances (max abs) against the reference CPU path; the
# reference's own fp32-vs-fp64 noise floor is 3e-6 / 5e-7 / 1e-5 / 1.6e-4 for these keys.
TOL = {"prediction": 1e-4, "mask": 1e-5, "occlusion_map": 1e-5, "deformation": 1e-5,
       "sparse_deformed": 5e-4, "deformed": 5e-4}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
