import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (still part of the default CPU suite unless deselected)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = np.load(os.path.join(here, "golden.npz"))
    with open(os.path.join(here, "golden_meta.json")) as f:
        meta = json.load(f)
    return data, meta


@pytest.fixture(scope="session")
def torch_first():
    """GPU tests that hand torch tensors to the library share one process with it: initialise torch's HIP runtime before
    the library touches the device ("import torch first"), whatever order the modules run in."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    return torch
