import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_the_gpu_first():
    """PyTorch's wheel carries its own HIP runtime; when a process uses both torch.cuda and libbisbm_hip.so (the
    pooled-marginals path hands a torch device tensor to the library), torch must bring the device up first -- the
    other order leaves torch without a GPU ("No HIP GPUs are available").  bench.py has that order by construction."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
