import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["bf16_3x", "highest"])
def prec(request):
    """Runs the test at both matmul precisions (serl_amd.set_matmul_precision) and restores the default."""
    import serl_amd
    serl_amd.set_matmul_precision(request.param)
    yield request.param
    serl_amd.set_matmul_precision("bf16_3x")
