import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """GPU tests call the HIP path through the C-ABI; never a fallback."""
    import torch

    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    lib.hip()  # raises loudly if libfdd_hip.so is missing
    return torch.device("cuda:0")
