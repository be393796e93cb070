import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: single-process oracle-parity tests first, multi-process tests last, so that under
# `pytest -x` a failure of the test infrastructure (rendezvous, spawn) can never hide the per-kernel parity evidence.
_FILE_ORDER = ("test_cpu_abi", "test_cpu_oracle", "test_cpu_amg", "test_gpu_kernels", "test_gpu_kernels_f32", "test_gpu_host", "test_gpu_amg",
               "test_gpu_fullsize", "test_cpu_multirank", "test_gpu_comm")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER) - 2  # unknown files: before the multi-process ones

    items.sort(key=key)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def gpu():
    """GPU tests call the HIP path through the C-ABI; never a fallback."""
    import torch

    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    lib.hip()  # raises loudly if libfdd_hip.so is missing
    return torch.device("cuda:0")
