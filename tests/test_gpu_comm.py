"""Communication back-ends on the GPU box.  One GPU is available to the tests,
so the process group has world size 1; the collectives are still driven
through their real code paths (device-pointer wrapping for torch.distributed
"nccl" = RCCL, and RCCL called directly from the host layer) by
fddh_comm_selftest, which bypasses the solver's size == 1 shortcuts.  The
world_size 2/4 logic is covered on CPU by tests/test_cpu_multirank.py."""
import os

import numpy as np
import pytest

from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group(gpu):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    H.init(0, use_torch_stream=True)
    H.set_print(False)
    yield dist
    H.comm_single()
    dist.destroy_process_group()


def test_torch_distributed_callbacks_on_device_buffers(group):
    H.comm_torch_callbacks(on_gpu=True)
    lib.host().call("fddh_comm_selftest", 100000)


def test_rccl_called_directly(group):
    H.comm_rccl_from_torch()
    lib.host().call("fddh_comm_selftest", 100000)


def test_solve_is_identical_under_every_backend(group):
    results = []
    for setup in (H.comm_single, lambda: H.comm_torch_callbacks(on_gpu=True), H.comm_rccl_from_torch):
        setup()
        p = H.Problem.box((4, 4, 4), (1, 1, 1), 3, 2, True)
        _, f = p.make_rhs(0, 0)
        u, its, hist = p.solve(f, "fcg")
        results.append((u, its, hist))
        p.close()
    for u, its, hist in results[1:]:
        assert its == results[0][1] and np.array_equal(hist, results[0][2]) and np.array_equal(u, results[0][0])
