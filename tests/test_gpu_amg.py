"""GPU parity of the low-order AMG V-cycle preconditioner (SURVEY.md 8 row a14,
config C5's inner preconditioner) against the oracle, through include/fdd_host.h.
The checks themselves live in amg_checks.py (shared with the CPU-shim test).

The host layer runs on its own HIP stream here so that the V-cycle goes through
hipGraph capture + replay (the default stream cannot be captured); a second
pass with "amg_graph" off runs the same launches eagerly and must give the
same bits."""
import numpy as np
import pytest

import amg_checks
import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def own_stream(gpu):
    H.init(0, use_torch_stream=False)
    H.comm_single()
    H.set_print(False)
    yield True
    H.init(0)  # back to torch's current stream for the other modules


def make_problem(E, N, red):
    p = H.Problem.box(E, (1, 1, 1), N, red, True)
    p.set_flag("sub_use_preconditioner", 0)  # switched on by the checks once the hierarchy under test is attached
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    return p


@pytest.mark.parametrize("E,N,red", [((4, 4, 4), 3, 2), ((3, 2, 2), 5, 2)])
def test_amg_preconditioner_matches_oracle(own_stream, E, N, red):
    p = make_problem(E, N, red)
    try:
        its = amg_checks.check_amg(p, N, red)
        assert its is not None and its <= 6
    finally:
        p.close()


def test_graph_replay_equals_eager_launches(own_stream):
    E, N, red = (4, 4, 4), 3, 2
    out = []
    for graph in (1, 0):
        p = make_problem(E, N, red)
        try:
            p.set_flag("amg_graph", graph)
            m = S.ArrayMesh.from_problem(p)
            dof = p.sub_point_dofs()
            p.amg_attach(S.low_order_hierarchy(m, dof, p.info["sub_num_dofs"]))
            r = S.seeded_uniform(p.n, 9) - 0.5
            out.append([p.amg_apply(r) for _ in range(3)])
        finally:
            p.close()
    for z in out[0][1:] + out[1]:
        assert np.array_equal(z, out[0][0])


def test_fused_smoother_equals_reference_launch_sequence(own_stream):
    """"amg_fused_smoother" only moves the element-wise smoother kernels into the SpMV
    epilogues: the V-cycle result must not change in any bit."""
    E, N, red = (4, 4, 4), 3, 2
    out = []
    for fused in (1, 0):
        p = make_problem(E, N, red)
        try:
            p.set_flag("amg_fused_smoother", fused)
            m = S.ArrayMesh.from_problem(p)
            dof = p.sub_point_dofs()
            p.amg_attach(S.low_order_hierarchy(m, dof, p.info["sub_num_dofs"]))
            r = S.seeded_uniform(p.n, 9) - 0.5
            out.append(p.amg_apply(r))
        finally:
            p.close()
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("E,N,red", [((4, 4, 4), 3, 2), ((3, 2, 2), 5, 2)])
def test_float_vcycle_matches_oracle(own_stream, E, N, red):
    """`Float = float` (AMG/config.hpp:4): f32 SpMV + fused smoother kernels, one hipGraph (amg_checks.check_amg_f32)."""
    p = make_problem(E, N, red)
    try:
        its = amg_checks.check_amg_f32(p, N, red)
        assert its <= 8
    finally:
        p.close()


def test_errors(own_stream):
    p = make_problem((2, 2, 2), 3, 2)
    try:
        with pytest.raises(Exception):
            p.amg_apply(np.zeros(p.n))  # nothing attached
        m = S.ArrayMesh.from_problem(p)
        levels = S.low_order_hierarchy(m, p.sub_point_dofs(), p.info["sub_num_dofs"])
        bad = [dict(levels[1])]
        with pytest.raises(Exception):
            p.amg_attach(bad)  # finest level must have the subdomain's dofs
    finally:
        p.close()


@pytest.mark.parametrize("E,N,red", [((4, 4, 4), 3, 2), ((3, 2, 2), 6, 2), ((3, 2, 2), 7, 2)])
def test_amg_with_the_host_layers_own_hierarchy(own_stream, E, N, red):
    """Low-order FEM matrix + smoothed-aggregation hierarchy built by host/low_order.hpp,
    V-cycle on the GPU (one hipGraph), against the oracle fed with the same arrays."""
    p = make_problem(E, N, red)
    try:
        its = amg_checks.check_amg(p, N, red, builder="product")
        assert its is not None and its <= 8
    finally:
        p.close()


@pytest.mark.parametrize("cheby_order,num_vcycles", [(1, 1), (2, 2), (3, 1), (4, 2)])
def test_reference_sweep_parameters_as_run_time_switches(own_stream, cheby_order, num_vcycles):
    """The parameters the reference's harness sweeps by rewriting header lines and rebuilding (run.py:150-156: subdomain.hpp:236
    `num_vcycles`, :237 `cheby_order`, clamped to 1..4 by subdomain.tpp:3477-3478) as run-time switches of the host layer
    (`amg_num_vcycles`, `amg_cheby_order`; `poisson --vcycles / --cheby`): the V-cycle with them against the oracle's with the
    same hierarchy, the inner solves and the outer solve iteration for iteration; the order cannot change under a built
    hierarchy (its coefficients are the hierarchy's)."""
    p = make_problem((4, 4, 4), 3, 2)
    try:
        its = amg_checks.check_amg(p, 3, 2, builder="product", cheby_order=cheby_order, num_vcycles=num_vcycles)
        assert its is not None and its <= 10
        with pytest.raises(lib.FddError):
            p.set_flag("amg_cheby_order", 1 if cheby_order != 1 else 2)
        p.set_flag("amg_num_vcycles", 1)  # any time: the captured graph is dropped
        with pytest.raises(lib.FddError):
            p.set_flag("amg_num_vcycles", 0)
    finally:
        p.close()
