import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)
E, N, red = (4, 4, 4), 3, 2
out = []
for graph in (1, 0, 1, 0):
    p = H.Problem.box(E, (1, 1, 1), N, red, True)
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    p.set_flag("amg_graph", graph)
    m = S.ArrayMesh.from_problem(p)
    dof = p.sub_point_dofs()
    p.amg_attach(S.low_order_hierarchy(m, dof, p.info["sub_num_dofs"]))
    r = S.seeded_uniform(p.n, 9) - 0.5
    out.append([p.amg_apply(r) for _ in range(3)])
    p.close()
ref = out[0][0]
for g, o in zip((1, 0, 1, 0), out):
    print(g, [float(np.abs(z - ref).max()) for z in o], float(np.abs(ref).max()))
