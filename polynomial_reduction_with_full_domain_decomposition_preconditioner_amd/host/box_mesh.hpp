/*
 * box_mesh.hpp -- synthetic input generator: the unit
 * cube cut into Ex x Ey x Ez affine hexes of degree N, partitioned into
 * Px x Py x Pz rank blocks with rank-contiguous element ids, producing exactly
 * the arrays Domain::initialize reads from a Nek5000 export
 * (domain.tpp:45-224).  Not part of the reference (its meshes come from
 * Nek5000 runs that are not in the repository); tests/support.py holds an
 * independent numpy statement of the same mesh that this one is checked
 * against.
 */
#ifndef FDD_BOX_MESH_HPP
#define FDD_BOX_MESH_HPP

#include <vector>

#include "element.hpp"
#include "gll.hpp"
#include "host_parallel.hpp"

namespace fdd
{

struct BoxSpec
{
    int E[3] = {4, 4, 4}; // global elements per direction
    int P[3] = {1, 1, 1}; // rank blocks per direction
};

// rank blocks for a cube of ranks: 1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2, ...
inline void default_rank_grid(int num_ranks, int P[3])
{
    P[0] = P[1] = P[2] = 1;
    int d = 0;
    while (num_ranks > 1 && num_ranks % 2 == 0)
    {
        P[d] *= 2;
        num_ranks /= 2;
        d = (d + 1) % 3;
    }
    P[0] *= num_ranks; // any odd remainder goes along x
}

template <typename DType>
MeshData<DType> make_box_mesh(const BoxSpec &spec, int N, int rank)
{
    MeshData<DType> m;
    m.dim = 3;
    m.poly_degree = N;
    const int n = N + 1;
    const int Ex = spec.E[0], Ey = spec.E[1], Ez = spec.E[2];
    const int Px = spec.P[0], Py = spec.P[1], Pz = spec.P[2];
    const int lx = Ex / Px, ly = Ey / Py, lz = Ez / Pz;
    const int rx = rank % Px, ry = (rank / Px) % Py, rz = rank / (Px * Py);
    const int ox = rx * lx, oy = ry * ly, oz = rz * lz;
    const long long Gx = (long long)Ex * N + 1, Gy = (long long)Ey * N + 1, Gz = (long long)Ez * N + 1;
    const double hx = 1.0 / Ex, hy = 1.0 / Ey, hz = 1.0 / Ez;

    m.num_local_elements = lx * ly * lz;
    const size_t P = (size_t)m.num_local_elements * n * n * n;

    std::vector<double> zg(n), wg(n);
    gll::zwgll(zg.data(), wg.data(), n);

    m.x.resize(P);
    m.y.resize(P);
    m.z.resize(P);
    m.glo_num.resize(P);
    m.node_degree.resize(P);
    m.p_mask.resize(P);
    for (int g = 0; g < NUM_GEOM_FACTS; g++) m.g[g].assign(P, 0.0);

    auto mult = [&](long long g, long long G) -> int { return (g % N == 0 && g > 0 && g < G - 1) ? 2 : 1; };

    // Global node ids the way Nek5000 hands them out: the element VERTICES first, 1..V in lexicographic order of
    // the (Ex+1)(Ey+1)(Ez+1) vertex grid -- the same ids at every polynomial degree, which is what lets elements of
    // different degree share their corners in the composite region (subdomain.tpp:930-966 restores the corner ids
    // after the per-level offset) -- then every other node, lexicographic on the degree-N grid.
    const long long V = (long long)(Ex + 1) * (Ey + 1) * (Ez + 1);

    // element ranges on the rank's host threads (every element writes its own n^3 points)
    const long long n3 = (long long)n * n * n;
    low_order::parallel_ranges(m.num_local_elements, low_order::range_parts(m.num_local_elements), [&](long long e0, long long e1, int) {
        for (long long e = e0; e < e1; e++)
            {
                const int ex = (int)(e % lx), ey = (int)((e / lx) % ly), ez = (int)(e / ((long long)lx * ly));
                const long long EX = ox + ex, EY = oy + ey, EZ = oz + ez;
                size_t p = (size_t)(e * n3);
                for (int k = 0; k < n; k++)
                    for (int j = 0; j < n; j++)
                        for (int i = 0; i < n; i++, p++)
                        {
                            const long long gi = EX * N + i, gj = EY * N + j, gk = EZ * N + k;
                            if (gi % N == 0 && gj % N == 0 && gk % N == 0)
                                m.glo_num[p] = 1 + gi / N + (long long)(Ex + 1) * (gj / N + (long long)(Ey + 1) * (gk / N));
                            else
                                m.glo_num[p] = V + 1 + gi + Gx * (gj + Gy * gk);
                            m.node_degree[p] = mult(gi, Gx) * mult(gj, Gy) * mult(gk, Gz);
                            const bool bd = gi == 0 || gi == Gx - 1 || gj == 0 || gj == Gy - 1 || gk == 0 || gk == Gz - 1;
                            m.p_mask[p] = bd ? 0.0 : 1.0;
                            m.x[p] = (EX + 0.5 * (zg[i] + 1.0)) * hx;
                            m.y[p] = (EY + 0.5 * (zg[j] + 1.0)) * hy;
                            m.z[p] = (EZ + 0.5 * (zg[k] + 1.0)) * hz;
                            // quadrature weights folded in: G_rr = w_i w_j w_k * hy*hz/(2*hx) ...
                            const double www = wg[i] * wg[j] * wg[k];
                            m.g[0][p] = www * (hy * hz) / (2.0 * hx);
                            m.g[1][p] = www * (hx * hz) / (2.0 * hy);
                            m.g[2][p] = www * (hx * hy) / (2.0 * hz);
                        }
            }
    });
    return m;
}

} // namespace fdd

#endif
