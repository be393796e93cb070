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

#include <atomic>
#include <stdexcept>
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
    // Kershaw deformation of the cube (the reference's own experiment geometry: every mesh of run.py:25-47 and run.sh:30 is
    // a Nek5000 "Kershaw" export with eps = 0.3).  1.0 = the uniform box.
    double kershaw_eps[2] = {1.0, 1.0}; // eps_y, eps_z in (0, 1]
    bool deformed() const { return kershaw_eps[0] != 1.0 || kershaw_eps[1] != 1.0; }
};

// The generalized Kershaw map of the unit cube onto itself (D. Kershaw, J. Comput. Phys. 39 (1981) 375-395, in the 3-D
// form the CEED bake-off problems and Nek5000's kershaw case use): x is kept; the x-range is cut into six layers whose
// yz-sections go left-to-left, left-to-right, right-to-left (two layers), left-to-right, right-to-right, where "left" /
// "right" compress the lower / upper half of [0,1] to eps/2.  Piecewise trilinear, continuous, boundary-preserving;
// eps = 1 is the identity.  The reference reads the deformed GLL coordinates and factors from Nek5000 exports that are not
// in the repository; here the map is applied to the GLL points of the box mesh and the factors follow isoparametrically.
namespace kershaw
{
inline double right(double eps, double x) { return x <= 0.5 ? (2.0 - eps) * x : 1.0 + eps * (x - 1.0); }
inline double left(double eps, double x) { return 1.0 - right(eps, 1.0 - x); }
inline double step(double a, double b, double t) { return t <= 0.0 ? a : (t >= 1.0 ? b : a + (b - a) * t); }
inline double section(double eps, double s, int layer, double lambda)
{
    switch (layer)
    {
    case 0: return left(eps, s);
    case 1:
    case 4: return step(left(eps, s), right(eps, s), lambda);
    case 2: return step(right(eps, s), left(eps, s), 0.5 * lambda);
    case 3: return step(right(eps, s), left(eps, s), 0.5 * (1.0 + lambda));
    default: return right(eps, s);
    }
}
inline void map(double eps_y, double eps_z, double x, double y, double z, double &X, double &Y, double &Z)
{
    X = x;
    int layer = (int)(x * 6.0);
    if (layer > 5) layer = 5;
    if (layer < 0) layer = 0;
    const double lambda = (x - layer / 6.0) * 6.0;
    Y = section(eps_y, y, layer, lambda);
    Z = section(eps_z, z, layer, lambda);
}
} // namespace kershaw

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

// Move the GLL points of a 3-D mesh by a global map and recompute the six geometric factors isoparametrically at the mesh's
// own degree: J = dX/dr through D_hat along r, s, t; G = w_i w_j w_k |J| J^-1 J^-T in the order rr, ss, tt, rs, rt, st
// (domain.okl:47-49), quadrature weights folded in as the reference's files have them.  The reference-cube factor 1/2 per
// direction of the affine box is part of J here (r in [-1,1]).
template <typename DType, typename Map>
void deform_isoparametric(MeshData<DType> &m, Map map)
{
    const int n = m.poly_degree + 1;
    const long long n3 = (long long)n * n * n;
    std::vector<double> zg(n), wg(n), D((size_t)n * n);
    gll::zwgll(zg.data(), wg.data(), n);
    gll::dgll(D.data(), zg.data(), n);
    std::atomic<bool> bad(false);
    low_order::parallel_ranges(m.num_local_elements, low_order::range_parts(m.num_local_elements), [&](long long e0, long long e1, int) {
        std::vector<double> c[3];
        for (auto &v : c) v.resize((size_t)n3);
        for (long long e = e0; e < e1; e++)
        {
            const size_t base = (size_t)(e * n3);
            for (long long q = 0; q < n3; q++) map((double)m.x[base + q], (double)m.y[base + q], (double)m.z[base + q], c[0][q], c[1][q], c[2][q]);
            for (int k = 0; k < n; k++)
                for (int j = 0; j < n; j++)
                    for (int i = 0; i < n; i++)
                    {
                        const long long q = i + (long long)n * (j + (long long)n * k);
                        double J[3][3]; // J[a][b] = d X_a / d r_b
                        for (int a = 0; a < 3; a++)
                        {
                            double dr = 0.0, ds = 0.0, dt = 0.0;
                            for (int p = 0; p < n; p++)
                            {
                                dr += D[(size_t)i * n + p] * c[a][p + (long long)n * (j + (long long)n * k)];
                                ds += D[(size_t)j * n + p] * c[a][i + (long long)n * (p + (long long)n * k)];
                                dt += D[(size_t)k * n + p] * c[a][i + (long long)n * (j + (long long)n * p)];
                            }
                            J[a][0] = dr;
                            J[a][1] = ds;
                            J[a][2] = dt;
                        }
                        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
                        if (!(det > 0.0)) bad.store(true); // reported after the threads have joined
                        double I[3][3]; // I = J^-1 = d r / d X
                        I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
                        I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
                        I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
                        I[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
                        I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
                        I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
                        I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
                        I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
                        I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
                        const double sc = wg[i] * wg[j] * wg[k] * det;
                        static const int pa[6] = {0, 1, 2, 0, 0, 1}, pb[6] = {0, 1, 2, 1, 2, 2};
                        for (int f = 0; f < 6; f++)
                        {
                            const int a = pa[f], b = pb[f];
                            m.g[f][base + q] = (DType)(sc * (I[a][0] * I[b][0] + I[a][1] * I[b][1] + I[a][2] * I[b][2]));
                        }
                    }
            for (long long q = 0; q < n3; q++)
            {
                m.x[base + q] = (DType)c[0][q];
                m.y[base + q] = (DType)c[1][q];
                m.z[base + q] = (DType)c[2][q];
            }
        }
    });
    if (bad.load()) throw std::runtime_error("deformed box mesh: non-positive Jacobian at a GLL point");
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
    if (spec.deformed()) deform_isoparametric(m, [&](double x, double y, double z, double &X, double &Y, double &Z) { kershaw::map(spec.kershaw_eps[0], spec.kershaw_eps[1], x, y, z, X, Y, Z); });
    return m;
}

} // namespace fdd

#endif
