/*
 * gll.hpp -- Gauss-Lobatto-Legendre nodes, weights, derivative matrix and
 * Lagrange interpolants.  The reference takes these from Nek5000's Fortran
 * speclib (special_functions.f: ZWGLL :108, DGLL :781, HGLL :816, PNLEG :856,
 * PNDLEG :888); this is an independent C++ implementation of the same
 * formulas.  tests/test_cpu_oracle.py pins it against tables produced by the
 * reference's own Fortran (tests/golden/gll_tables.json).
 */
#ifndef FDD_GLL_HPP
#define FDD_GLL_HPP

#include <cmath>
#include <vector>

namespace fdd
{
namespace gll
{

// Legendre polynomial P_n(z), three-term recurrence
inline double pnleg(double z, int n)
{
    if (n == 0) return 1.0;
    double p1 = 1.0, p2 = z, p3 = z;
    for (int k = 1; k < n; k++)
    {
        p3 = ((2.0 * k + 1.0) * z * p2 - k * p1) / (k + 1.0);
        p1 = p2;
        p2 = p3;
    }
    return p3;
}

// P_n'(z)
inline double pndleg(double z, int n)
{
    if (n == 0) return 0.0;
    double p1 = 1.0, p2 = z, p1d = 0.0, p2d = 1.0, p3d = 1.0;
    for (int k = 1; k < n; k++)
    {
        double p3 = ((2.0 * k + 1.0) * z * p2 - k * p1) / (k + 1.0);
        p3d = ((2.0 * k + 1.0) * p2 + (2.0 * k + 1.0) * z * p2d - k * p1d) / (k + 1.0);
        p1 = p2;
        p2 = p3;
        p1d = p2d;
        p2d = p3d;
    }
    return p3d;
}

// np GLL nodes z (ascending in [-1,1]) and quadrature weights w
inline void zwgll(double *z, double *w, int np)
{
    const int N = np - 1;
    if (np == 1)
    {
        z[0] = 0.0;
        w[0] = 2.0;
        return;
    }
    z[0] = -1.0;
    z[N] = 1.0;
    // interior nodes: roots of P_N', Newton from the Chebyshev-Lobatto guess
    for (int k = 1; k <= N / 2; k++)
    {
        double x = -std::cos(M_PI * k / N);
        for (int it = 0; it < 100; it++)
        {
            double p = pnleg(x, N);
            double dp = pndleg(x, N);
            double ddp = (2.0 * x * dp - N * (N + 1.0) * p) / (1.0 - x * x);
            double dx = dp / ddp;
            x -= dx;
            if (std::fabs(dx) < 1e-16) break;
        }
        z[k] = x;
        z[N - k] = -x;
    }
    if (N % 2 == 0) z[N / 2] = 0.0;
    for (int k = 0; k <= N; k++)
    {
        double p = pnleg(z[k], N);
        w[k] = 2.0 / (N * (N + 1.0) * p * p);
    }
}

// D[i*n + j] = d l_j / d xi (xi_i): the row-major D_hat of domain.tpp:311-316
inline void dgll(double *D, const double *z, int n)
{
    const int N = n - 1;
    if (n == 1)
    {
        D[0] = 0.0;
        return;
    }
    const double d0 = N * (N + 1.0) / 4.0;
    for (int i = 0; i < n; i++)
    {
        for (int j = 0; j < n; j++)
        {
            double v = 0.0;
            if (i != j) v = pnleg(z[i], N) / (pnleg(z[j], N) * (z[i] - z[j]));
            if (i == j && i == 0) v = -d0;
            if (i == j && i == N) v = d0;
            D[i * n + j] = v;
        }
    }
}

// value at x of the Lagrange interpolant through the n GLL points z that is 1 at z[i]
inline double hgll(int i, double x, const double *z, int n)
{
    const double eps = 1.0e-5;
    if (std::fabs(x - z[i]) < eps) return 1.0;
    const int N = n - 1;
    const double alfan = N * (N + 1.0);
    return -(1.0 - x * x) * pndleg(x, N) / (alfan * pnleg(z[i], N) * (x - z[i]));
}

// coarse-to-fine interpolator J[i*n_c + j] = h^c_j(xi^f_i) (subdomain.tpp:153-159)
inline std::vector<double> interpolator(int N_c, int N_f)
{
    const int n_c = N_c + 1, n_f = N_f + 1;
    std::vector<double> zc(n_c), wc(n_c), zf(n_f), wf(n_f), J((size_t)n_f * n_c);
    zwgll(zc.data(), wc.data(), n_c);
    zwgll(zf.data(), wf.data(), n_f);
    for (int i = 0; i < n_f; i++)
        for (int j = 0; j < n_c; j++) J[(size_t)i * n_c + j] = hgll(j, zf[i], zc.data(), n_c);
    return J;
}

} // namespace gll
} // namespace fdd

#endif
