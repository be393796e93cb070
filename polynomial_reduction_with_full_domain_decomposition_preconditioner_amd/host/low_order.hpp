/*
 * low_order.hpp -- the operators the AMG V-cycle of the FDD preconditioner runs
 * on, built on the host once per Subdomain:
 *
 *   assemble_fem   the low-order FEM matrix on the GLL sub-cells: every cell
 *                  between adjacent GLL nodes of a spectral element is cut into
 *                  6 tetrahedra (the reference's vertex table) and the P1
 *                  stiffness of each tetrahedron is summed onto the dofs
 *                  (subdomain.tpp:2823-3060, 3300-3412; conforming region: one
 *                  degree, boolean Q).
 *   build          an algebraic multigrid hierarchy for it.  The reference hands
 *                  the matrix to HYPRE BoomerAMG (subdomain.tpp:3383-3549); HYPRE
 *                  is not available, so this is this build's own hierarchy
 *                  (documented deviation): smoothed aggregation with
 *                  Galerkin coarse operators, Chebyshev data per level in the
 *                  form the V-cycle consumes (diagonal scaling D = diag(A)^-1/2
 *                  and the coefficients of p(DAD), hypre's `ds` / `coefs`).
 *
 * Pure host code (setup is outside the hot path).
 */
#ifndef FDD_LOW_ORDER_HPP
#define FDD_LOW_ORDER_HPP

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "host_parallel.hpp"

namespace fdd
{
namespace low_order
{

struct HostCSR
{
    int rows = 0, cols = 0;
    std::vector<int> ptr;
    pod_vector<int> col; // resize() does not initialise (host_parallel.hpp)
    pod_vector<double> val;
    long long nnz() const { return (long long)col.size(); }
};

// rows of (row, col, val) triplets -> CSR with duplicates summed, columns sorted.  The triplets come in `parts` lists
// (one per producer thread, in order); every step runs on row ranges: a thread counts / places the triplets of ITS rows
// while scanning the lists in order, so a row's entries arrive in the order of the one concatenated list.
struct TripletParts
{
    std::vector<std::vector<int>> ti, tj;
    std::vector<std::vector<double>> tv;
    std::vector<int> row_min, row_max; // per list: the rows it touches (lists whose range misses a thread's rows are skipped)
    explicit TripletParts(int parts) : ti(parts), tj(parts), tv(parts), row_min(parts, 0), row_max(parts, -1) {}
    void set(int part, std::vector<int> &&i, std::vector<int> &&j, std::vector<double> &&v)
    {
        int lo = INT_MAX, hi = -1;
        for (int r : i)
        {
            lo = std::min(lo, r);
            hi = std::max(hi, r);
        }
        row_min[part] = lo;
        row_max[part] = hi;
        ti[part] = std::move(i);
        tj[part] = std::move(j);
        tv[part] = std::move(v);
    }
};

inline HostCSR from_triplets(int rows, int cols, TripletParts &T)
{
    HostCSR A;
    A.rows = rows;
    A.cols = cols;
    A.ptr.assign(rows + 1, 0);
    size_t n = 0;
    for (const std::vector<int> &v : T.ti) n += v.size();
    const int lists = (int)T.ti.size();
    const int parts = range_parts(rows);
    parallel_ranges(rows, parts, [&](long long r0, long long r1, int) {
        for (int l = 0; l < lists; l++)
        {
            if (T.row_max[l] < r0 or T.row_min[l] >= r1) continue;
            for (int r : T.ti[l])
                if (r >= r0 and r < r1) A.ptr[r + 1]++;
        }
    });
    for (int i = 0; i < rows; i++) A.ptr[i + 1] += A.ptr[i];
    pod_vector<int> cj(n);
    pod_vector<double> cv(n);
    parallel_ranges(rows, parts, [&](long long r0, long long r1, int) {
        std::vector<int> fill(A.ptr.begin() + r0, A.ptr.begin() + r1);
        for (int l = 0; l < lists; l++)
        {
            if (T.row_max[l] < r0 or T.row_min[l] >= r1) continue;
            const std::vector<int> &ti = T.ti[l], &tj = T.tj[l];
            const std::vector<double> &tv = T.tv[l];
            for (size_t t = 0; t < ti.size(); t++)
            {
                const int r = ti[t];
                if (r < r0 or r >= r1) continue;
                const int p = fill[r - r0]++;
                cj[p] = tj[t];
                cv[p] = tv[t];
            }
        }
    });
    T = TripletParts(0);
    // sort each row by column (stable: insertion order of equal columns is kept) and merge duplicates;
    // row ranges in parallel, pieces concatenated in row order
    std::vector<std::vector<int>> pcol(parts);
    std::vector<std::vector<double>> pval(parts);
    std::vector<int> row_len(rows, 0);
    parallel_ranges(rows, parts, [&](long long r0, long long r1, int part) {
        std::vector<int> order;
        std::vector<int> oc; // thread-local while growing (the headers in pcol / pval share cache lines), moved out at the end
        std::vector<double> ov;
        oc.reserve((size_t)(A.ptr[r1] - A.ptr[r0]));
        ov.reserve((size_t)(A.ptr[r1] - A.ptr[r0]));
        for (long long i = r0; i < r1; i++)
        {
            const int a = A.ptr[i], b = A.ptr[i + 1];
            const bool short_row = (b - a) <= 64; // the usual case (a few entries per element): insertion sort in place, stable too
            if (short_row)
            {
                for (int k = a + 1; k < b; k++)
                {
                    const int c = cj[k];
                    const double v = cv[k];
                    int q = k - 1;
                    for (; q >= a and cj[q] > c; q--)
                    {
                        cj[q + 1] = cj[q];
                        cv[q + 1] = cv[q];
                    }
                    cj[q + 1] = c;
                    cv[q + 1] = v;
                }
            }
            else
            {
                order.resize(b - a);
                for (int k = 0; k < b - a; k++) order[k] = a + k;
                std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return cj[p] < cj[q]; });
            }
            int last = -1;
            const size_t before = oc.size();
            for (int k = 0; k < b - a; k++)
            {
                const int p = short_row ? a + k : order[k];
                if (cj[p] != last)
                {
                    oc.push_back(cj[p]);
                    ov.push_back(cv[p]);
                    last = cj[p];
                }
                else
                    ov.back() += cv[p];
            }
            row_len[i] = (int)(oc.size() - before);
        }
        pcol[part] = std::move(oc);
        pval[part] = std::move(ov);
    });
    A.ptr[0] = 0; // the old pointers (into cj / cv) are no longer needed
    for (int i = 0; i < rows; i++) A.ptr[i + 1] = A.ptr[i] + row_len[i];
    concatenate(A.col, pcol);
    concatenate(A.val, pval);
    return A;
}

inline HostCSR from_triplets(int rows, int cols, std::vector<int> &ti, std::vector<int> &tj, std::vector<double> &tv)
{
    TripletParts T(1);
    T.set(0, std::move(ti), std::move(tj), std::move(tv));
    return from_triplets(rows, cols, T);
}

// subdomain.tpp:2823-3060: P1 stiffness of the 6 tetrahedra of every GLL sub-cell, summed on the dofs.
// x, y, z: coordinates of the level-0 points (element-major, x fastest); point_dof[p] < 0: no dof (Dirichlet).
inline HostCSR assemble_fem(const double *x, const double *y, const double *z, const int *point_dof, int num_dofs, int poly_degree, int num_elements, double epsilon = 1.0e-12)
{
    // vertex offsets (i, j, k) of the 6 tetrahedra of a cell (subdomain.tpp:2878-2885)
    static const int tets[6][4][3] = {
        {{0, 0, 0}, {0, 1, 0}, {1, 0, 0}, {1, 0, 1}}, {{1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 1}}, {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 0, 1}},
        {{1, 0, 1}, {1, 1, 0}, {1, 1, 1}, {0, 1, 0}}, {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 1, 0}}, {{1, 0, 1}, {1, 1, 1}, {0, 1, 1}, {0, 1, 0}}};
    // gradients of the 4 P1 shape functions on the reference tetrahedron (subdomain.tpp:2841-2843: one row per direction)
    static const double Dref[3][4] = {{1.0, 0.0, 0.0, -1.0}, {0.0, 1.0, 0.0, -1.0}, {0.0, 0.0, 1.0, -1.0}};

    const int N = poly_degree, n = N + 1, n3 = n * n * n;
    // Element ranges in parallel.  The tetrahedra of one element are first summed into the element's own 27-neighbour
    // stencil (a vertex pair of a cell differs by at most one step per direction) and only then emitted as triplets:
    // 6 * 16 entries per cell would be 10^9 triplets (17 GB) at 32^3 elements of degree 7.
    const int parts = range_parts(num_elements);
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    TripletParts T(parts);
    parallel_ranges(num_elements, parts, [&](long long e0, long long e1, int part) {
    std::vector<int> ti, tj; // thread-local while growing, moved out at the end
    std::vector<double> tv;
    std::vector<double> K((size_t)n3 * 27);
    std::vector<unsigned char> touched((size_t)n3 * 27);
    const size_t guess = (size_t)(e1 - e0) * n3 * 8; // 7-point rows on a rectilinear mesh, 15 at most on a deformed one
    ti.reserve(guess);
    tj.reserve(guess);
    tv.reserve(guess);
    for (long long e = e0; e < e1; e++)
    {
        const size_t base = (size_t)e * n3;
        std::fill(K.begin(), K.end(), 0.0);
        std::fill(touched.begin(), touched.end(), (unsigned char)0);
        for (int sz = 0; sz < N; sz++)
            for (int sy = 0; sy < N; sy++)
                for (int sx = 0; sx < N; sx++)
                    for (int t = 0; t < 6; t++)
                    {
                        size_t loc[4];
                        double xs[4], ys[4], zs[4];
                        for (int v = 0; v < 4; v++)
                        {
                            loc[v] = base + (size_t)(sx + tets[t][v][0]) + (size_t)(sy + tets[t][v][1]) * n + (size_t)(sz + tets[t][v][2]) * n * n;
                            xs[v] = x[loc[v]];
                            ys[v] = y[loc[v]];
                            zs[v] = z[loc[v]];
                        }
                        // H = [x_v - x_3], inverse and determinant (subdomain.tpp:2975-2989, 2788-2817)
                        const double H[9] = {xs[0] - xs[3], xs[1] - xs[3], xs[2] - xs[3], ys[0] - ys[3], ys[1] - ys[3], ys[2] - ys[3], zs[0] - zs[3], zs[1] - zs[3], zs[2] - zs[3]};
                        const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
                        const double id = 1.0 / det;
                        const double iH[9] = {id * (H[4] * H[8] - H[7] * H[5]), id * (H[2] * H[7] - H[8] * H[1]), id * (H[1] * H[5] - H[4] * H[2]),
                                              id * (H[5] * H[6] - H[8] * H[3]), id * (H[0] * H[8] - H[6] * H[2]), id * (H[2] * H[3] - H[5] * H[0]),
                                              id * (H[3] * H[7] - H[6] * H[4]), id * (H[1] * H[6] - H[7] * H[0]), id * (H[0] * H[4] - H[3] * H[1])};
                        // G_mn = (det / 24) sum_k iH[m][k] iH[n][k] at each of the 4 quadrature points (subdomain.tpp:2994-3007)
                        double G[3][3];
                        for (int m = 0; m < 3; m++)
                            for (int nn = 0; nn < 3; nn++)
                            {
                                double g = 0.0;
                                for (int k = 0; k < 3; k++) g += (det / 24.0) * iH[m * 3 + k] * iH[nn * 3 + k];
                                G[m][nn] = g;
                            }
                        // A_tet = sum_mn D_m^T G_mn D_n over the 4 (identical) quadrature points (subdomain.tpp:3009-3027)
                        // Dref's entries are 0 and +-1 (row m: +1 in column m, -1 in column 3): the terms with a zero factor
                        // add +-0 and are skipped, the others are +-G_mn exactly, added in the reference's order (m, n, q)
                        double At[4][4];
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 4; j++)
                            {
                                double a = 0.0;
                                const int m0 = (i < 3) ? i : 0, m1 = (i < 3) ? i + 1 : 3, n0 = (j < 3) ? j : 0, n1 = (j < 3) ? j + 1 : 3;
                                const bool minus = (i < 3) != (j < 3);
                                for (int m = m0; m < m1; m++)
                                    for (int nn = n0; nn < n1; nn++)
                                    {
                                        const double g = minus ? -G[m][nn] : G[m][nn];
                                        for (int q = 0; q < 4; q++) a += g;
                                    }
                                At[i][j] = a;
                            }
                        for (int i = 0; i < 4; i++)
                        {
                            if (point_dof[loc[i]] < 0) continue;
                            const int li = (int)(loc[i] - base);
                            for (int j = 0; j < 4; j++)
                            {
                                if (point_dof[loc[j]] < 0 or not(std::abs(At[i][j]) > epsilon)) continue; // :3031
                                const int d0 = tets[t][j][0] - tets[t][i][0], d1 = tets[t][j][1] - tets[t][i][1], d2 = tets[t][j][2] - tets[t][i][2];
                                const size_t slot = (size_t)li * 27 + (size_t)((d0 + 1) + 3 * (d1 + 1) + 9 * (d2 + 1));
                                K[slot] += At[i][j];
                                touched[slot] = 1;
                            }
                        }
                    }
        // the element's merged entries, row by row, neighbours in ascending local index
        for (int li = 0; li < n3; li++)
        {
            const int di = point_dof[base + li];
            if (di < 0) continue;
            for (int s = 0; s < 27; s++)
            {
                if (not touched[(size_t)li * 27 + s]) continue;
                const int lj = li + (s % 3 - 1) + ((s / 3) % 3 - 1) * n + (s / 9 - 1) * n * n;
                ti.push_back(di);
                tj.push_back(point_dof[base + lj]);
                tv.push_back(K[(size_t)li * 27 + s]);
            }
        }
    }
    T.set(part, std::move(ti), std::move(tj), std::move(tv));
    });
    static const bool timing = getenv("FDD_SETUP_TIMING") != nullptr;
    const auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t1 = clock();
    HostCSR A = from_triplets(num_dofs, num_dofs, T);
    if (timing) printf("low_order: FEM matrix: element stencils %.3f s, rows from triplets %.3f s\n", t1 - t0, clock() - t1);
    return A;
}

// C = A * B (Gustavson, one dense accumulator row)
inline HostCSR multiply(const HostCSR &A, const HostCSR &B)
{
    HostCSR C;
    C.rows = A.rows;
    C.cols = B.cols;
    C.ptr.assign(A.rows + 1, 0);
    // row ranges in parallel (each with its own accumulator row), pieces concatenated in row order
    const int parts = range_parts(A.rows);
    std::vector<std::vector<int>> pcol(parts);
    std::vector<std::vector<double>> pval(parts);
    std::vector<int> row_len(A.rows, 0);
    parallel_ranges(A.rows, parts, [&](long long r0, long long r1, int part) {
        std::vector<long long> marker(B.cols, -1);
        std::vector<double> acc(B.cols, 0.0);
        std::vector<int> cols;
        std::vector<int> oc; // thread-local while growing, moved out at the end
        std::vector<double> ov;
        for (long long i = r0; i < r1; i++)
        {
            cols.clear();
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            {
                const int k = A.col[p];
                const double a = A.val[p];
                for (int q = B.ptr[k]; q < B.ptr[k + 1]; q++)
                {
                    const int j = B.col[q];
                    if (marker[j] != i)
                    {
                        marker[j] = i;
                        acc[j] = 0.0;
                        cols.push_back(j);
                    }
                    acc[j] += a * B.val[q];
                }
            }
            std::sort(cols.begin(), cols.end());
            for (int j : cols)
            {
                oc.push_back(j);
                ov.push_back(acc[j]);
            }
            row_len[i] = (int)cols.size();
        }
        pcol[part] = std::move(oc);
        pval[part] = std::move(ov);
    });
    for (int i = 0; i < A.rows; i++) C.ptr[i + 1] = C.ptr[i] + row_len[i];
    concatenate(C.col, pcol);
    concatenate(C.val, pval);
    return C;
}

// T = A^T.  Ranges of T's rows (A's columns) in parallel: a thread scans A in row order and places the entries of ITS
// columns, so every row of T comes out in ascending column order, as from the serial loop.
inline HostCSR transpose(const HostCSR &A)
{
    HostCSR T;
    T.rows = A.cols;
    T.cols = A.rows;
    T.ptr.assign((size_t)T.rows + 1, 0);
    const int parts = range_parts(T.rows);
    parallel_ranges(T.rows, parts, [&](long long c0, long long c1, int) {
        for (int c : A.col)
            if (c >= c0 and c < c1) T.ptr[c + 1]++;
    });
    for (int i = 0; i < T.rows; i++) T.ptr[i + 1] += T.ptr[i];
    T.col.resize(A.col.size());
    T.val.resize(A.val.size());
    parallel_ranges(T.rows, parts, [&](long long c0, long long c1, int) {
        std::vector<int> fill(T.ptr.begin() + c0, T.ptr.begin() + c1);
        for (int i = 0; i < A.rows; i++)
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            {
                const int c = A.col[p];
                if (c < c0 or c >= c1) continue;
                const int q = fill[c - c0]++;
                T.col[q] = i;
                T.val[q] = A.val[p];
            }
    });
    return T;
}

inline std::vector<double> diagonal(const HostCSR &A)
{
    std::vector<double> d(A.rows, 0.0);
    parallel_ranges(A.rows, range_parts(A.rows), [&](long long r0, long long r1, int) {
        for (long long i = r0; i < r1; i++)
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
                if (A.col[p] == i) d[i] = A.val[p];
    });
    return d;
}

inline void spmv(std::vector<double> &y, const HostCSR &A, const std::vector<double> &x)
{
    y.resize(A.rows);
    parallel_ranges(A.rows, range_parts(A.rows), [&](long long r0, long long r1, int) {
        for (long long i = r0; i < r1; i++)
        {
            double s = 0.0;
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++) s += A.val[p] * x[A.col[p]];
            y[i] = s;
        }
    });
}

// the power iteration's start vector: deterministic, no symmetry
inline std::vector<double> power_iteration_start(int n)
{
    std::vector<double> v(n);
    parallel_ranges(n, range_parts(n), [&](long long i0, long long i1, int) {
        for (long long i = i0; i < i1; i++) v[i] = 1.0 + 0.5 * std::sin(0.7 * (int)i + 0.3);
    });
    return v;
}

// largest eigenvalue of D A D, D = diag(A)^-1/2, by power iteration from a fixed start
inline double max_eigenvalue_scaled(const HostCSR &A, const std::vector<double> &D, int iterations)
{
    const int n = A.rows;
    std::vector<double> v = power_iteration_start(n), w(n), t(n);
    // the vector loops run on the host threads; sums are formed over fixed chunks of 65536 entries and the chunk sums
    // added in order, whatever the thread count: the hierarchy does not depend on the number of cores
    const int parts = range_parts(n);
    const long long chunk = 65536, chunks = (n + chunk - 1) / chunk;
    std::vector<double> partial((size_t)std::max(chunks, 1LL));
    auto reduce = [&](const std::function<double(long long, long long)> &f) {
        std::fill(partial.begin(), partial.end(), 0.0);
        parallel_ranges(chunks, (chunks < 2 * parts) ? 1 : parts, [&](long long c0, long long c1, int) {
            for (long long c = c0; c < c1; c++) partial[c] = f(c * chunk, std::min((long long)n, (c + 1) * chunk));
        });
        double s = 0.0;
        for (double x : partial) s += x;
        return s;
    };
    double lambda = 1.0;
    for (int it = 0; it < iterations; it++)
    {
        const double nrm = std::sqrt(reduce([&](long long i0, long long i1) {
            double s = 0.0;
            for (long long i = i0; i < i1; i++) s += v[i] * v[i];
            return s;
        }));
        parallel_ranges(n, parts, [&](long long i0, long long i1, int) {
            for (long long i = i0; i < i1; i++) t[i] = D[i] * (v[i] / nrm);
        });
        spmv(w, A, t);
        lambda = reduce([&](long long i0, long long i1) {
            double s = 0.0;
            for (long long i = i0; i < i1; i++)
            {
                w[i] *= D[i];
                s += w[i] * (v[i] / nrm);
            }
            return s;
        });
        v.swap(w);
    }
    return lambda;
}

struct Options
{
    int cheby_order = 2;          // subdomain.hpp:237
    int max_levels = 12;
    int coarsest_size = 400;      // stop coarsening at or below this many rows (solved with a dense inverse)
    double strength = 0.08;       // |a_ij| >= strength * sqrt(a_ii a_jj)
    double eig_ratio = 0.3;       // smoother targets [eig_ratio, 1.1] * lambda_max (hypre's Chebyshev defaults)
    double upper_factor = 1.1;
    int power_iterations = 25;
    // lambda_max(D A D) of a level by `power_iterations` steps of the power iteration from the fixed start vector below;
    // unset: on the host threads (max_eigenvalue_scaled).  Subdomain::amg_build runs the same iteration on the device.
    std::function<double(const HostCSR &, const std::vector<double> &, int)> lambda_max;
    bool smooth_prolongator = true;
    double geometric_eig_ratio = 0.15; // lower end of the Chebyshev smoother's interval, as a fraction of lambda_max, on the levels coarsened on the lattice
    int geometric_min_nodes = 5;  // lattices with fewer nodes per direction and element are left to the aggregation
    bool geometric_levels = true; // with a Lattice: the leading levels coarsen the GLL lattice itself (see geometric_level); false: aggregation from level 0
    int double_aggregation_levels = 0; // > 0: the finest levels aggregate TWICE (aggregates of aggregates, through the tentative Galerkin graph)
                                       // before the interpolator is smoothed: ~70 rows per aggregate on a 7-point stencil instead of ~6, operator
                                       // complexity 2.4 -> 1.1, V-cycle about half the time -- and 5 outer iterations instead of 3 (10^3 elements,
                                       // N = 7): measured, not the default
};

// What a matrix-free application of a geometric level's interpolator needs (fdd_lattice_prolong / _restrict): filled by
// geometric_level where every lattice point is a conforming dof's or a Dirichlet point (no hanging rows, no dofs outside
// the lattice: a rank's own conforming region), 3-D
struct Transfer
{
    int n = 0, m = 0;        // lattice nodes per direction and element: fine, kept
    std::vector<int> lo, hi; // fine node i between kept nodes lo[i] <= hi[i]
    std::vector<double> wl;  // weight of lo[i] (hi[i]: 1 - wl[i])
    long long num_elements = 0;
    std::vector<int> owner_dof;  // per fine lattice point: its dof where the point is the dof's first, -1 elsewhere
    std::vector<int> coarse_dof; // per kept node of every element: its coarse dof, -1 on a Dirichlet node
    bool active() const { return n > 0; }
};

struct Level
{
    HostCSR A, P; // P empty on the coarsest level
    std::vector<double> D, coefs;
    Transfer transfer; // active: P is this lattice interpolation
};

// entries below drop * (largest magnitude of the row) removed, the others scaled so that the row keeps its sum
inline HostCSR filter_rows(const HostCSR &P, double drop)
{
    HostCSR F;
    F.rows = P.rows;
    F.cols = P.cols;
    F.ptr.assign(P.rows + 1, 0);
    for (int i = 0; i < P.rows; i++)
    {
        double big = 0.0, sum = 0.0, kept = 0.0;
        for (int p = P.ptr[i]; p < P.ptr[i + 1]; p++)
        {
            big = std::max(big, std::abs(P.val[p]));
            sum += P.val[p];
        }
        const size_t first = F.col.size();
        for (int p = P.ptr[i]; p < P.ptr[i + 1]; p++)
            if (std::abs(P.val[p]) >= drop * big)
            {
                F.col.push_back(P.col[p]);
                F.val.push_back(P.val[p]);
                kept += P.val[p];
            }
        if (kept != 0.0 and sum != 0.0)
            for (size_t q = first; q < F.val.size(); q++) F.val[q] *= sum / kept;
        F.ptr[i + 1] = (int)F.col.size();
    }
    return F;
}

// greedy aggregation on the strength graph: returns the aggregate of every row and the number of aggregates
inline int aggregate(const HostCSR &A, double theta, std::vector<int> &agg)
{
    const int n = A.rows;
    const std::vector<double> d = diagonal(A);
    agg.assign(n, -1);
    auto strong = [&](int i, int p) {
        const int j = A.col[p];
        return j != i and std::abs(A.val[p]) >= theta * std::sqrt(std::abs(d[i] * d[j]));
    };
    int count = 0;
    // pass 1: a row whose strong neighbourhood is untouched seeds an aggregate with all of it
    for (int i = 0; i < n; i++)
    {
        if (agg[i] != -1) continue;
        bool free_nbhd = true;
        bool has_strong = false;
        for (int p = A.ptr[i]; p < A.ptr[i + 1] and free_nbhd; p++)
            if (strong(i, p))
            {
                has_strong = true;
                if (agg[A.col[p]] != -1) free_nbhd = false;
            }
        if (not free_nbhd or not has_strong) continue;
        agg[i] = count;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (strong(i, p)) agg[A.col[p]] = count;
        count++;
    }
    // pass 2: leftovers join the aggregate of their strongest aggregated neighbour
    std::vector<int> joined(n, -1);
    for (int i = 0; i < n; i++)
    {
        if (agg[i] != -1) continue;
        double best = 0.0;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (strong(i, p) and agg[A.col[p]] != -1 and std::abs(A.val[p]) > best)
            {
                best = std::abs(A.val[p]);
                joined[i] = agg[A.col[p]];
            }
    }
    for (int i = 0; i < n; i++)
        if (agg[i] == -1 and joined[i] != -1) agg[i] = joined[i];
    // pass 3: what is still alone (isolated rows) forms aggregates with its unaggregated strong neighbours
    for (int i = 0; i < n; i++)
    {
        if (agg[i] != -1) continue;
        agg[i] = count;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (strong(i, p) and agg[A.col[p]] == -1) agg[A.col[p]] = count;
        count++;
    }
    return count;
}

inline std::vector<double> chebyshev_coefficients(double lambda_max, const Options &o)
{
    // p(x) ~ 1/x on [a, b] = [eig_ratio, upper_factor] * lambda_max: `order` steps of the Chebyshev iteration
    // written as a polynomial in x, coefs[k] multiplying x^k (the form of subdomain.tpp:45-67)
    const double a = o.eig_ratio * lambda_max, b = o.upper_factor * lambda_max;
    const double theta = 0.5 * (a + b), delta = 0.5 * (b - a);
    if (o.cheby_order == 1) return {1.0 / theta};
    if (o.cheby_order == 2)
    {
        const double den = 2.0 * theta * theta - delta * delta;
        return {4.0 * theta / den, -2.0 / den};
    }
    // general order: expand p(x) = (1 - T_m((theta - x)/delta) / T_m(theta/delta)) / x in the monomial basis
    const int m = o.cheby_order;
    std::vector<std::vector<double>> T(m + 1); // T_k(t) coefficients in t
    T[0] = {1.0};
    T[1] = {0.0, 1.0};
    for (int k = 2; k <= m; k++)
    {
        T[k].assign(k + 1, 0.0);
        for (int i = 0; i < (int)T[k - 1].size(); i++) T[k][i + 1] += 2.0 * T[k - 1][i];
        for (int i = 0; i < (int)T[k - 2].size(); i++) T[k][i] -= T[k - 2][i];
    }
    // q(x) = T_m((theta - x)/delta): substitute t = theta/delta - x/delta
    std::vector<double> q(m + 1, 0.0), pw(1, 1.0); // pw = t^i as a polynomial in x
    const double c0 = theta / delta, c1 = -1.0 / delta;
    for (int i = 0; i <= m; i++)
    {
        for (int k = 0; k < (int)pw.size(); k++) q[k] += T[m][i] * pw[k];
        std::vector<double> nx(pw.size() + 1, 0.0);
        for (int k = 0; k < (int)pw.size(); k++)
        {
            nx[k] += c0 * pw[k];
            nx[k + 1] += c1 * pw[k];
        }
        pw.swap(nx);
    }
    double Tm0 = 0.0, tp = 1.0;
    for (int i = 0; i <= m; i++)
    {
        Tm0 += T[m][i] * tp;
        tp *= c0;
    }
    std::vector<double> coefs(m);
    for (int k = 0; k < m; k++) coefs[k] = -q[k + 1] / Tm0; // (1 - q(x)/Tm0)/x, q(0) = Tm0
    return coefs;
}

// ---------------------------------------------------------------------------------------------------------------
// Geometric leading levels.  The low-order matrix lives on a LATTICE: the (N+1)^dim GLL points of every degree-N
// element, joined across element faces.  Its leading levels are coarsened on that lattice, as a geometric multigrid
// would: per element and direction a symmetric subset of the nodes is kept (end points always: N = 7 keeps nodes
// {0, 3, 4, 7} of the 8, then {0, 7}, i.e. the degree-1 mesh), interpolation is multi-linear in the element's
// reference coordinates, coarse operators are Galerkin products.  Rows are short (at most 2^dim entries), coarse
// operators keep the 27-point shape, the operator complexity of the hierarchy is ~1.3 where smoothed aggregation of
// the 7-point level-0 matrix gives 2.4 (each of its 7-row aggregates reaches its second neighbours through the
// smoothed interpolator: ~48 entries per row on level 1).  Below the degree-1 lattice the aggregation takes over.
// (The reference asks HYPRE BoomerAMG for its hierarchy, subdomain.tpp:3383-3549; this is this build's stand-in.)
// ---------------------------------------------------------------------------------------------------------------
struct Lattice
{
    int dim = 3;
    int n = 0;               // lattice points per direction and element
    std::vector<double> ref; // their reference coordinates in [-1, 1], ascending
    long long num_elements = 0;
    // rows: (num_elements * n^dim) lattice points x dofs; a conforming point has one unit entry, a Dirichlet point none,
    // a hanging point (face / edge of a degree-N element against a lower-degree neighbour) its constraint row
    HostCSR rows;
    bool active(int min_nodes = 3) const { return n >= min_nodes and num_elements > 0; }
};

// the nodes a coarser lattice keeps, of n with reference coordinates ref: both ends, symmetric, about half of them
// (an even count when n is even), those nearest to equispaced targets
inline std::vector<int> coarse_nodes(int n, const std::vector<double> &ref)
{
    if (n <= 4) return {0, n - 1};
    int m = (n + 1) / 2;
    if (n % 2 == 0 and m % 2 == 1) m++;
    std::vector<char> taken(n, 0);
    std::vector<int> keep;
    for (int q = 0; q < (m + 1) / 2; q++)
    {
        static const int gll_targets = getenv("FDD_TUNE_AMG_GLL_TARGETS") ? atoi(getenv("FDD_TUNE_AMG_GLL_TARGETS")) : 0; // development: 1 = Chebyshev-like targets
        const double target = gll_targets ? -std::cos(3.14159265358979323846 * q / (m - 1)) : -1.0 + 2.0 * q / (m - 1);
        int best = -1;
        for (int i = 0; i <= (n - 1) / 2; i++)
            if (not taken[i] and (best < 0 or std::abs(ref[i] - target) < std::abs(ref[best] - target) - 1e-14)) best = i;
        taken[best] = taken[n - 1 - best] = 1;
    }
    for (int i = 0; i < n; i++)
        if (taken[i]) keep.push_back(i);
    return keep;
}

// One geometric level: interpolator P (dofs x coarse dofs) and the coarse lattice.  Coarse dofs = the dofs sitting on
// kept lattice nodes, plus every dof that is not a conforming lattice dof at all (lower-degree ring elements, superdomain
// dofs of a composite: carried through unchanged), numbered in the order of the fine dofs.
inline HostCSR geometric_level(const Lattice &fine, int num_dofs, Lattice &coarse, Transfer *transfer = nullptr)
{
    const int dim = fine.dim, n = fine.n;
    const std::vector<int> keep = coarse_nodes(n, fine.ref);
    const int m = (int)keep.size();
    long long np = 1, npc = 1;
    for (int d = 0; d < dim; d++)
    {
        np *= n;
        npc *= m;
    }
    // 1-D interpolation: node i from kept nodes lo[i], hi[i] with weights wl[i], 1 - wl[i] (a kept node from itself)
    std::vector<int> lo(n), hi(n), pos(n, -1);
    std::vector<double> wl(n);
    for (int a = 0; a < m; a++) pos[keep[a]] = a;
    for (int i = 0, a = 0; i < n; i++)
    {
        if (pos[i] >= 0)
        {
            lo[i] = hi[i] = pos[i];
            wl[i] = 1.0;
            a = pos[i];
            continue;
        }
        lo[i] = a;
        hi[i] = a + 1;
        wl[i] = (fine.ref[keep[a + 1]] - fine.ref[i]) / (fine.ref[keep[a + 1]] - fine.ref[keep[a]]);
    }
    const long long total = fine.num_elements * np;
    // where every conforming dof first sits, and whether some occurrence is on a kept node
    std::vector<long long> first(num_dofs, -1);
    std::vector<char> kept(num_dofs, 0);
    std::vector<char> node_kept((size_t)np, 0); // lattice node of an element: kept in every direction
    for (long long v0 = 0; v0 < np; v0++)
    {
        long long v = v0;
        bool all = true;
        for (int a = 0; a < dim; a++)
        {
            all = all and pos[v % n] >= 0;
            v /= n;
        }
        node_kept[v0] = all;
    }
    // ranges of dofs in parallel, each thread scanning the points in order for ITS dofs (first = the serial loop's)
    parallel_ranges(num_dofs, range_parts(num_dofs), [&](long long d0, long long d1, int) {
        const int *rp = fine.rows.ptr.data(), *rc = fine.rows.col.data();
        const double *rv = fine.rows.val.data();
        for (long long q = 0; q < total; q++)
        {
            const int t = rp[q];
            if (rp[q + 1] - t != 1) continue;
            const int d = rc[t];
            if (d < d0 or d >= d1 or rv[t] != 1.0) continue;
            if (first[d] < 0) first[d] = q;
            if (node_kept[q % np]) kept[d] = 1;
        }
    });
    std::vector<int> cmap(num_dofs, -1);
    int nc = 0;
    for (int d = 0; d < num_dofs; d++)
        if (first[d] < 0 or kept[d]) cmap[d] = nc++;

    HostCSR P;
    P.rows = num_dofs;
    P.cols = nc;
    P.ptr.assign(num_dofs + 1, 0);
    const int parts = range_parts(num_dofs);
    std::vector<std::vector<int>> pcol(parts);
    std::vector<std::vector<double>> pval(parts);
    std::vector<int> row_len(num_dofs, 0);
    parallel_ranges(num_dofs, parts, [&](long long d0, long long d1, int part) {
        std::vector<int> oc;
        std::vector<double> ov;
        std::vector<std::pair<int, double>> row;
        for (long long d = d0; d < d1; d++)
        {
            if (cmap[d] >= 0)
            {
                oc.push_back(cmap[d]);
                ov.push_back(1.0);
                row_len[d] = 1;
                continue;
            }
            const long long q = first[d], e = q / np;
            long long v = q % np;
            int idx[3] = {0, 0, 0};
            for (int a = 0; a < dim; a++)
            {
                idx[a] = (int)(v % n);
                v /= n;
            }
            row.clear();
            for (int corner = 0; corner < (1 << dim); corner++)
            {
                double w = 1.0;
                long long cq = 0, stride = 1;
                bool skip = false;
                for (int a = 0; a < dim; a++)
                {
                    const int side = (corner >> a) & 1, i = idx[a];
                    if (lo[i] == hi[i])
                    {
                        if (side) skip = true; // a kept coordinate has one parent in this direction
                        cq += (long long)keep[lo[i]] * stride;
                    }
                    else
                    {
                        w *= side ? 1.0 - wl[i] : wl[i];
                        cq += (long long)keep[side ? hi[i] : lo[i]] * stride;
                    }
                    stride *= n;
                }
                if (skip) continue;
                const long long point = e * np + cq;
                for (int t = fine.rows.ptr[point]; t < fine.rows.ptr[point + 1]; t++)
                {
                    const int c = cmap[fine.rows.col[t]];
                    if (c < 0)
                    {
                        fprintf(stderr, "ERROR: low_order::geometric_level: a kept lattice node depends on dof %d, which is not kept\n", fine.rows.col[t]);
                        exit(EXIT_FAILURE);
                    }
                    row.emplace_back(c, w * fine.rows.val[t]);
                }
            }
            std::sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
            int len = 0;
            for (size_t t = 0; t < row.size(); t++)
            {
                if (t > 0 and row[t].first == row[t - 1].first)
                    ov.back() += row[t].second;
                else
                {
                    oc.push_back(row[t].first);
                    ov.push_back(row[t].second);
                    len++;
                }
            }
            row_len[d] = len;
        }
        pcol[part] = std::move(oc);
        pval[part] = std::move(ov);
    });
    for (int d = 0; d < num_dofs; d++) P.ptr[d + 1] = P.ptr[d] + row_len[d];
    concatenate(P.col, pcol);
    concatenate(P.val, pval);

    // the coarse lattice: the kept nodes of every element, their rows over the coarse dofs
    coarse = Lattice();
    coarse.dim = dim;
    coarse.n = m;
    coarse.num_elements = fine.num_elements;
    coarse.ref.resize(m);
    for (int a = 0; a < m; a++) coarse.ref[a] = fine.ref[keep[a]];
    HostCSR &R = coarse.rows;
    R.rows = (int)(fine.num_elements * npc);
    R.cols = nc;
    R.ptr.assign((size_t)R.rows + 1, 0);
    std::vector<long long> fine_node((size_t)npc); // kept node cv of an element -> its fine lattice node
    for (long long cv = 0; cv < npc; cv++)
    {
        long long v = cv, fq = 0, stride = 1;
        for (int a = 0; a < dim; a++)
        {
            fq += (long long)keep[v % m] * stride;
            v /= m;
            stride *= n;
        }
        fine_node[cv] = fq;
    }
    const int eparts = range_parts(fine.num_elements);
    parallel_ranges(fine.num_elements, eparts, [&](long long e0, long long e1, int) {
        for (long long e = e0; e < e1; e++)
            for (long long cv = 0; cv < npc; cv++)
            {
                const long long point = e * np + fine_node[cv];
                R.ptr[e * npc + cv + 1] = fine.rows.ptr[point + 1] - fine.rows.ptr[point];
            }
    });
    for (long long q = 0; q < R.rows; q++) R.ptr[q + 1] += R.ptr[q];
    R.col.resize((size_t)R.ptr[R.rows]);
    R.val.resize((size_t)R.ptr[R.rows]);
    parallel_ranges(fine.num_elements, eparts, [&](long long e0, long long e1, int) {
        for (long long e = e0; e < e1; e++)
            for (long long cv = 0; cv < npc; cv++)
            {
                const long long point = e * np + fine_node[cv];
                int out = R.ptr[e * npc + cv];
                for (int t = fine.rows.ptr[point]; t < fine.rows.ptr[point + 1]; t++, out++)
                {
                    R.col[out] = cmap[fine.rows.col[t]];
                    R.val[out] = fine.rows.val[t];
                }
            }
    });
    // the maps of the matrix-free form, where this interpolator is nothing but the lattice interpolation
    if (transfer != nullptr)
    {
        *transfer = Transfer();
        bool plain = (dim == 3) and total < (1LL << 31);
        for (int d = 0; plain and d < num_dofs; d++) plain = first[d] >= 0;
        for (long long q = 0; plain and q < total; q++)
        {
            const int len = fine.rows.ptr[q + 1] - fine.rows.ptr[q];
            plain = len == 0 or (len == 1 and fine.rows.val[fine.rows.ptr[q]] == 1.0);
        }
        for (long long q = 0; plain and q < R.rows; q++)
        {
            const int len = R.ptr[q + 1] - R.ptr[q];
            plain = len == 0 or (len == 1 and R.val[R.ptr[q]] == 1.0);
        }
        if (plain)
        {
            Transfer &T = *transfer;
            T.n = n;
            T.m = m;
            T.lo = lo;
            T.hi = hi;
            T.wl = wl;
            T.num_elements = fine.num_elements;
            T.owner_dof.assign((size_t)total, -1);
            for (int d = 0; d < num_dofs; d++) T.owner_dof[(size_t)first[d]] = d;
            T.coarse_dof.assign((size_t)R.rows, -1);
            parallel_ranges(R.rows, range_parts(R.rows), [&](long long q0, long long q1, int) {
                for (long long q = q0; q < q1; q++)
                    if (R.ptr[q + 1] > R.ptr[q]) T.coarse_dof[(size_t)q] = R.col[R.ptr[q]];
            });
        }
    }
    return P;
}

inline std::vector<Level> build(HostCSR A0, const Options &o, bool verbose = false, Lattice lattice = Lattice())
{
    std::vector<Level> levels;
    HostCSR A = std::move(A0);
    static const int geometric_env = getenv("FDD_TUNE_AMG_GEOMETRIC") ? atoi(getenv("FDD_TUNE_AMG_GEOMETRIC")) : -1; // development override
    const bool geometric = geometric_env >= 0 ? geometric_env != 0 : o.geometric_levels;
    int geometric_done = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char *what, int l) {
        const double t = now();
        if (verbose) printf("low_order:   level %d %-24s %7.3f s\n", l, what, t - t_mark);
        t_mark = t;
    };
    for (int l = 0; l < o.max_levels; l++)
    {
        Level L;
        const int n = A.rows;
        std::vector<double> d = diagonal(A);
        L.D.resize(n);
        parallel_ranges(n, range_parts(n), [&](long long i0, long long i1, int) {
            for (long long i = i0; i < i1; i++) L.D[i] = 1.0 / std::sqrt(d[i]);
        });
        const double lmax = o.lambda_max ? o.lambda_max(A, L.D, o.power_iterations) : max_eigenvalue_scaled(A, L.D, o.power_iterations);
        {
            // the levels that are coarsened on the lattice lose 2.33 nodes per direction in one step (N = 7), more than an
            // aggregation level: the smoother in front of such a step has to reach further down the spectrum
            static const double geo_ratio_env = getenv("FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO") ? atof(getenv("FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO")) : 0.0; // development override
            Options ol = o;
            const bool lattice_next = geometric and lattice.active(geometric_done > 0 ? 3 : o.geometric_min_nodes) and lattice.rows.cols == n and n > o.coarsest_size;
            if (lattice_next) ol.eig_ratio = geo_ratio_env > 0.0 ? geo_ratio_env : o.geometric_eig_ratio;
            L.coefs = chebyshev_coefficients(lmax, ol);
        }
        if (verbose) printf("low_order: level %d rows %d nnz %lld lambda_max(DAD) %.4f\n", l, n, A.nnz(), lmax);
        lap("diagonal + lambda_max", l);

        bool last = (n <= o.coarsest_size) or (l == o.max_levels - 1);
        HostCSR P;
        static const int min_nodes_env = getenv("FDD_TUNE_AMG_GEOMETRIC_MIN_NODES") ? atoi(getenv("FDD_TUNE_AMG_GEOMETRIC_MIN_NODES")) : 0; // development override
        // a lattice of fewer than geometric_min_nodes nodes is coarsened geometrically only as the continuation of a
        // geometric level above it (N = 7: 8 -> 4 -> 2 nodes); on its own (degrees 2 and 3) the one factor-3 step is weaker
        // than the aggregation (the rod of DESIGN 5.3, N = 2: 23 against 19 iterations)
        if (not last and geometric and lattice.active(geometric_done > 0 ? 3 : (min_nodes_env > 0 ? min_nodes_env : o.geometric_min_nodes)) and lattice.rows.cols == n)
        {
            geometric_done++;
            Lattice next;
            P = geometric_level(lattice, n, next, &L.transfer);
            lattice = std::move(next);
            if (verbose) printf("low_order: level %d coarsened on the lattice: %d -> %d rows, %d nodes per direction and element left\n", l, n, P.cols, lattice.n);
        }
        else if (not last)
        {
            std::vector<int> agg;
            int nc = aggregate(A, o.strength * std::pow(0.5, l), agg); // Galerkin operators spread: weaker threshold per level (a trilinear 27-point stencil has edge / corner entries of 1/16 and 1/32 of its diagonal and no face entries)
            static const int double_levels_env = getenv("FDD_TUNE_AMG_DOUBLE_AGG") ? atoi(getenv("FDD_TUNE_AMG_DOUBLE_AGG")) : -1; // development override
            if (l < (double_levels_env >= 0 ? double_levels_env : o.double_aggregation_levels) and nc > 0 and nc < n)
            {
                // aggregates of aggregates: the graph of the tentative coarse operator T^T A T, aggregated again
                HostCSR T;
                T.rows = n;
                T.cols = nc;
                T.ptr.resize(n + 1);
                T.col.assign(agg.begin(), agg.end());
                T.val.assign(n, 1.0);
                for (int i = 0; i <= n; i++) T.ptr[i] = i;
                HostCSR C = multiply(transpose(T), multiply(A, T));
                std::vector<int> agg2;
                const int nc2 = aggregate(C, o.strength * 0.5, agg2);
                if (nc2 > 0 and nc2 < nc)
                {
                    for (int i = 0; i < n; i++) agg[i] = agg2[agg[i]];
                    nc = nc2;
                }
            }
            if (verbose) printf("low_order: level %d aggregated: %d -> %d rows\n", l, n, nc);
            if (nc >= n or nc == 0)
                last = true;
            else
            {
                // tentative prolongator: 1 on the aggregate of each row; smoothed: (I - omega D^-1 A) P_tent
                HostCSR T;
                T.rows = n;
                T.cols = nc;
                T.ptr.resize(n + 1);
                T.col.resize(n);
                T.val.assign(n, 1.0);
                for (int i = 0; i <= n; i++) T.ptr[i] = i;
                for (int i = 0; i < n; i++) T.col[i] = agg[i];
                if (o.smooth_prolongator)
                {
                    const double omega = (4.0 / 3.0) / lmax; // lambda_max(D^-1 A) = lambda_max(DAD)
                    HostCSR S = A; // I - omega D^-1 A
                    for (int i = 0; i < n; i++)
                        for (int p = S.ptr[i]; p < S.ptr[i + 1]; p++)
                        {
                            S.val[p] = -omega * S.val[p] / d[i];
                            if (S.col[p] == i) S.val[p] += 1.0;
                        }
                    P = multiply(S, T);
                    // development option: drop the small entries of the smoothed interpolator row by row and rescale the
                    // rest to the row's sum (sparser Galerkin operators); 0 = keep everything
                    static const double p_drop = getenv("FDD_TUNE_AMG_P_DROP") ? atof(getenv("FDD_TUNE_AMG_P_DROP")) : 0.0;
                    static const int p_drop_from = getenv("FDD_TUNE_AMG_P_DROP_FROM") ? atoi(getenv("FDD_TUNE_AMG_P_DROP_FROM")) : 0;
                    if (p_drop > 0.0 and l >= p_drop_from) P = filter_rows(P, p_drop);
                }
                else
                    P = T;
            }
        }
        if (last)
        {
            L.A = std::move(A);
            levels.push_back(std::move(L));
            break;
        }
        lap("interpolator", l);
        HostCSR R = transpose(P);
        lap("transpose", l);
        HostCSR AP = multiply(A, P);
        lap("A P", l);
        HostCSR Ac = multiply(R, AP);
        lap("R (A P)", l);
        L.A = std::move(A);
        L.P = std::move(P);
        levels.push_back(std::move(L));
        A = std::move(Ac);
    }
    return levels;
}

} // namespace low_order
} // namespace fdd

#endif
