/*
 * composite.hpp -- setup of the full-domain-decomposition composite of one rank
 * (the reference's Subdomain constructor, subdomain.tpp:198-2747), host side.
 *
 * Every rank owns a composite picture of the WHOLE domain:
 *   subdomain region   its own elements at degree N, `subdomain_overlap` rings of neighbour elements per
 *                      polynomial level at degrees N, N-r, ..., 1 (subdomain.tpp:455-510), plus one "extended"
 *                      ring at degree 1 whose dofs are copies of superdomain dofs (:512-531);
 *   superdomain        everything else at degree 1, coarsened algebraically with distance from the subdomain
 *                      (:1754-2576), plus "extended" copies of the subdomain dofs next to it.
 * Elements of different degree meet non-conformingly: face / edge points of the higher-degree side are
 * interpolated from the lower-degree side's dofs (rows of Q carrying J_cf weights, :1179-1585).
 *
 * What is built here, in the reference's order:
 *   - global element graph from the all-gathered corner ids (:198-453), regions (:455-579);
 *   - the pull of ring-element data (mask, geometric factors, global ids, coordinates) from their owners -- the
 *     reference's gs "tree" handle at setup (:601-805) becomes two point-to-point exchanges (Comm::exchange_host);
 *     the request lists are kept as the solve-time exchange plan of Subdomain::tree_operator (:4626);
 *   - region numbering (:920-1176) and the non-conforming Q (:1179-1585);
 *   - the coarse (degree 1) operator of the whole domain, Qt_coarse (:1632-1848);
 *   - the superdomain operator A and the interpolator Pt (:1850-2576).  DEVIATION, labelled: the reference takes
 *     the coarsening from HYPRE BoomerAMG's hierarchy (absent here).  This build grades the superdomain with its
 *     own smoothed aggregation: at every level the dofs within `superdomain_overlap` graph steps of what is
 *     already kept stay as they are, the rest is aggregated (greedy, strength 0.08, one damped-Jacobi smoothing
 *     step of the tentative interpolator on the aggregated rows only), and the next level repeats that on the
 *     Galerkin operator.  Same structure as the reference's composite (local | overlap per level), same
 *     operators downstream: A = P^T A_c P restricted to the superdomain dofs, Pt = P^T;
 *   - interface maps Q_int / Qt_int / QQt_int and the norm weights (:2581-2747).
 *
 * Dof numbering.  Subdomain dofs are ordered [regular | interface | extended] and superdomain dofs
 * [interface | regular | extended] as in the reference (:1100-1176, :2419-2424); inside "regular" the dofs that sit on
 * the rank's own elements come first, in the Domain's node order (so that the outer solve can hand its node
 * vectors to the inner solve without renumbering), where the reference ranks global ids.  Nothing outside the
 * Subdomain sees the numbering.
 *
 * Pure host code; pure setup.
 */
#ifndef FDD_COMPOSITE_HPP
#define FDD_COMPOSITE_HPP

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "comm.hpp"
#include "config.hpp"
#include "gll.hpp"
#include "low_order.hpp"

namespace fdd
{
namespace composite
{

using low_order::HostCSR;

struct RegionElem
{
    int id = 0;          // global element id
    int level = 0;       // index into poly_degree
    int N = 1;           // polynomial degree
    int n = 2;           // N + 1
    int num_points = 0;
    int offset = 0;      // first point in the region's point vectors
    int owner = 0;       // rank that owns the element
    int owner_elem = 0;  // its local element id there
};

// one packed message per peer and direction (solve-time ring pull, subdomain.tpp:4626)
struct PeerPlan
{
    int rank = 0;
    std::vector<int> send_level, send_elem; // what the peer pulls from this rank: (level, local element) in its region order
    std::vector<int> recv_elem;             // region elements (indices into `sub`) that come from the peer, in the same order
    long long send_points = 0, recv_points = 0;
};

// which lower-degree neighbour a hanging edge / face of a region element takes its values from
struct EdgeLink
{
    int elem_i, eid, elem_j, eid_j;
};
struct FaceLink
{
    int elem_i, fid, elem_j, fid_j;
};

struct Composite
{
    int dim = 3;
    int num_vertices = 8;
    int num_levels = 1;
    std::vector<int> poly_degree;

    int num_total_elements = 0;
    std::vector<int> proc_count, proc_offset; // elements per rank

    // subdomain region (own | rings by level | extended ring)
    std::vector<RegionElem> sub;
    int num_sub_elems = 0, num_sub_ext_elems = 0;
    int num_sub_points = 0, num_sub_ext_points = 0;
    std::vector<int> level_first_elem, level_num_elems; // per level: the contiguous run of region elements of that degree (extended ring included in the last level)
    int num_sup_elems = 0, num_sup_ext_elems = 0;
    std::vector<int> sup_ext_sub_index; // region index (into sub) of every superdomain-extended element

    // per point of the extended subdomain region
    std::vector<double> mask, x, y, z;
    std::vector<double> G[NUM_GEOM_FACTS];
    std::vector<long long> glo; // raw global id in the element's own level mesh
    std::vector<double> G_deg1[NUM_GEOM_FACTS]; // geometric factors of the degree-1 region elements only (the low-order matrix needs them after G is released)
    std::vector<int> deg1_offset;               // per region element: first entry in G_deg1, -1 for higher degrees

    // numbering and Q
    std::vector<int> point_dof; // direct dof of a point, -1: Dirichlet or hanging
    std::vector<int> Q_row, Q_col;
    std::vector<double> Q_val; // triplets of the non-conforming Q (rows = region points, cols = extended dofs)
    int sub_num_dofs = 0, sub_num_ext_dofs = 0;
    int num_interface_dofs = 0;
    int num_own_dofs = 0; // leading regular dofs that sit on own elements, in Domain node order
    std::vector<EdgeLink> edge_links;
    std::vector<FaceLink> face_links;

    // coarse level of the whole domain
    int num_coarse_dofs = 0;
    std::vector<int> dof_num_coarse; // 1-based coarse dof of every (element, vertex), 0: Dirichlet
    HostCSR Qt_coarse;               // num_coarse_dofs x (num_total_elements * num_vertices)

    // superdomain
    int sup_num_dofs = 0, sup_num_ext_dofs = 0;
    HostCSR A_sup;                // sup_num_ext_dofs x sup_num_ext_dofs
    HostCSR Pt_sup;               // sup_num_ext_dofs x num_coarse_dofs
    std::vector<int> dof_sup;     // 1-based superdomain dof of every coarse dof kept at level 0, 0: none
    std::vector<int> comp_levels; // composite dofs contributed per coarsening level (statistics)

    // interface maps: one unit entry per row, stored as the column of that entry
    int num_dofs = 0;                 // unique dofs of the composite
    std::vector<int> Q_int_col;       // [ext sub | ext sup] -> unique dof
    std::vector<int> Qt_int_col;      // unique dof -> position in [ext sub | ext sup]
    std::vector<int> QQt_int_col;     // [ext sub | ext sup] -> position of the owner's value in [ext sub | ext sup]
    std::vector<double> norm_weight;  // [ext sub | ext sup]

    std::vector<PeerPlan> peers;
};

// ---------------------------------------------------------------------------------------------------------------
// element topology in the reference's conventions (vertex v = i + 2j + 4k; edge and face tables subdomain.tpp:312-401)
// ---------------------------------------------------------------------------------------------------------------
static const int kEdgePairs3[12][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const int kEdgePairs2[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
static const int kFaceQuads[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 2, 4, 6}, {1, 3, 5, 7}};

inline int corner_index(int v, int n, int dim)
{
    const int i = (v & 1) ? n - 1 : 0, j = (v & 2) ? n - 1 : 0, k = (dim == 3 and (v & 4)) ? n - 1 : 0;
    return i + j * n + k * n * n;
}

// the n points of edge `eid` in increasing local coordinate (subdomain.tpp:1197-1308)
inline void edge_points(int eid, int n, int dim, std::vector<int> &idx)
{
    idx.resize(n);
    const int (*pairs)[2] = (dim == 2) ? kEdgePairs2 : kEdgePairs3;
    const int a = corner_index(pairs[eid][0], n, dim), b = corner_index(pairs[eid][1], n, dim);
    const int step = (b - a) / (n - 1);
    for (int k = 0; k < n; k++) idx[k] = a + k * step;
}

// the n*n points of face `fid`, first face coordinate fastest (subdomain.tpp:1366-1431)
inline void face_points(int fid, int n, std::vector<int> &idx)
{
    idx.resize((size_t)n * n);
    const int nn = n * n;
    for (int b = 0; b < n; b++)
        for (int a = 0; a < n; a++)
        {
            int p = 0;
            switch (fid)
            {
            case 0: p = a + b * n; break;
            case 1: p = a + b * n + (n - 1) * nn; break;
            case 2: p = a + b * nn; break;
            case 3: p = a + (n - 1) * n + b * nn; break;
            case 4: p = a * n + b * nn; break;
            default: p = (n - 1) + a * n + b * nn; break;
            }
            idx[a + b * n] = p;
        }
}

// dense 1-based rank of the positive values, 0 stays 0 (the reference's ranking lambda, subdomain.tpp:881-918, 1666-1704)
inline int rank_positive(const std::vector<long long> &v, std::vector<int> &rank)
{
    std::vector<long long> u;
    u.reserve(v.size());
    for (long long a : v)
        if (a > 0) u.push_back(a);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    rank.resize(v.size());
    for (size_t i = 0; i < v.size(); i++) rank[i] = (v[i] > 0) ? (int)(std::lower_bound(u.begin(), u.end(), v[i]) - u.begin()) + 1 : 0;
    return (int)u.size();
}

// ---------------------------------------------------------------------------------------------------------------
// graded aggregation of the superdomain (this build's stand-in for subdomain.tpp:1850-2400)
// ---------------------------------------------------------------------------------------------------------------
struct GradingOptions
{
    double strength = 0.08;  // |a_ij| >= strength * sqrt(a_ii a_jj)
    double omega = 2.0 / 3.0; // damped-Jacobi smoothing of the tentative interpolator
    int max_levels = 12;
    int keep_at_most = 8;    // this few remaining dofs are simply kept
    double tie = 1.0e-10;    // comparisons are decided with this relative slack, so that rounding cannot reorder ties
};

// Greedy aggregation of the `active` rows of A on its strength graph (the three passes of low_order::aggregate):
// aggregate of every active row in agg (-1 elsewhere), returns the number of aggregates.
inline int aggregate_active(const HostCSR &A, const std::vector<char> &active, const GradingOptions &o, std::vector<int> &agg)
{
    const int n = A.rows;
    const std::vector<double> d = low_order::diagonal(A);
    agg.assign(n, -1);
    auto strong = [&](int i, int p) {
        const int j = A.col[p];
        return j != i and active[j] and std::abs(A.val[p]) >= o.strength * std::sqrt(std::abs(d[i] * d[j])) * (1.0 - o.tie);
    };
    int count = 0;
    for (int i = 0; i < n; i++)
    {
        if (not active[i] or agg[i] != -1) continue;
        bool free_nbhd = true, has_strong = false;
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
    std::vector<int> joined(n, -1);
    for (int i = 0; i < n; i++)
    {
        if (not active[i] or agg[i] != -1) continue;
        double best = 0.0;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (strong(i, p) and agg[A.col[p]] != -1 and std::abs(A.val[p]) > best * (1.0 + o.tie))
            {
                best = std::abs(A.val[p]);
                joined[i] = agg[A.col[p]];
            }
    }
    for (int i = 0; i < n; i++)
        if (active[i] and agg[i] == -1 and joined[i] != -1) agg[i] = joined[i];
    for (int i = 0; i < n; i++)
    {
        if (not active[i] or agg[i] != -1) continue;
        agg[i] = count;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (strong(i, p) and agg[A.col[p]] == -1) agg[A.col[p]] = count;
        count++;
    }
    return count;
}

// The composite interpolator P (A_c.rows x composite dofs) and operator P^T A_c P.
//   marker[i] in {0,1,2,3,4}: the dof groups that lead the composite numbering (subdomain.tpp:1860-1905);
//   every marked dof is "local" at level 0 (:1929-1931).
// On return: comp_of_fine[i] = composite dof of a coarse dof kept at level 0 (-1: aggregated away),
// group_count[0..3] = sizes of the marker groups 1..4, kept_per_level = composite dofs added per level.
inline void grade_superdomain(const HostCSR &A_c, const std::vector<int> &marker, int superdomain_overlap, const GradingOptions &o, HostCSR &P_comp, HostCSR &A_comp, std::vector<int> &comp_of_fine, int group_count[4],
                              std::vector<int> &kept_per_level)
{
    const int n0 = A_c.rows;
    HostCSR A = A_c;
    std::vector<int> D(n0, 0); // 1: already in the composite, 2: overlap of this level, 0: remaining
    for (int i = 0; i < n0; i++) D[i] = (marker[i] > 0) ? 1 : 0;
    for (int g = 0; g < 4; g++) group_count[g] = 0;
    for (int i = 0; i < n0; i++)
        if (marker[i] > 0) group_count[marker[i] - 1]++;
    comp_of_fine.assign(n0, -1);
    kept_per_level.clear();
    bool have_P = false;
    int overlap = superdomain_overlap;

    for (int level = 0;; level++)
    {
        const int n = A.rows;
        // `overlap` sweeps of reach through the graph of A (subdomain.tpp:1942-1967)
        std::vector<double> w(n), w2(n);
        for (int i = 0; i < n; i++) w[i] = (D[i] > 0) ? 1.0 : 0.0;
        for (int nu = 0; nu < overlap; nu++)
        {
            for (int row = 0; row < n; row++)
            {
                double val = 0.0;
                for (int p = A.ptr[row]; p < A.ptr[row + 1]; p++) val += w[A.col[p]];
                w2[row] = val;
            }
            w.swap(w2);
        }
        if (overlap == 0) overlap = 1; // :1959
        int remaining = 0;
        for (int i = 0; i < n; i++)
        {
            if (D[i] == 0 and w[i] > 0.0) D[i] = 2;
            if (D[i] == 0) remaining++;
        }
        std::vector<char> active(n, 0);
        std::vector<int> agg;
        int num_agg = 0;
        if (remaining > 0 and remaining > o.keep_at_most and level < o.max_levels - 1)
        {
            for (int i = 0; i < n; i++) active[i] = (D[i] == 0);
            GradingOptions ol = o;
            ol.strength = o.strength * std::pow(0.5, level); // Galerkin operators spread: weaker threshold per level
            num_agg = aggregate_active(A, active, ol, agg);
        }
        if (remaining > 0 and (num_agg == 0 or num_agg >= remaining))
        {
            // nothing left to gain: what remains is kept as it is (the reference's last level, :1961-1963)
            for (int i = 0; i < n; i++)
                if (D[i] == 0) D[i] = 2;
            remaining = 0;
        }

        // composite position of the kept dofs: level 0 leads with the marker groups, later levels keep the order
        // they have (already-kept first) and append the new overlap
        std::vector<int> pos(n, -1);
        int nk = 0;
        if (level == 0)
        {
            for (int m = 1; m <= 4; m++)
                for (int i = 0; i < n; i++)
                    if (marker[i] == m) pos[i] = nk++;
            for (int i = 0; i < n; i++)
                if (D[i] == 2) pos[i] = nk++;
            for (int i = 0; i < n; i++) comp_of_fine[i] = pos[i];
            kept_per_level.push_back(nk);
        }
        else
        {
            for (int i = 0; i < n; i++)
                if (D[i] == 1) pos[i] = nk++;
            const int before = nk;
            for (int i = 0; i < n; i++)
                if (D[i] == 2) pos[i] = nk++;
            kept_per_level.push_back(nk - before);
        }

        // this level's interpolator: identity on the kept dofs, smoothed aggregation on the rest
        HostCSR P;
        P.rows = n;
        P.cols = nk + ((remaining > 0) ? num_agg : 0);
        P.ptr.assign(n + 1, 0);
        {
            const std::vector<double> d = low_order::diagonal(A);
            std::vector<double> acc(P.cols, 0.0);
            std::vector<int> stamp(P.cols, -1), cols;
            auto tcol = [&](int j) { return (pos[j] >= 0) ? pos[j] : nk + agg[j]; };
            for (int i = 0; i < n; i++)
            {
                if (pos[i] >= 0)
                {
                    P.col.push_back(pos[i]);
                    P.val.push_back(1.0);
                }
                else
                {
                    cols.clear();
                    auto add = [&](int c, double v) {
                        if (stamp[c] != i)
                        {
                            stamp[c] = i;
                            acc[c] = 0.0;
                            cols.push_back(c);
                        }
                        acc[c] += v;
                    };
                    add(tcol(i), 1.0);
                    for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++) add(tcol(A.col[p]), -o.omega * A.val[p] / d[i]);
                    std::sort(cols.begin(), cols.end());
                    for (int c : cols)
                    {
                        P.col.push_back(c);
                        P.val.push_back(acc[c]);
                    }
                }
                P.ptr[i + 1] = (int)P.col.size();
            }
        }

        HostCSR R = low_order::transpose(P);
        HostCSR AP = low_order::multiply(A, P);
        HostCSR An = low_order::multiply(R, AP);
        P_comp = have_P ? low_order::multiply(P_comp, P) : P;
        have_P = true;
        A = std::move(An);
        if (remaining == 0) break;
        D.assign(A.rows, 0);
        for (int i = 0; i < nk; i++) D[i] = 1;
    }
    A_comp = std::move(A);
}

// ---------------------------------------------------------------------------------------------------------------
// message helpers for the setup exchanges
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
inline void put(std::vector<char> &buf, const T *v, size_t n)
{
    const size_t at = buf.size();
    buf.resize(at + n * sizeof(T));
    if (n) memcpy(buf.data() + at, v, n * sizeof(T));
}
template <typename T>
inline void get(const std::vector<char> &buf, size_t &at, T *v, size_t n)
{
    if (n) memcpy(v, buf.data() + at, n * sizeof(T));
    at += n * sizeof(T);
}

// all-gather of per-element records (`per_elem` values of T for each local element), in global element order
template <typename T>
inline std::vector<T> allgather_elements(const std::vector<T> &local, int per_elem)
{
    static_assert(sizeof(T) == 8, "8-byte records");
    std::vector<long long> bits(local.size());
    if (not local.empty()) memcpy(bits.data(), local.data(), local.size() * 8);
    std::vector<int> counts;
    std::vector<long long> all = fdd::comm().allgatherv_host(bits, counts);
    std::vector<T> out(all.size());
    if (not all.empty()) memcpy(out.data(), all.data(), all.size() * 8);
    (void)per_elem;
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// the composite of this rank.  `domains[deg]` is the rank's Domain at degree deg (mesh arrays of its own elements);
// J_cf[(N_c, N_f)] the interpolators (n_f x n_c row-major), D_hat_coarse the 2 x 2 differentiation matrix of degree 1.
// own_point_node / own_num_nodes: the fine Domain's point -> node map (the order of the leading dofs).
// ---------------------------------------------------------------------------------------------------------------
template <typename DomainMap>
inline Composite build(DomainMap &domains, const std::vector<int> &poly_degree, int subdomain_overlap, int superdomain_overlap, double epsilon, const std::map<std::pair<int, int>, std::vector<double>> &J_cf,
                       const std::vector<double> &D_hat_coarse, const std::vector<int> &own_point_node, int own_num_nodes, const GradingOptions &grading = GradingOptions())
{
    Composite c;
    fdd::Comm &comm = fdd::comm();
    const int proc_id = comm.rank, num_procs = comm.size;
    // FDD_SETUP_TIMING=1: rank 0 prints the host time of every phase below
    const bool phase_timing = getenv("FDD_SETUP_TIMING") != nullptr and proc_id == 0;
    auto phase_clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double phase_t0 = phase_clock();
    auto phase = [&](const char *what) {
        if (not phase_timing) return;
        const double now = phase_clock();
        printf("composite::build %-28s %8.3f s\n", what, now - phase_t0);
        fflush(stdout);
        phase_t0 = now;
    };
    auto &domain = domains[poly_degree[0]];
    const int dim = domain.mesh.dim;
    const int nv = (dim == 2) ? 4 : 8;
    const int num_edges = (dim == 2) ? 4 : 12;
    const int num_faces = (dim == 2) ? 0 : 6;
    const int num_levels = (int)poly_degree.size();
    c.dim = dim;
    c.num_vertices = nv;
    c.num_levels = num_levels;
    c.poly_degree = poly_degree;
    const int num_local_elements = domain.num_local_elements;
    auto npts_of = [&](int N) { return (dim == 2) ? (N + 1) * (N + 1) : (N + 1) * (N + 1) * (N + 1); };

    // ---- corner ids of every element of the domain (subdomain.tpp:198-268) ----
    std::vector<long long> geometry_mesh;
    {
        const int n0 = poly_degree[0] + 1, np0 = npts_of(poly_degree[0]);
        std::vector<long long> local((size_t)num_local_elements * nv);
        for (int e = 0; e < num_local_elements; e++)
            for (int v = 0; v < nv; v++) local[(size_t)e * nv + v] = domain.mesh.glo_num[(size_t)e * np0 + corner_index(v, n0, dim)];
        std::vector<int> counts;
        geometry_mesh = comm.allgatherv_host(local, counts);
        c.proc_count.resize(num_procs);
        c.proc_offset.resize(num_procs);
        for (int p = 0; p < num_procs; p++) c.proc_count[p] = counts[p] / nv;
        c.proc_offset[0] = 0;
        for (int p = 1; p < num_procs; p++) c.proc_offset[p] = c.proc_offset[p - 1] + c.proc_count[p - 1];
    }
    const int num_total_elements = (int)(geometry_mesh.size() / nv);
    c.num_total_elements = num_total_elements;

    // element -> (owner, local id) (subdomain.tpp:270-280)
    std::vector<int> owner(num_total_elements), owner_elem(num_total_elements);
    for (int p = 0; p < num_procs; p++)
        for (int e = 0; e < c.proc_count[p]; e++)
        {
            owner[c.proc_offset[p] + e] = p;
            owner_elem[c.proc_offset[p] + e] = e;
        }

    // vertex -> elements (what the `expander` matrix encodes, subdomain.tpp:282-453: two elements are neighbours
    // when they share a vertex)
    std::vector<int> vert_of((size_t)num_total_elements * nv);
    std::vector<int> v2e_ptr, v2e;
    {
        std::vector<long long> ids(geometry_mesh);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        const int nvert = (int)ids.size();
        for (size_t i = 0; i < geometry_mesh.size(); i++) vert_of[i] = (int)(std::lower_bound(ids.begin(), ids.end(), geometry_mesh[i]) - ids.begin());
        v2e_ptr.assign(nvert + 1, 0);
        for (int v : vert_of) v2e_ptr[v + 1]++;
        for (int v = 0; v < nvert; v++) v2e_ptr[v + 1] += v2e_ptr[v];
        v2e.resize(vert_of.size());
        std::vector<int> fill(v2e_ptr.begin(), v2e_ptr.end() - 1);
        for (int e = 0; e < num_total_elements; e++)
            for (int v = 0; v < nv; v++) v2e[fill[vert_of[(size_t)e * nv + v]]++] = e;
    }

    phase("element graph");
    // ---- computational regions (subdomain.tpp:455-579) ----
    std::vector<int> mark(num_total_elements, 0); // > 0: in the subdomain (work_hst[1] of the reference)
    std::vector<char> reached(num_total_elements, 0);
    std::vector<int> frontier;
    auto add_elem = [&](int e, int level) {
        RegionElem r;
        r.id = e;
        r.level = level;
        r.N = poly_degree[level];
        r.n = r.N + 1;
        r.num_points = npts_of(r.N);
        r.owner = owner[e];
        r.owner_elem = owner_elem[e];
        c.sub.push_back(r);
    };
    for (int e = 0; e < num_local_elements; e++)
    {
        const int g = c.proc_offset[proc_id] + e;
        add_elem(g, 0);
        mark[g] = e + 1;
        reached[g] = 1;
        frontier.push_back(g);
    }
    c.num_sub_elems = num_local_elements;
    auto expand = [&]() {
        std::vector<int> next;
        for (int e : frontier)
            for (int v = 0; v < nv; v++)
            {
                const int vid = vert_of[(size_t)e * nv + v];
                for (int q = v2e_ptr[vid]; q < v2e_ptr[vid + 1]; q++)
                    if (not reached[v2e[q]])
                    {
                        reached[v2e[q]] = 1;
                        next.push_back(v2e[q]);
                    }
            }
        std::sort(next.begin(), next.end());
        return next;
    };
    {
        int overlap = subdomain_overlap;
        std::vector<int> pending; // reached but not yet given a degree
        for (int l = 0; l < num_levels; l++)
        {
            for (int nu = 0; nu < overlap; nu++)
            {
                std::vector<int> next = expand();
                pending.insert(pending.end(), next.begin(), next.end());
                frontier = next;
            }
            std::sort(pending.begin(), pending.end());
            for (int e : pending)
            {
                mark[e] = c.num_sub_elems + 1;
                add_elem(e, l);
                c.num_sub_elems++;
            }
            pending.clear();
            if (overlap == 0) overlap = 1; // :509
        }
    }
    c.num_sub_ext_elems = c.num_sub_elems;
    {
        // one more ring: the "extended" elements, degree 1 (subdomain.tpp:512-524); the frontier after an empty
        // expansion is empty, so the set the reference multiplies by the expander is rebuilt from the marks
        frontier.clear();
        for (int e = 0; e < num_total_elements; e++)
            if (mark[e] > 0) frontier.push_back(e);
        std::vector<int> ring = expand();
        for (int e : ring)
        {
            add_elem(e, num_levels - 1);
            c.num_sub_ext_elems++;
        }
    }
    // the region is sorted by level (own | rings level by level | extended ring at the last level): one contiguous
    // run of elements per polynomial degree, which is what the level-sorted stiffness launches walk
    c.level_first_elem.assign(num_levels, 0);
    c.level_num_elems.assign(num_levels, 0);
    for (int r = (int)c.sub.size() - 1; r >= 0; r--)
    {
        c.level_first_elem[c.sub[r].level] = r;
        c.level_num_elems[c.sub[r].level]++;
    }
    // superdomain: every unmarked element, then the subdomain elements next to one (subdomain.tpp:515-553)
    for (int e = 0; e < num_total_elements; e++)
        if (mark[e] == 0) c.num_sup_elems++;
    c.num_sup_ext_elems = c.num_sup_elems;
    {
        std::vector<int> region_index(num_total_elements, -1);
        for (int r = 0; r < c.num_sub_ext_elems; r++) region_index[c.sub[r].id] = r;
        for (int e = 0; e < num_total_elements; e++)
        {
            if (mark[e] == 0) continue;
            bool touches = false;
            for (int v = 0; v < nv and not touches; v++)
            {
                const int vid = vert_of[(size_t)e * nv + v];
                for (int q = v2e_ptr[vid]; q < v2e_ptr[vid + 1]; q++)
                    if (mark[v2e[q]] == 0)
                    {
                        touches = true;
                        break;
                    }
            }
            if (touches)
            {
                c.sup_ext_sub_index.push_back(region_index[e]);
                c.num_sup_ext_elems++;
            }
        }
    }
    {
        int off = 0;
        for (int r = 0; r < c.num_sub_ext_elems; r++)
        {
            c.sub[r].offset = off;
            off += c.sub[r].num_points;
            if (r + 1 == c.num_sub_elems) c.num_sub_points = off;
        }
        c.num_sub_ext_points = off;
        if (c.num_sub_elems == 0) c.num_sub_points = 0;
    }
    const int NP = c.num_sub_ext_points;

    phase("regions");
    // ---- data of the region's elements: own ones from the rank's meshes, the others pulled from their owners
    // (the gs exchanges of subdomain.tpp:644-805) ----
    c.mask.assign(NP, 0.0);
    c.x.assign(NP, 0.0);
    c.y.assign(NP, 0.0);
    c.z.assign(NP, 0.0);
    c.glo.assign(NP, 0);
    for (int g = 0; g < NUM_GEOM_FACTS; g++) c.G[g].assign(NP, 0.0);
    {
        // requests by owner, in region order (this order IS the solve-time message layout)
        std::vector<std::vector<int>> want(num_procs);
        std::vector<std::vector<int>> want_region(num_procs);
        for (int r = 0; r < c.num_sub_ext_elems; r++)
        {
            const RegionElem &el = c.sub[r];
            if (el.owner == proc_id) continue;
            want[el.owner].push_back(el.level);
            want[el.owner].push_back(el.owner_elem);
            want_region[el.owner].push_back(r);
        }
        std::vector<std::vector<char>> out(num_procs);
        for (int p = 0; p < num_procs; p++) put(out[p], want[p].data(), want[p].size());
        std::vector<std::vector<char>> in = comm.exchange_host(out);

        // answer: per requested element glo | mask | g_1..g_6 | x | y | z at the requested level
        std::vector<std::vector<char>> reply(num_procs);
        std::vector<std::vector<int>> asked(num_procs);
        for (int p = 0; p < num_procs; p++)
        {
            if (p == proc_id) continue;
            asked[p].resize(in[p].size() / sizeof(int));
            size_t at = 0;
            get(in[p], at, asked[p].data(), asked[p].size());
            for (size_t k = 0; k + 1 < asked[p].size(); k += 2)
            {
                const int l = asked[p][k], e = asked[p][k + 1];
                auto &m = domains[poly_degree[l]].mesh;
                const size_t np = (size_t)npts_of(poly_degree[l]), o = (size_t)e * np;
                put(reply[p], m.glo_num.data() + o, np);
                put(reply[p], m.p_mask.data() + o, np);
                for (int g = 0; g < NUM_GEOM_FACTS; g++) put(reply[p], m.g[g].data() + o, np);
                put(reply[p], m.x.data() + o, np);
                put(reply[p], m.y.data() + o, np);
                if (dim == 3) put(reply[p], m.z.data() + o, np);
            }
        }
        std::vector<std::vector<char>> data = comm.exchange_host(reply);
        for (int p = 0; p < num_procs; p++)
        {
            if (p == proc_id) continue;
            size_t at = 0;
            for (int r : want_region[p])
            {
                const RegionElem &el = c.sub[r];
                const size_t np = (size_t)el.num_points, o = (size_t)el.offset;
                get(data[p], at, c.glo.data() + o, np);
                get(data[p], at, c.mask.data() + o, np);
                for (int g = 0; g < NUM_GEOM_FACTS; g++) get(data[p], at, c.G[g].data() + o, np);
                get(data[p], at, c.x.data() + o, np);
                get(data[p], at, c.y.data() + o, np);
                if (dim == 3) get(data[p], at, c.z.data() + o, np);
            }
            if (want_region[p].empty() and asked[p].empty()) continue;
            PeerPlan plan;
            plan.rank = p;
            plan.recv_elem = want_region[p];
            for (int r : want_region[p]) plan.recv_points += c.sub[r].num_points;
            for (size_t k = 0; k + 1 < asked[p].size(); k += 2)
            {
                plan.send_level.push_back(asked[p][k]);
                plan.send_elem.push_back(asked[p][k + 1]);
                plan.send_points += npts_of(poly_degree[asked[p][k]]);
            }
            c.peers.push_back(plan);
        }
        for (int r = 0; r < c.num_sub_ext_elems; r++)
        {
            const RegionElem &el = c.sub[r];
            if (el.owner != proc_id) continue;
            auto &m = domains[poly_degree[el.level]].mesh;
            const size_t np = (size_t)el.num_points, o = (size_t)el.offset, s = (size_t)el.owner_elem * np;
            std::copy(m.glo_num.begin() + s, m.glo_num.begin() + s + np, c.glo.begin() + o);
            std::copy(m.p_mask.begin() + s, m.p_mask.begin() + s + np, c.mask.begin() + o);
            for (int g = 0; g < NUM_GEOM_FACTS; g++) std::copy(m.g[g].begin() + s, m.g[g].begin() + s + np, c.G[g].begin() + o);
            std::copy(m.x.begin() + s, m.x.begin() + s + np, c.x.begin() + o);
            std::copy(m.y.begin() + s, m.y.begin() + s + np, c.y.begin() + o);
            if (dim == 3) std::copy(m.z.begin() + s, m.z.begin() + s + np, c.z.begin() + o);
        }
    }

    c.deg1_offset.assign(c.num_sub_ext_elems, -1);
    {
        int at = 0;
        for (int r = 0; r < c.num_sub_ext_elems; r++)
            if (c.sub[r].N == 1)
            {
                c.deg1_offset[r] = at;
                at += c.sub[r].num_points;
            }
        for (int g = 0; g < NUM_GEOM_FACTS; g++)
        {
            c.G_deg1[g].resize(at);
            for (int r = 0; r < c.num_sub_ext_elems; r++)
                if (c.deg1_offset[r] >= 0) std::copy(c.G[g].begin() + c.sub[r].offset, c.G[g].begin() + c.sub[r].offset + c.sub[r].num_points, c.G_deg1[g].begin() + c.deg1_offset[r]);
        }
    }

    phase("region data pull");
    // ---- coarse level of the whole domain: geometric factors and masked ids of every element's vertices
    // (subdomain.tpp:1632-1713) ----
    auto &coarse_domain = domains[poly_degree[num_levels - 1]];
    std::vector<double> geom_fact_coarse[NUM_GEOM_FACTS];
    std::vector<long long> glo_num_coarse; // id * mask
    {
        const int npc = npts_of(poly_degree[num_levels - 1]); // == nv: the last level has degree 1
        for (int g = 0; g < NUM_GEOM_FACTS; g++)
        {
            std::vector<double> local(coarse_domain.mesh.g[g].begin(), coarse_domain.mesh.g[g].begin() + (size_t)num_local_elements * npc);
            geom_fact_coarse[g] = allgather_elements(local, nv);
        }
        std::vector<long long> local((size_t)num_local_elements * npc);
        for (size_t i = 0; i < local.size(); i++) local[i] = (coarse_domain.mesh.p_mask[i] > 0.0) ? coarse_domain.mesh.glo_num[i] : 0;
        glo_num_coarse = allgather_elements(local, nv);
    }
    c.num_coarse_dofs = rank_positive(glo_num_coarse, c.dof_num_coarse);
    const int num_coarse_dofs = c.num_coarse_dofs;
    {
        std::vector<int> ti, tj;
        std::vector<double> tv;
        for (int i = 0; i < num_total_elements * nv; i++)
            if (c.dof_num_coarse[i] > 0)
            {
                ti.push_back(c.dof_num_coarse[i] - 1);
                tj.push_back(i);
                tv.push_back(1.0);
            }
        c.Qt_coarse = low_order::from_triplets(num_coarse_dofs, num_total_elements * nv, ti, tj, tv);
    }

    phase("coarse level");
    // ---- interface nodes (subdomain.tpp:810-843): dofs of the subdomain's degree-1 elements that also sit on a
    // superdomain element ----
    std::unordered_set<long long> interface_glo_num;
    {
        std::unordered_set<long long> subdomain_glo_num;
        for (int r = 0; r < c.num_sub_elems; r++)
        {
            const RegionElem &el = c.sub[r];
            if (el.N != 1) continue;
            for (int v = 0; v < el.num_points; v++)
                if (c.mask[el.offset + v] > 0.0) subdomain_glo_num.insert(c.glo[el.offset + v]);
        }
        for (int e = 0; e < num_total_elements; e++)
        {
            if (mark[e] != 0) continue;
            for (int v = 0; v < nv; v++)
            {
                const long long g = glo_num_coarse[(size_t)e * nv + v];
                if (g > 0 and subdomain_glo_num.count(g)) interface_glo_num.insert(g);
            }
        }
    }
    c.num_interface_dofs = (int)interface_glo_num.size();

    phase("interface nodes");
    // ---- connectivity of the subdomain region through shared edges and faces (subdomain.tpp:845-878) ----
    const int RE = c.num_sub_ext_elems;
    typedef std::array<long long, 2> EdgeKey;
    typedef std::array<long long, 4> FaceKey;
    std::map<EdgeKey, std::vector<int>> edge_map;
    std::map<FaceKey, std::vector<int>> face_map;
    auto corner_glo = [&](int r, int v) { return c.glo[c.sub[r].offset + corner_index(v, c.sub[r].n, dim)]; };
    auto edge_key = [&](int r, int eid) {
        const int (*pairs)[2] = (dim == 2) ? kEdgePairs2 : kEdgePairs3;
        EdgeKey k = {corner_glo(r, pairs[eid][0]), corner_glo(r, pairs[eid][1])};
        if (k[0] > k[1]) std::swap(k[0], k[1]);
        return k;
    };
    auto face_key = [&](int r, int fid) {
        FaceKey k = {corner_glo(r, kFaceQuads[fid][0]), corner_glo(r, kFaceQuads[fid][1]), corner_glo(r, kFaceQuads[fid][2]), corner_glo(r, kFaceQuads[fid][3])};
        std::sort(k.begin(), k.end());
        return k;
    };
    for (int r = 0; r < RE; r++)
    {
        for (int eid = 0; eid < num_edges; eid++) edge_map[edge_key(r, eid)].push_back(r);
        for (int fid = 0; fid < num_faces; fid++) face_map[face_key(r, fid)].push_back(r);
    }

    phase("connectivity");
    // ---- region numbering (subdomain.tpp:920-1176) ----
    // key = global id made unique across levels (:921-967), 0 on the hanging side of a non-conforming edge / face
    std::vector<long long> key(NP);
    {
        std::vector<long long> global_offset(num_levels, 0);
        for (int l = 1; l < num_levels; l++) global_offset[l] = global_offset[l - 1] + (long long)num_total_elements * npts_of(poly_degree[l - 1]);
        std::vector<int> idx;
        for (int r = 0; r < RE; r++)
        {
            const RegionElem &el = c.sub[r];
            for (int v = 0; v < el.num_points; v++) key[el.offset + v] = c.glo[el.offset + v] + global_offset[el.level];
            for (int v = 0; v < nv; v++)
            {
                const int p = el.offset + corner_index(v, el.n, dim);
                key[p] = c.glo[p];
            }
        }
        for (int r = 0; r < RE; r++)
        {
            const RegionElem &el = c.sub[r];
            for (int eid = 0; eid < num_edges; eid++)
            {
                bool hanging = false;
                for (int rj : edge_map[edge_key(r, eid)])
                    if (rj != r and c.sub[rj].N < el.N) hanging = true;
                if (not hanging) continue;
                edge_points(eid, el.n, dim, idx);
                for (int k = 1; k < el.n - 1; k++) key[el.offset + idx[k]] = 0;
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                bool hanging = false;
                for (int rj : face_map[face_key(r, fid)])
                    if (rj != r and c.sub[rj].N < el.N) hanging = true;
                if (not hanging) continue;
                face_points(fid, el.n, idx);
                for (int b = 1; b < el.n - 1; b++)
                    for (int a = 1; a < el.n - 1; a++) key[el.offset + idx[a + b * el.n]] = 0;
            }
        }
    }
    // classes: 0 regular, 1 interface (degree-1 elements, :1107-1111), 2 extended (:1117-1124), -1 no dof
    std::vector<int> cls(NP, -1);
    for (int r = 0; r < RE; r++)
    {
        const RegionElem &el = c.sub[r];
        for (int v = 0; v < el.num_points; v++)
        {
            const int p = el.offset + v;
            if (key[p] == 0 or not(c.mask[p] > 0.0)) continue;
            const bool is_interface = (el.N == 1) and interface_glo_num.count(c.glo[p]) > 0;
            if (is_interface)
                cls[p] = 1;
            else if (r >= c.num_sub_elems)
                cls[p] = 2;
            else
                cls[p] = 0;
        }
    }
    // A node of an extended element that is not an interface node may still be a regular dof of the subdomain
    // proper (it cannot: the extended ring touches the subdomain only in interface nodes) -- but a key that occurs
    // in two classes takes the lower one, as the shifted ids of the reference would.
    c.point_dof.assign(NP, -1);
    {
        // every distinct key gets a slot (open addressing: 18 million points at C4's size, and a node-based hash map
        // spends 4 s there); the points remember their slot, so class and dof are array reads afterwards
        fdd::KeySlots slots((size_t)NP);
        std::vector<int> slot_of_point(NP, -1);
        std::vector<signed char> slot_class;
        for (int p = 0; p < NP; p++)
            if (cls[p] >= 0)
            {
                const int sl = slots.find_or_insert(key[p]);
                slot_of_point[p] = sl;
                if (sl == (int)slot_class.size())
                    slot_class.push_back((signed char)cls[p]);
                else
                    slot_class[sl] = std::min(slot_class[sl], (signed char)cls[p]);
            }
        const int num_slots = (int)slot_class.size();
        std::vector<int> slot_dof(num_slots, -1);
        // leading dofs: regular keys on the rank's own elements, in the Domain's node order
        int count = 0;
        {
            const int own_points = (int)own_point_node.size();
            std::vector<int> slot_of_node(own_num_nodes, -1);
            for (int p = 0; p < own_points and p < NP; p++)
                if (cls[p] >= 0 and slot_class[slot_of_point[p]] == 0) slot_of_node[own_point_node[p]] = slot_of_point[p];
            for (int nd = 0; nd < own_num_nodes; nd++)
                if (slot_of_node[nd] >= 0 and slot_dof[slot_of_node[nd]] < 0) slot_dof[slot_of_node[nd]] = count++;
            c.num_own_dofs = count;
        }
        for (int want = 0; want < 3; want++)
        {
            std::vector<std::pair<long long, int>> keys; // (key, slot) of the class's keys without a dof yet, ascending key
            for (int sl = 0; sl < num_slots; sl++)
                if (slot_class[sl] == want and slot_dof[sl] < 0) keys.emplace_back(slots.key_of_slot(sl), sl);
            std::sort(keys.begin(), keys.end());
            for (const std::pair<long long, int> &k : keys) slot_dof[k.second] = count++;
            if (want == 1) c.sub_num_dofs = count;
        }
        c.sub_num_ext_dofs = count;
        for (int p = 0; p < NP; p++)
            if (cls[p] >= 0) c.point_dof[p] = slot_dof[slot_of_point[p]];
    }

    phase("numbering");
    // ---- the non-conforming Q (subdomain.tpp:1496-1582) ----
    {
        std::vector<int> idx_i, idx_j;
        for (int p = 0; p < NP; p++)
            if (c.point_dof[p] >= 0)
            {
                c.Q_row.push_back(p);
                c.Q_col.push_back(c.point_dof[p]);
                c.Q_val.push_back(1.0);
            }
        for (int r = 0; r < RE; r++)
        {
            const RegionElem &ei = c.sub[r];
            for (int eid = 0; eid < num_edges; eid++)
            {
                // the lowest-degree element around the edge (:1525-1537)
                int rj = -1, N_j = ei.N;
                const EdgeKey ek = edge_key(r, eid);
                for (int cand : edge_map[ek])
                    if (cand != r and c.sub[cand].N < N_j)
                    {
                        rj = cand;
                        N_j = c.sub[cand].N;
                    }
                if (rj < 0) continue;
                const RegionElem &ej = c.sub[rj];
                int eid_j = -1;
                for (int q = 0; q < num_edges and eid_j < 0; q++)
                    if (edge_key(rj, q) == ek) eid_j = q; // matching_edge, :1179-1348
                if (eid_j < 0) continue;
                EdgeLink link = {r, eid, rj, eid_j};
                c.edge_links.push_back(link);
                edge_points(eid, ei.n, dim, idx_i);
                edge_points(eid_j, ej.n, dim, idx_j);
                const std::vector<double> &J = J_cf.at(std::pair<int, int>(ej.N, ei.N));
                for (int i = 1; i < ei.n - 1; i++)
                    for (int j = 0; j < ej.n; j++)
                    {
                        const int d = c.point_dof[ej.offset + idx_j[j]];
                        if (d < 0) continue;
                        c.Q_row.push_back(ei.offset + idx_i[i]);
                        c.Q_col.push_back(d);
                        c.Q_val.push_back(J[(size_t)i * ej.n + j]);
                    }
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                const FaceKey fk = face_key(r, fid);
                for (int rj : face_map[fk])
                {
                    if (rj == r or not(c.sub[rj].N < ei.N)) continue;
                    const RegionElem &ej = c.sub[rj];
                    int fid_j = -1;
                    for (int q = 0; q < num_faces and fid_j < 0; q++)
                        if (face_key(rj, q) == fk) fid_j = q; // matching_face, :1350-1494
                    if (fid_j < 0) continue;
                    FaceLink link = {r, fid, rj, fid_j};
                    c.face_links.push_back(link);
                    face_points(fid, ei.n, idx_i);
                    face_points(fid_j, ej.n, idx_j);
                    const std::vector<double> &J = J_cf.at(std::pair<int, int>(ej.N, ei.N));
                    for (int j = 1; j < ei.n - 1; j++)
                        for (int i = 1; i < ei.n - 1; i++)
                            for (int q = 0; q < ej.n; q++)
                                for (int pp = 0; pp < ej.n; pp++)
                                {
                                    const int d = c.point_dof[ej.offset + idx_j[pp + q * ej.n]];
                                    if (d < 0) continue;
                                    c.Q_row.push_back(ei.offset + idx_i[i + j * ei.n]);
                                    c.Q_col.push_back(d);
                                    c.Q_val.push_back(J[(size_t)i * ej.n + pp] * J[(size_t)j * ej.n + q]);
                                }
                }
            }
        }
    }

    phase("non-conforming Q");
    // ---- superdomain (subdomain.tpp:1715-2576) ----
    if (c.num_sup_elems > 0)
    {
        // degree-1 element matrices D^T G D with the vertex quadrature (:1715-1846), summed on the coarse dofs
        HostCSR A_coarse;
        {
            const double *Dh = D_hat_coarse.data(); // 2 x 2
            // Dd[d][row * nv + col]: differentiation along direction d on the 2^dim vertices (:1717-1748)
            std::vector<std::vector<double>> Dd(dim, std::vector<double>((size_t)nv * nv, 0.0));
            for (int row = 0; row < nv; row++)
                for (int col = 0; col < nv; col++)
                    for (int d = 0; d < dim; d++)
                    {
                        const int rest_mask = (nv - 1) & ~(1 << d);
                        if ((row & rest_mask) != (col & rest_mask)) continue;
                        Dd[d][(size_t)row * nv + col] = Dh[((row >> d) & 1) * 2 + ((col >> d) & 1)];
                    }
            // symmetric geometric tensor index: 3-D [[0,3,4],[3,1,5],[4,5,2]], 2-D [[0,2],[2,1]]
            const int g3[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}}, g2[2][2] = {{0, 2}, {2, 1}};
            std::vector<int> ti, tj;
            std::vector<double> tv;
            std::vector<double> GD((size_t)dim * nv * nv), A_e((size_t)nv * nv);
            for (int e = 0; e < num_total_elements; e++)
            {
                for (int a = 0; a < dim; a++)
                    for (int i = 0; i < nv; i++)
                        for (int j = 0; j < nv; j++)
                        {
                            double s = 0.0;
                            for (int b = 0; b < dim; b++)
                            {
                                const int g = (dim == 3) ? g3[a][b] : g2[a][b];
                                s += geom_fact_coarse[g][(size_t)e * nv + i] * Dd[b][(size_t)i * nv + j];
                            }
                            GD[((size_t)a * nv + i) * nv + j] = s;
                        }
                for (int i = 0; i < nv; i++)
                    for (int j = 0; j < nv; j++)
                    {
                        // per quadrature point the directions are summed first, then added (:1791, :1825)
                        double s = 0.0;
                        for (int k = 0; k < nv; k++)
                        {
                            double t = Dd[0][(size_t)k * nv + i] * GD[((size_t)0 * nv + k) * nv + j];
                            for (int a = 1; a < dim; a++) t += Dd[a][(size_t)k * nv + i] * GD[((size_t)a * nv + k) * nv + j];
                            s += t;
                        }
                        A_e[(size_t)i * nv + j] = s;
                    }
                for (int i = 0; i < nv; i++)
                    for (int j = 0; j < nv; j++)
                    {
                        const int row = c.dof_num_coarse[(size_t)e * nv + i] - 1, col = c.dof_num_coarse[(size_t)e * nv + j] - 1;
                        const double val = A_e[(size_t)i * nv + j];
                        if (row >= 0 and col >= 0 and std::abs(val) > epsilon)
                        {
                            ti.push_back(row);
                            tj.push_back(col);
                            tv.push_back(val);
                        }
                    }
            }
            A_coarse = low_order::from_triplets(num_coarse_dofs, num_coarse_dofs, ti, tj, tv);
        }

        // dof groups (:1860-1905): 1 subdomain, 2 interface, 3 on the extended ring only, 4 subdomain dofs next to the superdomain
        std::vector<int> dof_marker(num_coarse_dofs, 0);
        for (int r = 0; r < c.num_sub_elems; r++)
        {
            const int eid = c.sub[r].id;
            for (int v = 0; v < nv; v++)
            {
                const int dof = c.dof_num_coarse[(size_t)eid * nv + v];
                if (dof > 0) dof_marker[dof - 1] = 1;
            }
        }
        for (int r = 0; r < c.num_sub_elems; r++)
        {
            const int eid = c.sub[r].id;
            for (int v = 0; v < nv; v++)
            {
                const int dof = c.dof_num_coarse[(size_t)eid * nv + v];
                if (dof > 0 and interface_glo_num.count(glo_num_coarse[(size_t)eid * nv + v])) dof_marker[dof - 1] = 2;
            }
        }
        for (int r = c.num_sub_elems; r < c.num_sub_ext_elems; r++)
        {
            const int eid = c.sub[r].id;
            for (int v = 0; v < nv; v++)
            {
                const int dof = c.dof_num_coarse[(size_t)eid * nv + v];
                if (dof > 0 and dof_marker[dof - 1] == 0) dof_marker[dof - 1] = 3;
            }
        }
        for (int r : c.sup_ext_sub_index)
        {
            const int eid = c.sub[r].id;
            for (int v = 0; v < nv; v++)
            {
                const int dof = c.dof_num_coarse[(size_t)eid * nv + v];
                if (dof > 0 and dof_marker[dof - 1] == 1) dof_marker[dof - 1] = 4;
            }
        }

        HostCSR P_comp, A_comp;
        std::vector<int> comp_of_fine;
        int group[4];
        grade_superdomain(A_coarse, dof_marker, superdomain_overlap, grading, P_comp, A_comp, comp_of_fine, group, c.comp_levels);

        // drop the subdomain's own dofs and order [interface | regular | extended] (:2404-2424)
        const int ncomp = A_comp.rows;
        std::vector<int> R_sup(ncomp, -1);
        int dof = 0;
        const int off1 = group[0], off3 = group[0] + group[1] + group[2], off4 = off3 + group[3];
        for (int i = off1; i < off3; i++) R_sup[i] = dof++;
        for (int i = off4; i < ncomp; i++) R_sup[i] = dof++;
        for (int i = off3; i < off4; i++) R_sup[i] = dof++;
        c.sup_num_ext_dofs = dof;
        c.sup_num_dofs = dof - group[3];
        {
            std::vector<int> ti, tj;
            std::vector<double> tv;
            for (int i = 0; i < ncomp; i++)
                for (int p = A_comp.ptr[i]; p < A_comp.ptr[i + 1]; p++)
                    if (R_sup[i] >= 0 and R_sup[A_comp.col[p]] >= 0)
                    {
                        ti.push_back(R_sup[i]);
                        tj.push_back(R_sup[A_comp.col[p]]);
                        tv.push_back(A_comp.val[p]);
                    }
            c.A_sup = low_order::from_triplets(dof, dof, ti, tj, tv);
        }
        {
            std::vector<int> ti, tj;
            std::vector<double> tv;
            for (int row = 0; row < P_comp.rows; row++)
                for (int p = P_comp.ptr[row]; p < P_comp.ptr[row + 1]; p++)
                    if (R_sup[P_comp.col[p]] >= 0)
                    {
                        ti.push_back(R_sup[P_comp.col[p]]);
                        tj.push_back(row);
                        tv.push_back(P_comp.val[p]);
                    }
            c.Pt_sup = low_order::from_triplets(dof, num_coarse_dofs, ti, tj, tv);
        }
        c.dof_sup.assign(num_coarse_dofs, 0);
        for (int i = 0; i < num_coarse_dofs; i++)
            if (comp_of_fine[i] >= 0 and R_sup[comp_of_fine[i]] >= 0) c.dof_sup[i] = R_sup[comp_of_fine[i]] + 1;
    }
    else
    {
        c.dof_sup.assign(num_coarse_dofs, 0);
        c.A_sup = HostCSR();
        c.Pt_sup = HostCSR();
        c.Pt_sup.cols = num_coarse_dofs;
        c.Pt_sup.ptr.assign(1, 0);
        c.A_sup.ptr.assign(1, 0);
    }

    phase("superdomain");
    // ---- interface operators and weights (subdomain.tpp:2581-2747) ----
    {
        const int ns = c.sub_num_dofs, nse = c.sub_num_ext_dofs, nI = c.num_interface_dofs, nu = c.sup_num_dofs, nue = c.sup_num_ext_dofs;
        c.num_dofs = ns + nu - nI;
        const int shift = ns - nI; // superdomain dof d (1-based) is unique dof d + shift (1-based)
        c.Q_int_col.assign((size_t)nse + nue, -1);
        c.QQt_int_col.assign((size_t)nse + nue, -1);
        for (int i = 0; i < ns; i++)
        {
            c.Q_int_col[i] = i;
            c.QQt_int_col[i] = i;
        }
        for (int r = c.num_sub_elems; r < c.num_sub_ext_elems; r++)
        {
            const RegionElem &el = c.sub[r];
            for (int v = 0; v < el.num_points; v++)
            {
                // extended elements have degree 1: point v is vertex v
                const int d = c.point_dof[el.offset + v];
                const int cd = c.dof_num_coarse[(size_t)el.id * nv + v];
                if (d < 0 or cd <= 0 or c.dof_sup[cd - 1] <= 0) continue;
                if (d < ns)
                {
                    // an interface node seen from the extended ring: the reference overwrites the entry with the same
                    // value, because both regions number their interface dofs in ascending global id
                    if (c.dof_sup[cd - 1] + shift - 1 != d) fprintf(stderr, "WARNING: composite: interface dof %d is superdomain dof %d (expected %d)\n", d, c.dof_sup[cd - 1], d - shift + 1);
                    continue;
                }
                c.Q_int_col[d] = c.dof_sup[cd - 1] + shift - 1;
                if (c.QQt_int_col[d] < 0) c.QQt_int_col[d] = nse + c.dof_sup[cd - 1] - 1;
            }
        }
        for (int i = 0; i < nu; i++) c.Q_int_col[(size_t)nse + i] = i + shift;
        for (int i = 0; i < nI; i++) c.QQt_int_col[(size_t)nse + i] = ns - nI + i;
        for (int i = nI; i < nu; i++) c.QQt_int_col[(size_t)nse + i] = nse + i;
        for (int r : c.sup_ext_sub_index)
        {
            const RegionElem &el = c.sub[r];
            for (int v = 0; v < el.num_points; v++)
            {
                const int cd = c.dof_num_coarse[(size_t)el.id * nv + v];
                if (cd <= 0) continue;
                const int ds = c.dof_sup[cd - 1];
                if (ds <= nu) continue; // only the extended ones (marker 4)
                const int d = c.point_dof[el.offset + v]; // these elements have degree 1: point v is vertex v
                c.Q_int_col[(size_t)nse + ds - 1] = d;
                if (c.QQt_int_col[(size_t)nse + ds - 1] < 0) c.QQt_int_col[(size_t)nse + ds - 1] = d;
            }
        }
        c.Qt_int_col.assign(c.num_dofs, -1);
        for (int i = 0; i < ns; i++) c.Qt_int_col[i] = i;
        for (int i = 0; i < nu - nI; i++) c.Qt_int_col[ns + i] = nse + nI + i;

        c.norm_weight.assign((size_t)nse + nue, 1.0);
        for (int i = ns; i < nse; i++) c.norm_weight[i] = 0.0;
        for (int i = 0; i < nI; i++) c.norm_weight[(size_t)nse + i] = 0.0;
        for (int i = nu; i < nue; i++) c.norm_weight[(size_t)nse + i] = 0.0;
    }
    phase("interface maps and weights");
    return c;
}


// ---------------------------------------------------------------------------------------------------------------
// Low-order operator of the composite (subdomain.tpp:2749-3472): the matrix the AMG V-cycle of
// Subdomain::low_order_preconditioner runs on, over the composite's unique dofs.
//   subdomain part   every region element contributes its low-order matrix A_e -- P1 finite elements on the 6
//                    tetrahedra of every GLL sub-cell for degree > 1 (:2932-3039), the degree-1 spectral matrix
//                    itself otherwise (:3040-3125) -- through J_e^T A_e J_e (:3278-3409), J_e = the element's rows of
//                    the non-conforming Q with the PIECEWISE-LINEAR interpolation J_cf_fem (:2754-2783) on hanging
//                    edges and faces.  Here: sum_e J_e^T A_e J_e = Q_fem^T blockdiag(A_e) Q_fem, evaluated as two
//                    sparse products for the elements that have hanging points and by direct summation on the dofs
//                    for all the others (the bulk);
//   superdomain part the rows of the superdomain operator A of its regular dofs (:3444-3469);
// both renumbered to the unique dofs through Q_int (:3414-3417).
// ---------------------------------------------------------------------------------------------------------------
inline std::vector<double> interpolator_fem(int N_c, int N_f, const std::vector<double> &r_c, const std::vector<double> &r_f)
{
    const int n_c = N_c + 1, n_f = N_f + 1;
    std::vector<double> J((size_t)n_f * n_c, 0.0);
    J[0] = 1.0;
    for (int i = 1; i < N_f; i++)
        for (int j = 0; j < N_c; j++)
            if (r_c[j] <= r_f[i] and r_f[i] <= r_c[j + 1])
            {
                J[(size_t)i * n_c + j] = (r_c[j + 1] - r_f[i]) / (r_c[j + 1] - r_c[j]);
                J[(size_t)i * n_c + j + 1] = (r_f[i] - r_c[j]) / (r_c[j + 1] - r_c[j]);
            }
    J[(size_t)(n_f - 1) * n_c + (n_c - 1)] = 1.0;
    return J;
}

// gll_nodes[l]: the GLL nodes of level l; D_hat_coarse: the 2 x 2 differentiation matrix of degree 1
inline HostCSR assemble_low_order(const Composite &c, const std::vector<std::vector<double>> &gll_nodes, const std::vector<double> &D_hat_coarse, double epsilon)
{
    const int dim = c.dim, nv = c.num_vertices;
    const int RE = c.num_sub_ext_elems, NP = c.num_sub_ext_points, nse = c.sub_num_ext_dofs;
    if (dim != 3)
    {
        fprintf(stderr, "ERROR: the composite low-order operator is assembled for 3-D regions\n");
        exit(EXIT_FAILURE);
    }
    static const int tets[6][4][3] = {{{0, 0, 0}, {0, 1, 0}, {1, 0, 0}, {1, 0, 1}}, {{1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 1}}, {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 0, 1}},
                                      {{1, 0, 1}, {1, 1, 0}, {1, 1, 1}, {0, 1, 0}}, {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 1, 0}}, {{1, 0, 1}, {1, 1, 1}, {0, 1, 1}, {0, 1, 0}}}; // subdomain.tpp:2877-2882
    static const double Dref[3][4] = {{1.0, 0.0, 0.0, -1.0}, {0.0, 1.0, 0.0, -1.0}, {0.0, 0.0, 1.0, -1.0}}; // :2840-2842

    // elements with a hanging edge or face go through Q_fem^T A Q_fem, the others straight to the dofs
    std::vector<char> hanging(RE, 0);
    for (const EdgeLink &l : c.edge_links) hanging[l.elem_i] = 1;
    for (const FaceLink &l : c.face_links) hanging[l.elem_i] = 1;

    // degree-1 differentiation on the 8 vertices (as in build(), :1715-1748)
    std::vector<std::vector<double>> Dd(dim, std::vector<double>((size_t)nv * nv, 0.0));
    for (int row = 0; row < nv; row++)
        for (int col = 0; col < nv; col++)
            for (int d = 0; d < dim; d++)
            {
                const int rest_mask = (nv - 1) & ~(1 << d);
                if ((row & rest_mask) != (col & rest_mask)) continue;
                Dd[d][(size_t)row * nv + col] = D_hat_coarse[((row >> d) & 1) * 2 + ((col >> d) & 1)];
            }
    const int g3[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};

    // the local matrix of element r: entries (li, lj, value) handed to emit()
    auto element_matrix = [&](int r, std::vector<double> &K, std::vector<unsigned char> &touched, const std::function<void(int, int, double)> &emit) {
        const RegionElem &el = c.sub[r];
        const int N = el.N, n = el.n, n3 = el.num_points;
        const size_t base = (size_t)el.offset;
        if (N > 1)
        {
            K.assign((size_t)n3 * 27, 0.0);
            touched.assign((size_t)n3 * 27, (unsigned char)0);
            for (int sz = 0; sz < N; sz++)
                for (int sy = 0; sy < N; sy++)
                    for (int sx = 0; sx < N; sx++)
                        for (int t = 0; t < 6; t++)
                        {
                            int loc[4];
                            double xs[4], ys[4], zs[4];
                            for (int v = 0; v < 4; v++)
                            {
                                loc[v] = (sx + tets[t][v][0]) + (sy + tets[t][v][1]) * n + (sz + tets[t][v][2]) * n * n;
                                xs[v] = c.x[base + loc[v]];
                                ys[v] = c.y[base + loc[v]];
                                zs[v] = c.z[base + loc[v]];
                            }
                            const double H[9] = {xs[0] - xs[3], xs[1] - xs[3], xs[2] - xs[3], ys[0] - ys[3], ys[1] - ys[3], ys[2] - ys[3], zs[0] - zs[3], zs[1] - zs[3], zs[2] - zs[3]};
                            const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
                            const double id = 1.0 / det;
                            const double iH[9] = {id * (H[4] * H[8] - H[7] * H[5]), id * (H[2] * H[7] - H[8] * H[1]), id * (H[1] * H[5] - H[4] * H[2]),
                                                  id * (H[5] * H[6] - H[8] * H[3]), id * (H[0] * H[8] - H[6] * H[2]), id * (H[2] * H[3] - H[5] * H[0]),
                                                  id * (H[3] * H[7] - H[6] * H[4]), id * (H[1] * H[6] - H[7] * H[0]), id * (H[0] * H[4] - H[3] * H[1])};
                            double G[3][3];
                            for (int m = 0; m < 3; m++)
                                for (int nn = 0; nn < 3; nn++)
                                {
                                    double g = 0.0;
                                    for (int k = 0; k < 3; k++) g += (det / 24.0) * iH[m * 3 + k] * iH[nn * 3 + k];
                                    G[m][nn] = g;
                                }
                            for (int i = 0; i < 4; i++)
                                for (int j = 0; j < 4; j++)
                                {
                                    double a = 0.0;
                                    for (int m = 0; m < 3; m++)
                                        for (int nn = 0; nn < 3; nn++)
                                            for (int q = 0; q < 4; q++) a += Dref[m][i] * (G[m][nn] * Dref[nn][j]);
                                    if (not(std::abs(a) > epsilon)) continue; // :3025
                                    const int d0 = tets[t][j][0] - tets[t][i][0], d1 = tets[t][j][1] - tets[t][i][1], d2 = tets[t][j][2] - tets[t][i][2];
                                    const size_t slot = (size_t)loc[i] * 27 + (size_t)((d0 + 1) + 3 * (d1 + 1) + 9 * (d2 + 1));
                                    K[slot] += a;
                                    touched[slot] = 1;
                                }
                        }
            for (int li = 0; li < n3; li++)
                for (int s = 0; s < 27; s++)
                    if (touched[(size_t)li * 27 + s]) emit(li, li + (s % 3 - 1) + ((s / 3) % 3 - 1) * n + (s / 9 - 1) * n * n, K[(size_t)li * 27 + s]);
        }
        else
        {
            // the degree-1 spectral element matrix D^T G D (:3082-3123)
            const int go = c.deg1_offset[r];
            double GD[3][64];
            for (int a = 0; a < dim; a++)
                for (int i = 0; i < nv; i++)
                    for (int j = 0; j < nv; j++)
                    {
                        double s = 0.0;
                        for (int b = 0; b < dim; b++) s += c.G_deg1[g3[a][b]][(size_t)go + i] * Dd[b][(size_t)i * nv + j];
                        GD[a][i * nv + j] = s;
                    }
            for (int i = 0; i < nv; i++)
                for (int j = 0; j < nv; j++)
                {
                    double val = 0.0;
                    for (int k = 0; k < nv; k++)
                    {
                        double t = Dd[0][(size_t)k * nv + i] * GD[0][k * nv + j];
                        for (int a = 1; a < dim; a++) t += Dd[a][(size_t)k * nv + i] * GD[a][k * nv + j];
                        val += t;
                    }
                    if (std::abs(val) > epsilon) emit(i, j, val);
                }
        }
    };

    // (1) conforming elements: straight onto the dofs, element ranges on the host threads
    HostCSR A_direct;
    {
        const int parts = low_order::range_parts(RE);
        std::vector<std::vector<int>> pti(parts), ptj(parts);
        std::vector<std::vector<double>> ptv(parts);
        low_order::parallel_ranges(RE, parts, [&](long long e0, long long e1, int part) {
            std::vector<int> ti, tj;
            std::vector<double> tv;
            std::vector<double> K;
            std::vector<unsigned char> touched;
            for (long long r = e0; r < e1; r++)
            {
                if (hanging[r]) continue;
                const int off = c.sub[r].offset;
                element_matrix((int)r, K, touched, [&](int li, int lj, double v) {
                    const int di = c.point_dof[off + li], dj = c.point_dof[off + lj];
                    if (di < 0 or dj < 0) return;
                    ti.push_back(di);
                    tj.push_back(dj);
                    tv.push_back(v);
                });
            }
            pti[part] = std::move(ti);
            ptj[part] = std::move(tj);
            ptv[part] = std::move(tv);
        });
        std::vector<int> ti, tj;
        std::vector<double> tv;
        for (int t = 0; t < parts; t++)
        {
            ti.insert(ti.end(), pti[t].begin(), pti[t].end());
            tj.insert(tj.end(), ptj[t].begin(), ptj[t].end());
            tv.insert(tv.end(), ptv[t].begin(), ptv[t].end());
            std::vector<int>().swap(pti[t]);
            std::vector<int>().swap(ptj[t]);
            std::vector<double>().swap(ptv[t]);
        }
        A_direct = low_order::from_triplets(nse, nse, ti, tj, tv);
    }

    // (2) elements with hanging points: Q_fem^T A_blk Q_fem on their points (compressed to those elements)
    HostCSR A_hang;
    {
        std::vector<int> local_of(NP, -1); // compressed point index over the hanging elements
        int nloc = 0;
        for (int r = 0; r < RE; r++)
            if (hanging[r])
                for (int v = 0; v < c.sub[r].num_points; v++) local_of[c.sub[r].offset + v] = nloc++;
        if (nloc > 0)
        {
            std::vector<int> ti, tj;
            std::vector<double> tv;
            std::vector<double> K;
            std::vector<unsigned char> touched;
            for (int r = 0; r < RE; r++)
            {
                if (not hanging[r]) continue;
                const int lo = local_of[c.sub[r].offset];
                element_matrix(r, K, touched, [&](int li, int lj, double v) {
                    ti.push_back(lo + li);
                    tj.push_back(lo + lj);
                    tv.push_back(v);
                });
            }
            HostCSR A_blk = low_order::from_triplets(nloc, nloc, ti, tj, tv);

            // Q_fem rows of those points: direct dofs, and J_cf_fem rows on hanging edges / faces (:3287-3355)
            std::vector<int> qi, qj;
            std::vector<double> qv;
            for (int r = 0; r < RE; r++)
                if (hanging[r])
                    for (int v = 0; v < c.sub[r].num_points; v++)
                    {
                        const int p = c.sub[r].offset + v;
                        if (c.point_dof[p] >= 0)
                        {
                            qi.push_back(local_of[p]);
                            qj.push_back(c.point_dof[p]);
                            qv.push_back(1.0);
                        }
                    }
            std::vector<int> idx_i, idx_j;
            std::map<std::pair<int, int>, std::vector<double>> Jf;
            auto J_of = [&](const RegionElem &ej, const RegionElem &ei) -> const std::vector<double> & {
                const std::pair<int, int> key(ej.N, ei.N);
                auto it = Jf.find(key);
                if (it == Jf.end()) it = Jf.emplace(key, interpolator_fem(ej.N, ei.N, gll_nodes[ej.level], gll_nodes[ei.level])).first;
                return it->second;
            };
            for (const EdgeLink &l : c.edge_links)
            {
                const RegionElem &ei = c.sub[l.elem_i], &ej = c.sub[l.elem_j];
                edge_points(l.eid, ei.n, dim, idx_i);
                edge_points(l.eid_j, ej.n, dim, idx_j);
                const std::vector<double> &J = J_of(ej, ei);
                for (int i = 1; i < ei.n - 1; i++)
                    for (int j = 0; j < ej.n; j++)
                    {
                        const int d = c.point_dof[ej.offset + idx_j[j]];
                        const double w = J[(size_t)i * ej.n + j];
                        if (d < 0 or not(std::abs(w) > epsilon)) continue;
                        qi.push_back(local_of[ei.offset + idx_i[i]]);
                        qj.push_back(d);
                        qv.push_back(w);
                    }
            }
            for (const FaceLink &l : c.face_links)
            {
                const RegionElem &ei = c.sub[l.elem_i], &ej = c.sub[l.elem_j];
                face_points(l.fid, ei.n, idx_i);
                face_points(l.fid_j, ej.n, idx_j);
                const std::vector<double> &J = J_of(ej, ei);
                for (int j = 1; j < ei.n - 1; j++)
                    for (int i = 1; i < ei.n - 1; i++)
                        for (int q = 0; q < ej.n; q++)
                            for (int pp = 0; pp < ej.n; pp++)
                            {
                                const int d = c.point_dof[ej.offset + idx_j[pp + q * ej.n]];
                                const double w = J[(size_t)i * ej.n + pp] * J[(size_t)j * ej.n + q];
                                if (d < 0 or not(std::abs(w) > epsilon)) continue;
                                qi.push_back(local_of[ei.offset + idx_i[i + j * ei.n]]);
                                qj.push_back(d);
                                qv.push_back(w);
                            }
            }
            HostCSR Qf = low_order::from_triplets(nloc, nse, qi, qj, qv);
            HostCSR AQ = low_order::multiply(A_blk, Qf);
            HostCSR Qft = low_order::transpose(Qf);
            A_hang = low_order::multiply(Qft, AQ);
        }
    }

    // (3) the combined operator on the unique dofs (:3419-3472)
    std::vector<int> ti, tj;
    std::vector<double> tv;
    const int ns = c.sub_num_dofs, nI = c.num_interface_dofs, nu = c.sup_num_dofs;
    auto add_rows = [&](const HostCSR &M) {
        if (M.rows == 0) return;
        for (int i = 0; i < ns; i++)
            for (int p = M.ptr[i]; p < M.ptr[i + 1]; p++)
            {
                if (not(std::abs(M.val[p]) > epsilon)) continue; // :3395
                ti.push_back(c.Q_int_col[i]);
                tj.push_back(c.Q_int_col[M.col[p]]);
                tv.push_back(M.val[p]);
            }
    };
    add_rows(A_direct);
    add_rows(A_hang);
    for (int i = nI; i < nu; i++)
        for (int p = c.A_sup.ptr[i]; p < c.A_sup.ptr[i + 1]; p++)
        {
            ti.push_back(c.Q_int_col[(size_t)nse + i]);
            tj.push_back(c.Q_int_col[(size_t)nse + c.A_sup.col[p]]);
            tv.push_back(c.A_sup.val[p]);
        }
    return low_order::from_triplets(c.num_dofs, c.num_dofs, ti, tj, tv);
}

// The rows of the level-0 (degree-N) region points over the unique dofs, as assemble_low_order uses them: one unit
// entry on a conforming point, nothing on a Dirichlet point, the piecewise-linear constraint row (Q_fem, :3287-3355)
// on a point hanging on a lower-degree neighbour.  This is the lattice the geometric levels of the hierarchy coarsen
// (low_order.hpp); the degree-N elements are the region's first elements, one after the other.
inline HostCSR lattice_rows(const Composite &c, const std::vector<double> &gll_fine, double epsilon, long long &num_elements)
{
    const int dim = c.dim, nse = c.sub_num_ext_dofs;
    const int first = c.level_first_elem[0], count = c.level_num_elems[0];
    num_elements = count;
    HostCSR R;
    if (count == 0) return R;
    const int base = c.sub[first].offset, np = c.sub[first].num_points;
    const long long total = (long long)count * np;
    std::vector<int> qi, qj;
    std::vector<double> qv;
    auto unique_dof = [&](int ext) { return ext >= 0 ? c.Q_int_col[ext] : -1; };
    for (long long q = 0; q < total; q++)
    {
        const int u = unique_dof(c.point_dof[base + q]);
        if (u < 0) continue;
        qi.push_back((int)q);
        qj.push_back(u);
        qv.push_back(1.0);
    }
    auto is_fine = [&](int r) { return r >= first and r < first + count; };
    std::vector<int> idx_i, idx_j;
    std::map<int, std::vector<double>> Jf; // by the neighbour's degree
    auto J_of = [&](const RegionElem &ej, const RegionElem &ei) -> const std::vector<double> & {
        auto it = Jf.find(ej.N);
        if (it == Jf.end())
        {
            std::vector<double> coarse_nodes(ej.n), w(ej.n);
            gll::zwgll(coarse_nodes.data(), w.data(), ej.n);
            it = Jf.emplace(ej.N, interpolator_fem(ej.N, ei.N, coarse_nodes, gll_fine)).first;
        }
        return it->second;
    };
    for (const EdgeLink &l : c.edge_links)
    {
        if (not is_fine(l.elem_i)) continue;
        const RegionElem &ei = c.sub[l.elem_i], &ej = c.sub[l.elem_j];
        edge_points(l.eid, ei.n, dim, idx_i);
        edge_points(l.eid_j, ej.n, dim, idx_j);
        const std::vector<double> &J = J_of(ej, ei);
        for (int i = 1; i < ei.n - 1; i++)
            for (int j = 0; j < ej.n; j++)
            {
                const int d = unique_dof(c.point_dof[ej.offset + idx_j[j]]);
                const double w = J[(size_t)i * ej.n + j];
                if (d < 0 or not(std::abs(w) > epsilon)) continue;
                qi.push_back(ei.offset + idx_i[i] - base);
                qj.push_back(d);
                qv.push_back(w);
            }
    }
    if (dim == 3)
        for (const FaceLink &l : c.face_links)
        {
            if (not is_fine(l.elem_i)) continue;
            const RegionElem &ei = c.sub[l.elem_i], &ej = c.sub[l.elem_j];
            face_points(l.fid, ei.n, idx_i);
            face_points(l.fid_j, ej.n, idx_j);
            const std::vector<double> &J = J_of(ej, ei);
            for (int j = 1; j < ei.n - 1; j++)
                for (int i = 1; i < ei.n - 1; i++)
                    for (int q = 0; q < ej.n; q++)
                        for (int pp = 0; pp < ej.n; pp++)
                        {
                            const int d = unique_dof(c.point_dof[ej.offset + idx_j[pp + q * ej.n]]);
                            const double w = J[(size_t)i * ej.n + pp] * J[(size_t)j * ej.n + q];
                            if (d < 0 or not(std::abs(w) > epsilon)) continue;
                            qi.push_back(ei.offset + idx_i[i + j * ei.n] - base);
                            qj.push_back(d);
                            qv.push_back(w);
                        }
        }
    (void)nse;
    return low_order::from_triplets((int)total, c.num_dofs, qi, qj, qv);
}

} // namespace composite
} // namespace fdd

#endif
