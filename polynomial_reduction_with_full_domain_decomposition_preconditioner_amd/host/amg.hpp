/*
 * amg.hpp -- the low-order AMG V-cycle of the FDD preconditioner
 * (Subdomain::low_order_preconditioner, subdomain.tpp:3987-4159; smoother
 * pieces subdomain.tpp:19-83; containers AMG/csr_matrix.*, AMG/vector.*).
 *
 * The reference gets the hierarchy from HYPRE BoomerAMG (A_l, the Chebyshev
 * diagonal scaling and coefficients, P_l; subdomain.tpp:3474-3549), keeps
 * levels <= level_cutoff on the GPU and the rest on the host, solves the
 * coarsest level with hypre_GaussElimSolve on the host, and replays the two
 * device legs as CUDA graphs.  Here the hierarchy is handed in level by level
 * (HYPRE is not available: documented deviation), EVERY level lives
 * in HBM, the coarsest level is a dense inverse applied as one SpMV (computed
 * once on the host by Gaussian elimination), and the whole V-cycle -- not two
 * legs with host work in between -- is one hipGraph.
 *
 * Kernels: fdd_amg_* (AMG/kernels.cu), SpMV y = alpha*A*x + beta*y on the
 * LDS-staged row-block plan (cusparseSpMV, AMG/csr_matrix.cpp:129-131).
 */
#ifndef FDD_AMG_HPP
#define FDD_AMG_HPP

#include <chrono>
#include <map>
#include <utility>
#include <vector>

#include "config.hpp"
#include "csr_matrix.hpp"
#include "low_order.hpp"

namespace amg
{

// matrix-free form of a geometric level's interpolator (fdd_lattice_prolong / _restrict, csrc/fdd_transfer.hip): the maps
// of low_order::Transfer on the device, the element-local partial sums of the restriction and the boolean gather that adds
// the partial sums of a coarse dof
struct LatticeTransfer
{
    bool active = false;
    int n = 0, m = 0;
    std::vector<int> lo, hi;
    std::vector<double> wl;
    long long num_elements = 0;
    fdd::memory owner_dof, coarse_dof, partial, partial32;
    CSR_Matrix<double> gather; // coarse dofs x kept nodes of all elements
    fdd::memory gather_val32;
    fdd_csr_plan *gather_plan32 = nullptr;
    double prolong_bytes(int n_fine, int n_coarse, int vb) const { return 4.0 * (double)num_elements * n * n * n + 2.0 * vb * n_fine + (4.0 + vb) * (double)num_elements * m * m * m + 0.0 * n_coarse; }
    double restrict_bytes(int n_fine, int vb) const { return 4.0 * (double)num_elements * n * n * n + 1.0 * vb * n_fine + 1.0 * vb * (double)num_elements * m * m * m; }
};

struct Level
{
    int n = 0;
    LatticeTransfer T;
    CSR_Matrix<double> A;
    CSR_Matrix<double> P; // n x n_coarse
    CSR_Matrix<double> R; // P^T
    fdd::memory D_val;
    std::vector<double> coefs; // Chebyshev coefficients, coefs[p] multiplies (DAD)^p
    fdd::memory f, u, r, v, w, work;
    // Float = float (AMG/config.hpp:4): f32 copies of the values and vectors, row-block plans for the f32 SpMV
    std::vector<double> D_hst;
    fdd::memory A_val32, P_val32, R_val32, D_val32;
    fdd::memory f32, u32, r32, v32, work32;
    fdd_csr_plan *A_plan32 = nullptr, *P_plan32 = nullptr, *R_plan32 = nullptr;
};

class Hierarchy
{
  private:
    bool finalized = false;
    // one captured graph per OUTPUT vector of the cycle (vcycle_into: the caller's Krylov vectors stand in for
    // levels[0].u, so that the correction is written where it is wanted instead of being copied there)
    std::map<const void *, void *> graphs;
    bool graph_failed = false;

    void destroy_graphs()
    {
        for (auto &kv : graphs)
            if (kv.second != nullptr) (void)fdd_graph_destroy(kv.second);
        graphs.clear();
    }
    CSR_Matrix<double> coarse_inverse;

    // Chebyshev smoother, device branches of subdomain.tpp:19-83
    // u_is_zero: the level's u was just set to 0 (every pre-smoothing), so f - A u is f itself
    // bit for bit and the SpMV of scaled_residual is skipped (the reference multiplies by the zero vector)
    void smooth(int l, bool u_is_zero = false)
    {
        Level &L = levels[l];
        void *s = fdd::dev().stream;
        if (fused_smoother and cheby_order >= 2)
        {
            // the same statements as below, with the element-wise kernels running as epilogues of the SpMV in front
            // of them (bit-identical): cheby_order SpMV launches (one fewer, plus one element-wise, from u = 0) and
            // no vector in between goes through HBM.  work and v alternate as the SpMV operand.
            fdd::memory *in = &L.work, *out = &L.v;
            if (u_is_zero)
                FDD_CALL(fdd_amg_smooth_start(in->as<double>(), L.r.as<double>(), L.f.as<double>(), L.D_val.as<double>(), L.coefs[cheby_order - 1], L.n, s));
            else
                L.A.smooth_residual(*in, L.r, L.u, L.f, L.D_val, L.coefs[cheby_order - 1]);
            for (int p = cheby_order - 2; p >= 1; p--)
            {
                L.A.smooth_polynomial(*out, *in, L.r, L.D_val, L.coefs[p]);
                std::swap(in, out);
            }
            if (u_is_zero)
                L.A.smooth_update_from_zero(L.u, *in, L.r, L.D_val, L.coefs[0]); // u = 0 + D w: u is not read, and was not zeroed (vcycle_launches)
            else
                L.A.smooth_update(L.u, *in, L.r, L.D_val, L.coefs[0]);
            return;
        }
        // scaled_residual (:34-39): work = f - A u without the copy "work = f" in front of the SpMV; from u = 0 it is f
        if (not u_is_zero) L.A.matvec_to(L.work, L.f, L.u, -1.0, 1.0);
        FDD_CALL(fdd_amg_main_scaled_residual(L.r.as<double>(), L.w.as<double>(), (u_is_zero ? L.f : L.work).as<double>(), L.D_val.as<double>(), L.coefs[cheby_order - 1], L.n, s));
        // polynomial_evaluation (:62-67)
        for (int p = cheby_order - 2; p >= 0; p--)
        {
            FDD_CALL(fdd_amg_vector_multiplication(L.work.as<double>(), L.D_val.as<double>(), L.w.as<double>(), L.n, s));
            L.A.matvec(L.v, L.work, 1.0, 0.0);
            FDD_CALL(fdd_amg_main_polynomial_evaluation(L.w.as<double>(), L.v.as<double>(), L.r.as<double>(), L.D_val.as<double>(), L.coefs[p], L.n, s));
        }
        // update_field (:79-82)
        FDD_CALL(fdd_amg_main_update_field(L.u.as<double>(), L.w.as<double>(), L.D_val.as<double>(), L.n, s));
    }

    // subdomain.tpp:4015-4139, every level on the device
    void vcycle_launches()
    {
        const int nl = (int)levels.size();
        void *s = fdd::dev().stream;
        // the fused pre-smoother writes u = 0 + D w without reading u: the reference's zeroing of u (:4012, :4026, :4058)
        // would be 8 B per row written and read back for nothing
        const bool smoother_writes_u = fused_smoother and cheby_order >= 2;
        if (not smoother_writes_u or nl == 1) FDD_CALL(fdd_amg_vector_set_to_value(levels[0].u.as<double>(), 0.0, levels[0].n, s)); // :4012
        for (int iter = 0; iter < num_vcycles; iter++)
        {
            for (int l = 0; l < nl - 1; l++)
            {
                Level &L = levels[l];
                if (l > 0 and not smoother_writes_u) FDD_CALL(fdd_amg_vector_set_to_value(L.u.as<double>(), 0.0, L.n, s));
                smooth(l, l > 0 or iter == 0);
                L.A.matvec_to(L.v, L.f, L.u, -1.0, 1.0); // v = f - A u
                if (L.T.active and matrix_free_transfer)
                {
                    {
                        fdd::ProfileScope prof("lattice_restrict_kernel", L.T.restrict_bytes(L.n, 8));
                        FDD_CALL(fdd_lattice_restrict(L.T.partial.as<double>(), L.v.as<double>(), L.T.owner_dof.as<int>(), L.T.n, L.T.m, L.T.lo.data(), L.T.hi.data(), L.T.wl.data(), L.T.num_elements, s));
                    }
                    L.T.gather.matvec(levels[l + 1].f, L.T.partial, 1.0, 0.0);
                }
                else
                    L.R.matvec(levels[l + 1].f, L.v, 1.0, 0.0);
            }
            Level &C = levels[nl - 1];
            coarse_inverse.matvec(C.u, C.f, 1.0, 0.0); // hypre_GaussElimSolve's role (:4084)
            for (int l = nl - 1; l > 0; l--)
            {
                Level &F = levels[l - 1];
                if (F.T.active and matrix_free_transfer)
                {
                    fdd::ProfileScope prof("lattice_prolong_kernel", F.T.prolong_bytes(F.n, levels[l].n, 8));
                    FDD_CALL(fdd_lattice_prolong(F.u.as<double>(), levels[l].u.as<double>(), F.T.owner_dof.as<int>(), F.T.coarse_dof.as<int>(), F.T.n, F.T.m, F.T.lo.data(), F.T.hi.data(), F.T.wl.data(), F.T.num_elements, s));
                }
                else
                    F.P.matvec(F.u, levels[l].u, 1.0, 1.0); // u_{l-1} += P u_l
                smooth(l - 1);
            }
        }
    }

    // ---------------------------------------------------------------------------------------------
    // Float = float (AMG/config.hpp:4, swept by run.py:157): the same cycle on f32 values and vectors.
    // The residual is cast down on entry and the correction cast up on exit (the reference copies Float
    // data at subdomain.tpp:4008,4142); fused smoother sequence only.
    // ---------------------------------------------------------------------------------------------
    bool ready32 = false;
    fdd::memory coarse_inverse_val32;
    fdd_csr_plan *coarse_plan32 = nullptr;

    template <typename Vec>
    static fdd::memory to_f32(const Vec &v)
    {
        std::vector<float> t(v.begin(), v.end());
        fdd::memory m = fdd::dev().malloc<float>(std::max<size_t>(t.size(), 1));
        if (not t.empty()) m.copyFrom(t.data(), t.size() * sizeof(float));
        return m;
    }

    // plan of the f32 entries; with short even rows it gets the sliced-ELL copy like the fp64 plans (csr_matrix.hpp)
    static void blocked_plan(fdd_csr_plan **plan, CSR_Matrix<double> &M, fdd::memory &val32)
    {
        if (M.num_rows == 0 or M.num_cols == 0 or M.ptr_hst.empty()) return;
        FDD_CALL(fdd_csr_plan_create_f32(plan, M.ptr_hst.data(), M.num_rows, M.num_cols, M.num_nnz));
        int attached = 0;
        if (M.num_nnz > M.num_rows) FDD_CALL(fdd_csr_plan_attach_sell(*plan, M.ptr_hst.data(), M.ptr.as<int>(), M.col.as<int>(), val32.ptr(), 1.3, &attached, fdd::dev().stream));
    }

    void prepare32()
    {
        if (ready32) return;
        for (Level &L : levels)
        {
            L.A_val32 = to_f32(L.A.val_hst);
            blocked_plan(&L.A_plan32, L.A, L.A_val32);
            if (L.P.num_rows > 0 and not L.P.ptr_hst.empty())
            {
                L.P_val32 = to_f32(L.P.val_hst);
                L.R_val32 = to_f32(L.R.val_hst);
                blocked_plan(&L.P_plan32, L.P, L.P_val32);
                blocked_plan(&L.R_plan32, L.R, L.R_val32);
            }
            if (L.T.active)
            {
                L.T.partial32 = fdd::dev().malloc<float>((size_t)L.T.gather.num_cols);
                L.T.gather_val32 = to_f32(L.T.gather.val_hst);
                blocked_plan(&L.T.gather_plan32, L.T.gather, L.T.gather_val32);
            }
            L.D_val32 = to_f32(L.D_hst);
            for (fdd::memory *m : {&L.f32, &L.u32, &L.r32, &L.v32, &L.work32}) *m = fdd::dev().malloc<float>(L.n);
        }
        coarse_inverse_val32 = to_f32(coarse_inverse.val_hst);
        blocked_plan(&coarse_plan32, coarse_inverse, coarse_inverse_val32);
        ready32 = true;
    }

    // y = alpha*A*x + beta*y_in on f32 data
    static void matvec32(fdd_csr_plan *plan, CSR_Matrix<double> &M, fdd::memory &val32, fdd::memory &y, fdd::memory *y_in, fdd::memory &x, float alpha, float beta)
    {
        fdd::ProfileScope prof("csr_block_kernel<f32>", 8.0 * M.num_nnz + 8.0 * M.num_rows + 4.0 * M.num_cols + (beta != 0.0f ? 4.0 * M.num_rows : 0.0));
        FDD_CALL(fdd_csr_plan_matvec_to_f32(plan, y.as<float>(), y_in ? y_in->as<float>() : nullptr, M.ptr.as<int>(), M.col.as<int>(), val32.as<float>(), x.as<float>(), alpha, beta, fdd::dev().stream));
    }

    void smooth32(int l, bool u_is_zero)
    {
        Level &L = levels[l];
        void *s = fdd::dev().stream;
        const int *ptr = L.A.ptr.as<int>(), *col = L.A.col.as<int>();
        const float *val = L.A_val32.as<float>(), *D = L.D_val32.as<float>();
        fdd::memory *in = &L.work32, *out = &L.v32;
        if (u_is_zero)
            FDD_CALL(fdd_amg_smooth_start_f32(in->as<float>(), L.r32.as<float>(), L.f32.as<float>(), D, (float)L.coefs[cheby_order - 1], L.n, s));
        else
            FDD_CALL(fdd_amg_smooth_residual_matvec_f32(L.A_plan32, in->as<float>(), L.r32.as<float>(), ptr, col, val, L.u32.as<float>(), L.f32.as<float>(), D, (float)L.coefs[cheby_order - 1], s));
        for (int p = cheby_order - 2; p >= 1; p--)
        {
            FDD_CALL(fdd_amg_smooth_polynomial_matvec_f32(L.A_plan32, out->as<float>(), ptr, col, val, in->as<float>(), L.r32.as<float>(), D, (float)L.coefs[p], s));
            std::swap(in, out);
        }
        if (u_is_zero)
            FDD_CALL(fdd_amg_smooth_update_matvec_from_zero_f32(L.A_plan32, L.u32.as<float>(), ptr, col, val, in->as<float>(), L.r32.as<float>(), D, (float)L.coefs[0], s));
        else
            FDD_CALL(fdd_amg_smooth_update_matvec_f32(L.A_plan32, L.u32.as<float>(), ptr, col, val, in->as<float>(), L.r32.as<float>(), D, (float)L.coefs[0], s));
    }

    void vcycle_launches32()
    {
        const int nl = (int)levels.size();
        void *s = fdd::dev().stream;
        if (not f32_io) FDD_CALL(fdd_sub_copy_f32_f64(levels[0].f32.as<float>(), levels[0].f.as<double>(), levels[0].n, s));
        // the pre-smoother writes u = 0 + D w without reading u (smooth32): no zeroing of u
        if (nl == 1) FDD_CALL(fdd_amg_vector_set_to_value_f32(levels[0].u32.as<float>(), 0.0f, levels[0].n, s));
        for (int iter = 0; iter < num_vcycles; iter++)
        {
            for (int l = 0; l < nl - 1; l++)
            {
                Level &L = levels[l];
                smooth32(l, l > 0 or iter == 0);
                matvec32(L.A_plan32, L.A, L.A_val32, L.v32, &L.f32, L.u32, -1.0f, 1.0f);
                if (L.T.active and matrix_free_transfer)
                {
                    {
                        fdd::ProfileScope prof("lattice_restrict_kernel<f32>", L.T.restrict_bytes(L.n, 4));
                        FDD_CALL(fdd_lattice_restrict_f32(L.T.partial32.as<float>(), L.v32.as<float>(), L.T.owner_dof.as<int>(), L.T.n, L.T.m, L.T.lo.data(), L.T.hi.data(), L.T.wl.data(), L.T.num_elements, s));
                    }
                    matvec32(L.T.gather_plan32, L.T.gather, L.T.gather_val32, levels[l + 1].f32, nullptr, L.T.partial32, 1.0f, 0.0f);
                }
                else
                    matvec32(L.R_plan32, L.R, L.R_val32, levels[l + 1].f32, nullptr, L.v32, 1.0f, 0.0f);
            }
            Level &C = levels[nl - 1];
            matvec32(coarse_plan32, coarse_inverse, coarse_inverse_val32, C.u32, nullptr, C.f32, 1.0f, 0.0f);
            for (int l = nl - 1; l > 0; l--)
            {
                Level &F = levels[l - 1];
                if (F.T.active and matrix_free_transfer)
                {
                    fdd::ProfileScope prof("lattice_prolong_kernel<f32>", F.T.prolong_bytes(F.n, levels[l].n, 4));
                    FDD_CALL(fdd_lattice_prolong_f32(F.u32.as<float>(), levels[l].u32.as<float>(), F.T.owner_dof.as<int>(), F.T.coarse_dof.as<int>(), F.T.n, F.T.m, F.T.lo.data(), F.T.hi.data(), F.T.wl.data(), F.T.num_elements, s));
                }
                else
                    matvec32(F.P_plan32, F.P, F.P_val32, F.u32, &F.u32, levels[l].u32, 1.0f, 1.0f);
                smooth32(l - 1, false);
            }
        }
        if (not f32_io) FDD_CALL(fdd_sub_copy_f64_f32(levels[0].u.as<double>(), levels[0].u32.as<float>(), levels[0].n, s));
    }

  public:
    int cheby_order = 2; // subdomain.hpp:237
    int num_vcycles = 1; // subdomain.hpp:236
    bool use_graph = true; // AMG/config.hpp:6 USE_CUDA_GRAPH
    bool fused_smoother = true; // element-wise smoother kernels as SpMV epilogues (false: the reference's launch sequence)
    int precision = 64;         // AMG/config.hpp:4 `Float`: 64 = double, 32 = float (set_precision)
    bool f32_io = false;        // precision 32 with a caller that works in float itself (the single-precision inner solve): the right-hand
                                // side is given in levels[0].f32, the correction is read from levels[0].u32, no casts at the two ends

    void set_f32_io(bool on)
    {
        if (on != f32_io) destroy_graphs();
        f32_io = on;
    }
    fdd::memory &rhs32()
    {
        prepare32();
        return levels[0].f32;
    }
    fdd::memory &solution32() { return levels[0].u32; }

    // the interpolator of a geometric level applied matrix-free where the hierarchy's builder handed its maps over
    // (set_lattice_transfer); false: the CSR interpolator and its transpose, as on every other level.  The same operator
    // with its sums in another order: equal to rounding
    bool matrix_free_transfer = true;
    void set_matrix_free_transfer(bool on)
    {
        if (on != matrix_free_transfer) destroy_graphs();
        matrix_free_transfer = on;
    }

    // level l's interpolator IS the lattice interpolation these maps describe (low_order::geometric_level)
    void set_lattice_transfer(size_t l, fdd::low_order::Transfer &&t)
    {
        static const int env = getenv("FDD_TUNE_AMG_MATRIX_FREE_TRANSFER") ? atoi(getenv("FDD_TUNE_AMG_MATRIX_FREE_TRANSFER")) : 1; // development override
        int supported = 0;
        if (not t.active() or l >= levels.size() or not env) return;
        FDD_CALL(fdd_lattice_supported(t.n, t.m, &supported));
        if (not supported) return;
        Level &L = levels[l];
        LatticeTransfer &T = L.T;
        const long long mc = (long long)t.m * t.m * t.m, kept = t.num_elements * mc;
        const int nc = L.P.num_cols;
        if (kept >= (1LL << 31) or (long long)t.coarse_dof.size() != kept) return;
        T.n = t.n;
        T.m = t.m;
        T.lo = std::move(t.lo);
        T.hi = std::move(t.hi);
        T.wl = std::move(t.wl);
        T.num_elements = t.num_elements;
        T.owner_dof = fdd::dev().malloc<int>(t.owner_dof.size());
        T.owner_dof.copyFrom(t.owner_dof.data(), t.owner_dof.size() * sizeof(int));
        T.coarse_dof = fdd::dev().malloc<int>(t.coarse_dof.size());
        T.coarse_dof.copyFrom(t.coarse_dof.data(), t.coarse_dof.size() * sizeof(int));
        T.partial = fdd::dev().malloc<double>((size_t)kept);
        // coarse dof -> the kept nodes it sits on, in ascending (element, node) order: the boolean gather of the restriction
        std::vector<int> ptr((size_t)nc + 1, 0), col, fill;
        for (long long q = 0; q < kept; q++)
            if (t.coarse_dof[(size_t)q] >= 0) ptr[(size_t)t.coarse_dof[(size_t)q] + 1]++;
        for (int c = 0; c < nc; c++) ptr[(size_t)c + 1] += ptr[(size_t)c];
        col.resize((size_t)ptr[(size_t)nc]);
        fill.assign(ptr.begin(), ptr.end() - 1);
        for (long long q = 0; q < kept; q++)
            if (t.coarse_dof[(size_t)q] >= 0) col[(size_t)fill[(size_t)t.coarse_dof[(size_t)q]]++] = (int)q;
        std::vector<double> ones(col.size(), 1.0);
        T.gather.assemble_from_csr(nc, (int)kept, ptr.data(), col.data(), ones.data());
        T.active = true;
        destroy_graphs();
    }

    // subdomain.hpp:236 (swept by run.py:154): a captured graph belongs to one cycle count
    void set_num_vcycles(int v)
    {
        if (v != num_vcycles) destroy_graphs();
        num_vcycles = v;
    }

    // 32 needs the fused sequence with a Chebyshev order of at least 2; a captured graph belongs to one precision
    bool set_precision(int bits)
    {
        if (bits != 64 and bits != 32) return false;
        if (bits == 32 and cheby_order < 2) return false;
        if (bits != precision) destroy_graphs();
        precision = bits;
        return true;
    }
    std::vector<Level> levels;

    bool ready() const { return finalized; }
    int fine_size() const { return levels.empty() ? 0 : levels[0].n; }

    // CSR arrays on the host; P_* null on the coarsest level
    void add_level(int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs_, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val)
    {
        Level &L = new_level(n);
        L.A.assemble_from_csr(n, n, A_ptr, A_col, A_val);
        if (P_ptr) L.P.assemble_from_csr(n, n_coarse, P_ptr, P_col, P_val);
        finish_level(L, D_val, coefs_, P_ptr != nullptr);
    }

    // the same with the host arrays of the level handed over instead of copied (Subdomain::amg_build's own hierarchy)
    void add_level_adopt(int n, std::vector<int> &&A_ptr, fdd::low_order::pod_vector<int> &&A_col, fdd::low_order::pod_vector<double> &&A_val, const double *D_val, const double *coefs_, int n_coarse, std::vector<int> &&P_ptr,
                         fdd::low_order::pod_vector<int> &&P_col, fdd::low_order::pod_vector<double> &&P_val)
    {
        Level &L = new_level(n);
        const bool has_P = not P_ptr.empty();
        L.A.adopt_csr(n, n, std::move(A_ptr), std::move(A_col), std::move(A_val));
        if (has_P) L.P.adopt_csr(n, n_coarse, std::move(P_ptr), std::move(P_col), std::move(P_val));
        finish_level(L, D_val, coefs_, has_P);
    }

  private:

    Level &new_level(int n)
    {
        if (levels.capacity() < 32) levels.reserve(32); // a growing vector would copy every level built so far, host mirrors included
        levels.emplace_back();
        levels.back().n = n;
        return levels.back();
    }
    void finish_level(Level &L, const double *D_val, const double *coefs_, bool has_P)
    {
        const int n = L.n;
        if (has_P) L.P.transpose(L.R); // R_fem[l] = P^T (subdomain.tpp:3526-3545)
        L.D_val = fdd::dev().malloc<double>(n);
        L.D_val.copyFrom(D_val, (size_t)n * sizeof(double));
        L.D_hst.assign(D_val, D_val + n);
        L.coefs.assign(coefs_, coefs_ + cheby_order);
        for (fdd::memory *m : {&L.f, &L.u, &L.r, &L.v, &L.w, &L.work}) *m = fdd::dev().malloc<double>(n);
    }

  public:

    void finalize()
    {
        // dense inverse of the coarsest operator by Gaussian elimination (no pivoting), once, on the host
        Level &C = levels.back();
        const int n = C.n;
        std::vector<double> M((size_t)n * n, 0.0), Inv((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++)
            for (int j = C.A.ptr_hst[i]; j < C.A.ptr_hst[i + 1]; j++) M[(size_t)i * n + C.A.col_hst[j]] += C.A.val_hst[j];
        for (int i = 0; i < n; i++) Inv[(size_t)i * n + i] = 1.0;
        for (int k = 0; k < n; k++)
            for (int i = k + 1; i < n; i++)
            {
                const double m = M[(size_t)i * n + k] / M[(size_t)k * n + k];
                if (m == 0.0) continue;
                for (int j = k + 1; j < n; j++) M[(size_t)i * n + j] -= m * M[(size_t)k * n + j];
                for (int j = 0; j < n; j++) Inv[(size_t)i * n + j] -= m * Inv[(size_t)k * n + j];
            }
        for (int i = n - 1; i >= 0; i--)
            for (int c = 0; c < n; c++)
            {
                double s = Inv[(size_t)i * n + c];
                for (int j = i + 1; j < n; j++) s -= M[(size_t)i * n + j] * Inv[(size_t)j * n + c];
                Inv[(size_t)i * n + c] = s / M[(size_t)i * n + i];
            }
        std::vector<int> ptr(n + 1), col((size_t)n * n);
        for (int i = 0; i <= n; i++) ptr[i] = i * n;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) col[(size_t)i * n + j] = j;
        coarse_inverse.assemble_from_csr(n, n, ptr.data(), col.data(), Inv.data());
        finalized = true;
    }

    // algorithmic bytes of the SpMVs of one application (the element-wise kernels between them are not counted)
    double spmv_bytes(bool f32 = false) const
    {
        // f32: 4-byte values and vector entries, the same 4-byte indices
        auto bytes = [f32](const CSR_Matrix<double> &M, bool weighted) { return f32 ? 8.0 * M.num_nnz + 8.0 * M.num_rows + 4.0 * M.num_cols + (weighted ? 4.0 * M.num_rows : 0.0) : M.algorithmic_bytes(weighted); };
        double b = 0.0;
        const int nl = (int)levels.size();
        for (int iter = 0; iter < num_vcycles; iter++)
            for (int l = 0; l < nl - 1; l++)
            {
                const int with_A = 2 * cheby_order + ((l > 0 or iter == 0) ? 0 : 1); // pre-smoothing (no residual SpMV from u = 0), residual, post-smoothing
                b += with_A * bytes(levels[l].A, true) + bytes(levels[l].R, false) + bytes(levels[l].P, true);
            }
        return b + num_vcycles * bytes(coarse_inverse, false);
    }

    // u_fem[0] = V-cycle applied to f_fem[0] from u = 0; both live in levels[0]
    void vcycle()
    {
        void *s = fdd::dev().stream;
        const bool f32 = (precision == 32);
        if (f32) prepare32();
        if (use_graph and not graph_failed)
        {
            void *&graph = graphs[(f32 and f32_io) ? levels[0].u32.ptr() : levels[0].u.ptr()];
            if (graph == nullptr)
            {
                // captured once on a private stream (the caller's may be the default stream, which
                // cannot be captured); the launches record there while fdd::dev().stream points at it
                void *cs = nullptr;
                if (fdd_stream_create(&cs) == 0)
                {
                    if (fdd_graph_begin_capture(cs) == 0)
                    {
                        const bool profiling = fdd::profiler().enabled; // no event records inside a capture
                        fdd::profiler().enabled = false;
                        fdd::dev().stream = cs;
                        if (f32)
                            vcycle_launches32();
                        else
                            vcycle_launches();
                        fdd::dev().stream = s;
                        fdd::profiler().enabled = profiling;
                        if (fdd_graph_end_capture(cs, &graph) != 0) graph = nullptr;
                    }
                    (void)fdd_stream_destroy(cs);
                }
                if (graph == nullptr) graph_failed = true;
            }
            if (graph != nullptr)
            {
                fdd::ProfileScope prof(f32 ? "amg_vcycle<hipGraph,f32>" : "amg_vcycle<hipGraph>", spmv_bytes(f32));
                FDD_CALL(fdd_graph_launch(graph, s));
                return;
            }
        }
        if (f32)
            vcycle_launches32();
        else
            vcycle_launches();
    }

    // The same cycle with its correction written into `out` (at least fine_size() entries of the cycle's output
    // type: double, or float under f32_io) instead of levels[0].u / u32: the caller's vector stands in for the level's
    // for the duration of the call.  The first call with a given vector captures a graph of its own.
    void vcycle_into(fdd::memory &out)
    {
        fdd::memory &slot = (precision == 32 and f32_io) ? levels[0].u32 : levels[0].u;
        if (precision == 32) prepare32();
        const fdd::memory saved = slot;
        slot = fdd::memory(out.ptr(), (size_t)levels[0].n, (precision == 32 and f32_io) ? sizeof(float) : sizeof(double), false);
        vcycle();
        slot = saved;
    }
};

} // namespace amg

#endif
