/*
 * csr_matrix.hpp -- CSR_Matrix<DType>: the reference's sparse-matrix host
 * class (csr_matrix.hpp:15-56, csr_matrix.tpp:28-341) with the same public
 * members and methods, on the gfx950 SpMV kernels.
 *
 * Differences that matter:
 *   - assemble() also builds the SpMV row-block plan (fdd_csr_plan_create)
 *     while it still holds the host `ptr`; the multiply methods use it;
 *   - duplicates are summed in insertion order (a stable sort; the
 *     reference's std::sort leaves the order of equal keys unspecified);
 *   - host copies of ptr/col/val are kept (setup code walks them), so
 *     transpose()/diagonal()/print() need no device round trip.
 */
#ifndef FDD_CSR_MATRIX_HPP
#define FDD_CSR_MATRIX_HPP

#include <algorithm>
#include <cmath>
#include <tuple>
#include <type_traits>
#include <vector>

#include "config.hpp"
#include "host_parallel.hpp"

template <typename DType>
class CSR_Matrix
{
    static_assert(std::is_same<DType, double>::value, "the gfx950 SpMV kernels are fp64");

  private:
    int is_initialized = false;
    DType sparse_tolerance = 1.0e-12;
    std::vector<std::tuple<int, int, DType>> entries;
    fdd_csr_plan *plan = nullptr;
    int plan_kind = 0;
    bool plan_pipelined = false; // a short-row plan served by the persistent pipelined kernel (profile labels name the kernel that runs)
    bool lazy_identity = false; // initialize_identity(): the arrays do not exist yet

    void initialization_check()
    {
        if (not is_initialized)
        {
            printf("ERROR: CSR matrix has not been initialize\n");
            exit(EXIT_FAILURE);
        }
    }

  public:
    int num_rows = 0;
    int num_cols = 0;
    int num_nnz = 0;
    bool unit_values = false; // every stored value is exactly 1.0 (boolean gather/scatter matrices)
    bool is_identity = false;
    int sell = 0; // the plan carries a sliced-ELL copy (short even rows): the kernels named *_kernel<...> below are then sell_kernel
    fdd::memory ptr;
    fdd::memory col;
    fdd::memory val;

    // host mirrors (valid after assemble)
    std::vector<int> ptr_hst;
    fdd::low_order::pod_vector<int> col_hst; // resize() leaves new entries uninitialised (host_parallel.hpp)
    fdd::low_order::pod_vector<DType> val_hst;

    CSR_Matrix() {}
    CSR_Matrix(int num_rows_, int num_cols_) { initialize(num_rows_, num_cols_); }
    ~CSR_Matrix() {}

    void initialize(int num_rows_, int num_cols_)
    {
        num_rows = num_rows_;
        num_cols = num_cols_;
        num_nnz = 0;
        sparse_tolerance = 1.0e-12; // csr_matrix.tpp:61-64
        is_initialized = true;
    }

    void reserve(size_t n) { entries.reserve(n); }

    // The n x n identity (what add_entry(i, i, 1.0) for every i and assemble() give), WITHOUT its arrays: the fused launch
    // sequences look at `is_identity` and never multiply by it, and at C3's size three such matrices are 5 GB of device
    // memory and seconds of setup.  The first call that needs the arrays builds them (materialize()).
    void initialize_identity(int n)
    {
        initialize(n, n);
        num_nnz = n;
        unit_values = true;
        is_identity = true;
        lazy_identity = n > 0;
    }
    void materialize()
    {
        if (not lazy_identity) return;
        lazy_identity = false;
        const int n = num_rows;
        std::vector<int> p((size_t)n + 1);
        fdd::low_order::pod_vector<int> c((size_t)n);
        fdd::low_order::pod_vector<DType> v((size_t)n);
        fdd::low_order::parallel_ranges(n, fdd::low_order::range_parts(n), [&](long long i0, long long i1, int) {
            for (long long i = i0; i < i1; i++)
            {
                p[i] = (int)i;
                c[i] = (int)i;
                v[i] = (DType)1.0;
            }
        });
        p[n] = n;
        adopt_csr(n, n, std::move(p), std::move(c), std::move(v));
    }

    // val + col per non-zero, ptr + y per row, x once
    double algorithmic_bytes(bool weighted) const { return 12.0 * num_nnz + 12.0 * num_rows + 8.0 * num_cols + (weighted ? 8.0 * num_rows : 0.0); }

    void add_entry(int row, int col_, DType val_)
    {
        if ((row < 0) or (row >= num_rows) or (col_ < 0) or (col_ >= num_cols))
        {
            printf("ERROR: Entry at (%d, %d) is outside the matrix of size (%d, %d)\n", row, col_, num_rows, num_cols);
            exit(EXIT_FAILURE);
        }

        if (std::abs(val_) > sparse_tolerance) entries.push_back(std::tuple<int, int, DType>(row, col_, val_));
    }

    void assemble()
    {
        if ((num_rows == 0) or (num_cols == 0) or (entries.size() == 0)) return;

        initialization_check();

        const auto before = [](const std::tuple<int, int, DType> &a, const std::tuple<int, int, DType> &b) {
            if (std::get<0>(a) != std::get<0>(b)) return std::get<0>(a) < std::get<0>(b);
            return std::get<1>(a) < std::get<1>(b);
        };
        // entries added row by row (the scatter matrices Q: one per point) are in order already
        if (not std::is_sorted(entries.begin(), entries.end(), before)) std::stable_sort(entries.begin(), entries.end(), before);

        ptr_hst.assign(num_rows + 1, 0);
        col_hst.clear();
        val_hst.clear();
        col_hst.reserve(entries.size());
        val_hst.reserve(entries.size());

        int last_row = -1, last_col = -1;
        for (auto &entry : entries)
        {
            const int r = std::get<0>(entry), c = std::get<1>(entry);
            if (r != last_row or c != last_col)
            {
                ptr_hst[r + 1]++;
                col_hst.push_back(c);
                val_hst.push_back(std::get<2>(entry));
                last_row = r;
                last_col = c;
            }
            else
            {
                val_hst.back() += std::get<2>(entry);
            }
        }

        for (int i = 1; i <= num_rows; i++) ptr_hst[i] += ptr_hst[i - 1];

        entries.clear();
        entries.shrink_to_fit();
        upload();
    }

    // take finished CSR arrays as they are (AMG::CSR_Matrix::initialize copies HYPRE's i/j/data, AMG/csr_matrix.cpp:24-66)
    void assemble_from_csr(int num_rows_, int num_cols_, const int *ptr_, const int *col_, const DType *val_)
    {
        initialize(num_rows_, num_cols_);
        if ((num_rows == 0) or (num_cols == 0) or (ptr_[num_rows] == 0)) return;
        ptr_hst.assign(ptr_, ptr_ + num_rows + 1);
        col_hst.assign(col_, col_ + ptr_[num_rows]);
        val_hst.assign(val_, val_ + ptr_[num_rows]);
        upload();
    }

    // the same, taking the caller's arrays over instead of copying them (the setup's level matrices: a gigabyte at C2)
    void adopt_csr(int num_rows_, int num_cols_, std::vector<int> &&ptr_, fdd::low_order::pod_vector<int> &&col_, fdd::low_order::pod_vector<DType> &&val_)
    {
        initialize(num_rows_, num_cols_);
        if ((num_rows == 0) or (num_cols == 0) or (ptr_[num_rows] == 0)) return;
        ptr_hst = std::move(ptr_);
        col_hst = std::move(col_);
        val_hst = std::move(val_);
        upload();
    }

    // y = alpha*A*x + beta*y (AMG::CSR_Matrix::matvec, AMG/csr_matrix.cpp:129-131); beta == 0 never reads y
    void matvec(fdd::memory &y, fdd::memory &x, double alpha, double beta)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiAxpby>" : plan_kind == 0 ? "csr_row_kernel<EpiAxpby>" : (plan_pipelined ? "csr_short_pipelined_kernel<EpiAxpby>" : "csr_block_kernel<EpiAxpby>"), algorithmic_bytes(beta != 0.0));
        FDD_CALL(fdd_csr_plan_matvec(plan, y.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), x.as<double>(), alpha, beta, fdd::dev().stream));
    }

    // y = alpha*A*x + beta*y_in: the copy "y = y_in" that precedes the SpMV in the reference (subdomain.tpp:34-36) folded in
    void matvec_to(fdd::memory &y, fdd::memory &y_in, fdd::memory &x, double alpha, double beta)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiAxpby>" : plan_kind == 0 ? "csr_row_kernel<EpiAxpby>" : (plan_pipelined ? "csr_short_pipelined_kernel<EpiAxpby>" : "csr_block_kernel<EpiAxpby>"), algorithmic_bytes(beta != 0.0));
        FDD_CALL(fdd_csr_plan_matvec_to(plan, y.as<double>(), y_in.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), x.as<double>(), alpha, beta, fdd::dev().stream));
    }

    // --- the Chebyshev smoother's element-wise kernels as epilogues of this SpMV (subdomain.tpp:19-83) ---
    // Sr = D*(f - A u), work = D*(coef*Sr)
    void smooth_residual(fdd::memory &work, fdd::memory &Sr, fdd::memory &u, fdd::memory &f, fdd::memory &D, double coef)
    {
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiSmoothResidual>" : plan_kind == 0 ? "csr_row_kernel<EpiSmoothResidual>" : "csr_block_kernel<EpiSmoothResidual>", algorithmic_bytes(true) + 16.0 * num_rows);
        FDD_CALL(fdd_amg_smooth_residual_matvec(plan, work.as<double>(), Sr.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), u.as<double>(), f.as<double>(), D.as<double>(), coef, fdd::dev().stream));
    }

    // work_out = D*(coef*Sr + D*(A work_in))
    void smooth_polynomial(fdd::memory &work_out, fdd::memory &work_in, fdd::memory &Sr, fdd::memory &D, double coef)
    {
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiSmoothPoly>" : plan_kind == 0 ? "csr_row_kernel<EpiSmoothPoly>" : "csr_block_kernel<EpiSmoothPoly>", algorithmic_bytes(true) + 8.0 * num_rows);
        FDD_CALL(fdd_amg_smooth_polynomial_matvec(plan, work_out.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), work_in.as<double>(), Sr.as<double>(), D.as<double>(), coef, fdd::dev().stream));
    }

    // u += D*(coef*Sr + D*(A work_in))
    void smooth_update(fdd::memory &u, fdd::memory &work_in, fdd::memory &Sr, fdd::memory &D, double coef)
    {
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiSmoothUpdate>" : plan_kind == 0 ? "csr_row_kernel<EpiSmoothUpdate>" : "csr_block_kernel<EpiSmoothUpdate>", algorithmic_bytes(true) + 16.0 * num_rows);
        FDD_CALL(fdd_amg_smooth_update_matvec(plan, u.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), work_in.as<double>(), Sr.as<double>(), D.as<double>(), coef, fdd::dev().stream));
    }

    // the same from u = 0, which is not read (pre-smoothing)
    void smooth_update_from_zero(fdd::memory &u, fdd::memory &work_in, fdd::memory &Sr, fdd::memory &D, double coef)
    {
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        fdd::ProfileScope prof(sell ? "sell_kernel<EpiSmoothUpdate>" : plan_kind == 0 ? "csr_row_kernel<EpiSmoothUpdate>" : "csr_block_kernel<EpiSmoothUpdate>", algorithmic_bytes(true) + 8.0 * num_rows);
        FDD_CALL(fdd_amg_smooth_update_matvec_from_zero(plan, u.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), work_in.as<double>(), Sr.as<double>(), D.as<double>(), coef, fdd::dev().stream));
    }

  private:
    // host mirrors -> HBM + the SpMV plan
    void upload()
    {
        num_nnz = (int)col_hst.size();

        unit_values = true;
        for (const DType v : val_hst)
            if (v != (DType)1.0)
            {
                unit_values = false;
                break;
            }
        is_identity = unit_values and (num_rows == num_cols) and (num_nnz == num_rows);
        for (int i = 0; is_identity and i < num_rows; i++)
            if (col_hst[i] != i) is_identity = false;

        ptr = fdd::dev().malloc<int>(num_rows + 1);
        col = fdd::dev().malloc<int>(num_nnz);
        val = fdd::dev().malloc<DType>(num_nnz);

        ptr.copyFrom(ptr_hst.data(), (num_rows + 1) * sizeof(int));
        col.copyFrom(col_hst.data(), num_nnz * sizeof(int));
        val.copyFrom(val_hst.data(), num_nnz * sizeof(DType));

        if (plan) FDD_CALL(fdd_csr_plan_destroy(plan));
        FDD_CALL(fdd_csr_plan_create(&plan, ptr_hst.data(), num_rows, num_cols, num_nnz));
        FDD_CALL(fdd_csr_plan_kind(plan, &plan_kind));
        {
            int pipelined = 0;
            FDD_CALL(fdd_csr_plan_pipelined(plan, &pipelined));
            plan_pipelined = pipelined != 0;
        }
        FDD_CALL(fdd_csr_plan_set_unit_values(plan, unit_values ? 1 : 0));
        // short, even rows (AMG levels, interpolators): a sliced-ELL copy the SpMV entries then run on
        sell = 0;
        if (not unit_values and num_nnz > num_rows) FDD_CALL(fdd_csr_plan_attach_sell(plan, ptr_hst.data(), ptr.as<int>(), col.as<int>(), val.ptr(), 1.3, &sell, fdd::dev().stream));
    }

  public:

    // drop the host mirrors of a large matrix once setup no longer needs them
    void release_host()
    {
        std::vector<int>().swap(ptr_hst);
        fdd::low_order::pod_vector<int>().swap(col_hst);
        fdd::low_order::pod_vector<DType>().swap(val_hst);
    }

    // ... and fetch them back from the device copies when a later setup step wants to walk the matrix again
    void download_host()
    {
        materialize();
        if (num_nnz == 0 or not ptr.ptr()) return;
        ptr_hst.resize(num_rows + 1);
        col_hst.resize(num_nnz);
        val_hst.resize(num_nnz);
        ptr.copyTo(ptr_hst.data(), (size_t)(num_rows + 1) * sizeof(int));
        col.copyTo(col_hst.data(), (size_t)num_nnz * sizeof(int));
        val.copyTo(val_hst.data(), (size_t)num_nnz * sizeof(DType));
    }

    void print(FILE *file_ptr = NULL, int offset = 0)
    {
        FILE *out = file_ptr ? file_ptr : fdd::globals().pstdout_file;
        if (!out) out = stdout;
        fprintf(out, "num_rows = %d, num_cols = %d, num_nnz = %d\n", num_rows, num_cols, num_nnz);
        if ((num_rows == 0) or (num_cols == 0) or (num_nnz == 0)) return;
        for (int i = 0; i < num_rows; i++)
            for (int j = ptr_hst[i]; j < ptr_hst[i + 1]; j++) fprintf(out, "(%d, %d): %.16g\n", i + offset, col_hst[j] + offset, val_hst[j]);
    }

    void transpose(CSR_Matrix &At)
    {
        materialize();
        At.initialize(num_cols, num_rows);
        if ((num_rows == 0) or (num_cols == 0)) return;
        // What add_entry + assemble (csr_matrix.tpp:288-300) produces, by a counting transpose instead of a sort of
        // tuples: this matrix's rows are in ascending order with sorted, duplicate-free columns, so the transposed
        // rows come out sorted and duplicate-free; entries assemble would drop (|v| <= tolerance) are dropped here.
        // Ranges of the transposed rows on the host threads: a thread scans this matrix in row order and places the
        // entries of ITS columns (the serial loop's result).
        std::vector<int> tp(num_cols + 1, 0);
        const DType tol = At.sparse_tolerance;
        const int parts = fdd::low_order::range_parts(num_cols);
        fdd::low_order::parallel_ranges(num_cols, parts, [&](long long c0, long long c1, int) {
            const int nnz = ptr_hst[num_rows];
            for (int j = 0; j < nnz; j++)
            {
                const int c = col_hst[j];
                if (c >= c0 and c < c1 and std::abs(val_hst[j]) > tol) tp[c + 1]++;
            }
        });
        for (int c = 0; c < num_cols; c++) tp[c + 1] += tp[c];
        fdd::low_order::pod_vector<int> tc(tp[num_cols]);
        fdd::low_order::pod_vector<DType> tv(tp[num_cols]);
        fdd::low_order::parallel_ranges(num_cols, parts, [&](long long c0, long long c1, int) {
            std::vector<int> next(tp.begin() + c0, tp.begin() + c1);
            for (int i = 0; i < num_rows; i++)
                for (int j = ptr_hst[i]; j < ptr_hst[i + 1]; j++)
                {
                    const int c = col_hst[j];
                    if (c < c0 or c >= c1 or not(std::abs(val_hst[j]) > tol)) continue;
                    const int k = next[c - c0]++;
                    tc[k] = i;
                    tv[k] = val_hst[j];
                }
        });
        if (tp[num_cols] == 0) return;
        At.adopt_csr(num_cols, num_rows, std::move(tp), std::move(tc), std::move(tv));
    }

    void diagonal(fdd::memory D)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        std::vector<DType> work(num_rows, 0.0);
        for (int i = 0; i < num_rows; i++)
            for (int j = ptr_hst[i]; j < ptr_hst[i + 1]; j++)
                if (i == col_hst[j])
                {
                    work[i] = val_hst[j];
                    break;
                }
        D.copyFrom(work.data(), num_rows * sizeof(DType));
    }

    // --- boolean gather matrices only (unit_values): fused forms on the SpMV plan ---
    // out = Q (w .* (Qt u)) .* m with Q = (this)^T: one gather-scatter pass (fdd_hip.h);
    // mode 0 gather + scatter, 1 gather only (t out), 2 scatter only (t in); rows [row_lo, row_hi)
    void gather_scatter(double *out, double *t, const double *u, const double *node_weight, const double *point_mask, int row_lo, int row_hi, int mode)
    {
        if (row_hi <= row_lo or num_nnz == 0) return;
        const double rows = row_hi - row_lo, frac = rows / std::max(num_rows, 1);
        const double bytes = 4.0 * rows + frac * num_nnz * (4.0 + (mode != 2 ? 8.0 : 0.0) + (mode != 1 ? 8.0 : 0.0) + ((point_mask and mode != 1) ? 8.0 : 0.0)) +
                             ((node_weight and mode != 2) ? 8.0 * rows : 0.0) + ((t or mode != 0) ? 8.0 * rows : 0.0);
        const char *key = (plan_kind == 0) ? (mode == 0 ? "dssum_kernel<fused>" : mode == 1 ? "dssum_kernel<gather>" : "dssum_kernel<scatter>")
                                           : (mode == 0 ? "dssum_block_kernel<fused>" : mode == 1 ? ((plan_pipelined and not node_weight) ? "csr_short_pipelined_kernel<gather>" : "dssum_block_kernel<gather>") : "dssum_block_kernel<scatter>");
        fdd::ProfileScope prof(key, bytes);
        FDD_CALL(fdd_csr_plan_dssum(plan, out, t, ptr.as<int>(), col.as<int>(), u, node_weight, point_mask, row_lo, row_hi, mode, fdd::dev().stream));
    }

    // t[row] = sum of u over the row's entries on float vectors (the gather of a boolean matrix in the single-precision preconditioner)
    void gather_f32(float *t, const float *u, int row_lo, int row_hi)
    {
        if (row_hi <= row_lo or num_nnz == 0) return;
        fdd::ProfileScope prof((plan_pipelined ? "csr_short_pipelined_kernel<gather, f32>" : "gather_block_f32_kernel"), 8.0 * (row_hi - row_lo) + 8.0 * num_nnz * ((double)(row_hi - row_lo) / std::max(num_rows, 1)));
        FDD_CALL(fdd_csr_plan_gather_f32(plan, t, ptr.as<int>(), col.as<int>(), u, row_lo, row_hi, fdd::dev().stream));
    }

    // out_dev[0] = sum_rows s*s*w with s = (this u)[row]*w[row]
    void gather_weighted_norm2(double *out_dev, double *ws, const double *u, const double *node_weight)
    {
        fdd::ProfileScope prof(plan_kind == 0 ? "gather_norm2_kernel" : "gather_norm2_block_kernel", 4.0 * num_rows + 12.0 * num_nnz + 8.0 * num_rows);
        FDD_CALL(fdd_csr_plan_gather_weighted_norm2(plan, out_dev, ws, ptr.as<int>(), col.as<int>(), u, node_weight, fdd::dev().stream));
    }

    void multiply(fdd::memory &Au, fdd::memory &u)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        if (num_nnz == 0) // never assembled (csr_matrix.tpp:96 leaves no device arrays): A = 0
        {
            FDD_CALL(fdd_set_to_value(Au.as<double>(), 0.0, num_rows, 0, fdd::dev().stream));
            return;
        }
        fdd::ProfileScope prof(plan_kind == 0 ? "csr_row_kernel<EpiPlain>" : "csr_block_kernel<EpiPlain>", algorithmic_bytes(false));
        FDD_CALL(fdd_csr_plan_multiply(plan, Au.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), u.as<double>(), nullptr, fdd::dev().stream));
    }

    void multiply_range(fdd::memory &Au, fdd::memory &u, int row_start, int row_end)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        if (row_end < row_start)
        {
            printf("Row end (i_e = %d) has to be greater or equal to row start (i_s = %d)\n", row_end, row_start);
            exit(EXIT_FAILURE);
        }
        if (num_nnz == 0) return;
        FDD_CALL(fdd_csr_multiply_range(Au.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), u.as<double>(), row_start, row_end, fdd::dev().stream));
    }

    void multiply_weight(fdd::memory &Au, fdd::memory &u, fdd::memory &weight)
    {
        materialize();
        if ((num_rows == 0) or (num_cols == 0)) return;
        initialization_check();
        if (num_nnz == 0)
        {
            FDD_CALL(fdd_set_to_value(Au.as<double>(), 0.0, num_rows, 0, fdd::dev().stream));
            return;
        }
        fdd::ProfileScope prof(plan_kind == 0 ? "csr_row_kernel<EpiWeight>" : "csr_block_kernel<EpiWeight>", algorithmic_bytes(true));
        FDD_CALL(fdd_csr_plan_multiply(plan, Au.as<double>(), ptr.as<int>(), col.as<int>(), val.as<double>(), u.as<double>(), weight.as<double>(), fdd::dev().stream));
    }
};

#endif
