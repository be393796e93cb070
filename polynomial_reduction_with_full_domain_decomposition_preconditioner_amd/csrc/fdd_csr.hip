// CSR SpMV y = A x (optionally * weight, or alpha*A*x + beta*y) for gfx950.
// Replaces csr_matrix.okl (multiply / multiply_range / multiply_weight) and
// the cusparseSpMV call of AMG/csr_matrix.cpp:129-131.
//
// Two kernels, both HBM-bound (12 B per non-zero + 12 B per row + x once):
//
//  * csr_row_kernel: one lane per row, the reference's own mapping.  With
//    <= ~4 non-zeros per row (the boolean gather/scatter matrices Q, Qt,
//    Q_int ...) consecutive lanes read consecutive val/col entries, so every
//    128-B line is fetched from HBM once and this is already the minimum
//    traffic; no staging needed.
//
//  * csr_block_kernel: "LDS row staging".  The host-side plan cuts the rows
//    into blocks of <= FDD_CSR_BLOCK_NNZ non-zeros.  A 256-lane workgroup
//    streams its block's val/col fully coalesced (8 independent loads per lane
//    in flight), multiplies by the gathered x (L2 hits for stencil-like
//    matrices) and parks the products in 16 KiB of LDS; then one lane per row
//    adds that row's products in column order -- the reference's summation
//    order, so the result is bit-identical to the thread-per-row kernel.
//    A row longer than a block is reduced by the whole workgroup (shuffle
//    tree; order differs).
#include "fdd_common.h"

#include <vector>

namespace
{

constexpr int kBlock = 256;
constexpr int kBlockNnz = FDD_CSR_BLOCK_NNZ;
constexpr int kBlockRowsMax = 2048;

struct EpiPlain
{
    __device__ double apply(double s, int row, const double *y_old) const { return s; }
};
struct EpiWeight
{
    const double *weight;
    __device__ double apply(double s, int row, const double *y_old) const { return s * weight[row]; }
};
struct EpiAxpby // AMG/csr_matrix.cpp:112-134
{
    double alpha, beta;
    __device__ double apply(double s, int row, const double *y_old) const { return alpha * s + beta * y_old[row]; }
};

template <typename Epi>
__global__ __launch_bounds__(kBlock) void csr_row_kernel(double *__restrict__ Au, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const double *__restrict__ A_val, const double *__restrict__ u, Epi epi, int row_start, int row_end)
{
    // one row per lane, workgroups in XCD-chunked order: the x gathers of the
    // boolean gather/scatter matrices then stay inside one L2 (fdd_common.h)
    const int i = row_start + fdd_xcd_chunked_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (i < row_end)
    {
        const int j0 = A_ptr[i];
        const int j1 = A_ptr[i + 1];
        const double Au_i = fdd_row_sum<false>(A_col, A_val, u, j0, j1); // column order, loads of the row in flight together
        Au[i] = epi.apply(Au_i, i, Au);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = FDD_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, FDD_WAVE);
    return v;
}

template <typename Epi>
__global__ __launch_bounds__(kBlock) void csr_block_kernel(double *__restrict__ Au, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const double *__restrict__ A_val, const double *__restrict__ u, Epi epi, const int *__restrict__ row_blocks)
{
    __shared__ double prod[kBlockNnz];
    __shared__ double wsum[kBlock / FDD_WAVE];

    const int r0 = row_blocks[blockIdx.x];
    const int r1 = row_blocks[blockIdx.x + 1];
    const int base = A_ptr[r0];
    const int nnz = A_ptr[r1] - base;

    if (nnz <= kBlockNnz)
    {
        // phase 1: coalesced stream of the block's non-zeros
#pragma unroll
        for (int it = 0; it < kBlockNnz / kBlock; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            if (k < nnz) prod[k] = A_val[base + k] * u[A_col[base + k]];
        }
        __syncthreads();

        // phase 2: one lane per row, products added in column order
        for (int row = r0 + threadIdx.x; row < r1; row += kBlock)
        {
            const int j0 = A_ptr[row] - base;
            const int j1 = A_ptr[row + 1] - base;
            double Au_i = 0.0;
            for (int j = j0; j < j1; j++) Au_i += prod[j];
            Au[row] = epi.apply(Au_i, row, Au);
        }
    }
    else
    {
        // a single long row (the plan guarantees r1 == r0 + 1)
        double s = 0.0;
        for (int k = threadIdx.x; k < nnz; k += kBlock) s += A_val[base + k] * u[A_col[base + k]];
        s = wave_sum(s);
        if ((threadIdx.x & (FDD_WAVE - 1)) == 0) wsum[threadIdx.x / FDD_WAVE] = s;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            double t = wsum[0];
#pragma unroll
            for (int w = 1; w < kBlock / FDD_WAVE; w++) t += wsum[w];
            Au[r0] = epi.apply(t, r0, Au);
        }
    }
}

template <typename Epi>
int launch_rows(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const Epi &epi, int row_start, int row_end, void *stream)
{
    if (row_end <= row_start) return 0;
    int grid = (int)(((long long)row_end - row_start + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(csr_row_kernel<Epi>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_ptr, A_col, A_val, u, epi, row_start, row_end);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // namespace

struct fdd_csr_plan
{
    int num_rows;
    int num_cols;
    int num_nnz;
    int kind; // 0: thread-per-row, 1: LDS-staged row blocks
    int num_blocks;
    int *row_blocks_dev; // num_blocks + 1
};

extern "C" {

int fdd_csr_multiply(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(Au != nullptr && A_ptr != nullptr && u != nullptr);
    return launch_rows(Au, A_ptr, A_col, A_val, u, EpiPlain{}, 0, n, stream);
}

int fdd_csr_multiply_range(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int row_start, int row_end, void *stream)
{
    // csr_matrix.tpp:322-326: row_end < row_start is an error in the host class
    FDD_REQUIRE(row_start >= 0 && row_end >= row_start);
    FDD_REQUIRE(Au != nullptr && A_ptr != nullptr && u != nullptr);
    return launch_rows(Au, A_ptr, A_col, A_val, u, EpiPlain{}, row_start, row_end + 1, stream);
}

int fdd_csr_multiply_weight(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(Au != nullptr && A_ptr != nullptr && u != nullptr && weight != nullptr);
    return launch_rows(Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, 0, n, stream);
}

int fdd_amg_matvec(double *y, const int *ptr, const int *col, const double *val, const double *x, double alpha, double beta, int num_rows, void *stream)
{
    FDD_REQUIRE(num_rows >= 0);
    if (num_rows == 0) return 0;
    FDD_REQUIRE(y != nullptr && ptr != nullptr && x != nullptr && y != x);
    return launch_rows(y, ptr, col, val, x, EpiAxpby{alpha, beta}, 0, num_rows, stream);
}

int fdd_csr_plan_create(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz)
{
    FDD_REQUIRE(plan != nullptr && num_rows >= 0 && num_cols >= 0 && num_nnz >= 0);
    FDD_REQUIRE(num_rows == 0 || A_ptr_host != nullptr);
    FDD_REQUIRE(num_rows == 0 || A_ptr_host[num_rows] - A_ptr_host[0] == num_nnz);

    fdd_csr_plan *p = new fdd_csr_plan();
    p->num_rows = num_rows;
    p->num_cols = num_cols;
    p->num_nnz = num_nnz;
    p->kind = 0;
    p->num_blocks = 0;
    p->row_blocks_dev = nullptr;

    // boolean gather/scatter matrices: thread-per-row is already minimal traffic
    if (num_rows == 0 || (double)num_nnz <= 4.0 * (double)num_rows)
    {
        *plan = p;
        return 0;
    }

    std::vector<int> blocks;
    blocks.push_back(0);
    int r = 0;
    while (r < num_rows)
    {
        const int base = A_ptr_host[r];
        int e = r;
        while (e < num_rows && (e - r) < kBlockRowsMax && A_ptr_host[e + 1] - base <= kBlockNnz) e++;
        if (e == r) e = r + 1; // one row longer than a block: workgroup-reduced
        blocks.push_back(e);
        r = e;
    }

    p->kind = 1;
    p->num_blocks = (int)blocks.size() - 1;

    hipError_t err = hipMalloc((void **)&p->row_blocks_dev, blocks.size() * sizeof(int));
    if (err == hipSuccess) err = hipMemcpy(p->row_blocks_dev, blocks.data(), blocks.size() * sizeof(int), hipMemcpyHostToDevice);
    if (err != hipSuccess)
    {
        fdd_set_error("fdd_csr_plan_create: %s", hipGetErrorString(err));
        if (p->row_blocks_dev) (void)hipFree(p->row_blocks_dev);
        delete p;
        return (int)err;
    }

    *plan = p;
    return 0;
}

int fdd_csr_plan_destroy(fdd_csr_plan *plan)
{
    if (plan == nullptr) return 0;
    if (plan->row_blocks_dev) (void)hipFree(plan->row_blocks_dev);
    delete plan;
    return 0;
}

int fdd_csr_plan_num_blocks(const fdd_csr_plan *plan, int *num_blocks)
{
    FDD_REQUIRE(plan != nullptr && num_blocks != nullptr);
    *num_blocks = plan->num_blocks;
    return 0;
}

int fdd_csr_plan_kind(const fdd_csr_plan *plan, int *kind)
{
    FDD_REQUIRE(plan != nullptr && kind != nullptr);
    *kind = plan->kind;
    return 0;
}

int fdd_csr_plan_multiply(const fdd_csr_plan *plan, double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0; // csr_matrix.tpp:304,334
    FDD_REQUIRE(Au != nullptr && A_ptr != nullptr && u != nullptr);

    if (plan->kind == 0)
    {
        if (weight) return launch_rows(Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, 0, plan->num_rows, stream);
        return launch_rows(Au, A_ptr, A_col, A_val, u, EpiPlain{}, 0, plan->num_rows, stream);
    }

    if (weight)
        hipLaunchKernelGGL(csr_block_kernel<EpiWeight>, dim3(plan->num_blocks), dim3(kBlock), 0, fdd_stream(stream), Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, plan->row_blocks_dev);
    else
        hipLaunchKernelGGL(csr_block_kernel<EpiPlain>, dim3(plan->num_blocks), dim3(kBlock), 0, fdd_stream(stream), Au, A_ptr, A_col, A_val, u, EpiPlain{}, plan->row_blocks_dev);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
