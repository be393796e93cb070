// fdd_f32.hip -- the element-wise and gather kernels of the preconditioner's dof-space solve in single precision
// (the reference compiles Subdomain with DType = PTYPE = Float, config.hpp:19-20, poisson.cpp:206: with
// Float = float the whole inner solve runs on float data; subdomain.okl:268-282 cast at its two ends).
// Storage is float; device-resident scalars (Gram-Schmidt coefficients, 1/norm scales, the Givens state) stay
// double, as do the accumulators of the reductions (fdd_reduce.hip).  The stiffness kernel is the float
// instantiation of the fused kernel (fdd_stiffness.hip), the V-cycle the f32 cycle of host/amg.hpp.
#include <cstdint>

#include "fdd_common.h"

namespace
{

constexpr int kBlock = 256;

inline int grid_for(long long n)
{
    long long g = (n + kBlock - 1) / kBlock;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// au = (*scale) * u
__global__ __launch_bounds__(kBlock) void scale_dev_f32_kernel(float *__restrict__ au, const double *__restrict__ scale, const float *__restrict__ u, long long n)
{
    const float s = (float)(*scale);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) au[i] = s * u[i];
}

// z = d .* ((*scale) * u), scale optional
__global__ __launch_bounds__(kBlock) void diag_scale_dev_f32_kernel(float *__restrict__ z, const float *__restrict__ d, const double *__restrict__ scale, const float *__restrict__ u, long long n)
{
    const float s = scale ? (float)(*scale) : 1.0f;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) z[i] = scale ? d[i] * (s * u[i]) : d[i] * u[i];
}

// uv = alpha * u + beta * v
__global__ __launch_bounds__(kBlock) void axpby_f32_kernel(float *uv, float alpha, const float *u, float beta, const float *v, long long n)
{
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) uv[i] = alpha * u[i] + beta * v[i];
}

// q (+)= sum_k c[k] * (s[k] *) v_k, k < min(m, *last + 1): the solution update of the inner GMRES from device coefficients
template <int M>
struct LincombArgs
{
    const float *v[M];
};
template <int M>
__global__ __launch_bounds__(kBlock) void lincomb_f32_kernel(float *__restrict__ q, int q_is_zero, const double *__restrict__ c, LincombArgs<M> a, const double *__restrict__ vs, const double *__restrict__ last, int m, long long n)
{
    int use = last ? (int)(*last) + 1 : m;
    use = use > m ? m : use;
    double ck[M];
#pragma unroll
    for (int k = 0; k < M; k++) ck[k] = (k < use) ? c[k] * (vs ? vs[k] : 1.0) : 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    {
        double acc = q_is_zero ? 0.0 : (double)q[i];
#pragma unroll
        for (int k = 0; k < M; k++)
            if (k < use) acc += ck[k] * (double)a.v[k][i];
        q[i] = (float)acc;
    }
}

// t[row] = sum_j u[col_j] over the row's (boolean) entries: the gather Qt of the dof-space solve, one lane per row
__global__ __launch_bounds__(kBlock) void gather_rows_f32_kernel(float *__restrict__ t, const int *__restrict__ ptr, const int *__restrict__ col, const float *__restrict__ u, int row_lo, int row_hi)
{
    const int row = row_lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= row_hi) return;
    const int j0 = ptr[row], j1 = ptr[row + 1];
    float s = 0.0f;
    for (int j = j0; j < j1; j += 4)
    {
        // four gathers in flight; the sum stays in column order
        const int c0 = col[j], c1 = col[(j + 1 < j1) ? j + 1 : j], c2 = col[(j + 2 < j1) ? j + 2 : j], c3 = col[(j + 3 < j1) ? j + 3 : j];
        const float x0 = u[c0], x1 = u[c1], x2 = u[c2], x3 = u[c3];
        s += x0;
        s += (j + 1 < j1) ? x1 : 0.0f;
        s += (j + 2 < j1) ? x2 : 0.0f;
        s += (j + 3 < j1) ? x3 : 0.0f;
    }
    t[row] = s;
}

// out[i] = in[index[i]] (0 where index[i] < 0), float and float -> double forms
template <typename TO>
__global__ __launch_bounds__(kBlock) void gather_indexed_f32_kernel(TO *__restrict__ out, const float *__restrict__ in, const int *__restrict__ index, long long n)
{
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    {
        const int k = index[i];
        out[i] = (k < 0) ? TO(0) : (TO)in[k];
    }
}

} // namespace

extern "C" {

int fdd_vector_scaling_dev_f32(float *au, const double *scale_dev, const float *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(au != nullptr && u != nullptr && scale_dev != nullptr);
    hipLaunchKernelGGL(scale_dev_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), au, scale_dev, u, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_vector_diagonal_scaling_dev_f32(float *z, const float *d, const double *scale_dev, const float *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(z != nullptr && u != nullptr && d != nullptr);
    hipLaunchKernelGGL(diag_scale_dev_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), z, d, scale_dev, u, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_vector_vector_addition_f32(float *uv, float alpha, const float *u, float beta, const float *v, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(uv != nullptr && u != nullptr && v != nullptr);
    hipLaunchKernelGGL(axpby_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), uv, alpha, u, beta, v, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_multi_lincomb_limited_dev_f32(float *q, int q_is_zero, const double *coeffs_dev, const float *const *v, const double *v_scale_dev, const double *last_dev, int m, int n, void *stream)
{
    FDD_REQUIRE(n >= 0 && m >= 1 && m <= 8);
    if (n == 0) return 0;
    FDD_REQUIRE(q != nullptr && coeffs_dev != nullptr && v != nullptr);
    LincombArgs<8> a;
    for (int k = 0; k < 8; k++) a.v[k] = v[k < m ? k : 0];
    hipLaunchKernelGGL(lincomb_f32_kernel<8>, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), q, q_is_zero, coeffs_dev, a, v_scale_dev, last_dev, m, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_gather_rows_f32(float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi, void *stream)
{
    FDD_REQUIRE(row_lo >= 0 && row_hi >= row_lo);
    if (row_hi == row_lo) return 0;
    FDD_REQUIRE(t != nullptr && ptr != nullptr && col != nullptr && u != nullptr);
    const int rows = row_hi - row_lo;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((rows + kBlock - 1) / kBlock), dim3(kBlock), 0, fdd_stream(stream), t, ptr, col, u, row_lo, row_hi);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_gather_indexed_f32(float *out, const float *in, const int *index, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(out != nullptr && in != nullptr && index != nullptr);
    hipLaunchKernelGGL(gather_indexed_f32_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), out, in, index, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_gather_indexed_f32_f64(double *out, const float *in, const int *index, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(out != nullptr && in != nullptr && index != nullptr);
    hipLaunchKernelGGL(gather_indexed_f32_kernel<double>, dim3(grid_for(n)), dim3(kBlock), 0, fdd_stream(stream), out, in, index, (long long)n);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
