// Fused direct-stiffness summation (gather-scatter) for gfx950.
//
// The reference's direct_stiffness_summation (domain.tpp:582-600,
// subdomain.tpp:3969-3985) is two boolean CSR SpMVs with a vector in between:
//     t = (Qt u) .* node_weight ;  [gs_add on the boundary prefix] ;
//     out = (Q t) .* point_mask
// i.e. (12 nnz + 12 rows + 8 cols) bytes twice, 66-74 B per GLL point.  Q is the
// transpose of Qt with a single 1.0 per row, so the scatter needs no second
// matrix: one lane per assembled node sums its points (gather) and writes the
// sum back to the same points (scatter).  4 B/node of ptr, 4 B/point of col,
// 8 B/point in, 8 B/point out (+ 8 B/point mask, 8 B/node weight): ~36 B/point.
//
// Arithmetic is the reference's, operation for operation, so the result is
// bit-identical to the two SpMVs: the node sum starts from 0.0 and adds
// 1.0*u[col] in column order (csr_matrix.okl:9-14), is multiplied by
// weight[node] (:44), and the scatter is (0.0 + 1.0*t) * mask (:9-17, :44).
// Requires Qt to be boolean (all values 1.0), which the host class checks.
//
// Multi-rank: the first `num_bdary` nodes need gslib's gs_add across ranks
// between gather and scatter (domain.tpp:590-594).  `gather` fills t for a
// node range without scattering; `scatter` writes a node range from t; the
// fused kernel does both for the rank-interior nodes.
#include "fdd_common.h"

namespace
{
constexpr int kBlock = 256;

// MODE 0: gather + scatter, 1: gather only (t out), 2: scatter only (t in)
template <int MODE, bool WEIGHT, bool MASK>
__global__ __launch_bounds__(kBlock) void dssum_kernel(double *out, double *__restrict__ t, const int *__restrict__ Qt_ptr, const int *__restrict__ Qt_col, const double *u, const double *__restrict__ node_weight, const double *__restrict__ point_mask, int node_start, int node_end)
{
    // one node per lane, workgroups in XCD-chunked order (fdd_common.h)
    const int node = node_start + fdd_xcd_chunked_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (node < node_end)
    {
        const int j0 = Qt_ptr[node];
        const int j1 = Qt_ptr[node + 1];

        double s;
        if (MODE != 2)
        {
            s = fdd_row_sum<true>(Qt_col, nullptr, u, j0, j1); // all loads of the row in flight at once
            if (WEIGHT) s = s * node_weight[node];
            if (t) t[node] = s;
        }
        else
        {
            s = t[node];
        }

        if (MODE != 1)
        {
            const double v = 0.0 + 1.0 * s;
            for (int jb = j0; jb < j1; jb += FDD_ROW_CHUNK)
            {
                int p[FDD_ROW_CHUNK];
                double mk[FDD_ROW_CHUNK];
#pragma unroll
                for (int k = 0; k < FDD_ROW_CHUNK; k++) p[k] = (jb + k < j1) ? Qt_col[jb + k] : 0;
                if (MASK)
                {
#pragma unroll
                    for (int k = 0; k < FDD_ROW_CHUNK; k++) mk[k] = (jb + k < j1) ? point_mask[p[k]] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < FDD_ROW_CHUNK; k++)
                    if (jb + k < j1) out[p[k]] = MASK ? v * mk[k] : v;
            }
        }
    }
}

template <int MODE>
int launch(double *out, double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *w, const double *m, int n0, int n1, void *stream)
{
    if (n1 <= n0) return 0;
    const int grid = (int)(((long long)n1 - n0 + kBlock - 1) / kBlock);
    hipStream_t s = fdd_stream(stream);
#define FDD_DSSUM_LAUNCH(W, M) hipLaunchKernelGGL((dssum_kernel<MODE, W, M>), dim3(grid), dim3(kBlock), 0, s, out, t, Qt_ptr, Qt_col, u, w, m, n0, n1)
    if (w && m)
        FDD_DSSUM_LAUNCH(true, true);
    else if (w)
        FDD_DSSUM_LAUNCH(true, false);
    else if (m)
        FDD_DSSUM_LAUNCH(false, true);
    else
        FDD_DSSUM_LAUNCH(false, false);
#undef FDD_DSSUM_LAUNCH
    FDD_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void fill_indexed_kernel(double *__restrict__ out, const int *__restrict__ idx, double value, int n)
{
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[idx[i]] = value;
}
} // namespace

extern "C" {

int fdd_dssum_fused(double *QQtu, double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, const double *point_mask, int node_start, int node_end, void *stream)
{
    FDD_REQUIRE(node_start >= 0 && node_end >= node_start);
    if (node_end == node_start) return 0;
    FDD_REQUIRE(QQtu != nullptr && Qt_ptr != nullptr && Qt_col != nullptr && u != nullptr);
    return launch<0>(QQtu, t, Qt_ptr, Qt_col, u, node_weight, point_mask, node_start, node_end, stream);
}

int fdd_dssum_gather(double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, int node_start, int node_end, void *stream)
{
    FDD_REQUIRE(node_start >= 0 && node_end >= node_start);
    if (node_end == node_start) return 0;
    FDD_REQUIRE(t != nullptr && Qt_ptr != nullptr && Qt_col != nullptr && u != nullptr);
    return launch<1>(nullptr, t, Qt_ptr, Qt_col, u, node_weight, nullptr, node_start, node_end, stream);
}

int fdd_dssum_scatter(double *QQtu, const double *t, const int *Qt_ptr, const int *Qt_col, const double *point_mask, int node_start, int node_end, void *stream)
{
    FDD_REQUIRE(node_start >= 0 && node_end >= node_start);
    if (node_end == node_start) return 0;
    FDD_REQUIRE(QQtu != nullptr && t != nullptr && Qt_ptr != nullptr && Qt_col != nullptr);
    return launch<2>(QQtu, const_cast<double *>(t), Qt_ptr, Qt_col, nullptr, nullptr, point_mask, node_start, node_end, stream);
}

int fdd_fill_indexed(double *out, const int *idx, double value, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(out != nullptr && idx != nullptr);
    hipLaunchKernelGGL(fill_indexed_kernel, dim3(fdd_stream_grid(n, kBlock)), dim3(kBlock), 0, fdd_stream(stream), out, idx, value, n);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
