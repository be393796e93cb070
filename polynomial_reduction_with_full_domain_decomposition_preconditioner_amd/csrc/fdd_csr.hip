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
#include <type_traits>

#include "fdd_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace
{

constexpr int kBlock = 256;
// Non-zeros per row block: the kernels are compiled for two sizes and the plan picks one per matrix.  Boolean
// gather / scatter matrices (under 4 non-zeros per row) run best on 1024 (more workgroups in flight: Qt gather
// 89 -> 77 us at C2), wider rows on 2048 (27-point stencil 838 -> 803 us).  A block holds at most as many rows.
constexpr int kBlockNnzMax = FDD_CSR_BLOCK_NNZ;
constexpr int kBlockNnzSmall = FDD_CSR_BLOCK_NNZ / 2;

// Row epilogues.  apply() = operand() + finish(): the block kernel loads the operands of all its rows up front
// (unconditionally, with its other loads) and finishes once the row sums are known.
// kFreeOrder: the entry stands in for cusparseSpMV (AMG/csr_matrix.cpp:129-131), whose summation order is not
// defined: wide rows may be summed by several lanes.  The csr_matrix.okl replacements keep the column order.
struct EpiPlain
{
    static constexpr bool kFreeOrder = false;
    typedef double Opnd;
    __device__ double operand(int, const double *) const { return 0.0; }
    __device__ double finish(double s, double, int) const { return s; }
};
struct EpiWeight
{
    static constexpr bool kFreeOrder = false;
    typedef double Opnd;
    const double *weight;
    __device__ double operand(int row, const double *) const { return weight[row]; }
    __device__ double finish(double s, double w, int) const { return s * w; }
};
template <typename T>
struct EpiAxpbyT // AMG/csr_matrix.cpp:112-134
{
    static constexpr bool kFreeOrder = true;
    typedef T Opnd;
    T alpha, beta;
    const T *y_in; // optional: y = alpha*A*x + beta*y_in with y_in another vector (f - A u without copying f first)
    // beta == 0: y is output only (cusparseSpMV semantics), whatever it held is not read
    __device__ T operand(int row, const T *y_old) const { return (beta == T(0)) ? T(0) : (y_in ? y_in[row] : y_old[row]); }
    __device__ T finish(T s, T y, int) const { return (beta == T(0)) ? alpha * s : alpha * s + beta * y; }
};
typedef EpiAxpbyT<double> EpiAxpby;

// The Chebyshev smoother's element-wise kernels (subdomain.tpp:19-83, AMG/kernels.cu:25-94) as epilogues of the
// SpMV in front of them: the same products and sums in the same order (the statements of the unfused kernels are
// quoted), without the vectors in between going through HBM.
template <typename T>
struct Opnd2
{
    T a, b;
};
template <typename T>
struct Opnd3
{
    T a, b, c;
};
// work = f - A u (matvec -1, 1) | Sr = S*work; w = alpha*Sr (scaled_residual) | out = D*w (vector_multiplication)
template <typename T>
struct EpiSmoothResidualT
{
    static constexpr bool kFreeOrder = true;
    typedef Opnd2<T> Opnd;
    const T *f, *D;
    T *r; // Sr
    T coef;
    __device__ Opnd operand(int row, const T *) const { return Opnd{f[row], D[row]}; }
    __device__ T finish(T s, Opnd o, int row) const
    {
        const T work = T(-1) * s + T(1) * o.a;
        const T sr = o.b * work;
        r[row] = sr;
        const T w = coef * sr;
        return w * o.b;
    }
};
// v = A work (matvec 1, 0) | v *= D; w = alpha*r + v (polynomial_evaluation) | out = D*w (vector_multiplication)
template <typename T>
struct EpiSmoothPolyT
{
    static constexpr bool kFreeOrder = true;
    typedef Opnd2<T> Opnd;
    const T *r, *D;
    T coef;
    __device__ Opnd operand(int row, const T *) const { return Opnd{r[row], D[row]}; }
    __device__ T finish(T s, Opnd o, int) const
    {
        const T v = (T(1) * s) * o.b;
        const T w = coef * o.a + v;
        return w * o.b;
    }
};
// v = A work | v *= D; w = alpha*r + v (polynomial_evaluation) | u += D*w (update_field): the output vector is u
template <typename T>
struct EpiSmoothUpdateT
{
    static constexpr bool kFreeOrder = true;
    typedef Opnd3<T> Opnd;
    const T *r, *D;
    T coef;
    __device__ Opnd operand(int row, const T *u_old) const { return Opnd{r[row], D[row], u_old[row]}; }
    __device__ T finish(T s, Opnd o, int) const
    {
        const T v = (T(1) * s) * o.b;
        const T w = coef * o.a + v;
        return o.c + o.b * w;
    }
};
// the same from u = 0 (every pre-smoothing of a V-cycle): u = 0 + D*w without reading u -- the bits of the update above on
// a zeroed vector -- so that the level's u need not be set to zero first
template <typename T>
struct EpiSmoothUpdateZeroT
{
    static constexpr bool kFreeOrder = true;
    typedef Opnd2<T> Opnd;
    const T *r, *D;
    T coef;
    __device__ Opnd operand(int row, const T *) const { return Opnd{r[row], D[row]}; }
    __device__ T finish(T s, Opnd o, int) const
    {
        const T v = (T(1) * s) * o.b;
        const T w = coef * o.a + v;
        return T(0) + o.b * w;
    }
};
typedef EpiSmoothResidualT<double> EpiSmoothResidual;
typedef EpiSmoothPolyT<double> EpiSmoothPoly;
typedef EpiSmoothUpdateT<double> EpiSmoothUpdate;

// Lane-per-row SpMV.  Each lane owns NPT rows (strided by the workgroup size,
// so every access stays coalesced across lanes) and keeps the column loads,
// then the x gathers, of all of them in flight together (fdd_multi_row_sum);
// sums are in column order.  UNIT: every stored value is 1.0 (the plan knows),
// val is not read and 1.0*x is x.  Workgroups in XCD-chunked order.
template <typename Epi, bool UNIT, int NPT>
__global__ __launch_bounds__(kBlock) void csr_row_kernel(double *__restrict__ Au, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const double *__restrict__ A_val, const double *__restrict__ u, Epi epi, int row_start, int row_end)
{
    const int tile = fdd_xcd_chunked_block(blockIdx.x, gridDim.x) * (kBlock * NPT);
    int row[NPT], j0[NPT], j1[NPT];
    double s[NPT];
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
        row[r] = row_start + tile + r * kBlock + threadIdx.x;
        const bool on = row[r] < row_end;
        j0[r] = on ? A_ptr[row[r]] : 0;
        j1[r] = on ? A_ptr[row[r] + 1] : 0;
    }
    fdd_multi_row_sum<NPT, 4, UNIT>(A_col, A_val, u, j0, j1, s);
#pragma unroll
    for (int r = 0; r < NPT; r++)
        if (row[r] < row_end) Au[row[r]] = epi.finish(s[r], epi.operand(row[r], Au), row[r]);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = FDD_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, FDD_WAVE);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = FDD_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, FDD_WAVE);
    return v;
}

// T: the value type of the matrix and of both vectors (double everywhere except the Float = float V-cycle, AMG/config.hpp:4)
template <typename T, typename Epi, bool UNIT, int kBlockNnz>
__global__ __launch_bounds__(kBlock) void csr_block_kernel(T *__restrict__ Au, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const T *__restrict__ A_val, const T *__restrict__ u, Epi epi, const int *__restrict__ row_blocks, int xcd_chunked, int split_rows)
{
    __shared__ T prod[kBlockNnz];
    // Matrices with short rows (the small block size) have hundreds of rows per block: their row pointers and
    // epilogue operands are loaded up front with everything else and staged in LDS.  With wide rows a block has
    // few rows and the extra LDS would only cost occupancy (27-point stencil: 757 -> 905 us): those read them
    // in the row loop.
    constexpr bool kStageRows = (kBlockNnz == kBlockNnzSmall);
    __shared__ int sp[kStageRows ? kBlockNnz + 1 : 1];
    __shared__ T wsum[kBlock / FDD_WAVE];

    // Row blocks in plain dispatch order by default: the val/col streams of the 8 XCDs then advance through
    // adjacent memory.  XCD-chunked order (FDD_TUNE_CSR_XCD=1) does cut the 27-point stencil's L2 fetch traffic
    // from 4.58 to 4.00 GB per launch (x is no longer pulled through all eight L2s; algorithmic 3.80 GB) but
    // runs 3 % slower (785 vs 762 us): the re-fetches are Infinity Cache hits, the eight far-apart streams cost more.
    // XCD-windowed order (xcd_chunked >= 2 = consecutive row blocks per XCD inside a window of 8x as many; fdd_common.h)
    // keeps the eight streams adjacent AND puts neighbouring row blocks on one XCD.
    const int b = xcd_chunked == 1 ? fdd_xcd_chunked_block(blockIdx.x, gridDim.x) : fdd_xcd_windowed_block(blockIdx.x, gridDim.x, xcd_chunked);
    const int r0 = row_blocks[b];
    const int r1 = row_blocks[b + 1];
    const int base = A_ptr[r0];
    const int nnz = A_ptr[r1] - base;

    if (nnz <= kBlockNnz)
    {
        // phase 1: coalesced stream of the block's non-zeros: all column (and
        // value) loads first, then all x gathers, flat over the non-zeros --
        // no dependence on row lengths, every lane has 8 loads in flight
        constexpr int kIts = kBlockNnz / kBlock;
        constexpr int kRowIts = kBlockNnz / kBlock; // a block has at most as many rows as non-zeros (the plan)
        const int nrows = r1 - r0;
        int c[kIts], rp[kRowIts];
        T a[kIts];
        typename Epi::Opnd opnd[kRowIts];
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            // unconditional loads on a selected index (see fdd_multi_row_sum): all of a lane's loads in flight
            const int k = threadIdx.x + it * kBlock;
            // slots past the block's non-zeros re-read its first one; a block of empty rows only (nnz == 0: its `base`
            // may be one past the last stored entry) reads entry 0 of the matrix instead and uses none of it
            const int ks = (k < nnz) ? base + k : ((nnz > 0) ? base : 0);
            // short-row matrices: the matrix streams are read once per launch and x keeps the L2 (Qt at C2: 91 -> 79 us);
            // the 27-point stencil measured 2 % slower that way
            c[it] = kStageRows ? __builtin_nontemporal_load(A_col + ks) : A_col[ks];
            a[it] = UNIT ? T(1) : (kStageRows ? __builtin_nontemporal_load(A_val + ks) : A_val[ks]);
        }
        if (kStageRows)
        {
#pragma unroll
            for (int it = 0; it < kRowIts; it++)
            {
                // row pointers and epilogue operands of this lane's rows, with the same trick
                const int r = threadIdx.x + it * kBlock;
                const int rs = (r < nrows) ? r : 0;
                rp[it] = A_ptr[r0 + rs + 1];
                opnd[it] = epi.operand(r0 + rs, Au);
            }
        }
        // wide rows, free summation order: `lanes` (a power of two, as many as the block's rows leave room for) add one
        // row together -- with a lane per row a block of 2048 non-zeros in rows of 50 keeps 40 of its 256 lanes
        // busy with 50 dependent additions each
        int lanes = 1;
        if (Epi::kFreeOrder && !kStageRows && split_rows)
            while (lanes < 16 && nrows * lanes * 2 <= kBlock) lanes *= 2;
        const int my_row = threadIdx.x / lanes; // lanes == 1: the lane's first row
        // wide rows: the lane's first row (the only one, unless the block is full of short rows) has its
        // pointers and epilogue operand requested here, so that nothing is loaded from HBM after the barrier
        const int rs0 = (my_row < nrows) ? my_row : 0;
        const int first_j0 = kStageRows ? 0 : A_ptr[r0 + rs0] - base;
        const int first_j1 = kStageRows ? 0 : A_ptr[r0 + rs0 + 1] - base;
        const typename Epi::Opnd first_opnd = epi.operand(r0 + rs0, Au); // unused (and dropped) when the rows are staged
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            const T x = u[c[it]];
            if (k < nnz) prod[k] = a[it] * x;
        }
        if (kStageRows)
        {
            if (threadIdx.x == 0) sp[0] = 0;
#pragma unroll
            for (int it = 0; it < kRowIts; it++)
            {
                const int r = threadIdx.x + it * kBlock;
                if (r < nrows) sp[r + 1] = rp[it] - base;
            }
        }
        __syncthreads();

        // phase 2: one lane per row, products added in column order
        if (kStageRows)
        {
#pragma unroll
            for (int it = 0; it < kRowIts; it++)
            {
                const int r = threadIdx.x + it * kBlock;
                if (r < nrows)
                {
                    const int j0 = sp[r], j1 = sp[r + 1];
                    T Au_i = T(0);
                    for (int j = j0; j < j1; j++) Au_i += prod[j];
                    Au[r0 + r] = epi.finish(Au_i, opnd[it], r0 + r);
                }
            }
        }
        else if (lanes > 1)
        {
            // all rows of the block in one pass (nrows * lanes <= 256); lane s of a row adds its entries s, s + lanes, ...
            // (adjacent lanes read adjacent products), then the partial sums are folded by a shuffle tree
            const int s = threadIdx.x & (lanes - 1);
            const bool on = my_row < nrows;
            T acc = T(0);
            if (on)
            {
                for (int j = first_j0 + s; j < first_j1; j += 4 * lanes)
                {
                    const int j1 = j + lanes, j2 = j + 2 * lanes, j3 = j + 3 * lanes;
                    const T p0 = prod[j];
                    const T p1 = prod[(j1 < kBlockNnz) ? j1 : j];
                    const T p2 = prod[(j2 < kBlockNnz) ? j2 : j];
                    const T p3 = prod[(j3 < kBlockNnz) ? j3 : j];
                    acc += p0;
                    acc += (j1 < first_j1) ? p1 : T(0);
                    acc += (j2 < first_j1) ? p2 : T(0);
                    acc += (j3 < first_j1) ? p3 : T(0);
                }
            }
            for (int off = lanes >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, FDD_WAVE);
            if (on && s == 0) Au[r0 + my_row] = epi.finish(acc, first_opnd, r0 + my_row);
        }
        else
        {
            for (int row = r0 + threadIdx.x; row < r1; row += kBlock)
            {
                const bool first = (row == r0 + (int)threadIdx.x);
                const int j0 = first ? first_j0 : A_ptr[row] - base;
                const int j1 = first ? first_j1 : A_ptr[row + 1] - base;
                const typename Epi::Opnd y = first ? first_opnd : epi.operand(row, Au);
                // four LDS reads in flight per lane; the sum stays in column order (slots past the row add +0.0,
                // which leaves a sum that started from +0.0 unchanged bit for bit)
                T Au_i = T(0);
                for (int j = j0; j < j1; j += 4)
                {
                    const T p0 = prod[j];
                    const T p1 = prod[(j + 1 < kBlockNnz) ? j + 1 : j];
                    const T p2 = prod[(j + 2 < kBlockNnz) ? j + 2 : j];
                    const T p3 = prod[(j + 3 < kBlockNnz) ? j + 3 : j];
                    Au_i += p0;
                    Au_i += (j + 1 < j1) ? p1 : T(0);
                    Au_i += (j + 2 < j1) ? p2 : T(0);
                    Au_i += (j + 3 < j1) ? p3 : T(0);
                }
                Au[row] = epi.finish(Au_i, y, row);
            }
        }
    }
    else
    {
        // a single long row (the plan guarantees r1 == r0 + 1)
        T s = T(0);
        for (int k = threadIdx.x; k < nnz; k += kBlock) s += (UNIT ? T(1) : A_val[base + k]) * u[A_col[base + k]];
        s = wave_sum(s);
        if ((threadIdx.x & (FDD_WAVE - 1)) == 0) wsum[threadIdx.x / FDD_WAVE] = s;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            T t = wsum[0];
#pragma unroll
            for (int w = 1; w < kBlock / FDD_WAVE; w++) t += wsum[w];
            Au[r0] = epi.finish(t, epi.operand(r0, Au), r0);
        }
    }
}

// The gather half (MODE 1, no weight) on float vectors: the single-precision preconditioner's Qt (subdomain.okl's kernels
// instantiated with DType = float).  Same staging, the row sums in column order in IEEE single.
template <int kBlockNnz>
__global__ __launch_bounds__(kBlock) void gather_block_f32_kernel(float *__restrict__ t, const int *__restrict__ Qt_ptr, const int *__restrict__ Qt_col, const float *__restrict__ u, const int *__restrict__ row_blocks, int block_first, int row_lo, int row_hi, int xcd_window)
{
    __shared__ float x[kBlockNnz];
    __shared__ int sp[kBlockNnz + 1];
    constexpr int kIts = kBlockNnz / kBlock;
    // Row blocks follow the node order, which follows the element order: a block gathers the low-face points of the NEXT
    // element(s), whose 64-byte sectors hold seven more points that the next block gathers.  In XCD-windowed order that
    // next block runs on the same XCD (fdd_common.h).
    const int b = block_first + fdd_xcd_windowed_block(blockIdx.x, gridDim.x, xcd_window);
    const int r0 = row_blocks[b] > row_lo ? row_blocks[b] : row_lo;
    const int r1 = row_blocks[b + 1] < row_hi ? row_blocks[b + 1] : row_hi;
    if (r1 <= r0) return;
    const int nrows = r1 - r0;
    const int base = Qt_ptr[r0];
    const int nnz = Qt_ptr[r1] - base;
    int c[kIts], rp[kIts];
#pragma unroll
    for (int it = 0; it < kIts; it++)
    {
        const int k = threadIdx.x + it * kBlock;
        c[it] = __builtin_nontemporal_load(Qt_col + ((k < nnz) ? base + k : ((nnz > 0) ? base : 0)));
        rp[it] = Qt_ptr[r0 + ((k < nrows) ? k : 0) + 1];
    }
#pragma unroll
    for (int it = 0; it < kIts; it++)
    {
        const int k = threadIdx.x + it * kBlock;
        const float v = u[c[it]];
        if (k < nnz) x[k] = v;
    }
    if (threadIdx.x == 0) sp[0] = 0;
#pragma unroll
    for (int it = 0; it < kIts; it++)
    {
        const int r = threadIdx.x + it * kBlock;
        if (r < nrows) sp[r + 1] = rp[it] - base;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIts; it++)
    {
        const int r = threadIdx.x + it * kBlock;
        if (r < nrows)
        {
            float s = 0.0f;
            for (int j = sp[r]; j < sp[r + 1]; j++) s += x[j];
            t[r0 + r] = s;
        }
    }
}

template <typename Epi>
int launch_rows(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const Epi &epi, int row_start, int row_end, void *stream, bool unit_values = false)
{
    if (row_end <= row_start) return 0;
    const long long rows = (long long)row_end - row_start;
    if (unit_values)
    {
        constexpr int NPT = 4;
        const int grid = (int)((rows + kBlock * NPT - 1) / (kBlock * NPT));
        hipLaunchKernelGGL((csr_row_kernel<Epi, true, NPT>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_ptr, A_col, A_val, u, epi, row_start, row_end);
    }
    else
    {
        constexpr int NPT = 2;
        const int grid = (int)((rows + kBlock * NPT - 1) / (kBlock * NPT));
        hipLaunchKernelGGL((csr_row_kernel<Epi, false, NPT>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_ptr, A_col, A_val, u, epi, row_start, row_end);
    }
    FDD_LAUNCH_CHECK();
    return 0;
}

// Exactly one stored entry in every row (the scatter matrices Q: the plan checks its host row pointers): the row
// pointers are i and are not read, y[i] = val[i]*x[col[i]] is a two-deep load chain instead of three, and the index
// and value streams are non-temporal.  0.0 + a*x keeps the row sum's bits (a sum that starts from +0.0).
template <typename Epi, bool UNIT, int NPT>
__global__ __launch_bounds__(kBlock) void csr_one_per_row_kernel(double *__restrict__ Au, const int *__restrict__ A_col, const double *__restrict__ A_val, const double *__restrict__ u, Epi epi, int n)
{
    const int tile = fdd_xcd_chunked_block(blockIdx.x, gridDim.x) * (kBlock * NPT);
    int c[NPT];
    double a[NPT], x[NPT];
    typename Epi::Opnd o[NPT];
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
        const int row = tile + r * kBlock + threadIdx.x;
        const int rs = (row < n) ? row : 0; // unconditional loads on a selected index
        c[r] = __builtin_nontemporal_load(A_col + rs);
        a[r] = UNIT ? 1.0 : __builtin_nontemporal_load(A_val + rs);
        o[r] = epi.operand(rs, Au);
    }
#pragma unroll
    for (int r = 0; r < NPT; r++) x[r] = u[c[r]];
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
        const int row = tile + r * kBlock + threadIdx.x;
        if (row < n) Au[row] = epi.finish(0.0 + a[r] * x[r], o[r], row);
    }
}

// The same as a persistent, software-pipelined kernel (as gather_pipelined_kernel below): the index stream (and values,
// epilogue operands) of a workgroup's next tile are requested behind the gathers of the current one.
template <typename Epi, bool UNIT, int NPT>
__global__ __launch_bounds__(kBlock) void csr_one_per_row_pipelined_kernel(double *__restrict__ Au, const int *__restrict__ A_col, const double *__restrict__ A_val, const double *__restrict__ u, Epi epi, int n, int ntiles, int xcd_window)
{
    const int G = gridDim.x;
    int tile = fdd_xcd_windowed_block(blockIdx.x, G, xcd_window);
    if (tile >= ntiles) return;
    int c[NPT];
    double a[NPT];
    typename Epi::Opnd o[NPT];
    auto load_streams = [&](int tl, int (&cc)[NPT], double (&aa)[NPT], typename Epi::Opnd (&oo)[NPT]) {
#pragma unroll
        for (int r = 0; r < NPT; r++)
        {
            const int row = tl * (kBlock * NPT) + r * kBlock + threadIdx.x;
            const int rs = (row < n) ? row : 0; // unconditional loads on a selected index
            cc[r] = __builtin_nontemporal_load(A_col + rs);
            aa[r] = UNIT ? 1.0 : __builtin_nontemporal_load(A_val + rs);
            oo[r] = epi.operand(rs, Au);
        }
    };
    load_streams(tile, c, a, o);
    for (;;)
    {
        const int next = tile + G;
        const bool more = next < ntiles;
        double x[NPT];
#pragma unroll
        for (int r = 0; r < NPT; r++) x[r] = u[c[r]];
        int cn[NPT];
        double an[NPT];
        typename Epi::Opnd on[NPT];
        load_streams(more ? next : tile, cn, an, on);
#pragma unroll
        for (int r = 0; r < NPT; r++)
        {
            const int row = tile * (kBlock * NPT) + r * kBlock + threadIdx.x;
            if (row < n) Au[row] = epi.finish(0.0 + a[r] * x[r], o[r], row);
        }
        if (!more) break;
        tile = next;
#pragma unroll
        for (int r = 0; r < NPT; r++)
        {
            c[r] = cn[r];
            a[r] = an[r];
            o[r] = on[r];
        }
    }
}

template <typename Epi>
int launch_one_per_row(double *Au, const int *A_col, const double *A_val, const double *u, const Epi &epi, int n, void *stream, bool unit_values)
{
    // persistent pipelined form: FDD_TUNE_ONE_PER_ROW_PIPELINED = workgroups per CU (0: one tile per workgroup)
    static const int per_cu = fdd_env_int("FDD_TUNE_ONE_PER_ROW_PIPELINED", 0);
    if (per_cu > 0)
    {
        constexpr int NPT = 4;
        const int ntiles = (n + kBlock * NPT - 1) / (kBlock * NPT);
        int g = per_cu * FDD_CU_COUNT;
        if (g > ntiles) g = ntiles;
        if (unit_values)
            hipLaunchKernelGGL((csr_one_per_row_pipelined_kernel<Epi, true, NPT>), dim3(g), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n, ntiles, 32);
        else
            hipLaunchKernelGGL((csr_one_per_row_pipelined_kernel<Epi, false, NPT>), dim3(g), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n, ntiles, 32);
        FDD_LAUNCH_CHECK();
        return 0;
    }
    static const int npt = fdd_env_int("FDD_TUNE_CSR_ONE_NPT", 4);
    const int per = kBlock * ((npt == 8) ? 8 : 4);
    const int grid = (n + per - 1) / per;
    if (unit_values)
    {
        if (npt == 8)
            hipLaunchKernelGGL((csr_one_per_row_kernel<Epi, true, 8>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n);
        else
            hipLaunchKernelGGL((csr_one_per_row_kernel<Epi, true, 4>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n);
    }
    else
    {
        if (npt == 8)
            hipLaunchKernelGGL((csr_one_per_row_kernel<Epi, false, 8>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n);
        else
            hipLaunchKernelGGL((csr_one_per_row_kernel<Epi, false, 4>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, A_col, A_val, u, epi, n);
    }
    FDD_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// LDS-staged gather-scatter (dssum) on the row blocks of a boolean gather
// matrix Qt: the balanced form of fdd_dssum.hip's lane-per-node kernel.
//   phase 1 (gather):  flat over the block's entries, x[k] = u[col[k]] -> LDS
//   phase 2:           one lane per node: s = sum of its x in column order,
//                      s *= weight[node], t[node] = s; s is written back over
//                      the node's LDS entries
//   phase 3 (scatter): flat over the entries, out[col[k]] = (0.0+1.0*s)*mask
// Global accesses never depend on row lengths (8 independent loads per lane per
// phase), only the LDS phase sees the raggedness.  Rows [row_lo, row_hi) only:
// the boundary prefix of a multi-rank Domain is gathered / scattered apart.
// MODE 0: gather + scatter, 1: gather only, 2: scatter only (s read from t).
// ---------------------------------------------------------------------------
template <int MODE, bool WEIGHT, bool MASK, int kBlockNnz>
__global__ __launch_bounds__(kBlock) void dssum_block_kernel(double *out, double *__restrict__ t, const int *__restrict__ Qt_ptr, const int *__restrict__ Qt_col, const double *u, const double *__restrict__ node_weight, const double *__restrict__ point_mask, const int *__restrict__ row_blocks, int block_first, int row_lo, int row_hi, int xcd_window)
{
    __shared__ double x[kBlockNnz];
    constexpr int kBlockRowsMax = kBlockNnz;
    __shared__ int sp[kBlockRowsMax + 1]; // the block's row pointers, relative to its first non-zero
    constexpr int kIts = kBlockNnz / kBlock;
    constexpr int kRowIts = kBlockRowsMax / kBlock;

    const int b = block_first + fdd_xcd_windowed_block(blockIdx.x, gridDim.x, xcd_window); // as in gather_block_f32_kernel
    const int r0 = row_blocks[b] > row_lo ? row_blocks[b] : row_lo;
    const int r1 = row_blocks[b + 1] < row_hi ? row_blocks[b + 1] : row_hi;
    if (r1 <= r0) return;
    const int nrows = r1 - r0; // <= kBlockRowsMax (the plan)
    const int base = Qt_ptr[r0];
    const int nnz = Qt_ptr[r1] - base; // <= kBlockNnz: boolean gather rows are short (the plan checks)

    // Every global load of the block is issued up front and UNCONDITIONALLY (out-of-range slots re-read
    // entry 0): a load under a lane predicate compiles to a branch with its own s_waitcnt, and the loads
    // of a lane then complete one HBM latency after the other instead of together.
    int c[kIts], rp[kRowIts];
    double wn[kRowIts], tv[kRowIts];
#pragma unroll
    for (int it = 0; it < kIts; it++)
    {
        const int k = threadIdx.x + it * kBlock;
        c[it] = __builtin_nontemporal_load(Qt_col + ((k < nnz) ? base + k : ((nnz > 0) ? base : 0))); // the index stream is read once (an all-empty block reads entry 0 and uses none of it)
    }
#pragma unroll
    for (int it = 0; it < kRowIts; it++)
    {
        const int r = threadIdx.x + it * kBlock;
        const int rs = (r < nrows) ? r : 0;
        rp[it] = Qt_ptr[r0 + rs + 1];
        if (WEIGHT && MODE != 2) wn[it] = node_weight[r0 + rs];
        if (MODE == 2) tv[it] = t[r0 + rs];
    }
    if (MODE != 2)
    {
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            const double v = u[c[it]];
            if (k < nnz) x[k] = 1.0 * v;
        }
    }
    if (threadIdx.x == 0) sp[0] = 0;
#pragma unroll
    for (int it = 0; it < kRowIts; it++)
    {
        const int r = threadIdx.x + it * kBlock;
        if (r < nrows) sp[r + 1] = rp[it] - base;
    }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < kRowIts; it++)
    {
        const int r = threadIdx.x + it * kBlock;
        if (r < nrows)
        {
            const int j0 = sp[r], j1 = sp[r + 1];
            double s;
            if (MODE != 2)
            {
                s = 0.0;
                for (int j = j0; j < j1; j++) s += x[j];
                if (WEIGHT) s = s * wn[it];
                if (t) t[r0 + r] = s;
            }
            else
            {
                s = tv[it];
            }
            if (MODE != 1)
                for (int j = j0; j < j1; j++) x[j] = s;
        }
    }

    if (MODE != 1)
    {
        __syncthreads();
        double mk[kIts];
        if (MASK)
        {
#pragma unroll
            for (int it = 0; it < kIts; it++) mk[it] = point_mask[c[it]];
        }
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            if (k < nnz)
            {
                const double v = 0.0 + 1.0 * x[k];
                out[c[it]] = MASK ? v * mk[it] : v;
            }
        }
    }
}

// Short-row plans (blocks of kBlockNnzSmall non-zeros: the boolean gather `Qt` of every operator application of the solve,
// the hanging-point rows, the AMG's interpolators) as a PERSISTENT, software-pipelined kernel.  The one-block-per-workgroup
// form (csr_block_kernel / dssum_block_kernel above) is bound by neither bytes nor occupancy on such matrices (16 VGPRs,
// 8 waves per SIMD; taking 14 % of its L2 fetches away changed its time by 3 %): a row block is a chain of dependent round
// trips -- block bounds -> row pointers -> col (val) -> u[col] -> LDS -> barrier -> row sums -> store -- of which a
// workgroup has only one in flight at a time.  Here a workgroup walks its row blocks with three of them in flight: the
// 16-byte descriptor of block i + 2, the index / value streams, row pointers and epilogue operands of block i + 1 and the
// value gathers of block i are outstanding together, and the row sums of block i run under them.  Products and sums as in
// csr_block_kernel (column order, one lane per row), the same epilogues: same bits.
template <typename T, typename Epi, bool UNIT, int kBlockNnz>
__global__ __launch_bounds__(kBlock) void csr_short_pipelined_kernel(T *__restrict__ Au, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const T *__restrict__ A_val, const T *__restrict__ u, Epi epi, const int4 *__restrict__ meta, int block_first, int nblocks, int row_lo, int row_hi, int xcd_window)
{
    __shared__ T x[kBlockNnz];
    __shared__ int sp[kBlockNnz + 1];
    constexpr int kIts = kBlockNnz / kBlock;
    constexpr int kRowIts = kBlockNnz / kBlock;
    const int G = gridDim.x;
    int b = fdd_xcd_windowed_block(blockIdx.x, G, xcd_window);
    if (b >= nblocks) return;

    // descriptor -> (r0, nrows, base, nnz) of the rows of the block that lie in [row_lo, row_hi); only the two blocks that
    // straddle the range's ends take the branch
    auto bounds = [&](int4 m, int &r0, int &nrows, int &base, int &nnz) {
        r0 = m.x;
        int r1 = m.y;
        base = m.z;
        nnz = m.w;
        if (r0 < row_lo || r1 > row_hi)
        {
            r0 = r0 > row_lo ? r0 : row_lo;
            r1 = r1 < row_hi ? r1 : row_hi;
            if (r1 < r0) r1 = r0;
            base = A_ptr[r0];
            nnz = A_ptr[r1] - base;
        }
        nrows = r1 - r0;
    };
    auto load_streams = [&](int (&c)[kIts], T (&a)[kIts], int (&rp)[kRowIts], typename Epi::Opnd (&o)[kRowIts], int r0, int nrows, int base, int nnz) {
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            const int ks = (k < nnz) ? base + k : ((nnz > 0) ? base : 0); // unconditional loads on a selected index
            c[it] = __builtin_nontemporal_load(A_col + ks);
            a[it] = UNIT ? T(1) : __builtin_nontemporal_load(A_val + ks);
        }
#pragma unroll
        for (int it = 0; it < kRowIts; it++)
        {
            const int r = threadIdx.x + it * kBlock;
            const int rs = (r < nrows) ? r : 0;
            rp[it] = A_ptr[r0 + rs + 1];
            o[it] = epi.operand(r0 + rs, Au);
        }
    };

    int r0, nrows, base, nnz;
    bounds(meta[block_first + b], r0, nrows, base, nnz);
    int c[kIts], rp[kRowIts];
    T a[kIts];
    typename Epi::Opnd o[kRowIts];
    load_streams(c, a, rp, o, r0, nrows, base, nnz);
    int bn = b + G;
    int4 mn = meta[block_first + (bn < nblocks ? bn : b)]; // descriptor of the next block, one iteration ahead of its use

    for (;;)
    {
        const bool more = bn < nblocks;
        // value gathers of this block (its indices were requested one iteration ago)
        T v[kIts];
#pragma unroll
        for (int it = 0; it < kIts; it++) v[it] = u[c[it]];
        // streams of the next block, and the descriptor of the one after, behind the gathers
        int r0n, nrowsn, basen, nnzn;
        bounds(mn, r0n, nrowsn, basen, nnzn);
        int cn[kIts], rpn[kRowIts];
        T an[kIts];
        typename Epi::Opnd on[kRowIts];
        load_streams(cn, an, rpn, on, r0n, nrowsn, basen, nnzn);
        const int bnn = bn + G;
        const int4 mnn = meta[block_first + (bnn < nblocks ? bnn : (more ? bn : b))];

#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            if (k < nnz) x[k] = a[it] * v[it]; // UNIT: 1.0 * v, the reference's boolean rows, bit for bit
        }
        if (threadIdx.x == 0) sp[0] = 0;
#pragma unroll
        for (int it = 0; it < kRowIts; it++)
        {
            const int r = threadIdx.x + it * kBlock;
            if (r < nrows) sp[r + 1] = rp[it] - base;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kRowIts; it++)
        {
            const int r = threadIdx.x + it * kBlock;
            if (r < nrows)
            {
                T s = T(0);
                for (int j = sp[r]; j < sp[r + 1]; j++) s += x[j];
                Au[r0 + r] = epi.finish(s, o[it], r0 + r);
            }
        }
        if (!more) break; // every wave leaves here together: `more` is workgroup-uniform
        __syncthreads();  // x and sp are rewritten by the next block
        b = bn;
        bn = bnn;
        mn = mnn;
        r0 = r0n;
        nrows = nrowsn;
        base = basen;
        nnz = nnzn;
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            c[it] = cn[it];
            a[it] = an[it];
        }
#pragma unroll
        for (int it = 0; it < kRowIts; it++)
        {
            rp[it] = rpn[it];
            o[it] = on[it];
        }
    }
}

// the gather of the float preconditioner (subdomain.okl's kernels instantiated with DType = float): no epilogue
struct EpiPlainF32
{
    typedef float Opnd;
    __device__ float operand(int, const float *) const { return 0.0f; }
    __device__ float finish(float s, float, int) const { return s; }
};

// workgroups per CU of the persistent gather (0: one row block per workgroup, the forms above).  C2's Qt: 72 us with one
// block per workgroup, 73.6 / 59.1 / 62.9 / 63.3 us with 2 / 4 / 6 / 8 persistent workgroups per CU
inline int gather_pipelined_per_cu()
{
    static const int v = fdd_env_int("FDD_TUNE_GATHER_PIPELINED", 4);
    return v;
}

// sum_nodes s*s*w with s = (Qt u)[node]*w[node], on the same row blocks; a
// capped grid strides over the blocks, one partial per workgroup
template <int kBlockNnz>
__global__ __launch_bounds__(kBlock) void gather_norm2_block_kernel(double *__restrict__ ws, const int *__restrict__ Qt_ptr, const int *__restrict__ Qt_col, const double *__restrict__ u, const double *__restrict__ w, const int *__restrict__ row_blocks, int num_blocks)
{
    __shared__ double x[kBlockNnz];
    __shared__ double wsum[kBlock / FDD_WAVE];
    constexpr int kIts = kBlockNnz / kBlock;
    double acc = 0.0;

    for (int b = blockIdx.x; b < num_blocks; b += gridDim.x)
    {
        const int r0 = row_blocks[b];
        const int r1 = row_blocks[b + 1];
        const int base = Qt_ptr[r0];
        const int nnz = Qt_ptr[r1] - base;
        int c[kIts];
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            c[it] = Qt_col[(k < nnz) ? base + k : ((nnz > 0) ? base : 0)]; // unconditional on a selected index: all loads in flight (an all-empty block reads entry 0)
        }
        __syncthreads(); // previous block's readers of x are done
#pragma unroll
        for (int it = 0; it < kIts; it++)
        {
            const int k = threadIdx.x + it * kBlock;
            const double v = u[c[it]];
            if (k < nnz) x[k] = 1.0 * v;
        }
        __syncthreads();
        for (int row = r0 + threadIdx.x; row < r1; row += kBlock)
        {
            const int j0 = Qt_ptr[row] - base;
            const int j1 = Qt_ptr[row + 1] - base;
            double s = 0.0;
            for (int j = j0; j < j1; j++) s += x[j];
            const double wn = w[row];
            s = s * wn;
            acc += s * s * wn;
        }
    }

    acc = wave_sum(acc);
    if ((threadIdx.x & (FDD_WAVE - 1)) == 0) wsum[threadIdx.x / FDD_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double tsum = wsum[0];
#pragma unroll
        for (int k = 1; k < kBlock / FDD_WAVE; k++) tsum += wsum[k];
        ws[blockIdx.x] = tsum;
    }
}

__global__ __launch_bounds__(kBlock) void fold_partials_kernel(double *out, const double *ws, int n)
{
    __shared__ double wsum[kBlock / FDD_WAVE];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += ws[i];
    s = wave_sum(s);
    if ((threadIdx.x & (FDD_WAVE - 1)) == 0) wsum[threadIdx.x / FDD_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double tsum = wsum[0];
#pragma unroll
        for (int k = 1; k < kBlock / FDD_WAVE; k++) tsum += wsum[k];
        out[0] = tsum;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sliced ELL (slice = one wavefront of 64 rows, stored column-major, padded to the slice's longest row): the
// form for matrices whose rows are short and of nearly equal length -- the AMG levels of the low-order operator
// (7 entries per row on a rectilinear mesh, 15 on a deformed one, Galerkin stencils below), their interpolators.
// One lane per row: entry k of the 64 rows of a slice is one coalesced 256 B (columns) + 512 B (values) read,
// no row pointers, no LDS, no barrier; a row's products are added in column order (the CSR order), so results
// equal the row-block kernel's bit for bit (padding adds +0.0 * x[valid column]).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSellSlice = FDD_WAVE;

// Column readers of a slice.  Compact form (round 4): entry k of the 64 rows of a slice is stored as one wave-uniform base
// (the smallest column of the 64) + a 16-bit offset per lane -- on the lattice-numbered AMG levels the 64 rows of a slice
// read columns row + const in every slot, so the offsets are < 64 -- 10 instead of 12 bytes per entry (6 instead of 8 in
// single precision); a slice with a slot whose columns spread over more than 65535 keeps 32-bit columns.
struct SellCols32
{
    const int *c;
    __device__ __forceinline__ int at(int k) const { return __builtin_nontemporal_load(c + k * kSellSlice); }
};
struct SellCols16
{
    const unsigned short *c;
    const int *base; // wave-uniform address: scalar loads
    __device__ __forceinline__ int at(int k) const { return base[k] + (int)__builtin_nontemporal_load(c + k * kSellSlice); }
};

// B entries of the lane's row in flight together: B column and B value loads, then the B gathers, then the sum in column
// order.  FULL = false: only the first `count` (wave-uniform, >= 1) exist; the others re-read the last one (no branch around a
// load: a join would wait for everything in flight) and are left out of the sum by a select.
template <int B, bool FULL, typename T, typename Cols>
__device__ __forceinline__ void sell_batch(T &acc, const Cols cols, const T *__restrict__ v, const T *__restrict__ x, int k, int count)
{
    int c[B];
    T a[B], xv[B];
#pragma unroll
    for (int i = 0; i < B; i++)
    {
        const int ki = FULL ? k + i : min(k + i, k + count - 1);
        c[i] = cols.at(ki);
        a[i] = __builtin_nontemporal_load(v + ki * kSellSlice);
    }
#pragma unroll
    for (int i = 0; i < B; i++) xv[i] = x[c[i]];
#pragma unroll
    for (int i = 0; i < B; i++)
    {
        const T t = acc + a[i] * xv[i];
        acc = (FULL || i < count) ? t : acc;
    }
}

// A wave's time on a slice is a chain of dependent round trips (slice bounds -> columns -> gathers -> store) and a CU holds
// 32 waves: what counts is how few trips a row takes, not its bytes -- groups of eight entries (a level-0 row of 7 entries is
// one trip through columns and gathers; four per group, then one by one: three)
template <typename T, typename Cols>
__device__ __forceinline__ T sell_row_sum(const Cols cols, const T *__restrict__ v, const T *__restrict__ x, int width)
{
    T acc = T(0);
    int k = 0;
    for (; k + 8 <= width; k += 8) sell_batch<8, true>(acc, cols, v, x, k, 8);
    const int rem = width - k;
    if (rem > 4)
        sell_batch<8, false>(acc, cols, v, x, k, rem);
    else if (rem > 0)
        sell_batch<4, false>(acc, cols, v, x, k, rem);
    return acc;
}

template <typename T, typename Epi>
__global__ __launch_bounds__(kBlock) void sell_kernel(T *__restrict__ y, const int *__restrict__ slice_off, const int *__restrict__ slice_order, const int *__restrict__ col, const T *__restrict__ val, const T *__restrict__ x, Epi epi, int num_rows, int num_slices,
                                                      const unsigned short *__restrict__ col16, const int *__restrict__ slot_base, const int *__restrict__ wide_off)
{
    const int turn = blockIdx.x * (kBlock / kSellSlice) + threadIdx.x / kSellSlice;
    if (turn >= num_slices) return;
    // slices are visited widest first when their widths differ much (a few slices of long rows -- the interface and
    // superdomain rows of a composite's low-order operator -- would otherwise finish long after everything else)
    const int slice = __builtin_amdgcn_readfirstlane(slice_order ? slice_order[turn] : turn);
    const int lane = threadIdx.x & (kSellSlice - 1);
    const int row = slice * kSellSlice + lane;
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) / kSellSlice;
    const int rs = (row < num_rows) ? row : num_rows - 1;
    const typename Epi::Opnd opnd = epi.operand(rs, y); // requested with the first entries
    const T *v = val + off + lane;
    T acc;
    const int woff = col16 ? wide_off[slice] : 0;
    if (col16 != nullptr && woff < 0)
        acc = sell_row_sum<T>(SellCols16{col16 + off + lane, slot_base + off / kSellSlice}, v, x, width);
    else
        acc = sell_row_sum<T>(SellCols32{col + (col16 ? woff : off) + lane}, v, x, width);
    if (row < num_rows) y[row] = epi.finish(acc, opnd, row);
}

// which slices can take the compact column form: every slot's 64 columns within 65535 of the smallest
__global__ __launch_bounds__(kBlock) void sell_probe_kernel(int *__restrict__ wide, const int *__restrict__ slice_off, const int *__restrict__ A_ptr, const int *__restrict__ A_col, int num_rows, int num_slices)
{
    const int slice = blockIdx.x * (kBlock / kSellSlice) + threadIdx.x / kSellSlice;
    if (slice >= num_slices) return;
    const int lane = threadIdx.x & (kSellSlice - 1);
    const int row = slice * kSellSlice + lane;
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) / kSellSlice;
    const int p0 = (row < num_rows) ? A_ptr[row] : 0;
    const int len = (row < num_rows) ? A_ptr[row + 1] - p0 : 0;
    const int pad_col = (len > 0) ? A_col[p0 + len - 1] : -1;
    int too_wide = 0;
    for (int k = 0; k < width; k++)
    {
        const int c = (k < len) ? A_col[p0 + k] : pad_col;
        int lo = (c >= 0) ? c : 0x7fffffff, hi = c;
        for (int d = kSellSlice / 2; d > 0; d >>= 1)
        {
            lo = min(lo, __shfl_xor(lo, d));
            hi = max(hi, __shfl_xor(hi, d));
        }
        if (hi >= 0 && hi - lo > 65535) too_wide = 1;
    }
    if (lane == 0) wide[slice] = too_wide;
}

// CSR -> sliced ELL on the device: one wavefront per slice.  col16 != nullptr: the compact column form where wide_off[slice] < 0,
// 32-bit columns at scol + wide_off[slice] otherwise (scol then holds the wide slices only)
template <typename T>
__global__ __launch_bounds__(kBlock) void sell_fill_kernel(int *__restrict__ scol, T *__restrict__ sval, const int *__restrict__ slice_off, const int *__restrict__ A_ptr, const int *__restrict__ A_col, const T *__restrict__ A_val, int num_rows, int num_slices,
                                                           unsigned short *__restrict__ col16, int *__restrict__ slot_base, const int *__restrict__ wide_off)
{
    const int slice = blockIdx.x * (kBlock / kSellSlice) + threadIdx.x / kSellSlice;
    if (slice >= num_slices) return;
    const int lane = threadIdx.x & (kSellSlice - 1);
    const int row = slice * kSellSlice + lane;
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) / kSellSlice;
    const int p0 = (row < num_rows) ? A_ptr[row] : 0;
    const int len = (row < num_rows) ? A_ptr[row + 1] - p0 : 0;
    // padding repeats a column the row already reads -- its last: in the padded (late) slots the neighbouring rows read their
    // last columns too, which keeps a slot's columns close together -- with value 0; an empty row takes the slot's smallest
    // column of the other rows (column 0 where all are empty)
    const int pad_col = (len > 0) ? A_col[p0 + len - 1] : -1;
    const int woff = col16 ? wide_off[slice] : off;
    for (int k = 0; k < width; k++)
    {
        int c = (k < len) ? A_col[p0 + k] : pad_col;
        int lo = (c >= 0) ? c : 0x7fffffff;
        for (int d = kSellSlice / 2; d > 0; d >>= 1) lo = min(lo, __shfl_xor(lo, d));
        if (lo == 0x7fffffff) lo = 0;
        if (c < 0) c = lo;
        if (col16 != nullptr && woff < 0)
        {
            col16[off + k * kSellSlice + lane] = (unsigned short)(c - lo);
            if (lane == 0) slot_base[off / kSellSlice + k] = lo;
        }
        else
            scol[woff + k * kSellSlice + lane] = c;
        sval[off + k * kSellSlice + lane] = (k < len) ? A_val[p0 + k] : T(0);
    }
}

} // namespace

struct fdd_csr_plan
{
    int num_rows;
    int num_cols;
    int num_nnz;
    int kind; // 0: thread-per-row, 1: LDS-staged row blocks
    int unit_values; // every stored value is exactly 1.0: val need not be read
    int has_long_rows; // some row exceeds a block (workgroup-reduced in SpMV)
    int block_nnz;     // kBlockNnzSmall or kBlockNnzMax: non-zeros (and rows) per row block
    int xcd_chunked;   // SpMV row blocks in XCD-chunked order
    int one_per_row;   // kind 0 with exactly one entry in every row: the row pointers are not read
    int value_bytes;   // 8: the fp64 entries; 4: a plan of the f32 entries (always row blocks; twice the non-zeros per block measured no faster)
    int num_blocks;
    int *row_blocks_dev; // num_blocks + 1
    int4 *block_meta_dev = nullptr; // per row block {first row, end row, first non-zero, non-zeros}: one 16-byte scalar load instead of a chain row_blocks -> A_ptr (the persistent gather prefetches it two blocks ahead)
    std::vector<int> row_blocks_host;
    // sliced-ELL copy of the matrix (fdd_csr_plan_attach_sell): the SpMV entries then run on it
    int sell_slices = 0;
    int *sell_off_dev = nullptr; // sell_slices + 1 entry offsets
    int *sell_col_dev = nullptr;
    void *sell_val_dev = nullptr;
    int *sell_order_dev = nullptr; // slices by decreasing width (nullptr: widths are even, natural order)
    long long sell_entries = 0;
    // compact column form: 16-bit offsets from a per-(slice, slot) base; sell_col_dev then holds the wide slices' columns only
    unsigned short *sell_col16_dev = nullptr;
    int *sell_slot_base_dev = nullptr; // sell_entries / 64
    int *sell_wide_off_dev = nullptr;  // per slice: -1 = compact, else its offset in sell_col_dev
    int sell_compact_slices = 0;
};

// y[rows of blocks first..last) in [row_lo, row_hi)] = epi(A x) on a short-row plan: the persistent pipelined kernel.
// false: not applicable (the caller takes the one-block-per-workgroup form)
template <typename T, typename Epi>
static bool launch_short_pipelined(const fdd_csr_plan *plan, T *y, const int *ptr, const int *col, const T *val, const T *x, const Epi &epi, int first, int last, int row_lo, int row_hi, hipStream_t s, bool unit)
{
    const int per_cu = gather_pipelined_per_cu();
    if (per_cu <= 0 || plan->block_nnz != kBlockNnzSmall || plan->block_meta_dev == nullptr || plan->has_long_rows) return false;
    const int nblocks = last - first;
    if (nblocks <= 0) return true;
    int g = per_cu * FDD_CU_COUNT;
    if (g > nblocks) g = nblocks;
    static const int xcd_window = fdd_env_int("FDD_TUNE_DSSUM_XCD_WINDOW", 32);
    if (unit)
        hipLaunchKernelGGL((csr_short_pipelined_kernel<T, Epi, true, kBlockNnzSmall>), dim3(g), dim3(kBlock), 0, s, y, ptr, col, val, x, epi, plan->block_meta_dev, first, nblocks, row_lo, row_hi, xcd_window);
    else
        hipLaunchKernelGGL((csr_short_pipelined_kernel<T, Epi, false, kBlockNnzSmall>), dim3(g), dim3(kBlock), 0, s, y, ptr, col, val, x, epi, plan->block_meta_dev, first, nblocks, row_lo, row_hi, xcd_window);
    return true;
}

// the boolean gather t = Qt u (no values read, no epilogue)
static bool launch_gather_pipelined(const fdd_csr_plan *plan, double *t, const int *ptr, const int *col, const double *u, int first, int last, int row_lo, int row_hi, hipStream_t s)
{
    if (!plan->unit_values) return false;
    return launch_short_pipelined<double, EpiPlain>(plan, t, ptr, col, nullptr, u, EpiPlain{}, first, last, row_lo, row_hi, s, true);
}
static bool launch_gather_pipelined(const fdd_csr_plan *plan, float *t, const int *ptr, const int *col, const float *u, int first, int last, int row_lo, int row_hi, hipStream_t s)
{
    if (!plan->unit_values) return false;
    return launch_short_pipelined<float, EpiPlainF32>(plan, t, ptr, col, nullptr, u, EpiPlainF32{}, first, last, row_lo, row_hi, s, true);
}

template <typename T, typename Epi>
static int sell_launch(const fdd_csr_plan *plan, T *y, const T *x, const Epi &epi, void *stream)
{
    const int per_block = kBlock / kSellSlice;
    const dim3 grid((plan->sell_slices + per_block - 1) / per_block), block(kBlock);
    hipLaunchKernelGGL((sell_kernel<T, Epi>), grid, block, 0, fdd_stream(stream), y, plan->sell_off_dev, plan->sell_order_dev, plan->sell_col_dev, (const T *)plan->sell_val_dev, x, epi, plan->num_rows, plan->sell_slices,
                       plan->sell_col16_dev, plan->sell_slot_base_dev, plan->sell_wide_off_dev);
    FDD_LAUNCH_CHECK();
    return 0;
}

#define FDD_CSR_BLOCK(EPI, UNIT, ...)                                                          \
    do                                                                                         \
    {                                                                                          \
        if (plan->block_nnz == kBlockNnzSmall)                                                 \
            hipLaunchKernelGGL((csr_block_kernel<double, EPI, UNIT, kBlockNnzSmall>), __VA_ARGS__);    \
        else                                                                                   \
            hipLaunchKernelGGL((csr_block_kernel<double, EPI, UNIT, kBlockNnzMax>), __VA_ARGS__);      \
    } while (0)

template <typename Epi>
static int plan_launch(const fdd_csr_plan *plan, double *y, const int *A_ptr, const int *A_col, const double *A_val, const double *x, const Epi &epi, void *stream)
{
    if (plan->value_bytes != 8)
    {
        fdd_set_error("fp64 SpMV on a plan of fdd_csr_plan_create_f32");
        return 1;
    }
    if (plan->sell_slices > 0) return sell_launch<double, Epi>(plan, y, x, epi, stream);
    if (plan->one_per_row) return launch_one_per_row(y, A_col, A_val, x, epi, plan->num_rows, stream, plan->unit_values != 0);
    if (plan->kind == 0) return launch_rows(y, A_ptr, A_col, A_val, x, epi, 0, plan->num_rows, stream, plan->unit_values != 0);
    const dim3 grid(plan->num_blocks), block(kBlock);
    static const int split_rows = fdd_env_int("FDD_TUNE_CSR_SPLIT_ROWS", 1);
    if (launch_short_pipelined<double, Epi>(plan, y, A_ptr, A_col, A_val, x, epi, 0, plan->num_blocks, 0, plan->num_rows, fdd_stream(stream), plan->unit_values != 0))
    {
        // a short-row plan (the gather Qt, the hanging-point rows, the AMG's interpolators): the persistent pipelined
        // kernel, the same products and sums in the same order
    }
    else if (plan->unit_values)
        FDD_CSR_BLOCK(Epi, true, grid, block, 0, fdd_stream(stream), y, A_ptr, A_col, A_val, x, epi, plan->row_blocks_dev, plan->xcd_chunked, split_rows);
    else
        FDD_CSR_BLOCK(Epi, false, grid, block, 0, fdd_stream(stream), y, A_ptr, A_col, A_val, x, epi, plan->row_blocks_dev, plan->xcd_chunked, split_rows);
    FDD_LAUNCH_CHECK();
    return 0;
}

// the Float = float V-cycle (AMG/config.hpp:4): row-block plans only (fdd_csr_plan_create_blocked)
template <typename Epi>
static int plan_launch_f32(const fdd_csr_plan *plan, float *y, const int *A_ptr, const int *A_col, const float *A_val, const float *x, const Epi &epi, void *stream)
{
    if (plan->kind != 1 || plan->value_bytes != 4)
    {
        fdd_set_error("f32 SpMV: not a plan of fdd_csr_plan_create_f32");
        return 1;
    }
    if (plan->sell_slices > 0) return sell_launch<float, Epi>(plan, y, x, epi, stream);
    const dim3 grid(plan->num_blocks), block(kBlock);
    static const int split_rows = fdd_env_int("FDD_TUNE_CSR_SPLIT_ROWS", 1);
    if (launch_short_pipelined<float, Epi>(plan, y, A_ptr, A_col, A_val, x, epi, 0, plan->num_blocks, 0, plan->num_rows, fdd_stream(stream), false))
    {
    }
    else if (plan->block_nnz == kBlockNnzSmall)
        hipLaunchKernelGGL((csr_block_kernel<float, Epi, false, kBlockNnzSmall>), grid, block, 0, fdd_stream(stream), y, A_ptr, A_col, A_val, x, epi, plan->row_blocks_dev, plan->xcd_chunked, split_rows);
    else
        hipLaunchKernelGGL((csr_block_kernel<float, Epi, false, kBlockNnzMax>), grid, block, 0, fdd_stream(stream), y, A_ptr, A_col, A_val, x, epi, plan->row_blocks_dev, plan->xcd_chunked, split_rows);
    FDD_LAUNCH_CHECK();
    return 0;
}

static int plan_create(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz, bool f32);

// Host -> device copy of a small setup table, complete on return.  Not hipMemcpy: that runs on the legacy default stream,
// which is ordered against every other stream of the process -- with several ranks in one process (one host thread
// each) another rank may be capturing its V-cycle graph at that moment, and the runtime then refuses the copy
// (hipErrorStreamCaptureImplicit).  A non-blocking stream of the calling thread has no such implicit edges.
static hipError_t upload_table(void *dst, const void *src, size_t bytes)
{
    static thread_local hipStream_t s = nullptr;
    hipError_t err = hipSuccess;
    if (s == nullptr) err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    return err;
}

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
    return launch_rows(y, ptr, col, val, x, EpiAxpby{alpha, beta, nullptr}, 0, num_rows, stream);
}

int fdd_csr_plan_create(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz) { return plan_create(plan, A_ptr_host, num_rows, num_cols, num_nnz, false); }

/* a plan for the f32 entries: row blocks whatever the row lengths (no lane-per-row form there) */
int fdd_csr_plan_create_f32(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz) { return plan_create(plan, A_ptr_host, num_rows, num_cols, num_nnz, true); }

} // extern "C"

static int plan_create(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz, bool f32)
{
    FDD_REQUIRE(plan != nullptr && num_rows >= 0 && num_cols >= 0 && num_nnz >= 0);
    FDD_REQUIRE(num_rows == 0 || A_ptr_host != nullptr);
    FDD_REQUIRE(num_rows == 0 || A_ptr_host[num_rows] - A_ptr_host[0] == num_nnz);

    fdd_csr_plan *p = new fdd_csr_plan();
    p->num_rows = num_rows;
    p->num_cols = num_cols;
    p->num_nnz = num_nnz;
    p->kind = 0;
    p->unit_values = 0;
    p->has_long_rows = 0;
    p->block_nnz = ((double)num_nnz < 4.0 * (double)num_rows) ? kBlockNnzSmall : kBlockNnzMax;
    if (const char *e = getenv("FDD_TUNE_CSR_BLOCK_NNZ")) p->block_nnz = (atoi(e) <= kBlockNnzSmall) ? kBlockNnzSmall : kBlockNnzMax;
    p->value_bytes = f32 ? 4 : 8;
    p->one_per_row = 0;
    // 0: dispatch order; 1: XCD-chunked; >= 2: XCD-windowed with that many consecutive row blocks per XCD.  Short-row
    // (boolean gather) plans and wide-row plans have their own knobs: what neighbouring blocks share differs.
    p->xcd_chunked = (p->block_nnz == kBlockNnzSmall) ? fdd_env_int("FDD_TUNE_CSR_XCD_SHORT", 32) : fdd_env_int("FDD_TUNE_CSR_XCD", 0);
    p->num_blocks = 0;
    p->row_blocks_dev = nullptr;

    // boolean gather/scatter matrices: thread-per-row is already minimal traffic
    // rows with exactly one entry (the scatter matrices Q): lane-per-row is minimal
    // traffic and perfectly balanced.  Anything with longer / ragged rows goes
    // through LDS row staging.  FDD_TUNE_CSR_ROW_BLOCK_THRESHOLD overrides (development).
    double threshold = 1.0;
    if (const char *e = getenv("FDD_TUNE_CSR_ROW_BLOCK_THRESHOLD")) threshold = atof(e);
    if (num_rows == 0 || (not f32 and (double)num_nnz <= threshold * (double)num_rows))
    {
        bool one = num_rows > 0 && num_nnz == num_rows && A_ptr_host[0] == 0;
        for (int i = 0; one && i < num_rows; i++) one = (A_ptr_host[i + 1] == i + 1);
        p->one_per_row = (one && fdd_env_int("FDD_TUNE_CSR_ONE_PER_ROW", 1)) ? 1 : 0;
        *plan = p;
        return 0;
    }

    // wide-row blocks hold at most one row per lane: every row's pointers and epilogue operands are then requested
    // with the block's other loads and nothing is loaded after the barrier (7-entry rows of the AMG's finest level:
    // V-cycle 2.97 -> 2.90 ms).  FDD_TUNE_CSR_ROW_CAP overrides (development).
    static const int wide_cap = fdd_env_int("FDD_TUNE_CSR_ROW_CAP", kBlock);
    const int row_cap = (p->block_nnz == kBlockNnzMax && wide_cap > 0) ? wide_cap : p->block_nnz;
    std::vector<int> blocks;
    blocks.push_back(0);
    int r = 0;
    while (r < num_rows)
    {
        const int base = A_ptr_host[r];
        int e = r;
        while (e < num_rows && (e - r) < row_cap && A_ptr_host[e + 1] - base <= p->block_nnz) e++;
        if (e == r)
        {
            e = r + 1; // one row longer than a block: workgroup-reduced
            p->has_long_rows = 1;
        }
        blocks.push_back(e);
        r = e;
    }

    p->kind = 1;
    p->num_blocks = (int)blocks.size() - 1;
    p->row_blocks_host = blocks;

    hipError_t err = hipMalloc((void **)&p->row_blocks_dev, blocks.size() * sizeof(int));
    if (err == hipSuccess) err = upload_table(p->row_blocks_dev, blocks.data(), blocks.size() * sizeof(int));
    if (err == hipSuccess)
    {
        std::vector<int4> meta((size_t)p->num_blocks);
        for (int b = 0; b < p->num_blocks; b++) meta[b] = make_int4(blocks[b], blocks[b + 1], A_ptr_host[blocks[b]], A_ptr_host[blocks[b + 1]] - A_ptr_host[blocks[b]]);
        err = hipMalloc((void **)&p->block_meta_dev, std::max<size_t>(meta.size(), 1) * sizeof(int4));
        if (err == hipSuccess && not meta.empty()) err = upload_table(p->block_meta_dev, meta.data(), meta.size() * sizeof(int4));
    }
    if (err != hipSuccess)
    {
        fdd_set_error("fdd_csr_plan_create: %s", hipGetErrorString(err));
        if (p->block_meta_dev) (void)hipFree(p->block_meta_dev);
        if (p->row_blocks_dev) (void)hipFree(p->row_blocks_dev);
        delete p;
        return (int)err;
    }

    *plan = p;
    return 0;
}

extern "C" {

int fdd_csr_plan_destroy(fdd_csr_plan *plan)
{
    if (plan == nullptr) return 0;
    if (plan->row_blocks_dev) (void)hipFree(plan->row_blocks_dev);
    if (plan->block_meta_dev) (void)hipFree(plan->block_meta_dev);
    if (plan->sell_off_dev) (void)hipFree(plan->sell_off_dev);
    if (plan->sell_col_dev) (void)hipFree(plan->sell_col_dev);
    if (plan->sell_val_dev) (void)hipFree(plan->sell_val_dev);
    if (plan->sell_order_dev) (void)hipFree(plan->sell_order_dev);
    if (plan->sell_col16_dev) (void)hipFree(plan->sell_col16_dev);
    if (plan->sell_slot_base_dev) (void)hipFree(plan->sell_slot_base_dev);
    if (plan->sell_wide_off_dev) (void)hipFree(plan->sell_wide_off_dev);
    delete plan;
    return 0;
}

// Give the plan a sliced-ELL copy of the matrix when its rows are short and even enough for it (padding within
// `max_padding` of the stored entries, no row longer than 64): 0 = attached, the SpMV entries use it from now on;
// *attached = 0 and nothing changes otherwise.  A_ptr_host is the host row-pointer array, the others device arrays
// (values double, or float for a plan of fdd_csr_plan_create_f32).
int fdd_csr_plan_attach_sell(fdd_csr_plan *plan, const int *A_ptr_host, const int *A_ptr, const int *A_col, const void *A_val, double max_padding, int *attached, void *stream)
{
    FDD_REQUIRE(plan != nullptr && attached != nullptr);
    *attached = 0;
    if (plan->num_rows == 0 || plan->num_nnz == 0 || plan->unit_values || plan->sell_slices > 0) return 0;
    if (!fdd_env_int("FDD_TUNE_CSR_SELL", 1)) return 0;
    // one lane per row pays for short rows; wide-row matrices (the coarser Galerkin operators, 40-160 entries per row)
    // stay on the row-block kernel (measured equal at 164 per row, far slower at 41 with rows sorted by length)
    // (round 4: 27-entry rows -- AMG level 1 of the geometric hierarchy -- are faster on the row-block kernel: reference-default
    // PCG step 8.79 -> 8.50 ms at C2 with the threshold at 16 instead of 32; 15-entry rows -- level 0 on a deformed mesh --
    // stay here: with 12 the float V-cycle on the Kershaw mesh loses 9 %)
    static const int sell_max_row = fdd_env_int("FDD_TUNE_CSR_SELL_MAX_ROW", 16);
    if ((long long)plan->num_nnz > (long long)sell_max_row * plan->num_rows) return 0;
    FDD_REQUIRE(A_ptr_host != nullptr && A_ptr != nullptr && A_col != nullptr && A_val != nullptr);
    const int n = plan->num_rows, slices = (n + kSellSlice - 1) / kSellSlice;
    std::vector<int> off(slices + 1, 0), width(slices, 0);
    long long total = 0;
    int w_min = 1 << 30, w_max = 0;
    for (int s = 0; s < slices; s++)
    {
        int w = 0;
        for (int r = s * kSellSlice; r < n && r < (s + 1) * kSellSlice; r++) w = std::max(w, A_ptr_host[r + 1] - A_ptr_host[r]);
        width[s] = w;
        w_min = std::min(w_min, w);
        w_max = std::max(w_max, w);
        total += (long long)w * kSellSlice;
        if (total > 2000000000LL) return 0;
        off[s + 1] = (int)total;
    }
    if ((double)total > max_padding * (double)plan->num_nnz) return 0;
    // a few wide slices among narrow ones: visit the widest first (stable, so that equal widths keep their order and
    // neighbouring waves stream neighbouring memory)
    std::vector<int> order;
    if (w_max > 64 || w_max > 4 * std::max(w_min, 1))
    {
        order.resize(slices);
        for (int s = 0; s < slices; s++) order[s] = s;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return width[a] > width[b]; });
    }
    const size_t vb = (size_t)plan->value_bytes;
    const int per_block = kBlock / kSellSlice;
    const dim3 grid((slices + per_block - 1) / per_block), block(kBlock);
    auto release = [&]() {
        if (plan->sell_order_dev) (void)hipFree(plan->sell_order_dev);
        if (plan->sell_off_dev) (void)hipFree(plan->sell_off_dev);
        if (plan->sell_col_dev) (void)hipFree(plan->sell_col_dev);
        if (plan->sell_val_dev) (void)hipFree(plan->sell_val_dev);
        if (plan->sell_col16_dev) (void)hipFree(plan->sell_col16_dev);
        if (plan->sell_slot_base_dev) (void)hipFree(plan->sell_slot_base_dev);
        if (plan->sell_wide_off_dev) (void)hipFree(plan->sell_wide_off_dev);
        plan->sell_order_dev = plan->sell_off_dev = plan->sell_col_dev = plan->sell_slot_base_dev = plan->sell_wide_off_dev = nullptr;
        plan->sell_val_dev = nullptr;
        plan->sell_col16_dev = nullptr;
        plan->sell_compact_slices = 0;
    };
    hipError_t err = hipMalloc((void **)&plan->sell_off_dev, off.size() * sizeof(int));
    if (err == hipSuccess) err = upload_table(plan->sell_off_dev, off.data(), off.size() * sizeof(int));
    // compact column form (16-bit offsets from a per-slot base) for the slices whose slots allow it
    std::vector<int> wide_off;
    long long wide_total = total;
    int compact = 0;
    if (err == hipSuccess && fdd_env_int("FDD_TUNE_CSR_SELL_COL16", 1))
    {
        std::vector<int> wide(slices, 1);
        err = hipMalloc((void **)&plan->sell_wide_off_dev, (size_t)slices * sizeof(int));
        if (err == hipSuccess)
        {
            hipLaunchKernelGGL(sell_probe_kernel, grid, block, 0, fdd_stream(stream), plan->sell_wide_off_dev, plan->sell_off_dev, A_ptr, A_col, n, slices);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipMemcpyAsync(wide.data(), plan->sell_wide_off_dev, (size_t)slices * sizeof(int), hipMemcpyDeviceToHost, fdd_stream(stream));
        if (err == hipSuccess) err = hipStreamSynchronize(fdd_stream(stream));
        if (err == hipSuccess)
        {
            wide_off.assign(slices, -1);
            wide_total = 0;
            for (int s = 0; s < slices; s++)
                if (wide[s])
                {
                    wide_off[s] = (int)wide_total;
                    wide_total += off[s + 1] - off[s];
                }
                else
                    compact++;
            if (compact == 0)
            {
                (void)hipFree(plan->sell_wide_off_dev);
                plan->sell_wide_off_dev = nullptr;
                wide_total = total;
            }
            else
                err = upload_table(plan->sell_wide_off_dev, wide_off.data(), (size_t)slices * sizeof(int));
        }
    }
    if (err == hipSuccess) err = hipMalloc((void **)&plan->sell_col_dev, (size_t)std::max<long long>(wide_total, 1) * sizeof(int));
    if (err == hipSuccess && compact > 0) err = hipMalloc((void **)&plan->sell_col16_dev, (size_t)total * sizeof(unsigned short));
    if (err == hipSuccess && compact > 0) err = hipMalloc((void **)&plan->sell_slot_base_dev, (size_t)(total / kSellSlice) * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&plan->sell_val_dev, (size_t)total * vb);
    if (err == hipSuccess && !order.empty()) err = hipMalloc((void **)&plan->sell_order_dev, order.size() * sizeof(int));
    if (err == hipSuccess && !order.empty()) err = upload_table(plan->sell_order_dev, order.data(), order.size() * sizeof(int));
    if (err != hipSuccess)
    {
        fdd_set_error("fdd_csr_plan_attach_sell: %s", hipGetErrorString(err));
        release();
        return (int)err;
    }
    if (vb == 8)
        hipLaunchKernelGGL((sell_fill_kernel<double>), grid, block, 0, fdd_stream(stream), plan->sell_col_dev, (double *)plan->sell_val_dev, plan->sell_off_dev, A_ptr, A_col, (const double *)A_val, n, slices, plan->sell_col16_dev, plan->sell_slot_base_dev,
                           plan->sell_wide_off_dev);
    else
        hipLaunchKernelGGL((sell_fill_kernel<float>), grid, block, 0, fdd_stream(stream), plan->sell_col_dev, (float *)plan->sell_val_dev, plan->sell_off_dev, A_ptr, A_col, (const float *)A_val, n, slices, plan->sell_col16_dev, plan->sell_slot_base_dev,
                           plan->sell_wide_off_dev);
    FDD_LAUNCH_CHECK();
    plan->sell_compact_slices = compact;
    plan->sell_entries = total;
    plan->sell_slices = slices; // last: the SpMV entries switch over
    *attached = 1;
    return 0;
}

// what fdd_csr_plan_attach_sell made: slices of 64 rows (0: no sliced-ELL copy) and how many of them carry the compact
// column form (16-bit offsets from a per-slot base: 10 instead of 12 bytes per entry)
int fdd_csr_plan_sell_info(const fdd_csr_plan *plan, int *slices, int *compact_slices)
{
    FDD_REQUIRE(plan != nullptr && slices != nullptr && compact_slices != nullptr);
    *slices = plan->sell_slices;
    *compact_slices = plan->sell_compact_slices;
    return 0;
}

int fdd_csr_plan_num_blocks(const fdd_csr_plan *plan, int *num_blocks)
{
    FDD_REQUIRE(plan != nullptr && num_blocks != nullptr);
    *num_blocks = plan->num_blocks;
    return 0;
}

// t[row] = sum of u over the row's entries, rows [row_lo, row_hi), float vectors, on the plan of a boolean gather matrix
int fdd_csr_plan_gather_f32(const fdd_csr_plan *plan, float *t, const int *Qt_ptr, const int *Qt_col, const float *u, int row_lo, int row_hi, void *stream)
{
    FDD_REQUIRE(plan != nullptr && row_lo >= 0 && row_hi >= row_lo && row_hi <= plan->num_rows);
    if (row_hi == row_lo) return 0;
    FDD_REQUIRE(t != nullptr && Qt_ptr != nullptr && Qt_col != nullptr && u != nullptr);
    if (plan->kind == 0 || plan->has_long_rows) return fdd_gather_rows_f32(t, Qt_ptr, Qt_col, u, row_lo, row_hi, stream);
    const std::vector<int> &rb = plan->row_blocks_host;
    int first = (int)(std::upper_bound(rb.begin(), rb.end(), row_lo) - rb.begin()) - 1;
    int last = (int)(std::lower_bound(rb.begin(), rb.end(), row_hi) - rb.begin()); // exclusive
    if (first < 0) first = 0;
    if (last > plan->num_blocks) last = plan->num_blocks;
    if (last <= first) return 0;
    const dim3 grid(last - first), block(kBlock);
    static const int xcd_window = fdd_env_int("FDD_TUNE_DSSUM_XCD_WINDOW", 32);
    if (launch_gather_pipelined(plan, t, Qt_ptr, Qt_col, u, first, last, row_lo, row_hi, fdd_stream(stream)))
    {
    }
    else if (plan->block_nnz == kBlockNnzSmall)
        hipLaunchKernelGGL((gather_block_f32_kernel<kBlockNnzSmall>), grid, block, 0, fdd_stream(stream), t, Qt_ptr, Qt_col, u, plan->row_blocks_dev, first, row_lo, row_hi, xcd_window);
    else
        hipLaunchKernelGGL((gather_block_f32_kernel<kBlockNnzMax>), grid, block, 0, fdd_stream(stream), t, Qt_ptr, Qt_col, u, plan->row_blocks_dev, first, row_lo, row_hi, xcd_window);
    FDD_LAUNCH_CHECK();
    return 0;
}

// dssum on a plan of the boolean gather matrix Qt; mode 0 = gather + scatter,
// 1 = gather only (t out), 2 = scatter only (t in); rows [row_lo, row_hi)
int fdd_csr_plan_dssum(const fdd_csr_plan *plan, double *QQtu, double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, const double *point_mask, int row_lo, int row_hi, int mode, void *stream)
{
    FDD_REQUIRE(plan != nullptr && mode >= 0 && mode <= 2);
    FDD_REQUIRE(row_lo >= 0 && row_hi >= row_lo && row_hi <= plan->num_rows);
    if (row_hi == row_lo) return 0;
    FDD_REQUIRE(plan->unit_values != 0); // boolean matrices only
    FDD_REQUIRE(Qt_ptr != nullptr && Qt_col != nullptr);
    FDD_REQUIRE(mode == 1 || QQtu != nullptr);
    FDD_REQUIRE(mode == 2 || u != nullptr);
    FDD_REQUIRE(mode == 0 || t != nullptr);

    if (plan->kind == 0 || plan->has_long_rows)
    {
        // no row blocks (rows of one entry) or rows longer than a block: lane-per-node form
        if (mode == 0) return fdd_dssum_fused(QQtu, t, Qt_ptr, Qt_col, u, node_weight, point_mask, row_lo, row_hi, stream);
        if (mode == 1) return fdd_dssum_gather(t, Qt_ptr, Qt_col, u, node_weight, row_lo, row_hi, stream);
        return fdd_dssum_scatter(QQtu, t, Qt_ptr, Qt_col, point_mask, row_lo, row_hi, stream);
    }

    // blocks overlapping the row range
    const std::vector<int> &rb = plan->row_blocks_host;
    int first = (int)(std::upper_bound(rb.begin(), rb.end(), row_lo) - rb.begin()) - 1;
    int last = (int)(std::lower_bound(rb.begin(), rb.end(), row_hi) - rb.begin()); // exclusive
    if (first < 0) first = 0;
    if (last > plan->num_blocks) last = plan->num_blocks;
    if (last <= first) return 0;

    const dim3 grid(last - first), block(kBlock);
    hipStream_t s = fdd_stream(stream);
    const bool W = node_weight != nullptr && mode != 2, M = point_mask != nullptr && mode != 1;
    static const int xcd_window = fdd_env_int("FDD_TUNE_DSSUM_XCD_WINDOW", 32); // consecutive row blocks per XCD inside a window of 8x as many (0: dispatch order).  C2's Qt gather: L2 fetches 415 -> 356 MB per launch (1.23 -> 1.05 x the 338 MB it must move; profiles/r04_pmc_traffic_c2*.json), 74 -> 72 us
#define FDD_DSB(MODE, WW, MM)                                                                                                                                                                    \
    do                                                                                                                                                                                          \
    {                                                                                                                                                                                           \
        if (plan->block_nnz == kBlockNnzSmall)                                                                                                                                                  \
            hipLaunchKernelGGL((dssum_block_kernel<MODE, WW, MM, kBlockNnzSmall>), grid, block, 0, s, QQtu, t, Qt_ptr, Qt_col, u, node_weight, point_mask, plan->row_blocks_dev, first, row_lo, row_hi, xcd_window); \
        else                                                                                                                                                                                    \
            hipLaunchKernelGGL((dssum_block_kernel<MODE, WW, MM, kBlockNnzMax>), grid, block, 0, s, QQtu, t, Qt_ptr, Qt_col, u, node_weight, point_mask, plan->row_blocks_dev, first, row_lo, row_hi, xcd_window);   \
    } while (0)
    if (mode == 0)
    {
        if (W && M) FDD_DSB(0, true, true);
        else if (W) FDD_DSB(0, true, false);
        else if (M) FDD_DSB(0, false, true);
        else FDD_DSB(0, false, false);
    }
    else if (mode == 1)
    {
        if (W) FDD_DSB(1, true, false);
        else if (launch_gather_pipelined(plan, t, Qt_ptr, Qt_col, u, first, last, row_lo, row_hi, s))
        {
            // persistent, pipelined form (FDD_TUNE_GATHER_PIPELINED = workgroups per CU; 0: one row block per workgroup)
        }
        else FDD_DSB(1, false, false);
    }
    else
    {
        if (M) FDD_DSB(2, false, true);
        else FDD_DSB(2, false, false);
    }
#undef FDD_DSB
    FDD_LAUNCH_CHECK();
    return 0;
}

// out[0] = sum_nodes s*s*w, s = (Qt u)[node]*w[node], on the plan's row blocks
int fdd_csr_plan_gather_weighted_norm2(const fdd_csr_plan *plan, double *out, double *ws, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, void *stream)
{
    FDD_REQUIRE(plan != nullptr && out != nullptr && ws != nullptr);
    hipStream_t s = fdd_stream(stream);
    if (plan->num_rows == 0) return (int)hipMemsetAsync(out, 0, sizeof(double), s);
    FDD_REQUIRE(plan->unit_values != 0 && Qt_ptr != nullptr && Qt_col != nullptr && u != nullptr && node_weight != nullptr);
    if (plan->kind == 0 || plan->has_long_rows) return fdd_gather_weighted_norm2(out, ws, Qt_ptr, Qt_col, u, node_weight, plan->num_rows, stream);
    const int grid = plan->num_blocks < FDD_REDUCE_MAX_BLOCKS ? plan->num_blocks : FDD_REDUCE_MAX_BLOCKS;
    if (plan->block_nnz == kBlockNnzSmall)
        hipLaunchKernelGGL(gather_norm2_block_kernel<kBlockNnzSmall>, dim3(grid), dim3(kBlock), 0, s, ws, Qt_ptr, Qt_col, u, node_weight, plan->row_blocks_dev, plan->num_blocks);
    else
        hipLaunchKernelGGL(gather_norm2_block_kernel<kBlockNnzMax>, dim3(grid), dim3(kBlock), 0, s, ws, Qt_ptr, Qt_col, u, node_weight, plan->row_blocks_dev, plan->num_blocks);
    FDD_LAUNCH_CHECK();
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(kBlock), 0, s, out, ws, grid);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_csr_plan_set_unit_values(fdd_csr_plan *plan, int unit_values)
{
    FDD_REQUIRE(plan != nullptr);
    FDD_REQUIRE(plan->value_bytes == 8);
    FDD_REQUIRE(plan->value_bytes == 8);
    plan->unit_values = unit_values != 0;
    return 0;
}

int fdd_csr_plan_kind(const fdd_csr_plan *plan, int *kind)
{
    FDD_REQUIRE(plan != nullptr && kind != nullptr);
    *kind = plan->kind;
    return 0;
}

// 1 where the plan's SpMV / gather entries run on the persistent pipelined short-row kernel (csr_short_pipelined_kernel):
// what a profile label should name
int fdd_csr_plan_pipelined(const fdd_csr_plan *plan, int *pipelined)
{
    FDD_REQUIRE(plan != nullptr && pipelined != nullptr);
    *pipelined = (gather_pipelined_per_cu() > 0 && plan->kind == 1 && plan->block_nnz == kBlockNnzSmall && plan->block_meta_dev != nullptr && !plan->has_long_rows && plan->sell_slices == 0) ? 1 : 0;
    return 0;
}

// launch csr_block_kernel for the plan's block size

int fdd_csr_plan_multiply(const fdd_csr_plan *plan, double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    FDD_REQUIRE(plan->value_bytes == 8);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0; // csr_matrix.tpp:304,334
    FDD_REQUIRE(Au != nullptr && A_ptr != nullptr && u != nullptr);

    if (plan->one_per_row)
    {
        if (weight) return launch_one_per_row(Au, A_col, A_val, u, EpiWeight{weight}, plan->num_rows, stream, plan->unit_values != 0);
        return launch_one_per_row(Au, A_col, A_val, u, EpiPlain{}, plan->num_rows, stream, plan->unit_values != 0);
    }
    if (plan->kind == 0)
    {
        if (weight) return launch_rows(Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, 0, plan->num_rows, stream, plan->unit_values != 0);
        return launch_rows(Au, A_ptr, A_col, A_val, u, EpiPlain{}, 0, plan->num_rows, stream, plan->unit_values != 0);
    }

    const dim3 grid(plan->num_blocks), block(kBlock);
    hipStream_t s = fdd_stream(stream);
    if (plan->unit_values)
    {
        if (weight)
        {
            if (!launch_short_pipelined<double, EpiWeight>(plan, Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, 0, plan->num_blocks, 0, plan->num_rows, s, true))
                FDD_CSR_BLOCK(EpiWeight, true, grid, block, 0, s, Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, plan->row_blocks_dev, plan->xcd_chunked, 0);
        }
        else if (launch_gather_pipelined(plan, Au, A_ptr, A_col, u, 0, plan->num_blocks, 0, plan->num_rows, s))
        {
            // boolean short rows (the gather Qt): the persistent pipelined gather, same sums in the same order
        }
        else
            FDD_CSR_BLOCK(EpiPlain, true, grid, block, 0, s, Au, A_ptr, A_col, A_val, u, EpiPlain{}, plan->row_blocks_dev, plan->xcd_chunked, 0);
    }
    else
    {
        // valued matrices: short rows on the persistent pipelined kernel
        if (weight)
        {
            if (!launch_short_pipelined<double, EpiWeight>(plan, Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, 0, plan->num_blocks, 0, plan->num_rows, s, false))
                FDD_CSR_BLOCK(EpiWeight, false, grid, block, 0, s, Au, A_ptr, A_col, A_val, u, EpiWeight{weight}, plan->row_blocks_dev, plan->xcd_chunked, 0);
        }
        else if (!launch_short_pipelined<double, EpiPlain>(plan, Au, A_ptr, A_col, A_val, u, EpiPlain{}, 0, plan->num_blocks, 0, plan->num_rows, s, false))
            FDD_CSR_BLOCK(EpiPlain, false, grid, block, 0, s, Au, A_ptr, A_col, A_val, u, EpiPlain{}, plan->row_blocks_dev, plan->xcd_chunked, 0);
    }
    FDD_LAUNCH_CHECK();
    return 0;
}

// y = alpha*A*x + beta*y on a plan (the cusparseSpMV of AMG/csr_matrix.cpp:129-131); y must not alias x
int fdd_csr_plan_matvec(const fdd_csr_plan *plan, double *y, const int *A_ptr, const int *A_col, const double *A_val, const double *x, double alpha, double beta, void *stream)
{
    return fdd_csr_plan_matvec_to(plan, y, nullptr, A_ptr, A_col, A_val, x, alpha, beta, stream);
}

int fdd_csr_plan_matvec_to(const fdd_csr_plan *plan, double *y, const double *y_in, const int *A_ptr, const int *A_col, const double *A_val, const double *x, double alpha, double beta, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(y != nullptr && A_ptr != nullptr && x != nullptr && y != x);
    return plan_launch(plan, y, A_ptr, A_col, A_val, x, EpiAxpby{alpha, beta, y_in}, stream);
}

// The Chebyshev smoother with its element-wise kernels as SpMV epilogues (same arithmetic, statement for
// statement, as scaled_residual / polynomial_evaluation / update_field around matvec: subdomain.tpp:19-83).
// Sr = D*(f - A u), work = D*(coef*Sr)
int fdd_amg_smooth_residual_matvec(const fdd_csr_plan *plan, double *work, double *Sr, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *f, const double *D_val, double coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(work != nullptr && Sr != nullptr && A_ptr != nullptr && u != nullptr && f != nullptr && D_val != nullptr && work != u && Sr != u);
    return plan_launch(plan, work, A_ptr, A_col, A_val, u, EpiSmoothResidual{f, D_val, Sr, coef}, stream);
}

// work_out = D*(coef*Sr + D*(A work_in))
int fdd_amg_smooth_polynomial_matvec(const fdd_csr_plan *plan, double *work_out, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(work_out != nullptr && A_ptr != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && work_out != work_in);
    return plan_launch(plan, work_out, A_ptr, A_col, A_val, work_in, EpiSmoothPoly{Sr, D_val, coef}, stream);
}

// u += D*(coef*Sr + D*(A work_in))
int fdd_amg_smooth_update_matvec(const fdd_csr_plan *plan, double *u, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(u != nullptr && A_ptr != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && u != work_in);
    return plan_launch(plan, u, A_ptr, A_col, A_val, work_in, EpiSmoothUpdate{Sr, D_val, coef}, stream);
}

// u = 0 + D*(coef*Sr + D*(A work_in)): the update above from a zero u, which is not read
int fdd_amg_smooth_update_matvec_from_zero(const fdd_csr_plan *plan, double *u, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(u != nullptr && A_ptr != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && u != work_in);
    return plan_launch(plan, u, A_ptr, A_col, A_val, work_in, EpiSmoothUpdateZeroT<double>{Sr, D_val, coef}, stream);
}

// ---- Float = float (AMG/config.hpp:4): the V-cycle's SpMV and fused smoother on f32 values and vectors ----
int fdd_csr_plan_matvec_to_f32(const fdd_csr_plan *plan, float *y, const float *y_in, const int *A_ptr, const int *A_col, const float *A_val, const float *x, float alpha, float beta, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(y != nullptr && A_ptr != nullptr && A_val != nullptr && x != nullptr && y != x);
    return plan_launch_f32(plan, y, A_ptr, A_col, A_val, x, EpiAxpbyT<float>{alpha, beta, y_in}, stream);
}

int fdd_amg_smooth_residual_matvec_f32(const fdd_csr_plan *plan, float *work, float *Sr, const int *A_ptr, const int *A_col, const float *A_val, const float *u, const float *f, const float *D_val, float coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(work != nullptr && Sr != nullptr && A_ptr != nullptr && A_val != nullptr && u != nullptr && f != nullptr && D_val != nullptr && work != u && Sr != u);
    return plan_launch_f32(plan, work, A_ptr, A_col, A_val, u, EpiSmoothResidualT<float>{f, D_val, Sr, coef}, stream);
}

int fdd_amg_smooth_polynomial_matvec_f32(const fdd_csr_plan *plan, float *work_out, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(work_out != nullptr && A_ptr != nullptr && A_val != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && work_out != work_in);
    return plan_launch_f32(plan, work_out, A_ptr, A_col, A_val, work_in, EpiSmoothPolyT<float>{Sr, D_val, coef}, stream);
}

int fdd_amg_smooth_update_matvec_f32(const fdd_csr_plan *plan, float *u, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(u != nullptr && A_ptr != nullptr && A_val != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && u != work_in);
    return plan_launch_f32(plan, u, A_ptr, A_col, A_val, work_in, EpiSmoothUpdateT<float>{Sr, D_val, coef}, stream);
}

int fdd_amg_smooth_update_matvec_from_zero_f32(const fdd_csr_plan *plan, float *u, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream)
{
    FDD_REQUIRE(plan != nullptr);
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    FDD_REQUIRE(u != nullptr && A_ptr != nullptr && A_val != nullptr && work_in != nullptr && Sr != nullptr && D_val != nullptr && u != work_in);
    return plan_launch_f32(plan, u, A_ptr, A_col, A_val, work_in, EpiSmoothUpdateZeroT<float>{Sr, D_val, coef}, stream);
}

} // extern "C"
