// Dot-product style reductions (HBM-bound): the four Domain dot kernels
// (domain.okl:109-264), the Subdomain ones (subdomain.okl:103-258) and the
// cublasDdot replacement of AMG/vector.cpp.
//
// The reference writes one partial per 128-thread block, copies all of them to
// the host (131 072 doubles at config C2) and sums them there.  Here a capped
// grid (<= 2048 workgroups) strides over the vectors with 16-B loads, each
// lane keeps a private sum, a wavefront __shfl_down tree and a 4-wave LDS step
// give one partial per workgroup, and a second one-workgroup launch folds the
// partials into the final scalar on the device.  The result is deterministic
// for a given n (fixed grid, fixed tree) but its summation order differs from
// the reference's 128-wide tree + serial block sum: parity is
// tolerance-based (tests/).
#include "fdd_common.h"

namespace
{

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / FDD_WAVE;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = FDD_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, FDD_WAVE);
    return v;
}

// NV values per element (1 or 2 simultaneous sums)
template <int NV>
struct Acc
{
    double v[NV];
};

template <int NV>
__device__ __forceinline__ void block_reduce_store(Acc<NV> a, double *ws, int nblocks_stride)
{
    __shared__ double s[NV][kWaves];
    const int lane = threadIdx.x & (FDD_WAVE - 1);
    const int wave = threadIdx.x / FDD_WAVE;

#pragma unroll
    for (int k = 0; k < NV; k++)
    {
        double x = wave_sum(a.v[k]);
        if (lane == 0) s[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int k = 0; k < NV; k++)
        {
            double x = s[k][0];
#pragma unroll
            for (int w = 1; w < kWaves; w++) x += s[k][w];
            ws[blockIdx.x + k * nblocks_stride] = x;
        }
    }
}

template <typename Op>
__global__ __launch_bounds__(kBlock) void reduce_vec2_kernel(Op op, double *ws, long long n2, long long n)
{
    Acc<Op::NV> a;
#pragma unroll
    for (int k = 0; k < Op::NV; k++) a.v[k] = 0.0;

    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) op.vec2(i, a);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) op.one(n - 1, a);

    block_reduce_store<Op::NV>(a, ws, FDD_REDUCE_MAX_BLOCKS);
}

template <typename Op>
__global__ __launch_bounds__(kBlock) void reduce_scalar_kernel(Op op, double *ws, long long n)
{
    Acc<Op::NV> a;
#pragma unroll
    for (int k = 0; k < Op::NV; k++) a.v[k] = 0.0;

    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) op.one(i, a);

    block_reduce_store<Op::NV>(a, ws, FDD_REDUCE_MAX_BLOCKS);
}

// second stage: one workgroup folds `nblocks` partials per value
template <int NV>
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(double *out, const double *ws, int nblocks)
{
    __shared__ double s[NV][kWaves];
    const int lane = threadIdx.x & (FDD_WAVE - 1);
    const int wave = threadIdx.x / FDD_WAVE;

    // every partial this lane folds is requested before the first one is added (one round trip to the partials the
    // previous launch left in memory instead of one per partial); the sums and their order are those of the plain loop
    constexpr int kPer = FDD_REDUCE_MAX_BLOCKS / kBlock;
    double v[NV][kPer];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int i = 0; i < kPer; i++)
        {
            const int b = threadIdx.x + i * kBlock;
            v[k][i] = ws[(b < nblocks ? b : 0) + k * FDD_REDUCE_MAX_BLOCKS];
        }
#pragma unroll
    for (int k = 0; k < NV; k++)
    {
        double x = 0.0;
#pragma unroll
        for (int i = 0; i < kPer; i++)
            if (threadIdx.x + i * kBlock < nblocks) x += v[k][i];
        x = wave_sum(x);
        if (lane == 0) s[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int k = 0; k < NV; k++)
        {
            double x = s[k][0];
#pragma unroll
            for (int w = 1; w < kWaves; w++) x += s[k][w];
            out[k] = x;
        }
    }
}

template <typename Op>
int launch_reduce(const Op &op, double *out, double *ws, long long n, bool aligned, void *stream)
{
    hipStream_t s = fdd_stream(stream);
    if (n <= 0)
    {
        // empty sums are zero; keep it asynchronous
        return (int)hipMemsetAsync(out, 0, sizeof(double) * Op::NV, s);
    }

    int grid;
    if (aligned && n >= 2)
    {
        long long n2 = n / 2;
        grid = fdd_stream_grid(n2, kBlock);
        hipLaunchKernelGGL(reduce_vec2_kernel<Op>, dim3(grid), dim3(kBlock), 0, s, op, ws, n2, n);
    }
    else
    {
        grid = fdd_stream_grid(n, kBlock);
        hipLaunchKernelGGL(reduce_scalar_kernel<Op>, dim3(grid), dim3(kBlock), 0, s, op, ws, n);
    }
    FDD_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel<Op::NV>, dim3(1), dim3(kBlock), 0, s, out, ws, grid);
    FDD_LAUNCH_CHECK();
    return 0;
}

typedef double fdd_v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2(const double *p, long long i) // non-temporal: read once per launch
{
    const fdd_v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const fdd_v2f64 *>(p) + i);
    return make_double2(v.x, v.y);
}

struct Dot2Op // subdomain.okl:103-132 / cublasDdot
{
    static constexpr int NV = 1;
    const double *u, *v;
    __device__ void vec2(long long i, Acc<1> &a) const
    {
        double2 x = ld2(u, i), y = ld2(v, i);
        a.v[0] += x.x * y.x;
        a.v[0] += x.y * y.y;
    }
    __device__ void one(long long i, Acc<1> &a) const { a.v[0] += u[i] * v[i]; }
};

struct Dot3Op // domain.okl:109-138 (r*QQt_r*mask), :235-264 (u*v*mask), subdomain.okl:134-163 (u*v*w)
{
    static constexpr int NV = 1;
    const double *u, *v, *w;
    __device__ void vec2(long long i, Acc<1> &a) const
    {
        double2 x = ld2(u, i), y = ld2(v, i), z = ld2(w, i);
        a.v[0] += x.x * y.x * z.x;
        a.v[0] += x.y * y.y * z.y;
    }
    __device__ void one(long long i, Acc<1> &a) const { a.v[0] += u[i] * v[i] * w[i]; }
};

struct ProjOp // domain.okl:140-184
{
    static constexpr int NV = 2;
    const double *z, *r, *p, *q;
    __device__ void vec2(long long i, Acc<2> &a) const
    {
        double2 zz = ld2(z, i), rr = ld2(r, i), pp = ld2(p, i), qq = ld2(q, i);
        a.v[0] += zz.x * rr.x;
        a.v[0] += zz.y * rr.y;
        a.v[1] += pp.x * qq.x;
        a.v[1] += pp.y * qq.y;
    }
    __device__ void one(long long i, Acc<2> &a) const
    {
        a.v[0] += z[i] * r[i];
        a.v[1] += p[i] * q[i];
    }
};

struct ProjWOp // subdomain.okl:165-209
{
    static constexpr int NV = 2;
    const double *z, *r, *p, *q, *w;
    __device__ void vec2(long long i, Acc<2> &a) const
    {
        double2 zz = ld2(z, i), rr = ld2(r, i), pp = ld2(p, i), qq = ld2(q, i), ww = ld2(w, i);
        a.v[0] += zz.x * rr.x * ww.x;
        a.v[0] += zz.y * rr.y * ww.y;
        a.v[1] += pp.x * qq.x * ww.x;
        a.v[1] += pp.y * qq.y * ww.y;
    }
    __device__ void one(long long i, Acc<2> &a) const
    {
        a.v[0] += z[i] * r[i] * w[i];
        a.v[1] += p[i] * q[i] * w[i];
    }
};

struct FlexOp // domain.okl:195-224
{
    static constexpr int NV = 1;
    const double *r, *r1, *z;
    __device__ void vec2(long long i, Acc<1> &a) const
    {
        double2 a0 = ld2(r, i), a1 = ld2(r1, i), zz = ld2(z, i);
        a.v[0] += (a1.x - a0.x) * zz.x;
        a.v[0] += (a1.y - a0.y) * zz.y;
    }
    __device__ void one(long long i, Acc<1> &a) const { a.v[0] += (r1[i] - r[i]) * z[i]; }
};

// the flexible dot and, from the same three streams, the NEXT iteration's gamma = <z, r+> (domain.okl:140-184's first sum one
// iteration early: the projection kernel of that iteration is then left with <p, q> alone).  out = {gamma_next, theta}.
struct FlexGammaOp
{
    static constexpr int NV = 2;
    const double *r, *r1, *z;
    __device__ void vec2(long long i, Acc<2> &a) const
    {
        double2 a0 = ld2(r, i), a1 = ld2(r1, i), zz = ld2(z, i);
        a.v[0] += zz.x * a1.x;
        a.v[0] += zz.y * a1.y;
        a.v[1] += (a1.x - a0.x) * zz.x;
        a.v[1] += (a1.y - a0.y) * zz.y;
    }
    __device__ void one(long long i, Acc<2> &a) const
    {
        a.v[0] += z[i] * r1[i];
        a.v[1] += (r1[i] - r[i]) * z[i];
    }
};

struct FlexWOp // subdomain.okl:229-258
{
    static constexpr int NV = 1;
    const double *r, *r1, *z, *w;
    __device__ void vec2(long long i, Acc<1> &a) const
    {
        double2 a0 = ld2(r, i), a1 = ld2(r1, i), zz = ld2(z, i), ww = ld2(w, i);
        a.v[0] += (a1.x - a0.x) * zz.x * ww.x;
        a.v[0] += (a1.y - a0.y) * zz.y * ww.y;
    }
    __device__ void one(long long i, Acc<1> &a) const { a.v[0] += (r1[i] - r[i]) * z[i] * w[i]; }
};

inline bool al2(const void *a, const void *b) { return fdd_aligned16(a) && fdd_aligned16(b); }

// out[i] = sum a * b_i * w for i < M: `a` and `w` are read once for all M dots
// (the reference launches weighted_inner_product once per pair, subdomain.tpp:4389-4394)
template <int M>
struct MultiDotWOp
{
    static constexpr int NV = M;
    const double *a, *w;
    const double *b[M];
    const double *bs; // optional: b_k stands for bs[k] * b_k (a Krylov basis kept unnormalised, its 1/norm on the device)
    __device__ void vec2(long long i, Acc<M> &acc) const
    {
        const double2 aa = ld2(a, i), ww = w ? ld2(w, i) : make_double2(1.0, 1.0); // w == NULL: unit weights, not read (x * 1.0 is x)
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            double2 bb = (b[k] == a) ? aa : ld2(b[k], i); // <a, a>: one stream, not two
            if (bs)
            {
                const double sk = bs[k];
                bb.x = sk * bb.x;
                bb.y = sk * bb.y;
            }
            acc.v[k] += aa.x * bb.x * ww.x;
            acc.v[k] += aa.y * bb.y * ww.y;
        }
    }
    __device__ void one(long long i, Acc<M> &acc) const
    {
#pragma unroll
        for (int k = 0; k < M; k++) acc.v[k] += a[i] * (bs ? bs[k] * b[k][i] : b[k][i]) * (w ? w[i] : 1.0);
    }
};

template <int M>
int launch_multi_dot(double *out, double *ws, const double *a, const double *const *b, const double *b_scale_dev, const double *w, int n, void *stream)
{
    MultiDotWOp<M> op;
    op.a = a;
    op.w = w;
    op.bs = b_scale_dev;
    bool al = al2(a, w); // NULL counts as aligned
    for (int k = 0; k < M; k++)
    {
        op.b[k] = b[k];
        al = al && fdd_aligned16(b[k]);
    }
    return launch_reduce(op, out, ws, n, al, stream);
}

// Gram-Schmidt update and the norm of its result in one pass:
// y += sign * sum_k c[k] * x_k (the arithmetic of fdd_multi_axpy), out = sum y*y*w.
// The coefficients stay on the device (the output of fdd_multi_weighted_inner_product).
template <int M>
struct MultiAxpyNormOp
{
    static constexpr int NV = 1;
    double *dst;     // where the updated vector goes (y itself, or the next basis slot)
    const double *y;
    const double *w, *c;
    const double *x[M];
    const double *xs; // optional per-vector scales: x_k stands for xs[k] * x_k
    double sign;
    __device__ void vec2(long long i, Acc<1> &acc) const
    {
        double2 yy = ld2(y, i);
        const double2 ww = w ? ld2(w, i) : make_double2(1.0, 1.0);
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            const double ck = sign * c[k];
            double2 b = ld2(x[k], i);
            if (xs)
            {
                const double sk = xs[k];
                b.x = sk * b.x;
                b.y = sk * b.y;
            }
            yy.x = 1.0 * yy.x + ck * b.x;
            yy.y = 1.0 * yy.y + ck * b.y;
        }
        if (dst) reinterpret_cast<double2 *>(dst)[i] = yy; // dst == NULL: only the norm is wanted (the last Arnoldi step of a cycle)
        acc.v[0] += yy.x * yy.x * ww.x;
        acc.v[0] += yy.y * yy.y * ww.y;
    }
    __device__ void one(long long i, Acc<1> &acc) const
    {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < M; k++) v = 1.0 * v + (sign * c[k]) * (xs ? xs[k] * x[k][i] : x[k][i]);
        if (dst) dst[i] = v;
        acc.v[0] += v * v * (w ? w[i] : 1.0);
    }
};

template <int M>
int launch_multi_axpy_norm(double *out, double *ws, double *dst, const double *y, const double *c, double sign, const double *const *x, const double *x_scale_dev, const double *w, int n, void *stream)
{
    MultiAxpyNormOp<M> op;
    op.dst = dst;
    op.y = y;
    op.w = w;
    op.c = c;
    op.xs = x_scale_dev;
    op.sign = sign;
    bool al = al2(y, w) && fdd_aligned16(dst);
    for (int k = 0; k < M; k++)
    {
        op.x[k] = x[k];
        al = al && fdd_aligned16(x[k]);
    }
    return launch_reduce(op, out, ws, n, al, stream);
}

// ||Qt_w u||^2 in one pass: per assembled node s = (sum_j 1.0*u[col_j]) * w,
// accumulate s*s*w -- the gather of multiply_weight (csr_matrix.okl:35-48) and
// the weighted_inner_product (subdomain.okl:134-163) of Subdomain::residual_norm
// (subdomain.tpp:4491-4515) without the intermediate dof vector.
__global__ __launch_bounds__(kBlock) void gather_norm2_kernel(double *ws, const int *__restrict__ Qt_ptr, const int *__restrict__ Qt_col, const double *__restrict__ u, const double *__restrict__ w, int n)
{
    Acc<1> a;
    a.v[0] = 0.0;
    constexpr int NPT = 4;
    const int stride = gridDim.x * kBlock * NPT;
    for (int tile = blockIdx.x * kBlock * NPT; tile < n; tile += stride)
    {
        int j0[NPT], j1[NPT];
        double wn[NPT], s[NPT];
#pragma unroll
        for (int r = 0; r < NPT; r++)
        {
            const int node = tile + r * kBlock + threadIdx.x;
            const bool on = node < n;
            j0[r] = on ? Qt_ptr[node] : 0;
            j1[r] = on ? Qt_ptr[node + 1] : 0;
            wn[r] = on ? w[node] : 0.0;
        }
        fdd_multi_row_sum<NPT, 4, true>(Qt_col, nullptr, u, j0, j1, s);
#pragma unroll
        for (int r = 0; r < NPT; r++)
        {
            const double sw = s[r] * wn[r];
            a.v[0] += sw * sw * wn[r];
        }
    }
    block_reduce_store<1>(a, ws, FDD_REDUCE_MAX_BLOCKS);
}

// ---------------------------------------------------------------------------------------------------------------
// Single-precision preconditioner (the reference's PTYPE = Float, config.hpp:19-20): the same two reductions on
// float vectors.  Storage is float; products and sums are carried in double (the partial sums of a 10^7-term float
// dot would lose half their digits in float), the Gram-Schmidt update is rounded to float where it is stored.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 ld2f(const float *p, long long i) { return reinterpret_cast<const float2 *>(p)[i]; }

template <int M>
struct MultiDotOpF
{
    static constexpr int NV = M;
    const float *a;
    const float *b[M];
    const double *bs;
    __device__ void vec2(long long i, Acc<M> &acc) const
    {
        const float2 aa = ld2f(a, i);
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            const float2 bb = (b[k] == a) ? aa : ld2f(b[k], i);
            const double sk = bs ? bs[k] : 1.0;
            acc.v[k] += (double)aa.x * (sk * (double)bb.x);
            acc.v[k] += (double)aa.y * (sk * (double)bb.y);
        }
    }
    __device__ void one(long long i, Acc<M> &acc) const
    {
#pragma unroll
        for (int k = 0; k < M; k++) acc.v[k] += (double)a[i] * ((bs ? bs[k] : 1.0) * (double)b[k][i]);
    }
};

template <int M>
int launch_multi_dot_f32(double *out, double *ws, const float *a, const float *const *b, const double *b_scale_dev, int n, void *stream)
{
    MultiDotOpF<M> op;
    op.a = a;
    op.bs = b_scale_dev;
    bool al = (((uintptr_t)a) & 7) == 0;
    for (int k = 0; k < M; k++)
    {
        op.b[k] = b[k];
        al = al && ((((uintptr_t)b[k]) & 7) == 0);
    }
    return launch_reduce(op, out, ws, n, al, stream);
}

template <int M>
struct MultiAxpyNormOpF
{
    static constexpr int NV = 1;
    float *dst;
    const float *y;
    const float *x[M];
    const double *c;  // coefficients on the device
    const double *xs; // optional scales of the x_k
    double sign;
    __device__ void vec2(long long i, Acc<1> &acc) const
    {
        const float2 yy = ld2f(y, i);
        double v0 = yy.x, v1 = yy.y;
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            const float2 b = ld2f(x[k], i);
            const double ck = sign * c[k] * (xs ? xs[k] : 1.0);
            v0 += ck * (double)b.x;
            v1 += ck * (double)b.y;
        }
        const float2 r = make_float2((float)v0, (float)v1);
        if (dst) reinterpret_cast<float2 *>(dst)[i] = r;
        acc.v[0] += (double)r.x * (double)r.x;
        acc.v[0] += (double)r.y * (double)r.y;
    }
    __device__ void one(long long i, Acc<1> &acc) const
    {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < M; k++) v += sign * c[k] * (xs ? xs[k] : 1.0) * (double)x[k][i];
        const float r = (float)v;
        if (dst) dst[i] = r;
        acc.v[0] += (double)r * (double)r;
    }
};

template <int M>
int launch_multi_axpy_norm_f32(double *out, double *ws, float *dst, const float *y, const double *c, double sign, const float *const *x, const double *x_scale_dev, int n, void *stream)
{
    MultiAxpyNormOpF<M> op;
    op.dst = dst;
    op.y = y;
    op.c = c;
    op.xs = x_scale_dev;
    op.sign = sign;
    bool al = ((((uintptr_t)y) | ((uintptr_t)dst)) & 7) == 0;
    for (int k = 0; k < M; k++)
    {
        op.x[k] = x[k];
        al = al && ((((uintptr_t)x[k]) & 7) == 0);
    }
    return launch_reduce(op, out, ws, n, al, stream);
}

} // namespace

extern "C" {

size_t fdd_reduce_workspace_doubles(void) { return FDD_MULTI_MAX * (size_t)FDD_REDUCE_MAX_BLOCKS; }

int fdd_dom_residual_norm(double *out, double *ws, const double *r_k, const double *QQt_r_k, const double *dirichlet_mask, int num_points, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (r_k != nullptr && QQt_r_k != nullptr && dirichlet_mask != nullptr));
    return launch_reduce(Dot3Op{r_k, QQt_r_k, dirichlet_mask}, out, ws, num_points, al2(r_k, QQt_r_k) && fdd_aligned16(dirichlet_mask), stream);
}

int fdd_dom_inner_product(double *out, double *ws, const double *u_k, const double *v_k, const double *dirichlet_mask, int num_points, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (u_k != nullptr && v_k != nullptr && dirichlet_mask != nullptr));
    return launch_reduce(Dot3Op{u_k, v_k, dirichlet_mask}, out, ws, num_points, al2(u_k, v_k) && fdd_aligned16(dirichlet_mask), stream);
}

int fdd_dom_projection_inner_products(double *out2, double *ws, const double *z_k, const double *r_k, const double *p_k, const double *q_k, int num_points, void *stream)
{
    FDD_REQUIRE(out2 != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (z_k != nullptr && r_k != nullptr && p_k != nullptr && q_k != nullptr));
    return launch_reduce(ProjOp{z_k, r_k, p_k, q_k}, out2, ws, num_points, al2(z_k, r_k) && al2(p_k, q_k), stream);
}

int fdd_dom_inner_product_flexible(double *out, double *ws, const double *r_k, const double *r_kp1, const double *z_k, int num_points, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (r_k != nullptr && r_kp1 != nullptr && z_k != nullptr));
    return launch_reduce(FlexOp{r_k, r_kp1, z_k}, out, ws, num_points, al2(r_k, r_kp1) && fdd_aligned16(z_k), stream);
}

int fdd_dom_inner_product_flexible_gamma(double *out2, double *ws, const double *r_k, const double *r_kp1, const double *z_k, int num_points, void *stream)
{
    FDD_REQUIRE(out2 != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (r_k != nullptr && r_kp1 != nullptr && z_k != nullptr));
    return launch_reduce(FlexGammaOp{r_k, r_kp1, z_k}, out2, ws, num_points, al2(r_k, r_kp1) && fdd_aligned16(z_k), stream);
}

int fdd_sub_inner_product(double *out, double *ws, const double *u, const double *v, int num_values, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_values >= 0);
    FDD_REQUIRE(num_values == 0 || (u != nullptr && v != nullptr));
    return launch_reduce(Dot2Op{u, v}, out, ws, num_values, al2(u, v), stream);
}

int fdd_sub_weighted_inner_product(double *out, double *ws, const double *u, const double *v, const double *w, int num_values, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_values >= 0);
    FDD_REQUIRE(num_values == 0 || (u != nullptr && v != nullptr && w != nullptr));
    return launch_reduce(Dot3Op{u, v, w}, out, ws, num_values, al2(u, v) && fdd_aligned16(w), stream);
}

int fdd_sub_projection_inner_products(double *out2, double *ws, const double *z_k, const double *r_k, const double *p_k, const double *q_k, const double *weight, int num_values, void *stream)
{
    FDD_REQUIRE(out2 != nullptr && ws != nullptr && num_values >= 0);
    FDD_REQUIRE(num_values == 0 || (z_k != nullptr && r_k != nullptr && p_k != nullptr && q_k != nullptr && weight != nullptr));
    return launch_reduce(ProjWOp{z_k, r_k, p_k, q_k, weight}, out2, ws, num_values, al2(z_k, r_k) && al2(p_k, q_k) && fdd_aligned16(weight), stream);
}

int fdd_sub_search_update_inner_product(double *out, double *ws, const double *r_k, const double *r_kp1, const double *z_k, const double *weight, int num_points, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_points >= 0);
    FDD_REQUIRE(num_points == 0 || (r_k != nullptr && r_kp1 != nullptr && z_k != nullptr && weight != nullptr));
    return launch_reduce(FlexWOp{r_k, r_kp1, z_k, weight}, out, ws, num_points, al2(r_k, r_kp1) && al2(z_k, weight), stream);
}

int fdd_amg_dot(double *out, double *ws, const double *x, const double *y, int size, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && size >= 0);
    FDD_REQUIRE(size == 0 || (x != nullptr && y != nullptr));
    return launch_reduce(Dot2Op{x, y}, out, ws, size, al2(x, y), stream);
}

int fdd_multi_weighted_inner_product(double *out, double *ws, const double *a, const double *const *b, int m, const double *w, int n, void *stream)
{
    return fdd_multi_weighted_inner_product_scaled(out, ws, a, b, nullptr, m, w, n, stream);
}

int fdd_multi_weighted_inner_product_scaled(double *out, double *ws, const double *a, const double *const *b, const double *b_scale_dev, int m, const double *w, int n, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && n >= 0 && m >= 1 && m <= FDD_MULTI_MAX && b != nullptr);
    FDD_REQUIRE(n == 0 || a != nullptr); // w == NULL: unit weights
    for (int k = 0; k < m; k++) FDD_REQUIRE(n == 0 || b[k] != nullptr);
    switch (m)
    {
    case 1: return launch_multi_dot<1>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 2: return launch_multi_dot<2>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 3: return launch_multi_dot<3>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 4: return launch_multi_dot<4>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 5: return launch_multi_dot<5>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 6: return launch_multi_dot<6>(out, ws, a, b, b_scale_dev, w, n, stream);
    case 7: return launch_multi_dot<7>(out, ws, a, b, b_scale_dev, w, n, stream);
    default: return launch_multi_dot<8>(out, ws, a, b, b_scale_dev, w, n, stream);
    }
}

int fdd_multi_axpy_norm2_dev(double *out, double *ws, double *y, const double *coeffs_dev, double sign, const double *const *x, int m, const double *w, int n, void *stream)
{
    return fdd_multi_axpy_norm2_scaled_dev(out, ws, y, y, coeffs_dev, sign, x, nullptr, m, w, n, stream);
}

int fdd_multi_axpy_norm2_scaled_dev(double *out, double *ws, double *dst, const double *y, const double *coeffs_dev, double sign, const double *const *x, const double *x_scale_dev, int m, const double *w, int n, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && n >= 0 && m >= 1 && m <= FDD_MULTI_MAX && x != nullptr && coeffs_dev != nullptr);
    FDD_REQUIRE(n == 0 || y != nullptr); // w == NULL: unit weights; dst == NULL: the updated vector is not stored, only its norm formed
    for (int k = 0; k < m; k++) FDD_REQUIRE(n == 0 || x[k] != nullptr);
    switch (m)
    {
    case 1: return launch_multi_axpy_norm<1>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 2: return launch_multi_axpy_norm<2>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 3: return launch_multi_axpy_norm<3>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 4: return launch_multi_axpy_norm<4>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 5: return launch_multi_axpy_norm<5>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 6: return launch_multi_axpy_norm<6>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    case 7: return launch_multi_axpy_norm<7>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    default: return launch_multi_axpy_norm<8>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, w, n, stream);
    }
}

int fdd_gather_weighted_norm2(double *out, double *ws, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, int num_nodes, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && num_nodes >= 0);
    hipStream_t s = fdd_stream(stream);
    if (num_nodes == 0) return (int)hipMemsetAsync(out, 0, sizeof(double), s);
    FDD_REQUIRE(Qt_ptr != nullptr && Qt_col != nullptr && u != nullptr && node_weight != nullptr);
    const int grid = fdd_stream_grid((num_nodes + 3) / 4, kBlock); // 4 nodes per lane per pass
    hipLaunchKernelGGL(gather_norm2_kernel, dim3(grid), dim3(kBlock), 0, s, ws, Qt_ptr, Qt_col, u, node_weight, num_nodes);
    FDD_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel<1>, dim3(1), dim3(kBlock), 0, s, out, ws, grid);
    FDD_LAUNCH_CHECK();
    return 0;
}

#define FDD_M_SWITCH(CALL)                                                          \
    switch (m)                                                                      \
    {                                                                               \
    case 1: return CALL(1);                                                         \
    case 2: return CALL(2);                                                         \
    case 3: return CALL(3);                                                         \
    case 4: return CALL(4);                                                         \
    case 5: return CALL(5);                                                         \
    case 6: return CALL(6);                                                         \
    case 7: return CALL(7);                                                         \
    case 8: return CALL(8);                                                         \
    default: fdd_set_error("at most 8 vectors per multi-vector reduction"); return FDD_ERR_INVALID_ARGUMENT; \
    }

int fdd_multi_inner_product_scaled_f32(double *out, double *ws, const float *a, const float *const *b, const double *b_scale_dev, int m, int n, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && m >= 1 && n >= 0);
    FDD_REQUIRE(n == 0 || (a != nullptr && b != nullptr));
#define FDD_CALL_DOT(M_) launch_multi_dot_f32<M_>(out, ws, a, b, b_scale_dev, n, stream)
    FDD_M_SWITCH(FDD_CALL_DOT)
#undef FDD_CALL_DOT
}

int fdd_multi_axpy_norm2_scaled_dev_f32(double *out, double *ws, float *dst, const float *y, const double *coeffs_dev, double sign, const float *const *x, const double *x_scale_dev, int m, int n, void *stream)
{
    FDD_REQUIRE(out != nullptr && ws != nullptr && m >= 1 && n >= 0);
    FDD_REQUIRE(n == 0 || (y != nullptr && x != nullptr && coeffs_dev != nullptr)); // dst == NULL: norm only
#define FDD_CALL_AN(M_) launch_multi_axpy_norm_f32<M_>(out, ws, dst, y, coeffs_dev, sign, x, x_scale_dev, n, stream)
    FDD_M_SWITCH(FDD_CALL_AN)
#undef FDD_CALL_AN
}
#undef FDD_M_SWITCH

} // extern "C"
