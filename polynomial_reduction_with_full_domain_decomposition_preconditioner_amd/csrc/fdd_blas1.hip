// Element-wise streaming kernels (HBM-bound): math.okl, the fused vector
// updates of domain.okl / subdomain.okl, the precision-cast copies and the
// element-wise AMG smoother kernels of AMG/kernels.cu.
//
// One template drives them all: every lane moves 16 B per access (double2),
// the grid is capped at 2048 workgroups of 256 lanes and strides over the
// vector, so a 134 MB vector is 16 passes of fully coalesced 1 KiB
// wave-accesses.  Pointers that are not 16-B aligned (occa::memory::slice
// offsets, math.okl's `offset`) take the scalar 8-B path.
//
// Compiled with -ffp-contract=off: a*x + b*y is two multiplies and one add in
// source order, bit-identical to the OCCA-Serial arithmetic.
#include "fdd_common.h"

namespace
{

constexpr int kBlock = 256;

template <typename Op>
__global__ __launch_bounds__(kBlock) void ew_vec2_kernel(Op op, long long n2, long long n)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) op.vec2(i);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) op.one(n - 1);
}

template <typename Op>
__global__ __launch_bounds__(kBlock) void ew_scalar_kernel(Op op, long long n)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) op.one(i);
}

// Element-wise kernels have no per-workgroup epilogue: at C2 sizes a grid of one pair of doubles per lane (21 k workgroups)
// streams 5 % faster than 2048 workgroups walking the vector (6.7 against 6.3 TB/s on the axpy forms); the reductions,
// whose workgroups end in a fold, are best at the 2048 of FDD_REDUCE_MAX_BLOCKS (16384: 4.0-5.1 instead of 5.9 TB/s).
constexpr int kEwMaxBlocks = 32768;

template <typename Op>
int launch_ew(const Op &op, long long n, bool aligned, void *stream)
{
    if (n <= 0) return 0;
    if (aligned && n >= 2)
    {
        long long n2 = n / 2;
        int grid = fdd_stream_grid(n2, kBlock, kEwMaxBlocks);
        hipLaunchKernelGGL(ew_vec2_kernel<Op>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), op, n2, n);
    }
    else
    {
        int grid = fdd_stream_grid(n, kBlock, kEwMaxBlocks);
        hipLaunchKernelGGL(ew_scalar_kernel<Op>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), op, n);
    }
    FDD_LAUNCH_CHECK();
    return 0;
}

// 16-byte streaming loads are non-temporal: these kernels read every byte once per launch
// and the operands are far larger than the caches (measured +10-15 % on the multi-stream ops)
typedef double fdd_v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2(const double *p, long long i)
{
    const fdd_v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const fdd_v2f64 *>(p) + i);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st2(double *p, long long i, double2 v) { reinterpret_cast<double2 *>(p)[i] = v; } // default policy: the next kernel usually reads it

// ---------------------------------------------------------------- math.okl
struct SetOp // math.okl:5-11
{
    double *u;
    double alpha;
    __device__ void vec2(long long i) const { st2(u, i, make_double2(alpha, alpha)); }
    __device__ void one(long long i) const { u[i] = alpha; }
};

struct InvertOp // math.okl:13-19
{
    double *u;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(u, i);
        st2(u, i, make_double2(1.0 / a.x, 1.0 / a.y));
    }
    __device__ void one(long long i) const { u[i] = 1.0 / u[i]; }
};

struct AxpbyOp // math.okl:21-27
{
    double *uv;
    const double *u;
    const double *v;
    double alpha, beta;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(u, i), b = ld2(v, i);
        st2(uv, i, make_double2(alpha * a.x + beta * b.x, alpha * a.y + beta * b.y));
    }
    __device__ void one(long long i) const { uv[i] = alpha * u[i] + beta * v[i]; }
};

// x = (1 / sqrt(*s2)) * u with the squared norm still on the device
// (host sequence: alpha = sqrt(s2); vector_scaling(x, 1.0 / alpha, u), subdomain.tpp:4457)
struct ScaleRsqrtDevOp
{
    double *au;
    const double *u;
    const double *s2;
    __device__ void vec2(long long i) const
    {
        const double alpha = 1.0 / sqrt(*s2);
        double2 a = ld2(u, i);
        st2(au, i, make_double2(alpha * a.x, alpha * a.y));
    }
    __device__ void one(long long i) const { au[i] = (1.0 / sqrt(*s2)) * u[i]; }
};

// au = (*scale) * u: math.okl:29-35 with the factor read from device memory
struct ScaleDevOp
{
    double *au;
    const double *u;
    const double *scale;
    __device__ void vec2(long long i) const
    {
        const double alpha = *scale;
        double2 a = ld2(u, i);
        st2(au, i, make_double2(alpha * a.x, alpha * a.y));
    }
    __device__ void one(long long i) const { au[i] = (*scale) * u[i]; }
};

// z = d .* ((*scale) * u): the point-Jacobi preconditioner of the inner solve applied to a Krylov vector kept
// unnormalised (vector_scaling, math.okl:29-35, then AMG/kernels.cu:64-73 vector_multiplication, in that order)
struct DiagScaleDevOp
{
    double *z;
    const double *d;
    const double *u;
    const double *scale; // may be null: z = d .* u
    __device__ void vec2(long long i) const
    {
        const double alpha = scale ? *scale : 1.0;
        const double2 a = ld2(u, i), dd = ld2(d, i);
        st2(z, i, scale ? make_double2(dd.x * (alpha * a.x), dd.y * (alpha * a.y)) : make_double2(dd.x * a.x, dd.y * a.y));
    }
    __device__ void one(long long i) const { z[i] = scale ? d[i] * ((*scale) * u[i]) : d[i] * u[i]; }
};

// out = x + (*num / *den) * y: the p = z + beta*p half of residual_and_search_update (domain.okl:218-233)
// when r = r+ is a buffer swap instead of a copy
struct XpbyRatioDevOp
{
    double *out;
    const double *x, *y, *num, *den;
    __device__ void vec2(long long i) const
    {
        const double beta = *num / *den;
        const double2 a = ld2(x, i), b = ld2(y, i);
        st2(out, i, make_double2(a.x + beta * b.x, a.y + beta * b.y));
    }
    __device__ void one(long long i) const { out[i] = x[i] + (*num / *den) * y[i]; }
};

// out = x - (*num / *den) * y: the r+ = r - alpha*q half of solution_and_residual_update (domain.okl:186-193) on
// a vector the node-space solve keeps beside its node vectors (the point-space residual the degree tree is fed with)
struct XmayRatioDevOp
{
    double *out;
    const double *x, *y, *num, *den;
    __device__ void vec2(long long i) const
    {
        const double alpha = *num / *den;
        const double2 a = ld2(x, i), b = ld2(y, i);
        st2(out, i, make_double2(a.x - alpha * b.x, a.y - alpha * b.y));
    }
    __device__ void one(long long i) const { out[i] = x[i] - (*num / *den) * y[i]; }
};

} // namespace
__global__ void fdd_sqrt_sum_kernel(double *out, const double *parts, int nparts)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int k = 0; k < nparts; k++) s += parts[k];
    *out = sqrt(s);
}
namespace
{
struct ScaleOp // math.okl:29-35
{
    double *au;
    const double *u;
    double alpha;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(u, i);
        st2(au, i, make_double2(alpha * a.x, alpha * a.y));
    }
    __device__ void one(long long i) const { au[i] = alpha * u[i]; }
};

// ------------------------------------------------- domain.okl / subdomain.okl
struct InitOp // domain.okl:100-107, subdomain.okl:211-218
{
    double *u_k;
    double *r_k;
    const double *f;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(f, i);
        st2(u_k, i, make_double2(0.0, 0.0));
        st2(r_k, i, a);
    }
    __device__ void one(long long i) const
    {
        u_k[i] = 0.0;
        r_k[i] = f[i];
    }
};

// alpha either by value or as num/den read from device memory
struct ScalarByValue
{
    double v;
    __device__ double get() const { return v; }
};
struct ScalarRatioDev
{
    const double *num;
    const double *den;
    __device__ double get() const { return num[0] / den[0]; }
};

template <typename S>
struct UpdateUROp // domain.okl:186-193, subdomain.okl:220-227
{
    double *u_k;
    double *r_kp1;
    const double *r_k;
    const double *p_k;
    const double *q_k;
    S alpha;
    __device__ void vec2(long long i) const
    {
        const double a = alpha.get();
        double2 u = ld2(u_k, i), r = ld2(r_k, i), p = ld2(p_k, i), q = ld2(q_k, i);
        u.x += a * p.x;
        u.y += a * p.y;
        st2(u_k, i, u);
        st2(r_kp1, i, make_double2(r.x - a * q.x, r.y - a * q.y));
    }
    __device__ void one(long long i) const
    {
        const double a = alpha.get();
        u_k[i] += a * p_k[i];
        r_kp1[i] = r_k[i] - a * q_k[i];
    }
};

template <typename S>
struct UpdatePROp // domain.okl:226-233, subdomain.okl:259-266
{
    double *p_k;
    double *r_k;
    const double *z_k;
    const double *r_kp1;
    S beta;
    __device__ void vec2(long long i) const
    {
        const double b = beta.get();
        double2 z = ld2(z_k, i), p = ld2(p_k, i), r = ld2(r_kp1, i);
        st2(p_k, i, make_double2(z.x + b * p.x, z.y + b * p.y));
        st2(r_k, i, r);
    }
    __device__ void one(long long i) const
    {
        const double b = beta.get();
        p_k[i] = z_k[i] + b * p_k[i];
        r_k[i] = r_kp1[i];
    }
};

struct CopyOp // subdomain.okl:268-282 with DType = EType = double
{
    double *u;
    const double *v;
    __device__ void vec2(long long i) const { st2(u, i, ld2(v, i)); }
    __device__ void one(long long i) const { u[i] = v[i]; }
};

struct CopyF32F64Op // subdomain.okl:268-274, DType = float
{
    float *u;
    const double *v;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(v, i);
        reinterpret_cast<float2 *>(u)[i] = make_float2((float)a.x, (float)a.y);
    }
    __device__ void one(long long i) const { u[i] = (float)v[i]; }
};

struct CopyF64F32Op // subdomain.okl:276-282, DType = float
{
    double *u;
    const float *v;
    __device__ void vec2(long long i) const
    {
        float2 a = reinterpret_cast<const float2 *>(v)[i];
        st2(u, i, make_double2((double)a.x, (double)a.y));
    }
    __device__ void one(long long i) const { u[i] = (double)v[i]; }
};

// q <- (...((1.0*q + c_0 v_0) + c_1 v_1)...) + c_{M-1} v_{M-1}: the M successive
// vector_vector_addition(q, 1.0, q, c_i, v_i) launches of a Gram-Schmidt
// sweep (domain.tpp:817-822, subdomain.tpp:4396-4401) or of the solution
// update (domain.tpp:902-907) in one pass: q is read and written once, each
// element goes through the same operations in the same order.
template <int M>
struct MultiAxpyOp
{
    double *q;
    const double *v[M];
    double c[M];
    __device__ void vec2(long long i) const
    {
        double2 x = ld2(q, i);
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            const double2 b = ld2(v[k], i);
            x.x = 1.0 * x.x + c[k] * b.x;
            x.y = 1.0 * x.y + c[k] * b.y;
        }
        st2(q, i, x);
    }
    __device__ void one(long long i) const
    {
        double x = q[i];
#pragma unroll
        for (int k = 0; k < M; k++) x = 1.0 * x + c[k] * v[k][i];
        q[i] = x;
    }
};

template <int M>
int launch_multi_axpy(double *q, const double *coeffs, const double *const *v, int n, void *stream)
{
    MultiAxpyOp<M> op;
    op.q = q;
    bool al = fdd_aligned16(q);
    for (int k = 0; k < M; k++)
    {
        op.v[k] = v[k];
        op.c[k] = coeffs[k];
        al = al && fdd_aligned16(v[k]);
    }
    return launch_ew(op, n, al, stream);
}

// the same with the coefficients read from device memory (the output of fdd_gmres_finish_dev)
template <int M>
struct MultiAxpyDevOp
{
    double *q;
    const double *v[M];
    const double *c;
    const double *vs; // optional per-vector scales: v_k stands for vs[k] * v_k
    bool from_zero;   // q is known to be 0 (it was just cleared): it is not read, 1.0*0 + c*v is c*v all the same
    const double *last; // optional: only vectors 0..(int)*last enter (a count that never left the device)
    __device__ void vec2(long long i) const
    {
        double2 x = from_zero ? make_double2(0.0, 0.0) : ld2(q, i);
        const int kmax = last ? (int)*last : M - 1;
#pragma unroll
        for (int k = 0; k < M; k++)
        {
            if (k > kmax) break;
            const double ck = c[k];
            double2 b = ld2(v[k], i);
            if (vs)
            {
                const double sk = vs[k];
                b.x = sk * b.x;
                b.y = sk * b.y;
            }
            x.x = 1.0 * x.x + ck * b.x;
            x.y = 1.0 * x.y + ck * b.y;
        }
        st2(q, i, x);
    }
    __device__ void one(long long i) const
    {
        double x = from_zero ? 0.0 : q[i];
        const int kmax = last ? (int)*last : M - 1;
#pragma unroll
        for (int k = 0; k < M; k++)
            if (k <= kmax) x = 1.0 * x + c[k] * (vs ? vs[k] * v[k][i] : v[k][i]);
        q[i] = x;
    }
};

template <int M>
int launch_multi_axpy_dev(double *q, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, bool from_zero, const double *last_dev, int n, void *stream)
{
    MultiAxpyDevOp<M> op;
    op.q = q;
    op.c = coeffs_dev;
    op.vs = v_scale_dev;
    op.from_zero = from_zero;
    op.last = last_dev;
    bool al = fdd_aligned16(q);
    for (int k = 0; k < M; k++)
    {
        op.v[k] = v[k];
        al = al && fdd_aligned16(v[k]);
    }
    return launch_ew(op, n, al, stream);
}

// ------------------------------------------------------------ AMG/kernels.cu
struct ScaledResidualOp // AMG/kernels.cu:25-41
{
    double *Sr;
    double *w;
    const double *f_m_Au;
    const double *S;
    double alpha;
    __device__ void vec2(long long i) const
    {
        double2 s = ld2(S, i), r = ld2(f_m_Au, i);
        double2 sr = make_double2(s.x * r.x, s.y * r.y);
        st2(Sr, i, sr);
        st2(w, i, make_double2(alpha * sr.x, alpha * sr.y));
    }
    __device__ void one(long long i) const
    {
        Sr[i] = S[i] * f_m_Au[i];
        w[i] = alpha * Sr[i];
    }
};

struct SmoothStartOp // scaled_residual from u = 0 (f - A u is f) followed by vector_multiplication: Sr = S*f, work = D*(alpha*Sr)
{
    double *work;
    double *Sr;
    const double *f;
    const double *S;
    double alpha;
    __device__ void vec2(long long i) const
    {
        double2 s = ld2(S, i), r = ld2(f, i);
        double2 sr = make_double2(s.x * r.x, s.y * r.y);
        st2(Sr, i, sr);
        st2(work, i, make_double2(s.x * (alpha * sr.x), s.y * (alpha * sr.y)));
    }
    __device__ void one(long long i) const
    {
        const double sr = S[i] * f[i];
        Sr[i] = sr;
        work[i] = S[i] * (alpha * sr);
    }
};

// ---- Float = float (AMG/config.hpp:4): the two element-wise kernels the fused f32 V-cycle launches ----
struct SetF32Op // AMG/kernels.cu:11-23
{
    float *data;
    float value;
    __device__ void vec2(long long i) const { reinterpret_cast<float2 *>(data)[i] = make_float2(value, value); }
    __device__ void one(long long i) const { data[i] = value; }
};

struct SmoothStartF32Op // SmoothStartOp in f32
{
    float *work;
    float *Sr;
    const float *f;
    const float *S;
    float alpha;
    __device__ void vec2(long long i) const
    {
        const float2 s = reinterpret_cast<const float2 *>(S)[i], r = reinterpret_cast<const float2 *>(f)[i];
        const float2 sr = make_float2(s.x * r.x, s.y * r.y);
        reinterpret_cast<float2 *>(Sr)[i] = sr;
        reinterpret_cast<float2 *>(work)[i] = make_float2(s.x * (alpha * sr.x), s.y * (alpha * sr.y));
    }
    __device__ void one(long long i) const
    {
        const float sr = S[i] * f[i];
        Sr[i] = sr;
        work[i] = S[i] * (alpha * sr);
    }
};

struct PolyEvalOp // AMG/kernels.cu:43-59
{
    double *w;
    double *v;
    const double *r;
    const double *D_val;
    double alpha;
    __device__ void vec2(long long i) const
    {
        double2 vv = ld2(v, i), d = ld2(D_val, i), rr = ld2(r, i);
        vv.x *= d.x;
        vv.y *= d.y;
        st2(v, i, vv);
        st2(w, i, make_double2(alpha * rr.x + vv.x, alpha * rr.y + vv.y));
    }
    __device__ void one(long long i) const
    {
        v[i] *= D_val[i];
        w[i] = alpha * r[i] + v[i];
    }
};

struct UpdateFieldOp // AMG/kernels.cu:61-76
{
    double *u;
    const double *w;
    const double *D_val;
    __device__ void vec2(long long i) const
    {
        double2 uu = ld2(u, i), ww = ld2(w, i), d = ld2(D_val, i);
        uu.x += d.x * ww.x;
        uu.y += d.y * ww.y;
        st2(u, i, uu);
    }
    __device__ void one(long long i) const { u[i] += D_val[i] * w[i]; }
};

struct VMulOp // AMG/kernels.cu:79-94
{
    double *uv;
    const double *u;
    const double *v;
    __device__ void vec2(long long i) const
    {
        double2 a = ld2(u, i), b = ld2(v, i);
        st2(uv, i, make_double2(a.x * b.x, a.y * b.y));
    }
    __device__ void one(long long i) const { uv[i] = u[i] * v[i]; }
};

static int initialize_arrays(double *u_k, double *r_k, const double *f, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(u_k != nullptr && r_k != nullptr && f != nullptr);
    return launch_ew(InitOp{u_k, r_k, f}, n, fdd_aligned16(u_k) && fdd_aligned16(r_k) && fdd_aligned16(f), stream);
}

template <typename S>
static int update_ur(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, S alpha, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(u_k != nullptr && r_kp1 != nullptr && r_k != nullptr && p_k != nullptr && q_k != nullptr);
    bool al = fdd_aligned16(u_k) && fdd_aligned16(r_kp1) && fdd_aligned16(r_k) && fdd_aligned16(p_k) && fdd_aligned16(q_k);
    return launch_ew(UpdateUROp<S>{u_k, r_kp1, r_k, p_k, q_k, alpha}, n, al, stream);
}

template <typename S>
static int update_pr(double *p_k, double *r_k, const double *z_k, const double *r_kp1, S beta, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(p_k != nullptr && r_k != nullptr && z_k != nullptr && r_kp1 != nullptr);
    bool al = fdd_aligned16(p_k) && fdd_aligned16(r_k) && fdd_aligned16(z_k) && fdd_aligned16(r_kp1);
    return launch_ew(UpdatePROp<S>{p_k, r_k, z_k, r_kp1, beta}, n, al, stream);
}

} // namespace

extern "C" {

int fdd_set_to_value(double *u, double alpha, int n, int offset, void *stream)
{
    FDD_REQUIRE(n >= 0 && offset >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(u != nullptr);
    double *p = u + offset;
    return launch_ew(SetOp{p, alpha}, n, fdd_aligned16(p), stream);
}

int fdd_invert_vector_elements(double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(u != nullptr);
    return launch_ew(InvertOp{u}, n, fdd_aligned16(u), stream);
}

int fdd_vector_vector_addition(double *uv, double alpha, const double *u, double beta, const double *v, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(uv != nullptr && u != nullptr && v != nullptr);
    return launch_ew(AxpbyOp{uv, u, v, alpha, beta}, n, fdd_aligned16(uv) && fdd_aligned16(u) && fdd_aligned16(v), stream);
}

int fdd_vector_scaling(double *au, double alpha, const double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(au != nullptr && u != nullptr);
    return launch_ew(ScaleOp{au, u, alpha}, n, fdd_aligned16(au) && fdd_aligned16(u), stream);
}

int fdd_sqrt_sum_dev(double *out, const double *parts_dev, int nparts, void *stream)
{
    FDD_REQUIRE(out != nullptr && parts_dev != nullptr && nparts >= 1);
    hipLaunchKernelGGL(fdd_sqrt_sum_kernel, dim3(1), dim3(64), 0, fdd_stream(stream), out, parts_dev, nparts);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_xpby_ratio_dev(double *out, const double *x, const double *num_dev, const double *den_dev, const double *y, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(out != nullptr && x != nullptr && y != nullptr && num_dev != nullptr && den_dev != nullptr);
    return launch_ew(XpbyRatioDevOp{out, x, y, num_dev, den_dev}, n, fdd_aligned16(out) && fdd_aligned16(x) && fdd_aligned16(y), stream);
}

int fdd_xmay_ratio_dev(double *out, const double *x, const double *num_dev, const double *den_dev, const double *y, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(out != nullptr && x != nullptr && y != nullptr && num_dev != nullptr && den_dev != nullptr);
    return launch_ew(XmayRatioDevOp{out, x, y, num_dev, den_dev}, n, fdd_aligned16(out) && fdd_aligned16(x) && fdd_aligned16(y), stream);
}

int fdd_vector_scaling_dev(double *au, const double *scale_dev, const double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(au != nullptr && u != nullptr && scale_dev != nullptr);
    return launch_ew(ScaleDevOp{au, u, scale_dev}, n, fdd_aligned16(au) && fdd_aligned16(u), stream);
}

int fdd_vector_diagonal_scaling_dev(double *z, const double *d, const double *scale_dev, const double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(z != nullptr && u != nullptr && d != nullptr);
    return launch_ew(DiagScaleDevOp{z, d, u, scale_dev}, n, fdd_aligned16(z) && fdd_aligned16(u) && fdd_aligned16(d), stream);
}

int fdd_vector_scaling_rsqrt_dev(double *au, const double *norm2_dev, const double *u, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(au != nullptr && u != nullptr && norm2_dev != nullptr);
    return launch_ew(ScaleRsqrtDevOp{au, u, norm2_dev}, n, fdd_aligned16(au) && fdd_aligned16(u), stream);
}

int fdd_dom_initialize_arrays(double *u_k, double *r_k, const double *f, int num_points, void *stream)
{
    return initialize_arrays(u_k, r_k, f, num_points, stream);
}

int fdd_sub_initialize_arrays(double *u_k, double *r_k, const double *f, int num_values, void *stream)
{
    return initialize_arrays(u_k, r_k, f, num_values, stream);
}

int fdd_dom_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_points, void *stream)
{
    return update_ur(u_k, r_kp1, r_k, p_k, q_k, ScalarByValue{alpha_k}, num_points, stream);
}

int fdd_sub_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_values, void *stream)
{
    return update_ur(u_k, r_kp1, r_k, p_k, q_k, ScalarByValue{alpha_k}, num_values, stream);
}

int fdd_dom_solution_and_residual_update_dev(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, const double *alpha_num, const double *alpha_den, int num_points, void *stream)
{
    FDD_REQUIRE(alpha_num != nullptr && alpha_den != nullptr);
    return update_ur(u_k, r_kp1, r_k, p_k, q_k, ScalarRatioDev{alpha_num, alpha_den}, num_points, stream);
}

int fdd_dom_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_points, void *stream)
{
    return update_pr(p_k, r_k, z_k, r_kp1, ScalarByValue{beta_k}, num_points, stream);
}

int fdd_sub_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_values, void *stream)
{
    return update_pr(p_k, r_k, z_k, r_kp1, ScalarByValue{beta_k}, num_values, stream);
}

int fdd_dom_residual_and_search_update_dev(double *p_k, double *r_k, const double *z_k, const double *r_kp1, const double *beta_num, const double *beta_den, int num_points, void *stream)
{
    FDD_REQUIRE(beta_num != nullptr && beta_den != nullptr);
    return update_pr(p_k, r_k, z_k, r_kp1, ScalarRatioDev{beta_num, beta_den}, num_points, stream);
}

int fdd_sub_copy_f64_f64(double *u, const double *v, int num_points, void *stream)
{
    FDD_REQUIRE(num_points >= 0);
    if (num_points == 0) return 0;
    FDD_REQUIRE(u != nullptr && v != nullptr);
    return launch_ew(CopyOp{u, v}, num_points, fdd_aligned16(u) && fdd_aligned16(v), stream);
}

int fdd_sub_copy_f32_f64(float *u, const double *v, int num_points, void *stream)
{
    FDD_REQUIRE(num_points >= 0);
    if (num_points == 0) return 0;
    FDD_REQUIRE(u != nullptr && v != nullptr);
    bool al = fdd_aligned16(v) && ((reinterpret_cast<uintptr_t>(u) & 7u) == 0);
    return launch_ew(CopyF32F64Op{u, v}, num_points, al, stream);
}

int fdd_sub_copy_f64_f32(double *u, const float *v, int num_points, void *stream)
{
    FDD_REQUIRE(num_points >= 0);
    if (num_points == 0) return 0;
    FDD_REQUIRE(u != nullptr && v != nullptr);
    bool al = fdd_aligned16(u) && ((reinterpret_cast<uintptr_t>(v) & 7u) == 0);
    return launch_ew(CopyF64F32Op{u, v}, num_points, al, stream);
}

int fdd_multi_axpy(double *q, const double *coeffs, const double *const *v, int m, int n, void *stream)
{
    FDD_REQUIRE(n >= 0 && m >= 1 && m <= FDD_MULTI_MAX);
    if (n == 0) return 0;
    FDD_REQUIRE(q != nullptr && coeffs != nullptr && v != nullptr);
    for (int k = 0; k < m; k++) FDD_REQUIRE(v[k] != nullptr && v[k] != q);
    switch (m)
    {
    case 1: return launch_multi_axpy<1>(q, coeffs, v, n, stream);
    case 2: return launch_multi_axpy<2>(q, coeffs, v, n, stream);
    case 3: return launch_multi_axpy<3>(q, coeffs, v, n, stream);
    case 4: return launch_multi_axpy<4>(q, coeffs, v, n, stream);
    case 5: return launch_multi_axpy<5>(q, coeffs, v, n, stream);
    case 6: return launch_multi_axpy<6>(q, coeffs, v, n, stream);
    case 7: return launch_multi_axpy<7>(q, coeffs, v, n, stream);
    default: return launch_multi_axpy<8>(q, coeffs, v, n, stream);
    }
}

int fdd_multi_axpy_dev(double *q, const double *coeffs_dev, const double *const *v, int m, int n, void *stream)
{
    return fdd_multi_axpy_scaled_dev(q, coeffs_dev, v, nullptr, m, n, stream);
}

int fdd_multi_axpy_scaled_dev(double *q, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, int m, int n, void *stream)
{
    return fdd_multi_lincomb_scaled_dev(q, 0, coeffs_dev, v, v_scale_dev, m, n, stream);
}

int fdd_multi_lincomb_scaled_dev(double *q, int q_is_zero, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, int m, int n, void *stream)
{
    return fdd_multi_lincomb_limited_dev(q, q_is_zero, coeffs_dev, v, v_scale_dev, nullptr, m, n, stream);
}

int fdd_multi_lincomb_limited_dev(double *q, int q_is_zero, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, const double *last_dev, int m, int n, void *stream)
{
    const bool from_zero = q_is_zero != 0;
    FDD_REQUIRE(n >= 0 && m >= 1 && m <= FDD_MULTI_MAX);
    if (n == 0) return 0;
    FDD_REQUIRE(q != nullptr && coeffs_dev != nullptr && v != nullptr);
    for (int k = 0; k < m; k++) FDD_REQUIRE(v[k] != nullptr && v[k] != q);
    switch (m)
    {
    case 1: return launch_multi_axpy_dev<1>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 2: return launch_multi_axpy_dev<2>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 3: return launch_multi_axpy_dev<3>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 4: return launch_multi_axpy_dev<4>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 5: return launch_multi_axpy_dev<5>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 6: return launch_multi_axpy_dev<6>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    case 7: return launch_multi_axpy_dev<7>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    default: return launch_multi_axpy_dev<8>(q, coeffs_dev, v, v_scale_dev, from_zero, last_dev, n, stream);
    }
}

int fdd_amg_vector_set_to_value(double *data, double value, int size, void *stream)
{
    return fdd_set_to_value(data, value, size, 0, stream);
}

int fdd_amg_main_scaled_residual(double *Sr, double *w, const double *f_m_Au, const double *S, double alpha, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(Sr != nullptr && w != nullptr && f_m_Au != nullptr && S != nullptr);
    bool al = fdd_aligned16(Sr) && fdd_aligned16(w) && fdd_aligned16(f_m_Au) && fdd_aligned16(S);
    return launch_ew(ScaledResidualOp{Sr, w, f_m_Au, S, alpha}, size, al, stream);
}

int fdd_amg_smooth_start(double *work, double *Sr, const double *f, const double *D_val, double coef, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(work != nullptr && Sr != nullptr && f != nullptr && D_val != nullptr);
    bool al = fdd_aligned16(work) && fdd_aligned16(Sr) && fdd_aligned16(f) && fdd_aligned16(D_val);
    return launch_ew(SmoothStartOp{work, Sr, f, D_val, coef}, size, al, stream);
}

int fdd_amg_vector_set_to_value_f32(float *data, float value, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(data != nullptr);
    return launch_ew(SetF32Op{data, value}, size, fdd_aligned16(data), stream);
}

int fdd_amg_smooth_start_f32(float *work, float *Sr, const float *f, const float *D_val, float coef, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(work != nullptr && Sr != nullptr && f != nullptr && D_val != nullptr);
    bool al = fdd_aligned16(work) && fdd_aligned16(Sr) && fdd_aligned16(f) && fdd_aligned16(D_val);
    return launch_ew(SmoothStartF32Op{work, Sr, f, D_val, coef}, size, al, stream);
}

int fdd_amg_main_polynomial_evaluation(double *w, double *v, const double *r, const double *D_val, double alpha, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(w != nullptr && v != nullptr && r != nullptr && D_val != nullptr);
    bool al = fdd_aligned16(w) && fdd_aligned16(v) && fdd_aligned16(r) && fdd_aligned16(D_val);
    return launch_ew(PolyEvalOp{w, v, r, D_val, alpha}, size, al, stream);
}

int fdd_amg_main_update_field(double *u, const double *w, const double *D_val, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(u != nullptr && w != nullptr && D_val != nullptr);
    bool al = fdd_aligned16(u) && fdd_aligned16(w) && fdd_aligned16(D_val);
    return launch_ew(UpdateFieldOp{u, w, D_val}, size, al, stream);
}

int fdd_amg_vector_multiplication(double *uv, const double *u, const double *v, int size, void *stream)
{
    FDD_REQUIRE(size >= 0);
    if (size == 0) return 0;
    FDD_REQUIRE(uv != nullptr && u != nullptr && v != nullptr);
    bool al = fdd_aligned16(uv) && fdd_aligned16(u) && fdd_aligned16(v);
    return launch_ew(VMulOp{uv, u, v}, size, al, stream);
}

} // extern "C"
