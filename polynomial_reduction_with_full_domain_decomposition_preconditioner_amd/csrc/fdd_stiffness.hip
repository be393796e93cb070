// Matrix-free SEM stiffness Au = D^T G D u on GLL points, gfx950.
// Replaces domain.okl:5-98 (uniform degree) and subdomain.okl:4-101 (mixed
// degree with per-point indirection).
//
// (1) Reference two-launch form (stiffness_matrix_1 / _2, one lane per point,
//     global scratch GDu): kept for API parity and for degrees above 15.
//     112 B/point of HBM traffic (72 in 2-D).
//
// (2) Fused kernel (fdd_dom_stiffness_matrix / fdd_sub_stiffness_matrix):
//     64 B/point (u 8 + six geometric factors 48 + Au 8).  An element of
//     n = N+1 points per edge is owned by n*n lanes, lane (i,j) keeps its
//     k-column of u in registers.  For each k-slab the slab of u and then the
//     slabs of G*Du are staged through LDS (rows padded by one double: bank
//     conflict free for 8-B reads), x/y contractions read LDS, the z
//     contraction stays in registers.  With n = 8 an element is exactly one
//     64-lane wavefront and a 256-lane workgroup holds four elements.
//
//     Every output keeps the reference's operation order: Du_d and Au_d are
//     summed over p = 0..n-1 from 0.0, G*Du is evaluated left to right,
//     Au = (Au_1 + Au_2) + Au_3.  With -ffp-contract=off the fused kernel is
//     bit-identical to the two-launch form and to the OCCA-Serial arithmetic.
//
// Both are HBM-bound: 12n+17 flop/point is 1.77 flop/B at N=7.
#include "fdd_common.h"

namespace
{

constexpr int kBlock = 256;

template <typename T>
struct GPtrsT
{
    const T *g[FDD_NUM_GEOM_FACTS];
};

struct GPtrs
{
    const double *g[FDD_NUM_GEOM_FACTS];
};
struct GDuPtrs
{
    double *g[3];
};
struct LevelTable
{
    const double *D_hat[16];
    int poly_degree[16];
};

// ---------------------------------------------------------------------------
// (1) reference form, uniform degree: domain.okl:5-52 / :54-98
// ---------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(kBlock) void dom_stiffness_1_kernel(GDuPtrs GDu, const double *__restrict__ u, const double *__restrict__ D_hat, GPtrs G, int num_points, int n_x)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int n_xy = n_x * n_x;
    const int num_elem_points = (DIM == 2) ? n_xy : n_xy * n_x;
    const int e = idx / num_elem_points;
    const int v = idx % num_elem_points;
    const double *ue = u + (size_t)e * num_elem_points;

    if (DIM == 2)
    {
        const int i = v % n_x;
        const int j = v / n_x;
        double Du_1 = 0.0, Du_2 = 0.0;
        for (int k = 0; k < n_x; k++)
        {
            Du_1 += D_hat[k + i * n_x] * ue[k + j * n_x];
            Du_2 += D_hat[k + j * n_x] * ue[i + k * n_x];
        }
        GDu.g[0][idx] = G.g[0][idx] * Du_1 + G.g[2][idx] * Du_2;
        GDu.g[1][idx] = G.g[2][idx] * Du_1 + G.g[1][idx] * Du_2;
    }
    else
    {
        const int i = v % n_x;
        const int j = (v / n_x) % n_x;
        const int k = v / n_xy;
        double Du_1 = 0.0, Du_2 = 0.0, Du_3 = 0.0;
        for (int p = 0; p < n_x; p++)
        {
            Du_1 += D_hat[p + i * n_x] * ue[p + j * n_x + k * n_xy];
            Du_2 += D_hat[p + j * n_x] * ue[i + p * n_x + k * n_xy];
            Du_3 += D_hat[p + k * n_x] * ue[i + j * n_x + p * n_xy];
        }
        GDu.g[0][idx] = G.g[0][idx] * Du_1 + G.g[3][idx] * Du_2 + G.g[4][idx] * Du_3;
        GDu.g[1][idx] = G.g[3][idx] * Du_1 + G.g[1][idx] * Du_2 + G.g[5][idx] * Du_3;
        GDu.g[2][idx] = G.g[4][idx] * Du_1 + G.g[5][idx] * Du_2 + G.g[2][idx] * Du_3;
    }
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void dom_stiffness_2_kernel(double *__restrict__ Au, GDuPtrs GDu, const double *__restrict__ D_hat, int num_points, int n_x)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int n_xy = n_x * n_x;
    const int num_elem_points = (DIM == 2) ? n_xy : n_xy * n_x;
    const int e = idx / num_elem_points;
    const int v = idx % num_elem_points;
    const size_t o = (size_t)e * num_elem_points;

    if (DIM == 2)
    {
        const int i = v % n_x;
        const int j = v / n_x;
        double Au_1 = 0.0, Au_2 = 0.0;
        for (int k = 0; k < n_x; k++)
        {
            Au_1 += D_hat[i + k * n_x] * GDu.g[0][o + (k + j * n_x)];
            Au_2 += D_hat[j + k * n_x] * GDu.g[1][o + (i + k * n_x)];
        }
        Au[idx] = Au_1 + Au_2;
    }
    else
    {
        const int i = v % n_x;
        const int j = (v / n_x) % n_x;
        const int k = v / n_xy;
        double Au_1 = 0.0, Au_2 = 0.0, Au_3 = 0.0;
        for (int p = 0; p < n_x; p++)
        {
            Au_1 += D_hat[i + p * n_x] * GDu.g[0][o + (p + j * n_x + k * n_xy)];
            Au_2 += D_hat[j + p * n_x] * GDu.g[1][o + (i + p * n_x + k * n_xy)];
            Au_3 += D_hat[k + p * n_x] * GDu.g[2][o + (i + j * n_x + p * n_xy)];
        }
        Au[idx] = Au_1 + Au_2 + Au_3;
    }
}

// ---------------------------------------------------------------------------
// (1') reference form, per-point indirection: subdomain.okl:4-53 / :55-101
// ---------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(kBlock) void sub_stiffness_1_kernel(GDuPtrs GDu, const double *__restrict__ u, LevelTable T, const int *__restrict__ offset, const int *__restrict__ vert, const int *__restrict__ level, GPtrs G, int num_points)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int o = offset[idx];
    const int v = vert[idx];
    const int l = level[idx];
    const int n_x = T.poly_degree[l] + 1;
    const int n_xy = n_x * n_x;
    const double *D_hat = T.D_hat[l];

    if (DIM == 2)
    {
        const int i = v % n_x;
        const int j = v / n_x;
        double Du_1 = 0.0, Du_2 = 0.0;
        for (int k = 0; k < n_x; k++)
        {
            Du_1 += D_hat[k + i * n_x] * u[o + (k + j * n_x)];
            Du_2 += D_hat[k + j * n_x] * u[o + (i + k * n_x)];
        }
        GDu.g[0][idx] = G.g[0][idx] * Du_1 + G.g[2][idx] * Du_2;
        GDu.g[1][idx] = G.g[2][idx] * Du_1 + G.g[1][idx] * Du_2;
    }
    else
    {
        const int i = v % n_x;
        const int j = (v / n_x) % n_x;
        const int k = v / n_xy;
        double Du_1 = 0.0, Du_2 = 0.0, Du_3 = 0.0;
        for (int p = 0; p < n_x; p++)
        {
            Du_1 += D_hat[p + i * n_x] * u[o + (p + j * n_x + k * n_xy)];
            Du_2 += D_hat[p + j * n_x] * u[o + (i + p * n_x + k * n_xy)];
            Du_3 += D_hat[p + k * n_x] * u[o + (i + j * n_x + p * n_xy)];
        }
        GDu.g[0][idx] = G.g[0][idx] * Du_1 + G.g[3][idx] * Du_2 + G.g[4][idx] * Du_3;
        GDu.g[1][idx] = G.g[3][idx] * Du_1 + G.g[1][idx] * Du_2 + G.g[5][idx] * Du_3;
        GDu.g[2][idx] = G.g[4][idx] * Du_1 + G.g[5][idx] * Du_2 + G.g[2][idx] * Du_3;
    }
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void sub_stiffness_2_kernel(double *__restrict__ Au, GDuPtrs GDu, LevelTable T, const int *__restrict__ offset, const int *__restrict__ vert, const int *__restrict__ level, int num_points)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int o = offset[idx];
    const int v = vert[idx];
    const int l = level[idx];
    const int n_x = T.poly_degree[l] + 1;
    const int n_xy = n_x * n_x;
    const double *D_hat = T.D_hat[l];

    if (DIM == 2)
    {
        const int i = v % n_x;
        const int j = v / n_x;
        double Au_1 = 0.0, Au_2 = 0.0;
        for (int k = 0; k < n_x; k++)
        {
            Au_1 += D_hat[i + k * n_x] * GDu.g[0][o + (k + j * n_x)];
            Au_2 += D_hat[j + k * n_x] * GDu.g[1][o + (i + k * n_x)];
        }
        Au[idx] = Au_1 + Au_2;
    }
    else
    {
        const int i = v % n_x;
        const int j = (v / n_x) % n_x;
        const int k = v / n_xy;
        double Au_1 = 0.0, Au_2 = 0.0, Au_3 = 0.0;
        for (int p = 0; p < n_x; p++)
        {
            Au_1 += D_hat[i + p * n_x] * GDu.g[0][o + (p + j * n_x + k * n_xy)];
            Au_2 += D_hat[j + p * n_x] * GDu.g[1][o + (i + p * n_x + k * n_xy)];
            Au_3 += D_hat[k + p * n_x] * GDu.g[2][o + (i + j * n_x + p * n_xy)];
        }
        Au[idx] = Au_1 + Au_2 + Au_3;
    }
}

// ---------------------------------------------------------------------------
// (2) fused kernel, 3-D, n = N+1 in [2, 16]
// ---------------------------------------------------------------------------
template <int n>
struct FusedCfg
{
    static constexpr int nn = n * n;
    static constexpr int epb = (kBlock / nn) > 0 ? (kBlock / nn) : 1; // elements per workgroup
    static constexpr int ld = n + 1;                                  // padded LDS row
    static constexpr int slab = n * ld;
};

// Synchronisation between the LDS phases of one element.  When n*n divides 64
// an element never straddles a wavefront (n = 2, 4, 8: the headline N = 7), the
// LDS executes a wave's instructions in order, and no barrier is needed at all:
// every wave streams its elements independently.  Otherwise a barrier that
// waits on the LDS counter only -- __syncthreads() would also drain the vector
// memory counter, i.e. the geometric factors prefetched for the next slab.
template <bool kWaveLocal>
__device__ __forceinline__ void element_sync()
{
    if (kWaveLocal)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// kGather: u is read through point_dof (u[e,i,j,k] = v[point_dof[...]], 0 where
// the point has no dof): the boolean scatter Q of Subdomain fused into the load.
// T: double, or float for the single-precision preconditioner (the reference's PTYPE = Float, config.hpp:19-20)
// kAffine: the six factors of a point are NOT streamed (48 of the 64 B per point) but formed from six numbers per element
// and the GLL weights, G_f(e; i, j, k) = c_f(e) * (w_i w_j) w_k -- what they are on an element that is an affine image
// of the reference cube (every element of a box mesh).  G.g[0] then points to c (element-major, 6 per element of this
// list), G.g[1] to the n weights.  An option of this build (the reference always streams G): the host layer offers it
// only where the mesh's own factor arrays satisfy the product form to rounding (host/element.hpp: affine_factors).
template <typename T, int n, bool kGather, bool kNTStore, bool kAffine = false>
__global__ __launch_bounds__(kBlock, (kAffine && n == 8 && sizeof(T) == 8) ? 4 : 1) void fused_stiffness_kernel_t(T *__restrict__ Au, const T *__restrict__ u, const int *__restrict__ point_dof, const double *__restrict__ u_scale, const T *__restrict__ D_hat, GPtrsT<T> G, const int *__restrict__ elem_offset, int num_elements)
{
    using C = FusedCfg<n>;
    constexpr int nn = C::nn;
    constexpr int n3 = nn * n;
    constexpr bool kWaveLocal = (64 % nn == 0);
    // D_hat rows/columns a lane needs in every slab live in registers up to
    // n = 8 (4 x 8 doubles); above that they are re-read from LDS per slab.
    constexpr bool kDReg = (n <= 8);
    constexpr int nd = kDReg ? n : 1;

    __shared__ T s_D[n * n];
    __shared__ T s_u[C::epb][n * C::slab];   // the element, rows padded; slab k is reused for Au_1 + Au_2 once consumed
    __shared__ T s_g[2][2][C::epb][C::slab]; // GDu_1 / GDu_2 of the slab, T buffered: one sync per slab

    // n = 8: the element index is the wavefront index.  Telling the compiler so
    // (readfirstlane) keeps the element base, the `active` test and all array
    // bases in scalar registers: loads take the SGPR-base + 32-bit lane offset form.
    constexpr bool kWaveIsElement = (nn == 64);
    const int tid = threadIdx.x;
    const int e_loc = kWaveIsElement ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / nn;
    const int ij = tid - e_loc * nn;
    const int j = ij / n;
    const int i = ij - j * n;
    const int elem = blockIdx.x * C::epb + e_loc;
    const bool active = (e_loc < C::epb) && (elem < num_elements);

    size_t base = 0;
    if (active) base = elem_offset ? (size_t)elem_offset[elem] : (size_t)elem * n3;

    const int el = active ? e_loc : 0;
    T *su = s_u[el];
    const int lpos = i + j * C::ld;

    // this lane's k-column of u: registers for the z contraction, LDS for x and y;
    // geometric factors of the first kPF slabs right behind it (inside the loop slab
    // k + kPF is requested as soon as slab k's are consumed).  All of it is in
    // flight before D_hat is staged.
    constexpr int kPF = 1; // slabs of geometric factors in flight (2 measured no faster at n = 8: the kernel is not latency-bound)
    T r_u[n], r_3[n], gq[kPF][FDD_NUM_GEOM_FACTS];
    T cf[FDD_NUM_GEOM_FACTS], wij = T(0); // kAffine: the element's six numbers, w_i w_j of this lane
#pragma unroll
    for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) cf[f] = T(0);
#pragma unroll
    for (int k = 0; k < n; k++) r_u[k] = T(0);
#pragma unroll
    for (int s = 0; s < kPF; s++)
#pragma unroll
        for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) gq[s][f] = T(0);
    if (active)
    {
        if (kGather)
        {
            const int *pd = point_dof + base;
            int d[n];
#pragma unroll
            for (int k = 0; k < n; k++) d[k] = __builtin_nontemporal_load(pd + (ij + k * nn));
            // unconditional loads on a selected index (dof 0 stands in where the point has none): a load
            // under a lane predicate becomes a branch with its own wait and the n loads serialise
#pragma unroll
            for (int k = 0; k < n; k++) r_u[k] = u[d[k] < 0 ? 0 : d[k]];
#pragma unroll
            for (int k = 0; k < n; k++) r_u[k] = (d[k] < 0) ? T(0) : r_u[k];
            if (u_scale) // v stands for (*u_scale) * v: a Krylov vector kept unnormalised (math.okl:29-35 applied on load)
            {
                const T sc = (T)(*u_scale);
#pragma unroll
                for (int k = 0; k < n; k++) r_u[k] = sc * r_u[k];
            }
        }
        else
        {
            const T *up = u + base;
#pragma unroll
            for (int k = 0; k < n; k++) r_u[k] = __builtin_nontemporal_load(up + (ij + k * nn));
        }
        if (kAffine)
        {
#pragma unroll
            for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) cf[f] = G.g[0][(size_t)elem * FDD_NUM_GEOM_FACTS + f];
            wij = G.g[1][i] * G.g[1][j];
        }
        else
        {
#pragma unroll
            for (int s = 0; s < kPF; s++)
#pragma unroll
                for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) gq[s][f] = __builtin_nontemporal_load(G.g[f] + base + (ij + s * nn));
        }
    }

    for (int t = tid; t < n * n; t += kBlock) s_D[t] = D_hat[t];

#pragma unroll
    for (int k = 0; k < n; k++) r_3[k] = T(0);
    if (active)
    {
#pragma unroll
        for (int k = 0; k < n; k++) su[lpos + k * C::slab] = r_u[k];
    }

    __syncthreads(); // s_D (written across elements) and s_u

    T D_i[nd], D_j[nd], Dt_i[nd], Dt_j[nd];
    if (kDReg)
    {
#pragma unroll
        for (int p = 0; p < nd; p++)
        {
            D_i[p] = s_D[p + i * n];  // D_hat[p + i*n_x]: pass 1, x
            D_j[p] = s_D[p + j * n];  // D_hat[p + j*n_x]: pass 1, y
            Dt_i[p] = s_D[i + p * n]; // D_hat[i + p*n_x]: pass 2, x
            Dt_j[p] = s_D[j + p * n]; // D_hat[j + p*n_x]: pass 2, y
        }
    }

    // The slab loop is deliberately NOT unrolled beyond the prefetch depth:
    // unrolled, every D_hat entry becomes loop-invariant register state
    // (> 256 VGPRs and scratch spills).  A wave waits one full HBM latency per
    // slab unless enough slabs of geometric factors are already requested:
    // kPF slabs are in flight per lane (12 VGPRs each).
#pragma unroll 1
    for (int k0 = 0; k0 < n; k0 += kPF)
    {
#pragma unroll
        for (int s = 0; s < kPF; s++)
        {
            const int k = k0 + s;
            if ((n % kPF != 0) && k >= n) continue;
            // this slot's factors move to g; slab k + kPF is requested into the slot
            T g[FDD_NUM_GEOM_FACTS];
            if (kAffine)
            {
                const T w3 = wij * G.g[1][k]; // w_k: wave-uniform address
#pragma unroll
                for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) g[f] = cf[f] * w3;
            }
            else
            {
#pragma unroll
                for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) g[f] = gq[s][f];
            }
            if (!kAffine && active && k + kPF < n)
            {
                const int off = ij + (k + kPF) * nn;
#pragma unroll
                for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) gq[s][f] = __builtin_nontemporal_load(G.g[f] + base + off);
            }

            // row k of D_hat: wave-uniform address -> scalar loads, lives in SGPRs
            T Dk[n];
#pragma unroll
            for (int p = 0; p < n; p++) Dk[p] = D_hat[p + k * n];

            // opaque copies of i, j stop the compiler from hoisting the D_hat LDS
            // reads out of the slab loop when they are not meant to be registers
            int io = i, jo = j;
            if (!kDReg) asm volatile("" : "+v"(io), "+v"(jo));

            const T *suk = su + k * C::slab;
            T Du_1 = T(0), Du_2 = T(0), Du_3 = T(0);
#pragma unroll
            for (int p = 0; p < n; p++)
            {
                const T di = kDReg ? D_i[kDReg ? p : 0] : s_D[p + io * n];
                const T dj = kDReg ? D_j[kDReg ? p : 0] : s_D[p + jo * n];
                Du_1 += di * suk[p + j * C::ld];
                Du_2 += dj * suk[i + p * C::ld];
                Du_3 += Dk[p] * r_u[p];
            }

            const T GDu_1 = g[0] * Du_1 + g[3] * Du_2 + g[4] * Du_3;
            const T GDu_2 = g[3] * Du_1 + g[1] * Du_2 + g[5] * Du_3;
            const T GDu_3 = g[4] * Du_1 + g[5] * Du_2 + g[2] * Du_3;

            T *sg1 = s_g[k & 1][0][el];
            T *sg2 = s_g[k & 1][1][el];
            if (active)
            {
                sg1[lpos] = GDu_1;
                sg2[lpos] = GDu_2;
            }
            element_sync<kWaveLocal>();

            T Au_1 = T(0), Au_2 = T(0);
#pragma unroll
            for (int p = 0; p < n; p++)
            {
                const T dti = kDReg ? Dt_i[kDReg ? p : 0] : s_D[io + p * n];
                const T dtj = kDReg ? Dt_j[kDReg ? p : 0] : s_D[jo + p * n];
                Au_1 += dti * sg1[p + j * C::ld];
                Au_2 += dtj * sg2[i + p * C::ld];
            }
            // every lane of the element is past its reads of slab k of s_u (they
            // precede the sync above): the slot now holds Au_1 + Au_2 of this point
            if (active) su[lpos + k * C::slab] = Au_1 + Au_2;

            // Au_3(i,j,m) += D_hat[m + k*n_x] * GDu_3(i,j,k): the reference's p = k term
#pragma unroll
            for (int m = 0; m < n; m++) r_3[m] += Dk[m] * GDu_3;
        }
    }

    if (active)
    {
        // the s_u slots are read back by the lane that wrote them
        T *Aup = Au + base;
#pragma unroll
        for (int k = 0; k < n; k++)
        {
            const T v = su[lpos + k * C::slab] + r_3[k];
            if (kNTStore)
                __builtin_nontemporal_store(v, Aup + (ij + k * nn));
            else
                Aup[ij + k * nn] = v;
        }
    }
}

template <typename T, int n>
int launch_fused_t(T *Au, const T *u, const int *point_dof, const double *u_scale, const T *D_hat, const GPtrsT<T> &G, const int *elem_offset, int num_elements, void *stream)
{
    using C = FusedCfg<n>;
    const int grid = (num_elements + C::epb - 1) / C::epb;
    static const bool nt_store = fdd_env_int("FDD_TUNE_STIFFNESS_NT_STORE", 1) != 0;
    if (point_dof and nt_store)
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, true, true>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements);
    else if (point_dof)
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, true, false>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements);
    else if (nt_store)
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, false, true>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements);
    else
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, false, false>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements);
    FDD_LAUNCH_CHECK();
    return 0;
}

// One workgroup per element: c_f = G_f(p0) / W(p0) at the element's middle point, and the largest deviation of any
// factor of any point from c_f W(p), relative to max_f |c_f| W(p); W(p) = (w_i w_j) w_k as the kernel above forms it.
__global__ __launch_bounds__(kBlock) void affine_detect_kernel(double *__restrict__ c_out, double *__restrict__ deviation, GPtrs G, const int *__restrict__ elem_offset, const double *__restrict__ w, int n, int num_elements)
{
    __shared__ double s_c[FDD_NUM_GEOM_FACTS];
    __shared__ double s_max[kBlock / FDD_WAVE];
    const int e = blockIdx.x;
    if (e >= num_elements) return;
    const int n3 = n * n * n;
    const size_t base = elem_offset ? (size_t)elem_offset[e] : (size_t)e * n3;
    if (threadIdx.x < FDD_NUM_GEOM_FACTS)
    {
        const int m = n / 2, p0 = m + m * n + m * n * n;
        s_c[threadIdx.x] = G.g[threadIdx.x][base + p0] / ((w[m] * w[m]) * w[m]);
    }
    __syncthreads();
    double scale = 0.0;
#pragma unroll
    for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) scale = fmax(scale, fabs(s_c[f]));
    double worst = 0.0;
    for (int p = threadIdx.x; p < n3; p += kBlock)
    {
        const int i = p % n, j = (p / n) % n, k = p / (n * n);
        const double W = (w[i] * w[j]) * w[k];
#pragma unroll
        for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) worst = fmax(worst, fabs(G.g[f][base + p] - s_c[f] * W) / (scale * W));
    }
    if (!(scale > 0.0)) worst = 1.0; // all factors zero in the middle: not a usable element
#pragma unroll
    for (int off = FDD_WAVE / 2; off > 0; off >>= 1) worst = fmax(worst, __shfl_down(worst, off, FDD_WAVE));
    if ((threadIdx.x & (FDD_WAVE - 1)) == 0) s_max[threadIdx.x / FDD_WAVE] = worst;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int v = 1; v < kBlock / FDD_WAVE; v++) worst = fmax(worst, s_max[v]);
        deviation[e] = worst;
#pragma unroll
        for (int f = 0; f < FDD_NUM_GEOM_FACTS; f++) c_out[(size_t)e * FDD_NUM_GEOM_FACTS + f] = s_c[f];
    }
}

template <typename T, int n>
int launch_fused_affine_t(T *Au, const T *u, const int *point_dof, const double *u_scale, const T *D_hat, const T *elem_factors, const T *gll_weights, const int *elem_offset, int num_elements, void *stream)
{
    using C = FusedCfg<n>;
    const int grid = (num_elements + C::epb - 1) / C::epb;
    GPtrsT<T> g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = nullptr;
    g.g[0] = elem_factors;
    g.g[1] = gll_weights;
    if (point_dof)
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, true, true, true>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements);
    else
        hipLaunchKernelGGL((fused_stiffness_kernel_t<T, n, false, true, true>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements);
    FDD_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int fused_dispatch_affine(T *Au, const T *u, const int *point_dof, const double *u_scale, const T *D_hat, const T *elem_factors, const T *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && elem_factors != nullptr && gll_weights != nullptr);
    switch (poly_degree + 1)
    {
#define FDD_AFFINE_CASE(N_) \
    case N_: return launch_fused_affine_t<T, N_>(Au, u, point_dof, u_scale, D_hat, elem_factors, gll_weights, elem_offset, num_elements, stream);
        FDD_AFFINE_CASE(2)
        FDD_AFFINE_CASE(3)
        FDD_AFFINE_CASE(4)
        FDD_AFFINE_CASE(5)
        FDD_AFFINE_CASE(6)
        FDD_AFFINE_CASE(7)
        FDD_AFFINE_CASE(8)
        FDD_AFFINE_CASE(9)
        FDD_AFFINE_CASE(10)
        FDD_AFFINE_CASE(11)
        FDD_AFFINE_CASE(12)
        FDD_AFFINE_CASE(13)
        FDD_AFFINE_CASE(14)
        FDD_AFFINE_CASE(15)
        FDD_AFFINE_CASE(16)
#undef FDD_AFFINE_CASE
    default:
        fdd_set_error("fused stiffness kernel supports poly_degree 1..15, got %d", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

template <int n>
int launch_fused(double *Au, const double *u, const int *point_dof, const double *u_scale, const double *D_hat, const GPtrs &G, const int *elem_offset, int num_elements, void *stream)
{
    GPtrsT<double> g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = G.g[k];
    return launch_fused_t<double, n>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
}

// the single-precision form: gather-on-load only (the preconditioner's dof-space solve)
int fused_dispatch_f32(float *Au, const float *u, const int *point_dof, const double *u_scale, const float *D_hat, const float *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && G != nullptr);
    GPtrsT<float> g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++)
    {
        FDD_REQUIRE(G[k] != nullptr);
        g.g[k] = G[k];
    }
    switch (poly_degree + 1)
    {
#define FDD_F32_CASE(N_) \
    case N_: return launch_fused_t<float, N_>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
        FDD_F32_CASE(2)
        FDD_F32_CASE(3)
        FDD_F32_CASE(4)
        FDD_F32_CASE(5)
        FDD_F32_CASE(6)
        FDD_F32_CASE(7)
        FDD_F32_CASE(8)
        FDD_F32_CASE(9)
        FDD_F32_CASE(10)
        FDD_F32_CASE(11)
        FDD_F32_CASE(12)
        FDD_F32_CASE(13)
        FDD_F32_CASE(14)
        FDD_F32_CASE(15)
        FDD_F32_CASE(16)
#undef FDD_F32_CASE
    default:
        fdd_set_error("fused stiffness kernel supports poly_degree 1..15, got %d", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

int fused_dispatch(double *Au, const double *u, const int *point_dof, const double *u_scale, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && G != nullptr);
    GPtrs g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++)
    {
        FDD_REQUIRE(G[k] != nullptr);
        g.g[k] = G[k];
    }

    switch (poly_degree + 1)
    {
    case 2: return launch_fused<2>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 3: return launch_fused<3>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 4: return launch_fused<4>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 5: return launch_fused<5>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 6: return launch_fused<6>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 7: return launch_fused<7>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 8: return launch_fused<8>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 9: return launch_fused<9>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 10: return launch_fused<10>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 11: return launch_fused<11>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 12: return launch_fused<12>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 13: return launch_fused<13>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 14: return launch_fused<14>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 15: return launch_fused<15>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 16: return launch_fused<16>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    default:
        fdd_set_error("fused stiffness kernel supports poly_degree 1..15, got %d (use the two-launch form)", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------
// (2') fused kernel, 2-D, n = N+1 in [2, 16]: domain.okl:20-33 / :69-80 (the DIM == 2 branches)
// ---------------------------------------------------------------------------
// 40 B/point (u 8 + three geometric factors 24 + Au 8) against the two-launch form's 72.  One lane per point,
// floor(256 / n^2) elements per workgroup; the element and then G*Du go through LDS (rows padded by one double).
// Same operation order as the reference per point, so the result is bit-identical to the two-launch form.
template <int n, int kGroups, bool kNTStore>
__global__ __launch_bounds__(kBlock) void fused_stiffness_2d_kernel(double *__restrict__ Au, const double *__restrict__ u, const double *__restrict__ D_hat, GPtrs G, const int *__restrict__ elem_offset, int num_elements)
{
    constexpr int nn = n * n;
    constexpr int epb = kBlock / nn;
    constexpr int ld = n + 1;
    static_assert(epb >= 1, "an element must fit a workgroup");
    constexpr bool kWaveLocal = (64 % nn == 0); // an element never straddles a wavefront: no workgroup barrier between its LDS phases
    constexpr bool kDReg = (n <= 8);            // the four D_hat rows/columns of a lane in registers
    constexpr int nd = kDReg ? n : 1;

    __shared__ double s_D[nn];
    __shared__ double s_u[epb][n * ld];
    __shared__ double s_g[2][epb][n * ld];

    const int tid = threadIdx.x;
    const int e_loc = tid / nn;
    const int ij = tid - e_loc * nn;
    const int j = ij / n;
    const int i = ij - j * n;
    const int el = (e_loc < epb) ? e_loc : 0;
    const int lpos = i + j * ld;

    for (int t = tid; t < nn; t += kBlock) s_D[t] = D_hat[t];
    __syncthreads();
    double D_i[nd], D_j[nd], Dt_i[nd], Dt_j[nd];
    if (kDReg)
    {
#pragma unroll
        for (int k = 0; k < nd; k++)
        {
            D_i[k] = s_D[k + i * n];
            D_j[k] = s_D[k + j * n];
            Dt_i[k] = s_D[i + k * n];
            Dt_j[k] = s_D[j + k * n];
        }
    }

    // A workgroup takes kGroups consecutive groups of epb elements and requests all of their u and factors before it
    // computes the first: one point per lane and group is 32 B, too little in flight to cover the HBM latency.
    const int first = blockIdx.x * kGroups;
    bool active[kGroups];
    size_t at[kGroups];
    double r_u[kGroups], g0[kGroups], g1[kGroups], g2[kGroups];
#pragma unroll
    for (int q = 0; q < kGroups; q++)
    {
        const int elem = (first + q) * epb + e_loc;
        active[q] = (e_loc < epb) && (elem < num_elements);
        at[q] = 0;
        r_u[q] = g0[q] = g1[q] = g2[q] = 0.0;
        if (active[q])
        {
            at[q] = (elem_offset ? (size_t)elem_offset[elem] : (size_t)elem * nn) + ij;
            r_u[q] = __builtin_nontemporal_load(u + at[q]);
            g0[q] = __builtin_nontemporal_load(G.g[0] + at[q]);
            g1[q] = __builtin_nontemporal_load(G.g[1] + at[q]);
            g2[q] = __builtin_nontemporal_load(G.g[2] + at[q]);
        }
    }

    double *su = s_u[el];
    double *sg1 = s_g[0][el];
    double *sg2 = s_g[1][el];
#pragma unroll
    for (int q = 0; q < kGroups; q++)
    {
        if (active[q]) su[lpos] = r_u[q];
        element_sync<kWaveLocal>();

        int io = i, jo = j;
        if (!kDReg) asm volatile("" : "+v"(io), "+v"(jo));
        double Du_1 = 0.0, Du_2 = 0.0;
#pragma unroll
        for (int k = 0; k < n; k++)
        {
            const double di = kDReg ? D_i[kDReg ? k : 0] : s_D[k + io * n];
            const double dj = kDReg ? D_j[kDReg ? k : 0] : s_D[k + jo * n];
            Du_1 += di * su[k + j * ld];
            Du_2 += dj * su[i + k * ld];
        }
        if (active[q])
        {
            sg1[lpos] = g0[q] * Du_1 + g2[q] * Du_2;
            sg2[lpos] = g2[q] * Du_1 + g1[q] * Du_2;
        }
        element_sync<kWaveLocal>();

        double Au_1 = 0.0, Au_2 = 0.0;
#pragma unroll
        for (int k = 0; k < n; k++)
        {
            const double dti = kDReg ? Dt_i[kDReg ? k : 0] : s_D[io + k * n];
            const double dtj = kDReg ? Dt_j[kDReg ? k : 0] : s_D[jo + k * n];
            Au_1 += dti * sg1[k + j * ld];
            Au_2 += dtj * sg2[i + k * ld];
        }
        if (active[q])
        {
            if (kNTStore)
                __builtin_nontemporal_store(Au_1 + Au_2, Au + at[q]);
            else
                Au[at[q]] = Au_1 + Au_2;
        }
    }
}

template <int n>
int launch_fused_2d(double *Au, const double *u, const double *D_hat, const GPtrs &G, const int *elem_offset, int num_elements, void *stream)
{
    constexpr int epb = kBlock / (n * n);
    const int groups = (num_elements + epb - 1) / epb;
    static const bool nt_store = fdd_env_int("FDD_TUNE_STIFFNESS_NT_STORE", 1) != 0;
    static const int per_block = fdd_env_int("FDD_TUNE_STIFFNESS_2D_GROUPS", 4);
#define FDD_2D_LAUNCH(K_) \
    do \
    { \
        const int grid = (groups + K_ - 1) / K_; \
        if (nt_store) \
            hipLaunchKernelGGL((fused_stiffness_2d_kernel<n, K_, true>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, D_hat, G, elem_offset, num_elements); \
        else \
            hipLaunchKernelGGL((fused_stiffness_2d_kernel<n, K_, false>), dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, u, D_hat, G, elem_offset, num_elements); \
    } while (0)
    // few elements: one group per workgroup keeps more workgroups in the launch
    if (per_block >= 4 && groups >= 4 * 2048)
        FDD_2D_LAUNCH(4);
    else if (per_block >= 2 && groups >= 2 * 2048)
        FDD_2D_LAUNCH(2);
    else
        FDD_2D_LAUNCH(1);
#undef FDD_2D_LAUNCH
    FDD_LAUNCH_CHECK();
    return 0;
}

int fused_dispatch_2d(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && G != nullptr);
    GPtrs g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = G[k]; // 2-D reads G[0] = G11, G[1] = G22, G[2] = G12 only
    FDD_REQUIRE(g.g[0] != nullptr && g.g[1] != nullptr && g.g[2] != nullptr);
    switch (poly_degree + 1)
    {
#define FDD_2D_CASE(N_) \
    case N_: return launch_fused_2d<N_>(Au, u, D_hat, g, elem_offset, num_elements, stream);
        FDD_2D_CASE(2)
        FDD_2D_CASE(3)
        FDD_2D_CASE(4)
        FDD_2D_CASE(5)
        FDD_2D_CASE(6)
        FDD_2D_CASE(7)
        FDD_2D_CASE(8)
        FDD_2D_CASE(9)
        FDD_2D_CASE(10)
        FDD_2D_CASE(11)
        FDD_2D_CASE(12)
        FDD_2D_CASE(13)
        FDD_2D_CASE(14)
        FDD_2D_CASE(15)
        FDD_2D_CASE(16)
#undef FDD_2D_CASE
    default:
        fdd_set_error("fused 2-D stiffness kernel supports poly_degree 1..15, got %d (use the two-launch form)", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

int fill_level_table(LevelTable &T, const double *const *D_hat_ptr, const int *poly_degree, int num_levels)
{
    FDD_REQUIRE(D_hat_ptr != nullptr && poly_degree != nullptr && num_levels >= 1 && num_levels <= 16);
    for (int l = 0; l < 16; l++)
    {
        T.D_hat[l] = (l < num_levels) ? D_hat_ptr[l] : nullptr;
        T.poly_degree[l] = (l < num_levels) ? poly_degree[l] : 0;
    }
    return 0;
}

} // namespace

extern "C" {

int fdd_dom_stiffness_matrix_1(double *const GDu[3], const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], int num_points, int poly_degree, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && poly_degree >= 1 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(GDu != nullptr && u != nullptr && D_hat != nullptr && G != nullptr);
    GDuPtrs gd;
    GPtrs g;
    for (int k = 0; k < 3; k++) gd.g[k] = GDu[k];
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = G[k];
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(dom_stiffness_1_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), gd, u, D_hat, g, num_points, poly_degree + 1);
    else
        hipLaunchKernelGGL(dom_stiffness_1_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), gd, u, D_hat, g, num_points, poly_degree + 1);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_dom_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *D_hat, int num_points, int poly_degree, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && poly_degree >= 1 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(Au != nullptr && GDu != nullptr && D_hat != nullptr);
    GDuPtrs gd;
    for (int k = 0; k < 3; k++) gd.g[k] = const_cast<double *>(GDu[k]);
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(dom_stiffness_2_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, gd, D_hat, num_points, poly_degree + 1);
    else
        hipLaunchKernelGGL(dom_stiffness_2_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, gd, D_hat, num_points, poly_degree + 1);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_dom_stiffness_matrix(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], int num_elements, int poly_degree, void *stream)
{
    return fused_dispatch(Au, u, nullptr, nullptr, D_hat, G, nullptr, num_elements, poly_degree, stream);
}

int fdd_sub_stiffness_matrix(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return fused_dispatch(Au, u, nullptr, nullptr, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_stiffness_matrix_2d(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return fused_dispatch_2d(Au, u, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_sub_stiffness_matrix_gather(double *Au, const double *v, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(point_dof != nullptr);
    return fused_dispatch(Au, v, point_dof, nullptr, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_sub_stiffness_matrix_gather_scaled(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(point_dof != nullptr);
    return fused_dispatch(Au, v, point_dof, v_scale_dev, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_stiffness_matrix_affine(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *elem_factors, const double *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return fused_dispatch_affine<double>(Au, v, point_dof, v_scale_dev, D_hat, elem_factors, gll_weights, elem_offset, num_elements, poly_degree, stream);
}

int fdd_stiffness_affine_detect(double *elem_factors, double *deviation, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, const double *gll_weights, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0 && poly_degree >= 1);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(elem_factors != nullptr && deviation != nullptr && G != nullptr && gll_weights != nullptr);
    GPtrs g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++)
    {
        FDD_REQUIRE(G[k] != nullptr);
        g.g[k] = G[k];
    }
    hipLaunchKernelGGL(affine_detect_kernel, dim3(num_elements), dim3(kBlock), 0, fdd_stream(stream), elem_factors, deviation, g, elem_offset, gll_weights, poly_degree + 1, num_elements);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_stiffness_matrix_affine_f32(float *Au, const float *v, const double *v_scale_dev, const int *point_dof, const float *D_hat, const float *elem_factors, const float *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return fused_dispatch_affine<float>(Au, v, point_dof, v_scale_dev, D_hat, elem_factors, gll_weights, elem_offset, num_elements, poly_degree, stream);
}

int fdd_sub_stiffness_matrix_gather_scaled_f32(float *Au, const float *v, const double *v_scale_dev, const int *point_dof, const float *D_hat, const float *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(point_dof != nullptr);
    return fused_dispatch_f32(Au, v, point_dof, v_scale_dev, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_sub_stiffness_matrix_1(double *const GDu[3], const double *u, const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_levels, const double *const G[FDD_NUM_GEOM_FACTS], int num_points, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(GDu != nullptr && u != nullptr && offset != nullptr && vert != nullptr && level != nullptr && G != nullptr);
    LevelTable T;
    int rc = fill_level_table(T, D_hat_ptr, poly_degree, num_levels);
    if (rc) return rc;
    GDuPtrs gd;
    GPtrs g;
    for (int k = 0; k < 3; k++) gd.g[k] = GDu[k];
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = G[k];
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(sub_stiffness_1_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), gd, u, T, offset, vert, level, g, num_points);
    else
        hipLaunchKernelGGL(sub_stiffness_1_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), gd, u, T, offset, vert, level, g, num_points);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_sub_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_levels, int num_points, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(Au != nullptr && GDu != nullptr && offset != nullptr && vert != nullptr && level != nullptr);
    LevelTable T;
    int rc = fill_level_table(T, D_hat_ptr, poly_degree, num_levels);
    if (rc) return rc;
    GDuPtrs gd;
    for (int k = 0; k < 3; k++) gd.g[k] = const_cast<double *>(GDu[k]);
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(sub_stiffness_2_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, gd, T, offset, vert, level, num_points);
    else
        hipLaunchKernelGGL(sub_stiffness_2_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Au, gd, T, offset, vert, level, num_points);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
