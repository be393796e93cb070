// Matrix-free grid transfer of the AMG V-cycle's geometric leading level (host/low_order.hpp: geometric_level).
//
// The interpolator between the GLL lattice of the degree-N elements and its coarsened lattice is multi-linear in the
// element's reference coordinates: the same (N+1) x m table of 1-D weights in every element and direction.  Stored as a
// CSR matrix (the role of P_fem / R_fem, subdomain.tpp:3526-3545: u += P e on the way up, f_c = P^T v on the way down) it
// costs 12 B per entry -- 37.5 M entries between the two finest levels at 32^3 elements of degree 7, and the two SpMVs are
// a fifth of the cycle.  Here an element's lanes read its m^3 coarse values once and form its (N+1)^3 fine values from
// the weight table: what moves is one 4-byte map entry and the value per fine point (20 B per fine dof on the way up,
// 12 on the way down) and a few bytes per coarse node.
//
//   owner_dof[point]  the dof of a lattice point where the point is the FIRST of its dof (the interpolator's row of a
//                     dof is defined by that occurrence), -1 elsewhere (later occurrences, Dirichlet points)
//   coarse_dof[e*m^3 + t]  the coarse dof of the element's kept node t, -1 on a Dirichlet node
//
// The same operator as the CSR interpolator, with its sums formed in another order (weights multiplied direction by
// direction instead of as one product per entry; restriction summed per element, then over the elements of a coarse
// node in ascending element order): equal to rounding, deterministic.
#include "fdd_common.h"

namespace
{
constexpr int kBlock = 256;
constexpr int kMaxN = 16;

struct LatticeInterp
{
    int n, m;
    int lo[kMaxN], hi[kMaxN];
    double wl[kMaxN]; // weight of lo; hi gets 1 - wl (lo == hi: a kept node, weight 1)
};

// fine value of lattice node (i, j, k) of an element from its coarse values c[m^3] (x fastest)
template <typename T, int NF>
__global__ __launch_bounds__(kBlock) void lattice_prolong_kernel(T *u, const T *__restrict__ ec, const int *__restrict__ owner_dof, const int *__restrict__ coarse_dof, const LatticeInterp I, int num_elements)
{
    constexpr int LPE = NF * NF;      // lanes of an element: one per (i, j) column
    constexpr int EPW = kBlock / LPE; // elements of a workgroup
    constexpr int MC = (NF / 2 + 1) * (NF / 2 + 1) * (NF / 2 + 1);
    __shared__ T c[EPW][MC];
    const int m = I.m, mc = m * m * m;
    const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
    const int e = blockIdx.x * EPW + slot;
    const bool active = slot < EPW && e < num_elements;
    if (active)
        for (int t = lane; t < mc; t += LPE)
        {
            const int d = coarse_dof[(size_t)e * mc + t];
            c[slot][t] = (d >= 0) ? ec[d] : T(0);
        }
    __syncthreads();
    if (!active) return;
    const int i = lane % NF, j = lane / NF;
    const size_t p0 = (size_t)e * NF * NF * NF + lane;
    int d[NF];
    T uk[NF];
#pragma unroll
    for (int k = 0; k < NF; k++) d[k] = __builtin_nontemporal_load(owner_dof + p0 + (size_t)k * LPE);
#pragma unroll
    for (int k = 0; k < NF; k++) uk[k] = (d[k] >= 0) ? u[d[k]] : T(0);
    // the column's value on every coarse z level: the (at most four) coarse columns around (i, j)
    const int il = I.lo[i], ih = I.hi[i], jl = I.lo[j], jh = I.hi[j];
    const T wxl = (T)I.wl[i], wxh = (il == ih) ? T(0) : (T)(1.0 - I.wl[i]);
    const T wyl = (T)I.wl[j], wyh = (jl == jh) ? T(0) : (T)(1.0 - I.wl[j]);
    const T w00 = wxl * wyl, w10 = wxh * wyl, w01 = wxl * wyh, w11 = wxh * wyh;
    const T *cs = c[slot];
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
        const int kl = I.lo[k], kh = I.hi[k];
        const T wzl = (T)I.wl[k], wzh = (kl == kh) ? T(0) : (T)(1.0 - I.wl[k]);
        const T *cl = cs + kl * m * m, *ch = cs + kh * m * m;
        const T gl = w00 * cl[jl * m + il] + w10 * cl[jl * m + ih] + w01 * cl[jh * m + il] + w11 * cl[jh * m + ih];
        const T gh = w00 * ch[jl * m + il] + w10 * ch[jl * m + ih] + w01 * ch[jh * m + il] + w11 * ch[jh * m + ih];
        if (d[k] >= 0) u[d[k]] = uk[k] + (wzl * gl + wzh * gh);
    }
}

// element-local coarse sums of the fine values the element owns: partial[e*m^3 + t] = sum over the element's owned fine
// points of weight(point -> t) * fine[dof]
template <typename T, int NF>
__global__ __launch_bounds__(kBlock) void lattice_restrict_kernel(T *__restrict__ partial, const T *__restrict__ fine, const int *__restrict__ owner_dof, const LatticeInterp I, int num_elements)
{
    constexpr int LPE = NF * NF;
    constexpr int EPW = kBlock / LPE;
    constexpr int M = NF / 2 + 1; // bound of m
    __shared__ T W[NF][M];            // weight of fine node i for coarse node a
    __shared__ T g[EPW][M][LPE];      // after z: [cz][j][i]
    __shared__ T h[EPW][M][NF][M];    // after x: [cz][j][a]
    const int m = I.m, mc = m * m * m;
    for (int t = threadIdx.x; t < NF * M; t += kBlock)
    {
        const int i = t / M, a = t % M;
        T w = T(0);
        if (a == I.lo[i]) w = (T)I.wl[i];
        if (a == I.hi[i] && I.hi[i] != I.lo[i]) w = (T)(1.0 - I.wl[i]);
        W[i][a] = w;
    }
    const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
    const int e = blockIdx.x * EPW + slot;
    const bool active = slot < EPW && e < num_elements;
    T r[NF];
    if (active)
    {
        const size_t p0 = (size_t)e * NF * NF * NF + lane;
        int d[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) d[k] = __builtin_nontemporal_load(owner_dof + p0 + (size_t)k * LPE);
#pragma unroll
        for (int k = 0; k < NF; k++) r[k] = (d[k] >= 0) ? fine[d[k]] : T(0);
    }
    __syncthreads(); // W
    if (active)
        for (int cz = 0; cz < m; cz++)
        {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < NF; k++) s += W[k][cz] * r[k];
            g[slot][cz][lane] = s;
        }
    __syncthreads();
    if (active)
        for (int t = lane; t < m * NF * m; t += LPE) // (cz, j, a)
        {
            const int a = t % m, j = (t / m) % NF, cz = t / (m * NF);
            T s = T(0);
#pragma unroll
            for (int i = 0; i < NF; i++) s += W[i][a] * g[slot][cz][j * NF + i];
            h[slot][cz][j][a] = s;
        }
    __syncthreads();
    if (active)
        for (int t = lane; t < mc; t += LPE) // (cz, b, a), a fastest: the kept node's index
        {
            const int a = t % m, b = (t / m) % m, cz = t / (m * m);
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NF; j++) s += W[j][b] * h[slot][cz][j][a];
            partial[(size_t)e * mc + t] = s;
        }
}

static bool fill(LatticeInterp &I, int n, int m, const int *lo, const int *hi, const double *wl)
{
    if (n > kMaxN || m > n / 2 + 1 || m < 2) return false;
    I.n = n;
    I.m = m;
    for (int i = 0; i < kMaxN; i++)
    {
        I.lo[i] = (i < n) ? lo[i] : 0;
        I.hi[i] = (i < n) ? hi[i] : 0;
        I.wl[i] = (i < n) ? wl[i] : 1.0;
        if (I.lo[i] < 0 || I.hi[i] >= m || I.lo[i] > I.hi[i]) return false;
    }
    return true;
}

template <typename T, int NF>
static void launch_prolong(T *u, const T *ec, const int *owner_dof, const int *coarse_dof, const LatticeInterp &I, long long E, hipStream_t s)
{
    constexpr int EPW = kBlock / (NF * NF);
    hipLaunchKernelGGL((lattice_prolong_kernel<T, NF>), dim3((unsigned)((E + EPW - 1) / EPW)), dim3(kBlock), 0, s, u, ec, owner_dof, coarse_dof, I, (int)E);
}
template <typename T, int NF>
static void launch_restrict(T *partial, const T *fine, const int *owner_dof, const LatticeInterp &I, long long E, hipStream_t s)
{
    constexpr int EPW = kBlock / (NF * NF);
    hipLaunchKernelGGL((lattice_restrict_kernel<T, NF>), dim3((unsigned)((E + EPW - 1) / EPW)), dim3(kBlock), 0, s, partial, fine, owner_dof, I, (int)E);
}

template <typename T>
static int prolong(T *u, const T *ec, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long E, void *stream)
{
    FDD_REQUIRE(E >= 0 && E < (1LL << 31) / (n * n * n));
    if (E == 0) return 0;
    FDD_REQUIRE(u != nullptr && ec != nullptr && owner_dof != nullptr && coarse_dof != nullptr && lo != nullptr && hi != nullptr && wl != nullptr);
    LatticeInterp I;
    FDD_REQUIRE(fill(I, n, m, lo, hi, wl));
    if (n == 8)
        launch_prolong<T, 8>(u, ec, owner_dof, coarse_dof, I, E, fdd_stream(stream));
    else if (n == 16)
        launch_prolong<T, 16>(u, ec, owner_dof, coarse_dof, I, E, fdd_stream(stream));
    else
    {
        fdd_set_error("fdd_lattice_prolong: %d lattice nodes per direction (8 and 16 are built: fdd_lattice_supported)", n);
        return 1;
    }
    FDD_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int restrict_(T *partial, const T *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long E, void *stream)
{
    FDD_REQUIRE(E >= 0 && E < (1LL << 31) / (n * n * n));
    if (E == 0) return 0;
    FDD_REQUIRE(partial != nullptr && fine != nullptr && owner_dof != nullptr && lo != nullptr && hi != nullptr && wl != nullptr);
    LatticeInterp I;
    FDD_REQUIRE(fill(I, n, m, lo, hi, wl));
    if (n == 8)
        launch_restrict<T, 8>(partial, fine, owner_dof, I, E, fdd_stream(stream));
    else if (n == 16)
        launch_restrict<T, 16>(partial, fine, owner_dof, I, E, fdd_stream(stream));
    else
    {
        fdd_set_error("fdd_lattice_restrict: %d lattice nodes per direction (8 and 16 are built: fdd_lattice_supported)", n);
        return 1;
    }
    FDD_LAUNCH_CHECK();
    return 0;
}
} // namespace

extern "C" {

int fdd_lattice_supported(int n, int m, int *supported)
{
    FDD_REQUIRE(supported != nullptr);
    *supported = ((n == 8 || n == 16) && m >= 2 && m <= n / 2 + 1) ? 1 : 0;
    return 0;
}

int fdd_lattice_prolong(double *u, const double *coarse, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream)
{
    return prolong<double>(u, coarse, owner_dof, coarse_dof, n, m, lo, hi, wl, num_elements, stream);
}

int fdd_lattice_prolong_f32(float *u, const float *coarse, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream)
{
    return prolong<float>(u, coarse, owner_dof, coarse_dof, n, m, lo, hi, wl, num_elements, stream);
}

int fdd_lattice_restrict(double *partial, const double *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream)
{
    return restrict_<double>(partial, fine, owner_dof, n, m, lo, hi, wl, num_elements, stream);
}

int fdd_lattice_restrict_f32(float *partial, const float *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream)
{
    return restrict_<float>(partial, fine, owner_dof, n, m, lo, hi, wl, num_elements, stream);
}

} // extern "C"
