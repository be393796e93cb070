// Degree-tree restriction J^T (x) J^T (x) J^T of the FDD preconditioner
// (subdomain.okl:284-366; driven from Subdomain::tree_operator,
// subdomain.tpp:4576-4609).
//
// (1) the reference's three launches with global intermediates (any dim);
// (2) a fused 3-D kernel: one element per workgroup pass, the n_f^3 tensor is
//     read once into LDS, contracted along x, y, z inside LDS
//     (n_f^3 -> n_c n_f^2 -> n_c^2 n_f -> n_c^3) and only the n_c^3 result is
//     written: 8*(n_f^3 + n_c^3) bytes per element, HBM-bound.
//
// Each output is the reference's sum over l = 0..n_f-1 from 0.0 in that order,
// so with -ffp-contract=off both forms are bit-identical to OCCA-Serial.
#include "fdd_common.h"

namespace
{

constexpr int kBlock = 256;

template <int DIM>
__global__ __launch_bounds__(kBlock) void restriction_1_kernel(double *__restrict__ Ju, const double *__restrict__ J_cf, const double *__restrict__ u, int num_points, int n_f, int n_c)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int num_elem_points_fine = (DIM == 2) ? n_f * n_f : n_f * n_f * n_f;
    const int num_elem_points_coarse = (DIM == 2) ? n_f * n_c : n_f * n_f * n_c;
    const int e = idx / num_elem_points_coarse;
    const int v = idx % num_elem_points_coarse;
    const double *ue = u + (size_t)e * num_elem_points_fine;
    double *Je = Ju + (size_t)e * num_elem_points_coarse;
    double Ju_ij = 0.0;

    if (DIM == 2)
    {
        const int i = v % n_f;
        const int j = v / n_f;
        for (int k = 0; k < n_f; k++) Ju_ij += J_cf[j + k * n_c] * ue[i + k * n_f];
        Je[i + j * n_f] = Ju_ij;
    }
    else
    {
        const int i = v % n_c;
        const int j = (v / n_c) % n_f;
        const int k = v / (n_c * n_f);
        for (int l = 0; l < n_f; l++) Ju_ij += J_cf[i + l * n_c] * ue[l + j * n_f + k * (n_f * n_f)];
        Je[i + j * n_c + k * (n_c * n_f)] = Ju_ij;
    }
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void restriction_2_kernel(double *__restrict__ Ju, const double *__restrict__ J_cf, const double *__restrict__ u, int num_points, int n_f, int n_c)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int num_elem_points_fine = (DIM == 2) ? n_f * n_c : n_f * n_f * n_c;
    const int num_elem_points_coarse = (DIM == 2) ? n_c * n_c : n_f * n_c * n_c;
    const int e = idx / num_elem_points_coarse;
    const int v = idx % num_elem_points_coarse;
    const double *ue = u + (size_t)e * num_elem_points_fine;
    double *Je = Ju + (size_t)e * num_elem_points_coarse;
    double Ju_ij = 0.0;

    if (DIM == 2)
    {
        const int i = v % n_c;
        const int j = v / n_c;
        for (int k = 0; k < n_f; k++) Ju_ij += ue[j * n_f + k] * J_cf[k * n_c + i];
        Je[i + j * n_c] = Ju_ij;
    }
    else
    {
        const int i = v % n_c;
        const int j = (v / n_c) % n_c;
        const int k = v / (n_c * n_c);
        for (int l = 0; l < n_f; l++) Ju_ij += J_cf[j + l * n_c] * ue[i + l * n_c + k * (n_c * n_f)];
        Je[i + j * n_c + k * (n_c * n_c)] = Ju_ij;
    }
}

__global__ __launch_bounds__(kBlock) void restriction_3_kernel(double *__restrict__ Ju, const double *__restrict__ J_cf, const double *__restrict__ u, int num_points, int n_f, int n_c)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= num_points) return;

    const int num_elem_points_fine = n_f * n_c * n_c;
    const int num_elem_points_coarse = n_c * n_c * n_c;
    const int e = idx / num_elem_points_coarse;
    const int v = idx % num_elem_points_coarse;
    const double *ue = u + (size_t)e * num_elem_points_fine;
    double Ju_ij = 0.0;

    const int i = v % n_c;
    const int j = (v / n_c) % n_c;
    const int k = v / (n_c * n_c);
    for (int l = 0; l < n_f; l++) Ju_ij += J_cf[k + l * n_c] * ue[i + j * n_c + l * (n_c * n_c)];
    Ju[(size_t)e * num_elem_points_coarse + i + j * n_c + k * (n_c * n_c)] = Ju_ij;
}

// Fused: grid-stride over elements, dynamic LDS = (n_f*n_c + n_f^3 + n_c*n_f^2) doubles.
// Buffer A holds u (n_f^3) and later the y-contracted tensor (n_c^2 n_f);
// buffer B holds the x-contracted tensor (n_c n_f^2).
__global__ __launch_bounds__(kBlock) void restriction_fused_kernel(double *__restrict__ u_c, const double *__restrict__ J_cf, const double *__restrict__ u_f, int num_elements, int n_f, int n_c)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sJ = smem;                  // n_f * n_c
    double *sA = sJ + n_f * n_c;        // n_f^3
    double *sB = sA + n_f * n_f * n_f;  // n_c * n_f^2

    const int nf2 = n_f * n_f;
    const int nf3 = nf2 * n_f;
    const int nc2 = n_c * n_c;
    const int nc3 = nc2 * n_c;
    const int s1 = n_c * nf2; // after x
    const int s2 = nc2 * n_f; // after y

    for (int t = threadIdx.x; t < n_f * n_c; t += kBlock) sJ[t] = J_cf[t];

    for (int e = blockIdx.x; e < num_elements; e += gridDim.x)
    {
        const double *ue = u_f + (size_t)e * nf3;
        __syncthreads(); // previous element's readers of sA are done; sJ visible
        for (int t = threadIdx.x; t < nf3; t += kBlock) sA[t] = ue[t];
        __syncthreads();

        // restriction_1: (l, j, k) -> (i, j, k), i < n_c
        for (int v = threadIdx.x; v < s1; v += kBlock)
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_f;
            const int k = v / (n_c * n_f);
            double s = 0.0;
            for (int l = 0; l < n_f; l++) s += sJ[i + l * n_c] * sA[l + j * n_f + k * nf2];
            sB[i + j * n_c + k * (n_c * n_f)] = s;
        }
        __syncthreads();

        // restriction_2: (i, l, k) -> (i, j, k), j < n_c
        for (int v = threadIdx.x; v < s2; v += kBlock)
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_c;
            const int k = v / nc2;
            double s = 0.0;
            for (int l = 0; l < n_f; l++) s += sJ[j + l * n_c] * sB[i + l * n_c + k * (n_c * n_f)];
            sA[i + j * n_c + k * nc2] = s;
        }
        __syncthreads();

        // restriction_3: (i, j, l) -> (i, j, k), k < n_c
        for (int v = threadIdx.x; v < nc3; v += kBlock)
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_c;
            const int k = v / nc2;
            double s = 0.0;
            for (int l = 0; l < n_f; l++) s += sJ[k + l * n_c] * sA[i + j * n_c + l * nc2];
            u_c[(size_t)e * nc3 + v] = s;
        }
    }
}

// Fine elements of at most 512 points (n_f <= 8: every level pair below degree 8, e.g. 7 -> 1 of config C2): one
// WAVEFRONT per element, four elements per workgroup in flight, no workgroup barrier inside the loop (the LDS executes
// a wave's instructions in order), the element's 8 loads per lane issued together with non-temporal loads.  Same
// sums in the same order as restriction_1/2/3.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NF, int NC> // compile-time sizes (index arithmetic by constants, unrolled sums); 0, 0: sizes from the arguments
__global__ __launch_bounds__(kBlock) void restriction_wave_kernel(double *__restrict__ u_c, const double *__restrict__ J_cf, const double *__restrict__ u_f, int num_elements, int n_f_arg, int n_c_arg)
{
    const int n_f = NF > 0 ? NF : n_f_arg;
    const int n_c = NC > 0 ? NC : n_c_arg;
    constexpr int kWaves = kBlock / FDD_WAVE;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nf2 = n_f * n_f;
    const int nf3 = nf2 * n_f;
    const int nc2 = n_c * n_c;
    const int nc3 = nc2 * n_c;
    const int s1 = n_c * nf2; // after x
    const int s2 = nc2 * n_f; // after y
    const int wave = threadIdx.x / FDD_WAVE, lane = threadIdx.x % FDD_WAVE;
    double *sJ = smem;                                   // n_f * n_c
    double *sA = sJ + n_f * n_c + wave * (nf3 + s1);     // this wave's u, later the y-contracted tensor
    double *sB = sA + nf3;                               // this wave's x-contracted tensor

    for (int t = threadIdx.x; t < n_f * n_c; t += kBlock) sJ[t] = J_cf[t];
    __syncthreads();

    for (int e = blockIdx.x * kWaves + wave; e < num_elements; e += gridDim.x * kWaves)
    {
        const double *ue = u_f + (size_t)e * nf3;
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
            const int t = lane + q * FDD_WAVE;
            r[q] = __builtin_nontemporal_load(ue + (t < nf3 ? t : 0)); // slots past the element re-read its first point
        }
        wave_lds_sync(); // the previous element's readers of sA are done
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
            const int t = lane + q * FDD_WAVE;
            if (t < nf3) sA[t] = r[q];
        }
        wave_lds_sync();

        for (int v = lane; v < s1; v += FDD_WAVE) // restriction_1: (l, j, k) -> (i, j, k), i < n_c
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_f;
            const int k = v / (n_c * n_f);
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < n_f; l++) s += sJ[i + l * n_c] * sA[l + j * n_f + k * nf2];
            sB[i + j * n_c + k * (n_c * n_f)] = s;
        }
        wave_lds_sync();

        for (int v = lane; v < s2; v += FDD_WAVE) // restriction_2: (i, l, k) -> (i, j, k), j < n_c
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_c;
            const int k = v / nc2;
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < n_f; l++) s += sJ[j + l * n_c] * sB[i + l * n_c + k * (n_c * n_f)];
            sA[i + j * n_c + k * nc2] = s;
        }
        wave_lds_sync();

        for (int v = lane; v < nc3; v += FDD_WAVE) // restriction_3: (i, j, l) -> (i, j, k), k < n_c
        {
            const int i = v % n_c;
            const int j = (v / n_c) % n_c;
            const int k = v / nc2;
            double s = 0.0;
            for (int l = 0; l < n_f; l++) s += sJ[k + l * n_c] * sA[i + j * n_c + l * nc2];
            u_c[(size_t)e * nc3 + v] = s;
        }
    }
}

// 2-D form of the fused restriction (the `dim == 2` branches of subdomain.okl:284-344 in one launch): an element's n_f^2
// values go to LDS once, the y contraction (restriction_1's 2-D branch contracts the SECOND index) leaves n_f x n_c
// there, the x contraction writes the n_c^2 result.  Same statements per output, sums in ascending k: bit-identical to
// the two-launch form.  kBlock / n_f^2 elements per workgroup (n_f <= 16).
__global__ __launch_bounds__(kBlock) void restriction_2d_kernel(double *__restrict__ u_c, const double *__restrict__ J_cf, const double *__restrict__ u_f, int num_elements, int n_f, int n_c)
{
    extern __shared__ double lds2[];
    const int nf2 = n_f * n_f, nfc = n_f * n_c, nc2 = n_c * n_c;
    const int epb = kBlock / nf2;
    double *sJ = lds2;                      // n_f x n_c
    double *sU = sJ + nfc;                  // epb x n_f^2
    double *sT = sU + (size_t)epb * nf2;    // epb x (n_f x n_c)
    for (int t = threadIdx.x; t < nfc; t += kBlock) sJ[t] = J_cf[t];
    for (int g = blockIdx.x; g * epb < num_elements; g += gridDim.x)
    {
        const int e_loc = threadIdx.x / nf2, v = threadIdx.x - e_loc * nf2;
        const int e = g * epb + e_loc;
        const bool on = e_loc < epb and e < num_elements;
        __syncthreads(); // sJ staged; previous group's sU / sT consumed
        if (on) sU[e_loc * nf2 + v] = u_f[(size_t)e * nf2 + v];
        __syncthreads();
        if (on)
            for (int o = v; o < nfc; o += nf2)
            {
                // restriction_1, dim == 2: Ju(i, j) = sum_k J[j + k n_c] u(i, k), i < n_f, j < n_c
                const int i = o % n_f, j = o / n_f;
                double acc = 0.0;
                for (int k = 0; k < n_f; k++) acc += sJ[j + k * n_c] * sU[e_loc * nf2 + (i + k * n_f)];
                sT[e_loc * nfc + o] = acc;
            }
        __syncthreads();
        if (on)
            for (int o = v; o < nc2; o += nf2)
            {
                // restriction_2, dim == 2: Ju(i, j) = sum_k T(k, j) J[k n_c + i], i, j < n_c
                const int i = o % n_c, j = o / n_c;
                double acc = 0.0;
                for (int k = 0; k < n_f; k++) acc += sT[e_loc * nfc + (j * n_f + k)] * sJ[k * n_c + i];
                u_c[(size_t)e * nc2 + o] = acc;
            }
    }
}

} // namespace

extern "C" {

int fdd_sub_restriction_2d(double *u_c, const double *J_cf, const double *u_f, int num_elements, int n_f, int n_c, void *stream)
{
    FDD_REQUIRE(num_elements >= 0 && n_f >= 1 && n_c >= 1 && n_c <= n_f && n_f <= 16);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(u_c != nullptr && J_cf != nullptr && u_f != nullptr);
    const int epb = kBlock / (n_f * n_f);
    const int groups = (num_elements + epb - 1) / epb;
    const int grid = groups < 16 * FDD_CU_COUNT ? groups : 16 * FDD_CU_COUNT;
    const size_t lds = sizeof(double) * ((size_t)n_f * n_c + (size_t)epb * ((size_t)n_f * n_f + (size_t)n_f * n_c));
    hipLaunchKernelGGL(restriction_2d_kernel, dim3(grid), dim3(kBlock), lds, fdd_stream(stream), u_c, J_cf, u_f, num_elements, n_f, n_c);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_sub_restriction_1(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && n_f >= 1 && n_c >= 1 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(Ju != nullptr && J_cf != nullptr && u != nullptr);
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(restriction_1_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Ju, J_cf, u, num_points, n_f, n_c);
    else
        hipLaunchKernelGGL(restriction_1_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Ju, J_cf, u, num_points, n_f, n_c);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_sub_restriction_2(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && n_f >= 1 && n_c >= 1 && (dim == 2 || dim == 3));
    if (num_points == 0) return 0;
    FDD_REQUIRE(Ju != nullptr && J_cf != nullptr && u != nullptr);
    const int grid = (num_points + kBlock - 1) / kBlock;
    if (dim == 2)
        hipLaunchKernelGGL(restriction_2_kernel<2>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Ju, J_cf, u, num_points, n_f, n_c);
    else
        hipLaunchKernelGGL(restriction_2_kernel<3>, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Ju, J_cf, u, num_points, n_f, n_c);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_sub_restriction_3(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, void *stream)
{
    FDD_REQUIRE(num_points >= 0 && n_f >= 1 && n_c >= 1);
    if (num_points == 0) return 0;
    FDD_REQUIRE(Ju != nullptr && J_cf != nullptr && u != nullptr);
    const int grid = (num_points + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(restriction_3_kernel, dim3(grid), dim3(kBlock), 0, fdd_stream(stream), Ju, J_cf, u, num_points, n_f, n_c);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_sub_restriction(double *u_c, const double *J_cf, const double *u_f, int num_elements, int n_f, int n_c, void *stream)
{
    FDD_REQUIRE(num_elements >= 0 && n_f >= 1 && n_c >= 1 && n_c <= n_f && n_f <= 16);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(u_c != nullptr && J_cf != nullptr && u_f != nullptr);
    static const int wave_form = fdd_env_int("FDD_TUNE_RESTRICTION_WAVE", 1);
    if (n_f <= 8 && wave_form)
    {
        constexpr int kWaves = kBlock / FDD_WAVE;
        const size_t lds_w = sizeof(double) * ((size_t)n_f * n_c + (size_t)kWaves * ((size_t)n_f * n_f * n_f + (size_t)n_c * n_f * n_f));
        const int groups = (num_elements + kWaves - 1) / kWaves;
        const int grid_w = groups < 16 * FDD_CU_COUNT ? groups : 16 * FDD_CU_COUNT;
#define FDD_RESTRICT_CASE(NF_, NC_) \
    case NF_ * 16 + NC_: hipLaunchKernelGGL((restriction_wave_kernel<NF_, NC_>), dim3(grid_w), dim3(kBlock), lds_w, fdd_stream(stream), u_c, J_cf, u_f, num_elements, n_f, n_c); break;
        switch (n_f * 16 + n_c)
        {
            // the level pairs of the reference's degree lists N, N-r, ..., 1 below degree 8
            FDD_RESTRICT_CASE(8, 2)
            FDD_RESTRICT_CASE(8, 6)
            FDD_RESTRICT_CASE(8, 5)
            FDD_RESTRICT_CASE(8, 4)
            FDD_RESTRICT_CASE(8, 7)
            FDD_RESTRICT_CASE(7, 2)
            FDD_RESTRICT_CASE(7, 5)
            FDD_RESTRICT_CASE(6, 2)
            FDD_RESTRICT_CASE(6, 4)
            FDD_RESTRICT_CASE(5, 2)
            FDD_RESTRICT_CASE(5, 3)
            FDD_RESTRICT_CASE(4, 2)
            FDD_RESTRICT_CASE(3, 2)
        default: hipLaunchKernelGGL((restriction_wave_kernel<0, 0>), dim3(grid_w), dim3(kBlock), lds_w, fdd_stream(stream), u_c, J_cf, u_f, num_elements, n_f, n_c);
        }
#undef FDD_RESTRICT_CASE
        FDD_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = sizeof(double) * ((size_t)n_f * n_c + (size_t)n_f * n_f * n_f + (size_t)n_c * n_f * n_f);
    const int grid = num_elements < 8 * FDD_CU_COUNT ? num_elements : 8 * FDD_CU_COUNT;
    if (lds > 48 * 1024)
        FDD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(restriction_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(restriction_fused_kernel, dim3(grid), dim3(kBlock), lds, fdd_stream(stream), u_c, J_cf, u_f, num_elements, n_f, n_c);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
