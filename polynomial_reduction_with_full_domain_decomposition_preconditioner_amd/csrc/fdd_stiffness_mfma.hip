// High-order (N = 8..15) SEM stiffness Au = D^T G D u on the fp64 matrix cores
// of gfx950 (v_mfma_f64_16x16x4_f64).  Same operator as fdd_stiffness.hip's
// fused kernel (domain.okl:5-98 / subdomain.okl:4-101, both passes, 64 B/point
// of HBM traffic), for the degrees where the scalar kernel is LDS-bound
// (n = N+1 = 16: 128 LDS reads per point, 2.5 TB/s).
//
// The six 1-D contractions of an element are 16x16x16 matrix products
//     Y = Dm * X,   X = a 16x16 view of the element's 16^3 tensor in LDS whose
//                       ROW index is the contracted direction,
// i.e. 4 MFMAs each, 384 per element (6.1k cycles per CU, against ~22k cycles of
// HBM time per element per CU at 6.3 TB/s).  Measured: ~30k cycles per element
// whether its 229 KB come from HBM or from the Infinity Cache, with 4 % of the
// wave cycles waiting on LDS -- the sixteen wavefronts run in lock-step through
// six LDS-only barriers and nothing else is resident on the CU (DESIGN.md
// section 7): 70 % of the HBM roofline.  Degrees below 15 are zero-padded to 16
// (the extra products are exact zeros).
//
// One persistent 1024-lane workgroup per CU (16 wavefronts, wave w owns xy-slab
// k = w and xz-slab j = w), four padded 16^3 arrays in LDS (136 KiB):
//   P0  u(element)            -> sU                       (registers prefetched one element ahead; wave w writes slab k=w)
//   P1  Du_x, Du_y (slab k=w: the wave's own, no workgroup barrier), then Du_z (slab j=w) = D * views(sU) -> sA1, sA2, sA3
//   P2  point-wise G mixing, in place on sA1..3            (G prefetched one element ahead)
//   P3a Au_x + Au_y (slab k=w) = D^T * views(sA1, sA2)    -> sU
//   P3b Au_z (slab j=w)       = D^T * view(sA3)           += sU
//   P4  sU                     -> Au(element)
// Global loads of element e+1 are issued while element e computes.
//
// Numerics: MFMA fuses multiply-add and sums each 16-term contraction in its
// own order, so this kernel is NOT bit-identical to the reference arithmetic;
// it agrees to ~1e-15 * max|Au| (tolerance 1e-12 in tests/).  The top-level
// order (Au_x + Au_y) + Au_z and the G mixing expressions are the reference's.
#include <mutex>

#include "fdd_common.h"

namespace
{

#ifndef FDD_MFMA_UNCOND_PREFETCH
#define FDD_MFMA_UNCOND_PREFETCH 1 // 0: prefetch only when there is a next element (development A/B)
#endif
#ifndef FDD_MFMA_STAGGER
#define FDD_MFMA_STAGGER 0 // > 0: workgroups with an odd index start that many x 8k cycles late (development A/B: de-phase the CUs' load bursts)
#endif
#ifndef FDD_MFMA_TRACE
#define FDD_MFMA_TRACE 0 // 1: wave 0 of workgroup 0 accumulates the cycles spent in each phase and prints them (development)
#endif
#if FDD_MFMA_TRACE
#define FDD_TR(k)                                   \
    do                                              \
    {                                               \
        const unsigned long long c1_ = clock64();   \
        tr_[k] += c1_ - c0_;                        \
        c0_ = c1_;                                  \
    } while (0)
#else
#define FDD_TR(k)
#endif
constexpr int kThreads = 1024;
constexpr int LD = 17;        // padded row of 16
constexpr int PL = 16 * LD;   // plane stride
constexpr int ARR = 16 * PL;  // doubles per padded 16^3 array
constexpr int kPts = 4;       // padded points per lane (4096 / 1024)

typedef double v4f64 __attribute__((ext_vector_type(4)));

struct GPtrs
{
    const double *g[FDD_NUM_GEOM_FACTS];
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// the vector-memory counter, i.e. it would wait for the global loads prefetched
// for the NEXT element and for the stores of the previous one at every phase
// boundary; the phases here exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Ordering of one wave's own LDS traffic (the LDS executes a wave's instructions in order; this keeps the compiler
// from moving accesses across, and waits for the wave's outstanding LDS operations).
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Y = Dm * X for the view X[row][col] = src[base + row*rs + col*cs];
// a[s] = Dm[lane&15][4s + (lane>>4)] (A-operand fragments of the constant matrix).
__device__ __forceinline__ v4f64 tile_product(const double *src, int base, int rs, int cs, const double (&a)[4], int lane)
{
    const int kk = lane >> 4, c = lane & 15;
    double b[4];
#pragma unroll
    for (int s = 0; s < 4; s++) b[s] = src[base + (4 * s + kk) * rs + c * cs];
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// the same with the A-operand fragments read from LDS: sa[s*64 + lane]
__device__ __forceinline__ v4f64 tile_product_lds(const double *src, int base, int rs, int cs, const double *sa, int lane)
{
    const int kk = lane >> 4, c = lane & 15;
    double a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; s++)
    {
        a[s] = sa[s * 64 + lane];
        b[s] = src[base + (4 * s + kk) * rs + c * cs];
    }
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// C/D layout of v_mfma_f64_16x16x4_f64: register r of lane l is Y[(l>>4) + 4r][l&15]
__device__ __forceinline__ void tile_store(double *dst, int base, int rs, int cs, v4f64 y, int lane)
{
    const int kk = lane >> 4, c = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; r++) dst[base + (kk + 4 * r) * rs + c * cs] = y[r];
}

__device__ __forceinline__ void tile_add(double *dst, int base, int rs, int cs, v4f64 y, int lane)
{
    const int kk = lane >> 4, c = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int a = base + (kk + 4 * r) * rs + c * cs;
        dst[a] = dst[a] + y[r];
    }
}

// kGather: u[p] = (*u_scale) * v[point_dof[p]] (0 where the point has no dof), as in fused_stiffness_kernel
// kAffine: the factors of a point are c_f(e) (w_i w_j) w_k, formed in P2 from six numbers per element (G.g[0], element-
// major) and the GLL weights (G.g[1]) instead of prefetched from the six arrays -- 48 of the 64 bytes per point are not
// read (fdd_stiffness.hip, fused_stiffness_kernel_t; an option of this build for affine elements).
template <int n, bool kGather, bool kAffine = false>
__global__ __launch_bounds__(kThreads) void mfma_stiffness_kernel(double *__restrict__ Au, const double *__restrict__ u, const int *__restrict__ point_dof, const double *__restrict__ u_scale, const double *__restrict__ D_hat, GPtrs G, const int *__restrict__ elem_offset, int num_elements, int xcd_window)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sU = smem;
    double *sA1 = sU + ARR;
    double *sA2 = sA1 + ARR;
    double *sA3 = sA2 + ARR;
    double *sDt = sA3 + ARR; // A-operand fragments of D^T, [4][64]: kept out of the register file (see below)

    constexpr int n2 = n * n, n3 = n2 * n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // A-operand fragments of D (pass 1) and D^T (pass 2), zero-padded to 16x16.  Those of D stay in registers; those of
    // D^T are read from LDS at each use: with both in registers (16 VGPRs) the allocation spilled, and every scratch
    // reload waits on the in-order vector-memory counter, i.e. on the next element's prefetched loads and on the stores
    // of the previous one.
    double a_D[4];
    {
        const int row = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            const int p = 4 * s + kk;
            const bool in = (row < n) && (p < n);
            a_D[s] = in ? D_hat[p + row * n] : 0.0;                              // D[row][p]   = D_hat[p + row*n_x] (domain.okl:42)
            if (wave == 0) sDt[s * 64 + lane] = in ? D_hat[row + p * n] : 0.0;  // D^T[row][p] = D_hat[row + p*n_x] (domain.okl:90)
        }
    }
    __syncthreads();

    // A wave owns the xy-slab k = wave: its lanes' points are (i = lane & 15, j = (lane >> 4) + 4m, k = wave), so
    // that what the wave writes in P0 / P2 is what its own x and y products read -- those need no workgroup barrier,
    // only the z products (which cross the slabs) do.  Every index is a lane constant plus m times a literal.
    const int pi = lane & 15, pj0 = lane >> 4;
    const int l0 = pi + pj0 * LD + wave * PL;
    const int g0 = pi + pj0 * n + wave * n2;
    const bool vik = (pi < n) && (wave < n);
#define lidx(m) (l0 + 4 * (m)*LD)
#define goff(m) ((unsigned)(g0 + 4 * (m)*n)) // 32-bit lane offset behind a wave-uniform element base: SGPR base + VGPR offset loads
#define valid(m) (vik && (pj0 + 4 * (m) < n))

    auto elem_base = [&](int e) -> size_t { return elem_offset ? (size_t)elem_offset[e] : (size_t)e * n3; };
    const double uscale = (kGather && u_scale) ? *u_scale : 1.0;
    // kGather: the dof indices run one element ahead of the values they address (rd), so that the value
    // loads of the prefetch never wait on an index load issued in the same phase
    auto load_idx = [&](size_t eb, unsigned off, bool ok) -> int { return ok ? __builtin_nontemporal_load((point_dof + eb) + off) : -1; }; // the index stream is read once
    auto load_val = [&](int d) -> double { // unconditional load on a selected index, then the select
        const double v = u[d < 0 ? 0 : d];
        return (d < 0) ? 0.0 : (u_scale ? uscale * v : v);
    };

#if FDD_MFMA_STAGGER > 0
    if (blockIdx.x & 1)
        for (int w_ = 0; w_ < FDD_MFMA_STAGGER; w_++) __builtin_amdgcn_s_sleep(127);
#endif
    double ru[kPts], rg[FDD_NUM_GEOM_FACTS][kPts];
    int rd[kGather ? kPts : 1];
    // Which elements the persistent workgroups visit together.  Under round-robin dispatch workgroup b sits on XCD b % 8, so
    // in plain order the x-neighbours e and e + 1 -- which share a face whose 256 dofs are the LAST dof of 256 different
    // 120-byte runs of the left neighbour's numbering -- are always on different XCDs and both L2s fetch those lines.  In
    // XCD-windowed order (one window = the grid) the workgroups of an XCD take grid / 8 CONSECUTIVE elements of every sweep.
    int e = fdd_xcd_windowed_block(blockIdx.x, gridDim.x, xcd_window);
    if (e < num_elements)
    {
        const size_t base = elem_base(e);
        if (kGather)
        {
#pragma unroll
            for (int m = 0; m < kPts; m++) rd[m] = load_idx(base, goff(m), valid(m));
        }
        // u first, then the factors: the order the loop issues them in (u of the next element after P0, its factors in
        // P2).  The compiler's wait before P0's LDS writes is the weaker of the two paths into the loop head: with the
        // first element's loads interleaved it came out as vmcnt(0), i.e. every element waited for ALL outstanding
        // vector memory -- the 24 prefetched factor loads and the stores of the element before.
#pragma unroll
        for (int m = 0; m < kPts; m++) ru[m] = kGather ? load_val(rd[kGather ? m : 0]) : (valid(m) ? (u + base)[goff(m)] : 0.0);
        asm volatile("" ::: "memory"); // keep the order
        if (kGather) // the next element's indices before the factors, as in the loop (they are needed right after P0)
        {
            const int en0 = e + (int)gridDim.x;
            const size_t base_n = elem_base(en0 < num_elements ? en0 : e);
#pragma unroll
            for (int m = 0; m < kPts; m++) rd[m] = load_idx(base_n, goff(m), valid(m));
            asm volatile("" ::: "memory");
        }
        if (!kAffine)
        {
#pragma unroll
            for (int m = 0; m < kPts; m++)
#pragma unroll
                for (int g = 0; g < FDD_NUM_GEOM_FACTS; g++) rg[g][m] = valid(m) ? (G.g[g] + base)[goff(m)] : 0.0;
        }
    }
    // kAffine: (w_i w_j) of the lane's four points and w_k of its slab; zero on the padding
    double wij[kAffine ? kPts : 1], wk = 0.0;
    if (kAffine)
    {
#pragma unroll
        for (int m = 0; m < kPts; m++) wij[m] = valid(m) ? G.g[1][pi] * G.g[1][pj0 + 4 * m] : 0.0;
        wk = (wave < n) ? G.g[1][wave] : 0.0;
    }

#if FDD_MFMA_TRACE
    unsigned long long tr_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c0_ = clock64();
    int tr_elems = 0;
#endif
    for (; e < num_elements; e += gridDim.x)
    {
#if FDD_MFMA_TRACE
        tr_elems++;
        c0_ = clock64();
#endif
        const size_t base = elem_base(e);
        const int en = e + gridDim.x;
        const bool more = en < num_elements;
        // The prefetches below are issued on EVERY iteration (the last one re-reads its own element): with them under
        // `if (more)` the number of loads in flight at each wait differs between the paths, and the compiler has to wait
        // for the weakest case -- vmcnt(0) in P2, which also drains the loads just issued for the next element.
        const size_t base_n = elem_base(more ? en : e);

        // P0: element -> LDS (each lane rewrites the slots it read in P4: no barrier needed before)
#pragma unroll
        for (int m = 0; m < kPts; m++) sU[lidx(m)] = ru[m];
        wave_lds_sync(); // the slab is this wave's own
        if (FDD_MFMA_UNCOND_PREFETCH || more)
        {
#pragma unroll
            for (int m = 0; m < kPts; m++) ru[m] = kGather ? load_val(rd[kGather ? m : 0]) : (valid(m) ? (u + base_n)[goff(m)] : 0.0);
            if (kGather)
            {
                const int enn = en + (int)gridDim.x;
                const size_t base_nn = elem_base(enn < num_elements ? enn : (more ? en : e));
#pragma unroll
                for (int m = 0; m < kPts; m++) rd[m] = load_idx(base_nn, goff(m), valid(m));
            }
        }
        FDD_TR(0);

        // P1: first derivatives; x and y on the wave's own slab, then (every slab written) z across the slabs
        {
            v4f64 y = tile_product(sU, wave * PL, 1, LD, a_D, lane); // x: rows i, cols j, slab k = wave
            tile_store(sA1, wave * PL, 1, LD, y, lane);
            y = tile_product(sU, wave * PL, LD, 1, a_D, lane);       // y: rows j, cols i, slab k = wave
            tile_store(sA2, wave * PL, LD, 1, y, lane);
            FDD_TR(1);
            lds_barrier();
            FDD_TR(2);
            y = tile_product(sU, wave * LD, PL, 1, a_D, lane);       // z: rows k, cols i, slab j = wave
            tile_store(sA3, wave * LD, PL, 1, y, lane);
            FDD_TR(3);
        }
        lds_barrier();
        FDD_TR(4);

        // P2: geometric factors, point-wise, in place (domain.okl:47-49)
        if (kAffine)
        {
            double cf[FDD_NUM_GEOM_FACTS];
#pragma unroll
            for (int g = 0; g < FDD_NUM_GEOM_FACTS; g++) cf[g] = G.g[0][(size_t)e * FDD_NUM_GEOM_FACTS + g]; // wave-uniform address
#pragma unroll
            for (int m = 0; m < kPts; m++)
            {
                const double w3 = wij[kAffine ? m : 0] * wk;
#pragma unroll
                for (int g = 0; g < FDD_NUM_GEOM_FACTS; g++) rg[g][m] = cf[g] * w3;
            }
        }
#pragma unroll
        for (int m = 0; m < kPts; m++)
        {
            const double Du_1 = sA1[lidx(m)], Du_2 = sA2[lidx(m)], Du_3 = sA3[lidx(m)];
            sA1[lidx(m)] = rg[0][m] * Du_1 + rg[3][m] * Du_2 + rg[4][m] * Du_3;
            sA2[lidx(m)] = rg[3][m] * Du_1 + rg[1][m] * Du_2 + rg[5][m] * Du_3;
            sA3[lidx(m)] = rg[4][m] * Du_1 + rg[5][m] * Du_2 + rg[2][m] * Du_3;
        }
        if (!kAffine && (FDD_MFMA_UNCOND_PREFETCH || more))
        {
#pragma unroll
            for (int m = 0; m < kPts; m++)
#pragma unroll
                for (int g = 0; g < FDD_NUM_GEOM_FACTS; g++) rg[g][m] = valid(m) ? (G.g[g] + base_n)[goff(m)] : 0.0;
        }
        wave_lds_sync(); // P2 wrote this wave's slab of sA1 / sA2, which is all P3a reads
        FDD_TR(5);

        // P3a: Au_x then + Au_y on this wave's xy-slab (sU is free: the z products that read it are behind a barrier)
        {
            v4f64 y = tile_product_lds(sA1, wave * PL, 1, LD, sDt, lane);
            tile_store(sU, wave * PL, 1, LD, y, lane);
            wave_lds_sync();
            y = tile_product_lds(sA2, wave * PL, LD, 1, sDt, lane);
            tile_add(sU, wave * PL, LD, 1, y, lane);
        }
        FDD_TR(6);
        lds_barrier(); // every slab of sA3 (P2) and of sU (P3a) complete
        FDD_TR(7);

        // P3b: + Au_z on this wave's xz-slab
        {
            v4f64 y = tile_product_lds(sA3, wave * LD, PL, 1, sDt, lane);
            tile_add(sU, wave * LD, PL, 1, y, lane);
        }
        FDD_TR(8);
        lds_barrier();
        FDD_TR(9);

        // P4: LDS -> global
#pragma unroll
        for (int m = 0; m < kPts; m++)
            if (valid(m)) (Au + base)[goff(m)] = sU[lidx(m)]; // default cache policy: non-temporal accesses measured 4 % slower here
        FDD_TR(10);
    }
#if FDD_MFMA_TRACE
    if (blockIdx.x == 0 && tid == 0)
        printf("mfma trace (cycles per element, wave 0 of workgroup 0, %d elements): P0+issue %llu | xy %llu | wait %llu | z %llu | wait %llu | P2+issue %llu | P3a %llu | wait %llu | P3b %llu | wait %llu | P4 %llu\n", tr_elems,
               tr_[0] / tr_elems, tr_[1] / tr_elems, tr_[2] / tr_elems, tr_[3] / tr_elems, tr_[4] / tr_elems, tr_[5] / tr_elems, tr_[6] / tr_elems, tr_[7] / tr_elems, tr_[8] / tr_elems, tr_[9] / tr_elems, tr_[10] / tr_elems);
#endif
#undef lidx
#undef goff
#undef valid
}

template <int n, bool kAffine = false>
int launch_mfma(double *Au, const double *u, const int *point_dof, const double *u_scale, const double *D_hat, const GPtrs &G, const int *elem_offset, int num_elements, void *stream)
{
    const size_t lds = (4 * (size_t)ARR + 4 * 64) * sizeof(double);
    // once per kernel instance, whichever host thread gets here first
    static std::once_flag configured;
    hipError_t attr_a = hipSuccess, attr_b = hipSuccess;
    std::call_once(configured, [&] {
        attr_a = hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_stiffness_kernel<n, false, kAffine>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_b = hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_stiffness_kernel<n, true, kAffine>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    FDD_HIP_CHECK(attr_a);
    FDD_HIP_CHECK(attr_b);
    const int grid = num_elements < FDD_CU_COUNT ? num_elements : FDD_CU_COUNT;
    // consecutive elements per XCD and sweep (fdd_xcd_windowed_block; 0 = dispatch order, -1 = grid / 8: every XCD's
    // workgroups take one run of consecutive elements per sweep)
    // C3: L2 fetches of the gather form 10.03 -> 9.38 GB per launch with 8 (1.12 -> 1.05 x algorithmic), 1.74 -> 1.62 ms with grid / 8
    static const int xcd_env = fdd_env_int("FDD_TUNE_MFMA_XCD_WINDOW", -1);
    const int xcd_window = xcd_env < 0 ? grid / FDD_NUM_XCD : xcd_env;
    if (point_dof)
        hipLaunchKernelGGL((mfma_stiffness_kernel<n, true, kAffine>), dim3(grid), dim3(kThreads), lds, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements, xcd_window);
    else
        hipLaunchKernelGGL((mfma_stiffness_kernel<n, false, kAffine>), dim3(grid), dim3(kThreads), lds, fdd_stream(stream), Au, u, point_dof, u_scale, D_hat, G, elem_offset, num_elements, xcd_window);
    FDD_LAUNCH_CHECK();
    return 0;
}

int mfma_dispatch_affine(double *Au, const double *u, const int *point_dof, const double *u_scale, const double *D_hat, const double *elem_factors, const double *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && elem_factors != nullptr && gll_weights != nullptr && Au != u);
    GPtrs g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++) g.g[k] = nullptr;
    g.g[0] = elem_factors;
    g.g[1] = gll_weights;
    switch (poly_degree + 1)
    {
    case 9: return launch_mfma<9, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 10: return launch_mfma<10, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 11: return launch_mfma<11, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 12: return launch_mfma<12, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 13: return launch_mfma<13, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 14: return launch_mfma<14, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 15: return launch_mfma<15, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 16: return launch_mfma<16, true>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    default:
        fdd_set_error("fp64-MFMA stiffness kernel supports poly_degree 8..15, got %d", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

int mfma_dispatch(double *Au, const double *u, const int *point_dof, const double *u_scale, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(num_elements >= 0);
    if (num_elements == 0) return 0;
    FDD_REQUIRE(Au != nullptr && u != nullptr && D_hat != nullptr && G != nullptr && Au != u);
    GPtrs g;
    for (int k = 0; k < FDD_NUM_GEOM_FACTS; k++)
    {
        FDD_REQUIRE(G[k] != nullptr);
        g.g[k] = G[k];
    }
    switch (poly_degree + 1)
    {
    case 9: return launch_mfma<9>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 10: return launch_mfma<10>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 11: return launch_mfma<11>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 12: return launch_mfma<12>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 13: return launch_mfma<13>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 14: return launch_mfma<14>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 15: return launch_mfma<15>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    case 16: return launch_mfma<16>(Au, u, point_dof, u_scale, D_hat, g, elem_offset, num_elements, stream);
    default:
        fdd_set_error("fp64-MFMA stiffness kernel supports poly_degree 8..15, got %d", poly_degree);
        return FDD_ERR_UNSUPPORTED;
    }
}

} // namespace

extern "C" {

int fdd_stiffness_matrix_mfma(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return mfma_dispatch(Au, u, nullptr, nullptr, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

int fdd_stiffness_matrix_mfma_affine(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *elem_factors, const double *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    return mfma_dispatch_affine(Au, v, point_dof, v_scale_dev, D_hat, elem_factors, gll_weights, elem_offset, num_elements, poly_degree, stream);
}

int fdd_stiffness_matrix_mfma_gather(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream)
{
    FDD_REQUIRE(point_dof != nullptr);
    return mfma_dispatch(Au, v, point_dof, v_scale_dev, D_hat, G, elem_offset, num_elements, poly_degree, stream);
}

} // extern "C"
