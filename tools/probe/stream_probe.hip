// Development probe (not part of the product): what a persistent 1024-lane workgroup per CU can pull from HBM with
// the access pattern of the matrix-core stiffness kernel (csrc/fdd_stiffness_mfma.hip) -- eight streams of 32 KB per
// element (u, six factors in, Au out), loads issued one element ahead into registers -- with the compute phases
// replaced by nothing, by LDS barriers, or by barriers plus idle cycles.  Prints TB/s per mode.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/_build/stream_probe tools/probe/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do                                                                                 \
    {                                                                                  \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess)                                                          \
        {                                                                              \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct Ptrs
{
    const double *g[6];
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: stream only.  1: five LDS barriers per element.  2: barriers + `idle` x 64 cycles of sleep in three places.
// WIDE: 16-byte accesses (two doubles per lane and access) instead of 8-byte ones.
template <int MODE, bool WIDE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(double *__restrict__ Au, const double *__restrict__ u, Ptrs G, int num_elements, int idle)
{
    constexpr int n3 = 4096;
    constexpr int PTS = n3 / THREADS; // doubles per lane and array
    __shared__ double s[4096 + 64];
    const int tid = threadIdx.x;
    double ru[PTS], rg[6][PTS];
    auto off = [&](int m) -> unsigned { return WIDE ? (unsigned)(2 * tid + (m & 1) + (m >> 1) * 2 * THREADS) : (unsigned)(tid + m * THREADS); };
    auto load_all = [&](const double *p, double (&r)[PTS]) {
        if (WIDE)
        {
#pragma unroll
            for (int m = 0; m < PTS; m += 2)
            {
                const double2 v = *reinterpret_cast<const double2 *>(p + off(m));
                r[m] = v.x;
                r[m + 1] = v.y;
            }
        }
        else
        {
#pragma unroll
            for (int m = 0; m < PTS; m++) r[m] = p[off(m)];
        }
    };
    int e = blockIdx.x;
    if (e < num_elements)
    {
        load_all(u + (size_t)e * n3, ru);
#pragma unroll
        for (int g = 0; g < 6; g++) load_all(G.g[g] + (size_t)e * n3, rg[g]);
    }
    for (; e < num_elements; e += gridDim.x)
    {
        const int en = (e + (int)gridDim.x < num_elements) ? e + (int)gridDim.x : e;
        // "P0": u to LDS, next u requested
#pragma unroll
        for (int m = 0; m < PTS; m++) s[(tid + m * THREADS) & 4095] = ru[m];
        load_all(u + (size_t)en * n3, ru);
        if (MODE >= 1) lds_barrier();
        if (MODE >= 2)
            for (int w = 0; w < idle; w++) __builtin_amdgcn_s_sleep(64);
        if (MODE >= 1) lds_barrier();
        // "P2": factors consumed, next ones requested
        double acc[PTS];
#pragma unroll
        for (int m = 0; m < PTS; m++)
        {
            const double x = s[(tid + m * THREADS + 1) & 4095];
            acc[m] = rg[0][m] * x + rg[1][m] + rg[2][m] * rg[3][m] + rg[4][m] * rg[5][m];
        }
#pragma unroll
        for (int g = 0; g < 6; g++) load_all(G.g[g] + (size_t)en * n3, rg[g]);
        if (MODE >= 1) lds_barrier();
        if (MODE >= 2)
            for (int w = 0; w < idle; w++) __builtin_amdgcn_s_sleep(64);
        if (MODE >= 1) lds_barrier();
        if (MODE >= 2)
            for (int w = 0; w < idle; w++) __builtin_amdgcn_s_sleep(64);
        if (MODE >= 1) lds_barrier();
        // "P4": result out
        double *out = Au + (size_t)e * n3;
        if (WIDE)
        {
#pragma unroll
            for (int m = 0; m < PTS; m += 2) *reinterpret_cast<double2 *>(out + off(m)) = make_double2(acc[m], acc[m + 1]);
        }
        else
        {
#pragma unroll
            for (int m = 0; m < PTS; m++) out[off(m)] = acc[m];
        }
    }
}

// One element per workgroup, nothing persistent, nothing prefetched: several workgroups per CU overlap instead.
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void probe_flat(double *__restrict__ Au, const double *__restrict__ u, Ptrs G, int num_elements)
{
    constexpr int n3 = 4096, PTS = n3 / THREADS;
    const size_t base = (size_t)blockIdx.x * n3;
    const int tid = threadIdx.x;
#pragma unroll 4
    for (int m = 0; m < PTS; m++)
    {
        const size_t p = base + tid + m * THREADS;
        double a;
        if (NT)
            a = __builtin_nontemporal_load(G.g[0] + p) * u[p] + __builtin_nontemporal_load(G.g[1] + p) + __builtin_nontemporal_load(G.g[2] + p) * __builtin_nontemporal_load(G.g[3] + p) +
                __builtin_nontemporal_load(G.g[4] + p) * __builtin_nontemporal_load(G.g[5] + p);
        else
            a = G.g[0][p] * u[p] + G.g[1][p] + G.g[2][p] * G.g[3][p] + G.g[4][p] * G.g[5][p];
        if (NT)
            __builtin_nontemporal_store(a, Au + p);
        else
            Au[p] = a;
    }
}

template <int THREADS, bool NT>
static void run_flat(const char *name, double *Au, const double *u, Ptrs G, int ne)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((probe_flat<THREADS, NT>), dim3(ne), dim3(THREADS), 0, 0, Au, u, G, ne);
    CHECK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe_flat<THREADS, NT>), dim3(ne), dim3(THREADS), 0, 0, Au, u, G, ne);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 64.0 * 4096 * ne;
    printf("%-46s grid %5d x %4d         : %8.1f us  %6.2f TB/s\n", name, ne, THREADS, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
}

template <int MODE, bool WIDE, int THREADS>
static void run(const char *name, double *Au, const double *u, Ptrs G, int ne, int grid, int idle)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((probe<MODE, WIDE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, Au, u, G, ne, idle);
    CHECK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<MODE, WIDE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, Au, u, G, ne, idle);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 64.0 * 4096 * ne;
    printf("%-46s grid %4d x %4d idle %3d: %8.1f us  %6.2f TB/s\n", name, grid, THREADS, idle, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const int ne = 32768;
    const size_t n = (size_t)ne * 4096;
    double *Au, *u;
    Ptrs G;
    CHECK(hipMalloc(&Au, n * 8));
    CHECK(hipMalloc(&u, n * 8));
    CHECK(hipMemset(u, 0, n * 8));
    for (int g = 0; g < 6; g++)
    {
        double *p;
        CHECK(hipMalloc(&p, n * 8));
        CHECK(hipMemset(p, 0, n * 8));
        G.g[g] = p;
    }
    run<0, false, 1024>("stream, 8-byte accesses", Au, u, G, ne, 256, 0);
    run<0, true, 1024>("stream, 16-byte accesses", Au, u, G, ne, 256, 0);
    run<1, false, 1024>("+ five LDS barriers", Au, u, G, ne, 256, 0);
    run<1, true, 1024>("+ five LDS barriers, 16-byte", Au, u, G, ne, 256, 0);
    run<0, false, 1024>("stream, 8-byte, 512 workgroups (2 per CU slot?)", Au, u, G, ne, 512, 0);
    run<0, false, 512>("stream, 512 lanes, 2 workgroups per CU", Au, u, G, ne, 512, 0);
    run<0, false, 512>("stream, 512 lanes, 4 workgroups per CU", Au, u, G, ne, 1024, 0);
    run_flat<256, false>("one element per workgroup", Au, u, G, ne);
    run_flat<256, true>("one element per workgroup, non-temporal", Au, u, G, ne);
    run_flat<512, false>("one element per workgroup", Au, u, G, ne);
    run_flat<512, true>("one element per workgroup, non-temporal", Au, u, G, ne);
    run_flat<1024, false>("one element per workgroup", Au, u, G, ne);
    run_flat<1024, true>("one element per workgroup, non-temporal", Au, u, G, ne);
    // the same flat kernel on an eighth of the elements (C2's footprint, 1.07 GB) and on a quarter
    run_flat<512, true>("non-temporal, 4096 elements (1.07 GB)", Au, u, G, 4096);
    run_flat<512, true>("non-temporal, 8192 elements (2.1 GB)", Au, u, G, 8192);
    run_flat<512, true>("non-temporal, 16384 elements (4.3 GB)", Au, u, G, 16384);
    // is it the footprint of one launch, or the time at full rate?  The 32768 elements as 8 / 4 / 2 back-to-back launches
    for (int pieces : {8, 4, 2})
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        const int per = ne / pieces;
        auto go = [&] {
            for (int k = 0; k < pieces; k++)
            {
                Ptrs H;
                for (int g = 0; g < 6; g++) H.g[g] = G.g[g] + (size_t)k * per * 4096;
                hipLaunchKernelGGL((probe_flat<512, true>), dim3(per), dim3(512), 0, 0, Au + (size_t)k * per * 4096, u + (size_t)k * per * 4096, H, per);
            }
        };
        go();
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 10; i++) go();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("32768 elements as %d launches of %5d, non-temporal              : %8.1f us  %6.2f TB/s\n", pieces, per, ms / 10 * 1e3, 64.0 * 4096 * ne / (ms / 10 * 1e-3) / 1e12);
    }
    // the same eight arrays carved out of one allocation with `pad` bytes between them (separate hipMallocs of exactly
    // 1 GiB put every stream at the same offset modulo 1 GiB)
    for (size_t pad : {(size_t)0, (size_t)4096 * 37, (size_t)(2 << 20) + 4096 * 5, (size_t)(32 << 20) + 256 * 1024 * 3})
    {
        char *big;
        const size_t stride = n * 8 + pad;
        CHECK(hipMalloc(&big, 8 * stride + (1 << 20)));
        CHECK(hipMemset(big, 0, 8 * stride));
        Ptrs H;
        for (int g = 0; g < 6; g++) H.g[g] = (const double *)(big + (size_t)(g + 2) * stride);
        char name[128];
        snprintf(name, sizeof name, "one allocation, %zu KiB between arrays, nt", pad / 1024);
        run_flat<512, true>(name, (double *)big, (const double *)(big + stride), H, ne);
        CHECK(hipFree(big));
    }
    return 0;
}
