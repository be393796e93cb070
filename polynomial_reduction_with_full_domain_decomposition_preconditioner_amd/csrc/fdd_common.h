// Internal helpers shared by the gfx950 kernel translation units.
#ifndef FDD_COMMON_H
#define FDD_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstdint>
#include <cstdio>

#include "fdd_hip.h"

#define FDD_WAVE 64
#define FDD_CU_COUNT 256

void fdd_set_error(const char *fmt, ...);

#define FDD_HIP_CHECK(expr)                                                                         \
    do                                                                                              \
    {                                                                                               \
        hipError_t fdd_err_ = (expr);                                                               \
        if (fdd_err_ != hipSuccess)                                                                 \
        {                                                                                           \
            fdd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(fdd_err_), __FILE__, __LINE__); \
            return (int)fdd_err_;                                                                   \
        }                                                                                           \
    } while (0)

#define FDD_REQUIRE(cond)                                                              \
    do                                                                                 \
    {                                                                                  \
        if (!(cond))                                                                   \
        {                                                                              \
            fdd_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);  \
            return FDD_ERR_INVALID_ARGUMENT;                                           \
        }                                                                              \
    } while (0)

// after a kernel launch
#define FDD_LAUNCH_CHECK() FDD_HIP_CHECK(hipGetLastError())

static inline hipStream_t fdd_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a memory-bound grid-stride kernel: enough workgroups to fill the
// 256 CUs several times over, capped so the tail is a loop, not more blocks.
static inline int fdd_stream_grid(long long work_items, int block, int max_blocks = FDD_REDUCE_MAX_BLOCKS)
{
    long long blocks = (work_items + block - 1) / block;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}

static inline bool fdd_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// XCD-aware workgroup order for indexed (gather/scatter) kernels.  Workgroups
// are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has its
// own 4 MiB L2): remapping so that every XCD walks one contiguous eighth of
// the index range keeps the lines shared by neighbouring elements (the
// partner points of an interface node) in ONE L2, so scattered 8-byte stores
// merge there into full lines and partner loads hit instead of being fetched
// by two XCDs.  Bijective for any grid size; speed only, never correctness.
// Row sum of a CSR row by ONE lane with the loads of up to kRowChunk entries
// in flight at once.  A plain `for (j = j0; j < j1; j++) s += val[j]*u[col[j]]`
// is a chain of dependent loads (col -> u) per entry, and a wavefront runs as
// many serial round trips as its longest row: with 1-8 entries per row (the
// boolean gather matrices) that, not bandwidth, sets the time.  Here every
// lane first issues all its col (and val) loads of a chunk, then all its u
// gathers, then adds the products in column order -- the reference's
// summation order, so the result is unchanged.  Lanes past the end of their
// row issue nothing.
#define FDD_ROW_CHUNK 8
template <bool UNIT>
__device__ __forceinline__ double fdd_row_sum(const int *__restrict__ A_col, const double *__restrict__ A_val, const double *u, int j0, int j1)
{
    double s = 0.0;
    for (int jb = j0; jb < j1; jb += FDD_ROW_CHUNK)
    {
        int c[FDD_ROW_CHUNK];
        double a[FDD_ROW_CHUNK];
        double x[FDD_ROW_CHUNK];
#pragma unroll
        for (int k = 0; k < FDD_ROW_CHUNK; k++)
        {
            const bool on = jb + k < j1;
            c[k] = on ? A_col[jb + k] : 0;
            a[k] = (UNIT || !on) ? 1.0 : A_val[jb + k];
        }
#pragma unroll
        for (int k = 0; k < FDD_ROW_CHUNK; k++) x[k] = (jb + k < j1) ? u[c[k]] : 0.0;
#pragma unroll
        for (int k = 0; k < FDD_ROW_CHUNK; k++)
            if (jb + k < j1) s += a[k] * x[k];
    }
    return s;
}

// The same for NPT rows per lane: all col loads of a CH-wide slice of all NPT
// rows are issued, then all u gathers, then each row adds its products in
// column order.  NPT*CH gathers in flight per lane is what it takes to cover
// the HBM latency with three dependent round trips (ptr -> col -> u) per row.
// Rows to skip are passed with j0 == j1.
template <int NPT, int CH, bool UNIT>
__device__ __forceinline__ void fdd_multi_row_sum(const int *__restrict__ A_col, const double *__restrict__ A_val, const double *u, const int (&j0)[NPT], const int (&j1)[NPT], double (&s)[NPT])
{
    int len_max = 0;
#pragma unroll
    for (int r = 0; r < NPT; r++)
    {
        s[r] = 0.0;
        len_max = (j1[r] - j0[r] > len_max) ? j1[r] - j0[r] : len_max;
    }
    for (int off = 0; off < len_max; off += CH)
    {
        int c[NPT][CH];
        double a[NPT][CH];
        double x[NPT][CH];
#pragma unroll
        for (int r = 0; r < NPT; r++)
#pragma unroll
            for (int k = 0; k < CH; k++)
            {
                const int j = j0[r] + off + k;
                const bool on = j < j1[r];
                c[r][k] = on ? A_col[j] : 0;
                a[r][k] = (UNIT || !on) ? 1.0 : A_val[j];
            }
#pragma unroll
        for (int r = 0; r < NPT; r++)
#pragma unroll
            for (int k = 0; k < CH; k++) x[r][k] = (j0[r] + off + k < j1[r]) ? u[c[r][k]] : 0.0;
#pragma unroll
        for (int r = 0; r < NPT; r++)
#pragma unroll
            for (int k = 0; k < CH; k++)
                if (j0[r] + off + k < j1[r]) s[r] += a[r][k] * x[r][k];
    }
}

#define FDD_NUM_XCD 8
__device__ __forceinline__ int fdd_xcd_chunked_block(int bid, int nblocks)
{
    const int per = nblocks / FDD_NUM_XCD;
    const int rem = nblocks % FDD_NUM_XCD;
    const int xcd = bid % FDD_NUM_XCD;
    const int idx = bid / FDD_NUM_XCD;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// XCD-windowed order: the chunked order above sends the eight XCDs to eight far-apart eighths of the index range (eight
// distant streams: measured slower on the row-block SpMVs although it removes the double fetches).  Here the range is cut
// into windows of 8*C consecutive blocks and, inside a window, XCD x (= bid % 8 under round-robin dispatch) takes the C
// consecutive blocks [x*C, (x+1)*C), one after the other as its workgroups are dispatched: neighbours in index space run
// on ONE XCD close in time (the sectors / lines they share are fetched into one L2, once), while all eight XCDs stay inside
// the same window of memory.  C = nblocks / 8 is the chunked order.  A ragged last window keeps the plain order.
// Bijective; speed only, never correctness (HIP promises no block -> XCD placement).
__device__ __forceinline__ int fdd_xcd_windowed_block(int bid, int nblocks, int C)
{
    if (C <= 1) return bid;
    const int W = FDD_NUM_XCD * C;
    const int win = bid / W;
    if (win >= nblocks / W) return bid;
    const int in = bid - win * W;
    return win * W + (in % FDD_NUM_XCD) * C + in / FDD_NUM_XCD;
}

// integer tuning knob from the environment (read once by the caller: `static const`)
static inline int fdd_env_int(const char *name, int fallback)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : fallback;
}

#endif
