// Internal helpers shared by the gfx950 kernel translation units.
#ifndef FDD_COMMON_H
#define FDD_COMMON_H

#include <hip/hip_runtime.h>

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

#endif
