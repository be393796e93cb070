// Pack / unpack of a rank's boundary-node prefix into the dense interface-slot
// vector that is all-reduced over RCCL: the device-resident replacement of
// gslib's gs(gs_add) on the prefix (reference domain.tpp:590-594, which stages
// it D2H -> MPI -> H2D).  Indexed 8-B accesses; the prefix side is coalesced.
#include "fdd_common.h"

namespace
{
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void pack_kernel(double *__restrict__ slots, const int *__restrict__ slot_of, const double *__restrict__ prefix, int n)
{
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) slots[slot_of[i]] = prefix[i];
}

__global__ __launch_bounds__(kBlock) void unpack_kernel(double *__restrict__ prefix, const double *__restrict__ slots, const int *__restrict__ slot_of, int n)
{
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) prefix[i] = slots[slot_of[i]];
}
// Neighbour form of the same exchange (xGMI is point-to-point: every pair of GPUs of a node has its own link, so a rank
// sends each peer only the nodes the two share -- at C4 three faces of 225^2 nodes instead of a dense 3 * 449^2-slot vector
// carried around a ring).  One buffer per rank, `ncomp` (1 or 2) vectors interleaved:
//   [ own copies of the boundary prefix | the parts that go to the peers | the parts that arrive from them ]
// gather fills the first two regions from the prefix; sum adds a node's contributions in the order its row lists them
// (ascending rank, the own copy in its place: every sharer forms the same sum in the same order).
template <int NC>
__global__ __launch_bounds__(kBlock) void gather_kernel(double *__restrict__ buf, const int *__restrict__ index, int n, const double *__restrict__ a, const double *__restrict__ b)
{
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    {
        const int p = index[i];
        buf[(size_t)i * NC] = a[p];
        if (NC == 2) buf[(size_t)i * NC + 1] = b[p];
    }
}

template <int NC>
__global__ __launch_bounds__(kBlock) void sum_kernel(double *__restrict__ a, double *__restrict__ b, const int *__restrict__ ptr, const int *__restrict__ col, int rows, const double *__restrict__ buf)
{
    const int stride = gridDim.x * kBlock;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += stride)
    {
        double sa = 0.0, sb = 0.0;
        for (int k = ptr[r]; k < ptr[r + 1]; k++)
        {
            const size_t c = (size_t)col[k] * NC;
            sa += buf[c];
            if (NC == 2) sb += buf[c + 1];
        }
        a[r] = sa;
        if (NC == 2) b[r] = sb;
    }
}
} // namespace

extern "C" {

int fdd_interface_gather(double *buf, const int *index, int n, const double *a, const double *b, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(buf != nullptr && index != nullptr && a != nullptr);
    if (b)
        hipLaunchKernelGGL(gather_kernel<2>, dim3(fdd_stream_grid(n, kBlock)), dim3(kBlock), 0, fdd_stream(stream), buf, index, n, a, b);
    else
        hipLaunchKernelGGL(gather_kernel<1>, dim3(fdd_stream_grid(n, kBlock)), dim3(kBlock), 0, fdd_stream(stream), buf, index, n, a, b);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_interface_sum(double *a, double *b, const int *ptr, const int *col, int rows, const double *buf, void *stream)
{
    FDD_REQUIRE(rows >= 0);
    if (rows == 0) return 0;
    FDD_REQUIRE(a != nullptr && ptr != nullptr && col != nullptr && buf != nullptr);
    if (b)
        hipLaunchKernelGGL(sum_kernel<2>, dim3(fdd_stream_grid(rows, kBlock)), dim3(kBlock), 0, fdd_stream(stream), a, b, ptr, col, rows, buf);
    else
        hipLaunchKernelGGL(sum_kernel<1>, dim3(fdd_stream_grid(rows, kBlock)), dim3(kBlock), 0, fdd_stream(stream), a, b, ptr, col, rows, buf);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_interface_pack(double *slots, const int *slot_of, const double *prefix, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(slots != nullptr && slot_of != nullptr && prefix != nullptr);
    hipLaunchKernelGGL(pack_kernel, dim3(fdd_stream_grid(n, kBlock)), dim3(kBlock), 0, fdd_stream(stream), slots, slot_of, prefix, n);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_interface_unpack(double *prefix, const double *slots, const int *slot_of, int n, void *stream)
{
    FDD_REQUIRE(n >= 0);
    if (n == 0) return 0;
    FDD_REQUIRE(prefix != nullptr && slots != nullptr && slot_of != nullptr);
    hipLaunchKernelGGL(unpack_kernel, dim3(fdd_stream_grid(n, kBlock)), dim3(kBlock), 0, fdd_stream(stream), prefix, slots, slot_of, n);
    FDD_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
