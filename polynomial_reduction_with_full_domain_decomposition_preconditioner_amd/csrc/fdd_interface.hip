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
} // namespace

extern "C" {

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
