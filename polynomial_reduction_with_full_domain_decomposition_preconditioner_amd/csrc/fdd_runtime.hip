// Runtime entries of the C-ABI: device, memory, streams, events.  Replaces
// occa::device / occa::memory (reference config.hpp:52 and the usage inventory
// in SURVEY.md section 8(b)).
#include "fdd_common.h"

#include <cstdarg>
#include <cstring>

static thread_local char g_last_error[512] = "";

void fdd_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" {

const char *fdd_version(void) { return "fdd_hip 0.1 (gfx950)"; }

const char *fdd_last_error(void) { return g_last_error; }

int fdd_device_count(int *count)
{
    FDD_REQUIRE(count != nullptr);
    FDD_HIP_CHECK(hipGetDeviceCount(count));
    return 0;
}

int fdd_set_device(int device)
{
    FDD_HIP_CHECK(hipSetDevice(device));
    return 0;
}

int fdd_get_device(int *device)
{
    FDD_REQUIRE(device != nullptr);
    FDD_HIP_CHECK(hipGetDevice(device));
    return 0;
}

int fdd_device_name(char *buf, size_t buf_len)
{
    FDD_REQUIRE(buf != nullptr && buf_len > 0);
    int dev = 0;
    FDD_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FDD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int fdd_malloc(void **ptr, size_t bytes)
{
    FDD_REQUIRE(ptr != nullptr);
    *ptr = nullptr;
    if (bytes == 0) return 0;
    FDD_HIP_CHECK(hipMalloc(ptr, bytes));
    return 0;
}

int fdd_free(void *ptr)
{
    if (ptr == nullptr) return 0;
    FDD_HIP_CHECK(hipFree(ptr));
    return 0;
}

int fdd_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    FDD_REQUIRE(dst != nullptr && src != nullptr);
    FDD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, fdd_stream(stream)));
    FDD_HIP_CHECK(hipStreamSynchronize(fdd_stream(stream)));
    return 0;
}

int fdd_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    FDD_REQUIRE(dst != nullptr && src != nullptr);
    FDD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, fdd_stream(stream)));
    FDD_HIP_CHECK(hipStreamSynchronize(fdd_stream(stream)));
    return 0;
}

// Device -> host copy of a handful of reduction results (at most 4 KiB) through a
// pinned staging buffer owned by the library: a pageable destination costs an
// extra runtime-internal staging hop per call, and the solvers fetch scalars
// several times per iteration.
int fdd_fetch_scalars(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    FDD_REQUIRE(dst != nullptr && src != nullptr && bytes <= 4096);
    static thread_local void *staging = nullptr;
    if (staging == nullptr) FDD_HIP_CHECK(hipHostMalloc(&staging, 4096, hipHostMallocDefault));
    FDD_HIP_CHECK(hipMemcpyAsync(staging, src, bytes, hipMemcpyDeviceToHost, fdd_stream(stream)));
    FDD_HIP_CHECK(hipStreamSynchronize(fdd_stream(stream)));
    memcpy(dst, staging, bytes);
    return 0;
}

int fdd_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    FDD_REQUIRE(dst != nullptr && src != nullptr);
    FDD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, fdd_stream(stream)));
    return 0;
}

int fdd_memset(void *dst, int value, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    FDD_REQUIRE(dst != nullptr);
    FDD_HIP_CHECK(hipMemsetAsync(dst, value, bytes, fdd_stream(stream)));
    return 0;
}

int fdd_stream_create(void **stream)
{
    FDD_REQUIRE(stream != nullptr);
    hipStream_t s;
    FDD_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = reinterpret_cast<void *>(s);
    return 0;
}

int fdd_stream_destroy(void *stream)
{
    if (stream == nullptr) return 0;
    FDD_HIP_CHECK(hipStreamDestroy(fdd_stream(stream)));
    return 0;
}

int fdd_stream_sync(void *stream)
{
    FDD_HIP_CHECK(hipStreamSynchronize(fdd_stream(stream)));
    return 0;
}

int fdd_device_sync(void)
{
    FDD_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

// hipGraph capture of a launch sequence on a stream: the reference captures the
// down-leg and up-leg of the AMG V-cycle into CUDA graphs and replays them per
// application (subdomain.tpp:3644-3704, 4021, 4113).
int fdd_graph_begin_capture(void *stream)
{
    FDD_REQUIRE(stream != nullptr); // the default stream cannot be captured
    FDD_HIP_CHECK(hipStreamBeginCapture(fdd_stream(stream), hipStreamCaptureModeThreadLocal));
    return 0;
}

int fdd_graph_end_capture(void *stream, void **graph_exec)
{
    FDD_REQUIRE(graph_exec != nullptr);
    hipGraph_t graph = nullptr;
    FDD_HIP_CHECK(hipStreamEndCapture(fdd_stream(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (err != hipSuccess)
    {
        fdd_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(err));
        return (int)err;
    }
    *graph_exec = reinterpret_cast<void *>(exec);
    return 0;
}

int fdd_graph_launch(void *graph_exec, void *stream)
{
    FDD_REQUIRE(graph_exec != nullptr);
    FDD_HIP_CHECK(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), fdd_stream(stream)));
    return 0;
}

int fdd_graph_destroy(void *graph_exec)
{
    if (graph_exec == nullptr) return 0;
    FDD_HIP_CHECK(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return 0;
}

int fdd_event_create(void **event)
{
    FDD_REQUIRE(event != nullptr);
    hipEvent_t e;
    FDD_HIP_CHECK(hipEventCreate(&e));
    *event = reinterpret_cast<void *>(e);
    return 0;
}

int fdd_event_destroy(void *event)
{
    if (event == nullptr) return 0;
    FDD_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return 0;
}

int fdd_event_record(void *event, void *stream)
{
    FDD_REQUIRE(event != nullptr);
    FDD_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), fdd_stream(stream)));
    return 0;
}

int fdd_event_elapsed_ms(float *ms, void *start, void *stop)
{
    FDD_REQUIRE(ms != nullptr && start != nullptr && stop != nullptr);
    FDD_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    FDD_HIP_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return 0;
}

} // extern "C"
