// runtime.hip — device memory, streams, timers, device properties behind the C ABI.
// Replaces what the reference does inline through the CUDA runtime: CudaVector's
// cudaMalloc/cudaMemcpy/cudaFree (reference include/vector.h:119-169),
// cudaDeviceSynchronize (src/test.cu:77,89) and printGPUProperties (src/utils.cpp:5-15).
#include <cstdint>
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace rmd {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    // The error has been reported through the return value: take it out of the runtime's per-thread "last error", or the next
    // hipGetLastError() of anybody in the process (torch checks after each of its launches) blames an innocent call for it.
    (void)hipGetLastError();
    return (int)e;
}

int check_frame_geometry(const rmd_svgf_frame_desc* f)
{
    if (!f) return fail(RMD_E_NULL, "frame descriptor is NULL");
    if (f->width <= 0 || f->height <= 0) return fail(RMD_E_SHAPE, "frame %dx%d is not positive", f->width, f->height);
    if ((long long)f->width * (long long)f->height > 0x7fffffffLL)
        return fail(RMD_E_SHAPE, "frame %dx%d overflows int pixel indices", f->width, f->height);
    if (f->buf_rows <= 0 || f->buf_row0 < 0 || f->buf_row0 + f->buf_rows > f->height)
        return fail(RMD_E_ROWS, "buffer rows [%d,%d) outside frame height %d", f->buf_row0, f->buf_row0 + f->buf_rows, f->height);
    return RMD_OK;
}

int check_rows_in_buffer(const rmd_svgf_frame_desc* f, int lo, int hi, const char* what)
{
    if (lo < 0) lo = 0;
    if (hi > f->height) hi = f->height;
    if (lo < f->buf_row0 || hi > f->buf_row0 + f->buf_rows)
        return fail(RMD_E_ROWS, "%s needs rows [%d,%d) but the planes hold [%d,%d)", what, lo, hi,
                    f->buf_row0, f->buf_row0 + f->buf_rows);
    return RMD_OK;
}

struct Timer {
    hipEvent_t start, stop;
};

int current_device()
{
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

// Work counters of rmd_svgf_frame_atrous_next: a block of 64 words per device, handed out round robin (a counter is in use from
// the call's memset to the end of its last launch; 64 calls later it is long done)
unsigned* side_counter_on_device()
{
    static std::mutex mu;
    static unsigned* block[kMaxDevices] = {};
    static unsigned next[kMaxDevices] = {};
    const int d = current_device();
    if (d < 0 || d >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!block[d]) {
        void* q = nullptr;
        if (hipMalloc(&q, 64 * sizeof(unsigned)) != hipSuccess) return nullptr;
        block[d] = static_cast<unsigned*>(q);
    }
    return block[d] + (next[d]++ % 64u);
}

int device_cus()
{
    static int cus[kMaxDevices] = {};
    const int d = current_device();
    if (d < 0 || d >= kMaxDevices) return kCus;
    if (cus[d] == 0) {
        int n = 0;
        cus[d] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) ? n : kCus;
    }
    return cus[d];
}

#ifdef RMD_EXPERIMENTS
int tuning_env(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

bool first_use_on_device(const void* key)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, unsigned>> seen;      // (key, bit mask of devices)
    const int d = current_device();
    const unsigned bit = 1u << ((d < 0 ? 0 : d) % 32);
    std::lock_guard<std::mutex> lock(mu);
    for (auto& kv : seen)
        if (kv.first == key) {
            if (kv.second & bit) return false;
            kv.second |= bit;
            return true;
        }
    seen.emplace_back(key, bit);
    return true;
}

}  // namespace rmd

using namespace rmd;

extern "C" {

const char* rmd_last_error_string(void) { return g_err; }
#ifdef RMD_EXPERIMENTS
const char* rmd_version(void) { return "raymarchdenoisercuda_amd 0.4 (gfx950, experiments build)"; }
int rmd_has_experiments(void) { return 1; }
#else
const char* rmd_version(void) { return "raymarchdenoisercuda_amd 0.4 (gfx950)"; }
int rmd_has_experiments(void) { return 0; }
#endif

int rmd_malloc(void** ptr, size_t bytes)
{
    if (!ptr) return fail(RMD_E_NULL, "rmd_malloc: ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) return RMD_OK;
    RMD_HIP(hipMalloc(ptr, bytes));
    return RMD_OK;
}

int rmd_free(void* ptr)
{
    if (!ptr) return RMD_OK;
    RMD_HIP(hipFree(ptr));
    return RMD_OK;
}

int rmd_memset(void* ptr, int value, size_t bytes, void* stream)
{
    if (!ptr && bytes) return fail(RMD_E_NULL, "rmd_memset: ptr is NULL");
    RMD_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return RMD_OK;
}

int rmd_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
    if ((!dst || !src) && bytes) return fail(RMD_E_NULL, "rmd_memcpy_h2d: NULL pointer");
    RMD_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return RMD_OK;
}

int rmd_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
    if ((!dst || !src) && bytes) return fail(RMD_E_NULL, "rmd_memcpy_d2h: NULL pointer");
    RMD_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RMD_OK;
}

int rmd_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    if ((!dst || !src) && bytes) return fail(RMD_E_NULL, "rmd_memcpy_d2d: NULL pointer");
    RMD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return RMD_OK;
}

int rmd_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream)
{
    if ((!dst || !src) && bytes) return fail(RMD_E_NULL, "rmd_memcpy_h2d_async: NULL pointer");
    RMD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return RMD_OK;
}

int rmd_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream)
{
    if ((!dst || !src) && bytes) return fail(RMD_E_NULL, "rmd_memcpy_d2h_async: NULL pointer");
    RMD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return RMD_OK;
}

int rmd_host_alloc_pinned(void** ptr, size_t bytes)
{
    if (!ptr) return fail(RMD_E_NULL, "rmd_host_alloc_pinned: ptr is NULL");
    RMD_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return RMD_OK;
}

int rmd_host_free_pinned(void* ptr)
{
    if (!ptr) return RMD_OK;
    RMD_HIP(hipHostFree(ptr));
    return RMD_OK;
}

int rmd_stream_create(void** stream)
{
    if (!stream) return fail(RMD_E_NULL, "rmd_stream_create: stream is NULL");
    hipStream_t s;
    RMD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return RMD_OK;
}

int rmd_graph_capture_begin(void* stream)
{
    if (!stream) return fail(RMD_E_NULL, "rmd_graph_capture_begin: the NULL stream cannot be captured (rmd_stream_create one)");
    RMD_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return RMD_OK;
}

int rmd_graph_capture_end(void* stream, void** graph)
{
    if (!stream || !graph) return fail(RMD_E_NULL, "rmd_graph_capture_end: stream / graph is NULL");
    *graph = nullptr;
    hipGraph_t g = nullptr;
    RMD_HIP(hipStreamEndCapture(as_stream(stream), &g));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return hip_fail(e, "rmd_graph_capture_end: hipGraphInstantiate");
    *graph = exec;
    return RMD_OK;
}

int rmd_graph_launch(void* graph, void* stream)
{
    if (!graph) return fail(RMD_E_NULL, "rmd_graph_launch: graph is NULL");
    RMD_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph), as_stream(stream)));
    return RMD_OK;
}

int rmd_graph_destroy(void* graph)
{
    if (!graph) return RMD_OK;
    RMD_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph)));
    return RMD_OK;
}

int rmd_stream_destroy(void* stream)
{
    if (!stream) return RMD_OK;
    RMD_HIP(hipStreamDestroy(as_stream(stream)));
    return RMD_OK;
}

int rmd_stream_sync(void* stream)
{
    RMD_HIP(hipStreamSynchronize(as_stream(stream)));
    return RMD_OK;
}

int rmd_event_create(void** event)
{
    if (!event) return fail(RMD_E_NULL, "rmd_event_create: event is NULL");
    hipEvent_t e;
    RMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return RMD_OK;
}

int rmd_event_destroy(void* event)
{
    if (!event) return RMD_OK;
    RMD_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return RMD_OK;
}

int rmd_event_synchronize(void* event)
{
    if (!event) return fail(RMD_E_NULL, "rmd_event_synchronize: event is NULL");
    RMD_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return RMD_OK;
}

int rmd_event_record(void* event, void* stream)
{
    if (!event) return fail(RMD_E_NULL, "rmd_event_record: event is NULL");
    RMD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return RMD_OK;
}

int rmd_stream_wait_event(void* stream, void* event)
{
    if (!event) return fail(RMD_E_NULL, "rmd_stream_wait_event: event is NULL");
    RMD_HIP(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return RMD_OK;
}

int rmd_device_sync(void)
{
    RMD_HIP(hipDeviceSynchronize());
    return RMD_OK;
}

int rmd_device_count(int* count)
{
    if (!count) return fail(RMD_E_NULL, "rmd_device_count: count is NULL");
    RMD_HIP(hipGetDeviceCount(count));
    return RMD_OK;
}

int rmd_set_device(int device)
{
    RMD_HIP(hipSetDevice(device));
    return RMD_OK;
}

int rmd_print_device_properties(void)
{
    int dev = 0;
    RMD_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RMD_HIP(hipGetDeviceProperties(&prop, dev));
    // same facts the reference prints (name, shared memory, registers, warp size), restated for
    // CDNA: wavefront width, CU count, LDS per workgroup / per CU, L2, memory clock x bus.
    printf("Device name: %s (%s)\n", prop.name, prop.gcnArchName);
    printf("Compute units: %d\n", prop.multiProcessorCount);
    printf("Wavefront size: %d\n", prop.warpSize);
    printf("LDS per workgroup: %.1f KB\n", prop.sharedMemPerBlock / 1024.0);
    printf("LDS per compute unit: %.1f KB\n", prop.maxSharedMemoryPerMultiProcessor / 1024.0);
    printf("Registers per workgroup: %d\n", prop.regsPerBlock);
    printf("L2 cache: %.1f MB\n", prop.l2CacheSize / (1024.0 * 1024.0));
    printf("Global memory: %.1f GB\n", prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    printf("Memory clock: %d kHz, bus width %d bits\n\n", prop.memoryClockRate, prop.memoryBusWidth);
    return RMD_OK;
}

int rmd_timer_create(void** timer)
{
    if (!timer) return fail(RMD_E_NULL, "rmd_timer_create: timer is NULL");
    Timer* t = new Timer();
    hipError_t e = hipEventCreate(&t->start);
    if (e == hipSuccess) e = hipEventCreate(&t->stop);
    if (e != hipSuccess) { delete t; return hip_fail(e, "hipEventCreate"); }
    *timer = t;
    return RMD_OK;
}

int rmd_timer_destroy(void* timer)
{
    if (!timer) return RMD_OK;
    Timer* t = static_cast<Timer*>(timer);
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return RMD_OK;
}

int rmd_timer_start(void* timer, void* stream)
{
    if (!timer) return fail(RMD_E_NULL, "rmd_timer_start: timer is NULL");
    RMD_HIP(hipEventRecord(static_cast<Timer*>(timer)->start, as_stream(stream)));
    return RMD_OK;
}

int rmd_timer_stop(void* timer, void* stream)
{
    if (!timer) return fail(RMD_E_NULL, "rmd_timer_stop: timer is NULL");
    RMD_HIP(hipEventRecord(static_cast<Timer*>(timer)->stop, as_stream(stream)));
    return RMD_OK;
}

int rmd_timer_elapsed_ms(void* timer, float* ms)
{
    if (!timer || !ms) return fail(RMD_E_NULL, "rmd_timer_elapsed_ms: NULL argument");
    Timer* t = static_cast<Timer*>(timer);
    RMD_HIP(hipEventSynchronize(t->stop));
    RMD_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return RMD_OK;
}

}  // extern "C"
