// Runtime services of libgko_cdna4: error reporting, device query, memory and
// stream management.  These are the C-ABI equivalents of the HipExecutor
// member functions that Ginkgo stubs in core/device_hooks/hip_hooks.cpp:21-252
// and implements for its own backend in hip/base/executor.hip.cpp.
#include <atomic>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include <unistd.h>

#include "common.hpp"

namespace gkoc {

static thread_local char g_last_error[1024] = "";

void set_last_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    set_last_error("%s:%d: %s failed: %s (%d)", file, line, what,
                   hipGetErrorString(e), static_cast<int>(e));
    // clear the sticky "last error" so later launch checks start clean
    (void)hipGetLastError();
    return static_cast<int>(e) > 0 ? static_cast<int>(e) : GKOC_E_NO_DEVICE;
}

const device_props& current_device_props()
{
    static device_props cache[64];
    static bool have[64] = {};
    static std::mutex mtx;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> g(mtx);
    if (!have[dev]) {
        int cu = 256;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount,
                                  dev) != hipSuccess) {
            cu = 256;
        }
        cache[dev].num_cu = cu;
        cache[dev].num_xcd = 8;
        have[dev] = true;
    }
    return cache[dev];
}

// (device, stream) -> a zeroed device word; a slab of 1024 words per device, handed out once per stream
int stream_ticket(hipStream_t st, unsigned** word)
{
    struct slab {
        unsigned* base = nullptr;
        int used = 0;
    };
    static std::mutex mtx;
    static slab slabs[64];
    static thread_local struct {
        int dev;
        hipStream_t st;
        unsigned* word;
    } last = {-1, nullptr, nullptr};
    int dev = 0;
    GKOC_HIP(hipGetDevice(&dev));
    if (last.word && last.dev == dev && last.st == st) {
        *word = last.word;
        return GKOC_OK;
    }
    GKOC_REQUIRE(dev >= 0 && dev < 64, GKOC_E_NOT_SUPPORTED, "device id above 63");
    static std::vector<std::pair<std::pair<int, hipStream_t>, unsigned*>> table;
    std::lock_guard<std::mutex> g(mtx);
    for (const auto& e : table) {
        if (e.first.first == dev && e.first.second == st) {
            last = {dev, st, e.second};
            *word = e.second;
            return GKOC_OK;
        }
    }
    slab& sl = slabs[dev];
    if (!sl.base) {
        GKOC_HIP(hipMalloc(reinterpret_cast<void**>(&sl.base), 1024 * 64));
        GKOC_HIP(hipMemset(sl.base, 0, 1024 * 64));
    }
    // one word per 64-byte line; a process with more than 1024 streams shares the last one's word
    // with later streams (their reductions then must not run at the same time: not supported)
    GKOC_REQUIRE(sl.used < 1024, GKOC_E_NOT_SUPPORTED, "more than 1024 streams use one-kernel reductions");
    unsigned* w = sl.base + size_t(sl.used++) * 16;
    table.push_back({{dev, st}, w});
    last = {dev, st, w};
    *word = w;
    return GKOC_OK;
}

static int64_t g_tune[tune_num_keys] = {};
static bool g_tune_set[tune_num_keys] = {};

// (common.hpp) what a wave behind a gate word pays before its first read
static std::atomic<int> g_gate_fence_policy{0};
int gate_fence_policy() { return g_gate_fence_policy.load(std::memory_order_relaxed); }
void gate_fence_policy_set(int policy) { g_gate_fence_policy.store(policy < 0 ? 0 : policy > 2 ? 2 : policy); }

// process-wide tuning switches: gkoc_tune_set, else the environment variable
// GKOC_TUNE_<n>, else the default chosen by measurement
int64_t tune_value(int key)
{
    static const int64_t defaults[tune_num_keys] = {0, 0, 0, 2, 1, 0, 0, 0, 100, 0, 1, 0, 1, 0, 0, 0, 0, 1};   // measured: the XCD-contiguous order loses 1-8 %
    if (key < 0 || key >= tune_num_keys) return 0;
    if (!g_tune_set[key]) {
        char name[32];
        snprintf(name, sizeof(name), "GKOC_TUNE_%d", key);
        const char* e = getenv(name);
        g_tune[key] = e ? atoll(e) : defaults[key];
        g_tune_set[key] = true;
    }
    return g_tune[key];
}

__global__ void small_copy_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int words)
{
    if (int(threadIdx.x) < words) dst[threadIdx.x] = src[threadIdx.x];
}

}  // namespace gkoc

using namespace gkoc;

extern "C" {

const char* gkoc_last_error(void) { return g_last_error; }

int gkoc_version(void) { return GKOC_VERSION_MAJOR * 100 + GKOC_VERSION_MINOR; }

int gkoc_get_num_devices(int* count)
{
    GKOC_REQUIRE(count, GKOC_E_INVALID, "count == NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return GKOC_OK;
}

int gkoc_get_device_info(int device_id, gkoc_device_info* info)
{
    GKOC_REQUIRE(info, GKOC_E_INVALID, "info == NULL");
    hipDeviceProp_t p;
    GKOC_HIP(hipGetDeviceProperties(&p, device_id));
    std::memset(info, 0, sizeof(*info));
    info->device_id = device_id;
    info->num_cu = p.multiProcessorCount;
    info->wave_size = p.warpSize;
    info->num_xcd = 8;
    info->max_threads_per_block = p.maxThreadsPerBlock;
    info->major = p.major;
    info->minor = p.minor;
    info->lds_bytes_per_cu = static_cast<int32_t>(p.maxSharedMemoryPerMultiProcessor);
    info->hbm_bytes = static_cast<int64_t>(p.totalGlobalMem);
    std::strncpy(info->arch, p.gcnArchName, sizeof(info->arch) - 1);
    return GKOC_OK;
}

int gkoc_set_device(int device_id)
{
    GKOC_HIP(hipSetDevice(device_id));
    return GKOC_OK;
}

int gkoc_get_device(int* device_id)
{
    GKOC_REQUIRE(device_id, GKOC_E_INVALID, "device_id == NULL");
    GKOC_HIP(hipGetDevice(device_id));
    return GKOC_OK;
}

int gkoc_event_create(gkoc_event_t* e)
{
    GKOC_REQUIRE(e, GKOC_E_INVALID, "e == NULL");
    hipEvent_t ev;
    GKOC_HIP(hipEventCreate(&ev));
    *e = ev;
    return GKOC_OK;
}

int gkoc_event_destroy(gkoc_event_t e)
{
    if (e) GKOC_HIP(hipEventDestroy(static_cast<hipEvent_t>(e)));
    return GKOC_OK;
}

int gkoc_event_record(gkoc_event_t e, gkoc_stream_t s)
{
    GKOC_HIP(hipEventRecord(static_cast<hipEvent_t>(e), as_stream(s)));
    return GKOC_OK;
}

int gkoc_event_synchronize(gkoc_event_t e)
{
    GKOC_HIP(hipEventSynchronize(static_cast<hipEvent_t>(e)));
    return GKOC_OK;
}

int gkoc_event_elapsed_ns(gkoc_event_t start, gkoc_event_t stop, int64_t* ns)
{
    GKOC_REQUIRE(ns, GKOC_E_INVALID, "ns == NULL");
    float ms = 0;
    GKOC_HIP(hipEventElapsedTime(&ms, static_cast<hipEvent_t>(start),
                                 static_cast<hipEvent_t>(stop)));
    *ns = static_cast<int64_t>(double(ms) * 1e6);
    return GKOC_OK;
}

int gkoc_stream_wait_event(gkoc_stream_t s, gkoc_event_t e)
{
    GKOC_HIP(hipStreamWaitEvent(as_stream(s), static_cast<hipEvent_t>(e), 0));
    return GKOC_OK;
}

// ---- hipGraph: capture what is enqueued on a stream, replay it
int gkoc_stream_begin_capture(gkoc_stream_t s)
{
    GKOC_REQUIRE(s != nullptr, GKOC_E_INVALID,
                 "capture needs an explicit stream (not the NULL stream)");
    GKOC_HIP(hipStreamBeginCapture(as_stream(s), hipStreamCaptureModeThreadLocal));
    return GKOC_OK;
}

int gkoc_stream_end_capture(gkoc_stream_t s, gkoc_graph_t* graph)
{
    GKOC_REQUIRE(graph, GKOC_E_INVALID, "graph == NULL");
    *graph = nullptr;
    hipGraph_t g = nullptr;
    GKOC_HIP(hipStreamEndCapture(as_stream(s), &g));
    hipGraphExec_t e = nullptr;
    const hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    GKOC_HIP(err);
    *graph = e;
    return GKOC_OK;
}

int gkoc_graph_launch(gkoc_graph_t graph, gkoc_stream_t s)
{
    GKOC_REQUIRE(graph, GKOC_E_INVALID, "graph == NULL");
    GKOC_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph), as_stream(s)));
    return GKOC_OK;
}

int gkoc_graph_destroy(gkoc_graph_t graph)
{
    if (graph) GKOC_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph)));
    return GKOC_OK;
}

int gkoc_malloc(void** ptr, size_t bytes)
{
    GKOC_REQUIRE(ptr, GKOC_E_INVALID, "ptr == NULL");
    return arena_malloc(ptr, bytes, GKOC_MEM_AUTO);
}

int gkoc_malloc_host(void** ptr, size_t bytes)
{
    GKOC_REQUIRE(ptr, GKOC_E_INVALID, "ptr == NULL");
    *ptr = nullptr;
    if (bytes == 0) return GKOC_OK;
    GKOC_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return GKOC_OK;
}

int gkoc_free_host(void* ptr)
{
    if (ptr) GKOC_HIP(hipHostFree(ptr));
    return GKOC_OK;
}

int gkoc_malloc_managed(void** ptr, size_t bytes, unsigned int flags)
{
    GKOC_REQUIRE(ptr, GKOC_E_INVALID, "ptr == NULL");
    GKOC_REQUIRE(flags == hipMemAttachGlobal || flags == hipMemAttachHost,
                 GKOC_E_INVALID, "flags must be 1 (attach global) or 2 (attach host)");
    *ptr = nullptr;
    if (bytes == 0) return GKOC_OK;
    GKOC_HIP(hipMallocManaged(ptr, bytes, flags));
    return GKOC_OK;
}

int gkoc_free(void* ptr)
{
    gkoc::csr_long_rows_forget(ptr);     // what csr::spmv remembers about a matrix at this address (csr_spmv.hip)
    return arena_free(ptr);
}

int gkoc_tune_set(int key, int64_t value)
{
    GKOC_REQUIRE(key >= 0 && key < tune_num_keys, GKOC_E_INVALID, "unknown tuning key");
    g_tune[key] = value;
    g_tune_set[key] = true;
    return GKOC_OK;
}

int gkoc_gate_fence_policy(int set, int* now)
{
    GKOC_REQUIRE(set >= -1 && set <= 2, GKOC_E_INVALID, "policy must be -1 (query), 0, 1 or 2");
    if (set >= 0) gate_fence_policy_set(set);
    if (now) *now = gate_fence_policy();
    return GKOC_OK;
}

int gkoc_tune_get(int key, int64_t* value)
{
    GKOC_REQUIRE(key >= 0 && key < tune_num_keys && value, GKOC_E_INVALID,
                 "unknown tuning key or value == NULL");
    *value = tune_value(key);
    return GKOC_OK;
}

int gkoc_memcpy_h2d(void* dst, const void* src, size_t bytes, gkoc_stream_t s)
{
    if (bytes == 0) return GKOC_OK;
    GKOC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(s)));
    return GKOC_OK;
}

int gkoc_pointer_is_device(const void* ptr, int* is_device)
{
    GKOC_REQUIRE(is_device, GKOC_E_INVALID, "null result");
    *is_device = 0;
    if (!ptr) return GKOC_OK;
    if (arena_owns(ptr)) {
        *is_device = 1;
        return GKOC_OK;
    }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();      // an ordinary host pointer
        return GKOC_OK;
    }
    *is_device = attr.type == hipMemoryTypeDevice ? 1 : 0;
    return GKOC_OK;
}

int gkoc_device_identity(char* out, size_t out_bytes)
{
    GKOC_REQUIRE(out && out_bytes >= 64, GKOC_E_INVALID, "need 64 bytes");
    int dev = 0;
    GKOC_HIP(hipGetDevice(&dev));
    char bus[32] = {0};
    GKOC_HIP(hipDeviceGetPCIBusId(bus, sizeof(bus), dev));
    char host[24] = {0};
    if (gethostname(host, sizeof(host) - 1) != 0) host[0] = 0;
    snprintf(out, out_bytes, "%s/%s", host, bus);
    return GKOC_OK;
}

int gkoc_memcpy_d2h(void* dst, const void* src, size_t bytes, gkoc_stream_t s)
{
    if (bytes == 0) return GKOC_OK;
    GKOC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(s)));
    GKOC_HIP(hipStreamSynchronize(as_stream(s)));
    return GKOC_OK;
}

int gkoc_memcpy_d2d(void* dst, const void* src, size_t bytes, gkoc_stream_t s)
{
    if (bytes == 0) return GKOC_OK;
    // a scalar or two (a reduction's result handed on): one wave of our own instead of the
    // runtime's copy path
    // - only where a kernel on the CURRENT device may dereference both pointers: with several devices in
    // the process (HipExecutor::raw_copy_to between two executors, hip/base/executor.hip.cpp:150-190
    // uses hipMemcpyPeerAsync there) both must be device memory of this device; nothing here enables
    // peer access, so everything else takes the runtime's copy
    static const int n_devices = [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 2;
        (void)hipGetLastError();
        return n;
    }();
    auto local_device_memory = [](const void* p) {
        hipPointerAttribute_t a;
        int dev = -1;
        if (hipPointerGetAttributes(&a, p) != hipSuccess || hipGetDevice(&dev) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return a.type == hipMemoryTypeDevice && a.device == dev;
    };
    if (bytes <= 256 && bytes % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 &&
        reinterpret_cast<uintptr_t>(src) % 4 == 0 &&
        (n_devices == 1 || (local_device_memory(dst) && local_device_memory(src)))) {
        gkoc::small_copy_kernel<<<dim3(1), dim3(64), 0, as_stream(s)>>>(
            static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), int(bytes / 4));
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    GKOC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(s)));
    return GKOC_OK;
}

int gkoc_stream_query(gkoc_stream_t s, int* done)
{
    GKOC_REQUIRE(done, GKOC_E_INVALID, "done == NULL");
    const hipError_t e = hipStreamQuery(as_stream(s));
    if (e == hipSuccess) {
        *done = 1;
        return GKOC_OK;
    }
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *done = 0;
        return GKOC_OK;
    }
    GKOC_HIP(e);
    return GKOC_OK;
}

int gkoc_memset(void* dst, int value, size_t bytes, gkoc_stream_t s)
{
    if (bytes == 0) return GKOC_OK;
    GKOC_HIP(hipMemsetAsync(dst, value, bytes, as_stream(s)));
    return GKOC_OK;
}

int gkoc_stream_create(gkoc_stream_t* s)
{
    GKOC_REQUIRE(s, GKOC_E_INVALID, "s == NULL");
    hipStream_t st;
    GKOC_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = st;
    return GKOC_OK;
}

// a stream whose kernels are dispatched ahead of those of ordinary streams: for the few
// workgroups of a collective or of the boundary rows that must get onto the device while a
// device-filling kernel runs on another stream
int gkoc_stream_create_high_priority(gkoc_stream_t* s)
{
    GKOC_REQUIRE(s, GKOC_E_INVALID, "s == NULL");
    int least = 0, greatest = 0;
    GKOC_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st;
    GKOC_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
    *s = st;
    return GKOC_OK;
}

int gkoc_stream_destroy(gkoc_stream_t s)
{
    if (s) GKOC_HIP(hipStreamDestroy(as_stream(s)));
    return GKOC_OK;
}

int gkoc_stream_synchronize(gkoc_stream_t s)
{
    GKOC_HIP(hipStreamSynchronize(as_stream(s)));
    return GKOC_OK;
}

int gkoc_device_synchronize(void)
{
    GKOC_HIP(hipDeviceSynchronize());
    return GKOC_OK;
}

// ---- ROCTX ranges (gko::log::begin_roctx / end_roctx, hip/base/roctx.hip.cpp:30-36): bound at
// run time, so that the library has no link dependency on a profiler.  rocprofiler-sdk's roctx
// is tried first (rocprofv3 --marker-trace), then the roctracer one.
}  // extern "C"

namespace {
using roctx_push_t = int (*)(const char*);
using roctx_pop_t = int (*)();
struct roctx_api {
    roctx_push_t push = nullptr;
    roctx_pop_t pop = nullptr;
    roctx_api()
    {
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                 "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const roctx_api& roctx()
{
    static roctx_api api;
    return api;
}
}  // namespace

extern "C" {

int gkoc_range_push(const char* name)
{
    GKOC_REQUIRE(name, GKOC_E_INVALID, "name == NULL");
    if (roctx().push) roctx().push(name);
    return GKOC_OK;
}

int gkoc_range_pop(void)
{
    if (roctx().pop) roctx().pop();
    return GKOC_OK;
}

int gkoc_range_available(void) { return roctx().push != nullptr; }

}  // extern "C"
