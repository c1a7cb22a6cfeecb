// Error string, device info, hipGraph capture and hipEvent helpers of the C ABI.
#include <string.h>

#include "lt_common.h"

namespace lt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}
int device_cu_count8() {
    static std::atomic<int> cache[64];
    const int dev = current_device() & 63;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
        if (n <= 0) n = 256;
        n -= n % 8;   // the tile dealing of the persistent kernels assumes workgroup b runs on XCD b % 8
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
}  // namespace lt

using namespace lt;

#define LT_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));       \
            return LT_ERR_LAUNCH;                                           \
        }                                                                   \
    } while (0)

extern "C" const char* lt_last_error(void) { return g_err; }
extern "C" int lt_abi_version(void) { return LT_ABI_VERSION; }

extern "C" int lt_device_info(int* cu_count, int* lds_per_cu, char* arch, int arch_len) {
    int dev = 0;
    LT_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    LT_HIP(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return LT_OK;
}

extern "C" int lt_graph_begin(void* stream) {
    LT_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return LT_OK;
}
extern "C" int lt_graph_end(void* stream, void** graph_exec_out) {
    LT_REQUIRE(graph_exec_out, LT_ERR_INVALID, "lt_graph_end: null output");
    hipGraph_t g = nullptr;
    LT_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return LT_ERR_LAUNCH;
    }
    *graph_exec_out = (void*)ge;
    return LT_OK;
}
extern "C" int lt_graph_launch(void* graph_exec, void* stream) {
    LT_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return LT_OK;
}
extern "C" int lt_graph_destroy(void* graph_exec) {
    if (graph_exec) LT_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return LT_OK;
}
extern "C" int lt_event_create(void** ev_out) {
    LT_REQUIRE(ev_out, LT_ERR_INVALID, "lt_event_create: null output");
    hipEvent_t e;
    LT_HIP(hipEventCreate(&e));
    *ev_out = (void*)e;
    return LT_OK;
}
extern "C" int lt_event_record(void* ev, void* stream) {
    LT_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return LT_OK;
}
extern "C" int lt_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    LT_REQUIRE(ms_out, LT_ERR_INVALID, "lt_event_elapsed_ms: null output");
    LT_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    LT_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return LT_OK;
}
extern "C" int lt_event_destroy(void* ev) {
    if (ev) LT_HIP(hipEventDestroy((hipEvent_t)ev));
    return LT_OK;
}
