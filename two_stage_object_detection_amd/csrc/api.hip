// api.hip -- status strings, version and device query of libtsod.so.
#include "tsod_internal.h"

extern "C" const char *tsod_status_str(int status) {
    switch (status) {
        case TSOD_OK: return "ok";
        case TSOD_ERR_INVALID_ARG: return "invalid argument";
        case TSOD_ERR_UNSUPPORTED: return "unsupported configuration";
        case TSOD_ERR_ALIGNMENT: return "pointer, pitch or offset not aligned as required";
        case TSOD_ERR_WORKSPACE: return "workspace missing or too small";
        case TSOD_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

extern "C" int tsod_version(void) { return TSOD_VERSION; }

extern "C" int tsod_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return n;
}


// ---- a device word made readable by the host without a device call (serving: the slot's range flag, one read per request) ----
// tsod_host_mapped_pointer: the device-side address of page-locked host memory (hipHostMalloc / torch's pin_memory()); a query,
// made once when the buffer is created.  tsod_word_publish_i32: ONE thread stores *src to that address at the end of a forward
// (stream-ordered, capturable: a kernel node); the host reads its own memory once the forward's event has completed.
namespace {
__global__ void word_publish_kernel(const int *__restrict__ src, int *__restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src;
}
}  // namespace

extern "C" int tsod_host_mapped_pointer(void *host, void **device) {
    TSOD_REQUIRE(host != nullptr && device != nullptr, TSOD_ERR_INVALID_ARG);
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || d == nullptr) {
        (void)hipGetLastError();
        return TSOD_ERR_UNSUPPORTED;                  // not page-locked / not mapped into this device
    }
    *device = d;
    return TSOD_OK;
}

extern "C" int tsod_word_publish_i32(const int32_t *src_device, int32_t *dst_mapped, tsod_stream_t stream) {
    TSOD_REQUIRE(src_device != nullptr && dst_mapped != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((reinterpret_cast<uintptr_t>(src_device) & 3u) == 0 && (reinterpret_cast<uintptr_t>(dst_mapped) & 3u) == 0, TSOD_ERR_ALIGNMENT);
    hipLaunchKernelGGL(word_publish_kernel, dim3(1), dim3(64), 0, tsod_stream(stream), src_device, dst_mapped);
    return tsod_launch_status();
}
