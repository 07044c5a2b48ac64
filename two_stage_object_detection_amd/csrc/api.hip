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
