// tsod_internal.h -- helpers shared by the HIP translation units of libtsod.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tsod.h"

#define TSOD_WAVE 64

static inline hipStream_t tsod_stream(tsod_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int tsod_launch_status() {
    return hipGetLastError() == hipSuccess ? TSOD_OK : TSOD_ERR_LAUNCH;
}

static inline bool tsod_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t tsod_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define TSOD_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)
