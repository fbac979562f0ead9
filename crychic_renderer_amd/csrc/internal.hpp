// internal.hpp -- what the translation units behind the C ABI share: the context object and the error recorder.
#pragma once
#include <hip/hip_runtime_api.h>
#include "crychic_hip.h"

struct crychic_ctx {
    int device;
    char name[256];
    bool profiling;
    bool times_valid;
    hipEvent_t ev[4];  // start, after ssao, after blur, after light
    const uint32_t* rasterStatus;   // device status word of the most recent producer pass (crychic_raster_status)
    const unsigned long long* chainStatus;      // error word of the most recent single-launch blur chain (crychic_blur_chain_status)
    unsigned long long chainTag;                // (its frame stamp << 8) | 255: what the word holds if a workgroup gave up waiting
};

namespace cry {
// Stores the message crychic_last_error() returns (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace cry
