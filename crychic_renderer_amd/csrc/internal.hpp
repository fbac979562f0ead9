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

// api.cpp: crychic_draw_hot_path with the lighting pass in `nparts` row ranges and a hook behind each (comm.cpp's overlapped exchange).
typedef int (*PartHook)(void* user, uint32_t part, uint32_t row0, uint32_t rows);
int hot_path_parts(crychic_ctx* ctx, const crychic_ssao_constants* ssaoCB, const crychic_pass_constants* passCB,
                   const crychic_frame_desc* f, hipStream_t stream, uint32_t nparts, PartHook after, void* user);
}  // namespace cry
