// Shared host-side helpers of the C ABI (error text, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "scenenet_hip.h"

namespace sn {

char* error_buffer();  // thread-local, 512 bytes (defined in cabi.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SN_OK;
}

inline hipStream_t as_stream(sn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

int option_conv_skip_empty_tiles();  // cabi.hip (sn_set_option)

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per kernel (and raised when a launch needs more): the
// attribute call is not a stream operation, so steady-state launches stay free of it (cheaper, graph-capturable).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);  // cabi.hip

// Device-side gate of the conv launches made by this thread while a GateScope is alive: a kernel whose shape carries
// (gate, want) exits in its first instruction unless *gate == want (sn_forward_auto enqueues the int8 and the fp32
// form of the same forward and lets a device flag pick one -- no host synchronisation).
struct Gate {
    const int32_t* ptr;
    int want;
};
Gate current_gate();          // {nullptr, 0} outside a scope
struct GateScope {
    GateScope(const int32_t* ptr, int want);
    ~GateScope();
};

}  // namespace sn
