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
int option_conv_i8_fold();            // cabi.hip (sn_set_option "conv_i8_fold", default 1): 0 = never try the folded int8 kernel (conv_i8s.hip)
int option_conv_i8z_variant();        // cabi.hip (sn_set_option "conv_i8z_variant"): shape of the z-walk kernel's rounds (conv_i8z.inc)
int option_conv_i8_legacy();          // cabi.hip (sn_set_option "conv_i8_legacy"): 1 = the four-copy kernel of conv_i8.hip for every shape

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per kernel (and raised when a launch needs more): the
// attribute call is not a stream operation, so steady-state launches stay free of it (cheaper, graph-capturable).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);  // cabi.hip

// Device-side gate of the conv launches made by this thread while a GateScope is alive: a kernel whose shape carries
// (gate, want) exits in its first instruction unless *gate == want (sn_forward_auto enqueues the int8 and the fp32
// form of the same forward and lets a device flag pick one -- no host synchronisation).
// Scopes nest (up to kGateDepth): every condition of the enclosing scopes must hold as well -- sn_forward_auto's
// "grid is binary" around the int8 kernels' "quantisation bound exceeded, run the fp32 form" (conv_i8s.hip).
constexpr int kGateDepth = 3;
struct Gate {
    const int32_t* ptr[kGateDepth];
    int want[kGateDepth];
    __device__ __forceinline__ bool pass() const {
#pragma unroll
        for (int i = 0; i < kGateDepth; ++i)
            if (ptr[i] && *ptr[i] != want[i]) return false;
        return true;
    }
};
Gate current_gate();          // all null outside a scope
struct GateScope {
    GateScope(const int32_t* ptr, int want);   // a fourth nested scope is a programming error: it aborts the process
    ~GateScope();
  private:
    int slot_;
};

float option_conv_i8_tolerance();   // cabi.hip (sn_set_option "conv_i8_tolerance_ppb"): 0 = quantisation guard off
// One int of device memory per call, out of a per-device ring allocated once (1024 calls may be in flight before a
// slot is reused): the flag an int8 launch leaves for the gated fp32 launch enqueued behind it, dynamic tile tickets.
int32_t* device_flag_slot();        // cabi.hip; nullptr if the ring cannot be allocated

}  // namespace sn
