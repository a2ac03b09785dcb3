// Shared host-side helpers of the C ABI (error text, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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

int sticky_check(const char* what);   // below / cabi.hip

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return sticky_check(what);   // (a host read of pinned words; SN_OK unless a kernel has latched a status)
}

inline hipStream_t as_stream(sn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

int option_conv_skip_empty_tiles();  // cabi.hip (sn_set_option)
int option_conv_i8_fold();            // cabi.hip (sn_set_option "conv_i8_fold", default 1): 0 = never try the folded int8 kernel (conv_i8s.hip)
int option_voxel_onepass();           // cabi.hip (sn_set_option "voxel_onepass", default 1): 0 = the two-kernel form (bbox, then binning) on every grid
int option_voxel_onepass_spin();      // cabi.hip (sn_set_option "voxel_onepass_spin", default 64): polls of the box exchange before a workgroup goes alone
int option_conv_i8z_inject_fault();   // cabi.hip (sn_set_option "conv_i8z_inject_fault"): test hook, see conv_i8z.inc's prologue
int option_conv_i8z_variant();        // cabi.hip (sn_set_option "conv_i8z_variant"): shape of the z-walk kernel's rounds (conv_i8z.inc)
int option_corr_sparse_tile_bytes();  // cabi.hip (sn_set_option "corr_sparse_tile_bytes"): input bytes per job of the sparse correlation (0: 2048, the maximum)
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

// Result-preserving switches beside the named accessors above: set with sn_set_option(name, v); the environment variable
// listed with each is read ONCE, the first time the option is looked at (never per call: a getenv per launch sat on the
// production dispatch, and a stray variable could change kernels mid-run).
enum ExtraOption {
    kOptConvNoI8 = 0,        // "conv_no_i8"        SN_CONV_NO_I8=1         binary occupancy through the fp32 kernel
    kOptConvDoubleBuffer,    // "conv_double_buffer" SN_CONV_DOUBLE_BUFFER=1 fp32 kernel: double-buffered 4x4x64 tiles
    kOptConvLinNo24,         // "conv_lin_no24"     SN_CONV_LIN_NO24=1      K3L: 32-byte instead of 24-byte kernel rows
    kOptConvI8NoStage,       // "conv_i8_no_stage"  SN_CONV_I8_NO_STAGE=1   four-copy kernel without the LDS-DMA staging
    kOptCorrDense,           // "corr_dense"        SN_CORR_DENSE=1         backward correlation of binary occupancy as the GEMM (K4) instead of the gather (K4s)
    kOptCount
};
int option_extra(ExtraOption which);   // cabi.hip
// The kernels' side of those switches: SN_DBG(shape, bit) is a COMPILE-TIME false in the product build -- no kernel carries
// a runtime test of a debug word; in a -DSN_CONV_DEBUG build it reads the shape's dbg field.
#ifdef SN_CONV_DEBUG
#define SN_DBG(shape, bit) (((shape).dbg & (bit)) != 0)
#else
#define SN_DBG(shape, bit) false
#endif
// Wrong-result timing switches (a skipped epilogue, idle waves ...) exist only in builds made with -DSN_CONV_DEBUG
// (make EXTRA=-DSN_CONV_DEBUG): in the product they read as 0 whatever the environment says.
inline int debug_env_int(const char* name) {
#ifdef SN_CONV_DEBUG
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
#else
    (void)name;
    return 0;
#endif
}

// ---- sticky device status (round 4): a kernel that cannot keep its contract (a dependency spin that gave up, a launch
// made with `assume served` whose bank the guard declines) must not return plausible numbers.  It poisons what it owns
// (NaN) and LATCHES a status in four host-pinned words per device; every sn_* launch path looks at them (check_launch,
// a host memory read: no synchronisation) and fails with SN_ERR_DEVICE_STATUS until sn_device_status_clear().
//   words: [0] code (0: healthy)  [1] claim (first reporter wins)  [2], [3] detail (workgroup, ticket / verdict)
//   codes: 1 a dependency spin of the z-walk gave up;  2 a `served` launch of the z-walk declined its bank;
//          3 a tile kernel's (folded / stride-4 / four-copy) spin gave up
void take_timing_events(hipEvent_t& start, hipEvent_t& stop);   // cabi.hip: the pair set by sn_launch_timing_events, consumed
int32_t* sticky_device_ptr(hipStream_t stream);   // cabi.hip: the current device's words as a DEVICE pointer (nullptr: not allocated
                                                  // and `stream` is capturing -- nothing is ever allocated inside a capture)
int sticky_check(const char* what);   // cabi.hip: SN_OK, or SN_ERR_DEVICE_STATUS with the latched code in the error text
__device__ __forceinline__ void sticky_latch(int32_t* sticky, int code, int a, int b) {
    if (!sticky) return;
    // (a, b are wave-uniform; made opaque HERE so that hipcc does not materialise them in vector registers ahead of the hot
    // loop the latch sits behind -- [measured in the ISA] it hoisted a v_mov of blockIdx.x out of the z-walk's ticket loop and
    // spilled it: the kernel's only scratch use)
    a = __builtin_amdgcn_readfirstlane(a);   // (uniform by contract; the readfirstlane makes it so for the register allocator,
    b = __builtin_amdgcn_readfirstlane(b);   // whatever it thinks of the caller's expression)
    asm volatile("" : "+s"(a), "+s"(b));
    if (__hip_atomic_exchange(&sticky[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        __hip_atomic_store(&sticky[2], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&sticky[3], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&sticky[0], code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

float option_conv_i8_tolerance();   // cabi.hip (sn_set_option "conv_i8_tolerance_ppb"): 0 = quantisation guard off
// One int of device memory per call (the flag an int8 launch leaves for the gated fp32 launch enqueued behind it, dynamic
// tile tickets): from a recycled ring for eager launches, from a never-recycled pool while `stream` is capturing
// (cabi.hip).  nullptr if the pool cannot be allocated.
int32_t* device_flag_slot(hipStream_t stream);

}  // namespace sn
