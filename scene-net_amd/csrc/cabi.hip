// C-ABI housekeeping: version, last-error text, device probe.
#include "common.h"
#include <cstring>
#include <cstdlib>

#include <atomic>
#include <mutex>
#include <unordered_map>

namespace sn {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
static std::atomic<int> g_skip_empty{0};
int option_conv_skip_empty_tiles() { return g_skip_empty.load(std::memory_order_relaxed); }
static std::atomic<int> g_i8_legacy{-1};   // -1: not looked at yet; SN_CONV_I8_LEGACY=1 in the environment starts it at 1
int option_conv_i8_legacy() {
    int v = g_i8_legacy.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("SN_CONV_I8_LEGACY");
        int expected = -1;
        g_i8_legacy.compare_exchange_strong(expected, (e && e[0] == '1') ? 1 : 0, std::memory_order_relaxed);
        v = g_i8_legacy.load(std::memory_order_relaxed);
    }
    return v;
}
static std::atomic<int> g_vox_onepass{1};
int option_voxel_onepass() { return g_vox_onepass.load(std::memory_order_relaxed); }
static std::atomic<int> g_vox_spin{64};
int option_voxel_onepass_spin() { return g_vox_spin.load(std::memory_order_relaxed); }
static std::atomic<int> g_i8z_fault{0};
int option_conv_i8z_inject_fault() { return g_i8z_fault.load(std::memory_order_relaxed); }
static std::atomic<int> g_i8z_variant{2};
int option_conv_i8z_variant() { return g_i8z_variant.load(std::memory_order_relaxed); }
static std::atomic<int> g_corr_tile_bytes{0};
int option_corr_sparse_tile_bytes() { return g_corr_tile_bytes.load(std::memory_order_relaxed); }
static std::atomic<int> g_i8_fold{-1};   // -1: not looked at yet; SN_CONV_I8_NOFOLD=1 in the environment starts it at 0
int option_conv_i8_fold() {
    int v = g_i8_fold.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("SN_CONV_I8_NOFOLD");
        int expected = -1;
        g_i8_fold.compare_exchange_strong(expected, (e && e[0] == '1') ? 0 : 1, std::memory_order_relaxed);
        v = g_i8_fold.load(std::memory_order_relaxed);
    }
    return v;
}


namespace {
struct ExtraOpt {
    const char* name;
    const char* env;
    std::atomic<int> value;   // -1: not looked at yet (the environment decides at first use)
};
ExtraOpt g_extra[kOptCount] = {{"conv_no_i8", "SN_CONV_NO_I8", {-1}},
                               {"conv_double_buffer", "SN_CONV_DOUBLE_BUFFER", {-1}},
                               {"conv_lin_no24", "SN_CONV_LIN_NO24", {-1}},
                               {"conv_i8_no_stage", "SN_CONV_I8_NO_STAGE", {-1}},
                               {"corr_dense", "SN_CORR_DENSE", {-1}}};
}  // namespace
int option_extra(ExtraOption which) {
    ExtraOpt& o = g_extra[which];
    int v = o.value.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv(o.env);
        v = (e && e[0] == '1') ? 1 : 0;
        int expected = -1;
        o.value.compare_exchange_strong(expected, v, std::memory_order_relaxed);
        v = o.value.load(std::memory_order_relaxed);
    }
    return v;
}

static thread_local Gate g_gate{{nullptr, nullptr, nullptr}, {0, 0, 0}};
Gate current_gate() { return g_gate; }
GateScope::GateScope(const int32_t* ptr, int want) : slot_(-1) {
    for (int i = 0; i < kGateDepth; ++i)
        if (!g_gate.ptr[i]) { slot_ = i; break; }
    if (slot_ < 0) abort();
    g_gate.ptr[slot_] = ptr;
    g_gate.want[slot_] = want;
}
GateScope::~GateScope() {
    g_gate.ptr[slot_] = nullptr;
    g_gate.want[slot_] = 0;
}

// tolerance of the int8 kernels' quantisation guard, in units of 1e-9 (default 90 000 = 9e-5: the 1e-4 parity bar
// less the fp32 roundings of the recombination)
static std::atomic<int> g_i8_tol_ppb{90000};
float option_conv_i8_tolerance() { return 1e-9f * (float)g_i8_tol_ppb.load(std::memory_order_relaxed); }

// Flag / ticket words for the launches that need one int of device memory.  Two pools per device, allocated together
// the first time (sn_prepare_device, or the first call -- which must then be outside a stream capture):
//   * eager launches draw from a RING of 1024 words (a word is reused after 1024 further launches of this process on the
//     device: the launches that share a word are on one stream in practice, and each rewrites it before reading);
//   * a launch made while its stream is CAPTURING draws from a pool that is never recycled, so a hipGraph owns the words
//     baked into it and no eager call on another stream ever shares one with a replay (4096 words; when they are gone
//     captures fall back to the ring).  The prepared contraction (sn_conv_bank_prepared) needs neither: its flag lives in
//     the caller's blob.
namespace {
constexpr int kFlagRing = 1024, kFlagCapture = 4096, kFlagMaxDev = 16;
std::mutex g_flag_mu;
int32_t* g_flag_mem[kFlagMaxDev] = {nullptr};
unsigned g_flag_next[kFlagMaxDev] = {0}, g_flag_cap_next[kFlagMaxDev] = {0};
// The sticky status words of a device (common.h): 16 ints of pinned, device-mapped host memory, allocated with the flag pool
// (so the same rule holds: before the first stream capture).  Kernels latch into them with system-scope atomics; the host
// reads them at every launch check.  Never freed: a captured graph keeps the device pointer.
volatile int32_t* g_sticky_host[kFlagMaxDev] = {nullptr};
int32_t* g_sticky_dev[kFlagMaxDev] = {nullptr};
int32_t* flag_pool_locked(int dev) {
    if (!g_flag_mem[dev]) {
        if (hipMalloc((void**)&g_flag_mem[dev], (kFlagRing + kFlagCapture) * sizeof(int32_t)) != hipSuccess) {
            (void)hipGetLastError();
            g_flag_mem[dev] = nullptr;
            return nullptr;
        }
        (void)hipMemset(g_flag_mem[dev], 0, (kFlagRing + kFlagCapture) * sizeof(int32_t));
    }
    if (!g_sticky_host[dev]) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, 16 * sizeof(int32_t));
            g_sticky_dev[dev] = static_cast<int32_t*>(d);
            g_sticky_host[dev] = static_cast<volatile int32_t*>(h);
        } else {
            (void)hipGetLastError();   // no sticky words: kernels skip the latch (nullptr), the counters still count
        }
    }
    return g_flag_mem[dev];
}
}  // namespace

namespace {
thread_local hipEvent_t g_timing_start = nullptr, g_timing_stop = nullptr;
}
void take_timing_events(hipEvent_t& start, hipEvent_t& stop) {
    start = g_timing_start;
    stop = g_timing_stop;
    g_timing_start = g_timing_stop = nullptr;
}

int32_t* sticky_device_ptr(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kFlagMaxDev) return nullptr;
    if (g_sticky_dev[dev]) return g_sticky_dev[dev];   // (set once, never changed: no lock on the launch path)
    // not allocated yet: never allocate while `stream` is capturing (hipMalloc / hipHostMalloc are illegal there and would
    // poison the capture) -- such a launch simply has no status words to latch into (its kernel skips a null pointer)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive) return nullptr;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_flag_mu);
    (void)flag_pool_locked(dev);
    return g_sticky_dev[dev];
}

int sticky_check(const char* what) {
    // every device this process has touched: a latched status is a property of the process's results, not of one stream
    for (int dev = 0; dev < kFlagMaxDev; ++dev) {
        volatile int32_t* w = g_sticky_host[dev];
        if (!w || w[0] == 0) continue;
        static const char* const kText[] = {"", "a dependency spin of the z-walk contraction gave up (its workgroup's outputs are NaN)",
                                            "a launch made with `assume served` was declined by the bank's guard (its outputs are NaN): "
                                            "a cached verdict did not belong to these parameters",
                                            "a hand-over spin of an int8 tile kernel gave up (that launch's output is not to be trusted)"};
        const int code = w[0];
        return fail(SN_ERR_DEVICE_STATUS, "%s: device %d has a latched status %d -- %s [workgroup %d, detail %d]; "
                    "sn_device_status_clear() re-arms the device", what, dev, code,
                    (code >= 1 && code <= 3) ? kText[code] : "unknown", (int)w[2], (int)w[3]);
    }
    return SN_OK;
}

int32_t* device_flag_slot(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kFlagMaxDev) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
    if (!capturing) (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_flag_mu);
    int32_t* mem = flag_pool_locked(dev);
    if (!mem) return nullptr;
    if (capturing && g_flag_cap_next[dev] < (unsigned)kFlagCapture) return mem + kFlagRing + g_flag_cap_next[dev]++;
    return mem + (g_flag_next[dev]++ % kFlagRing);
}

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> configured;  // per process; one device kind (gfx950)
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* key = (const char*)kernel + dev;  // attributes are per device
    auto it = configured.find(key);
    if (it != configured.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) configured[key] = bytes;
    return e;
}
}  // namespace sn

extern "C" int sn_set_option(const char* name, int value) {
    if (!name) return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: null name");
    if (strcmp(name, "conv_skip_empty_tiles") == 0) {
        sn::g_skip_empty.store(value ? 1 : 0, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "conv_i8_tolerance_ppb") == 0) {
        if (value < 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: conv_i8_tolerance_ppb must be >= 0");
        sn::g_i8_tol_ppb.store(value, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "conv_i8_legacy") == 0) {
        sn::g_i8_legacy.store(value ? 1 : 0, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "conv_i8_fold") == 0) {
        sn::g_i8_fold.store(value ? 1 : 0, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "voxel_onepass") == 0) {
        sn::g_vox_onepass.store(value ? 1 : 0, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "voxel_onepass_spin") == 0) {
        if (value < 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: voxel_onepass_spin must be >= 0");
        sn::g_vox_spin.store(value, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "conv_i8z_inject_fault") == 0) {
        sn::g_i8z_fault.store(value ? 1 : 0, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "conv_i8z_variant") == 0) {
        if (value < 0 || value > 2) return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: conv_i8z_variant is 0, 1 or 2");
        sn::g_i8z_variant.store(value, std::memory_order_relaxed);
        return SN_OK;
    }
    if (strcmp(name, "corr_sparse_tile_bytes") == 0) {
        if (value < 0 || value > 2048)
            return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: corr_sparse_tile_bytes is 0 (default: 2048) .. 2048");
        sn::g_corr_tile_bytes.store(value, std::memory_order_relaxed);
        return SN_OK;
    }
    for (auto& o : sn::g_extra)
        if (strcmp(name, o.name) == 0) {
            o.value.store(value ? 1 : 0, std::memory_order_relaxed);
            return SN_OK;
        }
    return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: unknown option '%s'", name);
}

extern "C" int sn_get_option(const char* name) {
    if (name && strcmp(name, "conv_skip_empty_tiles") == 0) return sn::option_conv_skip_empty_tiles();
    if (name && strcmp(name, "conv_i8_tolerance_ppb") == 0) return sn::g_i8_tol_ppb.load(std::memory_order_relaxed);
    if (name && strcmp(name, "conv_i8_legacy") == 0) return sn::option_conv_i8_legacy();
    if (name && strcmp(name, "conv_i8_fold") == 0) return sn::option_conv_i8_fold();
    if (name && strcmp(name, "conv_i8z_variant") == 0) return sn::option_conv_i8z_variant();
    if (name && strcmp(name, "conv_i8z_inject_fault") == 0) return sn::option_conv_i8z_inject_fault();
    if (name && strcmp(name, "voxel_onepass") == 0) return sn::option_voxel_onepass();
    if (name && strcmp(name, "voxel_onepass_spin") == 0) return sn::option_voxel_onepass_spin();
    if (name && strcmp(name, "corr_sparse_tile_bytes") == 0) return sn::option_corr_sparse_tile_bytes();
    if (name)
        for (int i = 0; i < sn::kOptCount; ++i)
            if (strcmp(name, sn::g_extra[i].name) == 0) return sn::option_extra((sn::ExtraOption)i);
    return -1;
}

extern "C" int sn_prepare_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= sn::kFlagMaxDev)
        return sn::fail(SN_ERR_NO_DEVICE, "sn_prepare_device: no current HIP device");
    std::lock_guard<std::mutex> lock(sn::g_flag_mu);
    return sn::flag_pool_locked(dev) ? SN_OK : sn::fail(SN_ERR_LAUNCH, "sn_prepare_device: cannot allocate the flag pool");
}

extern "C" int sn_device_status(int* code, int* detail2) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= sn::kFlagMaxDev)
        return sn::fail(SN_ERR_NO_DEVICE, "sn_device_status: no current HIP device");
    volatile int32_t* w = sn::g_sticky_host[dev];
    if (code) *code = w ? (int)w[0] : 0;
    if (detail2) {
        detail2[0] = w ? (int)w[2] : 0;
        detail2[1] = w ? (int)w[3] : 0;
    }
    return SN_OK;
}

extern "C" int sn_launch_timing_events(void* start_event, void* stop_event) {
    sn::g_timing_start = static_cast<hipEvent_t>(start_event);
    sn::g_timing_stop = static_cast<hipEvent_t>(stop_event);
    return SN_OK;
}

extern "C" int sn_device_status_clear(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= sn::kFlagMaxDev)
        return sn::fail(SN_ERR_NO_DEVICE, "sn_device_status_clear: no current HIP device");
    // the kernels that latched have to be through before the words are re-armed
    if (hipDeviceSynchronize() != hipSuccess) {
        const hipError_t e = hipGetLastError();
        return sn::fail(SN_ERR_LAUNCH, "sn_device_status_clear: %s", hipGetErrorString(e));
    }
    volatile int32_t* w = sn::g_sticky_host[dev];
    if (w) {
        w[0] = 0; w[2] = 0; w[3] = 0;
        w[1] = 0;
    }
    return SN_OK;
}

extern "C" int sn_version(void) { return 103; }   // 103: round 4 (sticky device status, sn_launch_timing_events); 102: round-3 entries (riders, sn_conv_corr_ws, sn_conv_fused_v, sn_loss_*_m / _u)

extern "C" const char* sn_last_error(void) { return sn::error_buffer(); }

extern "C" int sn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
