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
static std::atomic<int> g_i8z_variant{2};
int option_conv_i8z_variant() { return g_i8z_variant.load(std::memory_order_relaxed); }
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
                               {"conv_i8_no_stage", "SN_CONV_I8_NO_STAGE", {-1}}};
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

int32_t* device_flag_slot() {
    constexpr int kRing = 1024, kMaxDev = 16;
    static std::mutex mu;
    static int32_t* ring[kMaxDev] = {nullptr};
    static unsigned next[kMaxDev] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ring[dev]) {
        if (hipMalloc((void**)&ring[dev], kRing * sizeof(int32_t)) != hipSuccess) {
            (void)hipGetLastError();
            ring[dev] = nullptr;
            return nullptr;
        }
        (void)hipMemset(ring[dev], 0, kRing * sizeof(int32_t));
    }
    return ring[dev] + (next[dev]++ % kRing);
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
    if (strcmp(name, "conv_i8z_variant") == 0) {
        if (value < 0 || value > 3) return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: conv_i8z_variant is 0 .. 3");
        sn::g_i8z_variant.store(value, std::memory_order_relaxed);
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
    if (name)
        for (int i = 0; i < sn::kOptCount; ++i)
            if (strcmp(name, sn::g_extra[i].name) == 0) return sn::option_extra((sn::ExtraOption)i);
    return -1;
}

extern "C" int sn_version(void) { return 100; }

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
