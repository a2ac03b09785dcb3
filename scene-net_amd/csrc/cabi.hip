// C-ABI housekeeping: version, last-error text, device probe.
#include "common.h"
#include <cstring>

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

static thread_local Gate g_gate{nullptr, 0};
Gate current_gate() { return g_gate; }
GateScope::GateScope(const int32_t* ptr, int want) { g_gate = Gate{ptr, want}; }
GateScope::~GateScope() { g_gate = Gate{nullptr, 0}; }

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
    return sn::fail(SN_ERR_INVALID_ARG, "sn_set_option: unknown option '%s'", name);
}

extern "C" int sn_get_option(const char* name) {
    if (name && strcmp(name, "conv_skip_empty_tiles") == 0) return sn::option_conv_skip_empty_tiles();
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
