// C-ABI housekeeping: version, last-error text, device probe.
#include "common.h"
#include <cstring>

namespace sn {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace sn

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
