// Error reporting / version / device probing for the C ABI.
#include "common.hpp"
#include <cstring>

namespace oodgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace oodgan

extern "C" int oodgan_version(void) { return 100; }
extern "C" const char* oodgan_last_error(void) { return oodgan::g_err; }
extern "C" int oodgan_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
