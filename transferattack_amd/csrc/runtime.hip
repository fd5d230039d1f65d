// Error reporting and ABI bookkeeping for libta_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include "ta_common.h"

namespace ta {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(err));
        return static_cast<int>(err);
    }
    return 0;
}

}  // namespace ta

extern "C" int ta_abi_version(void) { return TA_ABI_VERSION; }
extern "C" const char* ta_last_error(void) { return ta::g_error; }
