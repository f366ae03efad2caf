// Error plumbing and version of the libisg C ABI.
#include <stdarg.h>
#include <string.h>

#include "isg_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace isg {

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ISG_ELAUNCH;
    }
    return ISG_OK;
}

}  // namespace isg

extern "C" int isg_version(void) { return 100; }

extern "C" const char* isg_last_error(void) { return g_err; }
