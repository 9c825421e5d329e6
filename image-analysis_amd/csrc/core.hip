// Error plumbing of the C ABI (include/imgan.h).
#include "common.h"
#include "../../include/imgan.h"

namespace imgan {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return OK;
}

}  // namespace imgan

extern "C" const char* imgan_last_error(void) { return imgan::g_last_error.c_str(); }
extern "C" int imgan_version(void) { return 100; }
