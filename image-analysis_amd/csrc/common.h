// Shared host-side helpers for libimgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace imgan {

// Error codes returned through the C ABI (include/imgan.h).
enum { OK = 0, E_INVALID = -1, E_HIP = -2, E_RANGE = -3, E_NOMEM = -4 };

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_launch(const char* what);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace imgan

#define IMGAN_HIP(expr)                                                            \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess)                                                      \
            return imgan::fail(imgan::E_HIP, std::string(#expr) + ": " +           \
                                                 hipGetErrorString(_e));           \
    } while (0)

#define IMGAN_REQUIRE(cond, msg)                                                   \
    do {                                                                           \
        if (!(cond)) return imgan::fail(imgan::E_INVALID, msg);                    \
    } while (0)
