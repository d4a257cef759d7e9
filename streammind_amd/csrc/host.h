// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/streammind_hip.h"

extern thread_local char g_sm_err[512];

#define SM_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(g_sm_err, sizeof(g_sm_err), __VA_ARGS__);   \
        return (code);                                       \
    } while (0)

#define SM_REQUIRE(cond, ...)                                \
    do {                                                     \
        if (!(cond)) SM_FAIL(SM_EINVAL, __VA_ARGS__);        \
    } while (0)

#define SM_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) SM_FAIL(SM_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#define SM_LAUNCH_CHECK() SM_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
