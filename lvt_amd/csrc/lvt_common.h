// Shared helpers for the liblvt_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "lvt_hip.h"

void lvt_set_error(const char *fmt, ...);

#define LVT_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) {                                          \
            lvt_set_error(__VA_ARGS__);                         \
            return LVT_EINVAL;                                  \
        }                                                       \
    } while (0)

#define LVT_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            lvt_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return LVT_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

static inline bool lvt_aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
static inline __host__ __device__ long long lvt_cdiv(long long a, long long b) { return (a + b - 1) / b; }

// 256 CUs x 8 XCDs on MI355X; used only to size grids / split-K, never for correctness.
#define LVT_NUM_CU 256

