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


// ---- max |.| reporting of the kernels that produce engine operands (LVT_MATH_F16X2 scales, include/lvt_hip.h) -----------
// Non-negative floats order like their bit patterns, so the cross-workgroup step is an INTEGER atomic max: exact and
// order-independent (the library has no floating-point atomics).  One atomic per WORKGROUP, and only when the value would
// change what is already there (a racy read is fine: the atomic decides) -- same-address atomics serialise at ~6 ns each
// on this part, 8192 of them cost a stand-alone pass 50 us.  `scratch`: >= blockDim.x / 64 floats of LDS that every
// thread may overwrite; all threads of the workgroup must call.
#ifdef __HIPCC__
// |a| of a FINITE a, else 0: the max |.| of an operand is taken over its finite entries -- an inf / nan element poisons the
// products it takes part in through its own fp16 hi term (inf * s = inf), and must not collapse the scale of all the others
__device__ __forceinline__ float lvt_absf(float a) {
    const float b = fabsf(a);
    return b < __builtin_inff() ? b : 0.f;
}
__device__ __forceinline__ void lvt_block_amax_commit(float m, float *dst, float *scratch) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, scratch[w]);
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(reinterpret_cast<unsigned *>(dst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned *>(dst), bits);
    }
}
#endif
