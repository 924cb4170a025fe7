// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate as a function of the operand DATA (gfx950).
// 2 waves per SIMD, MFMAs only (no loads, no LDS, no VALU in the loop), 4 accumulators, 8 distinct A and 8 distinct
// B fragments cycled so that consecutive MFMAs see different operands, as in a real k-loop.
//   data 0: all zeros   1: small integers (exact, few set bits)   2: random bf16 in [-1,1]   3: random bit patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(256) void k(float *out, const uint4 *ops, int iters) {
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        uint4 ua = ops[(i * 256 + tid)], ub = ops[((8 + i) * 256 + tid)];
        memcpy(&a[i], &ua, 16); memcpy(&b[i], &ub, 16);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m * 3) & 7], acc[m & 3], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + tid] = t;
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    float *out; uint4 *ops;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ops, 16 * 256 * 16);
    std::vector<unsigned short> h(16 * 256 * 8);
    const char *names[4] = {"zeros", "small integers", "random bf16 in [-1,1]", "random bit patterns (finite)"};
    for (int data = 0; data < 4; ++data) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            if (data == 0) h[i] = 0;
            else if (data == 1) h[i] = f2bf((float)(rand() % 4));
            else if (data == 2) h[i] = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
            else { unsigned short v = (unsigned short)rand(); if (((v >> 7) & 0xff) == 0xff) v &= 0xbfff; h[i] = (v & 0x80ff) | (((v >> 7) & 0x0f) + 0x78) << 7; }
        }
        (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const int blocks = 512, iters = 20000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, ops, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, ops, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double tf = blocks * 4.0 * 48.0 * iters * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
        printf("%-30s %8.2f ms  %7.1f TF bf16  (%.1f%% of 2516.6; as bf16x3: %.1f TF fp32-equivalent)\n", names[data], ms, tf,
               tf / 2516.6 * 100, tf / 6);
    }
    return 0;
}
