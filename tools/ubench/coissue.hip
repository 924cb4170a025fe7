// Micro-benchmark: can the bf16 MFMAs and the VALU work of the bf16x3 split overlap on one SIMD?
//   mode 0: MFMA only   mode 1: VALU only   mode 2: both in every wave   mode 3: even waves MFMA, odd waves VALU
// build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue ; run: ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;
}

template <int MODE, int NMFMA, int NSPLIT>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    const int tid = threadIdx.x, wave = tid >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(tid + i); b[i] = (__bf16)(float)(tid * 3 + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[tid * 8 + i];
    unsigned sink = 0;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int s = 0; s < NSPLIT; ++s) {
                // one 3-way split of 2 floats: 3 cvt_pk + 4 sub + 4 shift/and  (11 VALU)
                float x0 = v[(2 * s) & 7], x1 = v[(2 * s + 1) & 7];
                unsigned p1 = cvt_pk(x0, x1);
                float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
                unsigned p2 = cvt_pk(r0, r1);
                float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
                unsigned p3 = cvt_pk(s0, s1);
                sink ^= p1 + p2 + p3;
                v[(2 * s) & 7] = s0 + 1.0f; v[(2 * s + 1) & 7] = s1 + 1.0f;
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + tid] = t + (float)sink + v[0];
}

template <int MODE, int NMFMA, int NSPLIT> static void run(const char *name, float *out, float *in, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NMFMA, NSPLIT>), dim3(blocks), dim3(256), 0, 0, out, in, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NMFMA, NSPLIT>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / (256 * 4);
    const double ns_per_iter = ms * 1e6 / iters;
    const double mfma_per_wave = (MODE == 1 ? 0 : NMFMA) * (MODE == 3 ? 0.5 : 1.0);
    const double tf = blocks * 4.0 * mfma_per_wave * iters * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %4d  %8.1f ns/iter  (%.1f waves/SIMD)  MFMA rate %7.1f TF  (%.1f%% of 2516)\n", name, blocks, ns_per_iter,
           waves_per_simd, tf, tf / 2516.6 * 100);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 256 * 8 * 4);
    hipMemset(in, 0, 256 * 8 * 4);
    // engine k-tile per wave: 48 MFMAs, ~350 VALU (= 32 splits of 11)
    for (int blocks : {256, 512}) {
        run<0, 48, 32>("MFMA only (48/iter)", out, in, blocks);
        run<1, 48, 32>("VALU only (32 splits = 352/iter)", out, in, blocks);
        run<2, 48, 32>("both, same wave", out, in, blocks);
        run<3, 48, 32>("specialised waves (even M, odd V)", out, in, blocks);
        run<2, 48, 16>("both, same wave, 16 splits", out, in, blocks);
    }
    return 0;
}
