// MFMA issue-rate probe (gfx950): v_mfma_f32_32x32x16_f16 back to back, NACC independent accumulators per wave, WPS waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int FILL>
__global__ __launch_bounds__(512) void probe(float *out, int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
    f16v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v0 = seed, v1 = seed + 1.f, v2 = seed + 2.f, v3 = seed + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
            if (FILL >= 1) v0 = fmaf(v0, 1.0001f, 0.5f);
            if (FILL >= 2) v1 = fmaf(v1, 1.0001f, 0.5f);
            if (FILL >= 3) v2 = fmaf(v2, 1.0001f, 0.5f);
            if (FILL >= 4) v3 = fmaf(v3, 1.0001f, 0.5f);
            if (FILL >= 5) v0 = fmaxf(v0, v2);
            if (FILL >= 6) v1 = fmaxf(v1, v3);
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FILL>
void run(int threads, const char *name) {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NACC, FILL>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * NACC * (threads / 64) / 4.0;
    printf("%-28s waves/SIMD %d  acc %d  fill %d : %8.1f us, %.2f ns per MFMA per SIMD  (= %.1f cycles at 2.4 GHz, %.0f TFLOP/s)\n", name, threads / 256, NACC,
           FILL, ms * 1e3, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, 256.0 * 4 * mfma_per_simd * 32768 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
template <int NACC, int NSEL>
__global__ __launch_bounds__(512) void probe_sel(float *out, int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
    f16v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float lb = seed; int lp = 0; float h = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NSEL; ++k) {
                const float s0 = fmaf(h, seed, lb * 0.5f + k), s1 = fmaf(h, seed, lb * 0.25f + k);
                unsigned long long m0, m1;
                asm("v_cmp_gt_f32_e64 %2, %4, %0\n\tv_max_f32_e32 %0, %0, %4\n\tv_cmp_gt_f32_e64 %3, %5, %0\n\tv_cndmask_b32_e64 %1, %1, 7, %2\n\t"
                    "v_max_f32_e32 %0, %0, %5\n\tv_cndmask_b32_e64 %1, %1, 9, %3" : "+v"(lb), "+v"(lp), "=&s"(m0), "=&s"(m1) : "v"(s0), "v"(s1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = lb + lp;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int NSEL>
void run_sel(int threads) {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe_sel<NACC, NSEL>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * NACC * (threads / 64) / 4.0;
    printf("select blocks: waves/SIMD %d  acc %d  blocks/MFMA %d (%d VALU) : %.2f ns per MFMA per SIMD\n", threads / 256, NACC, NSEL, NSEL * 10,
           ms * 1e6 / mfma_per_simd);
    hipFree(out);
}
int main() {
    run<1, 2>(512, "2 waves/SIMD + 2 VALU"); run<1, 4>(512, "2 waves/SIMD + 4 VALU"); run<1, 6>(512, "2 waves/SIMD + 6 VALU");
    run_sel<1, 1>(512); run_sel<2, 1>(512); run_sel<4, 1>(512); run_sel<4, 1>(256); run_sel<2, 0>(512); run_sel<1, 0>(512);
    run<4, 0>(256, "1 wave/SIMD"); run<4, 0>(512, "2 waves/SIMD"); run<2, 0>(512, "2 waves/SIMD"); run<1, 0>(512, "2 waves/SIMD");
    run<2, 2>(512, "2 waves/SIMD + 2 VALU"); run<2, 4>(512, "2 waves/SIMD + 4 VALU"); run<2, 6>(512, "2 waves/SIMD + 6 VALU");
    run<4, 6>(256, "1 wave/SIMD + 6 VALU"); run<4, 4>(256, "1 wave/SIMD + 4 VALU");
    return 0;
}
