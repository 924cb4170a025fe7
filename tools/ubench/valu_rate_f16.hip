// Micro-benchmark: issue cost of the VALU instructions that can form the f16x2 split (gfx950), relative to v_add_f32.
// Each wave runs 8 independent chains; cost = time / (instructions per wave).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    const int tid = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[tid * 16 + i];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(v[i]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[8 + c]));
                if (OP == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[c]) : "v"(v[c]), "v"(v[8 + c]));
                if (OP == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(u[c]) : "v"(v[c]), "v"(v[8 + c]));
                if (OP == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(v[c]) : "v"(v[8 + c]), "v"(v[8 + ((c + 1) & 7)]), "v"(u[c]));
                if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[c]) : "v"(v[c]), "v"(v[8 + c]));
                if (OP == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[c]) : "v"(u[c]));
                if (OP == 6) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[c]) : "v"(u[c]));
                if (OP == 7) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[c]) : "v"(v[c]));
                if (OP == 8) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double *)&v[2 * (c & 3)]) : "v"(*(double *)&v[8 + 2 * (c & 3)]), "v"(*(double *)&v[8 + 2 * ((c + 1) & 3)]));
                if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[8 + c]));
                if (OP == 10) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[c]) : "v"(u[c]));
                if (OP == 11) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(u[c]) : "v"(u[(c + 1) & 7]), "v"(u[(c + 2) & 7]));
            }
        }
    }
    float t = 0.f; unsigned s = 0;
    for (int i = 0; i < 16; ++i) t += v[i];
    for (int i = 0; i < 8; ++i) s ^= u[i];
    out[blockIdx.x * 256 + tid] = t + (float)s;
}
template <int OP> static double run(const char *name, float *out, float *in, int blocks, double base) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, in, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double ns = ms * 1e6 / (iters * 64.0 * waves_per_simd);
    printf("%-26s %.1f waves/SIMD  %6.3f ns/inst/SIMD  (%.2fx v_add_f32)\n", name, waves_per_simd, ns, base > 0 ? ns / base : 1.0);
    return ns;
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&in, 256 * 16 * 4);
    (void)hipMemset(in, 0, 256 * 16 * 4);
    const int blocks = 512;
    const double b = run<0>("v_add_f32", out, in, blocks, 0);
    run<1>("v_fma_mixlo_f16", out, in, blocks, b);
    run<2>("v_fma_mixhi_f16", out, in, blocks, b);
    run<3>("v_fma_mix_f32", out, in, blocks, b);
    run<4>("v_cvt_pk_f16_f32", out, in, blocks, b);
    run<5>("v_cvt_f32_f16", out, in, blocks, b);
    run<6>("v_cvt_f32_f16 sdwa WORD_1", out, in, blocks, b);
    run<7>("v_cvt_f16_f32", out, in, blocks, b);
    run<8>("v_pk_mul_f32", out, in, blocks, b);
    run<9>("v_fma_f32", out, in, blocks, b);
    run<10>("v_ldexp_f32", out, in, blocks, b);
    run<11>("v_pack_b32_f16", out, in, blocks, b);
    return 0;
}
