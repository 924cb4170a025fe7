// Dedicated kernel for the image-side ConvTranspose of the VQ-VAE decoder, whose output channel count (3, carried
// as 4) is degenerate for 32x32 MFMA tiles: on the generic engine it spends 8x its FLOPs on padding columns and
// re-reads its input once per stride phase and tap (PMC: 3.5 GB of fetches per launch for 268 MB of input).  It is
// HBM-bound by nature (one pass over the 128-channel activation), so it is an LDS-tiled fp32 FMA kernel that reads
// the big activation exactly once.  (The two 4-channel weight gradients run on the engine with a 64-row tile.)
#include "lvt_common.h"

// ------------------------------------------------------------------------------------------------
// y[n, 2q+r, :, c] = act( bias[c] + sum_{ci, taps} x[n, q+d, :, ci] * W[ci][c][kh][kw] ),  k4 s2 p1 ConvTranspose
// x (N, Hi, Wi, Ci) channels-last, y (N, 2Hi, 2Wi, 4) with channel 3 == act(0) padding, W in torch
// ConvTranspose2d layout (Ci, Cr, 4, 4) with Cr <= 4.
// A thread owns the 2x2 output block of TWO input positions (q and q + 8 rows); a workgroup owns a
// 16 x 32 input tile (+1 halo) and walks Ci in chunks of 16 staged through LDS.
// ------------------------------------------------------------------------------------------------
#define TC_CK 16
#define TC_LDX 20                       // floats per staged pixel (16 + 4 pad): b128 reads conflict-free
#define TC_TH 16
#define TC_TW 32

__device__ __forceinline__ int tc_kh(int r, int d) { return r == 0 ? (d == 0 ? 1 : 3) : (d == 0 ? 2 : 0); }

__global__ __launch_bounds__(256) void lvt_convt4_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, int N, int Hi, int Wi,
                                                             int Ci, int Cr, int act_tanh, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float xs[(TC_TH + 2) * (TC_TW + 2) * TC_LDX];
    __shared__ __attribute__((aligned(16))) float ws[16 * 4 * TC_CK];          // [tap][c][ci]
    const int tid = threadIdx.x;
    const int tiles_w = (Wi + TC_TW - 1) / TC_TW, tiles_h = (Hi + TC_TH - 1) / TC_TH;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; const int n = b / tiles_h;
    const int h0 = th * TC_TH, w0 = tw * TC_TW;
    const int lw = tid & 31, lh = tid >> 5;               // this thread: input (h0+lh, w0+lw) and (h0+lh+8, w0+lw)
    float acc[2][2][2][3];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[p][rh][rw][c] = 0.f;

    for (int c0 = 0; c0 < Ci; c0 += TC_CK) {
        // stage the (16+2) x (32+2) x 16 input tile (zero outside the image) and the weight chunk
        for (int u = tid; u < (TC_TH + 2) * (TC_TW + 2) * (TC_CK / 4); u += 256) {
            const int q4 = u % (TC_CK / 4); const int pix = u / (TC_CK / 4);
            const int pw = pix % (TC_TW + 2), ph = pix / (TC_TW + 2);
            const int hi = h0 + ph - 1, wi = w0 + pw - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi)
                v = *reinterpret_cast<const float4 *>(x + (((long long)n * Hi + hi) * Wi + wi) * Ci + c0 + q4 * 4);
            *reinterpret_cast<float4 *>(&xs[pix * TC_LDX + q4 * 4]) = v;
        }
        for (int u = tid; u < 16 * 4 * TC_CK; u += 256) {
            const int ci = u % TC_CK; const int c = (u / TC_CK) % 4; const int tap = u / (TC_CK * 4);
            ws[u] = (c < Cr) ? w[((long long)(c0 + ci) * Cr + c) * 16 + tap] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q4 = 0; q4 < TC_CK / 4; ++q4) {
#pragma unroll
            for (int dh = -1; dh <= 1; ++dh) {
#pragma unroll
                for (int dw = -1; dw <= 1; ++dw) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(&xs[((lh + 1 + dh) * (TC_TW + 2) + lw + 1 + dw) * TC_LDX + q4 * 4]);
                    const float4 a1 = *reinterpret_cast<const float4 *>(&xs[((lh + 9 + dh) * (TC_TW + 2) + lw + 1 + dw) * TC_LDX + q4 * 4]);
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        if ((rh == 0 && dh == 1) || (rh == 1 && dh == -1)) continue;
#pragma unroll
                        for (int rw = 0; rw < 2; ++rw) {
                            if ((rw == 0 && dw == 1) || (rw == 1 && dw == -1)) continue;
                            const int tap = tc_kh(rh, dh) * 4 + tc_kh(rw, dw);
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const float4 w4 = *reinterpret_cast<const float4 *>(&ws[(tap * 4 + c) * TC_CK + q4 * 4]);
                                float s0 = acc[0][rh][rw][c], s1 = acc[1][rh][rw][c];
                                s0 = fmaf(a0.x, w4.x, s0); s0 = fmaf(a0.y, w4.y, s0); s0 = fmaf(a0.z, w4.z, s0); s0 = fmaf(a0.w, w4.w, s0);
                                s1 = fmaf(a1.x, w4.x, s1); s1 = fmaf(a1.y, w4.y, s1); s1 = fmaf(a1.z, w4.z, s1); s1 = fmaf(a1.w, w4.w, s1);
                                acc[0][rh][rw][c] = s0; acc[1][rh][rw][c] = s1;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    const int Ho = 2 * Hi, Wo = 2 * Wi;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int qh = h0 + lh + 8 * p, qw = w0 + lw;
        if (qh >= Hi || qw >= Wi) continue;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                float o[4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = (c < Cr) ? acc[p][rh][rw][c] + bias[c] : 0.f;
                    o[c] = act_tanh ? tanhf(v) : v;
                }
                o[3] = 0.f;
                *reinterpret_cast<float4 *>(y + (((long long)n * Ho + 2 * qh + rh) * Wo + 2 * qw + rw) * 4) =
                    make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

extern "C" int lvt_convt4_fwd(const float *x, const float *w, const float *bias, int N, int Hi, int Wi, int Ci, int Cr,
                              int act_tanh, float *y, void *stream) {
    LVT_REQUIRE(x && w && bias && y && N > 0 && Hi > 0 && Wi > 0, "convT4_fwd: bad args");
    LVT_REQUIRE(Ci % TC_CK == 0 && Cr >= 1 && Cr <= 3, "convT4_fwd: needs Ci %% 16 == 0 and 1..3 output channels");
    const long long blocks = (long long)N * lvt_cdiv(Hi, TC_TH) * lvt_cdiv(Wi, TC_TW);
    LVT_REQUIRE(blocks < 0x7fffffffLL, "convT4_fwd: grid too large");
    hipLaunchKernelGGL(lvt_convt4_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, N,
                       Hi, Wi, Ci, Cr, act_tanh, y);
    LVT_CHECK_LAUNCH("lvt_convt4_fwd_kernel");
    return LVT_OK;
}
