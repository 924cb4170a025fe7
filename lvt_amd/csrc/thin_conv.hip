// Dedicated kernel for the image-side ConvTranspose of the VQ-VAE decoder, whose output channel count (3, carried
// as 4) is degenerate for 32x32 MFMA tiles: on the generic engine it spends 8x its FLOPs on padding columns and
// re-reads its input once per stride phase and tap (PMC: 3.5 GB of fetches per launch for 268 MB of input).  It is
// HBM-bound by nature (one pass over the 128-channel activation), so it is an LDS-tiled fp32 FMA kernel that reads
// the big activation exactly once.  (The two 4-channel weight gradients run on the engine with a 64-row tile.)
#include "lvt_common.h"

// ------------------------------------------------------------------------------------------------
// y[n, 2q+r, :, c] = act( bias[c] + sum_{ci, taps} x[n, q+d, :, ci] * W[ci][c][kh][kw] ),  k4 s2 p1 ConvTranspose
// x (N, Hi, Wi, Ci) channels-last, y (N, 2Hi, 2Wi, 4) with channel 3 == act(0) padding, W in torch
// ConvTranspose2d layout (Ci, Cr, 4, 4) with Cr <= 3.
// A thread owns the 2x2 output blocks of TC_P input positions; a workgroup owns a 16 x 32 input tile (+1 halo) and
// walks Ci in chunks of 16 staged through LDS.  Every kernel tap k (per dimension) belongs to exactly one
// (output phase r, neighbour offset d) pair -- k=0: (1,+1), k=1: (0,0), k=2: (1,0), k=3: (0,-1) -- so per (ci, c)
// the 16 taps are 16 FMAs into the 4 phase accumulators.  The weights are wave-uniform and are read with SCALAR
// loads (the 16 taps of one (ci, c) are contiguous: one s_load_dwordx16) and enter the FMAs as SGPR operands: the
// LDS pipe only carries the 9 neighbour float4 per 4 channels (the first version staged the weights in LDS too and
// was bound by its 48 broadcast reads per 4 channels: 332 us per launch at the bench shape, 268 us now; the
// remaining stall is the scalar-load latency in front of every 16-tap group).
// ------------------------------------------------------------------------------------------------
#define TC_CK 16
#define TC_LDX 20                       // floats per staged pixel (16 + 4 pad): b128 reads conflict-free
#ifndef TC_P
#define TC_P 1                         // positions per thread (rows lh + 8*p)
#endif
#define TC_TH (8 * TC_P)
#define TC_TW 32

__device__ __forceinline__ float tc_comp(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

__global__ __launch_bounds__(256) void lvt_convt4_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, int N, int Hi, int Wi,
                                                             int Ci, int Cr, int act_tanh, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float xs[(TC_TH + 2) * (TC_TW + 2) * TC_LDX];
    const int tid = threadIdx.x;
    const int tiles_w = (Wi + TC_TW - 1) / TC_TW, tiles_h = (Hi + TC_TH - 1) / TC_TH;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; const int n = b / tiles_h;
    const int h0 = th * TC_TH, w0 = tw * TC_TW;
    const int lw = tid & 31, lh = tid >> 5;               // this thread: input positions (h0+lh+8q, w0+lw)
    float acc[TC_P][2][2][3];
#pragma unroll
    for (int q = 0; q < TC_P; ++q)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[q][rh][rw][c] = 0.f;

    constexpr int PH[4] = {1, 0, 1, 0};                  // output phase of tap k
    constexpr int NB[4] = {2, 1, 1, 0};                  // staged neighbour index (offset + 1) of tap k
    // staging slots of this thread: fixed for the whole kernel, so the address decode runs once; the loads of chunk
    // c0 + 16 are issued before the FMAs of chunk c0 and land in LDS after them (register double buffer) -- with the
    // load -> store pairs inside the chunk loop every workgroup paid several dependent HBM latencies per chunk
    constexpr int NSLOT = (TC_TH + 2) * (TC_TW + 2) * (TC_CK / 4), NLD = (NSLOT + 255) / 256;
    const float *xn = x + (long long)n * Hi * Wi * Ci;
    int goff[NLD], loff[NLD];                   // global float offset inside the image (-1: zero fill), LDS float offset (-1: none)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int u = tid + 256 * i;
        const int q4 = u % (TC_CK / 4); const int pix = u / (TC_CK / 4);
        const int pw = pix % (TC_TW + 2), ph = pix / (TC_TW + 2);
        const int hi = h0 + ph - 1, wi = w0 + pw - 1;
        loff[i] = u < NSLOT ? pix * TC_LDX + q4 * 4 : -1;
        goff[i] = (u < NSLOT && (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi) ? (hi * Wi + wi) * Ci + q4 * 4 : -1;
    }
    float4 pre[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        pre[i] = goff[i] >= 0 ? *reinterpret_cast<const float4 *>(xn + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < Ci; c0 += TC_CK) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (loff[i] >= 0) *reinterpret_cast<float4 *>(&xs[loff[i]]) = pre[i];
        __syncthreads();
        if (c0 + TC_CK < Ci) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                pre[i] = goff[i] >= 0 ? *reinterpret_cast<const float4 *>(xn + goff[i] + c0 + TC_CK) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q4 = 0; q4 < TC_CK / 4; ++q4) {
            float4 a[TC_P][3][3];
#pragma unroll
            for (int q = 0; q < TC_P; ++q)
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
                        a[q][dh][dw] = *reinterpret_cast<const float4 *>(&xs[((lh + 8 * q + dh) * (TC_TW + 2) + lw + dw) * TC_LDX + q4 * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (c < Cr) {
                        const float *wt = w + ((long long)(c0 + q4 * 4 + j) * Cr + c) * 16;      // wave-uniform
#pragma unroll
                        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                            for (int kw = 0; kw < 4; ++kw) {
                                const float wv = wt[kh * 4 + kw];
#pragma unroll
                                for (int q = 0; q < TC_P; ++q)
                                    acc[q][PH[kh]][PH[kw]][c] = fmaf(tc_comp(a[q][NB[kh]][NB[kw]], j), wv, acc[q][PH[kh]][PH[kw]][c]);
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
    const int Ho = 2 * Hi, Wo = 2 * Wi;
#pragma unroll
    for (int q = 0; q < TC_P; ++q) {
        const int qh = h0 + lh + 8 * q, qw = w0 + lw;
        if (qh >= Hi || qw >= Wi) continue;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            float o[2][4];
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = (c < Cr) ? acc[q][rh][rw][c] + bias[c] : 0.f;
                    o[rw][c] = act_tanh ? tanhf(v) : v;
                }
                o[rw][3] = 0.f;
            }
            // the two horizontal phases are adjacent pixels: one 32-byte run per thread, 2 KB per wave row
            float4 *dst = reinterpret_cast<float4 *>(y + (((long long)n * Ho + 2 * qh + rh) * Wo + 2 * qw) * 4);
            dst[0] = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
            dst[1] = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
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
