// Dedicated kernels for the two image-side layers of the VQ-VAE, whose channel count (3, carried as 4) is
// degenerate for 32x32 MFMA tiles: on the generic engine the decoder's last ConvTranspose (128 -> 3) spends
// 8x its FLOPs on padding columns and re-reads its input once per stride phase and tap (PMC: 3.5 GB of
// fetches per launch for 268 MB of input), and the two weight gradients run a 64-row GEMM on a 128-row tile.
// Both are HBM-bound by nature (one pass over the 128-channel activation), so they are written as LDS-tiled
// fp32 FMA kernels that read the big activation exactly once.
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

// ------------------------------------------------------------------------------------------------
// weight gradient of a conv whose INPUT has 4 (padded) channels:
//   partial[blk][(tap*4 + ci)][co] = sum_{pixels of blk} x[in(pix, tap)][ci] * g[pix][co]
// g (N, Ho, Wo, Co) is read exactly once, fully coalesced; the 4-channel x patch of a pixel is a handful of
// 16-byte loads shared by the whole wave.  taps = Kh*Kw <= 16.  Partials use the engine's layout, so the
// generic unpack kernel (fixed-order reduction + transposition to (Co, Ci, Kh, Kw)) finishes the job.
// ------------------------------------------------------------------------------------------------
// A workgroup walks chunks of 32 consecutive output pixels of one output row: the chunk's g rows (32 x Co) and the
// input rows it touches ((Kh) x ((32-1)*sw + Kw) x 4 floats) are staged in LDS with coalesced 16-byte loads;
// thread (co, pixel lane) then accumulates its TAPS x 4 partial weights with broadcast LDS reads of the patch.
#define TW_PIX 32
template <int KH, int KW>
__global__ __launch_bounds__(256) void lvt_conv4_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                  lvt_conv_geom geo, int chunks_per_block,
                                                                  float *__restrict__ partial) {
    constexpr int TAPS = KH * KW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int Co = geo.Co;
    const int lanes = 256 / Co;                          // pixel lanes per workgroup
    const int co = tid % Co, pl = tid / Co;
    const int xcols = (TW_PIX - 1) * geo.sw + KW;        // input columns touched by one chunk
    float *gs = smem;                                     // [32][Co]
    float *xs = smem + TW_PIX * Co;                       // [KH][xcols][4]
    const int chunks_per_row = geo.Wo / TW_PIX;
    const long long nchunks = (long long)geo.N * geo.Ho * chunks_per_row;
    const long long c0 = (long long)blockIdx.x * chunks_per_block;
    const long long c1 = min(nchunks, c0 + chunks_per_block);
    float acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
    for (long long c = c0; c < c1; ++c) {
        long long r = c;
        const int wc = r % chunks_per_row; r /= chunks_per_row;
        const int ho = r % geo.Ho; const int n = r / geo.Ho;
        const int wo0 = wc * TW_PIX;
        const float *gp = g + (((long long)n * geo.Ho + ho) * geo.Wo + wo0) * Co;
        for (int u = tid; u < TW_PIX * Co / 4; u += 256)
            reinterpret_cast<float4 *>(gs)[u] = reinterpret_cast<const float4 *>(gp)[u];
        const int hi0 = ho * geo.sh - geo.ph, wi0 = wo0 * geo.sw - geo.pw;
        for (int u = tid; u < KH * xcols; u += 256) {
            const int kh = u / xcols, col = u % xcols;
            const int hi = hi0 + kh, wi = wi0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)hi < (unsigned)geo.Hi && (unsigned)wi < (unsigned)geo.Wi)
                v = *reinterpret_cast<const float4 *>(x + (((long long)n * geo.Hi + hi) * geo.Wi + wi) * 4);
            reinterpret_cast<float4 *>(xs)[u] = v;
        }
        __syncthreads();
        for (int p = pl; p < TW_PIX; p += lanes) {
            const float gv = gs[p * Co + co];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const float4 xv = reinterpret_cast<const float4 *>(xs)[kh * xcols + p * geo.sw + kw];
                    float *a = acc[kh * KW + kw];
                    a[0] = fmaf(xv.x, gv, a[0]); a[1] = fmaf(xv.y, gv, a[1]);
                    a[2] = fmaf(xv.z, gv, a[2]); a[3] = fmaf(xv.w, gv, a[3]);
                }
        }
        __syncthreads();
    }
    // combine the pixel lanes in lane order through LDS (reusing gs), then write the workgroup partial
    float *red = gs;
    float *dst = partial + (long long)blockIdx.x * TAPS * 4 * Co;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) red[(t * 4 + ci) * 256 + tid] = acc[t][ci];
    __syncthreads();
    for (int u = tid; u < TAPS * 4 * Co; u += 256) {
        const int row = u / Co, cc = u % Co;
        float sacc = red[row * 256 + cc];
        for (int k = 1; k < lanes; ++k) sacc += red[row * 256 + k * Co + cc];
        dst[u] = sacc;
    }
}

// out[i] = sum_k partial[k*n + i]: one wave per output element; lane l sums rows l, l+64, ... in order and the
// 64 lane sums are combined by a fixed butterfly -> deterministic, and every row is touched by a different lane
// so the reduction over ~1000 workgroup partials is not a serial chain of dependent loads.
__global__ void lvt_reduce_rows_wave_kernel(const float *__restrict__ partial, int rows, int n, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    float s = 0.f;
    for (int k = lane; k < rows; k += 64) s += partial[(long long)k * n + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[i] = s;
}

static long long thin_bw_blocks(const lvt_conv_geom *g, int *chunks_per_block) {
    const long long nchunks = (long long)g->N * g->Ho * (g->Wo / TW_PIX);
    long long blocks = 4 * LVT_NUM_CU;
    if (blocks > nchunks) blocks = nchunks;
    if (blocks < 1) blocks = 1;
    *chunks_per_block = (int)lvt_cdiv(nchunks, blocks);
    return lvt_cdiv(nchunks, *chunks_per_block);
}
static size_t thin_bw_smem(const lvt_conv_geom *g) {
    const size_t stage = (size_t)TW_PIX * g->Co + (size_t)g->Kh * ((TW_PIX - 1) * g->sw + g->Kw) * 4;
    const size_t red = (size_t)g->Kh * g->Kw * 4 * 256;
    return (stage > red ? stage : red) * sizeof(float);
}
extern "C" size_t lvt_conv4_bwd_weight_workspace_bytes(const lvt_conv_geom *g) {
    if (!g || g->Wo % TW_PIX) return 0;
    int cpb;
    const long long blocks = thin_bw_blocks(g, &cpb);
    const int n = g->Kh * g->Kw * 4 * g->Co;
    // workgroup partials + one reduced row + the scratch of the column-sum recursion
    return (size_t)(blocks + 1) * n * sizeof(float);
}
extern "C" int lvt_conv4_bwd_weight(const lvt_conv_geom *g, const float *x, const float *dy, float *dw, int Ci_real,
                                    int Co_real, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(g && x && dy && dw, "conv4_bwd_weight: null pointer");
    LVT_REQUIRE(g->Ci == 4 && g->Kt == 1 && g->Ti == 1 && g->To == 1 && g->Co <= 256 && 256 % g->Co == 0 && g->Co % 4 == 0,
                "conv4_bwd_weight: needs Ci == 4, a 2-D kernel and Co dividing 256");
    LVT_REQUIRE(g->Wo % TW_PIX == 0, "conv4_bwd_weight: Wo must be a multiple of %d", TW_PIX);
    const int taps = g->Kh * g->Kw;
    LVT_REQUIRE((g->Kh == 4 && g->Kw == 4) || (g->Kh == 3 && g->Kw == 3), "conv4_bwd_weight: %dx%d taps not instantiated", g->Kh, g->Kw);
    if (!workspace || workspace_bytes < lvt_conv4_bwd_weight_workspace_bytes(g)) {
        lvt_set_error("conv4_bwd_weight: workspace too small");
        return LVT_EWORKSPACE;
    }
    int cpb;
    const long long blocks = thin_bw_blocks(g, &cpb);
    const int smem = (int)thin_bw_smem(g);
    hipStream_t s = (hipStream_t)stream;
    if (g->Kh == 4) {
        (void)hipFuncSetAttribute((const void *)lvt_conv4_bwd_weight_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((lvt_conv4_bwd_weight_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), smem, s, x, dy, *g, cpb, (float *)workspace);
    } else {
        (void)hipFuncSetAttribute((const void *)lvt_conv4_bwd_weight_kernel<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((lvt_conv4_bwd_weight_kernel<3, 3>), dim3((unsigned)blocks), dim3(256), smem, s, x, dy, *g, cpb, (float *)workspace);
    }
    LVT_CHECK_LAUNCH("lvt_conv4_bwd_weight_kernel");
    // fixed-order reduction of the workgroup partials (one wave per output element), then the generic
    // transposition to (Co, Ci, Kh, Kw)
    const int n = taps * 4 * g->Co;
    float *reduced = (float *)workspace + blocks * n;
    hipLaunchKernelGGL(lvt_reduce_rows_wave_kernel, dim3((unsigned)lvt_cdiv((long long)n * 64, 256)), dim3(256), 0, s,
                       (const float *)workspace, (int)blocks, n, reduced);
    LVT_CHECK_LAUNCH("lvt_reduce_rows_wave_kernel");
    return lvt_unpack_wgrad(reduced, n, 1, dw, taps, 4, g->Co, Ci_real, Co_real, stream);
}
