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

// ------------------------------------------------------------------------------------------------
// The same layer on the matrix cores (round 4, LVT_MATH_F16X2): as a GEMM the rows are the INPUT positions, the 16 columns are
// (output phase rh, rw; channel c carried as 4) and the reduction runs over (neighbour dh, dw in 3 x 3; ci) = 9 x 128, with a
// zero weight where tap (phase, neighbour) does not exist (16 of the 36 (phase, neighbour) pairs do): 2.25x the useful products,
// on a pipe that has them to spare -- 58 GFLOP executed per 512 frames against an HBM pass of 268 MB.  v_mfma_f32_16x16x32_f16:
// A = 16 positions of an image row x 32 channels, B = 32 channels x the 16 columns, three products per block (hi hi, hi lo, lo hi;
// gemm_engine.hip describes the split).
//   * persistent workgroups (8 waves, one per CU) walk bands of 8 x 32 input positions; wave w owns row w of the band (two
//     16-position tiles);
//   * the weights are split ONCE per workgroup into fp16 planes [neighbour][column][ci] (78 KB of LDS, row pitch 136 halves:
//     conflict-free 16-byte fragment reads);
//   * the band's 10 x 34 halo patch is staged per 32-channel chunk as two planes (pixel pitch 40 halves), the next chunk's
//     global loads (or the next band's first chunk) in flight under the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
typedef _Float16 tm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tm_f16x2 __attribute__((ext_vector_type(2)));
typedef float tm_f32x4 __attribute__((ext_vector_type(4)));
typedef float tm_f32x2 __attribute__((ext_vector_type(2)));
#define TM_R 8
#define TM_W 32
#define TM_PW (TM_W + 2)
#define TM_PIX ((TM_R + 2) * TM_PW)
#define TM_CK 32
#define TM_CI 128
#define TM_XP (TM_CK + 8)
#define TM_XPL (TM_PIX * TM_XP)
#define TM_WP (TM_CI + 8)
#define TM_WPL (9 * 16 * TM_WP)
#define TM_THREADS 512

__device__ __forceinline__ float tm_mix(unsigned h, int hi, float c) {        // c - 2048 * half(h.lo | h.hi), one rounding (exact here)
    float r;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-2048.f), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-2048.f), "v"(c));
    return r;
}
// a s = hi + lo / 2048 (gemm_engine.hip: f16_split_pair<2048>)
__device__ __forceinline__ void tm_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const tm_f32x2 v = {a, b};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v * s, tm_f16x2));
    const tm_f32x2 t2 = v * (s * 2048.f);
    const tm_f32x2 r = {tm_mix(ph, 0, t2.x), tm_mix(ph, 1, t2.y)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, tm_f16x2));
}
__device__ __forceinline__ void tm_split4(const float4 v, float s, unsigned short *hi, unsigned short *lo) {
    uint2 ph, pl;
    tm_split_pair(v.x, v.y, s, ph.x, pl.x);
    tm_split_pair(v.z, v.w, s, ph.y, pl.y);
    *reinterpret_cast<uint2 *>(hi) = ph;
    *reinterpret_cast<uint2 *>(lo) = pl;
}
__device__ __forceinline__ float tm_scale(const float *amax, int &unscale) {       // = lvt_f16_scale (gemm_engine.hip)
    const int eb = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
    int se = 268 - eb;
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    unscale -= se - 127;
    return __uint_as_float((unsigned)se << 23);
}

__global__ __launch_bounds__(TM_THREADS) void lvt_convt4_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                     const float *__restrict__ bias, int N, int Hi, int Wi, int Cr,
                                                                     int act_tanh, float *__restrict__ y,
                                                                     const float *__restrict__ x_amax, const float *__restrict__ w_amax) {
    __shared__ __attribute__((aligned(16))) unsigned short Wp[2 * TM_WPL];
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2 * TM_XPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    int unscale = 0;
    const float sx = tm_scale(x_amax, unscale), sw = tm_scale(w_amax, unscale);

    // weight planes: column n = (rh, rw, c), neighbour (dh, dw); tap k of a dimension serves (phase, neighbour) =
    // k = 0: (1, 2), 1: (0, 1), 2: (1, 1), 3: (0, 0)
    for (int idx = tid; idx < 9 * 16 * (TM_CI / 4); idx += TM_THREADS) {
        const int ci4 = idx & 31, n = (idx >> 5) & 15, nbr = idx >> 9;
        const int dh = nbr / 3, dw = nbr - 3 * dh, rh = n >> 3, rw = (n >> 2) & 1, c = n & 3;
        const int kh = rh == 0 ? (dh == 1 ? 1 : dh == 0 ? 3 : -1) : (dh == 2 ? 0 : dh == 1 ? 2 : -1);
        const int kw = rw == 0 ? (dw == 1 ? 1 : dw == 0 ? 3 : -1) : (dw == 2 ? 0 : dw == 1 ? 2 : -1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kh >= 0 && kw >= 0 && c < Cr) {
            const float *wp = w + ((long long)(ci4 * 4) * Cr + c) * 16 + kh * 4 + kw;
            v = make_float4(wp[0], wp[Cr * 16], wp[2 * Cr * 16], wp[3 * Cr * 16]);
        }
        unsigned short *d = Wp + (nbr * 16 + n) * TM_WP + ci4 * 4;
        tm_split4(v, sw, d, d + TM_WPL);
    }

    const int nbw = Wi / TM_W, nbh = Hi / TM_R;
    const int nbands = N * nbh * nbw;
    constexpr int NSLOT = TM_PIX * (TM_CK / 4), NLD = (NSLOT + TM_THREADS - 1) / TM_THREADS;
    int loff[NLD], ph_[NLD], pw_[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int u = tid + TM_THREADS * i;
        const int pix = u >> 3, q4 = u & 7;
        loff[i] = u < NSLOT ? pix * TM_XP + q4 * 4 : -1;
        ph_[i] = pix / TM_PW; pw_[i] = pix - ph_[i] * TM_PW;
    }
    long long goff[NLD];                       // float offset of the slot's (pixel, channel quad) in x, chunk 0; -1: zero fill
    float4 pre[2][NLD];                        // two chunks in flight: chunk c + 2 is requested when chunk c has been stored
    auto setup = [&](int band, int &img, int &h0, int &w0) {
        const int bw = band % nbw; const int t = band / nbw;
        const int bh = t % nbh; img = t / nbh;
        h0 = bh * TM_R; w0 = bw * TM_W;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int hi = h0 + ph_[i] - 1, wi = w0 + pw_[i] - 1;
            goff[i] = (loff[i] >= 0 && (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi)
                          ? (((long long)img * Hi + hi) * Wi + wi) * TM_CI + (tid & 7) * 4 : -1;
        }
    };
    auto fetch = [&](int chunk, float4 *dst) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            dst[i] = goff[i] >= 0 ? *reinterpret_cast<const float4 *>(x + goff[i] + chunk * TM_CK) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int n_ = m, rh = n_ >> 3, rw = (n_ >> 2) & 1, c_ = n_ & 3;
    const float bias_c = c_ < Cr ? bias[c_] : 0.f;
    const int Ho = 2 * Hi, Wo = 2 * Wi;

    int band = blockIdx.x, img = 0, h0 = 0, w0 = 0;
    if (band < nbands) { setup(band, img, h0, w0); fetch(0, pre[0]); fetch(1, pre[1]); }
    for (; band < nbands; band += gridDim.x) {
        tm_f32x4 acc[2], acx[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[t][r] = 0.f; acx[t][r] = 0.f; }
        const int cimg = img, ch0 = h0, cw0 = w0;
        static_assert(TM_CI / TM_CK == 4, "the prefetch schedule below is written for four chunks per band");
        const bool more = band + (int)gridDim.x < nbands;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
#ifdef TM_X_NOSTAGE      // (timing experiments: the loaded values still have to arrive)
                if (loff[i] >= 0 && pre[c & 1][i].x == 12345.f) Xp[loff[i]] = 1;
#else
                if (loff[i] >= 0) tm_split4(pre[c & 1][i], sx, Xp + loff[i], Xp + TM_XPL + loff[i]);
#endif
            __syncthreads();                                   // (first pass: the weight planes as well)
            // chunk c + 2 into the registers just stored: chunks 2, 3 of this band, then chunks 0, 1 of the next one
            if (c < 2) fetch(c + 2, pre[c & 1]);
            else if (more) {
                if (c == 2) setup(band + gridDim.x, img, h0, w0);
                fetch(c - 2, pre[c & 1]);
            }
#ifndef TM_X_NOBLOCK
#pragma unroll
            for (int nbr = 0; nbr < 9; ++nbr) {
                const int dh = nbr / 3, dw = nbr - 3 * dh;
                const unsigned short *bp = Wp + (nbr * 16 + m) * TM_WP + c * TM_CK + 8 * kq;
                const tm_f16x8 bh = *reinterpret_cast<const tm_f16x8 *>(bp), bl = *reinterpret_cast<const tm_f16x8 *>(bp + TM_WPL);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned short *ap = Xp + ((wave + dh) * TM_PW + 16 * t + m + dw) * TM_XP + 8 * kq;
                    const tm_f16x8 ah = *reinterpret_cast<const tm_f16x8 *>(ap), al = *reinterpret_cast<const tm_f16x8 *>(ap + TM_XPL);
#ifdef TM_X_NOMFMA
                    acc[t][0] += (float)ah[0] + (float)bl[1] + (float)al[2] + (float)bh[3];
#else
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acx[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acx[t], 0, 0, 0);
#endif
                }
            }
#endif
            __syncthreads();                                   // every wave is done with this chunk's patch
        }
        // lane = column (rh, rw, c) of positions (row ch0 + wave, columns cw0 + 16 t + 4 kq + i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = ldexpf(fmaf(acx[t][i], 1.f / 2048.f, acc[t][i]), unscale) + bias_c;
                if (act_tanh) v = tanhf(v);
                if (c_ >= Cr) v = 0.f;
                const int qh = ch0 + wave, qw = cw0 + 16 * t + 4 * kq + i;
                y[(((long long)cimg * Ho + 2 * qh + rh) * Wo + 2 * qw + rw) * 4 + c_] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The image-side strided convolution 4 -> 128 channels, k4 s2 p1 (the first encoder layer and, as the backward-data of the
// ConvTranspose above, the first decoder gradient), LVT_MATH_F16X2.  Its reduction is 16 taps x 4 channels = 64 long: on the
// implicit-GEMM tile engine a 128 x 128 tile is two k-tiles between an im2col gather and a 64 KB epilogue (144 us per launch
// for a 268 MB output, 16 B gathers).  Here a wave owns 32 consecutive output pixels of one image row and all 128 channels:
//   * A operand straight from global memory: lane (pixel m, half h) needs the taps (kh = step, kw = 2h, 2h + 1) x 4 channels of a
//     k-step, i.e. TWO ADJACENT input pixels = one 32-byte run; split in registers, no LDS, next tile's loads under the MFMAs;
//   * B operand: the 64 x 128 weight split once per workgroup into fp16 planes [column][k] (37 KB of LDS);
//   * 48 MFMAs (32x32x16) per tile; the epilogue writes 128-byte runs per pixel (bias / residual / ReLU / mask as on the engine)
//     and folds max |y| for the next layer's scale.  Workgroups are persistent (waves are independent: no barrier in the loop).
// ------------------------------------------------------------------------------------------------
typedef float ic_f32x16 __attribute__((ext_vector_type(16)));
#define IC_KP (64 + 8)                 // halves per weight column
#define IC_PL (128 * IC_KP)
#define IC_THREADS 256
struct IcParams {
    const float *x, *wp, *bias, *res, *mask; float *y;
    int N, Hi, Wi, Ho, Wo, flags;
    const float *x_amax, *w_amax; float *y_amax;
};
template <bool RES, bool MASK>
__global__ __launch_bounds__(IC_THREADS, 2) void lvt_conv4s2_img_kernel(const IcParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short Bp[2 * IC_PL];
    __shared__ float amax_scratch[IC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    int unscale = 0;
    const float sx = tm_scale(p.x_amax, unscale), sw = tm_scale(p.w_amax, unscale);
    // weight planes: Bp[column n][k] = wp[k][n], k = tap * 4 + ci
    for (int idx = tid; idx < 16 * 128; idx += IC_THREADS) {
        const int n = idx & 127, k4 = idx >> 7;
        const float *w = p.wp + (long long)(k4 * 4) * 128 + n;
        unsigned short *d = Bp + n * IC_KP + k4 * 4;
        tm_split4(make_float4(w[0], w[128], w[256], w[384]), sw, d, d + IC_PL);
    }
    __syncthreads();

    const int nwt = p.Wo / 32;
    const long long ntiles = (long long)p.N * p.Ho * nwt, tstep = (long long)gridDim.x * (IC_THREADS / 64);
    float4 a[4][2], an[4][2];
    auto fetch = [&](long long t, float4 (&dst)[4][2]) {
        const int wt = (int)(t % nwt); const long long r = t / nwt;
        const int oh = (int)(r % p.Ho); const long long img = r / p.Ho;
        const int iw0 = 2 * (wt * 32 + m) - 1 + 2 * h;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int ih = 2 * oh - 1 + s_;
            const bool rok = (unsigned)ih < (unsigned)p.Hi;
            const float *src = p.x + ((img * p.Hi + ih) * p.Wi + iw0) * 4;
            dst[s_][0] = (rok && (unsigned)iw0 < (unsigned)p.Wi) ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            dst[s_][1] = (rok && (unsigned)(iw0 + 1) < (unsigned)p.Wi) ? *reinterpret_cast<const float4 *>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float am = 0.f;
    long long t = (long long)blockIdx.x * (IC_THREADS / 64) + wave;
    if (t < ntiles) fetch(t, a);
    for (; t < ntiles; t += tstep) {
        if (t + tstep < ntiles) fetch(t + tstep, an);
        ic_f32x16 acc[4], acx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; acx[j][r] = 0.f; }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            uint4 uh, ul;
            tm_split_pair(a[s_][0].x, a[s_][0].y, sx, uh.x, ul.x); tm_split_pair(a[s_][0].z, a[s_][0].w, sx, uh.y, ul.y);
            tm_split_pair(a[s_][1].x, a[s_][1].y, sx, uh.z, ul.z); tm_split_pair(a[s_][1].z, a[s_][1].w, sx, uh.w, ul.w);
            const tm_f16x8 ah = __builtin_bit_cast(tm_f16x8, uh), al = __builtin_bit_cast(tm_f16x8, ul);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned short *bp = Bp + (32 * j + m) * IC_KP + 16 * s_ + 8 * h;
                const tm_f16x8 bh = *reinterpret_cast<const tm_f16x8 *>(bp), bl = *reinterpret_cast<const tm_f16x8 *>(bp + IC_PL);
                acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acx[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
                acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acx[j], 0, 0, 0);
            }
        }
        // lane = channel 32 j + m of pixels (r & 3) + 8 (r >> 2) + 4 h of the tile: 128-byte runs per pixel and j
        const long long row0 = t * 32;                       // tiles are consecutive 32-pixel runs of the (N, Ho, Wo) raster
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 32 * j + m;
            const float bias = (p.flags & LVT_EPI_BIAS) ? p.bias[col] : 0.f;
            // the residual / mask values of the 16 rows are requested together, before the first store (the compiler must keep
            // a load behind every earlier store to memory it cannot tell apart: one by one they cost a round trip each --
            // 331 us per launch for the masked backward-data use against 69 us for the plain one)
            float rv[16], mv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (row0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + col;
                rv[r] = RES ? p.res[o] : 0.f;
                mv[r] = MASK ? p.mask[o] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (row0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + col;
                float v = ldexpf(fmaf(acx[j][r], 1.f / 2048.f, acc[j][r]), unscale) + bias + rv[r];
                if (p.flags & LVT_EPI_RELU) v = fmaxf(v, 0.f);
                v = mv[r] > 0.f ? v : 0.f;
                p.y[o] = v;
                am = fmaxf(am, lvt_absf(v));
            }
        }
        if (t + tstep < ntiles) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) { a[s_][0] = an[s_][0]; a[s_][1] = an[s_][1]; }
        }
    }
    if (p.y_amax) lvt_block_amax_commit(am, p.y_amax, amax_scratch);
}
// serves: Kt = 1, 4x4, stride 2, pad 1, Ci = 4 (3 carried as 4), Co = 128, Wo % 32 == 0, epilogue flags within
// BIAS | RESIDUAL | RELU | MASK, f16x2 arithmetic (called from lvt_conv3d_fwd, gemm_engine.hip)
bool lvt_conv4s2_img_ok(const lvt_conv_geom *g, int flags) {
    static const int off = getenv("LVT_NO_IMG_CONV") ? 1 : 0;
    return !off && (flags & LVT_MATH_F16X2) && !(flags & ~(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU | LVT_EPI_MASK | LVT_MATH_F16X2)) &&
           g->Kt == 1 && g->Kh == 4 && g->Kw == 4 && g->st == 1 && g->sh == 2 && g->sw == 2 && g->pt == 0 && g->ph == 1 && g->pw == 1 &&
           g->Ti == 1 && g->To == 1 && g->Ci == 4 && g->Co == 128 && g->Wo % 32 == 0 && g->Hi == 2 * g->Ho && g->Wi == 2 * g->Wo;
}
int lvt_conv4s2_img_launch(const lvt_conv_geom *g, const float *x, const float *wp, const float *bias, const float *res,
                           const float *mask, float *y, int flags, const float *x_amax, const float *w_amax, float *y_amax,
                           hipStream_t s) {
    IcParams p;
    p.x = x; p.wp = wp; p.bias = bias; p.res = res; p.mask = mask; p.y = y;
    p.N = g->N; p.Hi = g->Hi; p.Wi = g->Wi; p.Ho = g->Ho; p.Wo = g->Wo; p.flags = flags;
    p.x_amax = x_amax; p.w_amax = w_amax; p.y_amax = y_amax;
    const long long ntiles = (long long)g->N * g->Ho * (g->Wo / 32), wgs = lvt_cdiv(ntiles, IC_THREADS / 64);
    const unsigned grid = (unsigned)(wgs < 2 * LVT_NUM_CU ? wgs : 2 * LVT_NUM_CU);          // persistent: two workgroups per CU
    const bool r_ = flags & LVT_EPI_RESIDUAL, m_ = flags & LVT_EPI_MASK;
    if (r_ && m_) hipLaunchKernelGGL((lvt_conv4s2_img_kernel<true, true>), dim3(grid), dim3(IC_THREADS), 0, s, p);
    else if (r_) hipLaunchKernelGGL((lvt_conv4s2_img_kernel<true, false>), dim3(grid), dim3(IC_THREADS), 0, s, p);
    else if (m_) hipLaunchKernelGGL((lvt_conv4s2_img_kernel<false, true>), dim3(grid), dim3(IC_THREADS), 0, s, p);
    else hipLaunchKernelGGL((lvt_conv4s2_img_kernel<false, false>), dim3(grid), dim3(IC_THREADS), 0, s, p);
    LVT_CHECK_LAUNCH("lvt_conv4s2_img_kernel");
    return LVT_OK;
}

extern "C" int lvt_convt4_fwd(const float *x, const float *w, const float *bias, int N, int Hi, int Wi, int Ci, int Cr,
                              int act_tanh, float *y, int flags, const lvt_amax_io *ax, void *stream) {
    LVT_REQUIRE(x && w && bias && y && N > 0 && Hi > 0 && Wi > 0, "convT4_fwd: bad args");
    LVT_REQUIRE(Ci % TC_CK == 0 && Cr >= 1 && Cr <= 3, "convT4_fwd: needs Ci %% 16 == 0 and 1..3 output channels");
    LVT_REQUIRE(!(flags & LVT_MATH_F16X2) || (ax && ax->a && ax->b), "convT4_fwd: LVT_MATH_F16X2 needs ax->a = max |x|, ax->b = max |w|");
    static const int no_mfma = getenv("LVT_NO_CONVT4_MFMA") ? 1 : 0;
    if ((flags & LVT_MATH_F16X2) && !no_mfma && Ci == TM_CI && Hi % TM_R == 0 && Wi % TM_W == 0 && lvt_aligned16(x) && lvt_aligned16(y) &&
        (long long)N * (Hi / TM_R) * (Wi / TM_W) < 0x7fffffffLL) {
        const long long nbands = (long long)N * (Hi / TM_R) * (Wi / TM_W);
        const unsigned grid = (unsigned)(nbands < LVT_NUM_CU ? nbands : LVT_NUM_CU);       // persistent: one workgroup per CU
        hipLaunchKernelGGL(lvt_convt4_mfma_kernel, dim3(grid), dim3(TM_THREADS), 0, (hipStream_t)stream, x, w, bias, N, Hi, Wi, Cr,
                           act_tanh, y, ax->a, ax->b);
        LVT_CHECK_LAUNCH("lvt_convt4_mfma_kernel");
        return LVT_OK;
    }
    const long long blocks = (long long)N * lvt_cdiv(Hi, TC_TH) * lvt_cdiv(Wi, TC_TW);
    LVT_REQUIRE(blocks < 0x7fffffffLL, "convT4_fwd: grid too large");
    hipLaunchKernelGGL(lvt_convt4_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, N,
                       Hi, Wi, Ci, Cr, act_tanh, y);
    LVT_CHECK_LAUNCH("lvt_convt4_fwd_kernel");
    return LVT_OK;
}
