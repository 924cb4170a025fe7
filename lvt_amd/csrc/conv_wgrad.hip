// Frame-resident weight gradient of the 3x3 / stride 1 / pad 1 convolutions on 16x16 frames:
//     dW[tap][ci][co] = sum over frames and pixels of  x[pixel + tap][ci] * dy[pixel][co]
// The implicit-GEMM form (gemm_engine.hip, A_CONV_M) gathers a shifted copy of x for every tap and re-stages dy for
// every 128-row tile of (tap, ci): staging, not the matrix cores, bounds it (skipping the A-side staging on 7 of 8 k-tiles:
// +29 %, both sides: +48 %, profiles/r02_engine_staging_experiments.txt).  Here a workgroup (8 waves) owns a 32-channel
// chunk of the "patch" operand for ALL nine taps and all 256 channels of the "slab" operand:
//   * per frame the 18x18 patch chunk (frame + zero halo) is split into bf16x3 planes and staged ONCE, pixel-major;
//   * per image row the 16 pixels x 256 channels of the slab operand are staged once (double-buffered);
//   * the reduction index of the MFMAs is the pixel, i.e. the ROW index of both LDS images, so the operand fragments are
//     fetched with the gfx950 transposing read (ds_read_b64_tr_b16: a 16-lane group turns a [4 rows][16 columns] block into
//     per-column 4-vectors): a tap is a row offset into the patch, never a re-staging and never an unaligned access.
// Wave w accumulates the nine [32 patch channels] x [slab channels 32w..32w+31] tiles (144 accumulator registers).
// Frames are split over workgroups; the per-split partial sums go to the same [split][(tap, c_patch)][c_slab] buffer
// layout as the implicit-GEMM path and are reduced in a fixed order (no atomics).
// Roles: Co == 256 -> patch = x (chunks of ci), slab = dy; else Ci == 256 -> patch = dy (chunks of co), slab = x, taps reversed.
#include "lvt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define WG_THREADS 512
#define WG_SUM_PARTS 8                   // pixel ranges per split for the column sums of the slab operand (extra workgroups)
#define WG_PW 18
#define WG_PIX (WG_PW * WG_PW)
#define WG_PP 32                         // patch pixel pitch (bf16): 64 B -> the 4 rows of a transposing read tile the 64 banks
#define WG_PPL (WG_PIX * WG_PP)
#define WG_CQ 256
#define WG_QP (WG_CQ + 32)               // slab pixel pitch (bf16): 576 B = 16 dwords mod 64
#define WG_QPL (16 * WG_QP)
#ifndef LVT_WG_ANTIPHASE_MODES
#define LVT_WG_ANTIPHASE_MODES 2         // bit MODE: the stride-2 kernel (24 MFMAs per staged row) gains, the 3x3 one (54) does not
#endif
#ifndef LVT_WG_S2_PAIR
#define LVT_WG_S2_PAIR 1                 // stride-2 kernel in f16x2: two parity classes per workgroup (0: one, the round-4 form)
#endif
#define LVT_WG_ANTIPHASE(mode) (((LVT_WG_ANTIPHASE_MODES) >> (mode)) & 1)

struct WgParams {
    const float *P, *Q;
    int Cp, N, frames_per_split, nchunks;
    float *partial; long long partial_stride;
    const float *p_amax, *q_amax;        // MATH == 2: max |P|, max |Q| (device scalars)
    // bias gradient = column sums of dy (or, for a transposed layer whose operands arrive swapped, of x), by the same launch.
    // sum_mode 2: the summed tensor is the patch operand -- every workgroup adds up the 32-channel patches it stages (once per
    // frame: not in the row loop; MODE 1: one partial row per (split, parity class), the classes tile the pixels exactly once).  sum_mode 1: dy is the slab operand -- `nsum` EXTRA
    // workgroups (blockIdx >= nmain) stream one range of dy rows each and do nothing else; they run beside the matrix
    // workgroups (the rows are in L2 / MALL from their staging).  Summing inside slab_store instead put a branch into the row loop
    // and cost the stride-2 kernel 36 % (459 -> 625 us: its 24 MFMAs per row leave the scheduler one block to interleave).
    // colsum_partial[split or range][channel].
    int sum_mode, nmain, nsplits, nsum; float *colsum_partial;
};
// MODE 0: 3x3 / stride 1 / pad 1 on 16x16 frames: nine taps over the 18x18 patch, as described above.
// MODE 1: 4x4 / stride 2 / pad 1 between a 32x32 frame (patch operand, Cp channels) and a 16x16 frame (slab operand, 256
//         channels): a workgroup owns ONE parity class (ky & 1, kx & 1) of the sixteen taps -- its four taps (ky >> 1, kx >> 1)
//         are row / column offsets 0..1 into the 17x17 sub-image in[2r + py - 1][2c + px - 1] of the big frame, which is the
//         patch here.  blockIdx decodes to (class, chunk, split); partial rows are (tap16 = ky * 4 + kx, c_patch).
//         f16x2 (two planes: room for a second patch image): a workgroup owns the TWO classes (py, 0) and (py, 1) -- eight taps,
//         24 MFMAs per staged slab row instead of 12 (the slab row costs the same fetch + split + store whatever is multiplied
//         with it, and with 12 MFMAs per row that staging bounded the kernel: 218 TFLOP/s against 388 for the nine-tap form).

__device__ __forceinline__ unsigned wg_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void wg_split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3) {
    p1.x = wg_cvt_pk(v.x, v.y); p1.y = wg_cvt_pk(v.z, v.w);
    const float r0 = v.x - __uint_as_float(p1.x << 16), r1 = v.y - __uint_as_float(p1.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(p1.y << 16), r3 = v.w - __uint_as_float(p1.y & 0xffff0000u);
    p2.x = wg_cvt_pk(r0, r1); p2.y = wg_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(p2.x << 16), s1 = r1 - __uint_as_float(p2.x & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(p2.y << 16), s3 = r3 - __uint_as_float(p2.y & 0xffff0000u);
    p3.x = wg_cvt_pk(s0, s1); p3.y = wg_cvt_pk(s2, s3);
}
// f16x2 arithmetic (LVT_MATH_F16X2; gemm_engine.hip describes the split): a * s = hi + lo with lo = RN16(a s - hi) kept
// UNSCALED here, so that hi hi + hi lo + lo hi accumulate into ONE register set -- nine taps x two sets would not fit.
// lo then resolves down to 2^-24: all 22 + sign bits for the elements within 2^-16 of the operand's max, fewer below
// (absolute error <= 2^-40 max |a| per element: invisible in a reduction over 131072 pixels of mixed magnitudes).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wg_mix_lo(unsigned h, float c) {                  // c - half(h.lo), one rounding (exact here)
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float wg_mix_hi(unsigned h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
// per pair: t = v s (v_pk_mul_f32), hi = RN16(t) (v_cvt_pk_f16_f32), r = t - hi exactly (v_fma_mix_f32), lo = RN16(r)
// (instruction costs: gemm_engine.hip, f16_split_pair)
__device__ __forceinline__ void wg_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const f32x2 t = f32x2{a, b} * s;
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2v));
    const f32x2 r = {wg_mix_lo(ph, t.x), wg_mix_hi(ph, t.y)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
__device__ __forceinline__ void wg_split2(const float4 v, const float s, uint2 &ph, uint2 &pl) {
    wg_split_pair(v.x, v.y, s, ph.x, pl.x);
    wg_split_pair(v.z, v.w, s, ph.y, pl.y);
}
__device__ __forceinline__ float wg_f16_scale(const float *amax, int &unscale) {     // = lvt_f16_scale (gemm_engine.hip)
    if (!amax) return 1.f;
    const int eb = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
    int se = 268 - eb;
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    unscale -= se - 127;
    return __uint_as_float((unsigned)se << 23);
}
// 8 reduction rows (4 + 4) of the lane's column: two transposing reads, `pitch4` = 4 rows further (bf16 elements)
__device__ __forceinline__ bf16x8 wg_frag(const unsigned short *p, int pitch4) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(p + pitch4));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int MODE, int MATH>
__global__ __launch_bounds__(WG_THREADS) void lvt_conv_wgrad_frames_kernel(const WgParams p) {
    constexpr int NCLS = (MODE == 1 && MATH == 2 && LVT_WG_S2_PAIR) ? 2 : 1;         // parity classes (patch images) per workgroup
    constexpr int NTAPS = MODE == 0 ? 9 : 4 * NCLS;
    constexpr int NP = MATH == 2 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) unsigned short lds[NCLS * NP * WG_PPL + 2 * NP * WG_QPL];
    unsigned short *patch = lds, *slab0 = lds + NCLS * NP * WG_PPL;
    int unscale = 0;
    float sp = 1.f, sq = 1.f;
    if (MATH == 2) { sp = wg_f16_scale(p.p_amax, unscale); sq = wg_f16_scale(p.q_amax, unscale); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= p.nmain) {
        // column sums of the slab operand over the frames of split (blockIdx - nmain): thread = (pixel tid >> 6 of eight, channel
        // quad tid & 63), four independent running sums, combined in a fixed order
        // (one workgroup streams ~16 GB/s: the pixels are cut into nsum = 8 x splits ranges so that none outlasts the matrix work)
        const int split = blockIdx.x - p.nmain;                                    // index of the pixel range
        const long long per = (((long long)p.N * 256 + p.nsum - 1) / p.nsum + 7) / 8 * 8;
        const long long px0 = split * per, px1 = min((long long)p.N * 256, px0 + per);
        const float *src = p.Q + (tid & 63) * 4;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0, c3 = c0;
        long long px = px0 + (tid >> 6);
        for (; px + 24 < px1; px += 32) {
            const float4 a = *reinterpret_cast<const float4 *>(src + px * WG_CQ), b = *reinterpret_cast<const float4 *>(src + (px + 8) * WG_CQ);
            const float4 c = *reinterpret_cast<const float4 *>(src + (px + 16) * WG_CQ), d = *reinterpret_cast<const float4 *>(src + (px + 24) * WG_CQ);
            c0.x += a.x; c0.y += a.y; c0.z += a.z; c0.w += a.w; c1.x += b.x; c1.y += b.y; c1.z += b.z; c1.w += b.w;
            c2.x += c.x; c2.y += c.y; c2.z += c.z; c2.w += c.w; c3.x += d.x; c3.y += d.y; c3.z += d.z; c3.w += d.w;
        }
        for (; px < px1; px += 8) {
            const float4 a = *reinterpret_cast<const float4 *>(src + px * WG_CQ);
            c0.x += a.x; c0.y += a.y; c0.z += a.z; c0.w += a.w;
        }
        float *scratch = reinterpret_cast<float *>(lds);
        *reinterpret_cast<float4 *>(&scratch[tid * 4]) = make_float4((c0.x + c1.x) + (c2.x + c3.x), (c0.y + c1.y) + (c2.y + c3.y),
                                                                     (c0.z + c1.z) + (c2.z + c3.z), (c0.w + c1.w) + (c2.w + c3.w));
        __syncthreads();
        if (tid < WG_CQ) {
            float t = 0.f;
            for (int w = 0; w < WG_THREADS / 64; ++w) t += scratch[(w * 64 + (tid >> 2)) * 4 + (tid & 3)];
            p.colsum_partial[(long long)split * WG_CQ + tid] = t;
        }
        return;
    }
    // workgroup b runs on XCD b % 8 and every XCD has its own L2: the chunks of one split read the same frames (all of
    // the slab operand, the same patch pixels), so the split index is the fast one -- its low bits pick the XCD
    const int nsplits = p.nsplits;
    const int split = blockIdx.x % nsplits;
    const int job = blockIdx.x / nsplits;
    const int pc = MODE == 1 ? (NCLS == 2 ? job >> 1 : job >> 2) : job;                  // 32-channel chunk of the patch operand
    const int cls = MODE == 1 ? (NCLS == 2 ? (job & 1) << 1 : job & 3) : 0;              // (first) parity class (py, px)
    const int f0 = split * p.frames_per_split, f1 = min(p.N, f0 + p.frames_per_split);
    const int Cp = p.Cp;

    // (MODE 1 walks the 17x17 pixels its taps read, placed at the 18-pixel pitch: 5 passes instead of 6, 8 registers per image)
    constexpr int PWALK = MODE == 1 ? 17 : WG_PW;
    constexpr int PUNITS = PWALK * PWALK * 8, PPASS = (PUNITS + WG_THREADS - 1) / WG_THREADS;
    float4 pv[NCLS][PPASS], qv[2];
    // (a thread stages the same channel quad tid & 7 of every patch pixel it touches)
    const bool sum_p = p.sum_mode == 2;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto patch_fetch = [&](int f, int c) {         // c: image of class cls + c
        const float *xf = p.P + (long long)f * (MODE == 1 ? 1024 : 256) * Cp + pc * 32;
#pragma unroll
        for (int j = 0; j < PPASS; ++j) {
            const int u = tid + WG_THREADS * j;
            const int pp = u >> 3, q = u & 7;
            const int py = pp / PWALK, px = pp - py * PWALK;
            if (MODE == 1) {
                const int iy = 2 * py + (cls >> 1) - 1, ix = 2 * px + ((cls + c) & 1) - 1;
                const bool ok = u < PUNITS && (unsigned)iy < 32u && (unsigned)ix < 32u;
                pv[c][j] = ok ? *reinterpret_cast<const float4 *>(xf + (iy * 32 + ix) * Cp + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const bool ok = u < PUNITS && (unsigned)(py - 1) < 16u && (unsigned)(px - 1) < 16u;
                pv[c][j] = ok ? *reinterpret_cast<const float4 *>(xf + ((py - 1) * 16 + (px - 1)) * Cp + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto patch_store = [&](int c) {
#pragma unroll
        for (int j = 0; j < PPASS; ++j) {
            const int u = tid + WG_THREADS * j;
            if (u < PUNITS) {
                if (sum_p) { cs.x += pv[c][j].x; cs.y += pv[c][j].y; cs.z += pv[c][j].z; cs.w += pv[c][j].w; }    // (halo / out-of-range: zeros)
                const int pp = u >> 3;
                unsigned short *d = patch + c * (NP * WG_PPL) + (MODE == 1 ? pp + pp / PWALK : pp) * WG_PP + (u & 7) * 4;
                if (MATH == 2) {
                    uint2 ph, pl;
                    wg_split2(pv[c][j], sp, ph, pl);
                    *reinterpret_cast<uint2 *>(d) = ph;
                    *reinterpret_cast<uint2 *>(d + WG_PPL) = pl;
                } else {
                    uint2 p1, p2, p3;
                    wg_split4(pv[c][j], p1, p2, p3);
                    *reinterpret_cast<uint2 *>(d) = p1;
                    *reinterpret_cast<uint2 *>(d + WG_PPL) = p2;
                    *reinterpret_cast<uint2 *>(d + 2 * WG_PPL) = p3;
                }
            }
        }
    };
    auto slab_fetch = [&](int f, int y) {          // 16 pixels x 256 channels = 16 KB contiguous
        const float *src = p.Q + ((long long)f * 256 + y * 16) * WG_CQ;
#pragma unroll
        for (int j = 0; j < 2; ++j) qv[j] = *reinterpret_cast<const float4 *>(src + (tid + WG_THREADS * j) * 4);
    };
    auto slab_store = [&](unsigned short *slab) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = tid + WG_THREADS * j;
            unsigned short *d = slab + (u >> 6) * WG_QP + (u & 63) * 4;
            if (MATH == 2) {
                uint2 ph, pl;
                wg_split2(qv[j], sq, ph, pl);
                *reinterpret_cast<uint2 *>(d) = ph;
                *reinterpret_cast<uint2 *>(d + WG_QPL) = pl;
            } else {
                uint2 p1, p2, p3;
                wg_split4(qv[j], p1, p2, p3);
                *reinterpret_cast<uint2 *>(d) = p1;
                *reinterpret_cast<uint2 *>(d + WG_QPL) = p2;
                *reinterpret_cast<uint2 *>(d + 2 * WG_QPL) = p3;
            }
        }
    };

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment addressing of the transposing read: lane i of a 16-lane group supplies the address of row (i >> 2),
    // columns 4 * (i & 3) .. +3 of its group's [4][16] block and receives column i, rows 0..3 (tools/ubench/tr_probe.hip)
    const int li = lane & 15, g1 = (lane >> 4) & 1, half = lane >> 5;
    const int rowoff = 8 * half + (li >> 2), coloff = 16 * g1 + 4 * (li & 3);
    const int boff = rowoff * WG_QP + wave * 32 + coloff;
    const int aoff = rowoff * WG_PP + coloff;

    // The two waves of a SIMD (w and w + 4) belong to this workgroup and meet at every barrier, so left alone they would both
    // split / store the next slab row and then both queue on the matrix pipe.  Waves 4..7 ("early") therefore store row r + 1
    // BEFORE their MFMAs of row r (their fetch runs two rows ahead), waves 0..3 after them: on every SIMD one wave converts
    // while the other multiplies.  (Both orders sit between the same two barriers; each thread stores its own part of a row.)
    const bool early = LVT_WG_ANTIPHASE(MODE) && NCLS == 1 && ((wave >> 2) & 1);     // (two images: 24 MFMAs per row in f16x2, in phase is 2 % faster)
    const int r_end = f1 * 16;
    auto slab_fetch_row = [&](int r) { if (r < r_end) slab_fetch(r >> 4, r & 15); };
    if (f0 < f1) {
#pragma unroll
        for (int c = 0; c < NCLS; ++c) patch_fetch(f0, c);
        slab_fetch(f0, 0);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) patch_store(c);
        slab_store(slab0);
        if (early) slab_fetch_row(f0 * 16 + 1);
    }
    __syncthreads();
    int buf = 0;
    for (int f = f0; f < f1; ++f) {
        const bool next_frame = f + 1 < f1;
        for (int y = 0; y < 16; ++y) {
            const bool last_row = y == 15;
            const int r = f * 16 + y;
            if (early) {
                if (r + 1 < r_end) slab_store(slab0 + (buf ^ 1) * (NP * WG_QPL));
                slab_fetch_row(r + 2);
            } else {
                slab_fetch_row(r + 1);
            }
            // (two images: the second one is requested a row earlier, so that its 6 loads are not queued behind the first one's)
            if (NCLS == 2 && y == 14 && next_frame) patch_fetch(f + 1, 1);
            if (last_row && next_frame) patch_fetch(f + 1, 0);
            const unsigned short *slab = slab0 + buf * (NP * WG_QPL);
            bf16x8 b[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) b[q] = wg_frag(slab + boff + q * WG_QPL, 4 * WG_QP);
            constexpr int TD = MODE == 0 ? 3 : 2;                                         // taps per dimension
            if constexpr (MATH == 2) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c)
#pragma unroll
                for (int dy = 0; dy < TD; ++dy) {
                    bf16x8 a[TD][2];
#pragma unroll
                    for (int dx = 0; dx < TD; ++dx)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            a[dx][q] = wg_frag(patch + c * (NP * WG_PPL) + ((y + dy) * WG_PW + dx) * WG_PP + aoff + q * WG_PPL, 4 * WG_PP);
                    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};                    // lo hi, hi lo, hi hi
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int dx = 0; dx < TD; ++dx)
                            acc[(c * TD + dy) * TD + dx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(f16x8, a[dx][TA[t]]), __builtin_bit_cast(f16x8, b[TB[t]]), acc[(c * TD + dy) * TD + dx], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int dy = 0; dy < TD; ++dy) {
                // one tap row at a time: consecutive MFMAs go to different accumulators
                bf16x8 a[TD][3];
#pragma unroll
                for (int dx = 0; dx < TD; ++dx)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        a[dx][q] = wg_frag(patch + ((y + dy) * WG_PW + dx) * WG_PP + aoff + q * WG_PPL, 4 * WG_PP);
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};      // smallest terms first
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int dx = 0; dx < TD; ++dx)
                        acc[dy * TD + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dx][TA[t]], b[TB[t]], acc[dy * TD + dx], 0, 0, 0);
            }
            }
            if (!early && r + 1 < r_end) slab_store(slab0 + (buf ^ 1) * (NP * WG_QPL));
            if (last_row && next_frame) {
                __syncthreads();             // every wave is done with this frame's patch
#pragma unroll
                for (int c = 0; c < NCLS; ++c) patch_store(c);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    if (sum_p) {
        // threads that staged the same channel quad are combined in thread order through LDS (every wave is past the last barrier
        // of the loop, the operand images are dead): a fixed order
        float *scratch = reinterpret_cast<float *>(lds);
        *reinterpret_cast<float4 *>(&scratch[tid * 4]) = cs;              // [tid >> 3][quad 0..7]
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
            for (int i = 0; i < WG_THREADS / 8; ++i) t += scratch[(i * 8 + (tid >> 2)) * 4 + (tid & 3)];
            // (MODE 1: one partial row per (split, class) or, with two images per workgroup, per (split, py))
            p.colsum_partial[((long long)split * (MODE == 1 ? 4 / NCLS : 1) + (NCLS == 2 ? cls >> 1 : cls)) * Cp + pc * 32 + tid] = t;
        }
    }
    // partial[split][tap * Cp + pc * 32 + m][32 * wave + n]: lane = column n, 16 rows m per register file
    float *out = p.partial + (long long)split * p.partial_stride + (long long)(pc * 32) * WG_CQ + wave * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        // MODE 1: accumulator (a, b) of class (py, px) is tap (ky, kx) = (2a + py, 2b + px) of the 4x4 kernel
        // (two images: accumulator (c, a, b) belongs to class cls + c)
        const int trow = MODE == 0 ? t : (2 * ((t >> 1) & 1) + (cls >> 1)) * 4 + 2 * (t & 1) + ((cls + (t >> 2)) & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            out[((long long)trow * Cp + m) * WG_CQ] = MATH == 2 ? ldexpf(acc[t][r], unscale) : acc[t][r];
        }
    }
}

// swapped roles: partial[split][(t', co)][ci] -> dw[co][ci][8 - t'], fixed summation order over the splits
__global__ void lvt_unpack_wgrad_swapped_kernel(const float *__restrict__ partial, long long stride, int splits,
                                                float *__restrict__ dw, int Ci, int Co, int Ci_real, int Co_real) {
    const long long total = 9LL * Co * Ci;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            s0 += partial[k * stride + i]; s1 += partial[(k + 1) * stride + i];
            s2 += partial[(k + 2) * stride + i]; s3 += partial[(k + 3) * stride + i];
        }
        for (; k < splits; ++k) s0 += partial[k * stride + i];
        const int ci = i % Ci; long long t = i / Ci;
        const int co = t % Co; const int tr = t / Co;
        if (co < Co_real && ci < Ci_real) dw[((long long)co * Ci_real + ci) * 9 + (8 - tr)] = (s0 + s1) + (s2 + s3);
    }
}

// db[c] = sum over the splits of colsum_partial[split][c]: one wave per channel, lanes over the splits, butterfly (a fixed tree)
__global__ void lvt_wgrad_bias_reduce_kernel(const float *__restrict__ colsum_partial, int splits, int C, int Co_real,
                                             float *__restrict__ db) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (c >= Co_real) return;
    float s = 0.f;
    for (int k = lane; k < splits; k += 64) s += colsum_partial[(long long)k * C + c];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) db[c] = s;
}

// ---- host side (called from lvt_conv3d_bwd_weight in gemm_engine.hip) -----------------------------------------------
// 0: not served; 1: 3x3, patch = x / slab = dy; 2: 3x3 swapped; 3: 4x4 stride 2, patch = x (32x32 frames) / slab = dy (256 ch)
static int wg_role(const lvt_conv_geom *g, int flags = 0) {
    static const int off = getenv("LVT_NO_FRAME_WGRAD") ? 1 : 0;
    static const int off2 = getenv("LVT_NO_FRAME_WGRAD_S2") ? 1 : 0;
    if (off || (flags & LVT_MATH_F32)) return 0;                // (bf16x3 and f16x2 are both served)
    if (g->Kt != 1 || g->pt != 0 || g->Ti != 1 || g->To != 1 || g->st != 1 || g->Ho != 16 || g->Wo != 16) return 0;
    if (g->Kh == 3 && g->Kw == 3 && g->sh == 1 && g->sw == 1 && g->ph == 1 && g->pw == 1 && g->Hi == 16 && g->Wi == 16) {
        if (g->Co == WG_CQ && g->Ci % 32 == 0) return 1;
        if (g->Ci == WG_CQ && g->Co % 32 == 0) return 2;
    }
    if (!off2 && g->Kh == 4 && g->Kw == 4 && g->sh == 2 && g->sw == 2 && g->ph == 1 && g->pw == 1 && g->Hi == 32 && g->Wi == 32 &&
        g->Co == WG_CQ && g->Ci % 32 == 0)
        return 3;
    return 0;
}
static int wg_splits(const lvt_conv_geom *g, int role, bool f16) {
    // (role 3: four parity classes per chunk, two per workgroup in f16x2)
    const int jobs = role == 3 ? (f16 && LVT_WG_S2_PAIR ? 2 : 4) * (g->Ci / 32) : (role == 1 ? g->Ci : g->Co) / 32;
    int s = 256 / jobs;                            // one workgroup per CU (118 KB of LDS each): a single full wave
    if (s > g->N) s = g->N;
    return s < 1 ? 1 : s;
}
int lvt_wgrad_frames_role(const lvt_conv_geom *g, int flags) { return wg_role(g, flags); }
size_t lvt_wgrad_frames_workspace_bytes(const lvt_conv_geom *g) {
    const int role = wg_role(g);
    if (!role) return 0;
    const size_t bias_rows = (size_t)WG_SUM_PARTS * g->Co > (size_t)4 * g->Ci ? (size_t)WG_SUM_PARTS * g->Co : (size_t)4 * g->Ci;
    return (size_t)wg_splits(g, role, true) * ((size_t)g->Kh * g->Kw * g->Ci * g->Co + bias_rows) * sizeof(float);     // + the bias partials
}
int lvt_wgrad_frames_launch(const lvt_conv_geom *g, const float *x, const float *dy, float *dw, int Ci_real, int Co_real,
                            void *workspace, hipStream_t s, void (*unpack_plain)(const float *, long long, int, float *,
                                                                                 const lvt_conv_geom *, int, int, hipStream_t),
                            const float *x_amax, const float *dy_amax, float *db, int db_of_x) {
    const int role = wg_role(g);
    const bool f16 = x_amax && dy_amax;                  // LVT_MATH_F16X2 (checked by the caller)
    WgParams p;
    p.P = role == 2 ? dy : x; p.Q = role == 2 ? x : dy;
    p.p_amax = role == 2 ? dy_amax : x_amax; p.q_amax = role == 2 ? x_amax : dy_amax;
    p.Cp = role == 2 ? g->Co : g->Ci; p.N = g->N; p.nchunks = p.Cp / 32;
    const int splits = wg_splits(g, role, f16);
    const int ncls = role == 3 ? (f16 && LVT_WG_S2_PAIR ? 2 : 4) : 1;         // class jobs per chunk
    p.frames_per_split = (g->N + splits - 1) / splits;
    p.partial = (float *)workspace; p.partial_stride = (long long)g->Kh * g->Kw * g->Ci * g->Co;
    // dy is the patch operand in role 2, the slab (256 = Co channels) otherwise; x (db_of_x: role 3 only) is the patch operand there
    p.sum_mode = db ? ((role == 2 || db_of_x) ? 2 : 1) : 0;
    p.colsum_partial = p.partial + (long long)splits * p.partial_stride;
    p.nsplits = splits;
    p.nmain = ncls * p.nchunks * splits;
    p.nsum = p.sum_mode == 1 ? WG_SUM_PARTS * splits : ncls * splits;
    const unsigned grid = (unsigned)(p.nmain + (p.sum_mode == 1 ? p.nsum : 0));
    if (role == 3 && f16)
        hipLaunchKernelGGL((lvt_conv_wgrad_frames_kernel<1, 2>), dim3(grid), dim3(WG_THREADS), 0, s, p);
    else if (role == 3)
        hipLaunchKernelGGL((lvt_conv_wgrad_frames_kernel<1, 1>), dim3(grid), dim3(WG_THREADS), 0, s, p);
    else if (f16)
        hipLaunchKernelGGL((lvt_conv_wgrad_frames_kernel<0, 2>), dim3(grid), dim3(WG_THREADS), 0, s, p);
    else
        hipLaunchKernelGGL((lvt_conv_wgrad_frames_kernel<0, 1>), dim3(grid), dim3(WG_THREADS), 0, s, p);
    LVT_CHECK_LAUNCH("lvt_conv_wgrad_frames_kernel");
    if (role != 2) {
        unpack_plain(p.partial, p.partial_stride, splits, dw, g, Ci_real, Co_real, s);
    } else {
        const long long total = 9LL * g->Ci * g->Co;
        hipLaunchKernelGGL(lvt_unpack_wgrad_swapped_kernel, dim3((unsigned)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096)),
                           dim3(256), 0, s, (const float *)p.partial, p.partial_stride, splits, dw, g->Ci, g->Co, Ci_real, Co_real);
    }
    LVT_CHECK_LAUNCH("wgrad unpack");
    if (db) {
        hipLaunchKernelGGL(lvt_wgrad_bias_reduce_kernel, dim3((unsigned)lvt_cdiv(Co_real, 4)), dim3(256), 0, s,
                           (const float *)p.colsum_partial, p.nsum, db_of_x ? g->Ci : g->Co, db_of_x ? Ci_real : Co_real, db);
        LVT_CHECK_LAUNCH("lvt_wgrad_bias_reduce_kernel");
    }
    return LVT_OK;
}
