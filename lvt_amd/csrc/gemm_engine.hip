// MFMA tile engine for gfx950.
//
// (1) lvt_gemm_kernel: one LDS-tiled kernel template (128x128 output tile, 4 waves, BK = 32) that serves
//   * plain / batched GEMMs in NT, NN and TN form            (linear layers, QKV, attention matmuls)
//   * implicit-GEMM 3-D convolution forward                  (im2col gather in the A loader)
//   * convolution backward-data / ConvTranspose forward      (stride-phase decomposed gather)
//   * convolution backward-weight                            (gather on the M side, split-K over pixels)
//   in two arithmetic modes chosen per call (LVT_MATH_F32 in `flags`):
//   - bf16x3 (default): every fp32 operand is split exactly into three bf16 planes while it is staged in LDS and each
//     32x32x16 block is six v_mfma_f32_32x32x16_bf16 (fp32-class accuracy, ~1.6x the fp32 instruction's rate);
//   - f32: v_mfma_f32_32x32x2_f32, both operands k-major in LDS ([k][m], m contiguous) so that the per-MFMA operand
//     fetch is one conflict-free ds_read_b32 per lane; row pads keep the transposing stores conflict-free.
//   Global -> register -> LDS staging with the next tile's global loads issued before the MFMA block of the current tile;
//   16-byte loads along the contiguous dimension; two workgroups resident per CU; split-K writes per-split partial tiles
//   that a second kernel reduces in a fixed order (no float atomics: weight gradients are bit-reproducible).
//
// (2) lvt_conv_patch_kernel<MODE> (round 2): frame-resident convolutions of the 16x16 / 32x32 VQ-VAE frames -- the input
//   patch of a frame is staged ONCE per 32-channel chunk and the taps read it in place (3x3; the phases of the stride-2
//   transposed convolution; the parity classes of the stride-2 convolution).  Staging, not the matrix pipe, bounds (1):
//   profiles/r02_engine_staging_experiments.txt.  The matching weight-gradient kernel lives in conv_wgrad.hip.
#include "lvt_common.h"
#include "epilogue_fast.h"
#include <string.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef LVT_BK
#define LVT_BK 32
#endif
#define BK LVT_BK
#ifndef LVT_MINWAVES
#define LVT_MINWAVES 1
#endif
#define NTHREADS 256

enum { A_KPLAIN = 0, A_MPLAIN = 1, A_CONV_K = 2, A_CONVT_K = 3, A_CONV_M = 4, A_ONEHOT_M = 5, A_PATCH = 6, A_PATCHT = 7 };
enum { B_KPLAIN = 0, B_NPLAIN = 1, B_CONVT_W = 2 };

struct KParams {
    int M, N, K;
    const float *A; long long lda; int a_kb; long long a_skb;
    const float *B; long long ldb; int b_kb; long long b_skb;
    float *C; long long ldc; long long c_plane;
    int batch_inner;
    long long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    float alpha; int flags;
    const float *bias; const float *res; long long ldr; const float *mask; long long ldm;
    int splits; int k_per_split; float *partial; long long partial_stride;
    int vec_epi;             // C / res / mask / partial rows are 16-byte aligned and N % 4 == 0: float4 epilogue
    float *colsum_partial;   // bwd-weight: [splits][N] column sums of B (= bias gradient), written by the m0 == 0 tiles
    const float *a_amax, *b_amax;   // f16x2 arithmetic: device scalars >= max |A|, max |B| (NULL: the operand is used unscaled)
    const float *a_amax2, *b_amax2; // optional second bound per operand (the larger one counts)
    float *c_amax;                  // any arithmetic: when non-NULL, max |C| is folded in (integer atomic max on the bit pattern)
    lvt_conv_geom g;
    const char *b_img;       // frame-resident convolutions, f16x2: the weight tiles as ready LDS images (lvt_conv3d_weight_images), or NULL
    int Tq, Hq, Wq;          // A_CONVT_K: per-phase output extents
    int jT, jH, jW;          // A_CONVT_K: taps per phase and dimension
    // A_ONEHOT_M: A(m = slot*V + code, k = row) = (idx[b*bstride + off[slot] + pos*pstride] == code),
    // row = b*P + pos.  The transpose of a one-hot matrix, generated on the fly (never materialised).
    const long long *oh_idx; long long oh_bstride, oh_pstride; int oh_P, oh_V; int oh_off[32];
};

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// ---- bf16x3 split (opt-in math mode): a = a1 + a2 + a3 with three bf16 terms = 24 mantissa bits, every
// residual exact in fp32.  v_cvt_pk_bf16_f32 rounds to nearest even and packs (src0 -> low half).
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// 4 floats -> three planes of 4 packed bf16 (uint2 = elements 0..3 in memory order)
__device__ __forceinline__ void split3(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3) {
    p1.x = cvt_pk_bf16(v.x, v.y); p1.y = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - bf_lo(p1.x), r1 = v.y - bf_hi(p1.x), r2 = v.z - bf_lo(p1.y), r3 = v.w - bf_hi(p1.y);
    p2.x = cvt_pk_bf16(r0, r1); p2.y = cvt_pk_bf16(r2, r3);
    const float s0 = r0 - bf_lo(p2.x), s1 = r1 - bf_hi(p2.x), s2 = r2 - bf_lo(p2.y), s3 = r3 - bf_hi(p2.y);
    p3.x = cvt_pk_bf16(s0, s1); p3.y = cvt_pk_bf16(s2, s3);
}
#define HLD 40          // bf16 per staged row: BK (32) + 8 pad -> 80-byte rows
// Staged-row placement of the bf16 planes.  128-row tiles are stored "4x32 transposed" (logical row m lives at
// physical row (m & 3) * 32 + (m >> 2)) with 64 pad bytes after every 32 physical rows: the 16-byte MFMA operand
// reads and the 8-byte stores of the k-contiguous loaders are bank-conflict free, the 8-byte stores of the
// m-contiguous loaders (4 consecutive rows per lane) are 2-way.  Narrow (32-row) tiles keep the identity order.
template <int ROWS> __device__ __forceinline__ int hrow(int m) {
    if (ROWS >= 128) {
        const int r = (m & ~127) + ((m & 3) << 5) + ((m & 127) >> 2);
        return r * HLD + (r >> 5) * 32;
    }
    return m * HLD;
}
template <int ROWS> struct HPlane { static constexpr int SIZE = ROWS * HLD + (ROWS >= 128 ? ROWS : 0); };
// 4 consecutive k of one row -> one 8-byte store per plane
template <int ROWS> __device__ __forceinline__ void store_split_k(unsigned short *lds, int row, int k4, const float4 v) {
    uint2 p1, p2, p3;
    split3(v, p1, p2, p3);
    unsigned short *d = lds + hrow<ROWS>(row) + k4;
    *reinterpret_cast<uint2 *>(d) = p1;
    *reinterpret_cast<uint2 *>(d + HPlane<ROWS>::SIZE) = p2;
    *reinterpret_cast<uint2 *>(d + 2 * HPlane<ROWS>::SIZE) = p3;
}
// 4 consecutive rows at one k (narrow tiles only) -> transposing 2-byte stores
template <int ROWS> __device__ __forceinline__ void store_split_m(unsigned short *lds, int row4, int k, const float4 v) {
    uint2 p[3];
    split3(v, p[0], p[1], p[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        unsigned short *d = lds + q * HPlane<ROWS>::SIZE + k;
        d[hrow<ROWS>(row4)] = (unsigned short)(p[q].x & 0xffffu); d[hrow<ROWS>(row4 + 1)] = (unsigned short)(p[q].x >> 16);
        d[hrow<ROWS>(row4 + 2)] = (unsigned short)(p[q].y & 0xffffu); d[hrow<ROWS>(row4 + 3)] = (unsigned short)(p[q].y >> 16);
    }
}
// m-contiguous fetch with 4 consecutive k per lane: a 4(k) x 4(m) register block is transposed in registers
// and leaves as one 8-byte store per row and plane
template <int ROWS> __device__ __forceinline__ void store_split_block(unsigned short *lds, int row4, int k4, const float4 *v) {
    store_split_k<ROWS>(lds, row4 + 0, k4, make_float4(v[0].x, v[1].x, v[2].x, v[3].x));
    store_split_k<ROWS>(lds, row4 + 1, k4, make_float4(v[0].y, v[1].y, v[2].y, v[3].y));
    store_split_k<ROWS>(lds, row4 + 2, k4, make_float4(v[0].z, v[1].z, v[2].z, v[3].z));
    store_split_k<ROWS>(lds, row4 + 3, k4, make_float4(v[0].w, v[1].w, v[2].w, v[3].w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- f16x2 split (LVT_MATH_F16X2): a * s = hi + 2^-11 * lo with two fp16 terms.  s is an exact power of two taken from the
// operand's max |a| (lvt_f16_scale: max |a| * s in [2^14, 2^15), so hi never overflows), hi = RN16(a s), the residual
// a s - hi is exact in fp32 and lo = RN16(2^11 (a s - hi)) keeps it to 2^-23 |a| -- both planes live in the same binades,
// so the full 22 + sign bits hold for every element down to 2^-27 max |a| and degrade gradually below that.  A block is
// three v_mfma_f32_32x32x16_f16: hi hi into one accumulator, hi lo + lo hi into a second one that is added with weight 2^-11
// at the end (dropped: lo lo <= 2^-22 |a||b| worst case, 2^-24.6 rms -- the size of one fp32 rounding).
// Instruction choice (tools/ubench/valu_rate_f16.hip, issue cost relative to v_add_f32): v_fma_mixlo/hi_f16 3.3, v_fma_mix_f32
// 1.7, v_cvt_pk_f16_f32 1.7 (two elements), v_pk_mul_f32 1.75 (two elements).  Per PAIR of elements: t = v s (v_pk_mul_f32),
// hi = RN16(t) (v_cvt_pk_f16_f32), t2 = v (2048 s) (v_pk_mul_f32), r = t2 - 2048 hi exactly (v_fma_mix_f32 reads the fp16 half
// of hi directly), lo = RN16(r) (v_cvt_pk_f16_f32): 5.2 units per element; the first version (three v_fma_mix*_f16 forms per
// element) cost 8.3 and made the main loops VALU-bound (profiles/r04_wide_gemm_loop_experiments.txt).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16_mix_lo(unsigned h, float k, float c) {        // half(h.lo) * k + c, one rounding
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
__device__ __forceinline__ float f16_mix_hi(unsigned h, float k, float c) {        // half(h.hi) * k + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
// LOSCALE = 2048: the scaled low term of this file; LOSCALE = 1: the unscaled one of conv_wgrad.hip
template <int LOSCALE>
__device__ __forceinline__ void f16_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const f32x2 v = {a, b};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v * s, f16x2v));
    const f32x2 t2 = v * (s * (float)LOSCALE);
    const f32x2 r = {f16_mix_lo(ph, -(float)LOSCALE, t2.x), f16_mix_hi(ph, -(float)LOSCALE, t2.y)};       // exact
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
__device__ __forceinline__ void split2(const float4 v, const float s, uint2 &ph, uint2 &pl) {
    f16_split_pair<2048>(v.x, v.y, s, ph.x, pl.x);
    f16_split_pair<2048>(v.z, v.w, s, ph.y, pl.y);
}
// The same split with fewer live registers (three v_fma_mix* forms per element, 8.3 issue units): lvt_gemm_kernel sits at
// 230-250 registers with its two accumulator sets, and the 5.2-unit sequence above made its m-contiguous loaders spill
// (conv weight gradient by implicit GEMM: 58 -> 147 us).  LEAN = 1 selects it.
__device__ __forceinline__ unsigned f16_pair_mix(float a, float b, float s) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %3, %2, 0" : "=&v"(r) : "v"(a), "v"(s), "v"(b));
    return r;
}
__device__ __forceinline__ void split2_lean(const float4 v, const float s, uint2 &ph, uint2 &pl) {
    ph.x = f16_pair_mix(v.x, v.y, s); ph.y = f16_pair_mix(v.z, v.w, s);
    pl.x = f16_pair_mix(f16_mix_lo(ph.x, -1.f, v.x * s), f16_mix_hi(ph.x, -1.f, v.y * s), 2048.f);
    pl.y = f16_pair_mix(f16_mix_lo(ph.y, -1.f, v.z * s), f16_mix_hi(ph.y, -1.f, v.w * s), 2048.f);
}
// power-of-two scale of an operand from its max |a| (a device scalar; any upper bound works, a loose one costs range):
// returns s = 2^e with max * s in [2^14, 2^15) and adds -e to `unscale` (the exponent that undoes it on the result).
__device__ __forceinline__ float lvt_f16_scale(const float *amax, int &unscale, const float *amax2 = nullptr) {
    if (!amax) return 1.f;
    unsigned bits = __float_as_uint(*amax);
    if (amax2) bits = max(bits, __float_as_uint(*amax2));                            // non-negative floats order like their bits
    const int eb = (int)((bits >> 23) & 0xffu);                                      // biased exponent (255: inf / nan propagate)
    int se = 268 - eb;                                                               // 127 + 14 - (eb - 127)
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    unscale -= se - 127;
    return __uint_as_float((unsigned)se << 23);
}
template <int ROWS, int LEAN = 0> __device__ __forceinline__ void store_split2_k(unsigned short *lds, int row, int k4, const float4 v, float s) {
    uint2 ph, pl;
    if (LEAN) split2_lean(v, s, ph, pl);
    else split2(v, s, ph, pl);
    unsigned short *d = lds + hrow<ROWS>(row) + k4;
    *reinterpret_cast<uint2 *>(d) = ph;
    *reinterpret_cast<uint2 *>(d + HPlane<ROWS>::SIZE) = pl;
}
// (timing experiments LVT_WX_ARAW / LVT_WX_BRAW: the operand arrives as ready fp16 planes -- same bytes, no arithmetic)
template <int ROWS> __device__ __forceinline__ void store_raw2_k(unsigned short *lds, int row, int k4, const float4 v) {
    unsigned short *d = lds + hrow<ROWS>(row) + k4;
    *reinterpret_cast<uint2 *>(d) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
    *reinterpret_cast<uint2 *>(d + HPlane<ROWS>::SIZE) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
}
template <int ROWS, int LEAN = 0> __device__ __forceinline__ void store_split2_m(unsigned short *lds, int row4, int k, const float4 v, float s) {
    uint2 p[2];
    if (LEAN) split2_lean(v, s, p[0], p[1]);
    else split2(v, s, p[0], p[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        unsigned short *d = lds + q * HPlane<ROWS>::SIZE + k;
        d[hrow<ROWS>(row4)] = (unsigned short)(p[q].x & 0xffffu); d[hrow<ROWS>(row4 + 1)] = (unsigned short)(p[q].x >> 16);
        d[hrow<ROWS>(row4 + 2)] = (unsigned short)(p[q].y & 0xffffu); d[hrow<ROWS>(row4 + 3)] = (unsigned short)(p[q].y >> 16);
    }
}
template <int ROWS, int LEAN = 0> __device__ __forceinline__ void store_split2_block(unsigned short *lds, int row4, int k4, const float4 *v, float s) {
    store_split2_k<ROWS, LEAN>(lds, row4 + 0, k4, make_float4(v[0].x, v[1].x, v[2].x, v[3].x), s);
    store_split2_k<ROWS, LEAN>(lds, row4 + 1, k4, make_float4(v[0].y, v[1].y, v[2].y, v[3].y), s);
    store_split2_k<ROWS, LEAN>(lds, row4 + 2, k4, make_float4(v[0].z, v[1].z, v[2].z, v[3].z), s);
    store_split2_k<ROWS, LEAN>(lds, row4 + 3, k4, make_float4(v[0].w, v[1].w, v[2].w, v[3].w), s);
}

// Mixed-radix digits of a running GEMM-k index, k = ((t*nH + h)*nW + w)*nC + c.  Every loader decodes its
// k position ONCE (seek, with integer divisions) and then steps it tile by tile with compares only: the main
// loop is issue-bound (measured 10.6 VALU instructions per MFMA before this), and the three signed divisions
// per fetch were a fifth of that.
struct Cursor4 {
    int c, w, h, t;
    __device__ __forceinline__ void seek(int k, int nC, int nW, int nH) {
        int q = k / nC; c = k - q * nC;
        w = q % nW; q /= nW;
        h = q % nH; t = q / nH;
    }
    __device__ __forceinline__ void advance(int step, int nC, int nW, int nH) {
        c += step;
        while (c >= nC) {
            c -= nC;
            if (++w == nW) { w = 0; if (++h == nH) { h = 0; ++t; } }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// A-side loaders.  K-contiguous modes fetch float4 along k and scatter 4 scalars into the k-major
// LDS tile; M-contiguous modes fetch float4 along m and store it as one 16-byte LDS write.
// ------------------------------------------------------------------------------------------------
template <int MODE, int BM, int MATH> struct ALoader;

// ---- k-contiguous family: rows r0 + 32*i (i < BM/32), k quad kq = tid & 7 ------------------------
#define QPR (BK / 4)                 // float4 quads per tile row
#define RPP (NTHREADS / QPR)         // tile rows covered per pass
#define KPAD (BK == 32 ? 1 : 2)      // row pad that keeps the transposing stores conflict-free
template <int MODE, int BM> struct AKLoaderBase {
    static constexpr int ITERS = BM / RPP;
    static constexpr int LD = BM + KPAD;
    float4 v[ITERS];
    int r0, kq;
    __device__ __forceinline__ void store(float *lds) const {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int row = r0 + RPP * i;
            lds[(kq * 4 + 0) * LD + row] = v[i].x;
            lds[(kq * 4 + 1) * LD + row] = v[i].y;
            lds[(kq * 4 + 2) * LD + row] = v[i].z;
            lds[(kq * 4 + 3) * LD + row] = v[i].w;
        }
    }
    __device__ __forceinline__ void store_split(unsigned short *lds) const {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split_k<BM>(lds, r0 + RPP * i, kq * 4, v[i]);
    }
    __device__ __forceinline__ void store_split2(unsigned short *lds, float s) const {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split2_k<BM>(lds, r0 + RPP * i, kq * 4, v[i], s);
    }
};

template <int BM, int MATH> struct ALoader<A_KPLAIN, BM, MATH> : AKLoaderBase<A_KPLAIN, BM> {
    using Base = AKLoaderBase<A_KPLAIN, BM>;
    const float *rowp[Base::ITERS];
    bool rowok[Base::ITERS];
    int kb, kcur, kin; long long skb, kbase;
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *A, int) {
        this->r0 = tid / QPR; this->kq = tid % QPR;
        kb = p.a_kb; skb = p.a_skb;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const int m = m0 + this->r0 + RPP * i;
            rowok[i] = m < p.M;
            rowp[i] = A + (long long)m * p.lda;
        }
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kq * 4;
        const int blk = kcur / kb;
        kin = kcur - blk * kb; kbase = (long long)blk * skb;
    }
    __device__ __forceinline__ void fetch(int kend) {
        const bool kok = kcur < kend;
        const long long koff = kbase + kin;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i)
            this->v[i] = (kok && rowok[i]) ? ldg4(rowp[i] + koff) : zero4();
        kcur += BK; kin += BK;
        while (kin >= kb) { kin -= kb; kbase += skb; }
    }
};

// forward convolution: row m -> (n,to,ho,wo); k -> (tap, ci)
template <int BM, int MATH> struct ALoader<A_CONV_K, BM, MATH> : AKLoaderBase<A_CONV_K, BM> {
    using Base = AKLoaderBase<A_CONV_K, BM>;
    const float *rowp[Base::ITERS];          // x at (n, ti0, hi0, wi0, 0): the tap-(0,0,0) corner of the window
    int ti0[Base::ITERS], hi0[Base::ITERS], wi0[Base::ITERS];
    int Ci, Kw, Kh, Ti, Hi, Wi, kcur;
    Cursor4 cur;                              // (ci, kw, kh, kt)
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *A, int) {
        this->r0 = tid / QPR; this->kq = tid % QPR;
        const lvt_conv_geom &g = p.g;
        Ci = g.Ci; Kw = g.Kw; Kh = g.Kh; Ti = g.Ti; Hi = g.Hi; Wi = g.Wi;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            int m = m0 + this->r0 + RPP * i;
            if (m < p.M) {
                const int wo = m % g.Wo; m /= g.Wo;
                const int ho = m % g.Ho; m /= g.Ho;
                const int to = m % g.To; const int n = m / g.To;
                ti0[i] = to * g.st - g.pt; hi0[i] = ho * g.sh - g.ph; wi0[i] = wo * g.sw - g.pw;
                rowp[i] = A + ((((long long)n * Ti + ti0[i]) * Hi + hi0[i]) * Wi + wi0[i]) * Ci;
            } else {
                ti0[i] = -(1 << 28); hi0[i] = 0; wi0[i] = 0; rowp[i] = A;
            }
        }
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kq * 4;
        cur.seek(kcur, Ci, Kw, Kh);
    }
    __device__ __forceinline__ void fetch(int kend) {
        const bool kok = kcur < kend;
        const int tapoff = ((cur.t * Hi + cur.h) * Wi + cur.w) * Ci + cur.c;     // same for every row of the lane
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const bool ok = kok && (unsigned)(ti0[i] + cur.t) < (unsigned)Ti && (unsigned)(hi0[i] + cur.h) < (unsigned)Hi &&
                            (unsigned)(wi0[i] + cur.w) < (unsigned)Wi;
            this->v[i] = ok ? ldg4(rowp[i] + tapoff) : zero4();
        }
        kcur += BK;
        cur.advance(BK, Ci, Kw, Kh);
    }
};

// backward-data / ConvTranspose forward.  Rows are dx positions of ONE stride phase (blockIdx.z):
// i = s*q + r.  k -> (phase tap j, co); gathered dy position o = q + c - j per dimension.
template <int BM, int MATH> struct ALoader<A_CONVT_K, BM, MATH> : AKLoaderBase<A_CONVT_K, BM> {
    using Base = AKLoaderBase<A_CONVT_K, BM>;
    const float *rowp[Base::ITERS];          // dy at (n, ot0, oh0, ow0, 0): the phase-tap-(0,0,0) position
    int ot0[Base::ITERS], oh0[Base::ITERS], ow0[Base::ITERS];
    int Co, jH, jW, To, Ho, Wo, kcur;
    Cursor4 cur;                              // (co, jw, jh, jt)
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *A, int cls) {
        this->r0 = tid / QPR; this->kq = tid % QPR;
        const lvt_conv_geom &g = p.g;
        Co = g.Co; jH = p.jH; jW = p.jW; To = g.To; Ho = g.Ho; Wo = g.Wo;
        const int fw = cls % g.sw, fh = (cls / g.sw) % g.sh, ft = cls / (g.sw * g.sh);
        // r = (phi - p) mod s ; c = (r + p - phi) / s
        const int rt = ((ft - g.pt) % g.st + g.st) % g.st, ct = (rt + g.pt - ft) / g.st;
        const int rh = ((fh - g.ph) % g.sh + g.sh) % g.sh, ch = (rh + g.ph - fh) / g.sh;
        const int rw = ((fw - g.pw) % g.sw + g.sw) % g.sw, cw = (rw + g.pw - fw) / g.sw;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            int m = m0 + this->r0 + RPP * i;
            if (m < p.M) {
                const int qw = m % p.Wq; m /= p.Wq;
                const int qh = m % p.Hq; m /= p.Hq;
                const int qt = m % p.Tq; const int n = m / p.Tq;
                ot0[i] = qt + ct; oh0[i] = qh + ch; ow0[i] = qw + cw;
                rowp[i] = A + ((((long long)n * To + ot0[i]) * Ho + oh0[i]) * Wo + ow0[i]) * Co;
            } else {
                ot0[i] = -(1 << 28); oh0[i] = 0; ow0[i] = 0; rowp[i] = A;
            }
        }
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kq * 4;
        cur.seek(kcur, Co, jW, jH);
    }
    __device__ __forceinline__ void fetch(int kend) {
        const bool kok = kcur < kend;
        const int tapoff = cur.c - ((cur.t * Ho + cur.h) * Wo + cur.w) * Co;      // gathered position o = o0 - j
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const bool ok = kok && (unsigned)(ot0[i] - cur.t) < (unsigned)To && (unsigned)(oh0[i] - cur.h) < (unsigned)Ho &&
                            (unsigned)(ow0[i] - cur.w) < (unsigned)Wo;
            this->v[i] = ok ? ldg4(rowp[i] + tapoff) : zero4();
        }
        kcur += BK;
        cur.advance(BK, Co, jW, jH);
    }
};

// ---- m-contiguous family: k rows kk0 + KPP*i, m quad mq -----------------------------------------
template <int BM, int MATH> struct AMLoaderBase {
    static constexpr int UPK = BM / 4;              // float4 units per k row
    static constexpr int KPP = NTHREADS / UPK;      // k rows per pass
    static constexpr int ITERS = BK / KPP;
    static constexpr int LD = BM + 4;
    // k row of fetch i: strided over the passes for the fp32 tile, consecutive per lane for the bf16 planes
    static constexpr int KMUL = MATH ? ITERS : 1, KSTEP = MATH ? 1 : KPP;
    float4 v[ITERS];
    int kk0, mq;
    __device__ __forceinline__ void store(float *lds) const {
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
            *reinterpret_cast<float4 *>(&lds[(kk0 + KPP * i) * LD + mq * 4]) = v[i];
    }
    __device__ __forceinline__ void store_split(unsigned short *lds) const {
        if (ITERS == 4) { store_split_block<BM>(lds, mq * 4, kk0 * 4, v); return; }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split_m<BM>(lds, mq * 4, kk0 * KMUL + KSTEP * i, v[i]);
    }
    __device__ __forceinline__ void store_split2(unsigned short *lds, float s) const {
        if (ITERS == 4) { store_split2_block<BM, 1>(lds, mq * 4, kk0 * 4, v, s); return; }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split2_m<BM, 1>(lds, mq * 4, kk0 * KMUL + KSTEP * i, v[i], s);
    }
};

template <int BM, int MATH> struct ALoader<A_MPLAIN, BM, MATH> : AMLoaderBase<BM, MATH> {
    using Base = AMLoaderBase<BM, MATH>;
    const float *kp; bool mok; long long lda; int kcur;
    bool sum_on; float4 colacc;      // weight gradients: running sums over k of the fetched A tiles (bias gradient)
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *A, int) {
        this->kk0 = tid / Base::UPK; this->mq = tid % Base::UPK;
        const int m = m0 + this->mq * 4;
        mok = m < p.M; kp = A + m; lda = p.lda;
        sum_on = false; colacc = zero4();
    }
    // sums of everything this workgroup fetched, per m: lanes with the same m quad are added in kk0 order
    __device__ __forceinline__ void write_colsum(float *scratch, float *dst, int m0, int M, int tid) const {
        *reinterpret_cast<float4 *>(&scratch[(this->kk0 * Base::UPK + this->mq) * 4]) = colacc;
        __syncthreads();
        if (tid < BM) {
            float s_ = 0.f;
            for (int r = 0; r < Base::KPP; ++r) s_ += scratch[(r * Base::UPK + tid / 4) * 4 + (tid & 3)];
            if (m0 + tid < M) dst[m0 + tid] = s_;
        }
        __syncthreads();             // the epilogue reuses this memory
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kk0 * Base::KMUL;
        kp += (long long)kcur * lda;
    }
    __device__ __forceinline__ void fetch(int kend) {
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i)
            this->v[i] = (mok && kcur + Base::KSTEP * i < kend) ? ldg4(kp + (long long)(Base::KSTEP * i) * lda) : zero4();
        kcur += BK; kp += (long long)BK * lda;
        if (sum_on) {
#pragma unroll
            for (int i = 0; i < Base::ITERS; ++i) {
                colacc.x += this->v[i].x; colacc.y += this->v[i].y; colacc.z += this->v[i].z; colacc.w += this->v[i].w;
            }
        }
    }
};

// backward-weight: GEMM row = (tap, ci), GEMM k = output pixel (n,to,ho,wo)
template <int BM, int MATH> struct ALoader<A_CONV_M, BM, MATH> : AMLoaderBase<BM, MATH> {
    using Base = AMLoaderBase<BM, MATH>;
    const float *x; lvt_conv_geom g; bool mok; int kt, kh, kw, ci, kcur; long long sample;
    Cursor4 cur;                              // output pixel of the lane's first k row: (wo, ho, to, n)
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *A, int) {
        this->kk0 = tid / Base::UPK; this->mq = tid % Base::UPK;
        x = A; g = p.g;
        sample = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
        const int m = m0 + this->mq * 4;
        mok = m < p.M;
        int tap = m / g.Ci; ci = m - tap * g.Ci;
        kw = tap % g.Kw; tap /= g.Kw;
        kh = tap % g.Kh; kt = tap / g.Kh;
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kk0 * Base::KMUL;
        cur.seek(kcur, g.Wo, g.Ho, g.To);
    }
    __device__ __forceinline__ void fetch(int kend) {
        Cursor4 c = cur;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const int ti = c.h * g.st - g.pt + kt, hi = c.w * g.sh - g.ph + kh, wi = c.c * g.sw - g.pw + kw;
            const bool ok = mok && kcur + Base::KSTEP * i < kend && (unsigned)ti < (unsigned)g.Ti &&
                            (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            const int inner = ((ti * g.Hi + hi) * g.Wi + wi) * g.Ci + ci;            // within one sample: < 2^31
            this->v[i] = ok ? ldg4(x + c.t * sample + inner) : zero4();
            if (i + 1 < Base::ITERS) c.advance(Base::KSTEP, g.Wo, g.Ho, g.To);
        }
        kcur += BK;
        cur.advance(BK, g.Wo, g.Ho, g.To);
    }
};

// scatter-add as a GEMM: dTable[(slot, code)][:] = sum_rows onehot(row, slot)[code] * dOut[row][:]
template <int BM, int MATH> struct ALoader<A_ONEHOT_M, BM, MATH> : AMLoaderBase<BM, MATH> {
    using Base = AMLoaderBase<BM, MATH>;
    const long long *ip; long long bstride, pstride; int P, code0; bool mok;
    __device__ __forceinline__ void init(const KParams &p, int tid, int m0, const float *, int) {
        this->kk0 = tid / Base::UPK; this->mq = tid % Base::UPK;
        const int m = m0 + this->mq * 4;
        mok = m < p.M;
        const int slot = mok ? m / p.oh_V : 0;
        code0 = m - slot * p.oh_V;
        ip = p.oh_idx + p.oh_off[slot];
        bstride = p.oh_bstride; pstride = p.oh_pstride; P = p.oh_P;
    }
    int kcur;
    // f16x2 planes of a 0 / 1 operand need no arithmetic: hi = 1.0h (0x3C00) or 0, lo = 0 (the generic split spilled registers
    // here: 126 -> 228 us on the embedding weight gradients)
    __device__ __forceinline__ void store_split2(unsigned short *lds, float) const {
        static_assert(Base::ITERS == 4, "one-hot loader: 4 k rows per lane");
        const float c[4][4] = {{this->v[0].x, this->v[1].x, this->v[2].x, this->v[3].x}, {this->v[0].y, this->v[1].y, this->v[2].y, this->v[3].y},
                               {this->v[0].z, this->v[1].z, this->v[2].z, this->v[3].z}, {this->v[0].w, this->v[1].w, this->v[2].w, this->v[3].w}};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint2 h;
            h.x = (c[r][0] != 0.f ? 0x3C00u : 0u) | (c[r][1] != 0.f ? 0x3C000000u : 0u);
            h.y = (c[r][2] != 0.f ? 0x3C00u : 0u) | (c[r][3] != 0.f ? 0x3C000000u : 0u);
            unsigned short *d = lds + hrow<BM>(this->mq * 4 + r) + this->kk0 * 4;
            *reinterpret_cast<uint2 *>(d) = h;
            *reinterpret_cast<uint2 *>(d + HPlane<BM>::SIZE) = make_uint2(0u, 0u);
        }
    }
    __device__ __forceinline__ void seek(int k0) { kcur = k0 + this->kk0 * Base::KMUL; }
    __device__ __forceinline__ void fetch(int kend) {
        const int kbase = kcur;
        kcur += BK;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const int row = kbase + Base::KSTEP * i;
            float4 v = zero4();
            if (mok && row < kend) {
                const int b = row / P, pos = row - b * P;
                const int d = (int)(ip[b * bstride + pos * pstride]) - code0;
                v.x = d == 0 ? 1.f : 0.f; v.y = d == 1 ? 1.f : 0.f;
                v.z = d == 2 ? 1.f : 0.f; v.w = d == 3 ? 1.f : 0.f;
            }
            this->v[i] = v;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// B-side loaders
// ------------------------------------------------------------------------------------------------
template <int MODE, int BN, int MATH> struct BLoader;

template <int BN> struct BKLoaderBase {
    static constexpr int ITERS = (BN >= RPP) ? BN / RPP : 1;
    static constexpr int LD = BN + KPAD;
    float4 v[ITERS];
    int r0, kq;
    __device__ __forceinline__ void store(float *lds) const {
        if (r0 >= BN) return;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int row = r0 + RPP * i;
            lds[(kq * 4 + 0) * LD + row] = v[i].x;
            lds[(kq * 4 + 1) * LD + row] = v[i].y;
            lds[(kq * 4 + 2) * LD + row] = v[i].z;
            lds[(kq * 4 + 3) * LD + row] = v[i].w;
        }
    }
    __device__ __forceinline__ void store_split(unsigned short *lds) const {
        if (r0 >= BN) return;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split_k<BN>(lds, r0 + RPP * i, kq * 4, v[i]);
    }
    __device__ __forceinline__ void store_split2(unsigned short *lds, float s) const {
        if (r0 >= BN) return;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split2_k<BN>(lds, r0 + RPP * i, kq * 4, v[i], s);
    }
};

template <int BN, int MATH> struct BLoader<B_KPLAIN, BN, MATH> : BKLoaderBase<BN> {
    using Base = BKLoaderBase<BN>;
    const float *rowp[Base::ITERS]; bool rowok[Base::ITERS]; int kb, kcur, kin; long long skb, kbase;
    __device__ __forceinline__ void init(const KParams &p, int tid, int n0, const float *B, int) {
        this->r0 = tid / QPR; this->kq = tid % QPR;
        kb = p.b_kb; skb = p.b_skb;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const int n = n0 + this->r0 + RPP * i;
            rowok[i] = n < p.N && this->r0 < BN;
            rowp[i] = B + (long long)n * p.ldb;
        }
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kq * 4;
        const int blk = kcur / kb;
        kin = kcur - blk * kb; kbase = (long long)blk * skb;
    }
    __device__ __forceinline__ void fetch(int kend) {
        const bool kok = kcur < kend;
        const long long koff = kbase + kin;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i)
            this->v[i] = (kok && rowok[i]) ? ldg4(rowp[i] + koff) : zero4();
        kcur += BK; kin += BK;
        while (kin >= kb) { kin -= kb; kbase += skb; }
    }
};

// packed conv weights wp[tap][ci][co] read as B(k=(phase tap j, co), n=ci) for one stride phase
template <int BN, int MATH> struct BLoader<B_CONVT_W, BN, MATH> : BKLoaderBase<BN> {
    using Base = BKLoaderBase<BN>;
    const float *wp; lvt_conv_geom g; int jH, jW, ft, fh, fw, kcur;
    int rowoff[Base::ITERS]; bool rowok[Base::ITERS];
    Cursor4 cur;                              // (co, jw, jh, jt)
    __device__ __forceinline__ void init(const KParams &p, int tid, int n0, const float *B, int cls) {
        this->r0 = tid / QPR; this->kq = tid % QPR;
        wp = B; g = p.g; jH = p.jH; jW = p.jW;
        fw = cls % g.sw; fh = (cls / g.sw) % g.sh; ft = cls / (g.sw * g.sh);
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i) {
            const int n = n0 + this->r0 + RPP * i;
            rowok[i] = n < p.N && this->r0 < BN;
            rowoff[i] = n * g.Co;                    // packed weights hold taps*Ci*Co < 2^31 elements
        }
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + this->kq * 4;
        cur.seek(kcur, g.Co, jW, jH);
    }
    __device__ __forceinline__ void fetch(int kend) {
        const bool kok = kcur < kend;
        const int kt = ft + g.st * cur.t, kh = fh + g.sh * cur.h, kw = fw + g.sw * cur.w;
        const int base = ((kt * g.Kh + kh) * g.Kw + kw) * g.Ci * g.Co + cur.c;
#pragma unroll
        for (int i = 0; i < Base::ITERS; ++i)
            this->v[i] = (kok && rowok[i]) ? ldg4(wp + base + rowoff[i]) : zero4();
        kcur += BK;
        cur.advance(BK, g.Co, jW, jH);
    }
};

template <int BN, int MATH> struct BLoader<B_NPLAIN, BN, MATH> {
    static constexpr int UPK = BN / 4;
    static constexpr int KPP = (NTHREADS / UPK) > BK ? BK : (NTHREADS / UPK);
    static constexpr int ITERS = BK / KPP;
    static constexpr int LD = BN + 4;
    static constexpr int KMUL = MATH ? ITERS : 1, KSTEP = MATH ? 1 : KPP;
    float4 v[ITERS];
    int kk0, nq, kcur; bool active;
    const float *kp; bool nok; long long ldb;
    bool sum_on; float4 colacc;      // bwd-weight: running column sums of the fetched B tiles (bias gradient)
    __device__ __forceinline__ void init(const KParams &p, int tid, int n0, const float *B, int) {
        kk0 = tid / UPK; nq = tid % UPK;
        active = kk0 < BK;
        const int n = n0 + nq * 4;
        nok = active && n < p.N; kp = B + n; ldb = p.ldb;
        sum_on = false; colacc = zero4();
    }
    __device__ __forceinline__ void seek(int k0) {
        kcur = k0 + kk0 * KMUL;
        kp += (long long)kcur * ldb;
    }
    __device__ __forceinline__ void fetch(int kend) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
            v[i] = (nok && kcur + KSTEP * i < kend) ? ldg4(kp + (long long)(KSTEP * i) * ldb) : zero4();
        kcur += BK; kp += (long long)BK * ldb;
        if (sum_on) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) { colacc.x += v[i].x; colacc.y += v[i].y; colacc.z += v[i].z; colacc.w += v[i].w; }
        }
    }
    // column sums of everything this workgroup fetched: lanes with the same column quad (nq) are added in kk0 order
    __device__ __forceinline__ void write_colsum(float *scratch, float *dst, int n0, int N, int tid) const {
        constexpr int KROWS = (NTHREADS / UPK) > BK ? BK : (NTHREADS / UPK);
        if (active) *reinterpret_cast<float4 *>(&scratch[(kk0 * UPK + nq) * 4]) = colacc;
        __syncthreads();
        if (tid < BN) {
            float s_ = 0.f;
            for (int r = 0; r < KROWS; ++r) s_ += scratch[(r * UPK + tid / 4) * 4 + (tid & 3)];
            if (n0 + tid < N) dst[n0 + tid] = s_;
        }
        __syncthreads();             // the epilogue reuses this memory
    }
    __device__ __forceinline__ void store(float *lds) const {
        if (!active) return;
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
            *reinterpret_cast<float4 *>(&lds[(kk0 + KPP * i) * LD + nq * 4]) = v[i];
    }
    __device__ __forceinline__ void store_split(unsigned short *lds) const {
        if (!active) return;
        if (ITERS == 4) { store_split_block<BN>(lds, nq * 4, kk0 * 4, v); return; }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split_m<BN>(lds, nq * 4, kk0 * KMUL + KSTEP * i, v[i]);
    }
    __device__ __forceinline__ void store_split2(unsigned short *lds, float s) const {
        if (!active) return;
        if (ITERS == 4) { store_split2_block<BN, 1>(lds, nq * 4, kk0 * 4, v, s); return; }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) store_split2_m<BN, 1>(lds, nq * 4, kk0 * KMUL + KSTEP * i, v[i], s);
    }
};

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A_PATCH (frame-resident 3x3 convolution, below): GEMM row r of a 256-row frame tile is NOT pixel r.  The 32 rows of an
// MFMA tile t32 are the pixels of image rows 2*t32 and 2*t32+1, lane quad q (4 consecutive lanes) holding 4 consecutive
// x of image row 2*t32 + (popcount(q) & 1) starting at x = 4*(q >> 1): the lane groups a ds_read_b128 is serviced in
// ({0-3,12-15,20-27} / {4-11,16-19,28-31}) then each cover 16 consecutive pixels of ONE image row, which is what makes the
// shifted operand reads from the pixel-major patch bank-conflict free for every tap.
__device__ __forceinline__ long long patch_orow(int row) {
    const int f = row & 255, t32 = f >> 5, q = (f >> 2) & 7, e = f & 3;
    const int y = 2 * t32 + (__popc(q) & 1), x = 4 * (q >> 1) + e;
    return (long long)(row & ~255) + y * 16 + x;
}

// A_PATCHT: the same row order on the half-resolution grid of ONE output phase (py, px) of a stride-2 transposed
// convolution; the output pixel is (2y + py, 2x + px) of a 32x32 frame.
__device__ __forceinline__ long long patcht_orow(int row, int cls) {
    const int f = row & 255, t32 = f >> 5, q = (f >> 2) & 7, e = f & 3;
    const int y = 2 * t32 + (__popc(q) & 1), x = 4 * (q >> 1) + e;
    return (long long)(row >> 8) * 1024 + (2 * y + (cls >> 1)) * 32 + 2 * x + (cls & 1);
}

struct PatchRows { __device__ __forceinline__ long long operator()(int row) const { return patch_orow(row); } };
struct PatchTRows { int cls; __device__ __forceinline__ long long operator()(int row) const { return patcht_orow(row, cls); } };

// ------------------------------------------------------------------------------------------------
// epilogue shared by the kernels: alpha / bias / residual / ReLU / tanh / mask / accumulate, or split-K partials
// ------------------------------------------------------------------------------------------------
template <int AMODE, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void lvt_epilogue(const KParams &p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int m0, int n0,
                                             int wm, int wn, int l31, int half, int cls, long long coff, int z, int split) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // ConvTranspose phase: decode the phase-ordered row into the channels-last dx row.
    int fw = 0, fh = 0, ft = 0, rt = 0, rh = 0, rw = 0;
    if (AMODE == A_CONVT_K) {
        const lvt_conv_geom &g = p.g;
        fw = cls % g.sw; fh = (cls / g.sw) % g.sh; ft = cls / (g.sw * g.sh);
        rt = ((ft - g.pt) % g.st + g.st) % g.st;
        rh = ((fh - g.ph) % g.sh + g.sh) % g.sh;
        rw = ((fw - g.pw) % g.sw + g.sw) % g.sw;
    }
    const int flags = p.flags;
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.M) continue;
            long long orow = row;
            if (AMODE == A_PATCH) orow = patch_orow(row);
            if (AMODE == A_PATCHT) orow = patcht_orow(row, cls);
            if (AMODE == A_CONVT_K) {
                const lvt_conv_geom &g = p.g;
                int m = row;
                const int qw = m % p.Wq; m /= p.Wq;
                const int qh = m % p.Hq; m /= p.Hq;
                const int qt = m % p.Tq; const int n = m / p.Tq;
                orow = (((long long)n * g.Ti + (g.st * qt + rt)) * g.Hi + (g.sh * qh + rh)) * g.Wi +
                       (g.sw * qw + rw);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (TN * 32) + j * 32 + l31;
                if (col >= p.N) continue;
                float v = acc[i][j][r];
                if (p.splits > 1) {
                    p.partial[split * p.partial_stride + (long long)z * p.M * p.N + orow * p.N + col] = v;
                    continue;
                }
                v *= p.alpha;
                if (flags & LVT_EPI_BIAS) v += p.bias[col];
                if (flags & LVT_EPI_RESIDUAL) v += p.res[coff + orow * p.ldr + col];
                if (flags & LVT_EPI_RELU) v = fmaxf(v, 0.f);
                if (flags & LVT_EPI_TANH) v = tanhf(v);
                if (flags & LVT_EPI_MASK) v = (p.mask[coff + orow * p.ldm + col] > 0.f) ? v : 0.f;
                float *cp = p.C + coff + orow * p.ldc + col;
                if (flags & LVT_EPI_ACCUM) v += *cp;
                *cp = v;
                am = fmaxf(am, lvt_absf(v));
            }
        }
    }
    if (p.c_amax && p.splits <= 1) {
        __shared__ float amax_scratch[8];
        lvt_block_amax_commit(am, p.c_amax, amax_scratch);
    }
}

// blockIdx -> tile / batch / split coordinates (shared by the kernels)
struct TileCtx { int m0, n0, z, split, cls, kbeg, kend; const float *A, *B; long long coff; };
template <int AMODE, int BM, int BN>
__device__ __forceinline__ TileCtx lvt_tile_ctx(const KParams &p) {
    TileCtx t;
    // Workgroup b is dispatched to XCD b % 8 and every XCD has a private L2, so the linear id is first remapped
    // to give each XCD one CONTIGUOUS run of tiles (bijective for any grid size): the n tiles that share an A
    // panel, and the neighbouring m tiles that share im2col halo rows / the same image, then hit the same L2.
    const int ntn = (p.N + BN - 1) / BN;
    int wg = blockIdx.x;
    t.split = blockIdx.z;
    t.z = blockIdx.y;                   // batch (or conv phase class)
    if (p.splits > 1 && ((gridDim.y * gridDim.z) & 7) == 0) {
        // split-K of a few tiles (the weight gradients dW = dY^T X: 16 tiles x 32 k ranges): ALL tiles of one (batch, k range)
        // read the same rows of both operands, so such a group is given to ONE XCD (the dispatcher walks x, then y, then z:
        // linear id L -> XCD L % 8) and its tiles share the panels through that L2.  With the per-dimension remap below every
        // XCD held two tiles of every k range and fetched 4-5x the algorithmic bytes (profiles/r02_dsfvt_pmc_hbm_traffic.txt).
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = lin & 7, slot = lin >> 3;
        const unsigned grp = xcd + 8 * (slot / gridDim.x);          // (batch, k range), batch fastest
        t.z = (int)(grp % gridDim.y);
        t.split = (int)(grp / gridDim.y);
        wg = (int)(slot % gridDim.x);
    } else {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = wg & 7, slot = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    t.n0 = (wg % ntn) * BN; t.m0 = (wg / ntn) * BM;
    t.A = p.A; t.B = p.B; t.coff = 0; t.cls = 0;
    if (AMODE == A_CONVT_K) {
        t.cls = t.z;
    } else {
        const int zo = t.z / p.batch_inner, zi = t.z % p.batch_inner;
        t.A += zo * p.sA_o + zi * p.sA_i;
        t.B += zo * p.sB_o + zi * p.sB_i;
        t.coff = zo * p.sC_o + zi * p.sC_i;
    }
    t.kbeg = 0; t.kend = p.K;
    if (p.splits > 1) {
        t.kbeg = t.split * p.k_per_split;
        t.kend = min(p.K, t.kbeg + p.k_per_split);
    }
    // Causal attention products (square per-batch problems whose index k, m or n is a token position and whose operand is
    // known to vanish above / below the diagonal): the structurally-zero part of the reduction is not walked.
    if (p.flags & LVT_CAUSAL_KMAX) t.kend = min(t.kend, t.m0 + BM);             // A(m, k) == 0 for k > m   (dQ = dS K)
    if (p.flags & LVT_CAUSAL_KMIN) t.kbeg = max(t.kbeg, (t.m0 / BK) * BK);      // A(m, k) == 0 for k < m   (dV = P^T dO, dK = dS^T Q)
    if ((p.flags & LVT_CAUSAL_TILE) && t.n0 > t.m0 + BM - 1) t.kend = t.kbeg;   // C(m, n) is not needed for n > m: written as 0
    return t;
}

// The accumulator layout gives a lane ONE column and 16 rows per 32x32 tile, i.e. 4-byte stores and 4-byte residual /
// mask loads, 64 of them per thread, each with its own address arithmetic -- for the K = 512 GEMMs of the transformer
// this epilogue cost about as much as a third of the main loop.  Here every wave turns its sub-tile through LDS
// (32 rows at a time, wave-private region of the staging memory, which is free after the main loop): a lane then owns
// 4 consecutive columns of a row, the residual / mask / accumulate reads and the store are 16-byte accesses, a wave
// instruction covers whole 256-byte row segments, and the row decode (ConvTranspose phases) runs once per float4.
template <int AMODE, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void lvt_epilogue_vec(const KParams &p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float *lds,
                                                 int m0, int n0, int wm, int wn, int lane, int cls, long long coff, int z,
                                                 int split) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int SW = TN * 32;                    // sub-tile width (floats), row stride of the LDS turn-table
    constexpr int C4 = SW / 4;                     // float4 per row
    const int l31 = lane & 31, half = lane >> 5;
    float *tile = lds + (wm * WN + wn) * (32 * SW);
    int rt = 0, rh = 0, rw = 0;
    if (AMODE == A_CONVT_K) {
        const lvt_conv_geom &g = p.g;
        const int fw = cls % g.sw, fh = (cls / g.sw) % g.sh, ft = cls / (g.sw * g.sh);
        rt = ((ft - g.pt) % g.st + g.st) % g.st;
        rh = ((fh - g.ph) % g.sh + g.sh) % g.sh;
        rw = ((fw - g.pw) % g.sw + g.sw) % g.sw;
    }
    const int flags = p.flags;
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * half) * SW + 32 * j + l31] = acc[i][j][r];
        __syncthreads();
        if (AMODE != A_KPLAIN && AMODE != A_MPLAIN && p.splits <= 1 && (flags & LVT_EPI_MASK) && !(flags & (LVT_EPI_PLANES | LVT_EPI_ACCUM))) {
            constexpr int NU = 32 * C4 / 64;
            // MASK forms of the convolution kernels (ReLU-backward: the mask is an activation of the forward pass, cold in the
            // caches): every mask / residual value of the sub-tile is requested before the first store -- a load may not pass an
            // earlier store to memory the compiler cannot tell apart, and behind the per-iteration flag test each one sat in its own
            // block, one round trip per 256-byte row segment.  -0.15 ms per VQ-VAE step.  The plain GEMMs and the other forms keep the
            // loop below: in the DSFVT step the two-phase form gained nothing on the masked product and cost the others registers
            // (profiles/r04_wide_gemm_loop_experiments.txt, block 11).
            long long oaddr[NU];                       // orow of the unit, -1: outside the matrix
            int ocol[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = lane + 64 * u;
                const int rowl = idx / C4, c4 = idx % C4;
                const int row = m0 + wm * (TM * 32) + i * 32 + rowl;
                const int col = n0 + wn * SW + 4 * c4;
                ocol[u] = col;
                long long orow = -1;
                if (row < p.M && col < p.N) {
                    orow = row;
                    if (AMODE == A_PATCH) orow = patch_orow(row);
                    if (AMODE == A_PATCHT) orow = patcht_orow(row, cls);
                    if (AMODE == A_CONVT_K) {
                        const lvt_conv_geom &g = p.g;
                        int m = row;
                        const int qw = m % p.Wq; m /= p.Wq;
                        const int qh = m % p.Hq; m /= p.Hq;
                        const int qt = m % p.Tq; const int n = m / p.Tq;
                        orow = (((long long)n * g.Ti + (g.st * qt + rt)) * g.Hi + (g.sh * qh + rh)) * g.Wi + (g.sw * qw + rw);
                    }
                }
                oaddr[u] = orow;
            }
            float4 rv[NU], mv[NU];
            if (flags & LVT_EPI_RESIDUAL) {
#pragma unroll
                for (int u = 0; u < NU; ++u) rv[u] = oaddr[u] >= 0 ? ldg4(p.res + coff + oaddr[u] * p.ldr + ocol[u]) : zero4();
            } else {
#pragma unroll
                for (int u = 0; u < NU; ++u) rv[u] = zero4();
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) mv[u] = oaddr[u] >= 0 ? ldg4(p.mask + coff + oaddr[u] * p.ldm + ocol[u]) : zero4();
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = lane + 64 * u;
                const int rowl = idx / C4, c4 = idx % C4;
                const int col = ocol[u];
                const long long orow = oaddr[u];
                if (orow >= 0) {
                    float4 v = *reinterpret_cast<const float4 *>(&tile[rowl * SW + 4 * c4]);
                    v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
                    if (flags & LVT_EPI_BIAS) { const float4 b = ldg4(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                    v.x += rv[u].x; v.y += rv[u].y; v.z += rv[u].z; v.w += rv[u].w;
                    if (flags & LVT_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (flags & LVT_EPI_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                    v.x = mv[u].x > 0.f ? v.x : 0.f; v.y = mv[u].y > 0.f ? v.y : 0.f; v.z = mv[u].z > 0.f ? v.z : 0.f; v.w = mv[u].w > 0.f ? v.w : 0.f;
                    am = fmaxf(am, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
                    *reinterpret_cast<float4 *>(p.C + coff + orow * p.ldc + col) = v;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 32 * C4 / 64; ++u) {
                const int idx = lane + 64 * u;
                const int rowl = idx / C4, c4 = idx % C4;
                const int row = m0 + wm * (TM * 32) + i * 32 + rowl;
                const int col = n0 + wn * SW + 4 * c4;
                if (row < p.M && col < p.N) {
                    float4 v = *reinterpret_cast<const float4 *>(&tile[rowl * SW + 4 * c4]);
                    long long orow = row;
                    if (AMODE == A_PATCH) orow = patch_orow(row);
                    if (AMODE == A_PATCHT) orow = patcht_orow(row, cls);
                    if (AMODE == A_CONVT_K) {
                        const lvt_conv_geom &g = p.g;
                        int m = row;
                        const int qw = m % p.Wq; m /= p.Wq;
                        const int qh = m % p.Hq; m /= p.Hq;
                        const int qt = m % p.Tq; const int n = m / p.Tq;
                        orow = (((long long)n * g.Ti + (g.st * qt + rt)) * g.Hi + (g.sh * qh + rh)) * g.Wi + (g.sw * qw + rw);
                    }
                    if (p.splits > 1) {
                        *reinterpret_cast<float4 *>(p.partial + split * p.partial_stride + (long long)z * p.M * p.N + orow * p.N + col) = v;
                    } else {
                        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
                        if (flags & LVT_EPI_BIAS) { const float4 b = ldg4(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                        if (flags & LVT_EPI_RESIDUAL) { const float4 b = ldg4(p.res + coff + orow * p.ldr + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                        if (flags & LVT_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (flags & LVT_EPI_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                        if (flags & LVT_EPI_MASK) {
                            const float4 mk = ldg4(p.mask + coff + orow * p.ldm + col);
                            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                        }
                        if (!(flags & LVT_EPI_ACCUM)) am = fmaxf(am, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
                        if (flags & LVT_EPI_PLANES) {
                            // C is a bf16 image: the result leaves as its exact 3-way bf16 split, one plane c_plane elements
                            // after the other (the operand format of the fused attention kernels, attention_pipe.hip)
                            uint2 p1, p2, p3;
                            split3(v, p1, p2, p3);
                            unsigned short *cp = reinterpret_cast<unsigned short *>(p.C) + coff + orow * p.ldc + col;
                            *reinterpret_cast<uint2 *>(cp) = p1;
                            *reinterpret_cast<uint2 *>(cp + p.c_plane) = p2;
                            *reinterpret_cast<uint2 *>(cp + 2 * p.c_plane) = p3;
                        } else {
                            float4 *cp = reinterpret_cast<float4 *>(p.C + coff + orow * p.ldc + col);
                            if (flags & LVT_EPI_ACCUM) {
                                const float4 c = *cp; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
                                am = fmaxf(am, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
                            }
                            *cp = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (p.c_amax && p.splits <= 1) lvt_block_amax_commit(am, p.c_amax, lds);      // (the turn-table is free again)
}

// MATH == 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
// MATH == 1: bf16x3 split -- every fp32 operand is staged as three bf16 planes and each 32x32x16 block is six
//            v_mfma_f32_32x32x16_bf16 (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2: all product terms above 2^-24 |ab|,
//            each bf16 x bf16 product exact in the fp32 accumulator).  fp32-class accuracy at ~2.7x the rate.
// MATH == 2: f16x2 split -- two fp16 planes per operand after an exact power-of-two scale from the operand's max |.|
//            (split2 above), three v_mfma_f32_32x32x16_f16 per block into two accumulators.  Half the MFMAs of MATH == 1.
template <int AMODE, int BMODE, int BM, int BN, int WM, int WN, int MATH>
__global__ __launch_bounds__(NTHREADS, (MATH == 2 ? 2 : LVT_MINWAVES)) void lvt_gemm_kernel(const KParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    using AL = ALoader<AMODE, BM, MATH>;
    using BL = BLoader<BMODE, BN, MATH>;
    constexpr int LDA = AL::LD, LDB = BL::LD;
    constexpr int PSA = HPlane<BM>::SIZE, PSB = HPlane<BN>::SIZE;                       // bf16 plane strides (MATH == 1)
    constexpr int NP = MATH == 2 ? 2 : 3;                                               // 16-bit planes per operand
    constexpr int STAGE_FLOATS = MATH == 0 ? (BK * LDA + BK * LDB + 8) : (NP * (PSA + PSB) / 2 + 8);
    constexpr int TURN_FLOATS = WM * WN * 32 * (TN * 32);             // epilogue turn-table: 32 rows of every wave's sub-tile
    constexpr int LDS_FLOATS = STAGE_FLOATS > TURN_FLOATS ? STAGE_FLOATS : TURN_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *As = lds;
    float *Bs = lds + ((BK * LDA + 3) & ~3);
    unsigned short *Ah = reinterpret_cast<unsigned short *>(lds);
    unsigned short *Bh = Ah + NP * PSA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const TileCtx tc = lvt_tile_ctx<AMODE, BM, BN>(p);
    const int m0 = tc.m0, n0 = tc.n0, z = tc.z, split = tc.split, cls = tc.cls, kbeg = tc.kbeg, kend = tc.kend;
    const float *A = tc.A, *B = tc.B;
    const long long coff = tc.coff;

    AL al; BL bl;
    al.init(p, tid, m0, A, cls);
    bl.init(p, tid, n0, B, cls);

    f32x16 acc[TM][TN];
    f32x16 acx[MATH == 2 ? TM : 1][MATH == 2 ? TN : 1];      // f16x2: the hi lo + lo hi terms (weight 2^-11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (MATH == 2) acx[i][j][r] = 0.f;
            }
    int unscale = 0;
    float sa = 1.f, sb = 1.f;
    if (MATH == 2) { sa = lvt_f16_scale(p.a_amax, unscale, p.a_amax2); sb = lvt_f16_scale(p.b_amax, unscale, p.b_amax2); }

    constexpr bool COLSUM = (AMODE == A_CONV_M && BMODE == B_NPLAIN);
    constexpr bool COLSUM_A = (AMODE == A_MPLAIN);
    if constexpr (COLSUM) bl.sum_on = p.colsum_partial != nullptr && m0 == 0;
    if constexpr (COLSUM_A) al.sum_on = p.colsum_partial != nullptr && n0 == 0;          // per batch
    if (kbeg < kend) {
        al.seek(kbeg); bl.seek(kbeg);
        al.fetch(kend); bl.fetch(kend);
        if (MATH == 0) { al.store(As); bl.store(Bs); }
        else if (MATH == 1) { al.store_split(Ah); bl.store_split(Bh); }
        else { al.store_split2(Ah, sa); bl.store_split2(Bh, sb); }
    }
    __syncthreads();

    const float *Ard = As + half * LDA + wm * (TM * 32) + l31;
    const float *Brd = Bs + half * LDB + wn * (TN * 32) + l31;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool has_next = k0 + BK < kend;
        if (has_next) { al.fetch(kend); bl.fetch(kend); }
        if (MATH == 0) {
            // operand fragments are double-buffered in registers: the ds_reads of step kk+2 are in flight
            // while the MFMAs of step kk issue, so the LDS latency is not exposed once per step
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = Ard[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Brd[j * 32];
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + 1) / 2 + (TN + 1) / 2, 0);   // step-0 operand reads
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
                if (kk + 2 < BK) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = Ard[(kk + 2) * LDA + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = Brd[(kk + 2) * LDB + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                // pin the interleave: next step's operand reads are issued BEFORE this step's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, (TM + 1) / 2 + (TN + 1) / 2, 0);   // DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                       // MFMAs
            }
        } else if (MATH == 2) {
            const unsigned short *Arh[TM], *Brh[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) Arh[i] = Ah + hrow<BM>(wm * (TM * 32) + i * 32 + l31) + 8 * half;
#pragma unroll
            for (int j = 0; j < TN; ++j) Brh[j] = Bh + hrow<BN>(wn * (TN * 32) + j * 32 + l31) + 8 * half;
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                f16x8 a[2][TM], b[2][TN];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const f16x8 *>(Arh[i] + q * PSA + ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const f16x8 *>(Brh[j] + q * PSB + ks);
                }
                // three passes over the TM x TN accumulator pairs: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acx[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], acx[i][j], 0, 0, 0);
            }
        } else {
            // lane l feeds row (l & 31) and the 8 consecutive k starting at 8 * (l >> 5) of each 16-wide k step
            const unsigned short *Arh[TM], *Brh[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) Arh[i] = Ah + hrow<BM>(wm * (TM * 32) + i * 32 + l31) + 8 * half;
#pragma unroll
            for (int j = 0; j < TN; ++j) Brh[j] = Bh + hrow<BN>(wn * (TN * 32) + j * 32 + l31) + 8 * half;
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                bf16x8 a[3][TM], b[3][TN];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[q][i] = *reinterpret_cast<const bf16x8 *>(Arh[i] + q * PSA + ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[q][j] = *reinterpret_cast<const bf16x8 *>(Brh[j] + q * PSB + ks);
                }
                // term-major order: consecutive MFMAs go to different accumulators (TM*TN apart), so no MFMA
                // waits on the result of its predecessor; small terms first (2^-16, then 2^-8, then the leading one)
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (has_next) {
            if (MATH == 0) { al.store(As); bl.store(Bs); }
            else if (MATH == 1) { al.store_split(Ah); bl.store_split(Bh); }
            else { al.store_split2(Ah, sa); bl.store_split2(Bh, sb); }
        }
        __syncthreads();
    }
    // result = (hi hi + 2^-11 (hi lo + lo hi)) / (sa sb)
    if constexpr (MATH == 2) lvt_f16x2_finish<TM, TN>(acc, acx, unscale);

    if constexpr (COLSUM) {
        if (bl.sum_on) bl.write_colsum(lds, p.colsum_partial + (long long)split * p.N, n0, p.N, tid);   // staging LDS is free now
    }
    if constexpr (COLSUM_A) {
        if (al.sum_on) al.write_colsum(lds, p.colsum_partial + ((long long)split * gridDim.y + z) * p.M, m0, p.M, tid);
    }
#ifndef LVT_NO_FAST_EPILOGUE
    if constexpr (AMODE != A_CONVT_K && TM == 2 && TN == 2 && LDS_FLOATS >= TURN_FLOATS + 16) {
        if (p.vec_epi && !(p.flags & (LVT_EPI_PLANES | LVT_EPI_ACCUM | LVT_EPI_TANH))) {
            // every form whose tile rows ARE the rows of C: epilogue_fast.h (same arithmetic, see there)
            LvtEpi e;
            e.M = p.M; e.N = p.N;
            float *wave_tile = lds + wave * (32 * TN * 32);
            if (COLSUM || COLSUM_A) __syncthreads();                 // (the column sums above went through the same LDS)
            if (p.splits > 1) {
                e.C = p.partial; e.ldc = p.N; e.coff = split * p.partial_stride + (long long)z * p.M * p.N; e.alpha = 1.f; e.flags = 0;
                e.bias = nullptr; e.res = nullptr; e.ldr = 0; e.mask = nullptr; e.ldm = 0;
                (void)lvt_epi_fast_wave<0, TM, TN>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
                return;
            }
            const unsigned seen = lvt_amax_peek(p.c_amax);
            e.C = p.C; e.ldc = p.ldc; e.coff = coff; e.alpha = p.alpha; e.flags = p.flags;
            e.bias = p.bias; e.res = p.res; e.ldr = p.ldr; e.mask = p.mask; e.ldm = p.ldm;
            const float am_w = lvt_epi_fast_dispatch<TM, TN>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
            if (p.c_amax) lvt_block_amax_commit_seen(am_w, p.c_amax, lds + TURN_FLOATS, seen);
            return;
        }
    }
#endif
    if (p.vec_epi) lvt_epilogue_vec<AMODE, BM, BN, WM, WN>(p, acc, lds, m0, n0, wm, wn, lane, cls, coff, z, split);
    else lvt_epilogue<AMODE, BM, BN, WM, WN>(p, acc, m0, n0, wm, wn, l31, half, cls, coff, z, split);
}

// ------------------------------------------------------------------------------------------------
// Frame-resident 3x3 convolution (stride 1, pad 1, 16x16 frames): the implicit GEMM above re-stages every input
// element once per tap and per n-tile -- 9 x 2 trips global -> registers -> split -> LDS for the 256-channel layers, which
// is where a third of the engine time goes (staging A on every 8th k-tile only: +17 %, A on every 8th and B on every
// 2nd: +28 %, profiles/r02_engine_staging_experiments.txt).  Here one workgroup (8 waves) owns one whole FRAME x 128
// output channels: per 32-channel chunk the 18x18 input patch (frame + zero halo) is split and staged ONCE, pixel-major,
// and the nine taps read their A operands from it at a pixel offset; only the 32 x 128 weight tile of each (chunk, tap)
// step is staged (double-buffered, one barrier per step).  Data staged per MFMA: 1/3.5 of the implicit GEMM's.
// GEMM k order is (chunk, tap, channel-in-chunk); rows are pixels in the patch_orow order.
// ------------------------------------------------------------------------------------------------
#define PT_PW 18
#define PT_PIX (PT_PW * PT_PW)
#define PT_PLANE (PT_PIX * HLD)                 // bf16 per patch plane
#define PT_THREADS 512
// MODE 0: 3x3 / pad 1 convolution, nine taps at patch offsets (dy, dx) in 0..2.
// MODE 1: ONE output phase (py, px) of the 4x4 / stride 2 / pad 1 TRANSPOSED convolution of a 16x16 frame (ConvTranspose
//         forward = backward-data of the strided convolution): four taps at patch offsets (py + a, px + b), a, b in 0..1,
//         over the same 18x18 patch; weights packed [phase][tap][Cin][Cout] (lvt_conv3d_pack_weight_phases), rows written
//         to pixels (2y + py, 2x + px) of the 32x32 output frame.  The implicit GEMM re-stages the input per tap AND phase.
// MODE 2: the 4x4 / stride 2 / pad 1 convolution of a 32x32 frame (-> 16x16): the sixteen taps are four PARITY classes
//         (ky & 1, kx & 1) x four taps (ky >> 1, kx >> 1); the input pixels of one class, in[2r + py - 1][2c + px - 1], form
//         a 17x17 sub-image on which the class is a 2x2 / stride 1 convolution.  The "patch" is that sub-image, re-staged per
//         (chunk, class): one staging per four tap steps.  Weights packed [class][tap][Cin][Cout]
//         (lvt_conv3d_pack_weight_parity).
// MATH: 1 = bf16x3 planes (six MFMAs per block), 2 = f16x2 planes (three, two accumulators; lvt_gemm_kernel above).
// BIMG (f16x2 only, round 6): the weight tile of a step arrives as a ready LDS image -- the two fp16 planes of the 32 x 128 tile in
//         the hrow<128> order, pads included, 20992 contiguous bytes per tile, made once per pass by lvt_conv3d_weight_images -- and
//         goes global memory -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear on both sides): no registers, no split, no
//         ds_write, and no longer the work of the first four waves only.  Same bits in LDS as the in-kernel split.
typedef __attribute__((address_space(3))) void pt_lds_void;
typedef __attribute__((address_space(1))) const void pt_gl_void;
#define PT_BIMG_BYTES (2 * HPlane<128>::SIZE * 2)
template <int MODE, int MATH, int BIMG = 0>
__global__ __launch_bounds__(PT_THREADS) void lvt_conv_patch_kernel(const KParams p) {
    static_assert(!BIMG || MATH == 2, "weight images are f16x2 planes");
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, TM = 2, TN = 2;
    constexpr int NTAPS = MODE == 0 ? 9 : 4;
    constexpr int NP = MATH == 2 ? 2 : 3;
    constexpr int PSB = HPlane<BN>::SIZE;
    // f16x2 (two planes): TWO patch images -- the patch of chunk c + 1 is fetched during the second-to-last tap step of chunk c
    // and split + stored into the other image right behind the last tap step's MFMAs, instead of between two barriers
    // of its own with the matrix pipe idle (bf16x3: three planes, one image fits).  -DLVT_PX_ONE_PATCH=1: the round-4 form.
#ifndef LVT_PX_ONE_PATCH
#define LVT_PX_ONE_PATCH 0
#endif
    constexpr bool DBLA = MATH == 2 && !LVT_PX_ONE_PATCH;
    constexpr int A_IMG = NP * PT_PLANE;                                      // one patch image (bf16 / fp16 elements)
    constexpr int A_BYTES = A_IMG * 2 * (DBLA ? 2 : 1), B_BYTES = NP * PSB * 2;
    constexpr int STAGE_FLOATS = (A_BYTES + 2 * B_BYTES) / 4 + 8;
    constexpr int TURN_FLOATS = WM * WN * 32 * (TN * 32);
    constexpr int LDS_FLOATS = STAGE_FLOATS > TURN_FLOATS ? STAGE_FLOATS : TURN_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    unsigned short *Ah = reinterpret_cast<unsigned short *>(lds);
    unsigned short *Bh0 = Ah + A_IMG * (DBLA ? 2 : 1);
    int unscale = 0;
    float sa = 1.f, sb = 1.f;
    if (MATH == 2) { sa = lvt_f16_scale(p.a_amax, unscale); sb = lvt_f16_scale(p.b_amax, unscale); }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int Ci = p.g.Ci;
    const int ntn = p.N / BN;
    int wg = blockIdx.x;
    {   // XCD-contiguous tile order (see lvt_tile_ctx): the n-tiles of a frame share its patch in one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = wg & 7, slot = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n0 = (wg % ntn) * BN;
    const int frame = MODE == 1 ? (wg / ntn) >> 2 : wg / ntn, phase = MODE == 1 ? (wg / ntn) & 3 : 0, m0 = frame * BM;
    const float *xf = p.A + (long long)frame * (MODE == 2 ? 1024 : 256) * Ci;

    // ---- patch staging: unit u = pixel * 8 + channel quad; 2592 units over 512 threads -> 6 passes
    // (MODE 2 walks only the 17x17 sub-image pixels its taps read, placed at the 18-pixel pitch: 5 passes)
    constexpr int PWALK = MODE == 2 ? 17 : PT_PW;
    constexpr int PUNITS = PWALK * PWALK * 8, PPASS = (PUNITS + PT_THREADS - 1) / PT_THREADS;
    float4 pv[PPASS];
    // `chunk`: the 32-channel chunk (MODE 0 / 1), or chunk * 4 + parity class (MODE 2)
    auto patch_fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < PPASS; ++j) {
            const int u = tid + PT_THREADS * j;
            const int pp = u >> 3, q = u & 7;
            const int py = pp / PWALK, px = pp - py * PWALK;
            if (MODE == 2) {
                const int iy = 2 * py + ((chunk >> 1) & 1) - 1, ix = 2 * px + (chunk & 1) - 1;      // sub-image pixel -> input pixel
                const bool ok = u < PUNITS && (unsigned)iy < 32u && (unsigned)ix < 32u;
                pv[j] = ok ? ldg4(xf + (iy * 32 + ix) * Ci + (chunk >> 2) * 32 + q * 4) : zero4();
            } else {
                const bool ok = u < PUNITS && (unsigned)(py - 1) < 16u && (unsigned)(px - 1) < 16u;
                const float *src = xf + ((py - 1) * 16 + (px - 1)) * Ci + chunk * 32 + q * 4;
                pv[j] = ok ? ldg4(src) : zero4();
            }
        }
    };
    auto patch_store = [&](unsigned short *img) {
#pragma unroll
        for (int j = 0; j < PPASS; ++j) {
            const int u = tid + PT_THREADS * j;
            if (u < PUNITS) {
                const int pp = u >> 3;
                unsigned short *d = img + (MODE == 2 ? pp + pp / PWALK : pp) * HLD + (u & 7) * 4;
                if (MATH == 2) {
                    uint2 ph, pl;
                    split2(pv[j], sa, ph, pl);
                    *reinterpret_cast<uint2 *>(d) = ph;
                    *reinterpret_cast<uint2 *>(d + PT_PLANE) = pl;
                } else {
                    uint2 p1, p2, p3;
                    split3(pv[j], p1, p2, p3);
                    *reinterpret_cast<uint2 *>(d) = p1;
                    *reinterpret_cast<uint2 *>(d + PT_PLANE) = p2;
                    *reinterpret_cast<uint2 *>(d + 2 * PT_PLANE) = p3;
                }
            }
        }
    };
    // ---- weight tile of a step: 32 k rows x 128 columns, staged by the first 256 threads (4 k rows x 4 columns each,
    // transposed in registers into k-contiguous 8-byte LDS rows, as BLoader<B_NPLAIN> does)
    const bool bact = tid < 256;
    const int bkk0 = (tid >> 5) & 7, bnq = tid & 31;
    const float *bcol = p.B + n0 + bnq * 4;
    float4 bv[4];
    auto b_fetch = [&](int step) {
        const int cc = step / NTAPS, tap = step - cc * NTAPS;
        // MODE 2: cc = chunk * 4 + class -> weight rows ((class * 4 + tap) * Ci + chunk * 32 ...)
        const int wrow = MODE == 2 ? ((cc & 3) * NTAPS + tap) * Ci + (cc >> 2) * 32 : (phase * NTAPS + tap) * Ci + cc * 32;
        const float *src = bcol + (long long)(wrow + bkk0 * 4) * p.ldb;
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = ldg4(src + (long long)i * p.ldb);
    };

    f32x16 acc[TM][TN];
    f32x16 acx[MATH == 2 ? TM : 1][MATH == 2 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (MATH == 2) acx[i][j][r] = 0.f;
            }
    auto b_store = [&](unsigned short *dst) {
        if (MATH == 2) store_split2_block<BN>(dst, bnq * 4, bkk0 * 4, bv, sb);
        else store_split_block<BN>(dst, bnq * 4, bkk0 * 4, bv);
    };
    // BIMG: unit u (16 bytes) of the tile image -> LDS byte 16 u of the buffer; thread t moves units t, t + 512 and (t < 288) t + 1024
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    constexpr int B_UNITS = PT_BIMG_BYTES / 16;                       // 1312
    static_assert(PT_BIMG_BYTES == NP * PSB * 2 || !BIMG, "image = the LDS buffer");
    const char *bimg0 = BIMG ? p.b_img + (long long)(n0 / BN) * PT_BIMG_BYTES + tid * 16 : nullptr;
    auto b_dma = [&](int step, unsigned short *dst) {
        const int cc = step / NTAPS, tap = step - cc * NTAPS;
        const int wrow = MODE == 2 ? ((cc & 3) * NTAPS + tap) * Ci + (cc >> 2) * 32 : (phase * NTAPS + tap) * Ci + cc * 32;
        const char *src = bimg0 + (long long)(wrow >> 5) * ntn * PT_BIMG_BYTES;
        char *d = reinterpret_cast<char *>(dst) + wave_s * 1024;
        __builtin_amdgcn_global_load_lds((pt_gl_void *)src, (pt_lds_void *)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((pt_gl_void *)(src + 8192), (pt_lds_void *)(d + 8192), 16, 0, 0);
        if (tid + 1024 < B_UNITS) __builtin_amdgcn_global_load_lds((pt_gl_void *)(src + 16384), (pt_lds_void *)(d + 16384), 16, 0, 0);
    };
    // the LDS-DMA landing fence of gemm_p2.hip (profiles/r06_lds_dma_visibility.txt): vmcnt(0), then ONE LDS read by the issuing wave
    // from each chunk it requested (lane -> chunk lane % 3, or % 2 for the waves without a third one), before the barrier
    const int b_nprobe = wave_s * 64 + 1024 < B_UNITS ? 3 : 2;
    const unsigned b_probe = (unsigned)((lane % b_nprobe) * 8192 + wave_s * 1024 + (lane & 31) * 16);
    auto b_landed = [&](const unsigned short *dst) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int v = *reinterpret_cast<const volatile int *>(reinterpret_cast<const char *>(dst) + b_probe);
        (void)v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // A operand rows of this lane: MFMA tile i of the wave covers image rows 2*(2*wm + i) + {0, 1}
    int arow[TM];
    {
        const int q = l31 >> 2, e = l31 & 3;
        const int yy = __popc(q) & 1, x = 4 * (q >> 1) + e;
#pragma unroll
        for (int i = 0; i < TM; ++i) arow[i] = ((2 * (2 * wm + i) + yy) * PT_PW + x) * HLD + 8 * half;
    }
    int brow[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) brow[j] = hrow<BN>(wn * (TN * 32) + j * 32 + l31) + 8 * half;

    const int nchunks = (Ci / 32) * (MODE == 2 ? 4 : 1), nsteps = nchunks * NTAPS;
    patch_fetch(0);
    if (BIMG) b_dma(0, Bh0);
    else if (bact) b_fetch(0);
    patch_store(Ah);
    if (BIMG) b_landed(Bh0);
    else if (bact) b_store(Bh0);
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int cc = step / NTAPS, tap = step - cc * NTAPS;
        const bool has_next = step + 1 < nsteps;
        const bool new_chunk = has_next && tap == NTAPS - 1;
        // (BIMG: the other weight buffer was last read in step - 1, which every wave has left)
        if (BIMG) { if (has_next) b_dma(step + 1, Bh0 + ((step + 1) & 1) * (NP * PSB)); }
        else if (has_next && bact) b_fetch(step + 1);
        if (DBLA ? (cc + 1 < nchunks && tap == NTAPS - 2) : new_chunk) patch_fetch(cc + 1);
        const unsigned short *Bh = Bh0 + (step & 1) * (NP * PSB);
        const unsigned short *Acur = Ah + (DBLA ? (cc & 1) * A_IMG : 0);
        const unsigned short *Ap = MODE == 0 ? Acur + ((tap / 3) * PT_PW + (tap % 3)) * HLD
                                             : Acur + (((phase >> 1) + (tap >> 1)) * PT_PW + (phase & 1) + (tap & 1)) * HLD;   // (phase 0 in MODE 2)
        if constexpr (MATH == 2) {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                f16x8 a[2][TM], b[2][TN];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const f16x8 *>(Ap + arow[i] + q * PT_PLANE + ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const f16x8 *>(Bh + brow[j] + q * PSB + ks);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acx[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], acx[i][j], 0, 0, 0);
            }
            // the other image was last read in chunk cc - 1 (every wave has passed NTAPS barriers since): its split + store follows
            // the issue of this step's MFMAs with no barrier in between, the matrix pipe drains beside it.  (As part of the MFMA
            // block itself -- one basic block for the scheduler to interleave -- the allocator spilled ~320 registers.)
            if (DBLA && new_chunk) patch_store(Ah + ((cc + 1) & 1) * A_IMG);
        } else {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                bf16x8 a[3][TM], b[3][TN];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const bf16x8 *>(Ap + arow[i] + q * PT_PLANE + ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const bf16x8 *>(Bh + brow[j] + q * PSB + ks);
                }
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
            }
        }
        // the other weight buffer was last read in the previous step, which every wave has left (barrier below)
#ifndef LVT_PX_NOBSPLIT      // (timing experiment: the weight tile is split + stored for step 0 only)
        if (!BIMG && has_next && bact) b_store(Bh0 + ((step + 1) & 1) * (NP * PSB));
#endif
        if (!DBLA && new_chunk) {
            __syncthreads();              // every wave is done with the old patch
            patch_store(Ah);
        }
        if (BIMG && has_next) b_landed(Bh0 + ((step + 1) & 1) * (NP * PSB));
#ifdef LVT_PX_HALFBARRIERS   // (timing experiment, wrong results: a barrier every second step)
        if ((step & 1) || new_chunk)
#endif
        __syncthreads();
    }
    if constexpr (MATH == 2) lvt_f16x2_finish<TM, TN>(acc, acx, unscale);
#ifndef LVT_NO_FAST_EPILOGUE
    if (p.vec_epi && p.splits <= 1 && !(p.flags & (LVT_EPI_PLANES | LVT_EPI_ACCUM | LVT_EPI_TANH))) {
        // epilogue_fast.h with the pixel permutation of the frame as its row map
        const unsigned seen = lvt_amax_peek(p.c_amax);
        LvtEpi e;
        e.M = p.M; e.N = p.N; e.C = p.C; e.ldc = p.ldc; e.coff = 0; e.alpha = p.alpha; e.flags = p.flags;
        e.bias = p.bias; e.res = p.res; e.ldr = p.ldr; e.mask = p.mask; e.ldm = p.ldm;
        float *wave_tile = lds + wave * (32 * TN * 32);
        float am_w;
        if (MODE != 1) am_w = lvt_epi_fast_dispatch<TM, TN, PatchRows>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, PatchRows());
        else am_w = lvt_epi_fast_dispatch<TM, TN, PatchTRows>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, PatchTRows{phase});
        if (p.c_amax) lvt_block_amax_commit_seen(am_w, p.c_amax, lds + TURN_FLOATS, seen);
        return;
    }
#endif
    if (MODE != 1) lvt_epilogue_vec<A_PATCH, BM, BN, WM, WN>(p, acc, lds, m0, n0, wm, wn, lane, 0, 0, 0, 0);
    else lvt_epilogue_vec<A_PATCHT, BM, BN, WM, WN>(p, acc, lds, m0, n0, wm, wn, lane, phase, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Wide software-pipelined GEMM for the f16x2 arithmetic (round 4).  With three MFMAs per block instead of six the 128x128
// kernel above is bound by what surrounds the matrix instructions: per k-tile a wave waits for its global loads, splits,
// stores and passes two barriers before the next MFMA block can start (matrix pipe 39 % busy on the 16384 x 512 x 3072
// product).  Here
//   * one workgroup of 8 waves owns a 256 x 128 tile (wave = 64 x 64 as before): a third less staging per MFMA;
//   * the two fp16 planes of both operands are DOUBLE-BUFFERED in LDS (2 x 61.5 KB, one workgroup per CU): the split and
//     the stores of tile k+1 are independent of the MFMA block of tile k and share its issue window, and there is ONE
//     barrier per k-tile;
//   * the global loads of tile k+2 are issued as soon as the registers of tile k+1 have been split: a full iteration of
//     latency cover.
// Serves the plain forms (NT / NN / TN incl. 2-level k, batches, split-K with column sums); everything else and the other
// arithmetic modes stay on lvt_gemm_kernel.
// ------------------------------------------------------------------------------------------------
#define WIDE_THREADS 512
template <int TA, int TB>
__global__ __launch_bounds__(WIDE_THREADS, 2) void lvt_gemm_wide_kernel(const KParams p) {
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, TM = 2, TN = 2;
    constexpr int AMODE = TA ? A_MPLAIN : A_KPLAIN;
    constexpr int PSA = HPlane<BM>::SIZE, PSB = HPlane<BN>::SIZE;
    constexpr int STAGE = 2 * (PSA + PSB);                                  // fp16 elements per buffer
    constexpr int STAGE_FLOATS = (2 * STAGE) / 2 + 8;
    constexpr int TURN_FLOATS = WM * WN * 32 * (TN * 32);
    constexpr int LDS_FLOATS = STAGE_FLOATS > TURN_FLOATS ? STAGE_FLOATS : TURN_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    unsigned short *S0 = reinterpret_cast<unsigned short *>(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const TileCtx tc = lvt_tile_ctx<AMODE, BM, BN>(p);
    const int m0 = tc.m0, n0 = tc.n0, kbeg = tc.kbeg, kend = tc.kend;

    int unscale = 0;
    const float sa = lvt_f16_scale(p.a_amax, unscale, p.a_amax2), sb = lvt_f16_scale(p.b_amax, unscale, p.b_amax2);

    // ---- operand fetch state.  k-contiguous: thread = (row r0 + 64 i, k quad kq); m-contiguous: thread = (4 k rows, 4 columns).
    // The main loop carries NO bounds checks (one basic block: the split of tile k+1 and the loads of tile k+2 interleave
    // with the MFMAs of tile k): rows / columns beyond M / N are fetched from a clamped, valid address -- they only feed
    // output rows / columns that the epilogue does not store -- and the launcher sends K % 32 != 0 to lvt_gemm_kernel.
    float4 av[4], bv[TB ? 4 : 2];
    const int r0 = tid >> 3, kq = tid & 7;            // k-contiguous
    const int kk0 = TA ? tid >> 6 : 0, mq = tid & 63; // A m-contiguous: 8 k groups x 64 m quads
    const int bkk0 = (tid >> 5) & 7, bnq = tid & 31;  // B n-contiguous: 8 k groups x 32 n quads (threads 0..255)
    const bool bact = !TB || tid < 256;
    // Addresses = a workgroup-uniform base that walks k (scalar registers, scalar arithmetic) + per-thread BYTE offsets that
    // never change (the launcher keeps every operand below 4 GB per batch): the loop has no vector address arithmetic --
    // 43 of its 115 VALU instructions before this.
    const char *Ak = reinterpret_cast<const char *>(tc.A), *Bk = reinterpret_cast<const char *>(tc.B);
    unsigned aoff[4], boff_g[TB ? 4 : 2];
    int a_kin = 0, b_kin = 0;                           // position of the tile inside its k block (2-level k), uniform
    if (!TA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(((long long)min(m0 + r0 + 64 * i, p.M - 1) * p.lda + kq * 4) * 4);
        const int blk = kbeg / p.a_kb;
        a_kin = kbeg - blk * p.a_kb;
        Ak += ((long long)blk * p.a_skb + a_kin) * 4;
    } else {
        const int m = m0 + mq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(((long long)(kk0 * 4 + i) * p.lda + (m < p.M ? m : 0)) * 4);
        Ak += (long long)kbeg * p.lda * 4;
    }
    if (!TB) {
#pragma unroll
        for (int i = 0; i < 2; ++i) boff_g[i] = (unsigned)(((long long)min(n0 + r0 + 64 * i, p.N - 1) * p.ldb + kq * 4) * 4);
        const int blk = kbeg / p.b_kb;
        b_kin = kbeg - blk * p.b_kb;
        Bk += ((long long)blk * p.b_skb + b_kin) * 4;
    } else {
        const int n = n0 + bnq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) boff_g[i] = (unsigned)(((long long)(bkk0 * 4 + i) * p.ldb + (n < p.N ? n : 0)) * 4);
        Bk += (long long)kbeg * p.ldb * 4;
    }
    const bool sum_on = TA && p.colsum_partial != nullptr && n0 == 0;
    float4 colacc = zero4();
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const float4 *>(Ak + aoff[i]);
        if (!TA) {
            a_kin += BK;
            const bool wrap = a_kin >= p.a_kb;              // (a_kb is a multiple of BK: a tile never straddles two k blocks)
            Ak += wrap ? ((long long)p.a_skb - p.a_kb + BK) * 4 : (long long)BK * 4;
            a_kin = wrap ? 0 : a_kin;
        } else {
            Ak += (long long)BK * p.lda * 4;
            if (sum_on) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { colacc.x += av[i].x; colacc.y += av[i].y; colacc.z += av[i].z; colacc.w += av[i].w; }
            }
        }
        if (!TB) {
#pragma unroll
            for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const float4 *>(Bk + boff_g[i]);
            b_kin += BK;
            const bool wrap = b_kin >= p.b_kb;
            Bk += wrap ? ((long long)p.b_skb - p.b_kb + BK) * 4 : (long long)BK * 4;
            b_kin = wrap ? 0 : b_kin;
        } else {
            if (bact) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(Bk + boff_g[i]);
            }
            Bk += (long long)BK * p.ldb * 4;
        }
    };
    auto store = [&](unsigned short *buf) {
        unsigned short *Ah = buf, *Bh = buf + 2 * PSA;
        if (!TA) {
#pragma unroll
#ifdef LVT_WX_ARAW
            for (int i = 0; i < 4; ++i) store_raw2_k<BM>(Ah, r0 + 64 * i, kq * 4, av[i]);
#else
            for (int i = 0; i < 4; ++i) store_split2_k<BM>(Ah, r0 + 64 * i, kq * 4, av[i], sa);
#endif
        } else {
            store_split2_block<BM>(Ah, mq * 4, kk0 * 4, av, sa);
        }
        if (!TB) {
#pragma unroll
#ifdef LVT_WX_BRAW
            for (int i = 0; i < 2; ++i) store_raw2_k<BN>(Bh, r0 + 64 * i, kq * 4, bv[i]);
#else
            for (int i = 0; i < 2; ++i) store_split2_k<BN>(Bh, r0 + 64 * i, kq * 4, bv[i], sb);
#endif
        } else if (bact) {
            store_split2_block<BN>(Bh, bnq * 4, bkk0 * 4, bv, sb);
        }
    };

    f32x16 acc[TM][TN], acx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }

    int fa[TM], fb[TN];                     // LDS offsets of this lane's operand fragments
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = hrow<BM>(wm * (TM * 32) + i * 32 + l31) + 8 * half;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = 2 * PSA + hrow<BN>(wn * (TN * 32) + j * 32 + l31) + 8 * half;

    // one k-tile: the MFMA block on buffer `cur`; with STORE the registers (tile kt+1) are split into the other buffer, with
    // FETCH the loads of tile kt+2 follow -- no dependence on the MFMAs: one basic block that the scheduler interleaves
    // (after the first 12 MFMAs in program order).  Measured alternatives (profiles/r04_wide_gemm_loop_experiments.txt): the
    // two waves of a SIMD in anti-phase (waves 4..7 split before their MFMA block, 0..3 after it, straight-line blocks):
    // 7 % SLOWER than this interleave.
    auto tile = [&](const unsigned short *cur, unsigned short *nxt, auto do_store, auto do_fetch) {
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            f16x8 a[2][TM], b[2][TN];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const f16x8 *>(cur + fa[i] + q * PSA + ks);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const f16x8 *>(cur + fb[j] + q * PSB + ks);
            }
#ifdef LVT_WX_NOMFMA        // (timing experiments, tools/profile/build_variant.sh: what the main loop costs without one of its parts)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j][0] += (float)a[0][i][0] + (float)b[0][j][0] + (float)a[1][i][1] + (float)b[1][j][1];
                    acx[i][j][0] += (float)a[0][i][7] + (float)b[0][j][7];
                }
#else
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], acx[i][j], 0, 0, 0);
#endif
            if (ks == 0) {
#ifndef LVT_WX_NOSPLIT
                if constexpr (decltype(do_store)::value) store(nxt);
#endif
#ifndef LVT_WX_NOFETCH
                if constexpr (decltype(do_fetch)::value) fetch();
#endif
            }
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;

    const int ntiles = kbeg < kend ? (kend - kbeg) / BK : 0;
    if (ntiles > 0) {
        fetch();
        store(S0);
        if (ntiles > 1) fetch();
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < ntiles; ++kt) {
        tile(S0 + (kt & 1) * STAGE, S0 + ((kt + 1) & 1) * STAGE, yes_t(), yes_t());
#ifndef LVT_WX_NOBARRIER    // (timing experiment, wrong results)
        __syncthreads();
#endif
    }
    if (kt + 1 < ntiles) {
        tile(S0 + (kt & 1) * STAGE, S0 + ((kt + 1) & 1) * STAGE, yes_t(), no_t());
        __syncthreads();
        ++kt;
    }
    if (kt < ntiles) tile(S0 + (kt & 1) * STAGE, nullptr, no_t(), no_t());
    __syncthreads();
    lvt_f16x2_finish<TM, TN>(acc, acx, unscale);

    if (TA && sum_on) {
        // column sums of everything this workgroup fetched (bias gradient): lanes with the same m quad are added in kk0 order
        float *scratch = lds;
        *reinterpret_cast<float4 *>(&scratch[(kk0 * 64 + mq) * 4]) = colacc;
        __syncthreads();
        if (tid < BM) {
            float s_ = 0.f;
            for (int r = 0; r < 8; ++r) s_ += scratch[(r * 64 + tid / 4) * 4 + (tid & 3)];
            if (m0 + tid < p.M) p.colsum_partial[((long long)tc.split * gridDim.y + tc.z) * p.M + m0 + tid] = s_;
        }
        __syncthreads();
    }
#ifndef LVT_NO_FAST_EPILOGUE
    if (p.vec_epi && !(p.flags & (LVT_EPI_PLANES | LVT_EPI_ACCUM | LVT_EPI_TANH))) {
        // plain forms (every launch of the transformer): epilogue_fast.h -- same arithmetic, compile-time flag sets, no workgroup
        // barriers, max |C| peeked before the stores
        LvtEpi e;
        e.M = p.M; e.N = p.N;
        float *wave_tile = lds + wave * (32 * TN * 32);
        if (p.splits > 1) {
            e.C = p.partial; e.ldc = p.N; e.coff = tc.split * p.partial_stride + (long long)tc.z * p.M * p.N; e.alpha = 1.f; e.flags = 0;
            e.bias = nullptr; e.res = nullptr; e.ldr = 0; e.mask = nullptr; e.ldm = 0;
            (void)lvt_epi_fast_wave<0, TM, TN>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
            return;
        }
        const unsigned seen = lvt_amax_peek(p.c_amax);
        e.C = p.C; e.ldc = p.ldc; e.coff = tc.coff; e.alpha = p.alpha; e.flags = p.flags;
        e.bias = p.bias; e.res = p.res; e.ldr = p.ldr; e.mask = p.mask; e.ldm = p.ldm;
        const float am_w = lvt_epi_fast_dispatch<TM, TN>(e, acc, wave_tile, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
        if (p.c_amax) lvt_block_amax_commit_seen(am_w, p.c_amax, lds + TURN_FLOATS, seen);
        return;
    }
#endif
    if (p.vec_epi) lvt_epilogue_vec<AMODE, BM, BN, WM, WN>(p, acc, lds, m0, n0, wm, wn, lane, 0, tc.coff, tc.z, tc.split);
    else lvt_epilogue<AMODE, BM, BN, WM, WN>(p, acc, m0, n0, wm, wn, l31, half, 0, tc.coff, tc.z, tc.split);
}

// deterministic split-K reduction: out[i] (+)= sum_s partial[s][i]
__global__ void lvt_reduce_splits_kernel(const float *__restrict__ partial, long long n4, long long stride,
                                         int splits, float *__restrict__ out, int accumulate) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += step) {
        // four independent running sums (the loads of a thread do not wait on each other), combined in a fixed order
        float4 s0 = ldg4(partial + i * 4), s1 = zero4(), s2 = zero4(), s3 = zero4();
        int k = 1;
        for (; k + 3 < splits; k += 4) {
            const float4 t0 = ldg4(partial + k * stride + i * 4), t1 = ldg4(partial + (k + 1) * stride + i * 4);
            const float4 t2 = ldg4(partial + (k + 2) * stride + i * 4), t3 = ldg4(partial + (k + 3) * stride + i * 4);
            s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
            s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
            s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
            s3.x += t3.x; s3.y += t3.y; s3.z += t3.z; s3.w += t3.w;
        }
        for (; k < splits; ++k) {
            const float4 t = ldg4(partial + k * stride + i * 4);
            s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
        }
        float4 s;
        s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
        s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
        float4 *o = reinterpret_cast<float4 *>(out + i * 4);
        if (accumulate) { const float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        *o = s;
    }
}

// conv weight pack / unpack -----------------------------------------------------------------------
// w[co][ci][tap] -> wp[tap][ci_pad][co_pad]
__global__ void lvt_pack_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int taps, int Ci,
                                       int Co, int Ci_real, int Co_real) {
    const long long total = (long long)taps * Ci * Co;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int co = i % Co; long long t = i / Co;
        const int ci = t % Ci; const int tap = t / Ci;
        float v = 0.f;
        if (co < Co_real && ci < Ci_real) v = w[((long long)co * Ci_real + ci) * taps + tap];
        wp[i] = v;
    }
}
// w[co][ci][tap] -> wt[taps-1-tap][co_pad][ci_pad]: the packed weights of the convolution that IS the backward-data pass of a
// stride-1 convolution (input / output channels swapped, taps reversed)
__global__ void lvt_pack_weight_t_kernel(const float *__restrict__ w, float *__restrict__ wt, int taps, int Ci,
                                         int Co, int Ci_real, int Co_real) {
    const long long total = (long long)taps * Co * Ci;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % Ci; long long t = i / Ci;
        const int co = t % Co; const int tr = t / Co;
        float v = 0.f;
        if (co < Co_real && ci < Ci_real) v = w[((long long)co * Ci_real + ci) * taps + (taps - 1 - tr)];
        wt[i] = v;
    }
}
// w[co][ci][ky][kx] (4x4) -> wph[phase (py,px)][tap (a,b)][co_pad][ci_pad] with ky = 3 - py - 2a, kx = 3 - px - 2b: the
// four taps each output phase of the stride-2 transposed convolution uses, in the order the frame-resident kernel walks them
__global__ void lvt_pack_weight_phases_kernel(const float *__restrict__ w, float *__restrict__ wph, int Ci, int Co,
                                              int Ci_real, int Co_real) {
    const long long total = 16LL * Co * Ci;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % Ci; long long t = i / Ci;
        const int co = t % Co; const int pt = t / Co;         // pt = phase * 4 + tap
        const int ph = pt >> 2, tap = pt & 3;
        const int ky = 3 - (ph >> 1) - 2 * (tap >> 1), kx = 3 - (ph & 1) - 2 * (tap & 1);
        float v = 0.f;
        if (co < Co_real && ci < Ci_real) v = w[(((long long)co * Ci_real + ci) * 4 + ky) * 4 + kx];
        wph[i] = v;
    }
}
// w[co][ci][ky][kx] (4x4) -> wq[class (py,px)][tap (a,b)][ci_pad][co_pad] with ky = 2a + py, kx = 2b + px (MODE 2 above)
__global__ void lvt_pack_weight_parity_kernel(const float *__restrict__ w, float *__restrict__ wq, int Ci, int Co,
                                              int Ci_real, int Co_real) {
    const long long total = 16LL * Ci * Co;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = i % Co; long long t = i / Co;
        const int ci = t % Ci; const int ct = t / Ci;         // ct = class * 4 + tap
        const int cls = ct >> 2, tap = ct & 3;
        const int ky = 2 * (tap >> 1) + (cls >> 1), kx = 2 * (tap & 1) + (cls & 1);
        float v = 0.f;
        if (co < Co_real && ci < Ci_real) v = w[(((long long)co * Ci_real + ci) * 4 + ky) * 4 + kx];
        wq[i] = v;
    }
}
// partial[split][(tap,ci)][co] -> dw[co][ci][tap] for Co % 64 == 0, Ci % 4 == 0 and no channel padding (the 256 / 128-channel
// layers): a workgroup owns 64 output channels x 4 input channels x all taps.  Lanes run along co while the splits are
// summed (every load of a wave is one contiguous 256-byte run; four running sums per element, combined in a fixed order),
// the tile is turned through LDS, and the stores are contiguous runs of 4 * taps floats per output channel.  (The generic
// kernel below shares an element among L lanes over the splits: 16-byte fragments on the read side, one scattered 4-byte
// store per element -- 34-48 us for the 75 MB of partials of a 3x3 256-channel layer; this one is bound by reading them.)
#define UW_CO 64
#define UW_CI 4
#ifndef UW_WAVES
#define UW_WAVES 16                      // (tap, ci) rows in flight per pass: with 4 a CU held 4 waves, each waiting on its own loads
#endif
__global__ __launch_bounds__(64 * UW_WAVES) void lvt_unpack_wgrad_tiled_kernel(const float *__restrict__ partial, long long stride,
                                                                               int splits, float *__restrict__ dw, int taps, int Ci,
                                                                               int Co) {
    extern __shared__ float tile[];                       // [UW_CO][UW_CI * taps + 1]
    const int ld = UW_CI * taps + 1;
    const int nco = Co / UW_CO;
    const int co0 = (blockIdx.x % nco) * UW_CO, ci0 = (blockIdx.x / nco) * UW_CI;
    const int lane_co = threadIdx.x & (UW_CO - 1), r0 = threadIdx.x >> 6;
    const int rows = taps * UW_CI;
    for (int r = r0; r < rows; r += UW_WAVES) {
        const int tap = r / UW_CI, cil = r % UW_CI;
        const float *src = partial + ((long long)tap * Ci + ci0 + cil) * Co + co0 + lane_co;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            s0 += src[k * stride]; s1 += src[(k + 1) * stride]; s2 += src[(k + 2) * stride]; s3 += src[(k + 3) * stride];
        }
        for (; k < splits; ++k) s0 += src[k * stride];
        tile[lane_co * ld + cil * taps + tap] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int per_co = UW_CI * taps;                      // contiguous floats of dw per output channel
    for (int e = threadIdx.x; e < UW_CO * per_co; e += 64 * UW_WAVES) {
        const int col = e / per_co, off = e % per_co;
        dw[((long long)(co0 + col) * Ci + ci0) * taps + off] = tile[col * ld + off];
    }
}

// partial[split][(tap,ci)][co] -> dw[co][ci][tap]   (fixed summation order over splits).
// L lanes share one output element (L a power of two <= 64, chosen from the split count): lane s adds splits
// s, s+L, ... and the L lane sums are combined by a butterfly -- a fixed tree.  With one thread per element the
// image-side layers (1 tile, 512 splits) paid a chain of 512 dependent loads per thread.  Threads walk the SOURCE
// layout (co fastest) so the reads are contiguous per split; the one scattered write per element is the cheap side.
__global__ void lvt_unpack_wgrad_kernel(const float *__restrict__ partial, long long stride, int splits, int L,
                                        float *__restrict__ dw, int taps, int Ci, int Co, int Ci_real,
                                        int Co_real, const float *__restrict__ colsum_partial, float *__restrict__ db) {
    if (db) {
        // bias gradient: one wave per output channel, lanes over the splits, butterfly sum (a fixed tree)
        const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
        if (wave < Co_real) {
            float s = 0.f;
            for (int k = lane; k < splits; k += 64) s += colsum_partial[(long long)k * Co + wave];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
            if (lane == 0) db[wave] = s;
        }
    }
    const long long total = (long long)taps * Ci * Co;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(gtid & (L - 1));
    // the L lanes of an element share `i`, so they enter and leave the loop together and the butterfly only ever
    // exchanges inside such a group
    for (long long i = gtid / L; i < total; i += ((long long)gridDim.x * blockDim.x) / L) {
        float s = 0.f;
        for (int k = sub; k < splits; k += L) s += partial[k * stride + i];
        for (int d = L >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (sub == 0) {
            const int co = i % Co; long long t = i / Co;
            const int ci = t % Ci; const int tap = t / Ci;
            if (co < Co_real && ci < Ci_real) dw[((long long)co * Ci_real + ci) * taps + tap] = s;
        }
    }
}

// column sums (bias gradients).  One workgroup reduces a chunk of rows for ALL columns: thread
// (rl, c4) owns the float4 column group c4 and rows rl, rl+RL, ...; the RL row-lanes are combined
// through LDS in a fixed order.  Applied recursively (chunk partials -> final) so the result is
// bit-reproducible and every stage reads fully coalesced 16-byte lanes.
#define CS_THREADS 256
__global__ __launch_bounds__(CS_THREADS) void lvt_colsum_kernel(const float *__restrict__ g, long long M, int N,
                                                                long long ld, long long rows_per_block,
                                                                float *__restrict__ partial) {
    __shared__ float4 red[CS_THREADS];
    const int n4 = N / 4;
    const int rl_count = CS_THREADS / n4 > 0 ? CS_THREADS / n4 : 1;   // row lanes (N <= 1024)
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = min(M, r0 + rows_per_block);
    for (int cbase = 0; cbase < n4; cbase += CS_THREADS) {
        const int c4 = cbase + (int)(threadIdx.x % (n4 < CS_THREADS ? n4 : CS_THREADS));
        const int rl = threadIdx.x / (n4 < CS_THREADS ? n4 : CS_THREADS);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < n4 && rl < rl_count) {
            for (long long r = r0 + rl; r < r1; r += rl_count) {
                const float4 v = *reinterpret_cast<const float4 *>(g + r * ld + c4 * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0 && c4 < n4) {
            const int w = n4 < CS_THREADS ? n4 : CS_THREADS;
            for (int k = 1; k < rl_count; ++k) {
                const float4 v = red[k * w + threadIdx.x];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4 *>(partial + (long long)blockIdx.x * N + c4 * 4) = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// The arithmetic of a launch is chosen PER CALL by LVT_MATH_F32 in its `flags` (clear: bf16x3, the default; set: plain
// fp32 MFMA): the library keeps no mutable state, so concurrent callers (the autograd thread runs the backward launches
// of a forward issued from another thread) cannot influence each other.
static inline int math_of(int flags) { return (flags & LVT_MATH_F32) ? 0 : ((flags & LVT_MATH_F16X2) ? 2 : 1); }
// f16x2 needs the operands' max |.| (device scalars, lvt_amax_io); outputs may report theirs in any mode
static inline void set_amax(KParams &p, const lvt_amax_io *ax) {
    if (ax) { p.a_amax = ax->a; p.b_amax = ax->b; p.c_amax = ax->c; }
}
#define LVT_REQUIRE_AMAX(flags, ax, who)                                                                              \
    LVT_REQUIRE(math_of(flags) != 2 || ((ax) && (ax)->a && (ax)->b), "%s: LVT_MATH_F16X2 needs the operands' max |.| " \
                "(lvt_amax_io.a / .b: device scalars, see lvt_amax)", who)

template <int AMODE, int BMODE, int BM, int BN, int WM, int WN>
static int launch_tile(const KParams &p, int zcount, hipStream_t s) {
    const long long ntm = lvt_cdiv(p.M, BM), ntn = lvt_cdiv(p.N, BN);
    if (ntm * ntn > 0x7fffffffLL || zcount > 65535 || p.splits > 65535) {
        lvt_set_error("gemm: grid too large (%lld tiles, z=%d, splits=%d)", ntm * ntn, zcount, p.splits);
        return LVT_EINVAL;
    }
    dim3 grid((unsigned)(ntm * ntn), (unsigned)zcount, (unsigned)(p.splits > 1 ? p.splits : 1));
    KParams pv = p;
    {
        auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
        bool ok = p.N % 4 == 0;
        if (p.splits > 1) ok = ok && al16(p.partial) && p.partial_stride % 4 == 0 && ((long long)p.M * p.N) % 4 == 0;
        else {
            ok = ok && al16(p.C) && p.ldc % 4 == 0 && p.sC_o % 4 == 0 && p.sC_i % 4 == 0;
            if (p.flags & LVT_EPI_BIAS) ok = ok && al16(p.bias);
            if (p.flags & LVT_EPI_RESIDUAL) ok = ok && al16(p.res) && p.ldr % 4 == 0;
            if (p.flags & LVT_EPI_MASK) ok = ok && al16(p.mask) && p.ldm % 4 == 0;
        }
        pv.vec_epi = ok ? 1 : 0;
    }
    if ((p.flags & LVT_EPI_PLANES) && !pv.vec_epi) {
        // only the float4 epilogue writes planes: the scalar one would store fp32 into the bf16 image
        lvt_set_error("gemm: LVT_EPI_PLANES needs 16-byte aligned bias / res / mask and ldr, ldm %% 4 == 0");
        return LVT_EINVAL;
    }
    if (math_of(p.flags) == 2 && BK == 32)
        hipLaunchKernelGGL((lvt_gemm_kernel<AMODE, BMODE, BM, BN, WM, WN, 2>), grid, dim3(NTHREADS), 0, s, pv);
    else if (math_of(p.flags) == 1 && BK == 32)
        hipLaunchKernelGGL((lvt_gemm_kernel<AMODE, BMODE, BM, BN, WM, WN, 1>), grid, dim3(NTHREADS), 0, s, pv);
    else
        hipLaunchKernelGGL((lvt_gemm_kernel<AMODE, BMODE, BM, BN, WM, WN, 0>), grid, dim3(NTHREADS), 0, s, pv);
    LVT_CHECK_LAUNCH("lvt_gemm_kernel");
    return LVT_OK;
}

template <int TA, int TB>
static int launch_wide(const KParams &p, int zcount, hipStream_t s) {
    const long long ntm = lvt_cdiv(p.M, 256), ntn = lvt_cdiv(p.N, 128);
    if (ntm * ntn > 0x7fffffffLL || zcount > 65535 || p.splits > 65535) {
        lvt_set_error("gemm: grid too large (%lld tiles, z=%d, splits=%d)", ntm * ntn, zcount, p.splits);
        return LVT_EINVAL;
    }
    dim3 grid((unsigned)(ntm * ntn), (unsigned)zcount, (unsigned)(p.splits > 1 ? p.splits : 1));
    KParams pv = p;
    {
        auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
        bool ok = p.N % 4 == 0;
        if (p.splits > 1) ok = ok && al16(p.partial) && p.partial_stride % 4 == 0 && ((long long)p.M * p.N) % 4 == 0;
        else {
            ok = ok && al16(p.C) && p.ldc % 4 == 0 && p.sC_o % 4 == 0 && p.sC_i % 4 == 0;
            if (p.flags & LVT_EPI_BIAS) ok = ok && al16(p.bias);
            if (p.flags & LVT_EPI_RESIDUAL) ok = ok && al16(p.res) && p.ldr % 4 == 0;
            if (p.flags & LVT_EPI_MASK) ok = ok && al16(p.mask) && p.ldm % 4 == 0;
        }
        pv.vec_epi = ok ? 1 : 0;
    }
    if ((p.flags & LVT_EPI_PLANES) && !pv.vec_epi) {
        // only the float4 epilogue writes planes: the scalar one would store fp32 into the bf16 image
        lvt_set_error("gemm: LVT_EPI_PLANES needs 16-byte aligned bias / res / mask and ldr, ldm %% 4 == 0");
        return LVT_EINVAL;
    }
    hipLaunchKernelGGL((lvt_gemm_wide_kernel<TA, TB>), grid, dim3(WIDE_THREADS), 0, s, pv);
    LVT_CHECK_LAUNCH("lvt_gemm_wide_kernel");
    return LVT_OK;
}

static int choose_splits(long long tiles, int K, int min_k) {
    // Two workgroups are resident per CU (176 registers/lane), i.e. 512 slots on the chip: aim for the
    // largest split count whose grid still fits in ONE full wave of workgroups (tiles * splits <= 512), so that no
    // partially filled tail wave is paid; never less than min_k of reduction per split.  (Two waves, 1024, are no
    // faster on the conv weight gradients, 12-14 % slower on the 16384-row linear ones, and double the
    // partial-sum traffic; 1.5 waves is 20 % slower.)
    int s = (int)((2 * LVT_NUM_CU) / (tiles > 0 ? tiles : 1));
    const int maxs = K / min_k > 0 ? K / min_k : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return s;
}

static void kparams_from_desc(const lvt_gemm_desc *d, KParams &p) {
    memset(&p, 0, sizeof(p));
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.A = d->A; p.lda = d->lda; p.a_kb = d->a_kb > 0 ? d->a_kb : d->K; p.a_skb = d->a_skb;
    p.B = d->B; p.ldb = d->ldb; p.b_kb = d->b_kb > 0 ? d->b_kb : d->K; p.b_skb = d->b_skb;
    p.C = d->C; p.ldc = d->ldc; p.c_plane = d->c_plane;
    p.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    p.sA_o = d->sA_o; p.sA_i = d->sA_i; p.sB_o = d->sB_o; p.sB_i = d->sB_i; p.sC_o = d->sC_o; p.sC_i = d->sC_i;
    p.alpha = d->alpha; p.flags = d->flags; p.bias = d->bias; p.res = d->res; p.ldr = d->ldr;
    p.mask = d->mask; p.ldm = d->ldm;
    p.splits = d->splits > 1 ? d->splits : 1;
    p.a_amax = d->a_amax; p.b_amax = d->b_amax; p.c_amax = d->c_amax;
    p.a_amax2 = d->a_amax2; p.b_amax2 = d->b_amax2;
}

static int gemm_batch(const lvt_gemm_desc *d) {
    return (d->batch_outer > 0 ? d->batch_outer : 1) * (d->batch_inner > 0 ? d->batch_inner : 1);
}

extern "C" size_t lvt_gemm_workspace_bytes(const lvt_gemm_desc *d) {
    if (!d || d->splits <= 1) return 0;
    return (size_t)d->splits * gemm_batch(d) * ((size_t)d->M * d->N + (d->a_colsum ? (size_t)d->M : 0)) * sizeof(float);
}

extern "C" int lvt_gemm_f32(const lvt_gemm_desc *d, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(d && d->A && d->B && d->C, "gemm: null pointer");
    LVT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape %d %d %d", d->M, d->N, d->K);
    LVT_REQUIRE(d->K % 4 == 0, "gemm: K=%d must be a multiple of 4", d->K);
    LVT_REQUIRE(d->lda % 4 == 0 && d->ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4");
    LVT_REQUIRE(lvt_aligned16(d->A) && lvt_aligned16(d->B), "gemm: A/B must be 16-byte aligned");
    LVT_REQUIRE(d->ta == 0 || d->M % 4 == 0, "gemm: ta=1 needs M %% 4 == 0");
    LVT_REQUIRE(d->tb == 0 || d->N % 4 == 0, "gemm: tb=1 needs N %% 4 == 0");
    LVT_REQUIRE(d->a_kb <= 0 || (d->a_kb % 4 == 0 && d->a_skb % 4 == 0), "gemm: a_kb/a_skb %% 4");
    LVT_REQUIRE(d->b_kb <= 0 || (d->b_kb % 4 == 0 && d->b_skb % 4 == 0), "gemm: b_kb/b_skb %% 4");
    LVT_REQUIRE(!(d->ta == 1 && d->tb == 0), "gemm: (ta=1, tb=0) is not instantiated");
    LVT_REQUIRE(!(d->flags & LVT_EPI_BIAS) || d->bias, "gemm: BIAS flag without bias");
    LVT_REQUIRE(!(d->flags & LVT_EPI_RESIDUAL) || d->res, "gemm: RESIDUAL flag without res");
    LVT_REQUIRE(!(d->flags & LVT_EPI_MASK) || d->mask, "gemm: MASK flag without mask");
    LVT_REQUIRE(math_of(d->flags) != 2 || (d->a_amax && d->b_amax), "gemm: LVT_MATH_F16X2 needs a_amax and b_amax");
    LVT_REQUIRE(!d->c_amax || d->splits <= 1, "gemm: c_amax is not produced by split-K launches");
    if (d->flags & LVT_EPI_PLANES)
        LVT_REQUIRE(d->splits <= 1 && !(d->flags & LVT_EPI_ACCUM) && d->N % 4 == 0 && d->ldc % 4 == 0 && d->c_plane % 4 == 0 &&
                    d->sC_o % 4 == 0 && d->sC_i % 4 == 0 && lvt_aligned16(d->C) && !(d->flags & LVT_MATH_F32),
                    "gemm: PLANES needs the bf16x3 arithmetic, no split-K / ACCUM, N, ldc, c_plane, sC %% 4 == 0 and an aligned C");
    hipStream_t s = (hipStream_t)stream;
    KParams p; kparams_from_desc(d, p);
    const int zc = gemm_batch(d);
    if (p.splits > 1) {
        LVT_REQUIRE((d->flags & ~(LVT_EPI_ACCUM | LVT_MATH_F32 | LVT_MATH_F16X2)) == 0 && d->alpha == 1.0f, "gemm: split-K allows only ACCUM");
        LVT_REQUIRE(d->ldc == d->N && (zc == 1 || (d->sC_i == (long long)d->M * d->N)),
                    "gemm: split-K needs a dense C (ldc == N)");
        LVT_REQUIRE(((long long)d->M * d->N) % 4 == 0 && lvt_aligned16(d->C), "gemm: split-K C alignment");
        const size_t need = lvt_gemm_workspace_bytes(d);
        if (workspace_bytes < need || !workspace) {
            lvt_set_error("gemm: workspace %zu < %zu", workspace_bytes, need);
            return LVT_EWORKSPACE;
        }
        p.k_per_split = (int)(lvt_cdiv(lvt_cdiv(p.K, p.splits), BK) * BK);
        p.partial = (float *)workspace;
        p.partial_stride = (long long)zc * d->M * d->N;
        if (d->a_colsum) {
            LVT_REQUIRE(d->ta == 1 && d->M % 4 == 0, "gemm: a_colsum needs ta == 1 and M %% 4 == 0");
            p.colsum_partial = p.partial + (long long)p.splits * p.partial_stride;
        }
    } else {
        LVT_REQUIRE(!d->a_colsum, "gemm: a_colsum needs splits > 1");
    }
    int rc;
    static const int no_wide = getenv("LVT_NO_WIDE_GEMM") ? 1 : 0;
    // (the wide kernel addresses its operands with 32-bit byte offsets from a per-batch base: each must span < 4 GB)
    const long long a_span = d->ta ? (long long)d->K * d->lda : (long long)d->M * d->lda + (long long)(d->K / p.a_kb) * d->a_skb;
    const long long b_span = d->tb ? (long long)d->K * d->ldb : (long long)d->N * d->ldb + (long long)(d->K / p.b_kb) * d->b_skb;
    const bool wide = !no_wide && math_of(d->flags) == 2 && BK == 32 && d->M > 128 && d->K % BK == 0 && p.a_kb % BK == 0 &&
                      p.b_kb % BK == 0 && a_span < (1LL << 30) && b_span < (1LL << 30) &&
                      !(d->flags & (LVT_CAUSAL_KMAX | LVT_CAUSAL_KMIN | LVT_CAUSAL_TILE));
    if (wide && d->ta == 0 && d->tb == 0) rc = launch_wide<0, 0>(p, zc, s);
    else if (wide && d->ta == 0 && d->tb == 1) rc = launch_wide<0, 1>(p, zc, s);
    else if (wide) rc = launch_wide<1, 1>(p, zc, s);
    else if (d->ta == 0 && d->tb == 0) rc = launch_tile<A_KPLAIN, B_KPLAIN, 128, 128, 2, 2>(p, zc, s);
    else if (d->ta == 0 && d->tb == 1) rc = launch_tile<A_KPLAIN, B_NPLAIN, 128, 128, 2, 2>(p, zc, s);
    else rc = launch_tile<A_MPLAIN, B_NPLAIN, 128, 128, 2, 2>(p, zc, s);
    if (rc != LVT_OK) return rc;
    if (p.splits > 1) {
        const long long n4 = p.partial_stride / 4;
        const int blocks = (int)(lvt_cdiv(n4, 256) < 2048 ? lvt_cdiv(n4, 256) : 2048);
        // batches are laid out z-major in the partial buffer exactly like a dense C
        hipLaunchKernelGGL(lvt_reduce_splits_kernel, dim3(blocks), dim3(256), 0, s, p.partial, n4,
                           p.partial_stride, p.splits, d->C, (d->flags & LVT_EPI_ACCUM) ? 1 : 0);
        LVT_CHECK_LAUNCH("lvt_reduce_splits_kernel");
        if (p.colsum_partial) {
            // [split][batch][M] partials -> a_colsum (batch, M)
            hipLaunchKernelGGL(lvt_reduce_splits_kernel, dim3((unsigned)lvt_cdiv((long long)zc * d->M / 4, 256)), dim3(256), 0, s,
                               (const float *)p.colsum_partial, (long long)zc * d->M / 4, (long long)zc * d->M, p.splits,
                               d->a_colsum, 0);
            LVT_CHECK_LAUNCH("lvt_reduce_splits_kernel");
        }
    }
    return LVT_OK;
}

// ---- convolution entry points -------------------------------------------------------------------
// all weight packs of a convolution stack in ONE launch (blockIdx.y = entry): a VQ-VAE pass makes 24 of them, each a ~5 us launch
// in front of the layer that needs it; kind 0: lvt_pack_weight_kernel, 1: _t, 2: _phases, 3: _parity (same element functions)
struct PackTable { const float *w[64]; float *dst[64]; int kind[64], taps[64], Ci[64], Co[64], Ci_real[64], Co_real[64]; };
__global__ void lvt_pack_weights_multi_kernel(const PackTable t) {
    const int e = blockIdx.y;
    const float *__restrict__ w = t.w[e];
    float *__restrict__ dst = t.dst[e];
    const int kind = t.kind[e], taps = t.taps[e], Ci = t.Ci[e], Co = t.Co[e], Ci_real = t.Ci_real[e], Co_real = t.Co_real[e];
    const long long total = (long long)(kind >= 2 ? 16 : taps) * Ci * Co;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int co, ci, tap;
        if (kind == 0 || kind == 3) { co = i % Co; const long long q = i / Co; ci = q % Ci; tap = (int)(q / Ci); }      // [..][ci][co]
        else { ci = i % Ci; const long long q = i / Ci; co = q % Co; tap = (int)(q / Co); }                                  // [..][co][ci]
        int src_tap = tap;                                                      // index into the layer's own tap order
        if (kind == 1) src_tap = taps - 1 - tap;
        else if (kind == 2) { const int ph = tap >> 2, tp = tap & 3; src_tap = (3 - (ph >> 1) - 2 * (tp >> 1)) * 4 + 3 - (ph & 1) - 2 * (tp & 1); }
        else if (kind == 3) { const int cls = tap >> 2, tp = tap & 3; src_tap = (2 * (tp >> 1) + (cls >> 1)) * 4 + 2 * (tp & 1) + (cls & 1); }
        float v = 0.f;
        if (co < Co_real && ci < Ci_real) v = w[((long long)co * Ci_real + ci) * (kind >= 2 ? 16 : taps) + src_tap];
        dst[i] = v;
    }
}
extern "C" int lvt_conv3d_pack_weights_multi(const lvt_pack_entry *entries, int n, void *stream) {
    LVT_REQUIRE(entries && n > 0, "pack_weights_multi: bad args");
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        PackTable t;
        long long biggest = 0;
        for (int i = 0; i < cnt; ++i) {
            const lvt_pack_entry &e = entries[base + i];
            LVT_REQUIRE(e.w && e.dst && e.kind >= 0 && e.kind <= 3 && e.taps > 0 && e.Ci > 0 && e.Co > 0 && e.Ci_real <= e.Ci &&
                            e.Co_real <= e.Co && (e.kind < 2 || e.taps == 16), "pack_weights_multi: bad entry %d", base + i);
            t.w[i] = e.w; t.dst[i] = e.dst; t.kind[i] = e.kind; t.taps[i] = e.taps; t.Ci[i] = e.Ci; t.Co[i] = e.Co;
            t.Ci_real[i] = e.Ci_real; t.Co_real[i] = e.Co_real;
            const long long total = (long long)e.taps * e.Ci * e.Co;
            if (total > biggest) biggest = total;
        }
        const unsigned bx = (unsigned)(lvt_cdiv(biggest, 256) < 1024 ? lvt_cdiv(biggest, 256) : 1024);
        hipLaunchKernelGGL(lvt_pack_weights_multi_kernel, dim3(bx, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, t);
        LVT_CHECK_LAUNCH("lvt_pack_weights_multi_kernel");
    }
    return LVT_OK;
}

// ---- weight tiles as ready LDS images (round 6; lvt_conv_patch_kernel<*, 2, BIMG = 1>) ----------------------------------
// A packed weight wp[rows][cols] (rows % 32 == 0, cols % 128 == 0; any kind of the packs above) -> for every 32 x 128 tile (kt, nt)
// the 20992 bytes the kernel keeps in LDS for it: plane hi, plane lo of the f16x2 split under the weight's own scale, rows placed by
// hrow<128>, pads included (never read).  Same split function as the in-kernel path (store_split2_k), so the bytes in LDS are equal.
struct WImgTable { const float *wp[64]; char *img[64]; const float *amax[64]; int rows[64], cols[64]; };
__global__ __launch_bounds__(256) void lvt_weight_images_kernel(const WImgTable t) {
    const int e = blockIdx.y;
    const float *__restrict__ wp = t.wp[e];
    const int rows = t.rows[e], cols = t.cols[e];
    const int ntn = cols / 128, ntiles = (rows / 32) * ntn;
    int unscale = 0;
    const float sb = lvt_f16_scale(t.amax[e], unscale);
    const int n = threadIdx.x & 127, q0 = threadIdx.x >> 7;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int kt = tile / ntn, nt = tile - kt * ntn;
        unsigned short *img = reinterpret_cast<unsigned short *>(t.img[e] + (long long)tile * PT_BIMG_BYTES);
        const float *src = wp + (long long)(kt * 32) * cols + nt * 128 + n;
#pragma unroll
        for (int kq = q0; kq < 8; kq += 2) {
            const float4 v = make_float4(src[(long long)(4 * kq) * cols], src[(long long)(4 * kq + 1) * cols],
                                         src[(long long)(4 * kq + 2) * cols], src[(long long)(4 * kq + 3) * cols]);
            store_split2_k<128>(img, n, 4 * kq, v, sb);
        }
    }
}
extern "C" size_t lvt_conv3d_weight_image_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0 || rows % 32 || cols % 128) return 0;
    return (size_t)(rows / 32) * (cols / 128) * PT_BIMG_BYTES;
}
extern "C" int lvt_conv3d_weight_images(const lvt_weight_image_entry *entries, int n, void *stream) {
    LVT_REQUIRE(entries && n > 0, "weight_images: bad args");
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        WImgTable t;
        int most = 0;
        for (int i = 0; i < cnt; ++i) {
            const lvt_weight_image_entry &e = entries[base + i];
            LVT_REQUIRE(e.wp && e.amax && e.rows > 0 && e.cols > 0 && e.rows % 32 == 0 && e.cols % 128 == 0,
                        "weight_images: entry %d: rows %% 32 / cols %% 128 / pointers", base + i);
            // the image sits right behind the packed weight (lvt_conv3d_weight_image_bytes more bytes in the same buffer)
            t.wp[i] = e.wp; t.img[i] = (char *)(e.wp + (size_t)e.rows * e.cols); t.amax[i] = e.amax; t.rows[i] = e.rows; t.cols[i] = e.cols;
            LVT_REQUIRE(lvt_aligned16(t.img[i]), "weight_images: entry %d: the packed weight must be 16-byte aligned", base + i);
            const int tiles = (e.rows / 32) * (e.cols / 128);
            if (tiles > most) most = tiles;
        }
        hipLaunchKernelGGL(lvt_weight_images_kernel, dim3((unsigned)(most < 256 ? most : 256), (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, t);
        LVT_CHECK_LAUNCH("lvt_weight_images_kernel");
    }
    return LVT_OK;
}

static int check_geom(const lvt_conv_geom *g, const char *who) {
    LVT_REQUIRE(g, "%s: null geometry", who);
    LVT_REQUIRE(g->N > 0 && g->Ti > 0 && g->Hi > 0 && g->Wi > 0 && g->To > 0 && g->Ho > 0 && g->Wo > 0,
                "%s: bad extents", who);
    LVT_REQUIRE(g->Ci > 0 && g->Co > 0 && g->Ci % 4 == 0 && g->Co % 4 == 0, "%s: Ci=%d/Co=%d must be multiples of 4",
                who, g->Ci, g->Co);
    LVT_REQUIRE(g->Kt > 0 && g->Kh > 0 && g->Kw > 0 && g->st > 0 && g->sh > 0 && g->sw > 0, "%s: bad kernel/stride", who);
    return LVT_OK;
}

extern "C" int lvt_conv3d_pack_weight(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                      float *wp, void *stream) {
    int rc = check_geom(g, "pack_weight"); if (rc) return rc;
    LVT_REQUIRE(w && wp && Ci_real <= g->Ci && Co_real <= g->Co, "pack_weight: bad args");
    const int taps = g->Kt * g->Kh * g->Kw;
    const long long total = (long long)taps * g->Ci * g->Co;
    const int blocks = (int)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(lvt_pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, taps,
                       g->Ci, g->Co, Ci_real, Co_real);
    LVT_CHECK_LAUNCH("lvt_pack_weight_kernel");
    return LVT_OK;
}

extern "C" int lvt_conv3d_pack_weight_t(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                        float *wt, void *stream) {
    int rc = check_geom(g, "pack_weight_t"); if (rc) return rc;
    LVT_REQUIRE(w && wt && Ci_real <= g->Ci && Co_real <= g->Co, "pack_weight_t: bad args");
    LVT_REQUIRE(g->st == 1 && g->sh == 1 && g->sw == 1, "pack_weight_t: stride-1 convolutions only");
    const int taps = g->Kt * g->Kh * g->Kw;
    const long long total = (long long)taps * g->Ci * g->Co;
    const int blocks = (int)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(lvt_pack_weight_t_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, taps,
                       g->Ci, g->Co, Ci_real, Co_real);
    LVT_CHECK_LAUNCH("lvt_pack_weight_t_kernel");
    return LVT_OK;
}

// the frame-resident kernel serves 3x3 / stride 1 / pad 1 convolutions of 16x16 frames with Ci % 32 == 0, Co % 128 == 0
static bool patch_conv_eligible(const lvt_conv_geom *g, int flags) {
    static const int off = getenv("LVT_NO_PATCH_CONV") ? 1 : 0;
    return !off && math_of(flags) >= 1 && BK == 32 && g->Kt == 1 && g->Kh == 3 && g->Kw == 3 && g->st == 1 && g->sh == 1 &&
           g->sw == 1 && g->pt == 0 && g->ph == 1 && g->pw == 1 && g->Ti == 1 && g->Hi == 16 && g->Wi == 16 && g->To == 1 &&
           g->Ho == 16 && g->Wo == 16 && g->Ci % 32 == 0 && g->Co % 128 == 0;
}
extern "C" int lvt_conv3d_uses_patch_kernel(const lvt_conv_geom *g, int flags) { return g && patch_conv_eligible(g, flags) ? 1 : 0; }

// 4x4 / stride 2 / pad 1 convolution 32x32 -> 16x16 on the frame-resident kernel (parity classes)
static bool conv2x_eligible(const lvt_conv_geom *g, int flags) {
    static const int off = (getenv("LVT_NO_PATCH_CONV") || getenv("LVT_NO_PARITY_CONV")) ? 1 : 0;
    return !off && math_of(flags) >= 1 && BK == 32 && g->Kt == 1 && g->Kh == 4 && g->Kw == 4 && g->st == 1 && g->sh == 2 &&
           g->sw == 2 && g->pt == 0 && g->ph == 1 && g->pw == 1 && g->Ti == 1 && g->Hi == 32 && g->Wi == 32 && g->To == 1 &&
           g->Ho == 16 && g->Wo == 16 && g->Ci % 32 == 0 && g->Co % 128 == 0;
}
extern "C" int lvt_conv3d_fwd_uses_parity_kernel(const lvt_conv_geom *g, int flags) { return g && conv2x_eligible(g, flags) ? 1 : 0; }
extern "C" int lvt_conv3d_pack_weight_parity(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                             float *wq, void *stream) {
    int rc = check_geom(g, "pack_weight_parity"); if (rc) return rc;
    LVT_REQUIRE(w && wq && Ci_real <= g->Ci && Co_real <= g->Co && g->Kt == 1 && g->Kh == 4 && g->Kw == 4,
                "pack_weight_parity: 4x4 kernels only");
    const long long total = 16LL * g->Ci * g->Co;
    hipLaunchKernelGGL(lvt_pack_weight_parity_kernel, dim3((unsigned)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096)),
                       dim3(256), 0, (hipStream_t)stream, w, wq, g->Ci, g->Co, Ci_real, Co_real);
    LVT_CHECK_LAUNCH("lvt_pack_weight_parity_kernel");
    return LVT_OK;
}
extern "C" int lvt_conv3d_fwd_parity(const lvt_conv_geom *g, const float *x, const float *wq, const float *bias,
                                     const float *res, const float *mask, float *y, int flags, const lvt_amax_io *ax,
                                     void *stream) {
    int rc = check_geom(g, "conv3d_fwd_parity"); if (rc) return rc;
    const bool wimg = (flags & LVT_CONV_WEIGHT_IMAGE) != 0;
    flags &= ~LVT_CONV_WEIGHT_IMAGE;
    LVT_REQUIRE(x && wq && y, "conv3d_fwd_parity: null pointer");
    LVT_REQUIRE(conv2x_eligible(g, flags), "conv3d_fwd_parity: geometry not served (see lvt_conv3d_fwd_uses_parity_kernel)");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "conv3d_fwd_parity: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "conv3d_fwd_parity: RESIDUAL without res");
    LVT_REQUIRE(!(flags & LVT_EPI_MASK) || mask, "conv3d_fwd_parity: MASK without mask");
    LVT_REQUIRE(!(flags & LVT_EPI_ACCUM), "conv3d_fwd_parity: unsupported flag");
    LVT_REQUIRE(lvt_aligned16(x) && lvt_aligned16(wq) && lvt_aligned16(y) && lvt_aligned16(bias) && lvt_aligned16(res) &&
                lvt_aligned16(mask), "conv3d_fwd_parity: alignment");
    KParams p; memset(&p, 0, sizeof(p));
    p.M = g->N * 256; p.N = g->Co; p.K = 16 * g->Ci;
    p.A = x; p.B = wq; p.ldb = g->Co; p.C = y; p.ldc = g->Co; p.batch_inner = 1;
    p.alpha = 1.f; p.flags = flags; p.bias = bias; p.res = res; p.ldr = g->Co; p.mask = mask; p.ldm = g->Co;
    p.splits = 1; p.vec_epi = 1; p.g = *g;
    LVT_REQUIRE_AMAX(flags, ax, "conv3d_fwd_parity"); set_amax(p, ax);
    if (math_of(flags) == 2 && wimg) {
        p.b_img = (const char *)(wq + (size_t)p.K * p.N);
        hipLaunchKernelGGL((lvt_conv_patch_kernel<2, 2, 1>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    } else if (math_of(flags) == 2)
        hipLaunchKernelGGL((lvt_conv_patch_kernel<2, 2>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((lvt_conv_patch_kernel<2, 1>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    LVT_CHECK_LAUNCH("lvt_conv_patch_kernel<2>");
    return LVT_OK;
}

// image-side 4 -> 128 channel k4 s2 convolution of the f16x2 arithmetic (thin_conv.hip)
bool lvt_conv4s2_img_ok(const lvt_conv_geom *g, int flags);
int lvt_conv4s2_img_launch(const lvt_conv_geom *g, const float *x, const float *wp, const float *bias, const float *res,
                           const float *mask, float *y, int flags, const float *x_amax, const float *w_amax, float *y_amax,
                           hipStream_t s);
extern "C" int lvt_conv3d_fwd(const lvt_conv_geom *g, const float *x, const float *wp, const float *bias,
                              const float *res, const float *mask, float *y, int flags, const lvt_amax_io *ax, void *stream) {
    int rc = check_geom(g, "conv3d_fwd"); if (rc) return rc;
    const bool wimg = (flags & LVT_CONV_WEIGHT_IMAGE) != 0;       // (only the frame-resident f16x2 launch reads the image)
    flags &= ~LVT_CONV_WEIGHT_IMAGE;
    LVT_REQUIRE(x && wp && y, "conv3d_fwd: null pointer");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "conv3d_fwd: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "conv3d_fwd: RESIDUAL without res");
    LVT_REQUIRE(!(flags & LVT_EPI_MASK) || mask, "conv3d_fwd: MASK without mask");
    LVT_REQUIRE(!(flags & LVT_EPI_ACCUM), "conv3d_fwd: unsupported flag");
    const long long M = (long long)g->N * g->To * g->Ho * g->Wo;
    LVT_REQUIRE(M < 0x7fffffffLL, "conv3d_fwd: too many output positions");
    KParams p; memset(&p, 0, sizeof(p));
    p.M = (int)M; p.N = g->Co; p.K = g->Kt * g->Kh * g->Kw * g->Ci;
    p.A = x; p.B = wp; p.ldb = g->Co; p.C = y; p.ldc = g->Co; p.batch_inner = 1;
    p.alpha = 1.f; p.flags = flags; p.bias = bias; p.res = res; p.ldr = g->Co; p.mask = mask; p.ldm = g->Co;
    p.splits = 1; p.g = *g;
    LVT_REQUIRE_AMAX(flags, ax, "conv3d_fwd"); set_amax(p, ax);
    if (lvt_conv4s2_img_ok(g, flags) && lvt_aligned16(x) && lvt_aligned16(wp) && lvt_aligned16(y) &&
        (!(flags & LVT_EPI_BIAS) || lvt_aligned16(bias)) && (!(flags & LVT_EPI_RESIDUAL) || lvt_aligned16(res)) &&
        (!(flags & LVT_EPI_MASK) || lvt_aligned16(mask)) && M * g->Co < (1LL << 40))      // (an offset view of any of them: the tile kernel)
        return lvt_conv4s2_img_launch(g, x, wp, bias, res, mask, y, flags, ax->a, ax->b, ax->c, (hipStream_t)stream);
    {
        auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
        bool ok = patch_conv_eligible(g, flags) && al16(x) && al16(wp) && al16(y);
        if (flags & LVT_EPI_BIAS) ok = ok && al16(bias);
        if (flags & LVT_EPI_RESIDUAL) ok = ok && al16(res);
        if (flags & LVT_EPI_MASK) ok = ok && al16(mask);
        if (ok) {
            p.vec_epi = 1;
            if (math_of(flags) == 2 && wimg) {
                p.b_img = (const char *)(wp + (size_t)p.K * p.N);
                hipLaunchKernelGGL((lvt_conv_patch_kernel<0, 2, 1>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                                   (hipStream_t)stream, p);
            } else if (math_of(flags) == 2)
                hipLaunchKernelGGL((lvt_conv_patch_kernel<0, 2>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                                   (hipStream_t)stream, p);
            else
                hipLaunchKernelGGL((lvt_conv_patch_kernel<0, 1>), dim3((unsigned)(g->N * (g->Co / 128))), dim3(PT_THREADS), 0,
                                   (hipStream_t)stream, p);
            LVT_CHECK_LAUNCH("lvt_conv_patch_kernel");
            return LVT_OK;
        }
    }
    {
        // a 1x1x1 / stride 1 / unpadded convolution IS the product x (M x Ci) . wp (Ci x Co): in f16x2 mode it takes the wide
        // pipelined kernel (no im2col index arithmetic in the loader, 256-row tiles)
        static const int no_wide = getenv("LVT_NO_WIDE_GEMM") ? 1 : 0;
        if (!no_wide && math_of(flags) == 2 && BK == 32 && g->Kt == 1 && g->Kh == 1 && g->Kw == 1 && g->st == 1 && g->sh == 1 &&
            g->sw == 1 && g->pt == 0 && g->ph == 0 && g->pw == 0 && g->Ci % BK == 0 && p.M > 128 && lvt_aligned16(x) && lvt_aligned16(wp) &&
            (long long)p.M * g->Ci < (1LL << 30)) {
            p.lda = g->Ci; p.a_kb = p.K; p.b_kb = p.K;
            return launch_wide<0, 1>(p, 1, (hipStream_t)stream);
        }
    }
    if (g->Co <= 32) return launch_tile<A_CONV_K, B_NPLAIN, 128, 32, 4, 1>(p, 1, (hipStream_t)stream);
#ifdef LVT_CX_SMALLK64
    if (p.K <= 64) return launch_tile<A_CONV_K, B_NPLAIN, 64, 128, 2, 2>(p, 1, (hipStream_t)stream);
#endif
    return launch_tile<A_CONV_K, B_NPLAIN, 128, 128, 2, 2>(p, 1, (hipStream_t)stream);
}

// ---- stride-2 transposed convolution of 16x16 frames on the frame-resident kernel -------------------------------------
static bool convt2x_eligible(const lvt_conv_geom *g, int flags) {
    static const int off = (getenv("LVT_NO_PATCH_CONV") || getenv("LVT_NO_PHASE_CONV")) ? 1 : 0;
    // g is the geometry of the FORWARD strided convolution (Ci -> Co, 32x32 -> 16x16); the transposed pass maps Co -> Ci
    return !off && math_of(flags) >= 1 && BK == 32 && g->Kt == 1 && g->Kh == 4 && g->Kw == 4 && g->st == 1 && g->sh == 2 &&
           g->sw == 2 && g->pt == 0 && g->ph == 1 && g->pw == 1 && g->Ti == 1 && g->Hi == 32 && g->Wi == 32 && g->To == 1 &&
           g->Ho == 16 && g->Wo == 16 && g->Co % 32 == 0 && g->Ci % 128 == 0;
}
extern "C" int lvt_conv3d_bwd_data_uses_phase_kernel(const lvt_conv_geom *g, int flags) { return g && convt2x_eligible(g, flags) ? 1 : 0; }
extern "C" int lvt_conv3d_pack_weight_phases(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                             float *wph, void *stream) {
    int rc = check_geom(g, "pack_weight_phases"); if (rc) return rc;
    LVT_REQUIRE(w && wph && Ci_real <= g->Ci && Co_real <= g->Co && g->Kt == 1 && g->Kh == 4 && g->Kw == 4,
                "pack_weight_phases: 4x4 kernels only");
    const long long total = 16LL * g->Ci * g->Co;
    hipLaunchKernelGGL(lvt_pack_weight_phases_kernel, dim3((unsigned)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096)),
                       dim3(256), 0, (hipStream_t)stream, w, wph, g->Ci, g->Co, Ci_real, Co_real);
    LVT_CHECK_LAUNCH("lvt_pack_weight_phases_kernel");
    return LVT_OK;
}
extern "C" int lvt_conv3d_bwd_data_phases(const lvt_conv_geom *g, const float *dy, const float *wph, const float *bias,
                                          const float *res, const float *mask, float *dx, int flags, const lvt_amax_io *ax,
                                          void *stream) {
    int rc = check_geom(g, "conv3d_bwd_data_phases"); if (rc) return rc;
    const bool wimg = (flags & LVT_CONV_WEIGHT_IMAGE) != 0;
    flags &= ~LVT_CONV_WEIGHT_IMAGE;
    LVT_REQUIRE(dy && wph && dx, "conv3d_bwd_data_phases: null pointer");
    LVT_REQUIRE(convt2x_eligible(g, flags), "conv3d_bwd_data_phases: geometry not served (see lvt_conv3d_bwd_data_uses_phase_kernel)");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "conv3d_bwd_data_phases: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "conv3d_bwd_data_phases: RESIDUAL without res");
    LVT_REQUIRE(!(flags & LVT_EPI_MASK) || mask, "conv3d_bwd_data_phases: MASK without mask");
    LVT_REQUIRE(!(flags & LVT_EPI_ACCUM), "conv3d_bwd_data_phases: unsupported flag");
    LVT_REQUIRE(lvt_aligned16(dy) && lvt_aligned16(wph) && lvt_aligned16(dx) && lvt_aligned16(bias) && lvt_aligned16(res) &&
                lvt_aligned16(mask), "conv3d_bwd_data_phases: alignment");
    KParams p; memset(&p, 0, sizeof(p));
    p.M = g->N * 256; p.N = g->Ci; p.K = 4 * g->Co;            // per phase
    p.A = dy; p.B = wph; p.ldb = g->Ci; p.C = dx; p.ldc = g->Ci; p.batch_inner = 1;
    p.alpha = 1.f; p.flags = flags; p.bias = bias; p.res = res; p.ldr = g->Ci; p.mask = mask; p.ldm = g->Ci;
    p.splits = 1; p.vec_epi = 1;
    p.g = *g; p.g.Ci = g->Co;                                   // the kernel's input channel count
    LVT_REQUIRE_AMAX(flags, ax, "conv3d_bwd_data_phases"); set_amax(p, ax);
    if (math_of(flags) == 2 && wimg) {
        p.b_img = (const char *)(wph + (size_t)16 * g->Co * g->Ci);
        hipLaunchKernelGGL((lvt_conv_patch_kernel<1, 2, 1>), dim3((unsigned)(g->N * 4 * (g->Ci / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    } else if (math_of(flags) == 2)
        hipLaunchKernelGGL((lvt_conv_patch_kernel<1, 2>), dim3((unsigned)(g->N * 4 * (g->Ci / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((lvt_conv_patch_kernel<1, 1>), dim3((unsigned)(g->N * 4 * (g->Ci / 128))), dim3(PT_THREADS), 0,
                           (hipStream_t)stream, p);
    LVT_CHECK_LAUNCH("lvt_conv_patch_kernel<1>");
    return LVT_OK;
}

extern "C" int lvt_conv3d_bwd_data(const lvt_conv_geom *g, const float *dy, const float *wp, const float *bias,
                                   const float *res, const float *mask, float *dx, int flags, const lvt_amax_io *ax,
                                   void *stream) {
    int rc = check_geom(g, "conv3d_bwd_data"); if (rc) return rc;
    LVT_REQUIRE(dy && wp && dx, "conv3d_bwd_data: null pointer");
    LVT_REQUIRE(g->Kt % g->st == 0 && g->Kh % g->sh == 0 && g->Kw % g->sw == 0,
                "conv3d_bwd_data: kernel extents must be multiples of the strides");
    LVT_REQUIRE(g->Ti % g->st == 0 && g->Hi % g->sh == 0 && g->Wi % g->sw == 0,
                "conv3d_bwd_data: input extents must be multiples of the strides");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "conv3d_bwd_data: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "conv3d_bwd_data: RESIDUAL without res");
    LVT_REQUIRE(!(flags & LVT_EPI_MASK) || mask, "conv3d_bwd_data: MASK without mask");
    LVT_REQUIRE(!(flags & LVT_EPI_ACCUM), "conv3d_bwd_data: unsupported flag");
    KParams p; memset(&p, 0, sizeof(p));
    p.Tq = g->Ti / g->st; p.Hq = g->Hi / g->sh; p.Wq = g->Wi / g->sw;
    p.jT = g->Kt / g->st; p.jH = g->Kh / g->sh; p.jW = g->Kw / g->sw;
    const long long M = (long long)g->N * p.Tq * p.Hq * p.Wq;
    LVT_REQUIRE(M < 0x7fffffffLL, "conv3d_bwd_data: too many positions");
    p.M = (int)M; p.N = g->Ci; p.K = p.jT * p.jH * p.jW * g->Co;
    p.A = dy; p.B = wp; p.C = dx; p.ldc = g->Ci; p.batch_inner = 1;
    p.alpha = 1.f; p.flags = flags; p.bias = bias; p.res = res; p.ldr = g->Ci; p.mask = mask; p.ldm = g->Ci;
    p.splits = 1; p.g = *g;
    LVT_REQUIRE_AMAX(flags, ax, "conv3d_bwd_data"); set_amax(p, ax);
    {
        // 1x1x1 / stride 1 / unpadded: dx (M x Ci) = dy (M x Co) . wp^T, wp = (Ci x Co) read k-contiguous -- the wide kernel
        static const int no_wide = getenv("LVT_NO_WIDE_GEMM") ? 1 : 0;
        if (!no_wide && math_of(flags) == 2 && BK == 32 && g->Kt == 1 && g->Kh == 1 && g->Kw == 1 && g->st == 1 && g->sh == 1 &&
            g->sw == 1 && g->pt == 0 && g->ph == 0 && g->pw == 0 && g->Co % BK == 0 && p.M > 128 && lvt_aligned16(dy) && lvt_aligned16(wp) &&
            (long long)p.M * g->Co < (1LL << 30)) {
            p.lda = g->Co; p.ldb = g->Co; p.a_kb = p.K; p.b_kb = p.K;
            return launch_wide<0, 0>(p, 1, (hipStream_t)stream);
        }
    }
    const int ncls = g->st * g->sh * g->sw;
    if (g->Ci <= 32) return launch_tile<A_CONVT_K, B_CONVT_W, 128, 32, 4, 1>(p, ncls, (hipStream_t)stream);
    return launch_tile<A_CONVT_K, B_CONVT_W, 128, 128, 2, 2>(p, ncls, (hipStream_t)stream);
}

// frame-resident weight gradient of the 3x3 layers (conv_wgrad.hip)
int lvt_wgrad_frames_role(const lvt_conv_geom *g, int flags);
size_t lvt_wgrad_frames_workspace_bytes(const lvt_conv_geom *g);
int lvt_wgrad_frames_launch(const lvt_conv_geom *g, const float *x, const float *dy, float *dw, int Ci_real, int Co_real,
                            void *workspace, hipStream_t s, void (*unpack_plain)(const float *, long long, int, float *,
                                                                                 const lvt_conv_geom *, int, int, hipStream_t),
                            const float *x_amax, const float *dy_amax, float *db, int db_of_x);
// the tiled unpack kernel turns a [64 co][4 ci x taps] tile through dynamic LDS: served while that tile fits in 64 KB
static bool unpack_tiled_ok(const lvt_conv_geom *g, int Ci_real, int Co_real) {
    const int taps = g->Kt * g->Kh * g->Kw;
    return g->Co % UW_CO == 0 && g->Ci % UW_CI == 0 && Ci_real == g->Ci && Co_real == g->Co &&
           (size_t)UW_CO * (UW_CI * taps + 1) * sizeof(float) <= 64 * 1024;
}
static void unpack_plain_wgrad(const float *partial, long long stride, int splits, float *dw, const lvt_conv_geom *g,
                               int Ci_real, int Co_real, hipStream_t s) {
    const int taps = g->Kt * g->Kh * g->Kw;
    const long long total = (long long)taps * g->Ci * g->Co;
    if (unpack_tiled_ok(g, Ci_real, Co_real)) {
        hipLaunchKernelGGL(lvt_unpack_wgrad_tiled_kernel, dim3((unsigned)((g->Co / UW_CO) * (g->Ci / UW_CI))), dim3(64 * UW_WAVES),
                           (size_t)UW_CO * (UW_CI * taps + 1) * sizeof(float), s, partial, stride, splits, dw, taps, g->Ci, g->Co);
        return;
    }
    int L = 1;
    while (L < 64 && L * 4 <= splits) L <<= 1;
    long long blocks = lvt_cdiv(total * L, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lvt_unpack_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, stride, splits, L, dw, taps,
                       g->Ci, g->Co, Ci_real, Co_real, (const float *)nullptr, (float *)nullptr);
}
// 1 when lvt_conv3d_bwd_weight also produces the bias gradient (db) for this geometry: every served geometry does (the implicit-
// GEMM path sums the dy tiles it streams, the frame-resident kernels the dy rows / patches they stage)
// (LVT_NO_FRAME_BIAS=1: A/B switch -- the frame-resident geometries answer 0 again and callers fall back to lvt_colsum)
extern "C" int lvt_conv3d_bwd_weight_fuses_bias(const lvt_conv_geom *g, int flags) {
    static const int off = getenv("LVT_NO_FRAME_BIAS") ? 1 : 0;
    if (!g) return 0;
    if (flags & LVT_WGRAD_DB_OF_X) return !off && lvt_wgrad_frames_role(g, flags) == 3;      // (the stride-2 frame-resident kernel only)
    return off && lvt_wgrad_frames_role(g, flags) ? 0 : 1;
}

static int bwd_weight_splits(const lvt_conv_geom *g) {
    const long long Mg = (long long)g->Kt * g->Kh * g->Kw * g->Ci;
    const long long tiles = lvt_cdiv(Mg, Mg <= 64 ? 64 : 128) * lvt_cdiv(g->Co, 128);
    const long long pix = (long long)g->N * g->To * g->Ho * g->Wo;
    const int s = choose_splits(tiles, (int)(pix < 0x7fffffffLL ? pix : 0x7fffffff), 512);
    return s < 2 ? 2 : s;   // always go through the partial buffer (the unpack kernel reduces it)
}

extern "C" size_t lvt_conv3d_bwd_weight_workspace_bytes(const lvt_conv_geom *g) {
    if (!g) return 0;
    const long long Mg = (long long)g->Kt * g->Kh * g->Kw * g->Ci;
    const size_t generic = (size_t)bwd_weight_splits(g) * (Mg + 1) * g->Co * sizeof(float);      // + one row per split: column sums
    const size_t frames = lvt_wgrad_frames_workspace_bytes(g);
    return frames > generic ? frames : generic;
}

extern "C" int lvt_conv3d_bwd_weight(const lvt_conv_geom *g, const float *x, const float *dy, float *dw, float *db,
                                     int Ci_real, int Co_real, int flags, const lvt_amax_io *ax, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    int rc = check_geom(g, "conv3d_bwd_weight"); if (rc) return rc;
    LVT_REQUIRE(x && dy && dw && Ci_real <= g->Ci && Co_real <= g->Co, "conv3d_bwd_weight: bad args");
    const size_t need = lvt_conv3d_bwd_weight_workspace_bytes(g);
    if (!workspace || workspace_bytes < need) {
        lvt_set_error("conv3d_bwd_weight: workspace %zu < %zu", workspace_bytes, need);
        return LVT_EWORKSPACE;
    }
    const long long pix = (long long)g->N * g->To * g->Ho * g->Wo;
    LVT_REQUIRE(pix < 0x7fffffffLL, "conv3d_bwd_weight: too many positions");
    LVT_REQUIRE((flags & ~(LVT_MATH_F32 | LVT_MATH_F16X2 | LVT_WGRAD_DB_OF_X)) == 0, "conv3d_bwd_weight: only LVT_MATH_* / LVT_WGRAD_DB_OF_X are accepted in flags");
    const int db_of_x = (flags & LVT_WGRAD_DB_OF_X) ? 1 : 0;
    flags &= ~LVT_WGRAD_DB_OF_X;
    LVT_REQUIRE(!db_of_x || !db || (lvt_wgrad_frames_role(g, flags) == 3 && lvt_aligned16(x) && lvt_aligned16(dy)),
                "conv3d_bwd_weight: LVT_WGRAD_DB_OF_X is served by the stride-2 frame-resident kernel only (lvt_conv3d_bwd_weight_fuses_bias)");
    LVT_REQUIRE_AMAX(flags, ax, "conv3d_bwd_weight");
    const bool f16 = math_of(flags) == 2;
    if (lvt_wgrad_frames_role(g, flags) && lvt_aligned16(x) && lvt_aligned16(dy)) {
        rc = lvt_wgrad_frames_launch(g, x, dy, dw, Ci_real, Co_real, workspace, (hipStream_t)stream, unpack_plain_wgrad,
                                     f16 ? ax->a : nullptr, f16 ? ax->b : nullptr, db, db_of_x);
        if (rc) return rc;
        LVT_CHECK_LAUNCH("conv3d_bwd_weight (frame-resident)");
        return LVT_OK;
    }
    const int taps = g->Kt * g->Kh * g->Kw;
    KParams p; memset(&p, 0, sizeof(p));
    p.M = taps * g->Ci; p.N = g->Co; p.K = (int)pix;
    p.A = x; p.B = dy; p.ldb = g->Co; p.batch_inner = 1; p.alpha = 1.f; p.g = *g; p.flags = flags;
    if (f16) { p.a_amax = ax->a; p.b_amax = ax->b; }
    p.splits = bwd_weight_splits(g);
    p.k_per_split = (int)(lvt_cdiv(lvt_cdiv(p.K, p.splits), BK) * BK);
    p.partial = (float *)workspace;
    p.partial_stride = (long long)p.M * p.N;
    p.colsum_partial = db ? p.partial + (long long)p.splits * p.partial_stride : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (p.M <= 64) rc = launch_tile<A_CONV_M, B_NPLAIN, 64, 128, 2, 2>(p, 1, s);     // image-side layers: 16 taps x 4 channels
    else rc = launch_tile<A_CONV_M, B_NPLAIN, 128, 128, 2, 2>(p, 1, s);
    if (rc) return rc;
    const long long total = (long long)taps * g->Ci * g->Co;
    int L = 1;
    while (L < 64 && L * 4 <= p.splits) L <<= 1;                     // ~4 splits per lane
    if (unpack_tiled_ok(g, Ci_real, Co_real)) {
        // the weight part on the tiled kernel; the generic kernel then only reduces the bias gradient (taps = 0: no elements)
        hipLaunchKernelGGL(lvt_unpack_wgrad_tiled_kernel, dim3((unsigned)((g->Co / UW_CO) * (g->Ci / UW_CI))), dim3(64 * UW_WAVES),
                           (size_t)UW_CO * (UW_CI * taps + 1) * sizeof(float), s, (const float *)p.partial, p.partial_stride, p.splits,
                           dw, taps, g->Ci, g->Co);
        if (db)
            hipLaunchKernelGGL(lvt_unpack_wgrad_kernel, dim3((unsigned)lvt_cdiv(Co_real, 4)), dim3(256), 0, s, p.partial, p.partial_stride,
                               p.splits, L, dw, 0, g->Ci, g->Co, Ci_real, Co_real, (const float *)p.colsum_partial, db);
        LVT_CHECK_LAUNCH("lvt_unpack_wgrad_tiled_kernel");
        return LVT_OK;
    }
    long long blocks = lvt_cdiv(total * L, 256);
    if (blocks > 8192) blocks = 8192;
    if (db && blocks < lvt_cdiv(Co_real, 4)) blocks = lvt_cdiv(Co_real, 4);          // one wave per bias entry
    hipLaunchKernelGGL(lvt_unpack_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.partial, p.partial_stride,
                       p.splits, L, dw, taps, g->Ci, g->Co, Ci_real, Co_real, (const float *)p.colsum_partial, db);
    LVT_CHECK_LAUNCH("lvt_unpack_wgrad_kernel");
    return LVT_OK;
}

// ---- one-hot transposed GEMM (embedding / one-hot-linear weight gradients) ----------------------------
// (tables with enough rows go to the gather in transformer.hip: exact fp32 sums in row order, no zero multiplies)
bool lvt_onehot_gather_ok(int nslots, int V, int N, long long ldb, const float *dout);
int lvt_onehot_gather_launch(const long long *idx, int nslots, int V, const int *slot_off, long long bstride, long long pstride,
                             int P, long long rows, const float *dout, long long ldb, int N, float *out, hipStream_t s);
static int onehot_splits(int M, int N, long long rows) {
    const long long tiles = lvt_cdiv(M, 128) * lvt_cdiv(N, 128);
    const int s = choose_splits(tiles, (int)rows, 512);
    return s < 2 ? 2 : s;
}
extern "C" size_t lvt_onehot_tn_workspace_bytes(int nslots, int V, int N, long long rows) {
    return (size_t)onehot_splits(nslots * V, N, rows) * nslots * V * (size_t)N * sizeof(float);
}
extern "C" int lvt_onehot_tn_is_gather(int nslots, int V, int N, long long ldb, const float *dout, int flags) {
    return !(flags & LVT_ONEHOT_DENSE) && lvt_onehot_gather_ok(nslots, V, N, ldb, dout);
}
extern "C" int lvt_onehot_tn_gemm(const long long *idx, int nslots, int V, const int *slot_off, long long bstride,
                                  long long pstride, int P, long long rows, const float *dout, long long ldb, int N,
                                  float *out, int flags, const float *dout_amax, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    LVT_REQUIRE(idx && slot_off && dout && out && nslots > 0 && nslots <= 32 && V > 0 && V % 4 == 0,
                "onehot_tn_gemm: bad args");
    LVT_REQUIRE(rows > 0 && rows < 0x7fffffffLL && P > 0 && rows % P == 0 && N % 4 == 0 && ldb % 4 == 0,
                "onehot_tn_gemm: bad shape");
    const size_t need = lvt_onehot_tn_workspace_bytes(nslots, V, N, rows);
    if (!workspace || workspace_bytes < need) {
        lvt_set_error("onehot_tn_gemm: workspace %zu < %zu", workspace_bytes, need);
        return LVT_EWORKSPACE;
    }
    if (lvt_onehot_tn_is_gather(nslots, V, N, ldb, dout, flags))
        return lvt_onehot_gather_launch(idx, nslots, V, slot_off, bstride, pstride, P, rows, dout, ldb, N, out, (hipStream_t)stream);
    KParams p; memset(&p, 0, sizeof(p));
    p.M = nslots * V; p.N = N; p.K = (int)rows;
    LVT_REQUIRE(math_of(flags) != 2 || dout_amax, "onehot_tn_gemm: LVT_MATH_F16X2 needs dout_amax");
    p.B = dout; p.ldb = ldb; p.batch_inner = 1; p.alpha = 1.f; p.flags = flags & (LVT_MATH_F32 | LVT_MATH_F16X2);
    p.b_amax = dout_amax;             // (the one-hot operand is exact in fp16 unscaled: a_amax stays NULL)
    p.oh_idx = idx; p.oh_bstride = bstride; p.oh_pstride = pstride; p.oh_P = P; p.oh_V = V;
    for (int i = 0; i < nslots; ++i) p.oh_off[i] = slot_off[i];
    p.splits = onehot_splits(p.M, N, rows);
    p.k_per_split = (int)(lvt_cdiv(lvt_cdiv(p.K, p.splits), BK) * BK);
    p.partial = (float *)workspace;
    p.partial_stride = (long long)p.M * N;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_tile<A_ONEHOT_M, B_NPLAIN, 128, 128, 2, 2>(p, 1, s);
    if (rc) return rc;
    const long long n4 = p.partial_stride / 4;
    const int blocks = (int)(lvt_cdiv(n4, 256) < 2048 ? lvt_cdiv(n4, 256) : 2048);
    hipLaunchKernelGGL(lvt_reduce_splits_kernel, dim3(blocks), dim3(256), 0, s, p.partial, n4, p.partial_stride,
                       p.splits, out, 0);
    LVT_CHECK_LAUNCH("lvt_reduce_splits_kernel");
    return LVT_OK;
}

// rows handled by one workgroup per stage: small enough that stage 1 fills the chip (>= ~256 workgroups
// for 16k rows), large enough that the recursion is at most 3 launches deep for 2M rows.
static long long cs_rows(long long rows) { return rows >= (1 << 17) ? 128 : 64; }
extern "C" size_t lvt_colsum_workspace_bytes(long long M, int N) {
    long long total = 2, rows = M;
    while (rows > 1) { rows = lvt_cdiv(rows, cs_rows(rows)); total += rows; }
    return (size_t)total * N * sizeof(float);
}

extern "C" int lvt_colsum(const float *g, long long M, int N, long long ld, float *out, void *workspace,
                          size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(g && out && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && lvt_aligned16(g), "colsum: bad args");
    if (!workspace || workspace_bytes < lvt_colsum_workspace_bytes(M, N)) {
        lvt_set_error("colsum: workspace too small");
        return LVT_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const float *src = g;
    long long rows = M, lds_ = ld;
    float *buf = (float *)workspace;
    // stages: rows -> ceil(rows / CS_ROWS) until one row is left; the last stage writes `out`
    while (true) {
        const long long rpb = cs_rows(rows);
        const long long nblk = lvt_cdiv(rows, rpb);
        float *dst = (nblk == 1) ? out : buf;
        hipLaunchKernelGGL(lvt_colsum_kernel, dim3((unsigned)nblk), dim3(CS_THREADS), 0, s, src, rows, N, lds_,
                           rpb, dst);
        LVT_CHECK_LAUNCH("lvt_colsum_kernel");
        if (nblk == 1) break;
        src = dst; rows = nblk; lds_ = N; buf = dst + nblk * N;
    }
    return LVT_OK;
}
