// Plane-fed GEMM of the f16x2 arithmetic (round 6): operands that arrive as ready fp16 planes are staged by LDS-DMA
// (global_load_lds_dwordx4: global memory -> LDS with no register round trip, no split arithmetic, no ds_write).
//
// Why.  lvt_gemm_wide_kernel (gemm_engine.hip) stages both fp32 operands through registers and splits them on the vector
// ALU while the matrix pipe works on the previous k-tile: 6.2-7.9 VALU instructions per MFMA, and the staging stream alone
// takes as long as the MFMA stream alone (profiles/r04_wide_gemm_loop_experiments.txt, block 1: 201 us shipped, 127 us with
// neither split nor LDS stores).  Round 4 built ready planes for the weights and kept the register path (two 8-byte loads +
// two LDS stores per slot): no gain, the stream is bound by its memory / LDS instruction issue.  Here the planes reach LDS
// without touching the vector ALU at all.
//
// The "P2 image" of a matrix X[rows][K] (K % 32 == 0) under a power-of-two scale s (lvt_f16_scale of a device scalar
// >= max |X|, exactly the rule of the in-kernel split):
//     X s = hi + 2^-11 lo,  hi = RN16(X s),  lo = RN16(2^11 (X s - hi))               (gemm_engine.hip, f16_split_pair<2048>)
// stored row-major with the SAME footprint as the fp32 matrix (row pitch ld floats = 4 ld bytes): every group of 32 consecutive
// k of a row is one 128-byte line = [hi of k0 .. k0+31 (64 B) | lo of k0 .. k0+31 (64 B)].  A k-tile of a row is therefore
// ONE full cache line, and a wave-instruction of the loader copies 8 rows x 128 B.
//
// LDS image of a staged tile: [row][8 units of 16 B], unit u of row r at byte r * 128 + ((u ^ ((r >> 1) & 7)) << 4): the
// 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ... : MI355X_MICROARCH.md, LDS) then hit 16 distinct 16-byte slots for
// every (plane, k half) fragment.  LDS-DMA writes lane-linear (wave-uniform base + 16 lane), so the swizzle is applied to
// the SOURCE address of each lane: lane L of the instruction that fills rows 8 c .. 8 c + 7 fetches unit (L & 7) ^ x of row
// 8 c + (L >> 3).
//
// Forms: C[z] = epi(alpha A[z] B[z]^T), both operands k-contiguous (NT).  B is always a P2 image (weights: one
// lvt_p2_pack_multi launch per pass writes W and W^T images); A is a P2 image (LayerNorm outputs: lvt_layernorm_fwd_p2,
// a-priori bound) or plain fp32 (gradients, whose scale is not known before they are complete: fetched and split through
// registers as in the wide kernel, into the same LDS image).  Results are BIT-IDENTICAL to lvt_gemm_wide_kernel on the same
// operands: same split, same MFMA order, same accumulators (tests/test_gpu_p2.py).
#include "lvt_common.h"
#include "epilogue_fast.h"
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void p2_lds_void;
typedef __attribute__((address_space(1))) const void p2_gl_void;

#define P2_THREADS 512
#define P2_BM 256
#define P2_BN 128
#define P2_BK 32
#define P2_A_STAGE (P2_BM * 128)                 // bytes of one staged A tile
#define P2_B_STAGE (P2_BN * 128)
#define P2_B_BASE (2 * P2_A_STAGE)
#define P2_LDS_BYTES (2 * P2_A_STAGE + 2 * P2_B_STAGE)

struct P2Params {
    int M, N, K;
    const char *A; long long lda; int a_kb; long long a_skb;
    const char *B; long long ldb;
    float *C; long long ldc;
    int batch_inner;
    long long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    float alpha; int flags;
    const float *bias; const float *res; long long ldr; const float *mask; long long ldm;
    const float *a_amax, *b_amax; float *c_amax;
    char *Cp; long long ldcp; const float *cp_amax;     // optional second output: the P2 image of C under the scale of *cp_amax
    int stagger;                                        // experiment: first-wave workgroups start (id >> 3 & 3) * stagger * 64 cycles late
    unsigned long long *dbg;                            // P2_X_DBG builds: 4 timestamps per workgroup
};

// ---- the f16x2 split: identical to gemm_engine.hip (f16_split_pair<2048>, lvt_f16_scale) ----------------------------------
__device__ __forceinline__ float p2_mix_lo(unsigned h, float k, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
__device__ __forceinline__ float p2_mix_hi(unsigned h, float k, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
__device__ __forceinline__ void p2_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const f32x2 v = {a, b};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v * s, f16x2v));
    const f32x2 t2 = v * (s * 2048.f);
    const f32x2 r = {p2_mix_lo(ph, -2048.f, t2.x), p2_mix_hi(ph, -2048.f, t2.y)};       // exact
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
__device__ __forceinline__ void p2_split4(const float4 v, const float s, uint2 &ph, uint2 &pl) {
    p2_split_pair(v.x, v.y, s, ph.x, pl.x);
    p2_split_pair(v.z, v.w, s, ph.y, pl.y);
}
__device__ __forceinline__ float p2_scale(const float *amax, int &unscale) {
    unsigned bits = __float_as_uint(*amax);
    const int eb = (int)((bits >> 23) & 0xffu);
    int se = 268 - eb;
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    unscale -= se - 127;
    return __uint_as_float((unsigned)se << 23);
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------
template <int AP>
__global__ __launch_bounds__(P2_THREADS, 2) void lvt_gemm_p2_kernel(const P2Params p) {
    constexpr int BM = P2_BM, BN = P2_BN, BK = P2_BK, WN = 2, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(128))) char lds[P2_LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    // blockIdx -> tile: every XCD (workgroup b runs on XCD b % 8) gets one contiguous run of tiles (lvt_tile_ctx, gemm_engine.hip)
    const int ntn = (p.N + BN - 1) / BN;
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = wg & 7, slot = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n0 = (wg % ntn) * BN, m0 = (wg / ntn) * BM;
    const int z = blockIdx.y, zo = z / p.batch_inner, zi = z % p.batch_inner;
    const char *Ak = p.A + (zo * p.sA_o + zi * p.sA_i) * 4;
    const char *Bk = p.B + (zo * p.sB_o + zi * p.sB_i) * 4;
    const long long coff = zo * p.sC_o + zi * p.sC_i;

#ifdef P2_X_DBG
#define P2_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define P2_STAMP(i) do {} while (0)
#endif
    P2_STAMP(0);
    if (p.stagger && blockIdx.y == 0 && blockIdx.x < 256) {
        const int ph = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < ph * p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    int unscale = 0;
    const float sa = p2_scale(p.a_amax, unscale);
    (void)p2_scale(p.b_amax, unscale);

    // ---- loaders.  LDS-DMA: wave w fills chunks (8 rows x 128 B = 1 KB) 4 w .. 4 w + 3 of the A tile, 2 w, 2 w + 1 of the B tile.
    unsigned agoff[4], bgoff[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int row = (wave_s * 4 + c) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        agoff[c] = (unsigned)((long long)min(m0 + row, p.M - 1) * p.lda * 4 + u * 16);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int row = (wave_s * 2 + c) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        bgoff[c] = (unsigned)((long long)min(n0 + row, p.N - 1) * p.ldb * 4 + u * 16);
    }
    // fp32 A through registers (AP == 0): thread = (rows perm(q) + 64 i, k quad kq); perm swaps bits 0 and 3 of q = tid >> 3, so
    // that the two rows a 16-lane group of ds_write_b64 touches differ in bit 3 (their swizzles put them in disjoint 64-byte
    // halves of the 128-byte bank window: conflict-free).
    float4 av[4];
    unsigned aroff[4];
    int a_kin = 0;
    const int kq = tid & 7, q6 = tid >> 3;
    const int prow = (q6 & 0x36) | ((q6 & 1) << 3) | ((q6 >> 3) & 1);
    if (!AP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aroff[i] = (unsigned)(((long long)min(m0 + prow + 64 * i, p.M - 1) * p.lda + kq * 4) * 4);
    }
    const unsigned a_st = (unsigned)(prow * 128 + ((((kq >> 1) ^ ((prow >> 1) & 7)) << 4) | ((kq & 1) << 3)));   // hi; lo = ^ 64

    auto fetch_a = [&]() {              // AP == 0: the next k-tile of A into registers
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const float4 *>(Ak + aroff[i]);
        a_kin += BK;
        const bool wrap = a_kin >= p.a_kb;              // (a_kb is a multiple of BK: a tile never straddles two k blocks)
        Ak += wrap ? ((long long)p.a_skb - p.a_kb + BK) * 4 : (long long)BK * 4;
        a_kin = wrap ? 0 : a_kin;
    };
    auto store_a = [&](int stage) {     // AP == 0: split the registers into the LDS image of `stage`
        const unsigned base = (unsigned)(stage * P2_A_STAGE) + a_st;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 ph, pl;
            p2_split4(av[i], sa, ph, pl);
            const unsigned off = base + i * (64 * 128);
            *reinterpret_cast<uint2 *>(lds + off) = ph;
            *reinterpret_cast<uint2 *>(lds + (off ^ 64u)) = pl;
        }
    };
    auto dma_a = [&](int stage) {       // AP == 1
        char *dst = lds + stage * P2_A_STAGE + wave_s * 4096;
#ifdef P2_X_IMMOFF       // (experiment: one M0 per stage, the chunk through the instruction offset -- it moves BOTH addresses)
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Ak + agoff[0]), (p2_lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Ak + agoff[1] - 1024), (p2_lds_void *)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Ak + agoff[2] - 2048), (p2_lds_void *)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Ak + agoff[3] - 3072), (p2_lds_void *)dst, 16, 3072, 0);
#else
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __builtin_amdgcn_global_load_lds((p2_gl_void *)(Ak + agoff[c]), (p2_lds_void *)(dst + c * 1024), 16, 0, 0);
#ifdef P2_X_NOPS
            asm volatile("s_nop 7\n\ts_nop 7");
#endif
        }
#endif
        Ak += 128;
    };
    auto dma_b = [&](int stage) {
        char *dst = lds + P2_B_BASE + stage * P2_B_STAGE + wave_s * 2048;
#ifdef P2_X_IMMOFF
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Bk + bgoff[0]), (p2_lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((p2_gl_void *)(Bk + bgoff[1] - 1024), (p2_lds_void *)dst, 16, 1024, 0);
#else
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            __builtin_amdgcn_global_load_lds((p2_gl_void *)(Bk + bgoff[c]), (p2_lds_void *)(dst + c * 1024), 16, 0, 0);
#ifdef P2_X_NOPS
            asm volatile("s_nop 7\n\ts_nop 7");
#endif
        }
#endif
        Bk += 128;
    };

    // "Has my DMA landed?"  s_waitcnt vmcnt(0) is NOT enough on this part: with six LDS-DMA instructions per wave and k-tile in
    // flight, waves passed vmcnt(0) + s_barrier and read stale LDS -- single 1 KB chunks of a tile still missing, 900 of 1536
    // tiles of a 16384 x 3072 x 512 product wrong, fewer with a sleep behind the barrier, none with ONE LDS read by the issuing
    // wave behind the wait (profiles/r06_lds_dma_visibility.txt): the LDS serves a wave's read after that wave's pending DMA
    // writes.  So every wave reads one dword from each chunk it requested (one ds_read_b32: lane -> chunk lane % n) and waits for
    // it before the barrier.  (The two-DMA form, AP == 0, never failed; it gets the same fence.)
    constexpr int NPROBE = AP ? 6 : 2;
    const int pc = lane % NPROBE;
    const bool p_is_a = AP && pc < 4;
    const unsigned probe0 = p_is_a ? (unsigned)(wave_s * 4096 + pc * 1024 + lane * 16)
                                   : (unsigned)(P2_B_BASE + wave_s * 2048 + (pc - (AP ? 4 : 0)) * 1024 + lane * 16);
    const unsigned probe_st = p_is_a ? P2_A_STAGE : P2_B_STAGE;
    auto landed = [&](int stage) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int v = *reinterpret_cast<volatile int *>(lds + probe0 + stage * probe_st);
        (void)v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    f32x16 acc[TM][TN], acx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }

    // fragment addresses: row (.. + l31) of a 32-row block, unit u = 4 plane + 2 (ks / 16) + half, swizzled by x = (l31 >> 1) & 7
    const int fx = (l31 >> 1) & 7;
    unsigned fo[2][2];                  // [ks / 16][plane]: byte offset of the lane's unit inside its row
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) fo[s][qq] = (unsigned)(((4 * qq + 2 * s + half) ^ fx) << 4);
    const unsigned fa0 = (unsigned)((wm * (TM * 32) + l31) * 128), fb0 = (unsigned)(P2_B_BASE + (wn * (TN * 32) + l31) * 128);

    auto tile = [&](int stage, auto do_next, auto do_next2) {
        const char *ca = lds + stage * P2_A_STAGE, *cb = lds + stage * P2_B_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 a[2][TM], b[2][TN];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[qq][i] = *reinterpret_cast<const f16x8 *>(ca + fa0 + i * 4096 + fo[s][qq]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[qq][j] = *reinterpret_cast<const f16x8 *>(cb + fb0 + j * 4096 + fo[s][qq]);
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
            if (!AP && s == 0) {
                if constexpr (decltype(do_next)::value) store_a(stage ^ 1);
                if constexpr (decltype(do_next2)::value) fetch_a();
            }
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;

    const int ntiles = p.K / BK;
    // prologue: tile 0 into stage 0
    if (AP) dma_a(0);
    else fetch_a();
    dma_b(0);
    if (!AP) {
        store_a(0);
        if (ntiles > 1) fetch_a();
    }
    P2_STAMP(1);
    int kt = 0;
    if (ntiles == 1) { landed(0); __syncthreads(); P2_STAMP(2); }
    for (; kt + 2 < ntiles; ++kt) {
        landed(kt & 1);
        __syncthreads();                // tile kt has landed (LDS-DMA: vmcnt(0) of every wave, then the barrier); stage (kt + 1) & 1 is free
#ifdef P2_X_SLEEP
        __builtin_amdgcn_s_sleep(20); __syncthreads();
#endif
        if (AP) dma_a((kt + 1) & 1);
        dma_b((kt + 1) & 1);
#ifdef P2_X_SLEEP2       // (experiment: delay only the compute of tile kt, after the next tile's DMA is issued)
        __builtin_amdgcn_s_sleep(20);
#endif
        tile(kt & 1, yes_t(), yes_t());
    }
    if (kt + 1 < ntiles) {
        landed(kt & 1);
        __syncthreads();
        if (AP) dma_a((kt + 1) & 1);
        dma_b((kt + 1) & 1);
        tile(kt & 1, yes_t(), no_t());
        ++kt;
    }
    landed(kt & 1);
    __syncthreads();
    tile(kt & 1, no_t(), no_t());
    __syncthreads();
    P2_STAMP(3);

    lvt_f16x2_finish<TM, TN>(acc, acx, unscale);

    P2_STAMP(4);
    // ---- epilogue.  Plain forms: epilogue_fast.h (no workgroup barrier, compile-time flag sets, max |C| peeked up front)
#ifndef P2_NO_FAST_EPI
    if (!p.Cp) {
#ifdef P2_X_EPI_BARRIER
        __syncthreads();
#endif
        const unsigned seen = lvt_amax_peek(p.c_amax);
        LvtEpi e;
        e.M = p.M; e.N = p.N; e.C = p.C; e.ldc = p.ldc; e.coff = coff; e.alpha = p.alpha; e.flags = p.flags;
        e.bias = p.bias; e.res = p.res; e.ldr = p.ldr; e.mask = p.mask; e.ldm = p.ldm;
        const float am_w = lvt_epi_fast_dispatch<TM, TN>(e, acc, reinterpret_cast<float *>(lds) + wave * (32 * TN * 32),
                                                         m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
        P2_STAMP(5);
        if (p.c_amax) lvt_block_amax_commit_seen(am_w, p.c_amax, reinterpret_cast<float *>(lds + 8 * 32 * TN * 32 * 4), seen);
        P2_STAMP(6);
        return;
    }
#endif
    // ---- epilogue: every wave turns its 64 x 64 sub-tile through LDS 32 rows at a time (lvt_epilogue_vec, gemm_engine.hip):
    // a lane then owns 4 consecutive columns, all global accesses are 16-byte ones
    constexpr int SW = TN * 32, C4 = SW / 4;
    float *tile_f = reinterpret_cast<float *>(lds) + wave * (32 * SW);
    const int flags = p.flags;
    float am = 0.f;
    int cp_un = 0;
    const float cps = p.Cp ? p2_scale(p.cp_amax, cp_un) : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile_f[((r & 3) + 8 * (r >> 2) + 4 * half) * SW + 32 * j + l31] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 32 * C4 / 64; ++u) {
            const int idx = lane + 64 * u;
            const int rowl = idx / C4, c4 = idx % C4;
            const long long row = m0 + wm * (TM * 32) + i * 32 + rowl;
            const int col = n0 + wn * SW + 4 * c4;
            if (row < p.M && col < p.N) {
                float4 v = *reinterpret_cast<const float4 *>(&tile_f[rowl * SW + 4 * c4]);
                v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
                if (flags & LVT_EPI_BIAS) { const float4 b = *reinterpret_cast<const float4 *>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (flags & LVT_EPI_RESIDUAL) { const float4 b = *reinterpret_cast<const float4 *>(p.res + coff + row * p.ldr + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (flags & LVT_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (flags & LVT_EPI_MASK) {
                    const float4 mk = *reinterpret_cast<const float4 *>(p.mask + coff + row * p.ldm + col);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                am = fmaxf(am, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
#ifndef P2_X_NOSTORE        // (timing experiment: the epilogue without its global stores)
                *reinterpret_cast<float4 *>(p.C + coff + row * p.ldc + col) = v;
#else
                if (v.x == 1.2345e33f) *reinterpret_cast<float4 *>(p.C + coff + row * p.ldc + col) = v;
#endif
                if (p.Cp) {
                    uint2 ph, pl;
                    p2_split4(v, cps, ph, pl);
                    char *cp = p.Cp + (coff + row * p.ldcp) * 4 + (col >> 5) * 128 + (col & 31) * 2;
                    *reinterpret_cast<uint2 *>(cp) = ph;
                    *reinterpret_cast<uint2 *>(cp + 64) = pl;
                }
            }
        }
        __syncthreads();
    }
    P2_STAMP(5);
    if (p.c_amax) lvt_block_amax_commit(am, p.c_amax, reinterpret_cast<float *>(lds));
    P2_STAMP(6);
}
#ifdef P2_X_DBG
static unsigned long long *g_p2_dbg = nullptr;
extern "C" void lvt_p2_debug_buffer(void *p) { g_p2_dbg = (unsigned long long *)p; }
#endif

extern "C" int lvt_gemm_p2_f32(const lvt_gemm_p2_desc *d, void *stream) {
    LVT_REQUIRE(d && d->A && d->B && d->C, "gemm_p2: null pointer");
    LVT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->K % P2_BK == 0, "gemm_p2: bad shape %d %d %d (K %% 32)", d->M, d->N, d->K);
    LVT_REQUIRE(d->a_amax && d->b_amax, "gemm_p2: the operand scales (a_amax, b_amax) are required");
    LVT_REQUIRE(d->ldb % 32 == 0 && ((uintptr_t)d->B & 127) == 0 && d->sB_o % 32 == 0 && d->sB_i % 32 == 0,
                "gemm_p2: B is a P2 image: 128-byte aligned, ldb and batch strides %% 32 == 0");
    if (d->a_planes) {
        LVT_REQUIRE(d->lda % 32 == 0 && ((uintptr_t)d->A & 127) == 0 && d->sA_o % 32 == 0 && d->sA_i % 32 == 0,
                    "gemm_p2: A as a P2 image: 128-byte aligned, lda and batch strides %% 32 == 0");
        LVT_REQUIRE(d->a_kb <= 0 || d->a_kb == d->K, "gemm_p2: a P2 image has a single k level");
    } else {
        LVT_REQUIRE(d->lda % 4 == 0 && lvt_aligned16(d->A) && d->sA_o % 4 == 0 && d->sA_i % 4 == 0, "gemm_p2: A alignment");
        LVT_REQUIRE(d->a_kb <= 0 || (d->a_kb % P2_BK == 0 && d->a_skb % 4 == 0), "gemm_p2: a_kb %% 32, a_skb %% 4");
    }
    LVT_REQUIRE(d->N % 4 == 0 && lvt_aligned16(d->C) && d->ldc % 4 == 0 && d->sC_o % 4 == 0 && d->sC_i % 4 == 0,
                "gemm_p2: C must be 16-byte aligned with N, ldc, batch strides %% 4 == 0");
    LVT_REQUIRE(!(d->flags & LVT_EPI_BIAS) || (d->bias && lvt_aligned16(d->bias)), "gemm_p2: BIAS flag without an aligned bias");
    LVT_REQUIRE(!(d->flags & LVT_EPI_RESIDUAL) || (d->res && lvt_aligned16(d->res) && d->ldr % 4 == 0), "gemm_p2: RESIDUAL flag");
    LVT_REQUIRE(!(d->flags & LVT_EPI_MASK) || (d->mask && lvt_aligned16(d->mask) && d->ldm % 4 == 0), "gemm_p2: MASK flag");
    LVT_REQUIRE(!(d->flags & ~(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU | LVT_EPI_MASK | LVT_MATH_F16X2)),
                "gemm_p2: flags 0x%x not served (BIAS, RESIDUAL, RELU, MASK)", d->flags);
    LVT_REQUIRE(!d->Cp || (d->cp_amax && d->ldcp % 32 == 0 && ((uintptr_t)d->Cp & 127) == 0 && d->N % 32 == 0 &&
                           d->sC_o % 32 == 0 && d->sC_i % 32 == 0),
                "gemm_p2: the P2 image of C needs cp_amax, a 128-byte aligned Cp, ldcp, N and batch strides %% 32 == 0");
    const long long a_kb = d->a_kb > 0 ? d->a_kb : d->K;
    const long long a_span = (long long)d->M * d->lda + (long long)(d->K / a_kb) * d->a_skb, b_span = (long long)d->N * d->ldb;
    LVT_REQUIRE(a_span < (1LL << 30) && b_span < (1LL << 30), "gemm_p2: an operand spans 4 GB or more per batch");
    const int bi = d->batch_inner > 0 ? d->batch_inner : 1, zc = (d->batch_outer > 0 ? d->batch_outer : 1) * bi;
    const long long ntm = lvt_cdiv(d->M, P2_BM), ntn = lvt_cdiv(d->N, P2_BN);
    LVT_REQUIRE(ntm * ntn <= 0x7fffffffLL && zc <= 65535, "gemm_p2: grid too large");
    P2Params p;
    memset(&p, 0, sizeof(p));
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.A = (const char *)d->A; p.lda = d->lda; p.a_kb = (int)a_kb; p.a_skb = d->a_skb;
    p.B = (const char *)d->B; p.ldb = d->ldb;
    p.C = d->C; p.ldc = d->ldc;
    p.batch_inner = bi;
    p.sA_o = d->sA_o; p.sA_i = d->sA_i; p.sB_o = d->sB_o; p.sB_i = d->sB_i; p.sC_o = d->sC_o; p.sC_i = d->sC_i;
    p.alpha = d->alpha; p.flags = d->flags;
    p.bias = d->bias; p.res = d->res; p.ldr = d->ldr; p.mask = d->mask; p.ldm = d->ldm;
    p.a_amax = d->a_amax; p.b_amax = d->b_amax; p.c_amax = d->c_amax;
    p.Cp = (char *)d->Cp; p.ldcp = d->ldcp; p.cp_amax = d->cp_amax;
    { static const char *st = getenv("LVT_P2_STAGGER"); p.stagger = st ? atoi(st) : 0; }
#ifdef P2_X_DBG
    p.dbg = g_p2_dbg;
#endif
    dim3 grid((unsigned)(ntm * ntn), (unsigned)zc, 1);
    if (d->a_planes) hipLaunchKernelGGL((lvt_gemm_p2_kernel<1>), grid, dim3(P2_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lvt_gemm_p2_kernel<0>), grid, dim3(P2_THREADS), 0, (hipStream_t)stream, p);
    LVT_CHECK_LAUNCH("lvt_gemm_p2_kernel");
    return LVT_OK;
}

// ---- P2 images of many matrices in one launch (the weights of a model, once per pass) -----------------------------------------
// entry: src fp32 [rows][cols] (row pitch ld_src floats); transpose == 0: dst image rows = src rows, k = src columns (cols % 32 == 0);
// transpose == 1: dst image rows = src columns, k = src rows (rows % 32 == 0).  dst row pitch ld_dst floats (% 32 == 0), scale from
// *amax.  A workgroup owns a 64 (image rows) x 64 (k) block: the fp32 block goes through LDS, every thread then builds the hi and lo
// units (8 consecutive k) of one image row.
#define P2_PACK_MAX 32      // (32 entries of 80 bytes: the table travels as a kernel argument)
struct P2PackTable { lvt_p2_pack_entry e[P2_PACK_MAX]; int first_block[P2_PACK_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void lvt_p2_pack_multi_kernel(const P2PackTable t) {
    __shared__ float blk[64][65];
    int ei = 0;
    while (ei + 1 < t.n && (int)blockIdx.x >= t.first_block[ei + 1]) ++ei;
    lvt_p2_pack_entry e = t.e[ei];
    int local = blockIdx.x - t.first_block[ei];
    const int irows = e.transpose ? e.cols : e.rows, ik = e.transpose ? e.rows : e.cols;     // image rows, image k
    const int nkb = ik / 64 + (ik % 64 ? 1 : 0);
    const int per = ((irows + 63) / 64) * nkb;                                                // blocks per matrix of the entry
    const int zb = local / per;
    local -= zb * per;
    e.src += (long long)zb * e.bs_src;
    e.dst = reinterpret_cast<char *>(e.dst) + (long long)zb * e.bs_dst * 4;
    const int r0 = (local / nkb) * 64, k0 = (local % nkb) * 64;
    const int tid = threadIdx.x;
    // blk[image row][k]
    if (!e.transpose) {
        for (int idx = tid; idx < 64 * 16; idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < irows && k0 + 4 * c4 < ik) v = *reinterpret_cast<const float4 *>(e.src + (long long)(r0 + r) * e.ld_src + k0 + 4 * c4);
            blk[r][4 * c4] = v.x; blk[r][4 * c4 + 1] = v.y; blk[r][4 * c4 + 2] = v.z; blk[r][4 * c4 + 3] = v.w;
        }
    } else {
        for (int idx = tid; idx < 64 * 16; idx += 256) {
            const int kk = idx >> 4, c4 = idx & 15;                  // src row k0 + kk, src columns r0 + 4 c4 ..
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + kk < ik && r0 + 4 * c4 < irows) v = *reinterpret_cast<const float4 *>(e.src + (long long)(k0 + kk) * e.ld_src + r0 + 4 * c4);
            blk[4 * c4][kk] = v.x; blk[4 * c4 + 1][kk] = v.y; blk[4 * c4 + 2][kk] = v.z; blk[4 * c4 + 3][kk] = v.w;
        }
    }
    __syncthreads();
    int un = 0;
    const float s = p2_scale(e.amax, un);
    // 64 rows x 8 units of 8 k: two per thread
    for (int idx = tid; idx < 64 * 8; idx += 256) {
        const int r = idx >> 3, u = idx & 7, k = k0 + 8 * u;
        if (r0 + r >= irows || k >= ik) continue;
        uint2 h0, l0, h1, l1;
        p2_split4(make_float4(blk[r][8 * u], blk[r][8 * u + 1], blk[r][8 * u + 2], blk[r][8 * u + 3]), s, h0, l0);
        p2_split4(make_float4(blk[r][8 * u + 4], blk[r][8 * u + 5], blk[r][8 * u + 6], blk[r][8 * u + 7]), s, h1, l1);
        char *dst = reinterpret_cast<char *>(e.dst) + (long long)(r0 + r) * e.ld_dst * 4 + (k >> 5) * 128 + (k & 31) * 2;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4 *>(dst + 64) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}
extern "C" int lvt_p2_pack_multi(const lvt_p2_pack_entry *entries, int n, void *stream) {
    LVT_REQUIRE(entries && n > 0, "p2_pack_multi: no entries");
    for (int base = 0; base < n; base += P2_PACK_MAX) {
        P2PackTable t;
        t.n = n - base < P2_PACK_MAX ? n - base : P2_PACK_MAX;
        int blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            const lvt_p2_pack_entry &e = entries[base + i];
            const int irows = e.transpose ? e.cols : e.rows, ik = e.transpose ? e.rows : e.cols;
            LVT_REQUIRE(e.src && e.dst && e.amax && e.rows > 0 && e.cols > 0, "p2_pack_multi: entry %d: null pointer / empty", base + i);
            LVT_REQUIRE(ik % 32 == 0 && e.ld_dst % 32 == 0 && e.ld_dst >= ik && ((uintptr_t)e.dst & 127) == 0,
                        "p2_pack_multi: entry %d: image k = %d and ld_dst = %lld must be multiples of 32, dst 128-byte aligned", base + i, ik, e.ld_dst);
            LVT_REQUIRE(e.ld_src % 4 == 0 && e.cols % 4 == 0 && lvt_aligned16(e.src), "p2_pack_multi: entry %d: src alignment", base + i);
            LVT_REQUIRE(e.batch <= 1 || (e.bs_src % 4 == 0 && e.bs_dst % 32 == 0), "p2_pack_multi: entry %d: batch strides (src %% 4, dst %% 32)", base + i);
            t.e[i] = e;
            t.first_block[i] = blocks;
            blocks += (int)(lvt_cdiv(irows, 64) * lvt_cdiv(ik, 64)) * (e.batch > 1 ? e.batch : 1);
        }
        t.first_block[t.n] = blocks;
        hipLaunchKernelGGL(lvt_p2_pack_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        LVT_CHECK_LAUNCH("lvt_p2_pack_multi_kernel");
    }
    return LVT_OK;
}
