// The plain GEMM epilogue, straightened (round 6).
//
// lvt_epilogue_vec (gemm_engine.hip) turns a wave's 64 x 64 accumulator sub-tile through LDS so that a lane owns 4 consecutive
// columns, and then runs, for each of its 16 float4, the whole menu of the engine: run-time flag tests (a branch around every
// bias / residual / mask load), a 64-bit row * ldc product, the ConvTranspose row decode, the bf16 plane forms -- ~1500 vector
// and ~650 scalar instructions with 156 branches per workgroup tile, between four workgroup barriers.  Phase stamps inside
// lvt_gemm_p2_kernel (profiles/r06_gemm_fixed_cost.txt): 4.85 us of the 8.2 us a 256 x 128 tile costs beside its main loop, and
// removing its global stores changes 0.7 us -- the epilogue is bound by its own instruction stream, not by memory.  The K = 512
// products of a transformer layer are one such tile per CU and launch: a third of their time.
//
// Here, for the plain forms (row-major C or the pixel permutation of the frame-resident convolutions, flags in {BIAS, RESIDUAL,
// RELU, MASK}, or the split-K partial store):
//   * the flag set is a template parameter (the kernels switch over the six combinations the models use; anything else takes
//     the run-time form of the same body);
//   * a lane's column is fixed (bias loaded once), its rows step by 4;
//   * the residual / mask values of a 32-row round are requested before its first store;
//   * the LDS turn-table of a wave is private to the wave and LDS executes a wave's instructions in order, so there is NO
//     workgroup barrier inside: waves drain independently;
//   * the current max |C| scalar is read at the START of the epilogue (its latency hides behind the stores) instead of after
//     the reduction (lvt_block_amax_commit pays a global round trip on the critical path of every workgroup).
// Arithmetic and its order are those of lvt_epilogue_vec: results are bit-identical.
#pragma once
#include "lvt_common.h"

typedef float lvt_f32x16 __attribute__((ext_vector_type(16)));

// acc <- (acc + 2^-11 acx) 2^unscale for a wave's accumulator pair (the end of every f16x2 main loop).  ldexpf(fmaf(..)) is two
// full-rate-or-worse instructions per element (v_fma_f32 + v_ldexp_f32, 2.8 issue units); with s = 2^unscale a float, acc s and
// acx (2^-11 s) are exact products and fma(acx, 2^-11 s, acc s) rounds the same exact sum once -- the packed forms (v_pk_mul_f32,
// v_pk_fma_f32) do two elements each.  Same bits unless the result is subnormal (the two-step form rounds such a value twice).
template <int TM, int TN>
__device__ __forceinline__ void lvt_f16x2_finish(lvt_f32x16 (&acc)[TM][TN], lvt_f32x16 (&acx)[TM][TN], int unscale) {
    if (unscale >= -100 && unscale <= 120) {
        const float s = __uint_as_float((unsigned)(unscale + 127) << 23), sx = __uint_as_float((unsigned)(unscale + 116) << 23);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = acx[i][j] * sx + acc[i][j] * s;
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = ldexpf(fmaf(acx[i][j][r], 1.f / 2048.f, acc[i][j][r]), unscale);
    }
}

struct LvtEpi {
    int M, N;
    float *C; long long ldc;            // C + coff already applied by the caller? no: coff is added here (res / mask share it)
    long long coff;
    float alpha; int flags;
    const float *bias; const float *res; long long ldr; const float *mask; long long ldm;
};

// tile row -> row of C: identity for the GEMMs; the frame-resident convolutions permute the pixels of a frame (gemm_engine.hip)
struct LvtRowIdentity { __device__ __forceinline__ long long operator()(int row) const { return row; } };

// F >= 0: compile-time flag set; F < 0: the flags of `e` at run time.  Returns the wave's max |stored value|.
// wave_tile: 32 x 64 floats of LDS private to the wave.  (m_w, n_w): first row / column of the wave's 64 x 64 sub-tile.
template <int F, int TM, int TN, class RM = LvtRowIdentity>
__device__ __forceinline__ float lvt_epi_fast_wave(const LvtEpi &e, lvt_f32x16 (&acc)[TM][TN], float *wave_tile, int m_w, int n_w,
                                                   int lane, const RM rm = RM()) {
    static_assert(TM == 2 && TN == 2, "64 x 64 sub-tiles");
    constexpr int SW = TN * 32;
    const int flags = F >= 0 ? F : e.flags;
    const bool f_bias = flags & LVT_EPI_BIAS, f_res = flags & LVT_EPI_RESIDUAL, f_relu = flags & LVT_EPI_RELU, f_mask = flags & LVT_EPI_MASK;
    const int l31 = lane & 31, half = lane >> 5;
    const int c4 = lane & 15, r0 = lane >> 4;
    const int col = n_w + 4 * c4;
    const bool colok = col < e.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f_bias && colok) bias4 = *reinterpret_cast<const float4 *>(e.bias + col);
    float am = 0.f;
    // residual / mask rows of BOTH 32-row rounds are requested up front: requested per round (round 6, first form) a cold operand --
    // the saved hidden activation that masks the FFN data gradient comes from HBM -- exposed its latency once per round (the
    // 16384 x 512 x 512 product with a mask 51 us against 37 without).  LVT_EPI_LATE_LOADS=1: the per-round form (A/B switch).
#ifndef LVT_EPI_LATE_LOADS
#define LVT_EPI_LATE_LOADS 0
#endif
    long long orow_all[TM][8];
    float4 rv_all[TM][8], mv_all[TM][8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row0 = m_w + i * 32 + r0;
#pragma unroll
        for (int u = 0; u < 8; ++u) orow_all[i][u] = rm(row0 + 4 * u);
    }
    auto request = [&](int i) {
        const int row0 = m_w + i * 32 + r0;
        if (f_res) {
            const float *rp = e.res + e.coff + col;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                rv_all[i][u] = (colok && row0 + 4 * u < e.M) ? *reinterpret_cast<const float4 *>(rp + orow_all[i][u] * e.ldr) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (f_mask) {
            const float *mp = e.mask + e.coff + col;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                mv_all[i][u] = (colok && row0 + 4 * u < e.M) ? *reinterpret_cast<const float4 *>(mp + orow_all[i][u] * e.ldm) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (!LVT_EPI_LATE_LOADS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) request(i);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) wave_tile[((r & 3) + 8 * (r >> 2) + 4 * half) * SW + 32 * j + l31] = acc[i][j][r];
        const int row0 = m_w + i * 32 + r0;                                  // rows row0 + 4 u, u = 0 .. 7
        if (LVT_EPI_LATE_LOADS) request(i);
        const long long (&orow)[8] = orow_all[i];
        const float4 (&rv)[8] = rv_all[i];
        const float4 (&mv)[8] = mv_all[i];
        float *cp = e.C + e.coff + col;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float4 v = *reinterpret_cast<const float4 *>(&wave_tile[(r0 + 4 * u) * SW + 4 * c4]);
            v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
            if (f_bias) { v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w; }
            if (f_res) { v.x += rv[u].x; v.y += rv[u].y; v.z += rv[u].z; v.w += rv[u].w; }
            if (f_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (f_mask) { v.x = mv[u].x > 0.f ? v.x : 0.f; v.y = mv[u].y > 0.f ? v.y : 0.f; v.z = mv[u].z > 0.f ? v.z : 0.f; v.w = mv[u].w > 0.f ? v.w : 0.f; }
            if (colok && row0 + 4 * u < e.M) {
                am = fmaxf(am, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
                *reinterpret_cast<float4 *>(cp + orow[u] * e.ldc) = v;
            }
        }
    }
    return am;
}

// the switch over the flag sets of the models' launches
template <int TM, int TN, class RM = LvtRowIdentity>
__device__ __forceinline__ float lvt_epi_fast_dispatch(const LvtEpi &e, lvt_f32x16 (&acc)[TM][TN], float *wave_tile, int m_w, int n_w, int lane,
                                                       const RM rm = RM()) {
#define LVT_EPI_CASE(F) case (F): return lvt_epi_fast_wave<(F), TM, TN, RM>(e, acc, wave_tile, m_w, n_w, lane, rm)
    switch (e.flags & (LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU | LVT_EPI_MASK)) {
    LVT_EPI_CASE(0);
    LVT_EPI_CASE(LVT_EPI_BIAS);
    LVT_EPI_CASE(LVT_EPI_BIAS | LVT_EPI_RELU);
    LVT_EPI_CASE(LVT_EPI_BIAS | LVT_EPI_RESIDUAL);
    LVT_EPI_CASE(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU);
    LVT_EPI_CASE(LVT_EPI_RESIDUAL);
    LVT_EPI_CASE(LVT_EPI_MASK);
    LVT_EPI_CASE(LVT_EPI_RESIDUAL | LVT_EPI_MASK);
    default: return lvt_epi_fast_wave<-1, TM, TN, RM>(e, acc, wave_tile, m_w, n_w, lane, rm);
    }
#undef LVT_EPI_CASE
}

// max |C| of the workgroup into the device scalar: `seen` is the scalar's value read by thread 0 at the start of the epilogue (a
// stale value only costs an atomic that changes nothing).  scratch: >= blockDim.x / 64 floats of LDS no wave is still using --
// callers pass a region outside the turn-tables.  All threads call; one barrier.
__device__ __forceinline__ unsigned lvt_amax_peek(const float *dst) {
    return (dst && threadIdx.x == 0) ? __hip_atomic_load(reinterpret_cast<const unsigned *>(dst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
__device__ __forceinline__ void lvt_block_amax_commit_seen(float m, float *dst, float *scratch, unsigned seen) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, scratch[w]);
        const unsigned bits = __float_as_uint(m);
        if (bits > seen) atomicMax(reinterpret_cast<unsigned *>(dst), bits);
    }
}
