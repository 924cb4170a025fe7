// Product vector quantiser kernels (reference: vidgen/modeling/vq/vq_utils.py:5-65,
// vq_embedding.py:9-99).
//
//   lvt_vq_nearest      idx[n][g][p] = argmin_k ( |e_k|^2 + |x|^2 - 2 x.e_k ), lowest k on ties
//   lvt_vq_gather       out[row][g*D+d] = E[g][idx][d]            (z_q_st, z_q_bar, mode "emb")
//   lvt_vq_ema_accumulate   stats[g][k][0..D-1] = sum of x rows with idx k, stats[g][k][D] = count (LDS-private, no atomics)
//   lvt_vq_ema_finalize     decay-lerp of running_size / running_sum, Laplace smoothing, new codebook
//
// vq_nearest, three arithmetics (include/lvt_hip.h): the f16x2 kernel further down is the default of the Python side since
// round 5 (103 us per 512 frames against 185 us for the bf16x3 kernel; profiles/r05_vq_f16x2_notes.txt).
// f32 mode: one codebook group (512 x 64 fp32 = 128 KiB) is made LDS-resident
// ([dim][code], conflict-free operand reads) by a persistent 8-wave workgroup; each wave walks
// 32-row tiles, keeps its 32x64 activation fragment in VGPRs, runs the distance product on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chain) and folds a running (min, argmin)
// per row in registers; a 5-step wave shuffle finishes the row reduction.  z_e is read exactly once
// and only the int64 indices are written: 270,336 algorithmic bytes per 64x64 frame.
#include "lvt_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VQ_D 64
#define VQ_THREADS 512

template <int KC>
__global__ __launch_bounds__(VQ_THREADS) void lvt_vq_nearest_kernel(
    const float *__restrict__ z, long long rows, int ldz, int num, const float *__restrict__ codebooks,
    long long *__restrict__ idx_out, int P, int blocks_per_group) {
    constexpr int LDC = KC + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Es = smem;                 // [VQ_D][LDC]
    float *cbsq = smem + VQ_D * LDC;  // [KC]

    const int g = blockIdx.x / blocks_per_group;
    const int bg = blockIdx.x % blocks_per_group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const float *E = codebooks + (long long)g * KC * VQ_D;

    // codebook -> LDS, transposed to [dim][code]
    for (int u = tid; u < KC * (VQ_D / 4); u += VQ_THREADS) {
        const int code = u / (VQ_D / 4), dq = u % (VQ_D / 4);
        const float4 v = *reinterpret_cast<const float4 *>(E + (long long)code * VQ_D + dq * 4);
        Es[(dq * 4 + 0) * LDC + code] = v.x;
        Es[(dq * 4 + 1) * LDC + code] = v.y;
        Es[(dq * 4 + 2) * LDC + code] = v.z;
        Es[(dq * 4 + 3) * LDC + code] = v.w;
    }
    __syncthreads();
    for (int c = tid; c < KC; c += VQ_THREADS) {
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < VQ_D; ++d) { const float e = Es[d * LDC + c]; s = fmaf(e, e, s); }
        cbsq[c] = s;
    }
    __syncthreads();

    const long long ntiles = (rows + 31) / 32;
    const int wpb = VQ_THREADS / 64;
    for (long long tile = (long long)bg * wpb + wave; tile < ntiles; tile += (long long)blocks_per_group * wpb) {
        const long long row = tile * 32 + l31;
        const bool rok = row < rows;
        // MFMA k-slot pairing: step j consumes dims (j, j+32): lanes 0-31 hold dims 0..31 of their
        // row, lanes 32-63 dims 32..63 -> 8 contiguous 16-byte loads per lane.
        float a[32];
        const float *zp = z + (rok ? row : 0) * (long long)ldz + g * VQ_D + 32 * half;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = rok ? *reinterpret_cast<const float4 *>(zp + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            a[q * 4 + 0] = v.x; a[q * 4 + 1] = v.y; a[q * 4 + 2] = v.z; a[q * 4 + 3] = v.w;
        }
        float xs = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) xs = fmaf(a[j], a[j], xs);
        xs += __shfl_xor(xs, 32);
        float xsr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xsr[r] = __shfl(xs, (r & 3) + 8 * (r >> 2) + 4 * half);

        float best[16]; int bidx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { best[r] = 3.4e38f; bidx[r] = 0; }

        const float *Erd = Es + (32 * half) * LDC + l31;
        for (int nt = 0; nt < KC / 32; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], Erd[j * LDC + nt * 32], acc, 0, 0, 0);
            const int code = nt * 32 + l31;
            const float cs = cbsq[code];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (|e|^2 + |x|^2) + (-2) * (x.e): same algebraic form as torch.addmm(beta=1, alpha=-2)
                const float dist = fmaf(-2.0f, acc[r], cs + xsr[r]);
                if (dist < best[r]) { best[r] = dist; bidx[r] = code; }
            }
        }
        // reduce over the 32 lanes (columns) of each half; ties -> lowest code (torch.min semantics)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ob = __shfl_xor(best[r], off);
                const int oi = __shfl_xor(bidx[r], off);
                if (ob < best[r] || (ob == best[r] && oi < bidx[r])) { best[r] = ob; bidx[r] = oi; }
            }
        }
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long rr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (rr < rows) idx_out[((rr / P) * num + g) * (long long)P + rr % P] = bidx[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// vq_nearest, default arithmetic (bf16x3, see gemm_engine.hip): the distance product runs on the bf16 matrix cores
// with the exact 3-way split of both operands (six MFMAs per 16-wide step, fp32 accumulation) -- 384 MFMAs of 32
// cycles per 32 rows instead of 512 of 64.  The three bf16 planes of a 512 x 64 codebook (221 KB) do not fit in
// LDS, so a workgroup holds HALF a codebook (256 codes, 108 KB) and writes the (distance, index) of its best code
// per row; a merge pass picks the winner (ties -> lower index = lower half first, torch.min semantics).
// The product is computed transposed, D^T[code][row] = E X^T: a lane then owns one activation row (lane & 31) and
// sees codes in ascending order in its accumulator registers, so the running arg-min needs no cross-lane traffic
// until one exchange with the other half-wave at the end.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define VQH_CODES 256
#define VQH_LD (VQ_D + 8)              // plane row stride (bf16): 144 B, conflict-free 16-byte fragment reads

__device__ __forceinline__ unsigned vq_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void vq_split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3) {
    p1.x = vq_cvt_pk(v.x, v.y); p1.y = vq_cvt_pk(v.z, v.w);
    const float r0 = v.x - __uint_as_float(p1.x << 16), r1 = v.y - __uint_as_float(p1.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(p1.y << 16), r3 = v.w - __uint_as_float(p1.y & 0xffff0000u);
    p2.x = vq_cvt_pk(r0, r1); p2.y = vq_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(p2.x << 16), s1 = r1 - __uint_as_float(p2.x & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(p2.y << 16), s3 = r3 - __uint_as_float(p2.y & 0xffff0000u);
    p3.x = vq_cvt_pk(s0, s1); p3.y = vq_cvt_pk(s2, s3);
}

__global__ __launch_bounds__(VQ_THREADS) void lvt_vq_nearest_half_kernel(
    const float *__restrict__ z, long long rows, int ldz, int KC, const float *__restrict__ codebooks,
    float *__restrict__ pbest, int *__restrict__ pidx) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[3 * VQH_CODES * VQH_LD];
    __shared__ float cbsq[VQH_CODES];
    constexpr int PL = VQH_CODES * VQH_LD;
    const int g = blockIdx.z, part = blockIdx.y, nparts = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int code0 = part * VQH_CODES;
    const float *E = codebooks + ((long long)g * KC + code0) * VQ_D;

    // stage this half of the codebook as three bf16 planes [code][dim]; thread (code = tid >> 1, dims 32*(tid&1)..)
    {
        const int code = tid >> 1, d0 = 32 * (tid & 1);
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = *reinterpret_cast<const float4 *>(E + (long long)code * VQ_D + d0 + 4 * u);
            sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
            uint2 p1, p2, p3;
            vq_split4(v, p1, p2, p3);
            unsigned short *dst = planes + code * VQH_LD + d0 + 4 * u;
            *reinterpret_cast<uint2 *>(dst) = p1;
            *reinterpret_cast<uint2 *>(dst + PL) = p2;
            *reinterpret_cast<uint2 *>(dst + 2 * PL) = p3;
        }
        sq += __shfl_xor(sq, 1);
        if ((tid & 1) == 0) cbsq[code] = sq;
    }
    __syncthreads();

    const long long ntiles = (rows + 31) / 32;
    const int wpb = VQ_THREADS / 64;
    for (long long tile = (long long)blockIdx.x * wpb + wave; tile < ntiles; tile += (long long)gridDim.x * wpb) {
        const long long row = tile * 32 + l31;
        const bool rok = row < rows;
        // B operand: the lane's row, dims 16s + 8*half .. +7 for the four 16-wide steps, as three planes
        bf16x8 zb[4][3];
        float xs = 0.f;
        {
            const float *zp = z + (rok ? row : 0) * (long long)ldz + g * VQ_D + 8 * half;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 lo = rok ? *reinterpret_cast<const float4 *>(zp + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 hi = rok ? *reinterpret_cast<const float4 *>(zp + 16 * s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                xs = fmaf(lo.x, lo.x, xs); xs = fmaf(lo.y, lo.y, xs); xs = fmaf(lo.z, lo.z, xs); xs = fmaf(lo.w, lo.w, xs);
                xs = fmaf(hi.x, hi.x, xs); xs = fmaf(hi.y, hi.y, xs); xs = fmaf(hi.z, hi.z, xs); xs = fmaf(hi.w, hi.w, xs);
                uint2 a1, a2, a3, b1, b2, b3;
                vq_split4(lo, a1, a2, a3); vq_split4(hi, b1, b2, b3);
                const uint4 u1 = make_uint4(a1.x, a1.y, b1.x, b1.y), u2 = make_uint4(a2.x, a2.y, b2.x, b2.y),
                            u3 = make_uint4(a3.x, a3.y, b3.x, b3.y);
                zb[s][0] = *reinterpret_cast<const bf16x8 *>(&u1); zb[s][1] = *reinterpret_cast<const bf16x8 *>(&u2);
                zb[s][2] = *reinterpret_cast<const bf16x8 *>(&u3);
            }
            xs += __shfl_xor(xs, 32);
        }
        float best = 3.4e38f; int bidx = 0;
        // two code tiles advance together, term by term (independent accumulators)
        for (int ct = 0; ct < VQH_CODES / 32; ct += 2) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a0[3], a1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a0[pl] = *reinterpret_cast<const bf16x8 *>(planes + pl * PL + (ct * 32 + l31) * VQH_LD + 16 * s + 8 * half);
                    a1[pl] = *reinterpret_cast<const bf16x8 *>(planes + pl * PL + (ct * 32 + 32 + l31) * VQH_LD + 16 * s + 8 * half);
                }
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                for (int tm = 0; tm < 6; ++tm) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[TA[tm]], zb[s][TB[tm]], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[TA[tm]], zb[s][TB[tm]], acc1, 0, 0, 0);
                }
            }
            // codes ascend with (tile, r): a strict < keeps the lowest index among equal distances
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                // (|e|^2 + |x|^2) + (-2) * (x.e): same algebraic form as torch.addmm(beta=1, alpha=-2)
                const float dist = fmaf(-2.0f, acc0[r], cbsq[cl] + xs);
                if (dist < best) { best = dist; bidx = code0 + cl; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = ct * 32 + 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float dist = fmaf(-2.0f, acc1[r], cbsq[cl] + xs);
                if (dist < best) { best = dist; bidx = code0 + cl; }
            }
        }
        // the other half-wave saw the other codes of every tile
        const float ob = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bidx, 32);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        if (half == 0 && rok) {
            const long long o = ((long long)g * nparts + part) * rows + row;
            pbest[o] = best; pidx[o] = bidx;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Coarse-then-exact search (round 3; opt-in per call: LVT_VQ_COARSE in `flags`).  Nothing forces a full-precision scan of all 512 codes:
//   1. COARSE: score~_k = x~ . e~_k - |e_k|^2 / 2 with x~ = bf16(x), e~ = bf16(e): ONE bf16 MFMA per 16 dims instead of the
//      six of the exact split (the -|e|^2/2 term rides in as the fp32 initial accumulator).  RNE bf16 has a relative error of
//      at most 2^-9 per operand, so |score~_k - score_k| <= eps = (2^-8 + 2^-18) sum_d |x_d e_kd| <= 1.001 * 2^-8 |x| |e_k|.
//   2. Every code whose coarse score is within 2 eps (+ the fp32 evaluation noise of the exact distance) of the coarse
//      maximum is a CANDIDATE; the true nearest code always is one (score~_best >= score_best - eps >= score_j - eps >=
//      score~_j - 2 eps for every j).  The scores are recomputed by a second pass of the same MFMAs (bit-identical) instead of
//      being kept: 128 MFMAs per 32 rows in total against 384, and no 512-entry register file per row.
//   3. EXACT: the candidates (typically 1-3 per row, at most 8 per half-wave, else the row's tile takes the exhaustive path) are
//      evaluated as fp32 FMA chains dist = (|e|^2 + |x|^2) - 2 x.e, the reference's algebraic form (vq_utils.py:13-20), lowest
//      index on equal distances (torch.min).  The argmin is exact: which fp32 summation order decides a sub-rounding near-tie
//      is implementation-defined in the reference as well (tests/util_models.py:margin_ok).
// The whole codebook group fits as one bf16 plane (72 KB): one workgroup per (group, row range), no half / merge passes, and
// the indices are written directly in their final layout.
// Measured (1x MI355X, 512 frames): 176 us on well-separated codebooks against 190 us for the full scan -- the MFMA count fell 3x
// but the band test costs ~25 k vector instructions per wave, which is the new bound -- and 410-520 us on the DEGENERATE codebook
// of a freshly initialised EMA quantiser (|e| ~ 500-1000 against |x| ~ 0.1: every row's band holds 150-250 codes, every tile
// takes the exhaustive path).  The full scan does not depend on the data, so it stays the default.
// ------------------------------------------------------------------------------------------------
#define VQC_LD (VQ_D + 8)              // plane row stride (bf16): 144 B, conflict-free 16-byte fragment reads
#define VQC_MAXC 8                     // candidates kept per (row, half-wave); measured: 1.3-1.6 per ROW on average, > 8 per half ~never

template <int KC>
__global__ __launch_bounds__(VQ_THREADS) void lvt_vq_nearest_coarse_kernel(
    const float *__restrict__ z, long long rows, int ldz, const float *__restrict__ codebooks, long long *__restrict__ idx_out,
    int P, int num) {
    __shared__ __attribute__((aligned(16))) unsigned short plane[KC * VQC_LD];
    __shared__ __attribute__((aligned(16))) float cbsq[KC], hcb[KC];       // |e|^2 (FMA chain), -|e|^2 / 2
    __shared__ float emax2_s;
    __shared__ int cand[VQ_THREADS / 64][32][2][VQC_MAXC];
    const int g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const float *E = codebooks + (long long)g * KC * VQ_D;

    // stage the codebook group as ONE bf16 plane [code][dim]; thread (code = tid >> 1 (+256), dims 32 * (tid & 1) ..)
    for (int code = tid >> 1; code < KC; code += VQ_THREADS / 2) {
        const int d0 = 32 * (tid & 1);
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = *reinterpret_cast<const float4 *>(E + (long long)code * VQ_D + d0 + 4 * u);
            sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
            uint2 p1;
            p1.x = vq_cvt_pk(v.x, v.y); p1.y = vq_cvt_pk(v.z, v.w);
            *reinterpret_cast<uint2 *>(plane + code * VQC_LD + d0 + 4 * u) = p1;
        }
        sq += __shfl_xor(sq, 1);            // (dims 0..31) + (dims 32..63), the order of the exact phase below
        if ((tid & 1) == 0) { cbsq[code] = sq; hcb[code] = -0.5f * sq; }
    }
    __syncthreads();
    if (wave == 0) {
        float m = 0.f;
        for (int c = lane; c < KC; c += 64) m = fmaxf(m, cbsq[c]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) emax2_s = m;
    }
    __syncthreads();
    const float emax2 = emax2_s, emax = sqrtf(emax2);

    const long long ntiles = (rows + 31) / 32;
    const int wpb = VQ_THREADS / 64;
    for (long long tile = (long long)blockIdx.x * wpb + wave; tile < ntiles; tile += (long long)gridDim.x * wpb) {
        const long long row = tile * 32 + l31;
        const bool rok = row < rows;
        const float *zrow = z + (rok ? row : 0) * (long long)ldz + g * VQ_D;
        // exact-phase copy of the row: dims 32 half .. 32 half + 31 in fp32; |x|^2 as (dims 0..31) + (dims 32..63)
        float xr[32];
        float xs = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = rok ? *reinterpret_cast<const float4 *>(zrow + 32 * half + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
            xr[4 * u] = v.x; xr[4 * u + 1] = v.y; xr[4 * u + 2] = v.z; xr[4 * u + 3] = v.w;
            xs = fmaf(v.x, v.x, xs); xs = fmaf(v.y, v.y, xs); xs = fmaf(v.z, v.z, xs); xs = fmaf(v.w, v.w, xs);
        }
        xs += __shfl_xor(xs, 32);
        // coarse B operand: bf16(x), dims 16 s + 8 half .. + 7
        bf16x8 zb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 lo = rok ? *reinterpret_cast<const float4 *>(zrow + 16 * s + 8 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 hi = rok ? *reinterpret_cast<const float4 *>(zrow + 16 * s + 8 * half + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint4 u1 = make_uint4(vq_cvt_pk(lo.x, lo.y), vq_cvt_pk(lo.z, lo.w), vq_cvt_pk(hi.x, hi.y), vq_cvt_pk(hi.z, hi.w));
            zb[s] = *reinterpret_cast<const bf16x8 *>(&u1);
        }
        // FOUR 32 x 32 tiles of coarse scores at a time (rows = codes ct*32 .., columns = the wave's activation rows): four
        // independent accumulator chains keep the matrix pipe issuing back to back and sixteen LDS reads are in flight together
        // (one tile at a time, each of the four k-steps waited for its predecessor and for its own LDS round trip)
        auto coarse_tiles = [&](int ct, f32x16 (&acc)[4]) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 h4 = *reinterpret_cast<const float4 *>(&hcb[(ct + t4) * 32 + 8 * q + 4 * half]);
                    acc[t4][4 * q] = h4.x; acc[t4][4 * q + 1] = h4.y; acc[t4][4 * q + 2] = h4.z; acc[t4][4 * q + 3] = h4.w;
                }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    a[t4] = *reinterpret_cast<const bf16x8 *>(plane + ((ct + t4) * 32 + l31) * VQC_LD + 16 * s + 8 * half);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) acc[t4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t4], zb[s], acc[t4], 0, 0, 0);
            }
        };
        // ---- pass 1: coarse maximum ----
        float m = -3.4e38f;
        for (int ct = 0; ct < KC / 32; ct += 4) {
            f32x16 acc[4];
            coarse_tiles(ct, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(fmaxf(acc[0][r], acc[1][r]), fmaxf(acc[2][r], acc[3][r])));
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        // band: 2 eps of the coarse product + the fp32 rounding of the coarse accumulation and of the exact distances (both << 1e-5 (|x| + |e|)^2)
        const float xn = sqrtf(xs);
        const float thr = rok ? m - (2.0f * 1.002f * 0.00390625f * xn * emax + 1e-5f * (xs + emax2 + 2.f * xn * emax)) : 3.4e38f;
        // ---- pass 2: the same scores again; codes inside the band are appended to the lane's candidate list.  One test per
        // tile (its maximum) guards the sixteen per-code tests: most tiles hold no candidate of a given lane ----
        int cnt = 0;
        for (int ct = 0; ct < KC / 32; ct += 4) {
            f32x16 acc[4];
            coarse_tiles(ct, acc);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                float tm = acc[t4][0];
#pragma unroll
                for (int r = 1; r < 16; r += 3) tm = fmaxf(tm, fmaxf(fmaxf(acc[t4][r], acc[t4][r + 1]), acc[t4][r + 2]));
                if (tm >= thr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (acc[t4][r] >= thr) {
                            if (cnt < VQC_MAXC) cand[wave][l31][half][cnt] = (ct + t4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            ++cnt;
                        }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                                   // lgkmcnt(0): the list stores have landed (LDS is in order per wave)
        __builtin_amdgcn_wave_barrier();
        const int cnt_o = __shfl_xor(cnt, 32);
        const int c0 = half ? cnt_o : cnt, c1 = half ? cnt : cnt_o;            // entries of half 0 / half 1 for this row
        const bool overflow = c0 > VQC_MAXC || c1 > VQC_MAXC;
        // ---- exact phase: both half-waves walk the row's candidates together (32 dims each), two candidates per trip so
        // that their code rows (L2 hits, ~1 us away) are in flight together ----
        float best = 3.4e38f; int bidx = 0x7fffffff;
        const int total = overflow ? 0 : c0 + c1;
        int tmax = total;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) tmax = max(tmax, __shfl_xor(tmax, o));
        for (int j = 0; j < tmax; j += 2) {
            int code[2]; float4 ev[2][8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int jj = j + q < total ? j + q : (total > 0 ? total - 1 : 0);        // (clamped: a duplicate changes nothing)
                code[q] = total > 0 ? (jj < c0 ? cand[wave][l31][0][jj] : cand[wave][l31][1][jj - c0]) : 0;
                const float *e = E + (long long)code[q] * VQ_D + 32 * half;
#pragma unroll
                for (int u = 0; u < 8; ++u) ev[q][u] = *reinterpret_cast<const float4 *>(e + 4 * u);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float dot = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dot = fmaf(xr[4 * u], ev[q][u].x, dot); dot = fmaf(xr[4 * u + 1], ev[q][u].y, dot);
                    dot = fmaf(xr[4 * u + 2], ev[q][u].z, dot); dot = fmaf(xr[4 * u + 3], ev[q][u].w, dot);
                }
                dot += __shfl_xor(dot, 32);
                const float dist = fmaf(-2.0f, dot, cbsq[code[q]] + xs);
                if (j < total && (dist < best || (dist == best && code[q] < bidx))) { best = dist; bidx = code[q]; }
            }
        }
        // ---- a row whose band holds more than VQC_MAXC codes per half-wave (many codes (almost) equally near: dead or
        // duplicated codebook entries, e.g. the all-equal codes of an EMA codebook that has just been initialised): the whole
        // tile takes the exhaustive EXACT scan -- all codes, bf16x3 products (six MFMAs per step, code rows split on the fly
        // from the L2-resident fp32 codebook), running arg-min in ascending code order.  Worst case = the cost of the full
        // scan of rounds 1-2; an ordinary codebook never gets here.
        if (__builtin_amdgcn_ballot_w64(overflow) != 0) {
            bf16x8 zb3[4][3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 lo = rok ? *reinterpret_cast<const float4 *>(zrow + 16 * s + 8 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 hi = rok ? *reinterpret_cast<const float4 *>(zrow + 16 * s + 8 * half + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 a1, a2, a3, b1, b2, b3;
                vq_split4(lo, a1, a2, a3); vq_split4(hi, b1, b2, b3);
                const uint4 u1 = make_uint4(a1.x, a1.y, b1.x, b1.y), u2 = make_uint4(a2.x, a2.y, b2.x, b2.y), u3 = make_uint4(a3.x, a3.y, b3.x, b3.y);
                zb3[s][0] = *reinterpret_cast<const bf16x8 *>(&u1); zb3[s][1] = *reinterpret_cast<const bf16x8 *>(&u2);
                zb3[s][2] = *reinterpret_cast<const bf16x8 *>(&u3);
            }
            best = 3.4e38f; bidx = 0;
            for (int ct = 0; ct < KC / 32; ct += 2) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 a0[3], a1[3];
                    {
                        const float *e0 = E + (long long)(ct * 32 + l31) * VQ_D + 16 * s + 8 * half, *e1 = e0 + 32 * VQ_D;
                        uint2 p1, p2, p3, q1, q2, q3;
                        vq_split4(*reinterpret_cast<const float4 *>(e0), p1, p2, p3); vq_split4(*reinterpret_cast<const float4 *>(e0 + 4), q1, q2, q3);
                        uint4 u1 = make_uint4(p1.x, p1.y, q1.x, q1.y), u2 = make_uint4(p2.x, p2.y, q2.x, q2.y), u3 = make_uint4(p3.x, p3.y, q3.x, q3.y);
                        a0[0] = *reinterpret_cast<const bf16x8 *>(&u1); a0[1] = *reinterpret_cast<const bf16x8 *>(&u2); a0[2] = *reinterpret_cast<const bf16x8 *>(&u3);
                        vq_split4(*reinterpret_cast<const float4 *>(e1), p1, p2, p3); vq_split4(*reinterpret_cast<const float4 *>(e1 + 4), q1, q2, q3);
                        u1 = make_uint4(p1.x, p1.y, q1.x, q1.y); u2 = make_uint4(p2.x, p2.y, q2.x, q2.y); u3 = make_uint4(p3.x, p3.y, q3.x, q3.y);
                        a1[0] = *reinterpret_cast<const bf16x8 *>(&u1); a1[1] = *reinterpret_cast<const bf16x8 *>(&u2); a1[2] = *reinterpret_cast<const bf16x8 *>(&u3);
                    }
                    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                    for (int tm = 0; tm < 6; ++tm) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[TA[tm]], zb3[s][TB[tm]], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[TA[tm]], zb3[s][TB[tm]], acc1, 0, 0, 0);
                    }
                }
                // codes ascend with (tile, r): a strict < keeps the lowest index among equal distances
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float dist = fmaf(-2.0f, acc0[r], cbsq[cl] + xs);
                    if (dist < best) { best = dist; bidx = cl; }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = ct * 32 + 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float dist = fmaf(-2.0f, acc1[r], cbsq[cl] + xs);
                    if (dist < best) { best = dist; bidx = cl; }
                }
            }
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bidx, 32);
            if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (half == 0 && rok) idx_out[((row / P) * num + g) * (long long)P + row % P] = bidx;
        __builtin_amdgcn_wave_barrier();                                       // the lists are reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// vq_nearest, f16x2 arithmetic (round 5; LVT_MATH_F16X2 in `flags`, the default mode of the Python side).
//   argmin_k ( |e_k|^2 + |x|^2 - 2 x.e_k )  ==  argmax_k ( x.e_k - |e_k|^2 / 2 ):   |x|^2 is common to a row's 512 candidates,
//   so it is dropped instead of being added and rounded 512 times (the score's rounding is relative to |x.e| and |e|^2 only).
// Operands: the codebook group as TWO fp16 planes under ONE power-of-two scale sE (max |e| sE in [2^14, 2^15)): hi = fp16(e sE),
// lo = fp16(e sE - hi), 22 significant bits down to 2^-17 of the group's maximum; an activation row likewise under its OWN
// scale sx.  x.e sE sx = lo hi + hi lo + hi hi: three v_mfma_f32_32x32x16_f16 per 16 dims into one fp32 accumulator (six bf16
// MFMAs in the bf16x3 kernel above), and score = fma(-|e|^2 sE / 2, sx, acc): one FMA, one compare, one max, one select per
// candidate.  2 x 72 KB of planes hold the WHOLE group: one workgroup per (group, row range), no half / merge passes, indices
// written in their final layout.  Equal scores (duplicated codes give bit-identical planes, norms and accumulation orders)
// resolve to the lowest index.
// Schedule (what the measurements of round 5 asked for, profiles/r05_vq_f16x2_notes.txt):
//   * the two waves of a SIMD run in lock step (same code, no barrier in the scan), so a wave that issues its MFMAs in one run
//     and its selects in another leaves the matrix pipe idle during the selects of BOTH: the products of code tile i + 1 are
//     issued BESIDE the selects of tile i (two accumulators), in a hand-fixed order -- four groups of {2 fragment reads,
//     3 MFMAs, 4 FMAs, one 12-instruction select block}, each closed by a sched_barrier;
//   * a wave owns 32 rows at a time and the raw rows of its NEXT tile are requested right after the split of the current one:
//     z is read once, 134 MB per 512 frames, ~26 us of HBM time that otherwise sits in front of every scan (all waves load at
//     once); 32-row tiles (not 64) leave the registers for that (32 raw + 32 operand + 32 accumulator + 64 fragment).
// ------------------------------------------------------------------------------------------------
typedef _Float16 vqf_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vqf_h2 __attribute__((ext_vector_type(2)));
typedef float vqf_f2 __attribute__((ext_vector_type(2)));
typedef unsigned vqf_u4 __attribute__((ext_vector_type(4)));
#define VQF_LD (VQ_D + 8)              // plane row stride (fp16): 144 B, conflict-free 16-byte fragment reads

__device__ __forceinline__ float vqf_mix_lo(unsigned h, float c) {        // c - half(h.lo), exact
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.f), "v"(c));
    return r;
}
__device__ __forceinline__ float vqf_mix_hi(unsigned h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.f), "v"(c));
    return r;
}
// (a, b) * s -> packed fp16 hi pair and lo pair
__device__ __forceinline__ void vqf_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const vqf_f2 t = vqf_f2{a, b} * s;
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(t, vqf_h2));
    const vqf_f2 r = {vqf_mix_lo(ph, t.x), vqf_mix_hi(ph, t.y)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, vqf_h2));
}
__device__ __forceinline__ float vqf_absmax4(const float4 a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); }
__device__ __forceinline__ int vqf_ebits(float nonneg) { return (int)((__float_as_uint(nonneg) >> 23) & 0xffu); }
__device__ __forceinline__ float vqf_pow2(int field) { return __uint_as_float((unsigned)(field < 1 ? 1 : (field > 254 ? 254 : field)) << 23); }

// Two consecutive candidate codes of one row against its running maximum `lb` (position of the maximum inside the code tile:
// `lp`): compare, max, select per candidate.  Written out because the ORDER is the point: a VALU write of a lane mask needs two
// wait states before a VALU reads it (gfx940+), and the compiler's own selection (VCC for every compare, an s_nop before every
// select) spends more issue slots on s_nop than on MFMAs.  Here every select sits two instructions behind its compare.
// A strict > keeps the lowest position among equal scores.  POS, POS + 1 are inline constants (0 .. 64).
template <int POS>
__device__ __forceinline__ void vqf_select2(float s0, float s1, float &lb, int &lp) {
#ifdef VQF_X_NOSEL
    lb = fmaxf(lb, s0 + s1); return;
#endif
    unsigned long long m0, m1;
    asm("v_cmp_gt_f32_e64 %2, %4, %0\n\t"
        "v_max_f32_e32 %0, %0, %4\n\t"
        "v_cmp_gt_f32_e64 %3, %5, %0\n\t"
        "v_cndmask_b32_e64 %1, %1, %6, %2\n\t"
        "v_max_f32_e32 %0, %0, %5\n\t"
        "v_cndmask_b32_e64 %1, %1, %7, %3"
        : "+v"(lb), "+v"(lp), "=&s"(m0), "=&s"(m1)
        : "v"(s0), "v"(s1), "n"(POS), "n"(POS + 1));
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void vqf_static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); vqf_static_for<N, I + 1>(f); }
}

template <int KC>
__global__ __launch_bounds__(VQ_THREADS) void lvt_vq_nearest_f16x2_kernel(
    const float *__restrict__ z, long long rows, int ldz, const float *__restrict__ codebooks, long long *__restrict__ idx_out,
    int P, int num) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vqf_raw[];
    constexpr int PL = KC * VQF_LD;
    unsigned short *planes = reinterpret_cast<unsigned short *>(vqf_raw);            // [2][KC][VQF_LD]: hi, lo
    float *hcb = reinterpret_cast<float *>(planes + 2 * PL);                         // [KC]  -|e|^2 sE / 2
    float *red = hcb + KC;                                                           // [16]
    const int g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const float *E = codebooks + (long long)g * KC * VQ_D;

    // a wave's 32-row tiles; the raw rows of the first one are requested before the codebook is staged.  A lane holds dims
    // 16 s + 8 half .. + 7 (s = 0 .. 3) of row tile * 32 + l31: the B operand layout of the four 16-wide steps.  (Rows past the
    // end are clamped to the last row: their columns of the product are computed and not stored.)
    const long long ntiles = (rows + 31) / 32;
    const long long tile0 = (long long)blockIdx.x * (VQ_THREADS / 64) + wave, tstride = (long long)gridDim.x * (VQ_THREADS / 64);
    float4 raw[8];
    auto load_rows = [&](long long tile) {
        const long long row = tile * 32 + l31;
        const float *zp = z + (row < rows ? row : rows - 1) * (long long)ldz + g * VQ_D + 8 * half;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            raw[2 * s] = *reinterpret_cast<const float4 *>(zp + 16 * s);
            raw[2 * s + 1] = *reinterpret_cast<const float4 *>(zp + 16 * s + 4);
        }
    };
    if (tile0 < ntiles) load_rows(tile0);

    // ---- stage the group: thread (code = tid >> 1 (+ 256), dims 32 (tid & 1) ..); the group's max |e| first ----
    {
        constexpr int IT = KC > 256 ? KC / 256 : 1;
        float4 ev[IT][8];
        float sq[IT];
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int code = ((tid >> 1) + 256 * it) & (KC - 1);                     // (KC = 128: threads 256 .. repeat codes 0 ..)
            sq[it] = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ev[it][u] = *reinterpret_cast<const float4 *>(E + (long long)code * VQ_D + 32 * (tid & 1) + 4 * u);
                const float4 v = ev[it][u];
                sq[it] = fmaf(v.x, v.x, sq[it]); sq[it] = fmaf(v.y, v.y, sq[it]); sq[it] = fmaf(v.z, v.z, sq[it]); sq[it] = fmaf(v.w, v.w, sq[it]);
                m = fmaxf(m, vqf_absmax4(v));
            }
            sq[it] += __shfl_xor(sq[it], 1);        // (dims 0..31) + (dims 32..63)
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < VQ_THREADS / 64; ++w) m = fmaxf(m, red[w]);
        const float sE = vqf_pow2(268 - vqf_ebits(m));                               // max |e| sE in [2^14, 2^15)
        float hm = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int code = ((tid >> 1) + 256 * it) & (KC - 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint2 h, l;
                vqf_split_pair(ev[it][u].x, ev[it][u].y, sE, h.x, l.x);
                vqf_split_pair(ev[it][u].z, ev[it][u].w, sE, h.y, l.y);
                unsigned short *dst = planes + code * VQF_LD + 32 * (tid & 1) + 4 * u;
                *reinterpret_cast<uint2 *>(dst) = h;
                *reinterpret_cast<uint2 *>(dst + PL) = l;
            }
            const float h = -0.5f * sq[it] * sE;
            if ((tid & 1) == 0) hcb[code] = h;
            hm = fmaxf(hm, fabsf(h));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) hm = fmaxf(hm, __shfl_xor(hm, o));
        if (lane == 0) red[8 + wave] = hm;
        __syncthreads();
    }
    float hm = red[8];
#pragma unroll
    for (int w = 1; w < VQ_THREADS / 64; ++w) hm = fmaxf(hm, red[8 + w]);
    // a row's scale is capped so that |e|^2 sE sx / 2 stays below 2^100 (a row that is (nearly) zero against a large codebook)
    const int sx_cap = 353 - vqf_ebits(hm);

    // lane bases of the fragment reads (plane 1 lies 72 KB behind plane 0: beyond a DS offset field) and of the norm terms
    const unsigned short *fr0 = planes + l31 * VQF_LD + 8 * half, *fr1 = fr0 + PL;
    const float *hb = hcb + 4 * half;
    constexpr int NT = KC / 32;                                                      // 16, 8 or 4 code tiles
    constexpr std::true_type yes{};
    constexpr std::false_type no{};

    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        // B operand of the tile: hi and lo fragments of the four steps under the row's scale
        vqf_h8 zb[4][2];
        float sx;
        {
            float xm = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) xm = fmaxf(xm, vqf_absmax4(raw[u]));
            xm = fmaxf(xm, __shfl_xor(xm, 32));
            const int f = 268 - vqf_ebits(xm);
            sx = vqf_pow2(f < sx_cap ? f : sx_cap);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                vqf_u4 uh, ul;
                unsigned a, b;
                vqf_split_pair(raw[2 * s].x, raw[2 * s].y, sx, a, b); uh[0] = a; ul[0] = b;
                vqf_split_pair(raw[2 * s].z, raw[2 * s].w, sx, a, b); uh[1] = a; ul[1] = b;
                vqf_split_pair(raw[2 * s + 1].x, raw[2 * s + 1].y, sx, a, b); uh[2] = a; ul[2] = b;
                vqf_split_pair(raw[2 * s + 1].z, raw[2 * s + 1].w, sx, a, b); uh[3] = a; ul[3] = b;
                zb[s][0] = __builtin_bit_cast(vqf_h8, uh);
                zb[s][1] = __builtin_bit_cast(vqf_h8, ul);
            }
        }
#ifndef VQF_X_NOROWS
        if (tile + tstride < ntiles) load_rows(tile + tstride);                      // in flight during the scan
#endif

        float best = -INFINITY, lb = 0.f;
        int bidx = 0, lp = 0;
        float4 hnext = *reinterpret_cast<const float4 *>(hb);                         // -|e|^2 sE / 2 of the next four candidates
        // One STEP: the 12 MFMAs of code tile cp (accP) beside the selects of tile cs (accS, left by the previous step) and the
        // reads of the fragments of tile cn (afN, multiplied by the next step).
        auto step = [&](auto has_prod, auto has_sel, auto has_next, int cp, f32x16 &accP, const vqf_h8 (&afP)[4][2], int cs,
                        const f32x16 &accS, int cn, vqf_h8 (&afN)[4][2]) {
            constexpr bool PROD = decltype(has_prod)::value, SEL = decltype(has_sel)::value, NEXT = decltype(has_next)::value;
            if constexpr (SEL) { lb = best; lp = 0; }
            const unsigned short *f0 = fr0 + cn * 32 * VQF_LD, *f1 = fr1 + cn * 32 * VQF_LD;
            const float *hs = hb + cs * 32, *hp = hb + cp * 32;
            vqf_static_for<4>([&](auto ig) {
                constexpr int G = decltype(ig)::value;
                // The three MFMAs of a group are a dependent chain (one accumulator): a select block behind the first and the second,
                // the LDS reads behind the third, each piece pinned by a sched_barrier (the compiler's order is MFMA MFMA MFMA, then
                // all of the vector work: the wave then waits out two MFMA latencies with nothing to issue).
                const float4 h4 = hnext;
                if constexpr (PROD) {                                                 // smallest terms first: lo hi, hi lo, hi hi
                    if constexpr (G == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(afP[G][1], zb[G][0], zero, 0, 0, 0);
                    } else {
                        accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(afP[G][1], zb[G][0], accP, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (the norm terms of the next group are requested now: a just-in-time read stalls the wave for the LDS round trip)
                if constexpr (SEL && G < 3) hnext = *reinterpret_cast<const float4 *>(hs + 8 * (G + 1));
                if constexpr (PROD && G == 3) hnext = *reinterpret_cast<const float4 *>(hp);
                if constexpr (SEL) {                                                  // accumulator registers 4 G .. + 3 <-> codes 8 G .. + 3 (+ 4 half)
                    vqf_select2<8 * G>(fmaf(h4.x, sx, accS[4 * G]), fmaf(h4.y, sx, accS[4 * G + 1]), lb, lp);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PROD) {
                    accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(afP[G][0], zb[G][1], accP, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (SEL) {
                    vqf_select2<8 * G + 2>(fmaf(h4.z, sx, accS[4 * G + 2]), fmaf(h4.w, sx, accS[4 * G + 3]), lb, lp);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PROD) {
                    accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(afP[G][0], zb[G][0], accP, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef VQF_X_NOFRAG
                if constexpr (NEXT) if (cn == 0) {
#else
                if constexpr (NEXT) {
#endif
                    afN[G][0] = *reinterpret_cast<const vqf_h8 *>(f0 + 16 * G);
                    afN[G][1] = *reinterpret_cast<const vqf_h8 *>(f1 + 16 * G);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (SEL) {
                const bool imp = lb > best;
                best = lb;
                bidx = imp ? cs * 32 + lp : bidx;                                     // (+ 4 half: added once at the end)
            }
        };
        f32x16 accA, accB;
        vqf_h8 afA[4][2], afB[4][2];
        step(no, no, yes, 0, accA, afA, 0, accA, 0, afA);                             // fragments of tile 0
        step(yes, no, yes, 0, accA, afA, 0, accA, 1, afB);                            // products of tile 0, fragments of tile 1
#pragma unroll 1
        for (int ct = 1; ct < NT - 1; ct += 2) {
            step(yes, yes, yes, ct, accB, afB, ct - 1, accA, ct + 1, afA);
            step(yes, yes, yes, ct + 1, accA, afA, ct, accB, ct + 2 < NT ? ct + 2 : NT - 1, afB);
        }
        step(yes, yes, no, NT - 1, accB, afB, NT - 2, accA, 0, afA);
        step(no, yes, no, 0, accB, afB, NT - 1, accB, 0, afA);

        // the other half-wave saw the other codes of every tile
        bidx += 4 * half;
        const float ob = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bidx, 32);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        const long long row = tile * 32 + l31;
        if (half == 0 && row < rows) idx_out[((row / P) * num + g) * (long long)P + row % P] = bidx;
    }
}

__global__ void lvt_vq_nearest_merge_kernel(const float *__restrict__ pbest, const int *__restrict__ pidx, long long rows,
                                            int num, int nparts, int P, long long *__restrict__ idx_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * num) return;
    const int g = (int)(i / rows); const long long row = i % rows;
    float best = pbest[((long long)g * nparts) * rows + row]; int bi = pidx[((long long)g * nparts) * rows + row];
    for (int p = 1; p < nparts; ++p) {
        const float b = pbest[((long long)g * nparts + p) * rows + row];
        if (b < best) { best = b; bi = pidx[((long long)g * nparts + p) * rows + row]; }      // strict: lower half wins ties
    }
    idx_out[((row / P) * num + g) * (long long)P + row % P] = bi;
}

// out[row][g*D + d] = E[g][idx[n][g][p]][d]  (D == 64).  16 lanes move one (row, g) slice as float4, so a wave
// writes 1 KB of consecutive output per step (a whole row when num == 4) and every lane keeps GATHER_UNROLL
// independent index -> codebook -> store chains in flight; one lane per float with a single chain per wave ran at
// 1.75 TB/s of the 134 MB it writes.
#define GATHER_UNROLL 4
__global__ __launch_bounds__(256) void lvt_vq_gather_kernel(const long long *__restrict__ idx,
                                                            const float *__restrict__ codebooks, long long rows, int num,
                                                            int KC, int P, float *__restrict__ out, int ldo) {
    const long long pairs = rows * num;
    const int q = threadIdx.x & 15;                                   // float4 within the 64-float slice
    const long long stride = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long pr = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; pr < pairs; pr += stride * GATHER_UNROLL) {
        long long k[GATHER_UNROLL], row[GATHER_UNROLL]; int g[GATHER_UNROLL];
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u) {
            const long long pu = pr + stride * u;
            row[u] = pu / num; g[u] = (int)(pu % num);
            k[u] = pu < pairs ? idx[((row[u] / P) * num + g[u]) * (long long)P + row[u] % P] : 0;
        }
        float4 v[GATHER_UNROLL];
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u)
            v[u] = *reinterpret_cast<const float4 *>(codebooks + ((long long)g[u] * KC + k[u]) * VQ_D + q * 4);
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u)
            if (pr + stride * u < pairs) *reinterpret_cast<float4 *>(out + row[u] * ldo + g[u] * VQ_D + q * 4) = v[u];
    }
}

// EMA statistics: stats[g][k][0..D-1] = sum of the rows assigned to code k, stats[g][k][D] = their count.
// A workgroup owns one (group, row chunk) and keeps a PRIVATE 512 x 65 accumulator in LDS (133 KB of the
// 160 KB).  Wave w only accumulates rows whose code is congruent to w mod 8, walking them in row order:
// no two waves ever touch the same accumulator row, no atomics are needed, and the summation order is a
// pure function of the data (bit-reproducible).  Chunk partials are then summed in chunk order.
#define EMA_THREADS 512
template <int KC>
__global__ __launch_bounds__(EMA_THREADS) void lvt_vq_ema_partial_kernel(
    const long long *__restrict__ idx, const float *__restrict__ z, long long rows, int ldz, int num, int P,
    long long rows_per_chunk, int nchunks, float *__restrict__ partial) {
    constexpr int LDS_LD = VQ_D + 1;
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [KC][65], then a 64-row staging tile
    float *tile = acc + ((KC * LDS_LD + 3) & ~3);                   // [64 rows][VQ_D]
    const int g = blockIdx.x / nchunks, c = blockIdx.x % nchunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < KC * LDS_LD; i += EMA_THREADS) acc[i] = 0.f;
    const long long r0 = (long long)c * rows_per_chunk;
    const long long r1 = min(rows, r0 + rows_per_chunk);
    // 64 rows at a time: the workgroup fetches them coalesced (16 lanes per 256-byte row slice, the next tile is in
    // flight in registers), then the waves reduce them from LDS -- no wave waits on a global load per row any more.
    const int trow = tid >> 4, tq = (tid & 15) * 4;                  // staging: rows trow and trow + 32
    float4 pre[2];
    auto fetch = [&](long long base) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = base + trow + 32 * u;
            pre[u] = r < r1 ? *reinterpret_cast<const float4 *>(z + r * (long long)ldz + g * VQ_D + tq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (r0 < r1) fetch(r0);
    __syncthreads();
    for (long long base = r0; base < r1; base += 64) {
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<float4 *>(&tile[(trow + 32 * u) * VQ_D + tq]) = pre[u];
        const long long r = base + lane;
        int code = -1;
        if (r < r1) code = (int)idx[((r / P) * num + g) * (long long)P + r % P];
        __syncthreads();
        if (base + 64 < r1) fetch(base + 64);
        // Rows of the tile that share a code are summed in registers first and hit the LDS accumulator once, by
        // the wave that owns the code's FIRST row in the tile (row j belongs to wave j & 7): no two waves ever touch
        // the same accumulator row inside a tile, the work is spread by rows (a skewed code histogram no longer
        // serialises on one wave), and the order -- rows ascending inside a tile, tiles ascending -- is fixed.
        unsigned long long todo = __ballot(code >= 0);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;                // leader: first unprocessed row
            const int cj = __shfl(code, j);
            const unsigned long long same = __ballot(code == cj);
            todo &= ~same;
            if ((j & 7) != wave) continue;
            float sum = 0.f; int cnt = 0;
            unsigned long long mm = same;
            while (mm) { const int r_ = __ffsll((long long)mm) - 1; mm &= mm - 1; sum += tile[r_ * VQ_D + lane]; ++cnt; }
            acc[cj * LDS_LD + lane] += sum;
            if (lane == 0) acc[cj * LDS_LD + VQ_D] += (float)cnt;
        }
        __syncthreads();
    }
    __syncthreads();
    float *dst = partial + ((long long)g * nchunks + c) * (KC * LDS_LD);
    for (int i = tid; i < KC * LDS_LD; i += EMA_THREADS) dst[i] = acc[i];
}

// stats[g][i] = sum_c partial[g][c][i]  (chunk order)
__global__ void lvt_vq_ema_reduce_kernel(const float *__restrict__ partial, int nchunks, long long per_group,
                                         float *__restrict__ stats) {
    const int g = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_group) return;
    const float *p = partial + (long long)g * nchunks * per_group + i;
    // chunk order is fixed (bit-reproducible); the loads of 8 chunks are issued together
    float s = 0.f;
    int c = 0;
    for (; c + 7 < nchunks; c += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(long long)(c + u) * per_group];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; c < nchunks; ++c) s += p[(long long)c * per_group];
    stats[(long long)g * per_group + i] = s;
}

// one workgroup per codebook group (vq_embedding.py:48-59)
__global__ __launch_bounds__(512) void lvt_vq_ema_finalize_kernel(const float *__restrict__ stats, int KC,
                                                                  float decay, float one_minus_decay, float eps,
                                                                  float *__restrict__ running_size,
                                                                  float *__restrict__ running_sum,
                                                                  float *__restrict__ weight) {
    __shared__ float red[512];
    __shared__ float ntot;
    const int g = blockIdx.x, tid = threadIdx.x;
    float *rs = running_size + (long long)g * KC;
    float *rsum = running_sum + (long long)g * KC * VQ_D;
    float *w = weight + (long long)g * KC * VQ_D;
    const float *st = stats + (long long)g * KC * (VQ_D + 1);
    float part = 0.f;
    for (int k = tid; k < KC; k += blockDim.x) {
        const float v = rs[k] * decay + one_minus_decay * st[k * (VQ_D + 1) + VQ_D];
        rs[k] = v;
        part += v;
    }
    red[tid] = part;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) ntot = red[0];
    __syncthreads();
    const float n = ntot;
    // 8 elements per thread and pass, all loads issued before the first use: with one element per pass every
    // iteration paid a full memory latency (64 dependent round trips per thread, 38 us for 128 K floats)
    constexpr int FB = 8;
    for (int i0 = tid; i0 < KC * VQ_D; i0 += blockDim.x * FB) {
        float a[FB], b[FB], c[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = i0 + u * blockDim.x;
            const bool ok = i < KC * VQ_D;
            const int k = ok ? i / VQ_D : 0, d = i % VQ_D;
            a[u] = ok ? rsum[i] : 0.f; b[u] = st[k * (VQ_D + 1) + d]; c[u] = rs[k];
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < KC * VQ_D) {
                const float s = a[u] * decay + one_minus_decay * b[u];
                rsum[i] = s;
                const float size_ = (c[u] + eps) / (n + KC * eps) * n;
                w[i] = s / size_;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
static int vq_smem_bytes(int KC) { return (VQ_D * (KC + 1) + KC) * (int)sizeof(float); }

extern "C" size_t lvt_vq_nearest_workspace_bytes(long long rows, int num, int KC) {
    const int nparts = KC / VQH_CODES > 0 ? KC / VQH_CODES : 1;
    return (size_t)rows * num * nparts * (sizeof(float) + sizeof(int));
}

extern "C" int lvt_vq_nearest(const float *z, long long rows, int ldz, int num, int D, int KC,
                              const float *codebooks, long long *idx_out, int P, int flags, void *workspace,
                              size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(z && codebooks && idx_out, "vq_nearest: null pointer");
    LVT_REQUIRE(D == VQ_D, "vq_nearest: only D=%d per codebook is instantiated (got %d)", VQ_D, D);
    LVT_REQUIRE(KC == 512 || KC == 256 || KC == 128, "vq_nearest: codebook size %d not instantiated", KC);
    LVT_REQUIRE(rows > 0 && num > 0 && P > 0 && rows % P == 0, "vq_nearest: bad rows/P");
    LVT_REQUIRE(ldz % 4 == 0 && ldz >= num * D && lvt_aligned16(z) && lvt_aligned16(codebooks),
                "vq_nearest: alignment / ldz");
    if (!(flags & LVT_MATH_F32) && (flags & LVT_VQ_COARSE)) {
        const long long need_ = lvt_cdiv((rows + 31) / 32, VQ_THREADS / 64);
        int bpp = LVT_NUM_CU / num;                     // one workgroup per CU (LDS-bound)
        if (bpp > need_) bpp = (int)need_;
        if (bpp < 1) bpp = 1;
        hipStream_t s = (hipStream_t)stream;
        if (KC == 512) hipLaunchKernelGGL(lvt_vq_nearest_coarse_kernel<512>, dim3(bpp, num), dim3(VQ_THREADS), 0, s, z, rows, ldz, codebooks, idx_out, P, num);
        else if (KC == 256) hipLaunchKernelGGL(lvt_vq_nearest_coarse_kernel<256>, dim3(bpp, num), dim3(VQ_THREADS), 0, s, z, rows, ldz, codebooks, idx_out, P, num);
        else hipLaunchKernelGGL(lvt_vq_nearest_coarse_kernel<128>, dim3(bpp, num), dim3(VQ_THREADS), 0, s, z, rows, ldz, codebooks, idx_out, P, num);
        LVT_CHECK_LAUNCH("lvt_vq_nearest_coarse_kernel");
        return LVT_OK;
    }
    if ((flags & LVT_MATH_F16X2) && !(flags & LVT_MATH_F32)) {
        const long long need_ = lvt_cdiv((rows + 31) / 32, VQ_THREADS / 64);
        int bpp = LVT_NUM_CU / num;                     // one workgroup per CU (LDS-bound)
        if (bpp > need_) bpp = (int)need_;
        if (bpp < 1) bpp = 1;
        hipStream_t s = (hipStream_t)stream;
        const int smem = 2 * KC * VQF_LD * (int)sizeof(unsigned short) + (KC + 16) * (int)sizeof(float);
        hipError_t e;
#define VQF_LAUNCH(KCV)                                                                                                \
    e = hipFuncSetAttribute((const void *)lvt_vq_nearest_f16x2_kernel<KCV>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    if (e != hipSuccess) { lvt_set_error("vq_nearest: cannot get %d B LDS: %s", smem, hipGetErrorString(e));           \
        return LVT_ELAUNCH; }                                                                                          \
    hipLaunchKernelGGL(lvt_vq_nearest_f16x2_kernel<KCV>, dim3(bpp, num), dim3(VQ_THREADS), smem, s, z, rows, ldz, codebooks, idx_out, P, num);
        if (KC == 512) { VQF_LAUNCH(512) } else if (KC == 256) { VQF_LAUNCH(256) } else { VQF_LAUNCH(128) }
#undef VQF_LAUNCH
        LVT_CHECK_LAUNCH("lvt_vq_nearest_f16x2_kernel");
        return LVT_OK;
    }
    if (!(flags & LVT_MATH_F32) && KC % VQH_CODES == 0 && workspace &&
        workspace_bytes >= lvt_vq_nearest_workspace_bytes(rows, num, KC)) {
        const int nparts = KC / VQH_CODES;
        float *pbest = (float *)workspace;
        int *pidx = (int *)(pbest + rows * num * nparts);
        int bpp = LVT_NUM_CU / (num * nparts);        // one workgroup per CU (LDS-bound)
        const long long need_ = lvt_cdiv((rows + 31) / 32, VQ_THREADS / 64);
        if (bpp > need_) bpp = (int)need_;
        if (bpp < 1) bpp = 1;
        hipLaunchKernelGGL(lvt_vq_nearest_half_kernel, dim3(bpp, nparts, num), dim3(VQ_THREADS), 0, (hipStream_t)stream, z,
                           rows, ldz, KC, codebooks, pbest, pidx);
        LVT_CHECK_LAUNCH("lvt_vq_nearest_half_kernel");
        hipLaunchKernelGGL(lvt_vq_nearest_merge_kernel, dim3((unsigned)lvt_cdiv(rows * num, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const float *)pbest, (const int *)pidx, rows, num, nparts, P, idx_out);
        LVT_CHECK_LAUNCH("lvt_vq_nearest_merge_kernel");
        return LVT_OK;
    }
    const long long ntiles = (rows + 31) / 32;
    int bpg = LVT_NUM_CU / num;                       // one workgroup per CU (LDS-bound)
    const long long need = lvt_cdiv(ntiles, VQ_THREADS / 64);
    if (bpg > need) bpg = (int)need;
    if (bpg < 1) bpg = 1;
    const int smem = vq_smem_bytes(KC);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
#define VQ_LAUNCH(KCV)                                                                                          \
    e = hipFuncSetAttribute((const void *)lvt_vq_nearest_kernel<KCV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            smem);                                                                              \
    if (e != hipSuccess) { lvt_set_error("vq_nearest: cannot get %d B LDS: %s", smem, hipGetErrorString(e));    \
        return LVT_ELAUNCH; }                                                                                   \
    hipLaunchKernelGGL(lvt_vq_nearest_kernel<KCV>, dim3(bpg * num), dim3(VQ_THREADS), smem, s, z, rows, ldz, num, \
                       codebooks, idx_out, P, bpg);
    if (KC == 512) { VQ_LAUNCH(512) } else if (KC == 256) { VQ_LAUNCH(256) } else { VQ_LAUNCH(128) }
#undef VQ_LAUNCH
    LVT_CHECK_LAUNCH("lvt_vq_nearest_kernel");
    return LVT_OK;
}

extern "C" int lvt_vq_gather(const long long *idx, const float *codebooks, long long rows, int num, int D, int KC,
                             int P, float *out, int ldo, void *stream) {
    LVT_REQUIRE(idx && codebooks && out && D == VQ_D && rows > 0 && rows % P == 0, "vq_gather: bad args");
    const long long pairs = rows * num;
    LVT_REQUIRE(ldo % 4 == 0 && lvt_aligned16(out) && lvt_aligned16(codebooks), "vq_gather: out / codebooks must be 16-byte aligned");
    const int blocks = (int)(lvt_cdiv(pairs, 16 * GATHER_UNROLL) < 8192 ? lvt_cdiv(pairs, 16 * GATHER_UNROLL) : 8192);
    hipLaunchKernelGGL(lvt_vq_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, codebooks, rows,
                       num, KC, P, out, ldo);
    LVT_CHECK_LAUNCH("lvt_vq_gather_kernel");
    return LVT_OK;
}

static void ema_chunks(long long rows, int num, long long *rows_per_chunk, int *nchunks) {
    long long target = LVT_NUM_CU / (num > 0 ? num : 1);          // ~one workgroup per CU
    if (target < 1) target = 1;
    long long rpc = lvt_cdiv(rows, target);
    rpc = lvt_cdiv(rpc, 64) * 64;
    if (rpc < 256) rpc = 256;
    *rows_per_chunk = rpc;
    *nchunks = (int)lvt_cdiv(rows, rpc);
}
extern "C" size_t lvt_vq_ema_workspace_bytes(long long rows, int num, int D, int KC) {
    long long rpc; int nch;
    ema_chunks(rows, num, &rpc, &nch);
    return (size_t)num * nch * KC * (D + 1) * sizeof(float);
}
extern "C" int lvt_vq_ema_accumulate(const long long *idx, const float *z, long long rows, int ldz, int num, int D,
                                     int KC, int P, float *stats, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    LVT_REQUIRE(idx && z && stats && D == VQ_D && rows > 0 && rows % P == 0, "vq_ema_accumulate: bad args");
    LVT_REQUIRE(KC == 512 || KC == 256 || KC == 128, "vq_ema_accumulate: codebook size %d not instantiated", KC);
    long long rpc; int nch;
    ema_chunks(rows, num, &rpc, &nch);
    if (!workspace || workspace_bytes < lvt_vq_ema_workspace_bytes(rows, num, D, KC)) {
        lvt_set_error("vq_ema_accumulate: workspace too small");
        return LVT_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int smem = (((KC * (VQ_D + 1) + 3) & ~3) + 64 * VQ_D) * (int)sizeof(float);
    hipError_t e;
#define EMA_LAUNCH(KCV)                                                                                          \
    e = hipFuncSetAttribute((const void *)lvt_vq_ema_partial_kernel<KCV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            smem);                                                                               \
    if (e != hipSuccess) { lvt_set_error("vq_ema_accumulate: cannot get %d B LDS: %s", smem, hipGetErrorString(e)); \
        return LVT_ELAUNCH; }                                                                                    \
    hipLaunchKernelGGL(lvt_vq_ema_partial_kernel<KCV>, dim3(num * nch), dim3(EMA_THREADS), smem, s, idx, z, rows, ldz, \
                       num, P, rpc, nch, (float *)workspace);
    if (KC == 512) { EMA_LAUNCH(512) } else if (KC == 256) { EMA_LAUNCH(256) } else { EMA_LAUNCH(128) }
#undef EMA_LAUNCH
    LVT_CHECK_LAUNCH("lvt_vq_ema_partial_kernel");
    const long long per_group = (long long)KC * (D + 1);
    hipLaunchKernelGGL(lvt_vq_ema_reduce_kernel, dim3((unsigned)lvt_cdiv(per_group, 256), num), dim3(256), 0, s,
                       (const float *)workspace, nch, per_group, stats);
    LVT_CHECK_LAUNCH("lvt_vq_ema_reduce_kernel");
    return LVT_OK;
}

extern "C" int lvt_vq_ema_finalize(const float *stats, int num, int D, int KC, float decay, float eps,
                                   float *running_size, float *running_sum, float *weight, void *stream) {
    LVT_REQUIRE(stats && running_size && running_sum && weight && D == VQ_D, "vq_ema_finalize: bad args");
    // the reference evaluates (1 - decay) in double precision python and hands it to add_(alpha=...)
    const float omd = (float)(1.0 - (double)decay);
    hipLaunchKernelGGL(lvt_vq_ema_finalize_kernel, dim3(num), dim3(512), 0, (hipStream_t)stream, stats, KC, decay,
                       omd, eps, running_size, running_sum, weight);
    LVT_CHECK_LAUNCH("lvt_vq_ema_finalize_kernel");
    return LVT_OK;
}
