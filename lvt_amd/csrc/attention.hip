// Fused attention forward for one 256-token block (reference: ScaledDotProductAttention, vt_attention.py:52-81):
//   P = softmax_j( q.k_j / temper + (Bt + Bh) + Bw  |  fill where j > i ),   O = P V,
// one launch instead of QK^T GEMM -> softmax kernel -> PV GEMM: the 134 MB score tensor of a layer (b=64) no longer
// makes two extra round trips through HBM.  P is still written once because the backward pass consumes it.
//
// Work split: a workgroup (4 waves) owns 128 queries of one (sample, head); a wave owns 32 of them and all 256 keys.
// The products are computed TRANSPOSED so that the probabilities never leave the registers between the two GEMMs:
//   S^T[key][query] = K Q^T      A operand = K rows from LDS, B operand = the wave's Q rows (registers)
//   O^T[d][query]   = V^T P^T    A operand = V^T rows from LDS, B operand = P^T straight from the accumulators
// In the 32x32 accumulator layout a lane holds one query column (lane & 31) and keys (r&3) + 8(r>>2) + 4(lane>>5):
// registers 8s..8s+7 of a tile are exactly the 8 "k" values lane-half needs for MFMA step s, as long as the A
// operand (V^T) is read in the same key order {0-3, 8-11} + 4*half -- two 8-byte LDS reads per plane.
// The softmax reduces over keys = over a lane's own registers plus one exchange with the other half-wave.
// Arithmetic: fp32 in/out; every product is the exact bf16x3 split (six v_mfma_f32_32x32x16_bf16, fp32
// accumulation), as in gemm_engine.hip.  S == 256 and head dim 128 (all shipped configurations).
#include "attn_common.h"

// acc += a * b with the six products of the split, smallest terms first
__device__ __forceinline__ void at_mfma6(f32x16 &acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// BT/BH/BW > 0: block geometry known at compile time -- the bias of element (query i, key j) is then
// BTv[tj] + BHv[hj] + BWv[wj] with per-lane register tables and compile-time indices (the key index of accumulator
// register r is 32T + (r&3) + 8(r>>2) + 4*half: for BW % 8 == 0 the +4*half never carries, so it is folded into the
// table when it is built).  BT == 0: any geometry, coordinates and banks are looked up in LDS per element.
template <int BT, int BH, int BW>
__global__ __launch_bounds__(256, 1) void lvt_attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                              const float *__restrict__ v, int H, float inv_temper,
                                                              const float *__restrict__ dt, const float *__restrict__ dh,
                                                              const float *__restrict__ dw, AttnGeom g, int masked, float fill,
                                                              float *__restrict__ P, float *__restrict__ o) {
    // staging: K planes [3][64 keys][AT_KLD] in phase 1, V^T planes [3][128 d][AT_VLD] in phase 2 (same bytes)
    __shared__ __attribute__((aligned(16))) unsigned short stage[3 * AT_KC * AT_KLD];
    __shared__ float bank_t[64], bank_h[64], bank_w[64];
    __shared__ unsigned char cj_t[AT_S], cj_h[AT_S], cj_w[AT_S];
    constexpr int KPL = AT_KC * AT_KLD, VPL = AT_D * AT_VLD;          // plane sizes (bf16 elements)
    static_assert(3 * VPL <= 3 * KPL, "V^T planes must fit in the K staging area");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh_ = blockIdx.x >> 1, qhalf = blockIdx.x & 1;
    const int b = bh_ / H, h = bh_ % H;
    const int hd = H * AT_D;
    const long long row0 = (long long)b * AT_S;                      // first token row of the sample
    const int nt = 2 * g.bt - 1, nh = 2 * g.bh - 1, nw = 2 * g.bw - 1;

    // bias banks of this head and key coordinates
    if (tid < nt) bank_t[tid] = dt[h * nt + tid];
    if (tid < nh) bank_h[tid] = dh[h * nh + tid];
    if (tid < nw) bank_w[tid] = dw[h * nw + tid];
    { const int j = tid; cj_w[j] = (unsigned char)(j % g.bw); cj_h[j] = (unsigned char)((j / g.bw) % g.bh); cj_t[j] = (unsigned char)(j / (g.bw * g.bh)); }

    // the wave's queries: lane (l31, half) holds d = 16s + 8*half .. +7 of query i for every step s, as three planes
    const int i = qhalf * 128 + wave * 32 + l31;
    const int wi = i % g.bw, hi = (i / g.bw) % g.bh, ti = i / (g.bw * g.bh);
    bf16x8 qb[AT_D / 16][3];
    {
        const float *qrow = q + (row0 + i) * hd + h * AT_D + 8 * half;
#pragma unroll
        for (int s = 0; s < AT_D / 16; ++s)
            at_split8(*reinterpret_cast<const float4 *>(qrow + 16 * s), *reinterpret_cast<const float4 *>(qrow + 16 * s + 4),
                      qb[s][0], qb[s][1], qb[s][2]);
    }

    // ---------------- phase 1: S^T = K Q^T, all 256 keys ----------------
    f32x16 st[AT_S / 32];
#pragma unroll
    for (int T = 0; T < AT_S / 32; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[T][r] = 0.f;

    // staging of a 64-key chunk: thread (row = tid >> 2, part = tid & 3) owns 8 float4 of its row.  The loads of
    // chunk c+1 are issued before the MFMAs of chunk c (one wave per SIMD: nothing else would hide their latency).
    float4 greg[8];
    const int srow = tid >> 2, spart = tid & 3;
    auto load_k = [&](int c) {
        const float *krow = k + (row0 + c * AT_KC + srow) * hd + h * AT_D;
#pragma unroll
        for (int u = 0; u < 8; ++u) greg[u] = *reinterpret_cast<const float4 *>(krow + (u * 4 + spart) * 4);
    };
    auto park_k = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint2 p1, p2, p3;
            at_split4(greg[u], p1, p2, p3);
            unsigned short *dst = stage + srow * AT_KLD + (u * 4 + spart) * 4;
            *reinterpret_cast<uint2 *>(dst) = p1;
            *reinterpret_cast<uint2 *>(dst + KPL) = p2;
            *reinterpret_cast<uint2 *>(dst + 2 * KPL) = p3;
        }
    };
    // V^T staging: a thread transposes 4 keys x 4 d blocks (lanes along the keys: the 8-byte plane stores of a
    // 16-lane group are contiguous); greg holds 2 blocks of 4 float4
    const int vkq = tid & 15;
    auto load_v = [&](int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float *vp = v + (row0 + c * AT_KC + 4 * vkq) * hd + h * AT_D + 4 * ((tid >> 4) + 16 * u);
#pragma unroll
            for (int r = 0; r < 4; ++r) greg[4 * u + r] = *reinterpret_cast<const float4 *>(vp + (long long)r * hd);
        }
    };
    auto park_v = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int dq = (tid >> 4) + 16 * u;
            const float4 v0 = greg[4 * u], v1 = greg[4 * u + 1], v2 = greg[4 * u + 2], v3 = greg[4 * u + 3];
            const float4 tr[4] = {make_float4(v0.x, v1.x, v2.x, v3.x), make_float4(v0.y, v1.y, v2.y, v3.y),
                                  make_float4(v0.z, v1.z, v2.z, v3.z), make_float4(v0.w, v1.w, v2.w, v3.w)};
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                uint2 p1, p2, p3;
                at_split4(tr[dd], p1, p2, p3);
                unsigned short *dst = stage + (4 * dq + dd) * AT_VLD + 4 * vkq;
                *reinterpret_cast<uint2 *>(dst) = p1;
                *reinterpret_cast<uint2 *>(dst + VPL) = p2;
                *reinterpret_cast<uint2 *>(dst + 2 * VPL) = p3;
            }
        }
    };

    // causal layers: key j > query i carries exactly zero probability (exp(fill - max) underflows to 0), so key
    // chunks beyond the workgroup's last query are never staged and key tiles beyond a wave's last query are
    // never multiplied; their accumulators stay 0 and are overwritten with `fill` / 0 below.
    const int iw_last = qhalf * 128 + wave * 32 + 31;                     // last query of this wave
    const int nchunks = masked ? (qhalf * 128 + 127) / AT_KC + 1 : AT_S / AT_KC;
    load_k(0);
#pragma unroll
    for (int c = 0; c < AT_S / AT_KC; ++c) {
        if (c >= nchunks) break;
        park_k();
        __syncthreads();
        if (c + 1 < nchunks) load_k(c + 1); else load_v(0);               // in flight during the MFMAs (and the softmax)
        // the two key tiles of the chunk advance together, term by term: consecutive MFMAs never hit the same accumulator
#pragma unroll
        for (int s = 0; s < AT_D / 16; ++s) {
            bf16x8 a[AT_KC / 32][3];
#pragma unroll
            for (int kt = 0; kt < AT_KC / 32; ++kt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[kt][pl] = *reinterpret_cast<const bf16x8 *>(stage + (kt * 32 + l31) * AT_KLD + 8 * half + pl * KPL + 16 * s);
            constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};         // smallest terms first
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                for (int kt = 0; kt < AT_KC / 32; ++kt)
                    if (!masked || 32 * (c * (AT_KC / 32) + kt) <= iw_last)
                        st[c * (AT_KC / 32) + kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][TA[tm]], qb[s][TB[tm]],
                                                                                             st[c * (AT_KC / 32) + kt], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---------------- softmax over the keys of every query column ----------------
    float m = -3.4e38f;
    if constexpr (BT > 0) {
        static_assert(BW % 8 == 0 && BT * BH * BW == AT_S, "compile-time geometry: BW % 8 == 0, 256 tokens");
        float btv[BT], bhv[BH], bwv[BW / 2];
#pragma unroll
        for (int x = 0; x < BT; ++x) btv[x] = bank_t[ti - x + BT - 1];
#pragma unroll
        for (int x = 0; x < BH; ++x) bhv[x] = bank_h[hi - x + BH - 1];
#pragma unroll
        for (int x = 0; x < BW / 2; ++x) {                    // wj = (x & 3) + 8 * (x >> 2) + 4 * half
            const int wj = (x & 3) + 8 * (x >> 2) + 4 * half;
            bwv[x] = bank_w[wi - wj + BW - 1];
        }
#pragma unroll
        for (int T = 0; T < AT_S / 32; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jc = 32 * T + (r & 3) + 8 * (r >> 2);              // key index without the half offset
                const int wc = jc % BW, hj = (jc / BW) % BH, tj = jc / (BW * BH);
                const int wx = (wc & 3) + 4 * (wc >> 3);                     // slot of wj = wc + 4*half in bwv
                const float bias = (btv[tj] + bhv[hj]) + bwv[wx];
                float x = st[T][r] * inv_temper + bias;
                if (masked && jc + 4 * half > i) x = fill;
                st[T][r] = x;
                m = fmaxf(m, x);
            }
    } else {
#pragma unroll
        for (int T = 0; T < AT_S / 32; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * T + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float bias = (bank_t[ti - cj_t[j] + g.bt - 1] + bank_h[hi - cj_h[j] + g.bh - 1]) + bank_w[wi - cj_w[j] + g.bw - 1];
                float x = st[T][r] * inv_temper + bias;
                if (masked && j > i) x = fill;
                st[T][r] = x;
                m = fmaxf(m, x);
            }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int T = 0; T < AT_S / 32; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __expf(st[T][r] - m); st[T][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    {
        float *prow = P + (((long long)b * H + h) * AT_S + i) * AT_S + 4 * half;
#pragma unroll
        for (int T = 0; T < AT_S / 32; ++T)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float4 pv;
                pv.x = (st[T][4 * gq + 0] *= inv); pv.y = (st[T][4 * gq + 1] *= inv);
                pv.z = (st[T][4 * gq + 2] *= inv); pv.w = (st[T][4 * gq + 3] *= inv);
                *reinterpret_cast<float4 *>(prow + 32 * T + 8 * gq) = pv;
            }
    }

    // ---------------- phase 2: O^T = V^T P^T ----------------
    f32x16 oacc[AT_D / 32];
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 32; ++dtile)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dtile][r] = 0.f;

#pragma unroll
    for (int c = 0; c < AT_S / AT_KC; ++c) {
        if (c >= nchunks) break;
        park_v();
        __syncthreads();
        if (c + 1 < nchunks) load_v(c + 1);
#pragma unroll
        for (int kt = 0; kt < AT_KC / 32; ++kt) {
            const int T = c * (AT_KC / 32) + kt;
            if (masked && 32 * T > iw_last) continue;                     // P is exactly 0 on this key tile
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // B operand: P^T, registers 8s .. 8s+7 of tile T = keys {0-3, 8-11} + 16s + 4*half of the tile
                bf16x8 pb[3];
                at_split8(make_float4(st[T][8 * s + 0], st[T][8 * s + 1], st[T][8 * s + 2], st[T][8 * s + 3]),
                          make_float4(st[T][8 * s + 4], st[T][8 * s + 5], st[T][8 * s + 6], st[T][8 * s + 7]),
                          pb[0], pb[1], pb[2]);
                bf16x8 a[AT_D / 32][3];
#pragma unroll
                for (int dtile = 0; dtile < AT_D / 32; ++dtile) {
                    const unsigned short *vrow = stage + (dtile * 32 + l31) * AT_VLD + kt * 32 + 16 * s + 4 * half;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const uint2 lo = *reinterpret_cast<const uint2 *>(vrow + pl * VPL);
                        const uint2 hi2 = *reinterpret_cast<const uint2 *>(vrow + pl * VPL + 8);
                        const uint4 u = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
                        a[dtile][pl] = *reinterpret_cast<const bf16x8 *>(&u);
                    }
                }
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int dtile = 0; dtile < AT_D / 32; ++dtile)
                        oacc[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dtile][TA[tm]], pb[TB[tm]], oacc[dtile], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // O[i][d]: a lane holds 4 consecutive d per register group
    {
        float *orow = o + (row0 + i) * hd + h * AT_D + 4 * half;
#pragma unroll
        for (int dtile = 0; dtile < AT_D / 32; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4 *>(orow + 32 * dtile + 8 * gq) =
                    make_float4(oacc[dtile][4 * gq + 0], oacc[dtile][4 * gq + 1], oacc[dtile][4 * gq + 2], oacc[dtile][4 * gq + 3]);
    }
}

extern "C" int lvt_attn_fwd(const float *q, const float *k, const float *v, int B, int H, int S, int da, float temper,
                            const float *dt, const float *dh, const float *dw, int bt, int bh, int bw, int masked,
                            float fill, float *P, float *o, void *stream) {
    LVT_REQUIRE(q && k && v && dt && dh && dw && P && o && B > 0 && H > 0, "attn_fwd: bad args");
    LVT_REQUIRE(S == AT_S && da == AT_D && S == bt * bh * bw, "attn_fwd: S=%d da=%d unsupported (256 x 128 only)", S, da);
    LVT_REQUIRE(2 * bt - 1 <= 64 && 2 * bh - 1 <= 64 && 2 * bw - 1 <= 64 && bt < 256 && bh < 256 && bw < 256, "attn_fwd: block geometry");
    LVT_REQUIRE(lvt_aligned16(q) && lvt_aligned16(k) && lvt_aligned16(v) && lvt_aligned16(P) && lvt_aligned16(o), "attn_fwd: alignment");
    AttnGeom g = {bt, bh, bw};
    const dim3 grid((unsigned)(B * H * 2)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (bt == 1 && bh == 16 && bw == 16)
        hipLaunchKernelGGL((lvt_attn_fwd_kernel<1, 16, 16>), grid, blk, 0, s, q, k, v, H, 1.f / temper, dt, dh, dw, g, masked, fill, P, o);
    else if (bt == 4 && bh == 8 && bw == 8)
        hipLaunchKernelGGL((lvt_attn_fwd_kernel<4, 8, 8>), grid, blk, 0, s, q, k, v, H, 1.f / temper, dt, dh, dw, g, masked, fill, P, o);
    else
        hipLaunchKernelGGL((lvt_attn_fwd_kernel<0, 0, 0>), grid, blk, 0, s, q, k, v, H, 1.f / temper, dt, dh, dw, g, masked, fill, P, o);
    LVT_CHECK_LAUNCH("lvt_attn_fwd_kernel");
    return LVT_OK;
}
