// Building blocks shared by the fused attention kernels (attention.hip: generic forward; attention_pipe.hip: the
// software-pipelined forward and backward kernels): exact 3-way bf16 splits, the two LDS operand layouts and their
// staging ("park") routines, MFMA fragment loads.  One 256-token block, head dimension 128, fp32 in / out.
#pragma once
#include "lvt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define AT_S 256
#define AT_D 128
#define AT_KC 64                 // rows (keys or queries) staged per chunk
#define AT_KLD (AT_D + 8)        // row-major plane: row stride (bf16): 272 B, conflict-free 16-byte fragment reads
#define AT_VLD (AT_KC + 4)       // transposed plane: row stride (bf16): 136 B, conflict-free 8-byte fragment reads
#define AT_KPL (AT_KC * AT_KLD)  // plane sizes (bf16 elements); both layouts take 3 planes of 17408 B
#define AT_VPL (AT_D * AT_VLD)
#define AT_STAGE (3 * AT_KPL)    // one staged chunk (three planes), in bf16 elements = 52224 B

struct AttnGeom { int bt, bh, bw; };

__device__ __forceinline__ unsigned at_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 4 floats -> three planes of 4 packed bf16 (exact split, RNE at every level)
__device__ __forceinline__ void at_split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3) {
    p1.x = at_cvt_pk(v.x, v.y); p1.y = at_cvt_pk(v.z, v.w);
    const float r0 = v.x - __uint_as_float(p1.x << 16), r1 = v.y - __uint_as_float(p1.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(p1.y << 16), r3 = v.w - __uint_as_float(p1.y & 0xffff0000u);
    p2.x = at_cvt_pk(r0, r1); p2.y = at_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(p2.x << 16), s1 = r1 - __uint_as_float(p2.x & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(p2.y << 16), s3 = r3 - __uint_as_float(p2.y & 0xffff0000u);
    p3.x = at_cvt_pk(s0, s1); p3.y = at_cvt_pk(s2, s3);
}
__device__ __forceinline__ void at_split8(const float4 lo, const float4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
    uint2 a1, a2, a3, b1, b2, b3;
    at_split4(lo, a1, a2, a3); at_split4(hi, b1, b2, b3);
    const uint4 u1 = make_uint4(a1.x, a1.y, b1.x, b1.y), u2 = make_uint4(a2.x, a2.y, b2.x, b2.y),
                u3 = make_uint4(a3.x, a3.y, b3.x, b3.y);
    p1 = *reinterpret_cast<const bf16x8 *>(&u1); p2 = *reinterpret_cast<const bf16x8 *>(&u2);
    p3 = *reinterpret_cast<const bf16x8 *>(&u3);
}
// the six products of the split, smallest terms first: acc += a * b
#define AT_TA(t) ((t) == 0 ? 1 : (t) == 1 ? 0 : (t) == 2 ? 2 : (t) == 3 ? 0 : (t) == 4 ? 1 : 0)
#define AT_TB(t) ((t) == 0 ? 1 : (t) == 1 ? 2 : (t) == 2 ? 0 : (t) == 3 ? 1 : (t) == 4 ? 0 : 0)

// ---- staging of a 64-row chunk of a token-major (rows x 128) operand by 256 threads -----------------------------
// row-major planes  [64 rows][AT_KLD]: thread (row = tid >> 2, part = tid & 3) owns 8 float4 of its row
// transposed planes [128 d][AT_VLD]  : thread transposes two 4 (rows) x 4 (d) blocks
__device__ __forceinline__ void at_load_rows(float4 (&g)[8], const float *base, long long ld, int tid) {
    const float *row = base + (long long)(tid >> 2) * ld;
    const int part = tid & 3;
#pragma unroll
    for (int u = 0; u < 8; ++u) g[u] = *reinterpret_cast<const float4 *>(row + (u * 4 + part) * 4);
}
__device__ __forceinline__ void at_park_rows_one(const float4 gv, unsigned short *stage, int tid, int u) {
    uint2 p1, p2, p3;
    at_split4(gv, p1, p2, p3);
    unsigned short *dst = stage + (tid >> 2) * AT_KLD + (u * 4 + (tid & 3)) * 4;
    *reinterpret_cast<uint2 *>(dst) = p1;
    *reinterpret_cast<uint2 *>(dst + AT_KPL) = p2;
    *reinterpret_cast<uint2 *>(dst + 2 * AT_KPL) = p3;
}
__device__ __forceinline__ void at_load_cols(float4 (&g)[8], const float *base, long long ld, int tid) {
    const int rq = tid & 15;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float *vp = base + (long long)(4 * rq) * ld + 4 * ((tid >> 4) + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[4 * u + r] = *reinterpret_cast<const float4 *>(vp + (long long)r * ld);
    }
}
// one quarter (u in 0..1, dd pair) of the transposed park: block u, d offsets 2*hp, 2*hp+1
__device__ __forceinline__ void at_park_cols_one(const float4 (&g)[8], unsigned short *stage, int tid, int u, int dd) {
    const int rq = tid & 15, dq = (tid >> 4) + 16 * u;
    const float4 v0 = g[4 * u], v1 = g[4 * u + 1], v2 = g[4 * u + 2], v3 = g[4 * u + 3];
    const float4 tr = dd == 0 ? make_float4(v0.x, v1.x, v2.x, v3.x) : dd == 1 ? make_float4(v0.y, v1.y, v2.y, v3.y)
                    : dd == 2 ? make_float4(v0.z, v1.z, v2.z, v3.z) : make_float4(v0.w, v1.w, v2.w, v3.w);
    uint2 p1, p2, p3;
    at_split4(tr, p1, p2, p3);
    unsigned short *dst = stage + (4 * dq + dd) * AT_VLD + 4 * rq;
    *reinterpret_cast<uint2 *>(dst) = p1;
    *reinterpret_cast<uint2 *>(dst + AT_VPL) = p2;
    *reinterpret_cast<uint2 *>(dst + 2 * AT_VPL) = p3;
}

// A fragments.  Row-major planes: rows (tile * 32 + l31), k = 16 s + 8 half .. + 7: one 16-byte read per plane.
__device__ __forceinline__ void at_frag_rows(bf16x8 (&a)[3], const unsigned short *stage, int tile, int s, int l31, int half) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        a[pl] = *reinterpret_cast<const bf16x8 *>(stage + (tile * 32 + l31) * AT_KLD + 8 * half + pl * AT_KPL + 16 * s);
}
// Transposed planes, accumulator key order: rows (dtile * 32 + l31), k slots = columns {0-3, 8-11} + 16 s2 + 4 half of the
// 32-column tile kt (the order in which a lane holds a 32x32 accumulator tile): two 8-byte reads per plane.
__device__ __forceinline__ void at_frag_cols_acc(bf16x8 (&a)[3], const unsigned short *stage, int dtile, int kt, int s2, int l31, int half) {
    const unsigned short *vrow = stage + (dtile * 32 + l31) * AT_VLD + kt * 32 + 16 * s2 + 4 * half;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const uint2 lo = *reinterpret_cast<const uint2 *>(vrow + pl * AT_VPL);
        const uint2 hi2 = *reinterpret_cast<const uint2 *>(vrow + pl * AT_VPL + 8);
        const uint4 u = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
        a[pl] = *reinterpret_cast<const bf16x8 *>(&u);
    }
}
// Transposed planes, natural order: k = columns 16 s + 8 half .. + 7 of the chunk: two 8-byte reads per plane
// (rows are 136 B apart: 8-byte aligned only).
__device__ __forceinline__ void at_frag_cols(bf16x8 (&a)[3], const unsigned short *stage, int dtile, int s, int l31, int half) {
    const unsigned short *vrow = stage + (dtile * 32 + l31) * AT_VLD + 16 * s + 8 * half;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const uint2 lo = *reinterpret_cast<const uint2 *>(vrow + pl * AT_VPL);
        const uint2 hi2 = *reinterpret_cast<const uint2 *>(vrow + pl * AT_VPL + 4);
        const uint4 u = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
        a[pl] = *reinterpret_cast<const bf16x8 *>(&u);
    }
}
