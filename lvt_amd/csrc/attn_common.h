// Definitions shared by the fused attention kernels (attention.hip: forward on fp32 operands, any block geometry;
// attention_pipe.hip: pipelined forward / backward on bf16x3-plane operands): tile constants and the exact 3-way bf16 split.
// One 256-token block, head dimension 128, fp32 in / out.
#pragma once
#include "lvt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define AT_S 256
#define AT_D 128
#define AT_KC 64                 // rows (keys or queries) staged per chunk
#define AT_KLD (AT_D + 8)        // row-major plane: row stride (bf16): 272 B, conflict-free 16-byte fragment reads
#define AT_VLD (AT_KC + 4)       // transposed plane: row stride (bf16): 136 B, conflict-free 8-byte fragment reads

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
