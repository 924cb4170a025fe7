// Flash-style fused attention for one 256-token block, head dimension 128, on fp32 operands in the f16x2 arithmetic
// (reference: ScaledDotProductAttention / BlockLocalAttention, vidgen/modeling/autoregressive/vt_attention.py:52-81,142-174).
//
// What this file replaces.  The pipelined kernels of attention_pipe.hip take q / k / v / dO as three bf16 planes (6 bytes per
// element, six MFMAs per product), write the softmax P as 134 MB of fp32 per layer, read it back twice in the backward pass and
// exchange dS the same way: 490-760 MB per launch against 268 / 536 MB of algorithmic traffic (profiles/r04_dsfvt_pmc_hbm_traffic.txt).
// Here
//   * operands arrive as plain fp32 (4 bytes per element, what the projection GEMMs write anyway) and are split into TWO fp16
//     terms while they are staged into LDS: a s = hi + lo, hi = RN16(a s), lo = RN16(a s - hi), with an exact power-of-two
//     scale s PER ROW (one token x one head: 128 values) taken from the row's own max |a| -- no max |.| plumbing from the
//     producer, no a-priori bound.  A product is THREE v_mfma_f32_16x16x32_f16 (lo hi, hi lo, hi hi) into one fp32 accumulator;
//     all fp16 x fp16 products are exact there, the dropped lo lo term is <= 2^-22 |a||b|.  Envelope: 22 bits + sign for every
//     element within 2^-16 of ITS ROW's max, absolute error <= 2^-39 of the row max below that;
//   * the forward pass keeps nothing but two floats per query row (running max m and sum l of the online softmax over 32-key
//     chunks; O is rescaled when the max moves) -- P never leaves the registers;
//   * the backward pass recomputes S = q k^T and dP = dO v^T from the operands in both of its kernels:
//       A (one wave = 16 queries): pass 1 over the keys forms delta_i = sum_j P_ij dP_ij from THE dP values it will use
//         (vt_attention.py:59-81 under autograd: the softmax backward cancels row sums the same way), pass 2 forms
//         g = P o (dP - delta), the bias-bank gradients and dQ = g K / temper;
//       B (one wave = 16 keys, K fragments in registers, V fragments resident in LDS): dV = P^T dO, dK = g^T Q / temper over
//         32-query chunks, delta read from A's output;
//     nine score-sized products instead of four, each at half the MFMA cost of attention_pipe.hip's, for 1/3 of its traffic.
// Reductions over an index that carries per-row scales (keys in O = P V and dQ, queries in dV / dK) fold the row scale into
// the other operand (P, g) and keep a running power-of-two scale of the accumulator ("online scale", exact).
//
// Staging: an ITEM is 64 LDS rows = 32 rows of one operand + 32 rows of a second one (K|V chunk for the query-stationary
// kernels, Q|dO chunk for the key-stationary one), two fp16 planes with a 288-byte row pitch (conflict-free for the 16-row
// ds_read_b128 fragment reads and the transposing ds_read_b64_tr_b16), in a two-slot ring with two register sets: while the
// MFMAs of step t read slot t & 1, item t + 1 is split and stored into the other slot and the global loads of item t + 2 are
// in flight; one barrier per step.  16 lanes own a row: its max |.| is four DPP steps, no LDS traffic.
// Tiling as in attention_pipe.hip: 16-wide tiles, one wave = 16 queries (keys), eight waves per workgroup = two per SIMD.
#include "attn_common.h"

namespace {

template <int I> struct IC { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FA_LD 144                         // row pitch (fp16): 288 B
#define FA_PL (64 * FA_LD)                // one plane of an item (fp16 elements)
#define FA_SLOT (2 * FA_PL)               // hi plane, lo plane: 36864 B
#define FA_EMIN 30                        // clamp of the biased exponent of a row max: rows below 2^-97 share that scale, and every
                                          // derived factor (2^15 / 2^(e - 141), products of a scale and an inverse scale) stays finite
#define FA_EMAX 254
#define FA_LOG2E 1.4426950408889634f      // scores are kept in log2 units: p = 2^(x - m) is one v_exp_f32

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fa_u2f(unsigned u) { return __uint_as_float(u); }
// float with biased exponent field e (0 when e <= 0)
__device__ __forceinline__ float fa_exp_field(int e) { return e <= 0 ? 0.f : fa_u2f((unsigned)e << 23); }
__device__ __forceinline__ int fa_clamp_e(int e) { return e < FA_EMIN ? FA_EMIN : (e > FA_EMAX ? FA_EMAX : e); }
__device__ __forceinline__ int fa_ebits(float nonneg) { return fa_clamp_e((int)((__float_as_uint(nonneg) >> 23) & 0xffu)); }
template <int CTRL> __device__ __forceinline__ float fa_dpp_max(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
// max over the 16 lanes of a DPP row: quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float fa_row16_max(float v) {
    v = fa_dpp_max<0xB1>(v);
    v = fa_dpp_max<0x4E>(v);
    v = fa_dpp_max<0x141>(v);
    v = fa_dpp_max<0x140>(v);
    return v;
}
// sum / max over the four lanes (kg = lane >> 4) that share a 16-wide tile column: the gfx950 row swaps (VALU; a
// ds_bpermute shuffle goes through the LDS crossbar and waits on lgkmcnt).  v_permlane16_swap exchanges rows 1 <-> 0 and
// 3 <-> 2 of its two operands, v_permlane32_swap the upper half of one with the lower half of the other: with both operands =
// v the two results hold v of (row, row ^ 1) resp. (half, half ^ 1) side by side.
__device__ __forceinline__ float fa_kg_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(fa_u2f(r[0]), fa_u2f(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(fa_u2f(r[0]), fa_u2f(r[1]));
}
__device__ __forceinline__ float fa_kg_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fa_u2f(r[0]) + fa_u2f(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fa_u2f(r[0]) + fa_u2f(r[1]);
}
__device__ __forceinline__ float fa_max4(const f32x4v a) { return fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])); }
__device__ __forceinline__ float fa_absmax4(const float4 a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); }

__device__ __forceinline__ float fa_mix_lo(unsigned h, float c) {        // c - half(h.lo), exact
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.f), "v"(c));
    return r;
}
__device__ __forceinline__ float fa_mix_hi(unsigned h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.f), "v"(c));
    return r;
}
// (a, b) * s -> packed fp16 hi pair and lo pair (instruction choice: gemm_engine.hip, f16_split_pair)
__device__ __forceinline__ void fa_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const f32x2 t = f32x2{a, b} * s;
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2v));
    const f32x2 r = {fa_mix_lo(ph, t.x), fa_mix_hi(ph, t.y)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
// the same for values that are already scaled
__device__ __forceinline__ void fa_split_pair1(float a, float b, unsigned &ph, unsigned &pl) {
    const f32x2 t = {a, b};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2v));
    const f32x2 r = {fa_mix_lo(ph, a), fa_mix_hi(ph, b)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
// eight scaled values (k slots 0..7 of a B operand) -> hi / lo fragments
__device__ __forceinline__ void fa_split8(const float (&v)[8], f16x8 &h, f16x8 &l) {
    u32x4 uh, ul;
    unsigned a, b;
    fa_split_pair1(v[0], v[1], a, b); uh[0] = a; ul[0] = b;
    fa_split_pair1(v[2], v[3], a, b); uh[1] = a; ul[1] = b;
    fa_split_pair1(v[4], v[5], a, b); uh[2] = a; ul[2] = b;
    fa_split_pair1(v[6], v[7], a, b); uh[3] = a; ul[3] = b;
    h = __builtin_bit_cast(f16x8, uh);
    l = __builtin_bit_cast(f16x8, ul);
}
// The biased score of one (query, key) pair in log2 units, from the accumulator of the scaled operands.  ONE expression for the
// forward and both backward kernels: the backward recomputes p = 2^(x - m) / l against the forward's m and l, and only a
// bit-identical x makes the row maximum cancel exactly and the recomputed P equal the forward's (folding m into the bias
// instead tripled the error of dq / dv on heavy-tailed operands).
__device__ __forceinline__ float fa_score(float acc, float kq, float bias) { return fmaf(acc, kq, bias); }
__device__ __forceinline__ f32x4v fa_mfma(const f16x8 a, const f16x8 b, const f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// staging
// ---------------------------------------------------------------------------------------------------------------------------
// One row of 128 fp32 values held by 16 lanes (lane c: columns 4 c .. + 3 and 4 c + 64 .. + 3) -> its scale, then the two
// fp16 planes of LDS row `row` of `slot` and rs_inv[row] = 2^(e - 14), the inverse of the scale.  (An inf / nan element takes
// part in max |.| like any other -- v_max ignores a nan, an inf clamps the scale: such a row is lost either way.)  Two pieces,
// so that a kernel can spread the work between its MFMA groups.
__device__ __forceinline__ int fa_row_scale(const float4 a, const float4 b) {     // -> biased exponent of the row max
    return fa_ebits(fa_row16_max(fmaxf(fa_absmax4(a), fa_absmax4(b))));
}
__device__ __forceinline__ void fa_row_store(const float4 a, const float4 b, int eb, unsigned short *slot, float *rs_inv, int row, int c) {
    const float s = fa_u2f((unsigned)(268 - eb) << 23);               // row max * s in [2^14, 2^15)
    uint2 h0, l0, h1, l1;
    fa_split_pair(a.x, a.y, s, h0.x, l0.x); fa_split_pair(a.z, a.w, s, h0.y, l0.y);
    fa_split_pair(b.x, b.y, s, h1.x, l1.x); fa_split_pair(b.z, b.w, s, h1.y, l1.y);
    unsigned short *d = slot + row * FA_LD + 4 * c;
    *reinterpret_cast<uint2 *>(d) = h0;
    *reinterpret_cast<uint2 *>(d + 64) = h1;
    *reinterpret_cast<uint2 *>(d + FA_PL) = l0;
    *reinterpret_cast<uint2 *>(d + FA_PL + 64) = l1;
    rs_inv[c == 0 ? row : 64 + c] = fa_u2f((unsigned)(eb - 14) << 23);      // (entries 64..79: scratch, keeps the piece branch-free)
}
// 512-thread workgroups: a thread holds row r = tid >> 4 of operand X and row r of operand Y of a 32 + 32-row item
struct G4 { float4 v[4]; };
__device__ __forceinline__ void fa_load(G4 &g, const float *__restrict__ x, const float *__restrict__ y, long long ld, int tid) {
    const int r = tid >> 4, c = tid & 15;
    const float *px = x + (long long)r * ld + 4 * c, *py = y + (long long)r * ld + 4 * c;
    g.v[0] = *reinterpret_cast<const float4 *>(px);
    g.v[1] = *reinterpret_cast<const float4 *>(px + 64);
    g.v[2] = *reinterpret_cast<const float4 *>(py);
    g.v[3] = *reinterpret_cast<const float4 *>(py + 64);
}
__device__ __forceinline__ void fa_park(const G4 &g, unsigned short *slot, float *rs_inv, int tid) {
    const int r = tid >> 4, c = tid & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half)
        fa_row_store(g.v[2 * half], g.v[2 * half + 1], fa_row_scale(g.v[2 * half], g.v[2 * half + 1]), slot, rs_inv, 32 * half + r, c);
}
// 256-thread workgroups: rows r = tid >> 4 and r + 16 of X, then of Y (piece p = 0..3: LDS row 16 p + r)
struct G8 { float4 v[8]; };
__device__ __forceinline__ void fa_load8(G8 &g, const float *__restrict__ x, const float *__restrict__ y, long long ld, int tid) {
    const int r = tid >> 4, c = tid & 15;
    const float *px = x + (long long)r * ld + 4 * c, *py = y + (long long)r * ld + 4 * c;
    g.v[0] = *reinterpret_cast<const float4 *>(px);
    g.v[1] = *reinterpret_cast<const float4 *>(px + 64);
    g.v[2] = *reinterpret_cast<const float4 *>(px + 16 * ld);
    g.v[3] = *reinterpret_cast<const float4 *>(px + 16 * ld + 64);
    g.v[4] = *reinterpret_cast<const float4 *>(py);
    g.v[5] = *reinterpret_cast<const float4 *>(py + 64);
    g.v[6] = *reinterpret_cast<const float4 *>(py + 16 * ld);
    g.v[7] = *reinterpret_cast<const float4 *>(py + 16 * ld + 64);
}
__device__ __forceinline__ void fa_park8(const G8 &g, unsigned short *slot, float *rs_inv, int tid) {
    const int r = tid >> 4, c = tid & 15;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        fa_row_store(g.v[2 * p], g.v[2 * p + 1], fa_row_scale(g.v[2 * p], g.v[2 * p + 1]), slot, rs_inv, 16 * p + r, c);
}
// The wave's stationary operand straight from global memory in B-fragment layout: lane (tile column c16 = row `row` of the
// operand, k block kg) holds k = 32 s + 8 kg .. + 7 of each of the four k steps; the row's scale from the 32 values of the lane
// and the three other lanes of the column.  inv = 2^(e - 14).
// (two pieces, so that a prologue can request the row together with its other operands and split it when they are all on their way)
__device__ __forceinline__ void fa_bfrags_request(const float *__restrict__ row, int kg, float4 (&v)[8]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        v[2 * s] = *reinterpret_cast<const float4 *>(row + 32 * s + 8 * kg);
        v[2 * s + 1] = *reinterpret_cast<const float4 *>(row + 32 * s + 8 * kg + 4);
    }
}
__device__ __forceinline__ void fa_bfrags_split(const float4 (&v)[8], f16x8 (&fr)[4][2], float &inv) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fa_absmax4(v[e]));
    m = fa_kg_max(m);
    const int eb = fa_ebits(m);
    const float sc = fa_u2f((unsigned)(268 - eb) << 23);
    inv = fa_u2f((unsigned)(eb - 14) << 23);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 uh, ul;
        unsigned a, b;
        fa_split_pair(v[2 * s].x, v[2 * s].y, sc, a, b); uh[0] = a; ul[0] = b;
        fa_split_pair(v[2 * s].z, v[2 * s].w, sc, a, b); uh[1] = a; ul[1] = b;
        fa_split_pair(v[2 * s + 1].x, v[2 * s + 1].y, sc, a, b); uh[2] = a; ul[2] = b;
        fa_split_pair(v[2 * s + 1].z, v[2 * s + 1].w, sc, a, b); uh[3] = a; ul[3] = b;
        fr[s][0] = __builtin_bit_cast(f16x8, uh);
        fr[s][1] = __builtin_bit_cast(f16x8, ul);
    }
}
__device__ __forceinline__ void fa_load_bfrags(const float *__restrict__ row, int kg, f16x8 (&fr)[4][2], float &inv) {
    float4 v[8];
    fa_bfrags_request(row, kg, v);
    fa_bfrags_split(v, fr, inv);
}
// fragment of LDS row `row` (A operand: the row is the tile row; B operand: the row is the tile column), k = 32 s + 8 kg .. + 7
__device__ __forceinline__ f16x8 fa_rowfrag(const unsigned short *slot, int pl, int row, int s, int kg) {
    return *reinterpret_cast<const f16x8 *>(slot + pl * FA_PL + row * FA_LD + 32 * s + 8 * kg);
}
// A fragment TRANSPOSED: tile row = column 16 dtile + c16 of the staged rows, k slots 0..3 = rows rb + 4 kg + 0..3, 4..7 = the
// same rows + 16 (attention_pipe.hip, ap_tr)
__device__ __forceinline__ f16x8 fa_trfrag(const unsigned short *slot, int pl, int rb, int dtile, int c16, int kg) {
    const unsigned short *p = slot + pl * FA_PL + (rb + 4 * kg + (c16 >> 2)) * FA_LD + 16 * dtile + 4 * (c16 & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 16 * FA_LD));
    union { struct { s16x4 a, b; } s; f16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

struct FaSmem {
    unsigned short ring[2][FA_SLOT];
    float rs_inv[2][80];
    float dhs[32];                        // this head's dh bank (2 BH - 1 entries), in log2 units
};
template <int NR> struct FaSmemA {
    unsigned short ring[2][FA_SLOT];
    float rs_inv[2][80];
    float dhs[32];
    float red[16];
    float R[128 * 4 * NR];                // per-lane class sums of g (bias-bank gradient), [query][kg][NR]
};
struct FaSmem3 {                          // 256-thread workgroups: three-slot ring
    unsigned short ring[3][FA_SLOT];
    float rs_inv[3][80];
    float dhs[32];
};

// workgroup -> (sample, head) pair and half: the two halves of a pair get block ids 8 apart (same XCD, back to back)
__device__ __forceinline__ int fa_pair(unsigned x) { return (int)(((x >> 4) << 3) | (x & 7)); }
__device__ __forceinline__ int fa_half(unsigned x) { return (int)((x >> 3) & 1); }
// the same for four quarters per pair
__device__ __forceinline__ int fa_pair4(unsigned x) { return (int)(((x >> 5) << 3) | (x & 7)); }
__device__ __forceinline__ int fa_quarter(unsigned x) { return (int)((x >> 3) & 3); }
__device__ __forceinline__ int fa_opaque(int x) { asm volatile("" : "+s"(x)); return x; }

template <int BH, int BW> struct Geo16 {
    static_assert(BW == 16 || BW == 8, "16-wide tiles: BW is 8 or 16");
    static constexpr int HP = BW == 16 ? BH : BH / 2;
    static __device__ __forceinline__ int hj(int hslot, int kg) { return BW == 16 ? hslot : 2 * hslot + (kg >> 1); }
    static __device__ __forceinline__ int wj(int r, int kg) { return BW == 16 ? 4 * kg + r : 4 * (kg & 1) + r; }
};
template <int BT, int BH, int BW> struct BankIdx {
    static constexpr int NT = 2 * BT - 1, NH = 2 * BH - 1, NW = 2 * BW - 1, NB = NT + NH + NW;
};

struct FaArgs {
    const float *q, *k, *v, *d_o;        // token-major (B*S rows, ld floats per row), head h in columns h*128 ..
    long long ld;
    int H;
    float c1, fill2;                     // log2(e) / temper; the causal fill in log2 units
    float inv_temper;
    const float *dt, *dh, *dw;
    float *o, *m, *l;                    // forward outputs: o and the row statistics (B*H*S each): max m (log2 units), 1 / sum l
    float *dq, *dk, *dv, *delta, *scal, *bank_partial;
};

// ===========================================================================================================================
// forward
// ===========================================================================================================================
template <int BT, int BH, int BW, int MASKED, int NCH>
__device__ __forceinline__ float fa_fwd_body(const FaArgs &A, FaSmem &sm, int bh_, int qhalf) {
    using GE = Geo16<BH, BW>;
    constexpr int HP = GE::HP;
    static_assert(BT * BH * BW == AT_S, "256 tokens");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / A.H, h = bh_ % A.H;
    const long long row0 = (long long)b * AT_S;
    const int il = wave * 16 + c16, i = qhalf * 128 + il;
    const float *kbase = A.k + row0 * A.ld + h * AT_D, *vbase = A.v + row0 * A.ld + h * AT_D;
    constexpr int NIT = NCH;

    if (tid < 2 * BH - 1) sm.dhs[tid] = A.dh[h * (2 * BH - 1) + tid] * FA_LOG2E;
    G4 g0, g1;
#define FA_G(t) ((((t) & 1) == 0) ? g0 : g1)
    auto load_item = [&](int t, G4 &gg) { fa_load(gg, kbase + (long long)(32 * t) * A.ld, vbase + (long long)(32 * t) * A.ld, A.ld, tid); };
    load_item(0, g0);
    if (NIT > 1) load_item(1, g1);
    f16x8 qb[4][2];
    float qinv;
    fa_load_bfrags(A.q + (row0 + i) * A.ld + h * AT_D, kg, qb, qinv);
    const int wi = i % BW, hi = (i / BW) % BH, ti = i / (BW * BH);
    float bt_[BT], bw_[4];
#pragma unroll
    for (int x = 0; x < BT; ++x) bt_[x] = A.dt[h * (2 * BT - 1) + ti - x + BT - 1] * FA_LOG2E;
#pragma unroll
    for (int r = 0; r < 4; ++r) bw_[r] = A.dw[h * (2 * BW - 1) + wi - GE::wj(r, kg) + BW - 1] * FA_LOG2E;
    // dh[hi - hj(x, kg) + BH - 1] of h slot x, read from LDS per tile (a 16-entry per-lane table would cost 16 registers)
    const float *dhl = sm.dhs + (hi + BH - 1 - (BW == 16 ? 0 : (kg >> 1)));
    constexpr int HSTEP = BW == 16 ? 1 : 2;
    const float qc = qinv * A.c1;
    fa_park(g0, sm.ring[0], sm.rs_inv[0], tid);
    __syncthreads();

    f32x4v oacc[AT_D / 16];
#pragma unroll
    for (int d = 0; d < AT_D / 16; ++d) oacc[d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, lsum = 0.f;
    float vmax_run = fa_u2f((unsigned)(FA_EMIN - 14) << 23);         // largest 2^(e - 14) over the V rows met so far

    static_for<NIT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const unsigned short *cur = sm.ring[t & 1];
        const float *cinv = sm.rs_inv[t & 1];
        if constexpr (t + 2 < NIT) load_item(t + 2, FA_G(t));
        // ---- S^T = K Q^T for the 32 keys of this chunk (two 16-key tiles) ----
        f32x4v st[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f16x8 a[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[kt][pl] = fa_rowfrag(cur, pl, 16 * kt + c16, s, kg);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[kt] = fa_mfma(a[kt][1], qb[s][0], st[kt]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[kt] = fa_mfma(a[kt][0], qb[s][1], st[kt]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[kt] = fa_mfma(a[kt][0], qb[s][0], st[kt]);
        }
        if constexpr (t + 1 < NIT) fa_park(FA_G(t + 1), sm.ring[(t + 1) & 1], sm.rs_inv[(t + 1) & 1], tid);
        // ---- scores (log2 units), online softmax ----
        float x[8];
        f32x4v vinv[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int T = 2 * t + kt;
            const f32x4v kinv = *reinterpret_cast<const f32x4v *>(cinv + 16 * kt + 4 * kg);
            vinv[kt] = *reinterpret_cast<const f32x4v *>(cinv + 32 + 16 * kt + 4 * kg);
            const float bth = bt_[T / HP] + dhl[-HSTEP * (T % HP)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fa_score(st[kt][r], kinv[r] * qc, bth + bw_[r]);
                if (MASKED && 16 * T + 4 * kg + r > i) v = A.fill2;
                x[4 * kt + r] = v;
            }
        }
        float cmax = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
        cmax = fa_kg_max(cmax);
        const float vmax_c = fa_kg_max(fmaxf(fa_max4(vinv[0]), fa_max4(vinv[1])));   // = 2^(E_c - 141)
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        const float vmax_new = fmaxf(vmax_run, vmax_c);
        // accumulated so far at weight 2^(141 - E_run): bring it to the new chunk scale (exact) and to the new row max
        const float fo = alpha * (vmax_run * __builtin_amdgcn_rcpf(vmax_new));
        vmax_run = vmax_new;
        const float srun = 32768.f * __builtin_amdgcn_rcpf(vmax_new);                 // 2^15 * 2^(141 - E_run), exact (powers of two)
        float psum = 0.f, pw[8];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(x[4 * kt + r] - m_new);
                psum += p;
                pw[4 * kt + r] = p * (vinv[kt][r] * srun);
            }
        lsum = lsum * alpha + psum;
        f16x8 pbh, pbl;
        fa_split8(pw, pbh, pbl);
#pragma unroll
        for (int d = 0; d < AT_D / 16; ++d) oacc[d] *= fo;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dp = 0; dp < AT_D / 32; ++dp) {
            f16x8 a[2][2];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[dd][pl] = fa_trfrag(cur, pl, 32, 2 * dp + dd, c16, kg);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) oacc[2 * dp + dd] = fa_mfma(a[dd][1], pbh, oacc[2 * dp + dd]);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) oacc[2 * dp + dd] = fa_mfma(a[dd][0], pbl, oacc[2 * dp + dd]);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) oacc[2 * dp + dd] = fa_mfma(a[dd][0], pbh, oacc[2 * dp + dd]);
        }
        __syncthreads();
    });
#undef FA_G
    const float linv = 1.f / fa_kg_sum(lsum);
    // sum_j p V = acc 2^(E_run - 156) = acc vmax_run 2^-15 (vmax_run = 2^(E_run - 141))
    const float osc = (vmax_run * (1.f / 32768.f)) * linv;
    float am = 0.f;
    {
        float *orow = A.o + (row0 + i) * A.ld + h * AT_D + 4 * kg;
#pragma unroll
        for (int d = 0; d < AT_D / 16; ++d) {
            const f32x4v ov = oacc[d] * osc;
            *reinterpret_cast<f32x4v *>(orow + 16 * d) = ov;
            am = fmaxf(am, fmaxf(fmaxf(fabsf(ov[0]), fabsf(ov[1])), fmaxf(fabsf(ov[2]), fabsf(ov[3]))));
        }
    }
    if (kg == 0) {
        const long long si = ((long long)b * A.H + h) * AT_S + i;
        A.m[si] = m_run;
        A.l[si] = linv;
    }
    return am;
}

template <int BT, int BH, int BW, int MASKED>
__global__ __launch_bounds__(512, 1) void lvt_attn_fwd_flash_kernel(const FaArgs A, float *__restrict__ o_amax) {
    __shared__ __attribute__((aligned(16))) FaSmem sm;
    // the two waves of a SIMD (w, w + 4) run the same MFMA -> VALU -> MFMA sequence between the same barriers: a static
    // priority for one of them staggers the pair, so that one wave's softmax / staging VALU runs beside the other's MFMAs
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
    float am;
    if (MASKED) {        // one workgroup = both query halves of a (sample, head): 8 + 4 key chunks
        am = fa_fwd_body<BT, BH, BW, MASKED, 8>(A, sm, blockIdx.x, 1);
        __syncthreads();
        am = fmaxf(am, fa_fwd_body<BT, BH, BW, MASKED, 4>(A, sm, fa_opaque(blockIdx.x), 0));
    } else {
        am = fa_fwd_body<BT, BH, BW, MASKED, 8>(A, sm, fa_pair(blockIdx.x), fa_half(blockIdx.x));
    }
    if (o_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, o_amax, reinterpret_cast<float *>(sm.ring[0]));
    }
}

// ===========================================================================================================================
// backward A: delta, dQ, bias-bank partial sums (query-stationary)
// ===========================================================================================================================
// ONEP = 1 (round 5): ONE pass over the keys.  delta_i = sum_j P_ij dP_ij = dO_i . O_i comes from the forward's output row (the
// identity every flash backward uses; O carries the same 2^-22-class error as the recomputed dP), and what pass 1 also supplied --
// the bound of g and the largest K-row scale, both needed to put g K into fp16 terms -- becomes ONE running power-of-two scale per
// query row: t_ij = g_ij 2^(e_Kj - 14) is formed unscaled, the chunk's max |t| moves the exponent, and dQ's accumulator is
// rescaled (exactly) when it moves, as the forward rescales O.  Three score-sized products instead of five, K and V staged once.
// What the first pass also bought is CONSISTENCY: with delta2_i = sum_j p_ij dP_ij from the very dP values of pass 2 the row sums of
// g vanish; dO . O differs from it by eps_i ~ 2^-22 sum_j |p dP| (p and dO are rounded to 22 bits at different places of the two
// evaluations), and -eps_i sum_j p_ij K_j is an error COMMON to a row of dQ -- 1.06x over the gradient tolerance of the first
// encoder layer's w_q (tests/test_gpu_vt.py, G12).  The pass therefore also accumulates C_i = sum_j p_ij K_j at fp16 precision
// (one MFMA per 16 dims on the hi plane of the K^T fragments that dQ loads anyway) and the realised row sum eps_i = sum_j g_ij,
// returns dQ_i - eps_i C_i / temper, and hands kernel B delta + eps (B then forms g as the two-pass form did).
#ifndef FA_EARLY_LOADS
#define FA_EARLY_LOADS 1                    // kernel A (0: its prologue requests q, dO, o one after the other, the round-5 order)
#endif
#ifndef FB_EARLY_LOADS
#define FB_EARLY_LOADS 0                    // kernel B: measured 161.4 / 158.5 -> 164.5 / 160.7 us with it (A: 198.8 / 196.5 -> 196.9 / 192.9)
#endif
#ifndef LVT_FA_A_CORR
#define LVT_FA_A_CORR 1                   // (timing builds: 0 = one pass without the C_i accumulation -- fails G12's w_q bound)
#endif
template <int BT, int BH, int BW, int MASKED, int NCH, int ONEP>
__device__ __forceinline__ float fa_bwd_a_body(const FaArgs &A, FaSmemA<BT + Geo16<BH, BW>::HP + 4> &sm, int bh_, int qhalf) {
    using GE = Geo16<BH, BW>;
    using BI = BankIdx<BT, BH, BW>;
    constexpr int HP = GE::HP;
    constexpr int NR = BT + HP + 4;
    static_assert(BT * BH * BW == AT_S, "256 tokens");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / A.H, h = bh_ % A.H;
    const long long row0 = (long long)b * AT_S;
    const int il = wave * 16 + c16, i = qhalf * 128 + il;
    const float *kbase = A.k + row0 * A.ld + h * AT_D, *vbase = A.v + row0 * A.ld + h * AT_D;
    constexpr int NIT = ONEP ? NCH : 2 * NCH;                        // two passes: chunks 0 .. NCH-1, then the same again

    if (tid < 2 * BH - 1) sm.dhs[tid] = A.dh[h * (2 * BH - 1) + tid] * FA_LOG2E;
    float *mine = sm.R + (il * 4 + kg) * NR;                          // this lane's class sums: [t index | h slot | register]
#pragma unroll
    for (int x = 0; x < NR; ++x) mine[x] = 0.f;
    G4 g0;                             // one register set: item t + 1 is split + stored during step t, item t + 2 loaded right after
    auto load_item = [&](int t, G4 &gg) {
        const int c = t < NCH ? t : t - NCH;
        fa_load(gg, kbase + (long long)(32 * c) * A.ld, vbase + (long long)(32 * c) * A.ld, A.ld, tid);
    };
    load_item(0, g0);
    f16x8 qb[4][2], dob[4][2];
    float qinv, doinv;
#if FA_EARLY_LOADS
    // q, dO and (one pass) the forward's output row of this lane's query are requested TOGETHER, then split: one round of
    // global-load latency instead of three dependent ones (q -> split, dO -> split, and after the first barrier o / dO again for
    // delta = dO . O, which reads exactly the 32 columns of the dO fragments)
    float4 qraw[8], doraw[8], oraw[ONEP ? 8 : 1];
    fa_bfrags_request(A.q + (row0 + i) * A.ld + h * AT_D, kg, qraw);
    fa_bfrags_request(A.d_o + (row0 + i) * A.ld + h * AT_D, kg, doraw);
    if constexpr (ONEP) fa_bfrags_request(A.o + (row0 + i) * A.ld + h * AT_D, kg, oraw);
    fa_bfrags_split(qraw, qb, qinv);
    fa_bfrags_split(doraw, dob, doinv);
    float delta_early = 0.f;
    if constexpr (ONEP) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const float4 o0 = oraw[2 * s_], o1 = oraw[2 * s_ + 1], d0 = doraw[2 * s_], d1 = doraw[2 * s_ + 1];
            a0 = fmaf(o0.x, d0.x, a0); a0 = fmaf(o0.y, d0.y, a0); a0 = fmaf(o0.z, d0.z, a0); a0 = fmaf(o0.w, d0.w, a0);
            a1 = fmaf(o1.x, d1.x, a1); a1 = fmaf(o1.y, d1.y, a1); a1 = fmaf(o1.z, d1.z, a1); a1 = fmaf(o1.w, d1.w, a1);
        }
        delta_early = fa_kg_sum(a0 + a1);
    }
#else
    fa_load_bfrags(A.q + (row0 + i) * A.ld + h * AT_D, kg, qb, qinv);
    fa_load_bfrags(A.d_o + (row0 + i) * A.ld + h * AT_D, kg, dob, doinv);
#endif
    const long long si = ((long long)b * A.H + h) * AT_S + i;
    const float m_i = A.m[si], linv = A.l[si];
    const int wi = i % BW, hi = (i / BW) % BH, ti = i / (BW * BH);
    float bt_[BT];
    f32x4v bw4;
#pragma unroll
    for (int x = 0; x < BT; ++x) bt_[x] = A.dt[h * (2 * BT - 1) + ti - x + BT - 1] * FA_LOG2E;
#pragma unroll
    for (int r = 0; r < 4; ++r) bw4[r] = A.dw[h * (2 * BW - 1) + wi - GE::wj(r, kg) + BW - 1] * FA_LOG2E;
    const float *dhl = sm.dhs + (hi + BH - 1 - (BW == 16 ? 0 : (kg >> 1)));
    constexpr int HSTEP = BW == 16 ? 1 : 2;
    const float qc = qinv * A.c1;
    fa_park(g0, sm.ring[0], sm.rs_inv[0], tid);
    load_item(1, g0);
    __syncthreads();

    float delta = 0.f, gmax = 0.f;
    float kmax = fa_u2f((unsigned)(FA_EMIN - 14) << 23);             // largest 2^(e - 14) over the K rows (pass 1)
    float sgf = 0.f, kmr = 0.f, dl = 0.f;                             // pass 2: scale of g from its bound, 1 / kmax (applied one
                                                                      // after the other: their product can leave the float range), delta / l
    int ebG = FA_EMIN;                                                // ONEP: the running exponent of max |g 2^(e_K - 14)| of this query row
#if FA_EARLY_LOADS
    if constexpr (ONEP) { delta = delta_early; dl = delta * linv; }
#else
    if constexpr (ONEP) {
        // delta_i = dO_i . O_i: the lane's 32 columns (those of its B fragments), then the four lanes of the column
        const float *orow = A.o + (row0 + i) * A.ld + h * AT_D + 8 * kg, *drow = A.d_o + (row0 + i) * A.ld + h * AT_D + 8 * kg;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const float4 o0 = *reinterpret_cast<const float4 *>(orow + 32 * s_), o1 = *reinterpret_cast<const float4 *>(orow + 32 * s_ + 4);
            const float4 d0 = *reinterpret_cast<const float4 *>(drow + 32 * s_), d1 = *reinterpret_cast<const float4 *>(drow + 32 * s_ + 4);
            a0 = fmaf(o0.x, d0.x, a0); a0 = fmaf(o0.y, d0.y, a0); a0 = fmaf(o0.z, d0.z, a0); a0 = fmaf(o0.w, d0.w, a0);
            a1 = fmaf(o1.x, d1.x, a1); a1 = fmaf(o1.y, d1.y, a1); a1 = fmaf(o1.z, d1.z, a1); a1 = fmaf(o1.w, d1.w, a1);
        }
        delta = fa_kg_sum(a0 + a1);
        dl = delta * linv;
    }
#endif
    f32x4v qacc[AT_D / 16];
#pragma unroll
    for (int d = 0; d < AT_D / 16; ++d) qacc[d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    constexpr bool CORR = ONEP && LVT_FA_A_CORR;
    f32x4v cacc[CORR ? AT_D / 16 : 1];                               // ONEP: C_i = sum_j p_ij K_j under the scale 2^(268 - ebC)
#pragma unroll
    for (int d = 0; d < (CORR ? AT_D / 16 : 1); ++d) cacc[d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    int ebC = 16;                                                     // running max of the exponent field of the K rows' 2^(e - 14)
    float rst[BT], rsw[4];                                            // class sums over the t index and the register; h slots: LDS
#pragma unroll
    for (int x = 0; x < BT; ++x) rst[x] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rsw[r] = 0.f;

    static_for<NIT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int c = t < NCH ? t : t - NCH;
        constexpr bool PASS2 = ONEP || t >= NCH;
        const unsigned short *cur = sm.ring[t & 1];
        const float *cinv = sm.rs_inv[t & 1];
        constexpr bool PARK = t + 1 < NIT;
        unsigned short *nslot = sm.ring[(t + 1) & 1];
        float *ninv = sm.rs_inv[(t + 1) & 1];
        if constexpr (!ONEP && t == NCH) {                            // between the passes: delta, the scale bound of g, the bounds for B
            delta = fa_kg_sum(delta) * linv;
            gmax = fa_kg_max(gmax);
            kmax = fa_kg_max(kmax);
            const float G = gmax + fabsf(delta);                      // >= |p (dP - delta)|: p <= 1
            ebG = fa_ebits(G);
            sgf = fa_u2f((unsigned)(268 - ebG) << 23);
            kmr = __builtin_amdgcn_rcpf(kmax);
            dl = delta * linv;
            // per-workgroup bounds for kernel B: max 2^(e - 14) over the dO rows, max of G_i 2^(e_i - 14) over the q rows
            float r0 = doinv, r1 = G * qinv;
#pragma unroll
            for (int dd = 32; dd > 0; dd >>= 1) { r0 = fmaxf(r0, __shfl_xor(r0, dd, 64)); r1 = fmaxf(r1, __shfl_xor(r1, dd, 64)); }
            if (lane == 0) { sm.red[wave] = r0; sm.red[8 + wave] = r1; }
        }
        // ---- S^T = K Q^T, dP^T = V dO^T for the 32 keys of this chunk; the split + store of item t + 1 in four pieces between
        // the MFMAs of the four k steps ----
        f32x4v st[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
        f32x4v dp[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
        int eb0 = 0, eb1 = 0;
        static_for<4>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            f16x8 a[2][2], av[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    a[kt][pl] = fa_rowfrag(cur, pl, 16 * kt + c16, s, kg);
                    av[kt][pl] = fa_rowfrag(cur, pl, 32 + 16 * kt + c16, s, kg);
                }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { st[kt] = fa_mfma(a[kt][1], qb[s][0], st[kt]); dp[kt] = fa_mfma(av[kt][1], dob[s][0], dp[kt]); }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { st[kt] = fa_mfma(a[kt][0], qb[s][1], st[kt]); dp[kt] = fa_mfma(av[kt][0], dob[s][1], dp[kt]); }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { st[kt] = fa_mfma(a[kt][0], qb[s][0], st[kt]); dp[kt] = fa_mfma(av[kt][0], dob[s][0], dp[kt]); }
            if constexpr (PARK) {
                if constexpr (s == 0) eb0 = fa_row_scale(g0.v[0], g0.v[1]);
                if constexpr (s == 1) fa_row_store(g0.v[0], g0.v[1], eb0, nslot, ninv, tid >> 4, tid & 15);
                if constexpr (s == 2) eb1 = fa_row_scale(g0.v[2], g0.v[3]);
                if constexpr (s == 3) fa_row_store(g0.v[2], g0.v[3], eb1, nslot, ninv, 32 + (tid >> 4), tid & 15);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (s & 1) ? 3 : 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if constexpr (t + 2 < NIT) load_item(t + 2, g0);
        float w[8];
        u32x4 pku = {0u, 0u, 0u, 0u};                                  // CORR: the eight p 2^(e_K - 14) scc of the lane, fp16 pairs
        if constexpr (CORR) {
            // the chunk's largest K-row scale (all 32 rows: this lane's eight + the three other lanes of the column)
            const f32x4v k0 = *reinterpret_cast<const f32x4v *>(cinv + 4 * kg), k1 = *reinterpret_cast<const f32x4v *>(cinv + 16 + 4 * kg);
            const int ebN = max(ebC, (int)(__float_as_uint(fa_kg_max(fmaxf(fa_max4(k0), fa_max4(k1)))) >> 23));
            if (__builtin_amdgcn_ballot_w64(ebN != ebC) != 0) {
                const float f = fa_exp_field(127 + ebC - ebN);
#pragma unroll
                for (int d = 0; d < AT_D / 16; ++d) cacc[d] = cacc[d] * f;
                ebC = ebN;
            }
        }
        const float scc = CORR ? fa_u2f((unsigned)(268 - ebC) << 23) * linv : 0.f;      // p 2^(e_K - 14) scc <= 2^14
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int T = 2 * c + kt;
            const f32x4v kinv = *reinterpret_cast<const f32x4v *>(cinv + 16 * kt + 4 * kg);
            const f32x4v vinv = *reinterpret_cast<const f32x4v *>(cinv + 32 + 16 * kt + 4 * kg);
            const float bth = bt_[T / HP] + dhl[-HSTEP * (T % HP)];
            if constexpr (!PASS2) kmax = fmaxf(kmax, fa_max4(kinv));
            const f32x4v qk = kinv * qc, bm = bw4 + bth;
            const f32x4v dv = vinv * (PASS2 ? doinv * linv : doinv), kw = kinv * kmr;         // kw <= 1
            float gt = 0.f;
            const f32x4v ks = kinv * scc;
            f32x4v pkv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = fa_score(st[kt][r], qk[r], bm[r]);
                if (MASKED && 16 * T + 4 * kg + r > i) x = A.fill2;
                // (x - m as its own operation: folding m into the bias rounds every score at the size of m before the
                // subtraction -- dq of the heavy-tailed accuracy class went from 1.0x to 3.4x the fp32 evaluation's error)
                const float ex = __builtin_amdgcn_exp2f(x - m_i);     // p = ex linv
                if constexpr (!PASS2) {
                    const float dpv = dp[kt][r] * dv[r];
                    delta = fmaf(ex, dpv, delta);
                    gmax = fmaxf(gmax, fabsf(dpv));
                } else {
                    const float g = ex * fmaf(dp[kt][r], dv[r], -dl);  // p (dP - delta)
                    gt += g;
                    rsw[r] += g;
                    if constexpr (ONEP) { w[4 * kt + r] = g * kinv[r]; gmax = fmaxf(gmax, fabsf(g)); if constexpr (CORR) pkv[r] = ex * ks[r]; }
                    else w[4 * kt + r] = (g * sgf) * kw[r];
                }
            }
            if constexpr (PASS2) { rst[T / HP] += gt; mine[BT + T % HP] += gt; }
            if constexpr (CORR) {
                pku[2 * kt] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{pkv[0], pkv[1]}, f16x2v));
                pku[2 * kt + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{pkv[2], pkv[3]}, f16x2v));
            }
        }
        if constexpr (PASS2) {
            if constexpr (ONEP) {
                float tm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) tm = fmaxf(tm, fabsf(w[e]));
                const int ebN = max(ebG, fa_ebits(fa_kg_max(tm)));
                if (__builtin_amdgcn_ballot_w64(ebN != ebG) != 0) {   // (rare after the first chunks; exact: a power of two)
                    const float f = fa_exp_field(127 + ebG - ebN);
#pragma unroll
                    for (int d = 0; d < AT_D / 16; ++d) qacc[d] = qacc[d] * f;
                    ebG = ebN;
                }
                const float sct = fa_u2f((unsigned)(268 - ebG) << 23);       // max |w| in [2^14, 2^15)
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] *= sct;
            }
            f16x8 wbh, wbl;
            fa_split8(w, wbh, wbl);
            const f16x8 pcb = __builtin_bit_cast(f16x8, pku);
            // ---- dQ^T += K^T g^T ----
#pragma unroll
            for (int dq_ = 0; dq_ < AT_D / 32; ++dq_) {
                f16x8 a[2][2];
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) a[dd][pl] = fa_trfrag(cur, pl, 0, 2 * dq_ + dd, c16, kg);
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) qacc[2 * dq_ + dd] = fa_mfma(a[dd][1], wbh, qacc[2 * dq_ + dd]);
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) qacc[2 * dq_ + dd] = fa_mfma(a[dd][0], wbl, qacc[2 * dq_ + dd]);
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) qacc[2 * dq_ + dd] = fa_mfma(a[dd][0], wbh, qacc[2 * dq_ + dd]);
                if constexpr (CORR) {
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd) cacc[2 * dq_ + dd] = fa_mfma(a[dd][0], pcb, cacc[2 * dq_ + dd]);
                }
            }
        }
        __syncthreads();
    });
    // acc = sum_j g K 2^(14 - E_K) 2^(141 - ebG): dq = acc inv_temper kmax 2^(ebG - 141)   (kmax = 2^(E_K - 14), E_K unbiased)
    // (ONEP: acc = sum_j g 2^(e_K - 14) K~ 2^(141 - ebG), no common K scale)
    if constexpr (ONEP) {
        // per-workgroup bounds for kernel B: max 2^(e - 14) over the dO rows, max of max_j |g_ij| 2^(e_i - 14) over the q rows
        float r0 = doinv, r1 = fa_kg_max(gmax) * qinv;
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) { r0 = fmaxf(r0, __shfl_xor(r0, dd, 64)); r1 = fmaxf(r1, __shfl_xor(r1, dd, 64)); }
        if (lane == 0) { sm.red[wave] = r0; sm.red[8 + wave] = r1; }
        __syncthreads();
    }
    const float f1 = ONEP ? 1.f : kmax, f2 = fa_exp_field(ebG - 14) * A.inv_temper;
    // ONEP: eps_i = sum_j g_ij (every g sits in exactly one rsw register of one of the column's four lanes); C = cacc 2^(ebC - 141)
    float epsc = 0.f;
    if constexpr (CORR) {
        const float eps = fa_kg_sum((rsw[0] + rsw[1]) + (rsw[2] + rsw[3]));
        delta += eps;
        epsc = (eps * fa_exp_field(ebC - 14)) * A.inv_temper;
    }
    float am = 0.f;
    {
        float *qrow = A.dq + (row0 + i) * A.ld + h * AT_D + 4 * kg;
#pragma unroll
        for (int d = 0; d < AT_D / 16; ++d) {
            f32x4v ov = (qacc[d] * f1) * f2;
            if constexpr (CORR) ov = ov - cacc[d] * epsc;
            *reinterpret_cast<f32x4v *>(qrow + 16 * d) = ov;
            am = fmaxf(am, fmaxf(fmaxf(fabsf(ov[0]), fabsf(ov[1])), fmaxf(fabsf(ov[2]), fabsf(ov[3]))));
        }
    }
    if (kg == 0) A.delta[si] = delta;
    if (tid == 0) {
        float r0 = sm.red[0], r1 = sm.red[8];
#pragma unroll
        for (int x = 1; x < 8; ++x) { r0 = fmaxf(r0, sm.red[x]); r1 = fmaxf(r1, sm.red[8 + x]); }
        A.scal[(bh_ * 2 + qhalf) * 2] = r0;
        A.scal[(bh_ * 2 + qhalf) * 2 + 1] = r1;
    }
    // ---- bias-bank gradient of this (sample, head, query half): two fixed-order stages through LDS (attention_pipe.hip) ----
    float *R = sm.R;                                                  // [128 queries][4 kg][NR]
    float *R2 = reinterpret_cast<float *>(sm.ring[0]);                // [8 parts][NB]
#pragma unroll
    for (int x = 0; x < BT; ++x) mine[x] = rst[x];
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[BT + HP + r] = rsw[r];
    __syncthreads();
    {
        const int e = tid & 63, part = tid >> 6;
        if (e < BI::NB) {
            float acc = 0.f;
            for (int q = 16 * part; q < 16 * part + 16; ++q) {
                const int iq = qhalf * 128 + q;
                const int wq = iq % BW, hq = (iq / BW) % BH, tq = iq / (BW * BH);
                const float *r = R + (q * 4) * NR;
                if (e < BI::NT) {
                    const int tj = tq - e + BT - 1;
                    if (tj >= 0 && tj < BT) acc += (r[tj] + r[NR + tj]) + (r[2 * NR + tj] + r[3 * NR + tj]);
                } else if (e < BI::NT + BI::NH) {
                    const int hj = hq - (e - BI::NT) + BH - 1;
                    if (hj >= 0 && hj < BH) {
                        if (BW == 16) acc += (r[BT + hj] + r[NR + BT + hj]) + (r[2 * NR + BT + hj] + r[3 * NR + BT + hj]);
                        else acc += r[(2 * (hj & 1)) * NR + BT + (hj >> 1)] + r[(2 * (hj & 1) + 1) * NR + BT + (hj >> 1)];
                    }
                } else {
                    const int wj = wq - (e - BI::NT - BI::NH) + BW - 1;
                    if (wj >= 0 && wj < BW) {
                        if (BW == 16) acc += r[(wj >> 2) * NR + BT + HP + (wj & 3)];
                        else acc += r[(wj >> 2) * NR + BT + HP + (wj & 3)] + r[((wj >> 2) + 2) * NR + BT + HP + (wj & 3)];
                    }
                }
            }
            R2[part * BI::NB + e] = acc;
        }
    }
    __syncthreads();
    if (tid < BI::NB) {
        float acc = 0.f;
#pragma unroll
        for (int part = 0; part < 8; ++part) acc += R2[part * BI::NB + tid];
        A.bank_partial[((long long)bh_ * 2 + qhalf) * BI::NB + tid] = acc;
    }
    return am;
}

template <int BT, int BH, int BW, int MASKED, int ONEP>
__global__ __launch_bounds__(512, 1) void lvt_attn_bwd_flash_a_kernel(const FaArgs A, float *__restrict__ d_amax) {
    __shared__ __attribute__((aligned(16))) FaSmemA<BT + Geo16<BH, BW>::HP + 4> sm;
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);           // (see the forward kernel)
    float am;
    if (MASKED) {        // one workgroup = both query halves of a (sample, head): 8 + 4 key chunks (as separate workgroups: 219 us against 182)
        am = fa_bwd_a_body<BT, BH, BW, MASKED, 8, ONEP>(A, sm, blockIdx.x, 1);
        __syncthreads();
        am = fmaxf(am, fa_bwd_a_body<BT, BH, BW, MASKED, 4, ONEP>(A, sm, fa_opaque(blockIdx.x), 0));
    } else {
        am = fa_bwd_a_body<BT, BH, BW, MASKED, 8, ONEP>(A, sm, fa_pair(blockIdx.x), fa_half(blockIdx.x));
    }
    if (d_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, d_amax, reinterpret_cast<float *>(sm.ring[0]));
    }
}

// ===========================================================================================================================
// backward B: dK, dV (key-stationary): one wave = 16 keys (K fragments in registers, the workgroup's 128 V rows resident in
// LDS), 32-query chunks of Q | dO through the two-slot ring.  Per step: S = Q K^T and dP = dO V^T (48 MFMAs, the split +
// store of the next item spread between them), the element-wise work (scores, p, g, the two B operands), then dV^T += dO^T P
// and dK^T += Q^T g (48 MFMAs on transposed fragments).
// (A four-wave form -- one wave per SIMD with 512 registers, K and V fragments in registers, three slots, chunk t + 1's score
// products software-pipelined beside chunk t's element-wise work -- was built and measured: 241 us against 178 for this form.
// With one workgroup per CU its eight rounds of prologues are exposed, and a single in-order wave overlaps its own VALU and
// MFMA work far less than the instruction mix suggests: timing builds put the element-wise work at 91 us, the staging
// arithmetic at 51, all MFMAs at 59 and the bare loads / fragment reads / barriers at 72; profiles/r05_attn_flash_b4_ablation.txt.)
// ===========================================================================================================================
struct FaSmemB {
    unsigned short ring[2][FA_SLOT];
    unsigned short vres[2][FA_SLOT];      // the workgroup's 128 V rows, resident
    float rs_inv[2][80];
    float vinv[2][80];
    float dhs[32];
};

template <int BT, int BH, int BW, int MASKED, int C0, int NCH>
__device__ __forceinline__ float fa_bwd_b_body(const FaArgs &A, FaSmemB &sm, int bh_, int khalf) {
    using GE = Geo16<BH, BW>;
    constexpr int HP = GE::HP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / A.H, h = bh_ % A.H;
    const long long row0 = (long long)b * AT_S;
    const int jl = wave * 16 + c16, j = khalf * 128 + jl;             // this lane's key (tile column)
    const float *qbase = A.q + row0 * A.ld + h * AT_D, *dobase = A.d_o + row0 * A.ld + h * AT_D;
    const long long sb = ((long long)b * A.H + h) * AT_S;
    constexpr int NIT = NCH;

    if (tid < 2 * BH - 1) sm.dhs[tid] = A.dh[h * (2 * BH - 1) + tid] * FA_LOG2E;
    G4 g0;                             // one register set: item t + 1 is split + stored during step t, item t + 2 loaded right after
    auto load_item = [&](int t, G4 &gg) {
        fa_load(gg, qbase + (long long)(32 * (C0 + t)) * A.ld, dobase + (long long)(32 * (C0 + t)) * A.ld, A.ld, tid);
    };
    float4 kraw[8];
    {                                   // the workgroup's 128 V rows -> LDS, resident
        const float *rows = A.v + (row0 + khalf * 128) * A.ld + h * AT_D;
        G4 a, bq;
        fa_load(a, rows, rows + 32 * A.ld, A.ld, tid);
        fa_load(bq, rows + 64 * A.ld, rows + 96 * A.ld, A.ld, tid);
#if FB_EARLY_LOADS
        // the first Q | dO item and the wave's K rows are requested BEFORE the V rows are split and stored: one round of global-load
        // latency per workgroup instead of two (four workgroups per CU and launch, nothing beside a prologue to hide it)
        load_item(0, g0);
        fa_bfrags_request(A.k + (row0 + j) * A.ld + h * AT_D, kg, kraw);
#endif
#ifdef FB_X_NOVSTAGE
        if (A.H < 0)
#endif
        { fa_park(a, sm.vres[0], sm.vinv[0], tid); fa_park(bq, sm.vres[1], sm.vinv[1], tid); }
    }
    f16x8 kb[4][2];
    float kinv;
#if FB_EARLY_LOADS
    fa_bfrags_split(kraw, kb, kinv);
#else
    load_item(0, g0);
    fa_load_bfrags(A.k + (row0 + j) * A.ld + h * AT_D, kg, kb, kinv);
#endif
    // bounds from kernel A (both query halves): exponent of the dO rows, bound of g 2^(e_q - 14)
    const float *sc = A.scal + (long long)bh_ * 4;
    const float domax = fmaxf(sc[0], sc[2]), umax = fmaxf(sc[1], sc[3]);
    const float srun = 32768.f * __builtin_amdgcn_rcpf(domax);        // 2^15 2^(141 - E)
    const int zb = fa_ebits(umax);
    const float zrun = fa_u2f((unsigned)(268 - zb) << 23);
    const unsigned short *vs = sm.vres[jl >> 6];
    const int wj = j % BW, hj = (j / BW) % BH, tj = j / (BW * BH);
    float bt_[BT];                        // indexed by the QUERY's t index (Geo16 of the varying token)
    f32x4v bw4;
#pragma unroll
    for (int x = 0; x < BT; ++x) bt_[x] = A.dt[h * (2 * BT - 1) + x - tj + BT - 1] * FA_LOG2E;
#pragma unroll
    for (int r = 0; r < 4; ++r) bw4[r] = A.dw[h * (2 * BW - 1) + GE::wj(r, kg) - wj + BW - 1] * FA_LOG2E;
    const float *dhl = sm.dhs + (BH - 1 - hj + (BW == 16 ? 0 : (kg >> 1)));      // dh[hj(x, kg) - hj + BH - 1] of the query's h slot x
    constexpr int HSTEP = BW == 16 ? 1 : 2;
    const float kc = kinv * A.c1;
    fa_park(g0, sm.ring[0], sm.rs_inv[0], tid);
    if (NIT > 1) load_item(1, g0);
    __syncthreads();
    const float vinv = sm.vinv[jl >> 6][jl & 63];

    f32x4v accv[AT_D / 16], acck[AT_D / 16];
#pragma unroll
    for (int d = 0; d < AT_D / 16; ++d) { accv[d] = f32x4v{0.f, 0.f, 0.f, 0.f}; acck[d] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    static_for<NIT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int c = C0 + t;                                     // 32-query chunk of the (sample, head)
        constexpr bool PARK = t + 1 < NIT;
        const unsigned short *cur = sm.ring[t & 1];
        const float *cinv = sm.rs_inv[t & 1];
        unsigned short *nslot = sm.ring[(t + 1) & 1];
        float *ninv = sm.rs_inv[(t + 1) & 1];
        f32x4v m4[2], l4[2], d4[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const long long o = sb + 32 * c + 16 * qt + 4 * kg;
            m4[qt] = *reinterpret_cast<const f32x4v *>(A.m + o);
            l4[qt] = *reinterpret_cast<const f32x4v *>(A.l + o);
            d4[qt] = *reinterpret_cast<const f32x4v *>(A.delta + o);
        }
        // ---- S = Q K^T, dP = dO V^T for the 32 queries of this chunk (rows) x the wave's 16 keys (columns); the split + store
        // of item t + 1 in four pieces between the MFMAs of the four k steps ----
        f32x4v st[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
        f32x4v dp[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
        int eb0 = 0, eb1 = 0;
        // Fragment reads are software-pipelined by HALF k steps: while the six S MFMAs of step s issue, the dO / V fragments of
        // step s are in flight; while the six dP MFMAs issue, the Q fragments of step s + 1.  (Reading all ten fragments of a k
        // step and then multiplying made the LDS and the matrix pipe take turns: all eight waves read at once -- 320 LDS
        // cycles -- then multiply -- 2 x 192 cycles per SIMD.)
        f16x8 aq[2][2][2], ad[2][2], vb[2];
        auto load_s = [&](int s_, int bf) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) aq[bf][qt][pl] = fa_rowfrag(cur, pl, 16 * qt + c16, s_, kg);
#ifdef FB_X_NOFRAG1
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) aq[bf][qt][pl] = kb[s_ & 3][pl];
#endif
        };
        auto load_p = [&](int s_) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) vb[pl] = fa_rowfrag(vs, pl, jl & 63, s_, kg);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ad[qt][pl] = fa_rowfrag(cur, pl, 32 + 16 * qt + c16, s_, kg);
#ifdef FB_X_NOFRAG1
            vb[0] = kb[s_ & 3][1]; vb[1] = kb[s_ & 3][0];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ad[qt][pl] = kb[(s_ + 1) & 3][pl];
#endif
        };
        load_s(0, 0);
        static_for<4>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value, bf = s & 1;
            load_p(s);
#ifdef FB_X_NOMFMA1
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) { st[qt] += __builtin_bit_cast(f32x4v, aq[bf][qt][1]) + __builtin_bit_cast(f32x4v, aq[bf][qt][0]); dp[qt] += __builtin_bit_cast(f32x4v, ad[qt][0]) + __builtin_bit_cast(f32x4v, ad[qt][1]) + __builtin_bit_cast(f32x4v, vb[0]) + __builtin_bit_cast(f32x4v, vb[1]); }
            if constexpr (s < 3) load_s(s + 1, bf ^ 1);
#else
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) st[qt] = fa_mfma(aq[bf][qt][1], kb[s][0], st[qt]);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) st[qt] = fa_mfma(aq[bf][qt][0], kb[s][1], st[qt]);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) st[qt] = fa_mfma(aq[bf][qt][0], kb[s][0], st[qt]);
            if constexpr (s < 3) load_s(s + 1, bf ^ 1);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dp[qt] = fa_mfma(ad[qt][1], vb[0], dp[qt]);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dp[qt] = fa_mfma(ad[qt][0], vb[1], dp[qt]);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dp[qt] = fa_mfma(ad[qt][0], vb[0], dp[qt]);
#endif
#ifndef FB_X_NOPARK
            if constexpr (PARK) {
                if constexpr (s == 0) eb0 = fa_row_scale(g0.v[0], g0.v[1]);
                if constexpr (s == 1) fa_row_store(g0.v[0], g0.v[1], eb0, nslot, ninv, tid >> 4, tid & 15);
                if constexpr (s == 2) eb1 = fa_row_scale(g0.v[2], g0.v[3]);
                if constexpr (s == 3) fa_row_store(g0.v[2], g0.v[3], eb1, nslot, ninv, 32 + (tid >> 4), tid & 15);
            }
#endif
            // pin the order: [dO / V reads] [S MFMAs + staging arithmetic] [next Q reads] [dP MFMAs + staging arithmetic]
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (PARK) __builtin_amdgcn_sched_group_barrier(0x002, (s & 1) ? 3 : 2, 0);
            }
            if constexpr (s < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (PARK) __builtin_amdgcn_sched_group_barrier(0x002, (s & 1) ? 3 : 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (t + 2 < NIT) load_item(t + 2, g0);
        // ---- element-wise: per tile the products of the row / column factors, per element five operations + the mask ----
        float pw[8], u[8];
#ifdef FB_X_NOEW
#pragma unroll
        for (int e = 0; e < 8; ++e) { pw[e] = st[e >> 2][e & 3]; u[e] = dp[e >> 2][e & 3]; }
#else
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int T = 2 * c + qt;                                 // 16-query tile of the (sample, head)
            const f32x4v qinv = *reinterpret_cast<const f32x4v *>(cinv + 16 * qt + 4 * kg);
            const f32x4v doinv = *reinterpret_cast<const f32x4v *>(cinv + 32 + 16 * qt + 4 * kg);
            const float bth = bt_[T / HP] + dhl[HSTEP * (T % HP)];
            const f32x4v qk = qinv * kc, bm = bw4 + bth;
            const f32x4v lds = l4[qt] * (doinv * srun);               // P^T dO operand: p 2^(e_i - 14) 2^15 2^(141 - E) = exp lds
            const f32x4v qzl = l4[qt] * qinv;                         // g^T Q operand: g 2^(e_i - 14) 2^(141 - zb) = (exp (dp dvq - dq)) zrun
            const f32x4v dvq = (doinv * vinv) * qzl, dq_ = d4[qt] * qzl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = fa_score(st[qt][r], qk[r], bm[r]);
                if (MASKED && j > 16 * T + 4 * kg + r) x = A.fill2;
                const float ex = __builtin_amdgcn_exp2f(x - m4[qt][r]);
                pw[4 * qt + r] = ex * lds[r];
                u[4 * qt + r] = (ex * fmaf(dp[qt][r], dvq[r], -dq_[r])) * zrun;     // (zrun last: |g 2^(e_i - 14)| <= the bound it comes from)
            }
        }
#endif
        f16x8 pbh, pbl, ubh, ubl;
        fa_split8(pw, pbh, pbl);
        fa_split8(u, ubh, ubl);
        // ---- dV^T += dO^T P, dK^T += Q^T g: the transposed fragments pipelined the same way (Q^T of tile d in flight beside the dV
        // MFMAs of tile d, dO^T of tile d + 1 beside its dK MFMAs) ----
        f16x8 ao[2][2], aqt[2];
        auto load_o = [&](int d_, int bf) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ao[bf][pl] = fa_trfrag(cur, pl, 32, d_, c16, kg);
#ifdef FB_X_NOFRAG3
            ao[bf][0] = kb[d_ & 3][0]; ao[bf][1] = kb[d_ & 3][1];
#endif
        };
        load_o(0, 0);
        static_for<AT_D / 16>([&](auto dc) {
            constexpr int d = decltype(dc)::value, bf = d & 1;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) aqt[pl] = fa_trfrag(cur, pl, 0, d, c16, kg);
#ifdef FB_X_NOFRAG3
            aqt[0] = kb[(d + 1) & 3][0]; aqt[1] = kb[(d + 1) & 3][1];
#endif
#ifdef FB_X_NOMFMA3
            accv[d] += __builtin_bit_cast(f32x4v, ao[bf][1]) + __builtin_bit_cast(f32x4v, ao[bf][0]) + __builtin_bit_cast(f32x4v, pbh) + __builtin_bit_cast(f32x4v, pbl);
            if constexpr (d + 1 < AT_D / 16) load_o(d + 1, bf ^ 1);
            acck[d] += __builtin_bit_cast(f32x4v, aqt[1]) + __builtin_bit_cast(f32x4v, aqt[0]) + __builtin_bit_cast(f32x4v, ubh) + __builtin_bit_cast(f32x4v, ubl);
#else
            accv[d] = fa_mfma(ao[bf][1], pbh, accv[d]);
            accv[d] = fa_mfma(ao[bf][0], pbl, accv[d]);
            accv[d] = fa_mfma(ao[bf][0], pbh, accv[d]);
            if constexpr (d + 1 < AT_D / 16) load_o(d + 1, bf ^ 1);
            acck[d] = fa_mfma(aqt[1], ubh, acck[d]);
            acck[d] = fa_mfma(aqt[0], ubl, acck[d]);
            acck[d] = fa_mfma(aqt[0], ubh, acck[d]);
#endif
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if constexpr (d + 1 < AT_D / 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
    });
    // accv = sum_i p dO 2^15 2^(141 - E) = sum / (srun): dv = accv domax / 2^15;  acck = sum_i g Q 2^(141 - zb): dk = acck inv_temper 2^(zb - 141)
    const float fvs = domax * (1.f / 32768.f), fks = fa_exp_field(zb - 14) * A.inv_temper;
    float am = 0.f;
    {
        float *vrow = A.dv + (row0 + j) * A.ld + h * AT_D + 4 * kg, *krow = A.dk + (row0 + j) * A.ld + h * AT_D + 4 * kg;
#pragma unroll
        for (int d = 0; d < AT_D / 16; ++d) {
            const f32x4v ov = accv[d] * fvs, ok = acck[d] * fks;
#ifdef FB_X_NOSTORE
            if (A.H < 0)
#endif
            { *reinterpret_cast<f32x4v *>(vrow + 16 * d) = ov; *reinterpret_cast<f32x4v *>(krow + 16 * d) = ok; }
#pragma unroll
            for (int e = 0; e < 4; ++e) am = fmaxf(am, fmaxf(fabsf(ov[e]), fabsf(ok[e])));
        }
    }
    return am;
}

template <int BT, int BH, int BW, int MASKED>
__global__ __launch_bounds__(512, 1) void lvt_attn_bwd_flash_b_kernel(const FaArgs A, float *__restrict__ d_amax) {
    __shared__ __attribute__((aligned(16))) FaSmemB sm;
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);           // (see the forward kernel)
    // (A persistent form -- one workgroup per CU walking its items, the stores of item i draining beside the prologue of item
    // i + 1 -- measured the same 175 us: the ~70 us that a timing build with all step work removed still takes are not dispatch.)
    float am;
    if (MASKED) {        // key half 0 meets all eight query chunks, key half 1 the last four: separate workgroups (one workgroup
                         // running both halves back to back spilled ~95 registers inside its step loops)
        if (fa_half(blockIdx.x) == 0) am = fa_bwd_b_body<BT, BH, BW, MASKED, 0, 8>(A, sm, fa_pair(blockIdx.x), 0);
        else am = fa_bwd_b_body<BT, BH, BW, MASKED, 4, 4>(A, sm, fa_pair(blockIdx.x), 1);
    } else {
        am = fa_bwd_b_body<BT, BH, BW, MASKED, 0, 8>(A, sm, fa_pair(blockIdx.x), fa_half(blockIdx.x));
    }
    if (d_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, d_amax, reinterpret_cast<float *>(sm.ring[0]));
    }
}

// bank gradients: out[h][e] = sum over (sample, query half) of the workgroup partials, fixed order (attention_pipe.hip)
__global__ __launch_bounds__(256) void lvt_attn_flash_bank_reduce_kernel(const float *__restrict__ partial, int B, int H, int NB, int nt,
                                                                         int nh, float *__restrict__ ddt, float *__restrict__ ddh,
                                                                         float *__restrict__ ddw) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= H * NB) return;
    const int h = idx / NB, e = idx % NB;
    float s = 0.f;
    for (int jj = lane; jj < 2 * B; jj += 64) s += partial[(((long long)(jj >> 1) * H + h) * 2 + (jj & 1)) * NB + e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane) return;
    // A bank with ONE entry (the t bank of the (1, 16, 16) blocks) adds the same constant to every score of a row: softmax is
    // shift-invariant, its gradient sum_ij g_ij is identically zero and what the sum holds is the rounding residue of the row
    // cancellations (p (dP - delta) summed over j).  RMSprop divides a gradient by its own running magnitude, i.e. turns any
    // residue above its eps into lr-sized steps of a parameter that cannot change the output; the residue of the one-pass form
    // (delta from dO . O) is a few times the two-pass form's.  The exact value is returned instead.
    const int nw = NB - nt - nh;
    if (e < nt) ddt[h * nt + e] = nt == 1 ? 0.f : s;
    else if (e < nt + nh) ddh[h * nh + e - nt] = nh == 1 ? 0.f : s;
    else ddw[h * nw + e - nt - nh] = nw == 1 ? 0.f : s;
}

}  // namespace

extern "C" int lvt_attn_flash_supported(int S, int da, int bt, int bh, int bw) {
    return S == AT_S && da == AT_D && ((bt == 1 && bh == 16 && bw == 16) || (bt == 4 && bh == 8 && bw == 8)) ? 1 : 0;
}

#define LVT_FA_GEOMS(X) X(1, 16, 16) X(4, 8, 8)

extern "C" int lvt_attn_fwd_flash(const float *q, const float *k, const float *v, long long ld, int B, int H, int S, int da,
                                  float temper, const float *dt, const float *dh, const float *dw, int bt, int bh, int bw,
                                  int masked, float fill, float *o, float *stats, float *o_amax, void *stream) {
    LVT_REQUIRE(q && k && v && dt && dh && dw && o && stats && B > 0 && H > 0, "attn_fwd_flash: bad args");
    LVT_REQUIRE((B * H) % 8 == 0, "attn_fwd_flash: B * H = %d must be a multiple of 8 (workgroup pairing per XCD)", B * H);
    LVT_REQUIRE(lvt_attn_flash_supported(S, da, bt, bh, bw), "attn_fwd_flash: S=%d da=%d block (%d,%d,%d) has no instantiation", S, da, bt, bh, bw);
    LVT_REQUIRE(lvt_aligned16(q) && lvt_aligned16(k) && lvt_aligned16(v) && lvt_aligned16(o) && lvt_aligned16(stats) && ld % 4 == 0 &&
                ld >= (long long)H * da, "attn_fwd_flash: alignment / row stride");
    FaArgs A = {};
    A.q = q; A.k = k; A.v = v; A.ld = ld; A.H = H; A.inv_temper = 1.f / temper; A.c1 = FA_LOG2E / temper; A.fill2 = fill * FA_LOG2E;
    A.dt = dt; A.dh = dh; A.dw = dw; A.o = o; A.m = stats; A.l = stats + (size_t)B * H * S;
    const dim3 grid((unsigned)(B * H * (masked ? 1 : 2))), blk(512);
    hipStream_t s = (hipStream_t)stream;
#define LVT_X(BT, BH, BW)                                                                                       \
    if (bt == BT && bh == BH && bw == BW) {                                                                     \
        if (masked) hipLaunchKernelGGL((lvt_attn_fwd_flash_kernel<BT, BH, BW, 1>), grid, blk, 0, s, A, o_amax); \
        else hipLaunchKernelGGL((lvt_attn_fwd_flash_kernel<BT, BH, BW, 0>), grid, blk, 0, s, A, o_amax);        \
    }
    LVT_FA_GEOMS(LVT_X)
#undef LVT_X
    LVT_CHECK_LAUNCH("lvt_attn_fwd_flash_kernel");
    return LVT_OK;
}

extern "C" size_t lvt_attn_bwd_flash_workspace_bytes(int B, int H, int S, int bt, int bh, int bw) {
    const size_t nb = (size_t)(2 * bt - 1) + (2 * bh - 1) + (2 * bw - 1);
    return (size_t)B * H * S * sizeof(float) + (size_t)B * H * 4 * sizeof(float) + (size_t)B * H * 2 * nb * sizeof(float);
}

extern "C" int lvt_attn_bwd_flash(const float *q, const float *k, const float *v, const float *d_o, long long ld, const float *stats,
                                  const float *o, int B, int H, int S, int da, float temper, const float *dt, const float *dh, const float *dw,
                                  int bt, int bh, int bw, int masked, float fill, float *dq, float *dk, float *dv, float *ddt,
                                  float *ddh, float *ddw, float *d_amax, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(q && k && v && d_o && stats && dt && dh && dw && dq && dk && dv && ddt && ddh && ddw && B > 0 && H > 0, "attn_bwd_flash: bad args");
    LVT_REQUIRE((B * H) % 8 == 0, "attn_bwd_flash: B * H = %d must be a multiple of 8 (workgroup pairing per XCD)", B * H);
    LVT_REQUIRE(lvt_attn_flash_supported(S, da, bt, bh, bw), "attn_bwd_flash: S=%d da=%d block (%d,%d,%d) has no instantiation", S, da, bt, bh, bw);
    LVT_REQUIRE(lvt_aligned16(q) && lvt_aligned16(k) && lvt_aligned16(v) && lvt_aligned16(d_o) && lvt_aligned16(stats) && lvt_aligned16(dq) &&
                lvt_aligned16(dk) && lvt_aligned16(dv) && ld % 4 == 0 && ld >= (long long)H * da, "attn_bwd_flash: alignment / row stride");
    if (!workspace || workspace_bytes < lvt_attn_bwd_flash_workspace_bytes(B, H, S, bt, bh, bw) || !lvt_aligned16(workspace)) {
        lvt_set_error("attn_bwd_flash: workspace too small or misaligned");
        return LVT_EWORKSPACE;
    }
    FaArgs A = {};
    A.q = q; A.k = k; A.v = v; A.d_o = d_o; A.ld = ld; A.H = H; A.inv_temper = 1.f / temper; A.c1 = FA_LOG2E / temper; A.fill2 = fill * FA_LOG2E;
    A.dt = dt; A.dh = dh; A.dw = dw; A.m = const_cast<float *>(stats); A.l = const_cast<float *>(stats) + (size_t)B * H * S;
    A.dq = dq; A.dk = dk; A.dv = dv;
    // o (the forward's output, row stride ld) selects the one-pass form of kernel A; NULL (or LVT_FA_TWOPASS): delta from a first pass over the keys
    static const int twopass = getenv("LVT_FA_TWOPASS") ? 1 : 0;
    LVT_REQUIRE(!o || lvt_aligned16(o), "attn_bwd_flash: o must be 16-byte aligned");
    const bool onep = o && !twopass;
    A.o = const_cast<float *>(o);
    A.delta = (float *)workspace;
    A.scal = A.delta + (size_t)B * H * S;
    A.bank_partial = A.scal + (size_t)B * H * 4;
    const int nt = 2 * bt - 1, nh = 2 * bh - 1, nw = 2 * bw - 1, nb = nt + nh + nw;
    const dim3 grid((unsigned)(B * H * (masked ? 1 : 2))), blk(512);
    const dim3 gridb((unsigned)(B * H * 2));                              // kernel B: the two key halves are separate items
    hipStream_t s = (hipStream_t)stream;
#define LVT_X(BT, BH, BW)                                                                                         \
    if (bt == BT && bh == BH && bw == BW) {                                                                       \
        if (masked && onep) hipLaunchKernelGGL((lvt_attn_bwd_flash_a_kernel<BT, BH, BW, 1, 1>), grid, blk, 0, s, A, d_amax); \
        else if (masked) hipLaunchKernelGGL((lvt_attn_bwd_flash_a_kernel<BT, BH, BW, 1, 0>), grid, blk, 0, s, A, d_amax);    \
        else if (onep) hipLaunchKernelGGL((lvt_attn_bwd_flash_a_kernel<BT, BH, BW, 0, 1>), grid, blk, 0, s, A, d_amax);      \
        else hipLaunchKernelGGL((lvt_attn_bwd_flash_a_kernel<BT, BH, BW, 0, 0>), grid, blk, 0, s, A, d_amax);                \
    }
    LVT_FA_GEOMS(LVT_X)
#undef LVT_X
    LVT_CHECK_LAUNCH("lvt_attn_bwd_flash_a_kernel");
#define LVT_X(BT, BH, BW)                                                                                         \
    if (bt == BT && bh == BH && bw == BW) {                                                                       \
        if (masked) hipLaunchKernelGGL((lvt_attn_bwd_flash_b_kernel<BT, BH, BW, 1>), gridb, blk, 0, s, A, d_amax); \
        else hipLaunchKernelGGL((lvt_attn_bwd_flash_b_kernel<BT, BH, BW, 0>), grid, blk, 0, s, A, d_amax);        \
    }
    LVT_FA_GEOMS(LVT_X)
#undef LVT_X
    LVT_CHECK_LAUNCH("lvt_attn_bwd_flash_b_kernel");
    hipLaunchKernelGGL(lvt_attn_flash_bank_reduce_kernel, dim3((unsigned)lvt_cdiv((long long)H * nb, 4)), dim3(256), 0, s,
                       A.bank_partial, B, H, nb, nt, nh, ddt, ddh, ddw);
    LVT_CHECK_LAUNCH("lvt_attn_flash_bank_reduce_kernel");
    return LVT_OK;
}
