// Fused attention forward and backward for one 256-token block with a compile-time block geometry, on operands that arrive
// PRE-SPLIT as bf16x3 planes (reference: ScaledDotProductAttention / BlockLocalAttention, vt_attention.py:52-81,142-174).
//
// Why planes.  The phase-by-phase kernel of attention.hip keeps the matrix pipe busy 23 % of the time: of its ~3900 vector
// instructions per wave (768 MFMAs), 2600 are the 3-way bf16 split of K, V, Q and P, and K / V are split again by every
// workgroup and every kernel that stages them (profiles/r03_attention_instruction_mix.txt).  Here q, k, v and dO are split
// ONCE, by the epilogue of the GEMM that produces them (LVT_EPI_PLANES), and travel as three bf16 planes:
//   * staging a chunk is a copy (16-byte global loads -> ds_write_b128) into a two-slot LDS ring: while the MFMAs of step t
//     read slot t & 1 the chunk of step t+1 is copied into the other slot (one barrier per step) and the global loads of
//     step t+2 are in flight (two register sets);
//   * the B operands that live in registers (the wave's queries / dO rows) are loaded in fragment layout, no arithmetic;
//   * operands that are needed TRANSPOSED (V^T in O^T = V^T P^T, K^T in dQ^T = K^T dS^T, dO^T and Q^T in the dK / dV
//     products) are staged by rows and fetched with the gfx950 transposing LDS read (ds_read_b64_tr_b16);
//   * the softmax is ONLINE per 64-key chunk (running maximum, per-chunk correction factor, one multiply per element at the
//     end): the scores of chunk c are finished while the MFMAs of chunk c+1 run; the dS algebra of the backward pass needs
//     only delta_i = sum_d dO[i][d] O[i][d], known before the first product, and is chunk-wise in the same way.
//
// Tiling: 16-wide tiles on v_mfma_f32_16x16x32_bf16.  A first version used 32 x 32 x 16 tiles, one wave = 32 queries x 256
// keys = 128 score accumulators, ~370 registers, ONE wave per SIMD -- and was no faster than the unfused path although it
// removed two thirds of the vector instructions: with one wave per SIMD nothing covers a memory or LDS round trip (SQ: 64 %
// of the wave cycles in issue stalls, matrix pipe 29 % busy).  With 16 x 16 tiles a wave owns 16 queries (64 score + 32
// output accumulators, 190-250 registers), a 128-query workgroup is EIGHT waves = two per SIMD, and the same staging serves
// the same number of queries: forward 176 -> 147 us, backward ~410 -> 184 + 151 us per layer at the DSFVT shape.
// Operand layouts of the 16x16x32 MFMA: A / B: lane (row or column = lane & 15, k block kg = lane >> 4) holds 8 consecutive
// k; C: lane (column = lane & 15, row block = lane >> 4) holds rows 4 kg + 0..3.  A P^T / dS^T B operand built from two
// score tiles has the key order {tile0: 4 kg + 0..3, tile1: 4 kg + 0..3}; the transposing read fetches the same rows.
//
// Backward = two kernels over the saved P:
//   A (one wave = 16 queries): dP^T = V dO^T -> dS = P o (dP - delta) / temper -> dS to HBM, dQ^T = K^T dS^T, and the
//     bias-bank gradient of this (sample, head, query half) from the registers (two fixed-order stages through LDS);
//   B (one wave = 16 keys):    dV^T = dO^T P, dK^T = Q^T dS over the queries, P / dS fragments read in B-operand layout,
// and a fixed-order reduction of the per-workgroup bank sums (no atomics).
// Causal layers skip the chunks that are entirely above the diagonal; inside the remaining chunks the masked elements are
// computed and filled (exp(fill - m) == 0 exactly, P == 0 makes dS == 0), so the pipelined blocks contain no branches.
#include "attn_common.h"

namespace {

template <int I> struct IC { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}
// the two register sets of the staging ring, selected at compile time (an array indexed by the step would live in scratch)
#define AP_G(t) ((((t) & 1) == 0) ? g0 : g1)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 is a struct of unions and is not promoted to registers)

// A fragment TRANSPOSED (the reduction index is the LDS row): lane column ctile*32 + l31, k slots = rows r0 .. r0+3 and
// r1 .. r1+3.  Lane li of a 16-lane group supplies the address of row (li >> 2), columns 4 (li & 3) .. +3 of its group's
// [4 rows][16 columns] block and receives column li, rows 0..3 (tools/ubench/tr_probe.hip).
__device__ __forceinline__ bf16x8 ap_tr(const unsigned short *p, int hi_off) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + hi_off));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}
// accumulator order (B operand straight from a 32x32 accumulator tile): slots e = rows kt*32 + 16 s2 + 4 half + (e & 3) + 8 (e >> 2)


// Workgroup x runs on XCD x % 8 and every XCD has its own L2: the two halves of one (sample, head) stage the same K / V (or
// Q / dO) chunks, so they are given block ids 8 apart -- same XCD, dispatched back to back -- and the second reads from L2.
// (grid = 2 * pairs; the entry points require pairs % 8 == 0.)
// CAUSAL blocks: one half meets 4 key (query) chunks, the other 2.  As separate workgroups the short ones bought nothing (masked
// and unmasked launches took the same 145 / 180 / 150 us), so a causal launch has ONE workgroup per pair that runs the long half
// and then the short one: 3-6 % per kernel.  (A fully persistent forward -- one workgroup per CU walking its items, the ring
// running on into the next item's first chunks -- measured another 3 % at best: these kernels move 500-700 MB per launch at
// 3.6-4.3 TB/s and are bound by that, not by workgroup turnover; profiles/r03_attention_workgroup_timeline.txt.)
__device__ __forceinline__ int ap_pair(unsigned x) { return (int)(((x >> 4) << 3) | (x & 7)); }
__device__ __forceinline__ int ap_half(unsigned x) { return (int)((x >> 3) & 1); }
// the same value, but opaque to the optimiser: the second half of a causal pair recomputes its addresses instead of keeping the
// first half's alive across a whole kernel body (which spilled)
__device__ __forceinline__ int ap_opaque(int x) { asm volatile("" : "+s"(x)); return x; }

struct PlaneArgs {
    const unsigned short *q, *k, *v;     // plane 0 of each operand, token-major (M, hd)
    long long ps;                        // plane stride (elements)
};

template <int BT, int BH, int BW> struct BankIdx {
    static constexpr int NT = 2 * BT - 1, NH = 2 * BH - 1, NW = 2 * BW - 1, NB = NT + NH + NW, NR = BT + BH + BW / 2;
};


typedef float f32x4v __attribute__((ext_vector_type(4)));
#define A16_LD 144                       // row pitch (bf16): 288 B -- conflict-free for the 16-row ds_read_b128 AND the tr reads
#define A16_PL (AT_KC * A16_LD)
#define A16_SLOT (3 * A16_PL)            // 55296 B
#define A16_NG 6                         // 16-byte units per thread and chunk (512 threads)
struct G6 { u32x4 v[A16_NG]; };
__device__ __forceinline__ void a16_load(G6 &g, const unsigned short *base, long long ps, long long ld, int tid) {
    static_for<A16_NG>([&](auto ic) {
        constexpr int idx = decltype(ic)::value, pl = idx >> 1, j = idx & 1;
        const int u = tid + 512 * j;
        g.v[idx] = *reinterpret_cast<const u32x4 *>(base + pl * ps + (long long)(u >> 4) * ld + (u & 15) * 8);
    });
}
template <int U0, int N>
__device__ __forceinline__ void a16_park(const G6 &g, unsigned short *slot, int tid) {
    static_for<N>([&](auto ic) {
        constexpr int idx = U0 + decltype(ic)::value, pl = idx >> 1;
        const int u = tid + 512 * (idx & 1);
        *reinterpret_cast<u32x4 *>(slot + pl * A16_PL + (u >> 4) * A16_LD + (u & 15) * 8) = g.v[idx];
    });
}

// key j = 16 T + 4 kg + r (tile T, lane block kg, register r): for BW == 16 its coordinates are (t, h, w) = (T / HP, T % HP, 4 kg + r)
// with HP = BH tiles per t slab; for BW == 8 a tile spans two h rows: (T / HP, 2 (T % HP) + (kg >> 1), 4 (kg & 1) + r), HP = BH / 2.
// The t index and the h SLOT T % HP are compile-time per tile; what depends on the lane is folded into per-lane tables.
template <int BH, int BW> struct Geo16 {
    static_assert(BW == 16 || BW == 8, "16-wide tiles: BW is 8 or 16");
    static constexpr int HP = BW == 16 ? BH : BH / 2;
    static __device__ __forceinline__ int hj(int hslot, int kg) { return BW == 16 ? hslot : 2 * hslot + (kg >> 1); }
    static __device__ __forceinline__ int wj(int r, int kg) { return BW == 16 ? 4 * kg + r : 4 * (kg & 1) + r; }
};

template <int BT, int BH, int BW, int MASKED, int NCH>
__device__ __forceinline__ float attn_fwd16_body(const PlaneArgs pa, int H, float inv_temper, const float *__restrict__ dt,
                                                const float *__restrict__ dh, const float *__restrict__ dw, float fill,
                                                float *__restrict__ P, float *__restrict__ o, unsigned short *X, int bh_,
                                                int qhalf) {
    using GE = Geo16<BH, BW>;
    constexpr int HP = GE::HP;
    static_assert(BT * BH * BW == AT_S, "256 tokens");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / H, h = bh_ % H;
    const int hd = H * AT_D;
    const long long row0 = (long long)b * AT_S;
    const int i = qhalf * 128 + wave * 16 + c16;                      // this lane's query (accumulator column)
    const unsigned short *kbase = pa.k + row0 * hd + h * AT_D, *vbase = pa.v + row0 * hd + h * AT_D;
    constexpr int NIT = 2 * NCH;

    G6 g0, g1;
    auto load_item = [&](int t, G6 &gg) {
        a16_load(gg, (t < NCH ? kbase : vbase) + (long long)((t < NCH ? t : t - NCH) * AT_KC) * hd, pa.ps, hd, tid);
    };
    load_item(0, g0);
    load_item(1, g1);
    // the wave's queries as B operands: lane (query c16, k block kg) holds d = 32 s + 8 kg .. + 7
    bf16x8 qb[AT_D / 32][3];
    {
        const unsigned short *qrow = pa.q + (row0 + i) * hd + h * AT_D;
#pragma unroll
        for (int s = 0; s < AT_D / 32; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) qb[s][pl] = *reinterpret_cast<const bf16x8 *>(qrow + pl * pa.ps + 32 * s + 8 * kg);
    }
    // bias of (query i, key j = 16 T + 4 kg + r): per-lane tables over the t index, the h slot and the register (Geo16)
    const int wi = i % BW, hi = (i / BW) % BH, ti = i / (BW * BH);
    float bt_[BT], bh__[HP], bw_[4];
#pragma unroll
    for (int x = 0; x < BT; ++x) bt_[x] = dt[h * (2 * BT - 1) + ti - x + BT - 1];
#pragma unroll
    for (int x = 0; x < HP; ++x) bh__[x] = dh[h * (2 * BH - 1) + hi - GE::hj(x, kg) + BH - 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) bw_[r] = dw[h * (2 * BW - 1) + wi - GE::wj(r, kg) + BW - 1];
    a16_park<0, A16_NG>(g0, X, tid);
    __syncthreads();

    f32x4v st[AT_S / 16];
#pragma unroll
    for (int T = 0; T < AT_S / 16; ++T) st[T] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -3.4e38f, mc[NCH], sumc[NCH];
    float cmax = -3.4e38f, csum = 0.f;
    // online softmax of chunk c (tiles 4c .. 4c+3: 16 scores per lane) in 4 pieces, spread over the 4 k-steps of the next chunk
    auto softmax_piece = [&](auto cc, auto pc) {
        constexpr int c = decltype(cc)::value, piece = decltype(pc)::value;
        if constexpr (piece < 2) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                constexpr int dummy = 0; (void)dummy;
                const int T = 4 * c + 2 * piece + tt;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = st[T][r] * inv_temper + ((bt_[T / HP] + bh__[T % HP]) + bw_[r]);
                    if (MASKED && 16 * T + 4 * kg + r > i) x = fill;
                    st[T][r] = x;
                    cmax = fmaxf(cmax, x);
                }
            }
        }
        if constexpr (piece == 2) {
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            m_run = fmaxf(m_run, cmax);
            mc[c] = m_run;
            cmax = -3.4e38f;
            csum = 0.f;
        }
        if constexpr (piece >= 2) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int T = 4 * c + 2 * (piece - 2) + tt;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float ex = __expf(st[T][r] - m_run); st[T][r] = ex; csum += ex; }
            }
        }
        if constexpr (piece == 3) sumc[c] = csum;
    };

    // ---------------- phase 1: S^T = K Q^T ----------------
    static_for<NCH>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const unsigned short *cur = X + (t & 1) * A16_SLOT;
        unsigned short *nxt = X + ((t + 1) & 1) * A16_SLOT;
        if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));
        static_for<AT_D / 32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            bf16x8 a[4][3];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[kt][pl] = *reinterpret_cast<const bf16x8 *>(cur + pl * A16_PL + (kt * 16 + c16) * A16_LD + 32 * s + 8 * kg);
            if constexpr (s < 3) a16_park<2 * s, 2>(AP_G(t + 1), nxt, tid);
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    st[4 * t + kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt][AT_TA(tm)], qb[s][AT_TB(tm)], st[4 * t + kt], 0, 0, 0);
            if constexpr (t > 0) softmax_piece(IC<t - 1>{}, sc);
        });
        __syncthreads();
    });
    static_for<4>([&](auto pc) { softmax_piece(IC<NCH - 1>{}, pc); });
    {
        float total = 0.f, corr[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { corr[c] = __expf(mc[c] - m_run); total += sumc[c] * corr[c]; }
        total += __shfl_xor(total, 16, 64);
        total += __shfl_xor(total, 32, 64);
        const float inv = 1.f / total;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float f = corr[c] * inv;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[4 * c + kt] *= f;
        }
    }

    // ---------------- phase 2: O^T = V^T P^T ----------------
    f32x4v oacc[AT_D / 16];
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile) oacc[dtile] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float *prow = P + (((long long)b * H + h) * AT_S + i) * AT_S + 4 * kg;
    static_for<NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value, t = NCH + c;
        const unsigned short *cur = X + (t & 1) * A16_SLOT;
        unsigned short *nxt = X + ((t + 1) & 1) * A16_SLOT;
        if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));
        static_for<2>([&](auto pc) {                                  // key pair: tiles T0, T0 + 1 (32 keys) of the chunk
            constexpr int pr = decltype(pc)::value, T0 = 4 * c + 2 * pr;
            bf16x8 pb[3];
            at_split8(make_float4(st[T0][0], st[T0][1], st[T0][2], st[T0][3]),
                      make_float4(st[T0 + 1][0], st[T0 + 1][1], st[T0 + 1][2], st[T0 + 1][3]), pb[0], pb[1], pb[2]);
            *reinterpret_cast<f32x4v *>(prow + 16 * T0) = st[T0];
            *reinterpret_cast<f32x4v *>(prow + 16 * (T0 + 1)) = st[T0 + 1];
            static_for<2>([&](auto hc) {                              // d tiles 4 hq .. 4 hq + 3
                constexpr int hq = decltype(hc)::value;
                bf16x8 a[4][3];
#pragma unroll
                for (int dq = 0; dq < 4; ++dq) {
                    const unsigned short *p0 = cur + (32 * pr + 4 * kg + (c16 >> 2)) * A16_LD + (4 * hq + dq) * 16 + 4 * (c16 & 3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[dq][pl] = ap_tr(p0 + pl * A16_PL, 16 * A16_LD);
                }
                if constexpr (t + 1 < NIT && hq == 0) a16_park<3 * pr, 3>(AP_G(t + 1), nxt, tid);
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int dq = 0; dq < 4; ++dq)
                        oacc[4 * hq + dq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dq][AT_TA(tm)], pb[AT_TB(tm)], oacc[4 * hq + dq], 0, 0, 0);
            });
        });
        __syncthreads();
    });
    if (NCH < AT_S / AT_KC) {
#pragma unroll
        for (int T = 4 * NCH; T < AT_S / 16; ++T) *reinterpret_cast<f32x4v *>(prow + 16 * T) = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    {
        float *orow = o + (row0 + i) * hd + h * AT_D + 4 * kg;
#pragma unroll
        for (int dtile = 0; dtile < AT_D / 16; ++dtile) *reinterpret_cast<f32x4v *>(orow + 16 * dtile) = oacc[dtile];
    }
    float am = 0.f;          // max |o| of this lane's stores (reported through o_amax for the f16x2 engine launches downstream)
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile)
        am = fmaxf(am, fmaxf(fmaxf(lvt_absf(oacc[dtile][0]), lvt_absf(oacc[dtile][1])), fmaxf(lvt_absf(oacc[dtile][2]), lvt_absf(oacc[dtile][3]))));
    return am;
}

template <int BT, int BH, int BW, int MASKED>
__global__ __launch_bounds__(512, 1) void lvt_attn_fwd16_planes_kernel(const PlaneArgs pa, int H, float inv_temper,
                                                                       const float *__restrict__ dt, const float *__restrict__ dh,
                                                                       const float *__restrict__ dw, float fill,
                                                                       float *__restrict__ P, float *__restrict__ o,
                                                                       float *__restrict__ o_amax) {
    __shared__ __attribute__((aligned(16))) unsigned short X[2 * A16_SLOT];
    float am;
    if (MASKED) {        // one workgroup = both query halves of a (sample, head): 4 + 2 key chunks (see the note at ap_pair)
        am = attn_fwd16_body<BT, BH, BW, MASKED, 4>(pa, H, inv_temper, dt, dh, dw, fill, P, o, X, blockIdx.x, 1);
        __syncthreads();
        am = fmaxf(am, attn_fwd16_body<BT, BH, BW, MASKED, 2>(pa, H, inv_temper, dt, dh, dw, fill, P, o, X, ap_opaque(blockIdx.x), 0));
    } else {
        am = attn_fwd16_body<BT, BH, BW, MASKED, 4>(pa, H, inv_temper, dt, dh, dw, fill, P, o, X, ap_pair(blockIdx.x), ap_half(blockIdx.x));
    }
    if (o_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, o_amax, reinterpret_cast<float *>(X));
    }
}

// =====================================================================================================================
// backward on 16-wide tiles (eight waves per workgroup, two per SIMD): A16 = dS, dQ, bank sums; B16 = dV, dK.  BW == 16.
// =====================================================================================================================
template <int BT, int BH, int BW, int MASKED, int NCH>
__device__ __forceinline__ float attn_bwd_a16_body(const PlaneArgs pa, const unsigned short *__restrict__ dop, int H,
                                                  float inv_temper, const float *__restrict__ P, const float *__restrict__ o,
                                                  float *__restrict__ dS, float *__restrict__ dq, float *__restrict__ bank_partial,
                                                  unsigned short *X, int bh_, int qhalf) {
    static_assert(BT * BH * BW == AT_S, "256 tokens");
    using BI = BankIdx<BT, BH, BW>;
    using GE = Geo16<BH, BW>;
    constexpr int HP = GE::HP;
    constexpr int NR = BT + HP + 4;                                   // per-lane class sums: t index, h slot, register (Geo16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / H, h = bh_ % H;
    const int hd = H * AT_D;
    const long long row0 = (long long)b * AT_S;
    const int il = wave * 16 + c16, i = qhalf * 128 + il;
    const unsigned short *kbase = pa.k + row0 * hd + h * AT_D, *vbase = pa.v + row0 * hd + h * AT_D;
    constexpr int NIT = 2 * NCH;                                      // V chunks (by rows), then K chunks (transposed)

    G6 g0, g1;
    auto load_item = [&](int t, G6 &gg) {
        a16_load(gg, (t < NCH ? vbase : kbase) + (long long)((t < NCH ? t : t - NCH) * AT_KC) * hd, pa.ps, hd, tid);
    };
    load_item(0, g0);
    load_item(1, g1);
    bf16x8 dob[AT_D / 32][3];
    {
        const unsigned short *drow = dop + (row0 + i) * hd + h * AT_D;
#pragma unroll
        for (int s = 0; s < AT_D / 32; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dob[s][pl] = *reinterpret_cast<const bf16x8 *>(drow + pl * pa.ps + 32 * s + 8 * kg);
    }
    a16_park<0, A16_NG>(g0, X, tid);
    __syncthreads();

    f32x4v st[AT_S / 16];
#pragma unroll
    for (int T = 0; T < AT_S / 16; ++T) st[T] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const float *prow = P + (((long long)b * H + h) * AT_S + i) * AT_S + 4 * kg;
    float *dsrow = dS + (((long long)b * H + h) * AT_S + i) * AT_S + 4 * kg;
    f32x4v pv[4];                          // P of a chunk: slot k is consumed by piece k during step c+1 and refilled right after
    // delta_i = sum_j P_ij dP_ij, from THE dP values that dS is formed with: the row sums of dS then cancel the way the
    // reference's softmax backward makes them cancel (P o (dP - sum_j P dP)).  (Round 3 took delta_i = sum_d dO_id O_id,
    // the same number in exact arithmetic; its rounding error is a common offset of a whole row of dP - delta, i.e. an error
    // of dS along P_i that adds up coherently in dQ / dK: 6x the CPU fp32 oracle's distance from fp64 on the w_q gradient
    // of a near-uniform attention layer.)  dP of the whole row stays in registers; P is streamed twice (the second pass
    // hits L2) and O is not read at all.
    float delta = 0.f;
    auto delta_piece = [&](auto cc, auto pc, const f32x4v p4) {
        constexpr int T = 4 * decltype(cc)::value + decltype(pc)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) delta = fmaf(p4[e], st[T][e], delta);
    };
    auto ds_piece = [&](auto cc, auto pc, const f32x4v p4) {
        constexpr int T = 4 * decltype(cc)::value + decltype(pc)::value;
        f32x4v ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[e] = p4[e] * (st[T][e] - delta) * inv_temper;
        st[T] = ds;
        *reinterpret_cast<f32x4v *>(dsrow + 16 * T) = ds;
    };

    // ---------------- phase 1: dP^T = V dO^T ----------------
    static_for<NCH>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const unsigned short *cur = X + (t & 1) * A16_SLOT;
        unsigned short *nxt = X + ((t + 1) & 1) * A16_SLOT;
        if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));
        static_for<AT_D / 32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            bf16x8 a[4][3];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[kt][pl] = *reinterpret_cast<const bf16x8 *>(cur + pl * A16_PL + (kt * 16 + c16) * A16_LD + 32 * s + 8 * kg);
            if constexpr (s < 3) a16_park<2 * s, 2>(AP_G(t + 1), nxt, tid);
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    st[4 * t + kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt][AT_TA(tm)], dob[s][AT_TB(tm)], st[4 * t + kt], 0, 0, 0);
            if constexpr (t > 0) delta_piece(IC<t - 1>{}, sc, pv[s]);
            pv[s] = *reinterpret_cast<const f32x4v *>(prow + 16 * (4 * t + s));
        });
        __syncthreads();
    });
    static_for<4>([&](auto pc) { delta_piece(IC<NCH - 1>{}, pc, pv[decltype(pc)::value]); });
    delta += __shfl_xor(delta, 16, 64);          // the four lanes (kg) that share query i hold 64 keys each
    delta += __shfl_xor(delta, 32, 64);
    // second pass over P: dS = P o (dP - delta) / temper.  The last chunk's P is still in pv and goes first; the other chunks
    // are re-read (L2 hits), the four pieces of a chunk in flight together.
    static_for<4>([&](auto pc) { ds_piece(IC<NCH - 1>{}, pc, pv[decltype(pc)::value]); });
    static_for<NCH - 1>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) pv[s] = *reinterpret_cast<const f32x4v *>(prow + 16 * (4 * c + s));
        static_for<4>([&](auto pc) { ds_piece(cc, pc, pv[decltype(pc)::value]); });
    });
    if (NCH < AT_S / AT_KC) {
#pragma unroll
        for (int T = 4 * NCH; T < AT_S / 16; ++T) *reinterpret_cast<f32x4v *>(dsrow + 16 * T) = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    // per-lane class sums of dS (g = dS * temper) over the t index T / HP, the h slot T % HP and the register r
    float rsum[NR];
#pragma unroll
    for (int x = 0; x < NR; ++x) rsum[x] = 0.f;
    static_for<4 * NCH>([&](auto Tc) {
        constexpr int T = decltype(Tc)::value;
        const float tot = (st[T][0] + st[T][1]) + (st[T][2] + st[T][3]);
        rsum[T / HP] += tot;
        rsum[BT + T % HP] += tot;
#pragma unroll
        for (int r = 0; r < 4; ++r) rsum[BT + HP + r] += st[T][r];
    });

    // ---------------- phase 2: dQ^T = K^T dS^T ----------------
    f32x4v oacc[AT_D / 16];
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile) oacc[dtile] = f32x4v{0.f, 0.f, 0.f, 0.f};
    static_for<NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value, t = NCH + c;
        const unsigned short *cur = X + (t & 1) * A16_SLOT;
        unsigned short *nxt = X + ((t + 1) & 1) * A16_SLOT;
        if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));
        static_for<2>([&](auto pc) {
            constexpr int pr = decltype(pc)::value, T0 = 4 * c + 2 * pr;
            bf16x8 pb[3];
            at_split8(make_float4(st[T0][0], st[T0][1], st[T0][2], st[T0][3]),
                      make_float4(st[T0 + 1][0], st[T0 + 1][1], st[T0 + 1][2], st[T0 + 1][3]), pb[0], pb[1], pb[2]);
            static_for<2>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                bf16x8 a[4][3];
#pragma unroll
                for (int dq_ = 0; dq_ < 4; ++dq_) {
                    const unsigned short *p0 = cur + (32 * pr + 4 * kg + (c16 >> 2)) * A16_LD + (4 * hq + dq_) * 16 + 4 * (c16 & 3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[dq_][pl] = ap_tr(p0 + pl * A16_PL, 16 * A16_LD);
                }
                if constexpr (t + 1 < NIT && hq == 0) a16_park<3 * pr, 3>(AP_G(t + 1), nxt, tid);
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int dq_ = 0; dq_ < 4; ++dq_)
                        oacc[4 * hq + dq_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dq_][AT_TA(tm)], pb[AT_TB(tm)], oacc[4 * hq + dq_], 0, 0, 0);
            });
        });
        __syncthreads();
    });
    {
        float *qrow = dq + (row0 + i) * hd + h * AT_D + 4 * kg;
#pragma unroll
        for (int dtile = 0; dtile < AT_D / 16; ++dtile) *reinterpret_cast<f32x4v *>(qrow + 16 * dtile) = oacc[dtile];
    }
    float am = 0.f;          // max |dq| of this lane's stores
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile)
        am = fmaxf(am, fmaxf(fmaxf(lvt_absf(oacc[dtile][0]), lvt_absf(oacc[dtile][1])), fmaxf(lvt_absf(oacc[dtile][2]), lvt_absf(oacc[dtile][3]))));
    // ---- bias-bank gradient of this (sample, head, query half): two fixed-order stages through LDS ----
    float *R = reinterpret_cast<float *>(X);                          // [128 queries][4 kg][NR]
    float *R2 = R + 128 * 4 * NR;                                     // [8 parts][NB]
    {
        const float temper = 1.f / inv_temper;
        float *mine = R + (il * 4 + kg) * NR;
#pragma unroll
        for (int x = 0; x < NR; ++x) mine[x] = rsum[x] * temper;
    }
    __syncthreads();
    {
        const int e = tid & 63, part = tid >> 6;                      // entry, 16-query part
        if (e < BI::NB) {
            float acc = 0.f;
            for (int q = 16 * part; q < 16 * part + 16; ++q) {
                const int iq = qhalf * 128 + q;
                const int wi = iq % BW, hi = (iq / BW) % BH, ti = iq / (BW * BH);
                const float *r = R + (q * 4) * NR;
                if (e < BI::NT) {
                    const int tj = ti - e + BT - 1;
                    if (tj >= 0 && tj < BT) acc += (r[tj] + r[NR + tj]) + (r[2 * NR + tj] + r[3 * NR + tj]);
                } else if (e < BI::NT + BI::NH) {
                    const int hj = hi - (e - BI::NT) + BH - 1;
                    if (hj >= 0 && hj < BH) {
                        if (BW == 16) acc += (r[BT + hj] + r[NR + BT + hj]) + (r[2 * NR + BT + hj] + r[3 * NR + BT + hj]);
                        else acc += r[(2 * (hj & 1)) * NR + BT + (hj >> 1)] + r[(2 * (hj & 1) + 1) * NR + BT + (hj >> 1)];   // lanes with kg >> 1 == hj & 1
                    }
                } else {
                    const int wj = wi - (e - BI::NT - BI::NH) + BW - 1;
                    if (wj >= 0 && wj < BW) {
                        if (BW == 16) acc += r[(wj >> 2) * NR + BT + HP + (wj & 3)];
                        else acc += r[(wj >> 2) * NR + BT + HP + (wj & 3)] + r[((wj >> 2) + 2) * NR + BT + HP + (wj & 3)];          // lanes with kg & 1 == wj >> 2
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
        bank_partial[((long long)bh_ * 2 + qhalf) * BI::NB + tid] = acc;
    }
    return am;
}

template <int BT, int BH, int BW, int MASKED>
__global__ __launch_bounds__(512, 1) void lvt_attn_bwd_a16_kernel(const PlaneArgs pa, const unsigned short *__restrict__ dop, int H,
                                                                  float inv_temper, const float *__restrict__ P,
                                                                  const float *__restrict__ o, float *__restrict__ dS,
                                                                  float *__restrict__ dq, float *__restrict__ bank_partial,
                                                                  float *__restrict__ d_amax) {
    __shared__ __attribute__((aligned(16))) unsigned short X[2 * A16_SLOT];
    float am;
    if (MASKED) {
        am = attn_bwd_a16_body<BT, BH, BW, MASKED, 4>(pa, dop, H, inv_temper, P, o, dS, dq, bank_partial, X, blockIdx.x, 1);
        __syncthreads();
        am = fmaxf(am, attn_bwd_a16_body<BT, BH, BW, MASKED, 2>(pa, dop, H, inv_temper, P, o, dS, dq, bank_partial, X,
                                                                ap_opaque(blockIdx.x), 0));
    } else {
        am = attn_bwd_a16_body<BT, BH, BW, MASKED, 4>(pa, dop, H, inv_temper, P, o, dS, dq, bank_partial, X, ap_pair(blockIdx.x),
                                                      ap_half(blockIdx.x));
    }
    if (d_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, d_amax, reinterpret_cast<float *>(X));
    }
}

template <int C0, int NCH>
__device__ __forceinline__ float attn_bwd_b16_body(const PlaneArgs pa, const unsigned short *__restrict__ dop, int H,
                                                  const float *__restrict__ P, const float *__restrict__ dS,
                                                  float *__restrict__ dk, float *__restrict__ dv, unsigned short *X, int bh_,
                                                  int khalf) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    const int b = bh_ / H, h = bh_ % H;
    const int hd = H * AT_D;
    const long long row0 = (long long)b * AT_S;
    const int kb = khalf * 128 + wave * 16;                           // first key of this wave
    const unsigned short *qbase = pa.q + row0 * hd + h * AT_D, *dobase = dop + row0 * hd + h * AT_D;
    constexpr int NIT = 2 * NCH;                                      // dO chunk, Q chunk, dO chunk, ...

    G6 g0, g1;
    auto load_item = [&](int t, G6 &gg) {
        a16_load(gg, ((t & 1) ? qbase : dobase) + (long long)((C0 + (t >> 1)) * AT_KC) * hd, pa.ps, hd, tid);
    };
    // B fragments from P (even items) / dS (odd items): lane (key c16, kg), k slots j < 4: query 32 s + 4 kg + j,
    // j >= 4: query 32 s + 16 + 4 kg + (j - 4)  (the row blocks the transposing reads of dO / Q fetch)
    const float *pcol = P + (((long long)b * H + h) * AT_S) * AT_S + kb + c16;
    const float *dscol = dS + (((long long)b * H + h) * AT_S) * AT_S + kb + c16;
    float bf0[16], bf1[16];
#define AP_BF(t) ((((t) & 1) == 0) ? bf0 : bf1)
    auto load_b = [&](int t, float (&ff)[16]) {
        const float *src = ((t & 1) ? dscol : pcol) + (long long)((C0 + (t >> 1)) * AT_KC + 4 * kg) * AT_S;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) ff[8 * s + j] = src[(long long)(32 * s + 16 * (j >> 2) + (j & 3)) * AT_S];
    };
    load_item(0, g0);
    load_item(1, g1);
    load_b(0, bf0);
    a16_park<0, A16_NG>(g0, X, tid);
    __syncthreads();

    f32x4v accv[AT_D / 16], acck[AT_D / 16];
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile) { accv[dtile] = f32x4v{0.f, 0.f, 0.f, 0.f}; acck[dtile] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    static_for<NIT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const unsigned short *cur = X + (t & 1) * A16_SLOT;
        unsigned short *nxt = X + ((t + 1) & 1) * A16_SLOT;
        if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));
        if constexpr (t + 1 < NIT) load_b(t + 1, AP_BF(t + 1));
        static_for<2>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            bf16x8 pb[3];
            const float (&ff)[16] = AP_BF(t);
            at_split8(make_float4(ff[8 * s + 0], ff[8 * s + 1], ff[8 * s + 2], ff[8 * s + 3]),
                      make_float4(ff[8 * s + 4], ff[8 * s + 5], ff[8 * s + 6], ff[8 * s + 7]), pb[0], pb[1], pb[2]);
            static_for<2>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                bf16x8 a[4][3];
#pragma unroll
                for (int dq_ = 0; dq_ < 4; ++dq_) {
                    const unsigned short *p0 = cur + (32 * s + 4 * kg + (c16 >> 2)) * A16_LD + (4 * hq + dq_) * 16 + 4 * (c16 & 3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[dq_][pl] = ap_tr(p0 + pl * A16_PL, 16 * A16_LD);
                }
                if constexpr (t + 1 < NIT && hq == 0) a16_park<3 * s, 3>(AP_G(t + 1), nxt, tid);
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int dq_ = 0; dq_ < 4; ++dq_) {
                        if constexpr ((t & 1) == 0)
                            accv[4 * hq + dq_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dq_][AT_TA(tm)], pb[AT_TB(tm)], accv[4 * hq + dq_], 0, 0, 0);
                        else
                            acck[4 * hq + dq_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dq_][AT_TA(tm)], pb[AT_TB(tm)], acck[4 * hq + dq_], 0, 0, 0);
                    }
            });
        });
        __syncthreads();
    });
#undef AP_BF
    {
        float *vrow = dv + (row0 + kb + c16) * hd + h * AT_D + 4 * kg, *krow = dk + (row0 + kb + c16) * hd + h * AT_D + 4 * kg;
#pragma unroll
        for (int dtile = 0; dtile < AT_D / 16; ++dtile) {
            *reinterpret_cast<f32x4v *>(vrow + 16 * dtile) = accv[dtile];
            *reinterpret_cast<f32x4v *>(krow + 16 * dtile) = acck[dtile];
        }
    }
    float am = 0.f;          // max |dk|, |dv| of this lane's stores
#pragma unroll
    for (int dtile = 0; dtile < AT_D / 16; ++dtile)
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fmaxf(lvt_absf(accv[dtile][e]), lvt_absf(acck[dtile][e])));
    return am;
}

template <int MASKED>
__global__ __launch_bounds__(512, 1) void lvt_attn_bwd_b16_kernel(const PlaneArgs pa, const unsigned short *__restrict__ dop, int H,
                                                                  const float *__restrict__ P, const float *__restrict__ dS,
                                                                  float *__restrict__ dk, float *__restrict__ dv,
                                                                  float *__restrict__ d_amax) {
    __shared__ __attribute__((aligned(16))) unsigned short X[2 * A16_SLOT];
    float am;
    if (MASKED) {        // key half 0 meets all four query chunks, key half 1 the last two
        am = attn_bwd_b16_body<0, 4>(pa, dop, H, P, dS, dk, dv, X, blockIdx.x, 0);
        __syncthreads();
        am = fmaxf(am, attn_bwd_b16_body<2, 2>(pa, dop, H, P, dS, dk, dv, X, ap_opaque(blockIdx.x), 1));
    } else {
        am = attn_bwd_b16_body<0, 4>(pa, dop, H, P, dS, dk, dv, X, ap_pair(blockIdx.x), ap_half(blockIdx.x));
    }
    if (d_amax) {
        __syncthreads();
        lvt_block_amax_commit(am, d_amax, reinterpret_cast<float *>(X));
    }
}

// bank gradients: out[h][e] = sum over (sample, query half) of the workgroup partials.  One wave per output: lane l adds
// partials l, l + 64, ... in that order and the 64 lane sums meet in a fixed butterfly (deterministic, no atomics).
__global__ __launch_bounds__(256) void lvt_attn_bank_reduce_kernel(const float *__restrict__ partial, int B, int H, int NB, int nt,
                                                                   int nh, float *__restrict__ ddt, float *__restrict__ ddh,
                                                                   float *__restrict__ ddw) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= H * NB) return;
    const int h = idx / NB, e = idx % NB;
    float s = 0.f;
    for (int j = lane; j < 2 * B; j += 64) s += partial[(((long long)(j >> 1) * H + h) * 2 + (j & 1)) * NB + e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane) return;
    if (e < nt) ddt[h * nt + e] = s;
    else if (e < nt + nh) ddh[h * nh + e - nt] = s;
    else ddw[h * (NB - nt - nh) + e - nt - nh] = s;
}

}  // namespace

extern "C" int lvt_attn_planes_supported(int S, int da, int bt, int bh, int bw) {
    return S == AT_S && da == AT_D && ((bt == 1 && bh == 16 && bw == 16) || (bt == 4 && bh == 8 && bw == 8)) ? 1 : 0;
}

#define LVT_AP_GEOMS(X) X(1, 16, 16) X(4, 8, 8)

extern "C" int lvt_attn_fwd_planes(const void *qkv_planes, long long plane_stride, long long operand_stride, int B, int H, int S,
                                   int da, float temper, const float *dt, const float *dh, const float *dw, int bt, int bh,
                                   int bw, int masked, float fill, float *P, float *o, float *o_amax, void *stream) {
    LVT_REQUIRE(qkv_planes && dt && dh && dw && P && o && B > 0 && H > 0, "attn_fwd_planes: bad args");
    LVT_REQUIRE((B * H) % 8 == 0, "attn_fwd_planes: B * H = %d must be a multiple of 8 (workgroup pairing per XCD)", B * H);
    LVT_REQUIRE(lvt_attn_planes_supported(S, da, bt, bh, bw), "attn_fwd_planes: S=%d da=%d block (%d,%d,%d) has no instantiation", S, da, bt, bh, bw);
    LVT_REQUIRE(lvt_aligned16(qkv_planes) && plane_stride % 8 == 0 && operand_stride % 8 == 0 && lvt_aligned16(P) && lvt_aligned16(o),
                "attn_fwd_planes: alignment");
    const unsigned short *base = (const unsigned short *)qkv_planes;
    PlaneArgs pa = {base, base + operand_stride, base + 2 * operand_stride, plane_stride};
    const dim3 grid((unsigned)(B * H * (masked ? 1 : 2))), blk(512);
    hipStream_t s = (hipStream_t)stream;
    const float it = 1.f / temper;
#define LVT_X(BT, BH, BW)                                                                                                            \
    if (bt == BT && bh == BH && bw == BW) {                                                                                          \
        if (masked) hipLaunchKernelGGL((lvt_attn_fwd16_planes_kernel<BT, BH, BW, 1>), grid, blk, 0, s, pa, H, it, dt, dh, dw, fill, P, o, o_amax); \
        else hipLaunchKernelGGL((lvt_attn_fwd16_planes_kernel<BT, BH, BW, 0>), grid, blk, 0, s, pa, H, it, dt, dh, dw, fill, P, o, o_amax); \
    }
    LVT_AP_GEOMS(LVT_X)
#undef LVT_X
    LVT_CHECK_LAUNCH("lvt_attn_fwd16_planes_kernel");
    return LVT_OK;
}

extern "C" size_t lvt_attn_bwd_planes_workspace_bytes(int B, int H, int S, int bt, int bh, int bw) {
    const size_t nb = (size_t)(2 * bt - 1) + (2 * bh - 1) + (2 * bw - 1);
    return (size_t)B * H * S * S * sizeof(float) + (size_t)B * H * 2 * nb * sizeof(float);
}

extern "C" int lvt_attn_bwd_planes(const void *qkv_planes, long long plane_stride, long long operand_stride, const void *do_planes,
                                   const float *P, const float *o, int B, int H, int S, int da, float temper, int bt, int bh,
                                   int bw, int masked, float *dq, float *dk, float *dv, float *ddt, float *ddh, float *ddw,
                                   float *d_amax, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(qkv_planes && do_planes && P && o && dq && dk && dv && ddt && ddh && ddw && B > 0 && H > 0, "attn_bwd_planes: bad args");
    LVT_REQUIRE((B * H) % 8 == 0, "attn_bwd_planes: B * H = %d must be a multiple of 8 (workgroup pairing per XCD)", B * H);
    LVT_REQUIRE(lvt_attn_planes_supported(S, da, bt, bh, bw), "attn_bwd_planes: S=%d da=%d block (%d,%d,%d) has no instantiation", S, da, bt, bh, bw);
    LVT_REQUIRE(lvt_aligned16(qkv_planes) && lvt_aligned16(do_planes) && plane_stride % 8 == 0 && operand_stride % 8 == 0 &&
                lvt_aligned16(P) && lvt_aligned16(o) && lvt_aligned16(dq) && lvt_aligned16(dk) && lvt_aligned16(dv), "attn_bwd_planes: alignment");
    if (!workspace || workspace_bytes < lvt_attn_bwd_planes_workspace_bytes(B, H, S, bt, bh, bw) || !lvt_aligned16(workspace)) {
        lvt_set_error("attn_bwd_planes: workspace too small or misaligned");
        return LVT_EWORKSPACE;
    }
    const unsigned short *base = (const unsigned short *)qkv_planes, *dop = (const unsigned short *)do_planes;
    PlaneArgs pa = {base, base + operand_stride, base + 2 * operand_stride, plane_stride};
    float *dS = (float *)workspace, *partial = dS + (size_t)B * H * S * S;
    const int nt = 2 * bt - 1, nh = 2 * bh - 1, nw = 2 * bw - 1, nb = nt + nh + nw;
    const dim3 grid((unsigned)(B * H * (masked ? 1 : 2))), blk(512);
    hipStream_t s = (hipStream_t)stream;
    const float it = 1.f / temper;
#define LVT_X(BT, BH, BW)                                                                                                            \
    if (bt == BT && bh == BH && bw == BW) {                                                                                          \
        if (masked) hipLaunchKernelGGL((lvt_attn_bwd_a16_kernel<BT, BH, BW, 1>), grid, blk, 0, s, pa, dop, H, it, P, o, dS, dq, partial, d_amax);  \
        else hipLaunchKernelGGL((lvt_attn_bwd_a16_kernel<BT, BH, BW, 0>), grid, blk, 0, s, pa, dop, H, it, P, o, dS, dq, partial, d_amax);  \
    }
    LVT_AP_GEOMS(LVT_X)
#undef LVT_X
    LVT_CHECK_LAUNCH("lvt_attn_bwd_a16_kernel");
    if (masked) hipLaunchKernelGGL((lvt_attn_bwd_b16_kernel<1>), grid, blk, 0, s, pa, dop, H, P, dS, dk, dv, d_amax);
    else hipLaunchKernelGGL((lvt_attn_bwd_b16_kernel<0>), grid, blk, 0, s, pa, dop, H, P, dS, dk, dv, d_amax);
    LVT_CHECK_LAUNCH("lvt_attn_bwd_b16_kernel");
    hipLaunchKernelGGL(lvt_attn_bank_reduce_kernel, dim3((unsigned)lvt_cdiv((long long)H * nb, 4)), dim3(256), 0, s, partial, B, H, nb,
                       nt, nh, ddt, ddh, ddw);
    LVT_CHECK_LAUNCH("lvt_attn_bank_reduce_kernel");
    return LVT_OK;
}
