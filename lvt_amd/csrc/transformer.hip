// Latent-video-transformer specific HBM-bound kernels:
//   attention softmax with the learned relative-position bias banks and causal fill (vt_attention.py:59-81,
//   :142-174), its backward with the batch-reduced bias-bank gradient, embedding bags (the one-hot
//   Conv3d / Embedding sums / one-hot Linear inputs of videotransformer.py:41-57, 80-89, 139-160 executed as
//   gathers), and the fused cross-entropy (vt.py:305-313).
#include "lvt_common.h"

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}


struct BiasGeom { int bt, bh, bw; };   // attention block extents; S == bt*bh*bw

// P[b][h][i][:] = softmax_j( s/temper + (dt[h][..] + dh[h][..]) + dw[h][..]  |  fill where j > i )
// One wave per row; lane owns 4 consecutive columns per 256-column chunk (16-byte accesses).
#define SM_MAXC 4    // row length S <= 1024 (float4 chunks of 256 columns)
__global__ void lvt_attn_softmax_fwd_kernel(float *__restrict__ scores, int B, int H, int S, float temper,
                                            const float *__restrict__ dt, const float *__restrict__ dh,
                                            const float *__restrict__ dw, BiasGeom g, int masked, float fill) {
    const int lane = threadIdx.x & 63;
    const long long nrows = (long long)B * H * S;
    const int nc = S / 256;
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < nrows;
         row += ((long long)gridDim.x * blockDim.x) >> 6) {
        const int i = row % S; const int h = (row / S) % H;
        const int wi = i % g.bw, hi = (i / g.bw) % g.bh, ti = i / (g.bw * g.bh);
        const float *bt = dt + h * (2 * g.bt - 1), *bhp = dh + h * (2 * g.bh - 1), *bwp = dw + h * (2 * g.bw - 1);
        float *p = scores + row * S;
        float v[SM_MAXC][4];
        float m = -3.4e38f;
        const float inv_temper = 1.f / temper;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            if (c < nc) {
                const int j0 = c * 256 + lane * 4;
                const float4 x4 = *reinterpret_cast<const float4 *>(p + j0);
                const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
                // key coordinates of j0, then stepped with carries (no division per element)
                int wj = j0 % g.bw, q = j0 / g.bw;
                int hj = q % g.bh, tj = q / g.bh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bias = (bt[ti - tj + g.bt - 1] + bhp[hi - hj + g.bh - 1]) + bwp[wi - wj + g.bw - 1];
                    float x = xs[e] * inv_temper + bias;
                    if (masked && j0 + e > i) x = fill;
                    v[c][e] = x;
                    m = fmaxf(m, x);
                    if (++wj == g.bw) { wj = 0; if (++hj == g.bh) { hj = 0; ++tj; } }
                }
            }
        }
        m = wmax(m);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c)
            if (c < nc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[c][e] = expf(v[c][e] - m); s += v[c][e]; }
            }
        s = 1.f / wsum(s);
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c)
            if (c < nc)
                *reinterpret_cast<float4 *>(p + c * 256 + lane * 4) =
                    make_float4(v[c][0] * s, v[c][1] * s, v[c][2] * s, v[c][3] * s);
    }
}

// For one (h, i): loop over the batch; g = P * (dP - sum_j P dP); dS = g / temper (written over dP);
// G[h][i][:] = sum_b g  (gradient w.r.t. the bias matrix B, batch-reduced in a fixed order).
__global__ void lvt_attn_softmax_bwd_kernel(const float *__restrict__ P, float *__restrict__ dP, int B, int H, int S,
                                            float temper, float *__restrict__ G) {
    const int lane = threadIdx.x & 63;
    const int nc = S / 256;
    const long long nrows = (long long)H * S;
    for (long long hr = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; hr < nrows;
         hr += ((long long)gridDim.x * blockDim.x) >> 6) {
        float4 acc[SM_MAXC];
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < B; ++b) {
            const long long off = ((long long)b * H * S + hr) * S + lane * 4;
            float4 pv[SM_MAXC], gv[SM_MAXC];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < SM_MAXC; ++c)
                if (c < nc) {
                    pv[c] = *reinterpret_cast<const float4 *>(P + off + c * 256);
                    gv[c] = *reinterpret_cast<const float4 *>(dP + off + c * 256);
                    dot += (pv[c].x * gv[c].x + pv[c].y * gv[c].y) + (pv[c].z * gv[c].z + pv[c].w * gv[c].w);
                }
            dot = wsum(dot);
#pragma unroll
            for (int c = 0; c < SM_MAXC; ++c)
                if (c < nc) {
                    float4 gg;
                    gg.x = pv[c].x * (gv[c].x - dot); gg.y = pv[c].y * (gv[c].y - dot);
                    gg.z = pv[c].z * (gv[c].z - dot); gg.w = pv[c].w * (gv[c].w - dot);
                    acc[c].x += gg.x; acc[c].y += gg.y; acc[c].z += gg.z; acc[c].w += gg.w;
                    *reinterpret_cast<float4 *>(dP + off + c * 256) =
                        make_float4(gg.x / temper, gg.y / temper, gg.z / temper, gg.w / temper);
                }
        }
        if (G) {
#pragma unroll
            for (int c = 0; c < SM_MAXC; ++c)
                if (c < nc) *reinterpret_cast<float4 *>(G + hr * S + c * 256 + lane * 4) = acc[c];
        }
    }
}

// d bank[h][entry] = sum over (i, j) whose coordinate difference selects `entry`.  One workgroup of 16 waves per
// head; a wave stages one row G[h][i][:] in LDS at a time, lane e owns bank entry e (dt entries, then dh, then dw;
// at most 64 of them) and adds the row's elements that select it -- a strided walk over the row, no divisions in
// the inner loop, every element of G visited by exactly the lanes that need it.  The 16 wave partials are added
// in wave order: bit-reproducible, no atomics.
#define BG_WAVES 16
__global__ __launch_bounds__(64 * BG_WAVES) void lvt_attn_bank_grad_kernel(const float *__restrict__ G, int H, int S,
                                                                           BiasGeom g, float *__restrict__ ddt,
                                                                           float *__restrict__ ddh, float *__restrict__ ddw) {
    __shared__ __attribute__((aligned(16))) float rowbuf[BG_WAVES][1024];
    __shared__ float red[BG_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = 2 * g.bt - 1, nh = 2 * g.bh - 1, nw = 2 * g.bw - 1;
    const int h = blockIdx.x;
    const float *Gh = G + (long long)h * S * S;
    // which bank / which difference this lane accumulates
    int which = -1, e = lane;
    if (e < nt) which = 0;
    else if ((e -= nt) < nh) which = 1;
    else if ((e -= nh) < nw) which = 2;
    float acc = 0.f;
    float *rb = rowbuf[wave];
    // the next row is in flight in registers while the current one is reduced (S <= 1024: 4 float4 per lane)
    float4 nxt[4];
    auto fetch = [&](int i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane * 4 + 256 * u < S) nxt[u] = *reinterpret_cast<const float4 *>(Gh + (long long)i * S + lane * 4 + 256 * u);
    };
    if (wave < S) fetch(wave);
    for (int i = wave; i < S; i += BG_WAVES) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane * 4 + 256 * u < S) *reinterpret_cast<float4 *>(rb + lane * 4 + 256 * u) = nxt[u];
        if (i + BG_WAVES < S) fetch(i + BG_WAVES);
        __builtin_amdgcn_wave_barrier();
        const int wi = i % g.bw, hi = (i / g.bw) % g.bh, ti = i / (g.bw * g.bh);
        // four running sums per lane keep the adds independent; they are combined in a fixed order
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        // dt entries: a whole (h, w) plane each -- summed by the wave together (lane c holds elements 4c..4c+3 of
        // every 256-chunk), the owning lane keeps the total
        for (int tj = 0; tj < g.bt; ++tj) {
            const int lo = tj * g.bh * g.bw, hi_ = lo + g.bh * g.bw;
            float part = 0.f;
            for (int c = lane * 4; c < S; c += 256)
                if (c >= lo && c < hi_) { const float4 v = *reinterpret_cast<const float4 *>(rb + c); part += (v.x + v.y) + (v.z + v.w); }
            part = wsum(part);
            if (which == 0 && e == ti + g.bt - 1 - tj) a0 += part;
        }
        if (which == 0) {
        } else if (which == 1) {
            const int hj = hi + g.bh - 1 - e;
            if (hj >= 0 && hj < g.bh)
                for (int tj = 0; tj < g.bt; ++tj) {
                    const float *q = rb + (tj * g.bh + hj) * g.bw;
#pragma unroll 4
                    for (int k = 0; k + 3 < g.bw; k += 4) { a0 += q[k]; a1 += q[k + 1]; a2 += q[k + 2]; a3 += q[k + 3]; }
                    for (int k = g.bw & ~3; k < g.bw; ++k) a0 += q[k];
                }
        } else if (which == 2) {
            const int wj = wi + g.bw - 1 - e;
            if (wj >= 0 && wj < g.bw) {
                const int n = g.bt * g.bh;
#pragma unroll 4
                for (int k = 0; k + 3 < n; k += 4) {
                    a0 += rb[k * g.bw + wj]; a1 += rb[(k + 1) * g.bw + wj]; a2 += rb[(k + 2) * g.bw + wj]; a3 += rb[(k + 3) * g.bw + wj];
                }
                for (int k = n & ~3; k < n; ++k) a0 += rb[k * g.bw + wj];
            }
        }
        acc += (a0 + a1) + (a2 + a3);
        __builtin_amdgcn_wave_barrier();
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && which >= 0) {
        float v = red[0][lane];
        for (int w = 1; w < BG_WAVES; ++w) v += red[w][lane];
        float *dst = which == 0 ? ddt + h * nt : (which == 1 ? ddh + h * nh : ddw + h * nw);
        dst[e] = v;
    }
}

extern "C" int lvt_attn_softmax_fwd(float *scores, int B, int H, int S, float temper, const float *dt,
                                    const float *dh, const float *dw, int bt, int bh, int bw, int masked, float fill,
                                    void *stream) {
    LVT_REQUIRE(scores && dt && dh && dw && B > 0 && H > 0, "attn_softmax_fwd: bad args");
    LVT_REQUIRE(S == bt * bh * bw && S % 256 == 0 && S <= 256 * SM_MAXC, "attn_softmax_fwd: S=%d unsupported", S);
    BiasGeom g = {bt, bh, bw};
    const long long rows = (long long)B * H * S;
    const int blocks = (int)(lvt_cdiv(rows, 4) < 16384 ? lvt_cdiv(rows, 4) : 16384);
    hipLaunchKernelGGL(lvt_attn_softmax_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scores, B, H, S,
                       temper, dt, dh, dw, g, masked, fill);
    LVT_CHECK_LAUNCH("lvt_attn_softmax_fwd_kernel");
    return LVT_OK;
}

// dP is overwritten with dS; ddt/ddh/ddw receive the bias-bank gradients; G is an (H,S,S) scratch buffer.
extern "C" int lvt_attn_softmax_bwd(const float *P, float *dP, int B, int H, int S, float temper, int bt, int bh,
                                    int bw, float *G, float *ddt, float *ddh, float *ddw, void *stream) {
    LVT_REQUIRE(P && dP && G && ddt && ddh && ddw && B > 0 && H > 0, "attn_softmax_bwd: bad args");
    LVT_REQUIRE(S == bt * bh * bw && S % 256 == 0 && S <= 256 * SM_MAXC, "attn_softmax_bwd: S=%d unsupported", S);
    hipStream_t s = (hipStream_t)stream;
    const long long rows = (long long)H * S;
    hipLaunchKernelGGL(lvt_attn_softmax_bwd_kernel, dim3((unsigned)lvt_cdiv(rows, 4)), dim3(256), 0, s, P, dP, B, H, S,
                       temper, G);
    LVT_CHECK_LAUNCH("lvt_attn_softmax_bwd_kernel");
    BiasGeom g = {bt, bh, bw};
    const int per_h = (2 * bt - 1) + (2 * bh - 1) + (2 * bw - 1);
    LVT_REQUIRE(per_h <= 64 && S <= 1024 && (bh * bw) % 4 == 0, "attn_softmax_bwd: %d bank entries per head (max 64)", per_h);
    hipLaunchKernelGGL(lvt_attn_bank_grad_kernel, dim3(H), dim3(64 * BG_WAVES), 0, s, G, H, S, g, ddt, ddh, ddw);
    LVT_CHECK_LAUNCH("lvt_attn_bank_grad_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// embedding bag: out[b*P+pos][:] = bias + btable[bindex[b]] + sum_slots table[tab_row[s] + idx(b,pos,s)][:]
//   idx(b,pos,s) = idx[b*bstride + off[s] + pos];  negative indices (pad_value) contribute nothing.
// ------------------------------------------------------------------------------------------------
struct BagSlots { int n; int off[32]; int tab_row[32]; };

__global__ void lvt_embbag_fwd_kernel(const long long *__restrict__ idx, long long bstride, int P, long long rows,
                                      BagSlots sl, const float *__restrict__ table, int D,
                                      const float *__restrict__ bias, const float *__restrict__ btable,
                                      const long long *__restrict__ bindex, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int d4 = D / 4;
    for (long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < rows;
         r += ((long long)gridDim.x * blockDim.x) >> 6) {
        const long long b = r / P; const int pos = r % P;
        const long long *ip = idx + b * bstride + pos;
        for (int c = lane; c < d4; c += 64) {
            float4 acc = bias ? reinterpret_cast<const float4 *>(bias)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            // eight slots at a time: their indices, then their table rows, are requested together (a decode step has
            // ONE row per sample here, so the slot loop used to be a chain of 2 x n dependent round trips); rows are
            // added in slot order
            for (int s0 = 0; s0 < sl.n; s0 += 8) {
                long long id[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) id[u] = (s0 + u < sl.n) ? ip[sl.off[s0 + u]] : -1;
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (id[u] >= 0) ? reinterpret_cast<const float4 *>(table + (sl.tab_row[s0 + u] + id[u]) * D)[c]
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (id[u] >= 0) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            if (btable) {
                const float4 v = reinterpret_cast<const float4 *>(btable + bindex[b] * D)[c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            reinterpret_cast<float4 *>(out + r * D)[c] = acc;
        }
    }
}

extern "C" int lvt_embbag_fwd(const long long *idx, long long bstride, int P, long long rows, int nslots,
                              const int *slot_off, const int *tab_row, const float *table, int D, const float *bias,
                              const float *btable, const long long *bindex, float *out, void *stream) {
    LVT_REQUIRE(idx && slot_off && tab_row && table && out && nslots > 0 && nslots <= 32 && D % 4 == 0 && rows > 0 &&
                    P > 0 && rows % P == 0, "embbag_fwd: bad args");
    LVT_REQUIRE(!btable || bindex, "embbag_fwd: btable without bindex");
    BagSlots sl; sl.n = nslots;
    for (int i = 0; i < nslots; ++i) { sl.off[i] = slot_off[i]; sl.tab_row[i] = tab_row[i]; }
    const int blocks = (int)(lvt_cdiv(rows, 4) < 16384 ? lvt_cdiv(rows, 4) : 16384);
    hipLaunchKernelGGL(lvt_embbag_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, bstride, P, rows,
                       sl, table, D, bias, btable, bindex, out);
    LVT_CHECK_LAUNCH("lvt_embbag_fwd_kernel");
    return LVT_OK;
}

// ---- embedding-bag weight gradient as a gather --------------------------------------------------------
// dTable[(slot, code)][:] = sum over the rows whose index in that slot equals `code` of dOut[row][:]
// (the weight gradient of the one-hot Conv3d / Embedding / one-hot Linear inputs, videotransformer.py:41-57, 80-89,
// 139-160).  As a dense one-hot GEMM this is 2*nslots*V*N*rows flops of which one in V is not a multiply by zero; here
// every output row is owned by one wave, which adds its rows in ascending order (a fixed order, no atomics: bit-
// reproducible) - nslots*rows*N additions in all.
//   workgroup = 8 waves x CPW codes of one slot.  The slot's indices stream through LDS in 1024-row chunks (double
//   buffered, one barrier per chunk, the next chunk's indices already in flight); each wave compares a 64-row
//   group against its codes with one ballot per code and appends the hits to a per-(wave, code) row list in LDS.
//   The lists are drained eight rows at a time - eight independent row loads in flight per wave instead of one load
//   per hit, which is what a hit-by-hit loop would be bound by - and the rows are added in list (= row) order.
#define OHG_WAVES 8
#define OHG_CHUNK 1024
#define OHG_CAP 128
struct OhSlots { int n; int off[32]; };

template <int NV, int CPW>
__global__ __launch_bounds__(OHG_WAVES * 64) void lvt_onehot_gather_kernel(
    const long long *__restrict__ idx, OhSlots sl, int V, long long bstride, long long pstride, int P, int rows,
    const float *__restrict__ dout, long long ldb, float *__restrict__ out) {
    constexpr int VW = NV >= 4 ? 4 : NV, NVEC = NV / VW, N = NV * 64, NT = OHG_WAVES * 64, PER = OHG_CHUNK / NT;
    __shared__ int codes[2][OHG_CHUNK];
    __shared__ int hits[OHG_WAVES][CPW][OHG_CAP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int code0 = (blockIdx.x * OHG_WAVES + wave) * CPW;
    const long long *ip = idx + sl.off[blockIdx.y];

    float acc[CPW][NV];
    int cnt[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        cnt[u] = 0;
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[u][e] = 0.f;
    }
    auto fetch = [&](int row) -> int {
        if (row >= rows) return -1;
        const int b = row / P, pos = row - b * P;
        const long long c = ip[b * bstride + pos * pstride];
        return (c >= 0 && c < V) ? (int)c : -1;
    };
    auto drain = [&](int u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");       // the list was written by other lanes of this wave
        const int *h = hits[wave][u];
        const int n = cnt[u];                                        // wave-uniform
        for (int j = 0; j < n; j += 8) {
            float v[8][NV];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j + t < n) {
                    const float *src = dout + (long long)h[j + t] * ldb;
#pragma unroll
                    for (int q = 0; q < NVEC; ++q) {
                        const float *s = src + (q * 64 + lane) * VW;
                        if constexpr (VW == 4) {
                            const float4 f = *reinterpret_cast<const float4 *>(s);
                            v[t][q * 4] = f.x; v[t][q * 4 + 1] = f.y; v[t][q * 4 + 2] = f.z; v[t][q * 4 + 3] = f.w;
                        } else if constexpr (VW == 2) {
                            const float2 f = *reinterpret_cast<const float2 *>(s);
                            v[t][0] = f.x; v[t][1] = f.y;
                        } else {
                            v[t][0] = *s;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (j + t < n) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) acc[u][e] += v[t][e];
                }
        }
        cnt[u] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");       // reads done before the list is appended to again
    };

    int nxt[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) nxt[e] = fetch(e * NT + tid);
    int it = 0;
    for (int r0 = 0; r0 < rows; r0 += OHG_CHUNK, ++it) {
        int *cb = codes[it & 1];
#pragma unroll
        for (int e = 0; e < PER; ++e) cb[e * NT + tid] = nxt[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) nxt[e] = fetch(r0 + OHG_CHUNK + e * NT + tid);
        for (int g = 0; g < OHG_CHUNK / 64; ++g) {
            const int c = cb[g * 64 + lane];
#pragma unroll
            for (int u = 0; u < CPW; ++u) {
                const bool hit = c == code0 + u;
                const unsigned long long m = __ballot(hit);
                if (m) {
                    if (hit) hits[wave][u][cnt[u] + __popcll(m & ((1ull << lane) - 1ull))] = r0 + g * 64 + lane;
                    cnt[u] += __popcll(m);
                    if (cnt[u] > OHG_CAP - 64) drain(u);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        drain(u);
        if (code0 + u < V) {
            float *dst = out + ((long long)blockIdx.y * V + code0 + u) * N;
#pragma unroll
            for (int q = 0; q < NVEC; ++q) {
                float *d = dst + (q * 64 + lane) * VW;
                if constexpr (VW == 4)
                    *reinterpret_cast<float4 *>(d) = make_float4(acc[u][q * 4], acc[u][q * 4 + 1], acc[u][q * 4 + 2], acc[u][q * 4 + 3]);
                else if constexpr (VW == 2)
                    *reinterpret_cast<float2 *>(d) = make_float2(acc[u][0], acc[u][1]);
                else
                    *d = acc[u][0];
            }
        }
    }
}

// the gather needs enough output rows to fill the chip (one wave per output row); the few-row tables (class and
// slice embeddings, V = 16 .. 32) stay on the split-K one-hot GEMM
bool lvt_onehot_gather_ok(int nslots, int V, int N, long long ldb, const float *dout) {
    return (N == 64 || N == 128 || N == 256 || N == 512) && (long long)nslots * V >= 512 && ldb % 4 == 0 && lvt_aligned16(dout);
}
template <int NV>
static void ohg_launch(bool two, dim3 grid, hipStream_t s, const long long *idx, const OhSlots &sl, int V, long long bstride,
                       long long pstride, int P, int rows, const float *dout, long long ldb, float *out) {
    if (two)
        hipLaunchKernelGGL((lvt_onehot_gather_kernel<NV, 2>), grid, dim3(OHG_WAVES * 64), 0, s, idx, sl, V, bstride, pstride, P,
                           rows, dout, ldb, out);
    else
        hipLaunchKernelGGL((lvt_onehot_gather_kernel<NV, 1>), grid, dim3(OHG_WAVES * 64), 0, s, idx, sl, V, bstride, pstride, P,
                           rows, dout, ldb, out);
}
int lvt_onehot_gather_launch(const long long *idx, int nslots, int V, const int *slot_off, long long bstride, long long pstride,
                             int P, long long rows, const float *dout, long long ldb, int N, float *out, hipStream_t s) {
    OhSlots sl; sl.n = nslots;
    for (int i = 0; i < nslots; ++i) sl.off[i] = slot_off[i];
    // two codes per wave once that still leaves >= 512 workgroups (half the index scans per output row; measured at 16384
    // rows x 128: 28 slots x 512 codes 84 us with one code per wave - 1792 workgroups, 1.75 rounds of the chip - 32 slots 69 us)
    const bool two = (long long)nslots * lvt_cdiv(V, OHG_WAVES * 2) >= 512;
    const dim3 grid((unsigned)lvt_cdiv(V, OHG_WAVES * (two ? 2 : 1)), (unsigned)nslots);
    switch (N) {
    case 64: ohg_launch<1>(two, grid, s, idx, sl, V, bstride, pstride, P, (int)rows, dout, ldb, out); break;
    case 128: ohg_launch<2>(two, grid, s, idx, sl, V, bstride, pstride, P, (int)rows, dout, ldb, out); break;
    case 256: ohg_launch<4>(two, grid, s, idx, sl, V, bstride, pstride, P, (int)rows, dout, ldb, out); break;
    default: ohg_launch<8>(two, grid, s, idx, sl, V, bstride, pstride, P, (int)rows, dout, ldb, out); break;
    }
    LVT_CHECK_LAUNCH("lvt_onehot_gather_kernel");
    return LVT_OK;
}

// out[d2][d1][d0] (contiguous) = in[i0*s0 + i1*s1 + i2*s2] : generic 3-D permute for weight re-layouts
__global__ void lvt_permute3_kernel(const float *__restrict__ in, long long s0, long long s1, long long s2, int n0,
                                    int n1, int n2, float *__restrict__ out) {
    const long long total = (long long)n0 * n1 * n2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int i2 = i % n2; const long long t = i / n2;
        const int i1 = t % n1; const int i0 = t / n1;
        out[i] = in[i0 * s0 + i1 * s1 + i2 * s2];
    }
}
// out is contiguous (n0, n1, n2); element (i0,i1,i2) is read from in with the given element strides.
extern "C" int lvt_permute3(const float *in, long long s0, long long s1, long long s2, int n0, int n1, int n2,
                            float *out, void *stream) {
    LVT_REQUIRE(in && out && n0 > 0 && n1 > 0 && n2 > 0, "permute3: bad args");
    const long long total = (long long)n0 * n1 * n2;
    const int blocks = (int)(lvt_cdiv(total, 256) < 8192 ? lvt_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(lvt_permute3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, s0, s1, s2, n0, n1,
                       n2, out);
    LVT_CHECK_LAUNCH("lvt_permute3_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// cross entropy with ignore_index over rows of V logits (F.cross_entropy, vt.py:305-313)
// ------------------------------------------------------------------------------------------------
#define XE_MAXE 16   // V <= 4096
__global__ void lvt_xent_fwd_kernel(const float *__restrict__ logits, const long long *__restrict__ target,
                                    long long tstride_b, long long tstride_pos, int P, long long rows, int V,
                                    long long ignore, float *__restrict__ row_loss, float *__restrict__ lse_out) {
    const int lane = threadIdx.x & 63;
    const int ne = V / 256;   // float4 per lane
    for (long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < rows;
         r += ((long long)gridDim.x * blockDim.x) >> 6) {
        const float4 *p = reinterpret_cast<const float4 *>(logits + r * V);
        float4 v[XE_MAXE / 4 > 0 ? XE_MAXE : 1];
        float m = -3.4e38f;
#pragma unroll
        for (int e = 0; e < XE_MAXE; ++e)
            if (e < ne) { v[e] = p[lane + 64 * e]; m = fmaxf(m, fmaxf(fmaxf(v[e].x, v[e].y), fmaxf(v[e].z, v[e].w))); }
        m = wmax(m);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < XE_MAXE; ++e)
            if (e < ne) s += (expf(v[e].x - m) + expf(v[e].y - m)) + (expf(v[e].z - m) + expf(v[e].w - m));
        s = wsum(s);
        const float lse = m + logf(s);
        if (lane == 0) {
            const long long t = target[(r / P) * tstride_b + (r % P) * tstride_pos];
            lse_out[r] = lse;
            row_loss[r] = (t == ignore) ? 0.f : lse - logits[r * V + t];
        }
    }
}
// stage 1 of the loss reduction: partial sums + counts of non-ignored rows per workgroup
__global__ __launch_bounds__(256) void lvt_xent_partial_kernel(const float *__restrict__ row_loss,
                                                               const long long *__restrict__ target,
                                                               long long tstride_b, long long tstride_pos, int P,
                                                               long long rows, long long ignore,
                                                               float *__restrict__ psum, float *__restrict__ pcnt) {
    __shared__ float rs[256], rc[256];
    float s = 0.f, c = 0.f;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
        const long long t = target[(r / P) * tstride_b + (r % P) * tstride_pos];
        if (t != ignore) { s += row_loss[r]; c += 1.f; }
    }
    rs[threadIdx.x] = s; rc[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rc[threadIdx.x] += rc[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { psum[blockIdx.x] = rs[0]; pcnt[blockIdx.x] = rc[0]; }
}
__global__ void lvt_xent_finish_kernel(const float *__restrict__ psum, const float *__restrict__ pcnt, int n,
                                       float scale, float *__restrict__ loss, float *__restrict__ count) {
    __shared__ double rs[256], rc[256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s += psum[i]; c += pcnt[i]; }
    rs[threadIdx.x] = s; rc[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rc[threadIdx.x] += rc[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss[0] = (float)(rs[0] / rc[0] * scale); count[0] = (float)rc[0]; }
}
// dlogits = gout * scale / count * (softmax - onehot), 0 for ignored rows
__global__ void lvt_xent_bwd_kernel(const float *__restrict__ logits, const long long *__restrict__ target,
                                    long long tstride_b, long long tstride_pos, int P, long long rows, int V,
                                    long long ignore, const float *__restrict__ lse, const float *__restrict__ count,
                                    const float *__restrict__ gout, float scale, float *__restrict__ dlogits,
                                    float *__restrict__ dl_amax) {
    const int lane = threadIdx.x & 63;
    const float c = (gout ? gout[0] : 1.f) * scale / count[0];
    // |softmax - onehot| <= 1: |c| bounds every entry (an a-priori bound like LayerNorm's: one store, no reduction)
    if (dl_amax && blockIdx.x == 0 && threadIdx.x == 0) *dl_amax = lvt_absf(c);
    for (long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < rows;
         r += ((long long)gridDim.x * blockDim.x) >> 6) {
        const long long t = target[(r / P) * tstride_b + (r % P) * tstride_pos];
        const float l = lse[r];
        for (int j = lane * 4; j < V; j += 256) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t != ignore) {
                const float4 x = *reinterpret_cast<const float4 *>(logits + r * V + j);
                o.x = c * (expf(x.x - l) - (t == j ? 1.f : 0.f));
                o.y = c * (expf(x.y - l) - (t == j + 1 ? 1.f : 0.f));
                o.z = c * (expf(x.z - l) - (t == j + 2 ? 1.f : 0.f));
                o.w = c * (expf(x.w - l) - (t == j + 3 ? 1.f : 0.f));
            }
            *reinterpret_cast<float4 *>(dlogits + r * V + j) = o;
        }
    }
}

#define XE_BLOCKS 512
extern "C" size_t lvt_xent_workspace_bytes(void) { return (size_t)2 * XE_BLOCKS * sizeof(float); }
// loss[0] = scale * mean over non-ignored rows of (lse - logit[target]); count[0] = #non-ignored rows.
// target(b,pos) = target[b*tstride_b + pos*tstride_pos] with rows = B*P.
extern "C" int lvt_xent_fwd(const float *logits, const long long *target, long long tstride_b, long long tstride_pos,
                            int P, long long rows, int V, long long ignore, float scale, float *row_loss, float *lse,
                            float *loss, float *count, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(logits && target && row_loss && lse && loss && count && rows > 0 && P > 0 && rows % P == 0,
                "xent_fwd: bad args");
    LVT_REQUIRE(V % 256 == 0 && V <= 256 * XE_MAXE, "xent_fwd: V=%d unsupported", V);
    LVT_REQUIRE(workspace && workspace_bytes >= lvt_xent_workspace_bytes(), "xent_fwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)(lvt_cdiv(rows, 4) < 16384 ? lvt_cdiv(rows, 4) : 16384);
    hipLaunchKernelGGL(lvt_xent_fwd_kernel, dim3(blocks), dim3(256), 0, s, logits, target, tstride_b, tstride_pos, P,
                       rows, V, ignore, row_loss, lse);
    LVT_CHECK_LAUNCH("lvt_xent_fwd_kernel");
    const int pb = (int)(lvt_cdiv(rows, 256) < XE_BLOCKS ? lvt_cdiv(rows, 256) : XE_BLOCKS);
    float *psum = (float *)workspace, *pcnt = psum + XE_BLOCKS;
    hipLaunchKernelGGL(lvt_xent_partial_kernel, dim3(pb), dim3(256), 0, s, row_loss, target, tstride_b, tstride_pos, P,
                       rows, ignore, psum, pcnt);
    hipLaunchKernelGGL(lvt_xent_finish_kernel, dim3(1), dim3(256), 0, s, psum, pcnt, pb, scale, loss, count);
    LVT_CHECK_LAUNCH("lvt_xent_finish_kernel");
    return LVT_OK;
}
extern "C" int lvt_xent_bwd(const float *logits, const long long *target, long long tstride_b, long long tstride_pos,
                            int P, long long rows, int V, long long ignore, const float *lse, const float *count,
                            const float *gout, float scale, float *dlogits, float *dl_amax, void *stream) {
    LVT_REQUIRE(logits && target && lse && count && dlogits && rows > 0 && V % 4 == 0, "xent_bwd: bad args");
    const int blocks = (int)(lvt_cdiv(rows, 4) < 16384 ? lvt_cdiv(rows, 4) : 16384);
    hipLaunchKernelGGL(lvt_xent_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, target, tstride_b,
                       tstride_pos, P, rows, V, ignore, lse, count, gout, scale, dlogits, dl_amax);
    LVT_CHECK_LAUNCH("lvt_xent_bwd_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// single-query attention against a K/V cache (incremental sampling):
//   o[b][h][:] = softmax_j<=i( q.K_j / temper + bias(i, j) ) V_j     for the one query position i.
// One workgroup of 4 waves per (b, h).  q (B, H*DA), caches (B, S, H*DA) token-major, o (B, H*DA).  DA == 128.
// Same arithmetic as row i of the full causal layer (the masked columns j > i carry exp(-1e4 - m) == 0).
// Scores: 16 lanes per key (coalesced rows).  P.V: the keys are dealt to the 4 waves (j = w mod 4), every
// lane owns 2 of the 128 output dims, and the 4 partial sums are added in wave order through LDS.
// ------------------------------------------------------------------------------------------------
#define DEC_DA 128
#define DEC_WAVES 4
__global__ __launch_bounds__(64 * DEC_WAVES) void lvt_attn_decode_kernel(const float *__restrict__ q, long long ldq,
                                                             const float *__restrict__ Kc,
                                                             const float *__restrict__ Vc, int H, int S, int qi,
                                                             float temper, const float *__restrict__ dt,
                                                             const float *__restrict__ dh, const float *__restrict__ dw,
                                                             BiasGeom g, float *__restrict__ o,
                                                             const int *__restrict__ pos, long long q_pos) {
    __shared__ float qs[DEC_DA];
    __shared__ float ps[1024];
    __shared__ float redm[DEC_WAVES], reds[DEC_WAVES];
    __shared__ float acc[DEC_WAVES][DEC_DA];
    const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = H * DEC_DA;
    if (pos) {                                   // device-side cursor (hipGraph replay): clamped so that a stale value cannot
        qi = pos[0];                             // address outside the caches
        qi = qi < 0 ? 0 : (qi >= S ? S - 1 : qi);
        q += (long long)qi * q_pos;
    }
    const float *qp = q + (long long)b * ldq + h * DEC_DA;
    if (tid < DEC_DA) qs[tid] = qp[tid];
    __syncthreads();
    const int nk = qi + 1;
    const int wi = qi % g.bw, hi = (qi / g.bw) % g.bh, ti = qi / (g.bw * g.bh);
    const float *bt = dt + h * (2 * g.bt - 1), *bhp = dh + h * (2 * g.bh - 1), *bwp = dw + h * (2 * g.bw - 1);
    float m = -3.4e38f;
    // scores: 16 lanes share one key (a coalesced 512-byte row: 8 dims = two float4 per lane), 16 keys per pass of
    // the workgroup; the 16 partial dot products are combined with a butterfly inside the lane group
    {
        const int sub = tid & 15, grp = tid >> 4;                       // 16 groups of 16 lanes
        float qv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = qs[sub * 8 + d];
        for (int j0 = 0; j0 < nk; j0 += 16) {
            const int j = j0 + grp;
            float sc = 0.f;
            if (j < nk) {
                const float4 *kp = reinterpret_cast<const float4 *>(Kc + ((long long)b * S + j) * hd + h * DEC_DA + sub * 8);
                const float4 k0 = kp[0], k1 = kp[1];
                sc = fmaf(qv[0], k0.x, sc); sc = fmaf(qv[1], k0.y, sc); sc = fmaf(qv[2], k0.z, sc); sc = fmaf(qv[3], k0.w, sc);
                sc = fmaf(qv[4], k1.x, sc); sc = fmaf(qv[5], k1.y, sc); sc = fmaf(qv[6], k1.z, sc); sc = fmaf(qv[7], k1.w, sc);
            }
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) sc += __shfl_xor(sc, d, 64);
            if (sub == 0 && j < nk) {
                const int wj = j % g.bw, hj = (j / g.bw) % g.bh, tj = j / (g.bw * g.bh);
                const float x = sc / temper + ((bt[ti - tj + g.bt - 1] + bhp[hi - hj + g.bh - 1]) + bwp[wi - wj + g.bw - 1]);
                ps[j] = x;
                m = fmaxf(m, x);
            }
        }
    }
    m = wmax(m);
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sum = 0.f;
    for (int j = tid; j < nk; j += 64 * DEC_WAVES) { const float e = expf(ps[j] - m); ps[j] = e; sum += e; }
    sum = wsum(sum);
    if (lane == 0) reds[wave] = sum;
    __syncthreads();
    sum = ((reds[0] + reds[1]) + reds[2]) + reds[3];
    // value rows: a lane owns dims 2*lane, 2*lane+1 (one 8-byte load per row: a wave instruction is one whole 512-byte
    // row), eight rows in flight per wave in four independent accumulator pairs, combined in a fixed order
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f;
    const float *vp = Vc + (long long)b * S * hd + h * DEC_DA + 2 * lane;
    const long long rs = (long long)DEC_WAVES * hd;                     // stride between the rows of one wave
    int j = wave;
    for (; j + 7 * DEC_WAVES < nk; j += 8 * DEC_WAVES) {
        const float *v0 = vp + (long long)j * hd;
        float2 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float2 *>(v0 + u * rs);
        float pj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pj[u] = ps[j + u * DEC_WAVES];
        a0 = fmaf(pj[0], x[0].x, a0); a1 = fmaf(pj[0], x[0].y, a1);
        b0 = fmaf(pj[1], x[1].x, b0); b1 = fmaf(pj[1], x[1].y, b1);
        c0 = fmaf(pj[2], x[2].x, c0); c1 = fmaf(pj[2], x[2].y, c1);
        d0 = fmaf(pj[3], x[3].x, d0); d1 = fmaf(pj[3], x[3].y, d1);
        a0 = fmaf(pj[4], x[4].x, a0); a1 = fmaf(pj[4], x[4].y, a1);
        b0 = fmaf(pj[5], x[5].x, b0); b1 = fmaf(pj[5], x[5].y, b1);
        c0 = fmaf(pj[6], x[6].x, c0); c1 = fmaf(pj[6], x[6].y, c1);
        d0 = fmaf(pj[7], x[7].x, d0); d1 = fmaf(pj[7], x[7].y, d1);
    }
    for (; j < nk; j += DEC_WAVES) {
        const float p = ps[j];
        const float2 x = *reinterpret_cast<const float2 *>(vp + (long long)j * hd);
        a0 = fmaf(p, x.x, a0);
        a1 = fmaf(p, x.y, a1);
    }
    a0 = (a0 + b0) + (c0 + d0); a1 = (a1 + b1) + (c1 + d1);
    acc[wave][2 * lane] = a0; acc[wave][2 * lane + 1] = a1;
    __syncthreads();
    if (tid < DEC_DA)
        o[(long long)b * hd + h * DEC_DA + tid] = (((acc[0][tid] + acc[1][tid]) + acc[2][tid]) + acc[3][tid]) / sum;
}

// ------------------------------------------------------------------------------------------------
// categorical draw from logits with a caller-supplied uniform (videotransformer.py:176-181: softmax(logit / temp)
// followed by torch.multinomial; here the inverse-CDF rule of oracle.multinomial_from_uniform so that a draw is a
// pure function of (logits, u)): code = #{ j : cdf_j <= u * total }, clamped to V-1.  One wave per row, V <= 1024.
// The code is written as int64 at out[row * out_stride]; probabilities (row, V) are optional.
// ------------------------------------------------------------------------------------------------
#define SMP_MAXPER 16
__global__ __launch_bounds__(64) void lvt_sample_categorical_kernel(const float *__restrict__ logits, int V, float inv_temp,
                                                                    const float *__restrict__ u, long long *__restrict__ out,
                                                                    long long out_stride, float *__restrict__ probs,
                                                                    const int *__restrict__ pos, long long u_pos) {
    const int row = blockIdx.x, lane = threadIdx.x;
    if (pos) u += (long long)pos[0] * u_pos;
    const float *x = logits + (long long)row * V;
    const int per = (V + 63) / 64;                      // consecutive elements per lane: cdf order == memory order
    const int j0 = lane * per;
    float e[SMP_MAXPER];
    float m = -3.4e38f;
#pragma unroll
    for (int i = 0; i < SMP_MAXPER; ++i) { e[i] = (i < per && j0 + i < V) ? x[j0 + i] * inv_temp : -3.4e38f; m = fmaxf(m, e[i]); }
    m = wmax(m);
    float loc = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_MAXPER; ++i) { e[i] = (i < per && j0 + i < V) ? expf(e[i] - m) : 0.f; loc += e[i]; }
    // exclusive scan of the lane sums
    float incl = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const float total = __shfl(incl, 63, 64);
    const float thr = u[row] * total;
    float c = incl - loc;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < SMP_MAXPER; ++i) { c += e[i]; if (i < per && j0 + i < V && c <= thr) ++cnt; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0) out[row * out_stride] = cnt < V - 1 ? cnt : V - 1;
    if (probs)
#pragma unroll
        for (int i = 0; i < SMP_MAXPER; ++i) if (i < per && j0 + i < V) probs[(long long)row * V + j0 + i] = e[i] / total;
}

extern "C" int lvt_sample_categorical(const float *logits, long long rows, int V, float temp, const float *u,
                                      long long *out, long long out_stride, float *probs, const int *pos, long long u_pos,
                                      void *stream) {
    LVT_REQUIRE(logits && u && out && rows > 0 && V > 0 && V <= 64 * SMP_MAXPER && temp > 0.f, "sample_categorical: bad args");
    hipLaunchKernelGGL(lvt_sample_categorical_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, logits, V,
                       1.f / temp, u, out, out_stride, probs, pos, u_pos);
    LVT_CHECK_LAUNCH("lvt_sample_categorical_kernel");
    return LVT_OK;
}

extern "C" int lvt_attn_decode(const float *q, long long ldq, const float *Kc, const float *Vc, int B, int H, int S, int da, int qi,
                               float temper, const float *dt, const float *dh, const float *dw, int bt, int bh, int bw,
                               float *o, const int *pos, long long q_pos, void *stream) {
    LVT_REQUIRE(q && Kc && Vc && dt && dh && dw && o && B > 0 && H > 0, "attn_decode: bad args");
    LVT_REQUIRE(da == DEC_DA && S == bt * bh * bw && S <= 1024 && qi >= 0 && qi < S, "attn_decode: unsupported shape");
    BiasGeom g = {bt, bh, bw};
    LVT_REQUIRE(ldq >= (long long)H * da, "attn_decode: ldq");
    hipLaunchKernelGGL(lvt_attn_decode_kernel, dim3(B * H), dim3(64 * DEC_WAVES), 0, (hipStream_t)stream, q, ldq, Kc, Vc, H, S, qi,
                       temper, dt, dh, dw, g, o, pos, q_pos);
    LVT_CHECK_LAUNCH("lvt_attn_decode_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// Integer plumbing of a decode step whose position lives in DEVICE memory, so that one captured hipGraph serves every
// position of every slice (autoregressive/incremental.py):
//   lvt_decode_gather_codes: out[b][c][j] = codes[b][c][ nb[pos][j] ]   -- the codes of the causal-conv neighbours of
//       the current position (nb rows of `taps` int64 entries; entry == S points at the always-padded extra slot)
//   lvt_decode_commit:       codes[b][c][pos] = drawn[b][c] (optional), then pos += 1.  ONE workgroup, so that the
//       increment is ordered after every read of the cursor.
// codes is (B, nc, S1) int64 with S1 = S + 1 (the padded slot).
// ------------------------------------------------------------------------------------------------
__global__ void lvt_decode_gather_codes_kernel(const long long *__restrict__ codes, const long long *__restrict__ nb,
                                               const int *__restrict__ pos, int rows, int S1, int taps,
                                               long long *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * taps) return;
    int cur = pos[0];
    cur = cur < 0 ? 0 : (cur >= S1 - 1 ? S1 - 2 : cur);
    const int r = i / taps, j = i - r * taps;
    long long n = nb[(long long)cur * taps + j];
    n = n < 0 ? S1 - 1 : (n >= S1 ? S1 - 1 : n);
    out[i] = codes[(long long)r * S1 + n];
}
__global__ __launch_bounds__(256) void lvt_decode_commit_kernel(const long long *__restrict__ drawn, int rows, int S1,
                                                                long long *__restrict__ codes, int *__restrict__ pos) {
    const int cur = pos[0];
    if (drawn && cur >= 0 && cur < S1 - 1)
        for (int r = threadIdx.x; r < rows; r += blockDim.x) codes[(long long)r * S1 + cur] = drawn[r];
    __syncthreads();
    if (threadIdx.x == 0) pos[0] = cur + 1;
}
extern "C" int lvt_decode_gather_codes(const long long *codes, const long long *nb, const int *pos, int rows, int S1,
                                       int taps, long long *out, void *stream) {
    LVT_REQUIRE(codes && nb && pos && out && rows > 0 && S1 > 1 && taps > 0, "decode_gather_codes: bad args");
    hipLaunchKernelGGL(lvt_decode_gather_codes_kernel, dim3((unsigned)lvt_cdiv((long long)rows * taps, 256)), dim3(256), 0,
                       (hipStream_t)stream, codes, nb, pos, rows, S1, taps, out);
    LVT_CHECK_LAUNCH("lvt_decode_gather_codes_kernel");
    return LVT_OK;
}
extern "C" int lvt_decode_commit(const long long *drawn, int rows, int S1, long long *codes, int *pos, void *stream) {
    LVT_REQUIRE(codes && pos && rows > 0 && S1 > 1, "decode_commit: bad args");
    hipLaunchKernelGGL(lvt_decode_commit_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, drawn, rows, S1, codes, pos);
    LVT_CHECK_LAUNCH("lvt_decode_commit_kernel");
    return LVT_OK;
}
