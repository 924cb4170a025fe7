// HBM-bound helper kernels of the LVT hot path: layout conversion at the model boundary, losses,
// LayerNorm, attention softmax with the learned relative-position bias, embedding bags and
// cross-entropy.  All of them are 16-byte vectorised where the layout allows, one wave (64 lanes)
// per row for the row reductions, and every cross-workgroup reduction is two-stage with a fixed
// summation order (bit-reproducible run to run) unless stated otherwise.
#include "lvt_common.h"

static inline int grid_for(long long n, int per_block, int cap = 8192) {
    long long b = lvt_cdiv(n, per_block);
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// ------------------------------------------------------------------------------------------------
// layout: (B, R, C) -> (B, C, R) with channel padding and a per-channel affine (ae.py:36-37,151-168)
//   mode 0: y = x                      mode 1: y = (x - a[c]) / s[c]       (normalizer)
//   mode 2: y = clamp(x * s[c] + a[c]) (back_normalizer + clamp_)
// in is [B][R][ldi] (only the first C columns are read), out is [B][C_out][R] or, with to_last=1,
// in is [B][C][R] and out is [B][R][ldo] (columns C..ldo-1 zero-filled).
// ------------------------------------------------------------------------------------------------
__global__ void lvt_to_channels_last_kernel(const float *__restrict__ in, int B, int C, long long R, int ldo,
                                            int mode, const float *__restrict__ a, const float *__restrict__ s,
                                            float *__restrict__ out, float *__restrict__ out_amax) {
    __shared__ float amax_scratch[4];
    float am = 0.f;
    const long long total = (long long)B * R * ldo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = i % ldo; const long long t = i / ldo;
        const long long r = t % R; const long long b = t / R;
        float v = 0.f;
        if (c < C) {
            v = in[(b * C + c) * R + r];
            if (mode == 1) v = (v - a[c]) / s[c];
        }
        out[i] = v;
        am = fmaxf(am, lvt_absf(v));
    }
    if (out_amax) lvt_block_amax_commit(am, out_amax, amax_scratch);
}
__global__ void lvt_to_channels_first_kernel(const float *__restrict__ in, int B, int C, long long R, int ldi,
                                             int mode, const float *__restrict__ a, const float *__restrict__ s,
                                             float lo, float hi, float *__restrict__ out) {
    const long long total = (long long)B * C * R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i % R; const long long t = i / R;
        const int c = t % C; const long long b = t / C;
        float v = in[(b * R + r) * ldi + c];
        if (mode == 2) { v = v * s[c] + a[c]; v = fminf(fmaxf(v, lo), hi); }
        out[i] = v;
    }
}

extern "C" int lvt_to_channels_last(const float *in, int B, int C, long long R, int ldo, int mode, const float *a,
                                    const float *s, float *out, float *out_amax, void *stream) {
    LVT_REQUIRE(in && out && B > 0 && C > 0 && R > 0 && ldo >= C, "to_channels_last: bad args");
    LVT_REQUIRE(mode == 0 || (a && s), "to_channels_last: affine tables missing");
    hipLaunchKernelGGL(lvt_to_channels_last_kernel, dim3(grid_for((long long)B * R * ldo, 256, out_amax ? 2048 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, in, B, C, R, ldo, mode, a, s, out, out_amax);
    LVT_CHECK_LAUNCH("lvt_to_channels_last_kernel");
    return LVT_OK;
}
extern "C" int lvt_to_channels_first(const float *in, int B, int C, long long R, int ldi, int mode, const float *a,
                                     const float *s, float lo, float hi, float *out, void *stream) {
    LVT_REQUIRE(in && out && B > 0 && C > 0 && R > 0 && ldi >= C, "to_channels_first: bad args");
    LVT_REQUIRE(mode == 0 || (a && s), "to_channels_first: affine tables missing");
    hipLaunchKernelGGL(lvt_to_channels_first_kernel, dim3(grid_for((long long)B * C * R, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, B, C, R, ldi, mode, a, s, lo, hi, out);
    LVT_CHECK_LAUNCH("lvt_to_channels_first_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// squared-error loss (F.mse_loss; vqvae.py:79,86 / loss.py:19) -- fixed-order two-stage sum
// ------------------------------------------------------------------------------------------------
#define RED_BLOCKS 1024
template <int L1>          // L1 = 1: sum |a - b| (F.l1_loss, loss.py:11-12)
__global__ void lvt_sqdiff_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n4,
                                          float *__restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i];
        const float4 y = reinterpret_cast<const float4 *>(b)[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        if (L1) s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        else s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void lvt_scalar_finish_kernel(const float *__restrict__ partial, int n, float scale,
                                         const float *__restrict__ denom_dev, float *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double v = red[0] * (double)scale;
        if (denom_dev) v /= (double)fmaxf(denom_dev[0], 1.0f);
        out[0] = (float)v;
    }
}
extern "C" size_t lvt_reduce_workspace_bytes(void) { return (size_t)RED_BLOCKS * sizeof(float) * 2; }

// out[0] = scale * sum((a-b)^2) / denom
extern "C" int lvt_mse_fwd(const float *a, const float *b, long long n, double denom, float scale, float *out,
                           void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(a && b && out && n > 0 && n % 4 == 0 && denom > 0, "mse_fwd: bad args");
    LVT_REQUIRE(workspace && workspace_bytes >= lvt_reduce_workspace_bytes(), "mse_fwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = grid_for(n / 4, 256 * 4, RED_BLOCKS);
    hipLaunchKernelGGL(lvt_sqdiff_partial_kernel<0>, dim3(blocks), dim3(256), 0, s, a, b, n / 4, (float *)workspace);
    LVT_CHECK_LAUNCH("lvt_sqdiff_partial_kernel");
    hipLaunchKernelGGL(lvt_scalar_finish_kernel, dim3(1), dim3(256), 0, s, (const float *)workspace, blocks,
                       (float)((double)scale / denom), (const float *)nullptr, out);
    LVT_CHECK_LAUNCH("lvt_scalar_finish_kernel");
    return LVT_OK;
}
// out[0] = scale * sum(|a-b|) / denom      (F.l1_loss, vidgen/modeling/loss/loss.py:11-12)
extern "C" int lvt_l1_fwd(const float *a, const float *b, long long n, double denom, float scale, float *out,
                          void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(a && b && out && n > 0 && n % 4 == 0 && denom > 0, "l1_fwd: bad args");
    LVT_REQUIRE(workspace && workspace_bytes >= lvt_reduce_workspace_bytes(), "l1_fwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = grid_for(n / 4, 256 * 4, RED_BLOCKS);
    hipLaunchKernelGGL(lvt_sqdiff_partial_kernel<1>, dim3(blocks), dim3(256), 0, s, a, b, n / 4, (float *)workspace);
    LVT_CHECK_LAUNCH("lvt_sqdiff_partial_kernel<l1>");
    hipLaunchKernelGGL(lvt_scalar_finish_kernel, dim3(1), dim3(256), 0, s, (const float *)workspace, blocks,
                       (float)((double)scale / denom), (const float *)nullptr, out);
    LVT_CHECK_LAUNCH("lvt_scalar_finish_kernel");
    return LVT_OK;
}

// out = add + g * (2*scale/denom) * (a - b) [* (1 - a^2)]      g = gout_dev[0] (or 1)
template <int L1>          // L1 = 1: sign(a - b) instead of (a - b) (torch's l1_loss backward: 0 at a == b)
__global__ void lvt_mse_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n4, float c,
                                   const float *__restrict__ gout, const float *__restrict__ add, int tanh_of_a,
                                   float *__restrict__ out, float *__restrict__ out_amax) {
    __shared__ float amax_scratch[4];
    float am = 0.f;
    const float g = (gout ? gout[0] : 1.0f) * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i];
        const float4 y = reinterpret_cast<const float4 *>(b)[i];
        float4 r = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        if (L1) r = make_float4((float)((r.x > 0.f) - (r.x < 0.f)), (float)((r.y > 0.f) - (r.y < 0.f)), (float)((r.z > 0.f) - (r.z < 0.f)),
                                (float)((r.w > 0.f) - (r.w < 0.f)));
        r.x *= g; r.y *= g; r.z *= g; r.w *= g;
        if (tanh_of_a) {
            r.x *= 1.f - x.x * x.x; r.y *= 1.f - x.y * x.y; r.z *= 1.f - x.z * x.z; r.w *= 1.f - x.w * x.w;
        }
        if (add) {
            const float4 z = reinterpret_cast<const float4 *>(add)[i];
            r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w;
        }
        reinterpret_cast<float4 *>(out)[i] = r;
        am = fmaxf(am, fmaxf(fmaxf(lvt_absf(r.x), lvt_absf(r.y)), fmaxf(lvt_absf(r.z), lvt_absf(r.w))));
    }
    if (out_amax) lvt_block_amax_commit(am, out_amax, amax_scratch);
}
extern "C" int lvt_mse_bwd(const float *a, const float *b, long long n, double denom, float scale,
                           const float *gout_dev, const float *add, int tanh_of_a, float *out, float *out_amax, void *stream) {
    LVT_REQUIRE(a && b && out && n > 0 && n % 4 == 0 && denom > 0, "mse_bwd: bad args");
    hipLaunchKernelGGL(lvt_mse_bwd_kernel<0>, dim3(grid_for(n / 4, 256, out_amax ? 2048 : 8192)), dim3(256), 0, (hipStream_t)stream, a, b,
                       n / 4, (float)(2.0 * (double)scale / denom), gout_dev, add, tanh_of_a, out, out_amax);
    LVT_CHECK_LAUNCH("lvt_mse_bwd_kernel");
    return LVT_OK;
}
// out = add + g * (scale/denom) * sign(a - b) [* (1 - a^2)]
extern "C" int lvt_l1_bwd(const float *a, const float *b, long long n, double denom, float scale,
                          const float *gout_dev, const float *add, int tanh_of_a, float *out, float *out_amax, void *stream) {
    LVT_REQUIRE(a && b && out && n > 0 && n % 4 == 0 && denom > 0, "l1_bwd: bad args");
    hipLaunchKernelGGL(lvt_mse_bwd_kernel<1>, dim3(grid_for(n / 4, 256, out_amax ? 2048 : 8192)), dim3(256), 0, (hipStream_t)stream, a, b,
                       n / 4, (float)((double)scale / denom), gout_dev, add, tanh_of_a, out, out_amax);
    LVT_CHECK_LAUNCH("lvt_mse_bwd_kernel<l1>");
    return LVT_OK;
}

// y = alpha * x (+ add)   -- small glue (gradient scaling / accumulation)
__global__ void lvt_axpy_kernel(const float *__restrict__ x, const float *__restrict__ add, long long n,
                                const float *__restrict__ alpha_dev, float alpha, float *__restrict__ out) {
    const float a = alpha * (alpha_dev ? alpha_dev[0] : 1.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (add ? add[i] : 0.f);
}
extern "C" int lvt_axpy(const float *x, const float *add, long long n, const float *alpha_dev, float alpha,
                        float *out, void *stream) {
    LVT_REQUIRE(x && out && n > 0, "axpy: bad args");
    hipLaunchKernelGGL(lvt_axpy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, add, n,
                       alpha_dev, alpha, out);
    LVT_CHECK_LAUNCH("lvt_axpy_kernel");
    return LVT_OK;
}

// out = g * (1 - y^2)   (tanh backward at the end of the decoder chain)
__global__ void lvt_tanh_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y, long long n4,
                                    float *__restrict__ out, float *__restrict__ out_amax) {
    __shared__ float amax_scratch[4];
    float am = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(g)[i];
        const float4 t = reinterpret_cast<const float4 *>(y)[i];
        const float4 r = make_float4(a.x * (1.f - t.x * t.x), a.y * (1.f - t.y * t.y), a.z * (1.f - t.z * t.z), a.w * (1.f - t.w * t.w));
        reinterpret_cast<float4 *>(out)[i] = r;
        am = fmaxf(am, fmaxf(fmaxf(lvt_absf(r.x), lvt_absf(r.y)), fmaxf(lvt_absf(r.z), lvt_absf(r.w))));
    }
    if (out_amax) lvt_block_amax_commit(am, out_amax, amax_scratch);
}
extern "C" int lvt_tanh_bwd(const float *g, const float *y, long long n, float *out, float *out_amax, void *stream) {
    LVT_REQUIRE(g && y && out && n > 0 && n % 4 == 0, "tanh_bwd: bad args");
    hipLaunchKernelGGL(lvt_tanh_bwd_kernel, dim3(grid_for(n / 4, 256, out_amax ? 2048 : 8192)), dim3(256), 0, (hipStream_t)stream, g, y,
                       n / 4, out, out_amax);
    LVT_CHECK_LAUNCH("lvt_tanh_bwd_kernel");
    return LVT_OK;
}

// x[r][:] += table[r % P][:]   (positional encoding, vt_attention.py:25-50)
__global__ void lvt_add_periodic_kernel(float *__restrict__ x, const float *__restrict__ table, long long rows,
                                        int P, int d4) {
    const long long total = rows * d4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / d4; const int c = i % d4;
        float4 v = reinterpret_cast<float4 *>(x)[i];
        const float4 t = reinterpret_cast<const float4 *>(table)[(r % P) * d4 + c];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        reinterpret_cast<float4 *>(x)[i] = v;
    }
}
extern "C" int lvt_add_periodic(float *x, const float *table, long long rows, int P, int d, void *stream) {
    LVT_REQUIRE(x && table && rows > 0 && P > 0 && d % 4 == 0, "add_periodic: bad args");
    hipLaunchKernelGGL(lvt_add_periodic_kernel, dim3(grid_for(rows * (d / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, x, table, rows, P, d / 4);
    LVT_CHECK_LAUNCH("lvt_add_periodic_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (F.layer_norm, eps 1e-5; vt_attention.py:121,138; videotransformer.py:142)
// one wave per row, row kept in registers (d <= 1024), two-pass mean / variance
// ------------------------------------------------------------------------------------------------
#define LN_MAXV 4   // float4 per lane -> d <= 1024
// the a-priori bound of a LayerNorm output (ONE expression: the P2 form scales its image by the value the consumers read back)
__device__ __forceinline__ float ln_bound(const float *w_amax, const float *b_amax, int d) {
    return fmaf(*w_amax, sqrtf((float)(d - 1)), *b_amax);
}
__device__ __forceinline__ float ln_mix_lo(unsigned h, float k, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
__device__ __forceinline__ float ln_mix_hi(unsigned h, float k, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(k), "v"(c));
    return r;
}
typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ln_f16x2 __attribute__((ext_vector_type(2)));
// the f16x2 split of gemm_engine.hip (f16_split_pair<2048>), bit for bit
__device__ __forceinline__ void ln_split_pair(float a, float b, float s, unsigned &ph, unsigned &pl) {
    const ln_f32x2 v = {a, b};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v * s, ln_f16x2));
    const ln_f32x2 t2 = v * (s * 2048.f);
    const ln_f32x2 r = {ln_mix_lo(ph, -2048.f, t2.x), ln_mix_hi(ph, -2048.f, t2.y)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ln_f16x2));
}
template <int P2>
__global__ void lvt_layernorm_fwd_kernel(const float *__restrict__ x, long long rows, int d, float eps,
                                         const float *__restrict__ w, const float *__restrict__ b,
                                         float *__restrict__ y, float *__restrict__ mean_out,
                                         float *__restrict__ rstd_out, float *__restrict__ y_amax,
                                         const float *__restrict__ w_amax, const float *__restrict__ b_amax,
                                         char *__restrict__ yimg) {
    __shared__ float amax_scratch[4];
    const int lane = threadIdx.x & 63;
    const int d4 = d / 4;
    float am = 0.f;
    float ps = 1.f;
    if (P2) {
        // scale of the image: lvt_f16_scale (gemm_engine.hip) of the bound that block 0 stores below
        const int eb = (int)((__float_as_uint(ln_bound(w_amax, b_amax, d)) >> 23) & 0xffu);
        int se = 268 - eb;
        se = se < 2 ? 2 : (se > 252 ? 252 : se);
        ps = __uint_as_float((unsigned)se << 23);
    }
    if (y_amax && w_amax) {
        // a-priori bound instead of a reduction: |(x - mean) rstd| <= sqrt(d - 1) on every row, so
        // max |y| <= max |w| sqrt(d - 1) + max |b| -- ~4x above the actual maximum of a 16384 x 512 output (2 of the 27
        // binades an f16x2 operand scale has to spare), for one store instead of one atomic per workgroup
        if (blockIdx.x == 0 && threadIdx.x == 0) *y_amax = ln_bound(w_amax, b_amax, d);
        y_amax = nullptr;
    }
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < rows;
         row += ((long long)gridDim.x * blockDim.x) >> 6) {
        const float4 *xp = reinterpret_cast<const float4 *>(x + row * d);
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < d4 ? xp[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if (lane + 64 * i < d4) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / d + eps);
        float4 *yp = reinterpret_cast<float4 *>(y + row * d);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < d4) {
                const float4 ww = reinterpret_cast<const float4 *>(w)[c];
                const float4 bb = reinterpret_cast<const float4 *>(b)[c];
                float4 o;
                o.x = (v[i].x - mean) * rstd * ww.x + bb.x; o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
                o.z = (v[i].z - mean) * rstd * ww.z + bb.z; o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
                yp[c] = o;
                if (P2) {
                    // float4 c = elements 4 c .. 4 c + 3: group (4 c) / 32 of the row's image, 8 bytes into its hi and lo halves
                    uint2 ph, pl;
                    ln_split_pair(o.x, o.y, ps, ph.x, pl.x);
                    ln_split_pair(o.z, o.w, ps, ph.y, pl.y);
                    char *ip = yimg + (long long)row * d * 4 + (c >> 3) * 128 + (c & 7) * 8;
                    *reinterpret_cast<uint2 *>(ip) = ph;
                    *reinterpret_cast<uint2 *>(ip + 64) = pl;
                }
                am = fmaxf(am, fmaxf(fmaxf(lvt_absf(o.x), lvt_absf(o.y)), fmaxf(lvt_absf(o.z), lvt_absf(o.w))));
            }
        }
        if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
    if (y_amax) lvt_block_amax_commit(am, y_amax, amax_scratch);
}
#ifndef LN_FWD_MAX_BLOCKS
#define LN_FWD_MAX_BLOCKS 16384
#endif
extern "C" int lvt_layernorm_fwd(const float *x, long long rows, int d, float eps, const float *w, const float *b,
                                 float *y, float *mean, float *rstd, float *y_amax, const float *w_amax, const float *b_amax,
                                 void *stream) {
    LVT_REQUIRE(x && w && b && y && rows > 0 && d % 4 == 0 && d <= 256 * LN_MAXV, "layernorm_fwd: bad args (d=%d)", d);
    LVT_REQUIRE(!w_amax == !b_amax, "layernorm_fwd: w_amax and b_amax come together");
    hipLaunchKernelGGL(lvt_layernorm_fwd_kernel<0>, dim3(grid_for(rows, 4, (y_amax && !w_amax) ? 2048 : LN_FWD_MAX_BLOCKS)), dim3(256), 0,
                       (hipStream_t)stream, x, rows, d, eps, w, b, y, mean, rstd, y_amax, w_amax, b_amax, (char *)nullptr);
    LVT_CHECK_LAUNCH("lvt_layernorm_fwd_kernel");
    return LVT_OK;
}
extern "C" int lvt_layernorm_fwd_p2(const float *x, long long rows, int d, float eps, const float *w, const float *b,
                                    float *y, void *yp, float *mean, float *rstd, float *y_amax, const float *w_amax,
                                    const float *b_amax, void *stream) {
    LVT_REQUIRE(x && w && b && y && yp && rows > 0 && d % 32 == 0 && d <= 256 * LN_MAXV, "layernorm_fwd_p2: bad args (d=%d)", d);
    LVT_REQUIRE(y_amax && w_amax && b_amax, "layernorm_fwd_p2: the image is scaled by the a-priori bound: y_amax, w_amax, b_amax are required");
    LVT_REQUIRE(((uintptr_t)yp & 127) == 0, "layernorm_fwd_p2: the image must be 128-byte aligned");
    hipLaunchKernelGGL(lvt_layernorm_fwd_kernel<1>, dim3(grid_for(rows, 4, LN_FWD_MAX_BLOCKS)), dim3(256), 0,
                       (hipStream_t)stream, x, rows, d, eps, w, b, y, mean, rstd, y_amax, w_amax, b_amax, (char *)yp);
    LVT_CHECK_LAUNCH("lvt_layernorm_fwd_kernel<p2>");
    return LVT_OK;
}

// dx = rstd * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat)) (+ add);  partial dw/db per workgroup
// 512 workgroups (2 per CU): measured at 16384 x 512, bwd+add 28.0 us against 32.7 us at 1024 and 33 us at 256; more rows in flight
// per wave (LN_BWD_ROWS 2 / 4) measured 28.6 / 39.1 us (tools/profile/ln_time.py)
#ifndef LN_BWD_BLOCKS
#define LN_BWD_BLOCKS 512
#endif
#ifndef LN_BWD_ROWS
#define LN_BWD_ROWS 1
#endif
__global__ __launch_bounds__(256) void lvt_layernorm_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ w, long long rows, int d, const float *__restrict__ add,
    float *__restrict__ dx, float *__restrict__ pdw, float *__restrict__ pdb, float *__restrict__ dx_amax) {
    __shared__ float amax_scratch[4];
    float am = 0.f;
    __shared__ float sdw[4][256 * LN_MAXV];   // per-wave column partials, combined in wave order
    __shared__ float sdb[4][256 * LN_MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d4 = d / 4;
    float4 adw[LN_MAXV], adb[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) { adw[i] = make_float4(0.f, 0.f, 0.f, 0.f); adb[i] = adw[i]; }
    const long long rows_per_block = lvt_cdiv(rows, gridDim.x);
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    // LN_BWD_ROWS rows of a wave are in flight together: their loads are all requested before the first row reduction, and the column
    // partials still take the rows in ascending order (the sums do not depend on LN_BWD_ROWS)
    for (long long row0 = r0 + wave; row0 < r1; row0 += 4 * LN_BWD_ROWS) {
        float4 xh[LN_BWD_ROWS][LN_MAXV], gw[LN_BWD_ROWS][LN_MAXV], zv[LN_BWD_ROWS][LN_MAXV];
        float rs[LN_BWD_ROWS], s1[LN_BWD_ROWS], s2[LN_BWD_ROWS];
#pragma unroll
        for (int u = 0; u < LN_BWD_ROWS; ++u) {
            const long long row = row0 + 4 * u;
            if (row >= r1) continue;
            // the residual row is requested with the other two (it is only needed after the row reductions: loading it there put a
            // third dependent memory round trip into every row of an HBM-bound kernel)
            if (add) {
#pragma unroll
                for (int i = 0; i < LN_MAXV; ++i)
                    if (lane + 64 * i < d4) zv[u][i] = reinterpret_cast<const float4 *>(add + row * d)[lane + 64 * i];
            }
            const float m = mean[row];
            rs[u] = rstd[row];
            const float4 *xp = reinterpret_cast<const float4 *>(x + row * d);
            const float4 *gp = reinterpret_cast<const float4 *>(dy + row * d);
            s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < d4) {
                    const float4 xv = xp[c], gv = gp[c], ww = reinterpret_cast<const float4 *>(w)[c];
                    const float r = rs[u];
                    xh[u][i] = make_float4((xv.x - m) * r, (xv.y - m) * r, (xv.z - m) * r, (xv.w - m) * r);
                    gw[u][i] = make_float4(gv.x * ww.x, gv.y * ww.y, gv.z * ww.z, gv.w * ww.w);
                    const float4 h = xh[u][i], g = gw[u][i];
                    s1[u] += (g.x + g.y) + (g.z + g.w);
                    s2[u] += (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
                    adw[i].x += gv.x * h.x; adw[i].y += gv.y * h.y; adw[i].z += gv.z * h.z; adw[i].w += gv.w * h.w;
                    adb[i].x += gv.x; adb[i].y += gv.y; adb[i].z += gv.z; adb[i].w += gv.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LN_BWD_ROWS; ++u) {
            const long long row = row0 + 4 * u;
            if (row >= r1) continue;
            const float m1 = wave_sum(s1[u]) / d, m2 = wave_sum(s2[u]) / d;
            float4 *op = reinterpret_cast<float4 *>(dx + row * d);
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < d4) {
                    const float4 h = xh[u][i], g = gw[u][i];
                    const float r = rs[u];
                    float4 o;
                    o.x = r * (g.x - m1 - h.x * m2); o.y = r * (g.y - m1 - h.y * m2);
                    o.z = r * (g.z - m1 - h.z * m2); o.w = r * (g.w - m1 - h.w * m2);
                    if (add) {
                        const float4 z = zv[u][i];
                        o.x += z.x; o.y += z.y; o.z += z.z; o.w += z.w;
                    }
                    op[c] = o;
                    am = fmaxf(am, fmaxf(fmaxf(lvt_absf(o.x), lvt_absf(o.y)), fmaxf(lvt_absf(o.z), lvt_absf(o.w))));
                }
            }
        }
    }
    if (dx_amax) lvt_block_amax_commit(am, dx_amax, amax_scratch);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < d4) {
            reinterpret_cast<float4 *>(sdw[wave])[c] = adw[i];
            reinterpret_cast<float4 *>(sdb[wave])[c] = adb[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        pdw[(long long)blockIdx.x * d + c] = ((sdw[0][c] + sdw[1][c]) + sdw[2][c]) + sdw[3][c];
        pdb[(long long)blockIdx.x * d + c] = ((sdb[0][c] + sdb[1][c]) + sdb[2][c]) + sdb[3][c];
    }
}
// out[c] = sum_b partial[b][c]: 8 columns per workgroup, 32 row-lanes per column, four independent running sums per
// lane (the loads of a lane do not wait on each other); lanes are combined in lane order -> fixed summation order.
// blockIdx.y selects the (partial, out) pair, so dw and db of a LayerNorm share one launch.
__global__ __launch_bounds__(256) void lvt_rowsum_partials_kernel(const float *__restrict__ partial0, const float *__restrict__ partial1,
                                                                  int nblk, int n, float *__restrict__ out0, float *__restrict__ out1) {
    __shared__ float red[256];
    const float *partial = blockIdx.y ? partial1 : partial0;
    float *out = blockIdx.y ? out1 : out0;
    constexpr int cols = 8, RL = 256 / cols;
    const int c = blockIdx.x * cols + (threadIdx.x % cols);
    const int rl = threadIdx.x / cols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < n) {
        int b = rl;
        for (; b + 3 * RL < nblk; b += 4 * RL) {
            s0 += partial[(long long)b * n + c]; s1 += partial[(long long)(b + RL) * n + c];
            s2 += partial[(long long)(b + 2 * RL) * n + c]; s3 += partial[(long long)(b + 3 * RL) * n + c];
        }
        for (; b < nblk; b += RL) s0 += partial[(long long)b * n + c];
    }
    float s = (s0 + s1) + (s2 + s3);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < n) {
        for (int k = 1; k < RL; ++k) s += red[k * cols + threadIdx.x];
        out[c] = s;
    }
}
extern "C" size_t lvt_layernorm_bwd_workspace_bytes(int d) { return (size_t)2 * LN_BWD_BLOCKS * d * sizeof(float); }
extern "C" int lvt_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd,
                                 const float *w, long long rows, int d, const float *add, float *dx, float *dw,
                                 float *db, float *dx_amax, void *workspace, size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(dy && x && mean && rstd && w && dx && dw && db && rows > 0 && d % 4 == 0 && d <= 256 * LN_MAXV,
                "layernorm_bwd: bad args");
    LVT_REQUIRE(workspace && workspace_bytes >= lvt_layernorm_bwd_workspace_bytes(d), "layernorm_bwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)(lvt_cdiv(rows, 8) < LN_BWD_BLOCKS ? lvt_cdiv(rows, 8) : LN_BWD_BLOCKS);
    const long long rpb = lvt_cdiv(rows, blocks);
    blocks = (int)lvt_cdiv(rows, rpb);
    float *pdw = (float *)workspace, *pdb = pdw + (size_t)LN_BWD_BLOCKS * d;
    hipLaunchKernelGGL(lvt_layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, s, dy, x, mean, rstd, w, rows, d, add,
                       dx, pdw, pdb, dx_amax);
    LVT_CHECK_LAUNCH("lvt_layernorm_bwd_kernel");
    hipLaunchKernelGGL(lvt_rowsum_partials_kernel, dim3((d + 7) / 8, 2), dim3(256), 0, s, (const float *)pdw, (const float *)pdb,
                       blocks, d, dw, db);
    LVT_CHECK_LAUNCH("lvt_rowsum_partials_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// Subscale slice / context builder for a batch of code clips (reference: DatasetMapper.prepare_slices,
// vidgen/data/dataset_mapper.py:113-149 with vt_utils.py:24-57,104-128 -- per sample on the CPU there).
// One thread per output element; every output is a pure index function of (sample offset (a,b,c), position):
//   slice [b][ch][ti][hi][wi] = video[b][ti*st + a][ch][hi*sh + bo][wi*sw + c]
//   ctx   [b][ch][ot][oh][ow] = video at (ot - kt/2 + a, oh - kh/2 + bo, ow - kw/2 + c) when that position is inside the
//                               clip AND belongs to a slice generated strictly before (a,bo,c) in raster order of the
//                               offsets, else pad_value   (ss_shift of the visibility-masked clip)
//   ignore[b][0][ti][hi][wi] = (ti*st + a) < n_prime ;  slice_idx[b] = (a*sh + bo)*sw + c
// ------------------------------------------------------------------------------------------------
struct SliceGeom { int B, T, nc, H, W, st, sh, sw, kt, kh, kw, n_prime, t, h, w, Tc, Hc, Wc; long long pad; };
__global__ void lvt_slice_context_kernel(const long long *__restrict__ video, const int *__restrict__ abc, SliceGeom g,
                                         long long *__restrict__ ctx, long long *__restrict__ slice,
                                         long long *__restrict__ slice_idx, unsigned char *__restrict__ ignore) {
    const long long n_ctx = (long long)g.B * g.nc * g.Tc * g.Hc * g.Wc;
    const long long n_sl = (long long)g.B * g.nc * g.t * g.h * g.w;
    const long long n_ig = (long long)g.B * g.t * g.h * g.w;
    const long long total = n_ctx + n_sl + n_ig + g.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_ctx) {
            long long r = i;
            const int ow = r % g.Wc; r /= g.Wc;
            const int oh = r % g.Hc; r /= g.Hc;
            const int ot = r % g.Tc; r /= g.Tc;
            const int ch = r % g.nc; const int b = r / g.nc;
            const int a = abc[3 * b], bo = abc[3 * b + 1], c = abc[3 * b + 2];
            const int ti = ot - g.kt / 2 + a, hi = oh - g.kh / 2 + bo, wi = ow - g.kw / 2 + c;
            long long v = g.pad;
            if ((unsigned)ti < (unsigned)g.T && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W) {
                const int mine = ((ti % g.st) * g.sh + (hi % g.sh)) * g.sw + (wi % g.sw);
                if (mine < (a * g.sh + bo) * g.sw + c)
                    v = video[((((long long)b * g.T + ti) * g.nc + ch) * g.H + hi) * g.W + wi];
            }
            ctx[i] = v;
        } else if (i < n_ctx + n_sl) {
            long long r = i - n_ctx;
            const int wi = r % g.w; r /= g.w;
            const int hi = r % g.h; r /= g.h;
            const int ti = r % g.t; r /= g.t;
            const int ch = r % g.nc; const int b = r / g.nc;
            const int a = abc[3 * b], bo = abc[3 * b + 1], c = abc[3 * b + 2];
            slice[i - n_ctx] = video[((((long long)b * g.T + ti * g.st + a) * g.nc + ch) * g.H + hi * g.sh + bo) * g.W + wi * g.sw + c];
        } else if (i < n_ctx + n_sl + n_ig) {
            long long r = i - n_ctx - n_sl;
            r /= (long long)g.h * g.w;
            const int ti = r % g.t; const int b = r / g.t;
            ignore[i - n_ctx - n_sl] = (ti * g.st + abc[3 * b]) < g.n_prime ? 1 : 0;
        } else {
            const int b = (int)(i - n_ctx - n_sl - n_ig);
            slice_idx[b] = (abc[3 * b] * g.sh + abc[3 * b + 1]) * g.sw + abc[3 * b + 2];
        }
    }
}
extern "C" int lvt_slice_context(const long long *video, int B, int T, int nc, int H, int W, const int *abc, int st, int sh,
                                 int sw, int kt, int kh, int kw, int n_prime, long long pad_value, long long *ctx,
                                 long long *slice, long long *slice_idx, unsigned char *ignore, void *stream) {
    LVT_REQUIRE(video && abc && ctx && slice && slice_idx && ignore, "slice_context: null pointer");
    LVT_REQUIRE(B > 0 && nc > 0 && st > 0 && sh > 0 && sw > 0 && T % st == 0 && H % sh == 0 && W % sw == 0 &&
                kt > 0 && kh > 0 && kw > 0, "slice_context: bad geometry");
    SliceGeom g;
    g.B = B; g.T = T; g.nc = nc; g.H = H; g.W = W; g.st = st; g.sh = sh; g.sw = sw; g.kt = kt; g.kh = kh; g.kw = kw;
    g.n_prime = n_prime; g.pad = pad_value;
    g.t = T / st; g.h = H / sh; g.w = W / sw;
    // extent of the shifted clip: 2*(k/2) + (n/s - 1)*s + 1 per dimension, whatever the offset (vt_utils.py:104-128)
    g.Tc = 2 * (kt / 2) + (g.t - 1) * st + 1; g.Hc = 2 * (kh / 2) + (g.h - 1) * sh + 1; g.Wc = 2 * (kw / 2) + (g.w - 1) * sw + 1;
    const long long total = (long long)B * nc * g.Tc * g.Hc * g.Wc + (long long)B * (nc + 1) * g.t * g.h * g.w + B;
    const int blocks = (int)(lvt_cdiv(total, 256) < 4096 ? lvt_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(lvt_slice_context_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, video, abc, g, ctx, slice,
                       slice_idx, ignore);
    LVT_CHECK_LAUNCH("lvt_slice_context_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-sample row gather of a token matrix: out[b][i][:] = x[b][perm[i]][:] for (B, S, d) float32, d % 4 == 0.
// The block-split branch of BlockLocalAttention (vt_attention.py:189-200: view / permute / contiguous between the
// raster token order and the block-by-block order) is this gather; its inverse permutation is its own backward.
// ------------------------------------------------------------------------------------------------
__global__ void lvt_row_gather_kernel(const float *__restrict__ x, const long long *__restrict__ perm, long long B, int S,
                                      int d4, float *__restrict__ out) {
    const long long total = B * S * d4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = i % d4; const long long r = i / d4;
        const int s = r % S; const long long b = r / S;
        reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(x)[(b * S + perm[s]) * d4 + c];
    }
}
extern "C" int lvt_row_gather(const float *x, const long long *perm, long long B, int S, int d, float *out, void *stream) {
    LVT_REQUIRE(x && perm && out && B > 0 && S > 0 && d > 0 && d % 4 == 0 && lvt_aligned16(x) && lvt_aligned16(out),
                "row_gather: bad args");
    const long long total = B * S * (d / 4);
    const int blocks = (int)(lvt_cdiv(total, 256) < 8192 ? lvt_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(lvt_row_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, perm, B, S, d / 4, out);
    LVT_CHECK_LAUNCH("lvt_row_gather_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// max |x| of a tensor into a device scalar (LVT_MATH_F16X2 operand scales for tensors that no engine launch
// produced): *out = max(*out, max |x|).  Non-negative floats order like their bit patterns, so the cross-workgroup
// step is an INTEGER atomic max -- exact and order-independent (the library has no floating-point atomics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lvt_amax_kernel(const float *__restrict__ x, long long n, float *__restrict__ out) {
    __shared__ float scratch[4];
    const long long n4 = n >> 2;
    float m = 0.f;
    if ((((uintptr_t)x) & 15) == 0) {
        float m1 = 0.f;                  // two independent chains: the loads of a thread do not wait on one fmax
        long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long step = (long long)gridDim.x * blockDim.x;
        for (; i + step < n4; i += 2 * step) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i * 4), u = *reinterpret_cast<const float4 *>(x + (i + step) * 4);
            m = fmaxf(m, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
            m1 = fmaxf(m1, fmaxf(fmaxf(lvt_absf(u.x), lvt_absf(u.y)), fmaxf(lvt_absf(u.z), lvt_absf(u.w))));
        }
        if (i < n4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i * 4);
            m = fmaxf(m, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
        }
        m = fmaxf(m, m1);
        for (long long t = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += step) m = fmaxf(m, lvt_absf(x[t]));
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
            m = fmaxf(m, lvt_absf(x[i]));
    }
    // (lvt_absf: the max is over the FINITE entries; an inf / nan element poisons its products through its own hi term)
    lvt_block_amax_commit(m, out, scratch);
}
// the same for up to 64 tensors per launch (the weights of a model, once per pass): blockIdx.y = tensor
struct AmaxTable { const float *x[64]; long long n[64]; float *out[64]; };
__global__ __launch_bounds__(256) void lvt_amax_multi_kernel(const AmaxTable t) {
    __shared__ float scratch[4];
    const float *x = t.x[blockIdx.y];
    const long long n = t.n[blockIdx.y], n4 = (((uintptr_t)x) & 15) == 0 ? n >> 2 : 0;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i * 4);
        m = fmaxf(m, fmaxf(fmaxf(lvt_absf(v.x), lvt_absf(v.y)), fmaxf(lvt_absf(v.z), lvt_absf(v.w))));
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, lvt_absf(x[i]));
    lvt_block_amax_commit(m, t.out[blockIdx.y], scratch);
}
extern "C" int lvt_amax_multi(const lvt_amax_entry *entries, int n, void *stream) {
    LVT_REQUIRE(entries && n > 0, "amax_multi: bad args");
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        AmaxTable t;
        long long biggest = 0;
        for (int i = 0; i < cnt; ++i) {
            const lvt_amax_entry &e = entries[base + i];
            LVT_REQUIRE(e.x && e.out && e.n > 0, "amax_multi: bad entry %d", base + i);
            t.x[i] = e.x; t.n[i] = e.n; t.out[i] = e.out;
            if (e.n > biggest) biggest = e.n;
        }
        hipLaunchKernelGGL(lvt_amax_multi_kernel, dim3(grid_for(biggest / 4 + 1, 256 * 4, 64), cnt), dim3(256), 0, (hipStream_t)stream, t);
        LVT_CHECK_LAUNCH("lvt_amax_multi_kernel");
    }
    return LVT_OK;
}
// *out = max(*out, *a, *b): the bound of a launch whose operand spans two tensors (b may be NULL)
__global__ void lvt_amax_merge_kernel(const float *a, const float *b, float *out) {
    float m = fmaxf(*out, *a);
    if (b) m = fmaxf(m, *b);
    *out = m;
}
extern "C" int lvt_amax_merge(const float *a, const float *b, float *out, void *stream) {
    LVT_REQUIRE(a && out, "amax_merge: bad args");
    hipLaunchKernelGGL(lvt_amax_merge_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, b, out);
    LVT_CHECK_LAUNCH("lvt_amax_merge_kernel");
    return LVT_OK;
}
extern "C" int lvt_amax(const float *x, long long n, float *out, void *stream) {
    LVT_REQUIRE(x && out && n > 0, "amax: bad args");
    hipLaunchKernelGGL(lvt_amax_kernel, dim3(grid_for(n / 4 + 1, 256 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    LVT_CHECK_LAUNCH("lvt_amax_kernel");
    return LVT_OK;
}
