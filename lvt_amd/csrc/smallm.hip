// Small-M GEMM for incremental decoding: C[M<=64][N] = epi(alpha * A[M][K] . B), weights streamed once.
// The 128x128 MFMA tile engine needs ~30 us for an M=16 problem (one workgroup per 128 columns walking the
// whole K range); a decoding step is ~60 such GEMMs.
//
// tb == 0 (B[n][k], k contiguous -- every Linear weight): matrix-core kernel.  A workgroup owns 32 output columns
// for all rows; operands are staged through LDS with coalesced loads, split exactly into three bf16 terms in
// registers and multiplied with the six v_mfma_f32_32x32x16_bf16 of the fp32-accurate product (see
// gemm_engine.hip); the waves' K partials are summed through LDS in a fixed order.
// tb == 1 (B[k][n], n contiguous): LDS-staged fp32 FMA kernel, a workgroup owns 16 columns.
#include "lvt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SmallParams {
    int M, N, K, tb;
    const float *A; long long lda;
    const float *B; long long ldb;
    float *C; long long ldc;
    long long sA, sB, sC;             // per-batch (blockIdx.y) offsets of A (0: shared), B and C
    float alpha; int flags;
    const float *bias; const float *res; long long ldr;
    const int *pos; long long c_pos, r_pos;   // device-side row cursor: C += pos[0]*c_pos, res += pos[0]*r_pos (hipGraph replay)
};

#define SM_KC 128
#define SM_LD 132

template <int TB>
__global__ __launch_bounds__(256) void lvt_gemm_smallm_kernel(const SmallParams p) {
    __shared__ __attribute__((aligned(16))) float As[64 * SM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[16 * SM_LD];
    const int tid = threadIdx.x, nl = tid >> 4, mg = tid & 15;
    const int n0 = blockIdx.x * 16, z = blockIdx.y;
    const float *B = p.B + z * p.sB;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += SM_KC) {
        // A chunk: M rows x 128 floats
        for (int u = tid; u < p.M * (SM_KC / 4); u += 256) {
            const int m = u / (SM_KC / 4), q = u % (SM_KC / 4);
            const int k = k0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < p.K) v = *reinterpret_cast<const float4 *>(p.A + (long long)m * p.lda + k);
            *reinterpret_cast<float4 *>(&As[m * SM_LD + q * 4]) = v;
        }
        if (TB == 0) {                // B(k, n) at B[n*ldb + k]
            for (int u = tid; u < 16 * (SM_KC / 4); u += 256) {
                const int n = u / (SM_KC / 4), q = u % (SM_KC / 4);
                const int k = k0 + q * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K && n0 + n < p.N) v = *reinterpret_cast<const float4 *>(B + (long long)(n0 + n) * p.ldb + k);
                *reinterpret_cast<float4 *>(&Bs[n * SM_LD + q * 4]) = v;
            }
        } else {                      // B(k, n) at B[k*ldb + n]: 16 consecutive n per k row
            for (int u = tid; u < SM_KC * 4; u += 256) {
                const int kk = u >> 2, nq = u & 3;
                const int k = k0 + kk;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K && n0 + nq * 4 < p.N) v = *reinterpret_cast<const float4 *>(B + (long long)k * p.ldb + n0 + nq * 4);
                Bs[(nq * 4 + 0) * SM_LD + kk] = v.x; Bs[(nq * 4 + 1) * SM_LD + kk] = v.y;
                Bs[(nq * 4 + 2) * SM_LD + kk] = v.z; Bs[(nq * 4 + 3) * SM_LD + kk] = v.w;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < SM_KC; kk += 4) {
            const float4 b4 = *reinterpret_cast<const float4 *>(&Bs[nl * SM_LD + kk]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 a4 = *reinterpret_cast<const float4 *>(&As[(mg + 16 * r) * SM_LD + kk]);
                acc[r] = fmaf(a4.x, b4.x, acc[r]); acc[r] = fmaf(a4.y, b4.y, acc[r]);
                acc[r] = fmaf(a4.z, b4.z, acc[r]); acc[r] = fmaf(a4.w, b4.w, acc[r]);
            }
        }
        __syncthreads();
    }
    const int n = n0 + nl;
    if (n >= p.N) return;
    const long long cur = p.pos ? (long long)p.pos[0] : 0;
    float *Cp = p.C + cur * p.c_pos;
    const float *Rp = p.res ? p.res + cur * p.r_pos : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mg + 16 * r;
        if (m >= p.M) continue;
        float v = acc[r] * p.alpha;
        if (p.flags & LVT_EPI_BIAS) v += p.bias[n];
        if (p.flags & LVT_EPI_RESIDUAL) v += Rp[(long long)m * p.ldr + n];
        if (p.flags & LVT_EPI_RELU) v = fmaxf(v, 0.f);
        Cp[z * p.sC + (long long)m * p.ldc + n] = v;
    }
}

// ---- matrix-core path (tb == 0) -------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sm_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 8 floats -> three planes of 8 bf16 (exact 3-way split, round-to-nearest-even at every level)
__device__ __forceinline__ void sm_split8(const float4 lo, const float4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = sm_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __uint_as_float(a[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(a[i] & 0xffff0000u);
        b[i] = sm_cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(b[i] << 16), s1 = r1 - __uint_as_float(b[i] & 0xffff0000u);
        c[i] = sm_cvt_pk_bf16(s0, s1);
    }
    const uint4 ua = make_uint4(a[0], a[1], a[2], a[3]), ub = make_uint4(b[0], b[1], b[2], b[3]),
                uc = make_uint4(c[0], c[1], c[2], c[3]);
    p1 = *reinterpret_cast<const bf16x8 *>(&ua); p2 = *reinterpret_cast<const bf16x8 *>(&ub);
    p3 = *reinterpret_cast<const bf16x8 *>(&uc);
}
#define SMM_WAVES 8
#define SMM_KC 128                    // k per staged chunk: 8 MFMA steps, one per wave
#define SMM_LDK (SMM_KC + 4)          // LDS row stride (floats): 16-byte fragment reads of 16 rows hit 16 bank groups
// Direct per-lane fragment loads (lane = row) give the texture addresser one 16-byte request per lane -- measured
// ~1 lane per cycle per CU, 5.6 us for 64x512x512.  Instead the 256 threads fetch every 128-k chunk of A (64 rows)
// and B (32 rows) with fully coalesced 16-byte loads (a wave covers one 512-byte row segment), park it in LDS as
// fp32, and the waves read their MFMA fragments from there (conflict-free ds_read_b128).  Chunk c+1 is in flight
// in registers while chunk c is multiplied; each of the 8 waves takes one of the 8 k-steps (the split / MFMA chains
// of two waves share a SIMD and overlap) and the partial tiles are added in wave order through LDS.
template <int TM>      // TM = number of 32-row tiles (1: M <= 32, 2: M <= 64)
__global__ __launch_bounds__(64 * SMM_WAVES) void lvt_gemm_smallm_mfma_kernel(const SmallParams p) {
    constexpr int AR = 32 * TM;                                   // staged A rows
    constexpr int NA = AR * (SMM_KC / 4) / (64 * SMM_WAVES);      // float4 per thread per chunk (A)
    constexpr int NB = 32 * (SMM_KC / 4) / (64 * SMM_WAVES);      // (B)
    constexpr int STAGE = 2 * (AR + 32) * SMM_LDK, REDN = SMM_WAVES * TM * 16 * 64;
    __shared__ __attribute__((aligned(16))) float smem[STAGE > REDN ? STAGE : REDN];     // staging, then the reduction
    float (*As)[AR][SMM_LDK] = reinterpret_cast<float (*)[AR][SMM_LDK]>(smem);
    float (*Bs)[32][SMM_LDK] = reinterpret_cast<float (*)[32][SMM_LDK]>(smem + 2 * AR * SMM_LDK);
    float (*red)[TM][16][64] = reinterpret_cast<float (*)[TM][16][64]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32, z = blockIdx.y;
    const int mb = blockIdx.z * AR;                               // row block (M > 64: one workgroup per 64 rows and tile)
    const float *B = p.B + z * p.sB;
    const float *A = p.A + z * p.sA;
    const int nchunks = (p.K + SMM_KC - 1) / SMM_KC;
    constexpr int RP = 64 * SMM_WAVES / 32;                       // rows covered per pass of the workgroup
    const int q4 = (tid & 31) * 4, r0 = tid >> 5;                 // this thread's float4 column and first row

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 ra[NA], rb[NB];
    auto fetch = [&](int c) {
        const int k = c * SMM_KC + q4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mb + r0 + RP * i;
            ra[i] = (m < p.M && k < p.K) ? *reinterpret_cast<const float4 *>(A + (long long)m * p.lda + k)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + r0 + RP * i;
            rb[i] = (n < p.N && k < p.K) ? *reinterpret_cast<const float4 *>(B + (long long)n * p.ldb + k)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4 *>(&As[buf][r0 + RP * i][q4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4 *>(&Bs[buf][r0 + RP * i][q4]) = rb[i];
    };

    fetch(0);
    park(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch(c + 1);
        {
            const int kk = 16 * wave;                             // k offset of this wave's step inside the chunk
            const int ko = kk + 8 * half;
            if (c * SMM_KC + kk < p.K) {
            bf16x8 b[3], a[TM][3];
            sm_split8(*reinterpret_cast<const float4 *>(&Bs[buf][l31][ko]), *reinterpret_cast<const float4 *>(&Bs[buf][l31][ko + 4]),
                      b[0], b[1], b[2]);
#pragma unroll
            for (int t = 0; t < TM; ++t)
                sm_split8(*reinterpret_cast<const float4 *>(&As[buf][32 * t + l31][ko]),
                          *reinterpret_cast<const float4 *>(&As[buf][32 * t + l31][ko + 4]), a[t][0], a[t][1], a[t][2]);
            constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][TA[q]], b[TB[q]], acc[t], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            park(buf ^ 1);            // the other buffer was last read before the barrier that ended chunk c-1
            __syncthreads();
        }
    }
    // fixed-order sum of the waves' partial tiles (the staging memory is reused): wave w finalises accumulator
    // registers 2w, 2w+1
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    const int col = n0 + l31;
    if (col >= p.N) return;
    const long long cur = p.pos ? (long long)p.pos[0] : 0;
    float *Cp = p.C + cur * p.c_pos;
    const float *Rp = p.res ? p.res + cur * p.r_pos : nullptr;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int rr = 0; rr < 16 / SMM_WAVES; ++rr) {
            const int r = (16 / SMM_WAVES) * wave + rr;
            const int m = mb + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= p.M) continue;
            float v = red[0][t][r][lane];
#pragma unroll
            for (int w = 1; w < SMM_WAVES; ++w) v += red[w][t][r][lane];
            v *= p.alpha;
            if (p.flags & LVT_EPI_BIAS) v += p.bias[col];
            if (p.flags & LVT_EPI_RESIDUAL) v += Rp[(long long)m * p.ldr + col];
            if (p.flags & LVT_EPI_RELU) v = fmaxf(v, 0.f);
            Cp[z * p.sC + (long long)m * p.ldc + col] = v;
        }
}

extern "C" int lvt_gemm_smallm_f32(int M, int N, int K, int tb, const float *A, long long lda, const float *B,
                                   long long ldb, float *C, long long ldc, int batch, long long sB, long long sC,
                                   float alpha, int flags, const float *bias, const float *res, long long ldr,
                                   const int *pos, long long c_pos, long long r_pos, void *stream) {
    LVT_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "gemm_smallm: bad shape (M=%d)", M);
    LVT_REQUIRE(M <= 64 || (tb == 0 && K % 8 == 0), "gemm_smallm: M=%d > 64 needs the k-contiguous layout (tb == 0, K %% 8 == 0)", M);
    LVT_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lvt_aligned16(A) && lvt_aligned16(B), "gemm_smallm: alignment");
    LVT_REQUIRE(tb == 0 || N % 4 == 0, "gemm_smallm: tb=1 needs N %% 4 == 0");
    LVT_REQUIRE(!(flags & ~(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU)), "gemm_smallm: unsupported flag");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "gemm_smallm: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "gemm_smallm: RESIDUAL without res");
    SmallParams p;
    p.M = M; p.N = N; p.K = K; p.tb = tb; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.sA = 0; p.sB = sB; p.sC = sC; p.alpha = alpha; p.flags = flags; p.bias = bias; p.res = res; p.ldr = ldr;
    p.pos = pos; p.c_pos = c_pos; p.r_pos = r_pos;
    if (tb == 0 && K % 8 == 0) {
        dim3 grid((unsigned)lvt_cdiv(N, 32), (unsigned)batch, (unsigned)(M <= 32 ? 1 : lvt_cdiv(M, 64)));
        if (M <= 32) hipLaunchKernelGGL(lvt_gemm_smallm_mfma_kernel<1>, grid, dim3(64 * SMM_WAVES), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(lvt_gemm_smallm_mfma_kernel<2>, grid, dim3(64 * SMM_WAVES), 0, (hipStream_t)stream, p);
        LVT_CHECK_LAUNCH("lvt_gemm_smallm_mfma_kernel");
        return LVT_OK;
    }
    dim3 grid((unsigned)lvt_cdiv(N, 16), (unsigned)batch);
    if (tb == 0) hipLaunchKernelGGL(lvt_gemm_smallm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lvt_gemm_smallm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    LVT_CHECK_LAUNCH("lvt_gemm_smallm_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// Split-K form for long reductions.  One workgroup per 32 output columns walks K in 128-deep chunks at ~1.3 us per
// chunk (a memory round trip; the arithmetic of a chunk is a quarter of that), and a 64 x 512 x 2048 product has only
// 16 such workgroups: 23 us with 240 CUs idle.  Here the k range is cut into `splits` equal parts that run as the
// batch dimension of the same kernel (raw partial tiles into the workspace), and a second launch adds the parts in
// split order and applies the epilogue -- deterministic, ~8 us for the same product.
// ------------------------------------------------------------------------------------------------
__global__ void lvt_smallm_reduce_kernel(const float *__restrict__ ws, int splits, int M, int N, float *__restrict__ C,
                                         long long ldc, float alpha, int flags, const float *__restrict__ bias,
                                         const float *__restrict__ res, long long ldr, const int *__restrict__ pos,
                                         long long c_pos, long long r_pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // float4 index over M x N
    if (pos) { const long long cur = pos[0]; C += cur * c_pos; if (res) res += cur * r_pos; }
    const int n4 = N / 4;
    if (i >= M * n4) return;
    const int m = i / n4, c = (i - m * n4) * 4;
    const long long plane = (long long)M * N;
    float4 t[8];
    float4 s = *reinterpret_cast<const float4 *>(ws + (long long)m * N + c);
    for (int k0 = 1; k0 < splits; k0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            t[u] = (k0 + u < splits) ? *reinterpret_cast<const float4 *>(ws + (k0 + u) * plane + (long long)m * N + c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    float v[4] = {s.x * alpha, s.y * alpha, s.z * alpha, s.w * alpha};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (flags & LVT_EPI_BIAS) v[j] += bias[c + j];
        if (flags & LVT_EPI_RESIDUAL) v[j] += res[(long long)m * ldr + c + j];
        if (flags & LVT_EPI_RELU) v[j] = fmaxf(v[j], 0.f);
        C[(long long)m * ldc + c + j] = v[j];
    }
}

extern "C" size_t lvt_gemm_smallm_splitk_workspace_bytes(int M, int N, int splits) {
    return (size_t)(splits > 1 ? splits : 1) * M * N * sizeof(float);
}

static int smallm_partial(int M, int N, int K, int splits, const float *A, long long lda, const float *B,
                          long long ldb, void *workspace, size_t workspace_bytes, hipStream_t s, const char *who) {
    LVT_REQUIRE(A && B && M > 0 && N > 0 && K > 0, "%s: bad shape (M=%d)", who, M);
    LVT_REQUIRE(splits >= 2 && K % splits == 0 && (K / splits) % 8 == 0, "%s: K=%d is not %d ranges of a multiple of 8", who, K, splits);
    LVT_REQUIRE(N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lvt_aligned16(A) && lvt_aligned16(B), "%s: alignment", who);
    if (!workspace || workspace_bytes < lvt_gemm_smallm_splitk_workspace_bytes(M, N, splits) || !lvt_aligned16(workspace)) {
        lvt_set_error("%s: workspace too small or misaligned", who);
        return LVT_EWORKSPACE;
    }
    const int Kc = K / splits;
    SmallParams p;
    p.M = M; p.N = N; p.K = Kc; p.tb = 0; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb;
    p.C = (float *)workspace; p.ldc = N; p.sA = Kc; p.sB = Kc; p.sC = (long long)M * N;
    p.alpha = 1.f; p.flags = 0; p.bias = nullptr; p.res = nullptr; p.ldr = 0;
    p.pos = nullptr; p.c_pos = 0; p.r_pos = 0;
    dim3 grid((unsigned)lvt_cdiv(N, 32), (unsigned)splits, (unsigned)(M <= 32 ? 1 : lvt_cdiv(M, 64)));
    if (M <= 32) hipLaunchKernelGGL(lvt_gemm_smallm_mfma_kernel<1>, grid, dim3(64 * SMM_WAVES), 0, s, p);
    else hipLaunchKernelGGL(lvt_gemm_smallm_mfma_kernel<2>, grid, dim3(64 * SMM_WAVES), 0, s, p);
    LVT_CHECK_LAUNCH("lvt_gemm_smallm_mfma_kernel");
    return LVT_OK;
}

extern "C" int lvt_gemm_smallm_splitk_f32(int M, int N, int K, int splits, const float *A, long long lda, const float *B,
                                          long long ldb, float *C, long long ldc, float alpha, int flags,
                                          const float *bias, const float *res, long long ldr, const int *pos,
                                          long long c_pos, long long r_pos, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    LVT_REQUIRE(C, "gemm_smallm_splitk: null output");
    LVT_REQUIRE(!(flags & ~(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU)), "gemm_smallm_splitk: unsupported flag");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "gemm_smallm_splitk: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "gemm_smallm_splitk: RESIDUAL without res");
    hipStream_t s = (hipStream_t)stream;
    const int rc = smallm_partial(M, N, K, splits, A, lda, B, ldb, workspace, workspace_bytes, s, "gemm_smallm_splitk");
    if (rc) return rc;
    const int total4 = M * (N / 4);
    hipLaunchKernelGGL(lvt_smallm_reduce_kernel, dim3((unsigned)lvt_cdiv(total4, 256)), dim3(256), 0, s,
                       (const float *)workspace, splits, M, N, C, ldc, alpha, flags, bias, res, ldr, pos, c_pos, r_pos);
    LVT_CHECK_LAUNCH("lvt_smallm_reduce_kernel");
    return LVT_OK;
}

// ------------------------------------------------------------------------------------------------
// Partial products only + the reduction folded into the LayerNorm that consumes them.  In a decoder layer both
// k = 512..1024 products that end in a residual (attention output projection, FFN down-projection) feed a LayerNorm:
//     x = sum_s partial[s] (+ bias) (+ res);   y = LN(x) * w + b
// so the split-K reduction costs no launch of its own, and the products run as 4 x 16 workgroups of one 128-deep
// chunk each instead of 16 workgroups walking four chunks.  One wave per row; the split order is fixed.
// ------------------------------------------------------------------------------------------------
extern "C" int lvt_gemm_smallm_partial_f32(int M, int N, int K, int splits, const float *A, long long lda, const float *B,
                                           long long ldb, void *workspace, size_t workspace_bytes, void *stream) {
    return smallm_partial(M, N, K, splits, A, lda, B, ldb, workspace, workspace_bytes, (hipStream_t)stream,
                          "gemm_smallm_partial");
}

#define SLN_MAXV 4          // float4 per lane -> d <= 1024
__global__ __launch_bounds__(256) void lvt_splitsum_layernorm_kernel(const float *__restrict__ ws, int splits, int rows,
                                                                    int d, const float *__restrict__ bias,
                                                                    const float *__restrict__ res, long long ldr,
                                                                    float *__restrict__ x_out, float eps,
                                                                    const float *__restrict__ w, const float *__restrict__ b,
                                                                    float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const int d4 = d / 4;
    const long long plane = (long long)rows * d;
    float4 v[SLN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SLN_MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < d4) {
            const float *src = ws + (long long)row * d + c * 4;
            // the loads of up to eight partial tiles are issued together; they are added in split order
            float4 a = *reinterpret_cast<const float4 *>(src);
            for (int k0 = 1; k0 < splits; k0 += 8) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    t[u] = (k0 + u < splits) ? *reinterpret_cast<const float4 *>(src + (k0 + u) * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
            }
            if (bias) { const float4 t = reinterpret_cast<const float4 *>(bias)[c]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
            if (res) { const float4 t = *reinterpret_cast<const float4 *>(res + (long long)row * ldr + c * 4); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
            v[i] = a;
            reinterpret_cast<float4 *>(x_out + (long long)row * d)[c] = a;
            s += (a.x + a.y) + (a.z + a.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < SLN_MAXV; ++i) {
        if (lane + 64 * i < d4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / d + eps);
#pragma unroll
    for (int i = 0; i < SLN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < d4) {
            const float4 ww = reinterpret_cast<const float4 *>(w)[c], bb = reinterpret_cast<const float4 *>(b)[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * ww.x + bb.x; o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
            o.z = (v[i].z - mean) * rstd * ww.z + bb.z; o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
            reinterpret_cast<float4 *>(y + (long long)row * d)[c] = o;
        }
    }
}

extern "C" int lvt_splitsum_layernorm_fwd(const float *partials, int splits, int rows, int d, const float *bias,
                                          const float *res, long long ldr, float *x_out, float eps, const float *w,
                                          const float *b, float *y, void *stream) {
    LVT_REQUIRE(partials && x_out && w && b && y && splits >= 1 && rows > 0, "splitsum_layernorm: bad args");
    LVT_REQUIRE(d % 4 == 0 && d <= 256 * SLN_MAXV && (!res || ldr % 4 == 0), "splitsum_layernorm: d=%d must be a multiple of 4, <= 1024", d);
    LVT_REQUIRE(lvt_aligned16(partials) && lvt_aligned16(x_out) && lvt_aligned16(y) && lvt_aligned16(w) && lvt_aligned16(b) &&
                (!bias || lvt_aligned16(bias)) && (!res || lvt_aligned16(res)), "splitsum_layernorm: 16-byte alignment");
    hipLaunchKernelGGL(lvt_splitsum_layernorm_kernel, dim3((unsigned)lvt_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       partials, splits, rows, d, bias, res, ldr, x_out, eps, w, b, y);
    LVT_CHECK_LAUNCH("lvt_splitsum_layernorm_kernel");
    return LVT_OK;
}
