// Small-M GEMM for incremental decoding: C[M<=64][N] = epi(alpha * A[M][K] . B), weights streamed once.
// The 128x128 MFMA tile engine needs ~30 us for an M=16 problem (one workgroup per 128 columns walking the
// whole K range); a decoding step is ~75 such GEMMs.  Here a workgroup owns 16 output columns for all rows,
// stages A / B chunks of 128 k through LDS with 16-byte accesses, and the grid is N/16 (x heads) wide, so the
// weight matrix is read by many CUs in parallel.  fp32 FMA, k ascending: same summation order as a per-thread
// fmaf chain.
#include "lvt_common.h"

struct SmallParams {
    int M, N, K, tb;
    const float *A; long long lda;
    const float *B; long long ldb;
    float *C; long long ldc;
    long long sB, sC;                 // per-batch (blockIdx.y) offsets of B and C (A is shared)
    float alpha; int flags;
    const float *bias; const float *res; long long ldr;
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mg + 16 * r;
        if (m >= p.M) continue;
        float v = acc[r] * p.alpha;
        if (p.flags & LVT_EPI_BIAS) v += p.bias[n];
        if (p.flags & LVT_EPI_RESIDUAL) v += p.res[(long long)m * p.ldr + n];
        if (p.flags & LVT_EPI_RELU) v = fmaxf(v, 0.f);
        p.C[z * p.sC + (long long)m * p.ldc + n] = v;
    }
}

extern "C" int lvt_gemm_smallm_f32(int M, int N, int K, int tb, const float *A, long long lda, const float *B,
                                   long long ldb, float *C, long long ldc, int batch, long long sB, long long sC,
                                   float alpha, int flags, const float *bias, const float *res, long long ldr,
                                   void *stream) {
    LVT_REQUIRE(A && B && C && M > 0 && M <= 64 && N > 0 && K > 0 && batch > 0, "gemm_smallm: bad shape (M=%d)", M);
    LVT_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lvt_aligned16(A) && lvt_aligned16(B), "gemm_smallm: alignment");
    LVT_REQUIRE(tb == 0 || N % 4 == 0, "gemm_smallm: tb=1 needs N %% 4 == 0");
    LVT_REQUIRE(!(flags & ~(LVT_EPI_BIAS | LVT_EPI_RESIDUAL | LVT_EPI_RELU)), "gemm_smallm: unsupported flag");
    LVT_REQUIRE(!(flags & LVT_EPI_BIAS) || bias, "gemm_smallm: BIAS without bias");
    LVT_REQUIRE(!(flags & LVT_EPI_RESIDUAL) || res, "gemm_smallm: RESIDUAL without res");
    SmallParams p;
    p.M = M; p.N = N; p.K = K; p.tb = tb; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.sB = sB; p.sC = sC; p.alpha = alpha; p.flags = flags; p.bias = bias; p.res = res; p.ldr = ldr;
    dim3 grid((unsigned)lvt_cdiv(N, 16), (unsigned)batch);
    if (tb == 0) hipLaunchKernelGGL(lvt_gemm_smallm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lvt_gemm_smallm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    LVT_CHECK_LAUNCH("lvt_gemm_smallm_kernel");
    return LVT_OK;
}
