// Nearest code of ONE wide codebook (CODEBOOK.NUM == 1: `VQEmbedding` used directly, vidgen/modeling/meta_arch/vqvae.py:25-27,
// vq/vq_utils.py:13-20).  The product quantiser's search kernels (vq.hip) keep a whole 64-d codebook group in LDS; a 256-d
// codebook does not fit there, and this form is not on the headline path, so it is composed from what exists: the scores
// s = x E^T come from the GEMM engine (fp32-class arithmetic in every math mode), and this kernel finishes the search:
//     idx[r] = argmax_k ( s[r][k] - |e_k|^2 / 2 )          (= argmin_k |x_r - e_k|^2: the row constant |x_r|^2 dropped),
// lowest k on exact ties (torch.min).  One wave per row; the half norms are recomputed per workgroup into LDS
// (K x D fused multiply-adds, in ascending d order).
#include "lvt_common.h"

#define VQS_MAXK 2048
__global__ __launch_bounds__(256) void lvt_vq_argmax_scores_kernel(const float *__restrict__ s, long long rows, int K, long long ld,
                                                                   const float *__restrict__ E, int D, long long *__restrict__ idx) {
    __shared__ float hn[VQS_MAXK];
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float *e = E + (long long)k * D;
        float a = 0.f;
        for (int d = 0; d < D; ++d) a = fmaf(e[d], e[d], a);
        hn[k] = 0.5f * a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < rows; r += ((long long)gridDim.x * blockDim.x) >> 6) {
        const float *sr = s + r * ld;
        float best = -__builtin_inff();
        int bk = 0x7fffffff;
        for (int k = lane; k < K; k += 64) {
            const float v = sr[k] - hn[k];
            if (v > best || (v == best && k < bk) || bk == 0x7fffffff) { best = v; bk = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int ok = __shfl_xor(bk, o, 64);
            if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; }
        }
        if (lane == 0) idx[r] = bk;
    }
}

extern "C" int lvt_vq_argmax_scores(const float *scores, long long rows, int K, long long ld, const float *codebook, int D,
                                    long long *idx, void *stream) {
    LVT_REQUIRE(scores && codebook && idx && rows > 0, "vq_argmax_scores: null pointer / no rows");
    LVT_REQUIRE(K > 0 && K <= VQS_MAXK && D > 0 && ld >= K, "vq_argmax_scores: K=%d (<= %d), D=%d, ld=%lld", K, VQS_MAXK, D, ld);
    const long long blocks = lvt_cdiv(rows, 4) < 4096 ? lvt_cdiv(rows, 4) : 4096;
    hipLaunchKernelGGL(lvt_vq_argmax_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scores, rows, K, ld,
                       codebook, D, idx);
    LVT_CHECK_LAUNCH("lvt_vq_argmax_scores_kernel");
    return LVT_OK;
}
