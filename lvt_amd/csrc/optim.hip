// Multi-tensor optimizer step: a handful of launches update every parameter of a sub-network.
// The reference builds one torch.optim param group per parameter (vidgen/solver/build.py:12-43), which
// turns an optimizer step into hundreds of tiny launches (394 tensors x ~6 kernels for DSFVT).  Here up to
// 64 tensors travel BY VALUE in the kernel arguments (no table upload, no host sync) and each workgroup
// updates one 16K-element chunk.  Math follows torch.optim.Adam / torch.optim.RMSprop (non-amsgrad,
// non-centered) operation by operation.
#include "lvt_common.h"
#include <math.h>

typedef lvt_opt_entry LvtOptEntry;   // declared in lvt_hip.h; mirrored by ctypes in lvt_amd/hip/binding.py
#define OPT_CHUNK 16384
#define OPT_MAXT 64

struct OptArgs {
    int n;
    int chunk_prefix[OPT_MAXT + 1];
    LvtOptEntry e[OPT_MAXT];
};

__device__ __forceinline__ int find_tensor(const OptArgs &a, int blk) {
    int t = 0;
    while (t + 1 < a.n && a.chunk_prefix[t + 1] <= blk) ++t;
    return t;
}

__global__ __launch_bounds__(256) void lvt_adam_kernel(const OptArgs a, float beta1, float beta2, float eps,
                                                       float bc1, float bc2_sqrt) {
    const int t = find_tensor(a, blockIdx.x);
    const LvtOptEntry e = a.e[t];
    const long long lo = (long long)(blockIdx.x - a.chunk_prefix[t]) * OPT_CHUNK;
    const long long hi = min(e.n, lo + OPT_CHUNK);
    const float step = e.lr / bc1;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float g = e.g[i];
        const float p = e.p[i];
        if (e.wd != 0.f) g += e.wd * p;
        const float m = e.s0[i] + (1.f - beta1) * (g - e.s0[i]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float v = beta2 * e.s1[i] + (1.f - beta2) * g * g;          // mul_(beta2).addcmul_(g, g, 1 - beta2)
        e.s0[i] = m; e.s1[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        e.p[i] = p - step * (m / denom);
    }
}

__global__ __launch_bounds__(256) void lvt_rmsprop_kernel(const OptArgs a, float alpha, float eps, float momentum) {
    const int t = find_tensor(a, blockIdx.x);
    const LvtOptEntry e = a.e[t];
    const long long lo = (long long)(blockIdx.x - a.chunk_prefix[t]) * OPT_CHUNK;
    const long long hi = min(e.n, lo + OPT_CHUNK);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float g = e.g[i];
        const float p = e.p[i];
        if (e.wd != 0.f) g += e.wd * p;
        const float sq = alpha * e.s0[i] + (1.f - alpha) * g * g;
        e.s0[i] = sq;
        const float avg = sqrtf(sq) + eps;
        if (momentum > 0.f) {
            const float buf = momentum * e.s1[i] + g / avg;
            e.s1[i] = buf;
            e.p[i] = p - e.lr * buf;
        } else {
            e.p[i] = p - e.lr * (g / avg);
        }
    }
}

template <typename Launch>
static int for_each_group(const LvtOptEntry *host, int n, Launch launch) {
    for (int base = 0; base < n; base += OPT_MAXT) {
        OptArgs a;
        a.n = (n - base) < OPT_MAXT ? (n - base) : OPT_MAXT;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            a.e[i] = host[base + i];
            if (!a.e[i].p || !a.e[i].g || !a.e[i].s0 || a.e[i].n <= 0) {
                lvt_set_error("optimizer: bad entry %d", base + i);
                return LVT_EINVAL;
            }
            a.chunk_prefix[i] = blocks;
            blocks += (int)lvt_cdiv(a.e[i].n, OPT_CHUNK);
        }
        a.chunk_prefix[a.n] = blocks;
        launch(a, blocks);
        LVT_CHECK_LAUNCH("optimizer kernel");
    }
    return LVT_OK;
}

// p, exp_avg (s0), exp_avg_sq (s1) updated in place; `step` is the 1-based step count of these tensors.
extern "C" int lvt_adam_step(const LvtOptEntry *host, int n, float beta1, float beta2, float eps, int step,
                             void *stream) {
    LVT_REQUIRE(host && n > 0 && step > 0, "adam_step: bad args");
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    return for_each_group(host, n, [&](const OptArgs &a, int blocks) {
        hipLaunchKernelGGL(lvt_adam_kernel, dim3(blocks), dim3(256), 0, s, a, beta1, beta2, eps, (float)bc1,
                           (float)sqrt(bc2));
    });
}

// p, square_avg (s0), momentum_buffer (s1, may be NULL when momentum == 0) updated in place.
extern "C" int lvt_rmsprop_step(const LvtOptEntry *host, int n, float alpha, float eps, float momentum,
                                void *stream) {
    LVT_REQUIRE(host && n > 0, "rmsprop_step: bad args");
    hipStream_t s = (hipStream_t)stream;
    return for_each_group(host, n, [&](const OptArgs &a, int blocks) {
        hipLaunchKernelGGL(lvt_rmsprop_kernel, dim3(blocks), dim3(256), 0, s, a, alpha, eps, momentum);
    });
}
