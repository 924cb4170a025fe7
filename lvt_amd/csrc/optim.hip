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

// float4 lanes when the chunk of every stream of the tensor is 16-byte aligned and the tensor length is a multiple of 4
__device__ __forceinline__ bool opt_vec_ok(const LvtOptEntry &e, long long lo) {
    const uintptr_t a = (uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.s0 | (uintptr_t)(e.s1 ? e.s1 : e.s0);
    return (a & 15) == 0 && (e.n & 3) == 0 && (lo & 3) == 0;
}

__global__ __launch_bounds__(256) void lvt_adam_kernel(const OptArgs a, float beta1, float beta2, float eps,
                                                       float bc1, float bc2_sqrt) {
    const int t = find_tensor(a, blockIdx.x);
    const LvtOptEntry e = a.e[t];
    const long long lo = (long long)(blockIdx.x - a.chunk_prefix[t]) * OPT_CHUNK;
    const long long hi = min(e.n, lo + OPT_CHUNK);
    const float step = e.lr / bc1;
    auto upd = [&](float g, float p, float &m, float &v) -> float {
        if (e.wd != 0.f) g += e.wd * p;
        m = m + (1.f - beta1) * (g - m);                                   // exp_avg.lerp_(grad, 1 - beta1)
        v = beta2 * v + (1.f - beta2) * g * g;                             // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        return p - step * (m / denom);
    };
    if (opt_vec_ok(e, lo)) {                // 16-byte lanes: a wave instruction moves 1 KB instead of 256 B
        for (long long i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
            const float4 g4 = *reinterpret_cast<const float4 *>(e.g + i), p4 = *reinterpret_cast<const float4 *>(e.p + i);
            float4 m4 = *reinterpret_cast<const float4 *>(e.s0 + i), v4 = *reinterpret_cast<const float4 *>(e.s1 + i), o;
            o.x = upd(g4.x, p4.x, m4.x, v4.x); o.y = upd(g4.y, p4.y, m4.y, v4.y);
            o.z = upd(g4.z, p4.z, m4.z, v4.z); o.w = upd(g4.w, p4.w, m4.w, v4.w);
            *reinterpret_cast<float4 *>(e.s0 + i) = m4; *reinterpret_cast<float4 *>(e.s1 + i) = v4;
            *reinterpret_cast<float4 *>(e.p + i) = o;
        }
        return;
    }
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float m = e.s0[i], v = e.s1[i];
        e.p[i] = upd(e.g[i], e.p[i], m, v);
        e.s0[i] = m; e.s1[i] = v;
    }
}

__global__ __launch_bounds__(256) void lvt_rmsprop_kernel(const OptArgs a, float alpha, float eps, float momentum) {
    const int t = find_tensor(a, blockIdx.x);
    const LvtOptEntry e = a.e[t];
    const long long lo = (long long)(blockIdx.x - a.chunk_prefix[t]) * OPT_CHUNK;
    const long long hi = min(e.n, lo + OPT_CHUNK);
    auto upd = [&](float g, float p, float &sq, float &buf) -> float {
        if (e.wd != 0.f) g += e.wd * p;
        sq = alpha * sq + (1.f - alpha) * g * g;
        const float avg = sqrtf(sq) + eps;
        if (momentum > 0.f) { buf = momentum * buf + g / avg; return p - e.lr * buf; }
        return p - e.lr * (g / avg);
    };
    const bool mom = momentum > 0.f;
    if (opt_vec_ok(e, lo)) {
        for (long long i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
            const float4 g4 = *reinterpret_cast<const float4 *>(e.g + i), p4 = *reinterpret_cast<const float4 *>(e.p + i);
            float4 s4 = *reinterpret_cast<const float4 *>(e.s0 + i), o;
            float4 b4 = mom ? *reinterpret_cast<const float4 *>(e.s1 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x = upd(g4.x, p4.x, s4.x, b4.x); o.y = upd(g4.y, p4.y, s4.y, b4.y);
            o.z = upd(g4.z, p4.z, s4.z, b4.z); o.w = upd(g4.w, p4.w, s4.w, b4.w);
            *reinterpret_cast<float4 *>(e.s0 + i) = s4;
            if (mom) *reinterpret_cast<float4 *>(e.s1 + i) = b4;
            *reinterpret_cast<float4 *>(e.p + i) = o;
        }
        return;
    }
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float sq = e.s0[i], buf = mom ? e.s1[i] : 0.f;
        e.p[i] = upd(e.g[i], e.p[i], sq, buf);
        e.s0[i] = sq;
        if (mom) e.s1[i] = buf;
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
