// Fused HF-Trainer update on flat fp32 buffers: global grad-norm (two deterministic passes),
// clip_grad_norm_(max_norm) and AdamW in one sweep (7 x 4 B per parameter of HBM traffic).
// Reference: transformers/trainer.py (clip_grad_norm_ + optimizer.step), torch.optim.AdamW;
// flags from ref:SeqRec/tasks/train_SMB_decoder.py:396-428.
#include "common.h"

namespace gamer {

constexpr int OPT_THREADS = 256;

__global__ void __launch_bounds__(OPT_THREADS)
sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float sh[OPT_THREADS / 64];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t per4 = (per + 3) & ~(int64_t)3;                 // keep chunks 16-byte aligned
    const int64_t beg = (int64_t)blockIdx.x * per4;
    const int64_t end = beg + per4 < n ? beg + per4 : n;
    float acc = 0.f;
    if (beg < n) {
        const int64_t nvec = (end - beg) >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g + beg);
        for (int64_t i = threadIdx.x; i < nvec; i += OPT_THREADS) {
            const float4 v = g4[i];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = beg + (nvec << 2) + threadIdx.x; i < end; i += OPT_THREADS) acc += g[i] * g[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) s += sh[w];
        partial[blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(OPT_THREADS)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             int64_t n, int64_t n_decay, float lr, float beta1, float beta2, float eps, float weight_decay,
             float inv_bc1, float inv_sqrt_bc2, float max_norm, float grad_scale,
             const float* __restrict__ partial, int n_partial, float* __restrict__ norm_out) {
    __shared__ float sh[OPT_THREADS / 64];
    __shared__ float coef_s;
    // every workgroup folds the partial sums in the same fixed order
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += OPT_THREADS) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) s += sh[w];
        const float norm = sqrtf(s) * grad_scale;                   // norm of the scaled gradient
        float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        if (c > 1.f) c = 1.f;
        coef_s = c * grad_scale;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t idx = (i << 2) + e;
            const float gg = gp[e] * coef;
            const float wd = idx < n_decay ? weight_decay : 0.f;
            float x = pp[e] * (1.f - lr * wd);
            const float m1 = beta1 * mp[e] + (1.f - beta1) * gg;
            const float v1 = beta2 * vp[e] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(v1) * inv_sqrt_bc2 + eps;
            x -= lr * inv_bc1 * (m1 / denom);
            pp[e] = x; mp[e] = m1; vp[e] = v1;
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n % 4)
    if (blockIdx.x == 0) {
        for (int64_t idx = (n4 << 2) + threadIdx.x; idx < n; idx += OPT_THREADS) {
            const float gg = g[idx] * coef;
            const float wd = idx < n_decay ? weight_decay : 0.f;
            float x = p[idx] * (1.f - lr * wd);
            const float m1 = beta1 * m[idx] + (1.f - beta1) * gg;
            const float v1 = beta2 * v[idx] + (1.f - beta2) * gg * gg;
            x -= lr * inv_bc1 * (m1 / (sqrtf(v1) * inv_sqrt_bc2 + eps));
            p[idx] = x; m[idx] = m1; v[idx] = v1;
        }
    }
}

}  // namespace gamer

using namespace gamer;

extern "C" int gamer_sumsq(const float* g, int64_t n, float* partial, int n_partial, void* stream) {
    GAMER_CHECK_ARG(g && partial && n > 0 && n_partial > 0 && n_partial <= 4096, "gamer_sumsq: bad arguments n=%lld n_partial=%d", (long long)n, n_partial);
    GAMER_CHECK_ARG(aligned16(g), "gamer_sumsq: g must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_partial), dim3(OPT_THREADS), 0, (hipStream_t)stream, g, n, partial);
    GAMER_CHECK_LAUNCH("gamer_sumsq");
    return 0;
}

extern "C" int gamer_adamw(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                           float grad_scale, const float* partial, int n_partial, float* norm_out, void* stream) {
    GAMER_CHECK_ARG(p && g && m && v && partial && n > 0 && n_partial > 0 && step >= 1, "gamer_adamw: bad arguments");
    GAMER_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "gamer_adamw: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = ((n >> 2) + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, g, m, v, n, n_decay,
                       lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), max_norm,
                       grad_scale, partial, n_partial, norm_out);
    GAMER_CHECK_LAUNCH("gamer_adamw");
    return 0;
}
