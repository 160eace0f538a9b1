// Kernels for the post-LN BERT-style encoder of the discriminative baselines
// (ref:SeqRec/modules/layers/transformer.py:12-183: MultiHeadAttention, FeedForward, TransformerEncoderLayer).
// These models are small (hidden 64, 2 heads, 50 positions in the shipped configs), so the pieces are built for
// generality and HBM-coalesced access rather than MFMA: the three projections and the two FFN matmuls go through
// gamer_gemm_f32; what is here is everything around them.
//   bias + activation        y = act(x + b)            (nn.Linear bias, FeedForward.get_hidden_act, :100-109)
//   LayerNorm(x + residual)  nn.LayerNorm with affine  (:38, :80, :97)
//   dense attention          softmax(q k^T * scale + additive mask), dropout on the probabilities (:47-78), any
//                            head size <= 64, S <= 128, mask of any broadcastable shape [B|1, H|1, S|1, S]
#include "common.h"

namespace gamer {

constexpr int MOD_THREADS = 256;
constexpr int LN_MAXC = 16;                 // H <= 64 * 16

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SWISH = 3, ACT_TANH = 4, ACT_SIGMOID = 5, ACT_ELU = 6 };

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));          // F.gelu (exact form)
        case ACT_SWISH: return x / (1.f + __expf(-x));
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_GELU: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        case ACT_SWISH: { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
        case ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
        case ACT_SIGMOID: { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f - s); }
        case ACT_ELU: return x > 0.f ? 1.f : __expf(x);
        default: return 1.f;
    }
}

// x <- x + bias (kept for the backward), y = act(x + bias); y == nullptr: bias only
__global__ void __launch_bounds__(MOD_THREADS)
bias_act_fwd_kernel(float* __restrict__ x, const float* __restrict__ bias, int64_t total, int N, int act,
                    float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * MOD_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * MOD_THREADS) {
        const float v = x[i] + bias[i % N];
        x[i] = v;
        if (y) y[i] = act_fwd(v, act);
    }
}

// dx = dy * act'(pre); per-workgroup column sums of dx (bias gradient), reduced by gamer_colsum_reduce.
// Threads = (row lane, column): cw = min(N, 256) adjacent columns per row lane (coalesced rows), 256 / cw row lanes
// walk the workgroup's rows in parallel and are summed through LDS at the end.
__global__ void __launch_bounds__(MOD_THREADS)
bias_act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy, int T, int N, int act,
                    float* __restrict__ dx, float* __restrict__ db_partial) {
    __shared__ float red[MOD_THREADS];
    const int cw = N < MOD_THREADS ? N : MOD_THREADS;
    const int rl = MOD_THREADS / cw;
    const int c0 = threadIdx.x % cw, rlane = threadIdx.x / cw;
    const int rows_per_block = (T + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(T, r0 + rows_per_block);
    for (int cb = 0; cb < N; cb += cw) {
        const int c = cb + c0;
        float acc = 0.f;
        if (rlane < rl && c < N) {
            for (int r = r0 + rlane; r < r1; r += rl) {
                const int64_t i = (int64_t)r * N + c;
                const float g = dy[i] * (act == ACT_NONE ? 1.f : act_bwd(pre[i], act));
                dx[i] = g;
                acc += g;
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rlane == 0 && c < N) {
            float sum = 0.f;
            for (int k = 0; k < rl; ++k) sum += red[k * cw + c0];
            db_partial[(int64_t)blockIdx.x * N + c] = sum;
        }
        __syncthreads();
    }
}

// One wave per row: v = x (+ res); y = (v - mean) * rstd * w + b.  v is written out for the backward.
__global__ void __launch_bounds__(MOD_THREADS)
layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w,
                     const float* __restrict__ b, int T, int H, float eps, float* __restrict__ v_out,
                     float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * MOD_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * MOD_THREADS) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        float v[LN_MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = 0.f;
            if (c < H) {
                v[i] = x[(int64_t)t * H + c] + (res ? res[(int64_t)t * H + c] : 0.f);
                s += v[i];
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H) { const float d = v[i] - mean; ss += d * d; }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);        // biased variance, as nn.LayerNorm
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H) {
                if (v_out) v_out[(int64_t)t * H + c] = v[i];
                y[(int64_t)t * H + c] = (v[i] - mean) * rstd * w[c] + b[c];
            }
        }
        if (lane == 0) { mean_out[t] = mean; rstd_out[t] = rstd; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w; per-workgroup partial sums of dw = dy * xhat, db = dy
__global__ void __launch_bounds__(MOD_THREADS)
layernorm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ w, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ dy, int T, int H,
                     float* __restrict__ dx, float* __restrict__ dw_partial, float* __restrict__ db_partial) {
    __shared__ float red[2][MOD_THREADS / 64][64 * LN_MAXC];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = (blockIdx.x * MOD_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * MOD_THREADS) >> 6;
    float dwacc[LN_MAXC], dbacc[LN_MAXC], wv[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        dwacc[i] = 0.f; dbacc[i] = 0.f;
        wv[i] = c < H ? w[c] : 0.f;
    }
    for (int t = wave; t < T; t += nwaves) {
        const float mu = mean[t], rs = rstd[t];
        float xh[LN_MAXC], g[LN_MAXC];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            xh[i] = 0.f; g[i] = 0.f;
            if (c < H) {
                const float d = dy[(int64_t)t * H + c];
                xh[i] = (v[(int64_t)t * H + c] - mu) * rs;
                dwacc[i] += d * xh[i];
                dbacc[i] += d;
                g[i] = d * wv[i];
                sg += g[i];
                sgx += g[i] * xh[i];
            }
        }
        sg = wave_sum(sg) / (float)H;
        sgx = wave_sum(sgx) / (float)H;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H) dx[(int64_t)t * H + c] = rs * (g[i] - sg - xh[i] * sgx);
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) { red[0][wib][lane + 64 * i] = dwacc[i]; red[1][wib][lane + 64 * i] = dbacc[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += MOD_THREADS) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int k = 0; k < MOD_THREADS / 64; ++k) { a += red[0][k][c]; bsum += red[1][k][c]; }
        dw_partial[(int64_t)blockIdx.x * H + c] = a;
        db_partial[(int64_t)blockIdx.x * H + c] = bsum;
    }
}

// ---- dense attention: one workgroup per (batch, head), probabilities in LDS -------------------------------------
struct DenseAttnArgs {
    const float *q, *k, *v;      // [B*S, ld*]: head h at column h*dh
    int ldq, ldk, ldv;
    int stage;                   // 1: the (batch, head) operand tiles fit in LDS next to the probabilities
    const float* mask;           // additive, element (b,h,i,j) at mask[b*mb + h*mh + i*mi + j*mj]; nullptr: none
    int64_t mb, mh, mi, mj;
    int B, S, H, dh;
    float scale, p_drop;
    uint64_t seed;
};

// [S][dh] operand tile of one (batch, head): either the rows in global memory or a copy in LDS (row stride dh + 1,
// conflict-free for the column walks below).  The same generic pointer serves both.
struct OpTile {
    const float* p;
    int ld;
    __device__ __forceinline__ float at(int r, int d) const { return p[(int64_t)r * ld + d]; }
};
__device__ __forceinline__ OpTile op_tile(const float* g, int ld, int b, int h, int S, int dh, bool stage, float*& lds_top) {
    const float* base = g + (int64_t)b * S * ld + h * dh;
    if (!stage) return OpTile{base, ld};
    float* dst = lds_top;
    lds_top += S * (dh + 1);
    for (int e = threadIdx.x; e < S * dh; e += MOD_THREADS) dst[(e / dh) * (dh + 1) + e % dh] = base[(int64_t)(e / dh) * ld + e % dh];
    return OpTile{dst, dh + 1};
}

__device__ __forceinline__ float dense_score(const DenseAttnArgs& a, const OpTile& Q, const OpTile& K, int b, int h, int i, int j) {
    float s = 0.f;
    for (int d = 0; d < a.dh; ++d) s += Q.at(i, d) * K.at(j, d);
    s *= a.scale;
    if (a.mask) s += a.mask[b * a.mb + h * a.mh + i * a.mi + j * a.mj];
    return s;
}

__global__ void __launch_bounds__(MOD_THREADS)
attn_dense_fwd_kernel(const DenseAttnArgs a, float* __restrict__ o, int ldo, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* P = lds;                                                      // [S][S]
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int S = a.S, dh = a.dh;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const DropoutRng rng(a.p_drop, a.seed);
    float* top = lds + S * S;
    const OpTile Q = op_tile(a.q, a.ldq, b, h, S, dh, a.stage, top);
    const OpTile K = op_tile(a.k, a.ldk, b, h, S, dh, a.stage, top);
    const OpTile V = op_tile(a.v, a.ldv, b, h, S, dh, a.stage, top);
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += MOD_THREADS) P[e] = dense_score(a, Q, K, b, h, e / S, e % S);
    __syncthreads();
    for (int i = wib; i < S; i += MOD_THREADS / 64) {
        float m = -INFINITY;
        for (int j = lane; j < S; j += 64) m = fmaxf(m, P[i * S + j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < S; j += 64) sum += __expf(P[i * S + j] - m);
        sum = wave_sum(sum);
        const float l = m + __logf(sum);
        for (int j = lane; j < S; j += 64) {
            const float p = __expf(P[i * S + j] - l);
            P[i * S + j] = p * rng.mult((((uint64_t)b * a.H + h) * S + i) * S + j);
        }
        if (lane == 0) lse[((int64_t)b * a.H + h) * S + i] = l;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * dh; e += MOD_THREADS) {
        const int i = e / dh, d = e % dh;
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += P[i * S + j] * V.at(j, d);
        o[((int64_t)b * S + i) * ldo + h * dh + d] = acc;
    }
}

__global__ void __launch_bounds__(MOD_THREADS)
attn_dense_bwd_kernel(const DenseAttnArgs a, const float* __restrict__ o, const float* __restrict__ d_o, int ldo,
                      const float* __restrict__ lse, float* __restrict__ dq, int lddq, float* __restrict__ dk,
                      int lddk, float* __restrict__ dv, int lddv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S, dh = a.dh;
    float* P = lds;                   // [S][S] probabilities
    float* D = lds + S * S;           // [S][S] dropped probabilities, then dS
    float* delta = D + S * S;         // [S]
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const DropoutRng rng(a.p_drop, a.seed);
    float* top = delta + S;
    const OpTile Q = op_tile(a.q, a.ldq, b, h, S, dh, a.stage, top);
    const OpTile K = op_tile(a.k, a.ldk, b, h, S, dh, a.stage, top);
    const OpTile V = op_tile(a.v, a.ldv, b, h, S, dh, a.stage, top);
    const OpTile G = op_tile(d_o, ldo, b, h, S, dh, a.stage, top);
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += MOD_THREADS) {
        const int i = e / S, j = e % S;
        const float p = __expf(dense_score(a, Q, K, b, h, i, j) - lse[((int64_t)b * a.H + h) * S + i]);
        P[e] = p;
        D[e] = p * rng.mult((((uint64_t)b * a.H + h) * S + i) * S + j);
    }
    for (int i = threadIdx.x; i < S; i += MOD_THREADS) {
        const float* oi = o + ((int64_t)b * S + i) * ldo + h * dh;
        float acc = 0.f;
        for (int d = 0; d < dh; ++d) acc += oi[d] * G.at(i, d);
        delta[i] = acc;
    }
    __syncthreads();
    // dV[j][d] = sum_i Pd[i][j] dO[i][d]
    for (int e = threadIdx.x; e < S * dh; e += MOD_THREADS) {
        const int j = e / dh, d = e % dh;
        float acc = 0.f;
        for (int i = 0; i < S; ++i) acc += D[i * S + j] * G.at(i, d);
        dv[((int64_t)b * S + j) * lddv + h * dh + d] = acc;
    }
    __syncthreads();
    // dS[i][j] = Pd[i][j] * (dO_i . V_j) - P[i][j] * delta_i
    for (int e = threadIdx.x; e < S * S; e += MOD_THREADS) {
        const int i = e / S, j = e % S;
        float dp = 0.f;
        for (int d = 0; d < dh; ++d) dp += G.at(i, d) * V.at(j, d);
        D[e] = D[e] * dp - P[e] * delta[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * dh; e += MOD_THREADS) {
        const int r = e / dh, d = e % dh;
        float aq = 0.f, ak = 0.f;
        for (int t = 0; t < S; ++t) {
            aq += D[r * S + t] * K.at(t, d);       // dQ[r] += dS[r][t] K[t]
            ak += D[t * S + r] * Q.at(t, d);       // dK[r] += dS[t][r] Q[t]
        }
        dq[((int64_t)b * S + r) * lddq + h * dh + d] = aq * a.scale;
        dk[((int64_t)b * S + r) * lddk + h * dh + d] = ak * a.scale;
    }
}

}  // namespace gamer

using namespace gamer;
#define ST(s) ((hipStream_t)(s))

constexpr size_t DENSE_LDS_MAX = 150 * 1024;      // of the 160 KB per CU

static int mod_grid(int64_t n) {
    int64_t blocks = (n + MOD_THREADS - 1) / MOD_THREADS;
    return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

extern "C" int gamer_bias_act_fwd(float* x, const float* bias, int T, int N, int act, float* y, void* stream) {
    GAMER_CHECK_ARG(x && bias && T > 0 && N > 0, "gamer_bias_act_fwd: bad arguments T=%d N=%d", T, N);
    GAMER_CHECK_ARG(act >= ACT_NONE && act <= ACT_ELU, "gamer_bias_act_fwd: unknown activation %d", act);
    GAMER_CHECK_ARG(y || act == ACT_NONE, "gamer_bias_act_fwd: an activation needs an output buffer");
    hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(mod_grid((int64_t)T * N)), dim3(MOD_THREADS), 0, ST(stream), x, bias,
                       (int64_t)T * N, N, act, y);
    GAMER_CHECK_LAUNCH("gamer_bias_act_fwd");
    return 0;
}

extern "C" int gamer_bias_act_bwd(const float* pre, const float* dy, int T, int N, int act, float* dx,
                                  float* db_partial, int n_partial, void* stream) {
    GAMER_CHECK_ARG(dy && dx && db_partial && T > 0 && N > 0 && n_partial > 0 && (pre || act == ACT_NONE),
                    "gamer_bias_act_bwd: bad arguments T=%d N=%d n_partial=%d", T, N, n_partial);
    GAMER_CHECK_ARG(act >= ACT_NONE && act <= ACT_ELU, "gamer_bias_act_bwd: unknown activation %d", act);
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(n_partial), dim3(MOD_THREADS), 0, ST(stream), pre, dy, T, N, act, dx,
                       db_partial);
    GAMER_CHECK_LAUNCH("gamer_bias_act_bwd");
    return 0;
}

extern "C" int gamer_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, int T, int H,
                                   float eps, float* v_out, float* y, float* mean, float* rstd, void* stream) {
    GAMER_CHECK_ARG(x && w && b && y && mean && rstd, "gamer_layernorm_fwd: null pointer");
    GAMER_CHECK_ARG(T > 0 && H > 0 && H <= 64 * LN_MAXC, "gamer_layernorm_fwd: bad shape T=%d H=%d (H <= %d)", T, H, 64 * LN_MAXC);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(mod_grid((int64_t)T * 64)), dim3(MOD_THREADS), 0, ST(stream), x, res, w,
                       b, T, H, eps, v_out, y, mean, rstd);
    GAMER_CHECK_LAUNCH("gamer_layernorm_fwd");
    return 0;
}

extern "C" int gamer_layernorm_bwd(const float* v, const float* w, const float* mean, const float* rstd,
                                   const float* dy, int T, int H, float* dx, float* dw_partial, float* db_partial,
                                   int n_partial, void* stream) {
    GAMER_CHECK_ARG(v && w && mean && rstd && dy && dx && dw_partial && db_partial, "gamer_layernorm_bwd: null pointer");
    GAMER_CHECK_ARG(T > 0 && H > 0 && H <= 64 * LN_MAXC && n_partial > 0,
                    "gamer_layernorm_bwd: bad shape T=%d H=%d n_partial=%d", T, H, n_partial);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(n_partial), dim3(MOD_THREADS), 0, ST(stream), v, w, mean, rstd, dy, T,
                       H, dx, dw_partial, db_partial);
    GAMER_CHECK_LAUNCH("gamer_layernorm_bwd");
    return 0;
}

static int dense_args(const char* name, DenseAttnArgs& a, const float* q, int ldq, const float* k, int ldk,
                      const float* v, int ldv, const float* mask, const int64_t* mask_strides, int B, int S, int H,
                      int dh, float scale, float p_drop, uint64_t seed) {
    GAMER_CHECK_ARG(q && k && v, "%s: null pointer", name);
    GAMER_CHECK_ARG(B > 0 && S > 0 && S <= 128 && H > 0 && dh > 0 && dh <= 64,
                    "%s: bad shape B=%d S=%d H=%d head_dim=%d (S <= 128, head_dim <= 64)", name, B, S, H, dh);
    GAMER_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh, "%s: bad leading dims", name);
    GAMER_CHECK_ARG(!mask || mask_strides, "%s: a mask needs its four strides", name);
    GAMER_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p_drop=%f", name, p_drop);
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.stage = 0;
    a.mask = mask;
    a.mb = mask ? mask_strides[0] : 0; a.mh = mask ? mask_strides[1] : 0;
    a.mi = mask ? mask_strides[2] : 0; a.mj = mask ? mask_strides[3] : 0;
    a.B = B; a.S = S; a.H = H; a.dh = dh; a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    return 0;
}

extern "C" int gamer_attn_dense_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* mask, const int64_t* mask_strides, int B, int S, int H, int head_dim,
                                    float scale, float p_drop, uint64_t seed, float* o, int ldo, float* lse,
                                    void* stream) {
    DenseAttnArgs a;
    int rc = dense_args("gamer_attn_dense_fwd", a, q, ldq, k, ldk, v, ldv, mask, mask_strides, B, S, H, head_dim, scale,
                        p_drop, seed);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && lse && ldo >= H * head_dim, "gamer_attn_dense_fwd: bad output");
    size_t shmem = (size_t)S * S * sizeof(float);
    const size_t staged = shmem + (size_t)3 * S * (head_dim + 1) * sizeof(float);
    a.stage = staged <= DENSE_LDS_MAX ? 1 : 0;
    if (a.stage) shmem = staged;
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dense_fwd_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)DENSE_LDS_MAX);
        if (e != hipSuccess) { set_error("gamer_attn_dense_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    hipLaunchKernelGGL(attn_dense_fwd_kernel, dim3(B * H), dim3(MOD_THREADS), shmem, ST(stream), a, o, ldo, lse);
    GAMER_CHECK_LAUNCH("gamer_attn_dense_fwd");
    return 0;
}

extern "C" int gamer_attn_dense_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* mask, const int64_t* mask_strides, int B, int S, int H, int head_dim,
                                    float scale, float p_drop, uint64_t seed, const float* o, const float* d_o, int ldo,
                                    const float* lse, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                    void* stream) {
    DenseAttnArgs a;
    int rc = dense_args("gamer_attn_dense_bwd", a, q, ldq, k, ldk, v, ldv, mask, mask_strides, B, S, H, head_dim, scale,
                        p_drop, seed);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && d_o && lse && dq && dk && dv, "gamer_attn_dense_bwd: null pointer");
    size_t shmem = ((size_t)2 * S * S + S) * sizeof(float);
    const size_t staged = shmem + (size_t)4 * S * (head_dim + 1) * sizeof(float);
    a.stage = staged <= DENSE_LDS_MAX ? 1 : 0;
    if (a.stage) shmem = staged;
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dense_bwd_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)DENSE_LDS_MAX);
        if (e != hipSuccess) { set_error("gamer_attn_dense_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    hipLaunchKernelGGL(attn_dense_bwd_kernel, dim3(B * H), dim3(MOD_THREADS), shmem, ST(stream), a, o, d_o, ldo, lse, dq,
                       lddq, dk, lddk, dv, lddv);
    GAMER_CHECK_LAUNCH("gamer_attn_dense_bwd");
    return 0;
}
