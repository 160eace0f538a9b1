// HBM-bound pieces of the Qwen3Multi decoder layer: embedding, RMSNorm, behaviour tables,
// q/k norm + RoPE, residual/dropout, SwiGLU, output gate, temperature cross-entropy.
// Every kernel moves 16 B per lane and reduces with wave shuffles (wave = 64).
#include "common.h"

namespace gamer {

constexpr int EW_THREADS = 256;
constexpr int EW_WAVES = EW_THREADS / WAVE;

#ifndef EW_MAX_BLOCKS_W
#define EW_MAX_BLOCKS_W 8192
#endif
#ifndef EW_MAX_BLOCKS_T
#define EW_MAX_BLOCKS_T 16384
#endif
static inline int grid_for_waves(int64_t n_waves, int max_blocks = EW_MAX_BLOCKS_W) {
    int64_t blocks = (n_waves + EW_WAVES - 1) / EW_WAVES;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}
static inline int grid_for_threads(int64_t n, int max_blocks = EW_MAX_BLOCKS_T) {
    int64_t blocks = (n + EW_THREADS - 1) / EW_THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}

// ---------------------------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EW_THREADS)
embedding_fwd_kernel(const int64_t* __restrict__ ids, const float4* __restrict__ W, int V, int T, int H4,
                     float4* __restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * EW_THREADS) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        const int64_t id = ids[t];
        const bool ok = id >= 0 && id < V;
        for (int c = lane; c < H4; c += 64) {
            float4 v = ok ? W[id * H4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            x[(int64_t)t * H4 + c] = v;
        }
    }
}

// Scatter-add of d(embedding output) into the table gradient.  A few rows are hot (the behaviour tokens: every fifth
// token of the batch lands on one of NB rows) and same-address float atomics serialise in L2: 2.06 ms at T = 517k with
// one global atomic per element.  Each workgroup therefore takes a contiguous chunk of tokens, caches the first
// EMB_CACHE distinct ids of its chunk in LDS (the hot ids are among them: a chunk starts with a behaviour token),
// accumulates those rows with LDS atomics and flushes them once; every other row goes straight to global atomics.
constexpr int EMB_CACHE = 8;
constexpr int EMB_CHUNK = 1024;          // tokens per workgroup at large T; smaller chunks keep >= ~512 workgroups at small T
__global__ void __launch_bounds__(EW_THREADS)
embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx, int V, int T, int H,
                     int pad_id, float* __restrict__ dW, const int chunk) {
    extern __shared__ __attribute__((aligned(16))) float emb_cache[];       // [EMB_CACHE][H]
    __shared__ int64_t cache_id[EMB_CACHE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = blockIdx.x * chunk, t1 = min(T, t0 + chunk);
    for (int e = threadIdx.x; e < EMB_CACHE * H; e += EW_THREADS) emb_cache[e] = 0.f;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int i = 0; i < EMB_CACHE; ++i) cache_id[i] = -1;
        for (int t = t0; t < t1 && t < t0 + 64 && n < EMB_CACHE; ++t) {
            const int64_t id = ids[t];
            if (id == pad_id || id < 0 || id >= V) continue;
            bool seen = false;
            for (int i = 0; i < n; ++i) seen |= cache_id[i] == id;
            if (!seen) cache_id[n++] = id;
        }
    }
    __syncthreads();
    for (int t = t0 + w; t < t1; t += EW_THREADS / 64) {
        const int64_t id = ids[t];
        if (id == pad_id || id < 0 || id >= V) continue;   // wave-uniform
        int slot = -1;
#pragma unroll
        for (int i = 0; i < EMB_CACHE; ++i) slot = cache_id[i] == id ? i : slot;
        if (slot >= 0) {
            for (int c = lane; c < H; c += 64) atomicAdd(&emb_cache[slot * H + c], dx[(int64_t)t * H + c]);
        } else {
            for (int c = lane; c < H; c += 64) atomicAdd(&dW[id * H + c], dx[(int64_t)t * H + c]);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < EMB_CACHE * H; e += EW_THREADS) {
        const int64_t id = cache_id[e / H];
        const float v = emb_cache[e];
        if (id >= 0 && v != 0.f) atomicAdd(&dW[id * H + e % H], v);
    }
}

// ---- the same gradient without float atomics: a stable counting sort of the tokens by id, then sums in sorted order ----
// (1) per-chunk histograms of the ids (integer LDS atomics: the counts do not depend on their order), (2) one workgroup turns
// them into every chunk's write cursors (rows ascending, chunks ascending inside a row: a STABLE order) and cuts every row's
// token list into pieces of EMB_PIECE tokens, (3) every chunk writes its token indices to their sorted positions in token
// order, (4) one workgroup per piece adds its <= EMB_PIECE rows of dx in sorted order, (5) one workgroup per table row adds its
// pieces in order into dW.  Same bits on every run; 517k tokens x 256: five launches, ~0.3 ms (the atomic form: 0.75 ms).
constexpr int EMB_SORT_CHUNK = 1024;
constexpr int EMB_PIECE = 256;
constexpr int EMB_MAX_V = 8191;          // (V + 1 counters in LDS)
struct EmbSortWs { int* hist; int* row_start; int* row_count; int* piece_start; int* perm; float* partial; };

__global__ void __launch_bounds__(EW_THREADS)
emb_hist_kernel(const int64_t* __restrict__ ids, int V, int T, int pad_id, int* __restrict__ hist) {
    extern __shared__ int emb_cnt[];                             // [V + 1]: bucket V = padding / out-of-range ids
    for (int v = threadIdx.x; v <= V; v += EW_THREADS) emb_cnt[v] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * EMB_SORT_CHUNK, t1 = min(T, t0 + EMB_SORT_CHUNK);
    for (int t = t0 + threadIdx.x; t < t1; t += EW_THREADS) {
        const int64_t id = ids[t];
        atomicAdd(&emb_cnt[(id == pad_id || id < 0 || id >= V) ? V : (int)id], 1);
    }
    __syncthreads();
    for (int v = threadIdx.x; v <= V; v += EW_THREADS) hist[(int64_t)blockIdx.x * (V + 1) + v] = emb_cnt[v];
}

// one wave per table row v: hist[c][v] -> exclusive prefix over the chunks c, row_count[v] = the row's total
__global__ void __launch_bounds__(64)
emb_chunk_scan_kernel(int V, int n_chunks, int* __restrict__ hist, int* __restrict__ row_count) {
    const int v = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const int h = c < n_chunks ? hist[(int64_t)c * (V + 1) + v] : 0;
        int incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (c < n_chunks) hist[(int64_t)c * (V + 1) + v] = carry + incl - h;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) row_count[v] = carry;
}
// one workgroup: row_start[v] = tokens in front of row v, piece_start[v] = pieces in front of it (v = 0 .. V; [V + 1] = the end)
__global__ void __launch_bounds__(1024)
emb_row_scan_kernel(int V, const int* __restrict__ row_count, int* __restrict__ row_start, int* __restrict__ piece_start) {
    __shared__ int s_tok[1024], s_pc[1024];
    const int tid = threadIdx.x;
    const int per = (V + 1 + 1023) / 1024;                        // rows per thread (<= 8)
    int tok = 0, pc = 0;
    for (int i = 0; i < per; ++i) {
        const int v = tid * per + i;
        if (v <= V) {
            const int n = row_count[v];
            tok += n;
            if (v < V) pc += (n + EMB_PIECE - 1) / EMB_PIECE;
        }
    }
    s_tok[tid] = tok; s_pc[tid] = pc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                          // inclusive scans of the per-thread sums
        const int a = tid >= o ? s_tok[tid - o] : 0, b = tid >= o ? s_pc[tid - o] : 0;
        __syncthreads();
        s_tok[tid] += a; s_pc[tid] += b;
        __syncthreads();
    }
    int t_run = s_tok[tid] - tok, p_run = s_pc[tid] - pc;
    for (int i = 0; i < per; ++i) {
        const int v = tid * per + i;
        if (v <= V) {
            row_start[v] = t_run;
            piece_start[v] = p_run;
            const int n = row_count[v];
            t_run += n;
            if (v < V) p_run += (n + EMB_PIECE - 1) / EMB_PIECE;
        }
    }
    if (tid == 1023) piece_start[V + 1] = s_pc[1023];
}

__global__ void __launch_bounds__(EW_THREADS)
emb_scatter_kernel(const int64_t* __restrict__ ids, int V, int T, int pad_id, const int* __restrict__ hist,
                   const int* __restrict__ row_start, int* __restrict__ perm) {
    extern __shared__ int emb_cur[];                             // [V + 1] write cursors of this chunk, then its [1024] buckets
    int* bucket = emb_cur + V + 1;
    const int t0 = blockIdx.x * EMB_SORT_CHUNK, t1 = min(T, t0 + EMB_SORT_CHUNK);
    for (int v = threadIdx.x; v <= V; v += EW_THREADS) emb_cur[v] = row_start[v] + hist[(int64_t)blockIdx.x * (V + 1) + v];
    for (int t = t0 + threadIdx.x; t < t1; t += EW_THREADS) {
        const int64_t id = ids[t];
        bucket[t - t0] = (id == pad_id || id < 0 || id >= V) ? -1 : (int)id;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // token order inside the chunk: ONE wave walks it 64 tokens at a time.  A lane's position = its row's cursor + the number of
        // earlier lanes of the batch with the same id; the last lane of every id moves the cursor on.  (A single thread walking
        // the 1024 tokens took 120-150 us per launch - every step a dependent LDS round trip.)
        const int lane = threadIdx.x;
        for (int tb = t0; tb < t1; tb += 64) {
            const int t = tb + lane;
            const int b = t < t1 ? bucket[t - t0] : -1;
            int rank = 0, cnt = 0;
            for (int j = 0; j < 64; ++j) {
                const int bj = __shfl(b, j, 64);
                const bool same = bj == b;
                rank += (same && j < lane) ? 1 : 0;
                cnt += same ? 1 : 0;
            }
            if (b >= 0) {
                const int pos = emb_cur[b] + rank;               // (every lane reads before any lane of this batch writes)
                perm[pos] = t;
                __builtin_amdgcn_wave_barrier();
                if (rank == cnt - 1) emb_cur[b] = pos + 1;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ void __launch_bounds__(EW_THREADS)
emb_piece_sum_kernel(const float4* __restrict__ dx, int V, int H4, const int* __restrict__ row_start, const int* __restrict__ row_count,
                     const int* __restrict__ piece_start, const int* __restrict__ perm, float4* __restrict__ partial) {
    __shared__ float4 red[3][256];
    const int p = blockIdx.x;
    if (p >= piece_start[V]) return;
    int lo = 0, hi = V;                                          // the row of piece p: last v with piece_start[v] <= p
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (piece_start[mid] <= p) lo = mid; else hi = mid; }
    const int v = lo, k = p - piece_start[v];
    const int begin = row_start[v] + k * EMB_PIECE;
    const int count = min(EMB_PIECE, row_count[v] - k * EMB_PIECE);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // wave w adds tokens [w * 64, w * 64 + 64) of the piece in order; the four sums are then combined in a fixed order
    for (int c0 = 0; c0 < H4; c0 += 64) {
        const int c = c0 + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < H4) {
            const int j1 = min(count, (w + 1) * (EMB_PIECE / 4));
            int j = w * (EMB_PIECE / 4);
            for (; j + 4 <= j1; j += 4) {
                const int r0 = perm[begin + j], r1 = perm[begin + j + 1], r2 = perm[begin + j + 2], r3 = perm[begin + j + 3];
                const float4 a0 = dx[(int64_t)r0 * H4 + c], a1 = dx[(int64_t)r1 * H4 + c], a2 = dx[(int64_t)r2 * H4 + c],
                             a3 = dx[(int64_t)r3 * H4 + c];
                acc.x = (((acc.x + a0.x) + a1.x) + a2.x) + a3.x; acc.y = (((acc.y + a0.y) + a1.y) + a2.y) + a3.y;
                acc.z = (((acc.z + a0.z) + a1.z) + a2.z) + a3.z; acc.w = (((acc.w + a0.w) + a1.w) + a2.w) + a3.w;
            }
            for (; j < j1; ++j) {
                const float4 a0 = dx[(int64_t)perm[begin + j] * H4 + c];
                acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
            }
        }
        if (w > 0 && c < H4) red[w - 1][lane] = acc;
        __syncthreads();
        if (w == 0 && c < H4) {
            const float4 b = red[0][lane], d = red[1][lane], e = red[2][lane];
            float4 o;
            o.x = (acc.x + b.x) + (d.x + e.x); o.y = (acc.y + b.y) + (d.y + e.y);
            o.z = (acc.z + b.z) + (d.z + e.z); o.w = (acc.w + b.w) + (d.w + e.w);
            partial[(int64_t)p * H4 + c] = o;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64)
emb_row_sum_kernel(int H4, const int* __restrict__ piece_start, const float4* __restrict__ partial, float4* __restrict__ dW) {
    const int v = blockIdx.x;
    const int p0 = piece_start[v], p1 = piece_start[v + 1];
    if (p0 == p1) return;
    for (int c = threadIdx.x; c < H4; c += 64) {
        float4 s = partial[(int64_t)p0 * H4 + c];
        for (int p = p0 + 1; p < p1; ++p) {
            const float4 a = partial[(int64_t)p * H4 + c];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        float4 o = dW[(int64_t)v * H4 + c];
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        dW[(int64_t)v * H4 + c] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (hidden dim H <= 1024, H % 4 == 0); one wave per row, row cached in registers
// ---------------------------------------------------------------------------------------------
// float4 chunks per lane: 1 for H <= 256 (the shipped hidden size: one chunk per lane, 4x fewer registers - the
// 4-chunk form sat at 134 VGPRs = 3 waves per SIMD after the dtype templating, 9.2 -> 12.3 ms per step), 4 for H <= 1024
constexpr int RMS_MAXC_BIG = 4;

template <typename TY, int RMS_MAXC>
__global__ void __launch_bounds__(EW_THREADS)
rmsnorm_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ w, int T, int H4, float inv_h,
                   float eps, const int32_t* __restrict__ dst_rows, TY* __restrict__ y, int ldy, uint32_t* __restrict__ amax_out) {
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * EW_THREADS) >> 6;
    float4 wv[RMS_MAXC];
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        const int c = lane + 64 * i;
        wv[i] = (c < H4) ? w[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = wave; t < T; t += nwaves) {
        float4 v[RMS_MAXC];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < H4) ? x[(int64_t)t * H4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss * inv_h + eps);
        const int64_t r = dst_rows ? dst_rows[t] : t;
        TY* yrow = y + r * ldy;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H4) {
                float4 o;
                o.x = wv[i].x * (v[i].x * rstd);
                o.y = wv[i].y * (v[i].y * rstd);
                o.z = wv[i].z * (v[i].z * rstd);
                o.w = wv[i].w * (v[i].w * rstd);
                st4(yrow + 4 * c, o);
                am = amax_f4(am, o);
            }
        }
    }
    amax_block_commit(am, amax_out, amax_lds);
}

// dx (+)= rstd*(w*dy - xhat*mean(w*dy*xhat)); per-workgroup dw partial sums (deterministic).
template <typename TG, int RMS_MAXC>
__global__ void __launch_bounds__(EW_THREADS)
rmsnorm_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ w, const TG* __restrict__ dy,
                   int lddy, const int32_t* __restrict__ dy_rows, int T, int H4, float inv_h, float eps,
                   int accumulate_dx, float4* __restrict__ dx, float* __restrict__ dw_partial,
                   TG* __restrict__ mask_out, const int32_t* __restrict__ mask_rows, float p_drop, uint64_t seed,
                   uint32_t* __restrict__ amax_out) {
    // mask_out != nullptr: the next consumer of dx is a residual branch with dropout; its input gradient
    // mask(seed) * dx is written here (what gamer_residual_dropout_bwd would compute in a second pass over dx)
    // amax_out: max |mask_out| (the GEMM operand of that branch)
    const DropoutRng rng(mask_out ? p_drop : 0.f, seed);
    __shared__ float4 red[EW_WAVES][64 * RMS_MAXC];
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * EW_THREADS) >> 6;
    float4 wv[RMS_MAXC], dwacc[RMS_MAXC];
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        const int c = lane + 64 * i;
        wv[i] = (c < H4) ? w[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dwacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (Measured and not kept: two rows per iteration with the dx read hoisted in front of the reductions - 0.58 -> 0.39 ms on a
    // 512-workgroup grid, but the engine launches 2048 workgroups (8 waves per SIMD, 5.9 TB/s with the mask output) and there the
    // extra registers cost occupancy: 0.45 -> 0.51 ms per call.)
    for (int t = wave; t < T; t += nwaves) {
        float4 v[RMS_MAXC], g[RMS_MAXC];
        float ss = 0.f;
        const int64_t r = dy_rows ? dy_rows[t] : t;
        const TG* dyrow = dy + r * lddy;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < H4) ? x[(int64_t)t * H4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = (c < H4) ? ld4(dyrow + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss * inv_h + eps);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            // xhat
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;
            dwacc[i].x += g[i].x * v[i].x; dwacc[i].y += g[i].y * v[i].y;
            dwacc[i].z += g[i].z * v[i].z; dwacc[i].w += g[i].w * v[i].w;
            g[i].x *= wv[i].x; g[i].y *= wv[i].y; g[i].z *= wv[i].z; g[i].w *= wv[i].w;
            dot += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
        }
        dot = wave_sum(dot) * inv_h;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H4) {
                float4 o;
                o.x = rstd * (g[i].x - v[i].x * dot);
                o.y = rstd * (g[i].y - v[i].y * dot);
                o.z = rstd * (g[i].z - v[i].z * dot);
                o.w = rstd * (g[i].w - v[i].w * dot);
                if (accumulate_dx) {
                    float4 p = dx[(int64_t)t * H4 + c];
                    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                }
                dx[(int64_t)t * H4 + c] = o;
                if (mask_out) {
                    float m[4];
                    rng.mult4((uint32_t)((int64_t)t * H4 + c), m);
                    const int64_t mr = mask_rows ? mask_rows[t] : t;
                    const float4 mo = make_float4(m[0] * o.x, m[1] * o.y, m[2] * o.z, m[3] * o.w);
                    st4(mask_out + (mr * H4 + c) * 4, mo);
                    am = amax_f4(am, mo);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) red[wib][lane + 64 * i] = dwacc[i];
    __syncthreads();
    if (wib == 0) {
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < H4) {
                float4 s = red[0][c];
                for (int ww = 1; ww < EW_WAVES; ++ww) {
                    float4 o = red[ww][c];
                    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                }
                reinterpret_cast<float4*>(dw_partial + (int64_t)blockIdx.x * H4 * 4)[c] = s;
            }
        }
    }
    amax_block_commit(am, mask_out ? amax_out : nullptr, amax_lds);
}

// out[c] (+)= sum_r partial[r][c]; one workgroup per 32 columns, 32 row groups per workgroup, four independent
// loads in flight per thread, fixed summation order (deterministic).  (With 8 row groups and one load at a time
// the 2048-row partial tables of the norm backward took 82 us per call, 21 calls per step.)
constexpr int COLSUM_RG = 32;
__global__ void __launch_bounds__(32 * COLSUM_RG)
colsum_reduce_kernel(const float* __restrict__ partial, int rows, int cols, int accumulate,
                     float* __restrict__ out) {
    __shared__ float sh[COLSUM_RG][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 3 * COLSUM_RG < rows; r += 4 * COLSUM_RG) {
            s0 += partial[(int64_t)r * cols + c];
            s1 += partial[(int64_t)(r + COLSUM_RG) * cols + c];
            s2 += partial[(int64_t)(r + 2 * COLSUM_RG) * cols + c];
            s3 += partial[(int64_t)(r + 3 * COLSUM_RG) * cols + c];
        }
        for (; r < rows; r += COLSUM_RG) s0 += partial[(int64_t)r * cols + c];
    }
    sh[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < COLSUM_RG; ++g) t += sh[g][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// n tables [rows][cols] `stride` floats apart, each summed into its own output vector outs[i]: the RMSNorm weight gradients of
// a whole backward pass in ONE launch at its end (they belong to the tail bucket of the gradient all-reduce) instead of one
// 7-us launch behind every norm backward - 25 launches per step, 0.6 % of the step at per-GPU batch 128.
__global__ void __launch_bounds__(32 * COLSUM_RG)
colsum_reduce_batched_kernel(const float* __restrict__ partial, int64_t stride, int rows, int cols, int accumulate,
                             float* const* __restrict__ outs) {
    __shared__ float sh[COLSUM_RG][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const float* pt = partial + (int64_t)blockIdx.y * stride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 3 * COLSUM_RG < rows; r += 4 * COLSUM_RG) {
            s0 += pt[(int64_t)r * cols + c];
            s1 += pt[(int64_t)(r + COLSUM_RG) * cols + c];
            s2 += pt[(int64_t)(r + 2 * COLSUM_RG) * cols + c];
            s3 += pt[(int64_t)(r + 3 * COLSUM_RG) * cols + c];
        }
        for (; r < rows; r += COLSUM_RG) s0 += pt[(int64_t)r * cols + c];
    }
    sh[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < COLSUM_RG; ++g) t += sh[g][cl];
        float* out = outs[blockIdx.y];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---------------------------------------------------------------------------------------------
// small-table rows (behaviour embeddings)
// ---------------------------------------------------------------------------------------------
template <typename TY>
__global__ void __launch_bounds__(EW_THREADS)
rowtable_fwd_kernel(const float4* __restrict__ table, const int32_t* __restrict__ idx,
                    const int32_t* __restrict__ dst_rows, int T, int E4, TY* __restrict__ y, int ldy, int col0,
                    uint32_t* __restrict__ amax_out) {
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const int64_t total = (int64_t)T * E4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int t = (int)(i / E4);
        const int c = (int)(i % E4);
        const int64_t r = dst_rows ? dst_rows[t] : t;
        const float4 v = table[(int64_t)idx[t] * E4 + c];
        st4(y + r * ldy + col0 + 4 * c, v);
        am = amax_f4(am, v);
    }
    amax_block_commit(am, amax_out, amax_lds);
}

constexpr int TBL_MAXROWS = 8;   // num_behavior + 1 <= 8

// dtable[idx[t]][c] += dy[src(t)][col0+c]; per-thread predicated accumulators, LDS combine, one
// atomic per workgroup per table element.
// DET (a `partial` workspace was given): no atomics at all - the row groups of a workgroup are added in row-group order through
// LDS, the workgroup's table goes to partial[block] and gamer_colsum_reduce's kernel folds the blocks in block order.
template <typename TG, bool DET>
__global__ void __launch_bounds__(EW_THREADS)
rowtable_bwd_kernel(const TG* __restrict__ dy, int lddy, int col0, const int32_t* __restrict__ idx,
                    const int32_t* __restrict__ dy_rows, int T, int E4, int nrows, float* __restrict__ dtable,
                    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds_tbl[];   // [nrows][E4*4]; DET: [row groups][nrows][E4*4]
    const int E = E4 * 4;
    if (!DET) {
        for (int i = threadIdx.x; i < nrows * E; i += EW_THREADS) lds_tbl[i] = 0.f;
        __syncthreads();
    }
    const int c = threadIdx.x % E4;
    const int rl = threadIdx.x / E4;
    const int rows_per_block = EW_THREADS / E4;
    float4 acc[TBL_MAXROWS];
#pragma unroll
    for (int a = 0; a < TBL_MAXROWS; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < rows_per_block) {
        for (int64_t t = (int64_t)blockIdx.x * rows_per_block + rl; t < T; t += (int64_t)gridDim.x * rows_per_block) {
            const int64_t r = dy_rows ? dy_rows[t] : t;
            const float4 g = ld4(dy + r * lddy + col0 + 4 * c);
            const int a_t = idx[t];
#pragma unroll
            for (int a = 0; a < TBL_MAXROWS; ++a) {
                const float m = (a_t == a) ? 1.f : 0.f;
                acc[a].x += m * g.x; acc[a].y += m * g.y; acc[a].z += m * g.z; acc[a].w += m * g.w;
            }
        }
#pragma unroll
        for (int a = 0; a < TBL_MAXROWS; ++a) {
            if (a < nrows) {
                if (DET) {
                    *reinterpret_cast<float4*>(&lds_tbl[(rl * nrows + a) * E + c * 4]) = acc[a];
                } else {
                    atomicAdd(&lds_tbl[a * E + c * 4 + 0], acc[a].x);
                    atomicAdd(&lds_tbl[a * E + c * 4 + 1], acc[a].y);
                    atomicAdd(&lds_tbl[a * E + c * 4 + 2], acc[a].z);
                    atomicAdd(&lds_tbl[a * E + c * 4 + 3], acc[a].w);
                }
            }
        }
    }
    __syncthreads();
    if (DET) {
        for (int i = threadIdx.x; i < nrows * E; i += EW_THREADS) {
            float t = 0.f;
            for (int g = 0; g < rows_per_block; ++g) t += lds_tbl[g * nrows * E + i];
            partial[(int64_t)blockIdx.x * nrows * E + i] = t;
        }
    } else {
        for (int i = threadIdx.x; i < nrows * E; i += EW_THREADS) atomicAdd(&dtable[i], lds_tbl[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// q/k per-head RMSNorm + RoPE (+ behaviour bias), head_dim = 64: one wave per (token, head)
// ---------------------------------------------------------------------------------------------
// 16 lanes x float4 per (token, head) row, four rows per wave: 16-byte accesses, the 64-wide reductions
// are four xor-shuffles inside the 16-lane group, the RoPE partner (d +- 32) is lane +- 8 of the group.
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// TA = activation type of qkv / q_rot / k_rot.  With bf16 activations the arithmetic mirrors what the reference does
// under autocast (ref:SeqRec/tasks/train_SMB_decoder.py:114-118): the projection output is bf16; for the self attention
// Qwen3MoeRMSNorm sees a bf16 input and rounds the normalised value to bf16 before the fp32 weight multiply; for the
// cross attention the fp32 behaviour bias is added first, so the norm runs in fp32; RoPE is fp32; SDPA's inputs
// (q_rot, k_rot, v) are rounded to bf16.  The biased pre-norm values are NOT written back in bf16 (the backward adds
// the bias again, exactly); the fp32 kernels keep writing them back as in round 1.
template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_fwd_kernel(TA* __restrict__ qkv, int T, int S, int nq, int nkv,
                       const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                       const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                       const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                       const float* __restrict__ bias_v, const int32_t* __restrict__ act_idx,
                       TA* __restrict__ q_rot, TA* __restrict__ k_rot, const int32_t* __restrict__ pos_ids,
                       uint32_t* __restrict__ amax_q, uint32_t* __restrict__ amax_k, uint32_t* __restrict__ amax_v) {
    constexpr bool F32 = sizeof(TA) == 4;
    __shared__ uint32_t amax_lds[3][4];
    uint32_t amq = 0, amk = 0, amv = 0;                 // gamer_amax_sink(3): max |q_rot|, max |k_rot|, max |v + bias_v|
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * EW_THREADS) >> 6;
    const bool cross = bias_q != nullptr;
    const int NH = nq + nkv + (cross ? nkv : 0);
    const int ldqkv = (nq + 2 * nkv) * 64;
    const float4 wq4 = reinterpret_cast<const float4*>(wq)[g], wk4 = reinterpret_cast<const float4*>(wk)[g];
    const float sgn = g < 8 ? -1.f : 1.f;
    const int64_t total = (int64_t)T * NH;
    for (int64_t i0 = wave * 4; i0 < total; i0 += nwaves * 4) {
        const int64_t i = i0 + sub;
        const bool live = i < total;
        const int64_t ic = live ? i : total - 1;
        const int t = (int)(ic / NH);
        const int hd = (int)(ic % NH);
        const int a = cross ? act_idx[t] : 0;
        TA* row = qkv + (int64_t)t * ldqkv;
        if (hd < nq + nkv) {
            const bool isq = hd < nq;
            TA* src = row + hd * 64 + 4 * g;   // q heads then k heads are contiguous
            float4 x = ld4(src);
            if (cross) {
                const float4 b4 = isq ? reinterpret_cast<const float4*>(bias_q + (int64_t)a * nq * 64 + hd * 64)[g]
                                      : reinterpret_cast<const float4*>(bias_k + (int64_t)a * nkv * 64 + (hd - nq) * 64)[g];
                x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
                if (F32 && live) st4(src, x);             // fp32: keep the pre-norm (biased) value for the backward
            }
            const float ss = group16_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
            const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
            const float4 w4 = isq ? wq4 : wk4;
            float4 xn;
            xn.x = x.x * rstd; xn.y = x.y * rstd; xn.z = x.z * rstd; xn.w = x.w * rstd;
            if (!F32 && !cross) {                         // Qwen3MoeRMSNorm on a bf16 tensor: .to(input_dtype) before * weight
                xn.x = round_as<TA>(xn.x); xn.y = round_as<TA>(xn.y); xn.z = round_as<TA>(xn.z); xn.w = round_as<TA>(xn.w);
            }
            float4 y;
            y.x = w4.x * xn.x; y.y = w4.y * xn.y; y.z = w4.z * xn.z; y.w = w4.w * xn.w;
            float4 pr;
            pr.x = __shfl_xor(y.x, 8, 64); pr.y = __shfl_xor(y.y, 8, 64); pr.z = __shfl_xor(y.z, 8, 64); pr.w = __shfl_xor(y.w, 8, 64);
            const int pos = pos_ids ? pos_ids[t] : t % S;        // per-token RoPE position (session model) or the index
            const float4 c4 = reinterpret_cast<const float4*>(cos_t + pos * 64)[g];
            const float4 s4 = reinterpret_cast<const float4*>(sin_t + pos * 64)[g];
            float4 o;
            o.x = y.x * c4.x + sgn * pr.x * s4.x; o.y = y.y * c4.y + sgn * pr.y * s4.y;
            o.z = y.z * c4.z + sgn * pr.z * s4.z; o.w = y.w * c4.w + sgn * pr.w * s4.w;
            if (live) {
                if (isq) { st4(q_rot + (int64_t)t * nq * 64 + hd * 64 + 4 * g, o); amq = amax_f4(amq, o); }
                else { st4(k_rot + (int64_t)t * nkv * 64 + (hd - nq) * 64 + 4 * g, o); amk = amax_f4(amk, o); }
            }
        } else {
            // (only reached when cross) v += bias_v; the shuffles above are skipped by the whole 16-lane group
            const int hv = hd - nq - nkv;
            TA* dst = row + (nq + nkv + hv) * 64 + 4 * g;
            const float4 b4 = reinterpret_cast<const float4*>(bias_v + (int64_t)a * nkv * 64 + hv * 64)[g];
            float4 x = ld4(dst);
            x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
            if (live) { st4(dst, x); amv = amax_f4(amv, x); }
        }
    }
    amax_block_commit(amq, amax_q, amax_lds[0]);
    amax_block_commit(amk, amax_k, amax_lds[1]);
    amax_block_commit(amv, amax_v, amax_lds[2]);
}

// fp32, token-major: one 16-lane group per TOKEN, walking its q and k heads three at a time (three 16-byte loads in flight
// per lane, the token's cos / sin rows and behaviour index loaded once instead of once per head, no 64-bit division per row).
// Same arithmetic per element as the kernel above (bit-identical outputs).  The row-major form above reached 3.5 TB/s of
// 5.6 achievable: one dependent load -> reduce -> store chain per lane and per iteration, behind ~100 integer instructions.
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_fwd_tok_kernel(float* __restrict__ qkv, int T, int S, int nq, int nkv,
                           const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                           const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                           const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                           const float* __restrict__ bias_v, const int32_t* __restrict__ act_idx,
                           float* __restrict__ q_rot, float* __restrict__ k_rot, const int32_t* __restrict__ pos_ids,
                           uint32_t* __restrict__ amax_q, uint32_t* __restrict__ amax_k, uint32_t* __restrict__ amax_v) {
    __shared__ uint32_t amax_lds[3][4];
    uint32_t amq = 0, amk = 0, amv = 0;
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * EW_THREADS) >> 6;
    const bool cross = bias_q != nullptr;
    const int nqk = nq + nkv;
    const int ldqkv = (nq + 2 * nkv) * 64;
    const float4 wq4 = reinterpret_cast<const float4*>(wq)[g], wk4 = reinterpret_cast<const float4*>(wk)[g];
    const float sgn = g < 8 ? -1.f : 1.f;
    for (int64_t t0 = wave * 4; t0 < T; t0 += nwaves * 4) {
        const int64_t tt = t0 + sub;
        const bool live = tt < T;
        const int t = (int)(live ? tt : T - 1);
        const int a = cross ? act_idx[t] : 0;
        const int pos = pos_ids ? pos_ids[t] : t % S;
        const float4 c4 = reinterpret_cast<const float4*>(cos_t + pos * 64)[g];
        const float4 s4 = reinterpret_cast<const float4*>(sin_t + pos * 64)[g];
        float* row = qkv + (int64_t)t * ldqkv + 4 * g;
        const float* bq = cross ? bias_q + (int64_t)a * nq * 64 + 4 * g : nullptr;
        const float* bk = cross ? bias_k + (int64_t)a * nkv * 64 + 4 * g : nullptr;
        for (int h0 = 0; h0 < nqk; h0 += 3) {
            float4 x[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) x[u] = ld4(row + min(h0 + u, nqk - 1) * 64);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = h0 + u;
                if (hd >= nqk) break;                       // (uniform over the wave)
                const bool isq = hd < nq;
                float4 xx = x[u];
                if (cross) {
                    const float4 b4 = *reinterpret_cast<const float4*>(isq ? bq + hd * 64 : bk + (hd - nq) * 64);
                    xx.x += b4.x; xx.y += b4.y; xx.z += b4.z; xx.w += b4.w;
                    if (live) st4(row + hd * 64, xx);       // keep the pre-norm (biased) value for the backward
                }
                const float ss = group16_sum(xx.x * xx.x + xx.y * xx.y + xx.z * xx.z + xx.w * xx.w);
                const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
                const float4 w4 = isq ? wq4 : wk4;
                float4 xn;
                xn.x = xx.x * rstd; xn.y = xx.y * rstd; xn.z = xx.z * rstd; xn.w = xx.w * rstd;
                float4 y;
                y.x = w4.x * xn.x; y.y = w4.y * xn.y; y.z = w4.z * xn.z; y.w = w4.w * xn.w;
                float4 pr;
                pr.x = __shfl_xor(y.x, 8, 64); pr.y = __shfl_xor(y.y, 8, 64); pr.z = __shfl_xor(y.z, 8, 64); pr.w = __shfl_xor(y.w, 8, 64);
                float4 o;
                o.x = y.x * c4.x + sgn * pr.x * s4.x; o.y = y.y * c4.y + sgn * pr.y * s4.y;
                o.z = y.z * c4.z + sgn * pr.z * s4.z; o.w = y.w * c4.w + sgn * pr.w * s4.w;
                if (live) {
                    if (isq) { st4(q_rot + (int64_t)t * nq * 64 + hd * 64 + 4 * g, o); amq = amax_f4(amq, o); }
                    else { st4(k_rot + (int64_t)t * nkv * 64 + (hd - nq) * 64 + 4 * g, o); amk = amax_f4(amk, o); }
                }
            }
        }
        if (cross) {                                        // v += bias_v
            for (int hv = 0; hv < nkv; ++hv) {
                float* dst = row + (nqk + hv) * 64;
                const float4 b4 = *reinterpret_cast<const float4*>(bias_v + (int64_t)a * nkv * 64 + hv * 64 + 4 * g);
                float4 xv = ld4(dst);
                xv.x += b4.x; xv.y += b4.y; xv.z += b4.z; xv.w += b4.w;
                if (live) { st4(dst, xv); amv = amax_f4(amv, xv); }
            }
        }
    }
    amax_block_commit(amq, amax_q, amax_lds[0]);
    amax_block_commit(amk, amax_k, amax_lds[1]);
    amax_block_commit(amv, amax_v, amax_lds[2]);
}

// Each wave keeps one head for its whole life so that the norm-weight and bias gradients accumulate in
// registers (four tokens per iteration, one per 16-lane group).  At the end every wave writes its sums as one row of
// `partial` [waves_per_head][NH][1 + nb1][64] (slot 0: norm weight, slots 1..: bias rows) and
// qknorm_partial_reduce_kernel folds the rows in a fixed order.  (Round 1 finished with same-address float atomics
// from ~8000 waves into 64 + nb1*64 addresses per head: a 0.4 ms floor per call at every batch size.)
template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_bwd_kernel(const TA* __restrict__ qkv, const TA* __restrict__ dq_rot,
                       const TA* __restrict__ dk_rot, int T, int S, int nq, int nkv,
                       const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                       const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                       int cross, const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                       const int32_t* __restrict__ act_idx, int nb1,
                       TA* __restrict__ dqkv, float* __restrict__ partial,
                       int waves_per_head, const int32_t* __restrict__ pos_ids, uint32_t* __restrict__ amax_out) {
    constexpr bool F32 = sizeof(TA) == 4;
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int NH = nq + nkv + (cross ? nkv : 0);
    if (wave >= (int64_t)NH * waves_per_head) return;
    float am = 0.f;                                    // gamer_amax_sink: max |dq|, |dk| written here (one atomic per wave)
    const int hd = (int)(wave % NH);
    const int w0 = (int)(wave / NH);
    const int ldqkv = (nq + 2 * nkv) * 64;
    float4 dwacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dbacc[TBL_MAXROWS];
#pragma unroll
    for (int a = 0; a < TBL_MAXROWS; ++a) dbacc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float sgn = g < 8 ? 1.f : -1.f;             // transpose of the rotation
    if (hd < nq + nkv) {
        const bool isq = hd < nq;
        const float4 w4 = isq ? reinterpret_cast<const float4*>(wq)[g] : reinterpret_cast<const float4*>(wk)[g];
        for (int tb = w0 * 4; tb < T; tb += waves_per_head * 4) {
            const int t = tb + sub;
            const bool live = t < T;
            const int tc = live ? t : T - 1;
            float4 x = ld4(qkv + (int64_t)tc * ldqkv + hd * 64 + 4 * g);   // pre-norm (fp32: bias included)
            const int a_t = cross ? act_idx[tc] : 0;
            if (!F32 && cross) {
                const float4 b4 = isq ? reinterpret_cast<const float4*>(bias_q + (int64_t)a_t * nq * 64 + hd * 64)[g]
                                      : reinterpret_cast<const float4*>(bias_k + (int64_t)a_t * nkv * 64 + (hd - nq) * 64)[g];
                x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
            }
            const float4 d = isq ? ld4(dq_rot + (int64_t)tc * nq * 64 + hd * 64 + 4 * g)
                                 : ld4(dk_rot + (int64_t)tc * nkv * 64 + (hd - nq) * 64 + 4 * g);
            const int pos = pos_ids ? pos_ids[tc] : tc % S;
            const float4 c4 = reinterpret_cast<const float4*>(cos_t + pos * 64)[g];
            const float4 s4 = reinterpret_cast<const float4*>(sin_t + pos * 64)[g];
            float4 dp;
            dp.x = __shfl_xor(d.x, 8, 64); dp.y = __shfl_xor(d.y, 8, 64); dp.z = __shfl_xor(d.z, 8, 64); dp.w = __shfl_xor(d.w, 8, 64);
            // dy_j = dout_j cos_j + (j<32 ? +1 : -1) * dout_partner * sin_j
            float4 dy;
            dy.x = d.x * c4.x + sgn * dp.x * s4.x; dy.y = d.y * c4.y + sgn * dp.y * s4.y;
            dy.z = d.z * c4.z + sgn * dp.z * s4.z; dy.w = d.w * c4.w + sgn * dp.w * s4.w;
            const float ss = group16_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
            const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
            float4 xh;
            xh.x = x.x * rstd; xh.y = x.y * rstd; xh.z = x.z * rstd; xh.w = x.w * rstd;
            const float lv = live ? 1.f : 0.f;
            if (!F32 && !cross) {
                // the weight multiplied the bf16-rounded normalised value in the forward (the cast's gradient is identity)
                dwacc.x += lv * dy.x * round_as<TA>(xh.x); dwacc.y += lv * dy.y * round_as<TA>(xh.y);
                dwacc.z += lv * dy.z * round_as<TA>(xh.z); dwacc.w += lv * dy.w * round_as<TA>(xh.w);
            } else {
                dwacc.x += lv * dy.x * xh.x; dwacc.y += lv * dy.y * xh.y; dwacc.z += lv * dy.z * xh.z; dwacc.w += lv * dy.w * xh.w;
            }
            float4 gg;
            gg.x = dy.x * w4.x; gg.y = dy.y * w4.y; gg.z = dy.z * w4.z; gg.w = dy.w * w4.w;
            const float dot = group16_sum(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * (1.f / 64.f);
            float4 dx;
            dx.x = rstd * (gg.x - xh.x * dot); dx.y = rstd * (gg.y - xh.y * dot);
            dx.z = rstd * (gg.z - xh.z * dot); dx.w = rstd * (gg.w - xh.w * dot);
            if (live) {
                st4(dqkv + (int64_t)t * ldqkv + hd * 64 + 4 * g, dx);
                am = fmaxf(fmaxf(fmaxf(am, fabsf(dx.x)), fabsf(dx.y)), fmaxf(fabsf(dx.z), fabsf(dx.w)));
            }
            if (cross) {
                const int a_l = live ? a_t : -1;
#pragma unroll
                for (int a = 0; a < TBL_MAXROWS; ++a) {
                    const float m = (a_l == a) ? 1.f : 0.f;
                    dbacc[a].x += m * dx.x; dbacc[a].y += m * dx.y; dbacc[a].z += m * dx.z; dbacc[a].w += m * dx.w;
                }
            }
        }
    } else {
        const int hv = hd - nq - nkv;
        for (int tb = w0 * 4; tb < T; tb += waves_per_head * 4) {
            const int t = tb + sub;
            const bool live = t < T;
            const int tc = live ? t : T - 1;
            const float4 dv = ld4(dqkv + (int64_t)tc * ldqkv + (nq + nkv + hv) * 64 + 4 * g);
            const int a_t = live ? act_idx[t] : -1;
#pragma unroll
            for (int a = 0; a < TBL_MAXROWS; ++a) {
                const float m = (a_t == a) ? 1.f : 0.f;
                dbacc[a].x += m * dv.x; dbacc[a].y += m * dv.y; dbacc[a].z += m * dv.z; dbacc[a].w += m * dv.w;
            }
        }
    }
    if (amax_out) {
        uint32_t m = __float_as_uint(am);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0 && m) amax_publish(m, amax_out, (uint32_t)wave);
    }
    // fold the four 16-lane groups, then group 0 writes this wave's row of partial sums
    auto fold = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    const int SL = 1 + (cross ? nb1 : 0);
    float* prow = partial + ((int64_t)w0 * NH + hd) * SL * 64;
    {
        const float a0 = fold(dwacc.x), a1 = fold(dwacc.y), a2 = fold(dwacc.z), a3 = fold(dwacc.w);
        if (sub == 0) reinterpret_cast<float4*>(prow)[g] = make_float4(a0, a1, a2, a3);      // zeros for the v heads
    }
    if (cross) {
#pragma unroll
        for (int a = 0; a < TBL_MAXROWS; ++a) {
            const float b0 = fold(dbacc[a].x), b1 = fold(dbacc[a].y), b2 = fold(dbacc[a].z), b3 = fold(dbacc[a].w);
            if (a < nb1 && sub == 0) reinterpret_cast<float4*>(prow + (1 + a) * 64)[g] = make_float4(b0, b1, b2, b3);
        }
    }
}

// fp32, self attention (no behaviour biases): token-major like qknorm_rope_fwd_tok_kernel - one 16-lane group per token walks its
// q and k heads three at a time (six 16-byte loads in flight per lane), cos / sin once per token; the two norm-weight gradients
// accumulate in registers over all the heads of all the tokens of the wave and leave as ONE partial row [dwq | dwk] per wave
// (qknorm_partial_reduce_kernel with nq = nkv = 1 folds them).  Same arithmetic per element as qknorm_rope_bwd_kernel.
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_bwd_tok_kernel(const float* __restrict__ qkv, const float* __restrict__ dq_rot, const float* __restrict__ dk_rot,
                           int T, int S, int nq, int nkv, const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                           const float* __restrict__ cos_t, const float* __restrict__ sin_t, float* __restrict__ dqkv,
                           float* __restrict__ partial, int n_waves, const int32_t* __restrict__ pos_ids,
                           uint32_t* __restrict__ amax_out) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int wave = (int)(((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6);
    if (wave >= n_waves) return;
    float am = 0.f;
    const int nqk = nq + nkv;
    const int ldqkv = (nq + 2 * nkv) * 64;
    const float4 wq4 = reinterpret_cast<const float4*>(wq)[g], wk4 = reinterpret_cast<const float4*>(wk)[g];
    float4 dwq = make_float4(0.f, 0.f, 0.f, 0.f), dwk = dwq;
    const float sgn = g < 8 ? 1.f : -1.f;             // transpose of the rotation
    for (int tb = wave * 4; tb < T; tb += n_waves * 4) {
        const int t = tb + sub;
        const bool live = t < T;
        const int tc = live ? t : T - 1;
        const int pos = pos_ids ? pos_ids[tc] : tc % S;
        const float4 c4 = reinterpret_cast<const float4*>(cos_t + pos * 64)[g];
        const float4 s4 = reinterpret_cast<const float4*>(sin_t + pos * 64)[g];
        const float* xrow = qkv + (int64_t)tc * ldqkv + 4 * g;
        const float* dqrow = dq_rot + (int64_t)tc * nq * 64 + 4 * g;
        const float* dkrow = dk_rot + (int64_t)tc * nkv * 64 + 4 * g;
        const float lv = live ? 1.f : 0.f;
        for (int h0 = 0; h0 < nqk; h0 += 3) {
            float4 xs[3], ds[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = min(h0 + u, nqk - 1);
                xs[u] = ld4(xrow + hd * 64);
                ds[u] = hd < nq ? ld4(dqrow + hd * 64) : ld4(dkrow + (hd - nq) * 64);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = h0 + u;
                if (hd >= nqk) break;                       // (uniform over the wave)
                const bool isq = hd < nq;
                const float4 x = xs[u], d = ds[u];
                const float4 w4 = isq ? wq4 : wk4;
                float4 dp;
                dp.x = __shfl_xor(d.x, 8, 64); dp.y = __shfl_xor(d.y, 8, 64); dp.z = __shfl_xor(d.z, 8, 64); dp.w = __shfl_xor(d.w, 8, 64);
                float4 dy;
                dy.x = d.x * c4.x + sgn * dp.x * s4.x; dy.y = d.y * c4.y + sgn * dp.y * s4.y;
                dy.z = d.z * c4.z + sgn * dp.z * s4.z; dy.w = d.w * c4.w + sgn * dp.w * s4.w;
                const float ss = group16_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
                const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
                float4 xh;
                xh.x = x.x * rstd; xh.y = x.y * rstd; xh.z = x.z * rstd; xh.w = x.w * rstd;
                if (isq) { dwq.x += lv * dy.x * xh.x; dwq.y += lv * dy.y * xh.y; dwq.z += lv * dy.z * xh.z; dwq.w += lv * dy.w * xh.w; }
                else { dwk.x += lv * dy.x * xh.x; dwk.y += lv * dy.y * xh.y; dwk.z += lv * dy.z * xh.z; dwk.w += lv * dy.w * xh.w; }
                float4 gg;
                gg.x = dy.x * w4.x; gg.y = dy.y * w4.y; gg.z = dy.z * w4.z; gg.w = dy.w * w4.w;
                const float dot = group16_sum(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * (1.f / 64.f);
                float4 dx;
                dx.x = rstd * (gg.x - xh.x * dot); dx.y = rstd * (gg.y - xh.y * dot);
                dx.z = rstd * (gg.z - xh.z * dot); dx.w = rstd * (gg.w - xh.w * dot);
                if (live) {
                    st4(dqkv + (int64_t)t * ldqkv + hd * 64 + 4 * g, dx);
                    am = fmaxf(fmaxf(fmaxf(am, fabsf(dx.x)), fabsf(dx.y)), fmaxf(fabsf(dx.z), fabsf(dx.w)));
                }
            }
        }
    }
    if (amax_out) {
        uint32_t m = __float_as_uint(am);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0 && m) amax_publish(m, amax_out, (uint32_t)wave);
    }
    auto fold = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    float* prow = partial + (int64_t)wave * 128;
    const float a0 = fold(dwq.x), a1 = fold(dwq.y), a2 = fold(dwq.z), a3 = fold(dwq.w);
    const float b0 = fold(dwk.x), b1 = fold(dwk.y), b2 = fold(dwk.z), b3 = fold(dwk.w);
    if (sub == 0) {
        reinterpret_cast<float4*>(prow)[g] = make_float4(a0, a1, a2, a3);
        reinterpret_cast<float4*>(prow + 64)[g] = make_float4(b0, b1, b2, b3);
    }
}

// ---- bf16 activations: 8 lanes x 8 elements (16 bytes) per (token, head) row, 8 rows per wave --------------------
// Same arithmetic as the <bf16_t> instantiations of the kernels above (which move 8 bytes per lane and ran at
// 2.4 TB/s); the RoPE partner (d +- 32) is lane +- 4 of the 8-lane group.
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&f)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
__device__ __forceinline__ void ld8(const float* p, float (&f)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&f)[8]) {
    f32x8v v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f[e];
    *reinterpret_cast<bf16x8*>(p) = __builtin_convertvector(v, bf16x8);
}

__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_fwd_b8_kernel(bf16_t* __restrict__ qkv, int T, int S, int nq, int nkv,
                          const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                          const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                          const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                          const float* __restrict__ bias_v, const int32_t* __restrict__ act_idx,
                          bf16_t* __restrict__ q_rot, bf16_t* __restrict__ k_rot, const int32_t* __restrict__ pos_ids) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 7, sub = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * EW_THREADS) >> 6;
    const bool cross = bias_q != nullptr;
    const int NH = nq + nkv + (cross ? nkv : 0);
    const int ldqkv = (nq + 2 * nkv) * 64;
    float wqv[8], wkv[8];
    ld8(wq + 8 * g, wqv);
    ld8(wk + 8 * g, wkv);
    const float sgn = g < 4 ? -1.f : 1.f;
    const int64_t total = (int64_t)T * NH;
    for (int64_t i0 = wave * 8; i0 < total; i0 += nwaves * 8) {
        const int64_t i = i0 + sub;
        const bool live = i < total;
        const int64_t ic = live ? i : total - 1;
        const int t = (int)(ic / NH);
        const int hd = (int)(ic % NH);
        const int a = cross ? act_idx[t] : 0;
        bf16_t* row = qkv + (int64_t)t * ldqkv;
        if (hd < nq + nkv) {
            const bool isq = hd < nq;
            float x[8];
            ld8(row + hd * 64 + 8 * g, x);
            if (cross) {
                float b[8];
                ld8(isq ? bias_q + (int64_t)a * nq * 64 + hd * 64 + 8 * g : bias_k + (int64_t)a * nkv * 64 + (hd - nq) * 64 + 8 * g, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += b[e];
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
            ss = group8_sum(ss);
            const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
            float y[8], pr[8], c[8], sn[8], o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xn = x[e] * rstd;
                if (!cross) xn = round_as<bf16_t>(xn);        // Qwen3MoeRMSNorm on a bf16 tensor: .to(input_dtype) before * weight
                y[e] = (isq ? wqv[e] : wkv[e]) * xn;
                pr[e] = __shfl_xor(y[e], 4, 64);
            }
            const int pos = pos_ids ? pos_ids[t] : t % S;
            ld8(cos_t + pos * 64 + 8 * g, c);
            ld8(sin_t + pos * 64 + 8 * g, sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = y[e] * c[e] + sgn * pr[e] * sn[e];
            if (live) {
                if (isq) st8(q_rot + (int64_t)t * nq * 64 + hd * 64 + 8 * g, o);
                else st8(k_rot + (int64_t)t * nkv * 64 + (hd - nq) * 64 + 8 * g, o);
            }
        } else {
            const int hv = hd - nq - nkv;
            bf16_t* dst = row + (nq + nkv + hv) * 64 + 8 * g;
            float x[8], b[8];
            ld8(dst, x);
            ld8(bias_v + (int64_t)a * nkv * 64 + hv * 64 + 8 * g, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += b[e];
            if (live) st8(dst, x);
        }
    }
}

// token-major form of the kernel above (see qknorm_rope_fwd_tok_kernel): one 8-lane group per token, its heads three at a
// time, cos / sin / behaviour index once per token; same arithmetic per element
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_fwd_b8_tok_kernel(bf16_t* __restrict__ qkv, int T, int S, int nq, int nkv,
                              const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                              const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                              const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                              const float* __restrict__ bias_v, const int32_t* __restrict__ act_idx,
                              bf16_t* __restrict__ q_rot, bf16_t* __restrict__ k_rot, const int32_t* __restrict__ pos_ids) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 7, sub = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * EW_THREADS) >> 6;
    const bool cross = bias_q != nullptr;
    const int nqk = nq + nkv;
    const int ldqkv = (nq + 2 * nkv) * 64;
    float wqv[8], wkv[8];
    ld8(wq + 8 * g, wqv);
    ld8(wk + 8 * g, wkv);
    const float sgn = g < 4 ? -1.f : 1.f;
    for (int64_t t0 = wave * 8; t0 < T; t0 += nwaves * 8) {
        const int64_t tt = t0 + sub;
        const bool live = tt < T;
        const int t = (int)(live ? tt : T - 1);
        const int a = cross ? act_idx[t] : 0;
        const int pos = pos_ids ? pos_ids[t] : t % S;
        float c[8], sn[8];
        ld8(cos_t + pos * 64 + 8 * g, c);
        ld8(sin_t + pos * 64 + 8 * g, sn);
        bf16_t* row = qkv + (int64_t)t * ldqkv + 8 * g;
        for (int h0 = 0; h0 < nqk; h0 += 3) {
            float x[3][8];
#pragma unroll
            for (int u = 0; u < 3; ++u) ld8(row + min(h0 + u, nqk - 1) * 64, x[u]);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = h0 + u;
                if (hd >= nqk) break;                       // (uniform over the wave)
                const bool isq = hd < nq;
                if (cross) {
                    float b[8];
                    ld8(isq ? bias_q + (int64_t)a * nq * 64 + hd * 64 + 8 * g : bias_k + (int64_t)a * nkv * 64 + (hd - nq) * 64 + 8 * g, b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[u][e] += b[e];
                }
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x[u][e] * x[u][e];
                ss = group8_sum(ss);
                const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
                float y[8], pr[8], o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xn = x[u][e] * rstd;
                    if (!cross) xn = round_as<bf16_t>(xn);    // Qwen3MoeRMSNorm on a bf16 tensor: .to(input_dtype) before * weight
                    y[e] = (isq ? wqv[e] : wkv[e]) * xn;
                    pr[e] = __shfl_xor(y[e], 4, 64);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = y[e] * c[e] + sgn * pr[e] * sn[e];
                if (live) {
                    if (isq) st8(q_rot + (int64_t)t * nq * 64 + hd * 64 + 8 * g, o);
                    else st8(k_rot + (int64_t)t * nkv * 64 + (hd - nq) * 64 + 8 * g, o);
                }
            }
        }
        if (cross) {
            for (int hv = 0; hv < nkv; ++hv) {
                bf16_t* dst = row + (nqk + hv) * 64;
                float xv[8], b[8];
                ld8(dst, xv);
                ld8(bias_v + (int64_t)a * nkv * 64 + hv * 64 + 8 * g, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] += b[e];
                if (live) st8(dst, xv);
            }
        }
    }
}

// token-major form for the self-attention calls (see qknorm_rope_bwd_tok_kernel): one 8-lane group per token walks its heads three
// at a time, cos / sin once per token, the two norm-weight gradients accumulate over all heads and leave as one partial row
// [dwq | dwk] per wave; same arithmetic per element as the kernel below with cross = 0
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_bwd_b8_tok_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dq_rot, const bf16_t* __restrict__ dk_rot,
                              int T, int S, int nq, int nkv, const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                              const float* __restrict__ cos_t, const float* __restrict__ sin_t, bf16_t* __restrict__ dqkv,
                              float* __restrict__ partial, int n_waves, const int32_t* __restrict__ pos_ids) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 7, sub = lane >> 3;
    const int wave = (int)(((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6);
    if (wave >= n_waves) return;
    const int nqk = nq + nkv;
    const int ldqkv = (nq + 2 * nkv) * 64;
    float wq8[8], wk8[8], dwq[8], dwk[8];
    ld8(wq + 8 * g, wq8);
    ld8(wk + 8 * g, wk8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwq[e] = 0.f; dwk[e] = 0.f; }
    const float sgn = g < 4 ? 1.f : -1.f;             // transpose of the rotation
    for (int tb = wave * 8; tb < T; tb += n_waves * 8) {
        const int t = tb + sub;
        const bool live = t < T;
        const int tc = live ? t : T - 1;
        const int pos = pos_ids ? pos_ids[tc] : tc % S;
        float c[8], sn[8];
        ld8(cos_t + pos * 64 + 8 * g, c);
        ld8(sin_t + pos * 64 + 8 * g, sn);
        const bf16_t* xrow = qkv + (int64_t)tc * ldqkv + 8 * g;
        const bf16_t* dqrow = dq_rot + (int64_t)tc * nq * 64 + 8 * g;
        const bf16_t* dkrow = dk_rot + (int64_t)tc * nkv * 64 + 8 * g;
        const float lv = live ? 1.f : 0.f;
        for (int h0 = 0; h0 < nqk; h0 += 3) {
            float xs[3][8], ds[3][8];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = min(h0 + u, nqk - 1);
                ld8(xrow + hd * 64, xs[u]);
                ld8(hd < nq ? dqrow + hd * 64 : dkrow + (hd - nq) * 64, ds[u]);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int hd = h0 + u;
                if (hd >= nqk) break;                       // (uniform over the wave)
                const bool isq = hd < nq;
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xs[u][e] * xs[u][e];
                ss = group8_sum(ss);
                const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
                float gg[8], xh[8], dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dp = __shfl_xor(ds[u][e], 4, 64);
                    const float dy = ds[u][e] * c[e] + sgn * dp * sn[e];
                    xh[e] = xs[u][e] * rstd;
                    const float contrib = lv * dy * round_as<bf16_t>(xh[e]);
                    if (isq) dwq[e] += contrib; else dwk[e] += contrib;
                    gg[e] = dy * (isq ? wq8[e] : wk8[e]);
                    dot += gg[e] * xh[e];
                }
                dot = group8_sum(dot) * (1.f / 64.f);
                float dx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dx[e] = rstd * (gg[e] - xh[e] * dot);
                if (live) st8(dqkv + (int64_t)t * ldqkv + hd * 64 + 8 * g, dx);
            }
        }
    }
    auto fold = [&](float v) { v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    float* prow = partial + (int64_t)wave * 128;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = fold(dwq[e]), b = fold(dwk[e]);
        if (sub == 0) { prow[8 * g + e] = a; prow[64 + 8 * g + e] = b; }
    }
}

// NBR = compile-time bound of the bias-table rows (4 for the shipped three behaviours, 8 otherwise)
template <int NBR>
__global__ void __launch_bounds__(EW_THREADS)
qknorm_rope_bwd_b8_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dq_rot,
                          const bf16_t* __restrict__ dk_rot, int T, int S, int nq, int nkv,
                          const float* __restrict__ wq, const float* __restrict__ wk, float eps,
                          const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                          int cross, const float* __restrict__ bias_q, const float* __restrict__ bias_k,
                          const int32_t* __restrict__ act_idx, int nb1,
                          bf16_t* __restrict__ dqkv, float* __restrict__ partial,
                          int waves_per_head, const int32_t* __restrict__ pos_ids) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 7, sub = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int NH = nq + nkv + (cross ? nkv : 0);
    if (wave >= (int64_t)NH * waves_per_head) return;
    const int hd = (int)(wave % NH);
    const int w0 = (int)(wave / NH);
    const int ldqkv = (nq + 2 * nkv) * 64;
    float dwacc[8], dbacc[NBR][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dwacc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < NBR; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) dbacc[a][e] = 0.f;
    const float sgn = g < 4 ? 1.f : -1.f;             // transpose of the rotation
    if (hd < nq + nkv) {
        const bool isq = hd < nq;
        float w8[8];
        ld8((isq ? wq : wk) + 8 * g, w8);
        for (int tb = w0 * 8; tb < T; tb += waves_per_head * 8) {
            const int t = tb + sub;
            const bool live = t < T;
            const int tc = live ? t : T - 1;
            float x[8], d[8], c[8], sn[8];
            ld8(qkv + (int64_t)tc * ldqkv + hd * 64 + 8 * g, x);
            const int a_t = cross ? act_idx[tc] : 0;
            if (cross) {
                float b[8];
                ld8(isq ? bias_q + (int64_t)a_t * nq * 64 + hd * 64 + 8 * g : bias_k + (int64_t)a_t * nkv * 64 + (hd - nq) * 64 + 8 * g, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += b[e];
            }
            ld8(isq ? dq_rot + (int64_t)tc * nq * 64 + hd * 64 + 8 * g : dk_rot + (int64_t)tc * nkv * 64 + (hd - nq) * 64 + 8 * g, d);
            const int pos = pos_ids ? pos_ids[tc] : tc % S;
            ld8(cos_t + pos * 64 + 8 * g, c);
            ld8(sin_t + pos * 64 + 8 * g, sn);
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
            ss = group8_sum(ss);
            const float rstd = rsqrtf(ss * (1.f / 64.f) + eps);
            const float lv = live ? 1.f : 0.f;
            float gg[8], xh[8], dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dp = __shfl_xor(d[e], 4, 64);
                const float dy = d[e] * c[e] + sgn * dp * sn[e];
                xh[e] = x[e] * rstd;
                dwacc[e] += lv * dy * (cross ? xh[e] : round_as<bf16_t>(xh[e]));
                gg[e] = dy * w8[e];
                dot += gg[e] * xh[e];
            }
            dot = group8_sum(dot) * (1.f / 64.f);
            float dx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dx[e] = rstd * (gg[e] - xh[e] * dot);
            if (live) st8(dqkv + (int64_t)t * ldqkv + hd * 64 + 8 * g, dx);
            if (cross) {
                const int a_l = live ? a_t : -1;
#pragma unroll
                for (int a = 0; a < NBR; ++a) {
                    const float m = (a_l == a) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dbacc[a][e] += m * dx[e];
                }
            }
        }
    } else {
        const int hv = hd - nq - nkv;
        for (int tb = w0 * 8; tb < T; tb += waves_per_head * 8) {
            const int t = tb + sub;
            const bool live = t < T;
            const int tc = live ? t : T - 1;
            float dv[8];
            ld8(dqkv + (int64_t)tc * ldqkv + (nq + nkv + hv) * 64 + 8 * g, dv);
            const int a_t = live ? act_idx[t] : -1;
#pragma unroll
            for (int a = 0; a < NBR; ++a) {
                const float m = (a_t == a) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dbacc[a][e] += m * dv[e];
            }
        }
    }
    // fold the eight 8-lane groups, then group 0 writes this wave's row of partial sums (same layout as above)
    auto fold = [&](float v) { v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    const int SL = 1 + (cross ? nb1 : 0);
    float* prow = partial + ((int64_t)w0 * NH + hd) * SL * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s0 = fold(dwacc[e]);
        if (sub == 0) prow[8 * g + e] = s0;
    }
    if (cross) {
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s0 = fold(dbacc[a][e]);
                if (a < nb1 && sub == 0) prow[(1 + a) * 64 + 8 * g + e] = s0;
            }
    }
}

// dwq[c] += sum over (row, q head) of slot 0; dwk likewise over the k heads; dbias_x[a][head*64 + c] += sum over rows of
// slot 1 + a.  One workgroup per 32 output columns, 32 row groups, fixed summation order (deterministic).
// QKR_CW output columns per workgroup, 1024 / QKR_CW row groups: with 32 columns per workgroup the self-attention call
// (128 output columns) ran on FOUR workgroups, every thread walking ~170 dependent-latency loads: 54 us per call at every
// batch size, 0.65 ms per step.  8 columns per workgroup = 16 (self) / 400 (cross) workgroups, ~43 / 7 loads per thread;
// 32-byte row segments out of a 2-MB, L2-resident buffer.  Fixed summation order (deterministic).
constexpr int QKR_CW = 8;
constexpr int QKR_RG = 1024 / QKR_CW;
__global__ void __launch_bounds__(1024)
qknorm_partial_reduce_kernel(const float* __restrict__ partial, int n_rows, int nq, int nkv, int cross, int nb1,
                             float* __restrict__ dwq, float* __restrict__ dwk, float* __restrict__ dbias_q,
                             float* __restrict__ dbias_k, float* __restrict__ dbias_v) {
    __shared__ float sh[16][QKR_CW];
    const int cl = threadIdx.x & (QKR_CW - 1), rg = threadIdx.x / QKR_CW;
    const int NH = nq + nkv + (cross ? nkv : 0);
    const int SL = 1 + (cross ? nb1 : 0);
    const int64_t row_stride = (int64_t)NH * SL * 64;
    const int col = blockIdx.x * QKR_CW + cl;        // [0,64): dwq, [64,128): dwk, then bias columns (head, a, c)
    int h0, h1, slot, c;
    float* dst;
    if (col < 64) { h0 = 0; h1 = nq; slot = 0; c = col; dst = dwq + c; }
    else if (col < 128) { h0 = nq; h1 = nq + nkv; slot = 0; c = col - 64; dst = dwk + c; }
    else {
        const int b = col - 128;
        const int hd = b / (nb1 * 64), a = (b / 64) % nb1;
        c = b % 64; h0 = hd; h1 = hd + 1; slot = 1 + a;
        if (hd < nq) dst = dbias_q + (int64_t)a * nq * 64 + hd * 64 + c;
        else if (hd < nq + nkv) dst = dbias_k + (int64_t)a * nkv * 64 + (hd - nq) * 64 + c;
        else dst = dbias_v + (int64_t)a * nkv * 64 + (hd - nq - nkv) * 64 + c;
    }
    const int nh = h1 - h0;
    const int64_t n_items = (int64_t)n_rows * nh;     // (row, head) pairs to add up
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto at = [&](int64_t it) { return partial[(it / nh) * row_stride + ((int64_t)(h0 + it % nh) * SL + slot) * 64 + c]; };
    int64_t it = rg;
    for (; it + 3 * QKR_RG < n_items; it += 4 * QKR_RG) {
        s0 += at(it); s1 += at(it + QKR_RG); s2 += at(it + 2 * QKR_RG); s3 += at(it + 3 * QKR_RG);
    }
    for (; it < n_items; it += QKR_RG) s0 += at(it);
    // a wave holds 8 row groups of the same 8 columns: fold them with three shuffles, then the 16 waves through LDS
    float t = (s0 + s1) + (s2 + s3);
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < QKR_CW) sh[w][lane] = t;
    __syncthreads();
    if (threadIdx.x < QKR_CW) {
        float tot = 0.f;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) tot += sh[gi][threadIdx.x];
        *dst += tot;
    }
}

// ---------------------------------------------------------------------------------------------
// residual + dropout, SwiGLU, SiLU gate
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EW_THREADS)
residual_dropout_fwd_kernel(const float4* x_in, const float4* __restrict__ delta,
                            const int32_t* __restrict__ src_rows, int T, int H4, float p, uint64_t seed,
                            float4* x_out) {
    const DropoutRng rng(p, seed);
    const int64_t total = (int64_t)T * H4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int t = (int)(i / H4);
        const int c = (int)(i % H4);
        const int64_t r = src_rows ? src_rows[t] : t;
        const float4 d = delta[r * H4 + c];
        float4 v = x_in[i];
        float m[4];
        rng.mult4((uint32_t)i, m);
        v.x += m[0] * d.x; v.y += m[1] * d.y; v.z += m[2] * d.z; v.w += m[3] * d.w;
        x_out[i] = v;
    }
}

__global__ void __launch_bounds__(EW_THREADS)
residual_dropout_bwd_kernel(const float4* __restrict__ dx, const int32_t* __restrict__ src_rows, int T, int H4,
                            float p, uint64_t seed, float4* __restrict__ ddelta) {
    const DropoutRng rng(p, seed);
    const int64_t total = (int64_t)T * H4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int t = (int)(i / H4);
        const int c = (int)(i % H4);
        const int64_t r = src_rows ? src_rows[t] : t;
        const float4 g = dx[i];
        float m[4];
        rng.mult4((uint32_t)i, m);
        float4 o;
        o.x = m[0] * g.x; o.y = m[1] * g.y; o.z = m[2] * g.z; o.w = m[3] * g.w;
        ddelta[r * H4 + c] = o;
    }
}

template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
swiglu_fwd_kernel(const TA* __restrict__ g, const TA* __restrict__ u, int64_t n4, float p, uint64_t seed,
                  TA* __restrict__ hm, uint32_t* __restrict__ amax_out) {
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const DropoutRng rng(p, seed);
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const float4 a = ld4(g + 4 * i), b = ld4(u + 4 * i);
        float m[4];
        rng.mult4((uint32_t)i, m);
        float4 o;
        o.x = m[0] * (silu_f(a.x) * b.x);
        o.y = m[1] * (silu_f(a.y) * b.y);
        o.z = m[2] * (silu_f(a.z) * b.z);
        o.w = m[3] * (silu_f(a.w) * b.w);
        st4(hm + 4 * i, o);
        am = amax_f4(am, o);
    }
    amax_block_commit(am, amax_out, amax_lds);
}

template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
swiglu_bwd_kernel(TA* __restrict__ g, TA* __restrict__ u, const TA* __restrict__ dhm, int64_t n4,
                  float p, uint64_t seed, uint32_t* __restrict__ amax_g, uint32_t* __restrict__ amax_u) {
    __shared__ uint32_t amax_lds[2][4];
    uint32_t amg = 0, amu = 0;
    const DropoutRng rng(p, seed);
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const float4 a = ld4(g + 4 * i), b = ld4(u + 4 * i), d = ld4(dhm + 4 * i);
        float m[4];
        rng.mult4((uint32_t)i, m);
        const float d0 = m[0] * d.x, d1 = m[1] * d.y, d2 = m[2] * d.z, d3 = m[3] * d.w;
        float4 dg, du;
        dg.x = d0 * b.x * dsilu_f(a.x); du.x = d0 * silu_f(a.x);
        dg.y = d1 * b.y * dsilu_f(a.y); du.y = d1 * silu_f(a.y);
        dg.z = d2 * b.z * dsilu_f(a.z); du.z = d2 * silu_f(a.z);
        dg.w = d3 * b.w * dsilu_f(a.w); du.w = d3 * silu_f(a.w);
        st4(g + 4 * i, dg);
        st4(u + 4 * i, du);
        amg = amax_f4(amg, dg);
        amu = amax_f4(amu, du);
    }
    amax_block_commit(amg, amax_g, amax_lds[0]);
    amax_block_commit(amu, amax_u, amax_lds[1]);
}

// The same two kernels on the output of a FUSED gate|up projection: gu is [T][ld] with the gate in columns 0 .. I - 1 and the up
// projection in columns I .. 2 I - 1 of every row (ld >= 2 I); hm / dhm are [T][I] contiguous.  One wave per row; the dropout
// word of element (t, c) is that of the flat index t * I + c, as in the contiguous form (same masks).
template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
swiglu_fwd_ld_kernel(const TA* __restrict__ gu, int64_t ld, int T, int I4, float p, uint64_t seed, TA* __restrict__ hm,
                     uint32_t* __restrict__ amax_out, const float* __restrict__ tbl, const int32_t* __restrict__ row_group) {
    // tbl != nullptr (fp32 only): row t's gate | up values are gu[t] + tbl[row_group[t]] ([groups][2 I]: the share of the projection
    // that depends on the row's group only - gamer_inject_table_fwd); the sums are used, not stored
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const DropoutRng rng(p, seed);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6, nwaves = (gridDim.x * EW_THREADS) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        const TA* row = gu + (int64_t)t * ld;
        const float* trow = tbl ? tbl + (int64_t)row_group[t] * 8 * I4 : nullptr;
        for (int c = lane; c < I4; c += 64) {
            float4 a = ld4(row + 4 * c), b = ld4(row + 4 * (I4 + c));
            if (trow) {
                const float4 ta = *reinterpret_cast<const float4*>(trow + 4 * c), tb = *reinterpret_cast<const float4*>(trow + 4 * (I4 + c));
                a.x += ta.x; a.y += ta.y; a.z += ta.z; a.w += ta.w; b.x += tb.x; b.y += tb.y; b.z += tb.z; b.w += tb.w;
            }
            const int64_t i = (int64_t)t * I4 + c;
            float m[4];
            rng.mult4((uint32_t)i, m);
            float4 o;
            o.x = m[0] * (silu_f(a.x) * b.x);
            o.y = m[1] * (silu_f(a.y) * b.y);
            o.z = m[2] * (silu_f(a.z) * b.z);
            o.w = m[3] * (silu_f(a.w) * b.w);
            st4(hm + 4 * i, o);
            am = amax_f4(am, o);
        }
    }
    amax_block_commit(am, amax_out, amax_lds);
}

template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
swiglu_bwd_ld_kernel(TA* __restrict__ gu, int64_t ld, int T, int I4, const TA* __restrict__ dhm, float p, uint64_t seed,
                     uint32_t* __restrict__ amax_g, uint32_t* __restrict__ amax_u, const float* __restrict__ tbl,
                     const int32_t* __restrict__ row_group) {
    __shared__ uint32_t amax_lds[2][4];
    uint32_t amg = 0, amu = 0;
    const DropoutRng rng(p, seed);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6, nwaves = (gridDim.x * EW_THREADS) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        TA* row = gu + (int64_t)t * ld;
        const float* trow = tbl ? tbl + (int64_t)row_group[t] * 8 * I4 : nullptr;
        for (int c = lane; c < I4; c += 64) {
            const int64_t i = (int64_t)t * I4 + c;
            float4 a = ld4(row + 4 * c), b = ld4(row + 4 * (I4 + c));
            const float4 d = ld4(dhm + 4 * i);
            if (trow) {
                const float4 ta = *reinterpret_cast<const float4*>(trow + 4 * c), tb = *reinterpret_cast<const float4*>(trow + 4 * (I4 + c));
                a.x += ta.x; a.y += ta.y; a.z += ta.z; a.w += ta.w; b.x += tb.x; b.y += tb.y; b.z += tb.z; b.w += tb.w;
            }
            float m[4];
            rng.mult4((uint32_t)i, m);
            const float d0 = m[0] * d.x, d1 = m[1] * d.y, d2 = m[2] * d.z, d3 = m[3] * d.w;
            float4 dg, du;
            dg.x = d0 * b.x * dsilu_f(a.x); du.x = d0 * silu_f(a.x);
            dg.y = d1 * b.y * dsilu_f(a.y); du.y = d1 * silu_f(a.y);
            dg.z = d2 * b.z * dsilu_f(a.z); du.z = d2 * silu_f(a.z);
            dg.w = d3 * b.w * dsilu_f(a.w); du.w = d3 * silu_f(a.w);
            st4(row + 4 * c, dg);
            st4(row + 4 * (I4 + c), du);
            amg = amax_f4(amg, dg);
            amu = amax_f4(amu, du);
        }
    }
    amax_block_commit(amg, amax_g, amax_lds[0]);
    amax_block_commit(amu, amax_u, amax_lds[1]);
}

// TO = type of `out`: the residual stream (fp32) when resid is given, otherwise the activation type
template <typename TA, typename TO>
__global__ void __launch_bounds__(EW_THREADS)
silu_gate_fwd_kernel(const TA* __restrict__ a, const TA* __restrict__ gate, int64_t n4,
                     TO* __restrict__ out, const float4* __restrict__ resid, float p, uint64_t seed) {
    // resid != nullptr: out = resid + dropout(a * silu(gate)) - the residual add of the cross-attention block fused
    // in (same mask as gamer_residual_dropout_fwd with this seed: element index -> mult4)
    const DropoutRng rng(resid ? p : 0.f, seed);
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const float4 x = ld4(a + 4 * i), g = ld4(gate + 4 * i);
        float4 o;
        o.x = x.x * silu_f(g.x); o.y = x.y * silu_f(g.y); o.z = x.z * silu_f(g.z); o.w = x.w * silu_f(g.w);
        if (resid) {
            const float4 r = resid[i];
            float m[4];
            rng.mult4((uint32_t)i, m);
            o.x = r.x + m[0] * o.x; o.y = r.y + m[1] * o.y; o.z = r.z + m[2] * o.z; o.w = r.w + m[3] * o.w;
        }
        st4(out + 4 * i, o);
    }
}

template <typename TA>
__global__ void __launch_bounds__(EW_THREADS)
silu_gate_bwd_kernel(const TA* __restrict__ a, const TA* __restrict__ gate, const float4* __restrict__ dout,
                     int64_t n4, TA* __restrict__ da, TA* __restrict__ dgate, float p, uint64_t seed,
                     uint32_t* __restrict__ amax_a, uint32_t* __restrict__ amax_gate) {
    // p > 0: dout is the gradient of the residual stream; the dropout mask of the fused forward is applied first
    __shared__ uint32_t amax_lds[2][4];
    uint32_t ama = 0, amg = 0;
    const DropoutRng rng(p, seed);
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const float4 x = ld4(a + 4 * i), g = ld4(gate + 4 * i);
        float4 d = dout[i];
        float m[4];
        rng.mult4((uint32_t)i, m);
        d.x *= m[0]; d.y *= m[1]; d.z *= m[2]; d.w *= m[3];
        float4 oa, og;
        oa.x = d.x * silu_f(g.x); og.x = d.x * x.x * dsilu_f(g.x);
        oa.y = d.y * silu_f(g.y); og.y = d.y * x.y * dsilu_f(g.y);
        oa.z = d.z * silu_f(g.z); og.z = d.z * x.z * dsilu_f(g.z);
        oa.w = d.w * silu_f(g.w); og.w = d.w * x.w * dsilu_f(g.w);
        st4(da + 4 * i, oa);
        st4(dgate + 4 * i, og);
        ama = amax_f4(ama, oa);
        amg = amax_f4(amg, og);
    }
    amax_block_commit(ama, amax_a, amax_lds[0]);
    amax_block_commit(amg, amax_gate, amax_lds[1]);
}

// ---------------------------------------------------------------------------------------------
// temperature + shifted cross entropy; one wave per row of logits (TL = fp32, or bf16 as the reference's autocast
// lm_head produces them; the statistics are fp32 either way, loss_utils.py upcasts)
// ---------------------------------------------------------------------------------------------
constexpr int CE_MAXC = 20;              // register-resident rows up to V = 1280 (the shipped vocabulary is 1041)
template <typename T> constexpr int CE_EPL = 16 / (int)sizeof(T);      // elements per 16 bytes
// ... as 16-byte groups per lane: five cover 1280 fp32 values, three 1536 bf16 values (five for bf16 too meant 40 exponentials per
// lane for the 17 values it owns at V = 1041)
template <typename T> constexpr int CE_MAXQ = sizeof(T) == 2 ? 3 : 5;
__device__ __forceinline__ void ce_load16(const float* p, float (&x)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}
__device__ __forceinline__ void ce_store16(float* p, const float (&x)[4]) { *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ void ce_load16(const bf16_t* p, float (&x)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
}
__device__ __forceinline__ void ce_store16(bf16_t* p, const float (&x)[8]) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)x[e];
    *reinterpret_cast<bf16x8*>(p) = v;
}
template <typename TL>
__global__ void __launch_bounds__(EW_THREADS)
ce_fwd_kernel(TL* __restrict__ logits, int ldl, const int64_t* __restrict__ labels, int T, int S, int V,
              float inv_temp, int ignore_index, float* __restrict__ lse_out, float* __restrict__ row_loss, int vec4) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * EW_THREADS) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        TL* row = logits + (int64_t)t * ldl;
        const int s = t % S;
        const int64_t tgt = (s + 1 < S) ? labels[t + 1] : (int64_t)ignore_index;
        const bool valid = tgt != ignore_index && tgt >= 0 && tgt < V;
        float mx = -INFINITY;
        float zt = 0.f;
        if (vec4 && V <= 64 * CE_EPL<TL> * CE_MAXQ<TL>) {
            // rows with 16-byte aligned starts: the row in registers as 16-byte groups per lane (four fp32 / eight bf16 values; one wave
            // instruction moves 1 KB instead of 256 / 128 bytes: fp32 3.6 -> 5 TB/s, bf16 2.2 -> ...); the group that straddles V is
            // stored element by element - the padding columns of the row stay as they are
            constexpr int EPL = CE_EPL<TL>;
            float z[CE_MAXQ<TL>][EPL];
            const int c_last = ((V - 1) / EPL) * EPL;      // every load is unconditional (a group past the row reads the row's last group and
#pragma unroll                                            // is masked): loads under `if (c < V)` were each waited for right behind the load
            for (int i = 0; i < CE_MAXQ<TL>; ++i) {
                const int c = EPL * (lane + 64 * i);
                float x[EPL];
                ce_load16(row + min(c, c_last), x);
#pragma unroll
                for (int e = 0; e < EPL; ++e) z[i][e] = c + e < V ? round_as<TL>(x[e] * inv_temp) : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < CE_MAXQ<TL>; ++i) {
                const int c = EPL * (lane + 64 * i);
                if (inv_temp == 1.f) {
                    // the producer already divided by the temperature (the head GEMM's alpha): nothing to write back
                } else if (c + EPL <= V) {
                    ce_store16(row + c, z[i]);
                } else if (c < V) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e)
                        if (c + e < V) st1(row + c + e, z[i][e]);
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    mx = fmaxf(mx, z[i][e]);
                    if (valid && c + e == (int)tgt) zt = z[i][e];
                }
            }
            mx = wave_max(mx);
            zt = wave_sum(zt);
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < CE_MAXQ<TL>; ++i)
#pragma unroll
                for (int e = 0; e < EPL; ++e) se += expf(z[i][e] - mx);           // exp(-inf) = 0 past the row
            se = wave_sum(se);
            const float lse = mx + logf(se);
            if (lane == 0) {
                lse_out[t] = lse;
                row_loss[t] = valid ? (lse - zt) : 0.f;
            }
            continue;
        }
        if (V <= 64 * CE_MAXC) {
            // the whole row in registers: one read (all loads in flight together), one write, no second pass
            float z[CE_MAXC];
#pragma unroll
            for (int i = 0; i < CE_MAXC; ++i) {
                const int c = lane + 64 * i;
                z[i] = c < V ? round_as<TL>(ld1(row + c) * inv_temp) : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < CE_MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < V) st1(row + c, z[i]);
                mx = fmaxf(mx, z[i]);
                if (valid && c == (int)tgt) zt = z[i];
            }
            mx = wave_max(mx);
            zt = wave_sum(zt);
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < CE_MAXC; ++i) se += expf(z[i] - mx);           // exp(-inf) = 0 past the row
            se = wave_sum(se);
            const float lse = mx + logf(se);
            if (lane == 0) {
                lse_out[t] = lse;
                row_loss[t] = valid ? (lse - zt) : 0.f;
            }
            continue;
        }
        for (int c = lane; c < V; c += 64) {
            const float z = round_as<TL>(ld1(row + c) * inv_temp);
            st1(row + c, z);
            mx = fmaxf(mx, z);
            if (valid && c == (int)tgt) zt = z;
        }
        mx = wave_max(mx);
        zt = wave_sum(zt);
        float se = 0.f;
        for (int c = lane; c < V; c += 64) se += expf(ld1(row + c) - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        if (lane == 0) {
            lse_out[t] = lse;
            row_loss[t] = valid ? (lse - zt) : 0.f;
        }
    }
}

// deterministic single-workgroup reduction: out[0] = sum(row_loss), out[1] = #valid targets
__global__ void __launch_bounds__(1024)
ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, int T, int S, int V,
                 int ignore_index, float* __restrict__ loss_sum, float* __restrict__ count) {
    __shared__ float sh_l[1024];
    __shared__ float sh_c[1024];
    float l = 0.f, c = 0.f;
    // eight independent loads in flight per thread (one workgroup: the loop is latency-bound, 0.30 ms at T = 517k
    // with one load at a time); the summation order stays fixed
    int t = threadIdx.x;
    for (; t + 7 * 1024 < T; t += 8 * 1024) {
        float lv[8];
        int64_t tg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + u * 1024;
            lv[u] = row_loss[tt];
            tg[u] = (tt % S) + 1 < S ? labels[tt + 1] : (int64_t)ignore_index;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            l += lv[u];
            if (tg[u] != ignore_index && tg[u] >= 0 && tg[u] < V) c += 1.f;
        }
    }
    for (; t < T; t += 1024) {
        l += row_loss[t];
        const int s = t % S;
        if (s + 1 < S) {
            const int64_t tgt = labels[t + 1];
            if (tgt != ignore_index && tgt >= 0 && tgt < V) c += 1.f;
        }
    }
    sh_l[threadIdx.x] = l;
    sh_c[threadIdx.x] = c;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            sh_l[threadIdx.x] += sh_l[threadIdx.x + off];
            sh_c[threadIdx.x] += sh_c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_sum[0] = sh_l[0];
        count[0] = sh_c[0];
    }
}

template <typename TL>
__global__ void __launch_bounds__(EW_THREADS)
ce_bwd_kernel(TL* __restrict__ logits, int ldl, const int64_t* __restrict__ labels, int T, int S, int V,
              int ignore_index, const float* __restrict__ lse, const float* __restrict__ count_dev,
              float denom_host, float dloss_over_temp, const float* __restrict__ dloss_dev, uint32_t* __restrict__ amax_out,
              int vec4) {
    __shared__ uint32_t amax_lds[4];
    uint32_t am = 0;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * EW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * EW_THREADS) >> 6;
    const float denom = count_dev ? count_dev[0] : denom_host;
    const float gs = dloss_over_temp * (dloss_dev ? dloss_dev[0] : 1.f) / denom;
    for (int t = wave; t < T; t += nwaves) {
        TL* row = logits + (int64_t)t * ldl;
        const int s = t % S;
        const int64_t tgt = (s + 1 < S) ? labels[t + 1] : (int64_t)ignore_index;
        const bool valid = tgt != ignore_index && tgt >= 0 && tgt < V;
        const float l = lse[t];
        if (vec4 && V <= 64 * CE_EPL<TL> * CE_MAXQ<TL>) {       // 16 bytes per lane, the row's loads all in flight (see ce_fwd_kernel)
            constexpr int EPL = CE_EPL<TL>;
            const int c_last = ((V - 1) / EPL) * EPL;
            float xs[CE_MAXQ<TL>][EPL];
#pragma unroll
            for (int i = 0; i < CE_MAXQ<TL>; ++i) ce_load16(row + min(EPL * (lane + 64 * i), c_last), xs[i]);
#pragma unroll
            for (int i = 0; i < CE_MAXQ<TL>; ++i) {
                const int c = EPL * (lane + 64 * i);
                if (c >= V) continue;
                float g[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    float v = (valid && c + e < V) ? expf(xs[i][e] - l) : 0.f;
                    if (valid && c + e == (int)tgt) v -= 1.f;
                    g[e] = (valid && c + e < V) ? v * gs : 0.f;
                }
                if (c + EPL <= V) {
                    ce_store16(row + c, g);
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e)
                        if (c + e < V) st1(row + c + e, g[e]);
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) am = __float_as_uint(fmaxf(__uint_as_float(am), fabsf(g[e])));
            }
            continue;
        }
        for (int c = lane; c < V; c += 64) {
            float g = 0.f;
            if (valid) {
                g = expf(ld1(row + c) - l);
                if (c == (int)tgt) g -= 1.f;
                g *= gs;
            }
            st1(row + c, g);
            am = __float_as_uint(fmaxf(__uint_as_float(am), fabsf(g)));
        }
    }
    amax_block_commit(am, amax_out, amax_lds);
}

__global__ void __launch_bounds__(EW_THREADS)
fill_kernel(float* __restrict__ p, int64_t n, float value) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        p[i] = value;
}

}  // namespace gamer

namespace gamer {
static thread_local AmaxSink g_amax_sink = {{nullptr, nullptr, nullptr}};
AmaxSink take_amax_sink() {
    const AmaxSink s = g_amax_sink;
    g_amax_sink.out[0] = g_amax_sink.out[1] = g_amax_sink.out[2] = nullptr;
    return s;
}
}  // namespace gamer

using namespace gamer;

#define ST(s) ((hipStream_t)(s))

extern "C" int gamer_amax_sink(uint32_t* out0, uint32_t* out1) {
    g_amax_sink.out[0] = out0;
    g_amax_sink.out[1] = out1;
    g_amax_sink.out[2] = nullptr;
    return 0;
}
extern "C" int gamer_amax_sink3(uint32_t* out0, uint32_t* out1, uint32_t* out2) {
    g_amax_sink.out[0] = out0;
    g_amax_sink.out[1] = out1;
    g_amax_sink.out[2] = out2;
    return 0;
}

extern "C" int gamer_embedding_fwd(const int64_t* ids, const float* W, int V, int T, int H, float* x, void* stream) {
    GAMER_CHECK_ARG(ids && W && x, "gamer_embedding_fwd: null pointer");
    GAMER_CHECK_ARG(T > 0 && H > 0 && H % 4 == 0 && V > 0, "gamer_embedding_fwd: bad shape T=%d H=%d V=%d", T, H, V);
    GAMER_CHECK_ARG(aligned16(W) && aligned16(x), "gamer_embedding_fwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream),
                       ids, (const float4*)W, V, T, H / 4, (float4*)x);
    GAMER_CHECK_LAUNCH("gamer_embedding_fwd");
    return 0;
}

extern "C" int gamer_embedding_bwd(const int64_t* ids, const float* dx, int V, int T, int H, int pad_id, float* dW,
                                   void* stream) {
    GAMER_CHECK_ARG(ids && dx && dW, "gamer_embedding_bwd: null pointer");
    GAMER_CHECK_ARG(T > 0 && H > 0 && V > 0, "gamer_embedding_bwd: bad shape T=%d H=%d V=%d", T, H, V);
    GAMER_CHECK_ARG(H <= 1024, "gamer_embedding_bwd: H=%d > 1024 unsupported", H);
    // 1024-token chunks at T = 517k (505 workgroups); at the 8-GPU per-rank shape (T = 64,640) that would be 64 workgroups
    // walking 256 dependent iterations each (0.44 ms): chunks of >= 128 tokens sized for ~512 workgroups instead
    int chunk = EMB_CHUNK;
    while (chunk > 128 && (T + chunk - 1) / chunk < 512) chunk >>= 1;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((T + chunk - 1) / chunk), dim3(EW_THREADS),
                       (size_t)EMB_CACHE * H * sizeof(float), ST(stream), ids, dx, V, T, H, pad_id, dW, chunk);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd");
    return 0;
}

// bytes of the workspace gamer_embedding_bwd_ordered needs
extern "C" int64_t gamer_embedding_bwd_ordered_ws_bytes(int V, int T, int H) {
    if (V <= 0 || T <= 0 || H <= 0) return -1;
    const int64_t chunks = (T + EMB_SORT_CHUNK - 1) / EMB_SORT_CHUNK;
    const int64_t pieces = (int64_t)T / EMB_PIECE + V + 1;
    const int64_t ints = chunks * (V + 1) + 2 * (int64_t)(V + 1) + (V + 2) + T;
    return ((ints * 4 + 15) / 16) * 16 + pieces * H * 4;
}
extern "C" int gamer_embedding_bwd_ordered(const int64_t* ids, const float* dx, int V, int T, int H, int pad_id, float* dW,
                                           void* ws, int64_t ws_bytes, void* stream) {
    GAMER_CHECK_ARG(ids && dx && dW && ws, "gamer_embedding_bwd_ordered: null pointer");
    GAMER_CHECK_ARG(T > 0 && H > 0 && H % 4 == 0 && V > 0 && V <= EMB_MAX_V,
                    "gamer_embedding_bwd_ordered: bad shape T=%d H=%d V=%d (H %% 4 == 0, V <= %d)", T, H, V, EMB_MAX_V);
    GAMER_CHECK_ARG(aligned16(dx) && aligned16(dW) && aligned16(ws) && ws_bytes >= gamer_embedding_bwd_ordered_ws_bytes(V, T, H),
                    "gamer_embedding_bwd_ordered: dx / dW / ws must be 16-byte aligned and ws hold %lld bytes (got %lld)",
                    (long long)gamer_embedding_bwd_ordered_ws_bytes(V, T, H), (long long)ws_bytes);
    const int chunks = (T + EMB_SORT_CHUNK - 1) / EMB_SORT_CHUNK;
    const int pieces_max = T / EMB_PIECE + V + 1;
    int* ip = (int*)ws;
    EmbSortWs w;
    w.hist = ip; ip += (int64_t)chunks * (V + 1);
    w.row_start = ip; ip += V + 1;
    w.row_count = ip; ip += V + 1;
    w.piece_start = ip; ip += V + 2;
    w.perm = ip; ip += T;
    const int64_t ints = (int64_t)chunks * (V + 1) + 2 * (int64_t)(V + 1) + (V + 2) + T;
    w.partial = (float*)((char*)ws + ((ints * 4 + 15) / 16) * 16);
    hipStream_t st = ST(stream);
    const size_t lds = (size_t)(V + 1) * sizeof(int);
    hipLaunchKernelGGL(emb_hist_kernel, dim3(chunks), dim3(EW_THREADS), lds, st, ids, V, T, pad_id, w.hist);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/hist");
    hipLaunchKernelGGL(emb_chunk_scan_kernel, dim3(V + 1), dim3(64), 0, st, V, chunks, w.hist, w.row_count);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/chunk_scan");
    hipLaunchKernelGGL(emb_row_scan_kernel, dim3(1), dim3(1024), 0, st, V, w.row_count, w.row_start, w.piece_start);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/row_scan");
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(chunks), dim3(EW_THREADS), lds + EMB_SORT_CHUNK * sizeof(int), st, ids, V, T, pad_id,
                       w.hist, w.row_start, w.perm);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/scatter");
    hipLaunchKernelGGL(emb_piece_sum_kernel, dim3(pieces_max), dim3(EW_THREADS), 0, st, (const float4*)dx, V, H / 4, w.row_start,
                       w.row_count, w.piece_start, w.perm, (float4*)w.partial);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/pieces");
    hipLaunchKernelGGL(emb_row_sum_kernel, dim3(V), dim3(64), 0, st, H / 4, w.piece_start, (const float4*)w.partial, (float4*)dW);
    GAMER_CHECK_LAUNCH("gamer_embedding_bwd_ordered/rows");
    return 0;
}

template <typename TY>
static int rmsnorm_fwd_impl(const char* name, const float* x, const float* w, int T, int H, float eps,
                            const int32_t* dst_rows, TY* y, int ldy, void* stream) {
    GAMER_CHECK_ARG(x && w && y, "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && H > 0 && H % 4 == 0 && H <= 256 * RMS_MAXC_BIG && ldy >= H && ldy % 4 == 0,
                    "%s: bad shape T=%d H=%d ldy=%d (H%%4==0, H<=1024, ldy%%4==0)", name, T, H, ldy);
    GAMER_CHECK_ARG(aligned16(x) && aligned16(w) && aligned_vec4<TY>(y), "%s: pointers must be 16-byte aligned", name);
    const AmaxSink sink = take_amax_sink();
    if (H <= 256)
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<TY, 1>), dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream),
                           (const float4*)x, (const float4*)w, T, H / 4, 1.f / (float)H, eps, dst_rows, y, ldy, sink.out[0]);
    else
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<TY, RMS_MAXC_BIG>), dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream),
                           (const float4*)x, (const float4*)w, T, H / 4, 1.f / (float)H, eps, dst_rows, y, ldy, sink.out[0]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_rmsnorm_fwd(const float* x, const float* w, int T, int H, float eps, const int32_t* dst_rows,
                                 float* y, int ldy, void* stream) {
    return rmsnorm_fwd_impl<float>("gamer_rmsnorm_fwd", x, w, T, H, eps, dst_rows, y, ldy, stream);
}
extern "C" int gamer_rmsnorm_fwd_bf16(const float* x, const float* w, int T, int H, float eps, const int32_t* dst_rows,
                                      gamer_bf16* y, int ldy, void* stream) {
    return rmsnorm_fwd_impl<bf16_t>("gamer_rmsnorm_fwd_bf16", x, w, T, H, eps, dst_rows, (bf16_t*)y, ldy, stream);
}

template <typename TG>
static int rmsnorm_bwd_impl(const char* name, const float* x, const float* w, const TG* dy, int lddy,
                            const int32_t* dy_rows, int T, int H, float eps, int accumulate_dx, float* dx,
                            float* dw_partial, int n_partial, TG* mask_out, const int32_t* mask_rows, float p_drop,
                            uint64_t seed, void* stream) {
    GAMER_CHECK_ARG(x && w && dy && dx && dw_partial, "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && H > 0 && H % 4 == 0 && H <= 256 * RMS_MAXC_BIG && lddy >= H && lddy % 4 == 0 && n_partial > 0,
                    "%s: bad shape T=%d H=%d lddy=%d n_partial=%d", name, T, H, lddy, n_partial);
    GAMER_CHECK_ARG(aligned16(x) && aligned16(w) && aligned_vec4<TG>(dy) && aligned16(dx) && aligned16(dw_partial) &&
                    aligned_vec4<TG>(mask_out), "%s: pointers must be 16-byte aligned", name);
    const AmaxSink sink = take_amax_sink();
    if (H <= 256)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<TG, 1>), dim3(n_partial), dim3(EW_THREADS), 0, ST(stream),
                           (const float4*)x, (const float4*)w, dy, lddy, dy_rows, T, H / 4, 1.f / (float)H, eps,
                           accumulate_dx, (float4*)dx, dw_partial, mask_out, mask_rows, p_drop, seed, sink.out[0]);
    else
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<TG, RMS_MAXC_BIG>), dim3(n_partial), dim3(EW_THREADS), 0, ST(stream),
                           (const float4*)x, (const float4*)w, dy, lddy, dy_rows, T, H / 4, 1.f / (float)H, eps,
                           accumulate_dx, (float4*)dx, dw_partial, mask_out, mask_rows, p_drop, seed, sink.out[0]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_rmsnorm_bwd(const float* x, const float* w, const float* dy, int lddy, const int32_t* dy_rows,
                                 int T, int H, float eps, int accumulate_dx, float* dx, float* dw_partial,
                                 int n_partial, float* mask_out, const int32_t* mask_rows, float p_drop,
                                 uint64_t seed, void* stream) {
    return rmsnorm_bwd_impl<float>("gamer_rmsnorm_bwd", x, w, dy, lddy, dy_rows, T, H, eps, accumulate_dx, dx, dw_partial,
                                   n_partial, mask_out, mask_rows, p_drop, seed, stream);
}
extern "C" int gamer_rmsnorm_bwd_bf16(const float* x, const float* w, const gamer_bf16* dy, int lddy,
                                      const int32_t* dy_rows, int T, int H, float eps, int accumulate_dx, float* dx,
                                      float* dw_partial, int n_partial, gamer_bf16* mask_out, const int32_t* mask_rows,
                                      float p_drop, uint64_t seed, void* stream) {
    return rmsnorm_bwd_impl<bf16_t>("gamer_rmsnorm_bwd_bf16", x, w, (const bf16_t*)dy, lddy, dy_rows, T, H, eps,
                                    accumulate_dx, dx, dw_partial, n_partial, (bf16_t*)mask_out, mask_rows, p_drop, seed,
                                    stream);
}

extern "C" int gamer_colsum_reduce(const float* partial, int rows, int cols, int accumulate, float* out, void* stream) {
    GAMER_CHECK_ARG(partial && out && rows > 0 && cols > 0, "gamer_colsum_reduce: bad arguments");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((cols + 31) / 32), dim3(32 * COLSUM_RG), 0, ST(stream), partial, rows,
                       cols, accumulate, out);
    GAMER_CHECK_LAUNCH("gamer_colsum_reduce");
    return 0;
}

extern "C" int gamer_colsum_reduce_batched(const float* partial, int64_t stride, int rows, int cols, int n, float* const* outs,
                                           int accumulate, void* stream) {
    GAMER_CHECK_ARG(partial && outs && rows > 0 && cols > 0 && n > 0 && n <= 65535 && stride >= (int64_t)rows * cols,
                    "gamer_colsum_reduce_batched: bad arguments (rows=%d cols=%d n=%d stride=%lld)", rows, cols, n, (long long)stride);
    hipLaunchKernelGGL(colsum_reduce_batched_kernel, dim3((cols + 31) / 32, n), dim3(32 * COLSUM_RG), 0, ST(stream), partial, stride,
                       rows, cols, accumulate, outs);
    GAMER_CHECK_LAUNCH("gamer_colsum_reduce_batched");
    return 0;
}

template <typename TY>
static int rowtable_fwd_impl(const char* name, const float* table, const int32_t* idx, const int32_t* dst_rows, int T,
                             int E, TY* y, int ldy, int col0, void* stream) {
    GAMER_CHECK_ARG(table && idx && y, "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && E > 0 && E % 4 == 0 && ldy % 4 == 0 && col0 % 4 == 0 && col0 + E <= ldy,
                    "%s: bad shape T=%d E=%d ldy=%d col0=%d", name, T, E, ldy, col0);
    hipLaunchKernelGGL(rowtable_fwd_kernel<TY>, dim3(grid_for_threads((int64_t)T * E / 4)), dim3(EW_THREADS), 0,
                       ST(stream), (const float4*)table, idx, dst_rows, T, E / 4, y, ldy, col0, take_amax_sink().out[0]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_rowtable_fwd(const float* table, const int32_t* idx, const int32_t* dst_rows, int T, int E,
                                  float* y, int ldy, int col0, void* stream) {
    return rowtable_fwd_impl<float>("gamer_rowtable_fwd", table, idx, dst_rows, T, E, y, ldy, col0, stream);
}
extern "C" int gamer_rowtable_fwd_bf16(const float* table, const int32_t* idx, const int32_t* dst_rows, int T, int E,
                                       gamer_bf16* y, int ldy, int col0, void* stream) {
    return rowtable_fwd_impl<bf16_t>("gamer_rowtable_fwd_bf16", table, idx, dst_rows, T, E, (bf16_t*)y, ldy, col0, stream);
}

template <typename TG>
static int rowtable_bwd_impl(const char* name, const TG* dy, int lddy, int col0, const int32_t* idx,
                             const int32_t* dy_rows, int T, int E, int n_rows_table, float* dtable, float* partial,
                             int64_t partial_numel, void* stream) {
    GAMER_CHECK_ARG(dy && idx && dtable, "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && E > 0 && E % 4 == 0 && E / 4 <= EW_THREADS && lddy % 4 == 0 && col0 % 4 == 0 &&
                    n_rows_table > 0 && n_rows_table <= TBL_MAXROWS,
                    "%s: bad shape T=%d E=%d lddy=%d col0=%d rows=%d (rows<=8)", name, T, E, lddy, col0, n_rows_table);
    const int rows_per_block = EW_THREADS / (E / 4);
    int blocks = (T + rows_per_block - 1) / rows_per_block;
    if (blocks > 1024) blocks = 1024;
    if (partial) {
        const int64_t per_block = (int64_t)n_rows_table * E;
        if (blocks * per_block > partial_numel) blocks = (int)(partial_numel / per_block);
        GAMER_CHECK_ARG(blocks >= 1, "%s: partial holds %lld floats, one workgroup's table needs %lld", name, (long long)partial_numel,
                        (long long)per_block);
        const size_t lds = (size_t)rows_per_block * per_block * sizeof(float);
        GAMER_CHECK_ARG(lds <= 64 * 1024, "%s: the ordered form needs %zu bytes of LDS (E=%d, rows=%d)", name, lds, E, n_rows_table);
        hipLaunchKernelGGL((rowtable_bwd_kernel<TG, true>), dim3(blocks), dim3(EW_THREADS), lds, ST(stream), dy, lddy, col0, idx,
                           dy_rows, T, E / 4, n_rows_table, dtable, partial);
        GAMER_CHECK_LAUNCH(name);
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3(((int)per_block + 31) / 32), dim3(32 * COLSUM_RG), 0, ST(stream), partial, blocks,
                           (int)per_block, 1, dtable);
        GAMER_CHECK_LAUNCH(name);
        return 0;
    }
    hipLaunchKernelGGL((rowtable_bwd_kernel<TG, false>), dim3(blocks), dim3(EW_THREADS), (size_t)n_rows_table * E * sizeof(float),
                       ST(stream), dy, lddy, col0, idx, dy_rows, T, E / 4, n_rows_table, dtable, (float*)nullptr);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_rowtable_bwd(const float* dy, int lddy, int col0, const int32_t* idx, const int32_t* dy_rows,
                                  int T, int E, int n_rows_table, float* dtable, float* partial, int64_t partial_numel,
                                  void* stream) {
    return rowtable_bwd_impl<float>("gamer_rowtable_bwd", dy, lddy, col0, idx, dy_rows, T, E, n_rows_table, dtable, partial,
                                    partial_numel, stream);
}
extern "C" int gamer_rowtable_bwd_bf16(const gamer_bf16* dy, int lddy, int col0, const int32_t* idx,
                                       const int32_t* dy_rows, int T, int E, int n_rows_table, float* dtable,
                                       float* partial, int64_t partial_numel, void* stream) {
    return rowtable_bwd_impl<bf16_t>("gamer_rowtable_bwd_bf16", (const bf16_t*)dy, lddy, col0, idx, dy_rows, T, E,
                                     n_rows_table, dtable, partial, partial_numel, stream);
}

template <typename TA>
static int qknorm_rope_fwd_impl(const char* name, TA* qkv, int T, int S, int nq, int nkv, const float* wq,
                                const float* wk, float eps, const float* cos_t, const float* sin_t, const float* bias_q,
                                const float* bias_k, const float* bias_v, const int32_t* act_idx, TA* q_rot, TA* k_rot,
                                const int32_t* pos_ids, void* stream) {
    GAMER_CHECK_ARG(qkv && wq && wk && cos_t && sin_t && q_rot && k_rot, "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && S > 0 && nq > 0 && nkv > 0 && T % S == 0, "%s: bad shape T=%d S=%d nq=%d nkv=%d", name, T, S, nq, nkv);
    const bool cross = bias_q != nullptr;
    GAMER_CHECK_ARG(!cross || (bias_k && bias_v && act_idx), "%s: cross needs bias_k, bias_v, act_idx", name);
    const int NH = nq + nkv + (cross ? nkv : 0);
    static const bool row_major = getenv("GAMER_QKNORM_ROW_MAJOR") != nullptr;           // (the round-1 forms, kept for A/B runs)
    if constexpr (sizeof(TA) == 2) {
        if (row_major)
            hipLaunchKernelGGL(qknorm_rope_fwd_b8_kernel, dim3(grid_for_waves(((int64_t)T * NH + 7) / 8)), dim3(EW_THREADS), 0,
                               ST(stream), qkv, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q, bias_k, bias_v, act_idx, q_rot,
                               k_rot, pos_ids);
        else
            hipLaunchKernelGGL(qknorm_rope_fwd_b8_tok_kernel, dim3(grid_for_waves(((int64_t)T + 7) / 8)), dim3(EW_THREADS), 0,
                               ST(stream), qkv, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q, bias_k, bias_v, act_idx, q_rot,
                               k_rot, pos_ids);
    } else {
        const AmaxSink sink = take_amax_sink();
        if (row_major)
            hipLaunchKernelGGL(qknorm_rope_fwd_kernel<TA>, dim3(grid_for_waves(((int64_t)T * NH + 3) / 4)), dim3(EW_THREADS), 0,
                               ST(stream), qkv, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q, bias_k, bias_v, act_idx, q_rot,
                               k_rot, pos_ids, sink.out[0], sink.out[1], sink.out[2]);
        else
            hipLaunchKernelGGL(qknorm_rope_fwd_tok_kernel, dim3(grid_for_waves(((int64_t)T + 3) / 4)), dim3(EW_THREADS), 0,
                               ST(stream), qkv, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q, bias_k, bias_v, act_idx, q_rot,
                               k_rot, pos_ids, sink.out[0], sink.out[1], sink.out[2]);
    }
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_qknorm_rope_fwd(float* qkv, int T, int S, int nq, int nkv, const float* wq, const float* wk,
                                     float eps, const float* cos_t, const float* sin_t, const float* bias_q,
                                     const float* bias_k, const float* bias_v, const int32_t* act_idx, float* q_rot,
                                     float* k_rot, const int32_t* pos_ids, void* stream) {
    return qknorm_rope_fwd_impl<float>("gamer_qknorm_rope_fwd", qkv, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q,
                                       bias_k, bias_v, act_idx, q_rot, k_rot, pos_ids, stream);
}
extern "C" int gamer_qknorm_rope_fwd_bf16(gamer_bf16* qkv, int T, int S, int nq, int nkv, const float* wq,
                                          const float* wk, float eps, const float* cos_t, const float* sin_t,
                                          const float* bias_q, const float* bias_k, const float* bias_v,
                                          const int32_t* act_idx, gamer_bf16* q_rot, gamer_bf16* k_rot,
                                          const int32_t* pos_ids, void* stream) {
    return qknorm_rope_fwd_impl<bf16_t>("gamer_qknorm_rope_fwd_bf16", (bf16_t*)qkv, T, S, nq, nkv, wq, wk, eps, cos_t,
                                        sin_t, bias_q, bias_k, bias_v, act_idx, (bf16_t*)q_rot, (bf16_t*)k_rot, pos_ids,
                                        stream);
}

template <typename TA>
static int qknorm_rope_bwd_impl(const char* name, const TA* qkv, const TA* dq_rot, const TA* dk_rot, int T, int S,
                                int nq, int nkv, const float* wq, const float* wk, float eps, const float* cos_t,
                                const float* sin_t, const float* bias_q, const float* bias_k, const int32_t* act_idx,
                                int nb1, TA* dqkv, float* dwq, float* dwk, float* dbias_q, float* dbias_k,
                                float* dbias_v, const int32_t* pos_ids, float* partial, int64_t partial_numel,
                                void* stream) {
    GAMER_CHECK_ARG(qkv && dq_rot && dk_rot && wq && wk && cos_t && sin_t && dqkv && dwq && dwk && partial,
                    "%s: null pointer", name);
    GAMER_CHECK_ARG(T > 0 && S > 0 && nq > 0 && nkv > 0 && T % S == 0, "%s: bad shape", name);
    const int cross = bias_q != nullptr ? 1 : 0;
    GAMER_CHECK_ARG(!cross || (bias_k && act_idx && dbias_q && dbias_k && dbias_v && nb1 > 0 && nb1 <= TBL_MAXROWS),
                    "%s: cross needs bias_k, act_idx, dbias_*, 0<nb1<=8 (nb1=%d)", name, nb1);
    const int NH = nq + nkv + (cross ? nkv : 0);
    const int SL = 1 + (cross ? nb1 : 0);
    constexpr int RPW = sizeof(TA) == 2 ? 8 : 4;        // token rows per wave iteration
    int waves_per_head = 8192 / NH;
    if (waves_per_head > (T + RPW - 1) / RPW) waves_per_head = (T + RPW - 1) / RPW;
    const int64_t cap = partial_numel / ((int64_t)NH * SL * 64);
    if (waves_per_head > cap) waves_per_head = (int)cap;
    GAMER_CHECK_ARG(waves_per_head >= 1 && aligned16(partial),
                    "%s: partial needs at least %d floats, 16-byte aligned (got %lld)", name, NH * SL * 64,
                    (long long)partial_numel);
    const int64_t total_waves = (int64_t)NH * waves_per_head;
    const int blocks = (int)((total_waves + EW_WAVES - 1) / EW_WAVES);
    if constexpr (sizeof(TA) == 2) {
        static const bool row_major16 = getenv("GAMER_QKNORM_ROW_MAJOR") != nullptr;     // (the round-1 form, kept for A/B runs)
        int n_waves16 = (T + 7) / 8;
        if (n_waves16 > 8192) n_waves16 = 8192;
        if ((int64_t)n_waves16 * 128 > partial_numel) n_waves16 = (int)(partial_numel / 128);
        if (!cross && !row_major16 && n_waves16 >= 1) {
            hipLaunchKernelGGL(qknorm_rope_bwd_b8_tok_kernel, dim3((n_waves16 + EW_WAVES - 1) / EW_WAVES), dim3(EW_THREADS), 0,
                               ST(stream), qkv, dq_rot, dk_rot, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, dqkv, partial, n_waves16,
                               pos_ids);
            GAMER_CHECK_LAUNCH(name);
            hipLaunchKernelGGL(qknorm_partial_reduce_kernel, dim3(128 / QKR_CW), dim3(1024), 0, ST(stream), partial, n_waves16, 1, 1, 0,
                               0, dwq, dwk, dbias_q, dbias_k, dbias_v);
            GAMER_CHECK_LAUNCH(name);
            return 0;
        }
        if (!cross || nb1 <= 4)
            hipLaunchKernelGGL(qknorm_rope_bwd_b8_kernel<4>, dim3(blocks), dim3(EW_THREADS), 0, ST(stream), qkv, dq_rot, dk_rot,
                               T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, cross, bias_q, bias_k, act_idx, nb1, dqkv, partial,
                               waves_per_head, pos_ids);
        else
            hipLaunchKernelGGL(qknorm_rope_bwd_b8_kernel<TBL_MAXROWS>, dim3(blocks), dim3(EW_THREADS), 0, ST(stream), qkv,
                               dq_rot, dk_rot, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, cross, bias_q, bias_k, act_idx, nb1,
                               dqkv, partial, waves_per_head, pos_ids);
    } else {
        static const bool row_major = getenv("GAMER_QKNORM_ROW_MAJOR") != nullptr;       // (the round-1 form, kept for A/B runs)
        int n_waves = (T + 3) / 4;
        if (n_waves > 8192) n_waves = 8192;
        if ((int64_t)n_waves * 128 > partial_numel) n_waves = (int)(partial_numel / 128);
        if (!cross && !row_major && n_waves >= 1) {
            // self attention: token-major (one partial row [dwq | dwk] per wave; the reduce sees one q and one k "head")
            hipLaunchKernelGGL(qknorm_rope_bwd_tok_kernel, dim3((n_waves + EW_WAVES - 1) / EW_WAVES), dim3(EW_THREADS), 0, ST(stream),
                               qkv, dq_rot, dk_rot, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, dqkv, partial, n_waves, pos_ids,
                               take_amax_sink().out[0]);
            GAMER_CHECK_LAUNCH(name);
            hipLaunchKernelGGL(qknorm_partial_reduce_kernel, dim3(128 / QKR_CW), dim3(1024), 0, ST(stream), partial, n_waves, 1, 1, 0,
                               0, dwq, dwk, dbias_q, dbias_k, dbias_v);
            GAMER_CHECK_LAUNCH(name);
            return 0;
        }
        hipLaunchKernelGGL(qknorm_rope_bwd_kernel<TA>, dim3(blocks), dim3(EW_THREADS), 0, ST(stream), qkv, dq_rot, dk_rot, T,
                           S, nq, nkv, wq, wk, eps, cos_t, sin_t, cross, bias_q, bias_k, act_idx, nb1, dqkv, partial,
                           waves_per_head, pos_ids, take_amax_sink().out[0]);
    }
    GAMER_CHECK_LAUNCH(name);
    const int cols = 128 + (cross ? NH * nb1 * 64 : 0);
    hipLaunchKernelGGL(qknorm_partial_reduce_kernel, dim3(cols / QKR_CW), dim3(1024), 0, ST(stream), partial,
                       waves_per_head, nq, nkv, cross, nb1, dwq, dwk, dbias_q, dbias_k, dbias_v);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_qknorm_rope_bwd(const float* qkv, const float* dq_rot, const float* dk_rot, int T, int S, int nq,
                                     int nkv, const float* wq, const float* wk, float eps, const float* cos_t,
                                     const float* sin_t, const float* bias_q, const float* bias_k,
                                     const int32_t* act_idx, int nb1, float* dqkv, float* dwq, float* dwk,
                                     float* dbias_q, float* dbias_k, float* dbias_v, const int32_t* pos_ids,
                                     float* partial, int64_t partial_numel, void* stream) {
    return qknorm_rope_bwd_impl<float>("gamer_qknorm_rope_bwd", qkv, dq_rot, dk_rot, T, S, nq, nkv, wq, wk, eps, cos_t,
                                       sin_t, bias_q, bias_k, act_idx, nb1, dqkv, dwq, dwk, dbias_q, dbias_k, dbias_v,
                                       pos_ids, partial, partial_numel, stream);
}
extern "C" int gamer_qknorm_rope_bwd_bf16(const gamer_bf16* qkv, const gamer_bf16* dq_rot, const gamer_bf16* dk_rot,
                                          int T, int S, int nq, int nkv, const float* wq, const float* wk, float eps,
                                          const float* cos_t, const float* sin_t, const float* bias_q,
                                          const float* bias_k, const int32_t* act_idx, int nb1, gamer_bf16* dqkv,
                                          float* dwq, float* dwk, float* dbias_q, float* dbias_k, float* dbias_v,
                                          const int32_t* pos_ids, float* partial, int64_t partial_numel, void* stream) {
    return qknorm_rope_bwd_impl<bf16_t>("gamer_qknorm_rope_bwd_bf16", (const bf16_t*)qkv, (const bf16_t*)dq_rot,
                                        (const bf16_t*)dk_rot, T, S, nq, nkv, wq, wk, eps, cos_t, sin_t, bias_q, bias_k,
                                        act_idx, nb1, (bf16_t*)dqkv, dwq, dwk, dbias_q, dbias_k, dbias_v, pos_ids,
                                        partial, partial_numel, stream);
}

extern "C" int gamer_residual_dropout_fwd(const float* x_in, const float* delta, const int32_t* src_rows, int T, int H,
                                          float p_drop, uint64_t seed, float* x_out, void* stream) {
    GAMER_CHECK_ARG(x_in && x_out && delta && T > 0 && H > 0 && H % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "gamer_residual_dropout_fwd: bad arguments T=%d H=%d p=%f", T, H, p_drop);
    hipLaunchKernelGGL(residual_dropout_fwd_kernel, dim3(grid_for_threads((int64_t)T * H / 4)), dim3(EW_THREADS), 0,
                       ST(stream), (const float4*)x_in, (const float4*)delta, src_rows, T, H / 4, p_drop, seed,
                       (float4*)x_out);
    GAMER_CHECK_LAUNCH("gamer_residual_dropout_fwd");
    return 0;
}

extern "C" int gamer_residual_dropout_bwd(const float* dx, const int32_t* src_rows, int T, int H, float p_drop,
                                          uint64_t seed, float* ddelta, void* stream) {
    GAMER_CHECK_ARG(dx && ddelta && T > 0 && H > 0 && H % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "gamer_residual_dropout_bwd: bad arguments T=%d H=%d p=%f", T, H, p_drop);
    hipLaunchKernelGGL(residual_dropout_bwd_kernel, dim3(grid_for_threads((int64_t)T * H / 4)), dim3(EW_THREADS), 0,
                       ST(stream), (const float4*)dx, src_rows, T, H / 4, p_drop, seed, (float4*)ddelta);
    GAMER_CHECK_LAUNCH("gamer_residual_dropout_bwd");
    return 0;
}

template <typename TA>
static int swiglu_fwd_impl(const char* name, const TA* g, const TA* u, int64_t n, float p_drop, uint64_t seed, TA* hm,
                           void* stream) {
    GAMER_CHECK_ARG(g && u && hm && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f, "%s: bad arguments", name);
    hipLaunchKernelGGL(swiglu_fwd_kernel<TA>, dim3(grid_for_threads(n / 4)), dim3(EW_THREADS), 0, ST(stream), g, u, n / 4,
                       p_drop, seed, hm, take_amax_sink().out[0]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_swiglu_fwd(const float* g, const float* u, int64_t n, float p_drop, uint64_t seed, float* hm,
                                void* stream) {
    return swiglu_fwd_impl<float>("gamer_swiglu_fwd", g, u, n, p_drop, seed, hm, stream);
}
extern "C" int gamer_swiglu_fwd_bf16(const gamer_bf16* g, const gamer_bf16* u, int64_t n, float p_drop, uint64_t seed,
                                     gamer_bf16* hm, void* stream) {
    return swiglu_fwd_impl<bf16_t>("gamer_swiglu_fwd_bf16", (const bf16_t*)g, (const bf16_t*)u, n, p_drop, seed,
                                   (bf16_t*)hm, stream);
}

template <typename TA>
static int swiglu_bwd_impl(const char* name, TA* g, TA* u, const TA* dhm, int64_t n, float p_drop, uint64_t seed,
                           void* stream) {
    GAMER_CHECK_ARG(g && u && dhm && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f, "%s: bad arguments", name);
    const AmaxSink sink = take_amax_sink();
    hipLaunchKernelGGL(swiglu_bwd_kernel<TA>, dim3(grid_for_threads(n / 4)), dim3(EW_THREADS), 0, ST(stream), g, u, dhm,
                       n / 4, p_drop, seed, sink.out[0], sink.out[1]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_swiglu_bwd(float* g, float* u, const float* dhm, int64_t n, float p_drop, uint64_t seed,
                                void* stream) {
    return swiglu_bwd_impl<float>("gamer_swiglu_bwd", g, u, dhm, n, p_drop, seed, stream);
}
extern "C" int gamer_swiglu_bwd_bf16(gamer_bf16* g, gamer_bf16* u, const gamer_bf16* dhm, int64_t n, float p_drop,
                                     uint64_t seed, void* stream) {
    return swiglu_bwd_impl<bf16_t>("gamer_swiglu_bwd_bf16", (bf16_t*)g, (bf16_t*)u, (const bf16_t*)dhm, n, p_drop, seed,
                                   stream);
}

template <typename TA>
static int swiglu_fwd_ld_impl(const char* name, const TA* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, TA* hm,
                              void* stream, const float* tbl = nullptr, const int32_t* row_group = nullptr) {
    GAMER_CHECK_ARG((tbl == nullptr) == (row_group == nullptr) && aligned16(tbl), "%s: tbl (16-byte aligned) and row_group come together", name);
    GAMER_CHECK_ARG(gu && hm && T > 0 && I > 0 && I % 4 == 0 && ld >= 2 * (int64_t)I && ld % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "%s: bad arguments (T=%d I=%d ld=%lld)", name, T, I, (long long)ld);
    GAMER_CHECK_ARG(aligned_vec4<TA>(gu) && aligned_vec4<TA>(hm), "%s: pointers must be aligned to four elements", name);
    hipLaunchKernelGGL(swiglu_fwd_ld_kernel<TA>, dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream), gu, ld, T, I / 4,
                       p_drop, seed, hm, take_amax_sink().out[0], tbl, row_group);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_swiglu_fwd_ld(const float* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, float* hm, void* stream) {
    return swiglu_fwd_ld_impl<float>("gamer_swiglu_fwd_ld", gu, ld, T, I, p_drop, seed, hm, stream);
}
extern "C" int gamer_swiglu_fwd_ld_bf16(const gamer_bf16* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, gamer_bf16* hm,
                                        void* stream) {
    return swiglu_fwd_ld_impl<bf16_t>("gamer_swiglu_fwd_ld_bf16", (const bf16_t*)gu, ld, T, I, p_drop, seed, (bf16_t*)hm, stream);
}
template <typename TA>
static int swiglu_bwd_ld_impl(const char* name, TA* gu, int64_t ld, int T, int I, const TA* dhm, float p_drop, uint64_t seed,
                              void* stream, const float* tbl = nullptr, const int32_t* row_group = nullptr) {
    GAMER_CHECK_ARG((tbl == nullptr) == (row_group == nullptr) && aligned16(tbl), "%s: tbl (16-byte aligned) and row_group come together", name);
    GAMER_CHECK_ARG(gu && dhm && T > 0 && I > 0 && I % 4 == 0 && ld >= 2 * (int64_t)I && ld % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "%s: bad arguments (T=%d I=%d ld=%lld)", name, T, I, (long long)ld);
    GAMER_CHECK_ARG(aligned_vec4<TA>(gu) && aligned_vec4<TA>(dhm), "%s: pointers must be aligned to four elements", name);
    const AmaxSink sink = take_amax_sink();
    hipLaunchKernelGGL(swiglu_bwd_ld_kernel<TA>, dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream), gu, ld, T, I / 4, dhm,
                       p_drop, seed, sink.out[0], sink.out[1], tbl, row_group);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_swiglu_bwd_ld(float* gu, int64_t ld, int T, int I, const float* dhm, float p_drop, uint64_t seed, void* stream) {
    return swiglu_bwd_ld_impl<float>("gamer_swiglu_bwd_ld", gu, ld, T, I, dhm, p_drop, seed, stream);
}
extern "C" int gamer_swiglu_fwd_ld_tbl(const float* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, float* hm,
                                       const float* tbl, const int32_t* row_group, void* stream) {
    GAMER_CHECK_ARG(tbl && row_group, "gamer_swiglu_fwd_ld_tbl: null table");
    return swiglu_fwd_ld_impl<float>("gamer_swiglu_fwd_ld_tbl", gu, ld, T, I, p_drop, seed, hm, stream, tbl, row_group);
}
extern "C" int gamer_swiglu_bwd_ld_tbl(float* gu, int64_t ld, int T, int I, const float* dhm, float p_drop, uint64_t seed,
                                       const float* tbl, const int32_t* row_group, void* stream) {
    GAMER_CHECK_ARG(tbl && row_group, "gamer_swiglu_bwd_ld_tbl: null table");
    return swiglu_bwd_ld_impl<float>("gamer_swiglu_bwd_ld_tbl", gu, ld, T, I, dhm, p_drop, seed, stream, tbl, row_group);
}
extern "C" int gamer_swiglu_bwd_ld_bf16(gamer_bf16* gu, int64_t ld, int T, int I, const gamer_bf16* dhm, float p_drop, uint64_t seed,
                                        void* stream) {
    return swiglu_bwd_ld_impl<bf16_t>("gamer_swiglu_bwd_ld_bf16", (bf16_t*)gu, ld, T, I, (const bf16_t*)dhm, p_drop, seed, stream);
}

extern "C" int gamer_silu_gate_fwd(const float* a, const float* gate, int64_t n, float* out, const float* resid,
                                   float p_drop, uint64_t seed, void* stream) {
    GAMER_CHECK_ARG(a && gate && out && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f, "gamer_silu_gate_fwd: bad arguments");
    hipLaunchKernelGGL((silu_gate_fwd_kernel<float, float>), dim3(grid_for_threads(n / 4)), dim3(EW_THREADS), 0, ST(stream),
                       a, gate, n / 4, out, (const float4*)resid, p_drop, seed);
    GAMER_CHECK_LAUNCH("gamer_silu_gate_fwd");
    return 0;
}
extern "C" int gamer_silu_gate_fwd_bf16(const gamer_bf16* a, const gamer_bf16* gate, int64_t n, float* out,
                                        const float* resid, float p_drop, uint64_t seed, void* stream) {
    GAMER_CHECK_ARG(a && gate && out && resid && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "gamer_silu_gate_fwd_bf16: bad arguments (resid is required: out is the fp32 residual stream)");
    hipLaunchKernelGGL((silu_gate_fwd_kernel<bf16_t, float>), dim3(grid_for_threads(n / 4)), dim3(EW_THREADS), 0, ST(stream),
                       (const bf16_t*)a, (const bf16_t*)gate, n / 4, out, (const float4*)resid, p_drop, seed);
    GAMER_CHECK_LAUNCH("gamer_silu_gate_fwd_bf16");
    return 0;
}

template <typename TA>
static int silu_gate_bwd_impl(const char* name, const TA* a, const TA* gate, const float* dout, int64_t n, TA* da,
                              TA* dgate, float p_drop, uint64_t seed, void* stream) {
    GAMER_CHECK_ARG(a && gate && dout && da && dgate && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                    "%s: bad arguments", name);
    const AmaxSink sink = take_amax_sink();
    hipLaunchKernelGGL(silu_gate_bwd_kernel<TA>, dim3(grid_for_threads(n / 4)), dim3(EW_THREADS), 0, ST(stream), a, gate,
                       (const float4*)dout, n / 4, da, dgate, p_drop, seed, sink.out[0], sink.out[1]);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_silu_gate_bwd(const float* a, const float* gate, const float* dout, int64_t n, float* da,
                                   float* dgate, float p_drop, uint64_t seed, void* stream) {
    return silu_gate_bwd_impl<float>("gamer_silu_gate_bwd", a, gate, dout, n, da, dgate, p_drop, seed, stream);
}
extern "C" int gamer_silu_gate_bwd_bf16(const gamer_bf16* a, const gamer_bf16* gate, const float* dout, int64_t n,
                                        gamer_bf16* da, gamer_bf16* dgate, float p_drop, uint64_t seed, void* stream) {
    return silu_gate_bwd_impl<bf16_t>("gamer_silu_gate_bwd_bf16", (const bf16_t*)a, (const bf16_t*)gate, dout, n,
                                      (bf16_t*)da, (bf16_t*)dgate, p_drop, seed, stream);
}

template <typename TL>
static int ce_fwd_impl(const char* name, TL* logits, int ldl, const int64_t* labels, int B, int S, int V,
                       float temperature, int ignore_index, float* lse_out, float* row_loss, float* loss_sum,
                       float* count, void* stream) {
    GAMER_CHECK_ARG(logits && labels && lse_out && row_loss && loss_sum && count, "%s: null pointer", name);
    GAMER_CHECK_ARG(B > 0 && S > 0 && V > 0 && ldl >= V && temperature > 0.f, "%s: bad shape B=%d S=%d V=%d ldl=%d", name, B, S, V, ldl);
    const int T = B * S;
    const int vec4 = (ldl % (16 / (int)sizeof(TL)) == 0 && aligned16(logits)) ? 1 : 0;
    hipLaunchKernelGGL(ce_fwd_kernel<TL>, dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream), logits, ldl, labels, T,
                       S, V, 1.f / temperature, ignore_index, lse_out, row_loss, vec4);
    GAMER_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, ST(stream), row_loss, labels, T, S, V, ignore_index,
                       loss_sum, count);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_ce_fwd(float* logits, int ldl, const int64_t* labels, int B, int S, int V, float temperature,
                            int ignore_index, float* lse_out, float* row_loss, float* loss_sum, float* count,
                            void* stream) {
    return ce_fwd_impl<float>("gamer_ce_fwd", logits, ldl, labels, B, S, V, temperature, ignore_index, lse_out, row_loss,
                              loss_sum, count, stream);
}
extern "C" int gamer_ce_fwd_bf16(gamer_bf16* logits, int ldl, const int64_t* labels, int B, int S, int V,
                                 float temperature, int ignore_index, float* lse_out, float* row_loss, float* loss_sum,
                                 float* count, void* stream) {
    return ce_fwd_impl<bf16_t>("gamer_ce_fwd_bf16", (bf16_t*)logits, ldl, labels, B, S, V, temperature, ignore_index,
                               lse_out, row_loss, loss_sum, count, stream);
}

template <typename TL>
static int ce_bwd_impl(const char* name, TL* logits, int ldl, const int64_t* labels, int B, int S, int V,
                       float temperature, int ignore_index, const float* lse, const float* count_dev, float denom_host,
                       float dloss, const float* dloss_dev, void* stream) {
    GAMER_CHECK_ARG(logits && labels && lse, "%s: null pointer", name);
    GAMER_CHECK_ARG(B > 0 && S > 0 && V > 0 && ldl >= V && temperature > 0.f, "%s: bad shape", name);
    GAMER_CHECK_ARG(count_dev || denom_host > 0.f, "%s: need count_dev or a positive denom_host", name);
    const int T = B * S;
    hipLaunchKernelGGL(ce_bwd_kernel<TL>, dim3(grid_for_waves(T)), dim3(EW_THREADS), 0, ST(stream), logits, ldl, labels, T,
                       S, V, ignore_index, lse, count_dev, denom_host, dloss / temperature, dloss_dev, take_amax_sink().out[0],
                       (ldl % (16 / (int)sizeof(TL)) == 0 && aligned16(logits)) ? 1 : 0);
    GAMER_CHECK_LAUNCH(name);
    return 0;
}
extern "C" int gamer_ce_bwd(float* logits, int ldl, const int64_t* labels, int B, int S, int V, float temperature,
                            int ignore_index, const float* lse, const float* count_dev, float denom_host, float dloss,
                            const float* dloss_dev, void* stream) {
    return ce_bwd_impl<float>("gamer_ce_bwd", logits, ldl, labels, B, S, V, temperature, ignore_index, lse, count_dev,
                              denom_host, dloss, dloss_dev, stream);
}
extern "C" int gamer_ce_bwd_bf16(gamer_bf16* logits, int ldl, const int64_t* labels, int B, int S, int V,
                                 float temperature, int ignore_index, const float* lse, const float* count_dev,
                                 float denom_host, float dloss, const float* dloss_dev, void* stream) {
    return ce_bwd_impl<bf16_t>("gamer_ce_bwd_bf16", (bf16_t*)logits, ldl, labels, B, S, V, temperature, ignore_index, lse,
                               count_dev, denom_host, dloss, dloss_dev, stream);
}

extern "C" int gamer_fill_f32(float* p, int64_t n, float value, void* stream) {
    GAMER_CHECK_ARG(p && n >= 0, "gamer_fill_f32: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for_threads(n)), dim3(EW_THREADS), 0, ST(stream), p, n, value);
    GAMER_CHECK_LAUNCH("gamer_fill_f32");
    return 0;
}
