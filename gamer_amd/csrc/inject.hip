// The behaviour-embedding share of the injecting layers' gate|up projection as a TABLE (fp32, plain FMA arithmetic).
//
// Reference: MyQwen3SparseMLP.forward (ref:SeqRec/models/generative/Qwen3Moe/FFN.py:53-72) concatenates the behaviour embedding
// to the hidden state, h = cat(x, Eb[beh]) ([T, 256 + 64]), before the position's expert: gate|up = W_e h.  The 64 embedding
// columns take only NB + 1 different values, so their share of the product is W_e[:, 256:] Eb[b] - one row of 2 I numbers per
// (expert, behaviour) pair.  The engine sorts the expert rows by (expert, behaviour), runs the projection on the 256 hidden
// columns alone (K = 256: the activation-stationary / output-stationary / 256 x 256-tile kernels instead of the 128 x 128 one at
// K = 320) and adds the table row where the gate|up values are consumed (gamer_swiglu_fwd_ld_tbl, gamer_gemm_desc.sw_tbl).
// Backward: with SegSum[g] = the column sums of d(gate|up) over the rows of group g = (e, b) (gamer_segment_colsum),
//     dW_e[:, 256:] += sum_b SegSum[e, b]^T Eb[b]          dEb[b] += sum_e SegSum[e, b] W_e[:, 256:]
// - the weight gradient of the 64 columns and the embedding gradient without the [T, 64] input gradient ever existing.
#include "common.h"

namespace gamer {

constexpr int INJ_MAX_NB1 = 16;
constexpr int INJ_MAX_EB = 256;

// tbl[(e * NB1 + b) * twoI + n] = sum_j Eb[b][j] * W[(e * twoI + n) * ldw + col0 + j]: one thread per (e, n)
__global__ void __launch_bounds__(256)
inject_table_fwd_kernel(const float* __restrict__ Eb, const float* __restrict__ W, int64_t ldw, int col0, int rows_total, int twoI,
                        int NB1, int EB, float* __restrict__ tbl) {
    __shared__ float eb[INJ_MAX_NB1 * INJ_MAX_EB];
    for (int i = threadIdx.x; i < NB1 * EB; i += 256) eb[i] = Eb[i];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;            // e * twoI + n
    if (row >= rows_total) return;
    const int e = row / twoI, n = row - e * twoI;
    const float* w = W + (int64_t)row * ldw + col0;
    float acc[INJ_MAX_NB1];
#pragma unroll
    for (int b = 0; b < INJ_MAX_NB1; ++b) acc[b] = 0.f;
    for (int j = 0; j < EB; j += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + j);
#pragma unroll
        for (int b = 0; b < INJ_MAX_NB1; ++b) {
            if (b < NB1) {
                const float* e4 = eb + b * EB + j;
                acc[b] = fmaf(e4[0], w4.x, acc[b]); acc[b] = fmaf(e4[1], w4.y, acc[b]);
                acc[b] = fmaf(e4[2], w4.z, acc[b]); acc[b] = fmaf(e4[3], w4.w, acc[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < INJ_MAX_NB1; ++b)
        if (b < NB1) tbl[((int64_t)e * NB1 + b) * twoI + n] = acc[b];
}

// ---- column sums of the rows of every segment (rows sorted by segment, offsets[nseg + 1]) -------------------------------------------
// pass 1: a workgroup takes one chunk of SEG_CH rows of ONE segment (chunks never cross a segment boundary: the grid holds
// ceil(rows / SEG_CH) + nseg workgroups, the surplus ones leave) and writes its column sums to partial[workgroup]; pass 2 adds a
// segment's partial rows in workgroup order - a fixed summation order, no atomics.
constexpr int SEG_CH = 256;
__device__ __forceinline__ bool seg_find(const int32_t* __restrict__ offsets, int nseg, int wg, int& seg, int& r0, int& r1) {
    int prev = offsets[0], before = 0;
    for (int s = 0; s < nseg; ++s) {
        const int nxt = offsets[s + 1];
        const int chunks = (nxt - prev + SEG_CH - 1) / SEG_CH;
        if (wg < before + chunks) {
            seg = s; r0 = prev + (wg - before) * SEG_CH; r1 = min(nxt, r0 + SEG_CH);
            return true;
        }
        before += chunks;
        prev = nxt;
    }
    return false;
}
__global__ void __launch_bounds__(256)
segment_colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int cols4, const int32_t* __restrict__ offsets, int nseg,
                              float4* __restrict__ partial) {
    int seg, r0, r1;
    if (!seg_find(offsets, nseg, blockIdx.x, seg, r0, r1)) return;
    for (int c = threadIdx.x; c < cols4; c += 256) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        const float* p = x + (int64_t)r0 * ld + 4 * c;
        int r = r0;
        // (four independent loads in flight, fixed combination order.  Measured: eight in flight with chunks of 128 rows 2.83 ms per step
        // against 2.17 for this form - 3.9 TB/s)
        for (; r + 3 < r1; r += 4, p += 4 * ld) {
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + ld);
            const float4 v2 = *reinterpret_cast<const float4*>(p + 2 * ld), v3 = *reinterpret_cast<const float4*>(p + 3 * ld);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w; a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; r < r1; ++r, p += ld) {
            const float4 v0 = *reinterpret_cast<const float4*>(p);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
        float4 s;
        s.x = (a0.x + a1.x) + (a2.x + a3.x); s.y = (a0.y + a1.y) + (a2.y + a3.y);
        s.z = (a0.z + a1.z) + (a2.z + a3.z); s.w = (a0.w + a1.w) + (a2.w + a3.w);
        partial[(int64_t)blockIdx.x * cols4 + c] = s;
    }
}
// pass 2: out[seg][c] = sum over the segment's workgroups; grid (nseg, ceil(cols4 / 64)), a block = 64 columns x 4 row groups: group g
// adds the partial rows g, g + 4, ... (two independent chains), the groups are combined through LDS in group order - a fixed order.
// (One thread per column walking all of a segment's rows - up to ~400 dependent loads - took 0.17 ms per call.)
__global__ void __launch_bounds__(256)
segment_colsum_reduce_kernel(const float4* __restrict__ partial, int cols4, const int32_t* __restrict__ offsets, int nseg,
                             float4* __restrict__ out) {
    __shared__ float4 red[4][64];
    const int seg = blockIdx.x;
    int first = 0;
    for (int s = 0; s < seg; ++s) first += (offsets[s + 1] - offsets[s] + SEG_CH - 1) / SEG_CH;
    const int n = (offsets[seg + 1] - offsets[seg] + SEG_CH - 1) / SEG_CH;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c < cols4) {
        int i = g;
        for (; i + 4 < n; i += 8) {
            const float4 v = partial[(int64_t)(first + i) * cols4 + c], w = partial[(int64_t)(first + i + 4) * cols4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        if (i < n) {
            const float4 v = partial[(int64_t)(first + i) * cols4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[g][cl] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    if (g == 0 && c < cols4) {
        float4 s = red[0][cl];
#pragma unroll
        for (int k = 1; k < 4; ++k) { s.x += red[k][cl].x; s.y += red[k][cl].y; s.z += red[k][cl].z; s.w += red[k][cl].w; }
        out[(int64_t)seg * cols4 + c] = s;
    }
}

// dW[(e * twoI + n) * ldw + col0 + j] += sum_b seg[(e * NB1 + b) * twoI + n] * Eb[b][j]: thread per (row, four columns j)
__global__ void __launch_bounds__(256)
inject_table_bwd_w_kernel(const float* __restrict__ seg, const float* __restrict__ Eb, int rows_total, int twoI, int NB1, int EB,
                          float* __restrict__ dW, int64_t ldw, int col0) {
    __shared__ float eb[INJ_MAX_NB1 * INJ_MAX_EB];
    for (int i = threadIdx.x; i < NB1 * EB; i += 256) eb[i] = Eb[i];
    __syncthreads();
    const int q = EB / 4;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= (int64_t)rows_total * q) return;
    const int row = (int)(item / q), j = (int)(item % q) * 4;
    const int e = row / twoI, n = row - e * twoI;
    float* dst = dW + (int64_t)row * ldw + col0 + j;
    float4 a = *reinterpret_cast<const float4*>(dst);
    for (int b = 0; b < NB1; ++b) {
        const float s = seg[((int64_t)e * NB1 + b) * twoI + n];
        const float* e4 = eb + b * EB + j;
        a.x = fmaf(s, e4[0], a.x); a.y = fmaf(s, e4[1], a.y); a.z = fmaf(s, e4[2], a.z); a.w = fmaf(s, e4[3], a.w);
    }
    *reinterpret_cast<float4*>(dst) = a;
}
// dEb[b][j] += sum_e sum_n seg[(e * NB1 + b) * twoI + n] * W[(e * twoI + n) * ldw + col0 + j].  Pass 1: one workgroup of 1024 threads per
// (b, e): thread (j, part) - 16 parts - sums the rows n = part, part + 16, ... (four independent loads in flight), the parts are added in
// order through LDS -> scratch[(b * E + e) * EB + j]; pass 2: one thread per (b, j) adds the experts in order.
__global__ void __launch_bounds__(1024)
inject_table_bwd_e_kernel(const float* __restrict__ seg, const float* __restrict__ W, int64_t ldw, int col0, int E, int twoI, int NB1,
                          int EB, float* __restrict__ scratch) {
    __shared__ float red[16][64];
    const int b = blockIdx.x, e = blockIdx.y, jl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const float* sg = seg + ((int64_t)e * NB1 + b) * twoI;
    for (int j0 = 0; j0 < EB; j0 += 64) {
        const int j = j0 + jl;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (j < EB) {
            const float* w = W + (int64_t)e * twoI * ldw + col0 + j;
            int n = part;
            for (; n + 48 < twoI; n += 64) {
                a0 = fmaf(sg[n], w[(int64_t)n * ldw], a0); a1 = fmaf(sg[n + 16], w[(int64_t)(n + 16) * ldw], a1);
                a2 = fmaf(sg[n + 32], w[(int64_t)(n + 32) * ldw], a2); a3 = fmaf(sg[n + 48], w[(int64_t)(n + 48) * ldw], a3);
            }
            for (; n < twoI; n += 16) a0 = fmaf(sg[n], w[(int64_t)n * ldw], a0);
        }
        red[part][jl] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (part == 0 && j < EB) {
            float s = red[0][jl];
#pragma unroll
            for (int p2 = 1; p2 < 16; ++p2) s += red[p2][jl];
            scratch[((int64_t)b * E + e) * EB + j] = s;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
inject_table_bwd_e_reduce_kernel(const float* __restrict__ scratch, int E, int NB1, int EB, float* __restrict__ dEb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NB1 * EB) return;
    const int b = i / EB, j = i - b * EB;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += scratch[((int64_t)b * E + e) * EB + j];
    dEb[i] += s;
}

}  // namespace gamer

using namespace gamer;

#define ST(s) ((hipStream_t)(s))

extern "C" int gamer_inject_table_fwd(const float* Eb, const float* W, int64_t ldw, int col0, int E, int twoI, int NB1, int EB,
                                      float* tbl, void* stream) {
    GAMER_CHECK_ARG(Eb && W && tbl, "gamer_inject_table_fwd: null pointer");
    GAMER_CHECK_ARG(E > 0 && twoI > 0 && NB1 > 0 && NB1 <= INJ_MAX_NB1 && EB > 0 && EB % 4 == 0 && EB <= INJ_MAX_EB && col0 >= 0 &&
                    col0 % 4 == 0 && ldw % 4 == 0 && ldw >= col0 + EB && aligned16(W),
                    "gamer_inject_table_fwd: bad shape E=%d 2I=%d NB1=%d EB=%d col0=%d ldw=%lld", E, twoI, NB1, EB, col0, (long long)ldw);
    const int rows = E * twoI;
    hipLaunchKernelGGL(inject_table_fwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, ST(stream), Eb, W, ldw, col0, rows, twoI, NB1, EB,
                       tbl);
    GAMER_CHECK_LAUNCH("gamer_inject_table_fwd");
    return 0;
}

extern "C" int64_t gamer_segment_colsum_ws_floats(int rows, int cols, int nseg) {
    return ((int64_t)(rows + SEG_CH - 1) / SEG_CH + nseg) * (int64_t)cols;
}

extern "C" int gamer_segment_colsum(const float* x, int64_t ld, int rows, int cols, const int32_t* offsets, int nseg, float* ws,
                                    int64_t ws_floats, float* out, void* stream) {
    GAMER_CHECK_ARG(x && offsets && ws && out, "gamer_segment_colsum: null pointer");
    GAMER_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols && nseg > 0 && aligned16(x) && aligned16(ws) &&
                    aligned16(out), "gamer_segment_colsum: bad shape rows=%d cols=%d ld=%lld nseg=%d", rows, cols, (long long)ld, nseg);
    GAMER_CHECK_ARG(ws_floats >= gamer_segment_colsum_ws_floats(rows, cols, nseg),
                    "gamer_segment_colsum: ws holds %lld floats, needs %lld (gamer_segment_colsum_ws_floats)", (long long)ws_floats,
                    (long long)gamer_segment_colsum_ws_floats(rows, cols, nseg));
    const int wgs = (rows + SEG_CH - 1) / SEG_CH + nseg;
    hipLaunchKernelGGL(segment_colsum_partial_kernel, dim3(wgs), dim3(256), 0, ST(stream), x, ld, cols / 4, offsets, nseg, (float4*)ws);
    GAMER_CHECK_LAUNCH("gamer_segment_colsum/partial");
    hipLaunchKernelGGL(segment_colsum_reduce_kernel, dim3(nseg, (cols / 4 + 63) / 64), dim3(256), 0, ST(stream), (const float4*)ws,
                       cols / 4, offsets, nseg, (float4*)out);
    GAMER_CHECK_LAUNCH("gamer_segment_colsum/reduce");
    return 0;
}

extern "C" int gamer_inject_table_bwd(const float* seg, const float* Eb, const float* W, int64_t ldw, int col0, int E, int twoI, int NB1,
                                      int EB, float* dW, float* dEb, float* scratch, void* stream) {
    GAMER_CHECK_ARG(seg && Eb && W && dW && dEb && scratch, "gamer_inject_table_bwd: null pointer");
    GAMER_CHECK_ARG(E > 0 && twoI > 0 && NB1 > 0 && NB1 <= INJ_MAX_NB1 && EB > 0 && EB % 4 == 0 && EB <= INJ_MAX_EB && col0 >= 0 &&
                    col0 % 4 == 0 && ldw % 4 == 0 && ldw >= col0 + EB && aligned16(dW) && aligned16(W),
                    "gamer_inject_table_bwd: bad shape E=%d 2I=%d NB1=%d EB=%d col0=%d ldw=%lld", E, twoI, NB1, EB, col0, (long long)ldw);
    const int64_t items = (int64_t)E * twoI * (EB / 4);
    hipLaunchKernelGGL(inject_table_bwd_w_kernel, dim3((int)((items + 255) / 256)), dim3(256), 0, ST(stream), seg, Eb, E * twoI, twoI, NB1,
                       EB, dW, ldw, col0);
    GAMER_CHECK_LAUNCH("gamer_inject_table_bwd/dW");
    hipLaunchKernelGGL(inject_table_bwd_e_kernel, dim3(NB1, E), dim3(1024), 0, ST(stream), seg, W, ldw, col0, E, twoI, NB1, EB, scratch);
    GAMER_CHECK_LAUNCH("gamer_inject_table_bwd/dEb partial");
    hipLaunchKernelGGL(inject_table_bwd_e_reduce_kernel, dim3((NB1 * EB + 255) / 256), dim3(256), 0, ST(stream), scratch, E, NB1, EB, dEb);
    GAMER_CHECK_LAUNCH("gamer_inject_table_bwd/dEb");
    return 0;
}
