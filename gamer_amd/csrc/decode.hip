// Trie-constrained beam-search scoring for the SMB-decoder evaluation path (gfx950).
//
// Replaces, per generation step, HF's  log_softmax(next_token_logits)  followed by
// PrefixConstrainedLogitsProcessor driven by the reference's Trie
// (ref:SeqRec/generation/trie.py:5-104, ref:SeqRec/tasks/test_SMB_decoder.py:120,470-500): the reference walks a
// Python dict per (batch, beam) row on the host; here the item trie is a CSR array on the device and one
// wave scores one row.  Integer work + one HBM pass over the row's logits.
#include "common.h"

namespace gamer {

// scores[n][v] = -inf, except scores[n][tok] = logits[row_n][tok] - logsumexp(logits[row_n][0..V)) + beam_score[n]
// for every child token of trie node node[n] (node < 0: no token is allowed, as for an unknown prefix).
__global__ void __launch_bounds__(256)
trie_logprobs_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ row_index,
                     const float* __restrict__ beam_score, const int32_t* __restrict__ node,
                     const int32_t* __restrict__ child_start, const int32_t* __restrict__ child_tok, int N, int V,
                     float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int n = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    if (n >= N) return;
    const float* row = logits + (int64_t)row_index[n] * ld;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, row[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += __expf(row[v] - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    float* out = scores + (int64_t)n * V;
    for (int v = lane; v < V; v += 64) out[v] = -INFINITY;
    const int nd = node[n];
    if (nd < 0) return;
    const float add = beam_score[n] - lse;
    const int c0 = child_start[nd], c1 = child_start[nd + 1];
    // the fill above and the scatter below touch the same row from the same wave: order them
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int c = c0 + lane; c < c1; c += 64) {
        const int tok = child_tok[c];
        if (tok >= 0 && tok < V) out[tok] = row[tok] + add;
    }
}

// next[n] = child of node[n] reached by token[n]  (-1 when there is none: Trie.get returns [] from there on)
__global__ void __launch_bounds__(256)
trie_advance_kernel(const int32_t* __restrict__ node, const int64_t* __restrict__ token,
                    const int32_t* __restrict__ child_start, const int32_t* __restrict__ child_tok,
                    const int32_t* __restrict__ child_node, int N, int32_t* __restrict__ next) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int nd = node[n];
    int res = -1;
    if (nd >= 0) {
        const int64_t t = token[n];
        for (int c = child_start[nd]; c < child_start[nd + 1]; ++c)
            if ((int64_t)child_tok[c] == t) { res = child_node[c]; break; }
    }
    next[n] = res;
}


// ---------------------------------------------------------------------------------------------------
// Attention of ONE new token per beam against a K/V cache that is shared by the beams of a sample.
//
// The reference expands every prompt to num_beams copies and caches K/V per copy (HF generate); the prompt
// part of those copies is identical, so here it is stored once per sample ([B, L0] rows) and only the <= 4
// generated positions are per beam.  One workgroup = (sample, kv head): the prompt K/V tiles go through LDS
// once and serve all num_beams * G query rows.  HBM-bound on the prompt K/V of one layer (~0.2 GB per call at
// 256 users), the arithmetic is plain VALU (a few thousand FMAs per lane).
//   self attention : prompt key j allowed iff key_ok[b][j]; generated keys (own ancestors + itself) allowed
//   cross attention: prompt key j allowed iff key_ok[b][j] (= kept and lower level than the target behaviour, the
//                    cached last mask row of model.py:603-617); generated keys masked; uniform[b] = 1 when no key
//                    is allowed: the row is then the mean of V over ALL L0 + t keys (finfo.min quirk)
constexpr int DEC_MAXQ = 16;            // query rows per wave (4 waves): num_beams * G <= 64

template <int G>
__global__ void __launch_bounds__(256)
attn_decode_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kp, int ldkp,
                   const float* __restrict__ vp, int ldvp, const int32_t* __restrict__ key_ok,
                   const float* __restrict__ kg, const float* __restrict__ vg, int ldg, int tmax, int t, int gen_ok,
                   const int32_t* __restrict__ uniform, int nb, int L0, int nq, int nkv, float scale,
                   float* __restrict__ o) {
    __shared__ float Ks[64 * 65];
    __shared__ __attribute__((aligned(16))) float Vs[64 * 64];
    __shared__ __attribute__((aligned(16))) float qs[64 * 64];
    __shared__ float ps[4][64];
    __shared__ int32_t oks[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / nkv, kvh = blockIdx.x % nkv;
    const int Qn = nb * G;
    const bool uni = uniform != nullptr && uniform[b] != 0;
    for (int f = tid; f < Qn * 16; f += 256) {
        const int qi = f >> 4, c4 = (f & 15) << 2;
        const int n = b * nb + qi / G, head = kvh * G + qi % G;
        *reinterpret_cast<float4*>(&qs[qi * 64 + c4]) = *reinterpret_cast<const float4*>(q + (int64_t)n * ldq + head * 64 + c4);
    }
    float m[DEC_MAXQ], l[DEC_MAXQ], acc[DEC_MAXQ];
#pragma unroll
    for (int i = 0; i < DEC_MAXQ; ++i) { m[i] = -INFINITY; l[i] = 0.f; acc[i] = 0.f; }
    float vsum = 0.f;                        // sum over the prompt keys of V[j][lane] (uniform rows)
    for (int j0 = 0; j0 < L0; j0 += 64) {
        __syncthreads();
        for (int f = tid; f < 64 * 16; f += 256) {
            const int row = f >> 4, c4 = (f & 15) << 2;
            const int j = j0 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (j < L0) {
                kv = *reinterpret_cast<const float4*>(kp + ((int64_t)b * L0 + j) * ldkp + kvh * 64 + c4);
                vv = *reinterpret_cast<const float4*>(vp + ((int64_t)b * L0 + j) * ldvp + kvh * 64 + c4);
            }
            Ks[row * 65 + c4] = kv.x; Ks[row * 65 + c4 + 1] = kv.y; Ks[row * 65 + c4 + 2] = kv.z; Ks[row * 65 + c4 + 3] = kv.w;
            *reinterpret_cast<float4*>(&Vs[row * 64 + c4]) = vv;
        }
        if (tid < 64) oks[tid] = (j0 + tid < L0) ? key_ok[(int64_t)b * L0 + j0 + tid] : 0;
        __syncthreads();
        if (uni) {
            if (w == 0) for (int j = 0; j < 64; ++j) vsum += Vs[j * 64 + lane];
            continue;
        }
        const bool ok = oks[lane] != 0;
#pragma unroll
        for (int i = 0; i < DEC_MAXQ; ++i) {
            const int qi = w + 4 * i;
            if (qi >= Qn) break;
            float s = 0.f;
            for (int d = 0; d < 64; ++d) s += qs[qi * 64 + d] * Ks[lane * 65 + d];
            s = ok ? s * scale : -INFINITY;
            const float mt = wave_max(s);
            if (mt == -INFINITY) continue;                    // nothing allowed in this tile
            const float mn = fmaxf(m[i], mt);
            const float alpha = __expf(m[i] - mn);            // m = -inf -> 0
            const float pe = __expf(s - mn);
            l[i] = l[i] * alpha + wave_sum(pe);
            m[i] = mn;
            ps[w][lane] = pe;
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes landed
            __builtin_amdgcn_wave_barrier();
            float a = acc[i] * alpha;
            for (int j = 0; j < 64; ++j) a += ps[w][j] * Vs[j * 64 + lane];
            acc[i] = a;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (uni) {                               // wave 0 holds the prompt sum: publish it (block-uniform branch)
        if (w == 0) ps[0][lane] = vsum;
        __syncthreads();
    }
    // generated positions of each beam (its own ancestors and the new token), then the output
#pragma unroll
    for (int i = 0; i < DEC_MAXQ; ++i) {
        const int qi = w + 4 * i;
        if (qi >= Qn) break;
        const int n = b * nb + qi / G, head = kvh * G + qi % G;
        float res;
        if (uni) {
            float a = ps[0][lane];
            for (int g = 0; g < t; ++g) a += vg[((int64_t)n * tmax + g) * ldg + kvh * 64 + lane];
            res = a / (float)(L0 + t);
        } else {
            float mi = m[i], li = l[i], a = acc[i];
            if (gen_ok) {
                for (int g = 0; g < t; ++g) {
                    const float* krow = kg + ((int64_t)n * tmax + g) * ldg + kvh * 64;
                    const float s = wave_sum(qs[qi * 64 + lane] * krow[lane]) * scale;
                    const float mn = fmaxf(mi, s);
                    const float alpha = __expf(mi - mn), pe = __expf(s - mn);
                    li = li * alpha + pe;
                    a = a * alpha + pe * vg[((int64_t)n * tmax + g) * ldg + kvh * 64 + lane];
                    mi = mn;
                }
            }
            res = li > 0.f ? a / li : 0.f;
        }
        o[(int64_t)n * nq * 64 + head * 64 + lane] = res;
    }
}

}  // namespace gamer

using namespace gamer;

extern "C" int gamer_trie_logprobs(const float* logits, int64_t ld, const int32_t* row_index, const float* beam_score,
                                   const int32_t* node, const int32_t* child_start, const int32_t* child_tok, int N,
                                   int V, float* scores, void* stream) {
    GAMER_CHECK_ARG(logits && row_index && beam_score && node && child_start && child_tok && scores,
                    "gamer_trie_logprobs: null pointer");
    GAMER_CHECK_ARG(N > 0 && V > 0 && ld >= V, "gamer_trie_logprobs: bad shape N=%d V=%d ld=%lld", N, V, (long long)ld);
    const int blocks = (N + 3) / 4;
    hipLaunchKernelGGL(trie_logprobs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, row_index,
                       beam_score, node, child_start, child_tok, N, V, scores);
    GAMER_CHECK_LAUNCH("gamer_trie_logprobs");
    return 0;
}

extern "C" int gamer_trie_advance(const int32_t* node, const int64_t* token, const int32_t* child_start,
                                  const int32_t* child_tok, const int32_t* child_node, int N, int32_t* next,
                                  void* stream) {
    GAMER_CHECK_ARG(node && token && child_start && child_tok && child_node && next, "gamer_trie_advance: null pointer");
    GAMER_CHECK_ARG(N > 0, "gamer_trie_advance: N=%d", N);
    hipLaunchKernelGGL(trie_advance_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, node, token,
                       child_start, child_tok, child_node, N, next);
    GAMER_CHECK_LAUNCH("gamer_trie_advance");
    return 0;
}

extern "C" int gamer_attn_decode(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                                 const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t,
                                 int gen_ok, const int32_t* uniform, int B, int nb, int L0, int nq, int nkv,
                                 float scale, float* o, void* stream) {
    GAMER_CHECK_ARG(q && kp && vp && key_ok && kg && vg && o, "gamer_attn_decode: null pointer");
    GAMER_CHECK_ARG(B > 0 && nb > 0 && L0 > 0 && nq > 0 && nkv > 0 && nq % nkv == 0 && t >= 1 && t <= tmax,
                    "gamer_attn_decode: bad shape B=%d nb=%d L0=%d nq=%d nkv=%d t=%d tmax=%d", B, nb, L0, nq, nkv, t, tmax);
    const int G = nq / nkv;
    GAMER_CHECK_ARG((G == 1 || G == 2) && nb * G <= 4 * DEC_MAXQ, "gamer_attn_decode: num_beams * (nq/nkv) = %d > %d",
                    nb * G, 4 * DEC_MAXQ);
    GAMER_CHECK_ARG(ldq % 4 == 0 && ldkp % 4 == 0 && ldvp % 4 == 0 && aligned16(q) && aligned16(kp) && aligned16(vp),
                    "gamer_attn_decode: q / prompt K / V need 16-byte alignment and leading dims %% 4 == 0");
    dim3 grid(B * nkv);
    if (G == 1)
        hipLaunchKernelGGL(attn_decode_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                           key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o);
    else
        hipLaunchKernelGGL(attn_decode_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                           key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o);
    GAMER_CHECK_LAUNCH("gamer_attn_decode");
    return 0;
}
