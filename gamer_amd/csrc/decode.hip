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
