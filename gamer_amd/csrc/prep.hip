// Router maps, mask-predicate levels, empty-row flags and expert token lists (integer work).
// Reference behaviour restated: ref:SeqRec/models/generative/Qwen3Multi/router.py:74-201,
// ref:SeqRec/models/generative/Qwen3Multi/model.py:573-630,691-741,
// ref:SeqRec/models/generative/Qwen3Moe/FFN.py:63-68.
#include "common.h"
#include <atomic>

namespace gamer {

static thread_local char g_err[512] = {0};
static std::atomic<unsigned> g_env_epoch{1};
unsigned env_epoch() { return g_env_epoch.load(std::memory_order_acquire); }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    // every failing entry point comes through here: whatever the caller armed for "the next launch" (gamer_amax_sink,
    // gamer_attn_split_amax) must not survive a rejected call and attach itself to a later launch of another tensor
    take_amax_sink();
    disarm_attn_amax();
}

constexpr int ROUTER_THREADS = 256;
constexpr int INT_BIG = 0x7fffffff;

// One workgroup per sequence.  LDS prefix-min scan gives min_{j<=i} klevel[j] for the empty flags.
__global__ void __launch_bounds__(ROUTER_THREADS)
router_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ attn_mask,
              const int64_t* __restrict__ actions, const int32_t* __restrict__ lut, int vocab,
              int S, int P, int pad_id, int eos_id,
              int32_t* __restrict__ expert, int32_t* __restrict__ beh_idx, int32_t* __restrict__ act_idx,
              int32_t* __restrict__ kl_self, int32_t* __restrict__ kl_cross, int32_t* __restrict__ ql_cross,
              int32_t* __restrict__ empty_self, int32_t* __restrict__ empty_cross,
              int32_t* __restrict__ tile_empty_self, int32_t* __restrict__ tile_empty_cross,
              int32_t* __restrict__ bad_token) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    int32_t* pm_self = smem;            // [S] running min of kl_self
    int32_t* pm_cross = smem + S;       // [S]
    int32_t* tmp_a = smem + 2 * S;      // [S] scan ping-pong
    int32_t* tmp_b = smem + 3 * S;      // [S]
    const int b = blockIdx.x;
    const int64_t base = (int64_t)b * S;
    const int n_tiles = (S + 31) / 32;

    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const int64_t id = ids[base + t];
        const bool special = (id == pad_id) || (id == eos_id);
        const int e = special ? 0 : (t % P) + 1;
        // behaviour token of this token's item
        const int first = (t / P) * P;
        const int64_t btok = ids[base + first];
        int a;
        if (btok >= 0 && btok < vocab && lut[btok] >= 0) {
            a = lut[btok] + 1;
        } else {
            a = 0;
            if (!special && t == first) atomicAdd(bad_token, 1);
        }
        if (special) a = 0;
        expert[base + t] = e;
        act_idx[base + t] = a;
        beh_idx[base + t] = (t % P == 0) ? 0 : a;
        const bool keep = attn_mask ? (attn_mask[base + t] != 0) : true;
        int64_t lv64 = actions ? actions[base + t] : 0;
        if (lv64 > 0x3fffffff) lv64 = 0x3fffffff;
        if (lv64 < -0x3fffffff) lv64 = -0x3fffffff;
        const int lv = (int)lv64;
        const int ks = keep ? 0 : INT_BIG;
        const int kc = keep ? lv : INT_BIG;
        kl_self[base + t] = ks;
        kl_cross[base + t] = kc;
        ql_cross[base + t] = lv;
        pm_self[t] = ks;
        pm_cross[t] = kc;
    }
    __syncthreads();
    // inclusive prefix-min (Hillis-Steele) for both arrays
    for (int which = 0; which < 2; ++which) {
        int32_t* src = which ? pm_cross : pm_self;
        int32_t* a = src;
        int32_t* bb = tmp_a;
        for (int off = 1; off < S; off <<= 1) {
            for (int t = threadIdx.x; t < S; t += blockDim.x) {
                int v = a[t];
                if (t >= off) v = min(v, a[t - off]);
                bb[t] = v;
            }
            __syncthreads();
            int32_t* nx = (bb == tmp_a) ? tmp_b : tmp_a;
            a = bb;
            bb = nx;
        }
        // copy back if result is not in src
        if (a != src) {
            for (int t = threadIdx.x; t < S; t += blockDim.x) src[t] = a[t];
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const int lv = ql_cross[base + t];      // written by this same thread above
        const int es = (pm_self[t] < 1) ? 0 : 1;
        const int ec = (pm_cross[t] < lv) ? 0 : 1;
        empty_self[base + t] = es;
        empty_cross[base + t] = ec;
        tmp_a[t] = es;
        tmp_b[t] = ec;
    }
    __syncthreads();
    for (int qt = threadIdx.x; qt < n_tiles; qt += blockDim.x) {
        int es = 0, ec = 0;
        for (int t = qt * 32; t < min(S, qt * 32 + 32); ++t) {
            es |= tmp_a[t];
            ec |= tmp_b[t];
        }
        tile_empty_self[(int64_t)b * n_tiles + qt] = es;
        tile_empty_cross[(int64_t)b * n_tiles + qt] = ec;
    }
}

// ---- session spans (Qwen3SessionMulti) ------------------------------------------------------
// ref:SeqRec/models/generative/Qwen3SessionMulti/model.py:545-551 (in-item mask), :556-613 (cross mask),
// :676-728 (self mask), :983-984 (RoPE positions = extended_session_ids).
// The reference masks key j for query i unless sess[j] < sess[i] (plus, for the self attention, the keys of the
// query's own item up to the query).  Session ids that do not decrease along the kept tokens make
// {kept j : sess[j] < sess[i]} a prefix of the kept keys, so the masks become per-query spans:
//   lim_i = 1 + last kept j with sess[j] < sess[i]   (0 if none)
//   self : j <= i, minus the hole [lim_i, first token of i's item)        cross: j <= lim_i - 1 (and kl[j] < ql[i])
// One workgroup per sequence; each row scans the sequence in LDS (S^2 integer compares, ~0.1 ms at B = 1024).
// Rows whose span would not be causal / not a prefix (ids out of order) are counted in `violations` and clamped.
__global__ void __launch_bounds__(ROUTER_THREADS)
session_span_kernel(const int64_t* __restrict__ sess, const int64_t* __restrict__ ext_ids,
                    const int64_t* __restrict__ attn_mask, const int32_t* __restrict__ kl_cross,
                    const int32_t* __restrict__ ql_cross, int S, int P, int n_pos,
                    int32_t* __restrict__ span_self, int32_t* __restrict__ span_cross, int32_t* __restrict__ pos_ids,
                    int32_t* __restrict__ empty_self, int32_t* __restrict__ empty_cross,
                    int32_t* __restrict__ tile_empty_self, int32_t* __restrict__ tile_empty_cross,
                    int32_t* __restrict__ violations) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sraw[];
    int64_t* sess_s = reinterpret_cast<int64_t*>(sraw);                 // [S]
    int32_t* keep_s = reinterpret_cast<int32_t*>(sess_s + S);           // [S]
    int32_t* klc_s = keep_s + S;                                        // [S] cross key level (INT_MAX if padded)
    int32_t* es_s = klc_s + S;                                          // [S] empty flags for the tile reduction
    int32_t* ec_s = es_s + S;
    const int b = blockIdx.x;
    const int64_t base = (int64_t)b * S;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        sess_s[t] = sess[base + t];
        keep_s[t] = attn_mask ? (attn_mask[base + t] != 0 ? 1 : 0) : 1;
        klc_s[t] = kl_cross[base + t];
    }
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const int64_t si = sess_s[i];
        int last_lt = -1, first_ge = INT_BIG;
        for (int j = 0; j < S; ++j) {
            if (keep_s[j]) {
                if (sess_s[j] < si) last_lt = j;
                else first_ge = min(first_ge, j);
            }
        }
        if (last_lt > first_ge || last_lt >= i) ++bad;            // not a prefix of the kept keys / not causal
        const int lim = min(last_lt + 1, i);
        const int istart = i - i % P;
        bool any_before = false;
        int klmin = INT_BIG;
        for (int j = 0; j < lim; ++j) {
            any_before |= keep_s[j] != 0;
            klmin = min(klmin, klc_s[j]);
        }
        bool any_item = false;
        for (int j = istart; j <= i; ++j) any_item |= keep_s[j] != 0;
        const int es = (any_before || any_item) ? 0 : 1;
        const int ec = (klmin < ql_cross[base + i]) ? 0 : 1;
        const bool hole = lim < istart;
        reinterpret_cast<int4*>(span_self)[base + i] = make_int4(i, hole ? lim : INT_BIG, hole ? istart : 0, 0);
        reinterpret_cast<int4*>(span_cross)[base + i] = make_int4(lim - 1, INT_BIG, 0, 0);
        empty_self[base + i] = es;
        empty_cross[base + i] = ec;
        es_s[i] = es;
        ec_s[i] = ec;
        int64_t p = ext_ids ? ext_ids[base + i] : (int64_t)i;
        if (p < 0 || p >= n_pos) { ++bad; p = p < 0 ? 0 : n_pos - 1; }
        pos_ids[base + i] = (int32_t)p;
    }
    if (bad) atomicAdd(violations, bad);
    __syncthreads();
    const int n_tiles = (S + 31) / 32;
    for (int qt = threadIdx.x; qt < n_tiles; qt += blockDim.x) {
        int es = 0, ec = 0;
        for (int t = qt * 32; t < min(S, qt * 32 + 32); ++t) {
            es |= es_s[t];
            ec |= ec_s[t];
        }
        tile_empty_self[(int64_t)b * n_tiles + qt] = es;
        tile_empty_cross[(int64_t)b * n_tiles + qt] = ec;
    }
}

// ---- expert lists ---------------------------------------------------------------------------
// pass 1: per-sequence member counts   work[b*E + e]
__global__ void expert_count_kernel(const int32_t* __restrict__ expert, int S, int E,
                                    int32_t* __restrict__ work) {
    __shared__ int cnt[64];
    const int b = blockIdx.x;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        atomicAdd(&cnt[expert[(int64_t)b * S + t]], 1);
    }
    __syncthreads();
    if (threadIdx.x < E) work[(int64_t)b * E + threadIdx.x] = cnt[threadIdx.x];
}

// pass 2 (single workgroup): exclusive scan over sequences per expert, then expert bases.
// work[b*E+e] <- global start slot of (sequence b, expert e); offsets[e] = segment starts.
__global__ void __launch_bounds__(1024)
expert_scan_kernel(int B, int E, int T, int32_t* __restrict__ work, int32_t* __restrict__ offsets) {
    __shared__ int32_t sh[1024];
    __shared__ int32_t carry;
    __shared__ int32_t totals[65];
    const int tid = threadIdx.x;
    for (int e = 0; e < E; ++e) {
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int b0 = 0; b0 < B; b0 += 1024) {
            const int b = b0 + tid;
            const int v = (b < B) ? work[(int64_t)b * E + e] : 0;
            sh[tid] = v;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                int add = (tid >= off) ? sh[tid - off] : 0;
                __syncthreads();
                sh[tid] += add;
                __syncthreads();
            }
            const int incl = sh[tid];
            const int c = carry;
            if (b < B) work[(int64_t)b * E + e] = c + incl - v;   // exclusive, within this expert
            __syncthreads();
            if (tid == 1023) carry = c + incl;
            __syncthreads();
        }
        if (tid == 0) totals[e] = carry;
        __syncthreads();
    }
    __shared__ int32_t sbase[65];
    if (tid == 0) {
        int acc = 0;
        for (int e = 0; e < E; ++e) {
            offsets[e] = acc;
            sbase[e] = acc;
            acc += totals[e];
        }
        offsets[E] = acc;   // == T
    }
    __syncthreads();
    // add the expert base
    for (int i = tid; i < B * E; i += 1024) {
        work[i] += sbase[i % E];
    }
}

// pass 3: one wave per sequence; lane e (< E) walks the sequence in token order and emits slots.
__global__ void expert_emit_kernel(const int32_t* __restrict__ expert, int S, int E,
                                   const int32_t* __restrict__ work,
                                   int32_t* __restrict__ perm, int32_t* __restrict__ slot) {
    const int b = blockIdx.x;
    const int e = threadIdx.x;
    if (e >= E) return;
    int next = work[(int64_t)b * E + e];
    const int64_t base = (int64_t)b * S;
    for (int t = 0; t < S; ++t) {
        if (expert[base + t] == e) {
            perm[next] = (int32_t)(base + t);
            slot[base + t] = next;
            ++next;
        }
    }
}

// Labels outside [0, V) that are not ignore_index: torch's CrossEntropyLoss raises on them (a tokenizer / vocab_size
// mismatch); the CE kernels skip such rows, so they are counted here and the host raises (Engine.check_inputs).
__global__ void __launch_bounds__(256)
check_labels_kernel(const int64_t* __restrict__ labels, int64_t n, int V, int ignore_index, int32_t* __restrict__ bad) {
    int cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t y = labels[i];
        cnt += (y != ignore_index && (y < 0 || y >= V)) ? 1 : 0;
    }
    if (__any(cnt)) {
        cnt = wave_sum_i32(cnt);
        if ((threadIdx.x & 63) == 0) atomicAdd(bad, cnt);
    }
}

}  // namespace gamer

using namespace gamer;

extern "C" int gamer_abi_version(void) { return GAMER_ABI_VERSION; }
extern "C" int gamer_reload_env(void) { gamer::g_env_epoch.fetch_add(1, std::memory_order_acq_rel); return 0; }
extern "C" const char* gamer_last_error(void) { return gamer::g_err; }

extern "C" int gamer_router_fwd(const int64_t* ids, const int64_t* attn_mask, const int64_t* actions,
                                const int32_t* behavior_lut, int vocab, int B, int S, int num_positions,
                                int pad_id, int eos_id,
                                int32_t* expert, int32_t* beh_idx, int32_t* act_idx,
                                int32_t* kl_self, int32_t* kl_cross, int32_t* ql_cross,
                                int32_t* empty_self, int32_t* empty_cross,
                                int32_t* tile_empty_self, int32_t* tile_empty_cross,
                                int32_t* bad_token, void* stream) {
    GAMER_CHECK_ARG(ids && behavior_lut && expert && beh_idx && act_idx && kl_self && kl_cross &&
                    ql_cross && empty_self && empty_cross && tile_empty_self && tile_empty_cross && bad_token,
                    "gamer_router_fwd: null pointer");
    GAMER_CHECK_ARG(B > 0 && S > 0 && num_positions > 0, "gamer_router_fwd: bad shape B=%d S=%d P=%d", B, S, num_positions);
    GAMER_CHECK_ARG(S <= 8192, "gamer_router_fwd: S=%d > 8192 unsupported", S);
    const size_t shmem = (size_t)4 * S * sizeof(int32_t);
    hipLaunchKernelGGL(router_kernel, dim3(B), dim3(ROUTER_THREADS), shmem, (hipStream_t)stream,
                       ids, attn_mask, actions, behavior_lut, vocab, S, num_positions, pad_id, eos_id,
                       expert, beh_idx, act_idx, kl_self, kl_cross, ql_cross, empty_self, empty_cross,
                       tile_empty_self, tile_empty_cross, bad_token);
    GAMER_CHECK_LAUNCH("gamer_router_fwd");
    return 0;
}

extern "C" int gamer_session_spans(const int64_t* session_ids, const int64_t* extended_session_ids,
                                   const int64_t* attn_mask, const int32_t* kl_cross, const int32_t* ql_cross,
                                   int B, int S, int num_positions, int n_rope_positions,
                                   int32_t* span_self, int32_t* span_cross, int32_t* pos_ids,
                                   int32_t* empty_self, int32_t* empty_cross,
                                   int32_t* tile_empty_self, int32_t* tile_empty_cross, int32_t* violations,
                                   void* stream) {
    GAMER_CHECK_ARG(session_ids && kl_cross && ql_cross && span_self && span_cross && pos_ids && empty_self &&
                    empty_cross && tile_empty_self && tile_empty_cross && violations,
                    "gamer_session_spans: null pointer");
    GAMER_CHECK_ARG(B > 0 && S > 0 && num_positions > 0 && n_rope_positions > 0,
                    "gamer_session_spans: bad shape B=%d S=%d P=%d n_pos=%d", B, S, num_positions, n_rope_positions);
    GAMER_CHECK_ARG(S <= 2048, "gamer_session_spans: S=%d > 2048 unsupported", S);
    const size_t shmem = (size_t)S * (sizeof(int64_t) + 4 * sizeof(int32_t));
    hipLaunchKernelGGL(session_span_kernel, dim3(B), dim3(ROUTER_THREADS), shmem, (hipStream_t)stream,
                       session_ids, extended_session_ids, attn_mask, kl_cross, ql_cross, S, num_positions,
                       n_rope_positions, span_self, span_cross, pos_ids, empty_self, empty_cross,
                       tile_empty_self, tile_empty_cross, violations);
    GAMER_CHECK_LAUNCH("gamer_session_spans");
    return 0;
}

extern "C" int gamer_expert_lists(const int32_t* expert, int B, int S, int num_experts,
                                  int32_t* perm, int32_t* slot, int32_t* offsets, int32_t* work, void* stream) {
    GAMER_CHECK_ARG(expert && perm && slot && offsets && work, "gamer_expert_lists: null pointer");
    GAMER_CHECK_ARG(B > 0 && S > 0 && num_experts > 0 && num_experts <= 64,
                    "gamer_expert_lists: bad shape B=%d S=%d E=%d", B, S, num_experts);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(expert_count_kernel, dim3(B), dim3(256), 0, st, expert, S, num_experts, work);
    GAMER_CHECK_LAUNCH("gamer_expert_lists/count");
    hipLaunchKernelGGL(expert_scan_kernel, dim3(1), dim3(1024), 0, st, B, num_experts, B * S, work, offsets);
    GAMER_CHECK_LAUNCH("gamer_expert_lists/scan");
    hipLaunchKernelGGL(expert_emit_kernel, dim3(B), dim3(64), 0, st, expert, S, num_experts, work, perm, slot);
    GAMER_CHECK_LAUNCH("gamer_expert_lists/emit");
    return 0;
}

extern "C" int gamer_check_labels(const int64_t* labels, int64_t n, int V, int ignore_index, int32_t* bad_label,
                                  void* stream) {
    GAMER_CHECK_ARG(labels && bad_label && n > 0 && V > 0, "gamer_check_labels: bad arguments");
    int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(check_labels_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, labels, n, V,
                       ignore_index, bad_label);
    GAMER_CHECK_LAUNCH("gamer_check_labels");
    return 0;
}
