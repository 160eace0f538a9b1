// Multi-behaviour flash attention for gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32), head_dim 64.
//
// Replaces sdpa_attention_forward + the two additive [B,1,S,S] masks of the reference
// (ref:SeqRec/models/generative/Qwen3Multi/model.py:133-143, 573-630, 691-741): the masks are the
// predicate  allowed(i,j) = j<=i && kl[j] < ql[i]  evaluated in registers (the Qwen3SessionMulti variant adds a
// per-query key span, see QuerySpan), and a query row that has
// no allowed key ("empty" row, flagged by the router kernel) gets p = 1/S over ALL S keys exactly as
// the reference's finfo.min mask does, in the forward and in both backward kernels.
//
// Data flow per workgroup (4 waves).  G = nq/nkv query heads share one K/V head:
//   fwd / dq : workgroup = (b, kv head, tile of (4/G)*32 queries); wave = (query head, 32 queries);
//              K/V tiles of 32 keys stream through LDS (double buffered), shared by the 4 waves.
//              S^T = K Q^T puts the query on the lane, so the softmax state is lane-local and the
//              accumulator registers are directly the B operand of the second MFMA
//              (O^T += V^T P^T,  dQ^T += K^T dS^T): no LDS round trip for P.
//   dkv      : workgroup = (b, kv head, tile of (4/G)*32 keys); wave = (query head, 32 keys);
//              Q/dO tiles of 32 queries stream through LDS; S = Q K^T puts the key on the lane and
//              dV^T += dO^T P, dK^T += Q^T dS accumulate in registers; the G heads are summed
//              through LDS at the end.  No atomics anywhere (dQ is recomputed in its own kernel).
#include "attention_common.h"

namespace gamer {

// Optional per-workgroup trace (debug aid for residency studies): when g_trace != nullptr, wave 0 of
// every forward workgroup records {realtime start, realtime end, HW_ID, XCC_ID} (4 x u64 per workgroup).
__device__ unsigned long long* g_trace = nullptr;


// =============================================================================================
// forward
// =============================================================================================
#define GAMER_QK_TILE(ACC, KSLOT)                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                                \
        const float4 kf = *reinterpret_cast<const float4*>(&Ks[KSLOT][r * KLD + 8 * kk + 4 * h]);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kk][0], ACC, 0, 0, 0);                     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kk][1], ACC, 0, 0, 0);                     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kk][2], ACC, 0, 0, 0);                     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kk][3], ACC, 0, 0, 0);                     \
    }

template <int G, bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
attn_fwd_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
              const float* __restrict__ v, int ldv, const int32_t* __restrict__ kl,
              const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
              int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
              float* __restrict__ o, float* __restrict__ lse, const RowOrder ro, const int pair, const int qtile,
              const int uspan) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * 64];
    __shared__ __attribute__((aligned(16))) KeyMeta kms[3];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const int slot = q0 + sub * 32 + r;                 // sorted slot of this lane's query row
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iq_raw = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int iq = valid_q ? iq_raw : S;                // position in the sequence (S: beyond every key)
    const int iqc = iq_raw;
    const int64_t tok = (int64_t)b * S + iqc;

    // scores are kept in the log2 domain: q is pre-scaled by scale*log2(e), p = exp2(s - m_ref)
    unsigned long long* trace = g_trace;
    unsigned long long t_start = 0;
    if (trace) t_start = __builtin_amdgcn_s_memrealtime();
    const float qs = scale * 1.4426950408889634f;
    float qf[8][4];
    {
        const float* qrow = q + tok * ldq + head * 64;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 t4 = *reinterpret_cast<const float4*>(qrow + 8 * kk + 4 * h);
            qf[kk][0] = t4.x * qs; qf[kk][1] = t4.y * qs; qf[kk][2] = t4.z * qs; qf[kk][3] = t4.w * qs;
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq_raw, valid_q, S);
    const int wave_ql_min = wave_min_i32_dpp(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32_dpp(normal ? sp.hi : INT_BIG_A);   // every key up to here passes every row's limit
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;          // no normal row: no scores needed at all
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(normal ? sp.hole_hi : 0) : 0;
    const int n_all = (S + 31) / 32;
    int wave_q_hi, n_causal;
    bool wave_has_empty;
    if (ORD) {
        const int64_t tb = (int64_t)b * n_all + (q0 >> 5);
        const int wt = min(sub, n_all - 1 - (q0 >> 5));          // tiles past the end: reuse the last one
        wave_q_hi = (q0 + sub * 32 < S) ? ro.tile_maxpos[tb + wt] : -1;
        wave_has_empty = (q0 + sub * 32 < S) && (ro.tile_kind[tb + wt] & 2) != 0;
        int hi = -1;
#pragma unroll
        for (int ss = 0; ss < NSUB; ++ss)
            if (q0 + ss * 32 < S) hi = max(hi, ro.tile_maxpos[tb + ss]);
        n_causal = hi < 0 ? 0 : (hi >> 5) + 1;
    } else {
        wave_q_hi = wave_all_empty ? -1 : min(S - 1, q0 + sub * 32 + 31);
        wave_has_empty = __any(my_empty ? 1 : 0) != 0;
        n_causal = (min(S, q0 + R) + 31) / 32;
    }
    const bool block_has_empty = __syncthreads_or(my_empty ? 1 : 0) != 0;
    const int n_iter = block_has_empty ? n_all : n_causal;
    // An empty row is uniform over my_span keys: all S in training (uspan = S); when an evaluation loop re-runs
    // the whole sequence, over the keys that existed when the reference's cached decode computed the row
    // (uspan = prompt length for prompt rows, i + 1 for generated rows; model.py:603-617)
    const int my_span = min(S, max(uspan, iq + 1));
    const float invS = 1.f / (float)my_span;
    const AttnDropout rng(p_drop, seed);
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;

    const float* kbase = k + (int64_t)b * S * ldk + kvh * 64;
    const float* vbase = v + (int64_t)b * S * ldv + kvh * 64;
    const int32_t* klb = kl + (int64_t)b * S;

    float m_ref = 0.f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }

    // K tiles are staged one iteration earlier than V tiles (the scores of tile t+1 are taken at the end of
    // iteration t): K slot t&1 is loaded two iterations ahead, V slot t&1 one ahead, key metadata ring of 3.
    float4 rk[2], rv[2];
    int rmeta;
    load_tile32(kbase, ldk, 0, S, tid, rk);
    load_tile32(vbase, ldv, 0, S, tid, rv);
    rmeta = key_meta_load<DROP>(klb, 0, S, w, lane, rng);
    store_tile32<KLD>(Ks[0], tid, rk, S);
    store_tile32<64>(Vs[0], tid, rv, S);
    key_meta_store(kms[0], w, lane, rmeta);
    if (n_iter > 1) {
        load_tile32(kbase, ldk, 32, S, tid, rk);
        rmeta = key_meta_load<DROP>(klb, 32, S, w, lane, rng);
        store_tile32<KLD>(Ks[1], tid, rk, S - 32);
        key_meta_store(kms[1], w, lane, rmeta);
    }
    __syncthreads();

    f32x16 st_cur;
#pragma unroll
    for (int i = 0; i < 16; ++i) st_cur[i] = 0.f;
    if (!(0 > wave_q_hi)) { GAMER_QK_TILE(st_cur, 0) }
    // K(0) must stay intact until every wave has taken its first scores: a wave without work in this tile
    // (rows past the end of the sequence, or empty rows only) reaches the K(2) store of iteration 0 at once
    __syncthreads();

    for (int jt = 0; jt < n_iter; ++jt) {
        const int j0 = jt * 32;
        const bool have_next = jt + 1 < n_iter;
        const bool have_next2 = jt + 2 < n_iter;
        if (have_next2) {
            load_tile32(kbase, ldk, j0 + 64, S, tid, rk);
            rmeta = key_meta_load<DROP>(klb, j0 + 64, S, w, lane, rng);
        }
        if (have_next) load_tile32(vbase, ldv, j0 + 32, S, tid, rv);

        const bool beyond = j0 > wave_q_hi;                // every key of the tile is in every row's future
        const bool beyond_next = (j0 + 32) > wave_q_hi;
        if (!(beyond && !wave_has_empty)) {
            const KeyMeta& km = kms[jt % 3];
            const float* Vc = Vs[jt & 1];
            const bool full_span = (j0 + 32 <= uspan) || uspan >= S;   // every key of the tile is inside every span
            // O^T[d][query] += sum_key V[key][d] * P[query][key]   (rows of V past the sequence end are zero).
            // Called at the end of every variant below so that each one is a straight-line path (a join in
            // front of the MFMAs costs 16 register copies per tile, and VALU time is not hidden here).
            auto pv_tile = [&]() {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int key = rowmap(reg, h);
                    const float a0 = Vc[key * 64 + r];
                    const float a1 = Vc[key * 64 + 32 + r];
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, st_cur[reg], oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, st_cur[reg], oacc[1], 0, 0, 0);
                }
            };
            // MASK: evaluate the causal + level predicate per element (tiles on the diagonal, tiles holding a
            // blocked key); EMPTYSEL: some rows of the wave are "empty" rows (p = 1, normalised by 1/S at the end)
            auto softmax_tile = [&]<bool MASK, bool EMPTYSEL>() {
                if (MASK) {
                    int klv[16];
                    read_key_quads(km.kl, h, klv);
                    const int t_pos = sp.hi - j0 - 4 * h;        // key (reg&3)+8*(reg>>2) of the tile is <= hi (= iq)
                    const int t_lo = sp.hole_lo - j0 - 4 * h, t_hi = sp.hole_hi - j0 - 4 * h;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int ko = (reg & 3) + 8 * (reg >> 2);
                        bool allowed = (ko <= t_pos) & (klv[reg] < my_ql);
                        if (SPAN) allowed = allowed & !((ko >= t_lo) & (ko < t_hi));
                        st_cur[reg] = allowed ? st_cur[reg] : -INFINITY;
                    }
                }
                float mloc = st_cur[0];
#pragma unroll
                for (int reg = 1; reg < 16; ++reg) mloc = fmaxf(mloc, st_cur[reg]);
                mloc = xor32_max(mloc);
                bool need = (l_run == 0.f) ? (mloc > -INFINITY) : (mloc > RESCALE_TAU);
                if (EMPTYSEL) need = need && !my_empty;
                if (__any(need ? 1 : 0)) {
                    const float d = need ? mloc : 0.f;
                    const float alpha = (l_run == 0.f) ? 1.f : __builtin_amdgcn_exp2f(-d);
                    m_ref += d;
                    l_run *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; st_cur[i] -= d; }
                }
                float rowsum = 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float pe = __builtin_amdgcn_exp2f(st_cur[reg]);            // masked -> exp2(-inf) = 0
                    if (EMPTYSEL) {
                        const float one = (full_span || (j0 + rowmap(reg, h) < my_span)) ? 1.f : 0.f;
                        pe = my_empty ? one : pe;
                    }
                    rowsum += pe;
                    st_cur[reg] = pe;
                }
                rowsum = xor32_sum(rowsum);
                l_run += rowsum;
                if (DROP) {
                    int kwv[16];
                    read_key_quads(reinterpret_cast<const int32_t*>(km.kw), h, kwv);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        st_cur[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? st_cur[reg] : 0.f;
                }
                pv_tile();
            };
            if (wave_all_empty) {
                // every row of the wave is an empty row: P = keep / span, no scores, no softmax state
                if (DROP) {
                    int kwv[16];
                    read_key_quads(reinterpret_cast<const int32_t*>(km.kw), h, kwv);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) st_cur[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? 1.f : 0.f;
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) st_cur[reg] = 1.f;
                }
                if (!full_span) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) st_cur[reg] = (j0 + rowmap(reg, h) < my_span) ? st_cur[reg] : 0.f;
                }
                pv_tile();
            } else if (wave_has_empty) {
                softmax_tile.template operator()<true, true>();
            } else {
                const int klmax = __builtin_amdgcn_readfirstlane(km.klmax);
                const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                       (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
                if (free_tile) softmax_tile.template operator()<false, false>();
                else softmax_tile.template operator()<true, false>();
            }
        }
        if (have_next && !beyond_next) {
            // scores of the next tile (minus the reference), taken BEFORE the barrier: after it other waves may
            // already overwrite this K slot with K(jt+3)
            const float init = -m_ref;
#pragma unroll
            for (int i = 0; i < 16; ++i) st_cur[i] = init;
            if ((jt + 1) & 1) { GAMER_QK_TILE(st_cur, 1) } else { GAMER_QK_TILE(st_cur, 0) }
        }
        if (have_next2) {
            store_tile32<KLD>(Ks[jt & 1], tid, rk, S - (j0 + 64));   // K(jt) is dead: its scores were taken last iteration
            key_meta_store(kms[(jt + 2) % 3], w, lane, rmeta);
        }
        if (have_next) store_tile32<64>(Vs[(jt + 1) & 1], tid, rv, S - (j0 + 32));
        __syncthreads();
    }

    if (valid_q) {
        float linv = my_empty ? invS : (l_run > 0.f ? 1.f / l_run : 0.f);
        if (DROP) linv *= rng.scale;
        float* orow = o + tok * (int64_t)nq * 64 + head * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = oacc[dh][4 * g4 + 0] * linv; t4.y = oacc[dh][4 * g4 + 1] * linv;
                t4.z = oacc[dh][4 * g4 + 2] * linv; t4.w = oacc[dh][4 * g4 + 3] * linv;
                *reinterpret_cast<float4*>(orow + 32 * dh + 8 * g4 + 4 * h) = t4;
            }
        }
        if (trace && tid == 0) {
            unsigned long long* rec = trace + 4ull * ((size_t)pair * ((S + R - 1) / R) + qtile);
            rec[0] = t_start;
            rec[1] = __builtin_amdgcn_s_memrealtime();
            rec[2] = __builtin_amdgcn_s_getreg(63492);       // HW_REG_HW_ID
            rec[3] = __builtin_amdgcn_s_getreg(63508);       // HW_REG_XCC_ID
        }
        // natural-log LSE of the scaled scores (what the backward kernels consume)
        if (h == 0) lse[((int64_t)b * nq + head) * S + iqc] =
            my_empty ? 0.f : (m_ref + __log2f(l_run)) * 0.6931471805599453f;
    }
}

template <int G, bool DROP, bool ORD, bool SPAN>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                const float* __restrict__ v, int ldv, const int32_t* __restrict__ kl,
                const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                float* __restrict__ o, float* __restrict__ lse, const RowOrder ro, const int uspan) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        // With the row order the tiles of a sequence are NOT a causal ramp (normal rows first: short key ranges, then the
        // empty rows: all S keys, one product), so "tile u + tile n-1-u" is not equal work for every u: u = 0 pairs a
        // short tile with a long one, the middle u two long ones.  The workgroups that share a pair still cover every u in
        // every round, but each takes a different u from round to round: over its ~24 pairs every workgroup sees the mix.
        const int halves = (n_tiles + 1) >> 1;
        const int uu = ORD ? (wl.u + it) % halves : wl.u;
        const int heavy = n_tiles - 1 - uu, light = uu;           // later query tiles see more keys
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {                    // one inlined body (register pressure)
            if (pass == 1 && light == heavy) break;
            attn_fwd_tile<G, DROP, ORD, SPAN>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, S, nq, nkv, scale, p_drop, seed, o, lse,
                                        ro, pair, pass == 0 ? heavy : light, uspan);
        }
    }
}

// =============================================================================================
// backward: dQ
// =============================================================================================
template <int G, bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
attn_bwd_dq_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                   const float* __restrict__ v, int ldv, const float* __restrict__ o, const float* __restrict__ d_o,
                   const float* __restrict__ lse, float* __restrict__ delta,
                   const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                   const int32_t* __restrict__ row_empty, int S, int nq, int nkv, float scale,
                   float p_drop, uint64_t seed, float* __restrict__ dq, int lddq, const RowOrder ro, const int pair,
                   const int qtile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * KLD];
    __shared__ __attribute__((aligned(16))) KeyMeta kms[2];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const int slot = q0 + sub * 32 + r;                 // sorted slot of this lane's query row
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iq_raw = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int iqc = iq_raw;
    const int64_t tok = (int64_t)b * S + iqc;

    // log2 domain: q is pre-scaled by scale*log2(e) and the score accumulators start at -lse*log2(e), so
    // p = exp2(accumulator)
    const float qs = scale * 1.4426950408889634f;
    float qf[8][4], dof[8][4];
    float my_delta = 0.f;                 // delta_i = dO_i . O_i, computed here and published for the dK/dV kernel
    {
        const float* qrow = q + tok * ldq + head * 64;
        const float* drow = d_o + tok * (int64_t)nq * 64 + head * 64;
        const float* orow = o + tok * (int64_t)nq * 64 + head * 64;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 t4 = *reinterpret_cast<const float4*>(qrow + 8 * kk + 4 * h);
            qf[kk][0] = t4.x * qs; qf[kk][1] = t4.y * qs; qf[kk][2] = t4.z * qs; qf[kk][3] = t4.w * qs;
            const float4 u4 = *reinterpret_cast<const float4*>(drow + 8 * kk + 4 * h);
            dof[kk][0] = u4.x; dof[kk][1] = u4.y; dof[kk][2] = u4.z; dof[kk][3] = u4.w;
            const float4 o4 = *reinterpret_cast<const float4*>(orow + 8 * kk + 4 * h);
            my_delta += u4.x * o4.x + u4.y * o4.y + u4.z * o4.z + u4.w * o4.w;
        }
        my_delta = xor32_sum(my_delta);
        if (valid_q && h == 0) delta[((int64_t)b * nq + head) * S + iqc] = my_delta;
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq_raw, valid_q, S);
    const float neg_lse2 = -lse[((int64_t)b * nq + head) * S + iqc] * 1.4426950408889634f;
    const int wave_ql_min = wave_min_i32_dpp(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32_dpp(normal ? sp.hi : INT_BIG_A);
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(normal ? sp.hole_hi : 0) : 0;
    const int n_all = (S + 31) / 32;
    int wave_q_hi, n_causal;
    bool wave_has_empty;
    if (ORD) {
        const int64_t tb = (int64_t)b * n_all + (q0 >> 5);
        const int wt = min(sub, n_all - 1 - (q0 >> 5));          // tiles past the end: reuse the last one
        wave_q_hi = (q0 + sub * 32 < S) ? ro.tile_maxpos[tb + wt] : -1;
        wave_has_empty = (q0 + sub * 32 < S) && (ro.tile_kind[tb + wt] & 2) != 0;
        int hi = -1;
#pragma unroll
        for (int ss = 0; ss < NSUB; ++ss)
            if (q0 + ss * 32 < S) hi = max(hi, ro.tile_maxpos[tb + ss]);
        n_causal = hi < 0 ? 0 : (hi >> 5) + 1;
    } else {
        wave_q_hi = wave_all_empty ? -1 : min(S - 1, q0 + sub * 32 + 31);
        wave_has_empty = __any(my_empty ? 1 : 0) != 0;
        n_causal = (min(S, q0 + R) + 31) / 32;
    }
    const bool block_has_empty = __syncthreads_or(my_empty ? 1 : 0) != 0;
    const int n_iter = block_has_empty ? n_all : n_causal;
    const float invS = 1.f / (float)S;
    const AttnDropout rng(p_drop, seed);
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;
    const float sd = rng.scale;
    const float neg_delta = -my_delta;

    const float* kbase = k + (int64_t)b * S * ldk + kvh * 64;
    const float* vbase = v + (int64_t)b * S * ldv + kvh * 64;
    const int32_t* klb = kl + (int64_t)b * S;

    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }

    float4 rk[2], rv[2];
    int rmeta;
    load_tile32(kbase, ldk, 0, S, tid, rk);
    load_tile32(vbase, ldv, 0, S, tid, rv);
    rmeta = key_meta_load<DROP>(klb, 0, S, w, lane, rng);
    store_tile32<KLD>(Ks[0], tid, rk, S);
    store_tile32<KLD>(Vs[0], tid, rv, S);
    key_meta_store(kms[0], w, lane, rmeta);
    __syncthreads();

    for (int jt = 0; jt < n_iter; ++jt) {
        const int cur = jt & 1;
        const bool more = jt + 1 < n_iter;
        const int j0 = jt * 32;
        if (more) {
            load_tile32(kbase, ldk, j0 + 32, S, tid, rk);
            load_tile32(vbase, ldv, j0 + 32, S, tid, rv);
            rmeta = key_meta_load<DROP>(klb, j0 + 32, S, w, lane, rng);
        }
        const bool beyond = j0 > wave_q_hi;
        if (!(beyond && !wave_has_empty)) {
            const KeyMeta& km = kms[cur];
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = neg_lse2; dp[i] = 0.f; }
            // S^T = K Q^T and dP^T[key][query] = sum_d V[key][d] dO[query][d]: two independent accumulation chains,
            // issued alternately (a single dependent chain of this MFMA runs 5-13 % below the pipe rate)
            if (!beyond) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 kf = *reinterpret_cast<const float4*>(&Ks[cur][r * KLD + 8 * kk + 4 * h]);
                    const float4 vf = *reinterpret_cast<const float4*>(&Vs[cur][r * KLD + 8 * kk + 4 * h]);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kk][0], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, dof[kk][0], dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kk][1], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, dof[kk][1], dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kk][2], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, dof[kk][2], dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kk][3], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, dof[kk][3], dp, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 vf = *reinterpret_cast<const float4*>(&Vs[cur][r * KLD + 8 * kk + 4 * h]);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, dof[kk][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, dof[kk][1], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, dof[kk][2], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, dof[kk][3], dp, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of the later phases below this point
            // u = mult * dP - delta  (mult = keep / (1 - p))
            if (DROP) {
                int kwv[16];
                read_key_quads(reinterpret_cast<const int32_t*>(km.kw), h, kwv);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float t = rng.keep(aw, (uint32_t)kwv[reg]) ? dp[reg] : 0.f;
                    dp[reg] = fmaf(t, sd, neg_delta);
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) dp[reg] += neg_delta;
            }
            // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]   (rows of K past the sequence end are zero)
            auto dq_tile = [&](const f32x16& ds) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int key = rowmap(reg, h);
                    const float a0 = Ks[cur][key * KLD + r];
                    const float a1 = Ks[cur][key * KLD + 32 + r];
                    dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ds[reg], dqacc[0], 0, 0, 0);
                    dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ds[reg], dqacc[1], 0, 0, 0);
                }
            };
            // dS^T = p * u ; an empty row has p = 1 here and 1/S in the final scale
            auto ds_tile = [&]<bool MASK, bool EMPTYSEL>() {
                if (MASK) {
                    int klv[16];
                    read_key_quads(km.kl, h, klv);
                    const int t_pos = sp.hi - j0 - 4 * h;
                    const int t_lo = sp.hole_lo - j0 - 4 * h, t_hi = sp.hole_hi - j0 - 4 * h;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int ko = (reg & 3) + 8 * (reg >> 2);
                        bool allowed = (ko <= t_pos) & (klv[reg] < my_ql);
                        if (SPAN) allowed = allowed & !((ko >= t_lo) & (ko < t_hi));
                        st[reg] = allowed ? st[reg] : -INFINITY;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float pe = __builtin_amdgcn_exp2f(st[reg]);
                    if (EMPTYSEL) pe = my_empty ? 1.f : pe;
                    st[reg] = pe * dp[reg];
                }
            };
            if (wave_all_empty) {
                st = dp;
            } else if (wave_has_empty) {
                ds_tile.template operator()<true, true>();
            } else {
                const int klmax = __builtin_amdgcn_readfirstlane(km.klmax);
                const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                       (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
                if (free_tile) ds_tile.template operator()<false, false>();
                else ds_tile.template operator()<true, false>();
            }
            __builtin_amdgcn_sched_barrier(0);
            dq_tile(st);
        }
        if (more) {
            store_tile32<KLD>(Ks[cur ^ 1], tid, rk, S - (j0 + 32));
            store_tile32<KLD>(Vs[cur ^ 1], tid, rv, S - (j0 + 32));
            key_meta_store(kms[cur ^ 1], w, lane, rmeta);
        }
        __syncthreads();
    }

    if (valid_q) {
        const float fs = my_empty ? scale * invS : scale;
        float* drow = dq + tok * lddq + head * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = dqacc[dh][4 * g4 + 0] * fs; t4.y = dqacc[dh][4 * g4 + 1] * fs;
                t4.z = dqacc[dh][4 * g4 + 2] * fs; t4.w = dqacc[dh][4 * g4 + 3] * fs;
                *reinterpret_cast<float4*>(drow + 32 * dh + 8 * g4 + 4 * h) = t4;
            }
        }
    }
}

template <int G, bool DROP, bool ORD, bool SPAN>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_bwd_dq_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                   const float* __restrict__ v, int ldv, const float* __restrict__ o, const float* __restrict__ d_o,
                   const float* __restrict__ lse, float* __restrict__ delta,
                   const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                   const int32_t* __restrict__ row_empty, int nbatch, int S, int nq, int nkv, float scale,
                   float p_drop, uint64_t seed, float* __restrict__ dq, int lddq, const RowOrder ro) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int heavy = n_tiles - 1 - wl.u, light = wl.u;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dq_tile<G, DROP, ORD, SPAN>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, S, nq, nkv, scale,
                                           p_drop, seed, dq, lddq, ro, pair, pass == 0 ? heavy : light);
        }
    }
}

// =============================================================================================
// backward: dK, dV
// =============================================================================================
template <int G>
struct DkvSmem {
    float Kt[(4 / G) * 32 * KLD];      // this workgroup's keys, fragment image (read every tile: keeps 64 VGPRs free)
    float Vt[(4 / G) * 32 * KLD];
    float Qs[G][32 * KLD];
    float dOs[G][32 * KLD];
    float nlse2_s[G][32];       // -lse * log2(e) of the staged query rows
    float ndelta_s[G][32];      // -delta
    uint32_t aw_s[G][32];       // dropout row words
    int32_t ql_s[32];
    int32_t empty_s[32];
    int32_t pos_s[32];          // key limit of the staged query rows: their position (QuerySpan::hi), S past the end
    int32_t hole_lo_s[32];      // QuerySpan hole of the staged query rows (SPAN kernels only)
    int32_t hole_hi_s[32];
    int32_t qlmin;              // smallest query level / key limit over the normal rows of the tile (INT_MAX: none)
    int32_t posmin;
    int32_t hole_lo_min;        // over the normal rows: smallest hole start, largest hole end
    int32_t hole_hi_max;
};

template <int G, bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
attn_bwd_dkv_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                    const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                    const float* __restrict__ lse, const float* __restrict__ delta,
                    const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                    const int32_t* __restrict__ row_empty, const int32_t* __restrict__ tile_empty,
                    int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                    float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, const RowOrder ro, const int pair,
                    const int ktile, float* __restrict__ ds_out) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char dkv_raw[];
    DkvSmem<G>& sm = *reinterpret_cast<DkvSmem<G>*>(dkv_raw);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    __syncthreads();                 // the previous tile's head reduction may still be reading the LDS region
    const int b = pair / nkv, kvh = pair % nkv, k0 = ktile * R;
    const int r = lane & 31, h = lane >> 5;
    const int jk = k0 + sub * 32 + r;                 // this lane's key
    const bool valid_k = jk < S;
    const int jkc = valid_k ? jk : S - 1;
    const int64_t ktok = (int64_t)b * S + jkc;

    // stage this workgroup's K and V rows (R keys) once
    for (int f = tid; f < R * 16; f += AT_THREADS) {
        const int row = f >> 4, c4 = (f & 15) << 2;
        const int j = k0 + row;
        float4 kv4 = make_float4(0.f, 0.f, 0.f, 0.f), vv4 = kv4;
        if (j < S) {
            kv4 = *reinterpret_cast<const float4*>(k + ((int64_t)b * S + j) * ldk + kvh * 64 + c4);
            vv4 = *reinterpret_cast<const float4*>(v + ((int64_t)b * S + j) * ldv + kvh * 64 + c4);
        }
        *reinterpret_cast<float4*>(&sm.Kt[row * KLD + c4]) = kv4;
        *reinterpret_cast<float4*>(&sm.Vt[row * KLD + c4]) = vv4;
    }
    const float* Kw = &sm.Kt[(sub * 32 + r) * KLD + 4 * h];
    const float* Vw = &sm.Vt[(sub * 32 + r) * KLD + 4 * h];
    const int my_kl = valid_k ? kl[ktok] : INT_BIG_A;
    const int wave_kl_max = wave_max_i32_dpp(my_kl);
    const int wave_k_lo = k0 + sub * 32, wave_k_hi = wave_k_lo + 31;
    const float invS = 1.f / (float)S;
    const float c2 = scale * 1.4426950408889634f;
    const AttnDropout rng(p_drop, seed);
    const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
    const float sd = rng.scale;
    const uint32_t drop_head0 = (uint32_t)(((int64_t)b * nq + kvh * G) * S);

    const int n_qt = (S + 31) / 32;
    const int32_t* te = tile_empty + (int64_t)b * n_qt;
    const int32_t* tkind = ORD ? ro.tile_kind + (int64_t)b * n_qt : nullptr;
    const int32_t* tmax = ORD ? ro.tile_maxpos + (int64_t)b * n_qt : nullptr;
    const int32_t* pmap = ORD ? ro.perm + (int64_t)b * S : nullptr;
    // a query tile matters to this key tile if it has an empty row (attends every key) or a normal row at
    // or after the first key
    auto tile_empty_rows = [&](int qt) { return ORD ? (tkind[qt] & 2) != 0 : te[qt] != 0; };
    auto tile_last_pos = [&](int qt) { return ORD ? tmax[qt] : min(S - 1, qt * 32 + 31); };

    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

    // staged registers: per head one Q tile and one dO tile (2 float4 each per thread)
    float4 rq[G][2], rdo[G][2];
    float rl = 0.f, rd = 0.f;
    uint32_t raw = 0;
    int rql = 1, rem = 0, rpos = 0, rqlmin = INT_BIG_A, rposmin = INT_BIG_A;
    int rhlo = INT_BIG_A, rhhi = 0, rhlomin = INT_BIG_A, rhhimax = 0;

    auto next_tile = [&](int qt) {
        while (qt < n_qt && !tile_empty_rows(qt) && tile_last_pos(qt) < k0) ++qt;
        return qt;
    };
    auto load_q_tile = [&](int qt) {
        const int i0 = qt * 32;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int hd = kvh * G + g;
            if (ORD) {
                load_tile32_rows(q + (int64_t)b * S * ldq + hd * 64, ldq, pmap, i0, S, tid, rq[g]);
                load_tile32_rows(d_o + (int64_t)b * S * nq * 64 + hd * 64, (int64_t)nq * 64, pmap, i0, S, tid, rdo[g]);
            } else {
                load_tile32(q + (int64_t)b * S * ldq + hd * 64, ldq, i0, S, tid, rq[g]);
                load_tile32(d_o + (int64_t)b * S * nq * 64 + hd * 64, (int64_t)nq * 64, i0, S, tid, rdo[g]);
            }
        }
        // per-query scalars: waves 0 (and 1 for G = 2) take (g = wave, row = lane & 31); both lane halves
        // compute the same values so that the wave-wide minima need no masking
        if (w < G) {
            const int g = w, row = lane & 31;
            const int sl = i0 + row;
            const bool in = sl < S;
            const int i = in ? (ORD ? pmap[sl] : sl) : 0;          // position of the row
            const int hd = kvh * G + g;
            rl = in ? -lse[((int64_t)b * nq + hd) * S + i] * 1.4426950408889634f : 0.f;
            rd = in ? -delta[((int64_t)b * nq + hd) * S + i] : 0.f;
            if (DROP) raw = rng.row_word(drop_head0 + (uint32_t)(g * S + i));
            if (g == 0) {
                rql = in ? (ql ? ql[(int64_t)b * S + i] : 1) : 0;
                rem = in ? row_empty[(int64_t)b * S + i] : 0;
                const QuerySpan qs = QuerySpan::load<SPAN>(ro.span, (int64_t)b * S + i, i, in, S);
                rpos = qs.hi;
                if (SPAN) { rhlo = qs.hole_lo; rhhi = qs.hole_hi; }
            }
        }
    };
    auto store_q_tile = [&](int qt) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            store_tile32<KLD>(sm.Qs[g], tid, rq[g], S - qt * 32);
            store_tile32<KLD>(sm.dOs[g], tid, rdo[g], S - qt * 32);
        }
        if (w == 0) {                    // (the tile's statistics: DPP reductions over the values wave 0 loaded for head 0)
            const bool normal = qt * 32 + (lane & 31) < S && rem == 0;
            rqlmin = wave_min_i32_dpp(normal ? rql : INT_BIG_A);
            rposmin = wave_min_i32_dpp(normal ? rpos : INT_BIG_A);
            if (SPAN) {
                rhlomin = wave_min_i32_dpp(normal ? rhlo : INT_BIG_A);
                rhhimax = wave_max_i32_dpp(normal ? rhhi : 0);
            }
        }
        if (w < G && lane < 32) {
            const int g = w, row = lane;
            sm.nlse2_s[g][row] = rl;
            sm.ndelta_s[g][row] = rd;
            if (DROP) sm.aw_s[g][row] = raw;
            if (g == 0) {
                sm.ql_s[row] = rql; sm.empty_s[row] = rem; sm.pos_s[row] = rpos;
                if (SPAN) { sm.hole_lo_s[row] = rhlo; sm.hole_hi_s[row] = rhhi; }
                if (row == 0) {
                    sm.qlmin = rqlmin; sm.posmin = rposmin;
                    if (SPAN) { sm.hole_lo_min = rhlomin; sm.hole_hi_max = rhhimax; }
                }
            }
        }
    };

    // No register prefetch across tiles here: the four accumulator tiles already take 64 VGPRs and the
    // second resident workgroup of the CU covers the load latency.
    int qt = next_tile(0);
    while (qt < n_qt) {
        __syncthreads();                 // previous tile fully consumed (and K/V staging visible)
        load_q_tile(qt);
        store_q_tile(qt);
        __syncthreads();
        const bool tile_has_empty = tile_empty_rows(qt);
        const int qt_next = next_tile(qt + 1);
        const int posmin = __builtin_amdgcn_readfirstlane(sm.posmin);
        const int qlmin = __builtin_amdgcn_readfirstlane(sm.qlmin);
        const bool tile_all_empty = tile_has_empty && posmin == INT_BIG_A;

        const bool before = tile_last_pos(qt) < wave_k_lo;   // every normal query of the tile precedes this wave's keys
        if (!(before && !tile_has_empty)) {
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
            const float* Qh = sm.Qs[hg];
            const float* dOh = sm.dOs[hg];
            // S[query][key] = sum_d Q[query][d] K[key][d] and dP[query][key] = sum_d dO[query][d] V[key][d]:
            // two independent accumulation chains, issued alternately
            if (!before && !tile_all_empty) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 a4 = *reinterpret_cast<const float4*>(&Qh[r * KLD + 8 * kk + 4 * h]);
                    const float4 b4 = *reinterpret_cast<const float4*>(Kw + 8 * kk);
                    const float4 c4 = *reinterpret_cast<const float4*>(&dOh[r * KLD + 8 * kk + 4 * h]);
                    const float4 d4 = *reinterpret_cast<const float4*>(Vw + 8 * kk);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.x, d4.x, dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.y, d4.y, dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.z, d4.z, dp, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.w, d4.w, dp, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 c4 = *reinterpret_cast<const float4*>(&dOh[r * KLD + 8 * kk + 4 * h]);
                    const float4 d4 = *reinterpret_cast<const float4*>(Vw + 8 * kk);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.x, d4.x, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.y, d4.y, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.z, d4.z, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.w, d4.w, dp, 0, 0, 0);
                }
            }
            // per element (query = register, key = lane):  P -> st (for dV, 1/(1-p) applied at the end),
            //                                               dS -> dp (for dK)
            // Rows past the end of the sequence have Q = dO = 0, lse = delta = 0: p = 1 but dS = 0 and dO = 0.
            auto elem_tile = [&]<bool MASK, bool EMPTYSEL, bool ALL_EMPTY>() {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int qb = 8 * g4 + 4 * h;
                    const float4 d4 = *reinterpret_cast<const float4*>(&sm.ndelta_s[hg][qb]);
                    const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
                    float nl[4] = {0.f, 0.f, 0.f, 0.f};
                    int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0}, emv[4] = {0, 0, 0, 0};
                    int hlo[4] = {0, 0, 0, 0}, hhi[4] = {0, 0, 0, 0};
                    uint32_t awv[4] = {0, 0, 0, 0};
                    if (!ALL_EMPTY) {
                        const float4 l4 = *reinterpret_cast<const float4*>(&sm.nlse2_s[hg][qb]);
                        nl[0] = l4.x; nl[1] = l4.y; nl[2] = l4.z; nl[3] = l4.w;
                    }
                    if (MASK) {
                        const int4 q4 = *reinterpret_cast<const int4*>(&sm.ql_s[qb]);
                        const int4 p4 = *reinterpret_cast<const int4*>(&sm.pos_s[qb]);
                        qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
                        posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
                        if (SPAN) {
                            const int4 a4 = *reinterpret_cast<const int4*>(&sm.hole_lo_s[qb]);
                            const int4 b4 = *reinterpret_cast<const int4*>(&sm.hole_hi_s[qb]);
                            hlo[0] = a4.x; hlo[1] = a4.y; hlo[2] = a4.z; hlo[3] = a4.w;
                            hhi[0] = b4.x; hhi[1] = b4.y; hhi[2] = b4.z; hhi[3] = b4.w;
                        }
                    }
                    if (EMPTYSEL) {
                        const int4 e4 = *reinterpret_cast<const int4*>(&sm.empty_s[qb]);
                        emv[0] = e4.x; emv[1] = e4.y; emv[2] = e4.z; emv[3] = e4.w;
                    }
                    if (DROP) {
                        const uint4 a4 = *reinterpret_cast<const uint4*>(&sm.aw_s[hg][qb]);
                        awv[0] = a4.x; awv[1] = a4.y; awv[2] = a4.z; awv[3] = a4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * g4 + e;
                        float pe;
                        if (ALL_EMPTY) {
                            pe = invS;
                        } else {
                            pe = __builtin_amdgcn_exp2f(fmaf(st[reg], c2, nl[e]));
                            if (MASK) {
                                bool allowed = (jk <= posv[e]) & (my_kl < qlv[e]);
                                if (SPAN) allowed = allowed & !((jk >= hlo[e]) & (jk < hhi[e]));
                                pe = allowed ? pe : 0.f;
                            }
                            if (EMPTYSEL) pe = (emv[e] != 0) ? invS : pe;
                        }
                        if (DROP) {
                            const bool keep = rng.keep(awv[e], bw);
                            const float t = keep ? dp[reg] : 0.f;
                            dp[reg] = pe * fmaf(t, sd, ndl[e]);          // dS[query][key]
                            st[reg] = keep ? pe : 0.f;                   // dropped P[query][key] * (1 - p)
                        } else {
                            dp[reg] = pe * (dp[reg] + ndl[e]);
                            st[reg] = pe;
                        }
                    }
                }
            };
            if (tile_all_empty) {
                elem_tile.template operator()<false, false, true>();
            } else if (tile_has_empty) {
                elem_tile.template operator()<true, true, false>();
            } else {
                bool free_tile = (posmin >= wave_k_hi) && (wave_kl_max < qlmin);
                if (SPAN) {
                    const int hl = __builtin_amdgcn_readfirstlane(sm.hole_lo_min);
                    const int hh = __builtin_amdgcn_readfirstlane(sm.hole_hi_max);
                    free_tile = free_tile && (wave_k_hi < hl || wave_k_lo >= hh);
                }
                if (free_tile) elem_tile.template operator()<false, false, false>();
                else elem_tile.template operator()<true, false, false>();
            }
            if (ds_out != nullptr && (k0 >> 5) + sub < n_qt) {
                // spill dS for the dQ kernel: tile (head, query tile qt, this wave's 32-key tile) as [query][key]
                // floats; a store instruction writes two 128-byte rows (a layout in the reader's register order
                // makes its loads contiguous but costs this kernel +14 % for 16-byte scattered stores: measured)
                const int head = kvh * G + hg;
                float* tile = ds_out + ((((int64_t)b * nq + head) * n_qt + qt) * n_qt + (k0 >> 5) + sub) * 1024;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) tile[rowmap(reg, h) * 32 + r] = dp[reg];
            }
            // dV^T[d][key] += sum_query dO[query][d] Pd[query][key] ; dK^T[d][key] += sum_query Q[query][d] dS[query][key]
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int qrow = rowmap(reg, h);
                const float o0 = dOh[qrow * KLD + r];
                const float o1 = dOh[qrow * KLD + 32 + r];
                dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o0, st[reg], dvacc[0], 0, 0, 0);
                dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o1, st[reg], dvacc[1], 0, 0, 0);
                const float q0v = Qh[qrow * KLD + r];
                const float q1v = Qh[qrow * KLD + 32 + r];
                dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0v, dp[reg], dkacc[0], 0, 0, 0);
                dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1v, dp[reg], dkacc[1], 0, 0, 0);
            }
        }
        qt = qt_next;
    }

    // ---- sum the G query heads of this kv head through LDS, then store ---------------------------
    // LDS image [sub][key 32][128] floats (dK 64 | dV 64) reuses the tile region.
    float* red = reinterpret_cast<float*>(dkv_raw);
#pragma unroll
    for (int gsrc = G - 1; gsrc >= 1; --gsrc) {
        __syncthreads();
        if (hg == gsrc) {
            float* dst = red + (sub * 32 + r) * 132;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * dh + 8 * g4 + 4 * h;
                    *reinterpret_cast<float4*>(dst + d) = make_float4(dkacc[dh][4 * g4], dkacc[dh][4 * g4 + 1],
                                                                      dkacc[dh][4 * g4 + 2], dkacc[dh][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(dst + 64 + d) = make_float4(dvacc[dh][4 * g4], dvacc[dh][4 * g4 + 1],
                                                                           dvacc[dh][4 * g4 + 2], dvacc[dh][4 * g4 + 3]);
                }
        }
        __syncthreads();
        if (hg == 0) {
            const float* src = red + (sub * 32 + r) * 132;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * dh + 8 * g4 + 4 * h;
                    const float4 a = *reinterpret_cast<const float4*>(src + d);
                    const float4 c = *reinterpret_cast<const float4*>(src + 64 + d);
                    dkacc[dh][4 * g4] += a.x; dkacc[dh][4 * g4 + 1] += a.y; dkacc[dh][4 * g4 + 2] += a.z; dkacc[dh][4 * g4 + 3] += a.w;
                    dvacc[dh][4 * g4] += c.x; dvacc[dh][4 * g4 + 1] += c.y; dvacc[dh][4 * g4 + 2] += c.z; dvacc[dh][4 * g4 + 3] += c.w;
                }
        }
    }
    if (hg == 0 && valid_k) {
        float* dkrow = dk + ktok * lddk + kvh * 64;
        float* dvrow = dv + ktok * lddv + kvh * 64;
        const float vs = DROP ? sd : 1.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dh + 8 * g4 + 4 * h;
                *reinterpret_cast<float4*>(dkrow + d) = make_float4(dkacc[dh][4 * g4] * scale, dkacc[dh][4 * g4 + 1] * scale,
                                                                    dkacc[dh][4 * g4 + 2] * scale, dkacc[dh][4 * g4 + 3] * scale);
                *reinterpret_cast<float4*>(dvrow + d) = make_float4(dvacc[dh][4 * g4] * vs, dvacc[dh][4 * g4 + 1] * vs,
                                                                    dvacc[dh][4 * g4 + 2] * vs, dvacc[dh][4 * g4 + 3] * vs);
            }
    }
}

template <int G, bool DROP, bool ORD, bool SPAN>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_bwd_dkv_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                    const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                    const float* __restrict__ lse, const float* __restrict__ delta,
                    const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                    const int32_t* __restrict__ row_empty, const int32_t* __restrict__ tile_empty,
                    int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                    float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, const RowOrder ro,
                    float* __restrict__ ds_out) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int heavy = wl.u, light = n_tiles - 1 - wl.u;       // earlier key tiles are seen by more queries
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dkv_tile<G, DROP, ORD, SPAN>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, tile_empty, S, nq, nkv,
                                            scale, p_drop, seed, dk, lddk, dv, lddv, ro, pair, pass == 0 ? heavy : light, ds_out);
        }
    }
}

// =============================================================================================
// backward with a dS spill: delta, then dK/dV (+ dS to HBM), then dQ = dS K from the spilled tiles
// =============================================================================================
// The recompute pair above issues 7 matmuls per (query tile, key tile): S and dP twice.  With a workspace of
// B*nq*ceil(S/32)^2 KiB*4 the dK/dV kernel writes its dS tiles (3 GB per call at B = 1024, hidden behind its
// MFMA work) and dQ becomes one matmul per tile that streams them back (HBM-bound, ~0.6 ms): 5 matmuls, no
// second softmax / dropout pass, still no atomics.

// delta[b,h,i] = sum_d dO * O
__global__ void __launch_bounds__(AT_THREADS)
attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, int B, int S, int nq,
                  float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;                  // 16 lanes x float4 = one head row, 4 rows per wave
    const int64_t wave = ((int64_t)blockIdx.x * AT_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * AT_THREADS) >> 6;
    const int64_t total = (int64_t)B * S * nq;
    for (int64_t i0 = wave * 4; i0 < total; i0 += nwaves * 4) {
        const int64_t i = i0 + sub;
        const bool live = i < total;
        const int64_t ic = live ? i : total - 1;
        const float4 a = reinterpret_cast<const float4*>(o + ic * 64)[g];
        const float4 c = reinterpret_cast<const float4*>(d_o + ic * 64)[g];
        float s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (live && g == 0) {
            const int64_t t = i / nq;
            const int hd = (int)(i % nq);
            const int64_t bb = t / S, ii = t % S;
            delta[(bb * nq + hd) * S + ii] = s;
        }
    }
}

template <int G, bool ORD>
__device__ __forceinline__ void
attn_bwd_dq2_tile(const float* __restrict__ k, int ldk, const float* __restrict__ ds_in,
                  const int32_t* __restrict__ row_empty, int S, int nq, int nkv, float scale,
                  float* __restrict__ dq, int lddq, const RowOrder ro, const int pair, const int qtile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * KLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const int slot = q0 + sub * 32 + r;
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iqc = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int64_t tok = (int64_t)b * S + iqc;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const int n_all = (S + 31) / 32;
    const int qt32 = (q0 >> 5) + sub;                            // this wave's 32-row query tile
    int wave_q_hi, n_causal;
    bool wave_has_empty;
    if (ORD) {
        const int64_t tb = (int64_t)b * n_all + (q0 >> 5);
        const int wt = min(sub, n_all - 1 - (q0 >> 5));
        wave_q_hi = (q0 + sub * 32 < S) ? ro.tile_maxpos[tb + wt] : -1;
        wave_has_empty = (q0 + sub * 32 < S) && (ro.tile_kind[tb + wt] & 2) != 0;
        int hi = -1;
#pragma unroll
        for (int ss = 0; ss < NSUB; ++ss)
            if (q0 + ss * 32 < S) hi = max(hi, ro.tile_maxpos[tb + ss]);
        n_causal = hi < 0 ? 0 : (hi >> 5) + 1;
    } else {
        wave_q_hi = (q0 + sub * 32 < S) ? min(S - 1, q0 + sub * 32 + 31) : -1;
        wave_has_empty = __any(my_empty ? 1 : 0) != 0;
        n_causal = (min(S, q0 + R) + 31) / 32;
    }
    const bool block_has_empty = __syncthreads_or(my_empty ? 1 : 0) != 0;
    const int n_iter = block_has_empty ? n_all : n_causal;
    const float* kbase = k + (int64_t)b * S * ldk + kvh * 64;
    // dS tiles of this wave: (head, qt32, key tile jt) -> 1024 floats [query][key]; the lane takes query r and the
    // keys rowmap(reg, h), i.e. four 16-byte pieces of its row
    const float* ds_row = ds_in + ((((int64_t)b * nq + head) * n_all + min(qt32, n_all - 1)) * n_all) * 1024 + r * 32 + 4 * h;

    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }
    auto live = [&](int jt) { return !((jt * 32 > wave_q_hi) && !wave_has_empty); };   // mirrors the dK/dV kernel
    auto load_ds = [&](int jt, float4 (&t4)[4]) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) t4[g4] = *reinterpret_cast<const float4*>(ds_row + (int64_t)jt * 1024 + 8 * g4);
    };
    // the dS tiles come straight from HBM and one tile of MFMA work (~1 us) is shorter than the load latency:
    // three tiles are kept in flight per wave
    float4 rk[2], t0[4], t1[4], t2[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) { t0[g4] = make_float4(0.f, 0.f, 0.f, 0.f); t1[g4] = t0[g4]; t2[g4] = t0[g4]; }
    load_tile32(kbase, ldk, 0, S, tid, rk);
    store_tile32<KLD>(Ks[0], tid, rk, S);
    if (n_iter > 0 && live(0)) load_ds(0, t0);
    if (n_iter > 1 && live(1)) load_ds(1, t1);
    if (n_iter > 2 && live(2)) load_ds(2, t2);
    __syncthreads();
    // iteration jt consumes ring slot jt % 3 and refills it with tile jt + 3 (no register rotation: a copy of a
    // register with a load in flight would wait for the load)
    auto step = [&](int jt, float4 (&tu)[4]) {
        const int cur = jt & 1;
        const bool more = jt + 1 < n_iter;
        if (more) load_tile32(kbase, ldk, jt * 32 + 32, S, tid, rk);
        if (live(jt)) {
            const float st[16] = {tu[0].x, tu[0].y, tu[0].z, tu[0].w, tu[1].x, tu[1].y, tu[1].z, tu[1].w,
                                  tu[2].x, tu[2].y, tu[2].z, tu[2].w, tu[3].x, tu[3].y, tu[3].z, tu[3].w};
            // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]   (rows of K past the sequence end are zero)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int key = rowmap(reg, h);
                const float a0 = Ks[cur][key * KLD + r];
                const float a1 = Ks[cur][key * KLD + 32 + r];
                dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, st[reg], dqacc[0], 0, 0, 0);
                dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, st[reg], dqacc[1], 0, 0, 0);
            }
        }
        if (jt + 3 < n_iter && live(jt + 3)) load_ds(jt + 3, tu);
        if (more) store_tile32<KLD>(Ks[cur ^ 1], tid, rk, S - (jt * 32 + 32));
        __syncthreads();
    };
    for (int jt = 0; jt < n_iter; jt += 3) {
        step(jt, t0);
        if (jt + 1 < n_iter) step(jt + 1, t1);
        if (jt + 2 < n_iter) step(jt + 2, t2);
    }
    if (valid_q) {
        float* drow = dq + tok * lddq + head * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = dqacc[dh][4 * g4 + 0] * scale; t4.y = dqacc[dh][4 * g4 + 1] * scale;
                t4.z = dqacc[dh][4 * g4 + 2] * scale; t4.w = dqacc[dh][4 * g4 + 3] * scale;
                *reinterpret_cast<float4*>(drow + 32 * dh + 8 * g4 + 4 * h) = t4;
            }
        }
    }
}

template <int G, bool ORD>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_bwd_dq2_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ ds_in,
                    const int32_t* __restrict__ row_empty, int nbatch, int S, int nq, int nkv, float scale,
                    float* __restrict__ dq, int lddq, const RowOrder ro) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int heavy = n_tiles - 1 - wl.u, light = wl.u;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dq2_tile<G, ORD>(k, ldk, ds_in, row_empty, S, nq, nkv, scale, dq, lddq, ro, pair,
                                      pass == 0 ? heavy : light);
        }
    }
}

// Streaming form of the dQ kernel for sequences whose K fits in LDS (S <= 573 at head_dim 64): one workgroup per
// (sample, kv head) keeps ALL its keys in LDS (137 KB at S = 505), so the loop has no staging, no barrier and no
// branch - four dS loads for tile t+5, 32 LDS fragment reads and 32 MFMAs per step - and each wave keeps five dS
// tiles (20 KB) in flight: the kernel runs at the rate the spilled dS comes back from HBM.
constexpr int DQ3_THREADS = 512;          // two waves per SIMD: twice the dS tiles in flight per CU
template <int G, bool ORD>
__global__ void __launch_bounds__(DQ3_THREADS, 1)
attn_bwd_dq3_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ ds_in,
                    const int32_t* __restrict__ row_empty, int S, int nq, int nkv, float scale,
                    float* __restrict__ dq, int lddq, const RowOrder ro) {
    extern __shared__ __attribute__((aligned(16))) float Kall[];          // [n_all*32][KLD]
    constexpr int NS = (DQ3_THREADS / 64) / G;                             // query-tile streams per head
    constexpr int RING = 5;                                                // dS tiles in flight per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / nkv, kvh = blockIdx.x % nkv;
    const int r = lane & 31, h = lane >> 5;
    const int n_all = (S + 31) / 32;
    // fill: eight loads in flight per thread (the matrix pipe idles until K is in place)
    for (int f0 = tid; f0 < n_all * 32 * 16; f0 += DQ3_THREADS * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + DQ3_THREADS * u;
            const int row = min(f >> 4, S - 1), c4 = (f & 15) << 2;
            v[u] = *reinterpret_cast<const float4*>(k + ((int64_t)b * S + row) * ldk + kvh * 64 + c4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + DQ3_THREADS * u;
            const int row = f >> 4, c4 = (f & 15) << 2;
            if (row < n_all * 32) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Kall[row * KLD + c4]) = row < S ? v[u] : z;
            }
        }
    }
    __syncthreads();
    const int head = kvh * G + (w % G), stream = w / G;
    const float* ds_head = ds_in + ((int64_t)b * nq + head) * n_all * n_all * 1024 + r * 32 + 4 * h;
    for (int m = 0; m * NS < n_all; ++m) {
        // zigzag over the query tiles so that the streams of a head see the same number of (query, key) tile pairs
        const int qt = m * NS + ((m & 1) ? NS - 1 - stream : stream);
        if (qt >= n_all) continue;
        const int slot = qt * 32 + r;
        const bool valid_q = slot < S;
        const int slotc = valid_q ? slot : S - 1;
        const int iqc = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
        const int64_t tok = (int64_t)b * S + iqc;
        const bool my_empty = valid_q && row_empty[tok] != 0;
        int q_hi;
        bool has_empty;
        if (ORD) {
            q_hi = ro.tile_maxpos[(int64_t)b * n_all + qt];
            has_empty = (ro.tile_kind[(int64_t)b * n_all + qt] & 2) != 0;
        } else {
            q_hi = min(S - 1, qt * 32 + 31);
            has_empty = __any(my_empty ? 1 : 0) != 0;
        }
        // the key tiles the dK/dV kernel visited for this query tile (and spilled dS for)
        const int n_iter = has_empty ? n_all : (q_hi < 0 ? 0 : (q_hi >> 5) + 1);
        const float* ds_q = ds_head + (int64_t)qt * n_all * 1024;
        f32x16 dqacc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }
        auto load_ds = [&](int jt, float4 (&t4)[4]) {                        // tile index clamped: loads are unconditional
            const float* p = ds_q + (int64_t)min(jt, n_iter - 1) * 1024;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) t4[g4] = *reinterpret_cast<const float4*>(p + 8 * g4);
        };
        auto step = [&](int jt, float4 (&tu)[4]) {
            const float st[16] = {tu[0].x, tu[0].y, tu[0].z, tu[0].w, tu[1].x, tu[1].y, tu[1].z, tu[1].w,
                                  tu[2].x, tu[2].y, tu[2].z, tu[2].w, tu[3].x, tu[3].y, tu[3].z, tu[3].w};
            const float* Kt = Kall + jt * 32 * KLD;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int key = rowmap(reg, h);
                dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[key * KLD + r], st[reg], dqacc[0], 0, 0, 0);
                dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[key * KLD + 32 + r], st[reg], dqacc[1], 0, 0, 0);
            }
            // pin the refill here: left alone, the scheduler sinks these loads next to their use three steps later
            __builtin_amdgcn_sched_barrier(0);
            load_ds(jt + RING, tu);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (n_iter > 0) {
            float4 ring[RING][4];
#pragma unroll
            for (int u = 0; u < RING; ++u) load_ds(u, ring[u]);
            int jt = 0;
            for (; jt + RING <= n_iter; jt += RING) {
#pragma unroll
                for (int u = 0; u < RING; ++u) step(jt + u, ring[u]);
            }
#pragma unroll
            for (int u = 0; u < RING - 1; ++u)
                if (jt + u < n_iter) step(jt + u, ring[u]);
        }
        if (valid_q) {
            float* drow = dq + tok * lddq + head * 64;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float4 t4;
                    t4.x = dqacc[dh][4 * g4 + 0] * scale; t4.y = dqacc[dh][4 * g4 + 1] * scale;
                    t4.z = dqacc[dh][4 * g4 + 2] * scale; t4.w = dqacc[dh][4 * g4 + 3] * scale;
                    *reinterpret_cast<float4*>(drow + 32 * dh + 8 * g4 + 4 * h) = t4;
                }
            }
        }
    }
}

// Stable partition of the query rows of every sequence: normal rows first (ascending position), empty rows
// behind them; plus the per-32-slot tile summaries the attention kernels schedule by.
__global__ void __launch_bounds__(256)
attn_row_order_kernel(const int32_t* __restrict__ row_empty, int S, int32_t* __restrict__ perm,
                      int32_t* __restrict__ tile_kind, int32_t* __restrict__ tile_maxpos) {
    extern __shared__ __attribute__((aligned(16))) int32_t osm[];
    int32_t* a = osm;            // scan ping
    int32_t* c = osm + S;        // scan pong
    int32_t* slot_pos = osm + 2 * S;
    const int b = blockIdx.x;
    const int32_t* re = row_empty + (int64_t)b * S;
    for (int t = threadIdx.x; t < S; t += blockDim.x) a[t] = re[t] ? 0 : 1;
    __syncthreads();
    int32_t* src = a;
    int32_t* dst = c;
    for (int off = 1; off < S; off <<= 1) {
        for (int t = threadIdx.x; t < S; t += blockDim.x) dst[t] = src[t] + (t >= off ? src[t - off] : 0);
        __syncthreads();
        int32_t* tmp = src; src = dst; dst = tmp;
    }
    const int n_normal = src[S - 1];
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const int incl = src[t];
        const int sl = re[t] ? n_normal + (t - incl) : incl - 1;
        slot_pos[sl] = t;
        perm[(int64_t)b * S + sl] = t;
    }
    __syncthreads();
    const int n_tiles = (S + 31) / 32;
    for (int qt = threadIdx.x; qt < n_tiles; qt += blockDim.x) {
        int kind = 0, mx = -1;
        for (int sl = qt * 32; sl < min(S, qt * 32 + 32); ++sl) {
            const int pos = slot_pos[sl];
            if (re[pos]) kind |= 2; else { kind |= 1; mx = max(mx, pos); }
        }
        tile_kind[(int64_t)b * n_tiles + qt] = kind;
        tile_maxpos[(int64_t)b * n_tiles + qt] = mx;
    }
}

template <int G>
static int launch_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* kl,
                      const int32_t* ql, const int32_t* row_empty, int B, int S, int nq, int nkv, float scale,
                      float p_drop, uint64_t seed, float* o, float* lse, RowOrder ro, int uspan, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    dim3 grid(worklist_grid(B * nkv, (S + R - 1) / R));
#define GAMER_LAUNCH_FWD(DROPV, ORDV, SPANV)                                                                              \
    hipLaunchKernelGGL((attn_fwd_kernel<G, DROPV, ORDV, SPANV>), grid, dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, \
                       kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan)
#define GAMER_LAUNCH_FWD2(DROPV, ORDV) do { if (ro.span) GAMER_LAUNCH_FWD(DROPV, ORDV, true); else GAMER_LAUNCH_FWD(DROPV, ORDV, false); } while (0)
    if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD2(true, true); else GAMER_LAUNCH_FWD2(true, false); }
    else { if (ro.perm) GAMER_LAUNCH_FWD2(false, true); else GAMER_LAUNCH_FWD2(false, false); }
#undef GAMER_LAUNCH_FWD2
#undef GAMER_LAUNCH_FWD
    GAMER_CHECK_LAUNCH("gamer_attn_fwd");
    return 0;
}

template <int G, bool DROP, bool ORD, bool SPAN>
static int launch_bwd_variant(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                              const float* d_o, const float* lse, float* delta, const int32_t* kl, const int32_t* ql,
                              const int32_t* row_empty, const int32_t* tile_empty, int B, int S, int nq, int nkv,
                              float scale, float p_drop, uint64_t seed, float* dq, int lddq, float* dk, int lddk,
                              float* dv, int lddv, RowOrder ro, float* ds_work, int delta_ready, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    dim3 grid(worklist_grid(B * nkv, (S + R - 1) / R));
    if (ds_work == nullptr) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<G, DROP, ORD, SPAN>), grid, dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, o, d_o,
                           lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, ro);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd/dq");
    } else if (!delta_ready) {
        hipLaunchKernelGGL(attn_delta_kernel, dim3(2048), dim3(AT_THREADS), 0, st, o, d_o, B, S, nq, delta);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd/delta");
    }
    size_t shmem = sizeof(DkvSmem<G>);
    const size_t red_bytes = (size_t)R * 132 * sizeof(float);
    if (shmem < red_bytes) shmem = red_bytes;
    static bool attr_dev[MAX_DEVICES] = {};   // one set of flags per template instantiation, one flag per device
    bool& attr_set = attr_dev[current_device()];
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<G, DROP, ORD, SPAN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<G, DROP, ORD, SPAN>), grid, dim3(AT_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, d_o, lse,
                       delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv, lddv, ro,
                       ds_work);
    GAMER_CHECK_LAUNCH("gamer_attn_bwd/dkv");
    if (ds_work != nullptr) {
        const size_t kall = (size_t)((S + 31) / 32) * 32 * KLD * sizeof(float);
        if (kall <= 156 * 1024) {
            static bool attr3_dev[MAX_DEVICES] = {};
            bool& attr3 = attr3_dev[current_device()];
            if (!attr3) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq3_kernel<G, ORD>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                if (e != hipSuccess) {
                    set_error("gamer_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                    return (int)e;
                }
                attr3 = true;
            }
            hipLaunchKernelGGL((attn_bwd_dq3_kernel<G, ORD>), dim3(B * nkv), dim3(DQ3_THREADS), kall, st, k, ldk, ds_work,
                               row_empty, S, nq, nkv, scale, dq, lddq, ro);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<G, ORD>), grid, dim3(AT_THREADS), 0, st, k, ldk, ds_work, row_empty, B, S,
                               nq, nkv, scale, dq, lddq, ro);
        }
        GAMER_CHECK_LAUNCH("gamer_attn_bwd/dq2");
    }
    return 0;
}

template <int G>
static int launch_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                      const float* d_o, const float* lse, float* delta, const int32_t* kl, const int32_t* ql,
                      const int32_t* row_empty, const int32_t* tile_empty, int B, int S, int nq, int nkv, float scale,
                      float p_drop, uint64_t seed, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                      RowOrder ro, float* ds_work, int delta_ready, hipStream_t st) {
#define GAMER_LAUNCH_BWD(DROPV, ORDV, SPANV)                                                                                \
    return launch_bwd_variant<G, DROPV, ORDV, SPANV>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, \
                                              S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, ds_work, delta_ready, st)
#define GAMER_LAUNCH_BWD2(DROPV, ORDV) do { if (ro.span) GAMER_LAUNCH_BWD(DROPV, ORDV, true); else GAMER_LAUNCH_BWD(DROPV, ORDV, false); } while (0)
    if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_BWD2(true, true); else GAMER_LAUNCH_BWD2(true, false); }
    else { if (ro.perm) GAMER_LAUNCH_BWD2(false, true); else GAMER_LAUNCH_BWD2(false, false); }
#undef GAMER_LAUNCH_BWD2
#undef GAMER_LAUNCH_BWD
    return -1;      // not reached
}

}  // namespace gamer

using namespace gamer;

static int check_attn_common(const char* name, const void* q, const void* k, const void* v, const void* kl,
                             const void* row_empty, int ldq, int ldk, int ldv, int B, int S, int nq, int nkv,
                             float p_drop) {
    GAMER_CHECK_ARG(q && k && v && kl && row_empty, "%s: null pointer", name);
    GAMER_CHECK_ARG(B > 0 && S > 0 && nq > 0 && nkv > 0 && nq % nkv == 0, "%s: bad shape B=%d S=%d nq=%d nkv=%d", name, B, S, nq, nkv);
    const int G = nq / nkv;
    GAMER_CHECK_ARG(G == 1 || G == 2, "%s: GQA group %d not built (1 or 2)", name, G);
    GAMER_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldq >= nq * 64 && ldk >= nkv * 64 && ldv >= nkv * 64,
                    "%s: bad leading dims ldq=%d ldk=%d ldv=%d", name, ldq, ldk, ldv);
    GAMER_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v), "%s: q/k/v must be 16-byte aligned", name);
    GAMER_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p_drop=%f", name, p_drop);
    return 0;
}

extern "C" int gamer_attn_row_order(const int32_t* row_empty, int B, int S, int32_t* perm, int32_t* tile_kind,
                                    int32_t* tile_maxpos, void* stream) {
    GAMER_CHECK_ARG(row_empty && perm && tile_kind && tile_maxpos && B > 0 && S > 0 && S <= 8192,
                    "gamer_attn_row_order: bad arguments B=%d S=%d", B, S);
    hipLaunchKernelGGL(attn_row_order_kernel, dim3(B), dim3(256), (size_t)3 * S * sizeof(int32_t), (hipStream_t)stream,
                       row_empty, S, perm, tile_kind, tile_maxpos);
    GAMER_CHECK_LAUNCH("gamer_attn_row_order");
    return 0;
}

extern "C" int gamer_debug_set_trace(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &v, sizeof(v));
}

// Debug aid (not part of the public header): resident workgroups per CU the runtime reports.
extern "C" int gamer_debug_attn_occupancy(int which) {
    int n = -1;
    hipError_t e = hipSuccess;
    switch (which) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<2, true, false, false>, AT_THREADS, 0); break;
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<2, false, false, false>, AT_THREADS, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dq_kernel<2, true, false, false>, AT_THREADS, 0); break;
        case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dkv_kernel<2, true, false, false>, AT_THREADS, sizeof(DkvSmem<2>)); break;
    }
    return e == hipSuccess ? n : -(int)e;
}

extern "C" int gamer_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                              const int32_t* kl, const int32_t* ql, const int32_t* row_empty,
                              const int32_t* tile_empty, int B, int S, int nq, int nkv, float scale, float p_drop,
                              uint64_t seed, float* o, float* lse, const int32_t* row_perm, const int32_t* tile_kind,
                              const int32_t* tile_maxpos, int uniform_len, const int32_t* q_span, void* stream) {
    (void)tile_empty;
    GAMER_CHECK_ARG(uniform_len >= 0 && uniform_len <= S, "gamer_attn_fwd: uniform_len=%d outside 0..S", uniform_len);
    const int uspan = uniform_len == 0 ? S : uniform_len;
    GAMER_CHECK_ARG(!row_perm || (tile_kind && tile_maxpos), "gamer_attn_fwd: row_perm needs tile_kind and tile_maxpos");
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_fwd: q_span must be 16-byte aligned");
    const RowOrder ro{row_perm, tile_kind, tile_maxpos, q_span};
    int rc = check_attn_common("gamer_attn_fwd", q, k, v, kl, row_empty, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && lse && aligned16(o), "gamer_attn_fwd: null/unaligned output");
    hipStream_t st = (hipStream_t)stream;
    switch (nq / nkv) {
        case 1: return launch_fwd<1>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, st);
        case 2: return launch_fwd<2>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, st);
        default: return launch_fwd<2>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, st);
    }
}

extern "C" int gamer_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                              const float* o, const float* d_o, const float* lse, const int32_t* kl,
                              const int32_t* ql, const int32_t* row_empty, const int32_t* tile_empty, int B, int S,
                              int nq, int nkv, float scale, float p_drop, uint64_t seed, float* delta, float* dq,
                              int lddq, float* dk, int lddk, float* dv, int lddv, const int32_t* row_perm,
                              const int32_t* tile_kind, const int32_t* tile_maxpos, float* ds_work,
                              const int32_t* q_span, int delta_ready, void* stream) {
    int rc = check_attn_common("gamer_attn_bwd", q, k, v, kl, row_empty, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && d_o && lse && tile_empty && delta && dq && dk && dv, "gamer_attn_bwd: null pointer");
    GAMER_CHECK_ARG(!row_perm || (tile_kind && tile_maxpos), "gamer_attn_bwd: row_perm needs tile_kind and tile_maxpos");
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_bwd: q_span must be 16-byte aligned");
    const RowOrder ro{row_perm, tile_kind, tile_maxpos, q_span};
    GAMER_CHECK_ARG(lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && aligned16(dq) && aligned16(dk) && aligned16(dv) &&
                    aligned16(d_o) && aligned16(o),
                    "gamer_attn_bwd: gradient buffers must be 16-byte aligned with leading dims %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    switch (nq / nkv) {
        case 1: return launch_bwd<1>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, ds_work, delta_ready, st);
        case 2: return launch_bwd<2>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, ds_work, delta_ready, st);
        default: return launch_bwd<2>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, ds_work, delta_ready, st);
    }
}
