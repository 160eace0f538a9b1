// Multi-behaviour flash attention for gfx950 with fp32 inputs / outputs and every matrix product formed on the bf16 matrix
// pipe from an EXACT three-way cut of both operands (the attention counterpart of gamer_gemm_f32_split, csrc/gemm.hip):
// an fp32 value x is the sum of three bf16 values x0 = upper 16 bits of x, x1 = upper 16 bits of (x - x0), x2 = x - x0 - x1
// (both subtractions are exact), every piece product a_i b_j (8 x 8 significant bits) is exact in fp32, and the six
// products with i + j <= 2 are accumulated by v_mfma_f32_32x32x16_bf16 in fp32 - what is dropped is below 2^-24 |a b|,
// the size of one fp32 rounding.  The bf16 pipe is 16 x wider than the fp32 one, so six products cost 3/8 of the fp32
// MFMA's time, and on gfx950 fp32 MFMA and fp32 VALU work share a pipe (tools/ubench_pipes.hip).
//
// Same algorithm, semantics, work partition and entry-point signatures as attention.hip (Qwen3Multi masks as predicates,
// "empty" rows uniform over all S keys, lazily rescaled online softmax, row order of the cross attention, dropout mask
// function): replaces ref:SeqRec/models/generative/Qwen3Multi/model.py:133-143 + HF sdpa_attention_forward in the
// Engine(matmul="split6") step.  What differs from attention.hip:
//   * K / V (Q / dO in the dK/dV kernel) tiles are cut on their way into LDS: three bf16 images [rows][64] per tile,
//     16-byte chunks XOR-swizzled (the layout of attention_bf16.hip: conflict-free row reads and transposing reads);
//   * q (pre-scaled), dO rows are cut once per tile into register fragments; P and dS are cut in registers;
//   * the backward comes in the two forms of attention.hip: recompute (dQ kernel + dK/dV kernel, 7 products per tile pair)
//     and, with a workspace, the dS spill (the dK/dV kernel writes its fp32 dS tiles, dQ = dS K is one product per tile
//     pair in the tiled kernel - the fp32 design's streaming dQ kernel would need the whole K of a sequence as three
//     images in LDS, 194 KB).
// Not built here (the Engine keeps those calls on the fp32-MFMA kernels): per-query key spans (Qwen3SessionMulti) and
// uniform_len of the evaluation re-run path.
#include "attention_common.h"

#include "attention_split_common.h"

namespace gamer {

#ifndef SPA_STAMP
#define SPA_STAMP 0       // diagnostic builds (tools/stamp_attn_fwd.py): every wave of the forward kernel accumulates shader-clock cycles per
#endif                    // phase of its key loop into g_attn_stamp[(4 * blockIdx.x + wave) * 8 + phase] (set by gamer_debug_attn_stamp)
#if SPA_STAMP
__device__ unsigned long long* g_attn_stamp = nullptr;
#define SPA_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         ph_[i] += (unsigned)(t_ - tprev_); tprev_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SPA_MARK(i) do { } while (0)
#endif

// ACC += (three K images of one slot)(rows = this tile's 32 keys) . q^T, six piece products, smallest first.
// (Measured and not kept: two interleaved accumulation chains here and in the second products - the extra 16 accumulator
// registers cost more in spills than the independent chains gained: forward 0.63 -> 0.76 ms at batch 256.)
#define SPLIT_QK_TILE(ACC, KIMG)                                                                                          \
    if (!(SPA_ABLATE & 1)) {                                                                                                \
        /* the fragments of k-step s + 1 are requested BEFORE the six products of k-step s (left alone the compiler issues  \
           every ds_read right in front of the MFMA that needs it and waits for it: the 24 MFMAs of a tile cost 3.3 x their \
           pipe time, tools/ablate_attn_split.sh) */                                                                        \
        constexpr int NP_ = H2 ? 2 : 3;                                                                                     \
        bf16x8 kf_[2][3];                                                                                                   \
        _Pragma("unroll") for (int pc_ = 0; pc_ < NP_; ++pc_) kf_[0][pc_] = read_row8(KIMG + pc_ * SIMG, lo, 0, 0);         \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                     \
            if (s + 1 < 4) {                                                                                                \
                _Pragma("unroll") for (int pc_ = 0; pc_ < NP_; ++pc_) kf_[(s + 1) & 1][pc_] = read_row8(KIMG + pc_ * SIMG, lo, 0, s + 1); \
            }                                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            const bf16x8 qs_[3] = {qf[0][s], qf[1][s], qf[NP_ - 1][s]};                                                     \
            ACC = mfma_pieces<H2>(kf_[s & 1], qs_, ACC);                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                   \
    }

// =============================================================================================
// forward (attention.hip: attn_fwd_tile, with the split fragments)
// =============================================================================================
template <int G, bool DROP, bool ORD, bool SPAN, bool H2>
__device__ __forceinline__ void
attn_fwd_s_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
              const float* __restrict__ v, int ldv, const int32_t* __restrict__ kl,
              const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
              int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
              float* __restrict__ o, float* __restrict__ lse, const RowOrder ro, const int pair, const int qtile,
              const int uspan, uint32_t* __restrict__ amax_lds, const H2Scales& sc) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    constexpr int NP = H2 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][NP * SIMG];     // slot x (NP piece images of 32 keys x 64)
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][NP * SIMG];
    __shared__ __attribute__((aligned(16))) KeyMeta kms[3];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#if SPA_STAMP
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int slot = q0 + sub * 32 + r;                 // sorted slot of this lane's query row
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iq_raw = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int iq = valid_q ? iq_raw : S;                // position in the sequence (S: beyond every key)
    const int iqc = iq_raw;
    const int64_t tok = (int64_t)b * S + iqc;

    // scores are kept in the log2 domain: q is pre-scaled by scale*log2(e), p = exp2(s - m_ref)
    const float qs = scale * 1.4426950408889634f;
    bf16x8 qf[3][4];                   // piece x k-step: lane (r, h) holds d = 16 s + 8 h .. + 7 of its query row
    {
        const float* qrow = q + tok * ldq + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = *reinterpret_cast<const float4*>(qrow + 16 * s);
            float4 b4 = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
            a4.x *= qs; a4.y *= qs; a4.z *= qs; a4.w *= qs; b4.x *= qs; b4.y *= qs; b4.z *= qs; b4.w *= qs;
            bf16x8 pq[3];
            cut8_t<H2>(a4, b4, sc.q, pq);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) qf[pc][s] = pq[pc];
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq_raw, valid_q, S);
    const int wave_ql_min = wave_min_i32(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32(normal ? sp.hi : INT_BIG_A);   // every key up to here passes every row's limit
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;          // no normal row: no scores needed at all
    const int wave_hole_lo = SPAN ? wave_min_i32(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32(normal ? sp.hole_hi : 0) : 0;
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
    store_tile32_t<H2>(Ks[0], tid, rk, S, sc.k);
    store_tile32_t<H2>(Vs[0], tid, rv, S, sc.v);
    key_meta_store(kms[0], w, lane, rmeta);
    if (n_iter > 1) {
        load_tile32(kbase, ldk, 32, S, tid, rk);
        rmeta = key_meta_load<DROP>(klb, 32, S, w, lane, rng);
        store_tile32_t<H2>(Ks[1], tid, rk, S - 32, sc.k);
        key_meta_store(kms[1], w, lane, rmeta);
    }
    __syncthreads();

    f32x16 st_cur;
#pragma unroll
    for (int i = 0; i < 16; ++i) st_cur[i] = 0.f;
    if (!(0 > wave_q_hi)) {
        SPLIT_QK_TILE(st_cur, Ks[0])
        if (H2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st_cur[i] *= sc.inv_qk;
        }
    }
    // K(0) must stay intact until every wave has taken its first scores: a wave without work in this tile
    // (rows past the end of the sequence, or empty rows only) reaches the K(2) store of iteration 0 at once
    __syncthreads();

    SPA_MARK(0);                                         // 0: the tile's prologue (q, row data, first K / V tiles, first scores)
    for (int jt = 0; jt < n_iter; ++jt) {
        const int j0 = jt * 32;
        const bool have_next = jt + 1 < n_iter;
        const bool have_next2 = jt + 2 < n_iter;
        if (have_next2 && !(SPA_ABLATE & 512)) {
            load_tile32(kbase, ldk, j0 + 64, S, tid, rk);
            rmeta = key_meta_load<DROP>(klb, j0 + 64, S, w, lane, rng);
        }
        if (have_next && !(SPA_ABLATE & 512)) load_tile32(vbase, ldv, j0 + 32, S, tid, rv);

        SPA_MARK(1);                                       // 1: issue of the K / V / metadata loads
        const bool beyond = j0 > wave_q_hi;                // every key of the tile is in every row's future
        const bool beyond_next = (j0 + 32) > wave_q_hi;
        if (!(beyond && !wave_has_empty)) {
            const KeyMeta& km = kms[jt % 3];
            const bf16_t* Vc = Vs[jt & 1];
            const bool full_span = (j0 + 32 <= uspan) || uspan >= S;   // every key of the tile is inside every span
            // O^T[d][query] += sum_key V[key][d] * P[query][key]   (rows of V past the sequence end are zero).
            // Called at the end of every variant below so that each one is a straight-line path (a join in
            // front of the MFMAs costs 16 register copies per tile, and VALU time is not hidden here).
            auto pv_tile = [&]<bool ONE_PIECE>() {
                // P (fp32, in registers) cut into three bf16 pieces; ONE_PIECE: every P is 0 or 1 (exact in bf16)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf[3];
                    cut8_regs_t<H2>(st_cur, 8 * s2, H2Scales::P, pf);
                    if (SPA_ABLATE & 2) { asm volatile("" :: "v"(pf[0]), "v"(pf[1])); continue; }
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        bf16x8 vf[3];
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc) vf[pc] = read_tr8(Vc + pc * SIMG, lo, 16 * s2, db);
                        // smallest piece products first: (v2 p0), (v1 p1), (v0 p2), (v1 p0), (v0 p1), (v0 p0); a 0 / 1 P has one piece
                        oacc[db] = mfma_pieces<H2, ONE_PIECE>(vf, pf, oacc[db]);
                    }
                }
            };
            // MASK: evaluate the causal + level predicate per element (tiles on the diagonal, tiles holding a
            // blocked key); EMPTYSEL: some rows of the wave are "empty" rows (p = 1, normalised by 1/S at the end)
            auto softmax_tile = [&]<bool MASK, bool EMPTYSEL>() {
                if (SPA_ABLATE & 256) { pv_tile.template operator()<false>(); return; }     // timing-only: no softmax, no dropout
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
                // (H2: P is cut into fp16 pieces of P * 2^13 - it must stay below 2^3, so the reference follows the maximum closely)
                bool need = (l_run == 0.f) ? (mloc > -INFINITY) : (mloc > (H2 ? 2.f : RESCALE_TAU));
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
                    float pe = (SPA_ABLATE & 64) ? st_cur[reg] : __builtin_amdgcn_exp2f(st_cur[reg]);            // masked -> exp2(-inf) = 0
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
                pv_tile.template operator()<false>();
            };
            if (wave_all_empty) {
                // every row of the wave is an empty row: P = keep / span, no scores, no softmax state.  (One pass over the 16
                // values: with a second select pass over st_cur for the span the compiler kept sixteen 16-register copies of
                // it in scratch in the self-attention variants.)
                int kwv[16];
                if (DROP) read_key_quads(reinterpret_cast<const int32_t*>(km.kw), h, kwv);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    bool on = DROP ? rng.keep(aw, (uint32_t)kwv[reg]) : true;
                    on = on && (full_span || (j0 + rowmap(reg, h) < my_span));
                    st_cur[reg] = on ? 1.f : 0.f;
                }
                pv_tile.template operator()<true>();
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
        SPA_MARK(2);                                       // 2: softmax + dropout + cut of P + second products (issue)
        if (have_next && !beyond_next) {
            // scores of the next tile (minus the reference), taken BEFORE the barrier: after it other waves may
            // already overwrite this K slot with K(jt+3)
            const float init = H2 ? -m_ref * (sc.q * sc.k) : -m_ref;        // (H2: the products carry the operands' scales)
#pragma unroll
            for (int i = 0; i < 16; ++i) st_cur[i] = init;
            if ((jt + 1) & 1) { SPLIT_QK_TILE(st_cur, Ks[1]) } else { SPLIT_QK_TILE(st_cur, Ks[0]) }
            if (H2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st_cur[i] *= sc.inv_qk;
            }
        }
        SPA_MARK(3);                                       // 3: scores of the next tile (K fragment reads + products + scale)
        if (have_next2) {
            store_tile32_t<H2>(Ks[jt & 1], tid, rk, S - (j0 + 64), sc.k);   // K(jt) is dead: its scores were taken last iteration
            key_meta_store(kms[(jt + 2) % 3], w, lane, rmeta);
        }
        if (have_next) store_tile32_t<H2>(Vs[(jt + 1) & 1], tid, rv, S - (j0 + 32), sc.v);
        SPA_MARK(4);                                       // 4: wait for the loads, cut, LDS stores
        if (!(SPA_ABLATE & 128)) __syncthreads();
        SPA_MARK(5);                                       // 5: barrier
    }

    float omax = 0.f;
    if (valid_q) {
        float linv = my_empty ? invS : (l_run > 0.f ? 1.f / l_run : 0.f);
        if (DROP) linv *= rng.scale;
        if (H2) linv *= H2Scales::INV_P * sc.inv_v;
        float* orow = o + tok * (int64_t)nq * 64 + head * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = oacc[dh][4 * g4 + 0] * linv; t4.y = oacc[dh][4 * g4 + 1] * linv;
                t4.z = oacc[dh][4 * g4 + 2] * linv; t4.w = oacc[dh][4 * g4 + 3] * linv;
                *reinterpret_cast<float4*>(orow + 32 * dh + 8 * g4 + 4 * h) = t4;
                omax = fmaxf(fmaxf(fmaxf(omax, fabsf(t4.x)), fabsf(t4.y)), fmaxf(fabsf(t4.z), fabsf(t4.w)));
            }
        }

        // natural-log LSE of the scaled scores (what the backward kernels consume)
        if (h == 0) lse[((int64_t)b * nq + head) * S + iqc] =
            my_empty ? 0.f : (m_ref + __log2f(l_run)) * 0.6931471805599453f;
    }
    if (amax_lds) {                                                     // (gamer_amax_sink: max |o| for the o_proj GEMM's scale)
        uint32_t mw = __float_as_uint(omax);                            // one LDS atomic per wave, not 64 to one address
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
        if (lane == 0 && mw) atomicMax(amax_lds, mw);
    }
#if SPA_STAMP
    SPA_MARK(6);                                           // 6: the tile's epilogue (normalise, store o and lse)
    ph_[7] = (unsigned long long)n_iter;                   // 7: key-loop iterations of this tile
    if (g_attn_stamp && lane == 0) {
        unsigned long long* rec = g_attn_stamp + ((size_t)blockIdx.x * 4 + w) * 8;
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) rec[i_] += ph_[i_];
    }
#endif
}


template <int G, bool DROP, bool ORD, bool SPAN, bool H2>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_fwd_s_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                const float* __restrict__ v, int ldv, const int32_t* __restrict__ kl,
                const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                float* __restrict__ o, float* __restrict__ lse, const RowOrder ro, const int uspan,
                uint32_t* __restrict__ amax_out, const AttnAmax am) {
    constexpr int R = (4 / G) * 32;
    const H2Scales sc(H2 ? am.q : nullptr, am.k, am.v, nullptr, scale * 1.4426950408889634f);
    const int n_tiles = (S + R - 1) / R;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    __shared__ uint32_t amax_word;
    if (amax_out) {
        if (threadIdx.x == 0) amax_word = 0;
        __syncthreads();
    }
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
            attn_fwd_s_tile<G, DROP, ORD, SPAN, H2>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, S, nq, nkv, scale, p_drop, seed, o, lse,
                                        ro, pair, pass == 0 ? heavy : light, uspan, amax_out ? &amax_word : nullptr, sc);
        }
    }
    if (amax_out) {
        __syncthreads();
        if (threadIdx.x == 0 && amax_word) amax_publish(amax_word, amax_out, blockIdx.x);
    }
}


// =============================================================================================
// backward: dQ (attention.hip: attn_bwd_dq_tile, with the split fragments; K / V single-buffered, dO pieces in LDS)
// =============================================================================================
template <int G, bool DROP, bool ORD, bool SPAN, bool H2>
__device__ __forceinline__ void
attn_bwd_dq_s_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                   const float* __restrict__ v, int ldv, const float* __restrict__ o, const float* __restrict__ d_o,
                   const float* __restrict__ lse, float* __restrict__ delta,
                   const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                   const int32_t* __restrict__ row_empty, int S, int nq, int nkv, float scale,
                   float p_drop, uint64_t seed, float* __restrict__ dq, int lddq, const RowOrder ro, const int pair,
                   const int qtile, const int delta_ready, const H2Scales& sc) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    constexpr int NP = H2 ? 2 : 3;
    // K / V: ONE slot of three piece images each (the next tile waits in registers while this one is multiplied); dO: three
    // piece images of every wave's 32 query rows (the B operand of dP = V dO^T: in registers they would be 48 more VGPRs
    // next to the 48 of q's pieces).  24 + 24 + 48 KB: two workgroups per CU.
    __shared__ __attribute__((aligned(16))) bf16_t Ks[NP * SIMG];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[NP * SIMG];
    __shared__ __attribute__((aligned(16))) bf16_t dOs[4][NP * SIMG];
    __shared__ __attribute__((aligned(16))) KeyMeta kms[2];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int slot = q0 + sub * 32 + r;                 // sorted slot of this lane's query row
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iq_raw = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int iqc = iq_raw;
    const int64_t tok = (int64_t)b * S + iqc;

    // log2 domain: q is pre-scaled by scale*log2(e) and the score accumulators start at -lse*log2(e), so
    // p = exp2(accumulator)
    const float qs = scale * 1.4426950408889634f;
    bf16x8 qf[3][4];                      // piece x k-step of the pre-scaled q row (lane = query)
    float my_delta = 0.f;                 // delta_i = dO_i . O_i: computed here and published for the dK/dV kernel unless ready
    // H2: dO and dS are scaled PER QUERY ROW here - the query is on the lane in both products that consume them (dP^T = V dO^T,
    // dQ^T = K^T dS^T: the row index is never contracted), so the scale is a per-lane factor of the accumulators.  A row of dO
    // far below the tensor's maximum (a low-gradient token, or every other row once one row holds an outlier) then keeps its
    // full relative precision; the dK/dV kernel contracts OVER the rows and keeps the tensor's scale (a small row is a small term).
    float my_sdo = sc.d_o, my_inv_do = sc.inv_do, my_sds = sc.ds, my_inv_ds = sc.inv_ds;
    {
        const float* qrow = q + tok * ldq + head * 64 + 8 * h;
        const float* drow = d_o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
        const float* orow = o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
        bf16_t* dimg = dOs[w];
        float4 du[4], dw[4];
        float rowmax = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = *reinterpret_cast<const float4*>(qrow + 16 * s);
            float4 b4 = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
            a4.x *= qs; a4.y *= qs; a4.z *= qs; a4.w *= qs; b4.x *= qs; b4.y *= qs; b4.z *= qs; b4.w *= qs;
            bf16x8 pq[3];
            cut8_t<H2>(a4, b4, sc.q, pq);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) qf[pc][s] = pq[pc];
            float4 u4 = *reinterpret_cast<const float4*>(drow + 16 * s);
            float4 w4 = *reinterpret_cast<const float4*>(drow + 16 * s + 4);
            if (!delta_ready) {
                const float4 o4 = *reinterpret_cast<const float4*>(orow + 16 * s);
                const float4 p4 = *reinterpret_cast<const float4*>(orow + 16 * s + 4);
                my_delta += u4.x * o4.x + u4.y * o4.y + u4.z * o4.z + u4.w * o4.w + w4.x * p4.x + w4.y * p4.y + w4.z * p4.z + w4.w * p4.w;
            }
            if (!valid_q) { u4 = make_float4(0.f, 0.f, 0.f, 0.f); w4 = u4; }
            du[s] = u4; dw[s] = w4;
            if (H2) {
                rowmax = fmaxf(fmaxf(fmaxf(rowmax, fabsf(u4.x)), fabsf(u4.y)), fmaxf(fabsf(u4.z), fabsf(u4.w)));
                rowmax = fmaxf(fmaxf(fmaxf(rowmax, fabsf(w4.x)), fabsf(w4.y)), fmaxf(fabsf(w4.z), fabsf(w4.w)));
            }
        }
        if (H2) {
            rowmax = xor32_max(rowmax);                                    // the row's other 32 columns
            if (rowmax > 0.f) {                                          // (an all-zero row keeps the tensor's scales: its pieces are 0)
                scale_from_amax(__float_as_uint(rowmax), my_sdo, my_inv_do);
                scale_from_amax(__float_as_uint(sc.ds_coef * rowmax), my_sds, my_inv_ds);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 dpc[3];
            cut8_t<H2>(du[s], dw[s], my_sdo, dpc);
            // this lane's 8 values are columns 16 s + 8 h .. + 7 of row r: exactly what read_row8 hands back to it
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<bf16x8*>(dimg + pc * SIMG + lo.row[s]) = dpc[pc];
        }
        if (!delta_ready) {
            my_delta = xor32_sum(my_delta);
            if (valid_q && h == 0) delta[((int64_t)b * nq + head) * S + iqc] = my_delta;
        } else {
            my_delta = delta[((int64_t)b * nq + head) * S + iqc];
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq_raw, valid_q, S);
    const float neg_lse2 = -lse[((int64_t)b * nq + head) * S + iqc] * 1.4426950408889634f;
    const int wave_ql_min = wave_min_i32(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32(normal ? sp.hi : INT_BIG_A);
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;
    const int wave_hole_lo = SPAN ? wave_min_i32(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32(normal ? sp.hole_hi : 0) : 0;
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

    for (int jt = 0; jt < n_iter; ++jt) {
        const int cur = jt & 1;
        const bool more = jt + 1 < n_iter;
        const int j0 = jt * 32;
        __syncthreads();                              // the previous tile has been consumed by every wave
        store_tile32_t<H2>(Ks, tid, rk, S - j0, sc.k);
        store_tile32_t<H2>(Vs, tid, rv, S - j0, sc.v);
        key_meta_store(kms[cur], w, lane, rmeta);
        __syncthreads();
        if (more) {                                   // in flight while this tile is multiplied
            load_tile32(kbase, ldk, j0 + 32, S, tid, rk);
            load_tile32(vbase, ldv, j0 + 32, S, tid, rv);
            rmeta = key_meta_load<DROP>(klb, j0 + 32, S, w, lane, rng);
        }
        const bool beyond = j0 > wave_q_hi;
        if (!(beyond && !wave_has_empty)) {
            const KeyMeta& km = kms[cur];
            f32x16 st, dp;
            const float st0 = (H2 && !beyond) ? neg_lse2 * (sc.q * sc.k) : neg_lse2;   // (H2: the products carry the operands' scales)
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = st0; dp[i] = 0.f; }
            // S^T = K Q^T and dP^T[key][query] = sum_d V[key][d] dO[query][d]: two independent accumulation chains,
            // issued alternately (a single dependent chain of this MFMA runs 5-13 % below the pipe rate)
            const bf16_t* dimg = dOs[w];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 vf[3], df0[3];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) { vf[pc] = read_row8(Vs + pc * SIMG, lo, 0, s); df0[pc] = read_row8(dimg + pc * SIMG, lo, 0, s); }
                dp = mfma_pieces<H2>(vf, df0, dp);
            }
            if (!beyond) { SPLIT_QK_TILE(st, Ks) }
            if (H2) {
                const float cdp = sc.inv_v * my_inv_do;
#pragma unroll
                for (int i = 0; i < 16; ++i) dp[i] *= cdp;
                if (!beyond) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[i] *= sc.inv_qk;
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
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 df[3];
                    cut8_regs_t<H2>(ds, 8 * s2, my_sds, df);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        bf16x8 kf[3];
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc) kf[pc] = read_tr8(Ks + pc * SIMG, lo, 16 * s2, db);
                        dqacc[db] = mfma_pieces<H2>(kf, df, dqacc[db]);
                    }
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
    }

    if (valid_q) {
        float fs = my_empty ? scale * invS : scale;
        if (H2) fs *= my_inv_ds * sc.inv_k;
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

template <int G, bool DROP, bool ORD, bool SPAN, bool H2>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_bwd_dq_s_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                   const float* __restrict__ v, int ldv, const float* __restrict__ o, const float* __restrict__ d_o,
                   const float* __restrict__ lse, float* __restrict__ delta,
                   const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                   const int32_t* __restrict__ row_empty, int nbatch, int S, int nq, int nkv, float scale,
                   float p_drop, uint64_t seed, float* __restrict__ dq, int lddq, const RowOrder ro, const int delta_ready,
                   const AttnAmax am) {
    constexpr int R = (4 / G) * 32;
    const H2Scales sc(H2 ? am.q : nullptr, am.k, am.v, am.d_o, scale * 1.4426950408889634f, 1.f / (1.f - p_drop));
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
            attn_bwd_dq_s_tile<G, DROP, ORD, SPAN, H2>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, S, nq, nkv, scale,
                                           p_drop, seed, dq, lddq, ro, pair, pass == 0 ? heavy : light, delta_ready, sc);
        }
    }
}


// =============================================================================================
// backward: dK, dV (attention.hip: attn_bwd_dkv_tile without the dS spill, with the split fragments)
// =============================================================================================
// Workgroup = 4 G waves (G = 2: 512 threads), 128 keys of one (sequence, kv head): wave = (query head hg, 32 keys).
// LDS: the workgroup's K and V rows as three piece images each (96 KB: they are the B operands of S = Q K^T and
// dP = dO V^T for every query tile), the staged query tile's Q and dO images per head (12 KB each), per-query scalars.
// 144 KB at G = 2: ONE workgroup per CU, eight waves = two per SIMD, as the two four-wave workgroups of the fp32 form.
constexpr int DKV_S_KEYS = 128;
// one staged query tile: the Q and dO piece images of the G heads and the per-query scalars
template <int G, int NP>
struct DkvQTile {
    bf16_t Qs[G][NP * SIMG];
    bf16_t dOs[G][NP * SIMG];
    float nlse2_s[G][32];       // -lse * log2(e) of the staged query rows
    float ndelta_s[G][32];      // -delta
    uint32_t aw_s[G][32];       // dropout row words
    int32_t ql_s[32];
    int32_t empty_s[32];
    int32_t pos_s[32];          // key limit of the staged query rows: their position (session masks: QuerySpan.hi), S past the end
    int32_t hole_lo_s[32];      // QuerySpan hole of the staged query rows (SPAN kernels only)
    int32_t hole_hi_s[32];
    int32_t qlmin;              // smallest query level / key limit over the normal rows of the tile (INT_MAX: none)
    int32_t posmin;
    int32_t hole_lo_min;        // smallest hole start / largest hole end over the normal rows (SPAN kernels only)
    int32_t hole_hi_max;
};
// NP = 3 (six-product bf16 form): 96 KB of K / V images + one 48-KB query tile = 144 KB.  NP = 2 (three-product fp16 form):
// 64 KB + TWO 32-KB query tiles = 130 KB - tile t + 1 is cut and stored while tile t is multiplied, one barrier per tile.
template <int G, int NP = 3>
struct DkvSmemS {
    static constexpr int NBUF = NP == 2 ? 2 : 1;
    bf16_t Kt[NP][DKV_S_KEYS * 64];
    bf16_t Vt[NP][DKV_S_KEYS * 64];
    DkvQTile<G, NP> qt[NBUF];
};

template <int G, bool DROP, bool ORD, bool H2, bool SPAN = false>
__device__ __forceinline__ void
attn_bwd_dkv_s_tile(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                    const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                    const float* __restrict__ lse, const float* __restrict__ delta,
                    const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                    const int32_t* __restrict__ row_empty, const int32_t* __restrict__ tile_empty,
                    int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                    float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, const RowOrder ro, const int pair,
                    const int ktile, float* __restrict__ ds_out, uint32_t* __restrict__ amax_lds, const H2Scales& sc) {
    constexpr int NP = H2 ? 2 : 3;
    constexpr int THREADS = 256 * G;
    constexpr int R = DKV_S_KEYS;
    constexpr int NLD = 512 / THREADS;               // float4 per thread, tensor and head of a staged 32 x 64 tile
    extern __shared__ __attribute__((aligned(16))) unsigned char dkvs_raw[];
    DkvSmemS<G, NP>& sm = *reinterpret_cast<DkvSmemS<G, NP>*>(dkvs_raw);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w >> 2, sub = w & 3;
    __syncthreads();                 // the previous item's head reduction may still be reading the LDS region
    const int b = pair / nkv, kvh = pair % nkv, k0 = ktile * R;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int jk = k0 + sub * 32 + r;                 // this lane's key
    const bool valid_k = jk < S;
    const int jkc = valid_k ? jk : S - 1;
    const int64_t ktok = (int64_t)b * S + jkc;

    // stage this workgroup's K and V rows once: cut into the three piece images (zero rows past the end of the sequence)
    for (int f = tid; f < R * 16; f += THREADS) {
        const int row = f >> 4, c4 = (f & 15) << 2;
        const int j = k0 + row;
        float4 kv4 = make_float4(0.f, 0.f, 0.f, 0.f), vv4 = kv4;
        if (j < S) {
            kv4 = *reinterpret_cast<const float4*>(k + ((int64_t)b * S + j) * ldk + kvh * 64 + c4);
            vv4 = *reinterpret_cast<const float4*>(v + ((int64_t)b * S + j) * ldv + kvh * 64 + c4);
        }
        const int off = sl_off(row, c4);
        if (H2) {
            uint32_t a0, a1, b0, b1;
            cut2h_quad(kv4.x, kv4.y, kv4.z, kv4.w, sc.k, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.Kt[0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.Kt[1] + off) = make_uint2(a1, b1);
            cut2h_quad(vv4.x, vv4.y, vv4.z, vv4.w, sc.v, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.Vt[0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.Vt[1] + off) = make_uint2(a1, b1);
        } else {
            const float xk[4] = {kv4.x, kv4.y, kv4.z, kv4.w}, xv[4] = {vv4.x, vv4.y, vv4.z, vv4.w};
            uint32_t wk[4][3], wv[4][3];
#pragma unroll
            for (int e = 0; e < 4; ++e) { cut3(xk[e], wk[e]); cut3(xv[e], wv[e]); }
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                *reinterpret_cast<uint2*>(sm.Kt[s] + off) = make_uint2(hi16_pair(wk[0][s], wk[1][s]), hi16_pair(wk[2][s], wk[3][s]));
                *reinterpret_cast<uint2*>(sm.Vt[s] + off) = make_uint2(hi16_pair(wv[0][s], wv[1][s]), hi16_pair(wv[2][s], wv[3][s]));
            }
        }
    }
    const int my_kl = valid_k ? kl[ktok] : INT_BIG_A;
    const int wave_kl_max = wave_max_i32(my_kl);
    const int wave_k_lo = k0 + sub * 32, wave_k_hi = wave_k_lo + 31;
    const float invS = 1.f / (float)S;
    // (H2: the products carry their operands' scales - folded into the constants that multiply them anyway)
    const float c2 = scale * 1.4426950408889634f * (H2 ? sc.inv_qk : 1.f);
    const AttnDropout rng(p_drop, seed);
    const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
    const float cdp = H2 ? sc.inv_v * sc.inv_do : 1.f;     // dP = (dO' V'^T) cdp
    const float sd = rng.scale * cdp;
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

    // staged registers: per head one Q tile and one dO tile (NLD float4 each per thread); the NEXT tile's loads are in
    // flight while the current one is multiplied
    float4 rq[G][NLD], rdo[G][NLD];
    float rl = 0.f, rd = 0.f;
    uint32_t raw = 0;
    int rql = 1, rem = 0, rpos = 0, rhlo = INT_BIG_A, rhhi = 0;

    auto next_tile = [&](int qt) {
        while (qt < n_qt && !tile_empty_rows(qt) && tile_last_pos(qt) < k0) ++qt;
        return qt;
    };
    auto load_q_tile = [&](int qt) {
        const int i0 = qt * 32;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int hd = kvh * G + g;
#pragma unroll
            for (int jj = 0; jj < NLD; ++jj) {
                const int f = tid + THREADS * jj;
                const int sl = min(i0 + (f >> 4), S - 1);
                const int pos = ORD ? pmap[sl] : sl;
                rq[g][jj] = *reinterpret_cast<const float4*>(q + ((int64_t)b * S + pos) * ldq + hd * 64 + ((f & 15) << 2));
                rdo[g][jj] = *reinterpret_cast<const float4*>(d_o + ((int64_t)b * S + pos) * (int64_t)nq * 64 + hd * 64 + ((f & 15) << 2));
            }
        }
        // per-query scalars: waves 0 .. G-1 take (g = wave, row = lane & 31); both lane halves compute the same values
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
                const QuerySpan qsp = QuerySpan::load<SPAN>(ro.span, (int64_t)b * S + i, i, in, S);
                rpos = qsp.hi;
                if (SPAN) { rhlo = qsp.hole_lo; rhhi = qsp.hole_hi; }
            }
        }
    };
    auto store_q_tile = [&](int qt, DkvQTile<G, NP>& dst) {
        const int n_valid = S - qt * 32;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int jj = 0; jj < NLD; ++jj) {
                const int f = tid + THREADS * jj;
                const int row = f >> 4;
                const bool ok = row < n_valid;
                const float4 a = rq[g][jj], c = rdo[g][jj];
                const float xa[4] = {ok ? a.x : 0.f, ok ? a.y : 0.f, ok ? a.z : 0.f, ok ? a.w : 0.f};
                const float xc[4] = {ok ? c.x : 0.f, ok ? c.y : 0.f, ok ? c.z : 0.f, ok ? c.w : 0.f};
                const int off = sl_off(row, (f & 15) << 2);
                if (H2 && (SPA_ABLATE & 32)) { asm volatile("" :: "v"(xa[0]), "v"(xa[3]), "v"(xc[1]), "v"(xc[2])); continue; }
                if (H2) {
                    uint32_t a0, a1, b0, b1;
                    cut2h_quad(xa[0], xa[1], xa[2], xa[3], sc.q, a0, a1, b0, b1);
                    *reinterpret_cast<uint2*>(dst.Qs[g] + off) = make_uint2(a0, b0);
                    *reinterpret_cast<uint2*>(dst.Qs[g] + SIMG + off) = make_uint2(a1, b1);
                    cut2h_quad(xc[0], xc[1], xc[2], xc[3], sc.d_o, a0, a1, b0, b1);
                    *reinterpret_cast<uint2*>(dst.dOs[g] + off) = make_uint2(a0, b0);
                    *reinterpret_cast<uint2*>(dst.dOs[g] + SIMG + off) = make_uint2(a1, b1);
                } else {
                    uint32_t wa[4][3], wc[4][3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cut3(xa[e], wa[e]); cut3(xc[e], wc[e]); }
#pragma unroll
                    for (int s = 0; s < NP; ++s) {
                        *reinterpret_cast<uint2*>(dst.Qs[g] + s * SIMG + off) = make_uint2(hi16_pair(wa[0][s], wa[1][s]), hi16_pair(wa[2][s], wa[3][s]));
                        *reinterpret_cast<uint2*>(dst.dOs[g] + s * SIMG + off) = make_uint2(hi16_pair(wc[0][s], wc[1][s]), hi16_pair(wc[2][s], wc[3][s]));
                    }
                }
            }
        if (w < G) {
            const int g = w, row = lane & 31;
            int qlmin = 0, posmin = 0, hlomin = INT_BIG_A, hhimax = 0;
            if (g == 0) {
                const bool in = qt * 32 + row < S;
                const bool normal = in && rem == 0;
                qlmin = wave_min_i32(normal ? rql : INT_BIG_A);
                posmin = wave_min_i32(normal ? rpos : INT_BIG_A);
                if (SPAN) {
                    hlomin = wave_min_i32(normal ? rhlo : INT_BIG_A);
                    hhimax = wave_max_i32(normal ? rhhi : 0);
                }
            }
            if (lane < 32) {
                dst.nlse2_s[g][row] = rl;
                dst.ndelta_s[g][row] = rd;
                if (DROP) dst.aw_s[g][row] = raw;
                if (g == 0) {
                    dst.ql_s[row] = rql; dst.empty_s[row] = rem; dst.pos_s[row] = rpos;
                    if (SPAN) { dst.hole_lo_s[row] = rhlo; dst.hole_hi_s[row] = rhhi; }
                    if (row == 0) {
                        dst.qlmin = qlmin; dst.posmin = posmin;
                        if (SPAN) { dst.hole_lo_min = hlomin; dst.hole_hi_max = hhimax; }
                    }
                }
            }
        }
    };

    // the products of one staged query tile `tq` (qt) against this wave's 32 keys
    auto compute_tile = [&](const int qt, const DkvQTile<G, NP>& tq, bool& prefetched, const int qt_next) {
        const bool tile_has_empty = tile_empty_rows(qt);
        const int posmin = __builtin_amdgcn_readfirstlane(tq.posmin);
        const int qlmin = __builtin_amdgcn_readfirstlane(tq.qlmin);
        const bool tile_all_empty = tile_has_empty && posmin == INT_BIG_A;

        const bool before = tile_last_pos(qt) < wave_k_lo;   // every normal query of the tile precedes this wave's keys
        if (!(before && !tile_has_empty)) {
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
            const bf16_t* Qh = tq.Qs[hg];
            const bf16_t* dOh = tq.dOs[hg];
            // S[query][key] = sum_d Q[query][d] K[key][d] and dP[query][key] = sum_d dO[query][d] V[key][d]; six piece
            // products each, smallest first
            const bool need_s = !before && !tile_all_empty;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 cf[3], vf[3];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) { cf[pc] = read_row8(dOh + pc * SIMG, lo, 0, s); vf[pc] = read_row8(sm.Vt[pc], lo, 32 * sub, s); }
                dp = mfma_pieces<H2>(cf, vf, dp);
                if (need_s) {
                    bf16x8 af[3], bfr[3];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) { af[pc] = read_row8(Qh + pc * SIMG, lo, 0, s); bfr[pc] = read_row8(sm.Kt[pc], lo, 32 * sub, s); }
                    st = mfma_pieces<H2>(af, bfr, st);
                }
            }
            // per element (query = register, key = lane):  P -> st (for dV, 1/(1-p) applied at the end),
            //                                               dS -> dp (for dK)
            // Rows past the end of the sequence have Q = dO = 0, lse = delta = 0: p = 1 but dS = 0 and dO = 0.
            auto elem_tile = [&]<bool MASK, bool EMPTYSEL, bool ALL_EMPTY>() {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int qb = 8 * g4 + 4 * h;
                    const float4 d4 = *reinterpret_cast<const float4*>(&tq.ndelta_s[hg][qb]);
                    const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
                    float nl[4] = {0.f, 0.f, 0.f, 0.f};
                    int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0}, emv[4] = {0, 0, 0, 0};
                    int hlo[4] = {0, 0, 0, 0}, hhi[4] = {0, 0, 0, 0};
                    uint32_t awv[4] = {0, 0, 0, 0};
                    if (!ALL_EMPTY) {
                        const float4 l4 = *reinterpret_cast<const float4*>(&tq.nlse2_s[hg][qb]);
                        nl[0] = l4.x; nl[1] = l4.y; nl[2] = l4.z; nl[3] = l4.w;
                    }
                    if (MASK) {
                        const int4 q4 = *reinterpret_cast<const int4*>(&tq.ql_s[qb]);
                        const int4 p4 = *reinterpret_cast<const int4*>(&tq.pos_s[qb]);
                        qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
                        posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
                        if (SPAN) {
                            const int4 a4 = *reinterpret_cast<const int4*>(&tq.hole_lo_s[qb]);
                            const int4 b4 = *reinterpret_cast<const int4*>(&tq.hole_hi_s[qb]);
                            hlo[0] = a4.x; hlo[1] = a4.y; hlo[2] = a4.z; hlo[3] = a4.w;
                            hhi[0] = b4.x; hhi[1] = b4.y; hhi[2] = b4.z; hhi[3] = b4.w;
                        }
                    }
                    if (EMPTYSEL) {
                        const int4 e4 = *reinterpret_cast<const int4*>(&tq.empty_s[qb]);
                        emv[0] = e4.x; emv[1] = e4.y; emv[2] = e4.z; emv[3] = e4.w;
                    }
                    if (DROP) {
                        const uint4 a4 = *reinterpret_cast<const uint4*>(&tq.aw_s[hg][qb]);
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
                            dp[reg] = pe * (H2 ? fmaf(dp[reg], cdp, ndl[e]) : dp[reg] + ndl[e]);
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
                    const int hl = __builtin_amdgcn_readfirstlane(tq.hole_lo_min);
                    const int hh = __builtin_amdgcn_readfirstlane(tq.hole_hi_max);
                    free_tile = free_tile && (wave_k_hi < hl || wave_k_lo >= hh);
                }
                if (free_tile) elem_tile.template operator()<false, false, false>();
                else elem_tile.template operator()<true, false, false>();
            }
            if (ds_out != nullptr && (k0 >> 5) + sub < n_qt) {
                // spill dS (fp32) for the dQ kernel: tile (head, query tile qt, this wave's 32-key tile) as [query][key]
                const int head = kvh * G + hg;
                float* tile = ds_out + ((((int64_t)b * nq + head) * n_qt + qt) * n_qt + (k0 >> 5) + sub) * 1024;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) tile[rowmap(reg, h) * 32 + r] = dp[reg];
            }
            if (!prefetched) {
                prefetched = true;
                if (qt_next < n_qt) load_q_tile(qt_next);
            }
            // dV^T[d][key] += sum_query dO[query][d] Pd[query][key] ; dK^T[d][key] += sum_query Q[query][d] dS[query][key]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf[3], df[3];
                cut8_regs_t<H2>(st, 8 * s2, H2Scales::P, pf);
                cut8_regs_t<H2>(dp, 8 * s2, sc.ds, df);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    bf16x8 of[3], qfr[3];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) of[pc] = read_tr8(dOh + pc * SIMG, lo, 16 * s2, db);
                    dvacc[db] = mfma_pieces<H2>(of, pf, dvacc[db]);
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) qfr[pc] = read_tr8(Qh + pc * SIMG, lo, 16 * s2, db);
                    dkacc[db] = mfma_pieces<H2>(qfr, df, dkacc[db]);
                }
            }
        }
    };

    int qt = next_tile(0);
    if (H2) {
        // two query-tile buffers: tile t + 1 (loaded from global memory during tile t - 1's products ... ) is cut and stored into the
        // other buffer right after this wave's products of tile t - ONE barrier per tile, and no phase in which the whole
        // workgroup only stores
        int buf = 0;
        if (qt < n_qt) {
            load_q_tile(qt);
            store_q_tile(qt, sm.qt[0]);
        }
        __syncthreads();                 // K / V staging and the first query tile visible
        int qt_next = qt < n_qt ? next_tile(qt + 1) : n_qt;
        if (qt_next < n_qt) load_q_tile(qt_next);
        while (qt < n_qt) {
            bool prefetched = true;
            compute_tile(qt, sm.qt[buf], prefetched, qt_next);
            if (qt_next < n_qt) store_q_tile(qt_next, sm.qt[DkvSmemS<G, NP>::NBUF - 1 - buf]);
            __syncthreads();             // tile qt consumed by every wave, tile qt_next visible
            qt = qt_next;
            buf = DkvSmemS<G, NP>::NBUF - 1 - buf;
            qt_next = qt < n_qt ? next_tile(qt + 1) : n_qt;
            if (qt_next < n_qt) load_q_tile(qt_next);
        }
    } else {
        if (qt < n_qt) load_q_tile(qt);
        while (qt < n_qt) {
            __syncthreads();                 // previous tile fully consumed (and K/V staging visible)
            store_q_tile(qt, sm.qt[0]);
            __syncthreads();
            const int qt_next = next_tile(qt + 1);
            // the next tile's global loads are in flight while this one is multiplied.  With the dS spill they are issued BEHIND
            // the spill stores (the wait for them in front of the next LDS store is s_waitcnt vmcnt(0): issued ahead of the
            // stores, every tile would wait for its own spill to be acknowledged)
            bool prefetched = ds_out == nullptr;
            if (prefetched && qt_next < n_qt) load_q_tile(qt_next);
            compute_tile(qt, sm.qt[0], prefetched, qt_next);
            if (!prefetched && qt_next < n_qt) load_q_tile(qt_next);
            qt = qt_next;
        }
    }

    // ---- sum the G query heads of this kv head through LDS, then store ---------------------------
    // LDS image [sub][key 32][132] floats (dK 64 | dV 64) reuses the tile region.
    float* red = reinterpret_cast<float*>(dkvs_raw);
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
        const float vs = (DROP ? rng.scale : 1.f) * (H2 ? sc.inv_do * H2Scales::INV_P : 1.f);
        const float ks = scale * (H2 ? sc.inv_q * sc.inv_ds : 1.f);
        float vmax = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dh + 8 * g4 + 4 * h;
                *reinterpret_cast<float4*>(dkrow + d) = make_float4(dkacc[dh][4 * g4] * ks, dkacc[dh][4 * g4 + 1] * ks,
                                                                    dkacc[dh][4 * g4 + 2] * ks, dkacc[dh][4 * g4 + 3] * ks);
                const float4 dv4 = make_float4(dvacc[dh][4 * g4] * vs, dvacc[dh][4 * g4 + 1] * vs,
                                               dvacc[dh][4 * g4 + 2] * vs, dvacc[dh][4 * g4 + 3] * vs);
                *reinterpret_cast<float4*>(dvrow + d) = dv4;
                vmax = fmaxf(fmaxf(fmaxf(vmax, fabsf(dv4.x)), fabsf(dv4.y)), fmaxf(fabsf(dv4.z), fabsf(dv4.w)));
            }
        if (amax_lds) atomicMax(amax_lds, __float_as_uint(vmax));      // (gamer_amax_sink: max |dv|, the v columns of d(q|k|v); once per key tile)
    }
}

// one workgroup per CU: the work list hands out (heavy key tile + light key tile) items, 32 slots per XCD
static inline int worklist_grid_1(int n_pairs, int n_tiles) {
    const int halves = (n_tiles + 1) / 2;
    int per_xcd = (n_pairs + 7) / 8;
    int rounds_cap = 32 / halves;
    if (rounds_cap < 1) rounds_cap = 1;
    const int ppr = per_xcd < rounds_cap ? per_xcd : rounds_cap;
    return 8 * ppr * halves;
}

template <int G, bool DROP, bool ORD, bool H2, bool SPAN = false>
__global__ void __launch_bounds__(256 * G, 1)
attn_bwd_dkv_s_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                      const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                      const float* __restrict__ lse, const float* __restrict__ delta,
                      const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                      const int32_t* __restrict__ row_empty, const int32_t* __restrict__ tile_empty,
                      int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                      float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, const RowOrder ro,
                      float* __restrict__ ds_out, uint32_t* __restrict__ amax_out, const AttnAmax am) {
    const H2Scales sc(H2 ? am.q : nullptr, am.k, am.v, am.d_o, 1.f, 1.f / (1.f - p_drop));
    const int n_tiles = (S + DKV_S_KEYS - 1) / DKV_S_KEYS;
    const WorkList wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    __shared__ uint32_t amax_word;
    if (amax_out) {
        if (threadIdx.x == 0) amax_word = 0;
        __syncthreads();
    }
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int heavy = wl.u, light = n_tiles - 1 - wl.u;       // earlier key tiles are seen by more queries
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dkv_s_tile<G, DROP, ORD, H2, SPAN>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, tile_empty, S, nq, nkv,
                                              scale, p_drop, seed, dk, lddk, dv, lddv, ro, pair, pass == 0 ? heavy : light, ds_out,
                                              amax_out ? &amax_word : nullptr, sc);
        }
    }
    if (amax_out) {
        __syncthreads();
        if (threadIdx.x == 0 && amax_word) amax_publish(amax_word, amax_out, blockIdx.x);
    }
}

// =============================================================================================
// backward: dQ = dS K from the dS tiles the dK/dV kernel spilled (attention.hip: attn_bwd_dq2_tile, split fragments):
// one product per tile pair instead of the three of the recompute form, for 2 x 4 KB of HBM traffic per tile pair
// =============================================================================================
// delta[b,h,i] = sum_d dO * O
__global__ void __launch_bounds__(AT_THREADS)
attn_delta_s_kernel(const float* __restrict__ o, const float* __restrict__ d_o, int B, int S, int nq, float* __restrict__ delta) {
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
        float sum = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (live && g == 0) {
            const int64_t t = i / nq;
            delta[((t / S) * nq + (int)(i % nq)) * S + t % S] = sum;
        }
    }
}

template <int G, bool ORD>
__device__ __forceinline__ void
attn_bwd_dq2_s_tile(const float* __restrict__ k, int ldk, const float* __restrict__ ds_in,
                  const int32_t* __restrict__ row_empty, int S, int nq, int nkv, float scale,
                  float* __restrict__ dq, int lddq, const RowOrder ro, const int pair, const int qtile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][3 * SIMG];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
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
    store_tile32_split(Ks[0], tid, rk, S);
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
            f32x16 st;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { st[4 * g4] = tu[g4].x; st[4 * g4 + 1] = tu[g4].y; st[4 * g4 + 2] = tu[g4].z; st[4 * g4 + 3] = tu[g4].w; }
            // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]   (rows of K past the sequence end are zero)
            const bf16_t* Kc = Ks[cur];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 df[3];
                cut8_regs(st, 8 * s2, df[0], df[1], df[2]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 k0 = read_tr8(Kc, lo, 16 * s2, db), k1 = read_tr8(Kc + SIMG, lo, 16 * s2, db), k2 = read_tr8(Kc + 2 * SIMG, lo, 16 * s2, db);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, df[2], dqacc[db], 0, 0, 0);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, df[1], dqacc[db], 0, 0, 0);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, df[0], dqacc[db], 0, 0, 0);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, df[1], dqacc[db], 0, 0, 0);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, df[0], dqacc[db], 0, 0, 0);
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, df[0], dqacc[db], 0, 0, 0);
                }
            }
        }
        if (jt + 3 < n_iter && live(jt + 3)) load_ds(jt + 3, tu);
        if (more) store_tile32_split(Ks[cur ^ 1], tid, rk, S - (jt * 32 + 32));
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
attn_bwd_dq2_s_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ ds_in,
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
            attn_bwd_dq2_s_tile<G, ORD>(k, ldk, ds_in, row_empty, S, nq, nkv, scale, dq, lddq, ro, pair,
                                      pass == 0 ? heavy : light);
        }
    }
}

// the sink armed for the entry point in progress (gamer_amax_sink): max |o| of the forward, max |dv| of the backward
static thread_local uint32_t* t_amax_out = nullptr;
// the operand maxima armed for the entry point in progress (gamer_attn_split_amax): non-null q selects the H2 form
static thread_local AttnAmax t_attn_amax = {nullptr, nullptr, nullptr, nullptr};
static thread_local AttnAmax g_attn_amax_armed = {nullptr, nullptr, nullptr, nullptr};
void disarm_attn_amax() { g_attn_amax_armed = AttnAmax{nullptr, nullptr, nullptr, nullptr}; }

template <int G>
static int launch_fwd_s(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* kl,
                        const int32_t* ql, const int32_t* row_empty, int B, int S, int nq, int nkv, float scale,
                        float p_drop, uint64_t seed, float* o, float* lse, RowOrder ro, int uspan, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    dim3 grid(worklist_grid(B * nkv, (S + R - 1) / R));
#define GAMER_LAUNCH_FWD_SP(DROPV, ORDV, SPANV, H2V)                                                                         \
    hipLaunchKernelGGL((attn_fwd_s_kernel<G, DROPV, ORDV, SPANV, H2V>), grid, dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, \
                       kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, t_amax_out, t_attn_amax)
#define GAMER_LAUNCH_FWD_S(DROPV, ORDV, H2V) GAMER_LAUNCH_FWD_SP(DROPV, ORDV, false, H2V)
    if (ro.span) {
        // per-query key spans (Qwen3SessionMulti): built for the three-product form (the entry point has checked that it is armed)
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_SP(true, true, true, true); else GAMER_LAUNCH_FWD_SP(true, false, true, true); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_SP(false, true, true, true); else GAMER_LAUNCH_FWD_SP(false, false, true, true); }
    } else if (t_attn_amax.q) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_S(true, true, true); else GAMER_LAUNCH_FWD_S(true, false, true); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_S(false, true, true); else GAMER_LAUNCH_FWD_S(false, false, true); }
    } else {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_S(true, true, false); else GAMER_LAUNCH_FWD_S(true, false, false); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_S(false, true, false); else GAMER_LAUNCH_FWD_S(false, false, false); }
    }
#undef GAMER_LAUNCH_FWD_S
#undef GAMER_LAUNCH_FWD_SP
    GAMER_CHECK_LAUNCH("gamer_attn_fwd_split");
    return 0;
}

template <int G, bool DROP, bool ORD>
static int launch_bwd_s_variant(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                const float* d_o, const float* lse, float* delta, const int32_t* kl, const int32_t* ql,
                                const int32_t* row_empty, const int32_t* tile_empty, int B, int S, int nq, int nkv,
                                float scale, float p_drop, uint64_t seed, float* dq, int lddq, float* dk, int lddk,
                                float* dv, int lddv, RowOrder ro, int delta_ready, float* ds_work, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    const bool h2 = t_attn_amax.q != nullptr;       // gamer_attn_split_amax: the three-product fp16 form (recompute form only)
    // key spans on the resident kernels: the GQA group of two, both kernels resident (the conditions of the two launches below)
    const bool span_res = ro.span && G == 2 && h2 && ds_work == nullptr && res_enabled() && S <= 2048 &&
                          GAMER_RES_PART("GAMER_ATTN_RES_SPAN") && GAMER_RES_PART("GAMER_ATTN_RES_DQ") &&
                          GAMER_RES_PART("GAMER_ATTN_RES_DKV") && (res_fill(B * nkv) >= 0.85 || res_split_forced());
    if (ro.span && !span_res) {
        // per-query key spans (Qwen3SessionMulti): the three-product recompute form (checked by the entry point)
        hipLaunchKernelGGL((attn_bwd_dq_s_kernel<G, DROP, ORD, true, true>), dim3(worklist_grid(B * nkv, (S + R - 1) / R)),
                           dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv,
                           scale, p_drop, seed, dq, lddq, ro, delta_ready, t_attn_amax);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dq");
        const size_t shmem = sizeof(DkvSmemS<G, 2>);
        static bool attr_dev[MAX_DEVICES] = {};
        bool& attr_set = attr_dev[current_device()];
        if (!attr_set) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_s_kernel<G, DROP, ORD, true, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) {
                set_error("gamer_attn_bwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return (int)e;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_bwd_dkv_s_kernel<G, DROP, ORD, true, true>), dim3(worklist_grid_1(B * nkv, (S + DKV_S_KEYS - 1) / DKV_S_KEYS)),
                           dim3(256 * G), shmem, st, q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv,
                           scale, p_drop, seed, dk, lddk, dv, lddv, ro, (float*)nullptr, t_amax_out, t_attn_amax);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dkv");
        return 0;
    }
    if (ds_work == nullptr) {
        if (h2 && res_enabled() && S <= 2048 && GAMER_RES_PART("GAMER_ATTN_RES_DQ")) {
            // K / V of a (sequence, kv head) resident in LDS (attention_res.hip)
            const int rc_dq = launch_dq_res(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop,
                                            seed, dq, lddq, ro, delta_ready, t_attn_amax, st);
            if (rc_dq) return rc_dq;
        } else if (h2)
            hipLaunchKernelGGL((attn_bwd_dq_s_kernel<G, DROP, ORD, false, true>), dim3(worklist_grid(B * nkv, (S + R - 1) / R)),
                               dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv,
                               scale, p_drop, seed, dq, lddq, ro, delta_ready, t_attn_amax);
        else
            hipLaunchKernelGGL((attn_bwd_dq_s_kernel<G, DROP, ORD, false, false>), dim3(worklist_grid(B * nkv, (S + R - 1) / R)),
                               dim3(AT_THREADS), 0, st, q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv,
                               scale, p_drop, seed, dq, lddq, ro, delta_ready, t_attn_amax);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dq");
    } else if (!delta_ready) {
        hipLaunchKernelGGL(attn_delta_s_kernel, dim3(2048), dim3(AT_THREADS), 0, st, o, d_o, B, S, nq, delta);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/delta");
    }
    const int n_kt = (S + DKV_S_KEYS - 1) / DKV_S_KEYS;
    auto launch_dkv = [&]<bool H2V>() -> int {
        const size_t shmem = sizeof(DkvSmemS<G, H2V ? 2 : 3>);
        static bool attr_dev[MAX_DEVICES] = {};   // one set of flags per template instantiation, one flag per device
        bool& attr_set = attr_dev[current_device()];
        if (!attr_set) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_s_kernel<G, DROP, ORD, H2V>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) {
                set_error("gamer_attn_bwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return (int)e;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_bwd_dkv_s_kernel<G, DROP, ORD, H2V>), dim3(worklist_grid_1(B * nkv, n_kt)), dim3(256 * G), shmem, st, q, ldq,
                           k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dk,
                           lddk, dv, lddv, ro, ds_work, t_amax_out, t_attn_amax);
        return 0;
    };
    // (the resident dK/dV kernel takes whole pairs: when they do not fill the CUs evenly - per-GPU batch 128: 384 pairs on 256 CUs -
    // the tiled kernel below is faster than either half-filled rounds or pairs split over two workgroups; measured)
    if (h2 && ds_work == nullptr && res_enabled() && S <= 2048 && GAMER_RES_PART("GAMER_ATTN_RES_DKV") &&
        (res_fill(B * nkv) >= 0.85 || res_split_forced()))
        // Q / dO of a (sequence, head, query block) resident in LDS, key tiles from a queue (attention_res.hip)
        return launch_dkv_res(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv,
                              lddv, ro, t_amax_out, t_attn_amax, st);
    const int rc_dkv = h2 ? launch_dkv.template operator()<true>() : launch_dkv.template operator()<false>();
    if (rc_dkv) return rc_dkv;
    GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dkv");
    if (ds_work != nullptr) {
        hipLaunchKernelGGL((attn_bwd_dq2_s_kernel<G, ORD>), dim3(worklist_grid(B * nkv, (S + R - 1) / R)), dim3(AT_THREADS), 0, st,
                           k, ldk, ds_work, row_empty, B, S, nq, nkv, scale, dq, lddq, ro);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dq2");
    }
    return 0;
}

template <int G>
static int launch_bwd_s(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                        const float* d_o, const float* lse, float* delta, const int32_t* kl, const int32_t* ql,
                        const int32_t* row_empty, const int32_t* tile_empty, int B, int S, int nq, int nkv, float scale,
                        float p_drop, uint64_t seed, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                        RowOrder ro, int delta_ready, float* ds_work, hipStream_t st) {
#define GAMER_LAUNCH_BWD_S(DROPV, ORDV)                                                                                       \
    return launch_bwd_s_variant<G, DROPV, ORDV>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, \
                                                nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, delta_ready, ds_work, st)
    if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_BWD_S(true, true); else GAMER_LAUNCH_BWD_S(true, false); }
    else { if (ro.perm) GAMER_LAUNCH_BWD_S(false, true); else GAMER_LAUNCH_BWD_S(false, false); }
#undef GAMER_LAUNCH_BWD_S
    return -1;      // not reached
}

}  // namespace gamer

using namespace gamer;

static int check_attn_split(const char* name, const void* q, const void* k, const void* v, const void* kl,
                            const void* row_empty, int ldq, int ldk, int ldv, int B, int S, int nq, int nkv, float p_drop) {
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

extern "C" int gamer_attn_fwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const int32_t* kl, const int32_t* ql, const int32_t* row_empty, int B, int S, int nq,
                                    int nkv, float scale, float p_drop, uint64_t seed, float* o, float* lse,
                                    const int32_t* row_perm, const int32_t* tile_kind, const int32_t* tile_maxpos,
                                    int uniform_len, const int32_t* q_span, void* stream) {
    GAMER_CHECK_ARG(!row_perm || (tile_kind && tile_maxpos), "gamer_attn_fwd_split: row_perm needs tile_kind and tile_maxpos");
    GAMER_CHECK_ARG(uniform_len >= 0 && uniform_len <= S, "gamer_attn_fwd_split: uniform_len=%d outside [0, S=%d]", uniform_len, S);
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_fwd_split: q_span must be 16-byte aligned");
    const int uspan = uniform_len == 0 ? S : uniform_len;
    const RowOrder ro{row_perm, tile_kind, tile_maxpos, q_span};
    int rc = check_attn_split("gamer_attn_fwd_split", q, k, v, kl, row_empty, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && lse && aligned16(o), "gamer_attn_fwd_split: null/unaligned output");
    hipStream_t st = (hipStream_t)stream;
    t_amax_out = take_amax_sink().out[0];               // gamer_amax_sink: max |o|
    t_attn_amax = g_attn_amax_armed;                    // gamer_attn_split_amax: the three-product fp16 form
    g_attn_amax_armed = AttnAmax{nullptr, nullptr, nullptr, nullptr};
    GAMER_CHECK_ARG(!t_attn_amax.q || (t_attn_amax.k && t_attn_amax.v), "gamer_attn_fwd_split: gamer_attn_split_amax needs q, k and v");
    GAMER_CHECK_ARG(!t_attn_amax.q || p_drop < 0.75f, "gamer_attn_fwd_split: the three-product fp16 form needs p_drop < 0.75 (p_drop=%f): "
                    "its probabilities are cut at a fixed 2^13 scale; do not arm gamer_attn_split_amax for this call", p_drop);
    GAMER_CHECK_ARG(!q_span || t_attn_amax.q, "gamer_attn_fwd_split: q_span is built for the three-product form (arm gamer_attn_split_amax)");
    if (t_attn_amax.q && (!q_span || (nq / nkv == 2 && GAMER_RES_PART("GAMER_ATTN_RES_SPAN"))) && uspan == S && S <= 2048 && res_enabled())
        // the three-product form with training semantics: K / V of a (sequence, kv head) resident in LDS (attention_res.hip; key
        // spans - Qwen3SessionMulti - for the GQA group of two)
        rc = launch_fwd_res(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, t_amax_out, t_attn_amax, st);
    else
    rc = (nq / nkv == 1) ? launch_fwd_s<1>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, st)
                         : launch_fwd_s<2>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, uspan, st);
    t_amax_out = nullptr;
    t_attn_amax = AttnAmax{nullptr, nullptr, nullptr, nullptr};
    return rc;
}

#if SPA_STAMP
extern "C" int gamer_debug_attn_stamp(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(gamer::g_attn_stamp), &v, sizeof(v));
}
#endif

extern "C" int gamer_attn_split_amax(const uint32_t* q, const uint32_t* k, const uint32_t* v, const uint32_t* d_o) {
    g_attn_amax_armed = AttnAmax{q, k, v, d_o};
    return 0;
}

extern "C" int gamer_attn_bwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* o, const float* d_o, const float* lse, const int32_t* kl,
                                    const int32_t* ql, const int32_t* row_empty, const int32_t* tile_empty, int B, int S,
                                    int nq, int nkv, float scale, float p_drop, uint64_t seed, float* delta, float* dq,
                                    int lddq, float* dk, int lddk, float* dv, int lddv, const int32_t* row_perm,
                                    const int32_t* tile_kind, const int32_t* tile_maxpos, int delta_ready, float* ds_work,
                                    const int32_t* q_span, void* stream) {
    int rc = check_attn_split("gamer_attn_bwd_split", q, k, v, kl, row_empty, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && d_o && lse && tile_empty && delta && dq && dk && dv, "gamer_attn_bwd_split: null pointer");
    GAMER_CHECK_ARG(!row_perm || (tile_kind && tile_maxpos), "gamer_attn_bwd_split: row_perm needs tile_kind and tile_maxpos");
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_bwd_split: q_span must be 16-byte aligned");
    const RowOrder ro{row_perm, tile_kind, tile_maxpos, q_span};
    GAMER_CHECK_ARG(lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && aligned16(dq) && aligned16(dk) && aligned16(dv) &&
                    aligned16(d_o) && aligned16(o),
                    "gamer_attn_bwd_split: gradient buffers must be 16-byte aligned with leading dims %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    t_amax_out = take_amax_sink().out[0];               // gamer_amax_sink: max |dv|
    t_attn_amax = g_attn_amax_armed;
    g_attn_amax_armed = AttnAmax{nullptr, nullptr, nullptr, nullptr};
    GAMER_CHECK_ARG(!t_attn_amax.q || (t_attn_amax.k && t_attn_amax.v && t_attn_amax.d_o && !ds_work),
                    "gamer_attn_bwd_split: gamer_attn_split_amax needs q, k, v and d_o, and the recompute form (ds_work = NULL)");
    GAMER_CHECK_ARG(!t_attn_amax.q || p_drop < 0.75f, "gamer_attn_bwd_split: the three-product fp16 form needs p_drop < 0.75 (p_drop=%f)", p_drop);
    GAMER_CHECK_ARG(!q_span || (t_attn_amax.q && !ds_work), "gamer_attn_bwd_split: q_span is built for the three-product recompute form "
                    "(arm gamer_attn_split_amax, ds_work = NULL)");
    rc = (nq / nkv == 1)
        ? launch_bwd_s<1>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, delta_ready, ds_work, st)
        : launch_bwd_s<2>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, dk, lddk, dv, lddv, ro, delta_ready, ds_work, st);
    t_amax_out = nullptr;
    t_attn_amax = AttnAmax{nullptr, nullptr, nullptr, nullptr};
    return rc;
}
