// Multi-behaviour flash attention, three-product fp16 form (H2, csrc/attention_split.hip), with the K and V rows of a
// (sequence, kv head) RESIDENT in LDS: replaces ref:SeqRec/models/generative/Qwen3Multi/model.py:75-150 (+ masks :573-630,
// :691-741, HF sdpa_attention_forward) in the Engine(matmul="split3") train step, same semantics and entry points as
// attention_split.hip (predicate masks, "empty" rows uniform over all S keys, lazily rescaled online softmax, row order of the
// cross attention, dropout mask function, operand maxima / sinks).
//
// Why (round 4 measurements, docs/DESIGN_rounds1-4.md section 20): the tiled kernels re-stage the causal prefix of K / V for every 64-query
// tile (4.5 x per sequence at S = 505); per 32 x 32 tile and wave that is ~215 of ~350 vector instructions (address arithmetic of
// the loads, the cut, the LDS stores) next to 24 MFMAs, plus one workgroup barrier per key tile - 49 % of the wave cycles parked
// at waits / barriers, 17 % MFMA-busy.  Here one workgroup of eight waves owns a (sequence, kv head) pair:
//   * keys are processed in BLOCKS of 256: the block's K and V rows are cut ONCE into their two fp16 piece images
//     (4 x 32 KB of LDS, one workgroup per CU) and every query row tile of both query heads runs against them;
//   * inside a block a wave walks its own 32-query row tiles with NO barrier and no global load in the key loop: K fragments by
//     row reads, V^T fragments by transposing reads of the resident images, key levels / dropout words from LDS;
//   * a row tile that needs keys of more than one block (S = 505: the upper half of the queries) carries its softmax state
//     through the output buffers: the block-0 pass stores O / l and m + log2 l (the state of an attention over the keys seen so
//     far, in final form up to constants), the next pass reloads it as (m_ref = that LSE, l = 1) - the same wave, the same
//     lanes, so program order makes it visible; the last pass writes the final o and lse;
//   * row tiles are dealt to the waves of a query head boustrophedon (causal ramp: equal work per wave in every block).
// Sessions' key spans (SPAN), uniform_len of the evaluation re-run and the six-product bf16 form stay on attention_split.hip.
#include "attention_split_common.h"

namespace gamer {

#ifndef RES_WARM
#define RES_WARM 0        // 1: L2 warm-up loads of the block a workgroup stages next.  Measured SLOWER (batch 1024: forward 1.33 vs 1.28 ms, backward
                          // 4.14 vs 4.05): vmcnt counts in order, so the next wait for a real load also waits for the warm-up loads
#endif
#ifndef RES_TOUCH_Q
#define RES_TOUCH_Q 0     // 1: L2 warm-up loads of the next row tile q rows and row data in the forward (slower as well: 1.28 -> 1.33 ms)
#endif
#ifndef RES_PREFETCH_Q
#define RES_PREFETCH_Q 0  // 1: the next row tile q row is requested into registers at the start of the current tile key loop.  Measured SLOWER (1.31 -> 1.48 ms:
                          // 32 more live registers in a 256-register kernel spill 14-27 of them in the loop)
#endif
#ifndef RES_STAMP
#define RES_STAMP 0       // diagnostic builds (tools/stamp_attn_res.py): every wave of the forward kernel accumulates shader-clock cycles per phase
#endif                    // into g_res_stamp[(8 * blockIdx.x + wave) * 8 + phase] (set by gamer_debug_res_stamp); never in the shipped library
#if RES_STAMP
__device__ unsigned long long* g_res_stamp = nullptr;
#define RES_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         ph_[i] += (t_ - tprev_); tprev_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define RES_STAMP_ARGS , unsigned long long (&ph_)[8], unsigned long long& tprev_
#define RES_STAMP_PASS , ph_, tprev_
#else
#define RES_MARK(i) do { } while (0)
#define RES_STAMP_ARGS
#define RES_STAMP_PASS
#endif

#ifndef RES_KEEP
#define RES_KEEP 1        // attn_bwd_dkv_r2_kernel: a wave keeps the key tile it swept last across the stage boundary (0: round-5 order)
#endif

constexpr int RB_KEYS = 256;                 // keys of one resident block
constexpr int RB_TILES = RB_KEYS / 32;
constexpr int RES_THREADS = 512;
constexpr int RIMG = RB_KEYS * 64;           // 16-bit elements of one piece image of a block ([256][64], swizzled like a tile image)
constexpr int RES_MAX_TILES = 64;            // 32-query row tiles per head the kernels are built for (S <= 2048)

struct ResSmem {
    bf16_t K[2][RIMG];
    bf16_t V[2][RIMG];
    int32_t kl[RB_KEYS];                     // key levels (INT_MAX past the end of the sequence)
    uint32_t kw[RB_KEYS];                    // dropout key words
    int32_t klmax[RB_TILES];                 // largest key level per 32-key tile
    uint32_t amax_word;
    int32_t ctr;                             // the block's row-tile queue: next item (waves take items with an LDS atomic)
    int32_t pad_[6];
    int32_t ntot[2 * RES_MAX_TILES];         // key tiles every (head of the pair, row tile) needs, written by the block-0 pass
};
// the row tile a wave will process after the current one: its q rows (and row data) are requested into the L2 while the current
// tile is multiplied - a row tile's prologue is otherwise two dependent HBM round trips with nothing to overlap them
struct ResNext { bool valid; int head, t, perm; };
__device__ __forceinline__ int res_grab_ctr(int32_t* ctr, int lane) {
    int v = 0;
    if (lane == 0) v = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int res_grab(ResSmem& sm, int lane) { return res_grab_ctr(&sm.ctr, lane); }

// Cut and store the K / V rows j0 .. j0 + 255 of one (sequence, kv head) and their metadata.  Every thread: 8 (row, quad) items per
// tensor, four loads of each tensor in flight.
template <bool DROP>
__device__ __forceinline__ void res_stage_block(ResSmem& sm, const float* __restrict__ kbase, int ldk, const float* __restrict__ vbase,
                                                int ldv, const int32_t* __restrict__ klb, int j0, int S, int tid, const H2Scales& sc,
                                                const AttnDropout& rng) {
    // all sixteen loads of a thread are requested before the first cut waits for one (no row-tile state is live here: 64 registers)
    float4 rk[8], rv[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int f = tid + RES_THREADS * p;
        const int j = min(j0 + (f >> 4), S - 1);
        rk[p] = *reinterpret_cast<const float4*>(kbase + (int64_t)j * ldk + ((f & 15) << 2));
        rv[p] = *reinterpret_cast<const float4*>(vbase + (int64_t)j * ldv + ((f & 15) << 2));
    }
    int klv = INT_BIG_A;
    if (tid < RB_KEYS && j0 + tid < S) klv = klb[j0 + tid];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int f = tid + RES_THREADS * p;
        const int row = f >> 4;
        const bool ok = j0 + row < S;
        const int off = sl_off(row, (f & 15) << 2);
        uint32_t a0, a1, b0, b1;
        cut2h_quad(ok ? rk[p].x : 0.f, ok ? rk[p].y : 0.f, ok ? rk[p].z : 0.f, ok ? rk[p].w : 0.f, sc.k, a0, a1, b0, b1);
        *reinterpret_cast<uint2*>(sm.K[0] + off) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(sm.K[1] + off) = make_uint2(a1, b1);
        cut2h_quad(ok ? rv[p].x : 0.f, ok ? rv[p].y : 0.f, ok ? rv[p].z : 0.f, ok ? rv[p].w : 0.f, sc.v, a0, a1, b0, b1);
        *reinterpret_cast<uint2*>(sm.V[0] + off) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(sm.V[1] + off) = make_uint2(a1, b1);
    }
    if (tid < RB_KEYS) {
        const int v = klv;
        sm.kl[tid] = v;
        int m = v;
#pragma unroll
        for (int o2 = 16; o2 > 0; o2 >>= 1) m = max(m, __shfl_xor(m, o2, 64));
        if ((tid & 31) == 0) sm.klmax[tid >> 5] = m;
    } else if (DROP) {
        sm.kw[tid - RB_KEYS] = rng.key_word((uint32_t)(j0 + tid - RB_KEYS));
    }
}

// st += K(tile) q^T: 4 k-steps x 3 piece products; the fragments of k-step s + 1 are requested before the products of k-step s
__device__ __forceinline__ f32x16 res_qk_tile(const ResSmem& sm, int tl, const SlOffsets& lo, const bf16x8 (&qf)[2][4], f32x16 st) {
    const bf16_t* k0 = sm.K[0] + tl * SIMG;
    const bf16_t* k1 = sm.K[1] + tl * SIMG;
    bf16x8 kf[2][2];
    kf[0][0] = *reinterpret_cast<const bf16x8*>(k0 + lo.row[0]);
    kf[0][1] = *reinterpret_cast<const bf16x8*>(k1 + lo.row[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
            kf[(s + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(k0 + lo.row[s + 1]);
            kf[(s + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(k1 + lo.row[s + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // smallest piece products first: (k1 q0), (k0 q1), (k0 q0)
        st = mfma_piece<true>(kf[s & 1][1], qf[0][s], st);
        st = mfma_piece<true>(kf[s & 1][0], qf[1][s], st);
        st = mfma_piece<true>(kf[s & 1][0], qf[0][s], st);
        __builtin_amdgcn_sched_barrier(0);
    }
    return st;
}

// One 32-query row tile of one query head against the resident key block kb.  SPAN: per-query key spans (Qwen3SessionMulti,
// attention_common.h: QuerySpan) - the row's causal limit becomes its span end and a hole may be cut out of its keys.
template <int G, bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
res_fwd_rowtile(ResSmem& sm, const float* __restrict__ q, int ldq, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                int S, int nq, float scale, const AttnDropout& rng, float* o, float* lse, const RowOrder& ro, const int b,
                const int head, const int t, const int kb, const bool want_amax, const H2Scales& sc, const int perm_cur,
                const ResNext& nx, float4 (&qraw)[8], const bool q_ready RES_STAMP_ARGS) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int n_all = (S + 31) >> 5;
    const int slot = t * 32 + r;                        // sorted slot of this lane's query row
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iqc = ORD ? perm_cur : slotc;
    const int iq = valid_q ? iqc : S;                   // position in the sequence (S: beyond every key)
    const int64_t tok = (int64_t)b * S + iqc;

    // every global load of the prologue is requested before anything waits for one: the q row, the row's data, the carried state
    // (in source order they came as three dependent round trips: row data -> wave reductions -> q -> state)
    // (qraw: this tile's q row, requested by the PREVIOUS tile of this wave right after it had cut its own - RES_PREFETCH_Q - or here)
    if (!q_ready) {
        const float* qrow = q + tok * ldq + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qraw[2 * s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
            qraw[2 * s + 1] = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    float* orow = o + tok * (int64_t)nq * 64 + head * 64;
    float* lse_p = lse + ((int64_t)b * nq + head) * S + iqc;
    f32x4v carry[8];
    float carry_l = -INFINITY;
    if (kb > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            carry[i] = f32x4v{0.f, 0.f, 0.f, 0.f};         // (possibly stored by another wave of this workgroup: read past the L1)
            if (valid_q) carry[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(orow + 32 * (i >> 2) + 8 * (i & 3) + 4 * h));
        }
        if (valid_q) carry_l = __builtin_nontemporal_load(lse_p);
    }
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq, valid_q, S);     // (no spans: hi = the row's position)
    const int wave_ql_min = wave_min_i32_dpp(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32_dpp(normal ? sp.hi : INT_BIG_A);  // every key up to here passes every row's causal limit
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;               // no normal row: no scores needed at all
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(normal ? sp.hole_hi : 0) : 0;
    int wave_q_hi;
    bool wave_has_empty;
    if (ORD) {
        wave_q_hi = ro.tile_maxpos[(int64_t)b * n_all + t];
        wave_has_empty = (ro.tile_kind[(int64_t)b * n_all + t] & 2) != 0;
    } else {
        wave_q_hi = wave_all_empty ? -1 : min(S - 1, t * 32 + 31);
        wave_has_empty = __any(my_empty ? 1 : 0) != 0;
    }
    // key tiles this row tile needs: all of them with an empty row (uniform over all S keys), else the causal prefix
    const int n_total = wave_has_empty ? n_all : (wave_q_hi < 0 ? 0 : (wave_q_hi >> 5) + 1);
    const int jt_lo = kb * RB_TILES;
    const int jt_hi = min(n_total, jt_lo + RB_TILES);
    const bool first = kb == 0, last = jt_lo + RB_TILES >= n_total;
    if (first && lane == 0) sm.ntot[(head % G) * RES_MAX_TILES + t] = n_total;     // (later blocks skip finished tiles without loading anything)

    // scores are kept in the log2 domain: q is pre-scaled by scale*log2(e), p = exp2(s - m_ref)
    const float qs = scale * 1.4426950408889634f;
    bf16x8 qf[2][4];                   // piece x k-step: lane (r, h) holds d = 16 s + 8 h .. + 7 of its query row
    {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = qraw[2 * s];
            float4 b4 = qraw[2 * s + 1];
            a4.x *= qs; a4.y *= qs; a4.z *= qs; a4.w *= qs; b4.x *= qs; b4.y *= qs; b4.z *= qs; b4.w *= qs;
            bf16x8 pq[3];
            cut8_t<true>(a4, b4, sc.q, pq);
            qf[0][s] = pq[0]; qf[1][s] = pq[1];
        }
    }
    if (RES_PREFETCH_Q && nx.valid) {
        // the NEXT row tile's q row into the registers this tile's raw q has just left: in flight during this tile's key loop
        const int64_t ntok = (int64_t)b * S + (ORD ? nx.perm : min(nx.t * 32 + r, S - 1));
        const float* qrow = q + ntok * ldq + nx.head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qraw[2 * s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
            qraw[2 * s + 1] = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
        }
    }
    const float invS = 1.f / (float)S;
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;

    float m_ref = 0.f, l_run = 0.f;
    f32x16 oacc[2];
    if (first) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    } else {
        // the state the previous block's pass left in o / lse: O / l (times the powers of two of the final form) and m + log2 l
        const float up = H2Scales::P * sc.v;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4v t4 = carry[4 * dh + g4];
                oacc[dh][4 * g4 + 0] = t4[0] * up; oacc[dh][4 * g4 + 1] = t4[1] * up;
                oacc[dh][4 * g4 + 2] = t4[2] * up; oacc[dh][4 * g4 + 3] = t4[3] * up;
            }
        const float L = normal ? carry_l : -INFINITY;
        const bool had = L > -INFINITY;                 // (a row whose keys so far were all blocked: nothing accumulated)
        m_ref = had ? L : 0.f;
        l_run = had ? 1.f : 0.f;
    }
    RES_MARK(1);                                           // 1: row-tile prologue (row data, q load + cut, carried state)

    uint32_t tq = 0, tm = 0;                               // (destinations of the next tile's L2 warm-up loads: never read)
#pragma unroll 1
    for (int jt = jt_lo; jt < jt_hi; ++jt) {
        const int j0 = jt * 32, tl = jt - jt_lo;
        if (RES_TOUCH_Q && jt == jt_lo && nx.valid) {
            // one 128-byte line per lane: lane (r, h) takes half h of row r of the next tile's q rows; its row data by lane half
            const int64_t ntok = (int64_t)b * S + (ORD ? nx.perm : min(nx.t * 32 + r, S - 1));
            tq = *reinterpret_cast<const uint32_t*>(q + ntok * ldq + nx.head * 64 + 32 * h);
            tm = *reinterpret_cast<const uint32_t*>(((h == 1 && ql) ? ql : row_empty) + ntok);
        }
        const bool beyond = j0 > wave_q_hi;                // every key of the tile is in every normal row's future
        if (beyond && !wave_has_empty) continue;
        f32x16 st;
        if (!wave_all_empty && !beyond) {
            const float init = -m_ref * (sc.q * sc.k);     // (the products carry the operands' scales)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = init;
            st = res_qk_tile(sm, tl, lo, qf, st);
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] *= sc.inv_qk;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = 0.f;      // (beyond: the mask below blocks every normal row)
        }
        const bf16_t* v0 = sm.V[0] + tl * SIMG;
        const bf16_t* v1 = sm.V[1] + tl * SIMG;
        const int32_t* klt = sm.kl + tl * 32;
        const int32_t* kwt = reinterpret_cast<const int32_t*>(sm.kw) + tl * 32;
        // O^T[d][query] += sum_key V[key][d] * P[query][key]   (rows of V past the sequence end are zero)
        auto pv_tile = [&]<bool ONE_PIECE>() {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf[3];
                cut8_regs_t<true>(st, 8 * s2, H2Scales::P, pf);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    bf16x8 vf[3];
                    vf[0] = read_tr8(v0, lo, 16 * s2, db);
                    vf[1] = read_tr8(v1, lo, 16 * s2, db);
                    vf[2] = vf[1];
                    oacc[db] = mfma_pieces<true, ONE_PIECE>(vf, pf, oacc[db]);
                }
            }
        };
        auto softmax_tile = [&]<bool MASK, bool EMPTYSEL>() {
            if (MASK) {
                int klv[16];
                read_key_quads(klt, h, klv);
                const int t_pos = sp.hi - j0 - 4 * h;        // key (reg&3)+8*(reg>>2) of the tile is <= hi (= iq without spans)
                const int t_lo = sp.hole_lo - j0 - 4 * h, t_hi = sp.hole_hi - j0 - 4 * h;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int ko = (reg & 3) + 8 * (reg >> 2);
                    bool allowed = (ko <= t_pos) & (klv[reg] < my_ql);
                    if (SPAN) allowed = allowed & !((ko >= t_lo) & (ko < t_hi));
                    st[reg] = allowed ? st[reg] : -INFINITY;
                }
            }
            float mloc = st[0];
#pragma unroll
            for (int reg = 1; reg < 16; ++reg) mloc = fmaxf(mloc, st[reg]);
            mloc = xor32_max(mloc);
            // (P is cut into fp16 pieces of P * 2^13 - it must stay below 2^3, so the reference follows the maximum closely)
            bool need = (l_run == 0.f) ? (mloc > -INFINITY) : (mloc > 2.f);
            if (EMPTYSEL) need = need && !my_empty;
            if (__any(need ? 1 : 0)) {
                const float d = need ? mloc : 0.f;
                const float alpha = (l_run == 0.f) ? 1.f : __builtin_amdgcn_exp2f(-d);
                m_ref += d;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; st[i] -= d; }
            }
            float rowsum = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float pe = __builtin_amdgcn_exp2f(st[reg]);            // masked -> exp2(-inf) = 0
                if (EMPTYSEL) pe = my_empty ? 1.f : pe;
                rowsum += pe;
                st[reg] = pe;
            }
            rowsum = xor32_sum(rowsum);
            l_run += rowsum;
            if (DROP) {
                int kwv[16];
                read_key_quads(kwt, h, kwv);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) st[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? st[reg] : 0.f;
            }
            pv_tile.template operator()<false>();
        };
        if (wave_all_empty) {
            // every row of the wave is an empty row: P = keep, no scores, no softmax state
            int kwv[16];
            if (DROP) read_key_quads(kwt, h, kwv);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const bool on = DROP ? rng.keep(aw, (uint32_t)kwv[reg]) : true;
                st[reg] = on ? 1.f : 0.f;
            }
            pv_tile.template operator()<true>();
        } else if (wave_has_empty) {
            softmax_tile.template operator()<true, true>();
        } else {
            const int klmax = __builtin_amdgcn_readfirstlane(sm.klmax[tl]);
            const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                   (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
            if (free_tile) softmax_tile.template operator()<false, false>();
            else softmax_tile.template operator()<true, false>();
        }
#if RES_STAMP
        ph_[6] += 1;                                       // 6: key tiles
#endif
    }
    RES_MARK(2);                                           // 2: key loop

    if (last) {
        float omax = 0.f;
        if (valid_q) {
            float linv = my_empty ? invS : (l_run > 0.f ? 1.f / l_run : 0.f);
            if (DROP) linv *= rng.scale;
            linv *= H2Scales::INV_P * sc.inv_v;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float4 t4;
                    t4.x = oacc[dh][4 * g4 + 0] * linv; t4.y = oacc[dh][4 * g4 + 1] * linv;
                    t4.z = oacc[dh][4 * g4 + 2] * linv; t4.w = oacc[dh][4 * g4 + 3] * linv;
                    *reinterpret_cast<float4*>(orow + 32 * dh + 8 * g4 + 4 * h) = t4;
                    omax = fmaxf(fmaxf(fmaxf(omax, fabsf(t4.x)), fabsf(t4.y)), fmaxf(fabsf(t4.z), fabsf(t4.w)));
                }
            // natural-log LSE of the scaled scores (what the backward kernels consume)
            if (h == 0) *lse_p = my_empty ? 0.f : (m_ref + __log2f(l_run)) * 0.6931471805599453f;
        }
        if (want_amax) {                                                    // (gamer_amax_sink: max |o| for the o_proj GEMM's scale)
            uint32_t mw = __float_as_uint(omax);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
            if (lane == 0 && mw) atomicMax(&sm.amax_word, mw);
        }
    } else if (valid_q) {
        // carry: the attention over the keys seen so far, normalised (an empty row: its plain sum)
        float f = my_empty ? 1.f : (l_run > 0.f ? 1.f / l_run : 0.f);
        f *= H2Scales::INV_P * sc.inv_v;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = oacc[dh][4 * g4 + 0] * f; t4.y = oacc[dh][4 * g4 + 1] * f;
                t4.z = oacc[dh][4 * g4 + 2] * f; t4.w = oacc[dh][4 * g4 + 3] * f;
                *reinterpret_cast<float4*>(orow + 32 * dh + 8 * g4 + 4 * h) = t4;
            }
        if (h == 0) *lse_p = my_empty ? 0.f : m_ref + __log2f(l_run);       // (log2 domain; -inf: nothing accumulated yet)
    }
    asm volatile("" :: "v"(tq), "v"(tm));
    RES_MARK(3);                                           // 3: epilogue (final or carried state)
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(RES_THREADS, 1)
attn_fwd_r_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                  const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                  int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* o, float* lse,
                  const RowOrder ro, uint32_t* __restrict__ amax_out, const AttnAmax am, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char res_raw[];
    ResSmem& sm = *reinterpret_cast<ResSmem*>(res_raw);
    const H2Scales sc(am.q, am.k, am.v, nullptr, scale * 1.4426950408889634f);
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x;
    const int n_all = (S + 31) >> 5;
    const int nblk = (S + RB_KEYS - 1) / RB_KEYS;
    // unit of a workgroup: a (sequence, kv head) pair with its G query heads, or - split, when the pairs do not fill the CUs evenly
    // (per-GPU batch 128: 384 pairs on 256 CUs) - ONE query head of it (K / V are then staged once per head)
    const int hpu = split ? 1 : G, upp = G / hpu;
    const int n_items = hpu * n_all;                        // (head of the unit, row tile) items of one key block
    const int n_pairs = nbatch * nkv * upp;                 // (units)
    const int lane = tid & 63, r = lane & 31;
    if (tid == 0) sm.amax_word = 0;
#if RES_STAMP
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const int b = (pair / upp) / nkv, kvh = (pair / upp) % nkv, h0 = (pair % upp) * hpu;
        const float* kbase = k + (int64_t)b * S * ldk + kvh * 64;
        const float* vbase = v + (int64_t)b * S * ldv + kvh * 64;
        const int32_t* klb = kl + (int64_t)b * S;
#pragma unroll 1
        for (int kb = 0; kb < nblk; ++kb) {
            RES_MARK(5);                                    // 5: loop control between row tiles
            __syncthreads();                                // every wave is done with the previous block's images
            RES_MARK(4);                                    // 4: waiting for the slowest wave of the block
            if (tid == 0) sm.ctr = 0;
            res_stage_block<DROP>(sm, kbase, ldk, vbase, ldv, klb, kb * RB_KEYS, S, tid, sc, rng);
            __syncthreads();
            RES_MARK(0);                                    // 0: staging of the block (loads, cut, LDS stores, barrier)
            // warm the L2 with the K / V rows of the block this workgroup stages NEXT (one 128-byte line per thread and tensor): all
            // workgroups run in step, so unprefetched the staging is an HBM burst every CU waits for
            uint32_t tk0 = 0, tk1 = 0;
            {
                int npair = pair, nkb = kb + 1;
                if (nkb == nblk) { nkb = 0; npair += gridDim.x; }
                if (RES_WARM && npair < n_pairs) {
                    const int j = min(nkb * RB_KEYS + (tid >> 1), S - 1);
                    const int64_t nrow = (int64_t)((npair / upp) / nkv) * S + j;
                    tk0 = *reinterpret_cast<const uint32_t*>(k + nrow * ldk + ((npair / upp) % nkv) * 64 + (tid & 1) * 32);
                    tk1 = *reinterpret_cast<const uint32_t*>(v + nrow * ldv + ((npair / upp) % nkv) * 64 + (tid & 1) * 32);
                }
            }
            // the block's row tiles: late (long) tiles first, taken from a queue - whichever wave is free takes the next one (the
            // carried state of a tile travels through global memory, workgroup barriers in between: any wave may continue it)
            int cur = res_grab(sm, lane);
            int perm_cur = 0;
            float4 qraw[8];
            bool q_ready = false;
            if (ORD && cur < n_items) perm_cur = ro.perm[(int64_t)b * S + min((n_all - 1 - cur / hpu) * 32 + r, S - 1)];
#pragma unroll 1
            while (cur < n_items) {
                const int nxt = res_grab(sm, lane);
                const int hg = h0 + cur % hpu, t = n_all - 1 - cur / hpu;
                ResNext nx;
                nx.t = n_all - 1 - nxt / hpu;
                nx.head = kvh * G + h0 + nxt % hpu;
                nx.valid = nxt < n_items && (kb == 0 || sm.ntot[(h0 + nxt % hpu) * RES_MAX_TILES + nx.t] > kb * RB_TILES);
                nx.perm = 0;
                if (ORD && nxt < n_items) nx.perm = ro.perm[(int64_t)b * S + min(nx.t * 32 + r, S - 1)];
                if (kb == 0 || sm.ntot[hg * RES_MAX_TILES + t] > kb * RB_TILES) {
                    res_fwd_rowtile<G, DROP, ORD, SPAN>(sm, q, ldq, ql, row_empty, S, nq, scale, rng, o, lse, ro, b, kvh * G + hg, t, kb,
                                                  amax_out != nullptr, sc, perm_cur, nx, qraw, q_ready RES_STAMP_PASS);
                    q_ready = RES_PREFETCH_Q && nx.valid;       // (nx.valid: the next item is a tile this block's pass processes)
                } else {
                    q_ready = false;
                }
                cur = nxt;
                perm_cur = nx.perm;
            }
            asm volatile("" :: "v"(tk0), "v"(tk1));
        }
    }
    if (amax_out) {
        __syncthreads();
        if (tid == 0 && sm.amax_word) amax_publish(sm.amax_word, amax_out, blockIdx.x);
    }
#if RES_STAMP
    RES_MARK(5);
    if (g_res_stamp && (tid & 63) == 0) {
        unsigned long long* rec = g_res_stamp + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) rec[i_] += ph_[i_];
    }
#endif
}


// =============================================================================================
// backward: dQ (attention_split.hip: attn_bwd_dq_s_tile) against the resident K / V block.  Per key tile: dP^T = V dO^T and
// S^T = K q^T (query on the lane, the dO pieces in registers like q's), dS = P (mult dP - delta), dQ^T += K^T dS^T with the K^T
// fragments by transposing reads of the resident image; dO and dS are scaled per query row.  A row tile that needs more than one
// key block carries its raw accumulators through dq (fp32, exact: the sum continues as if the key loop had not been cut).
// =============================================================================================
template <int G, bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
res_dq_rowtile(ResSmem& sm, const float* __restrict__ q, int ldq, const float* __restrict__ o, const float* __restrict__ d_o,
               const float* __restrict__ lse, float* delta, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
               int S, int nq, float scale, const AttnDropout& rng, float* dq, int lddq, const RowOrder& ro, const int b,
               const int head, const int t, const int kb, const int delta_ready, const H2Scales& sc) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int n_all = (S + 31) >> 5;
    const int slot = t * 32 + r;
    const bool valid_q = slot < S;
    const int slotc = valid_q ? slot : S - 1;
    const int iqc = ORD ? ro.perm[(int64_t)b * S + slotc] : slotc;
    const int iq = valid_q ? iqc : S;
    const int64_t tok = (int64_t)b * S + iqc;

    // every global load of the prologue is requested before anything waits for one (see res_fwd_rowtile)
    const bool need_delta = !delta_ready && kb == 0;
    float* delta_p = delta + ((int64_t)b * nq + head) * S + iqc;
    float* dqrow = dq + tok * lddq + head * 64;
    float4 qraw[8], draw[8];
    {
        const float* qrow = q + tok * ldq + head * 64 + 8 * h;
        const float* drow = d_o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qraw[2 * s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
            qraw[2 * s + 1] = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
            draw[2 * s] = *reinterpret_cast<const float4*>(drow + 16 * s);
            draw[2 * s + 1] = *reinterpret_cast<const float4*>(drow + 16 * s + 4);
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    const bool my_empty = valid_q && row_empty[tok] != 0;
    const float my_lse = lse[((int64_t)b * nq + head) * S + iqc];
    float my_delta = 0.f;
    if (!need_delta) my_delta = __builtin_nontemporal_load(delta_p);      // (possibly stored by another wave of this workgroup: past the L1)
    f32x4v carry[8];
    if (kb > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            carry[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (valid_q) carry[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dqrow + 32 * (i >> 2) + 8 * (i & 3) + 4 * h));
        }
    }
    const bool normal = valid_q && !my_empty;
    const QuerySpan sp = QuerySpan::load<SPAN>(ro.span, tok, iq, valid_q, S);     // (no spans: hi = the row's position)
    const int wave_ql_min = wave_min_i32_dpp(normal ? my_ql : INT_BIG_A);
    const int wave_q_lo = wave_min_i32_dpp(normal ? sp.hi : INT_BIG_A);
    const bool wave_all_empty = wave_q_lo == INT_BIG_A;
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(normal ? sp.hole_lo : INT_BIG_A) : INT_BIG_A;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(normal ? sp.hole_hi : 0) : 0;
    int wave_q_hi;
    bool wave_has_empty;
    if (ORD) {
        wave_q_hi = ro.tile_maxpos[(int64_t)b * n_all + t];
        wave_has_empty = (ro.tile_kind[(int64_t)b * n_all + t] & 2) != 0;
    } else {
        wave_q_hi = wave_all_empty ? -1 : min(S - 1, t * 32 + 31);
        wave_has_empty = __any(my_empty ? 1 : 0) != 0;
    }
    const int n_total = wave_has_empty ? n_all : (wave_q_hi < 0 ? 0 : (wave_q_hi >> 5) + 1);
    const int jt_lo = kb * RB_TILES;
    if (kb > 0 && jt_lo >= n_total) return;
    const int jt_hi = min(n_total, jt_lo + RB_TILES);
    const bool first = kb == 0, last = jt_lo + RB_TILES >= n_total;
    if (first && lane == 0) sm.ntot[(head % G) * RES_MAX_TILES + t] = n_total;     // (later blocks skip finished tiles without loading anything)

    // log2 domain: q is pre-scaled by scale*log2(e) and the score accumulators start at -lse*log2(e), so p = exp2(accumulator)
    const float qs = scale * 1.4426950408889634f;
    bf16x8 qf[2][4], dof[2][4];           // piece x k-step of the pre-scaled q row and of the dO row (lane = query)
    float my_sdo = sc.d_o, my_inv_do = sc.inv_do, my_sds = sc.ds, my_inv_ds = sc.inv_ds;
    {
        const float* orow = o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
        float4 du[4], dw[4];
        float rowmax = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = qraw[2 * s];
            float4 b4 = qraw[2 * s + 1];
            a4.x *= qs; a4.y *= qs; a4.z *= qs; a4.w *= qs; b4.x *= qs; b4.y *= qs; b4.z *= qs; b4.w *= qs;
            bf16x8 pq[3];
            cut8_t<true>(a4, b4, sc.q, pq);
            qf[0][s] = pq[0]; qf[1][s] = pq[1];
            float4 u4 = draw[2 * s];
            float4 w4 = draw[2 * s + 1];
            if (need_delta) {
                const float4 o4 = *reinterpret_cast<const float4*>(orow + 16 * s);
                const float4 p4 = *reinterpret_cast<const float4*>(orow + 16 * s + 4);
                my_delta += u4.x * o4.x + u4.y * o4.y + u4.z * o4.z + u4.w * o4.w + w4.x * p4.x + w4.y * p4.y + w4.z * p4.z + w4.w * p4.w;
            }
            if (!valid_q) { u4 = make_float4(0.f, 0.f, 0.f, 0.f); w4 = u4; }
            du[s] = u4; dw[s] = w4;
            rowmax = fmaxf(fmaxf(fmaxf(rowmax, fabsf(u4.x)), fabsf(u4.y)), fmaxf(fabsf(u4.z), fabsf(u4.w)));
            rowmax = fmaxf(fmaxf(fmaxf(rowmax, fabsf(w4.x)), fabsf(w4.y)), fmaxf(fabsf(w4.z), fabsf(w4.w)));
        }
        rowmax = xor32_max(rowmax);                                    // the row's other 32 columns
        if (rowmax > 0.f) {                                          // (an all-zero row keeps the tensor's scales: its pieces are 0)
            scale_from_amax(__float_as_uint(rowmax), my_sdo, my_inv_do);
            scale_from_amax(__float_as_uint(sc.ds_coef * rowmax), my_sds, my_inv_ds);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 dpc[3];
            cut8_t<true>(du[s], dw[s], my_sdo, dpc);
            dof[0][s] = dpc[0]; dof[1][s] = dpc[1];
        }
        if (need_delta) {
            my_delta = xor32_sum(my_delta);
            if (valid_q && h == 0) *delta_p = my_delta;
        }
    }
    const float neg_lse2 = -my_lse * 1.4426950408889634f;
    const float invS = 1.f / (float)S;
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;
    const float sd = rng.scale;
    const float neg_delta = -my_delta;

    f32x16 dqacc[2];
    if (first) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }
    } else {
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4v t4 = carry[4 * dh + g4];
                dqacc[dh][4 * g4 + 0] = t4[0]; dqacc[dh][4 * g4 + 1] = t4[1]; dqacc[dh][4 * g4 + 2] = t4[2]; dqacc[dh][4 * g4 + 3] = t4[3];
            }
    }

#pragma unroll 1
    for (int jt = jt_lo; jt < jt_hi; ++jt) {
        const int j0 = jt * 32, tl = jt - jt_lo;
        const bool beyond = j0 > wave_q_hi;
        if (beyond && !wave_has_empty) continue;
        const bf16_t* k0 = sm.K[0] + tl * SIMG;
        const bf16_t* k1 = sm.K[1] + tl * SIMG;
        const bf16_t* v0 = sm.V[0] + tl * SIMG;
        const bf16_t* v1 = sm.V[1] + tl * SIMG;
        const int32_t* klt = sm.kl + tl * 32;
        const int32_t* kwt = reinterpret_cast<const int32_t*>(sm.kw) + tl * 32;
        f32x16 st, dp;
        const float st0 = !beyond ? neg_lse2 * (sc.q * sc.k) : neg_lse2;   // (the products carry the operands' scales)
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = st0; dp[i] = 0.f; }
        // dP^T[key][query] = sum_d V[key][d] dO[query][d]
        {
            bf16x8 vf[2][2];
            vf[0][0] = *reinterpret_cast<const bf16x8*>(v0 + lo.row[0]);
            vf[0][1] = *reinterpret_cast<const bf16x8*>(v1 + lo.row[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) {
                    vf[(s + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(v0 + lo.row[s + 1]);
                    vf[(s + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(v1 + lo.row[s + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                dp = mfma_piece<true>(vf[s & 1][1], dof[0][s], dp);
                dp = mfma_piece<true>(vf[s & 1][0], dof[1][s], dp);
                dp = mfma_piece<true>(vf[s & 1][0], dof[0][s], dp);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!beyond && !wave_all_empty) st = res_qk_tile(sm, tl, lo, qf, st);
        {
            const float cdp = sc.inv_v * my_inv_do;
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] *= cdp;
            if (!beyond) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i] *= sc.inv_qk;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // u = mult * dP - delta  (mult = keep / (1 - p))
        if (DROP) {
            int kwv[16];
            read_key_quads(kwt, h, kwv);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float tt = rng.keep(aw, (uint32_t)kwv[reg]) ? dp[reg] : 0.f;
                dp[reg] = fmaf(tt, sd, neg_delta);
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) dp[reg] += neg_delta;
        }
        // dS^T = p * u ; an empty row has p = 1 here and 1/S in the final scale
        auto ds_tile = [&]<bool MASK, bool EMPTYSEL>() {
            if (MASK) {
                int klv[16];
                read_key_quads(klt, h, klv);
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
            const int klmax = __builtin_amdgcn_readfirstlane(sm.klmax[tl]);
            const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                   (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
            if (free_tile) ds_tile.template operator()<false, false>();
            else ds_tile.template operator()<true, false>();
        }
        __builtin_amdgcn_sched_barrier(0);
        // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]   (rows of K past the sequence end are zero)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 df[3];
            cut8_regs_t<true>(st, 8 * s2, my_sds, df);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 kf[3];
                kf[0] = read_tr8(k0, lo, 16 * s2, db);
                kf[1] = read_tr8(k1, lo, 16 * s2, db);
                kf[2] = kf[1];
                dqacc[db] = mfma_pieces<true>(kf, df, dqacc[db]);
            }
        }
    }

    if (valid_q) {
        float fs = 1.f;
        if (last) fs = (my_empty ? scale * invS : scale) * (my_inv_ds * sc.inv_k);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 t4;
                t4.x = dqacc[dh][4 * g4 + 0] * fs; t4.y = dqacc[dh][4 * g4 + 1] * fs;
                t4.z = dqacc[dh][4 * g4 + 2] * fs; t4.w = dqacc[dh][4 * g4 + 3] * fs;
                *reinterpret_cast<float4*>(dqrow + 32 * dh + 8 * g4 + 4 * h) = t4;
            }
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(RES_THREADS, 1)
attn_bwd_dq_r_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                     const float* __restrict__ o, const float* __restrict__ d_o, const float* __restrict__ lse, float* delta,
                     const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                     int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dq, int lddq,
                     const RowOrder ro, const int delta_ready, const AttnAmax am, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char res_raw[];
    ResSmem& sm = *reinterpret_cast<ResSmem*>(res_raw);
    const H2Scales sc(am.q, am.k, am.v, am.d_o, scale * 1.4426950408889634f, 1.f / (1.f - p_drop));
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_all = (S + 31) >> 5;
    const int nblk = (S + RB_KEYS - 1) / RB_KEYS;
    const int hpu = split ? 1 : G, upp = G / hpu;           // (units: see attn_fwd_r_kernel)
    const int n_items = hpu * n_all;
    const int n_units = nbatch * nkv * upp;
#pragma unroll 1
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int b = (unit / upp) / nkv, kvh = (unit / upp) % nkv, h0 = (unit % upp) * hpu;
        const float* kbase = k + (int64_t)b * S * ldk + kvh * 64;
        const float* vbase = v + (int64_t)b * S * ldv + kvh * 64;
        const int32_t* klb = kl + (int64_t)b * S;
#pragma unroll 1
        for (int kb = 0; kb < nblk; ++kb) {
            __syncthreads();
            if (tid == 0) sm.ctr = 0;
            res_stage_block<DROP>(sm, kbase, ldk, vbase, ldv, klb, kb * RB_KEYS, S, tid, sc, rng);
            __syncthreads();
            // warm the L2 with the K / V rows of the block this workgroup stages next (see attn_fwd_r_kernel)
            uint32_t tk0 = 0, tk1 = 0;
            {
                int nunit = unit, nkb = kb + 1;
                if (nkb == nblk) { nkb = 0; nunit += gridDim.x; }
                if (RES_WARM && nunit < n_units) {
                    const int j = min(nkb * RB_KEYS + (tid >> 1), S - 1);
                    const int64_t nrow = (int64_t)((nunit / upp) / nkv) * S + j;
                    tk0 = *reinterpret_cast<const uint32_t*>(k + nrow * ldk + ((nunit / upp) % nkv) * 64 + (tid & 1) * 32);
                    tk1 = *reinterpret_cast<const uint32_t*>(v + nrow * ldv + ((nunit / upp) % nkv) * 64 + (tid & 1) * 32);
                }
            }
            // row tiles from the block's queue, late (long) tiles first
#pragma unroll 1
            for (int cur = res_grab(sm, lane); cur < n_items; cur = res_grab(sm, lane)) {
                const int hg = h0 + cur % hpu, t = n_all - 1 - cur / hpu;
                if (kb == 0 || sm.ntot[hg * RES_MAX_TILES + t] > kb * RB_TILES)
                    res_dq_rowtile<G, DROP, ORD, SPAN>(sm, q, ldq, o, d_o, lse, delta, ql, row_empty, S, nq, scale, rng, dq, lddq, ro, b,
                                                 kvh * G + hg, t, kb, delta_ready, sc);
            }
            asm volatile("" :: "v"(tk0), "v"(tk1));
        }
    }
}


// =============================================================================================
// backward: dK, dV - the mirror image of the forward: the Q and dO rows of ONE query head and a block of 256 (sorted) query slots
// are resident in LDS as fp16 piece images with their per-row scalars, a wave owns a 32-key tile (key on the lane, its K / V
// fragments in registers for the whole sweep, dK^T / dV^T in 64 accumulator registers) and walks the block's query tiles with no
// barrier.  Key tiles are taken from a queue (early keys - seen by the most queries - first).  A key tile's sums continue from
// stage to stage (the G heads of the kv head x the query blocks, in this fixed order) through dk / dv themselves, raw fp32, so the
// result does not depend on which wave took which item; the last stage scales and stores every key tile.
// =============================================================================================
struct DkvRSmem {
    bf16_t Q[2][RIMG];
    bf16_t dO[2][RIMG];
    float nlse2[RB_KEYS];        // -lse * log2(e) of the staged query rows
    float ndelta[RB_KEYS];       // -delta
    uint32_t aw[RB_KEYS];        // dropout row words
    int32_t ql[RB_KEYS];         // query levels (0 past the end of the sequence)
    int32_t empty[RB_KEYS];
    int32_t pos[RB_KEYS];        // position of the row in the sequence = its causal key limit (S past the end)
    int32_t t_qlmin[RB_TILES];   // per 32-row tile: smallest query level / position over its normal rows (INT_MAX: none),
    int32_t t_posmin[RB_TILES];
    int32_t t_maxpos[RB_TILES];  //                  largest position of a normal row (-1: none), has "empty" rows
    int32_t t_empty[RB_TILES];
    int32_t started[RES_MAX_TILES];      // key tile has a partial sum in dk / dv
    uint32_t amax_word;
    int32_t ctr;
    int32_t pad_[6];
};

// Cut and store the Q / dO rows of query slots s0 .. s0 + 255 of one head, their scalars and the per-tile summaries.
constexpr int DKV_R_THREADS = 256;            // four waves, one per SIMD: the whole 512-register file per wave (K / V fragments, four
                                              // accumulator tiles and the tile in flight do not fit 256 registers: 85-99 spilled)
template <bool DROP, bool ORD>
__device__ __forceinline__ void
res_stage_queries(DkvRSmem& sm, const float* __restrict__ q, int ldq, const float* __restrict__ d_o, const float* __restrict__ lse,
                  const float* delta, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty, const RowOrder& ro,
                  int b, int head, int s0, int S, int nq, int tid, const H2Scales& sc, const AttnDropout& rng) {
    const int32_t* pmap = ORD ? ro.perm + (int64_t)b * S : nullptr;
    float4 rq[8], rd[8];
    auto load8 = [&](int p0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = tid + DKV_R_THREADS * (p0 + p);
            const int sl = min(s0 + (f >> 4), S - 1);
            const int64_t tok = (int64_t)b * S + (ORD ? pmap[sl] : sl);
            rq[p] = *reinterpret_cast<const float4*>(q + tok * ldq + head * 64 + ((f & 15) << 2));
            rd[p] = *reinterpret_cast<const float4*>(d_o + tok * (int64_t)nq * 64 + head * 64 + ((f & 15) << 2));
        }
    };
    auto store8 = [&](int p0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = tid + DKV_R_THREADS * (p0 + p);
            const int row = f >> 4;
            const bool ok = s0 + row < S;
            const int off = sl_off(row, (f & 15) << 2);
            uint32_t a0, a1, b0, b1;
            cut2h_quad(ok ? rq[p].x : 0.f, ok ? rq[p].y : 0.f, ok ? rq[p].z : 0.f, ok ? rq[p].w : 0.f, sc.q, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.Q[0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.Q[1] + off) = make_uint2(a1, b1);
            cut2h_quad(ok ? rd[p].x : 0.f, ok ? rd[p].y : 0.f, ok ? rd[p].z : 0.f, ok ? rd[p].w : 0.f, sc.d_o, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.dO[0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.dO[1] + off) = make_uint2(a1, b1);
        }
    };
    load8(0);
    if (tid < RB_KEYS) {
        const int sl = s0 + tid;
        const bool in = sl < S;
        const int pos = in ? (ORD ? pmap[sl] : sl) : 0;
        const int64_t hrow = ((int64_t)b * nq + head) * S + pos;
        const int em = in ? row_empty[(int64_t)b * S + pos] : 0;
        const int qlv = in ? (ql ? ql[(int64_t)b * S + pos] : 1) : 0;
        sm.nlse2[tid] = in ? -lse[hrow] * 1.4426950408889634f : 0.f;
        sm.ndelta[tid] = in ? -delta[hrow] : 0.f;
        if (DROP) sm.aw[tid] = rng.row_word((uint32_t)hrow);
        sm.ql[tid] = qlv;
        sm.empty[tid] = em;
        sm.pos[tid] = in ? pos : S;
        const bool normal = in && em == 0;
        int qlmin = normal ? qlv : INT_BIG_A, posmin = normal ? pos : INT_BIG_A, maxpos = normal ? pos : -1, anye = em != 0 ? 1 : 0;
#pragma unroll
        for (int o2 = 16; o2 > 0; o2 >>= 1) {
            qlmin = min(qlmin, __shfl_xor(qlmin, o2, 64));
            posmin = min(posmin, __shfl_xor(posmin, o2, 64));
            maxpos = max(maxpos, __shfl_xor(maxpos, o2, 64));
            anye |= __shfl_xor(anye, o2, 64);
        }
        if ((tid & 31) == 0) {
            sm.t_qlmin[tid >> 5] = qlmin; sm.t_posmin[tid >> 5] = posmin; sm.t_maxpos[tid >> 5] = maxpos; sm.t_empty[tid >> 5] = anye;
        }
    }
    store8(0);
    load8(8);
    store8(8);
}

// One 32-key tile against the resident query block.
template <bool DROP>
__device__ __forceinline__ void
res_dkv_keytile(DkvRSmem& sm, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv, const int32_t* __restrict__ kl,
                int S, float scale, const AttnDropout& rng, float* dk, int lddk, float* dv, int lddv, const int b, const int kvh,
                const int kt, const int n_qt, const bool fresh, const bool final_stage, const bool want_amax, const H2Scales& sc) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const SlOffsets lo(lane);
    const int jk = kt * 32 + r;                          // this lane's key
    const bool valid_k = jk < S;
    const int64_t ktok = (int64_t)b * S + (valid_k ? jk : S - 1);

    bf16x8 kf[2][4], vf[2][4];            // piece x k-step: lane (r, h) holds d = 16 s + 8 h .. + 7 of its key's K / V row
    {
        const float* krow = k + ktok * ldk + kvh * 64 + 8 * h;
        const float* vrow = v + ktok * ldv + kvh * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = *reinterpret_cast<const float4*>(krow + 16 * s);
            float4 b4 = *reinterpret_cast<const float4*>(krow + 16 * s + 4);
            float4 c4 = *reinterpret_cast<const float4*>(vrow + 16 * s);
            float4 d4 = *reinterpret_cast<const float4*>(vrow + 16 * s + 4);
            if (!valid_k) { a4 = make_float4(0.f, 0.f, 0.f, 0.f); b4 = a4; c4 = a4; d4 = a4; }
            bf16x8 pc[3];
            cut8_t<true>(a4, b4, sc.k, pc);
            kf[0][s] = pc[0]; kf[1][s] = pc[1];
            cut8_t<true>(c4, d4, sc.v, pc);
            vf[0][s] = pc[0]; vf[1][s] = pc[1];
        }
    }
    const int my_kl = valid_k ? kl[ktok] : INT_BIG_A;
    const int wave_kl_max = wave_max_i32(my_kl);
    const int wave_k_lo = kt * 32, wave_k_hi = wave_k_lo + 31;
    const float invS = 1.f / (float)S;
    // (the products carry their operands' scales - folded into the constants that multiply them anyway)
    const float c2 = scale * 1.4426950408889634f * sc.inv_qk;
    const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
    const float cdp = sc.inv_v * sc.inv_do;              // dP = (dO' V'^T) cdp
    const float sd = rng.scale * cdp;
    float* dkrow = dk + ktok * lddk + kvh * 64;
    float* dvrow = dv + ktok * lddv + kvh * 64;

    f32x16 dkacc[2], dvacc[2];
    if (fresh) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }
    } else {
        // the raw sums an earlier stage left (possibly another wave: read past the L1)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dh + 8 * g4 + 4 * h;
                f32x4v a = {0.f, 0.f, 0.f, 0.f}, c = a;
                if (valid_k) {
                    a = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dkrow + d));
                    c = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dvrow + d));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { dkacc[dh][4 * g4 + e] = a[e]; dvacc[dh][4 * g4 + e] = c[e]; }
            }
    }

#pragma unroll 1
    for (int qt = 0; qt < n_qt; ++qt) {
        const bool tile_has_empty = __builtin_amdgcn_readfirstlane(sm.t_empty[qt]) != 0;
        const int posmin = __builtin_amdgcn_readfirstlane(sm.t_posmin[qt]);
        const int qlmin = __builtin_amdgcn_readfirstlane(sm.t_qlmin[qt]);
        const int maxpos = __builtin_amdgcn_readfirstlane(sm.t_maxpos[qt]);
        const bool tile_all_empty = tile_has_empty && posmin == INT_BIG_A;
        const bool before = maxpos < wave_k_lo;            // every normal query of the tile precedes this wave's keys
        if (before && !tile_has_empty) continue;
        const bf16_t* q0 = sm.Q[0] + qt * SIMG;
        const bf16_t* q1 = sm.Q[1] + qt * SIMG;
        const bf16_t* o0 = sm.dO[0] + qt * SIMG;
        const bf16_t* o1 = sm.dO[1] + qt * SIMG;
        const float* ndl_t = sm.ndelta + qt * 32;
        const float* nl_t = sm.nlse2 + qt * 32;
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
        // S[query][key] = sum_d Q[query][d] K[key][d] and dP[query][key] = sum_d dO[query][d] V[key][d]
        const bool need_s = !before && !tile_all_empty;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(o0 + lo.row[s]);
            const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(o1 + lo.row[s]);
            dp = mfma_piece<true>(c1, vf[0][s], dp);
            dp = mfma_piece<true>(c0, vf[1][s], dp);
            dp = mfma_piece<true>(c0, vf[0][s], dp);
            if (need_s) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(q0 + lo.row[s]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(q1 + lo.row[s]);
                st = mfma_piece<true>(a1, kf[0][s], st);
                st = mfma_piece<true>(a0, kf[1][s], st);
                st = mfma_piece<true>(a0, kf[0][s], st);
            }
        }
        // per element (query = register, key = lane):  P -> st (for dV, 1/(1-p) applied at the end), dS -> dp (for dK)
        // Rows past the end of the sequence have Q = dO = 0, lse = delta = 0: p = 1 but dS = 0 and dO = 0.
        auto elem_tile = [&]<bool MASK, bool EMPTYSEL, bool ALL_EMPTY>() {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qb = 8 * g4 + 4 * h;
                const float4 d4 = *reinterpret_cast<const float4*>(ndl_t + qb);
                const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
                float nl[4] = {0.f, 0.f, 0.f, 0.f};
                int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0}, emv[4] = {0, 0, 0, 0};
                uint32_t awv[4] = {0, 0, 0, 0};
                if (!ALL_EMPTY) {
                    const float4 l4 = *reinterpret_cast<const float4*>(nl_t + qb);
                    nl[0] = l4.x; nl[1] = l4.y; nl[2] = l4.z; nl[3] = l4.w;
                }
                if (MASK) {
                    const int4 q4 = *reinterpret_cast<const int4*>(sm.ql + qt * 32 + qb);
                    const int4 p4 = *reinterpret_cast<const int4*>(sm.pos + qt * 32 + qb);
                    qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
                    posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
                }
                if (EMPTYSEL) {
                    const int4 e4 = *reinterpret_cast<const int4*>(sm.empty + qt * 32 + qb);
                    emv[0] = e4.x; emv[1] = e4.y; emv[2] = e4.z; emv[3] = e4.w;
                }
                if (DROP) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(sm.aw + qt * 32 + qb);
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
                            const bool allowed = (jk <= posv[e]) & (my_kl < qlv[e]);
                            pe = allowed ? pe : 0.f;
                        }
                        if (EMPTYSEL) pe = (emv[e] != 0) ? invS : pe;
                    }
                    if (DROP) {
                        const bool keep = rng.keep(awv[e], bw);
                        const float tt = keep ? dp[reg] : 0.f;
                        dp[reg] = pe * fmaf(tt, sd, ndl[e]);         // dS[query][key]
                        st[reg] = keep ? pe : 0.f;                   // dropped P[query][key] * (1 - p)
                    } else {
                        dp[reg] = pe * fmaf(dp[reg], cdp, ndl[e]);
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
            const bool free_tile = (posmin >= wave_k_hi) && (wave_kl_max < qlmin);
            if (free_tile) elem_tile.template operator()<false, false, false>();
            else elem_tile.template operator()<true, false, false>();
        }
        // dV^T[d][key] += sum_query dO[query][d] Pd[query][key] ; dK^T[d][key] += sum_query Q[query][d] dS[query][key]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf[3], df[3];
            cut8_regs_t<true>(st, 8 * s2, H2Scales::P, pf);
            cut8_regs_t<true>(dp, 8 * s2, sc.ds, df);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 of[3], qfr[3];
                of[0] = read_tr8(o0, lo, 16 * s2, db); of[1] = read_tr8(o1, lo, 16 * s2, db); of[2] = of[1];
                dvacc[db] = mfma_pieces<true>(of, pf, dvacc[db]);
                qfr[0] = read_tr8(q0, lo, 16 * s2, db); qfr[1] = read_tr8(q1, lo, 16 * s2, db); qfr[2] = qfr[1];
                dkacc[db] = mfma_pieces<true>(qfr, df, dkacc[db]);
            }
        }
    }

    float vmax = 0.f;
    if (valid_k) {
        const float vs = final_stage ? (DROP ? rng.scale : 1.f) * (sc.inv_do * H2Scales::INV_P) : 1.f;
        const float ks = final_stage ? scale * (sc.inv_q * sc.inv_ds) : 1.f;
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
    }
    if (final_stage && want_amax) {                       // (gamer_amax_sink: max |dv|, the v columns of d(q|k|v))
        uint32_t mw = __float_as_uint(vmax);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
        if (lane == 0 && mw) atomicMax(&sm.amax_word, mw);
    }
}

template <int G, bool DROP, bool ORD>
__global__ void __launch_bounds__(DKV_R_THREADS, 1)
attn_bwd_dkv_r_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                      const float* __restrict__ d_o, const float* __restrict__ lse, const float* delta,
                      const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                      int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dk, int lddk, float* dv,
                      int lddv, const RowOrder ro, uint32_t* __restrict__ amax_out, const AttnAmax am) {
    extern __shared__ __attribute__((aligned(16))) unsigned char res_raw[];
    DkvRSmem& sm = *reinterpret_cast<DkvRSmem*>(res_raw);
    const H2Scales sc(am.q, am.k, am.v, am.d_o, 1.f, 1.f / (1.f - p_drop));
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_all = (S + 31) >> 5;
    const int nqb = (S + RB_KEYS - 1) / RB_KEYS;
    const int n_pairs = nbatch * nkv;
    if (tid == 0) sm.amax_word = 0;
#pragma unroll 1
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const int b = pair / nkv, kvh = pair % nkv;
#pragma unroll 1
        for (int st_i = 0; st_i < G * nqb; ++st_i) {
            const int g = st_i / nqb, qb = st_i % nqb;
            __syncthreads();                                // every wave is done with the previous stage's images (and flags)
            if (tid == 0) sm.ctr = 0;
            if (st_i == 0 && tid < RES_MAX_TILES) sm.started[tid] = 0;
            res_stage_queries<DROP, ORD>(sm, q, ldq, d_o, lse, delta, ql, row_empty, ro, b, kvh * G + g, qb * RB_KEYS, S, nq, tid, sc, rng);
            __syncthreads();
            const bool final_stage = st_i == G * nqb - 1;
            const int n_qt = min(RB_TILES, n_all - qb * RB_TILES);
            int blk_maxpos = -1, blk_empty = 0;
#pragma unroll 1
            for (int i = 0; i < n_qt; ++i) { blk_maxpos = max(blk_maxpos, sm.t_maxpos[i]); blk_empty |= sm.t_empty[i]; }
            blk_maxpos = __builtin_amdgcn_readfirstlane(blk_maxpos);
            blk_empty = __builtin_amdgcn_readfirstlane(blk_empty);
            // key tiles from a queue, early keys (seen by the most queries) first; the last stage visits every key tile
#pragma unroll 1
            for (int kt = res_grab_ctr(&sm.ctr, lane); kt < n_all; kt = res_grab_ctr(&sm.ctr, lane)) {
                if (!(final_stage || blk_empty != 0 || blk_maxpos >= kt * 32)) break;      // (nor does any later key tile)
                const bool fresh = __builtin_amdgcn_readfirstlane(sm.started[kt]) == 0;
                res_dkv_keytile<DROP>(sm, k, ldk, v, ldv, kl, S, scale, rng, dk, lddk, dv, lddv, b, kvh, kt, n_qt, fresh,
                                      final_stage, amax_out != nullptr, sc);
                if (lane == 0) sm.started[kt] = 1;
            }
        }
    }
    if (amax_out) {
        __syncthreads();
        if (tid == 0 && sm.amax_word) amax_publish(sm.amax_word, amax_out, blockIdx.x);
    }
}


// =============================================================================================
// backward: dK, dV for a GQA group of two query heads - TWO chains per wave.  tools/ubench_fill.hip on MI355X: inside one wave up to
// ~6 plain vector instructions (or 2 v_exp) issue in the shadow of one 32 x 32 x 16 MFMA at no cost; an in-order wave with ONE chain
// (products -> element-wise -> products, each depending on the one before) can never use that, and that is where the tiled and the
// single-chain resident kernels spend 60 % of their cycles.  Here a wave owns 32 keys and, for every 32-query tile, BOTH heads of the
// kv group (the same rows: Q / dO of both heads for a block of 128 query slots are resident, 128 KB): four slots per tile,
//     F(A)   |   E(A) beside F(B)   |   S(A) beside E(B)   |   S(B)
// (F: S = Q K^T and dP = dO V^T, 24 MFMAs; E: the ~230 vector instructions per tile that turn them into the fp16 pieces of P and dS;
// S: dV^T += dO^T P, dK^T += Q^T dS, 24 MFMAs), so the vector work of one head sits beside the matrix work of the other in one
// basic block.  Four waves per workgroup = one per SIMD with the whole 512-register file: the wave's K / V fragments (64 registers),
// dK^T / dV^T of its keys (64; both heads add into them - the GQA sum needs no LDS pass), two heads' S / dP tiles and pieces.
// Stages (query blocks of 128 slots, in order), the key-tile queue and the carry of a key tile's raw sums through dk / dv are those of
// the single-chain kernel above.
// =============================================================================================
constexpr int QB2 = 128;                      // query slots of one resident block
constexpr int QB2_TILES = QB2 / 32;
constexpr int QIMG2 = QB2 * 64;               // 16-bit elements of one piece image of one head
struct DkvR2Smem {
    bf16_t Q[2][2][QIMG2];                    // head x piece
    bf16_t dO[2][2][QIMG2];
    float nlse2[2][QB2];
    float ndelta[2][QB2];
    uint32_t aw[2][QB2];
    int32_t ql[QB2];
    int32_t empty[QB2];
    int32_t pos[QB2];                         // the row's causal key limit: its position (its span end with key spans), S past the end
    int32_t t_qlmin[QB2_TILES];
    int32_t t_posmin[QB2_TILES];
    int32_t t_maxpos[QB2_TILES];
    int32_t t_empty[QB2_TILES];
    int32_t started[RES_MAX_TILES];
    uint32_t amax_word;
    int32_t ctr;
    int32_t pad_[6];
    int32_t keeper_stage[RES_MAX_TILES];      // stage in which the tile is swept by the wave that kept it (-1: nobody's)
    int32_t hole_lo[QB2];                     // SPAN kernels only: the hole of every staged row, and per 32-row tile the smallest
    int32_t hole_hi[QB2];                     // hole start / largest hole end over its normal rows
    int32_t t_hlomin[QB2_TILES];
    int32_t t_hhimax[QB2_TILES];
};

template <bool DROP, bool ORD, bool SPAN>
__device__ __forceinline__ void
res2_stage_queries(DkvR2Smem& sm, const float* __restrict__ q, int ldq, const float* __restrict__ d_o, const float* __restrict__ lse,
                   const float* delta, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty, const RowOrder& ro,
                   int b, int head0, int s0, int S, int nq, int tid, const H2Scales& sc, const AttnDropout& rng) {
    const int32_t* pmap = ORD ? ro.perm + (int64_t)b * S : nullptr;
    // all 32 loads of a thread (both heads) are requested before the first cut waits for one
    float4 rq[2][8], rd[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = tid + DKV_R_THREADS * p;
            const int sl = min(s0 + (f >> 4), S - 1);
            const int64_t tok = (int64_t)b * S + (ORD ? pmap[sl] : sl);
            rq[g][p] = *reinterpret_cast<const float4*>(q + tok * ldq + (head0 + g) * 64 + ((f & 15) << 2));
            rd[g][p] = *reinterpret_cast<const float4*>(d_o + tok * (int64_t)nq * 64 + (head0 + g) * 64 + ((f & 15) << 2));
        }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g == 0) {
            // per-row scalars: threads 0..127 head 0 and the rows' own data, threads 128..255 head 1
            const int row = tid & (QB2 - 1), hh = tid >> 7;
            const int sl = s0 + row;
            const bool in = sl < S;
            const int pos = in ? (ORD ? pmap[sl] : sl) : 0;
            const int64_t hrow = ((int64_t)b * nq + head0 + hh) * S + pos;
            sm.nlse2[hh][row] = in ? -lse[hrow] * 1.4426950408889634f : 0.f;
            sm.ndelta[hh][row] = in ? -delta[hrow] : 0.f;
            if (DROP) sm.aw[hh][row] = rng.row_word((uint32_t)hrow);
            if (hh == 0) {
                const int em = in ? row_empty[(int64_t)b * S + pos] : 0;
                const int qlv = in ? (ql ? ql[(int64_t)b * S + pos] : 1) : 0;
                const QuerySpan qsp = QuerySpan::load<SPAN>(ro.span, (int64_t)b * S + pos, pos, in, S);   // hi = pos without spans, S past the end
                sm.ql[row] = qlv;
                sm.empty[row] = em;
                sm.pos[row] = qsp.hi;
                const bool normal = in && em == 0;
                int qlmin = normal ? qlv : INT_BIG_A, posmin = normal ? qsp.hi : INT_BIG_A, maxpos = normal ? qsp.hi : -1, anye = em != 0 ? 1 : 0;
                int hlomin = normal ? qsp.hole_lo : INT_BIG_A, hhimax = normal ? qsp.hole_hi : 0;
#pragma unroll
                for (int o2 = 16; o2 > 0; o2 >>= 1) {
                    qlmin = min(qlmin, __shfl_xor(qlmin, o2, 64));
                    posmin = min(posmin, __shfl_xor(posmin, o2, 64));
                    maxpos = max(maxpos, __shfl_xor(maxpos, o2, 64));
                    anye |= __shfl_xor(anye, o2, 64);
                    if (SPAN) { hlomin = min(hlomin, __shfl_xor(hlomin, o2, 64)); hhimax = max(hhimax, __shfl_xor(hhimax, o2, 64)); }
                }
                if (SPAN) { sm.hole_lo[row] = qsp.hole_lo; sm.hole_hi[row] = qsp.hole_hi; }
                if ((row & 31) == 0) {
                    sm.t_qlmin[row >> 5] = qlmin; sm.t_posmin[row >> 5] = posmin; sm.t_maxpos[row >> 5] = maxpos; sm.t_empty[row >> 5] = anye;
                    if (SPAN) { sm.t_hlomin[row >> 5] = hlomin; sm.t_hhimax[row >> 5] = hhimax; }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = tid + DKV_R_THREADS * p;
            const int row = f >> 4;
            const bool ok = s0 + row < S;
            const int off = sl_off(row, (f & 15) << 2);
            uint32_t a0, a1, b0, b1;
            cut2h_quad(ok ? rq[g][p].x : 0.f, ok ? rq[g][p].y : 0.f, ok ? rq[g][p].z : 0.f, ok ? rq[g][p].w : 0.f, sc.q, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.Q[g][0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.Q[g][1] + off) = make_uint2(a1, b1);
            cut2h_quad(ok ? rd[g][p].x : 0.f, ok ? rd[g][p].y : 0.f, ok ? rd[g][p].z : 0.f, ok ? rd[g][p].w : 0.f, sc.d_o, a0, a1, b0, b1);
            *reinterpret_cast<uint2*>(sm.dO[g][0] + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(sm.dO[g][1] + off) = make_uint2(a1, b1);
        }
    }
}

// classes of a query tile (the same for both heads: they share the rows)
enum { QT_FREE = 0, QT_MASK = 1, QT_EMPTYSEL = 2, QT_ALL_EMPTY = 3 };

// What a wave holds of its key tile between calls: the K / V fragments and the raw dK / dV sums.  A wave keeps the tile it swept last
// across the stage boundary (RES_KEEP): the next stage sweeps it first, without the prologue (K / V rows, cut, carried sums) and
// without the carry through dk / dv in between.
struct DkvTileState {
    bf16x8 kf[2][4], vf[2][4];            // piece x k-step: lane (r, h) holds d = 16 s + 8 h .. + 7 of its key's K / V row
    f32x16 dkacc[2], dvacc[2];
    int my_kl;
};

// One step of a wave: `release` - the sums of the held tile kt_rel leave (raw, or final: scaled, with the maximum); then, for kt >= 0,
// `acquire` - its K / V fragments and carried sums are fetched (else T already holds them) - and the sweep over the staged query tiles.
template <bool DROP, bool SPAN>
__device__ __forceinline__ void
res2_dkv_step(DkvR2Smem& sm, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
              const int32_t* __restrict__ kl, int S, float scale, const AttnDropout& rng, float* dk, int lddk, float* dv, int lddv,
              const int b, const int kvh, const int kt_rel, const bool release, const bool final_stage, const int kt,
              const bool acquire, const bool fresh, const int n_qt, const bool want_amax, const H2Scales& sc,
              DkvTileState& T RES_STAMP_ARGS) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    if (release) {
        const int jr = kt_rel * 32 + r;
        const bool valid_r = jr < S;
        const int64_t rtok = (int64_t)b * S + (valid_r ? jr : S - 1);
        float* dkrow_r = dk + rtok * lddk + kvh * 64;
        float* dvrow_r = dv + rtok * lddv + kvh * 64;
        float vmax = 0.f;
        if (valid_r) {
            const float vs = final_stage ? (DROP ? rng.scale : 1.f) * (sc.inv_do * H2Scales::INV_P) : 1.f;
            const float ks = final_stage ? scale * (sc.inv_q * sc.inv_ds) : 1.f;
    #pragma unroll
            for (int dh = 0; dh < 2; ++dh)
    #pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * dh + 8 * g4 + 4 * h;
                    *reinterpret_cast<float4*>(dkrow_r + d) = make_float4(T.dkacc[dh][4 * g4] * ks, T.dkacc[dh][4 * g4 + 1] * ks,
                                                                        T.dkacc[dh][4 * g4 + 2] * ks, T.dkacc[dh][4 * g4 + 3] * ks);
                    const float4 dv4 = make_float4(T.dvacc[dh][4 * g4] * vs, T.dvacc[dh][4 * g4 + 1] * vs,
                                                   T.dvacc[dh][4 * g4 + 2] * vs, T.dvacc[dh][4 * g4 + 3] * vs);
                    *reinterpret_cast<float4*>(dvrow_r + d) = dv4;
                    vmax = fmaxf(fmaxf(fmaxf(vmax, fabsf(dv4.x)), fabsf(dv4.y)), fmaxf(fabsf(dv4.z), fabsf(dv4.w)));
                }
        }
        if (final_stage && want_amax) {                       // (gamer_amax_sink: max |dv|, the v columns of d(q|k|v))
            uint32_t mw = __float_as_uint(vmax);
    #pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
            if (lane == 0 && mw) atomicMax(&sm.amax_word, mw);
        }
        RES_MARK(3);                                           // 3: key-tile epilogue (store)

    }
    if (kt < 0) return;
    const SlOffsets lo(lane);
    const int jk = kt * 32 + r;                          // this lane's key
    const bool valid_k = jk < S;
    const int64_t ktok = (int64_t)b * S + (valid_k ? jk : S - 1);

    if (acquire) {
        // every global load of the prologue is requested before anything waits for one: the K / V rows, the key's level, the carried sums
        float* dkrow = dk + ktok * lddk + kvh * 64;
        float* dvrow = dv + ktok * lddv + kvh * 64;
        float4 kraw[8], vraw[8];
        {
            const float* krow = k + ktok * ldk + kvh * 64 + 8 * h;
            const float* vrow = v + ktok * ldv + kvh * 64 + 8 * h;
    #pragma unroll
            for (int s = 0; s < 4; ++s) {
                kraw[2 * s] = *reinterpret_cast<const float4*>(krow + 16 * s);
                kraw[2 * s + 1] = *reinterpret_cast<const float4*>(krow + 16 * s + 4);
                vraw[2 * s] = *reinterpret_cast<const float4*>(vrow + 16 * s);
                vraw[2 * s + 1] = *reinterpret_cast<const float4*>(vrow + 16 * s + 4);
            }
        }
        T.my_kl = valid_k ? kl[ktok] : INT_BIG_A;
        if (fresh) {
    #pragma unroll
            for (int i = 0; i < 16; ++i) { T.dkacc[0][i] = 0.f; T.dkacc[1][i] = 0.f; T.dvacc[0][i] = 0.f; T.dvacc[1][i] = 0.f; }
        } else {
            // the raw sums an earlier stage left (possibly another wave: read past the L1)
    #pragma unroll
            for (int dh = 0; dh < 2; ++dh)
    #pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * dh + 8 * g4 + 4 * h;
                    f32x4v a = {0.f, 0.f, 0.f, 0.f}, c = a;
                    if (valid_k) {
                        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dkrow + d));
                        c = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dvrow + d));
                    }
    #pragma unroll
                    for (int e = 0; e < 4; ++e) { T.dkacc[dh][4 * g4 + e] = a[e]; T.dvacc[dh][4 * g4 + e] = c[e]; }
                }
        }
    #pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4 = kraw[2 * s], b4 = kraw[2 * s + 1], c4 = vraw[2 * s], d4 = vraw[2 * s + 1];
            if (!valid_k) { a4 = make_float4(0.f, 0.f, 0.f, 0.f); b4 = a4; c4 = a4; d4 = a4; }
            bf16x8 pc[3];
            cut8_t<true>(a4, b4, sc.k, pc);
            T.kf[0][s] = pc[0]; T.kf[1][s] = pc[1];
            cut8_t<true>(c4, d4, sc.v, pc);
            T.vf[0][s] = pc[0]; T.vf[1][s] = pc[1];
        }

    }
    bf16x8 (&kf)[2][4] = T.kf;
    bf16x8 (&vf)[2][4] = T.vf;
    f32x16 (&dkacc)[2] = T.dkacc;
    f32x16 (&dvacc)[2] = T.dvacc;
    const int my_kl = T.my_kl;
    const int wave_kl_max = wave_max_i32_dpp(my_kl);
    const int wave_k_lo = kt * 32, wave_k_hi = wave_k_lo + 31;
    const float invS = 1.f / (float)S;
    const float c2 = scale * 1.4426950408889634f * sc.inv_qk;
    const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
    const float cdp = sc.inv_v * sc.inv_do;              // dP = (dO' V'^T) cdp
    const float sd = rng.scale * cdp;

    // ---- the pieces of a tile's work ------------------------------------------------------------------------------------------
    // F, k-step s of head g / tile qt: the row fragments of dO and Q (A operands)
    auto f_reads = [&](int g, int qt, int s, bf16x8 (&a)[4]) {
        a[0] = *reinterpret_cast<const bf16x8*>(sm.dO[g][0] + qt * SIMG + lo.row[s]);
        a[1] = *reinterpret_cast<const bf16x8*>(sm.dO[g][1] + qt * SIMG + lo.row[s]);
        a[2] = *reinterpret_cast<const bf16x8*>(sm.Q[g][0] + qt * SIMG + lo.row[s]);
        a[3] = *reinterpret_cast<const bf16x8*>(sm.Q[g][1] + qt * SIMG + lo.row[s]);
    };
    // S, group (s2, db) of head g / tile qt: dV^T += dO^T P, dK^T += Q^T dS (transposing reads of the resident images)
    auto s_reads = [&](int g, int qt, int s2, int db, bf16x8 (&a)[4]) {
        a[0] = read_tr8(sm.dO[g][0] + qt * SIMG, lo, 16 * s2, db);
        a[1] = read_tr8(sm.dO[g][1] + qt * SIMG, lo, 16 * s2, db);
        a[2] = read_tr8(sm.Q[g][0] + qt * SIMG, lo, 16 * s2, db);
        a[3] = read_tr8(sm.Q[g][1] + qt * SIMG, lo, 16 * s2, db);
    };
    auto e_cut = [&](int s2, const f32x16& st, const f32x16& dp, bf16x8 (&pf)[2][2], bf16x8 (&df)[2][2]) {
        bf16x8 t3[3];
        cut8_regs_t<true>(st, 8 * s2, H2Scales::P, t3);
        pf[s2][0] = t3[0]; pf[s2][1] = t3[1];
        cut8_regs_t<true>(dp, 8 * s2, sc.ds, t3);
        df[s2][0] = t3[0]; df[s2][1] = t3[1];
    };

    RES_MARK(1);                                           // 1: key-tile prologue (K / V rows + cut, key data, carried sums)
    // F of one k-step / S of one (s2, db) group: six MFMAs each
    auto f_step = [&]<bool NEED_S>(const bf16x8 (&a)[4], int s, f32x16& st, f32x16& dp) {
        dp = mfma_piece<true>(a[1], vf[0][s], dp);
        if (NEED_S) st = mfma_piece<true>(a[3], kf[0][s], st);
        dp = mfma_piece<true>(a[0], vf[1][s], dp);
        if (NEED_S) st = mfma_piece<true>(a[2], kf[1][s], st);
        dp = mfma_piece<true>(a[0], vf[0][s], dp);
        if (NEED_S) st = mfma_piece<true>(a[2], kf[0][s], st);
    };
    auto s_step = [&](const bf16x8 (&a)[4], const bf16x8 (&pf)[2][2], const bf16x8 (&df)[2][2], int s2, int db) {
        dvacc[db] = mfma_piece<true>(a[1], pf[s2][0], dvacc[db]);
        dkacc[db] = mfma_piece<true>(a[3], df[s2][0], dkacc[db]);
        dvacc[db] = mfma_piece<true>(a[0], pf[s2][1], dvacc[db]);
        dkacc[db] = mfma_piece<true>(a[2], df[s2][1], dkacc[db]);
        dvacc[db] = mfma_piece<true>(a[0], pf[s2][0], dvacc[db]);
        dkacc[db] = mfma_piece<true>(a[2], df[s2][0], dkacc[db]);
    };
    // E, registers 4 g4 .. 4 g4 + 3 of head g / tile qt (query = register, key = lane):  P -> st (1 / (1 - p) applied at the end), dS -> dp
    // Rows past the end of the sequence have Q = dO = 0, lse = delta = 0: p = 1 but dS = 0 and dO = 0.
    auto e_chunk = [&]<int CLS>(int g, int qt, int g4, f32x16& st, f32x16& dp) {
        constexpr bool MASK = CLS == QT_MASK || CLS == QT_EMPTYSEL, EMPTYSEL = CLS == QT_EMPTYSEL, ALL_EMPTY = CLS == QT_ALL_EMPTY;
        const int qb = qt * 32 + 8 * g4 + 4 * h;
        const float4 d4 = *reinterpret_cast<const float4*>(&sm.ndelta[g][qb]);
        const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
        float nl[4] = {0.f, 0.f, 0.f, 0.f};
        int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0}, emv[4] = {0, 0, 0, 0};
        uint32_t awv[4] = {0, 0, 0, 0};
        if (!ALL_EMPTY) {
            const float4 l4 = *reinterpret_cast<const float4*>(&sm.nlse2[g][qb]);
            nl[0] = l4.x; nl[1] = l4.y; nl[2] = l4.z; nl[3] = l4.w;
        }
        int hlo[4] = {0, 0, 0, 0}, hhi[4] = {0, 0, 0, 0};
        if (MASK) {
            const int4 q4 = *reinterpret_cast<const int4*>(&sm.ql[qb]);
            const int4 p4 = *reinterpret_cast<const int4*>(&sm.pos[qb]);
            qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
            posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
            if (SPAN) {
                const int4 a4 = *reinterpret_cast<const int4*>(&sm.hole_lo[qb]);
                const int4 b4 = *reinterpret_cast<const int4*>(&sm.hole_hi[qb]);
                hlo[0] = a4.x; hlo[1] = a4.y; hlo[2] = a4.z; hlo[3] = a4.w;
                hhi[0] = b4.x; hhi[1] = b4.y; hhi[2] = b4.z; hhi[3] = b4.w;
            }
        }
        if (EMPTYSEL) {
            const int4 e4 = *reinterpret_cast<const int4*>(&sm.empty[qb]);
            emv[0] = e4.x; emv[1] = e4.y; emv[2] = e4.z; emv[3] = e4.w;
        }
        if (DROP) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(&sm.aw[g][qb]);
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
                const float tt = keep ? dp[reg] : 0.f;
                dp[reg] = pe * fmaf(tt, sd, ndl[e]);         // dS[query][key]
                st[reg] = keep ? pe : 0.f;                   // dropped P[query][key] * (1 - p)
            } else {
                dp[reg] = pe * fmaf(dp[reg], cdp, ndl[e]);
                st[reg] = pe;
            }
        }
    };
    // first query tile that reaches this wave's keys (slots are sorted: normal rows by position, then the "empty" rows)
    int q_first = 0;
    while (q_first < n_qt && sm.t_maxpos[q_first] < wave_k_lo && sm.t_empty[q_first] == 0) ++q_first;
    q_first = __builtin_amdgcn_readfirstlane(q_first);

#pragma unroll 1
    for (int qt = q_first; qt < n_qt; ++qt) {
        const bool tile_has_empty = __builtin_amdgcn_readfirstlane(sm.t_empty[qt]) != 0;
        const int posmin = __builtin_amdgcn_readfirstlane(sm.t_posmin[qt]);
        const int qlmin = __builtin_amdgcn_readfirstlane(sm.t_qlmin[qt]);
        const int maxpos = __builtin_amdgcn_readfirstlane(sm.t_maxpos[qt]);
        const bool tile_all_empty = tile_has_empty && posmin == INT_BIG_A;
        const bool before = maxpos < wave_k_lo;            // every normal query of the tile precedes this wave's keys
        if (before && !tile_has_empty) continue;
        const bool need_s = !before && !tile_all_empty;
        bool free_tile = !tile_has_empty && (posmin >= wave_k_hi) && (wave_kl_max < qlmin);
        if (SPAN && free_tile) {                            // (no normal row's hole may touch this wave's keys)
            const int hlomin = __builtin_amdgcn_readfirstlane(sm.t_hlomin[qt]), hhimax = __builtin_amdgcn_readfirstlane(sm.t_hhimax[qt]);
            free_tile = wave_k_hi < hlomin || wave_k_lo >= hhimax;
        }

        f32x16 stA, dpA, stB, dpB;
#pragma unroll
        for (int i = 0; i < 16; ++i) { stA[i] = 0.f; dpA[i] = 0.f; stB[i] = 0.f; dpB[i] = 0.f; }
        bf16x8 pfA[2][2], dfA[2][2], pfB[2][2], dfB[2][2];

        // ---- slot 1: F(A) -------------------------------------------------------------------------------------------------
        {
            bf16x8 a[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) f_reads(0, qt, s, a[s]);
            if (need_s) {
#pragma unroll
                for (int s = 0; s < 4; ++s) f_step.template operator()<true>(a[s], s, stA, dpA);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) f_step.template operator()<false>(a[s], s, stA, dpA);
            }
        }
        // ---- slot 2: E(A) beside F(B) -----------------------------------------------------------------------------------
        auto slot2 = [&]<int CLS, bool NEED_S>() {
            bf16x8 a[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) f_reads(1, qt, s, a[s]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f_step.template operator()<NEED_S>(a[s], s, stB, dpB);
                e_chunk.template operator()<CLS>(0, qt, s, stA, dpA);
                if (s & 1) e_cut(s >> 1, stA, dpA, pfA, dfA);
            }
        };
        // ---- slot 3: S(A) beside E(B) -----------------------------------------------------------------------------------
        auto slot3 = [&]<int CLS>() {
            bf16x8 a[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s_reads(0, qt, i >> 1, i & 1, a[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_step(a[i], pfA, dfA, i >> 1, i & 1);
                e_chunk.template operator()<CLS>(1, qt, i, stB, dpB);
                if (i & 1) e_cut(i >> 1, stB, dpB, pfB, dfB);
            }
        };
        if (tile_all_empty) { slot2.template operator()<QT_ALL_EMPTY, false>(); slot3.template operator()<QT_ALL_EMPTY>(); }
        else if (tile_has_empty) {
            if (need_s) slot2.template operator()<QT_EMPTYSEL, true>(); else slot2.template operator()<QT_EMPTYSEL, false>();
            slot3.template operator()<QT_EMPTYSEL>();
        } else if (free_tile) { slot2.template operator()<QT_FREE, true>(); slot3.template operator()<QT_FREE>(); }
        else { slot2.template operator()<QT_MASK, true>(); slot3.template operator()<QT_MASK>(); }
        // ---- slot 4: S(B) -------------------------------------------------------------------------------------------------
        {
            bf16x8 a[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s_reads(1, qt, i >> 1, i & 1, a[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) s_step(a[i], pfB, dfB, i >> 1, i & 1);
        }
#if RES_STAMP
        ph_[6] += 1;                                       // 6: (query tile x two heads) steps
#endif
    }
    RES_MARK(2);                                           // 2: query-tile loop

}

template <bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(DKV_R_THREADS, 1)
attn_bwd_dkv_r2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                       const float* __restrict__ d_o, const float* __restrict__ lse, const float* delta,
                       const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, const int32_t* __restrict__ row_empty,
                       int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dk, int lddk, float* dv,
                       int lddv, const RowOrder ro, uint32_t* __restrict__ amax_out, const AttnAmax am, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char res_raw[];
    DkvR2Smem& sm = *reinterpret_cast<DkvR2Smem*>(res_raw);
    const H2Scales sc(am.q, am.k, am.v, am.d_o, 1.f, 1.f / (1.f - p_drop));
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_all = (S + 31) >> 5;
    const int nqb = (S + QB2 - 1) / QB2;
    // unit of a workgroup: a (sequence, kv head) pair, or - split - the even / odd key tiles of one (the query blocks are then staged
    // by both workgroups of the pair)
    const int upp = split ? 2 : 1;
    const int n_pairs = nbatch * nkv * upp;                 // (units)
    if (tid == 0) sm.amax_word = 0;
#if RES_STAMP
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    DkvTileState T;
#pragma unroll 1
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const int b = (pair / upp) / nkv, kvh = (pair / upp) % nkv, kt0 = pair % upp;
        int held = -1;                                      // the key tile this wave's registers hold (T), -1: none
#pragma unroll 1
        for (int qb = 0; qb < nqb; ++qb) {
            RES_MARK(5);                                    // 5: queue / loop control
            __syncthreads();                                // every wave is done with the previous stage's images (and flags)
            RES_MARK(4);                                    // 4: waiting for the slowest wave of the stage
            if (tid == 0) sm.ctr = 0;
            if (qb == 0 && tid < RES_MAX_TILES) { sm.started[tid] = 0; sm.keeper_stage[tid] = -1; }
            res2_stage_queries<DROP, ORD, SPAN>(sm, q, ldq, d_o, lse, delta, ql, row_empty, ro, b, kvh * 2, qb * QB2, S, nq, tid, sc, rng);
            __syncthreads();
            RES_MARK(0);                                    // 0: staging (loads, cut, LDS stores, barrier)
            // warm the L2 with the Q / dO rows of the block this workgroup stages next: 128 slots x 2 heads x 2 tensors x two 128-byte
            // lines = 4 per thread (natural row order only: the sorted order would need the permutation first)
            uint32_t tw[4] = {0, 0, 0, 0};
            if (RES_WARM && !ORD) {
                int npair = pair, nqb_ = qb + 1;
                if (nqb_ == nqb) { nqb_ = 0; npair += gridDim.x; }
                if (npair < n_pairs) {
                    const int sl = min(nqb_ * QB2 + (tid >> 1), S - 1);
                    const int64_t ntok = (int64_t)((npair / upp) / nkv) * S + sl;
                    const int nh0 = ((npair / upp) % nkv) * 2;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        tw[2 * g] = *reinterpret_cast<const uint32_t*>(q + ntok * ldq + (nh0 + g) * 64 + (tid & 1) * 32);
                        tw[2 * g + 1] = *reinterpret_cast<const uint32_t*>(d_o + ntok * (int64_t)nq * 64 + (nh0 + g) * 64 + (tid & 1) * 32);
                    }
                }
            }
            const bool final_stage = qb == nqb - 1;
            const int n_qt = min(QB2_TILES, n_all - qb * QB2_TILES);
            int blk_maxpos = -1, blk_empty = 0;
#pragma unroll 1
            for (int i = 0; i < n_qt; ++i) { blk_maxpos = max(blk_maxpos, sm.t_maxpos[i]); blk_empty |= sm.t_empty[i]; }
            blk_maxpos = __builtin_amdgcn_readfirstlane(blk_maxpos);
            blk_empty = __builtin_amdgcn_readfirstlane(blk_empty);
            // Key tiles from the stage's queue, early keys (seen by the most queries) first; the last stage visits every key tile.  A wave
            // keeps the tile it swept last in its registers across the stage boundary (RES_KEEP): it stamps the tile as its own for the
            // next stage (the queue's other takers skip it), sweeps it first there - no K / V prologue, no carried sums - and lets it go
            // when it takes another tile, or for good at the end of the last stage.
            auto needed = [&](int kt_) { return final_stage || blk_empty != 0 || blk_maxpos >= kt_ * 32; };
            bool first = true;
#pragma unroll 1
            while (true) {
                int kt = -1;
                if (RES_KEEP && first && held >= 0 && needed(held)) {
                    kt = held;
                } else {
                    int c = upp * res_grab_ctr(&sm.ctr, lane) + kt0;
                    // (>= qb: a keeper that ends the stage still holding its tile re-stamps it qb + 1 while others may still be reading the stamp)
                    while (RES_KEEP && c < n_all && __builtin_amdgcn_readfirstlane(sm.keeper_stage[c]) >= qb) c = upp * res_grab_ctr(&sm.ctr, lane) + kt0;
                    if (c < n_all && needed(c)) kt = c;             // (else: nor does any later key tile)
                }
                first = false;
                const bool release = held >= 0 && kt != held && (kt >= 0 || final_stage);
                const bool acquire = kt >= 0 && kt != held;
                if (!release && kt < 0) break;
                const bool fresh = acquire && __builtin_amdgcn_readfirstlane(sm.started[kt]) == 0;
                RES_MARK(5);
                res2_dkv_step<DROP, SPAN>(sm, k, ldk, v, ldv, kl, S, scale, rng, dk, lddk, dv, lddv, b, kvh, held, release, final_stage, kt,
                                          acquire, fresh, n_qt, amax_out != nullptr, sc, T RES_STAMP_PASS);
                if (release) held = -1;
                if (kt < 0) break;
                if (acquire && lane == 0) sm.started[kt] = 1;
                held = kt;
                if (!RES_KEEP) {                                    // (the round-5 order: every tile leaves right behind its sweep)
                    res2_dkv_step<DROP, SPAN>(sm, k, ldk, v, ldv, kl, S, scale, rng, dk, lddk, dv, lddv, b, kvh, held, true, final_stage, -1,
                                              false, false, n_qt, amax_out != nullptr, sc, T RES_STAMP_PASS);
                    held = -1;
                }
            }
            if (RES_KEEP && held >= 0 && lane == 0) sm.keeper_stage[held] = qb + 1;      // (not the last stage: it let every tile go)
            asm volatile("" :: "v"(tw[0]), "v"(tw[1]), "v"(tw[2]), "v"(tw[3]));
        }
    }
    if (amax_out) {
        __syncthreads();
        if (tid == 0 && sm.amax_word) amax_publish(sm.amax_word, amax_out, blockIdx.x);
    }
#if RES_STAMP
    RES_MARK(5);
    if (g_res_stamp && lane == 0) {
        unsigned long long* rec = g_res_stamp + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) rec[i_] += ph_[i_];
    }
#endif
}

template <int G, bool DROP, bool SPAN, bool ORD>
static int launch_fwd_r_variant(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* kl,
                                const int32_t* ql, const int32_t* row_empty, int B, int S, int nq, int nkv, float scale, float p_drop,
                                uint64_t seed, float* o, float* lse, RowOrder ro, uint32_t* amax_out, AttnAmax am, hipStream_t st) {
    const size_t shmem = sizeof(ResSmem);
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr_set = attr_dev[current_device()];
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_r_kernel<G, DROP, SPAN, ORD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_fwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int split = res_split(B * nkv, G);
    const int n_units = B * nkv * (split ? G : 1);
    const int grid = n_units < res_grid_cap() ? n_units : res_grid_cap();
    hipLaunchKernelGGL((attn_fwd_r_kernel<G, DROP, SPAN, ORD>), dim3(grid), dim3(RES_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, kl, ql,
                       row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, ro, amax_out, am, split);
    GAMER_CHECK_LAUNCH("gamer_attn_fwd_split/resident");
    return 0;
}

int launch_fwd_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* kl, const int32_t* ql,
                   const int32_t* row_empty, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* o,
                   float* lse, RowOrder ro, uint32_t* amax_out, AttnAmax am, hipStream_t st) {
#define GAMER_LAUNCH_FWD_RS(GV, DROPV, SPANV, ORDV)                                                                           \
    return launch_fwd_r_variant<GV, DROPV, SPANV, ORDV>(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, \
                                                        lse, ro, amax_out, am, st)
#define GAMER_LAUNCH_FWD_R(GV, DROPV, ORDV) GAMER_LAUNCH_FWD_RS(GV, DROPV, false, ORDV)
    if (ro.span) {                          // per-query key spans (Qwen3SessionMulti): built for the GQA group of two (the caller checks)
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_RS(2, true, true, true); else GAMER_LAUNCH_FWD_RS(2, true, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_RS(2, false, true, true); else GAMER_LAUNCH_FWD_RS(2, false, true, false); }
    }
    if (nq / nkv == 1) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_R(1, true, true); else GAMER_LAUNCH_FWD_R(1, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_R(1, false, true); else GAMER_LAUNCH_FWD_R(1, false, false); }
    } else {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_FWD_R(2, true, true); else GAMER_LAUNCH_FWD_R(2, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_FWD_R(2, false, true); else GAMER_LAUNCH_FWD_R(2, false, false); }
    }
#undef GAMER_LAUNCH_FWD_R
#undef GAMER_LAUNCH_FWD_RS
    return -1;      // not reached
}

template <int G, bool DROP, bool SPAN, bool ORD>
static int launch_dq_r_variant(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                               const float* d_o, const float* lse, float* delta, const int32_t* kl, const int32_t* ql,
                               const int32_t* row_empty, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                               float* dq, int lddq, RowOrder ro, int delta_ready, AttnAmax am, hipStream_t st) {
    const size_t shmem = sizeof(ResSmem);
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr_set = attr_dev[current_device()];
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_r_kernel<G, DROP, SPAN, ORD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_bwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int split = res_split(B * nkv, G);
    const int n_units = B * nkv * (split ? G : 1);
    const int grid = n_units < res_grid_cap() ? n_units : res_grid_cap();
    hipLaunchKernelGGL((attn_bwd_dq_r_kernel<G, DROP, SPAN, ORD>), dim3(grid), dim3(RES_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, o, d_o,
                       lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, ro, delta_ready, am, split);
    GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dq resident");
    return 0;
}

int launch_dq_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                  const float* lse, float* delta, const int32_t* kl, const int32_t* ql, const int32_t* row_empty, int B, int S,
                  int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dq, int lddq, RowOrder ro, int delta_ready,
                  AttnAmax am, hipStream_t st) {
#define GAMER_LAUNCH_DQ_RS(GV, DROPV, SPANV, ORDV)                                                                            \
    return launch_dq_r_variant<GV, DROPV, SPANV, ORDV>(q, ldq, k, ldk, v, ldv, o, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, \
                                                       p_drop, seed, dq, lddq, ro, delta_ready, am, st)
#define GAMER_LAUNCH_DQ_R(GV, DROPV, ORDV) GAMER_LAUNCH_DQ_RS(GV, DROPV, false, ORDV)
    if (ro.span) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DQ_RS(2, true, true, true); else GAMER_LAUNCH_DQ_RS(2, true, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DQ_RS(2, false, true, true); else GAMER_LAUNCH_DQ_RS(2, false, true, false); }
    }
    if (nq / nkv == 1) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DQ_R(1, true, true); else GAMER_LAUNCH_DQ_R(1, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DQ_R(1, false, true); else GAMER_LAUNCH_DQ_R(1, false, false); }
    } else {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DQ_R(2, true, true); else GAMER_LAUNCH_DQ_R(2, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DQ_R(2, false, true); else GAMER_LAUNCH_DQ_R(2, false, false); }
    }
#undef GAMER_LAUNCH_DQ_R
#undef GAMER_LAUNCH_DQ_RS
    return -1;      // not reached
}

template <int G, bool DROP, bool SPAN, bool ORD>
static int launch_dkv_r_variant(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* d_o,
                                const float* lse, const float* delta, const int32_t* kl, const int32_t* ql, const int32_t* row_empty,
                                int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dk, int lddk,
                                float* dv, int lddv, RowOrder ro, uint32_t* amax_out, AttnAmax am, hipStream_t st) {
    const size_t shmem = sizeof(DkvRSmem);
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr_set = attr_dev[current_device()];
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_r_kernel<G, DROP, ORD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_bwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int n_pairs = B * nkv;
    const int grid = n_pairs < res_grid_cap() ? n_pairs : res_grid_cap();
    if (G == 2 && (SPAN || GAMER_RES_PART("GAMER_ATTN_RES_DKV2"))) {
        // two heads per wave (the element-wise work of one beside the products of the other)
        const size_t shmem2 = sizeof(DkvR2Smem);
        static bool attr2_dev[MAX_DEVICES] = {};
        bool& attr2_set = attr2_dev[current_device()];
        if (!attr2_set) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_r2_kernel<DROP, SPAN, ORD>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2);
            if (e != hipSuccess) {
                set_error("gamer_attn_bwd_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return (int)e;
            }
            attr2_set = true;
        }
        const int split = res_split(n_pairs, 2);
        const int n_units = n_pairs * (split ? 2 : 1);
        const int grid2 = n_units < res_grid_cap() ? n_units : res_grid_cap();
        hipLaunchKernelGGL((attn_bwd_dkv_r2_kernel<DROP, SPAN, ORD>), dim3(grid2), dim3(DKV_R_THREADS), shmem2, st, q, ldq, k, ldk, v, ldv, d_o,
                           lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv, lddv, ro, amax_out, am, split);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dkv resident, two heads per wave");
        return 0;
    }
    if constexpr (!SPAN) {                  // (the single-chain kernel takes no spans: SPAN calls are G = 2 and ended above)
        hipLaunchKernelGGL((attn_bwd_dkv_r_kernel<G, DROP, ORD>), dim3(grid), dim3(DKV_R_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, d_o, lse,
                           delta, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv, lddv, ro, amax_out, am);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_split/dkv resident");
    }
    return 0;
}

int launch_dkv_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* d_o, const float* lse,
                   const float* delta, const int32_t* kl, const int32_t* ql, const int32_t* row_empty, int B, int S, int nq, int nkv,
                   float scale, float p_drop, uint64_t seed, float* dk, int lddk, float* dv, int lddv, RowOrder ro, uint32_t* amax_out,
                   AttnAmax am, hipStream_t st) {
#define GAMER_LAUNCH_DKV_RS(GV, DROPV, SPANV, ORDV)                                                                           \
    return launch_dkv_r_variant<GV, DROPV, SPANV, ORDV>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, row_empty, B, S, nq, nkv, scale, \
                                                        p_drop, seed, dk, lddk, dv, lddv, ro, amax_out, am, st)
#define GAMER_LAUNCH_DKV_R(GV, DROPV, ORDV) GAMER_LAUNCH_DKV_RS(GV, DROPV, false, ORDV)
    if (ro.span) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DKV_RS(2, true, true, true); else GAMER_LAUNCH_DKV_RS(2, true, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DKV_RS(2, false, true, true); else GAMER_LAUNCH_DKV_RS(2, false, true, false); }
    }
    if (nq / nkv == 1) {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DKV_R(1, true, true); else GAMER_LAUNCH_DKV_R(1, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DKV_R(1, false, true); else GAMER_LAUNCH_DKV_R(1, false, false); }
    } else {
        if (p_drop > 0.f) { if (ro.perm) GAMER_LAUNCH_DKV_R(2, true, true); else GAMER_LAUNCH_DKV_R(2, true, false); }
        else { if (ro.perm) GAMER_LAUNCH_DKV_R(2, false, true); else GAMER_LAUNCH_DKV_R(2, false, false); }
    }
#undef GAMER_LAUNCH_DKV_R
#undef GAMER_LAUNCH_DKV_RS
    return -1;      // not reached
}

}  // namespace gamer

#if RES_STAMP
extern "C" int gamer_debug_res_stamp(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(gamer::g_res_stamp), &v, sizeof(v));
}
#endif
