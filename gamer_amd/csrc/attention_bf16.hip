// Multi-behaviour flash attention for gfx950 on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), head_dim 64: the
// attention of the reference's --bf16 run (ref:SeqRec/tasks/train_SMB_decoder.py:114-118: HF Trainer autocast, under
// which sdpa_attention_forward gets bf16 q, k, v AND a bf16 mask; ref:SeqRec/models/generative/Qwen3Multi/model.py:133-143).
//
// Semantics that differ from the fp32 kernels (attention.hip), pinned by tests/golden/*_bf16.npz: the additive
// finfo(float32).min mask is not representable in bf16 and becomes -inf, and SDPA's softmax gives 0 for a row whose
// scores are all -inf.  So a query row with no allowed key ("empty" row) produces an all-zero output and no gradient -
// which is simply what the masked online softmax yields (l = 0), no special path.  lse of such a row is stored as 0.
//
// Arithmetic: bf16 q / k / v / dO, fp32 scores and softmax statistics, un-normalised probabilities rounded to bf16 for
// the second product (as the fused SDPA kernels do), fp32 accumulators, bf16 o / dq / dk / dv.
//
// Data flow (4 waves per workgroup, G = nq / nkv query heads share a K/V head):
//   fwd / dq : workgroup = (b, kv head, tile of (4/G)*32 queries); wave = (query head, 32 queries).  K/V tiles of 64
//              keys stream through LDS (double buffered, one barrier per tile).  S^T = K Q^T puts the query on the
//              lane: softmax state is lane-local and the score registers, converted pairwise to bf16, ARE the B operand
//              of O^T += V^T P^T (dQ^T += K^T dS^T).  V^T (K^T) fragments come from the row-major LDS image through
//              ds_read_b64_tr_b16.
//   dkv      : workgroup = (b, kv head, tile of (4/G)*32 keys); wave = (query head, 32 keys), K/V fragments in
//              registers; Q/dO tiles of 32 queries stream through LDS and are read by rows for S = Q K^T, dP = dO V^T
//              and transposed for dV^T += dO^T P, dK^T += Q^T dS.  The G heads are summed through LDS at the end.
//   No atomics: dQ has its own kernel (the bf16 products are cheap; the exponentials are recomputed).
// One LDS image serves row reads and transposing reads without bank conflicts: 128-byte rows, the 16-byte chunk index
// XOR-ed with a 3-bit permutation of the row's bits 1..3 (lds_off below).
#include "common.h"
#include <stdlib.h>

namespace gamer {

constexpr int AB_THREADS = 256;
#ifndef AB_ABLATE
#define AB_ABLATE 0       // timing-only: 1 = the forward re-stages its first K/V tile instead of loading the next ones
#endif
#ifndef AB_OCC_FWD
#define AB_OCC_FWD 2      // workgroups per CU the forward / dQ / dK-dV kernels are compiled for (register budget 512 / (2 x this))
#endif
#ifndef AB_OCC_DQ
#define AB_OCC_DQ 2
#endif
#ifndef AB_ABLATE_DKV
#define AB_ABLATE_DKV 0   // timing-only builds of attn_bwd_dkv_b_kernel (tools/ablate_attn_bf16_dkv.sh): bits 1 S/dP products, 2 element-wise, 4 dV/dK products, 8 Q/dO loads, 16 barriers | final stores, 32 LDS stores | K/V fragment loads, 64 park, 128 row scalars (tiled | head-resident kernel)
#endif
#ifndef AB_OCC_DKV
#define AB_OCC_DKV 2
#endif
constexpr int INT_BIG_B = 0x7fffffff;
constexpr float RESCALE_TAU_B = 16.f;      // log2 domain: p <= 2^16 relative to the running reference

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int rowmap_b(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// element offset of (row, column) in a [rows][64] bf16 image with swizzled 16-byte chunks.
//   row reads (ds_read_b128, one row per lane, same chunk): the rows of a hardware lane group get 16 distinct
//   (row parity, chunk ^ f) bank slots; transposing reads (4 consecutive rows x 32 columns per half wave): bit 1 of the
//   row flips the 64-byte half, its parity the 128-byte half of the 256-byte bank line.
__device__ __forceinline__ int lds_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int lds_off(int row, int col) { return row * 64 + ((((col >> 3) ^ lds_f(row))) << 3) + (col & 7); }

struct QuerySpanB {
    int hi, hole_lo, hole_hi;
    template <bool SPAN>
    static __device__ __forceinline__ QuerySpanB load(const int32_t* __restrict__ span, int64_t tok, int pos, bool valid) {
        QuerySpanB q;
        q.hi = valid ? pos : -1;             // rows past the end of the sequence attend nothing
        q.hole_lo = INT_BIG_B;
        q.hole_hi = 0;
        if (SPAN && valid) {
            const int4 t = reinterpret_cast<const int4*>(span)[tok];
            q.hi = t.x; q.hole_lo = t.y; q.hole_hi = t.z;
        }
        return q;
    }
};

// Optional query-row order (behaviour-level "cross" attention; the fp32 kernels' RowOrder): the rows of every sequence
// are visited through `perm` (sorted slot -> position: rows with an allowed key first, ascending; rows without one -
// "empty" rows, whose bf16 output and gradient are 0 - behind them), so that the tiles behind the last normal row of a
// sequence cost nothing and no tile mixes a short key range with dead rows.  `tile_maxpos[b][t]` = largest position of
// a normal row in the 32-slot tile t, -1 if it has none; `row_empty[b][position]` != 0 marks the empty rows.
struct QOrdB {
    const int32_t* perm;
    const int32_t* tile_maxpos;
    const int32_t* row_empty;
};

// persistent workgroups walking equal-work (heavy tile + light tile) items; see attention.hip for the measurements
struct WorkListB {
    int xcd, pair_slot, u, pairs_per_round, n_pairs, n_tiles;
    bool valid;
    __device__ __forceinline__ WorkListB(int n_pairs_, int n_tiles_) : n_pairs(n_pairs_), n_tiles(n_tiles_) {
        const int halves = (n_tiles + 1) >> 1;
        const int nlocal = (int)gridDim.x >> 3;
        xcd = blockIdx.x & 7;
        const int local = blockIdx.x >> 3;
        pairs_per_round = nlocal / halves;
        pair_slot = local / halves;
        u = local % halves;
        valid = pair_slot < pairs_per_round;
    }
    __device__ __forceinline__ int pair_at(int it) const { return (it * pairs_per_round + pair_slot) * 8 + xcd; }
};
static inline int worklist_grid_b(int n_pairs, int n_tiles, int wg_per_cu) {
    const int halves = (n_tiles + 1) / 2;
    int per_xcd = (n_pairs + 7) / 8;
    int rounds_cap = (32 * wg_per_cu) / halves;          // 32 CUs per XCD
    if (rounds_cap < 1) rounds_cap = 1;
    const int ppr = per_xcd < rounds_cap ? per_xcd : rounds_cap;
    return 8 * ppr * halves;
}

// ---- staging: [NR rows][64 columns] bf16 tile = NR * 8 chunks of 16 bytes; thread -> chunks tid + 256 * jj --------
// Rows at or past r_end are read from the last valid row and stored as zeros (clamped address, no branch at the load).
template <int NR>
__device__ __forceinline__ void load_rows(const bf16_t* __restrict__ base, int64_t ld, int r0, int r_end, int tid,
                                          uint4 (&rg)[NR / 32]) {
#pragma unroll
    for (int jj = 0; jj < NR / 32; ++jj) {
        const int f = tid + AB_THREADS * jj;
        const int row = min(r0 + (f >> 3), r_end - 1);
        rg[jj] = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + ((f & 7) << 3));
    }
}
template <int NR>
__device__ __forceinline__ void store_rows(bf16_t* __restrict__ lds, int tid, const uint4 (&rg)[NR / 32], int n_valid) {
#pragma unroll
    for (int jj = 0; jj < NR / 32; ++jj) {
        const int f = tid + AB_THREADS * jj;
        const int row = f >> 3;
        uint4 v = rg[jj];
        if (row >= n_valid) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(lds + lds_off(row, (f & 7) << 3)) = v;
    }
}
// Per-lane element offsets of the two read patterns inside a 32-row image; every other row block / buffer is a
// compile-time constant away (lds_f ignores row bits 0 and >= 4, so +16 / +32 rows are plain offsets), which keeps the
// address registers of a kernel at 8 instead of one per (k-step, block).
struct LdsOffsets {
    int row[4];        // row read:  row (lane & 31), columns 16 s + 8 h .. + 7        (s = 0..3)
    int tr[2][2];      // transposing read [c][db]: row 8 c + 4 h + q4, column 32 db + 16 gsel + 4 pp
    __device__ __forceinline__ LdsOffsets(int lane) {
        const int r = lane & 31, h = lane >> 5, gsel = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int s = 0; s < 4; ++s) row[s] = lds_off(r, 16 * s + 8 * h);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int db = 0; db < 2; ++db) tr[c][db] = lds_off(8 * c + 4 * h + q4, 32 * db + 16 * gsel + 4 * pp);
    }
};
// row read: the 8 k-values (columns 16 s + 8 h ..) of row rblk + (lane & 31); rblk a multiple of 32
__device__ __forceinline__ bf16x8 read_row8(const bf16_t* __restrict__ img, const LdsOffsets& lo, int rblk, int s) {
    return *reinterpret_cast<const bf16x8*>(img + rblk * 64 + lo.row[s]);
}
// transposing read: for the MFMA whose k index runs over ROWS of the image.  Returns, for column 32 db + (lane & 31),
// the 8 rows rbase + 8 * (j >> 2) + 4 * h + (j & 3), j = 0..7 (rbase a multiple of 16) - the k order of an accumulator
// tile used as operand.
__device__ __forceinline__ bf16x8 read_tr8(const bf16_t* __restrict__ img, const LdsOffsets& lo, int rbase, int db) {
    bf16x8 out;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (bf16x4 __attribute__((address_space(3)))*)(img + rbase * 64 + lo.tr[c][db]));
        out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
    }
    return out;
}
// Two fp32 -> one register of two bf16 (round to nearest even).  Spelled as the instruction: left to the compiler, the
// conversion of a vector that has just gone through a select came out as one v_cvt_pk_bf16_f32 PER ELEMENT (upper half
// unused), the select on the 16-bit result and a v_perm_b32 per pair - 40 vector instructions per 16 probabilities where
// 8 do the work, in kernels whose time is vector-ALU time (docs/DESIGN_rounds1-4.md section 11).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s2) {
    u32x4b u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = cvt_pk_bf16(a[8 * s2 + 2 * j], a[8 * s2 + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, u);
}
// x[i] = exp2(x[i] * c + d) for the 16 registers of a tile, the multiply-adds as 8 packed v_pk_fma_f32
__device__ __forceinline__ void exp2_affine16(f32x16& x, const float c, const float d) {
    const f32x2 cv = {c, c}, dv = {d, d};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x2 t = {x[2 * j], x[2 * j + 1]};
        t = __builtin_elementwise_fma(t, cv, dv);
        x[2 * j] = __builtin_amdgcn_exp2f(t[0]);
        x[2 * j + 1] = __builtin_amdgcn_exp2f(t[1]);
    }
}
__device__ __forceinline__ float sum16(const f32x16& x) {
    f32x2 a = {x[0], x[1]}, b = {x[2], x[3]};
#pragma unroll
    for (int j = 2; j < 8; j += 2) {
        a += (f32x2){x[2 * j], x[2 * j + 1]};
        b += (f32x2){x[2 * j + 2], x[2 * j + 3]};
    }
    a += b;
    return a[0] + a[1];
}

// per staged 64-key tile: key levels, dropout key words, the largest level of each 32-key half
struct KeyMetaB {
    int32_t kl[64];
    uint32_t kw[64];
    int32_t klmax[2];
    int32_t pad_[2];
};
template <bool DROP>
__device__ __forceinline__ int key_meta_load_b(const int32_t* __restrict__ klb, int j0, int S, int w, int lane,
                                               const AttnDropout& rng) {
    const int j = j0 + lane;
    int v = 0;
    if (w == 0 || w == 1) v = j < S ? klb[j] : INT_BIG_B;
    else if (w == 2) { if (DROP) v = (int)rng.key_word((uint32_t)j); }
    return v;
}
__device__ __forceinline__ void key_meta_store_b(KeyMetaB& km, int w, int lane, int v) {
    if (w == 0) km.kl[lane] = v;
    else if (w == 1) {
        // max over each 32-lane half
        int m = v;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if ((lane & 31) == 0) km.klmax[lane >> 5] = m;
    } else if (w == 2) km.kw[lane] = (uint32_t)v;
}
__device__ __forceinline__ void read_key_quads_b(const int32_t* __restrict__ base, int h, int (&out)[16]) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int4 t = *reinterpret_cast<const int4*>(base + 8 * g4 + 4 * h);
        out[4 * g4] = t.x; out[4 * g4 + 1] = t.y; out[4 * g4 + 2] = t.z; out[4 * g4 + 3] = t.w;
    }
}

// =============================================================================================
// forward
// =============================================================================================
template <int G, bool DROP, bool SPAN, bool ORD>
__device__ __forceinline__ void
attn_fwd_b_tile(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                const bf16_t* __restrict__ v, int ldv, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, bf16_t* __restrict__ o,
                float* __restrict__ lse, const int32_t* __restrict__ span, const QOrdB ord, const int pair, const int qtile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][64 * 64];
    __shared__ __attribute__((aligned(16))) KeyMetaB kms[2];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int iq = q0 + sub * 32 + r;                 // sorted slot (= position without a row order)
    const bool valid_q = iq < S;
    const int iqc = ORD ? ord.perm[(int64_t)b * S + (valid_q ? iq : S - 1)] : (valid_q ? iq : S - 1);
    const int64_t tok = (int64_t)b * S + iqc;

    bf16x8 qf[4];
    {
        const bf16_t* qrow = q + tok * ldq + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
    }
    const int my_ql = ql ? ql[tok] : 1;
    QuerySpanB sp = QuerySpanB::load<SPAN>(span, tok, iqc, valid_q);
    if (ORD && valid_q && ord.row_empty[tok] != 0) sp.hi = -1;
    const int wave_q_hi = wave_max_i32_dpp(sp.hi);
    const int wave_q_lo = wave_min_i32_dpp(valid_q ? sp.hi : INT_BIG_B);
    const int wave_ql_min = wave_min_i32_dpp(valid_q ? my_ql : INT_BIG_B);
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(valid_q ? sp.hole_lo : INT_BIG_B) : INT_BIG_B;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(valid_q ? sp.hole_hi : 0) : 0;
    int n_iter = (min(S, q0 + R) + 63) / 64;                    // allowed keys are <= the query position
    if (ORD) {                                                  // ... = the largest position of a normal row of the tile
        const int n32 = (S + 31) >> 5;
        int mp = -1;
#pragma unroll
        for (int t = 0; t < NSUB; ++t)
            if (qtile * NSUB + t < n32) mp = max(mp, ord.tile_maxpos[(int64_t)b * n32 + qtile * NSUB + t]);
        n_iter = mp < 0 ? 0 : (min(S, mp + 1) + 63) / 64;
    }
    const float c2 = scale * 1.4426950408889634f;               // scores in the log2 domain
    const AttnDropout rng(p_drop, seed);
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;

    const bf16_t* kbase = k + (int64_t)b * S * ldk + kvh * 64;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + kvh * 64;
    const int32_t* klb = kl + (int64_t)b * S;

    float m_ref = 0.f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }

    uint4 rk[2], rv[2];
    int rmeta = 0;
    __syncthreads();                                             // the previous item of this workgroup is done with LDS
    if (n_iter > 0) {
        load_rows<64>(kbase, ldk, 0, S, tid, rk);
        load_rows<64>(vbase, ldv, 0, S, tid, rv);
        rmeta = key_meta_load_b<DROP>(klb, 0, S, w, lane, rng);
        store_rows<64>(Ks[0], tid, rk, S);
        store_rows<64>(Vs[0], tid, rv, S);
        key_meta_store_b(kms[0], w, lane, rmeta);
    }
    __syncthreads();

    for (int jt = 0; jt < n_iter; ++jt) {
        const int cur = jt & 1;
        const bool more = jt + 1 < n_iter;
        if (more && !(AB_ABLATE & 1)) {
            load_rows<64>(kbase, ldk, jt * 64 + 64, S, tid, rk);
            load_rows<64>(vbase, ldv, jt * 64 + 64, S, tid, rv);
            rmeta = key_meta_load_b<DROP>(klb, jt * 64 + 64, S, w, lane, rng);
        }
        const KeyMetaB& km = kms[cur];
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            const int j0 = jt * 64 + kb * 32;
            if (j0 > wave_q_hi) continue;                        // every key of the sub-tile is in every row's future
            f32x16 st;
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(Ks[cur], lo, 32 * kb, s), qf[s], st, 0, 0, 0);
            const int klmax = __builtin_amdgcn_readfirstlane(km.klmax[kb]);
            const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                   (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
            if (!free_tile) {
                int klv[16];
                read_key_quads_b(km.kl + 32 * kb, h, klv);
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
            float mloc = st[0];
#pragma unroll
            for (int reg = 1; reg < 16; ++reg) mloc = fmaxf(mloc, st[reg]);
            mloc = xor32_max(mloc) * c2;
            const bool need = (l_run == 0.f) ? (mloc > -INFINITY) : (mloc > m_ref + RESCALE_TAU_B);
            if (__any(need ? 1 : 0)) {
                const float alpha = (need && l_run != 0.f) ? __builtin_amdgcn_exp2f(m_ref - mloc) : 1.f;
                m_ref = need ? mloc : m_ref;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
            }
            exp2_affine16(st, c2, -m_ref);                          // masked: exp2(-inf) = 0
            float rowsum = sum16(st);
            rowsum = xor32_sum(rowsum);
            l_run += rowsum;
            if (DROP) {
                int kwv[16];
                read_key_quads_b(reinterpret_cast<const int32_t*>(km.kw) + 32 * kb, h, kwv);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) st[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? st[reg] : 0.f;
            }
            // O^T[d][query] += sum_key V[key][d] * P[query][key]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(st, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(Vs[cur], lo, 32 * kb + 16 * s2, db), pf,
                                                                       oacc[db], 0, 0, 0);
            }
        }
        if (more) {
            store_rows<64>(Ks[cur ^ 1], tid, rk, S - (jt * 64 + 64));
            store_rows<64>(Vs[cur ^ 1], tid, rv, S - (jt * 64 + 64));
            key_meta_store_b(kms[cur ^ 1], w, lane, rmeta);
        }
        __syncthreads();
    }

    if (valid_q) {
        float linv = l_run > 0.f ? 1.f / l_run : 0.f;
        if (DROP) linv *= rng.scale;
        bf16_t* orow = o + tok * (int64_t)nq * 64 + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t4 = make_float4(oacc[db][4 * g4] * linv, oacc[db][4 * g4 + 1] * linv,
                                              oacc[db][4 * g4 + 2] * linv, oacc[db][4 * g4 + 3] * linv);
                st4(orow + 32 * db + 8 * g4 + 4 * h, t4);
            }
        // natural-log LSE of the scaled scores; 0 for a row without an allowed key
        if (h == 0) lse[((int64_t)b * nq + head) * S + iqc] =
            l_run > 0.f ? (m_ref + __log2f(l_run)) * 0.6931471805599453f : 0.f;
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(AB_THREADS, AB_OCC_FWD)
attn_fwd_b_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                  const bf16_t* __restrict__ v, int ldv, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                  int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                  bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ span, const QOrdB ord) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkListB wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        // (with a row order the tiles are not a causal ramp: every workgroup takes another slot u from pair to pair)
        const int uu = ORD ? (wl.u + it) % ((n_tiles + 1) >> 1) : wl.u;
        const int heavy = n_tiles - 1 - uu, light = uu;           // later query tiles see more keys
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_fwd_b_tile<G, DROP, SPAN, ORD>(q, ldq, k, ldk, v, ldv, kl, ql, S, nq, nkv, scale, p_drop, seed, o, lse, span,
                                                ord, pair, pass == 0 ? heavy : light);
        }
    }
}

// =============================================================================================
// forward with the K and V rows of a (sequence, kv head) RESIDENT in LDS (the structure of csrc/attention_res.hip: one workgroup of
// eight waves per pair, every wave takes 32-query row tiles of the pair's query heads from a queue and walks the keys with no
// barrier and no global load in the loop).  In bf16 the whole sequence fits: S <= 576 keys x 128 bytes x (K + V) = 144 KB; longer
// sequences stay on the tiled kernel above.  The arithmetic of a row tile is the tiled kernel's, 32-key sub-tile by sub-tile in
// the same order: results are bit-identical.
// =============================================================================================
#ifdef BR_THREADS_OVERRIDE
constexpr int BR_THREADS = BR_THREADS_OVERRIDE;
#else
constexpr int BR_THREADS = 512;
#endif
constexpr int BR_MAXKEYS = 576;
struct BrSmem {
    bf16_t K[BR_MAXKEYS * 64];
    bf16_t V[BR_MAXKEYS * 64];
    int32_t kl[BR_MAXKEYS];
    uint32_t kw[BR_MAXKEYS];
    int32_t klmax[BR_MAXKEYS / 32];
    int32_t ctr;
    int32_t pad_[5];
};
static inline bool br_enabled() {
    static EnvSwitch sw("GAMER_ATTN_RES");             // (the switch of the fp32 resident kernels: A/B runs)
    return sw.get(1) != 0;
}
#define GAMER_BR_PART(name) ([]() -> bool { static EnvSwitch sw(name); return sw.get(1) != 0; }())
static inline int br_grid_cap() {
    static EnvSwitch grid("GAMER_ATTN_RES_GRID");       // (tests force a few persistent workgroups to walk many units)
    if (grid.get(0) > 0) return grid.get(0);
    static int cap[MAX_DEVICES] = {};
    int& c = cap[current_device()];
    if (c == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess) n = 256;
        c = n > 0 ? n : 256;
    }
    return c;
}
// one query head per workgroup when whole pairs do not fill the CUs evenly (see res_split in attention_split_common.h)
static inline int br_split(int n_pairs, int ways) {
    if (ways < 2) return 0;
    static EnvSwitch sp("GAMER_ATTN_RES_SPLIT");
    if (sp.is_set()) return sp.get(0) != 0;
    const int ncu = br_grid_cap();
    const double whole = (double)((n_pairs + ncu - 1) / ncu), halves = 0.58 * (double)((ways * n_pairs + ncu - 1) / ncu);
    return halves < whole ? 1 : 0;
}

template <bool DROP>
__device__ __forceinline__ void br_stage(BrSmem& sm, const bf16_t* __restrict__ kbase, int ldk, const bf16_t* __restrict__ vbase,
                                         int ldv, const int32_t* __restrict__ klb, int S, int tid, const AttnDropout& rng) {
    const int n_rows = ((S + 31) >> 5) << 5;
    // chunks of 16 bytes: n_rows x 8 per tensor; eight of each tensor in flight per thread and pass
#pragma unroll 1
    for (int c0 = 0; c0 < n_rows * 8; c0 += BR_THREADS * 8) {
        uint4 rk[8], rv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = c0 + tid + BR_THREADS * p;
            const int row = min(f >> 3, S - 1);
            rk[p] = *reinterpret_cast<const uint4*>(kbase + (int64_t)row * ldk + ((f & 7) << 3));
            rv[p] = *reinterpret_cast<const uint4*>(vbase + (int64_t)row * ldv + ((f & 7) << 3));
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int f = c0 + tid + BR_THREADS * p;
            const int row = f >> 3;
            if (row < n_rows) {
                const bool ok = row < S;
                const int off = lds_off(row, (f & 7) << 3);
                *reinterpret_cast<uint4*>(sm.K + off) = ok ? rk[p] : make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(sm.V + off) = ok ? rv[p] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    for (int j = tid; j < n_rows; j += BR_THREADS) {
        const int v = j < S ? klb[j] : INT_BIG_B;
        sm.kl[j] = v;
        if (DROP) sm.kw[j] = rng.key_word((uint32_t)j);
        int m = v;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if ((j & 31) == 0) sm.klmax[j >> 5] = m;
    }
}

// one 32-query row tile of one query head against the resident keys
template <bool DROP, bool SPAN, bool ORD>
__device__ __forceinline__ void
br_fwd_rowtile(const BrSmem& sm, const bf16_t* __restrict__ q, int ldq, const int32_t* __restrict__ ql, int S, int nq, float scale,
               const AttnDropout& rng, bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ span,
               const QOrdB& ord, const int b, const int head, const int t) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int iq = t * 32 + r;                        // sorted slot (= position without a row order)
    const bool valid_q = iq < S;
    const int iqc = ORD ? ord.perm[(int64_t)b * S + (valid_q ? iq : S - 1)] : (valid_q ? iq : S - 1);
    const int64_t tok = (int64_t)b * S + iqc;

    bf16x8 qf[4];
    {
        const bf16_t* qrow = q + tok * ldq + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
    }
    const int my_ql = ql ? ql[tok] : 1;
    QuerySpanB sp = QuerySpanB::load<SPAN>(span, tok, iqc, valid_q);
    if (ORD && valid_q && ord.row_empty[tok] != 0) sp.hi = -1;
    const int wave_q_hi = wave_max_i32_dpp(sp.hi);
    const int wave_q_lo = wave_min_i32_dpp(valid_q ? sp.hi : INT_BIG_B);
    const int wave_ql_min = wave_min_i32_dpp(valid_q ? my_ql : INT_BIG_B);
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(valid_q ? sp.hole_lo : INT_BIG_B) : INT_BIG_B;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(valid_q ? sp.hole_hi : 0) : 0;
    const int n_sub = wave_q_hi < 0 ? 0 : (min(S - 1, wave_q_hi) >> 5) + 1;     // 32-key sub-tiles up to the last allowed key
    const float c2 = scale * 1.4426950408889634f;               // scores in the log2 domain
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;

    float m_ref = 0.f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }

#pragma unroll 1
    for (int js = 0; js < n_sub; ++js) {
        const int j0 = js * 32;
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(sm.K, lo, j0, s), qf[s], st, 0, 0, 0);
        const int klmax = __builtin_amdgcn_readfirstlane(sm.klmax[js]);
        const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                               (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
        if (!free_tile) {
            int klv[16];
            read_key_quads_b(sm.kl + j0, h, klv);
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
        float mloc = st[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mloc = fmaxf(mloc, st[reg]);
        mloc = xor32_max(mloc) * c2;
        const bool need = (l_run == 0.f) ? (mloc > -INFINITY) : (mloc > m_ref + RESCALE_TAU_B);
        if (__any(need ? 1 : 0)) {
            const float alpha = (need && l_run != 0.f) ? __builtin_amdgcn_exp2f(m_ref - mloc) : 1.f;
            m_ref = need ? mloc : m_ref;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        }
        exp2_affine16(st, c2, -m_ref);                          // masked: exp2(-inf) = 0
        float rowsum = sum16(st);
        rowsum = xor32_sum(rowsum);
        l_run += rowsum;
        if (DROP) {
            int kwv[16];
            read_key_quads_b(reinterpret_cast<const int32_t*>(sm.kw) + j0, h, kwv);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) st[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? st[reg] : 0.f;
        }
        // O^T[d][query] += sum_key V[key][d] * P[query][key]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(st, s2);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(sm.V, lo, j0 + 16 * s2, db), pf, oacc[db], 0, 0, 0);
        }
    }

    if (valid_q) {
        float linv = l_run > 0.f ? 1.f / l_run : 0.f;
        if (DROP) linv *= rng.scale;
        bf16_t* orow = o + tok * (int64_t)nq * 64 + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t4 = make_float4(oacc[db][4 * g4] * linv, oacc[db][4 * g4 + 1] * linv,
                                              oacc[db][4 * g4 + 2] * linv, oacc[db][4 * g4 + 3] * linv);
                st4(orow + 32 * db + 8 * g4 + 4 * h, t4);
            }
        // natural-log LSE of the scaled scores; 0 for a row without an allowed key
        if (h == 0) lse[((int64_t)b * nq + head) * S + iqc] =
            l_run > 0.f ? (m_ref + __log2f(l_run)) * 0.6931471805599453f : 0.f;
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(BR_THREADS, 1)
attn_fwd_br_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk, const bf16_t* __restrict__ v, int ldv,
                   const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, int nbatch, int S, int nq, int nkv, float scale,
                   float p_drop, uint64_t seed, bf16_t* __restrict__ o, float* __restrict__ lse, const int32_t* __restrict__ span,
                   const QOrdB ord, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char br_raw[];
    BrSmem& sm = *reinterpret_cast<BrSmem*>(br_raw);
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_all = (S + 31) >> 5;
    const int hpu = split ? 1 : G, upp = G / hpu;
    const int n_items = hpu * n_all;
    const int n_units = nbatch * nkv * upp;
#pragma unroll 1
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int b = (unit / upp) / nkv, kvh = (unit / upp) % nkv, h0 = (unit % upp) * hpu;
        __syncthreads();                                        // every wave is done with the previous unit's images
        if (tid == 0) sm.ctr = 0;
        br_stage<DROP>(sm, k + (int64_t)b * S * ldk + kvh * 64, ldk, v + (int64_t)b * S * ldv + kvh * 64, ldv, kl + (int64_t)b * S, S,
                       tid, rng);
        __syncthreads();
        // row tiles from the unit's queue, late (long) tiles first
#pragma unroll 1
        for (;;) {
            int cur = 0;
            if (lane == 0) cur = atomicAdd(&sm.ctr, 1);
            cur = __builtin_amdgcn_readfirstlane(cur);
            if (cur >= n_items) break;
            br_fwd_rowtile<DROP, SPAN, ORD>(sm, q, ldq, ql, S, nq, scale, rng, o, lse, span, ord, b, kvh * G + h0 + cur % hpu,
                                            n_all - 1 - cur / hpu);
        }
    }
}

// resident-K/V dQ (attn_bwd_dq_b_tile's arithmetic, sub-tile by sub-tile in the same order: bit-identical results)
template <bool DROP, bool SPAN, bool ORD>
__device__ __forceinline__ void
br_dq_rowtile(const BrSmem& sm, const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
              const float* __restrict__ delta, const int32_t* __restrict__ ql, int S, int nq, float scale, const AttnDropout& rng,
              bf16_t* __restrict__ dq, int lddq, const int32_t* __restrict__ span, const QOrdB& ord, const int b, const int head,
              const int t) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int iq = t * 32 + r;
    const bool valid_q = iq < S;
    const int iqc = ORD ? ord.perm[(int64_t)b * S + (valid_q ? iq : S - 1)] : (valid_q ? iq : S - 1);
    const int64_t tok = (int64_t)b * S + iqc;

    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qrow = q + tok * ldq + head * 64 + 8 * h;
        const bf16_t* drow = d_o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8*>(drow + 16 * s);
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    QuerySpanB sp = QuerySpanB::load<SPAN>(span, tok, iqc, valid_q);
    if (ORD && valid_q && ord.row_empty[tok] != 0) sp.hi = -1;
    const float neg_lse2 = -lse[((int64_t)b * nq + head) * S + iqc] * 1.4426950408889634f;
    const float neg_delta = -delta[((int64_t)b * nq + head) * S + iqc];
    const int wave_q_hi = wave_max_i32_dpp(sp.hi);
    const int wave_q_lo = wave_min_i32_dpp(valid_q ? sp.hi : INT_BIG_B);
    const int wave_ql_min = wave_min_i32_dpp(valid_q ? my_ql : INT_BIG_B);
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(valid_q ? sp.hole_lo : INT_BIG_B) : INT_BIG_B;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(valid_q ? sp.hole_hi : 0) : 0;
    const int n_sub = wave_q_hi < 0 ? 0 : (min(S - 1, wave_q_hi) >> 5) + 1;
    const float c2 = scale * 1.4426950408889634f;
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;
    const float sd = rng.scale;

    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }

#pragma unroll 1
    for (int js = 0; js < n_sub; ++js) {
        const int j0 = js * 32;
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
        // S^T = K Q^T and dP^T[key][query] = sum_d V[key][d] dO[query][d]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(sm.K, lo, j0, s), qf[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(sm.V, lo, j0, s), dof[s], dp, 0, 0, 0);
        }
        const int klmax = __builtin_amdgcn_readfirstlane(sm.klmax[js]);
        const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                               (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
        if (!free_tile) {
            int klv[16];
            read_key_quads_b(sm.kl + j0, h, klv);
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
        // dS^T = p * (mult * dP - delta), p = exp2(c2 * s - lse2); a row without an allowed key has p = 0 everywhere
        if (DROP) {
            int kwv[16];
            read_key_quads_b(reinterpret_cast<const int32_t*>(sm.kw) + j0, h, kwv);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) dp[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? dp[reg] : 0.f;
        }
        exp2_affine16(st, c2, neg_lse2);
        {
            const f32x2 sdv = {DROP ? sd : 1.f, DROP ? sd : 1.f}, ndv = {neg_delta, neg_delta};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x2 tt = {dp[2 * j], dp[2 * j + 1]};
                tt = __builtin_elementwise_fma(tt, sdv, ndv);
                st[2 * j] *= tt[0];
                st[2 * j + 1] *= tt[1];
            }
        }
        // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 df = pack8(st, s2);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(sm.K, lo, j0 + 16 * s2, db), df, dqacc[db], 0, 0, 0);
        }
    }

    if (valid_q) {
        bf16_t* drow = dq + tok * lddq + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t4 = make_float4(dqacc[db][4 * g4] * scale, dqacc[db][4 * g4 + 1] * scale,
                                              dqacc[db][4 * g4 + 2] * scale, dqacc[db][4 * g4 + 3] * scale);
                st4(drow + 32 * db + 8 * g4 + 4 * h, t4);
            }
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(BR_THREADS, 1)
attn_bwd_dq_br_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk, const bf16_t* __restrict__ v,
                      int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                      const int32_t* __restrict__ kl, const int32_t* __restrict__ ql, int nbatch, int S, int nq, int nkv, float scale,
                      float p_drop, uint64_t seed, bf16_t* __restrict__ dq, int lddq, const int32_t* __restrict__ span,
                      const QOrdB ord, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char br_raw[];
    BrSmem& sm = *reinterpret_cast<BrSmem*>(br_raw);
    const AttnDropout rng(p_drop, seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_all = (S + 31) >> 5;
    const int hpu = split ? 1 : G, upp = G / hpu;
    const int n_items = hpu * n_all;
    const int n_units = nbatch * nkv * upp;
#pragma unroll 1
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int b = (unit / upp) / nkv, kvh = (unit / upp) % nkv, h0 = (unit % upp) * hpu;
        __syncthreads();
        if (tid == 0) sm.ctr = 0;
        br_stage<DROP>(sm, k + (int64_t)b * S * ldk + kvh * 64, ldk, v + (int64_t)b * S * ldv + kvh * 64, ldv, kl + (int64_t)b * S, S,
                       tid, rng);
        __syncthreads();
#pragma unroll 1
        for (;;) {
            int cur = 0;
            if (lane == 0) cur = atomicAdd(&sm.ctr, 1);
            cur = __builtin_amdgcn_readfirstlane(cur);
            if (cur >= n_items) break;
            br_dq_rowtile<DROP, SPAN, ORD>(sm, q, ldq, d_o, lse, delta, ql, S, nq, scale, rng, dq, lddq, span, ord, b,
                                           kvh * G + h0 + cur % hpu, n_all - 1 - cur / hpu);
        }
    }
}

// (A resident dK / dV in the same style - Q and dO of one head resident, four waves owning four key tiles each, the two heads
// added in registers - was built and measured SLOWER than the tiled kernel below: 2.40 against 2.16 ms per backward at batch 1024;
// 256 accumulator registers per wave leave one wave per SIMD with nothing to hide its LDS latency.  Not kept.)
// =============================================================================================
// backward: delta = dO . O
// =============================================================================================
__global__ void __launch_bounds__(AB_THREADS)
attn_delta_b_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, int B, int S, int nq,
                    float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 7, sub = lane >> 3;                     // 8 lanes x 8 elements = one head row, 8 rows per wave
    const int64_t wave = ((int64_t)blockIdx.x * AB_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * AB_THREADS) >> 6;
    const int64_t total = (int64_t)B * S * nq;
    for (int64_t i0 = wave * 8; i0 < total; i0 += nwaves * 8) {
        const int64_t i = i0 + sub;
        const bool live = i < total;
        const int64_t ic = live ? i : total - 1;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + ic * 64 + 8 * g);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(d_o + ic * 64 + 8 * g);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (live && g == 0) {
            const int64_t t = i / nq;
            const int hd = (int)(i % nq);
            const int64_t bb = t / S, ii = t % S;
            delta[(bb * nq + hd) * S + ii] = s;
        }
    }
}

// =============================================================================================
// backward: dQ
// =============================================================================================
template <int G, bool DROP, bool SPAN, bool ORD>
__device__ __forceinline__ void
attn_bwd_dq_b_tile(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                   const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                   const float* __restrict__ delta, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                   int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, bf16_t* __restrict__ dq, int lddq,
                   const int32_t* __restrict__ span, const QOrdB ord, const int pair, const int qtile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][64 * 64];
    __shared__ __attribute__((aligned(16))) KeyMetaB kms[2];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, q0 = qtile * R;
    const int head = kvh * G + hg;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int iq = q0 + sub * 32 + r;                 // sorted slot (= position without a row order)
    const bool valid_q = iq < S;
    const int iqc = ORD ? ord.perm[(int64_t)b * S + (valid_q ? iq : S - 1)] : (valid_q ? iq : S - 1);
    const int64_t tok = (int64_t)b * S + iqc;

    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qrow = q + tok * ldq + head * 64 + 8 * h;
        const bf16_t* drow = d_o + tok * (int64_t)nq * 64 + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8*>(drow + 16 * s);
        }
    }
    const int my_ql = ql ? ql[tok] : 1;
    QuerySpanB sp = QuerySpanB::load<SPAN>(span, tok, iqc, valid_q);
    if (ORD && valid_q && ord.row_empty[tok] != 0) sp.hi = -1;
    const float neg_lse2 = -lse[((int64_t)b * nq + head) * S + iqc] * 1.4426950408889634f;
    const float neg_delta = -delta[((int64_t)b * nq + head) * S + iqc];
    const int wave_q_hi = wave_max_i32_dpp(sp.hi);
    const int wave_q_lo = wave_min_i32_dpp(valid_q ? sp.hi : INT_BIG_B);
    const int wave_ql_min = wave_min_i32_dpp(valid_q ? my_ql : INT_BIG_B);
    const int wave_hole_lo = SPAN ? wave_min_i32_dpp(valid_q ? sp.hole_lo : INT_BIG_B) : INT_BIG_B;
    const int wave_hole_hi = SPAN ? wave_max_i32_dpp(valid_q ? sp.hole_hi : 0) : 0;
    int n_iter = (min(S, q0 + R) + 63) / 64;
    if (ORD) {                                                  // ... = the largest position of a normal row of the tile
        const int n32 = (S + 31) >> 5;
        int mp = -1;
#pragma unroll
        for (int t = 0; t < NSUB; ++t)
            if (qtile * NSUB + t < n32) mp = max(mp, ord.tile_maxpos[(int64_t)b * n32 + qtile * NSUB + t]);
        n_iter = mp < 0 ? 0 : (min(S, mp + 1) + 63) / 64;
    }
    const float c2 = scale * 1.4426950408889634f;
    const AttnDropout rng(p_drop, seed);
    const uint32_t aw = DROP ? rng.row_word((uint32_t)(((int64_t)b * nq + head) * S + iqc)) : 0u;
    const float sd = rng.scale;

    const bf16_t* kbase = k + (int64_t)b * S * ldk + kvh * 64;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + kvh * 64;
    const int32_t* klb = kl + (int64_t)b * S;

    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }

    uint4 rk[2], rv[2];
    int rmeta = 0;
    __syncthreads();
    if (n_iter > 0) {
        load_rows<64>(kbase, ldk, 0, S, tid, rk);
        load_rows<64>(vbase, ldv, 0, S, tid, rv);
        rmeta = key_meta_load_b<DROP>(klb, 0, S, w, lane, rng);
        store_rows<64>(Ks[0], tid, rk, S);
        store_rows<64>(Vs[0], tid, rv, S);
        key_meta_store_b(kms[0], w, lane, rmeta);
    }
    __syncthreads();

    for (int jt = 0; jt < n_iter; ++jt) {
        const int cur = jt & 1;
        const bool more = jt + 1 < n_iter;
        if (more) {
            load_rows<64>(kbase, ldk, jt * 64 + 64, S, tid, rk);
            load_rows<64>(vbase, ldv, jt * 64 + 64, S, tid, rv);
            rmeta = key_meta_load_b<DROP>(klb, jt * 64 + 64, S, w, lane, rng);
        }
        const KeyMetaB& km = kms[cur];
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            const int j0 = jt * 64 + kb * 32;
            if (j0 > wave_q_hi) continue;
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
            // S^T = K Q^T and dP^T[key][query] = sum_d V[key][d] dO[query][d]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(Ks[cur], lo, 32 * kb, s), qf[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(Vs[cur], lo, 32 * kb, s), dof[s], dp, 0, 0, 0);
            }
            const int klmax = __builtin_amdgcn_readfirstlane(km.klmax[kb]);
            const bool free_tile = (j0 + 31 <= wave_q_lo) && (klmax < wave_ql_min) &&
                                   (!SPAN || j0 + 31 < wave_hole_lo || j0 >= wave_hole_hi);
            if (!free_tile) {
                int klv[16];
                read_key_quads_b(km.kl + 32 * kb, h, klv);
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
            // dS^T = p * (mult * dP - delta), p = exp2(c2 * s - lse2); a row without an allowed key has p = 0 everywhere
            if (DROP) {
                int kwv[16];
                read_key_quads_b(reinterpret_cast<const int32_t*>(km.kw) + 32 * kb, h, kwv);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) dp[reg] = rng.keep(aw, (uint32_t)kwv[reg]) ? dp[reg] : 0.f;
            }
            exp2_affine16(st, c2, neg_lse2);
            {
                const f32x2 sdv = {DROP ? sd : 1.f, DROP ? sd : 1.f}, ndv = {neg_delta, neg_delta};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x2 t = {dp[2 * j], dp[2 * j + 1]};
                    t = __builtin_elementwise_fma(t, sdv, ndv);
                    st[2 * j] *= t[0];
                    st[2 * j + 1] *= t[1];
                }
            }
            // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 df = pack8(st, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(Ks[cur], lo, 32 * kb + 16 * s2, db), df,
                                                                        dqacc[db], 0, 0, 0);
            }
        }
        if (more) {
            store_rows<64>(Ks[cur ^ 1], tid, rk, S - (jt * 64 + 64));
            store_rows<64>(Vs[cur ^ 1], tid, rv, S - (jt * 64 + 64));
            key_meta_store_b(kms[cur ^ 1], w, lane, rmeta);
        }
        __syncthreads();
    }

    if (valid_q) {
        bf16_t* drow = dq + tok * lddq + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t4 = make_float4(dqacc[db][4 * g4] * scale, dqacc[db][4 * g4 + 1] * scale,
                                              dqacc[db][4 * g4 + 2] * scale, dqacc[db][4 * g4 + 3] * scale);
                st4(drow + 32 * db + 8 * g4 + 4 * h, t4);
            }
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(AB_THREADS, AB_OCC_DQ)
attn_bwd_dq_b_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                     const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                     const float* __restrict__ delta, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                     int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                     bf16_t* __restrict__ dq, int lddq, const int32_t* __restrict__ span, const QOrdB ord) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkListB wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int uu = ORD ? (wl.u + it) % ((n_tiles + 1) >> 1) : wl.u;
        const int heavy = n_tiles - 1 - uu, light = uu;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dq_b_tile<G, DROP, SPAN, ORD>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, S, nq, nkv, scale, p_drop, seed,
                                                   dq, lddq, span, ord, pair, pass == 0 ? heavy : light);
        }
    }
}

// =============================================================================================
// backward: dK, dV
// =============================================================================================
template <int G>
struct DkvSmemB {
    bf16_t Kt[(4 / G) * 32 * 64];  // this workgroup's keys / values (read as B operands every tile: 32 VGPRs less)
    bf16_t Vt[(4 / G) * 32 * 64];
    bf16_t Qs[2][G][32 * 64];
    bf16_t dOs[2][G][32 * 64];
    float nlse2_s[2][G][32];      // -lse * log2(e) of the staged query rows
    float ndelta_s[2][G][32];     // -delta
    uint32_t aw_s[2][G][32];      // dropout row words
    int32_t ql_s[2][32];
    int32_t pos_s[2][32];         // key limit of the staged query rows (QuerySpanB::hi; -1 past the end)
    int32_t hole_lo_s[2][32];
    int32_t hole_hi_s[2][32];
    int32_t qlmin[2];             // over the rows of the tile that are inside the sequence
    int32_t posmin[2];
    int32_t posmax[2];
    int32_t hole_lo_min[2];
    int32_t hole_hi_max[2];
};

template <int G, bool DROP, bool SPAN, bool ORD>
__device__ __forceinline__ void
attn_bwd_dkv_b_tile(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                    const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                    const float* __restrict__ delta, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                    int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, bf16_t* __restrict__ dk, int lddk,
                    bf16_t* __restrict__ dv, int lddv, const int32_t* __restrict__ span, const QOrdB ord, const int pair,
                    const int ktile) {
    constexpr int NSUB = 4 / G;
    constexpr int R = NSUB * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char dkvb_raw[];
    DkvSmemB<G>& sm = *reinterpret_cast<DkvSmemB<G>*>(dkvb_raw);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hg = w / NSUB, sub = w % NSUB;
    const int b = pair / nkv, kvh = pair % nkv, k0 = ktile * R;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int jk = k0 + sub * 32 + r;                 // this lane's key
    const bool valid_k = jk < S;
    const int jkc = valid_k ? jk : S - 1;
    const int64_t ktok = (int64_t)b * S + jkc;

    // this workgroup's K and V rows (MFMA B operands, key on the lane) staged once, zero past the end of the sequence
    __syncthreads();                                   // the previous item's head reduction is done with LDS
#pragma unroll
    for (int jj = 0; jj < R / 32; ++jj) {
        const int f = tid + AB_THREADS * jj;
        const int row = f >> 3;
        const int j = min(k0 + row, S - 1);
        uint4 a = *reinterpret_cast<const uint4*>(k + ((int64_t)b * S + j) * ldk + kvh * 64 + ((f & 7) << 3));
        uint4 c = *reinterpret_cast<const uint4*>(v + ((int64_t)b * S + j) * ldv + kvh * 64 + ((f & 7) << 3));
        if (k0 + row >= S) { a = make_uint4(0u, 0u, 0u, 0u); c = a; }
        *reinterpret_cast<uint4*>(sm.Kt + lds_off(row, (f & 7) << 3)) = a;
        *reinterpret_cast<uint4*>(sm.Vt + lds_off(row, (f & 7) << 3)) = c;
    }
    const int my_kl = valid_k ? kl[ktok] : INT_BIG_B;
    const int wave_kl_max = wave_max_i32_dpp(my_kl);
    const int wave_k_lo = k0 + sub * 32, wave_k_hi = wave_k_lo + 31;
    const float c2 = scale * 1.4426950408889634f;
    const AttnDropout rng(p_drop, seed);
    const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
    const float sd = rng.scale;
    const uint32_t drop_head0 = (uint32_t)(((int64_t)b * nq + kvh * G) * S);
    int n_qt = (S + 31) / 32;
    int qt_first = k0 >> 5;                            // earlier query tiles only hold queries before every key here
    if (ORD) {
        // sorted tiles: the normal rows come first in ascending position, so tile_maxpos is non-decreasing over them and
        // -1 behind them: visit [first tile that reaches this workgroup's first key, last tile with a normal row]
        const int32_t* mp = ord.tile_maxpos + (int64_t)b * n_qt;
        int first = 0, end = 0;
        while (end < n_qt && mp[end] >= 0) ++end;
        while (first < end && mp[first] < k0) ++first;
        qt_first = first;
        n_qt = end;
    }

    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

    // staged registers: per head one Q tile and one dO tile (32 rows x 128 bytes = 256 chunks: one uint4 per thread)
    uint4 rq[G], rdo[G];
    float rl = 0.f, rd = 0.f;
    uint32_t raw = 0;
    int rql = 0, rpos = -1, rhlo = INT_BIG_B, rhhi = 0;
    int rqlmin = INT_BIG_B, rposmin = INT_BIG_B, rposmax = -1, rhlomin = INT_BIG_B, rhhimax = 0;

    auto load_q_tile = [&](int qt) {
        const int i0 = qt * 32;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int hd = kvh * G + g;
            int row = min(i0 + (tid >> 3), S - 1);
            if (ORD) row = ord.perm[(int64_t)b * S + row];
            rq[g] = *reinterpret_cast<const uint4*>(q + ((int64_t)b * S + row) * ldq + hd * 64 + ((tid & 7) << 3));
            rdo[g] = *reinterpret_cast<const uint4*>(d_o + ((int64_t)b * S + row) * (int64_t)nq * 64 + hd * 64 + ((tid & 7) << 3));
        }
        if (w < G) {
            // per-query scalars: wave g takes head g, both lane halves compute the same row (lane & 31)
            const int g = w, row = lane & 31;
            const int i = i0 + row;
            const bool in = i < S;
            int ic = in ? i : S - 1;
            if (ORD) ic = ord.perm[(int64_t)b * S + ic];
            const int hd = kvh * G + g;
            rl = in ? -lse[((int64_t)b * nq + hd) * S + ic] * 1.4426950408889634f : 0.f;
            rd = in ? -delta[((int64_t)b * nq + hd) * S + ic] : 0.f;
            if (DROP) raw = rng.row_word(drop_head0 + (uint32_t)(g * S + ic));
            if (g == 0) {
                rql = in ? (ql ? ql[(int64_t)b * S + ic] : 1) : 0;
                const QuerySpanB qs = QuerySpanB::load<SPAN>(span, (int64_t)b * S + ic, ic, in);
                rpos = qs.hi;
                if (ORD && in && ord.row_empty[(int64_t)b * S + ic] != 0) rpos = -1;
                if (SPAN) { rhlo = qs.hole_lo; rhhi = qs.hole_hi; }
            }
        }
    };
    auto store_q_tile = [&](int qt, int buf) {
        const int n_valid = S - qt * 32;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int row = tid >> 3;
            uint4 a = rq[g], c = rdo[g];
            if (row >= n_valid) { a = make_uint4(0u, 0u, 0u, 0u); c = a; }
            *reinterpret_cast<uint4*>(sm.Qs[buf][g] + lds_off(row, (tid & 7) << 3)) = a;
            *reinterpret_cast<uint4*>(sm.dOs[buf][g] + lds_off(row, (tid & 7) << 3)) = c;
        }
        if (w == 0) {
            // the tile's statistics, here and not behind the loads: the values have arrived by now (the loads were issued a whole tile
            // ago), and the reductions are DPP steps - as ds_bpermute chains behind the loads they were ~2000 cycles in wave 0 for
            // which the other three waves waited at every tile's barrier
            const bool in = qt * 32 + (lane & 31) < S;
            rqlmin = wave_min_i32_dpp(in ? rql : INT_BIG_B);
            rposmin = wave_min_i32_dpp(in ? rpos : INT_BIG_B);
            rposmax = wave_max_i32_dpp(rpos);
            if (SPAN) {
                rhlomin = wave_min_i32_dpp(in ? rhlo : INT_BIG_B);
                rhhimax = wave_max_i32_dpp(in ? rhhi : 0);
            }
        }
        if (w < G && lane < 32) {
            const int g = w, row = lane;
            sm.nlse2_s[buf][g][row] = rl;
            sm.ndelta_s[buf][g][row] = rd;
            if (DROP) sm.aw_s[buf][g][row] = raw;
            if (g == 0) {
                sm.ql_s[buf][row] = rql; sm.pos_s[buf][row] = rpos;
                if (SPAN) { sm.hole_lo_s[buf][row] = rhlo; sm.hole_hi_s[buf][row] = rhhi; }
                if (row == 0) {
                    sm.qlmin[buf] = rqlmin; sm.posmin[buf] = rposmin; sm.posmax[buf] = rposmax;
                    if (SPAN) { sm.hole_lo_min[buf] = rhlomin; sm.hole_hi_max[buf] = rhhimax; }
                }
            }
        }
    };

    if (qt_first < n_qt) {
        load_q_tile(qt_first);
        store_q_tile(qt_first, 0);
    }
    __syncthreads();
    for (int qt = qt_first; qt < n_qt; ++qt) {
        const int cur = (qt - qt_first) & 1;
        const bool more = qt + 1 < n_qt;
        if (more && !(AB_ABLATE_DKV & 8)) load_q_tile(qt + 1);
        const int posmax = __builtin_amdgcn_readfirstlane(sm.posmax[cur]);
        if (posmax >= wave_k_lo) {                     // some query of the tile reaches this wave's first key
            const bf16_t* Qh = sm.Qs[cur][hg];
            const bf16_t* dOh = sm.dOs[cur][hg];
            f32x16 st, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
            // S[query][key] = sum_d Q[query][d] K[key][d],  dP[query][key] = sum_d dO[query][d] V[key][d]
            if (!(AB_ABLATE_DKV & 1))
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(Qh, lo, 0, s), read_row8(sm.Kt, lo, 32 * sub, s), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(dOh, lo, 0, s), read_row8(sm.Vt, lo, 32 * sub, s), dp, 0, 0, 0);
            }
            const int posmin = __builtin_amdgcn_readfirstlane(sm.posmin[cur]);
            const int qlmin = __builtin_amdgcn_readfirstlane(sm.qlmin[cur]);
            bool free_tile = (posmin >= wave_k_hi) && (wave_kl_max < qlmin) && (qt * 32 + 32 <= S);
            if (SPAN) {
                const int hl = __builtin_amdgcn_readfirstlane(sm.hole_lo_min[cur]);
                const int hh = __builtin_amdgcn_readfirstlane(sm.hole_hi_max[cur]);
                free_tile = free_tile && (wave_k_hi < hl || wave_k_lo >= hh);
            }
            // per element (query = register, key = lane): P -> st (for dV, 1/(1-p) applied at the end), dS -> dp (for dK)
            if (!(AB_ABLATE_DKV & 2))
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                __builtin_amdgcn_sched_barrier(0);       // one group of four queries at a time (register pressure)
                const int qb = 8 * g4 + 4 * h;
                const float4 d4 = *reinterpret_cast<const float4*>(&sm.ndelta_s[cur][hg][qb]);
                const float4 l4 = *reinterpret_cast<const float4*>(&sm.nlse2_s[cur][hg][qb]);
                const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
                const float nl[4] = {l4.x, l4.y, l4.z, l4.w};
                int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0}, hlo[4] = {0, 0, 0, 0}, hhi[4] = {0, 0, 0, 0};
                uint32_t awv[4] = {0, 0, 0, 0};
                if (!free_tile) {
                    const int4 q4 = *reinterpret_cast<const int4*>(&sm.ql_s[cur][qb]);
                    const int4 p4 = *reinterpret_cast<const int4*>(&sm.pos_s[cur][qb]);
                    qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
                    posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
                    if (SPAN) {
                        const int4 a4 = *reinterpret_cast<const int4*>(&sm.hole_lo_s[cur][qb]);
                        const int4 b4 = *reinterpret_cast<const int4*>(&sm.hole_hi_s[cur][qb]);
                        hlo[0] = a4.x; hlo[1] = a4.y; hlo[2] = a4.z; hlo[3] = a4.w;
                        hhi[0] = b4.x; hhi[1] = b4.y; hhi[2] = b4.z; hhi[3] = b4.w;
                    }
                }
                if (DROP) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&sm.aw_s[cur][hg][qb]);
                    awv[0] = a4.x; awv[1] = a4.y; awv[2] = a4.z; awv[3] = a4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int reg = 4 * g4 + e;
                    float pe = __builtin_amdgcn_exp2f(fmaf(st[reg], c2, nl[e]));
                    if (!free_tile) {
                        bool allowed = (jk <= posv[e]) & (my_kl < qlv[e]);
                        if (SPAN) allowed = allowed & !((jk >= hlo[e]) & (jk < hhi[e]));
                        pe = allowed ? pe : 0.f;
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
            // dV^T[d][key] += sum_query dO[query][d] Pd[query][key];  dK^T[d][key] += sum_query Q[query][d] dS[query][key]
            if (!(AB_ABLATE_DKV & 4))
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(st, s2);
                const bf16x8 df = pack8(dp, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(dOh, lo, 16 * s2, db), pf, dvacc[db], 0, 0, 0);
                    dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(Qh, lo, 16 * s2, db), df, dkacc[db], 0, 0, 0);
                }
            }
        }
        if (more && !(AB_ABLATE_DKV & 32)) store_q_tile(qt + 1, cur ^ 1);
        if (!(AB_ABLATE_DKV & 16)) __syncthreads();
    }

    // ---- sum the G query heads of this kv head through LDS, then store ---------------------------
    float* red = reinterpret_cast<float*>(dkvb_raw);             // [sub][key 32][132] floats (dK 64 | dV 64)
#pragma unroll
    for (int gsrc = G - 1; gsrc >= 1; --gsrc) {
        __syncthreads();
        if (hg == gsrc) {
            float* dst = red + (sub * 32 + r) * 132;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * db + 8 * g4 + 4 * h;
                    *reinterpret_cast<float4*>(dst + d) = make_float4(dkacc[db][4 * g4], dkacc[db][4 * g4 + 1],
                                                                      dkacc[db][4 * g4 + 2], dkacc[db][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(dst + 64 + d) = make_float4(dvacc[db][4 * g4], dvacc[db][4 * g4 + 1],
                                                                           dvacc[db][4 * g4 + 2], dvacc[db][4 * g4 + 3]);
                }
        }
        __syncthreads();
        if (hg == 0) {
            const float* src = red + (sub * 32 + r) * 132;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * db + 8 * g4 + 4 * h;
                    const float4 a = *reinterpret_cast<const float4*>(src + d);
                    const float4 c = *reinterpret_cast<const float4*>(src + 64 + d);
                    dkacc[db][4 * g4] += a.x; dkacc[db][4 * g4 + 1] += a.y; dkacc[db][4 * g4 + 2] += a.z; dkacc[db][4 * g4 + 3] += a.w;
                    dvacc[db][4 * g4] += c.x; dvacc[db][4 * g4 + 1] += c.y; dvacc[db][4 * g4 + 2] += c.z; dvacc[db][4 * g4 + 3] += c.w;
                }
        }
    }
    if (hg == 0 && valid_k) {
        bf16_t* dkrow = dk + ktok * lddk + kvh * 64;
        bf16_t* dvrow = dv + ktok * lddv + kvh * 64;
        const float vs = DROP ? sd : 1.f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * db + 8 * g4 + 4 * h;
                st4(dkrow + d, make_float4(dkacc[db][4 * g4] * scale, dkacc[db][4 * g4 + 1] * scale,
                                           dkacc[db][4 * g4 + 2] * scale, dkacc[db][4 * g4 + 3] * scale));
                st4(dvrow + d, make_float4(dvacc[db][4 * g4] * vs, dvacc[db][4 * g4 + 1] * vs,
                                           dvacc[db][4 * g4 + 2] * vs, dvacc[db][4 * g4 + 3] * vs));
            }
    }
}

template <int G, bool DROP, bool SPAN, bool ORD>
__global__ void __launch_bounds__(AB_THREADS, AB_OCC_DKV)
attn_bwd_dkv_b_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                      const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                      const float* __restrict__ delta, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                      int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                      bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv,
                      const int32_t* __restrict__ span, const QOrdB ord) {
    constexpr int R = (4 / G) * 32;
    const int n_tiles = (S + R - 1) / R;
    const WorkListB wl(nbatch * nkv, n_tiles);
    if (!wl.valid) return;
    for (int it = 0;; ++it) {
        const int pair = wl.pair_at(it);
        if (pair >= wl.n_pairs) break;
        const int heavy = wl.u, light = n_tiles - 1 - wl.u;       // earlier key tiles are seen by more queries
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && light == heavy) break;
            attn_bwd_dkv_b_tile<G, DROP, SPAN, ORD>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, S, nq, nkv, scale, p_drop,
                                                    seed, dk, lddk, dv, lddv, span, ord, pair, pass == 0 ? heavy : light);
        }
    }
}

// =============================================================================================
// backward: dK, dV with ONE QUERY HEAD'S Q / dO resident in LDS (S <= 512, no key spans)
//   The tiled kernel above re-stages a 32-query tile of Q / dO through registers and LDS for every 64 keys and synchronises its four
//   waves per tile; timing-only builds (tools/ablate_attn_bf16_dkv.sh) price that skeleton - loads, LDS stores, barriers, the K / V
//   prologue and the head reduction of 24,576 work items - at 0.86 of the kernel's 1.45 ms, the products and the element-wise work at
//   0.59.  Here a workgroup (8 waves, one per CU) takes a whole (sequence, kv head): per query head it stages all rows of Q and dO once
//   (2 x 64 KB, in the order of the row permutation if there is one), then every wave sweeps the query tiles for its TWO key tiles
//   (w and 15 - w: 17 tile visits per wave under the causal mask) with K / V fragments in registers and no barrier; the heads of the
//   group add into the same accumulators (no head reduction), dK / dV leave once per pair.
// =============================================================================================
constexpr int BH_THREADS = 512, BH_ROWS = 512, BH_TILES = BH_ROWS / 32;
struct BhSmem {
    bf16_t Q[BH_ROWS * 64];       // slot-major (slot = position, or the row permutation's slot), chunks swizzled as everywhere here
    bf16_t dO[BH_ROWS * 64];
    float nlse2[BH_ROWS];         // -lse * log2(e) of the slot's row, current head
    float ndelta[BH_ROWS];
    uint32_t aw[BH_ROWS];         // dropout row words, current head
    int32_t ql[BH_ROWS];          // (head independent)
    int32_t pos[BH_ROWS];         // key limit of the slot's row; -1: none (past the end, or an "empty" row)
    int32_t t_qlmin[BH_TILES], t_posmin[BH_TILES], t_posmax[BH_TILES];
};
static_assert(sizeof(BhSmem) <= 160 * 1024, "BhSmem must fit the CU's LDS");

template <int G, bool DROP, bool ORD>
__global__ void __launch_bounds__(BH_THREADS, 1)
attn_bwd_dkv_bh_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                       const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                       const float* __restrict__ delta, const int32_t* __restrict__ kl, const int32_t* __restrict__ ql,
                       int nbatch, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                       bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv, const QOrdB ord) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bh_raw[];
    BhSmem& sm = *reinterpret_cast<BhSmem*>(bh_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const LdsOffsets lo(lane);
    const int n_qt_all = (S + 31) >> 5;
    const float c2 = scale * 1.4426950408889634f;
    const AttnDropout rng(p_drop, seed);
    const float sd = rng.scale;

    for (int pair = blockIdx.x; pair < nbatch * nkv; pair += gridDim.x) {
        const int b = pair / nkv, kvh = pair % nkv;
        const int32_t* mp = ORD ? ord.tile_maxpos + (int64_t)b * n_qt_all : nullptr;
        int n_qt = n_qt_all;
        if (ORD) {                                         // sorted tiles: the tiles with a normal row come first
            int end = 0;
            while (end < n_qt_all && mp[end] >= 0) ++end;
            n_qt = end;
        }
        const int n_slots = n_qt * 32;

        // ONE set of accumulators: the tile being swept.  Heads alternate the order of the wave's two tiles (A B | B A | ...), so the
        // sums of the tile swept last continue into the next head; the other tile's sums wait in `park` (private memory, written and
        // read once per head boundary).  With both sets in registers (128 + the loop's ~130) the compiler kept three of the ACTIVE
        // tile's four accumulators in scratch and moved them in and out around every query tile: 2.1 ms against the tiled kernel's 1.45.
        f32x16 dkacc[2], dvacc[2];                         // [columns 32 db ..]
        float park_mem[64];
        float __attribute__((address_space(5)))* park = (float __attribute__((address_space(5)))*)park_mem;
        asm volatile("" : "+v"(park));                     // (the array stays in memory: plain 16-byte scratch stores / loads, one wait)

#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            const int hd = kvh * G + g;
            __syncthreads();                               // the previous head's (pair's) sweeps are done with the images
            // ---- stage the head's Q / dO rows and per-row scalars: two rounds of four 16-byte chunks per thread and operand.  (Issuing
            // the loads in front of the barrier - so that they fly while the slower waves finish - measured 2.27-2.30 against 1.90 ms:
            // the values that stay live across the barrier push the address arithmetic of the whole head loop into scratch.)
#pragma unroll 1
            for (int half = 0; half < ((AB_ABLATE_DKV & 8) ? 0 : 2); ++half) {
                uint4 rq[4], rdo[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int f = tid + BH_THREADS * (4 * half + jj);
                    const int slot = min(f >> 3, S - 1);   // (slots past the sequence re-read its last row: masked through pos = -1)
                    const int row = ORD ? ord.perm[(int64_t)b * S + slot] : slot;
                    rq[jj] = *reinterpret_cast<const uint4*>(q + ((int64_t)b * S + row) * ldq + hd * 64 + ((f & 7) << 3));
                    rdo[jj] = *reinterpret_cast<const uint4*>(d_o + ((int64_t)b * S + row) * (int64_t)nq * 64 + hd * 64 + ((f & 7) << 3));
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int f = tid + BH_THREADS * (4 * half + jj);
                    if ((f >> 3) < n_slots) {
                        *reinterpret_cast<uint4*>(sm.Q + lds_off(f >> 3, (f & 7) << 3)) = rq[jj];
                        *reinterpret_cast<uint4*>(sm.dO + lds_off(f >> 3, (f & 7) << 3)) = rdo[jj];
                    }
                }
            }
            if (tid < n_slots) {
                const int slot = tid;
                const bool in = slot < S;
                int ic = in ? slot : S - 1;
                if (ORD) ic = ord.perm[(int64_t)b * S + ic];
                sm.nlse2[slot] = (in && !(AB_ABLATE_DKV & 128)) ? -lse[((int64_t)b * nq + hd) * S + ic] * 1.4426950408889634f : 0.f;
                sm.ndelta[slot] = (in && !(AB_ABLATE_DKV & 128)) ? -delta[((int64_t)b * nq + hd) * S + ic] : 0.f;
                if (DROP) sm.aw[slot] = rng.row_word((uint32_t)(((int64_t)b * nq + hd) * S) + (uint32_t)ic);
                if (g == 0) {
                    const int rql = in ? (ql ? ql[(int64_t)b * S + ic] : 1) : 0;
                    int rpos = in ? ic : -1;
                    if (ORD && in && ord.row_empty[(int64_t)b * S + ic] != 0) rpos = -1;
                    sm.ql[slot] = rql;
                    sm.pos[slot] = rpos;
                    // the statistics of the wave's two tiles (a tile = two rows of 16 lanes)
                    int a = in ? rql : INT_BIG_B, c = in ? rpos : INT_BIG_B, d = rpos;
                    GAMER_DPP_ROW_REDUCE(min, a)
                    GAMER_DPP_ROW_REDUCE(min, c)
                    GAMER_DPP_ROW_REDUCE(max, d)
                    const int a0 = min(__builtin_amdgcn_readlane(a, 0), __builtin_amdgcn_readlane(a, 16));
                    const int a1 = min(__builtin_amdgcn_readlane(a, 32), __builtin_amdgcn_readlane(a, 48));
                    const int c0 = min(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 16));
                    const int c1 = min(__builtin_amdgcn_readlane(c, 32), __builtin_amdgcn_readlane(c, 48));
                    const int d0 = max(__builtin_amdgcn_readlane(d, 0), __builtin_amdgcn_readlane(d, 16));
                    const int d1 = max(__builtin_amdgcn_readlane(d, 32), __builtin_amdgcn_readlane(d, 48));
                    if (lane == 0) {
                        sm.t_qlmin[2 * w] = a0; sm.t_posmin[2 * w] = c0; sm.t_posmax[2 * w] = d0;
                        sm.t_qlmin[2 * w + 1] = a1; sm.t_posmin[2 * w + 1] = c1; sm.t_posmax[2 * w + 1] = d1;
                    }
                }
            }
            __syncthreads();

            // ---- the wave's two key tiles ------------------------------------------------------------------------------
#pragma unroll 1
            for (int step = 0; step < 2; ++step) {
                const int kk = (g & 1) ? 1 - step : step;
                const int kt = kk == 0 ? w : BH_TILES - 1 - w;
                const int k0 = kt * 32;
                if (k0 >= S) continue;                     // (then the wave has one tile: its sums simply continue from head to head)
                const bool two_tiles = (BH_TILES - 1 - w) * 32 < S;
                if (g == 0 || (step == 1 && !two_tiles)) {
                    if (g == 0) {
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int i = 0; i < 16; ++i) { dkacc[db][i] = 0.f; dvacc[db][i] = 0.f; }
                    }
                } else if (step == 1 && !(AB_ABLATE_DKV & 64)) {   // back to the tile that waited since the previous head
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { dkacc[db][i] = park[32 * db + i]; dvacc[db][i] = park[32 * db + 16 + i]; }
                }
                const int jk = k0 + r;
                const bool valid_k = jk < S;
                const int64_t ktok = (int64_t)b * S + (valid_k ? jk : S - 1);
                bf16x8 kf[4], vf[4];                       // B operands: the lane's key, columns 16 s + 8 h ..
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    uint4 a = make_uint4(0u, 0u, 0u, 0u), c = a;
                    if (!(AB_ABLATE_DKV & 32)) {
                        a = *reinterpret_cast<const uint4*>(k + ktok * ldk + kvh * 64 + 16 * s4 + 8 * h);
                        c = *reinterpret_cast<const uint4*>(v + ktok * ldv + kvh * 64 + 16 * s4 + 8 * h);
                    }
                    if (!valid_k) { a = make_uint4(0u, 0u, 0u, 0u); c = a; }
                    kf[s4] = __builtin_bit_cast(bf16x8, a);
                    vf[s4] = __builtin_bit_cast(bf16x8, c);
                }
                const int my_kl = (valid_k && !(AB_ABLATE_DKV & 32)) ? kl[ktok] : INT_BIG_B;
                const int wave_kl_max = wave_max_i32_dpp(my_kl);
                const uint32_t bw = DROP ? rng.key_word((uint32_t)jk) : 0u;
                int qt_first = kt;                         // earlier query tiles only hold queries before every key of the tile
                if (ORD) {
                    int first = 0;
                    while (first < n_qt && mp[first] < k0) ++first;
                    qt_first = first;
                }
#pragma unroll 1
                for (int qt = qt_first; qt < n_qt; ++qt) {
                    const int posmax = __builtin_amdgcn_readfirstlane(sm.t_posmax[qt]);
                    if (posmax < k0) continue;             // no query of the tile reaches the wave's first key
                    const bf16_t* Qh = sm.Q + qt * 32 * 64;
                    const bf16_t* dOh = sm.dO + qt * 32 * 64;
                    f32x16 st, dp;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
                    if (!(AB_ABLATE_DKV & 1))
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(Qh, lo, 0, s4), kf[s4], st, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_row8(dOh, lo, 0, s4), vf[s4], dp, 0, 0, 0);
                    }
                    const int posmin = __builtin_amdgcn_readfirstlane(sm.t_posmin[qt]);
                    const int qlmin = __builtin_amdgcn_readfirstlane(sm.t_qlmin[qt]);
                    const bool free_tile = (posmin >= k0 + 31) && (wave_kl_max < qlmin) && (qt * 32 + 32 <= S);
                    if (!(AB_ABLATE_DKV & 2))
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        __builtin_amdgcn_sched_barrier(0);   // one group of four queries at a time (register pressure)
                        const int qb = qt * 32 + 8 * g4 + 4 * h;
                        const float4 d4 = *reinterpret_cast<const float4*>(&sm.ndelta[qb]);
                        const float4 l4 = *reinterpret_cast<const float4*>(&sm.nlse2[qb]);
                        const float ndl[4] = {d4.x, d4.y, d4.z, d4.w};
                        const float nl[4] = {l4.x, l4.y, l4.z, l4.w};
                        int qlv[4] = {0, 0, 0, 0}, posv[4] = {0, 0, 0, 0};
                        uint32_t awv[4] = {0, 0, 0, 0};
                        if (!free_tile) {
                            const int4 q4 = *reinterpret_cast<const int4*>(&sm.ql[qb]);
                            const int4 p4 = *reinterpret_cast<const int4*>(&sm.pos[qb]);
                            qlv[0] = q4.x; qlv[1] = q4.y; qlv[2] = q4.z; qlv[3] = q4.w;
                            posv[0] = p4.x; posv[1] = p4.y; posv[2] = p4.z; posv[3] = p4.w;
                        }
                        if (DROP) {
                            const uint4 a4 = *reinterpret_cast<const uint4*>(&sm.aw[qb]);
                            awv[0] = a4.x; awv[1] = a4.y; awv[2] = a4.z; awv[3] = a4.w;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int reg = 4 * g4 + e;
                            float pe = __builtin_amdgcn_exp2f(fmaf(st[reg], c2, nl[e]));
                            if (!free_tile) {
                                const bool allowed = (jk <= posv[e]) & (my_kl < qlv[e]);
                                pe = allowed ? pe : 0.f;
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
                    if (!(AB_ABLATE_DKV & 4))
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 pf = pack8(st, s2);
                        const bf16x8 df = pack8(dp, s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(dOh, lo, 16 * s2, db), pf, dvacc[db], 0, 0, 0);
                            dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr8(Qh, lo, 16 * s2, db), df, dkacc[db], 0, 0, 0);
                        }
                    }
                }
                if (g == G - 1) {
                    // ---- dK, dV of the tile: every head of the group is in the sums ----
                    if (valid_k && !(AB_ABLATE_DKV & 16)) {
                        bf16_t* dkrow = dk + ktok * lddk + kvh * 64;
                        bf16_t* dvrow = dv + ktok * lddv + kvh * 64;
                        const float vs = DROP ? sd : 1.f;
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int d = 32 * db + 8 * g4 + 4 * h;
                                st4(dkrow + d, make_float4(dkacc[db][4 * g4] * scale, dkacc[db][4 * g4 + 1] * scale,
                                                           dkacc[db][4 * g4 + 2] * scale, dkacc[db][4 * g4 + 3] * scale));
                                st4(dvrow + d, make_float4(dvacc[db][4 * g4] * vs, dvacc[db][4 * g4 + 1] * vs,
                                                           dvacc[db][4 * g4 + 2] * vs, dvacc[db][4 * g4 + 3] * vs));
                            }
                    }
                } else if (step == 0 && two_tiles && !(AB_ABLATE_DKV & 64)) {
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { park[32 * db + i] = dkacc[db][i]; park[32 * db + 16 + i] = dvacc[db][i]; }
                }
            }
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------
template <int G>
static int launch_fwd_b(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv, const int32_t* kl,
                        const int32_t* ql, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                        bf16_t* o, float* lse, const int32_t* span, const QOrdB ord, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    if (S <= BR_MAXKEYS && br_enabled()) {
        // K / V of a (sequence, kv head) resident in LDS
        const int split = br_split(B * nkv, G);
        const int n_units = B * nkv * (split ? G : 1);
        const dim3 rgrid(n_units < br_grid_cap() ? n_units : br_grid_cap());
        const size_t shmem = sizeof(BrSmem);
#define GAMER_LAUNCH_FWD_BR(DROPV, SPANV, ORDV)                                                                             \
        do {                                                                                                                \
            static bool attr_dev[MAX_DEVICES] = {};                                                                         \
            if (!attr_dev[current_device()]) {                                                                              \
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_br_kernel<G, DROPV, SPANV, ORDV>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);          \
                if (e != hipSuccess) { set_error("gamer_attn_fwd_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; } \
                attr_dev[current_device()] = true;                                                                          \
            }                                                                                                               \
            hipLaunchKernelGGL((attn_fwd_br_kernel<G, DROPV, SPANV, ORDV>), rgrid, dim3(BR_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, \
                               kl, ql, B, S, nq, nkv, scale, p_drop, seed, o, lse, span, ord, split);                       \
        } while (0)
        if (ord.perm) { if (p_drop > 0.f) GAMER_LAUNCH_FWD_BR(true, false, true); else GAMER_LAUNCH_FWD_BR(false, false, true); }
        else if (p_drop > 0.f) { if (span) GAMER_LAUNCH_FWD_BR(true, true, false); else GAMER_LAUNCH_FWD_BR(true, false, false); }
        else { if (span) GAMER_LAUNCH_FWD_BR(false, true, false); else GAMER_LAUNCH_FWD_BR(false, false, false); }
#undef GAMER_LAUNCH_FWD_BR
        GAMER_CHECK_LAUNCH("gamer_attn_fwd_bf16/resident");
        return 0;
    }
    dim3 grid(worklist_grid_b(B * nkv, (S + R - 1) / R, AB_OCC_FWD));
#define GAMER_LAUNCH_FWD_B(DROPV, SPANV, ORDV)                                                                              \
    hipLaunchKernelGGL((attn_fwd_b_kernel<G, DROPV, SPANV, ORDV>), grid, dim3(AB_THREADS), 0, st, q, ldq, k, ldk, v, ldv,   \
                       kl, ql, B, S, nq, nkv, scale, p_drop, seed, o, lse, span, ord)
    if (ord.perm) { if (p_drop > 0.f) GAMER_LAUNCH_FWD_B(true, false, true); else GAMER_LAUNCH_FWD_B(false, false, true); }
    else if (p_drop > 0.f) { if (span) GAMER_LAUNCH_FWD_B(true, true, false); else GAMER_LAUNCH_FWD_B(true, false, false); }
    else { if (span) GAMER_LAUNCH_FWD_B(false, true, false); else GAMER_LAUNCH_FWD_B(false, false, false); }
#undef GAMER_LAUNCH_FWD_B
    GAMER_CHECK_LAUNCH("gamer_attn_fwd_bf16");
    return 0;
}

template <int G, bool DROP, bool SPAN, bool ORD>
static int launch_bwd_b_variant(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv,
                                const bf16_t* d_o, const float* lse, const float* delta, const int32_t* kl,
                                const int32_t* ql, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                                bf16_t* dq, int lddq, bf16_t* dk, int lddk, bf16_t* dv, int lddv, const int32_t* span,
                                const QOrdB ord, hipStream_t st) {
    constexpr int R = (4 / G) * 32;
    dim3 grid(worklist_grid_b(B * nkv, (S + R - 1) / R, 2));
    bool dkv_done = false;
    if constexpr (!SPAN) {
        // dK / dV with a query head's Q / dO resident in LDS: whole (sequence, kv head) pairs on persistent workgroups, one per CU;
        // taken when the pairs fill the CUs' rounds (384 pairs on 256 CUs - per-GPU batch 128 - would leave the second round half empty)
        const int pairs = B * nkv, cap = br_grid_cap();
        const int rounds = (pairs + cap - 1) / cap;
        const bool fills = (double)pairs >= 0.85 * (double)rounds * (double)cap || pairs <= cap;
        if (S <= BH_ROWS && br_enabled() && GAMER_BR_PART("GAMER_ATTN_RES_DKV_BF16") && fills) {
            const size_t hshmem = sizeof(BhSmem);
            static bool hattr_dev[MAX_DEVICES] = {};
            if (!hattr_dev[current_device()]) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_bh_kernel<G, DROP, ORD>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)hshmem);
                if (e != hipSuccess) {
                    set_error("gamer_attn_bwd_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                    return (int)e;
                }
                hattr_dev[current_device()] = true;
            }
            hipLaunchKernelGGL((attn_bwd_dkv_bh_kernel<G, DROP, ORD>), dim3(pairs < cap ? pairs : cap), dim3(BH_THREADS), hshmem, st, q, ldq,
                               k, ldk, v, ldv, d_o, lse, delta, kl, ql, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv, lddv, ord);
            GAMER_CHECK_LAUNCH("gamer_attn_bwd_bf16/dkv head-resident");
            dkv_done = true;
        }
    }
    if (!dkv_done) {
    size_t shmem = sizeof(DkvSmemB<G>);
    const size_t red_bytes = (size_t)R * 132 * sizeof(float);
    if (shmem < red_bytes) shmem = red_bytes;
    static bool attr_dev[MAX_DEVICES] = {};   // one set of flags per template instantiation, one flag per device
    bool& attr_set = attr_dev[current_device()];
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_b_kernel<G, DROP, SPAN, ORD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_bwd_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_dkv_b_kernel<G, DROP, SPAN, ORD>), grid, dim3(AB_THREADS), shmem, st, q, ldq, k, ldk, v, ldv, d_o,
                       lse, delta, kl, ql, B, S, nq, nkv, scale, p_drop, seed, dk, lddk, dv, lddv, span, ord);
    GAMER_CHECK_LAUNCH("gamer_attn_bwd_bf16/dkv");
    }
    if (S <= BR_MAXKEYS && br_enabled() && GAMER_BR_PART("GAMER_ATTN_RES_DQ")) {
        // dQ with K / V of a (sequence, kv head) resident in LDS
        const int split = br_split(B * nkv, G);
        const int n_units = B * nkv * (split ? G : 1);
        const size_t rshmem = sizeof(BrSmem);
        static bool rattr_dev[MAX_DEVICES] = {};
        if (!rattr_dev[current_device()]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_br_kernel<G, DROP, SPAN, ORD>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)rshmem);
            if (e != hipSuccess) {
                set_error("gamer_attn_bwd_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return (int)e;
            }
            rattr_dev[current_device()] = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_br_kernel<G, DROP, SPAN, ORD>), dim3(n_units < br_grid_cap() ? n_units : br_grid_cap()),
                           dim3(BR_THREADS), rshmem, st, q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, B, S, nq, nkv, scale, p_drop,
                           seed, dq, lddq, span, ord, split);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_bf16/dq resident");
        return 0;
    }
    hipLaunchKernelGGL((attn_bwd_dq_b_kernel<G, DROP, SPAN, ORD>), grid, dim3(AB_THREADS), 0, st, q, ldq, k, ldk, v, ldv, d_o, lse,
                       delta, kl, ql, B, S, nq, nkv, scale, p_drop, seed, dq, lddq, span, ord);
    GAMER_CHECK_LAUNCH("gamer_attn_bwd_bf16/dq");
    return 0;
}

template <int G>
static int launch_bwd_b(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv, const bf16_t* d_o,
                        const float* lse, const float* delta, const int32_t* kl, const int32_t* ql, int B, int S, int nq,
                        int nkv, float scale, float p_drop, uint64_t seed, bf16_t* dq, int lddq, bf16_t* dk, int lddk,
                        bf16_t* dv, int lddv, const int32_t* span, const QOrdB ord, hipStream_t st) {
#define GAMER_LAUNCH_BWD_B(DROPV, SPANV, ORDV)                                                                              \
    return launch_bwd_b_variant<G, DROPV, SPANV, ORDV>(q, ldq, k, ldk, v, ldv, d_o, lse, delta, kl, ql, B, S, nq, nkv, scale, \
                                                       p_drop, seed, dq, lddq, dk, lddk, dv, lddv, span, ord, st)
    if (ord.perm) { if (p_drop > 0.f) GAMER_LAUNCH_BWD_B(true, false, true); else GAMER_LAUNCH_BWD_B(false, false, true); }
    else if (p_drop > 0.f) { if (span) GAMER_LAUNCH_BWD_B(true, true, false); else GAMER_LAUNCH_BWD_B(true, false, false); }
    else { if (span) GAMER_LAUNCH_BWD_B(false, true, false); else GAMER_LAUNCH_BWD_B(false, false, false); }
#undef GAMER_LAUNCH_BWD_B
    return -1;      // not reached
}

}  // namespace gamer

using namespace gamer;

static int check_attn_b(const char* name, const void* q, const void* k, const void* v, const void* kl, int ldq, int ldk,
                        int ldv, int B, int S, int nq, int nkv, float p_drop) {
    GAMER_CHECK_ARG(q && k && v && kl, "%s: null pointer", name);
    GAMER_CHECK_ARG(B > 0 && S > 0 && nq > 0 && nkv > 0 && nq % nkv == 0, "%s: bad shape B=%d S=%d nq=%d nkv=%d", name, B, S, nq, nkv);
    const int G = nq / nkv;
    GAMER_CHECK_ARG(G == 1 || G == 2, "%s: GQA group %d not built (1 or 2)", name, G);
    GAMER_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldq >= nq * 64 && ldk >= nkv * 64 && ldv >= nkv * 64,
                    "%s: bad leading dims ldq=%d ldk=%d ldv=%d (multiples of 8)", name, ldq, ldk, ldv);
    GAMER_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v), "%s: q/k/v must be 16-byte aligned", name);
    GAMER_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p_drop=%f", name, p_drop);
    return 0;
}

extern "C" int gamer_attn_fwd_bf16(const gamer_bf16* q, int ldq, const gamer_bf16* k, int ldk, const gamer_bf16* v, int ldv,
                                   const int32_t* kl, const int32_t* ql, int B, int S, int nq, int nkv, float scale,
                                   float p_drop, uint64_t seed, gamer_bf16* o, float* lse, const int32_t* q_span,
                                   const int32_t* perm, const int32_t* tile_maxpos, const int32_t* row_empty, void* stream) {
    int rc = check_attn_b("gamer_attn_fwd_bf16", q, k, v, kl, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(o && lse && aligned16(o), "gamer_attn_fwd_bf16: null/unaligned output");
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_fwd_bf16: q_span must be 16-byte aligned");
    GAMER_CHECK_ARG(!perm || (tile_maxpos && row_empty && !q_span),
                    "gamer_attn_fwd_bf16: a row order needs perm, tile_maxpos and row_empty, and excludes q_span");
    const QOrdB ord{perm, tile_maxpos, row_empty};
    hipStream_t st = (hipStream_t)stream;
    if (nq / nkv == 1)
        return launch_fwd_b<1>((const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, kl, ql, B, S, nq, nkv,
                               scale, p_drop, seed, (bf16_t*)o, lse, q_span, ord, st);
    return launch_fwd_b<2>((const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, kl, ql, B, S, nq, nkv, scale,
                           p_drop, seed, (bf16_t*)o, lse, q_span, ord, st);
}

extern "C" int gamer_attn_bwd_bf16(const gamer_bf16* q, int ldq, const gamer_bf16* k, int ldk, const gamer_bf16* v, int ldv,
                                   const gamer_bf16* o, const gamer_bf16* d_o, const float* lse, const int32_t* kl,
                                   const int32_t* ql, int B, int S, int nq, int nkv, float scale, float p_drop,
                                   uint64_t seed, float* delta, gamer_bf16* dq, int lddq, gamer_bf16* dk, int lddk,
                                   gamer_bf16* dv, int lddv, const int32_t* q_span, int delta_ready, const int32_t* perm,
                                   const int32_t* tile_maxpos, const int32_t* row_empty, void* stream) {
    int rc = check_attn_b("gamer_attn_bwd_bf16", q, k, v, kl, ldq, ldk, ldv, B, S, nq, nkv, p_drop);
    if (rc) return rc;
    GAMER_CHECK_ARG(d_o && lse && delta && dq && dk && dv && (o || delta_ready), "gamer_attn_bwd_bf16: null pointer");
    GAMER_CHECK_ARG(lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && aligned16(d_o),
                    "gamer_attn_bwd_bf16: gradient leading dims must be multiples of 4, dO 16-byte aligned");
    GAMER_CHECK_ARG(!q_span || aligned16(q_span), "gamer_attn_bwd_bf16: q_span must be 16-byte aligned");
    GAMER_CHECK_ARG(!perm || (tile_maxpos && row_empty && !q_span),
                    "gamer_attn_bwd_bf16: a row order needs perm, tile_maxpos and row_empty, and excludes q_span");
    const QOrdB ord{perm, tile_maxpos, row_empty};
    hipStream_t st = (hipStream_t)stream;
    if (!delta_ready) {
        hipLaunchKernelGGL(attn_delta_b_kernel, dim3(2048), dim3(AB_THREADS), 0, st, (const bf16_t*)o, (const bf16_t*)d_o, B,
                           S, nq, delta);
        GAMER_CHECK_LAUNCH("gamer_attn_bwd_bf16/delta");
    }
    if (nq / nkv == 1)
        return launch_bwd_b<1>((const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lse,
                               delta, kl, ql, B, S, nq, nkv, scale, p_drop, seed, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk,
                               (bf16_t*)dv, lddv, q_span, ord, st);
    return launch_bwd_b<2>((const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lse, delta,
                           kl, ql, B, S, nq, nkv, scale, p_drop, seed, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv,
                           lddv, q_span, ord, st);
}
