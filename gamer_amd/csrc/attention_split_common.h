// Device helpers shared by the split-product attention kernels (attention_split.hip: tiled, K / V streamed per query tile;
// attention_res.hip: K / V of a (sequence, kv head) resident in LDS): the swizzled 16-bit piece images, the exact three-way bf16
// cut, the two-way fp16 cut with per-tensor scales (H2), the piece-product chains and the scales of one call.
#pragma once
#include "attention_common.h"

namespace gamer {

#ifndef SPA_ABLATE
#define SPA_ABLATE 0      // timing-only builds (tools/ablate_attn_split.sh): 1 no S = K Q^T MFMAs, 2 no second-product MFMAs, 4 no cut of P / dS,
#endif                    // 8 no cut arithmetic anywhere (pieces = the raw word), 16 no LDS tile stores, 32 (H2) no staging of K / V / Q / dO tiles at all,
                          // forward only: 64 no exp2, 128 no barrier in the key loop, 256 no softmax / dropout at all, 512 no global loads in the key loop
constexpr int SIMG = 32 * 64;              // bf16 elements of one piece image of a 32-row tile

// element offset of (row, column) in a [rows][64] bf16 image with swizzled 16-byte chunks (attention_bf16.hip: lds_off)
__device__ __forceinline__ int sl_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int sl_off(int row, int col) { return row * 64 + ((((col >> 3) ^ sl_f(row))) << 3) + (col & 7); }
struct SlOffsets {
    int row[4];        // row read:  row (lane & 31), columns 16 s + 8 h .. + 7        (s = 0..3)
    int tr[2][2];      // transposing read [c][db]: row 8 c + 4 h + q4, column 32 db + 16 gsel + 4 pp
    __device__ __forceinline__ SlOffsets(int lane) {
        const int r = lane & 31, h = lane >> 5, gsel = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int s = 0; s < 4; ++s) row[s] = sl_off(r, 16 * s + 8 * h);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int db = 0; db < 2; ++db) tr[c][db] = sl_off(8 * c + 4 * h + q4, 32 * db + 16 * gsel + 4 * pp);
    }
};
// row read: the 8 k-values (columns 16 s + 8 h ..) of row rblk + (lane & 31); rblk a multiple of 32
__device__ __forceinline__ bf16x8 read_row8(const bf16_t* __restrict__ img, const SlOffsets& lo, int rblk, int s) {
    return *reinterpret_cast<const bf16x8*>(img + rblk * 64 + lo.row[s]);
}
// transposing read (the MFMA's k index runs over ROWS of the image): for column 32 db + (lane & 31), the 8 rows
// rbase + 8 * (j >> 2) + 4 * h + (j & 3), j = 0..7 (rbase a multiple of 16) - the k order of an accumulator tile
__device__ __forceinline__ bf16x8 read_tr8(const bf16_t* __restrict__ img, const SlOffsets& lo, int rbase, int db) {
    bf16x8 out;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (bf16x4 __attribute__((address_space(3)))*)(img + rbase * 64 + lo.tr[c][db]));
        out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
    }
    return out;
}

// ---- the exact three-way cut (csrc/gemm.hip: split3) ---------------------------------------------------------------
// upper halves of (w1, w0) -> one register {bf16(w0), bf16(w1)}
__device__ __forceinline__ uint32_t hi16_pair(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x07060302u); }
// x = p0 + p1 + p2 exactly, p_i = the value of the upper 16 bits of w[i].  (Measured and not kept: the two subtractions on
// PAIRS of values as v_pk_add_f32 - fewer instructions, but the 64-bit register pairs it forces cost the forward kernel a
// spill cascade: 0.58 -> 0.91 ms at batch 256; the GEMM's cut got 2-7 % slower the same way.)
__device__ __forceinline__ void cut3(float x, uint32_t (&w)[3]) {
    if (SPA_ABLATE & 8) { w[0] = w[1] = w[2] = __float_as_uint(x); return; }
    w[0] = __float_as_uint(x);
    const float r1 = x - __uint_as_float(w[0] & 0xffff0000u);
    w[1] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(w[1] & 0xffff0000u);
    w[2] = __float_as_uint(r2);
}
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
// eight consecutive values (two float4) -> the three bf16x8 fragments
__device__ __forceinline__ void cut8(const float4& a, const float4& b, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t w[8][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) cut3(x[e], w[e]);
    u32x4s u[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[s][j] = hi16_pair(w[2 * j][s], w[2 * j + 1][s]);
    p0 = __builtin_bit_cast(bf16x8, u[0]); p1 = __builtin_bit_cast(bf16x8, u[1]); p2 = __builtin_bit_cast(bf16x8, u[2]);
}
// registers first .. first + 7 of an accumulator tile -> the three bf16x8 fragments (operand of the second product)
__device__ __forceinline__ void cut8_regs(const f32x16& t, const int first, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    if (SPA_ABLATE & 4) {
        u32x4s z;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = __float_as_uint(t[first + 2 * j]);
        p0 = p1 = p2 = __builtin_bit_cast(bf16x8, z);
        return;
    }
    uint32_t w[8][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) cut3(t[first + e], w[e]);
    u32x4s u[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[s][j] = hi16_pair(w[2 * j][s], w[2 * j + 1][s]);
    p0 = __builtin_bit_cast(bf16x8, u[0]); p1 = __builtin_bit_cast(bf16x8, u[1]); p2 = __builtin_bit_cast(bf16x8, u[2]);
}
// a staged 32 x 64 fp32 tile (load_tile32's registers: thread -> row f >> 4, columns 4 (f & 15) .. + 3) -> three images
__device__ __forceinline__ void store_tile32_split(bf16_t* __restrict__ img, int tid, const float4 (&rg)[2], int n_valid) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int f = tid + AT_THREADS * jj;
        const int row = f >> 4;
        const bool ok = row < n_valid;
        const float x[4] = {ok ? rg[jj].x : 0.f, ok ? rg[jj].y : 0.f, ok ? rg[jj].z : 0.f, ok ? rg[jj].w : 0.f};
        uint32_t w[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e) cut3(x[e], w[e]);
        bf16_t* dst = img + sl_off(row, (f & 15) << 2);
        if (SPA_ABLATE & 16) { asm volatile("" :: "v"(w[0][0]), "v"(w[1][1]), "v"(w[2][2]), "v"(w[3][0])); continue; }
#pragma unroll
        for (int s = 0; s < 3; ++s)
            *reinterpret_cast<uint2*>(dst + s * SIMG) = make_uint2(hi16_pair(w[0][s], w[1][s]), hi16_pair(w[2][s], w[3][s]));
    }
}
// ---- H2: the two-way fp16 cut of csrc/gemm.hip's SPLIT == 3 form (x s = h0 + h1 + r, |r| <= 2^-22 |x s|; s = a power of two that
// brings the tensor's largest magnitude into [2^13, 2^14)), three piece products h0 h0' + h0 h1' + h1 h0'.  Fragments travel in
// the same bf16x8 containers (bits), images 0 and 1 of the three-image layouts.
// piece fragments of eight consecutive values; H2: the values are multiplied by s first, p[2] is not written
template <bool H2>
__device__ __forceinline__ void cut8_t(const float4& a, const float4& b, float s, bf16x8 (&p)[3]) {
    if (!H2) { cut8(a, b, p[0], p[1], p[2]); return; }
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32x4s u0, u1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint32_t a0, a1, b0, b1;
        cut2h_quad(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3], s, a0, a1, b0, b1);
        u0[2 * j] = a0; u0[2 * j + 1] = b0; u1[2 * j] = a1; u1[2 * j + 1] = b1;
    }
    p[0] = __builtin_bit_cast(bf16x8, u0); p[1] = __builtin_bit_cast(bf16x8, u1);
}
template <bool H2>
__device__ __forceinline__ void cut8_regs_t(const f32x16& t, const int first, float s, bf16x8 (&p)[3]) {
    if (!H2) { cut8_regs(t, first, p[0], p[1], p[2]); return; }
    if (SPA_ABLATE & 4) {
        u32x4s z;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = __float_as_uint(t[first + 2 * j]);
        p[0] = p[1] = __builtin_bit_cast(bf16x8, z);
        return;
    }
    u32x4s u0, u1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint32_t a0, a1, b0, b1;
        cut2h_quad(t[first + 4 * j], t[first + 4 * j + 1], t[first + 4 * j + 2], t[first + 4 * j + 3], s, a0, a1, b0, b1);
        u0[2 * j] = a0; u0[2 * j + 1] = b0; u1[2 * j] = a1; u1[2 * j + 1] = b1;
    }
    p[0] = __builtin_bit_cast(bf16x8, u0); p[1] = __builtin_bit_cast(bf16x8, u1);
}
template <bool H2>
__device__ __forceinline__ void store_tile32_t(bf16_t* __restrict__ img, int tid, const float4 (&rg)[2], int n_valid, float s) {
    if (!H2) { store_tile32_split(img, tid, rg, n_valid); return; }
    if (SPA_ABLATE & 32) { asm volatile("" :: "v"(rg[0].x), "v"(rg[0].w), "v"(rg[1].y), "v"(rg[1].z)); return; }   // timing-only: no cut, no store
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int f = tid + AT_THREADS * jj;
        const int row = f >> 4;
        const bool ok = row < n_valid;
        uint32_t a0, a1, b0, b1;
        cut2h_quad(ok ? rg[jj].x : 0.f, ok ? rg[jj].y : 0.f, ok ? rg[jj].z : 0.f, ok ? rg[jj].w : 0.f, s, a0, a1, b0, b1);
        bf16_t* dst = img + sl_off(row, (f & 15) << 2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(dst + SIMG) = make_uint2(a1, b1);
    }
}
template <bool H2>
__device__ __forceinline__ f32x16 mfma_piece(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// acc += sum over piece pairs i + j < NP of A_i . B_j, smallest products first (NP = 3: the six products of the exact cut in
// the order (2,0) (1,1) (0,2) (1,0) (0,1) (0,0); NP = 2: (1,0) (0,1) (0,0)); ONE_B: B has one non-zero piece (0 / 1 values)
template <bool H2, bool ONE_B = false>
__device__ __forceinline__ f32x16 mfma_pieces(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
    constexpr int NP = H2 ? 2 : 3;
#pragma unroll
    for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
        for (int i = sum; i >= 0; --i) {
            const int j = sum - i;
            if (ONE_B && j != 0) continue;
            acc = mfma_piece<H2>(a[i], b[j], acc);
        }
    return acc;
}
// the scales of one attention call in the H2 form (powers of two; kernel-uniform)
struct H2Scales {
    float q, k, v, d_o, ds;          // operand tensors (from their maxima); dS: from the bound |dS| <= 128 / (1 - p) max|dO| max|V|
    float inv_qk, inv_v, inv_q, inv_k, inv_do, inv_ds;
    float ds_coef;                   // 128 / (1 - p) max|V|: times a ROW's max|dO| = that row's dS bound (the dQ kernel scales per row)
    static constexpr float P = 8192.f, INV_P = 1.f / 8192.f;       // probabilities (<= 1 / (1 - p_drop) < 4)
    __device__ __forceinline__ H2Scales(const uint32_t* __restrict__ aq, const uint32_t* __restrict__ ak,
                                        const uint32_t* __restrict__ av, const uint32_t* __restrict__ ado, float q_prescale,
                                        float drop_scale = 1.f) {
        q = k = v = d_o = ds = inv_q = inv_k = inv_v = inv_do = inv_ds = inv_qk = 1.f;
        ds_coef = 0.f;
        if (aq) {
            // q enters its products multiplied by q_prescale (the softmax scale, log2 domain): scale that product's maximum
            const uint32_t mv = amax_read(av);
            const uint32_t qb = __float_as_uint(__uint_as_float(amax_read(aq)) * q_prescale);
            scale_from_amax(qb, q, inv_q);
            scale_from_amax(amax_read(ak), k, inv_k);
            scale_from_amax(mv, v, inv_v);
            if (ado) {
                const uint32_t mdo = amax_read(ado);
                scale_from_amax(mdo, d_o, inv_do);
                // dS = p (mult dP - delta), p <= 1, |dP| <= 64 max|dO| max|V|, |delta| <= 64 max|dO| max|O|, |O| <= mult max|V|
                ds_coef = 128.f * drop_scale * __uint_as_float(mv);
                const float bound = ds_coef * __uint_as_float(mdo);
                scale_from_amax(__float_as_uint(bound), ds, inv_ds);
            }
            inv_qk = inv_q * inv_k;
        }
    }
};


// the maxima of q, k, v (and dO in the backward) of the H2 form: device words (gamer_absmax_f32 / gamer_amax_sink)
struct AttnAmax { const uint32_t* q; const uint32_t* k; const uint32_t* v; const uint32_t* d_o; };

// ---- attention_res.hip: the H2 kernels with K / V of a (sequence, kv head) resident in LDS ------------------------------------------
// GAMER_ATTN_RES=0 keeps every call on the tiled kernels of attention_split.hip (A/B runs); default on
static inline bool res_enabled() {
    static EnvSwitch sw("GAMER_ATTN_RES");             // (cached: gamer_reload_env() after a change inside the process)
    return sw.get(1) != 0;
}
// GAMER_ATTN_RES_DQ / _DKV / _DKV2 = 0: that kernel stays tiled (A/B runs)
#define GAMER_RES_PART(name) ([]() -> bool { static EnvSwitch sw(name); return sw.get(1) != 0; }())
// persistent workgroups: one per CU (GAMER_ATTN_RES_GRID overrides)
static inline int res_grid_cap() {
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
// Split a (sequence, kv head) pair's work over `ways` workgroups?  Persistent workgroups take whole units, so the launch lasts
// ceil(units / CUs) rounds: 384 pairs on 256 CUs (per-GPU batch 128) are two rounds with half the CUs idle in the second, 768 half
// units three full rounds at ~0.58 of a pair's time each (the split units stage the shared operand once more).  GAMER_ATTN_RES_SPLIT=0/1 forces.
static inline int res_split(int n_pairs, int ways) {
    if (ways < 2) return 0;
    static EnvSwitch sp("GAMER_ATTN_RES_SPLIT");
    if (sp.is_set()) return sp.get(0) != 0;
    const int ncu = res_grid_cap();
    const double whole = (double)((n_pairs + ncu - 1) / ncu), halves = 0.58 * (double)((ways * n_pairs + ncu - 1) / ncu);
    return halves < whole ? 1 : 0;
}
static inline bool res_split_forced() { static EnvSwitch sp("GAMER_ATTN_RES_SPLIT"); return sp.is_set(); }
// share of the CU-rounds that whole-pair units keep busy
static inline double res_fill(int n_pairs) {
    const int ncu = res_grid_cap();
    if (n_pairs <= ncu) return 1.0;          // (a single round: nothing to balance)
    return (double)n_pairs / ((double)((n_pairs + ncu - 1) / ncu) * ncu);
}
int launch_fwd_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* kl, const int32_t* ql,
                   const int32_t* row_empty, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed, float* o,
                   float* lse, RowOrder ro, uint32_t* amax_out, AttnAmax am, hipStream_t st);
int launch_dq_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                  const float* lse, float* delta, const int32_t* kl, const int32_t* ql, const int32_t* row_empty, int B, int S,
                  int nq, int nkv, float scale, float p_drop, uint64_t seed, float* dq, int lddq, RowOrder ro, int delta_ready,
                  AttnAmax am, hipStream_t st);
int launch_dkv_res(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* d_o, const float* lse,
                   const float* delta, const int32_t* kl, const int32_t* ql, const int32_t* row_empty, int B, int S, int nq, int nkv,
                   float scale, float p_drop, uint64_t seed, float* dk, int lddk, float* dv, int lddv, RowOrder ro, uint32_t* amax_out,
                   AttnAmax am, hipStream_t st);

}  // namespace gamer
