// Trie-constrained beam-search scoring for the SMB-decoder evaluation path (gfx950).
//
// Replaces, per generation step, HF's  log_softmax(next_token_logits)  followed by
// PrefixConstrainedLogitsProcessor driven by the reference's Trie
// (ref:SeqRec/generation/trie.py:5-104, ref:SeqRec/tasks/test_SMB_decoder.py:120,470-500): the reference walks a
// Python dict per (batch, beam) row on the host; here the item trie is a CSR array on the device and one
// wave scores one row.  Integer work + one HBM pass over the row's logits.
#include "attention_split_common.h"

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
// generated positions are per beam.  One workgroup = (sample, kv head); its num_beams * G <= 64 query rows are
// two 32-column MFMA tiles, so the prompt part is the forward kernel's inner loop (S^T = K Q^T with the query
// on the lane, scores straight into O^T += V^T P^T, v_mfma_f32_32x32x2_f32) with the KEY range split over the
// four waves: every wave streams its own 32-key tiles through a private LDS region (no workgroup barrier in
// the loop) and the four partial (max, sum, O) states are merged through LDS at the end, flash-decoding style.
// (The first version did the dot products on the VALU out of LDS and was LDS-bandwidth bound: 0.39 ms per call
// at 256 users x 20 beams against 0.1 ms for this one.)
//   self attention : prompt key j allowed iff key_ok[b][j]; generated keys (own ancestors + itself) allowed
//   cross attention: prompt key j allowed iff key_ok[b][j] (= kept and lower level than the target behaviour, the
//                    cached last mask row of model.py:603-617); generated keys masked; uniform[b] = 1 when no key
//                    is allowed: the row is then the mean of V over ALL L0 + t keys (finfo.min quirk)
// Sum over the 64 lanes for the tail's dot products with the generated keys (up to tmax per query row, one after the other): the pairs of
// the xor 32, 16, 8, 4, 2, 1 butterfly in that order - v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows
// of 16, the rest are DPP row rotations and quad permutations (their partners hold the xor partner's VALUE once the earlier steps have
// made the lanes 8- and 4-periodic) - so the bits are those of wave_sum's __shfl_xor chain (checked on random data), without its six
// dependent ds_bpermute round trips per dot product.
__device__ __forceinline__ float wave_sum_x(float v) {
#define GAMER_DEC_DPP(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    v += GAMER_DEC_DPP(0x128);            // row_ror:8
    v += GAMER_DEC_DPP(0x124);            // row_ror:4
    v += GAMER_DEC_DPP(0x4E);             // quad_perm [2,3,0,1]
    v += GAMER_DEC_DPP(0xB1);             // quad_perm [1,0,3,2]
#undef GAMER_DEC_DPP
    return v;
}
constexpr int DEC_MAXQ = 16;            // query rows per wave in the tail (4 waves): num_beams * G <= 64
constexpr int DEC_KLD = 68;             // floats per row of the K tile image (16-byte aligned rows, conflict-free)
constexpr int DEC_WAVE_LDS = 32 * DEC_KLD + 32 * 64 + 32;          // K tile + V tile + key_ok tile
constexpr int DEC_OLD = 65;             // row stride of the merged-state image [64 queries][64 d]
constexpr int DEC_MERGE_LDS = 4 * (64 * DEC_OLD + 128);
constexpr int DEC_LDS_FLOATS = DEC_MERGE_LDS > 4 * DEC_WAVE_LDS ? DEC_MERGE_LDS : 4 * DEC_WAVE_LDS;

typedef float dec_f32x16 __attribute__((ext_vector_type(16)));

template <int G>
__global__ void __launch_bounds__(256, 2)
attn_decode_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kp, int ldkp,
                   const float* __restrict__ vp, int ldvp, const int32_t* __restrict__ key_ok,
                   const float* __restrict__ kg, const float* __restrict__ vg, int ldg, int tmax, int t, int gen_ok,
                   const int32_t* __restrict__ uniform, int nb, int L0, int nq, int nkv, float scale,
                   float* __restrict__ o) {
    extern __shared__ __attribute__((aligned(16))) float dec_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / nkv, kvh = blockIdx.x % nkv;
    const int Qn = nb * G;
    const bool uni = uniform != nullptr && uniform[b] != 0;
    const float c2 = scale * 1.4426950408889634f;               // scores live in the log2 domain
    float* mrg_m = dec_lds;                                     // [4][64] running max of every wave
    float* mrg_l = dec_lds + 256;                               // [4][64] running sum
    float* mrg_o = dec_lds + 512;                               // [4][64][DEC_OLD] partial O

    if (uni) {
        // mean of V over the prompt keys (+ the generated ones in the tail): wave w sums keys w, w+4, ...
        float vsum = 0.f;
        for (int j = w; j < L0; j += 4) vsum += vp[((int64_t)b * L0 + j) * ldvp + kvh * 64 + lane];
        mrg_o[w * 64 + lane] = vsum;
        __syncthreads();
    } else {
        float* Kw = dec_lds + w * DEC_WAVE_LDS;
        float* Vw = Kw + 32 * DEC_KLD;
        int32_t* okw = reinterpret_cast<int32_t*>(Vw + 32 * 64);
        // query fragments of the two 32-column tiles (columns past Qn are zero and never stored)
        float qf[2][8][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qi = qt * 32 + r;
            const bool live = qi < Qn;
            const int qc = live ? qi : 0;
            const float* qrow = q + (int64_t)(b * nb + qc / G) * ldq + (kvh * G + qc % G) * 64;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float4 t4 = *reinterpret_cast<const float4*>(qrow + 8 * kk + 4 * h);
                qf[qt][kk][0] = live ? t4.x * c2 : 0.f; qf[qt][kk][1] = live ? t4.y * c2 : 0.f;
                qf[qt][kk][2] = live ? t4.z * c2 : 0.f; qf[qt][kk][3] = live ? t4.w * c2 : 0.f;
            }
        }
        float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
        dec_f32x16 oacc[2][2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][0][i] = 0.f; oacc[0][1][i] = 0.f; oacc[1][0][i] = 0.f; oacc[1][1][i] = 0.f; }
        const int n_tiles = (L0 + 31) >> 5;
        for (int jt = w; jt < n_tiles; jt += 4) {
            const int j0 = jt * 32;
            // this wave's K / V tile: 512 float4 each, 8 per lane; rows past the prompt are zero
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = lane + 64 * i, row = f >> 4, c4 = (f & 15) << 2;
                const int j = j0 + row;
                float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
                if (j < L0) {
                    kv = *reinterpret_cast<const float4*>(kp + ((int64_t)b * L0 + j) * ldkp + kvh * 64 + c4);
                    vv = *reinterpret_cast<const float4*>(vp + ((int64_t)b * L0 + j) * ldvp + kvh * 64 + c4);
                }
                *reinterpret_cast<float4*>(&Kw[row * DEC_KLD + c4]) = kv;
                *reinterpret_cast<float4*>(&Vw[row * 64 + c4]) = vv;
            }
            if (lane < 32) okw[lane] = (j0 + lane < L0) ? key_ok[(int64_t)b * L0 + j0 + lane] : 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes landed
            __builtin_amdgcn_wave_barrier();
            int okv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 t4 = *reinterpret_cast<const int4*>(okw + 8 * g4 + 4 * h);
                okv[4 * g4] = t4.x; okv[4 * g4 + 1] = t4.y; okv[4 * g4 + 2] = t4.z; okv[4 * g4 + 3] = t4.w;
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                dec_f32x16 st;
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 kf = *reinterpret_cast<const float4*>(&Kw[r * DEC_KLD + 8 * kk + 4 * h]);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[qt][kk][0], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[qt][kk][1], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[qt][kk][2], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[qt][kk][3], st, 0, 0, 0);
                }
                // register reg of lane (r, h) = score of key (reg&3) + 8*(reg>>2) + 4*h for query column r
                float mloc = -INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    st[reg] = okv[reg] != 0 ? st[reg] : -INFINITY;
                    mloc = fmaxf(mloc, st[reg]);
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mn = fmaxf(m_run[qt], mloc);
                const float base = mn == -INFINITY ? 0.f : mn;               // nothing allowed so far: p = 0
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - base);     // m = -inf -> 0
                float rowsum = 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    st[reg] = __builtin_amdgcn_exp2f(st[reg] - base);
                    rowsum += st[reg];
                }
                rowsum += __shfl_xor(rowsum, 32, 64);
                l_run[qt] = l_run[qt] * alpha + rowsum;
                m_run[qt] = mn;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[qt][0][i] *= alpha; oacc[qt][1][i] *= alpha; }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int key = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    oacc[qt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vw[key * 64 + r], st[reg], oacc[qt][0], 0, 0, 0);
                    oacc[qt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vw[key * 64 + 32 + r], st[reg], oacc[qt][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();                  // the tile is consumed before the next one overwrites it
        }
        __syncthreads();                                      // every wave is done with its tile region
        // publish this wave's state: O[query qt*32 + r][d = 32*dh + (reg&3) + 8*(reg>>2) + 4*h]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (h == 0) { mrg_m[w * 64 + qt * 32 + r] = m_run[qt]; mrg_l[w * 64 + qt * 32 + r] = l_run[qt]; }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    mrg_o[(w * 64 + qt * 32 + r) * DEC_OLD + 32 * dh + (reg & 3) + 8 * (reg >> 2) + 4 * h] = oacc[qt][dh][reg];
        }
        __syncthreads();
    }
    // tail: merge the four key ranges, add the generated positions of each beam (its own ancestors and the new
    // token), normalise and store.  Wave w takes query rows w, w + 4, ...; lane = d.
#pragma unroll 1
    for (int i = 0; i < DEC_MAXQ; ++i) {
        const int qi = w + 4 * i;
        if (qi >= Qn) break;
        const int n = b * nb + qi / G, head = kvh * G + qi % G;
        float res;
        if (uni) {
            float a = mrg_o[lane] + mrg_o[64 + lane] + mrg_o[128 + lane] + mrg_o[192 + lane];
            for (int g = 0; g < t; ++g) a += vg[((int64_t)n * tmax + g) * ldg + kvh * 64 + lane];
            res = a / (float)(L0 + t);
        } else {
            float mi = fmaxf(fmaxf(mrg_m[qi], mrg_m[64 + qi]), fmaxf(mrg_m[128 + qi], mrg_m[192 + qi]));
            float li = 0.f, a = 0.f;
            if (mi > -INFINITY) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float sc = __builtin_amdgcn_exp2f(mrg_m[ww * 64 + qi] - mi);        // -inf -> 0
                    li += mrg_l[ww * 64 + qi] * sc;
                    a += mrg_o[(ww * 64 + qi) * DEC_OLD + lane] * sc;
                }
            }
            if (gen_ok) {
                // the generated positions' key / value rows of this beam, four at a time, all requested before the first dot product
                // waits for one (a load per position inside the dependent softmax update was a memory round trip per position and row)
                const float qd = q[(int64_t)n * ldq + head * 64 + lane] * c2;
                for (int g0 = 0; g0 < t; g0 += 4) {
                    float kv4[4], vv4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t row = ((int64_t)n * tmax + min(g0 + u, t - 1)) * ldg + kvh * 64 + lane;
                        kv4[u] = kg[row];
                        vv4[u] = vg[row];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (g0 + u < t) {
                            const float s = wave_sum_x(qd * kv4[u]);
                            const float mn = fmaxf(mi, s);
                            const float alpha = __builtin_amdgcn_exp2f(mi - mn), pe = __builtin_amdgcn_exp2f(s - mn);
                            li = li * alpha + pe;
                            a = a * alpha + pe * vv4[u];
                            mi = mn;
                        }
                    }
                }
            }
            res = li > 0.f ? a / li : 0.f;
        }
        o[(int64_t)n * nq * 64 + head * 64 + lane] = res;
    }
}


// ---------------------------------------------------------------------------------------------------
// The same kernel with its two products in the three-piece fp16 form of the train step (csrc/attention_split.hip, H2): the prompt K / V
// tiles are cut into fp16 piece images (per-TENSOR scales from the maxima of the prompt caches, measured once per session), the query
// fragments once per workgroup with the workgroup's own scale, P at the fixed 2^13.  12 + 12 MFMAs of 32 cycles per (32 keys x 32 query
// columns) instead of 32 + 32 fp32 MFMAs of 64: the fp32 kernel ran 144 us per call at 256 users x 20 beams, bound by the fp32 matrix
// pipe (6.3 GFLOP per call on a 157 TFLOP/s pipe shared with the loads it waits for).  Tail (merge of the four key ranges, the <= 4
// generated positions in fp32 on the vector unit) as in attn_decode_kernel.
constexpr int DEC2_WAVE_BYTES = 4 * SIMG * (int)sizeof(bf16_t) + 32 * (int)sizeof(int32_t);          // K h0 | K h1 | V h0 | V h1 | key_ok
constexpr int DEC2_LDS_BYTES = (DEC_MERGE_LDS * (int)sizeof(float) > 4 * DEC2_WAVE_BYTES) ? DEC_MERGE_LDS * (int)sizeof(float) : 4 * DEC2_WAVE_BYTES;

#ifndef DEC2_WGS
#define DEC2_WGS 2          // workgroups per CU the kernel is built for
#endif
#ifndef DEC2_PREFETCH
#define DEC2_PREFETCH 0     // 1: the next tile's K / V rows are requested before the current tile is multiplied (64 more live registers)
#endif
template <int G>
__global__ void __launch_bounds__(256, DEC2_WGS)
attn_decode_h2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kp, int ldkp,
                      const float* __restrict__ vp, int ldvp, const int32_t* __restrict__ key_ok,
                      const float* __restrict__ kg, const float* __restrict__ vg, int ldg, int tmax, int t, int gen_ok,
                      const int32_t* __restrict__ uniform, int nb, int L0, int nq, int nkv, float scale,
                      float* __restrict__ o, const uint32_t* __restrict__ amax_k, const uint32_t* __restrict__ amax_v,
                      uint32_t* __restrict__ amax_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dec2_raw[];
    float* dec_lds = reinterpret_cast<float*>(dec2_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / nkv, kvh = blockIdx.x % nkv;
    const int Qn = nb * G;
    const bool uni = uniform != nullptr && uniform[b] != 0;
    const float c2 = scale * 1.4426950408889634f;               // scores live in the log2 domain
    float* mrg_m = dec_lds;                                     // [4][64] running max of every wave
    float* mrg_l = dec_lds + 256;                               // [4][64] running sum
    float* mrg_o = dec_lds + 512;                               // [4][64][DEC_OLD] partial O

    if (uni) {
        float vsum = 0.f;
        for (int j = w; j < L0; j += 4) vsum += vp[((int64_t)b * L0 + j) * ldvp + kvh * 64 + lane];
        mrg_o[w * 64 + lane] = vsum;
        __syncthreads();
    } else {
        bf16_t* Kw = reinterpret_cast<bf16_t*>(dec2_raw + (size_t)w * DEC2_WAVE_BYTES);      // pieces h0 | h1, SIMG apart
        bf16_t* Vw = Kw + 2 * SIMG;
        int32_t* okw = reinterpret_cast<int32_t*>(Vw + 2 * SIMG);
        const SlOffsets lo(lane);
        float sk, isk, sv, isv;
        scale_from_amax(amax_read(amax_k), sk, isk);
        scale_from_amax(amax_read(amax_v), sv, isv);
        // query fragments of the two 32-column tiles (columns past Qn are zero and never stored): B operands, lane = query column,
        // 8 d-values per k-step; one scale for the workgroup's queries (every wave derives the same one from the same rows)
        float4 qraw[2][8];
        float qmax = 0.f;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qi = qt * 32 + r;
            const bool live = qi < Qn;
            const int qc = live ? qi : 0;
            const float* qrow = q + (int64_t)(b * nb + qc / G) * ldq + (kvh * G + qc % G) * 64 + 8 * h;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float4 a4 = *reinterpret_cast<const float4*>(qrow + 16 * s), b4 = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
                if (!live) { a4 = make_float4(0.f, 0.f, 0.f, 0.f); b4 = a4; }
                a4.x *= c2; a4.y *= c2; a4.z *= c2; a4.w *= c2; b4.x *= c2; b4.y *= c2; b4.z *= c2; b4.w *= c2;
                qraw[qt][2 * s] = a4; qraw[qt][2 * s + 1] = b4;
                qmax = fmaxf(fmaxf(fmaxf(qmax, fabsf(a4.x)), fabsf(a4.y)), fmaxf(fabsf(a4.z), fabsf(a4.w)));
                qmax = fmaxf(fmaxf(fmaxf(qmax, fabsf(b4.x)), fabsf(b4.y)), fmaxf(fabsf(b4.z), fabsf(b4.w)));
            }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, o2, 64));
        float sq = 1.f, isq = 1.f;
        if (qmax > 0.f) scale_from_amax(__float_as_uint(qmax), sq, isq);
        bf16x8 qf[2][2][4];                                     // query tile x piece x k-step
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 pq[3];
                cut8_t<true>(qraw[qt][2 * s], qraw[qt][2 * s + 1], sq, pq);
                qf[qt][0][s] = pq[0]; qf[qt][1][s] = pq[1];
            }
        const float inv_qk = isq * isk;
        float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
        f32x16 oacc[2][2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][0][i] = 0.f; oacc[0][1][i] = 0.f; oacc[1][0][i] = 0.f; oacc[1][1][i] = 0.f; }
        const int n_tiles = (L0 + 31) >> 5;
        float4 kr[8], vr[8];
        int okv_g = 0;
        auto request = [&](int jt_) {
            const int jb = jt_ * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = lane + 64 * i, row = f >> 4, c4 = (f & 15) << 2;
                const int j = min(jb + row, L0 - 1);
                kr[i] = *reinterpret_cast<const float4*>(kp + ((int64_t)b * L0 + j) * ldkp + kvh * 64 + c4);
                vr[i] = *reinterpret_cast<const float4*>(vp + ((int64_t)b * L0 + j) * ldvp + kvh * 64 + c4);
            }
            okv_g = key_ok[(int64_t)b * L0 + min(jb + (lane & 31), L0 - 1)];
        };
        if (DEC2_PREFETCH && w < n_tiles) request(w);
        for (int jt = w; jt < n_tiles; jt += 4) {
            const int j0 = jt * 32;
            // this wave's K / V tile: 512 float4 each, 8 per lane, cut into the two piece images; rows past the prompt are zero.  Every
            // load is unconditional (clamped row) and requested before the first cut.
            if (!DEC2_PREFETCH) request(jt);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = lane + 64 * i, row = f >> 4, c4 = (f & 15) << 2;
                const bool in = j0 + row < L0;
                uint32_t a0, a1, b0, b1;
                cut2h_quad(in ? kr[i].x : 0.f, in ? kr[i].y : 0.f, in ? kr[i].z : 0.f, in ? kr[i].w : 0.f, sk, a0, a1, b0, b1);
                bf16_t* dk_ = Kw + sl_off(row, c4);
                *reinterpret_cast<uint2*>(dk_) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(dk_ + SIMG) = make_uint2(a1, b1);
                cut2h_quad(in ? vr[i].x : 0.f, in ? vr[i].y : 0.f, in ? vr[i].z : 0.f, in ? vr[i].w : 0.f, sv, a0, a1, b0, b1);
                bf16_t* dv_ = Vw + sl_off(row, c4);
                *reinterpret_cast<uint2*>(dv_) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(dv_ + SIMG) = make_uint2(a1, b1);
            }
            if (lane < 32) okw[lane] = (j0 + lane < L0) ? okv_g : 0;
            if (DEC2_PREFETCH) request(min(jt + 4, n_tiles - 1));           // (past the last tile: its rows once more, never used)
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes landed
            __builtin_amdgcn_wave_barrier();
            int okv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 t4 = *reinterpret_cast<const int4*>(okw + 8 * g4 + 4 * h);
                okv[4 * g4] = t4.x; okv[4 * g4 + 1] = t4.y; okv[4 * g4 + 2] = t4.z; okv[4 * g4 + 3] = t4.w;
            }
            // the tile's K row fragments (A operands: lane = key) serve both query tiles
            bf16x8 kf[2][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { kf[0][s] = read_row8(Kw, lo, 0, s); kf[1][s] = read_row8(Kw + SIMG, lo, 0, s); }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x16 st;
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {                 // smallest piece products first
                    st = mfma_piece<true>(kf[1][s], qf[qt][0][s], st);
                    st = mfma_piece<true>(kf[0][s], qf[qt][1][s], st);
                    st = mfma_piece<true>(kf[0][s], qf[qt][0][s], st);
                }
                // register reg of lane (r, h) = score of key (reg&3) + 8*(reg>>2) + 4*h for query column r
                float mloc = -INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    st[reg] = okv[reg] != 0 ? st[reg] * inv_qk : -INFINITY;
                    mloc = fmaxf(mloc, st[reg]);
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mn = fmaxf(m_run[qt], mloc);
                const float base = mn == -INFINITY ? 0.f : mn;               // nothing allowed so far: p = 0
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - base);     // m = -inf -> 0
                float rowsum = 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    st[reg] = __builtin_amdgcn_exp2f(st[reg] - base);
                    rowsum += st[reg];
                }
                rowsum += __shfl_xor(rowsum, 32, 64);
                l_run[qt] = l_run[qt] * alpha + rowsum;
                m_run[qt] = mn;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[qt][0][i] *= alpha; oacc[qt][1][i] *= alpha; }
                // O^T[d][query] += sum_key V[key][d] * P[query][key]
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf[3];
                    cut8_regs_t<true>(st, 8 * s2, H2Scales::P, pf);
                    pf[2] = pf[1];
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        bf16x8 vf[3];
                        vf[0] = read_tr8(Vw, lo, 16 * s2, db);
                        vf[1] = read_tr8(Vw + SIMG, lo, 16 * s2, db);
                        vf[2] = vf[1];
                        oacc[qt][db] = mfma_pieces<true>(vf, pf, oacc[qt][db]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                  // the tile is consumed before the next one overwrites it
        }
        __syncthreads();                                      // every wave is done with its tile region
        // publish this wave's state: O[query qt*32 + r][d = 32*dh + (reg&3) + 8*(reg>>2) + 4*h] (the products' scales taken out)
        const float osc = isv * H2Scales::INV_P;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (h == 0) { mrg_m[w * 64 + qt * 32 + r] = m_run[qt]; mrg_l[w * 64 + qt * 32 + r] = l_run[qt]; }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    mrg_o[(w * 64 + qt * 32 + r) * DEC_OLD + 32 * dh + (reg & 3) + 8 * (reg >> 2) + 4 * h] = oacc[qt][dh][reg] * osc;
        }
        __syncthreads();
    }
    // tail: as attn_decode_kernel
    float omax = 0.f;
#pragma unroll 1
    for (int i = 0; i < DEC_MAXQ; ++i) {
        const int qi = w + 4 * i;
        if (qi >= Qn) break;
        const int n = b * nb + qi / G, head = kvh * G + qi % G;
        float res;
        if (uni) {
            float a = mrg_o[lane] + mrg_o[64 + lane] + mrg_o[128 + lane] + mrg_o[192 + lane];
            for (int g = 0; g < t; ++g) a += vg[((int64_t)n * tmax + g) * ldg + kvh * 64 + lane];
            res = a / (float)(L0 + t);
        } else {
            float mi = fmaxf(fmaxf(mrg_m[qi], mrg_m[64 + qi]), fmaxf(mrg_m[128 + qi], mrg_m[192 + qi]));
            float li = 0.f, a = 0.f;
            if (mi > -INFINITY) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float sc = __builtin_amdgcn_exp2f(mrg_m[ww * 64 + qi] - mi);        // -inf -> 0
                    li += mrg_l[ww * 64 + qi] * sc;
                    a += mrg_o[(ww * 64 + qi) * DEC_OLD + lane] * sc;
                }
            }
            if (gen_ok) {
                // the generated positions' key / value rows of this beam, four at a time, all requested before the first dot product
                // waits for one (a load per position inside the dependent softmax update was a memory round trip per position and row)
                const float qd = q[(int64_t)n * ldq + head * 64 + lane] * c2;
                for (int g0 = 0; g0 < t; g0 += 4) {
                    float kv4[4], vv4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t row = ((int64_t)n * tmax + min(g0 + u, t - 1)) * ldg + kvh * 64 + lane;
                        kv4[u] = kg[row];
                        vv4[u] = vg[row];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (g0 + u < t) {
                            const float s = wave_sum_x(qd * kv4[u]);
                            const float mn = fmaxf(mi, s);
                            const float alpha = __builtin_amdgcn_exp2f(mi - mn), pe = __builtin_amdgcn_exp2f(s - mn);
                            li = li * alpha + pe;
                            a = a * alpha + pe * vv4[u];
                            mi = mn;
                        }
                    }
                }
            }
            res = li > 0.f ? a / li : 0.f;
        }
        o[(int64_t)n * nq * 64 + head * 64 + lane] = res;
        omax = fmaxf(omax, fabsf(res));
    }
    if (amax_out) {                                           // (gamer_amax_sink: max |o| for the o_proj GEMM's scale)
        uint32_t mw = __float_as_uint(omax);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
        // one word, thousands of waves: an atomic only when it would raise the maximum (a stale read costs one atomic more, never a wrong
        // result - the maximum only grows); unconditional, the 3072 atomics on one address took 19 us of a 85-us kernel
        if (lane == 0 && mw > __atomic_load_n(amax_out, __ATOMIC_RELAXED)) atomicMax(amax_out, mw);
    }
}

// the new token's rotated keys and (biased) values join the generated part of the cache: kg / vg [N][tmax][C] <- k / v rows, position g
__global__ void __launch_bounds__(256)
kv_append_kernel(const float4* __restrict__ k, int ldk4, const float4* __restrict__ v, int ldv4, float4* __restrict__ kg,
                 float4* __restrict__ vg, int ldg4, int tmax, int g, int N, int C4) {
    const int64_t total = (int64_t)N * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / C4), c = (int)(i % C4);
        const int64_t dst = ((int64_t)n * tmax + g) * ldg4 + c;
        kg[dst] = k[(int64_t)n * ldk4 + c];
        vg[dst] = v[(int64_t)n * ldv4 + c];
    }
}

}  // namespace gamer

using namespace gamer;

extern "C" int gamer_kv_append(const float* k, int ldk, const float* v, int ldv, float* kg, float* vg, int ldg, int tmax, int g,
                               int N, int C, void* stream) {
    GAMER_CHECK_ARG(k && v && kg && vg, "gamer_kv_append: null pointer");
    GAMER_CHECK_ARG(N > 0 && C > 0 && C % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldg % 4 == 0 && tmax > 0 && g >= 0 && g < tmax,
                    "gamer_kv_append: bad shape N=%d C=%d ldk=%d ldv=%d ldg=%d tmax=%d g=%d (multiples of 4)", N, C, ldk, ldv, ldg, tmax, g);
    GAMER_CHECK_ARG(aligned16(k) && aligned16(v) && aligned16(kg) && aligned16(vg), "gamer_kv_append: pointers must be 16-byte aligned");
    const int64_t total = (int64_t)N * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(kv_append_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(k), ldk / 4,
                       reinterpret_cast<const float4*>(v), ldv / 4, reinterpret_cast<float4*>(kg), reinterpret_cast<float4*>(vg), ldg / 4,
                       tmax, g, N, C / 4);
    GAMER_CHECK_LAUNCH("gamer_kv_append");
    return 0;
}

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

static int attn_decode_impl(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                            const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t,
                            int gen_ok, const int32_t* uniform, int B, int nb, int L0, int nq, int nkv,
                            float scale, float* o, const uint32_t* amax_k, const uint32_t* amax_v, void* stream) {
    GAMER_CHECK_ARG(q && kp && vp && key_ok && kg && vg && o, "gamer_attn_decode: null pointer");
    GAMER_CHECK_ARG(B > 0 && nb > 0 && L0 > 0 && nq > 0 && nkv > 0 && nq % nkv == 0 && t >= 0 && t <= tmax,
                    "gamer_attn_decode: bad shape B=%d nb=%d L0=%d nq=%d nkv=%d t=%d tmax=%d", B, nb, L0, nq, nkv, t, tmax);
    const int G = nq / nkv;
    GAMER_CHECK_ARG((G == 1 || G == 2) && nb * G <= 4 * DEC_MAXQ, "gamer_attn_decode: num_beams * (nq/nkv) = %d > %d",
                    nb * G, 4 * DEC_MAXQ);
    GAMER_CHECK_ARG(ldq % 4 == 0 && ldkp % 4 == 0 && ldvp % 4 == 0 && aligned16(q) && aligned16(kp) && aligned16(vp),
                    "gamer_attn_decode: q / prompt K / V need 16-byte alignment and leading dims %% 4 == 0");
    dim3 grid(B * nkv);
    if (amax_k && amax_v) {
        // the three-piece fp16 form (prompt K / V maxima given); max |o| goes to the slot gamer_amax_sink armed, if any
        uint32_t* sink_o = take_amax_sink().out[0];
        static bool attr2_dev[MAX_DEVICES] = {};
        bool& attr2 = attr2_dev[current_device()];
        if (!attr2) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_h2_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, DEC2_LDS_BYTES);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_h2_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, DEC2_LDS_BYTES);
            if (e != hipSuccess) {
                set_error("gamer_attn_decode_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return (int)e;
            }
            attr2 = true;
        }
        if (G == 1)
            hipLaunchKernelGGL(attn_decode_h2_kernel<1>, grid, dim3(256), DEC2_LDS_BYTES, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                               key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o, amax_k, amax_v, sink_o);
        else
            hipLaunchKernelGGL(attn_decode_h2_kernel<2>, grid, dim3(256), DEC2_LDS_BYTES, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                               key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o, amax_k, amax_v, sink_o);
        GAMER_CHECK_LAUNCH("gamer_attn_decode_split");
        return 0;
    }
    const size_t shmem = (size_t)DEC_LDS_FLOATS * sizeof(float);
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_kernel<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) {
            set_error("gamer_attn_decode: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr = true;
    }
    if (G == 1)
        hipLaunchKernelGGL(attn_decode_kernel<1>, grid, dim3(256), shmem, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                           key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o);
    else
        hipLaunchKernelGGL(attn_decode_kernel<2>, grid, dim3(256), shmem, (hipStream_t)stream, q, ldq, kp, ldkp, vp, ldvp,
                           key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, nb, L0, nq, nkv, scale, o);
    GAMER_CHECK_LAUNCH("gamer_attn_decode");
    return 0;
}

extern "C" int gamer_attn_decode(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                                 const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t,
                                 int gen_ok, const int32_t* uniform, int B, int nb, int L0, int nq, int nkv,
                                 float scale, float* o, void* stream) {
    return attn_decode_impl(q, ldq, kp, ldkp, vp, ldvp, key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, B, nb, L0, nq, nkv, scale, o,
                            nullptr, nullptr, stream);
}
extern "C" int gamer_attn_decode_split(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                                       const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t,
                                       int gen_ok, const int32_t* uniform, int B, int nb, int L0, int nq, int nkv,
                                       float scale, float* o, const uint32_t* amax_k, const uint32_t* amax_v, void* stream) {
    GAMER_CHECK_ARG(amax_k && amax_v, "gamer_attn_decode_split: the maxima of the prompt K and V caches (gamer_absmax_f32 slots) are required");
    return attn_decode_impl(q, ldq, kp, ldkp, vp, ldvp, key_ok, kg, vg, ldg, tmax, t, gen_ok, uniform, B, nb, L0, nq, nkv, scale, o,
                            amax_k, amax_v, stream);
}
