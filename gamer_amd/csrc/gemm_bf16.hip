// bf16 MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate) - the AMP variant of the nn.Linear sites.
//
// Two kernels, both 128 x 128 workgroup tiles, 4 waves as 2 x 2, each wave a 64 x 64 patch = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator registers), operand tiles global -> registers -> LDS, double buffered, one barrier per
// K-step of 64:
//
//   gemm_bf16_kernel   C[m][n] = sum_k A[m][k] * B[n][k]      both operands K-CONTIGUOUS (Linear forward; the dgrad
//       GEMMs run in this form too, against a transposed bf16 copy of the weights that the optimizer refreshes once per
//       step - 49 MB against GB-sized activations).  LDS image [row][64 + 8] bf16: the 8 k-values of one MFMA operand
//       are one conflict-free ds_read_b128.  Epilogues: bf16 store (optionally +=), fp32 residual + dropout store
//       (the decoder layer's three residual sites), bf16 store + row-dot (attention backward's delta).
//   gemm_bf16_wgrad_kernel   C[m][n] += sum_t A[t][m] * B[t][n]   the contraction runs over TOKENS, so both operands
//       are strided in k.  Tiles are staged as they lie in memory ([token][128 columns], 256-byte rows, 16-byte chunks
//       XOR-swizzled by the token's low bits) and read with ds_read_b64_tr_b16: the hardware transposing read hands
//       every lane 4 consecutive tokens of its column, two reads = the 8 k-values of a fragment, conflict-free.
//       Split over token chunks, combined with fp32 atomics (one accumulator register = two 128-byte row segments,
//       the shape the memory-side atomic units take at full rate).
#include "common.h"
#include <stdlib.h>

namespace gamer {

constexpr int HB_BM = 128, HB_BN = 128, HB_BK = 64;
constexpr int HB_THREADS = 256;
constexpr int HB_MAX_GROUPS = 64;
#ifndef HB_ABLATE
#define HB_ABLATE 0            // tools/ablate_gemm.sh: 1 no epilogue stores, 2 no global loads, 4 no MFMA / LDS reads, 8 no LDS stores
#endif
// K-contiguous kernel: K-step and workgroups per CU (tools/ablate_gemm.sh builds other combinations)
// (measured: K-step 32 with three workgroups per CU - 41 KB of LDS, 164 VGPRs - is 2-8 % SLOWER on every shape)
#ifndef HF_BK
#define HF_BK 64
#endif
#ifndef HF_WGS
#define HF_WGS 2
#endif
constexpr int HB_LD = HF_BK + 8;                       // bf16 elements per row of a K-contiguous tile image (80 / 144 B)
constexpr int HB_TILE = HB_BM * HB_LD;                 // elements per operand image
constexpr int HB_LDS_BYTES = 4 * HB_TILE * 2;          // 2 operands x 2 buffers: 40,960 B (K-step 32) / 73,728 B (64)
constexpr int HF_NJ = HF_BK / 16;                      // 16-byte chunks per thread, operand and K-step
constexpr int HF_CH = HF_BK / 8;                       // chunks per tile row
constexpr int HF_CS = HF_BK == 64 ? 3 : 2;
static_assert(HF_BK == 32 || HF_BK == 64, "K-step of the K-contiguous bf16 GEMM");
constexpr int HW_TILE = HB_BK * 128;                   // wgrad: [64 tokens][128 columns] elements per operand image
constexpr int HW_LDS_BYTES = 4 * HW_TILE * 2;          // 65,536 B

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Staging helpers as functions over array references (kept in registers; the same code written as lambdas over captured
// uint4 arrays was placed in scratch memory by hipcc: 144 bytes per lane, every K-step through private memory).
template <int NJ>
__device__ __forceinline__ void hb_load4(const bf16_t* __restrict__ base, const int64_t (&off)[NJ], int64_t k, u32x4 (&r)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) r[j] = *reinterpret_cast<const u32x4*>(base + off[j] + k);
}
template <int NJ>
__device__ __forceinline__ void hb_store4(bf16_t* __restrict__ dst, const int (&loff)[NJ], const u32x4 (&r)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4*>(dst + loff[j]) = r[j];
}

struct GemmBf16Params {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    int accumulate;
    int groups;
    const int32_t* group_offsets;
    int64_t strideB, strideC;
    int kchunk;
    int m_tiles, n_tiles;
    const float* resid;
    const int32_t* row_map;
    float p_drop;
    uint64_t seed;
    const bf16_t* rowdot_other;
    float* rowdot_out;
    int rowdot_S;
    // EPI 3 (q|k|v projection): per-head RMSNorm x weight + RoPE of the q / k heads (+ behaviour biases), see hb_epilogue
    const float* qk_wq; const float* qk_wk; float qk_eps;
    const float* qk_cos; const float* qk_sin;
    const float* qk_bias_q; const float* qk_bias_k; const float* qk_bias_v;
    const int32_t* qk_act_idx; const int32_t* qk_pos_ids;
    bf16_t* qk_q_rot; bf16_t* qk_k_rot;
    int qk_S, qk_nq, qk_nkv;
    // EPI 4 (input gradient of the experts' down projection): C = d(hm) is rounded to bf16 and consumed by the SwiGLU backward in
    // the epilogue - sw_gu[row][col] <- d gate, sw_gu[row][N + col] <- d up (gamer_swiglu_bwd_ld_bf16's arithmetic) - not stored
    bf16_t* sw_gu; int64_t sw_ld;
    // weight gradient, ordered form: chunk partial tiles [chunk][tile][128][128] (plain stores) instead of fp32 atomics into C;
    // gemm.hip's wgrad_reduce_kernel adds them in chunk order afterwards
    float* wgrad_ws;
};

__device__ __forceinline__ int xcd_remap_b(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, idx = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// =================================================================================================
// K-contiguous x K-contiguous
// =================================================================================================
// EPI 0: C bf16 (= or +=); EPI 1: C fp32 = resid + dropout(acc), rows optionally scattered through row_map;
// EPI 2: C bf16 + rowdot_out[b][head][i] = sum over the head's 64 columns of C * other (full tiles only)
//
// Persistent workgroups: at K = 256 a tile is four K-steps, so the first global load of a tile (nothing to hide it
// behind) and the epilogue were most of its time (first version: 12 % of the matrix pipe).  Each workgroup now walks a
// list of tiles and requests the first K-step of tile n+1 during the last K-step of tile n, so the load is in flight
// while the epilogue of tile n runs - and that epilogue no longer touches LDS: the MFMA is issued with its operands
// swapped (acc = C^T tile: the lane owns a ROW of C, its registers run over columns), v_permlane32_swap pairs the two
// lane halves so that every lane holds 8 consecutive columns, and the tile leaves as 16-byte stores straight from
// registers.  The residual epilogue reads and writes its float4 pieces the same way.
struct HbTile { int row0, row_end, col0, g; bool valid; };

// `offs` = the segment offsets in LDS (read once at kernel start: a global load here would put an s_waitcnt vmcnt(0)
// - a drain of the whole operand prefetch - at every tile boundary)
__device__ __forceinline__ HbTile hb_locate(const GemmBf16Params& p, const int32_t* offs, int L) {
    HbTile t;
    const int mt = L / p.n_tiles;
    t.col0 = (L % p.n_tiles) * HB_BN;
    t.g = 0;
    int seg_beg = 0, seg_end = p.M, tiles_before = 0;
    bool found = false;
    if (p.group_offsets) {
        int prev = offs[0];
        for (int gi = 0; gi < p.groups; ++gi) {
            const int nxt = offs[gi + 1];
            const int tiles = (nxt - prev + HB_BM - 1) / HB_BM;
            if (!found && mt < tiles_before + tiles) { t.g = gi; seg_beg = prev; seg_end = nxt; found = true; }
            if (!found) tiles_before += tiles;
            prev = nxt;
        }
    } else {
        found = mt < (p.M + HB_BM - 1) / HB_BM;
    }
    t.valid = found;
    t.row0 = seg_beg + (mt - tiles_before) * HB_BM;
    t.row_end = seg_end;
    return t;
}

// epilogue of one tile from registers: lane = row (i*32 + r32), acc[i][j][4*g4 + e] = column j*32 + 8*g4 + 4*h + e
// (wm, wn) = position of the 64 x 64 patch inside the workgroup tile, `interior` = the whole workgroup tile is in range
template <int EPI, bool ACCUM>
__device__ __forceinline__ void hb_epilogue(const GemmBf16Params& p, f32x16 (&acc)[2][2], const int row0, const int row_end,
                                            const int col0, const int col_end, const int wm, const int wn, const int r32,
                                            const int h, const bool wave_live, const bool interior) {
    // ---- epilogue from registers: lane = row (i*32 + r32), acc[i][j][4*g4 + e] = column j*32 + 8*g4 + 4*h + e ----
    if (wave_live) {
        if (EPI == 1) {
            float* Cp = reinterpret_cast<float*>(p.C);
            const DropoutRng rng(p.p_drop, p.seed);
            if (interior && (p.ldc & 3) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    const int64_t rc = p.row_map ? (int64_t)p.row_map[row] : (int64_t)row;
                    const int64_t e0 = rc * p.ldc + col0 + wn * 64 + 4 * h;
                    // (all eight residual quads of the row requested at once: two at a time - what the 256 x 256 form had registers for, and
                    // it does not use this epilogue - left four memory round trips in a row per 32-row half)
                    float4 x[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) x[j][g4] = *reinterpret_cast<const float4*>(p.resid + e0 + j * 32 + 8 * g4);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2) {
                                const int g4 = 2 * gp + g2;
                                const int64_t e = e0 + j * 32 + 8 * g4;
                                float m[4];
                                rng.mult4((uint32_t)(e >> 2), m);
                                const float4 xx = x[j][g4];
                                float4 o;
                                o.x = xx.x + m[0] * acc[i][j][4 * g4 + 0]; o.y = xx.y + m[1] * acc[i][j][4 * g4 + 1];
                                o.z = xx.z + m[2] * acc[i][j][4 * g4 + 2]; o.w = xx.w + m[3] * acc[i][j][4 * g4 + 3];
                                *reinterpret_cast<float4*>(Cp + e) = o;
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    if (row >= row_end) continue;
                    const int64_t rc = p.row_map ? (int64_t)p.row_map[row] : (int64_t)row;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = col0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (col < col_end) {
                                const int64_t e = rc * p.ldc + col;
                                float m[4];
                                rng.mult4((uint32_t)(e >> 2), m);
                                Cp[e] = p.resid[e] + m[e & 3] * acc[i][j][r];
                            }
                        }
                }
            }
        } else if (EPI == 3) {
            // q|k|v projection of the AMP step with qknorm_rope_fwd_b8's arithmetic in the epilogue (full tiles only, checked
            // on the host): the wave's 64-column patch is ONE head, the lane owns a row and - after the lane-half swap - the 32
            // columns 32 j + 16 t2 + 8 h + e of it, so the 64-wide sum of squares is a local sum + one cross-half add, and the
            // RoPE partner d +- 32 is the same (t2, e) of the other j in the SAME lane.  As under autocast
            // (ref:SeqRec/models/generative/Qwen3Multi/model.py:88-101): the projection output is rounded to bf16 (stored: the
            // backward reads it), self attention normalises that bf16 value and rounds the normalised value before the fp32
            // weight, the cross attention adds the fp32 behaviour bias first and stays fp32; RoPE in fp32; q_rot / k_rot bf16.
            bf16_t* Cp = reinterpret_cast<bf16_t*>(p.C);
            const int hd = (col0 + wn * 64) >> 6;
            const bool cross = p.qk_bias_q != nullptr;
            const bool isq = hd < p.qk_nq, isk = !isq && hd < p.qk_nq + p.qk_nkv;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = row0 + wm * 64 + i * 32 + r32;
                bf16_t* crow = Cp + (int64_t)row * p.ldc + col0 + wn * 64 + 8 * h;
                const int a = cross ? p.qk_act_idx[row] : 0;
                float x[2][2][8];
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x8v v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * t2 + e]),
                                                                             __float_as_uint(acc[i][j][8 * t2 + 4 + e]), false, false);
                            v[e] = __uint_as_float(sw[0]);
                            v[4 + e] = __uint_as_float(sw[1]);
                        }
                        bf16x8 o = __builtin_convertvector(v, bf16x8);
                        const int c = j * 32 + 16 * t2 + 8 * h;                 // first of this lane's 8 columns of the head
                        if (isq || isk) {
                            *reinterpret_cast<bf16x8*>(crow + j * 32 + 16 * t2) = o;          // raw projection output
                            const float* bias = !cross ? nullptr
                                : (isq ? p.qk_bias_q + (int64_t)a * p.qk_nq * 64 + hd * 64 + c
                                       : p.qk_bias_k + (int64_t)a * p.qk_nkv * 64 + (hd - p.qk_nq) * 64 + c);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float xe = (float)o[e];
                                if (cross) xe += bias[e];
                                x[j][t2][e] = xe;
                                ss += xe * xe;
                            }
                        } else {
                            if (cross) {                                                      // v += bias_v, rounded once more
                                const float* bias = p.qk_bias_v + (int64_t)a * p.qk_nkv * 64 + (hd - p.qk_nq - p.qk_nkv) * 64 + c;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (float)o[e] + bias[e];
                                o = __builtin_convertvector(v, bf16x8);
                            }
                            *reinterpret_cast<bf16x8*>(crow + j * 32 + 16 * t2) = o;
                        }
                    }
                if (isq || isk) {
                    ss = xor32_sum(ss);
                    const float rstd = rsqrtf(ss * (1.f / 64.f) + p.qk_eps);
                    const float* wn_ = isq ? p.qk_wq : p.qk_wk;
                    const int pos = p.qk_pos_ids ? p.qk_pos_ids[row] : row % p.qk_S;
                    bf16_t* rot = isq ? p.qk_q_rot + (int64_t)row * p.qk_nq * 64 + hd * 64
                                      : p.qk_k_rot + (int64_t)row * p.qk_nkv * 64 + (hd - p.qk_nq) * 64;
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const int c = 16 * t2 + 8 * h;                               // columns c + e (j = 0) and 32 + c + e (j = 1)
                        float y0[8], y1[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float n0 = x[0][t2][e] * rstd, n1 = x[1][t2][e] * rstd;
                            if (!cross) { n0 = round_as<bf16_t>(n0); n1 = round_as<bf16_t>(n1); }
                            y0[e] = wn_[c + e] * n0;
                            y1[e] = wn_[32 + c + e] * n1;
                        }
                        f32x8v o0, o1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float cs = p.qk_cos[pos * 64 + c + e], sn = p.qk_sin[pos * 64 + c + e];   // cos[d] = cos[d + 32]
                            o0[e] = y0[e] * cs - y1[e] * sn;                          // rotate_half: d < 32 pairs with -x[d + 32]
                            o1[e] = y1[e] * cs + y0[e] * sn;
                        }
                        *reinterpret_cast<bf16x8*>(rot + c) = __builtin_convertvector(o0, bf16x8);
                        *reinterpret_cast<bf16x8*>(rot + 32 + c) = __builtin_convertvector(o1, bf16x8);
                    }
                }
            }
        } else if (EPI == 4) {
            const DropoutRng rng(p.p_drop, p.seed);
            if (interior) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    const int cbase = col0 + wn * 64 + 8 * h;
                    bf16_t* grow = p.sw_gu + (int64_t)row * p.sw_ld + cbase;
                    bf16x8 pa[2][2], pb[2][2];       // (the row's gate | up groups requested at once: four round trips -> one)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            pa[j][t2] = *reinterpret_cast<const bf16x8*>(grow + j * 32 + 16 * t2);
                            pb[j][t2] = *reinterpret_cast<const bf16x8*>(grow + j * 32 + 16 * t2 + p.N);
                        }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            // (the lane-half swap of the plain store: 8 consecutive columns 16 t2 + 8 h .. per lane)
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * t2 + e]),
                                                                                 __float_as_uint(acc[i][j][8 * t2 + 4 + e]), false, false);
                                v[e] = __uint_as_float(sw[0]);
                                v[4 + e] = __uint_as_float(sw[1]);
                            }
                            bf16_t* ga = grow + j * 32 + 16 * t2;
                            const bf16x8 a8 = pa[j][t2], b8 = pb[j][t2];
                            const int64_t e0 = (int64_t)row * p.N + cbase + j * 32 + 16 * t2;      // flat index of d(hm): the dropout word
                            float m0[4], m1[4];
                            rng.mult4((uint32_t)(e0 >> 2), m0);
                            rng.mult4((uint32_t)(e0 >> 2) + 1u, m1);
                            const float m[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
                            f32x8v dg, du;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float d = m[e] * round_as<bf16_t>(v[e]);       // (d(hm) is a bf16 tensor in the unfused path)
                                const float a = (float)a8[e], b = (float)b8[e];
                                dg[e] = d * b * dsilu_f(a);
                                du[e] = d * silu_f(a);
                            }
                            *reinterpret_cast<bf16x8*>(ga) = __builtin_convertvector(dg, bf16x8);
                            *reinterpret_cast<bf16x8*>(ga + p.N) = __builtin_convertvector(du, bf16x8);
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    if (row >= row_end) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = col0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (col < col_end) {
                                bf16_t* ga = p.sw_gu + (int64_t)row * p.sw_ld + col;
                                const float a = (float)ga[0], b = (float)ga[p.N];
                                const int64_t e = (int64_t)row * p.N + col;
                                float m[4];
                                rng.mult4((uint32_t)(e >> 2), m);
                                const float d = m[e & 3] * round_as<bf16_t>(acc[i][j][r]);
                                ga[0] = (bf16_t)(d * b * dsilu_f(a));
                                ga[p.N] = (bf16_t)(d * silu_f(a));
                            }
                        }
                }
            }
        } else {
            bf16_t* Cp = reinterpret_cast<bf16_t*>(p.C);
            if (interior && (p.ldc & 7) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    bf16_t* crow = Cp + (int64_t)row * p.ldc + col0 + wn * 64 + 8 * h;
                    const bf16_t* orow = EPI == 2 ? p.rowdot_other + (int64_t)row * p.ldc + col0 + wn * 64 + 8 * h : nullptr;
                    float dot = 0.f;
                    bf16x8 pot[2][2];                // (row-dot: the row's four groups of `other` requested at once, not one per store)
                    if (EPI == 2) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int t2 = 0; t2 < 2; ++t2) pot[j][t2] = *reinterpret_cast<const bf16x8*>(orow + j * 32 + 16 * t2);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            // groups g4 = 2*t2 (columns 16*t2 + 4h..) and 2*t2 + 1 (columns 16*t2 + 8 + 4h..): after the
                            // swap the lower lane half holds columns 16*t2 + 0..7, the upper half 16*t2 + 8..15
                            f32x8v v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * t2 + e]),
                                                                                 __float_as_uint(acc[i][j][8 * t2 + 4 + e]), false, false);
                                v[e] = __uint_as_float(sw[0]);
                                v[4 + e] = __uint_as_float(sw[1]);
                            }
                            bf16_t* dst = crow + j * 32 + 16 * t2;
                            if (ACCUM) {
                                const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
                            }
                            const bf16x8 o = __builtin_convertvector(v, bf16x8);
                            *reinterpret_cast<bf16x8*>(dst) = o;
                            if (EPI == 2) {
                                // delta uses the ROUNDED dO (what the attention backward reads), times O
                                const bf16x8 ot = pot[j][t2];
#pragma unroll
                                for (int e = 0; e < 8; ++e) dot += (float)o[e] * (float)ot[e];
                            }
                        }
                    if (EPI == 2) {
                        // the wave's 64-column patch is exactly one head; the two lane halves hold its two halves
                        dot = xor32_sum(dot);
                        if (h == 0) {
                            const int heads = p.N >> 6, head = (col0 + wn * 64) >> 6;
                            p.rowdot_out[((int64_t)(row / p.rowdot_S) * heads + head) * p.rowdot_S + row % p.rowdot_S] = dot;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = row0 + wm * 64 + i * 32 + r32;
                    if (row >= row_end) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = col0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (col < col_end) {
                                bf16_t* dst = Cp + (int64_t)row * p.ldc + col;
                                const float v = acc[i][j][r];
                                *dst = (bf16_t)(ACCUM ? v + (float)*dst : v);
                            }
                        }
                }
            }
        }
    }
}

template <int EPI, bool ACCUM>
__global__ void __launch_bounds__(HB_THREADS, HF_WGS)
gemm_bf16_kernel(const GemmBf16Params p, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);           // buffer b: A image at 2*b*HB_TILE, B image behind it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;
    // tile list: the logical tile range is cut into 8 contiguous parts, one per XCD (workgroup ids are dealt
    // round-robin over the XCDs), and the workgroups of an XCD walk their part side by side, so that the column tiles
    // of one row panel of A run at the same time on one L2
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
    const int per_xcd = (total_tiles + 7) >> 3;
    const int t_hi = min(total_tiles, (xcd + 1) * per_xcd);
    const int col_end = p.N;
    const int nkt = p.K / HF_BK;                                  // K % HF_BK == 0 (checked on the host)
    __shared__ int32_t s_offs[HB_MAX_GROUPS + 1];
    if (p.group_offsets && tid <= p.groups) s_offs[tid] = p.group_offsets[tid];
    __syncthreads();

    auto next_valid = [&](int& L) {
        HbTile t;
        t.valid = false;
        while (L < t_hi) {
            t = hb_locate(p, s_offs, L);
            if (t.valid) break;
            L += nlocal;
        }
        if (L >= t_hi) t.valid = false;
        return t;
    };
    // chunk f = tid + 256 j of a tile: row f >> 3, 16-byte k-chunk f & 7.  Rows past the end are read from the last
    // valid row (their results are never stored), so there is no predicate anywhere in the loads.
    int64_t oa[HF_NJ], ob[HF_NJ];
    int lo[HF_NJ];
#pragma unroll
    for (int j = 0; j < HF_NJ; ++j) {
        const int f = tid + HB_THREADS * j;
        lo[j] = (f >> HF_CS) * HB_LD + ((f & (HF_CH - 1)) << 3);
    }
    auto offsets = [&](const HbTile& t) {
#pragma unroll
        for (int j = 0; j < HF_NJ; ++j) {
            const int f = tid + HB_THREADS * j;
            oa[j] = (int64_t)min(t.row0 + (f >> HF_CS), t.row_end - 1) * p.lda + ((f & (HF_CH - 1)) << 3);
            ob[j] = (int64_t)t.g * p.strideB + (int64_t)min(t.col0 + (f >> HF_CS), col_end - 1) * p.ldb + ((f & (HF_CH - 1)) << 3);
        }
    };
    // The (tile, K-step) pairs of this workgroup form ONE stream of steps; the operands of step s+2 are requested while
    // step s is multiplied and step s+1 waits in the other register set: the memory latency under load is several
    // K-steps of bf16 MFMA time, and with one step of prefetch every K-step waited for its operands (0.32 ms for the
    // q|k|v shape against 0.21 ms of HBM time).  Step s lives in LDS buffer s & 1; r0a / r0b carry even steps, r1a / r1b
    // odd ones.  The load side (ld, ld_kt) runs two steps ahead of the compute side (cur, kt) over the same tile list.
    u32x4 r0a[HF_NJ], r0b[HF_NJ], r1a[HF_NJ], r1b[HF_NJ];
    int Lld = xcd * per_xcd + local;
    HbTile ld = next_valid(Lld);
    int ld_kt = 0;
    if (!ld.valid || nkt == 0) return;
    offsets(ld);
    // The loads are issued UNCONDITIONALLY (past the end of the tile list the last step is simply read again): with a
    // conditional issue the compiler has to assume "not issued" when it counts how many younger loads may stay in
    // flight behind the ones a ds_write needs, and waits for all of them.
#define HB_ISSUE(RA, RB)                                                                       \
    {                                                                                          \
        if (!(HB_ABLATE & 2)) {                                                                \
        hb_load4(p.A, oa, (int64_t)ld_kt * HF_BK, RA);                                         \
        hb_load4(p.B, ob, (int64_t)ld_kt * HF_BK, RB);                                         \
        }                                                                                      \
        if (ld.valid && ld_kt + 1 == nkt) {                                                    \
            Lld += nlocal;                                                                     \
            ld = next_valid(Lld);                                                              \
            if (ld.valid) { offsets(ld); ld_kt = 0; }                                          \
        } else if (ld.valid) {                                                                 \
            ++ld_kt;                                                                           \
        }                                                                                      \
    }
    int Lc = Lld;
    HbTile cur = ld;
    HB_ISSUE(r0a, r0b)                                             // step 0
    HB_ISSUE(r1a, r1b)                                             // step 1 (if there is one)
    hb_store4(smem, lo, r0a);
    hb_store4(smem + HB_TILE, lo, r0b);
    __syncthreads();
    int Ln = Lc + nlocal;
    HbTile nxt = next_valid(Ln);
    int kt = 0;
    bool wave_live = (cur.row0 + wm * 64 < cur.row_end) && (cur.col0 + wn * 64 < col_end);
    f32x16 acc[2][2];               // acc[i][j] = C^T tile: lane -> row i*32 + r32, register -> column j*32 + rowmap
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one step: BUF = LDS buffer of this step, (NA, NB) = registers of the next step, (FA, FB) = registers free for step + 2
#define HB_STEP(BUF, NA, NB, FA, FB)                                                                                   \
    {                                                                                                                  \
        HB_ISSUE(FA, FB)                                                                                               \
        if (wave_live && !(HB_ABLATE & 4)) {                                                                           \
            const bf16_t* as = smem + 2 * (BUF) * HB_TILE;                                                             \
            const bf16_t* bs = as + HB_TILE;                                                                           \
            _Pragma("unroll") for (int s = 0; s < HF_BK / 16; ++s) {                                                   \
                bf16x8 af[2], bf[2];                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
                    af[i] = *reinterpret_cast<const bf16x8*>(as + (wm * 64 + i * 32 + r32) * HB_LD + 16 * s + 8 * h);  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                    bf[j] = *reinterpret_cast<const bf16x8*>(bs + (wn * 64 + j * 32 + r32) * HB_LD + 16 * s + 8 * h);  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);         \
            }                                                                                                          \
        }                                                                                                              \
        const bool tile_done = ++kt == nkt;                                                                            \
        if (!(HB_ABLATE & 8)) {                                                                                        \
        hb_store4(smem + 2 * ((BUF) ^ 1) * HB_TILE, lo, NA);       /* past the last step: a harmless re-store */       \
        hb_store4(smem + 2 * ((BUF) ^ 1) * HB_TILE + HB_TILE, lo, NB);                                                 \
        } else { asm volatile("" :: "v"(NA[0]), "v"(NA[1]), "v"(NB[0]), "v"(NB[1])); }                                 \
        if (tile_done) {                                                                                               \
            if (!(HB_ABLATE & 1) || acc[0][0][0] == 1.2345e-30f)                                                       \
            hb_epilogue<EPI, ACCUM>(p, acc, cur.row0, cur.row_end, cur.col0, col_end, wm, wn, r32, h, wave_live,       \
                                    (cur.row0 + HB_BM <= cur.row_end) && (cur.col0 + HB_BN <= col_end));               \
            if (!nxt.valid) break;                                                                                     \
            cur = nxt;                                                                                                 \
            Ln += nlocal;                                                                                              \
            nxt = next_valid(Ln);                                                                                      \
            kt = 0;                                                                                                    \
            wave_live = (cur.row0 + wm * 64 < cur.row_end) && (cur.col0 + wn * 64 < col_end);                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                                 \
        }                                                                                                              \
        __syncthreads();                                                                                               \
    }
    while (true) {
        HB_STEP(0, r1a, r1b, r0a, r0b)
        HB_STEP(1, r0a, r0b, r1a, r1b)
    }
#undef HB_STEP
#undef HB_ISSUE
}

// =================================================================================================
// 256 x 256 tile form of the same GEMM (plain bf16 store): 8 waves as 2 x 4, each a 128 x 64 patch (4 x 2 MFMA tiles,
// 128 accumulator registers), one workgroup per CU, persistent.  Against the 128 x 128 form a K-step moves half the
// operand bytes from L2 to LDS per MFMA and needs 0.75 LDS fragment reads per MFMA instead of 1.
// =================================================================================================
constexpr int HG_BM = 256, HG_BN = 256, HG_THREADS = 512;
constexpr int HG_TILE = HG_BM * (HB_BK + 8);                   // elements per operand image (144-byte rows)
constexpr int HG_LDS_BYTES = 4 * HG_TILE * 2;                  // 147,456 B

template <int EPI, bool ACCUM>
__global__ void __launch_bounds__(HG_THREADS, 1)
gemm_bf16_big_kernel(const GemmBf16Params p, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    constexpr int LD = HB_BK + 8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int r32 = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
    const int per_xcd = (total_tiles + 7) >> 3;
    const int t_hi = min(total_tiles, (xcd + 1) * per_xcd);
    const int nkt = p.K / HB_BK;
    __shared__ int32_t s_offs[HB_MAX_GROUPS + 1];
    if (p.group_offsets && tid <= p.groups) s_offs[tid] = p.group_offsets[tid];
    __syncthreads();
    auto locate = [&](int L) {
        HbTile t;
        const int mt = L / p.n_tiles;
        t.col0 = (L % p.n_tiles) * HG_BN;
        t.g = 0;
        int seg_beg = 0, seg_end = p.M, tiles_before = 0;
        bool found = false;
        if (p.group_offsets) {
            int prev = s_offs[0];
            for (int gi = 0; gi < p.groups; ++gi) {
                const int nxt = s_offs[gi + 1];
                const int tiles = (nxt - prev + HG_BM - 1) / HG_BM;
                if (!found && mt < tiles_before + tiles) { t.g = gi; seg_beg = prev; seg_end = nxt; found = true; }
                if (!found) tiles_before += tiles;
                prev = nxt;
            }
        } else {
            found = mt < (p.M + HG_BM - 1) / HG_BM;
        }
        t.valid = found;
        t.row0 = seg_beg + (mt - tiles_before) * HG_BM;
        t.row_end = seg_end;
        return t;
    };
    auto next_valid = [&](int& L) {
        HbTile t;
        t.valid = false;
        while (L < t_hi) {
            t = locate(L);
            if (t.valid) break;
            L += nlocal;
        }
        if (L >= t_hi) t.valid = false;
        return t;
    };
    int64_t oa[4], ob[4];
    int lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + HG_THREADS * j;
        lo[j] = (f >> 3) * LD + ((f & 7) << 3);
    }
    auto offsets = [&](const HbTile& t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + HG_THREADS * j;
            oa[j] = (int64_t)min(t.row0 + (f >> 3), t.row_end - 1) * p.lda + ((f & 7) << 3);
            ob[j] = (int64_t)t.g * p.strideB + (int64_t)min(t.col0 + (f >> 3), p.N - 1) * p.ldb + ((f & 7) << 3);
        }
    };
    u32x4 ra[4], rb[4];
    int Lld = xcd * per_xcd + local;
    HbTile ld = next_valid(Lld);
    int ld_kt = 0;
    if (!ld.valid || nkt == 0) return;
    offsets(ld);
    auto issue = [&]() {
        hb_load4(p.A, oa, (int64_t)ld_kt * HB_BK, ra);
        hb_load4(p.B, ob, (int64_t)ld_kt * HB_BK, rb);
        if (ld.valid && ld_kt + 1 == nkt) {
            Lld += nlocal;
            ld = next_valid(Lld);
            if (ld.valid) { offsets(ld); ld_kt = 0; }
        } else if (ld.valid) {
            ++ld_kt;
        }
    };
    int Lc = Lld;
    HbTile cur = ld;
    issue();
    hb_store4(smem, lo, ra);
    hb_store4(smem + HG_TILE, lo, rb);
    __syncthreads();
    int Ln = Lc + nlocal;
    HbTile nxt = next_valid(Ln);
    int kt = 0, buf = 0;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    while (true) {
        issue();
        {
            const bf16_t* as = smem + 2 * buf * HG_TILE;
            const bf16_t* bs = as + HG_TILE;
#pragma unroll
            for (int s = 0; s < HB_BK / 16; ++s) {
                bf16x8 af[4], bf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(as + (wm * 128 + i * 32 + r32) * LD + 16 * s + 8 * h);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(bs + (wn * 64 + j * 32 + r32) * LD + 16 * s + 8 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
        const bool tile_done = ++kt == nkt;
        hb_store4(smem + 2 * (buf ^ 1) * HG_TILE, lo, ra);
        hb_store4(smem + 2 * (buf ^ 1) * HG_TILE + HG_TILE, lo, rb);
        if (tile_done) {
            // the 128 x 64 patch of the wave = two 64 x 64 patches of the common epilogue
            const bool interior = (cur.row0 + HG_BM <= cur.row_end) && (cur.col0 + HG_BN <= p.N);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bool live = (cur.row0 + wm * 128 + half * 64 < cur.row_end) && (cur.col0 + wn * 64 < p.N);
                hb_epilogue<EPI, ACCUM>(p, reinterpret_cast<f32x16(&)[2][2]>(acc[2 * half]), cur.row0, cur.row_end, cur.col0, p.N,
                                        2 * wm + half, wn, r32, h, live, interior);
            }
            if (!nxt.valid) break;
            cur = nxt;
            Ln += nlocal;
            nxt = next_valid(Ln);
            kt = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        buf ^= 1;
        __syncthreads();
    }
}

// =================================================================================================
// Wave-specialised form (plain bf16 store / +=): 256 x 128 tiles, EIGHT waves per workgroup, one workgroup per CU.
//   waves 4-7 (loaders): fetch the operand tiles with LDS-DMA (global_load_lds_dwordx4: no destination registers, no
//       ds_write pass) into a ring of three 48-KB stages, two K-steps ahead of the multiply, and never store;
//   waves 0-3 (compute): 64 x 128 patch each (2 x 4 MFMA tiles, 128 accumulator registers), fragments by ds_read_b128,
//       and the epilogue's global stores - and never load an operand from global memory.
// Why: a wave's memory operations retire in order, so a wave that stores a C tile and then loads the next operands waits
// for the store acknowledgements (~6 us per 256 x 256 tile at one workgroup per CU) before its loads count as done; in
// the one-role kernels the load, multiply and store phases of a workgroup therefore ADD (tools/ablate_gemm.sh).  Here the
// stores of the compute waves are fire-and-forget and the loader's only wait is a counted vmcnt for its own loads.
// LDS image: rows of 128 bytes (64 bf16) as they lie in memory, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
// - the permutation is applied on the SOURCE address of the LDS-DMA (its destination is lane-linear) and again in the
// fragment reads, which are conflict-free.  One raw s_barrier per K-step for all eight waves.
// =================================================================================================
constexpr int WS_BM = 256, WS_BN = 128, WS_LOADERS = 4, WS_THREADS = 256 + 64 * WS_LOADERS, WS_STAGES = 3;
constexpr int WS_A_BYTES = WS_BM * 128, WS_B_BYTES = WS_BN * 128, WS_STAGE = WS_A_BYTES + WS_B_BYTES;     // 49,152
constexpr int WS_GLDS = WS_STAGE / 1024 / WS_LOADERS;                 // LDS-DMA instructions per stage and loader wave (12)
static_assert(2 * WS_GLDS < 64, "a wave cannot have more than 63 memory operations in flight (6-bit vmcnt)");
constexpr int WS_LDS_BYTES = WS_STAGES * WS_STAGE + 512;              // + the segment offsets

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool ACCUM>
__global__ void __launch_bounds__(WS_THREADS, 1)
gemm_bf16_ws_kernel(const GemmBf16Params p, const int total_tiles) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw;
    int32_t* s_offs = reinterpret_cast<int32_t*>(smem + WS_STAGES * WS_STAGE);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
    const int per_xcd = (total_tiles + 7) >> 3;
    const int t_hi = min(total_tiles, (xcd + 1) * per_xcd);
    const int nkt = p.K / HB_BK;
    if (p.group_offsets && tid <= p.groups) s_offs[tid] = p.group_offsets[tid];
    __syncthreads();
    auto locate = [&](int L) {
        HbTile t;
        const int mt = L / p.n_tiles;
        t.col0 = (L % p.n_tiles) * WS_BN;
        t.g = 0;
        int seg_beg = 0, seg_end = p.M, tiles_before = 0;
        bool found = false;
        if (p.group_offsets) {
            int prev = s_offs[0];
            for (int gi = 0; gi < p.groups; ++gi) {
                const int nxt = s_offs[gi + 1];
                const int tiles = (nxt - prev + WS_BM - 1) / WS_BM;
                if (!found && mt < tiles_before + tiles) { t.g = gi; seg_beg = prev; seg_end = nxt; found = true; }
                if (!found) tiles_before += tiles;
                prev = nxt;
            }
        } else {
            found = mt < (p.M + WS_BM - 1) / WS_BM;
        }
        t.valid = found;
        t.row0 = seg_beg + (mt - tiles_before) * WS_BM;
        t.row_end = seg_end;
        return t;
    };
    auto next_valid = [&](int& L) {
        HbTile t;
        t.valid = false;
        while (L < t_hi) {
            t = locate(L);
            if (t.valid) break;
            L += nlocal;
        }
        if (L >= t_hi) t.valid = false;
        return t;
    };
    // both roles walk the same list of tiles; the number of K-steps of this workgroup fixes the barrier count
    int n_steps = 0;
    {
        int L = xcd * per_xcd + local;
        for (HbTile t = next_valid(L); t.valid; L += nlocal, t = next_valid(L)) n_steps += nkt;
    }
    if (n_steps == 0) return;

    if (wid >= 4) {
        // ------------------------------------------------ loaders ------------------------------------------------
        // piece g of the A image = rows 8g .. 8g+7 (1 KiB), lane -> row 8g + (lane >> 3), stored chunk lane & 7, which
        // holds the memory chunk (lane & 7) ^ ((row >> 1) & 7); B likewise.  Loader q takes the pieces g = q (mod 4): with
        // one loader the 6-bit vmcnt (63 operations in flight per wave) would cap the ring at 63 KB in flight per CU.
        constexpr int NA = WS_A_BYTES / 1024 / WS_LOADERS, NB = WS_B_BYTES / 1024 / WS_LOADERS;
        const int lq = wid - 4;
        const bf16_t* pa[NA];
        const bf16_t* pb[NB];
        int Lld = xcd * per_xcd + local;
        HbTile ld = next_valid(Lld);
        int ld_kt = 0;
        auto set_tile = [&](const HbTile& t) {
#pragma unroll
            for (int g = 0; g < NA; ++g) {
                const int row = 8 * (WS_LOADERS * g + lq) + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                pa[g] = p.A + (int64_t)min(t.row0 + row, t.row_end - 1) * p.lda + 8 * c;
            }
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                const int col = 8 * (WS_LOADERS * g + lq) + (lane >> 3);
                const int c = (lane & 7) ^ ((col >> 1) & 7);
                pb[g] = p.B + (int64_t)t.g * p.strideB + (int64_t)min(t.col0 + col, p.N - 1) * p.ldb + 8 * c;
            }
        };
        set_tile(ld);
        auto issue_stage = [&](int buf) -> bool {
            if (!ld.valid) return false;
            unsigned char* dst = smem + buf * WS_STAGE + lq * 1024;
            const int64_t kof = (int64_t)ld_kt * HB_BK;
            if (!(HB_ABLATE & 2))
#pragma unroll
            for (int g = 0; g < NA; ++g)
                __builtin_amdgcn_global_load_lds((gptr_t)(pa[g] + kof), (lptr_t)(dst + g * WS_LOADERS * 1024), 16, 0, 0);
            if (!(HB_ABLATE & 2))
#pragma unroll
            for (int g = 0; g < NB; ++g)
                __builtin_amdgcn_global_load_lds((gptr_t)(pb[g] + kof), (lptr_t)(dst + WS_A_BYTES + g * WS_LOADERS * 1024), 16, 0, 0);
            if (++ld_kt == nkt) {
                ld_kt = 0;
                Lld += nlocal;
                ld = next_valid(Lld);
                if (ld.valid) set_tile(ld);
            }
            return true;
        };
        issue_stage(0);
        const bool two = issue_stage(1);
        // stage 0 has landed when at most this wave's 12 pieces of stage 1 are still in flight
        static_assert(WS_GLDS == 12, "the counted waits below are written for 12 pieces per stage and loader");
        if (two) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int s = 0; s < n_steps; ++s) {
            // the compute waves multiply stage s; the buffer of stage s+2 was read in step s-1, which the last barrier ended
            const bool have = issue_stage((s + 2) % WS_STAGES);
            if (have) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");      // stage s+1 has landed
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    }
    // ---------------------------------------------------- compute ----------------------------------------------------
    const int r32 = lane & 31, h = lane >> 5;
    int Lc = xcd * per_xcd + local;
    HbTile cur = next_valid(Lc);
    int kt = 0;
    f32x16 acc[2][2][2];                       // [column half][i][jj]: two 64 x 64 patches of the common epilogue
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    const int swz = (r32 >> 1) & 7;
    int coff[HB_BK / 16];
#pragma unroll
    for (int s = 0; s < HB_BK / 16; ++s) coff[s] = ((2 * s + h) ^ swz) << 4;
    const int a_row = (wid * 64 + r32) * 128;
    const int b_row = WS_A_BYTES + r32 * 128;
    asm volatile("s_barrier" ::: "memory");
    for (int s = 0; s < n_steps; ++s) {
        const unsigned char* st = smem + (s % WS_STAGES) * WS_STAGE;
#pragma unroll
        for (int kb = 0; kb < ((HB_ABLATE & 4) ? 0 : HB_BK / 16); ++kb) {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_row + i * 32 * 128 + coff[kb]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(st + b_row + j * 32 * 128 + coff[kb]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j >> 1][i][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[j >> 1][i][j & 1], 0, 0, 0);
        }
        if (++kt == nkt) {
            // (holding the rounded tile in 64 registers and storing a quarter of it with every K-step of the next tile
            // was measured too: 0.451 against 0.407 ms on the q|k|v shape - the burst is not what costs)
            const bool interior = (cur.row0 + WS_BM <= cur.row_end) && (cur.col0 + WS_BN <= p.N);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if ((HB_ABLATE & 1) && acc[a][0][0][0] != 1.2345e-30f) continue;
                const bool live = (cur.row0 + wid * 64 < cur.row_end) && (cur.col0 + a * 64 < p.N);
                hb_epilogue<0, ACCUM>(p, acc[a], cur.row0, cur.row_end, cur.col0, p.N, wid, a, r32, h, live, interior);
            }
            Lc += nlocal;
            cur = next_valid(Lc);
            kt = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
        }
        asm volatile("s_barrier" ::: "memory");
    }
}

// =================================================================================================
// wgrad: contraction over tokens, both operands token-major
// =================================================================================================
// chunk c (16 bytes = 8 columns) of token row t lives at physical chunk c ^ ((t & 3) << 2) of its 256-byte LDS row
__device__ __forceinline__ int hw_off(int t, int chunk) { return t * 128 + ((chunk ^ ((t & 3) << 2)) << 3); }
// chunk j of the thread: token k0 + tk[j] (zero page past kend), element offset off[j] relative to token row 0
__device__ __forceinline__ void hw_load4(const bf16_t* __restrict__ base, int64_t ld, const int64_t (&off)[4], const int (&tk)[4],
                                         int k0, int kend, const bf16_t* __restrict__ zeros, u32x4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16_t* s = (k0 + tk[j] < kend) ? base + (int64_t)k0 * ld + off[j] : zeros;
        r[j] = *reinterpret_cast<const u32x4*>(s);
    }
}

__global__ void __launch_bounds__(HB_THREADS, 2)
gemm_bf16_wgrad_kernel(const GemmBf16Params p, const bf16_t* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);           // buffer b: A image at 2*b*HW_TILE, B image behind it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int L = xcd_remap_b(blockIdx.x, gridDim.x);

    const int tiles_mn = p.m_tiles * p.n_tiles;
    const int chunk = L / tiles_mn;
    const int tile = L % tiles_mn;
    const int row0 = (tile / p.n_tiles) * HB_BM;       // first column of A (= row of C)
    const int col0 = (tile % p.n_tiles) * HB_BN;       // first column of B (= column of C)
    int g = 0, seg_beg = 0, seg_end = p.K, chunks_before = 0;
    bool found = false;
    if (p.group_offsets) {
        int prev = p.group_offsets[0];
        for (int gi = 0; gi < p.groups; ++gi) {
            const int nxt = p.group_offsets[gi + 1];
            const int chunks = (nxt - prev + p.kchunk - 1) / p.kchunk;
            if (!found && chunk < chunks_before + chunks) { g = gi; seg_beg = prev; seg_end = nxt; found = true; }
            if (!found) chunks_before += chunks;
            prev = nxt;
        }
    } else {
        found = chunk < (p.K + p.kchunk - 1) / p.kchunk;
    }
    if (!found) return;
    const int kbeg = seg_beg + (chunk - chunks_before) * p.kchunk;
    const int kend = min(seg_end, kbeg + p.kchunk);
    float* Cp = reinterpret_cast<float*>(p.C) + (int64_t)g * p.strideC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // chunk f = tid + 256 j of a tile: token f >> 4, 16-byte column chunk f & 15.  Column chunks past the operand's
    // width are read from its last chunk (those rows / columns of C are never stored); tokens past the end of the
    // segment are read from a page of zeros (they must not contribute).
    const int a_cmax = ((p.M + 7) >> 3) - 1, b_cmax = ((p.N + 7) >> 3) - 1;        // last readable chunk (ld >= round_up(., 8))
    int64_t oa[4], ob[4];
    int lo[4], tk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + HB_THREADS * j;
        tk[j] = f >> 4;
        oa[j] = (int64_t)tk[j] * p.lda + (min((row0 >> 3) + (f & 15), a_cmax) << 3);
        ob[j] = (int64_t)tk[j] * p.ldb + (min((col0 >> 3) + (f & 15), b_cmax) << 3);
        lo[j] = hw_off(f >> 4, f & 15);
    }
    u32x4 ra[4], rb[4];
    // token rows at or past kend come from a page of zeros (a select on the ADDRESS: the load itself has no predicate)
    const bool wave_live = (row0 + wm * 64 < p.M) && (col0 + wn * 64 < p.N);
    // transposing read: lane 16g + 4q + pp of a wave supplies the address of (token 4-block row q, columns 4pp..4pp+3)
    // of its group's 4 x 16 block and receives column (lane & 15) of the four tokens
    const int q = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
    auto frag = [&](const bf16_t* img, int cbase, int sub) {
        // 8 k-values (tokens 16*sub + 8*h + 0..7) of column cbase + r32
        const int col = cbase + 16 * gsel + 4 * pp;
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = 16 * sub + 8 * h + 4 * c + q;
            const bf16_t* src = img + hw_off(t, col >> 3) + (col & 7);
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(src));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return out;
    };
    auto mfma_tile = [&](const bf16_t* as) {
        if (!wave_live) return;
        const bf16_t* bs = as + HW_TILE;
#pragma unroll
        for (int s = 0; s < HB_BK / 16; ++s) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag(as, wm * 64 + i * 32, s);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = frag(bs, wn * 64 + j * 32, s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    // Two K-steps of loads in flight: step k+2 is requested while step k is multiplied and step k+1 sits in the other
    // register set (the memory latency under load is several K-steps of bf16 MFMA time; one step of prefetch left every
    // K-step waiting for its operands).  ra / rb hold odd steps, ra2 / rb2 even ones.
    u32x4 ra2[4], rb2[4];
    const int nkt = (kend - kbeg + HB_BK - 1) / HB_BK;
    auto kof = [&](int kt) { return kbeg + kt * HB_BK; };
    // Every load and LDS store below is UNCONDITIONAL (steps past the end read the page of zeros and are never
    // multiplied): a conditional issue makes the compiler wait for the youngest loads too when an LDS store needs the
    // older ones.
    if (nkt <= 0) return;
    hw_load4(p.A, p.lda, oa, tk, kof(0), kend, zeros, ra2);
    hw_load4(p.B, p.ldb, ob, tk, kof(0), kend, zeros, rb2);
    hw_load4(p.A, p.lda, oa, tk, kof(1), kend, zeros, ra);
    hw_load4(p.B, p.ldb, ob, tk, kof(1), kend, zeros, rb);
    hb_store4(smem, lo, ra2);
    hb_store4(smem + HW_TILE, lo, rb2);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // even step kt: LDS buffer 0; step kt+1 is in ra / rb; request step kt+2 into ra2 / rb2
        hw_load4(p.A, p.lda, oa, tk, kof(kt + 2), kend, zeros, ra2);
        hw_load4(p.B, p.ldb, ob, tk, kof(kt + 2), kend, zeros, rb2);
        mfma_tile(smem);
        hb_store4(smem + 2 * HW_TILE, lo, ra);
        hb_store4(smem + 2 * HW_TILE + HW_TILE, lo, rb);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        // odd step kt+1: LDS buffer 1; step kt+2 is in ra2 / rb2; request step kt+3 into ra / rb
        hw_load4(p.A, p.lda, oa, tk, kof(kt + 3), kend, zeros, ra);
        hw_load4(p.B, p.ldb, ob, tk, kof(kt + 3), kend, zeros, rb);
        mfma_tile(smem + 2 * HW_TILE);
        hb_store4(smem, lo, ra2);
        hb_store4(smem + HW_TILE, lo, rb2);
        __syncthreads();
    }
    if (!wave_live) return;
    if (p.wgrad_ws) {
        // (elements past M / N hold sums of re-read columns: the second pass never adds them to C)
        float* blk = p.wgrad_ws + ((int64_t)chunk * tiles_mn + tile) * (int64_t)(HB_BM * HB_BN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lc = wn * 64 + j * 32 + r32;
                const int lr0 = wm * 64 + i * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) blk[(lr0 + (r & 3) + 8 * (r >> 2)) * HB_BN + lc] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + r32;
            const int rbase = row0 + wm * 64 + i * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M && col < p.N) atomicAdd(Cp + (int64_t)row * p.ldc + col, acc[i][j][r]);
            }
        }
    }
}

// [rows][cols] fp32 -> bf16 copy and/or transposed bf16 copy ([cols][ldt], columns >= rows left untouched): the
// operand copies of the parameters that the bf16 GEMMs read.  One launch covers every matrix of a table.
struct CastEntry { int64_t src, dst, dst_t; int rows, cols, ldt, tile0; };
__global__ void __launch_bounds__(256)
cast_params_kernel(const float* __restrict__ flat, bf16_t* __restrict__ out, bf16_t* __restrict__ out_t,
                   const CastEntry* __restrict__ table, int n_entries) {
    __shared__ float tile[32][33];
    int e = 0;
    for (int i = 1; i < n_entries; ++i) e = ((int)blockIdx.x >= table[i].tile0) ? i : e;
    const CastEntry ce = table[e];
    const int tiles_c = (ce.cols + 31) / 32;
    const int tl = blockIdx.x - ce.tile0;
    const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const float* src = flat + ce.src;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < ce.rows && c < ce.cols) ? src[(int64_t)r * ce.cols + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        if (out && r < ce.rows && c < ce.cols) out[ce.dst + (int64_t)r * ce.cols + c] = (bf16_t)v;
    }
    __syncthreads();
    if (out_t && ce.dst_t >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;               // transposed: row index c, column index r
            if (c < ce.cols && r < ce.rows) out_t[ce.dst_t + (int64_t)c * ce.ldt + r] = (bf16_t)tile[tx][ty + 8 * k];
        }
    }
}

}  // namespace gamer

using namespace gamer;

// 4 KB of zeros the wgrad kernel points out-of-range loads at.  A device symbol: present (zero-initialised by the loader)
// on every device the library runs on, no allocation or memset at first use (a first use inside a stream capture is fine).
__device__ __attribute__((aligned(4096))) uint16_t g_zero_page[2048];

static const bf16_t* zero_page() {
    static const bf16_t* page[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!page[dev]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
        page[dev] = static_cast<const bf16_t*>(ptr);
    }
    return page[dev];
}

template <int EPI, bool ACCUM>
static int launch_hb(const GemmBf16Params& p, int tiles, hipStream_t st) {
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    auto kfn = gemm_bf16_kernel<EPI, ACCUM>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           HB_LDS_BYTES);
        if (e != hipSuccess) { set_error("gamer_gemm_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    // HF_WGS workgroups per CU x 256 CUs, in multiples of 8 (one share per XCD); small problems: one workgroup per tile
    int blocks = 256 * HF_WGS;
    if (tiles < blocks) blocks = ((tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(HB_THREADS), HB_LDS_BYTES, st, p, tiles);
    GAMER_CHECK_LAUNCH("gamer_gemm_bf16");
    return 0;
}

template <bool ACCUM>
static int launch_ws(const GemmBf16Params& p, int tiles, hipStream_t st) {
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    auto kfn = gemm_bf16_ws_kernel<ACCUM>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           WS_LDS_BYTES);
        if (e != hipSuccess) { set_error("gamer_gemm_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    int blocks = 256;                                   // one workgroup per CU, in multiples of 8 (one share per XCD)
    if (tiles < blocks) blocks = ((tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(WS_THREADS), WS_LDS_BYTES, st, p, tiles);
    GAMER_CHECK_LAUNCH("gamer_gemm_bf16/ws");
    return 0;
}

template <int EPI, bool ACCUM>
static int launch_big(const GemmBf16Params& p, int tiles, hipStream_t st) {
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    auto kfn = gemm_bf16_big_kernel<EPI, ACCUM>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           HG_LDS_BYTES);
        if (e != hipSuccess) { set_error("gamer_gemm_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    int blocks = 256;                                   // one workgroup per CU, in multiples of 8 (one share per XCD)
    if (tiles < blocks) blocks = ((tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(HG_THREADS), HG_LDS_BYTES, st, p, tiles);
    GAMER_CHECK_LAUNCH("gamer_gemm_bf16/256");
    return 0;
}

extern "C" int gamer_gemm_bf16(const gamer_gemm_bf16_desc* d, void* stream) {
    GAMER_CHECK_ARG(d, "gamer_gemm_bf16: null descriptor");
    GAMER_CHECK_ARG(d->A && d->B && d->C, "gamer_gemm_bf16: null matrix pointer");
    GAMER_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gamer_gemm_bf16: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    GAMER_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0 && aligned16(d->A) && aligned16(d->B),
                    "gamer_gemm_bf16: A and B must be 16-byte aligned with leading dims %% 8 == 0 (lda=%lld ldb=%lld)",
                    (long long)d->lda, (long long)d->ldb);
    GAMER_CHECK_ARG(d->groups >= 1 && (d->groups == 1 || d->group_offsets), "gamer_gemm_bf16: groups=%d needs group_offsets", d->groups);
    GAMER_CHECK_ARG(d->group_mode == 0 || d->group_mode == 1, "gamer_gemm_bf16: group_mode=%d", d->group_mode);
    GemmBf16Params p;
    p.A = (const bf16_t*)d->A; p.lda = d->lda;
    p.B = (const bf16_t*)d->B; p.ldb = d->ldb;
    p.C = d->C; p.ldc = d->ldc;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.accumulate = d->accumulate;
    p.groups = d->groups; p.group_offsets = d->group_offsets;
    p.strideB = d->strideB; p.strideC = d->strideC;
    p.kchunk = d->kchunk;
    p.resid = d->resid; p.row_map = d->row_map; p.p_drop = d->p_drop; p.seed = d->seed;
    p.rowdot_other = (const bf16_t*)d->rowdot_other; p.rowdot_out = d->rowdot_out; p.rowdot_S = d->rowdot_S;
    p.qk_wq = d->qk_wq; p.qk_wk = d->qk_wk; p.qk_eps = d->qk_eps; p.qk_cos = d->qk_cos; p.qk_sin = d->qk_sin;
    p.qk_bias_q = d->qk_bias_q; p.qk_bias_k = d->qk_bias_k; p.qk_bias_v = d->qk_bias_v; p.qk_act_idx = d->qk_act_idx;
    p.qk_pos_ids = d->qk_pos_ids; p.qk_q_rot = (bf16_t*)d->qk_q_rot; p.qk_k_rot = (bf16_t*)d->qk_k_rot;
    p.qk_S = d->qk_S; p.qk_nq = d->qk_nq; p.qk_nkv = d->qk_nkv;
    p.sw_gu = (bf16_t*)d->sw_gu; p.sw_ld = d->sw_ld;
    p.wgrad_ws = nullptr;
    p.n_tiles = (d->N + HB_BN - 1) / HB_BN;
    hipStream_t st = (hipStream_t)stream;
    if (d->group_mode == 0) {
        GAMER_CHECK_ARG(d->K % 64 == 0 && d->lda >= d->K && d->ldb >= d->K,
                        "gamer_gemm_bf16: the K-contiguous form needs K %% 64 == 0 (K=%d) and lda, ldb >= K", d->K);
        GAMER_CHECK_ARG(d->ldc >= d->N && d->strideB % 8 == 0, "gamer_gemm_bf16: ldc=%lld < N or strideB %% 8 != 0", (long long)d->ldc);
        GAMER_CHECK_ARG(!d->resid || (!d->accumulate && !d->rowdot_out && d->ldc % 4 == 0 && aligned16(d->resid) &&
                                      aligned16(d->C) && d->p_drop >= 0.f && d->p_drop < 1.f),
                        "gamer_gemm_bf16: the residual epilogue writes fp32 C: ldc %% 4 == 0, no accumulate / row-dot");
        GAMER_CHECK_ARG(!d->rowdot_out || (d->rowdot_other && d->rowdot_S > 0 && d->groups == 1 && !d->accumulate &&
                                           d->M % HB_BM == 0 && d->N % HB_BN == 0 && d->M % d->rowdot_S == 0 &&
                                           d->ldc % 8 == 0 && aligned16(d->rowdot_other) && aligned16(d->C)),
                        "gamer_gemm_bf16: the row-dot epilogue needs full 128 x 128 tiles (M=%d N=%d), one group, ldc %% 8 == 0",
                        d->M, d->N);
        p.m_tiles = (d->M + HB_BM - 1) / HB_BM + (d->group_offsets ? d->groups : 0);
        const int64_t blocks = (int64_t)p.m_tiles * p.n_tiles;
        GAMER_CHECK_ARG(blocks < (1LL << 31), "gamer_gemm_bf16: grid too large");
        if (d->qk_q_rot) {
            // q|k|v projection with the per-head RMSNorm + RoPE epilogue: full tiles, one group, plain bf16 store
            GAMER_CHECK_ARG(d->qk_k_rot && d->qk_wq && d->qk_wk && d->qk_cos && d->qk_sin && d->qk_S > 0 && d->qk_nq > 0 && d->qk_nkv > 0 &&
                            d->N == (d->qk_nq + 2 * d->qk_nkv) * 64 && d->M % HB_BM == 0 && d->N % HB_BN == 0 && d->groups == 1 &&
                            !d->accumulate && !d->resid && !d->rowdot_out && d->ldc % 8 == 0 && aligned16(d->C) &&
                            aligned16(d->qk_q_rot) && aligned16(d->qk_k_rot) &&
                            (!d->qk_bias_q || (d->qk_bias_k && d->qk_bias_v && d->qk_act_idx)),
                            "gamer_gemm_bf16: the q|k|v epilogue needs N = (nq + 2 nkv) * 64, full 128 x 128 tiles (M=%d N=%d), one "
                            "group, a plain bf16 C with ldc %% 8 == 0, and bias_k / bias_v / act_idx with bias_q", d->M, d->N);
            static int big3 = -1;
            if (big3 < 0) { const char* e = getenv("GAMER_GEMM_BF16_QK_BIG"); big3 = e ? atoi(e) : 1; }
            if (big3 && d->M % HG_BM == 0 && d->N % HG_BN == 0 && d->M >= 16 * HG_BM) {
                GemmBf16Params q = p;
                q.n_tiles = d->N / HG_BN;
                q.m_tiles = d->M / HG_BM;
                return launch_big<3, false>(q, q.m_tiles * q.n_tiles, st);
            }
            return launch_hb<3, false>(p, (int)blocks, st);
        }
        if (d->sw_gu) {
            // input gradient of the down projection with the SwiGLU backward in the epilogue (128 x 128 form: the epilogue's
            // temporaries do not fit beside the 256 x 256 form's accumulators)
            GAMER_CHECK_ARG(!d->accumulate && !d->resid && !d->rowdot_out && d->N % 8 == 0 && d->sw_ld % 8 == 0 &&
                            d->sw_ld >= 2 * (int64_t)d->N && aligned16(d->sw_gu) && d->p_drop >= 0.f && d->p_drop < 1.f,
                            "gamer_gemm_bf16: the SwiGLU-backward epilogue needs a plain input-gradient GEMM, N %% 8 == 0 and "
                            "sw_ld >= 2 N, sw_ld %% 8 == 0 (N=%d sw_ld=%lld)", d->N, (long long)d->sw_ld);
            return launch_hb<4, false>(p, (int)blocks, st);
        }
        static int big = -1;
        // 0: 128^2 only; 1 (default): 256^2 where its last column tile is well filled, else the wave-specialised 256 x 128
        // form (N = 320: 0.321 against 0.345 ms), else 128^2; 2: wave-specialised wherever it applies.  Measured per shape
        // with tools/kbench_bf16.py; the vendor library (torch.mm, tools/vendor_gemm.py) is within +-8 % of the best of
        // these on every shape except the head forward (0.755 ms there, 0.55 here).
        if (big < 0) { const char* e = getenv("GAMER_GEMM_BF16_BIG"); big = e ? atoi(e) : 1; }
        const int rem = d->N % HG_BN;
        const bool big_ok = (rem == 0 || rem >= 3 * HG_BN / 4 || d->N >= 4 * HG_BN) && d->M >= 16 * HG_BM && !d->resid && !d->rowdot_out;
        if ((big == 2 || (big == 1 && !big_ok)) && d->M >= 16 * WS_BM && !d->resid && !d->rowdot_out) {
            GemmBf16Params q = p;
            q.n_tiles = (d->N + WS_BN - 1) / WS_BN;
            q.m_tiles = (d->M + WS_BM - 1) / WS_BM + (d->group_offsets ? d->groups : 0);
            const int tiles = q.m_tiles * q.n_tiles;
            if (d->accumulate) return launch_ws<true>(q, tiles, st);
            return launch_ws<false>(q, tiles, st);
        }
        // 256 x 256 tiles when they are well filled: at least 3/4 of the last column tile in use and enough row tiles to give
        // every CU work (the fp32 residual and the row-dot epilogues measured 3-15 % slower in this form - their epilogues
        // need the registers the 128 accumulators take - and stay with 128 x 128)
        if (big == 1 && big_ok) {
            GemmBf16Params q = p;
            q.n_tiles = (d->N + HG_BN - 1) / HG_BN;
            q.m_tiles = (d->M + HG_BM - 1) / HG_BM + (d->group_offsets ? d->groups : 0);
            const int tiles = q.m_tiles * q.n_tiles;
            if (d->accumulate) return launch_big<0, true>(q, tiles, st);
            return launch_big<0, false>(q, tiles, st);
        }
        if (d->resid) return launch_hb<1, false>(p, (int)blocks, st);
        if (d->rowdot_out) return launch_hb<2, false>(p, (int)blocks, st);
        if (d->accumulate) return launch_hb<0, true>(p, (int)blocks, st);
        return launch_hb<0, false>(p, (int)blocks, st);
    }
    // wgrad
    GAMER_CHECK_ARG(d->kchunk >= HB_BK && d->kchunk % HB_BK == 0, "gamer_gemm_bf16: kchunk=%d must be a positive multiple of 64", d->kchunk);
    GAMER_CHECK_ARG(d->lda >= ((d->M + 7) / 8) * 8 && d->ldb >= ((d->N + 7) / 8) * 8,
                    "gamer_gemm_bf16: wgrad operands need leading dims >= their width rounded up to 8 (lda=%lld M=%d ldb=%lld N=%d)",
                    (long long)d->lda, d->M, (long long)d->ldb, d->N);
    GAMER_CHECK_ARG(!d->resid && !d->rowdot_out, "gamer_gemm_bf16: no fused epilogue in the wgrad form");
    const bf16_t* zp = zero_page();
    GAMER_CHECK_ARG(zp, "gamer_gemm_bf16: could not allocate the zero page");
    p.m_tiles = (d->M + HB_BM - 1) / HB_BM;
    const int64_t chunks = (d->K + d->kchunk - 1) / d->kchunk + (d->group_offsets ? d->groups : 0);
    const int64_t blocks = chunks * p.m_tiles * p.n_tiles;
    GAMER_CHECK_ARG(blocks < (1LL << 31), "gamer_gemm_bf16: grid too large");
    static bool attr_dev[MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_wgrad_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, HW_LDS_BYTES);
        if (e != hipSuccess) { set_error("gamer_gemm_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    if (d->wgrad_ws) {
        const int64_t need = blocks * (int64_t)(HB_BM * HB_BN);
        GAMER_CHECK_ARG(d->wgrad_ws_floats >= need && aligned16(d->wgrad_ws),
                        "gamer_gemm_bf16: wgrad_ws holds %lld floats, this weight gradient needs %lld (chunks x tiles x 16384)",
                        (long long)d->wgrad_ws_floats, (long long)need);
        p.wgrad_ws = d->wgrad_ws;
    }
    hipLaunchKernelGGL(gemm_bf16_wgrad_kernel, dim3((int)blocks), dim3(HB_THREADS), HW_LDS_BYTES, st, p, zp);
    GAMER_CHECK_LAUNCH("gamer_gemm_bf16/wgrad");
    if (p.wgrad_ws) {
        const int e = launch_wgrad_reduce(p.wgrad_ws, reinterpret_cast<float*>(d->C), d->ldc, d->M, d->N, d->groups, d->group_offsets,
                                          d->K, d->kchunk, d->strideC, st);
        if (e) { set_error("gamer_gemm_bf16/wgrad_reduce: %s", hipGetErrorString((hipError_t)e)); return e; }
    }
    return 0;
}

extern "C" int gamer_cast_params_bf16(const float* flat, gamer_bf16* out, gamer_bf16* out_t, const int64_t* table,
                                      int n_entries, int n_tiles, void* stream) {
    GAMER_CHECK_ARG(flat && table && (out || out_t) && n_entries > 0 && n_tiles > 0, "gamer_cast_params_bf16: bad arguments");
    static_assert(sizeof(CastEntry) == 5 * sizeof(int64_t), "CastEntry layout");
    hipLaunchKernelGGL(cast_params_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, flat, (bf16_t*)out,
                       (bf16_t*)out_t, reinterpret_cast<const CastEntry*>(table), n_entries);
    GAMER_CHECK_LAUNCH("gamer_cast_params_bf16");
    return 0;
}
