// Activation-stationary Linear forward for short contractions (K <= 256: every nn.Linear of the decoder whose input is the hidden
// state - the q|k|v and output-gate projections, the tied head; ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 145-149, 1001),
// three-product fp16 form (gamer_gemm_f32_split, terms = 3).  Same results contract as csrc/gemm.hip: fp32 in, fp32 out,
// C[m][n] = alpha * sum_k A[m][k] W[n][k]; error against fp64 at or below the fp32 MFMA's.
//
// Why a second kernel.  The 128 x 128 x 32 tile kernel of gemm.hip re-cuts the activation tile for every column tile (six times at
// N = 768), stages both operands through LDS every K-step with a barrier pair per two slices, and at K = 256 pays its prologue and
// epilogue every eight K-steps: 26 % MFMA-busy, 5.8 vector instructions per MFMA (profiles/r05a_mfma_busy.md).  The resident
// attention kernels (attention_res.hip) showed what the same matrix pipe does when nothing is re-staged.  Here:
//   * a wave owns 32 rows of A for the WHOLE launch: it loads them once (the lane owns a row: 8 k-values per k-step, the layout of
//     the attention kernels' q fragments), cuts them once into fp16 pieces held in registers (128 registers at K = 256) - with a
//     power-of-two scale PER ROW (the row index is never contracted, so the scale is a per-lane factor of the epilogue): a row
//     far below the tensor's maximum keeps its full relative precision by construction, no range guard and no operand maximum needed;
//   * the workgroup (4 waves = 128 rows) walks the weight matrix in slabs of 32 output columns: a slab's packed pieces
//     (gamer_split2h_planes_multi: the parameters are cut once per pass) go global -> registers -> LDS into one of two buffers while
//     the other is multiplied - one barrier per slab of 48 MFMAs per wave (K = 256), no vector work in the loop but the epilogue's scale;
//   * C^T = W A^T puts the output row on the lane: the tile leaves as 16-byte stores per lane.
// A is read once, C written once, W (<= 1 MB) comes from the L2 once per 128 rows.
//
// Measured (batch 1024 x 505 tokens, tools/dev_gemm_as_time.py): q|k|v [517120 x 256] x [768 x 256]^T 0.94 ms (tile kernel) -> 0.70,
// tied head (N = 1041) 1.36 -> 0.95, the experts' gate|up (N = 1024, grouped) 1.37 -> 0.97; the forward GEMMs of a step 39.0 -> 33.4 ms
// (ideal for q|k|v: 0.24 ms of MFMAs, 0.27 ms of HBM bytes).
#include "common.h"
#include <stdlib.h>
#include <atomic>

namespace gamer {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));

#ifndef AS_WAVES
#define AS_WAVES 4        // default waves per workgroup (32 rows each): two workgroups per CU, out of step with each other - one loads its rows of
#endif                    // A while the other multiplies; the launcher takes 8 for the wide ungrouped forwards (template parameter NW of the kernel)
#ifndef AS_NT
#define AS_NT 1           // 32-column MFMA tiles per slab (LDS per workgroup: 2 buffers x 2 pieces x K x 32 AS_NT x 2 bytes)
#endif
#ifndef AS_DEPTH
#define AS_DEPTH 1        // slabs of W in flight in registers (2: three LDS buffers, two register sets; with 8 waves 0.739 against 0.764 ms
                          // on one box, 4 waves 1.06: the per-slab fetch latency is not what is left)
#endif
#ifndef AS_ABLATE
#define AS_ABLATE 0       // timing-only builds: 1 no C stores, 2 no MFMAs, 4 no W staging after the first slab
#endif
constexpr int AS_SLAB = 32 * AS_NT;            // output columns per slab
constexpr int AS_PANEL = AS_SLAB * 64;         // 16-bit elements of one [slab n][64 k] piece image

// element offset of (row, column) in a [64][64] 16-bit image with swizzled 16-byte chunks (attention_split_common.h: sl_off)
__device__ __forceinline__ int as_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int as_off(int row, int col) { return row * 64 + ((((col >> 3) ^ as_f(row))) << 3) + (col & 7); }

struct AsParams {
    const float* A; int64_t lda;
    const uint16_t* Wp;                        // packed pieces at W's offsets: 16 bytes = {h0 x 4 | h1 x 4} of four consecutive k
    int64_t ldw;                               // elements between rows of W
    float* C; int64_t ldc;
    int M, N, K;
    float alpha;
    const uint32_t* amax_w;                    // bits of max |W| (the scale the pieces were cut with)
    uint32_t* amax_c; int amax_c_col0;
    // grouped form (the position-routed experts: rows sorted by expert, group g = rows group_offsets[g] .. group_offsets[g + 1] - 1
    // with its own weight matrix strideW elements further): a workgroup never crosses a segment boundary
    int groups; const int32_t* group_offsets; int64_t strideW;
    // W row-contiguous (WRC: the input-gradient layout, W [K][N]): epilogues of csrc/gemm.hip that the layers with 256 OUTPUT features need
    int accumulate;                            // C += (the cross block's gate)
    const float* rowdot_other; float* rowdot_out; int rowdot_S;      // EPI 2: per (row, head of 64 columns) dot of C with `other` (o_proj)
    float* sw_gu; int64_t sw_ld; float p_drop; uint64_t seed;        // EPI 4: SwiGLU backward in place of the store (down projection)
    // EPI 5 (W k-contiguous, N = 2 I, the experts' fused gate|up projection): besides C = gate | up the kernel stores
    // hm[m][c] = dropout(silu(gate[m][c] + tg) * (up[m][c] + tu)) - gamer_swiglu_fwd_ld(_tbl)'s result without its pass over C.
    // The slabs are walked gate 0, up 0, gate 1, up 1, ...; tbl (optional): [groups][N] added to the values hm is computed from (not to
    // C); group_div: that many consecutive row groups share one W.
    float* hm; const float* tbl; int group_div;
};

// KP = K / 64 panels.  WRC = false: W [N][K] (k-contiguous, the Linear-forward layout): piece images [n][64 k] per panel, fragments by
// 16-byte reads.  WRC = true: W [K][N] (the input-gradient layout of a layer with K <= 256 OUTPUT features: o_proj, the experts' down
// projection, the cross block's gate): piece images [k][32 n], fragments by transposing reads (as gemm_os.hip).  EPI: 0 store (or
// accumulate), 2 row-dot, 4 SwiGLU backward (the epilogues of csrc/gemm.hip, same definitions).
// NW = waves per workgroup (32 rows each): 4 (two workgroups per CU, out of step) or 8 (one; W is fetched once per 256 rows - faster for
// the wide ungrouped launches, q|k|v 0.73 -> 0.68 ms, head 0.97 -> 0.93; slower for the grouped gate|up, 0.97 -> 1.15, and the 256-column gate)
template <int KP, bool WRC, int EPI, int NW>
__global__ void __launch_bounds__(64 * NW, AS_DEPTH == 2 ? 1 : 8 / NW)
gemm_as_kernel(const AsParams p) {
    constexpr int AS_WAVES_K = NW, AS_THREADS_K = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char as_raw[];
    bf16_t* const wimg = reinterpret_cast<bf16_t*>(as_raw);                // [buffer 2][piece 2][panel KP][AS_PANEL]
    __shared__ uint32_t amax_word;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int row0 = blockIdx.x * (32 * AS_WAVES_K), row_end = p.M, grp = 0;
    if (p.group_offsets) {
        int prev = p.group_offsets[0], tiles_before = 0;
        bool found = false;
        for (int gi = 0; gi < p.groups; ++gi) {
            const int nxt = p.group_offsets[gi + 1];
            const int tiles = (nxt - prev + 32 * AS_WAVES_K - 1) / (32 * AS_WAVES_K);
            if (!found && (int)blockIdx.x < tiles_before + tiles) {
                grp = gi; row0 = prev + ((int)blockIdx.x - tiles_before) * (32 * AS_WAVES_K); row_end = nxt; found = true;
            }
            if (!found) tiles_before += tiles;
            prev = nxt;
        }
        if (!found) return;
    }
    const int m = row0 + w * 32 + r;
    const bool valid_m = m < row_end;
    if (tid == 0) amax_word = 0;

    // ---- this lane's row of A: 8 k-values per k-step, cut once with the row's own scale --------------------------------------
    bf16x8 af[2][KP * 4];
    float row_inv;
    {
        const float* arow = p.A + (int64_t)(valid_m ? m : row_end - 1) * p.lda + 8 * h;
        float4 raw[KP * 8];
#pragma unroll
        for (int s = 0; s < KP * 4; ++s) {
            raw[2 * s] = *reinterpret_cast<const float4*>(arow + 16 * s);
            raw[2 * s + 1] = *reinterpret_cast<const float4*>(arow + 16 * s + 4);
        }
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < KP * 8; ++i) {
            if (!valid_m) raw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            mx = fmaxf(fmaxf(fmaxf(mx, fabsf(raw[i].x)), fabsf(raw[i].y)), fmaxf(fabsf(raw[i].z), fabsf(raw[i].w)));
        }
        mx = xor32_max(mx);                                              // the row's other half
        float s_row = 1.f;
        row_inv = 1.f;
        if (mx > 0.f) scale_from_amax(__float_as_uint(mx), s_row, row_inv);
#pragma unroll
        for (int s = 0; s < KP * 4; ++s) {
            u32x4a u0, u1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 x = raw[2 * s + j];
                uint32_t a0, a1, b0, b1;
                cut2h_quad(x.x, x.y, x.z, x.w, s_row, a0, a1, b0, b1);
                u0[2 * j] = a0; u0[2 * j + 1] = b0; u1[2 * j] = a1; u1[2 * j + 1] = b1;
            }
            af[0][s] = __builtin_bit_cast(bf16x8, u0);
            af[1][s] = __builtin_bit_cast(bf16x8, u1);
        }
    }
    float w_s, w_inv;
    scale_from_amax(amax_read(p.amax_w), w_s, w_inv);
    const float out_scale = row_inv * w_inv * p.alpha;

    // ---- slabs of 64 output columns ---------------------------------------------------------------------------------------------
    const int n_slabs = (p.N + AS_SLAB - 1) / AS_SLAB;
    constexpr int QPR = KP * 16;                                         // 16-byte groups (four k) per row of W
    constexpr int NLD = AS_SLAB * QPR / AS_THREADS_K;                      // groups per thread and slab (KP * 2)
    uint4 rw[NLD], rw2[NLD];
    // per thread: the NLD (row of the slab, four-k group) items it stages, as offsets computed ONCE (a slab later = a constant stride)
    int lds_off[NLD], nl_[NLD];
    const uint4* wsrc[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int f = tid + AS_THREADS_K * i;
        if (!WRC) {
            const int nl = f / QPR, g = f % QPR;
            nl_[i] = nl;
            lds_off[i] = (g >> 4) * AS_PANEL + as_off(nl, (4 * g) & 63);
            wsrc[i] = reinterpret_cast<const uint4*>(p.Wp) + (((int64_t)(grp / p.group_div) * p.strideW + (int64_t)nl * p.ldw + 4 * g) >> 2);
        } else {
            // item f: k row f / (AS_SLAB / 4), quad f % (AS_SLAB / 4) of the slab's columns; image [k][AS_SLAB n] (rows of 2 AS_SLAB bytes:
            // a half wave's transposing read covers four whole rows = 256 bytes = every bank once)
            const int k = f / (AS_SLAB / 4), nq = f % (AS_SLAB / 4);
            nl_[i] = 4 * nq;
            lds_off[i] = k * AS_SLAB + 4 * nq;                             // (16-bit elements)
            wsrc[i] = reinterpret_cast<const uint4*>(p.Wp) + (((int64_t)(grp / p.group_div) * p.strideW + (int64_t)k * p.ldw + 4 * nq) >> 2);
        }
    }
    const int64_t slab_stride = WRC ? (AS_SLAB >> 2) : (((int64_t)AS_SLAB * p.ldw) >> 2);         // uint4 between slabs
    auto load_slab_into = [&](int j, uint4 (&dst)[NLD]) {
        const bool full = (j + 1) * AS_SLAB <= p.N;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const uint4* src = wsrc[i] + (int64_t)j * slab_stride;
            if (!full && j * AS_SLAB + nl_[i] >= p.N) src = wsrc[i];       // (rows past N: any valid address; stored as zeros)
            dst[i] = *src;
        }
    };
    auto store_slab_from = [&](int j, int buf, const uint4 (&rw)[NLD]) {
        bf16_t* img = wimg + buf * (2 * KP * AS_PANEL);
        const bool full = (j + 1) * AS_SLAB <= p.N;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool ok = full || j * AS_SLAB + nl_[i] < p.N;
            *reinterpret_cast<uint2*>(img + lds_off[i]) = ok ? make_uint2(rw[i].x, rw[i].y) : make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(img + KP * AS_PANEL + lds_off[i]) = ok ? make_uint2(rw[i].z, rw[i].w) : make_uint2(0u, 0u);
        }
    };
    auto load_slab = [&](int j) { load_slab_into(j, rw); };
    auto store_slab = [&](int j, int buf) { store_slab_from(j, buf, rw); };
    int row_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) row_off[s] = as_off(r, 16 * s + 8 * h);
    // WRC: lane 16 g + 4 q + pp supplies the address of (k row q of a 4-block, columns 4 pp .. 4 pp + 3) of its group's 4 x 16 block
    const int tr_lane = (8 * h + ((lane >> 2) & 3)) * AS_SLAB + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // (16-bit elements)
    auto read_tr = [&](const bf16_t* base) -> bf16x8 {
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + c * 4 * AS_SLAB));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return out;
    };
    float hd_dot = 0.f;                                                   // EPI 2: the head's dot over the slabs seen so far
    float emax = 0.f;                                                     // EPI 4: max |d gate|, |d up| stored; EPI 5: max |hm|
    float4 gsave[EPI == 5 ? 4 * AS_NT : 1];                               // EPI 5: the gate slab's values (+ table) until the up slab is done
    // EPI 5 with a table: this workgroup's rows are ONE group - its table row waits in LDS (behind the W buffers)
    float* const tbl_lds = reinterpret_cast<float*>(as_raw + (size_t)(AS_DEPTH == 2 ? 3 : 2) * 2 * KP * AS_PANEL * sizeof(bf16_t));
    if (EPI == 5 && p.tbl) {
        const float4* src = reinterpret_cast<const float4*>(p.tbl + (int64_t)grp * p.N);
        for (int i = tid; i < p.N / 4; i += AS_THREADS_K) reinterpret_cast<float4*>(tbl_lds)[i] = src[i];
    }

    float cmax = 0.f;
    float* crow = p.C + (int64_t)m * p.ldc;
    f32x16 acc[AS_NT];
    auto mfma_slab = [&](const int j, const int buf) {
        const bf16_t* img = wimg + buf * (2 * KP * AS_PANEL);
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int nt = 0; nt < AS_NT; ++nt) acc[nt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KP * 4; ++ks) {
                const int pn = ks >> 2, s = ks & 3;
#pragma unroll
                for (int nt = 0; nt < AS_NT; ++nt) {
                    bf16x8 w0, w1;
                    if (!WRC) {
                        w0 = *reinterpret_cast<const bf16x8*>(img + pn * AS_PANEL + nt * 32 * 64 + row_off[s]);
                        w1 = *reinterpret_cast<const bf16x8*>(img + (KP + pn) * AS_PANEL + nt * 32 * 64 + row_off[s]);
                    } else {
                        w0 = read_tr(img + ks * 16 * AS_SLAB + nt * 32 + tr_lane);
                        w1 = read_tr(img + KP * AS_PANEL + ks * 16 * AS_SLAB + nt * 32 + tr_lane);
                    }
                    // smallest piece products first
                    if (AS_ABLATE & 2) { asm volatile("" :: "v"(w0), "v"(w1)); continue; }
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w1), __builtin_bit_cast(f16x8, af[0][ks]), acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, af[1][ks]), acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, af[0][ks]), acc[nt], 0, 0, 0);
                }
            }
    };
    auto epilogue_slab = [&](const int j) {
            // epilogue: lane = output row m, acc[nt][reg] = column slab + 32 nt + (reg & 3) + 8 (reg >> 2) + 4 h
            float part = 0.f;                                             // EPI 2: this lane's share of the slab's row-dot
            if (valid_m) {
                const bool track = EPI != 5 && p.amax_c != nullptr && j * AS_SLAB >= p.amax_c_col0;
#pragma unroll
                for (int nt = 0; nt < AS_NT; ++nt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = j * AS_SLAB + nt * 32 + 8 * g4 + 4 * h;
                        if (col < p.N) {
                            float4 t4 = make_float4(acc[nt][4 * g4] * out_scale, acc[nt][4 * g4 + 1] * out_scale,
                                                    acc[nt][4 * g4 + 2] * out_scale, acc[nt][4 * g4 + 3] * out_scale);
                            if (AS_ABLATE & 1) { asm volatile("" :: "v"(t4.x), "v"(t4.y), "v"(t4.z), "v"(t4.w)); continue; }
                            if (EPI == 4) {
                                // C = d(hm) is not stored: with mult = the dropout multiplier of flat element m N + col, the gate | up values
                                // a, b stored at sw_gu[m][col], sw_gu[m][N + col] become mult C b silu'(a), mult C silu(a)  (gemm.hip, EPI == 4)
                                float* ga = p.sw_gu + (int64_t)m * p.sw_ld + col;
                                const float4 a4 = *reinterpret_cast<const float4*>(ga), b4 = *reinterpret_cast<const float4*>(ga + p.N);
                                const DropoutRng rng(p.p_drop, p.seed);
                                float mu[4];
                                rng.mult4((uint32_t)(((int64_t)m * p.N + col) >> 2), mu);
                                const float d0 = mu[0] * t4.x, d1 = mu[1] * t4.y, d2 = mu[2] * t4.z, d3 = mu[3] * t4.w;
                                const float4 dg = make_float4(d0 * b4.x * dsilu_f(a4.x), d1 * b4.y * dsilu_f(a4.y), d2 * b4.z * dsilu_f(a4.z),
                                                              d3 * b4.w * dsilu_f(a4.w));
                                const float4 du = make_float4(d0 * silu_f(a4.x), d1 * silu_f(a4.y), d2 * silu_f(a4.z), d3 * silu_f(a4.w));
                                *reinterpret_cast<float4*>(ga) = dg;
                                *reinterpret_cast<float4*>(ga + p.N) = du;
                                emax = fmaxf(fmaxf(fmaxf(emax, fmaxf(fabsf(dg.x), fabsf(du.x))), fmaxf(fabsf(dg.y), fabsf(du.y))),
                                             fmaxf(fmaxf(fabsf(dg.z), fabsf(du.z)), fmaxf(fabsf(dg.w), fabsf(du.w))));
                                continue;
                            }
                            if (p.accumulate) {                           // (N % 4 == 0 with accumulate: checked by the launcher)
                                const float4 o = *reinterpret_cast<const float4*>(crow + col);
                                t4.x += o.x; t4.y += o.y; t4.z += o.z; t4.w += o.w;
                            }
                            if (col + 4 <= p.N) {
                                *reinterpret_cast<float4*>(crow + col) = t4;
                            } else {                                              // (columns from N on belong to the caller)
                                const float e[4] = {t4.x, t4.y, t4.z, t4.w};
                                for (int q = 0; q < 4 && col + q < p.N; ++q) crow[col + q] = e[q];
                            }
                            if (EPI == 2) {
                                const float4 o4 = *reinterpret_cast<const float4*>(p.rowdot_other + (int64_t)m * p.ldc + col);
                                part += (t4.x * o4.x + t4.y * o4.y) + (t4.z * o4.z + t4.w * o4.w);
                            }
                            if (EPI == 5) {
                                const int half = p.N >> 1;
                                float4 x4 = t4;
                                if (p.tbl) {
                                    const float4 tb = *reinterpret_cast<const float4*>(tbl_lds + col);
                                    x4.x += tb.x; x4.y += tb.y; x4.z += tb.z; x4.w += tb.w;
                                }
                                if (col < half) {
                                    gsave[EPI == 5 ? nt * 4 + g4 : 0] = x4;
                                } else {
                                    const float4 g = gsave[EPI == 5 ? nt * 4 + g4 : 0];
                                    const int hc = col - half;
                                    const DropoutRng rng(p.p_drop, p.seed);
                                    float mu[4];
                                    rng.mult4((uint32_t)(((int64_t)m * half + hc) >> 2), mu);
                                    const float4 o = make_float4(mu[0] * (silu_f(g.x) * x4.x), mu[1] * (silu_f(g.y) * x4.y),
                                                                 mu[2] * (silu_f(g.z) * x4.z), mu[3] * (silu_f(g.w) * x4.w));
                                    *reinterpret_cast<float4*>(p.hm + (int64_t)m * half + hc) = o;
                                    emax = fmaxf(fmaxf(fmaxf(emax, fabsf(o.x)), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
                                }
                            }
                            if (track) cmax = fmaxf(fmaxf(fmaxf(cmax, fabsf(t4.x)), fabsf(t4.y)), fmaxf(fabsf(t4.z), fabsf(t4.w)));
                        }
                    }
            }
            if (EPI == 2) {
                // a head = 64 columns = 64 / AS_SLAB slabs: the lane pair (m, h = 0 / 1) holds the slab's two halves
                const float sl = xor32_sum(part);
                constexpr int SPH = 64 / AS_SLAB;                          // slabs per head
                hd_dot = (j % SPH == 0) ? sl : hd_dot + sl;
                if (j % SPH == SPH - 1 && valid_m && h == 0) {
                    const int heads = p.N >> 6, head = (j * AS_SLAB) >> 6;
                    p.rowdot_out[((int64_t)(m / p.rowdot_S) * heads + head) * p.rowdot_S + m % p.rowdot_S] = hd_dot;
                }
            }
    };
    if (AS_DEPTH == 1) {
        load_slab(0);
        store_slab(0, 0);
        __syncthreads();
        // (the next slab's loads and stores are UNCONDITIONAL - after the last slab it is requested and stored once more, into the buffer
        // nobody reads any more: under `if (j + 1 < n_slabs)` the loaded registers became a loop-carried merge, the compiler copied
        // them right behind the loads and the kernel waited vmcnt(0) there - the L2 round trip of W in front of every slab)
        // Order inside a slab: request slab j + 1 | MFMAs of slab j | slab j + 1 -> LDS | epilogue of slab j.  The LDS store waits for
        // the W loads with nothing younger in the memory queue; with the epilogue's C stores in front of it the compiler's counted wait
        // (a variable number of stores in branches: it assumes the worst) also waited for the stores it had just issued.
        // EPI 5 walks the slabs gate 0, up 0, gate 1, up 1, ... (the up slab of a column meets its gate slab's values in gsave)
        auto slab_of = [&](int i) { return EPI == 5 ? (i >> 1) + (i & 1) * (n_slabs >> 1) : i; };
#pragma unroll 1
        for (int i = 0; i < n_slabs; ++i) {
            const int j = slab_of(i), buf = i & 1, jn = slab_of(min(i + 1, n_slabs - 1));
            if (!(AS_ABLATE & 4)) load_slab(jn);
            __builtin_amdgcn_sched_barrier(0);             // (or the scheduler sinks the loads to their use, behind the MFMAs)
            mfma_slab(j, buf);
            __builtin_amdgcn_sched_barrier(0);
            if (!(AS_ABLATE & 4)) store_slab(jn, buf ^ 1);
            epilogue_slab(j);
            __syncthreads();
        }
    } else {
        // two slabs in flight: slab j + 1 waits in one register set, j + 2 in the other; three LDS buffers (slab j in buffer j % 3)
        load_slab_into(0, rw);
        store_slab_from(0, 0, rw);
        if (n_slabs > 1) load_slab_into(1, rw);
        if (n_slabs > 2) load_slab_into(2, rw2);
        __syncthreads();
        int bufc = 0;
#pragma unroll 1
        for (int j = 0; j < n_slabs; j += 2) {
            const int b1 = bufc == 2 ? 0 : bufc + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            mfma_slab(j, bufc); epilogue_slab(j);
            if (j + 1 < n_slabs) store_slab_from(j + 1, b1, rw);
            if (j + 3 < n_slabs) load_slab_into(j + 3, rw);
            __syncthreads();
            if (j + 1 >= n_slabs) break;
            mfma_slab(j + 1, b1); epilogue_slab(j + 1);
            if (j + 2 < n_slabs) store_slab_from(j + 2, b2, rw2);
            if (j + 4 < n_slabs) load_slab_into(j + 4, rw2);
            __syncthreads();
            bufc = b2;
        }
    }
    if (p.amax_c) {
        uint32_t mw = __float_as_uint((EPI == 4 || EPI == 5) ? emax : cmax);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
        if (lane == 0 && mw) atomicMax(&amax_word, mw);
        __syncthreads();
        if (tid == 0 && amax_word) amax_publish(amax_word, p.amax_c, blockIdx.x);
    }
}

static std::atomic<long long> g_as_launches{0};

static inline bool gemm_as_enabled() {
    static EnvSwitch sw("GAMER_GEMM_AS");                // (cached: gamer_reload_env() after a change inside the process)
    return sw.get(1) != 0;
}

// Does this descriptor take the activation-stationary kernel?  A k-contiguous with K a multiple of 64 up to 256, packed weight pieces,
// enough rows to fill the chip; W k-contiguous (Linear forward): plain store, one group or the experts' row segments; W row-contiguous
// (input gradient of a layer with 256 output features): K = 256, store / accumulate, row-dot or SwiGLU-backward epilogue.
bool gemm_as_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc, const uint16_t* b_planes) {
    if (!gemm_as_enabled() || !a_kc || !b_planes || !d->amax_b) return false;
    if (d->group_mode != 0 || d->resid || d->qk_q_rot) return false;
    if ((d->group_div > 1 || d->sw_tbl) && !d->sw_hm) return false;      // (row groups that share a W / the row table: the fused SwiGLU forward only)
    if (d->sw_hm) {
        // the experts' gate|up projection with the SwiGLU forward in its epilogue: W k-contiguous, N = 2 I in whole slab pairs, no other epilogue
        static EnvSwitch sw5("GAMER_GEMM_AS_SWIGLU");        // (0: the projection and gamer_swiglu_fwd_ld(_tbl) as two launches - A/B runs)
        if (sw5.get(1) == 0 || !b_kc || d->accumulate || d->rowdot_out || d->sw_gu || d->N % (2 * AS_SLAB) != 0 || d->alpha != 1.f ||
            !aligned16(d->sw_hm) || (d->sw_tbl && (!aligned16(d->sw_tbl) || !d->group_offsets)) || d->p_drop < 0.f || d->p_drop >= 1.f)
            return false;
        if (d->group_div > 1 && (!d->group_offsets || d->groups % d->group_div != 0)) return false;
    }
    // (a workgroup takes 128 rows: below ~16 k rows the launch leaves most CUs idle and the tile kernel is the better fit;
    // GAMER_GEMM_AS_MIN_M lowers the bar for tests)
    static EnvSwitch min_m("GAMER_GEMM_AS_MIN_M");
    if (d->K % 64 != 0 || d->K < 64 || d->K > 256 || d->M < min_m.get(16384)) return false;
    if (d->a_rs % 4 != 0 || d->ldc % 4 != 0 || !aligned16(d->C) || !aligned16(b_planes)) return false;
    if (d->groups != 1 && (!d->group_offsets || d->strideC != 0)) return false;
    if (d->groups == 1 && d->group_offsets) return false;      // (a one-group row window: the tile kernel honours it, this one would take all M rows)
    if (b_kc) {
        if (d->accumulate || d->rowdot_out || d->sw_gu || d->b_rs % 4 != 0) return false;
        // (grouped: the experts' gate|up projection at d_in = 256: 1.37 -> 1.30 ms, W is 1 MB per 128 rows there; the injecting layers'
        // d_in = 320 would need 160 piece registers = one workgroup per CU: measured 1.62 -> 2.4 ms, stays on the tile kernel)
        if (d->groups != 1 && d->amax_c && !d->sw_hm) return false;      // (with sw_hm amax_c is the slot of max |hm|)
        if (d->amax_c && !d->sw_hm && d->amax_c_col0 % AS_SLAB != 0) return false;
        return true;
    }
    // W row-contiguous.  Measured in the step (same box): the plain / accumulating form (the cross block's gate) 0.22 ms per launch
    // faster than the tile kernel; the row-dot and SwiGLU-backward epilogues not (o_proj 0.59 -> 0.65 ms, down projection 1.28 ->
    // 1.31): their inputs are requested in the epilogue of every 32-column slab and waited for there, and requesting them before the
    // slab's MFMAs spills (256 registers: 1.30 -> 2.2 ms).  They stay on the tile kernel unless GAMER_GEMM_AS_RC=2 (tests).
    static EnvSwitch rc("GAMER_GEMM_AS_RC");
    const int rc_mode = rc.get(1);
    if (rc_mode == 0) return false;
    if (d->K != 256 || d->N % 4 != 0 || d->b_ks % 4 != 0) return false;
    if ((d->sw_gu || d->rowdot_out) && rc_mode < 2) return false;
    if (d->sw_gu) {                                    // (gemm.hip checked: alpha = 1, N % 4 == 0, sw_ld >= 2 N, alignment, no other epilogue)
        return !d->amax_c || d->amax_c_col0 == 0;
    }
    if (d->groups != 1) return false;
    if (d->rowdot_out) return !d->accumulate && d->N % 64 == 0 && 64 % AS_SLAB == 0 && (!d->amax_c || d->amax_c_col0 % AS_SLAB == 0);
    return !d->amax_c || (!d->accumulate && d->amax_c_col0 % AS_SLAB == 0);
}

int launch_gemm_as(const gamer_gemm_desc* d, const uint16_t* b_planes, bool b_kc, hipStream_t st) {
    AsParams p;
    p.A = d->A; p.lda = d->a_rs;
    p.Wp = b_planes; p.ldw = b_kc ? d->b_rs : d->b_ks;
    p.C = d->C; p.ldc = d->ldc;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.alpha = d->alpha;
    p.amax_w = d->amax_b;
    p.amax_c = d->amax_c; p.amax_c_col0 = d->amax_c_col0;
    p.groups = d->groups; p.group_offsets = d->groups > 1 ? d->group_offsets : nullptr; p.strideW = d->strideB;
    p.accumulate = d->accumulate;
    p.rowdot_other = d->rowdot_other; p.rowdot_out = d->rowdot_out; p.rowdot_S = d->rowdot_S;
    p.sw_gu = d->sw_gu; p.sw_ld = d->sw_ld; p.p_drop = d->p_drop; p.seed = d->seed;
    p.hm = d->sw_hm; p.tbl = d->sw_hm ? d->sw_tbl : nullptr; p.group_div = d->group_div > 1 ? d->group_div : 1;
    const int kp = d->K / 64;
    const size_t shmem = (size_t)(AS_DEPTH == 2 ? 3 : 2) * 2 * kp * AS_PANEL * sizeof(bf16_t) + (d->sw_hm ? (size_t)d->N * sizeof(float) : 0);
    // eight waves per workgroup for the wide ungrouped Linear forwards (q|k|v, head), four otherwise (see the kernel)
    static EnvSwitch wv("GAMER_GEMM_AS_WAVES");
    const int nw = d->sw_hm ? 4 : (wv.is_set() ? (wv.get(4) == 8 ? 8 : 4) : ((b_kc && d->groups == 1 && d->N >= 512 && AS_WAVES == 4) ? 8 : AS_WAVES));
    const dim3 grid((d->M + 32 * nw - 1) / (32 * nw) + (p.group_offsets ? d->groups : 0));
#define GAMER_LAUNCH_AS_W(KPV, WRCV, EPIV, NWV)                                                                               \
    do {                                                                                                                      \
        static bool attr_dev[MAX_DEVICES] = {};                                                                               \
        if (!attr_dev[current_device()]) {                                                                                    \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_as_kernel<KPV, WRCV, EPIV, NWV>),     \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                 \
            if (e != hipSuccess) { set_error("gamer_gemm_f32_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; } \
            attr_dev[current_device()] = true;                                                                                \
        }                                                                                                                     \
        hipLaunchKernelGGL((gemm_as_kernel<KPV, WRCV, EPIV, NWV>), grid, dim3(64 * NWV), shmem, st, p);                        \
    } while (0)
#define GAMER_LAUNCH_AS(KPV, WRCV, EPIV)                                                                                      \
    do { if (nw == 8) GAMER_LAUNCH_AS_W(KPV, WRCV, EPIV, 8); else GAMER_LAUNCH_AS_W(KPV, WRCV, EPIV, 4); } while (0)
    if (b_kc && d->sw_hm) {
        // (one shared-memory size per instantiation is registered once: always ask for the table's room)
        switch (kp) {
            case 1: GAMER_LAUNCH_AS_W(1, false, 5, 4); break;
            case 2: GAMER_LAUNCH_AS_W(2, false, 5, 4); break;
            case 3: GAMER_LAUNCH_AS_W(3, false, 5, 4); break;
            default: GAMER_LAUNCH_AS_W(4, false, 5, 4); break;
        }
    } else if (b_kc) {
        switch (kp) {
            case 1: GAMER_LAUNCH_AS(1, false, 0); break;
            case 2: GAMER_LAUNCH_AS(2, false, 0); break;
            case 3: GAMER_LAUNCH_AS(3, false, 0); break;
            default: GAMER_LAUNCH_AS(4, false, 0); break;
        }
    } else if (d->sw_gu) {
        GAMER_LAUNCH_AS(4, true, 4);
    } else if (d->rowdot_out) {
        GAMER_LAUNCH_AS(4, true, 2);
    } else {
        GAMER_LAUNCH_AS(4, true, 0);
    }
#undef GAMER_LAUNCH_AS
#undef GAMER_LAUNCH_AS_W
    GAMER_CHECK_LAUNCH("gamer_gemm_f32_split/activation-stationary");
    g_as_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

}  // namespace gamer

// Diagnostic (not in include/gamer_hip.h): launches of the activation-stationary kernel by this process so far.  On well-scaled data
// the kernel returns the BITS of the tile kernel (a power-of-two scale does not change an fp16 rounding), so a test cannot tell from
// the results which of the two ran.
extern "C" long long gamer_debug_gemm_as_launches(void) { return gamer::g_as_launches.load(std::memory_order_relaxed); }
