// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
//   C[m][n] (=|+=) alpha * sum_k A(m,k) * B(n,k)
//
// Workgroup tile 128 x 128 x 32, 4 waves as 2 x 2, each wave a 64 x 64 patch = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator registers).  Operand tiles go global -> registers -> LDS (double
// buffered, one barrier per K-step) with 16-byte accesses on both sides:
//   * K-contiguous operand ("KC", e.g. activations X[t][k] or weights W[n][k]):
//       LDS image [row][32+4]; a lane reads one float4 = 4 consecutive k of its row.  The MFMA's two
//       k slots are the two lane halves, so lane half h takes k = 8*kk + 4*h + s for step s: a
//       permutation of k that is applied identically to both operands.
//   * row-contiguous operand ("RC", e.g. W[k][n] in dgrad, dY[t][m] / X[t][n] in wgrad):
//       LDS image [k][128]; a lane reads 4 scalars (same k permutation), conflict-free because the
//       32 lanes of a half read 32 consecutive floats.
// Grouped forms cover the position-routed experts without host synchronisation: segment offsets are
// read from device memory and surplus workgroups exit.
//
// SPLIT = 6 | 9 (gamer_gemm_f32_split): the same kernel with the products formed on the bf16 matrix pipe, which is
// 16 x wider than the fp32 one.  Every fp32 operand value is cut into three bf16 pieces x = x0 + x1 + x2 - EXACTLY:
// x0 = the upper 16 bits of x (8 significant bits), x1 = the upper 16 bits of x - x0 (the subtraction is exact), x2 =
// x - x0 - x1, which has at most 8 significant bits left - while the tile is written to LDS (three bf16 images per
// operand), and a.b is accumulated as the bf16 products a_i.b_j (each exact in fp32: 8 x 8 significant bits) in the
// fp32 accumulators of v_mfma_f32_32x32x16_bf16: all nine of them (SPLIT = 9: every bit of both mantissas takes part,
// the only rounding left is the fp32 accumulation, as in the fp32 MFMA), or the six with i + j <= 2 (SPLIT = 6: the
// three dropped products are below 2^-24 of |a.b| each - the size of one fp32 rounding).  Inf / NaN operands give NaN.
//
// SPLIT = 3 (gamer_gemm_f32_split, terms = 3): a TWO-way fp16 cut and three products.  Each operand TENSOR has a power-of-two
// scale s = 2^e from its largest magnitude (gamer_absmax_f32 -> `amax_a` / `amax_b`, read here from device memory: max |x| s
// in [2^13, 2^14)); x s = h0 + h1 + r with h0 = fp16(x s), h1 = fp16(x s - h0) (the subtraction is exact), |r| <= 2^-22 |x s|
// for every element within 2^17 of the tensor's largest (smaller ones keep an ABSOLUTE error of 2^-25 of the scaled
// unit, i.e. < 2^-38 of the largest); a.b = (a0.b0 + a0.b1 + a1.b0) / (s_a s_b) in the fp32 accumulators of
// v_mfma_f32_32x32x16_f16 (11 x 11-bit products are exact in fp32; the dropped a1.b1 and the residuals are ~2^-22 |a.b| per term,
// measured against fp64: rms error 1.2e-8 of sum |a_k b_k|, the six-product form's 0.9e-8, the fp32 MFMA's 4.4e-8).  Half the
// MFMAs, two thirds of the LDS bytes and ~60 % of the cut's vector instructions of SPLIT = 6.
#include "common.h"
#include <stdlib.h>

namespace gamer {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int KC_LD = BK + 4;                  // floats per row of a KC tile image
constexpr int TILE_FLOATS = BM * KC_LD;        // 4608 >= BK*BM (RC image)
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * (int)sizeof(float);   // 2 operands x 2 buffers

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- SPLIT mode: LDS images and the exact three-way cut ------------------------------------------------------------
constexpr int SP_KC_ROW = 80;                  // bytes per row of a K-contiguous bf16 image: 32 k + 16 bytes of padding
constexpr int SP_KC_IMG = BM * SP_KC_ROW;      // 10,240
constexpr int SP_RC_IMG = BK * 256;            // [k][128 rows] bf16, 256-byte rows, 16-byte chunks XOR-swizzled: 8,192
constexpr int SP_OPERAND = 3 * SP_KC_IMG;      // one operand = three images (the larger layout)
constexpr int SP_LDS_BYTES = 2 * SP_OPERAND;   // 61,440: ONE stage (two barriers per K-step), two workgroups per CU

#ifndef SP_ABLATE
#define SP_ABLATE 0     // timing-only builds (tools/ablate_gemm.sh): 1 no split arithmetic, 2 no LDS stores, 4 no MFMA, 8 no global loads
#endif
// upper halves of (w1, w0) -> one register {bf16(w0), bf16(w1)}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x07060302u); }
// x = p0 + p1 + p2 exactly, p_i = the value of the upper 16 bits of w[i]
__device__ __forceinline__ void split3(float x, uint32_t (&w)[3]) {
    if (SP_ABLATE & 1) { w[0] = w[1] = w[2] = __float_as_uint(x); return; }
    w[0] = __float_as_uint(x);
    const float r1 = x - __uint_as_float(w[0] & 0xffff0000u);
    w[1] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(w[1] & 0xffff0000u);
    w[2] = __float_as_uint(r2);
}
// the three bf16 images of 4 consecutive values: img[s] = {lo dword, hi dword}
__device__ __forceinline__ void split3_quad(const float4& v, uint2 (&img)[3]) {
    uint32_t a[3], b[3], c[3], d[3];
    split3(v.x, a); split3(v.y, b); split3(v.z, c); split3(v.w, d);
#pragma unroll
    for (int s = 0; s < 3; ++s) img[s] = make_uint2(pack_hi16(a[s], b[s]), pack_hi16(c[s], d[s]));
}
// byte offset of (k, row) in an RC image: chunk (row >> 3) of the k row is XORed with (k & 3) << 2 so that both the
// 8-byte stores of a float4's pieces and the transposing reads spread over the banks
__device__ __forceinline__ int sp_rc_off(int k, int row) { return k * 256 + ((((row >> 3) ^ ((k & 3) << 2))) << 4) + ((row & 7) << 1); }
__device__ __forceinline__ void store_kc_split(unsigned char* __restrict__ lds, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[3];
        split3_quad(r[j], img);
        unsigned char* dst = lds + (f >> 3) * SP_KC_ROW + ((f & 7) << 3);
        if (SP_ABLATE & 2) { asm volatile("" :: "v"(img[0].x), "v"(img[1].x), "v"(img[2].x), "v"(img[0].y), "v"(img[1].y), "v"(img[2].y)); continue; }
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(dst + s * SP_KC_IMG) = img[s];
    }
}
__device__ __forceinline__ void store_rc_split(unsigned char* __restrict__ lds, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[3];
        split3_quad(r[j], img);
        unsigned char* dst = lds + sp_rc_off(f >> 5, (f & 31) << 2);
        if (SP_ABLATE & 2) { asm volatile("" :: "v"(img[0].x), "v"(img[1].x), "v"(img[2].x), "v"(img[0].y), "v"(img[1].y), "v"(img[2].y)); continue; }
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(dst + s * SP_RC_IMG) = img[s];
    }
}
// ---- SPLIT == 3: two fp16 pieces of x * s (s = the tensor's power-of-two scale); images 0 and 1 of the three-image layouts
__device__ __forceinline__ void split2h_quad(const float4& v, float s, uint2 (&img)[2]) {
    uint32_t a0, a1, b0, b1;
    cut2h_quad(v.x, v.y, v.z, v.w, s, a0, a1, b0, b1);
    img[0] = make_uint2(a0, b0);
    img[1] = make_uint2(a1, b1);
}
__device__ __forceinline__ void store_kc_h2(unsigned char* __restrict__ lds, int tid, const float4 (&r)[4], float s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[2];
        split2h_quad(r[j], s, img);
        unsigned char* dst = lds + (f >> 3) * SP_KC_ROW + ((f & 7) << 3);
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(dst + q * SP_KC_IMG) = img[q];
    }
}
__device__ __forceinline__ void store_rc_h2(unsigned char* __restrict__ lds, int tid, const float4 (&r)[4], float s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[2];
        split2h_quad(r[j], s, img);
        unsigned char* dst = lds + sp_rc_off(f >> 5, (f & 31) << 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(dst + q * SP_RC_IMG) = img[q];
    }
}
// ---- two-stage form (SP_TWO_STAGE): K-steps of 16, each stage = both operands' three images of a 128 x 16 slice.  A KC image
// row is 32 B of k + 16 B of padding (48-byte rows: the 16-byte fragment reads of 16 consecutive rows fall on distinct banks),
// an RC image is [16 k][128 rows] with the chunk swizzle of sp_rc_off.  2 x 2 x 3 x 6144 B = 73,728 B = the fp32 kernel's
// allocation, two workgroups per CU.
#ifndef SP_TWO_STAGE
#define SP_TWO_STAGE 0
#endif
#ifndef SP_PREFETCH2
#define SP_PREFETCH2 0  // one-stage form with the global loads two K-steps ahead (two register sets, ~230 VGPRs)
#endif
#ifndef SP_WIDE3
#define SP_WIDE3 1      // SPLIT == 3, forward layout: two 32-deep slices (two LDS stages) per barrier pair (the other layouts: measured slower)
#endif
#ifndef SP_PINGPONG
#define SP_PINGPONG 0   // instantiate the eight-wave ping-pong form (PP below); GAMER_GEMM_PP=1 then selects it
#endif
// The three forms above are bit-identical to the default one-stage form and were each measured within +-3 % of it on every
// GEMM shape of the step (docs/DESIGN_rounds1-4.md section 13): the split kernels run against the matrix pipe's POWER limit, not against
// their loop structure (tools/ubench_memtime.hip: random operand bits sustain 58-67 % of the nominal bf16 rate).
constexpr int SPH_KC_ROW = 48;
constexpr int SPH_KC_IMG = BM * SPH_KC_ROW;    // 6,144
constexpr int SPH_RC_IMG = 16 * 256;           // 4,096
constexpr int SPH_OPERAND = 3 * SPH_KC_IMG;    // (the larger layout)
constexpr int SPH_STAGE = 2 * SPH_OPERAND;     // 36,864
// KC slice: 128 rows x 16 k = 512 float4; f = tid + 256 j: row f >> 2, k-quad f & 3
__device__ __forceinline__ void store_kc_half(unsigned char* __restrict__ lds, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[3];
        split3_quad(r[j], img);
        unsigned char* dst = lds + (f >> 2) * SPH_KC_ROW + ((f & 3) << 3);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(dst + s * SPH_KC_IMG) = img[s];
    }
}
// RC slice: 16 k x 128 rows = 512 float4; f: k f >> 5, row-quad f & 31
__device__ __forceinline__ void store_rc_half(unsigned char* __restrict__ lds, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = tid + GEMM_THREADS * j;
        uint2 img[3];
        split3_quad(r[j], img);
        unsigned char* dst = lds + sp_rc_off(f >> 5, (f & 31) << 2);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(dst + s * SPH_RC_IMG) = img[s];
    }
}
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag_half(const unsigned char* __restrict__ img, int rbase, int lane) {
    const int r32 = lane & 31, h = lane >> 5;
    if (KC) {
        return *reinterpret_cast<const bf16x8*>(img + (rbase + r32) * SPH_KC_ROW + (h << 4));
    } else {
        const int q = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
        const int row = rbase + 16 * gsel + 4 * pp;
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = 8 * h + 4 * c + q;
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(img + sp_rc_off(k, row)));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return out;
    }
}

// fragment of `v_mfma_f32_32x32x16_bf16`: lane (r32, h) holds k = 16 * sub + 8 * h + 0..7 of operand row `rbase + r32`
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag_split(const unsigned char* __restrict__ img, int rbase, int sub, int lane) {
    const int r32 = lane & 31, h = lane >> 5;
    if (KC) {
        return *reinterpret_cast<const bf16x8*>(img + (rbase + r32) * SP_KC_ROW + ((2 * sub + h) << 4));
    } else {
        // transposing read: lane 16g + 4q + pp supplies the address of (k 4-block row q, rows 4pp..4pp+3) of its group's
        // 4 x 16 block and receives row (lane & 15) of the four k
        const int q = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
        const int row = rbase + 16 * gsel + 4 * pp;
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = 16 * sub + 8 * h + 4 * c + q;
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(img + sp_rc_off(k, row)));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return out;
    }
}

// Diagnostic build only (STAMP = true, selected by gamer_debug_gemm_stamp): wave 0 of every workgroup
// accumulates shader-clock cycles per phase of the K loop into g_gemm_stamp[8 * blockIdx.x + phase].
__device__ unsigned long long* g_gemm_stamp = nullptr;
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

struct GemmParams {
    const float* A; int64_t a_rs, a_ks;
    const float* B; int64_t b_rs, b_ks;
    float* C; int64_t ldc;
    int M, N, K;
    float alpha;
    int accumulate;
    int groups;
    const int32_t* group_offsets;
    int64_t strideB, strideC;
    int kchunk;
    int m_tiles, n_tiles;      // tile counts of the C matrix (mode 1) / upper bound of m tiles (mode 0)
    // fused epilogue (mode 0): C[map(row)] = resid[map(row)] + dropout(alpha*acc)
    const float* resid;
    const int32_t* row_map;
    float p_drop;
    uint64_t seed;
    // row-dot epilogue (EPI == 2): rowdot_out[b][head][i] = sum over the head's 64 columns of C[m][.] * other[m][.]
    const float* rowdot_other;
    float* rowdot_out;
    int rowdot_S;
    // q|k|v epilogue (EPI == 3): per-head RMSNorm + RoPE of the q and k heads while the tile is in LDS (see gamer_gemm_desc)
    const float* qk_wq; const float* qk_wk;
    float qk_eps;
    const float* qk_cos; const float* qk_sin;
    const float* qk_bias_q; const float* qk_bias_k; const float* qk_bias_v;
    const int32_t* qk_act_idx;
    const int32_t* qk_pos_ids;
    float* qk_q_rot; float* qk_k_rot;
    int qk_S, qk_nq, qk_nkv;
    const uint32_t* amax_a; const uint32_t* amax_b;   // SPLIT == 3: bits of max |A|, max |B| (gamer_absmax_f32)
    const uint16_t* b_planes;      // SPLIT, MODE 0: B pre-cut into three bf16 planes indexed like B (nullptr: cut in the kernel)
    int64_t b_plane_stride;
    int guard;                     // SPLIT == 3, MODE 0: row-range guard of the A operand (gamer_split3_guard; default on)
    uint32_t* amax_c;              // MODE 0, plain / row-dot epilogues: fold the bits of max |C stored| over columns >= amax_c_col0 into this slot
    int amax_c_col0;
    float* wgrad_ws;               // MODE 1, deterministic form: chunk partial tiles [chunk][tile][128][128] instead of fp32 atomics
    // SwiGLU-backward epilogue (EPI == 4): C = d(hm) is NOT stored; with m = the dropout multiplier of flat element row N + col,
    // sw_gu[row][col] <- m C b silu'(a), sw_gu[row][N + col] <- m C silu(a) for a, b = the gate / up values stored there
    float* sw_gu; int64_t sw_ld;
    int group_div;                 // MODE 0 with group_offsets: group g multiplies with B + (g / group_div) * strideB (>= 1)
    const float* sw_tbl;           // EPI == 4: [groups][2 N] added to the gate | up values of group g's rows before the backward
};

// SPLIT == 3 scales a whole operand TENSOR by one power of two, so a row of A whose largest magnitude lies more than 2^16 below
// the tensor's keeps only an absolute precision (2^-38 of the tensor's maximum): relative to that row's own result the error
// grows as 2^(r - 38) for a row 2^r below the maximum - an outlier element or a loss spike anywhere in the tensor does that to
// every other row, and the input-gradient GEMMs see it on low-gradient rows.  Guard (Linear-forward and input-gradient
// layouts, where a row of A is a row of C): every thread tracks max |a| of the rows it stages (one v_max3 per two values), and
// a tile that holds a non-zero row below 2^-16 of the tensor's maximum is computed AGAIN from the fp32 operands in global
// memory with v_mfma_f32_32x32x2_f32 (exact fp32 products; slow, rare) before the epilogue - decided on the device, no host
// synchronisation, results of unaffected tiles bit-identical to the unguarded kernel.  Not guarded: the B operand (parameters;
// their rows and columns stay within a few powers of two of each other) and the weight-gradient layout (its contraction runs
// over the tokens: a small row contributes a small term, the error stays relative to sum |a_k b_k|).
constexpr float SPLIT3_GUARD_RATIO = 1.f / 65536.f;
#ifndef GAMER_GEMM_CAMAX_BUILD
#define GAMER_GEMM_CAMAX_BUILD 1        // 0: timing-only builds without the amax_c epilogue code (tools: what it costs the kernels)
#endif
#ifndef GAMER_SPLIT3_GUARD_BUILD
#define GAMER_SPLIT3_GUARD_BUILD 1      // 0: timing-only builds without the row-maximum tracking (tools: what the guard costs)
#endif
static int g_split3_guard = 1;

// blockIdx -> logical tile id such that consecutive logical ids run on one XCD (ids are dealt
// round-robin over the 8 XCDs); bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, idx = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- global -> registers ---------------------------------------------------------------------
// KC tile: 128 rows x 32 k; float4 f = tid + 256*j covers row f>>3, k-quad f&7.
// INTERIOR = the whole 128 x 32 tile is in range: no predicates at all.  On gfx950 the fp32 MFMA runs at
// the vector rate and every VALU instruction issued beside it costs matrix throughput (measured: a
// branch-free clamp+select version of these loads, ~100 more VALU per K-step, lowered the GEMM from 98 to
// 93 TFLOP/s), so the common path keeps the instruction count minimal.
template <bool INTERIOR>
__device__ __forceinline__ void load_kc(const float* __restrict__ p, int64_t rs, int row0, int row_end, int k0,
                                        int k_end, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        const int row = row0 + (f >> 3);
        const int k = k0 + ((f & 7) << 2);
        if (INTERIOR) {
            r[j] = *reinterpret_cast<const float4*>(p + (int64_t)row * rs + k);
            continue;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < row_end && k < k_end) {
            v = *reinterpret_cast<const float4*>(p + (int64_t)row * rs + k);
            if (k + 3 >= k_end) {      // ragged K tail (leading dim is a multiple of 4, so the load is in bounds)
                if (k + 1 >= k_end) v.y = 0.f;
                if (k + 2 >= k_end) v.z = 0.f;
                v.w = 0.f;
            }
        }
        r[j] = v;
    }
}
// RC tile: 32 k x 128 rows; float4 f covers k f>>5, row-quad f&31.
template <bool INTERIOR>
__device__ __forceinline__ void load_rc(const float* __restrict__ p, int64_t ks, int row0, int row_end, int k0,
                                        int k_end, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        const int k = k0 + (f >> 5);
        const int row = row0 + ((f & 31) << 2);
        if (INTERIOR) {
            r[j] = *reinterpret_cast<const float4*>(p + (int64_t)k * ks + row);
            continue;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_end && row < row_end) {
            v = *reinterpret_cast<const float4*>(p + (int64_t)k * ks + row);
            if (row + 3 >= row_end) {
                if (row + 1 >= row_end) v.y = 0.f;
                if (row + 2 >= row_end) v.z = 0.f;
                v.w = 0.f;
            }
        }
        r[j] = v;
    }
}
__device__ __forceinline__ void store_kc(float* __restrict__ lds, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        *reinterpret_cast<float4*>(lds + (f >> 3) * KC_LD + ((f & 7) << 2)) = r[j];
    }
}
__device__ __forceinline__ void store_rc(float* __restrict__ lds, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + GEMM_THREADS * j;
        *reinterpret_cast<float4*>(lds + (f >> 5) * BM + ((f & 31) << 2)) = r[j];
    }
}

template <bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int row, int kk, int h, float (&f)[4]) {
    if (KC) {
        const float4 v = *reinterpret_cast<const float4*>(lds + row * KC_LD + 8 * kk + 4 * h);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = lds[(8 * kk + 4 * h + s) * BM + row];
    }
}

// EPI: 0 plain, 1 residual + dropout (Linear-forward sites of the decoder layer), 2 row-dot ("delta = dO . O" of the
// attention backward, taken while the o_proj dgrad tile is still in LDS: one pass over dO and one over O less),
// 3 q|k|v projection: (+ behaviour bias,) per-head RMSNorm * weight and RoPE of the q / k heads written to q_rot / k_rot
// next to the raw q|k|v - the row-major rewrite hands 16 lanes x float4 = one head row, the layout of qknorm_rope_fwd
// PP ("ping-pong", SPLIT forms): ONE workgroup of eight waves per CU = two groups of four waves, each group computing its
// own 128 x 128 tile from its own LDS region with the K loop of the four-wave kernel - but group 1 runs half a K-step behind
// group 0, and every barrier is a barrier of all eight waves: while one group's waves cut and store their tile, the other
// group's issue MFMAs, by construction.  (Two independent four-wave workgroups on a CU fall INTO phase instead - both in
// their MFMA phase at half rate each, then both in their store phase with the matrix pipe idle: tools/stamp_gemm_split.py.)
template <bool A_KC, bool B_KC, int MODE, bool ACCUM, bool STAMP, int NBUF, int EPI, int SPLIT = 0, bool PP = false>
__device__ __forceinline__ void
gemm_f32_tile(const GemmParams& p, float* __restrict__ smem_all, const int vblock, const int vgrid) {
    // one tile (PP: two) of the launch: `vblock` of `vgrid` is what blockIdx.x of gridDim.x is in a one-tile-per-workgroup launch
    // buffer b: A image at smem + 2*b*TILE_FLOATS, B image right behind it

    const int grp = PP ? (int)(threadIdx.x >> 8) : 0;             // wave-uniform
    const int tid = threadIdx.x & (GEMM_THREADS - 1);
    float* const smem = smem_all + (PP ? grp * (GEMM_LDS_BYTES / 4) : 0);
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int Lx = xcd_remap(vblock, vgrid);
    const int L = (PP && MODE == 0) ? 2 * Lx + grp : Lx;           // mode 0: the two groups take neighbouring tiles

    // ---- which tile / which group -------------------------------------------------------------
    int row0, row_end;          // A/C row range of this tile (mode 0) or C row tile (mode 1)
    int col0;                   // C column tile start
    int kbeg, kend;             // contraction range
    const float* Bp = p.B;
    const float* Bsrc = p.B;    // the fp32 values of this tile's B (Bp may point at their packed pieces)
    float* Cp = p.C;
    int64_t ws_block = 0;       // MODE 1: index of this workgroup's partial tile in p.wgrad_ws
    int tile_group = 0;         // MODE 0: the row group of this tile (EPI 4: its row of sw_tbl)
    if (MODE == 0) {
        const int mt = L / p.n_tiles;
        col0 = (L % p.n_tiles) * BN;
        int g = 0, seg_beg = 0, seg_end = p.M, tiles_before = 0;
        bool found = false;
        if (p.group_offsets) {
            int prev = p.group_offsets[0];
            for (int gi = 0; gi < p.groups; ++gi) {
                const int nxt = p.group_offsets[gi + 1];
                const int tiles = (nxt - prev + BM - 1) / BM;
                if (!found && mt < tiles_before + tiles) {
                    g = gi; seg_beg = prev; seg_end = nxt; found = true;
                }
                if (!found) tiles_before += tiles;
                prev = nxt;
            }
        } else {
            found = mt < (p.M + BM - 1) / BM;
        }
        if (!found) return;
        row0 = seg_beg + (mt - tiles_before) * BM;
        row_end = seg_end;
        kbeg = 0;
        kend = p.K;
        tile_group = g;
        Bp += (int64_t)(g / p.group_div) * p.strideB;
        // SPLIT == 3 with pre-cut parameters: `b_planes` holds, at the SAME offsets as B, 16 bytes per four elements = their four
        // h0 pieces and their four h1 pieces (gamer_split2h_planes_multi) - the loads below stay what they are, the B half of the
        // cut becomes two 8-byte LDS stores
        Bsrc = Bp;
        if (SPLIT == 3 && p.b_planes) Bp = reinterpret_cast<const float*>(p.b_planes) + (Bp - p.B);
    } else {
        const int tiles_mn = p.m_tiles * p.n_tiles;
        const int per_chunk = PP ? (tiles_mn + 1) / 2 : tiles_mn;  // PP: the two groups take two C tiles of the SAME k chunk
        const int chunk = L / per_chunk;
        const int tile = PP ? 2 * (L % per_chunk) + grp : L % per_chunk;
        if (PP && tile >= tiles_mn) return;                        // (terminated waves do not take part in barriers)
        ws_block = (int64_t)chunk * tiles_mn + tile;
        row0 = (tile / p.n_tiles) * BM;
        row_end = p.M;
        col0 = (tile % p.n_tiles) * BN;
        int g = 0, seg_beg = 0, seg_end = p.K, chunks_before = 0;
        bool found = false;
        if (p.group_offsets) {
            int prev = p.group_offsets[0];
            for (int gi = 0; gi < p.groups; ++gi) {
                const int nxt = p.group_offsets[gi + 1];
                const int chunks = (nxt - prev + p.kchunk - 1) / p.kchunk;
                if (!found && chunk < chunks_before + chunks) {
                    g = gi; seg_beg = prev; seg_end = nxt; found = true;
                }
                if (!found) chunks_before += chunks;
                prev = nxt;
            }
        } else {
            found = chunk < (p.K + p.kchunk - 1) / p.kchunk;
        }
        if (!found) return;
        kbeg = seg_beg + (chunk - chunks_before) * p.kchunk;
        kend = min(seg_end, kbeg + p.kchunk);
        Cp += (int64_t)g * p.strideC;
    }
    const int col_end = p.N;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (kend - kbeg + BK - 1) / BK;
    const bool a_in = row0 + BM <= row_end, b_in = col0 + BN <= col_end;     // workgroup-uniform
    float4 ra[4], rb[4];
    float scale_a = 1.f, scale_b = 1.f, unscale = 1.f;
    constexpr bool GUARD = GAMER_SPLIT3_GUARD_BUILD && SPLIT == 3 && MODE == 0 && A_KC && !PP && !STAMP;
    float rmax[4] = {0.f, 0.f, 0.f, 0.f};          // GUARD: max |a| of this thread's part of rows (tid >> 3) + 32 j
    uint32_t amax_a_bits = 0;
    if (SPLIT == 3) {
        float ia, ib;
        amax_a_bits = amax_read(p.amax_a);
        scale_from_amax(amax_a_bits, scale_a, ia);
        scale_from_amax(amax_read(p.amax_b), scale_b, ib);
        unscale = ia * ib;
    }
    auto load_tile = [&](int k0) {
        const bool k_in = k0 + BK <= kend;
        if (a_in && k_in) {
            if (A_KC) load_kc<true>(p.A, p.a_rs, row0, row_end, k0, kend, tid, ra);
            else load_rc<true>(p.A, p.a_ks, row0, row_end, k0, kend, tid, ra);
        } else {
            if (A_KC) load_kc<false>(p.A, p.a_rs, row0, row_end, k0, kend, tid, ra);
            else load_rc<false>(p.A, p.a_ks, row0, row_end, k0, kend, tid, ra);
        }
        if (b_in && k_in) {
            if (B_KC) load_kc<true>(Bp, p.b_rs, col0, col_end, k0, kend, tid, rb);
            else load_rc<true>(Bp, p.b_ks, col0, col_end, k0, kend, tid, rb);
        } else {
            if (B_KC) load_kc<false>(Bp, p.b_rs, col0, col_end, k0, kend, tid, rb);
            else load_rc<false>(Bp, p.b_ks, col0, col_end, k0, kend, tid, rb);
        }
    };
    // SPLIT == 3: two images per operand, B right behind A's two (a stage is 2 (A_IMG + B_IMG) bytes: 40,960 / 36,864 / 32,768 for
    // the forward / input-gradient / weight-gradient layouts)
    constexpr int H2_B_OFF = 2 * (A_KC ? SP_KC_IMG : SP_RC_IMG);
    constexpr int H2_STAGE = H2_B_OFF + 2 * (B_KC ? SP_KC_IMG : SP_RC_IMG);
    auto store_tile_from = [&](float* dst, const float4 (&qa)[4], const float4 (&qb)[4]) {
        if (SPLIT == 3) {
            unsigned char* d8 = reinterpret_cast<unsigned char*>(dst);
            if (A_KC) store_kc_h2(d8, tid, qa, scale_a); else store_rc_h2(d8, tid, qa, scale_a);
            if (GUARD) {
                // two v_max3_f32 per float4 (|x| is a source modifier; the C form costs a canonicalising v_max per operand: 7 per float4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|"
                        : "+v"(rmax[j]) : "v"(qa[j].x), "v"(qa[j].y), "v"(qa[j].z), "v"(qa[j].w));
            }
            if (MODE == 0 && p.b_planes) {          // packed pieces: {h0 x 4 | h1 x 4} per float4 slot
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = tid + GEMM_THREADS * j;
                    unsigned char* dstb = B_KC ? d8 + H2_B_OFF + (f >> 3) * SP_KC_ROW + ((f & 7) << 3)
                                               : d8 + H2_B_OFF + sp_rc_off(f >> 5, (f & 31) << 2);
                    *reinterpret_cast<uint2*>(dstb) = make_uint2(__float_as_uint(qb[j].x), __float_as_uint(qb[j].y));
                    *reinterpret_cast<uint2*>(dstb + (B_KC ? SP_KC_IMG : SP_RC_IMG)) = make_uint2(__float_as_uint(qb[j].z), __float_as_uint(qb[j].w));
                }
                return;
            }
            if (B_KC) store_kc_h2(d8 + H2_B_OFF, tid, qb, scale_b); else store_rc_h2(d8 + H2_B_OFF, tid, qb, scale_b);
            return;
        }
        if (SPLIT) {
            unsigned char* d8 = reinterpret_cast<unsigned char*>(dst);
            if (A_KC) store_kc_split(d8, tid, qa); else store_rc_split(d8, tid, qa);
            if (SP_ABLATE & 16) { asm volatile("" :: "v"(qb[0].x), "v"(qb[1].y), "v"(qb[2].z), "v"(qb[3].w)); return; }   // timing-only: B never enters LDS
            if (B_KC) store_kc_split(d8 + SP_OPERAND, tid, qb); else store_rc_split(d8 + SP_OPERAND, tid, qb);
            return;
        }
        if (A_KC) store_kc(dst, tid, qa); else store_rc(dst, tid, qa);
        if (B_KC) store_kc(dst + TILE_FLOATS, tid, qb); else store_rc(dst + TILE_FLOATS, tid, qb);
    };
    auto store_tile = [&](float* dst) { store_tile_from(dst, ra, rb); };
    unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t0 = 0, t1 = 0;
    const unsigned long long t_begin = STAMP ? stamp_now() : 0;
    // a wave whose 64 x 64 patch lies entirely past the edge of C (N = 320 leaves half of the last column tile
    // empty) issues no MFMAs: its share of the matrix pipe goes to the other resident workgroup
    const bool wave_live = (row0 + wm * 64 < row_end) && (col0 + wn * 64 < col_end);
    auto mfma_tile = [&](const float* as) {
        if (!wave_live) return;
        if (SPLIT != 0 && (SP_ABLATE & 4) != 0) return;
        if (SPLIT == 3) {
            const unsigned char* a8 = reinterpret_cast<const unsigned char*>(as);
            const unsigned char* b8 = a8 + H2_B_OFF;
            constexpr int A_IMG = A_KC ? SP_KC_IMG : SP_RC_IMG, B_IMG = B_KC ? SP_KC_IMG : SP_RC_IMG;
#pragma unroll
            for (int sub = 0; sub < BK / 16; ++sub) {
                f16x8 af[2][2], bf[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i][q] = __builtin_bit_cast(f16x8, read_frag_split<A_KC>(a8 + q * A_IMG, wm * 64 + i * 32, sub, lane));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[j][q] = __builtin_bit_cast(f16x8, read_frag_split<B_KC>(b8 + q * B_IMG, wn * 64 + j * 32, sub, lane));
                }
                // a0.b1, a1.b0 (small) first, then a0.b0; the weight-gradient layout: a1.b0, a0.b0, a0.b1 - the order of gemm_wg.hip,
                // whose single fragment set needs it (the step's sums are added to a running fp32 total either way)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int qa = MODE == 1 ? (t == 0 ? 1 : 0) : (t == 1 ? 1 : 0), qb = MODE == 1 ? (t == 2 ? 1 : 0) : (t == 0 ? 1 : 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][qa], bf[j][qb], acc[i][j], 0, 0, 0);
                }
            }
            return;
        }
        if (SPLIT) {
            const unsigned char* a8 = reinterpret_cast<const unsigned char*>(as);
            const unsigned char* b8 = a8 + SP_OPERAND;
            constexpr int A_IMG = A_KC ? SP_KC_IMG : SP_RC_IMG, B_IMG = B_KC ? SP_KC_IMG : SP_RC_IMG;
#pragma unroll
            for (int sub = 0; sub < BK / 16; ++sub) {
                bf16x8 af[2][3], bf[2][3];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i][s] = read_frag_split<A_KC>(a8 + s * A_IMG, wm * 64 + i * 32, sub, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        bf[j][s] = (SP_ABLATE & 16) ? af[j][s] : read_frag_split<B_KC>(b8 + s * B_IMG, wn * 64 + j * 32, sub, lane);
                }
                // smallest products first; sa + sb <= 2 for SPLIT == 6, all nine for SPLIT == 9
#pragma unroll
                for (int order = (SPLIT == 9 ? 4 : 2); order >= 0; --order)
#pragma unroll
                    for (int sa = 0; sa < 3; ++sa) {
                        const int sb = order - sa;
                        if (sb < 0 || sb > 2) continue;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][sa], bf[j][sb], acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
        // measured (tools/ab_libs.sh): +0.5-1 % on the forward / dgrad layouts, -2 % on the split-K wgrad
        if (MODE == 0) __builtin_amdgcn_s_setprio(1);
        const float* bs = as + TILE_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float af[2][4], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) read_frag<A_KC>(as, wm * 64 + i * 32 + r32, kk, h, af[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) read_frag<B_KC>(bs, wn * 64 + j * 32 + r32, kk, h, bf[j]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (MODE == 0) __builtin_amdgcn_s_setprio(0);
    };
    const bool fast = a_in && b_in && ((kend - kbeg) % BK == 0);
    if (fast) {
        // Interior tile: no predicates, per-thread base pointers advanced by BK per step, the eight 16-byte
        // loads of the next step are the only address-dependent instructions in the loop.
        const float* pa = A_KC ? p.A + (int64_t)(row0 + (tid >> 3)) * p.a_rs + kbeg + ((tid & 7) << 2)
                               : p.A + (int64_t)(kbeg + (tid >> 5)) * p.a_ks + row0 + ((tid & 31) << 2);
        const float* pb = B_KC ? Bp + (int64_t)(col0 + (tid >> 3)) * p.b_rs + kbeg + ((tid & 7) << 2)
                               : Bp + (int64_t)(kbeg + (tid >> 5)) * p.b_ks + col0 + ((tid & 31) << 2);
        const int64_t ja = A_KC ? 32 * p.a_rs : 8 * p.a_ks;       // float4 #j sits 32 rows (KC) / 8 k (RC) further
        const int64_t jb = B_KC ? 32 * p.b_rs : 8 * p.b_ks;
        const int64_t sa = A_KC ? BK : BK * p.a_ks;               // one K-step
        const int64_t sb = B_KC ? BK : BK * p.b_ks;
        auto fast_load = [&]() {
            if (SPLIT != 0 && (SP_ABLATE & 8) != 0) return;
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4*>(pa + j * ja);
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const float4*>(pb + j * jb);
            pa += sa;
            pb += sb;
        };
        if (SPLIT && SPLIT != 3 && SP_TWO_STAGE && !PP && !(MODE == 0 && p.b_planes != nullptr)) {
            // Two LDS stages of 16-k slices: while the MFMAs of slice j run from one stage, slice j + 1 is cut and stored into
            // the other and slice j + 2 is in flight from global memory - ONE barrier per slice and no phase in which a whole
            // workgroup only stores (the one-stage form below: store | barrier | multiply | barrier).
            const float* pa2 = A_KC ? p.A + (int64_t)(row0 + (tid >> 2)) * p.a_rs + kbeg + ((tid & 3) << 2)
                                    : p.A + (int64_t)(kbeg + (tid >> 5)) * p.a_ks + row0 + ((tid & 31) << 2);
            const float* pb2 = B_KC ? Bp + (int64_t)(col0 + (tid >> 2)) * p.b_rs + kbeg + ((tid & 3) << 2)
                                    : Bp + (int64_t)(kbeg + (tid >> 5)) * p.b_ks + col0 + ((tid & 31) << 2);
            const int64_t ja2 = A_KC ? 64 * p.a_rs : 8 * p.a_ks;      // float4 #j sits 64 rows (KC) / 8 k (RC) further
            const int64_t jb2 = B_KC ? 64 * p.b_rs : 8 * p.b_ks;
            const int64_t sa2 = A_KC ? 16 : 16 * p.a_ks;              // one 16-k slice
            const int64_t sb2 = B_KC ? 16 : 16 * p.b_ks;
            const int n16 = (kend - kbeg) / 16;
            float4 a0[2], b0[2], a1[2], b1[2];
            auto load16 = [&](float4 (&ra_)[2], float4 (&rb_)[2]) {
#pragma unroll
                for (int j = 0; j < 2; ++j) ra_[j] = *reinterpret_cast<const float4*>(pa2 + j * ja2);
#pragma unroll
                for (int j = 0; j < 2; ++j) rb_[j] = *reinterpret_cast<const float4*>(pb2 + j * jb2);
                pa2 += sa2;
                pb2 += sb2;
            };
            auto store16 = [&](unsigned char* st8, const float4 (&ra_)[2], const float4 (&rb_)[2]) {
                if (A_KC) store_kc_half(st8, tid, ra_); else store_rc_half(st8, tid, ra_);
                if (B_KC) store_kc_half(st8 + SPH_OPERAND, tid, rb_); else store_rc_half(st8 + SPH_OPERAND, tid, rb_);
            };
            auto mfma16 = [&](const unsigned char* st8) {
                if (!wave_live) return;
                const unsigned char* a8 = st8;
                const unsigned char* b8 = st8 + SPH_OPERAND;
                constexpr int A_IMG = A_KC ? SPH_KC_IMG : SPH_RC_IMG, B_IMG = B_KC ? SPH_KC_IMG : SPH_RC_IMG;
                bf16x8 af[2][3], bf[2][3];
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i][s3] = read_frag_half<A_KC>(a8 + s3 * A_IMG, wm * 64 + i * 32, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[j][s3] = read_frag_half<B_KC>(b8 + s3 * B_IMG, wn * 64 + j * 32, lane);
                }
#pragma unroll
                for (int order = (SPLIT == 9 ? 4 : 2); order >= 0; --order)
#pragma unroll
                    for (int sa3 = 0; sa3 < 3; ++sa3) {
                        const int sb3 = order - sa3;
                        if (sb3 < 0 || sb3 > 2) continue;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][sa3], bf[j][sb3], acc[i][j], 0, 0, 0);
                    }
            };
            unsigned char* st0 = reinterpret_cast<unsigned char*>(smem);
            unsigned char* st1 = st0 + SPH_STAGE;
            load16(a0, b0);
            if (n16 > 1) load16(a1, b1);
            store16(st0, a0, b0);
            __syncthreads();
            for (int j16 = 0; j16 < n16; j16 += 2) {
                if (j16 + 2 < n16) load16(a0, b0);                 // slice j + 2 (its registers were stored one slice ago)
                mfma16(st0);
                if (j16 + 1 < n16) store16(st1, a1, b1);           // slice j + 1 -> the other stage
                __syncthreads();
                if (j16 + 1 >= n16) break;
                if (j16 + 3 < n16) load16(a1, b1);
                mfma16(st1);
                if (j16 + 2 < n16) store16(st0, a0, b0);
                __syncthreads();
            }
        } else if (SPLIT && SPLIT != 3 && MODE == 0 && p.b_planes != nullptr) {
            // B = weights pre-cut into their three bf16 planes (gamer_split3_planes, once per step): the planes are staged as
            // they are - 8-byte loads, 8-byte LDS stores, no cut.  (The SPLIT == 3 form reads PACKED pieces instead: below.)
            constexpr int NPL = 3;
            const uint16_t* pq = p.b_planes + (pb - p.B);
            const int64_t plane = p.b_plane_stride;
            uint2 rq[NPL][4];
            auto load_a = [&]() {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4*>(pa + j * ja);
                pa += sa;
            };
            auto load_bq = [&]() {
#pragma unroll
                for (int s = 0; s < NPL; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rq[s][j] = *reinterpret_cast<const uint2*>(pq + s * plane + j * jb);
                pq += sb;
            };
            auto store_bq = [&](unsigned char* d8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = tid + GEMM_THREADS * j;
                    unsigned char* dst = B_KC ? d8 + (f >> 3) * SP_KC_ROW + ((f & 7) << 3) : d8 + sp_rc_off(f >> 5, (f & 31) << 2);
#pragma unroll
                    for (int s = 0; s < NPL; ++s) *reinterpret_cast<uint2*>(dst + s * (B_KC ? SP_KC_IMG : SP_RC_IMG)) = rq[s][j];
                }
            };
            load_a();
            load_bq();
            for (int kt = 0; kt < nkt; ++kt) {
                unsigned char* d8 = reinterpret_cast<unsigned char*>(smem);
                if (A_KC) store_kc_split(d8, tid, ra); else store_rc_split(d8, tid, ra);
                store_bq(d8 + SP_OPERAND);
                __syncthreads();
                if (kt + 1 < nkt) { load_a(); load_bq(); }
                mfma_tile(smem);
                __syncthreads();
            }
        } else if (SPLIT == 3 && SP_WIDE3 && A_KC && B_KC) {
            // TWO 32-deep slices per barrier pair, each in its own stage: 48 MFMAs between barriers as in the six-product form (with
            // 24 the fixed cost of a K-step - two barriers, the wait for the loads - is half of the step)
            float4 ra2[4], rb2[4];
            float* const stage1 = smem + H2_STAGE / 4;
            auto load_into = [&](float4 (&qa)[4], float4 (&qb)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) qa[j] = *reinterpret_cast<const float4*>(pa + j * ja);
#pragma unroll
                for (int j = 0; j < 4; ++j) qb[j] = *reinterpret_cast<const float4*>(pb + j * jb);
                pa += sa;
                pb += sb;
            };
            load_into(ra, rb);
            if (nkt > 1) load_into(ra2, rb2);
            for (int kt = 0; kt < nkt; kt += 2) {
                const bool two = kt + 1 < nkt;
                store_tile(smem);
                if (two) store_tile_from(stage1, ra2, rb2);
                __syncthreads();
                if (kt + 2 < nkt) load_into(ra, rb);
                if (kt + 3 < nkt) load_into(ra2, rb2);
                mfma_tile(smem);
                if (two) mfma_tile(stage1);
                __syncthreads();
            }
        } else if (SPLIT && SP_PREFETCH2) {
            // one stage of three-image tiles (60 KB), global loads TWO K-steps ahead in two register sets: a split K-step
            // is ~0.4x as long as the fp32 kernel's, and one step of distance no longer covers the load latency under load
            float4 ra2[4], rb2[4];
            auto load_into = [&](float4 (&qa)[4], float4 (&qb)[4]) {
                if (SP_ABLATE & 8) return;
#pragma unroll
                for (int j = 0; j < 4; ++j) qa[j] = *reinterpret_cast<const float4*>(pa + j * ja);
#pragma unroll
                for (int j = 0; j < 4; ++j) qb[j] = *reinterpret_cast<const float4*>(pb + j * jb);
                pa += sa;
                pb += sb;
            };
            auto store_from = [&](const float4 (&qa)[4], const float4 (&qb)[4]) {
                unsigned char* d8 = reinterpret_cast<unsigned char*>(smem);
                if (SPLIT == 3) {
                    if (A_KC) store_kc_h2(d8, tid, qa, scale_a); else store_rc_h2(d8, tid, qa, scale_a);
                    if (B_KC) store_kc_h2(d8 + H2_B_OFF, tid, qb, scale_b); else store_rc_h2(d8 + H2_B_OFF, tid, qb, scale_b);
                    return;
                }
                if (A_KC) store_kc_split(d8, tid, qa); else store_rc_split(d8, tid, qa);
                if (B_KC) store_kc_split(d8 + SP_OPERAND, tid, qb); else store_rc_split(d8 + SP_OPERAND, tid, qb);
            };
            load_into(ra, rb);
            if (nkt > 1) load_into(ra2, rb2);
            if (PP && grp == 1) __syncthreads();                   // half a K-step behind group 0
            for (int kt = 0; kt < nkt; kt += 2) {
                store_from(ra, rb);
                __syncthreads();
                if (kt + 2 < nkt) load_into(ra, rb);
                mfma_tile(smem);
                __syncthreads();
                if (kt + 1 >= nkt) break;
                store_from(ra2, rb2);
                __syncthreads();
                if (kt + 3 < nkt) load_into(ra2, rb2);
                mfma_tile(smem);
                __syncthreads();
            }
        } else if (SPLIT) {
            // one stage of three-image tiles (60 KB): split + store, barrier, request the next step, multiply, barrier
            fast_load();
            if (PP && grp == 1) __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                store_tile(smem);
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xc07f); t1 = stamp_now(); t_ph[0] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                __syncthreads();
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[1] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                if (kt + 1 < nkt) fast_load();
                mfma_tile(smem);
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[2] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                __syncthreads();
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[3] += t1 - t0; __builtin_amdgcn_sched_barrier(0); }
            }
        } else if (NBUF == 1) {
            // one LDS image (36 KB): three workgroups per CU; two barriers per K-step, covered by the other
            // two waves of the SIMD
            fast_load();
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt > 0) __syncthreads();
                store_tile(smem);
                __syncthreads();
                if (kt + 1 < nkt) fast_load();
                mfma_tile(smem);
            }
        } else {
            fast_load();
            store_tile(smem);
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const int cur = kt & 1;
                const bool more = kt + 1 < nkt;
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                if (more) fast_load();
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[0] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                mfma_tile(smem + 2 * cur * TILE_FLOATS);
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[1] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                if (more) store_tile(smem + 2 * (cur ^ 1) * TILE_FLOATS);
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[2] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); }
                __syncthreads();
                if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); t_ph[3] += t1 - t0; __builtin_amdgcn_sched_barrier(0); }
            }
        }
    } else {
        if (SPLIT) {
            if (nkt > 0) load_tile(kbeg);
            if (PP && grp == 1) __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                store_tile(smem);
                __syncthreads();
                if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BK);
                mfma_tile(smem);
                __syncthreads();
            }
        } else if (NBUF == 1) {
            if (nkt > 0) load_tile(kbeg);
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt > 0) __syncthreads();
                store_tile(smem);
                __syncthreads();
                if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BK);
                mfma_tile(smem);
            }
        } else {
            if (nkt > 0) {
                load_tile(kbeg);
                store_tile(smem);
            }
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const int cur = kt & 1;
                const bool more = kt + 1 < nkt;
                if (more) load_tile(kbeg + (kt + 1) * BK);
                mfma_tile(smem + 2 * cur * TILE_FLOATS);
                if (more) store_tile(smem + 2 * (cur ^ 1) * TILE_FLOATS);
                __syncthreads();
            }
        }
    }
    if (PP && grp == 0) __syncthreads();                           // pairs with group 1's last barrier
    const unsigned long long t_loop_end = STAMP ? stamp_now() : 0;

    // ---- SPLIT == 3 row-range guard (see SPLIT3_GUARD_RATIO): a tile with a row too far below the tensor's maximum is redone in fp32
    bool redo = false;
    if (GUARD && p.guard) {
        const float thr = __uint_as_float(amax_a_bits) * SPLIT3_GUARD_RATIO;      // (Inf / NaN maximum: every tile takes the fp32 path)
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m = rmax[j];                                                    // the row's 8 threads are 8 consecutive lanes
            m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
            bad |= (m > 0.f) & !(m >= thr);
        }
        // workgroup OR through the tail of the DYNAMIC allocation (behind the epilogue's four 64 x 68 patches; the K loop's last
        // barrier has retired every read of the images): a static __shared__ word - what __syncthreads_or takes - would push the
        // forward layout's 2 x 40,960-byte stages past half of the CU's 160 KB and leave ONE workgroup per CU (+47 % measured)
        int* flags = reinterpret_cast<int*>(smem + 4 * 64 * 68);
        const bool wave_bad = __any(bad ? 1 : 0) != 0;
        if (lane == 0) flags[wid] = wave_bad ? 1 : 0;
        __syncthreads();
        redo = (flags[0] | flags[1] | flags[2] | flags[3]) != 0;
        if (redo) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            if (wave_live) {
                // lane (r32, h) supplies A(row, k + h) and B(col, k + h) of v_mfma_f32_32x32x2_f32, straight from global memory
                const int ar0 = row0 + wm * 64 + r32, bc0 = col0 + wn * 64 + r32;
#pragma unroll 1
                for (int k = kbeg; k < kend; k += 2) {
                    const int kk = k + h;
                    const bool kin = kk < kend;
                    float av[2], bv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = ar0 + i * 32;
                        av[i] = (kin && row < row_end) ? p.A[(int64_t)row * p.a_rs + kk] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = bc0 + j * 32;
                        bv[j] = (kin && col < col_end) ? (B_KC ? Bsrc[(int64_t)col * p.b_rs + kk] : Bsrc[(int64_t)kk * p.b_ks + col]) : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: acc[i][j][r] is C[row = (r&3)+8*(r>>2)+4*h][col = lane&31] of its 32x32 tile --
    const bool interior = a_in && b_in;
    const float alpha_eff = (SPLIT == 3 && !redo) ? p.alpha * unscale : p.alpha;      // (alpha and 1 / (s_a s_b) are powers of two or 1 in the step)
    if (MODE == 0 && (NBUF == 2 || SPLIT) && interior && (p.ldc & 3) == 0) {
        // Row-major rewrite through LDS (the K-loop images are dead after its last barrier): each wave
        // parks its 64x64 patch as [64][68] floats and reads it back one 16-byte row chunk per lane, so the
        // patch leaves in 16 x 1-KiB store instructions (4 rows x 256 B each) instead of 64 dword stores.
        float* patch = smem + wid * (64 * 68);
        // row-dot epilogue: the 16 float4 of the other operand are requested now, ahead of the LDS rewrite of the
        // accumulators, so that they arrive while the patch is written and read back
        float4 oth[EPI == 2 ? 16 : 1];
        if (EPI == 2) {
            const float* ob = p.rowdot_other + (int64_t)(row0 + wm * 64 + (lane >> 4)) * p.ldc + col0 + wn * 64 + ((lane & 15) << 2);
#pragma unroll
            for (int it = 0; it < 16; ++it) oth[it] = *reinterpret_cast<const float4*>(ob + (int64_t)it * 4 * p.ldc);
        }
        // (alpha == 1 for every GEMM of the train step: no multiply; every VALU instruction here is matrix time)
        if (SPLIT != 3 && p.alpha == 1.f) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + j * 32 + r32] = acc[i][j][r];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + j * 32 + r32] = alpha_eff * acc[i][j][r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): own LDS writes landed (wave-private patch)
        const int c4 = (lane & 15) << 2;
        const DropoutRng rng((EPI == 1 || EPI == 4) ? p.p_drop : 0.f, p.seed);
        // one 64-bit address per wave patch, then a constant stride per store (a per-store row * ldc product costs
        // two quarter-rate integer multiplies each)
        const int row_first = row0 + wm * 64 + (lane >> 4);
        const int col = col0 + wn * 64 + c4;
        float* dst0 = Cp + (int64_t)row_first * p.ldc + col;
        const int64_t step = 4 * p.ldc;
        const int qk_pos0 = EPI == 3 ? row_first % max(p.qk_S, 1) : 0;
        // (a wave's 64-column patch lies on one side of amax_c_col0 when that is a multiple of 64 - the q|k / v boundary is)
        const bool amax_on = GAMER_GEMM_CAMAX_BUILD && p.amax_c != nullptr && !ACCUM && col0 + wn * 64 >= p.amax_c_col0;
        float cmax = 0.f;
        // The epilogues that read global memory request ALL sixteen rows' values here, before the loop (the accumulators are in the LDS
        // patch by now: the registers are free).  Requested inside the loop, every iteration waited for its own load - sixteen memory
        // round trips in a row per tile, thirty-two with the row map (index, then residual) - with nothing but the other resident
        // workgroup to hide them.
        int pre_rc[EPI == 1 ? 16 : 1];
        float4 pre_x[EPI == 1 ? 16 : (EPI == 4 ? 8 : 1)], pre_b[EPI == 4 ? 8 : 1];      // (EPI 4: two rounds of eight rows - 128 registers spill)
        if (EPI == 1) {
            if (p.row_map) {
#pragma unroll
                for (int it = 0; it < 16; ++it) pre_rc[it] = p.row_map[row_first + 4 * it];
            } else {
#pragma unroll
                for (int it = 0; it < 16; ++it) pre_rc[it] = row_first + 4 * it;
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) pre_x[it] = *reinterpret_cast<const float4*>(p.resid + (int64_t)pre_rc[it] * p.ldc + col);
        }
        // (EPI 4 with sw_tbl: every row of the tile is in ONE group and the lane's four columns are fixed - two quads per tile)
        float4 tb_g = make_float4(0.f, 0.f, 0.f, 0.f), tb_u = tb_g;
        if (EPI == 4 && p.sw_tbl) {
            const float* tb = p.sw_tbl + (int64_t)tile_group * 2 * p.N + col;
            tb_g = *reinterpret_cast<const float4*>(tb);
            tb_u = *reinterpret_cast<const float4*>(tb + p.N);
        }
        auto pre_sw = [&](int it0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const float* ga = p.sw_gu + (int64_t)(row_first + 4 * (it0 + it)) * p.sw_ld + col;
                pre_x[EPI == 4 ? it : 0] = *reinterpret_cast<const float4*>(ga);
                pre_b[EPI == 4 ? it : 0] = *reinterpret_cast<const float4*>(ga + p.N);
            }
        };
        if (EPI == 4) pre_sw(0);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (EPI == 4 && it == 8) pre_sw(8);
            const int lr = (lane >> 4) + 4 * it;             // row inside the patch
            const float4 v = *reinterpret_cast<const float4*>(patch + lr * 68 + c4);
            if (EPI == 3) {
                const int row = row_first + 4 * it;
                const int hd = (col0 + wn * 64) >> 6;                 // the wave's patch is one head
                const int g = lane & 15;
                float4 x = v;
                const bool cross = p.qk_bias_q != nullptr;
                const int a = cross ? p.qk_act_idx[row] : 0;
                if (hd < p.qk_nq + p.qk_nkv) {
                    const bool isq = hd < p.qk_nq;
                    if (cross) {
                        const float4 b4 = isq ? reinterpret_cast<const float4*>(p.qk_bias_q + (int64_t)a * p.qk_nq * 64 + hd * 64)[g]
                                              : reinterpret_cast<const float4*>(p.qk_bias_k + (int64_t)a * p.qk_nkv * 64 + (hd - p.qk_nq) * 64)[g];
                        x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
                    }
                    *reinterpret_cast<float4*>(dst0 + it * step) = x;         // raw (biased) q / k: what the backward reads
                    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                    ss = row16_sum_dpp(ss);
                    const float rstd = rsqrtf(ss * (1.f / 64.f) + p.qk_eps);
                    const float4 w4 = reinterpret_cast<const float4*>(isq ? p.qk_wq : p.qk_wk)[g];
                    float4 y;
                    y.x = w4.x * (x.x * rstd); y.y = w4.y * (x.y * rstd); y.z = w4.z * (x.z * rstd); y.w = w4.w * (x.w * rstd);
                    float4 pr;
                    pr.x = __shfl_xor(y.x, 8, 64); pr.y = __shfl_xor(y.y, 8, 64); pr.z = __shfl_xor(y.z, 8, 64); pr.w = __shfl_xor(y.w, 8, 64);
                    // (the rows of a tile are consecutive: one division per wave, then an add and a conditional subtract)
                    int pos = qk_pos0 + 4 * it;
                    if (p.qk_S >= 64) pos = pos >= p.qk_S ? pos - p.qk_S : pos;   // a 64-row patch crosses at most one sequence end
                    else pos %= p.qk_S;                                           // short sequences: several ends per patch
                    if (p.qk_pos_ids) pos = p.qk_pos_ids[row];
                    const float4 cc = reinterpret_cast<const float4*>(p.qk_cos + pos * 64)[g];
                    const float4 sn = reinterpret_cast<const float4*>(p.qk_sin + pos * 64)[g];
                    const float sgn = g < 8 ? -1.f : 1.f;
                    float4 o;
                    o.x = y.x * cc.x + sgn * pr.x * sn.x; o.y = y.y * cc.y + sgn * pr.y * sn.y;
                    o.z = y.z * cc.z + sgn * pr.z * sn.z; o.w = y.w * cc.w + sgn * pr.w * sn.w;
                    float* rot = isq ? p.qk_q_rot + (int64_t)row * p.qk_nq * 64 + hd * 64
                                     : p.qk_k_rot + (int64_t)row * p.qk_nkv * 64 + (hd - p.qk_nq) * 64;
                    *reinterpret_cast<float4*>(rot + 4 * g) = o;
                } else {
                    if (cross) {
                        const float4 b4 = reinterpret_cast<const float4*>(p.qk_bias_v + (int64_t)a * p.qk_nkv * 64 +
                                                                          (hd - p.qk_nq - p.qk_nkv) * 64)[g];
                        x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
                    }
                    *reinterpret_cast<float4*>(dst0 + it * step) = x;
                }
            } else if (EPI == 4) {
                // v = d(hm) of four consecutive columns of one row: the SwiGLU backward right here (gamer_swiglu_bwd_ld's arithmetic)
                const int row = row_first + 4 * it;
                float* ga = p.sw_gu + (int64_t)row * p.sw_ld + col;
                float* ua = ga + p.N;
                float4 a = pre_x[EPI == 4 ? (it & 7) : 0], b = pre_b[EPI == 4 ? (it & 7) : 0];
                a.x += tb_g.x; a.y += tb_g.y; a.z += tb_g.z; a.w += tb_g.w;
                b.x += tb_u.x; b.y += tb_u.y; b.z += tb_u.z; b.w += tb_u.w;
                float m[4];
                rng.mult4((uint32_t)(((int64_t)row * p.N + col) >> 2), m);
                const float d0 = m[0] * v.x, d1 = m[1] * v.y, d2 = m[2] * v.z, d3 = m[3] * v.w;
                float4 dg, du;
                dg.x = d0 * b.x * dsilu_f(a.x); du.x = d0 * silu_f(a.x);
                dg.y = d1 * b.y * dsilu_f(a.y); du.y = d1 * silu_f(a.y);
                dg.z = d2 * b.z * dsilu_f(a.z); du.z = d2 * silu_f(a.z);
                dg.w = d3 * b.w * dsilu_f(a.w); du.w = d3 * silu_f(a.w);
                *reinterpret_cast<float4*>(ga) = dg;
                *reinterpret_cast<float4*>(ua) = du;
                if (amax_on) {
                    asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(cmax) : "v"(dg.x), "v"(dg.y), "v"(dg.z), "v"(dg.w));
                    asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(cmax) : "v"(du.x), "v"(du.y), "v"(du.z), "v"(du.w));
                }
            } else if (EPI == 1) {
                const int64_t e = (int64_t)pre_rc[EPI == 1 ? it : 0] * p.ldc + col;
                float* dst = Cp + e;
                const float4 x = pre_x[EPI == 1 ? it : 0];
                float m[4];
                rng.mult4((uint32_t)(e >> 2), m);
                float4 o;
                o.x = x.x + m[0] * v.x; o.y = x.y + m[1] * v.y; o.z = x.z + m[2] * v.z; o.w = x.w + m[3] * v.w;
                *reinterpret_cast<float4*>(dst) = o;
            } else {
                float* dst = dst0 + it * step;
                if (EPI != 1 && EPI != 3 && amax_on)
                    asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(cmax) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                if (EPI == 2) {
                    // the wave's 64-column patch is exactly one head: 16 lanes hold one row of it
                    const float4 o4 = oth[EPI == 2 ? it : 0];
                    float d = v.x * o4.x + v.y * o4.y + v.z * o4.z + v.w * o4.w;
                    d = row16_sum_dpp(d);
                    if ((lane & 15) == 0) {
                        const int row = row_first + 4 * it;
                        const int heads = p.N >> 6, head = (col0 + wn * 64) >> 6;
                        p.rowdot_out[((int64_t)(row / p.rowdot_S) * heads + head) * p.rowdot_S + row % p.rowdot_S] = d;
                    }
                }
                if (ACCUM) {
                    float4 o = *reinterpret_cast<const float4*>(dst);
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                    *reinterpret_cast<float4*>(dst) = o;
                } else {
                    *reinterpret_cast<float4*>(dst) = v;
                }
            }
        }
        if (EPI != 1 && EPI != 3 && amax_on) {
            uint32_t mw = __float_as_uint(cmax);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
            if (lane == 0 && mw) amax_publish(mw, p.amax_c, (uint32_t)(vblock * 4 + wid));
        }
    } else if (MODE == 1 && p.wgrad_ws != nullptr) {
        // deterministic weight gradient: the chunk's 128 x 128 partial tile goes to the workspace with plain stores (out-of-range
        // rows / columns hold exact zeros: their operand rows were staged as zeros), wgrad_reduce_kernel adds the chunks of a
        // tile in chunk order - no fp32 atomics, the same bits on every run
        float* blk = p.wgrad_ws + ws_block * (int64_t)(BM * BN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lc = wn * 64 + j * 32 + r32;
                const int lr0 = wm * 64 + i * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) blk[(lr0 + (r & 3) + 8 * (r >> 2)) * BN + lc] = alpha_eff * acc[i][j][r];
            }
    } else {
        float emax = 0.f;
        if (EPI == 4 && p.sw_tbl != nullptr) {
            // a partial tile of the SwiGLU-backward epilogue with a row table (at most one per row group): the table row is added to
            // the tile's gate | up values IN PLACE first (a rolled loop: anything more inside `emit`, which is unrolled 64 times,
            // keeps the compiler from unrolling it and sends the accumulators to scratch), then the plain epilogue runs on the sums
            const float* tb0 = p.sw_tbl + (int64_t)tile_group * 2 * p.N;
            const int nr = min(BM, row_end - row0), nc = min(BN, p.N - col0);
            for (int i = threadIdx.x; i < nr * nc; i += GEMM_THREADS) {
                const int rr = row0 + i / nc, cc = col0 + i % nc;
                float* ga = p.sw_gu + (int64_t)rr * p.sw_ld + cc;
                ga[0] += tb0[cc];
                ga[p.N] += tb0[p.N + cc];
            }
            __syncthreads();
        }
        auto emit = [&](int row, int col, float v) {
            if (EPI == 4) {
                float* ga = p.sw_gu + (int64_t)row * p.sw_ld + col;
                float a = ga[0], b = ga[p.N];
                const int64_t e = (int64_t)row * p.N + col;
                const DropoutRng rng(p.p_drop, p.seed);
                float m[4];
                rng.mult4((uint32_t)(e >> 2), m);
                const float d = m[e & 3] * v;
                const float dg = d * b * dsilu_f(a), du = d * silu_f(a);
                ga[0] = dg; ga[p.N] = du;
                if (GAMER_GEMM_CAMAX_BUILD && p.amax_c != nullptr) emax = fmaxf(emax, fmaxf(fabsf(dg), fabsf(du)));
                return;
            }
            if (GAMER_GEMM_CAMAX_BUILD && EPI != 1 && MODE == 0 && !ACCUM && p.amax_c != nullptr && col >= p.amax_c_col0) emax = fmaxf(emax, fabsf(v));
            if (EPI == 1) {
                const int64_t rc = p.row_map ? p.row_map[row] : row;
                const int64_t e = rc * p.ldc + col;
                const DropoutRng rng(p.p_drop, p.seed);
                float m[4];
                rng.mult4((uint32_t)(e >> 2), m);
                Cp[e] = p.resid[e] + m[e & 3] * v;
            } else {
                float* dst = Cp + (int64_t)row * p.ldc + col;
                if (MODE == 1) atomicAdd(dst, v);
                else if (ACCUM) *dst += v;
                else *dst = v;
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = col0 + wn * 64 + j * 32 + r32;
                const int rbase = row0 + wm * 64 + i * 32 + 4 * h;
                if (interior) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) emit(rbase + (r & 3) + 8 * (r >> 2), col, alpha_eff * acc[i][j][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2);
                        if (rbase + rr < row_end && col < col_end) emit(rbase + rr, col, alpha_eff * acc[i][j][r]);
                    }
                }
            }
        }
        if (GAMER_GEMM_CAMAX_BUILD && MODE == 0 && !ACCUM && p.amax_c != nullptr) {
            uint32_t mw = __float_as_uint(emax);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
            if (lane == 0 && mw) amax_publish(mw, p.amax_c, (uint32_t)(vblock * 4 + wid));
        }
    }
    if (STAMP) {
        unsigned long long* st = g_gemm_stamp;
        if (st && threadIdx.x == 0) {
            const unsigned long long t_end = stamp_now();
            unsigned long long* rec = st + 8ull * vblock;
            rec[0] = t_ph[0]; rec[1] = t_ph[1]; rec[2] = t_ph[2]; rec[3] = t_ph[3];
            rec[4] = t_loop_end - t_begin;        // whole K loop
            rec[5] = t_end - t_loop_end;          // epilogue
            rec[6] = t_begin;                     // absolute start (for gaps between workgroups)
            rec[7] = t_end;
        }
    }
}

// grid = `vgrid` workgroups: one tile each; a smaller grid (GAMER_GEMM_PERSIST): every workgroup walks tiles blockIdx.x,
// blockIdx.x + gridDim.x, ... - the stores of one tile's epilogue drain while the next tile's first loads are on their way,
// and no workgroup start-up sits between them (gridDim.x a multiple of 8: a workgroup's tiles stay on its XCD's share)
template <bool A_KC, bool B_KC, int MODE, bool ACCUM, bool STAMP, int NBUF, int EPI, int SPLIT = 0, bool PP = false>
__global__ void __launch_bounds__(PP ? 2 * GEMM_THREADS : GEMM_THREADS, (NBUF == 1 && SPLIT == 0) ? 3 : 2)
gemm_f32_kernel(const GemmParams p, const int vgrid) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    for (int v = blockIdx.x; v < vgrid; v += gridDim.x) {
        gemm_f32_tile<A_KC, B_KC, MODE, ACCUM, STAMP, NBUF, EPI, SPLIT, PP>(p, smem_all, v, vgrid);
        if (v + (int)gridDim.x < vgrid) __syncthreads();       // the next tile's staging overwrites this tile's epilogue patches
    }
}

template <bool A_KC, bool B_KC, int MODE, bool ACCUM, bool STAMP, int NBUF, int EPI = 0, int SPLIT = 0, bool PP = false>
static int launch_gemm_t(const GemmParams& p, int blocks, hipStream_t st, int lds) {
    static int attr_lds_dev[MAX_DEVICES] = {};
    int& attr_lds = attr_lds_dev[current_device()];
    auto kfn = gemm_f32_kernel<A_KC, B_KC, MODE, ACCUM, STAMP, NBUF, EPI, SPLIT, PP>;
    if (attr_lds != lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            set_error("gamer_gemm_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_lds = lds;
    }
    static int persist = -1;
    if (persist < 0) { const char* e = getenv("GAMER_GEMM_PERSIST"); persist = e ? atoi(e) : 0; }
    // GAMER_GEMM_PERSIST=n (n a multiple of 8): at most n workgroups, each walking several tiles (plain split forms only)
    const int grid = (persist > 0 && SPLIT != 0 && !PP && !STAMP && blocks > persist) ? persist : blocks;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(PP ? 2 * GEMM_THREADS : GEMM_THREADS), lds, st, p, blocks);
    GAMER_CHECK_LAUNCH("gamer_gemm_f32");
    return 0;
}

// the bf16-split forms: one LDS stage; the epilogue's row-major rewrite needs the full 72 KB
template <bool A_KC, bool B_KC, int MODE, int SPLIT>
static int launch_gemm_split(const GemmParams& p, int blocks, hipStream_t st) {
    // (SPLIT == 3 with two stages: 81,920 / 73,728 / 65,536 bytes for the forward / input-gradient / weight-gradient layouts; the
    // epilogue's row-major rewrite needs 69,632)
    constexpr int h2_stage = 2 * ((A_KC ? SP_KC_IMG : SP_RC_IMG) + (B_KC ? SP_KC_IMG : SP_RC_IMG));
    const int lds = (SPLIT == 3 && SP_WIDE3 && A_KC && B_KC && 2 * h2_stage > GEMM_LDS_BYTES) ? 2 * h2_stage : GEMM_LDS_BYTES;
    static_assert(SP_LDS_BYTES <= GEMM_LDS_BYTES, "split images must fit the fp32 kernel's LDS allocation");
    const bool acc = MODE == 0 && p.accumulate;
    static int stamp = -1, pp = 0;
    if (stamp < 0) {
        const char* e = getenv("GAMER_GEMM_STAMP");
        stamp = (e && e[0] == '1') ? 1 : 0;
        const char* e2 = getenv("GAMER_GEMM_PP");
        pp = (SP_PINGPONG && e2 && e2[0] == '1') ? 1 : 0;
    }
    (void)pp;
#if SP_PINGPONG
    if ((SPLIT == 6 || SPLIT == 3) && pp && !(MODE == 0 && p.b_planes)) {
        // eight-wave ping-pong form: one workgroup per two tiles, both groups' LDS regions
        int pblocks;
        if (MODE == 0) pblocks = (blocks + 1) / 2;
        else {
            const int tiles_mn = p.m_tiles * p.n_tiles;
            pblocks = blocks / tiles_mn * ((tiles_mn + 1) / 2);
        }
        const int plds = 2 * GEMM_LDS_BYTES;
        if (stamp) return launch_gemm_t<A_KC, B_KC, MODE, false, true, 2, 0, SPLIT, true>(p, pblocks, st, plds);
        if (MODE == 0 && p.resid) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 1, SPLIT, true>(p, pblocks, st, plds);
        if (MODE == 0 && p.rowdot_out) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 2, SPLIT, true>(p, pblocks, st, plds);
        if (MODE == 0 && A_KC && B_KC && p.qk_q_rot) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 3, SPLIT, true>(p, pblocks, st, plds);
        if (acc) return launch_gemm_t<A_KC, B_KC, MODE, true, false, 2, 0, SPLIT, true>(p, pblocks, st, plds);
        return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 0, SPLIT, true>(p, pblocks, st, plds);
    }
#endif
    if (stamp && SPLIT == 6) return launch_gemm_t<A_KC, B_KC, MODE, false, true, 2, 0, SPLIT>(p, blocks, st, lds);
    if constexpr (SPLIT == 3 && MODE == 0 && A_KC) {
        if (p.sw_gu) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 4, SPLIT>(p, blocks, st, lds);
    }
    if (MODE == 0 && p.resid) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 1, SPLIT>(p, blocks, st, lds);
    if (MODE == 0 && p.rowdot_out) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 2, SPLIT>(p, blocks, st, lds);
    if (MODE == 0 && A_KC && B_KC && p.qk_q_rot) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 3, SPLIT>(p, blocks, st, lds);
    if (acc) return launch_gemm_t<A_KC, B_KC, MODE, true, false, 2, 0, SPLIT>(p, blocks, st, lds);
    return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 0, SPLIT>(p, blocks, st, lds);
}

template <bool A_KC, bool B_KC, int MODE>
static int launch_gemm(const GemmParams& p, int blocks, hipStream_t st, int split = 0) {
    if (split == 6) return launch_gemm_split<A_KC, B_KC, MODE, 6>(p, blocks, st);
    if (split == 9) return launch_gemm_split<A_KC, B_KC, MODE, 9>(p, blocks, st);
    if (split == 3) return launch_gemm_split<A_KC, B_KC, MODE, 3>(p, blocks, st);
    static int stamp = -1, solo = 0, nbuf = 2;
    if (stamp < 0) {
        const char* e = getenv("GAMER_GEMM_STAMP");
        stamp = (e && e[0] == '1') ? 1 : 0;
        const char* s2 = getenv("GAMER_GEMM_SOLO");
        solo = (s2 && s2[0] == '1') ? 1 : 0;
        const char* s3 = getenv("GAMER_GEMM_NBUF");
        nbuf = (s3 && s3[0] == '1') ? 1 : 2;
    }
    const int lds = solo ? 150 * 1024 : GEMM_LDS_BYTES;          // solo: one workgroup per CU (diagnostics)
    const bool acc = MODE == 0 && p.accumulate;
    if (stamp) return launch_gemm_t<A_KC, B_KC, MODE, false, true, 2>(p, blocks, st, lds);
    if (MODE == 0 && p.resid) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 1>(p, blocks, st, lds);
    if (MODE == 0 && p.rowdot_out) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 2>(p, blocks, st, lds);
    if (MODE == 0 && A_KC && B_KC && p.qk_q_rot) return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2, 3>(p, blocks, st, lds);
    if (nbuf == 1) {
        const int lds1 = 2 * TILE_FLOATS * (int)sizeof(float);
        if (acc) return launch_gemm_t<A_KC, B_KC, MODE, true, false, 1>(p, blocks, st, lds1);
        return launch_gemm_t<A_KC, B_KC, MODE, false, false, 1>(p, blocks, st, lds1);
    }
    if (acc) return launch_gemm_t<A_KC, B_KC, MODE, true, false, 2>(p, blocks, st, lds);
    return launch_gemm_t<A_KC, B_KC, MODE, false, false, 2>(p, blocks, st, lds);
}

// Second pass of the deterministic weight gradient: C tile (+)= sum over the chunks of its group, in chunk order.  One workgroup
// per (group, C tile, strip of 8 rows): thread -> (row tid >> 5, float4 column tid & 31); four chunk loads in flight.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc, int M, int N, int m_tiles, int n_tiles,
                    int groups, const int32_t* __restrict__ group_offsets, int K, int kchunk, int64_t strideC) {
    const int tiles_mn = m_tiles * n_tiles;
    int id = blockIdx.x;
    const int strip = id & 15; id >>= 4;
    const int tile = id % tiles_mn, g = id / tiles_mn;
    int c_beg = 0, c_n;
    if (group_offsets) {
        int prev = group_offsets[0];
        c_n = 0;
        for (int gi = 0; gi <= g; ++gi) {
            const int nxt = group_offsets[gi + 1];
            const int chunks = (nxt - prev + kchunk - 1) / kchunk;
            if (gi < g) c_beg += chunks; else c_n = chunks;
            prev = nxt;
        }
    } else {
        c_n = (K + kchunk - 1) / kchunk;
    }
    const int lr = strip * 8 + (threadIdx.x >> 5), lc = (threadIdx.x & 31) << 2;
    const float4* src = reinterpret_cast<const float4*>(ws + ((int64_t)c_beg * tiles_mn + tile) * (int64_t)(BM * BN) + lr * BN + lc);
    const int64_t step4 = (int64_t)tiles_mn * (BM * BN) / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0;
    for (; c + 4 <= c_n; c += 4) {
        const float4 a0 = src[(int64_t)c * step4], a1 = src[(int64_t)(c + 1) * step4], a2 = src[(int64_t)(c + 2) * step4],
                     a3 = src[(int64_t)(c + 3) * step4];
        s.x = (((s.x + a0.x) + a1.x) + a2.x) + a3.x; s.y = (((s.y + a0.y) + a1.y) + a2.y) + a3.y;
        s.z = (((s.z + a0.z) + a1.z) + a2.z) + a3.z; s.w = (((s.w + a0.w) + a1.w) + a2.w) + a3.w;
    }
    for (; c < c_n; ++c) {
        const float4 a0 = src[(int64_t)c * step4];
        s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
    }
    const int row = (tile / n_tiles) * BM + lr, col = (tile % n_tiles) * BN + lc;
    if (row < M && c_n > 0) {
        float* dst = C + (int64_t)g * strideC + (int64_t)row * ldc + col;
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < N) dst[e] += v[e];
    }
}

// host side of the second pass, shared with the bf16 weight-gradient GEMM (gemm_bf16.hip: same tiles, same chunk order)
int launch_wgrad_reduce(const float* ws, float* C, int64_t ldc, int M, int N, int groups, const int32_t* group_offsets, int K,
                        int kchunk, int64_t strideC, hipStream_t st) {
    const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
    const int64_t rblocks = (int64_t)groups * m_tiles * n_tiles * 16;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)rblocks), dim3(256), 0, st, ws, C, ldc, M, N, m_tiles, n_tiles, groups,
                       group_offsets, K, kchunk, strideC);
    return (int)hipGetLastError();
}

// x[i] -> its three bf16 pieces, planes[s * stride + i]  (the cut of split3, four values per thread)
__global__ void __launch_bounds__(256)
split3_planes_kernel(const float4* __restrict__ x, uint16_t* __restrict__ planes, int64_t n4, int64_t stride) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        uint2 img[3];
        split3_quad(x[i], img);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(planes + s * stride + 4 * i) = img[s];
    }
}

}  // namespace gamer

using namespace gamer;

extern "C" int gamer_split3_planes(const float* x, gamer_bf16* planes, int64_t n, int64_t plane_stride, void* stream) {
    GAMER_CHECK_ARG(x && planes && n > 0 && n % 4 == 0 && plane_stride >= n && plane_stride % 4 == 0 && aligned16(x) &&
                    (reinterpret_cast<uintptr_t>(planes) & 7u) == 0,
                    "gamer_split3_planes: n=%lld and plane_stride=%lld must be multiples of 4 (stride >= n), x 16-byte and planes 8-byte aligned",
                    (long long)n, (long long)plane_stride);
    const int64_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split3_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<uint16_t*>(planes), n4, plane_stride);
    GAMER_CHECK_LAUNCH("gamer_split3_planes");
    return 0;
}

extern "C" int gamer_debug_gemm_stamp(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), &v, sizeof(v));
}

// the diagnostic builds' switches (stamps, one LDS stage, ping-pong) rule out the epilogue forms
static inline bool diag_build_switches() {
    static EnvSwitch stamp("GAMER_GEMM_STAMP"), nbuf("GAMER_GEMM_NBUF");
    return stamp.is_set() || nbuf.is_set();
}
static inline bool diag_pp_switch() { static EnvSwitch pp("GAMER_GEMM_PP"); return pp.is_set(); }
static int gemm_entry(const gamer_gemm_desc* d, void* stream, int split) {
    GAMER_CHECK_ARG(d, "gamer_gemm_f32: null descriptor");
    GAMER_CHECK_ARG(d->A && d->B && d->C, "gamer_gemm_f32: null matrix pointer");
    GAMER_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gamer_gemm_f32: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    GAMER_CHECK_ARG((d->a_rs == 1) != (d->a_ks == 1) || (d->a_rs == 1 && d->a_ks == 1 && (d->M == 1 || d->K == 1)),
                    "gamer_gemm_f32: exactly one of a_rs/a_ks must be 1 (rs=%lld ks=%lld)", (long long)d->a_rs, (long long)d->a_ks);
    GAMER_CHECK_ARG((d->b_rs == 1) != (d->b_ks == 1) || (d->b_rs == 1 && d->b_ks == 1 && (d->N == 1 || d->K == 1)),
                    "gamer_gemm_f32: exactly one of b_rs/b_ks must be 1 (rs=%lld ks=%lld)", (long long)d->b_rs, (long long)d->b_ks);
    const bool a_kc = d->a_ks == 1, b_kc = d->b_ks == 1;
    const int64_t a_ld = a_kc ? d->a_rs : d->a_ks, b_ld = b_kc ? d->b_rs : d->b_ks;
    GAMER_CHECK_ARG(a_ld % 4 == 0 && b_ld % 4 == 0 && d->ldc >= d->N,
                    "gamer_gemm_f32: leading dims must be multiples of 4 (lda=%lld ldb=%lld ldc=%lld)",
                    (long long)a_ld, (long long)b_ld, (long long)d->ldc);
    GAMER_CHECK_ARG(aligned16(d->A) && aligned16(d->B), "gamer_gemm_f32: A and B must be 16-byte aligned");
    GAMER_CHECK_ARG(d->groups >= 1 && (d->groups == 1 || d->group_offsets), "gamer_gemm_f32: groups=%d needs group_offsets", d->groups);
    GAMER_CHECK_ARG(d->strideB % 4 == 0, "gamer_gemm_f32: strideB must be a multiple of 4");
    GAMER_CHECK_ARG(d->group_mode == 0 || d->group_mode == 1, "gamer_gemm_f32: group_mode=%d", d->group_mode);

    GemmParams p;
    p.A = d->A; p.a_rs = d->a_rs; p.a_ks = d->a_ks;
    p.B = d->B; p.b_rs = d->b_rs; p.b_ks = d->b_ks;
    p.C = d->C; p.ldc = d->ldc;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.alpha = d->alpha; p.accumulate = d->accumulate;
    p.groups = d->groups; p.group_offsets = d->group_offsets;
    p.strideB = d->strideB; p.strideC = d->strideC;
    p.kchunk = d->kchunk;
    p.resid = d->resid; p.row_map = d->row_map; p.p_drop = d->p_drop; p.seed = d->seed;
    p.rowdot_other = d->rowdot_other; p.rowdot_out = d->rowdot_out; p.rowdot_S = d->rowdot_S;
    p.qk_wq = d->qk_wq; p.qk_wk = d->qk_wk; p.qk_eps = d->qk_eps; p.qk_cos = d->qk_cos; p.qk_sin = d->qk_sin;
    p.qk_bias_q = d->qk_bias_q; p.qk_bias_k = d->qk_bias_k; p.qk_bias_v = d->qk_bias_v;
    p.qk_act_idx = d->qk_act_idx; p.qk_pos_ids = d->qk_pos_ids; p.qk_q_rot = d->qk_q_rot; p.qk_k_rot = d->qk_k_rot;
    p.qk_S = d->qk_S; p.qk_nq = d->qk_nq; p.qk_nkv = d->qk_nkv;
    p.amax_a = d->amax_a; p.amax_b = d->amax_b;
    p.guard = g_split3_guard;
    p.amax_c = d->amax_c; p.amax_c_col0 = d->amax_c_col0;
    p.wgrad_ws = nullptr;
    p.sw_gu = d->sw_gu; p.sw_ld = d->sw_ld;
    p.group_div = d->group_div > 1 ? d->group_div : 1; p.sw_tbl = d->sw_tbl;
    GAMER_CHECK_ARG(d->group_div <= 1 || (d->group_mode == 0 && d->group_offsets && d->groups % d->group_div == 0),
                    "gamer_gemm_f32: group_div = %d needs group_mode 0, group_offsets and groups (%d) a multiple of it", d->group_div, d->groups);
    GAMER_CHECK_ARG(!d->sw_tbl || ((d->sw_gu || d->sw_hm) && aligned16(d->sw_tbl)),
                    "gamer_gemm_f32: sw_tbl is an option of the SwiGLU epilogues (sw_gu, sw_hm)");
    GAMER_CHECK_ARG(!d->sw_gu || (split == 3 && d->group_mode == 0 && !d->accumulate && !d->resid && !d->rowdot_out && !d->qk_q_rot &&
                                  d->alpha == 1.f && d->N % 4 == 0 && d->sw_ld >= 2 * (int64_t)d->N && d->sw_ld % 4 == 0 &&
                                  aligned16(d->sw_gu) && d->p_drop >= 0.f && d->p_drop < 1.f),
                    "gamer_gemm_f32: the SwiGLU-backward epilogue (sw_gu) is built for gamer_gemm_f32_split(terms = 3), a plain input-"
                    "gradient GEMM with alpha = 1, N %% 4 == 0, sw_ld >= 2 N (N=%d sw_ld=%lld)", d->N, (long long)d->sw_ld);
    GAMER_CHECK_ARG(!d->amax_c || (d->group_mode == 0 && !d->accumulate && !d->resid && !d->qk_q_rot && d->amax_c_col0 >= 0 &&
                                   d->amax_c_col0 % 64 == 0),
                    "gamer_gemm_f32: amax_c needs a plain or row-dot Linear-forward / input-gradient GEMM (no accumulate, residual or "
                    "q|k|v epilogue) and amax_c_col0 a multiple of 64 (amax_c_col0=%d)", d->amax_c_col0);
    GAMER_CHECK_ARG(split != 3 || (d->amax_a && d->amax_b), "gamer_gemm_f32_split: terms = 3 needs amax_a and amax_b (gamer_absmax_f32)");
    p.b_planes = (split && d->group_mode == 0) ? (const uint16_t*)d->b_planes : nullptr;
    p.b_plane_stride = d->b_plane_stride;
    GAMER_CHECK_ARG(!p.b_planes || split == 3 || ((reinterpret_cast<uintptr_t>(p.b_planes) & 7u) == 0 && d->b_plane_stride % 4 == 0 && d->b_plane_stride > 0),
                    "gamer_gemm_f32_split: b_planes must be 8-byte aligned with b_plane_stride %% 4 == 0");
    GAMER_CHECK_ARG(!p.b_planes || split != 3 || (reinterpret_cast<uintptr_t>(p.b_planes) & 15u) == 0,
                    "gamer_gemm_f32_split: terms = 3 takes b_planes as packed pieces at B's offsets (16-byte aligned)");
    // packed pieces hold FOUR elements per 16 bytes ({h0 x 4 | h1 x 4}): the ragged-quad masking of the edge tiles works on
    // plain fp32 lanes, so a B whose contiguous extent is not a multiple of 4 is cut in the kernel instead (same bits)
    if (split == 3 && p.b_planes && ((b_kc ? d->K : d->N) % 4) != 0) p.b_planes = nullptr;
    GAMER_CHECK_ARG(!d->qk_q_rot || (d->qk_k_rot && d->qk_wq && d->qk_wk && d->qk_cos && d->qk_sin && d->group_mode == 0 &&
                                     d->groups == 1 && a_kc && b_kc && !d->resid && !d->rowdot_out && !d->accumulate &&
                                     d->alpha == 1.f && d->M % BM == 0 && d->qk_nq > 0 && d->qk_nkv > 0 && d->qk_S > 0 &&
                                     d->N == (d->qk_nq + 2 * d->qk_nkv) * 64 && d->N % BN == 0 && d->ldc % 4 == 0 &&
                                     aligned16(d->C) && aligned16(d->qk_q_rot) && aligned16(d->qk_k_rot) &&
                                     (!d->qk_bias_q || (d->qk_bias_k && d->qk_bias_v && d->qk_act_idx)) &&
                                     !diag_build_switches()),
                    "gamer_gemm_f32: the q|k|v epilogue needs a Linear-forward layout, one group, M %% 128 == 0 (M=%d), "
                    "N = (nq + 2 nkv) * 64 a multiple of 128 (N=%d), alpha = 1 and no other epilogue", d->M, d->N);
    GAMER_CHECK_ARG(!d->rowdot_out || (d->rowdot_other && d->rowdot_S > 0 && d->group_mode == 0 && d->groups == 1 &&
                                       !d->resid && !d->accumulate && d->M % BM == 0 && d->N % BN == 0 &&
                                       d->M % d->rowdot_S == 0 && d->ldc % 4 == 0 && aligned16(d->rowdot_other) &&
                                       aligned16(d->C) && !diag_build_switches()),
                    "gamer_gemm_f32: the row-dot epilogue needs full 128 x 128 tiles (M=%d N=%d), one group, ldc %% 4 == 0",
                    d->M, d->N);
    GAMER_CHECK_ARG(!d->resid || (d->group_mode == 0 && !d->accumulate && a_kc && b_kc && d->ldc % 4 == 0 &&
                                  aligned16(d->resid) && aligned16(d->C) && d->p_drop >= 0.f && d->p_drop < 1.f),
                    "gamer_gemm_f32: the fused residual epilogue needs a Linear-forward layout, ldc %% 4 == 0, no accumulate");
    GAMER_CHECK_ARG(!d->sw_hm || (d->group_mode == 0 && a_kc && b_kc && !d->accumulate && !d->resid && !d->rowdot_out && !d->qk_q_rot &&
                                  !d->sw_gu && d->alpha == 1.f && d->N % 8 == 0 && d->ldc >= d->N && d->ldc % 4 == 0 && aligned16(d->sw_hm) &&
                                  aligned16(d->C) && (!d->sw_tbl || d->sw_row_group) && d->p_drop >= 0.f && d->p_drop < 1.f),
                    "gamer_gemm_f32: the SwiGLU-forward epilogue (sw_hm) needs a Linear-forward layout with N = 2 I (N=%d), alpha = 1, no other "
                    "epilogue, and sw_row_group with sw_tbl", d->N);
    // short-contraction Linear forward from packed weight pieces: the activation-stationary kernel (csrc/gemm_as.hip)
    if (split == 3 && gemm_as_eligible(d, a_kc, b_kc, p.b_planes)) return launch_gemm_as(d, p.b_planes, b_kc, (hipStream_t)stream);
    if (d->sw_hm) {
        // sw_hm not taken by that kernel (few rows, no packed pieces yet, another product form): the projection as it would run without
        // it, then gamer_swiglu_fwd_ld(_tbl) on its output - the same results by definition
        gamer_gemm_desc d2 = *d;
        d2.sw_hm = nullptr; d2.sw_row_group = nullptr; d2.sw_tbl = nullptr; d2.amax_c = nullptr;
        const int rc = gemm_entry(&d2, stream, split);
        if (rc) return rc;
        if (d->amax_c) gamer_amax_sink(d->amax_c, nullptr);
        return d->sw_tbl ? gamer_swiglu_fwd_ld_tbl(d->C, d->ldc, d->M, d->N / 2, d->p_drop, d->seed, d->sw_hm, d->sw_tbl, d->sw_row_group, stream)
                         : gamer_swiglu_fwd_ld(d->C, d->ldc, d->M, d->N / 2, d->p_drop, d->seed, d->sw_hm, stream);
    }
    // plain input gradient with 256 input features: the output-stationary kernel (csrc/gemm_os.hip)
    if (split == 3 && gemm_os_eligible(d, a_kc, b_kc, p.b_planes)) return launch_gemm_os(d, p.b_planes, g_split3_guard, (hipStream_t)stream);
    // Linear forward with 256 output features and K > 256 (o_proj, the experts' down projection): the same kernel on W's transposed pieces
    if (split == 3 && gemm_os_fwd_eligible(d, a_kc, b_kc))
        return launch_gemm_os(d, reinterpret_cast<const uint16_t*>(d->b_planes_t), g_split3_guard, (hipStream_t)stream, true);
    p.n_tiles = (d->N + BN - 1) / BN;
    hipStream_t st = (hipStream_t)stream;

    if (d->group_mode == 0) {
        p.m_tiles = (d->M + BM - 1) / BM + (d->group_offsets ? d->groups : 0);
        const int64_t blocks = (int64_t)p.m_tiles * p.n_tiles;
        GAMER_CHECK_ARG(blocks < (1LL << 31), "gamer_gemm_f32: grid too large");
        if (a_kc && b_kc) return launch_gemm<true, true, 0>(p, (int)blocks, st, split);
        if (a_kc && !b_kc) return launch_gemm<true, false, 0>(p, (int)blocks, st, split);
        if (!a_kc && !b_kc) return launch_gemm<false, false, 0>(p, (int)blocks, st, split);
        GAMER_CHECK_ARG(false, "gamer_gemm_f32: layout (A row-contiguous, B k-contiguous) is not built");
    } else {
        GAMER_CHECK_ARG(!a_kc && !b_kc, "gamer_gemm_f32: group_mode 1 (wgrad) needs row-contiguous A and B");
        GAMER_CHECK_ARG(d->kchunk >= BK && d->kchunk % BK == 0, "gamer_gemm_f32: kchunk=%d must be a positive multiple of 32", d->kchunk);
        p.m_tiles = (d->M + BM - 1) / BM;
        const int64_t chunks = (d->K + d->kchunk - 1) / d->kchunk + (d->group_offsets ? d->groups : 0);
        const int64_t blocks = chunks * p.m_tiles * p.n_tiles;
        GAMER_CHECK_ARG(blocks < (1LL << 31), "gamer_gemm_f32: grid too large");
        p.wgrad_ws = d->wgrad_ws;
        if (p.wgrad_ws) {
            const int64_t need = chunks * p.m_tiles * p.n_tiles * (int64_t)(BM * BN);
            GAMER_CHECK_ARG(d->wgrad_ws_floats >= need && aligned16(d->wgrad_ws),
                            "gamer_gemm_f32: wgrad_ws holds %lld floats, this weight gradient needs %lld (chunks x tiles x 16384)",
                            (long long)d->wgrad_ws_floats, (long long)need);
            GAMER_CHECK_ARG(!diag_pp_switch(), "gamer_gemm_f32: the deterministic weight gradient is not built for the ping-pong form");
        }
        // three-product form with the chunk workspace: the 256 x 256-tile kernel (csrc/gemm_wg.hip; same partial tiles, same bits)
        const int rc = (split == 3 && p.wgrad_ws && gemm_wg_eligible(d, a_kc, b_kc)) ? launch_gemm_wg(d, st)
                                                                                     : launch_gemm<false, false, 1>(p, (int)blocks, st, split);
        if (rc || !p.wgrad_ws) return rc;
        const int64_t rblocks = (int64_t)d->groups * p.m_tiles * p.n_tiles * 16;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)rblocks), dim3(256), 0, st, p.wgrad_ws, d->C, d->ldc, d->M, d->N, p.m_tiles,
                           p.n_tiles, d->groups, d->group_offsets, d->K, d->kchunk, d->strideC);
        GAMER_CHECK_LAUNCH("gamer_gemm_f32/wgrad_reduce");
        return 0;
    }
    return 0;
}

// bits of max |x| over `batch` matrices [rows, cols] (leading dimension ld, `stride` elements apart) -> atomicMax(out):
// non-negative floats order like their bit patterns, a NaN's pattern is above every number's (it wins, on purpose)
__device__ __forceinline__ uint32_t amax4(uint32_t m, const uint4 v) {
    return max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
}
// one atomic per workgroup
__device__ __forceinline__ void amax_commit(uint32_t m, uint32_t* __restrict__ out) {
    __shared__ uint32_t part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m) amax_publish(m, out, blockIdx.x);
    }
}
// dense storage (ld == cols, matrices back to back): one 16-byte load per thread and iteration, two in flight
__global__ void __launch_bounds__(256)
absmax_flat_kernel(const uint4* __restrict__ x, int64_t n4, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step < n4; i += 2 * step) {
        const uint4 a = x[i], b = x[i + step];
        m = amax4(amax4(m, a), b);
    }
    if (i < n4) m = amax4(m, x[i]);
    amax_commit(m, out);
}
// padded rows: one wave per row at a time, lanes across the row's 16-byte chunks
__global__ void __launch_bounds__(256)
absmax_rows_kernel(const float* __restrict__ x, int batch, int64_t stride, int rows, int cols, int64_t ld, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * 4, nrows = (int64_t)batch * rows;
    uint32_t m = 0;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nrows; w += waves) {
        const int64_t b = w / rows, r = w - b * rows;
        const float* src = x + b * stride + r * ld;
        for (int c = lane << 2; c < cols; c += 256) {
            if (c + 3 < cols) m = amax4(m, *reinterpret_cast<const uint4*>(src + c));
            else for (int q = 0; c + q < cols; ++q) m = max(m, __float_as_uint(src[c + q]) & 0x7fffffffu);
        }
    }
    amax_commit(m, out);
}
extern "C" int gamer_absmax_f32(const float* x, int batch, int64_t stride, int rows, int cols, int64_t ld, uint32_t* out, void* stream) {
    GAMER_CHECK_ARG(x && out && batch >= 1 && rows >= 1 && cols >= 1 && ld >= cols && ld % 4 == 0 && stride % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(x) & 15u) == 0, "gamer_absmax_f32: bad arguments (rows=%d cols=%d ld=%lld)",
                    rows, cols, (long long)ld);
    if (ld == cols && (batch == 1 || stride == (int64_t)rows * ld)) {
        const int64_t n4 = (int64_t)batch * rows * cols / 4;
        int64_t blocks = (n4 + 511) / 512;
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(absmax_flat_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const uint4*>(x), n4, out);
    } else {
        int64_t blocks = ((int64_t)batch * rows + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(absmax_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, batch, stride, rows, cols, ld, out);
    }
    GAMER_CHECK_LAUNCH("gamer_absmax_f32");
    return 0;
}

// several dense tensors inside one buffer in ONE launch (the weights of a pass): table[e] = {offset, numel} in elements (both
// multiples of 4), `bpe` workgroups per entry, out + e * GAMER_AMAX_WORDS = the slot of entry e
__global__ void __launch_bounds__(256)
absmax_multi_kernel(const float* __restrict__ base, const int64_t* __restrict__ table, int bpe, uint32_t* __restrict__ out) {
    const int e = blockIdx.x / bpe, part = blockIdx.x % bpe;
    const uint4* x = reinterpret_cast<const uint4*>(base + table[2 * e]);
    const int64_t n4 = table[2 * e + 1] >> 2;
    uint32_t m = 0;
    for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n4; i += (int64_t)bpe * 256) m = amax4(m, x[i]);
    __shared__ uint32_t part4[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) part4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part4[0], part4[1]), max(part4[2], part4[3]));
        if (m) amax_publish(m, out + (int64_t)e * 256, (uint32_t)part);
    }
}
extern "C" int gamer_absmax_multi_f32(const float* base, const int64_t* table, int n, uint32_t* out, void* stream) {
    GAMER_CHECK_ARG(base && table && out && n >= 1 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0, "gamer_absmax_multi_f32: bad arguments");
    const int bpe = 16;
    hipLaunchKernelGGL(absmax_multi_kernel, dim3(n * bpe), dim3(256), 0, (hipStream_t)stream, base, table, bpe, out);
    GAMER_CHECK_LAUNCH("gamer_absmax_multi_f32");
    return 0;
}

// the fp16 pieces of several dense tensors of one buffer (the parameters), each scaled by ITS power of two, PACKED at the
// offsets of the values they replace: the 16 bytes of elements 4 i .. 4 i + 3 hold their four h0 pieces, then their four h1
// pieces (table as gamer_absmax_multi_f32, slots = its output)
__global__ void __launch_bounds__(256)
split2h_planes_multi_kernel(const float* __restrict__ base, const int64_t* __restrict__ table, int bpe,
                            const uint32_t* __restrict__ slots, uint16_t* __restrict__ planes) {
    const int e = blockIdx.x / bpe, part = blockIdx.x % bpe;
    const int64_t off = table[2 * e], n4 = table[2 * e + 1] >> 2;
    float s, inv;
    scale_from_amax(amax_read(slots + (int64_t)e * 256), s, inv);
    const float4* x = reinterpret_cast<const float4*>(base + off);
    for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n4; i += (int64_t)bpe * 256) {
        uint2 img[2];
        split2h_quad(x[i], s, img);
        reinterpret_cast<uint4*>(planes)[(off >> 2) + i] = make_uint4(img[0].x, img[0].y, img[1].x, img[1].y);
    }
}
extern "C" int gamer_split2h_planes_multi(const float* base, const int64_t* table, int n, const uint32_t* slots, gamer_bf16* planes,
                                          void* stream) {
    GAMER_CHECK_ARG(base && table && slots && planes && n >= 1 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(planes) & 15u) == 0, "gamer_split2h_planes_multi: bad arguments");
    const int bpe = 16;
    hipLaunchKernelGGL(split2h_planes_multi_kernel, dim3(n * bpe), dim3(256), 0, (hipStream_t)stream, base, table, bpe, slots,
                       reinterpret_cast<uint16_t*>(planes));
    GAMER_CHECK_LAUNCH("gamer_split2h_planes_multi");
    return 0;
}

// The packed pieces of W^T from the packed pieces of W: table = n entries (element offset of a row-major [rows][cols] matrix, rows,
// cols), rows and cols multiples of 4; planes_t receives, at the SAME offset, the pieces of the [cols][rows] transpose - unit (c, r4) =
// {h0 of elements (4 r4 .. 4 r4 + 3, c) | their h1}.  Same bits as the source (nothing is cut again).
__global__ void __launch_bounds__(256)
split2h_transpose_multi_kernel(const uint4* __restrict__ planes, const int64_t* __restrict__ table, int bpe, uint4* __restrict__ planes_t) {
    const int e = blockIdx.x / bpe, part = blockIdx.x % bpe;
    const int64_t off4 = table[3 * e] >> 2;
    const int rows = (int)table[3 * e + 1], cols = (int)table[3 * e + 2];
    const int r4n = rows >> 2, c4n = cols >> 2;
    const int64_t units = (int64_t)r4n * cols;
    for (int64_t u = (int64_t)part * 256 + threadIdx.x; u < units; u += (int64_t)bpe * 256) {
        const int r4 = (int)(u / cols), c = (int)(u % cols);      // (consecutive threads: consecutive columns of four source rows)
        uint32_t h0[4], h1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 s = planes[off4 + (int64_t)(4 * r4 + i) * c4n + (c >> 2)];
            const uint32_t w0 = (c & 2) ? s.y : s.x, w1 = (c & 2) ? s.w : s.z;
            h0[i] = (c & 1) ? (w0 >> 16) : (w0 & 0xffffu);
            h1[i] = (c & 1) ? (w1 >> 16) : (w1 & 0xffffu);
        }
        planes_t[off4 + (int64_t)c * r4n + r4] = make_uint4(h0[0] | (h0[1] << 16), h0[2] | (h0[3] << 16), h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16));
    }
}
extern "C" int gamer_split2h_transpose_multi(const gamer_bf16* planes, const int64_t* table, int n, gamer_bf16* planes_t, void* stream) {
    GAMER_CHECK_ARG(planes && table && planes_t && n >= 1 && aligned16(planes) && aligned16(planes_t), "gamer_split2h_transpose_multi: bad arguments");
    const int bpe = 16;
    hipLaunchKernelGGL(split2h_transpose_multi_kernel, dim3(n * bpe), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(planes),
                       table, bpe, reinterpret_cast<uint4*>(planes_t));
    GAMER_CHECK_LAUNCH("gamer_split2h_transpose_multi");
    return 0;
}

extern "C" int gamer_gemm_f32(const gamer_gemm_desc* d, void* stream) { return gemm_entry(d, stream, 0); }

extern "C" int gamer_split3_guard(int on) {
    const int prev = g_split3_guard;
    if (on >= 0) g_split3_guard = on ? 1 : 0;
    return prev;
}

extern "C" int gamer_gemm_f32_split(const gamer_gemm_desc* d, int terms, void* stream) {
    GAMER_CHECK_ARG(terms == 3 || terms == 6 || terms == 9, "gamer_gemm_f32_split: terms=%d (3, 6 or 9)", terms);
    return gemm_entry(d, stream, terms);
}
