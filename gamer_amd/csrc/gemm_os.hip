// Output-stationary input gradient of an nn.Linear with 256 input features, three-product fp16 form (gamer_gemm_f32_split, terms = 3):
// dX[m][n] = alpha * sum_k dY[m][k] W[k][n], n < 256, any k - the input gradients of the q|k|v projections and of the tied head
// (autograd of ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 1001).  Same results contract as csrc/gemm.hip.
//
// The mirror image of gemm_as.hip (where the activation row is short and stays in registers): here the OUTPUT row is short.  A wave owns
// 32 rows of dX for the whole launch - all 256 columns, 128 accumulator registers, C^T = W^T dY^T puts the row on the lane - and
// walks the contraction: the lane streams its own row of dY from global memory straight into registers (64 contiguous bytes per
// 32-k block and half wave: k = block + 16 h + 0 .. 15; the contraction order is free as long as both operands use it), cuts it there
// - dY never passes through LDS and is cut ONCE (the 128 x 128 kernel cuts it for each of its two column tiles) - and multiplies it with
// W^T fragments read with transposing LDS reads from the block's [32 k][256 n] piece images, which the workgroup's eight waves (256
// rows) stage from the parameters' packed pieces (gamer_split2h_planes_multi), two buffers, one barrier per block of 48 MFMAs per wave.
// Scale of dY: the tensor's (gamer_gemm_desc.amax_a), as in gemm.hip; the row-range guard of that kernel becomes: every lane tracks
// its row's largest magnitude as it streams, and a workgroup holding a non-zero row below 2^-16 of the tensor's maximum runs its
// contraction a second time with a power-of-two scale PER ROW (exact, since the row index is not contracted) - decided on the device.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace gamer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef OS_ABLATE
#define OS_ABLATE 0       // timing-only builds: 1 no C stores, 2 no MFMAs, 4 no W staging after the first block, 8 no dY loads after the first
#endif
#ifndef OS_WAVES
#define OS_WAVES 8        // waves per workgroup (32 rows each); 4: two workgroups per CU, out of step with each other
#endif
constexpr int OS_THREADS = 64 * OS_WAVES;
constexpr int OS_N = 256;                       // output columns (all of them: a wave holds 8 tiles of 32)
constexpr int OS_BK = 32;                       // contraction block
constexpr int OS_WLD = OS_BK * 64 / OS_THREADS; // 16-byte groups of W per thread and block
constexpr int OS_ROWB = 2 * OS_N;               // bytes of one k row of a piece image
constexpr int OS_IMG = OS_BK * OS_ROWB;         // 16 KB
constexpr int OS_STAGE = 2 * OS_IMG;            // h0 | h1
constexpr int OS_LDS = 2 * OS_STAGE;            // 64 KB

struct OsParams {
    const float* A; int64_t lda;                // dY [M][K]
    const uint16_t* Wp; int64_t ldw;            // packed pieces at W's offsets (W [K][N] row-major): 16 bytes = {h0 x 4 | h1 x 4} of four consecutive n
    float* C; int64_t ldc;
    int M, K;
    float alpha;
    const uint32_t* amax_a; const uint32_t* amax_w;
    uint32_t* amax_c; int amax_c_col0;
    int guard;
    // grouped form (the experts' gate|up at d_in = 256): rows sorted by expert, group g = rows group_offsets[g] .. group_offsets[g + 1] - 1
    // with its own W strideW elements further; a workgroup never crosses a segment boundary
    int groups; const int32_t* group_offsets; int64_t strideW;
    // Linear FORWARD of a layer with 256 output features and a long contraction (o_proj K = 384, the experts' down projection K = 512):
    // the same kernel on the TRANSPOSED packed pieces of W (gamer_split2h_transpose_multi: [K][256] from W [256][K]), with the residual
    // epilogue of csrc/gemm.hip: C[map(m)] = resid[map(m)] + dropout(alpha * acc)  (row_map: the expert rows back to token order)
    const float* resid; const int32_t* row_map; float p_drop; uint64_t seed;
};

// byte offset of (k, n) in a piece image [32 k][256 n] of 16-bit values: the 16-byte chunk (n >> 3) of the k row XORed with (k & 3) << 2
// (gemm_wg.hip: wg_off - stores of 8-byte quads and transposing reads both conflict-free)
__device__ __forceinline__ int os_off(int k, int n) { return k * OS_ROWB + ((((n >> 3) ^ ((k & 3) << 2))) << 4) + ((n & 7) << 1); }

__global__ void __launch_bounds__(OS_THREADS, 8 / OS_WAVES)
gemm_os_kernel(const OsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char os_smem[];
    __shared__ uint32_t os_word[2];             // [0] amax of C, [1] guard vote
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int row0 = blockIdx.x * (32 * OS_WAVES), row_end = p.M, grp = 0;
    if (p.group_offsets) {
        int prev = p.group_offsets[0], tiles_before = 0;
        bool found = false;
        for (int gi = 0; gi < p.groups; ++gi) {
            const int nxt = p.group_offsets[gi + 1];
            const int tiles = (nxt - prev + 32 * OS_WAVES - 1) / (32 * OS_WAVES);
            if (!found && (int)blockIdx.x < tiles_before + tiles) {
                grp = gi; row0 = prev + ((int)blockIdx.x - tiles_before) * (32 * OS_WAVES); row_end = nxt; found = true;
            }
            if (!found) tiles_before += tiles;
            prev = nxt;
        }
        if (!found) return;
    }
    const int m = row0 + w * 32 + r;
    const bool valid_m = m < row_end;
    if (tid < 2) os_word[tid] = 0;

    const uint32_t amax_a_bits = amax_read(p.amax_a);
    float s_a, i_a, s_w, i_w;
    scale_from_amax(amax_a_bits, s_a, i_a);
    scale_from_amax(amax_read(p.amax_w), s_w, i_w);
    float sc = s_a, inv = i_a;                  // this lane's scale of dY (pass 0: the tensor's)

    // ---- dY: this lane's 16 floats of a block, clamped so that every load is unconditional and in range
    const float* arow = p.A + (int64_t)(valid_m ? m : row_end - 1) * p.lda;
    const int k_last4 = (p.K - 1) & ~3;
    auto load_row = [&](int kb, float4 (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const float4*>(arow + min(kb + 16 * h + 4 * j, k_last4));
    };
    // ---- W: thread f = tid + OS_THREADS i -> k = f >> 6, n = 4 (f & 63): 16 bytes of packed pieces -> two 8-byte stores
    const int wk = tid >> 6, wn4 = (tid & 63) << 2;
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.Wp) + (((int64_t)grp * p.strideW + (int64_t)wk * p.ldw + wn4) >> 2);
    const int w_lds = os_off(wk, wn4);                                // (+ OS_WAVES i rows: k & 3 unchanged)
    auto load_w = [&](int kb, uint4 (&dst)[OS_WLD]) {
#pragma unroll
        for (int i = 0; i < OS_WLD; ++i) {
            const int k = min(kb + wk + OS_WAVES * i, p.K - 1);              // (rows past K: any valid row; stored as zeros)
            dst[i] = wsrc[(((int64_t)(k - wk) * p.ldw) >> 2)];
        }
    };
    auto store_w = [&](int kb, unsigned char* st, const uint4 (&src)[OS_WLD]) {
#pragma unroll
        for (int i = 0; i < OS_WLD; ++i) {
            const bool ok = kb + wk + OS_WAVES * i < p.K;
            unsigned char* d = st + w_lds + i * OS_WAVES * OS_ROWB;
            *reinterpret_cast<uint2*>(d) = ok ? make_uint2(src[i].x, src[i].y) : make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(d + OS_IMG) = ok ? make_uint2(src[i].z, src[i].w) : make_uint2(0u, 0u);
        }
    };
    // ---- W^T fragments: transposing reads, lane 16 g + 4 q + pp -> (k 4-block row q, n 4 pp .. 4 pp + 3) of its 4 x 16 block;
    // k = 16 h + 8 s + 4 c + q (k & 3 = q), n = 32 nt + 16 g + 4 pp: chunk = (4 nt + 2 g + (pp >> 1)) ^ (q << 2) = 4 (nt ^ q) + ...
    const int q = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
    const int fr_lane = (16 * h + q) * OS_ROWB + ((2 * gsel + (pp >> 1)) << 4) + ((pp & 1) << 3);
    int fw[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) fw[nt] = fr_lane + (((nt & 4) | ((nt & 3) ^ q)) << 6);
    auto read_frag = [&](const unsigned char* st, int off) -> f16x8 {
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(st + off + c * 4 * OS_ROWB));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return __builtin_bit_cast(f16x8, out);
    };

    f32x16 acc[8];
    float rmax = 0.f;                            // largest |dY| of this lane's half of the row
    const int n_blk = (p.K + OS_BK - 1) / OS_BK;
    float out_scale = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
        // two register sets of dY, blocks alternate between them: a set is re-requested (for block b + 2) as soon as its last values
        // are cut, half a block before the block ends - every load is unconditional (clamped: past the end the last block is read once
        // more and never used) and pinned in place: a load under `if (b + 2 < n_blk)` becomes a loop-carried merge the compiler copies
        // and waits for right behind the load, a free one is sunk to its use (gemm_as.hip)
        float4 R[2][4];
        uint4 rw[OS_WLD];
        load_row(0, R[0]);
        load_w(0, rw);
        store_w(0, os_smem, rw);
        load_row(OS_BK, R[1]);
        load_w(OS_BK, rw);
        __syncthreads();
        auto block = [&](auto par_c, const int b) {
            constexpr int P = decltype(par_c)::value;
            const unsigned char* cur = os_smem + P * OS_STAGE;
            unsigned char* nxt = os_smem + (P ^ 1) * OS_STAGE;
            const int kb = b * OS_BK;
            if (kb + OS_BK > p.K) {              // the last, partial block: zeros past K (wave-uniform branch, no memory access inside)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k0 = kb + 16 * h + 4 * j;
                    R[P][j].x = k0 < p.K ? R[P][j].x : 0.f; R[P][j].y = k0 + 1 < p.K ? R[P][j].y : 0.f;
                    R[P][j].z = k0 + 2 < p.K ? R[P][j].z : 0.f; R[P][j].w = k0 + 3 < p.K ? R[P][j].w : 0.f;
                }
            }
            if (pass == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|"
                        : "+v"(rmax) : "v"(R[P][j].x), "v"(R[P][j].y), "v"(R[P][j].z), "v"(R[P][j].w));
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
                cut2h_quad(R[P][2 * s].x, R[P][2 * s].y, R[P][2 * s].z, R[P][2 * s].w, sc, a0, a1, b0, b1);
                cut2h_quad(R[P][2 * s + 1].x, R[P][2 * s + 1].y, R[P][2 * s + 1].z, R[P][2 * s + 1].w, sc, c0, c1, d0, d1);
                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                const u32x4v u0 = {a0, b0, c0, d0}, u1 = {a1, b1, c1, d1};
                const f16x8 y0 = __builtin_bit_cast(f16x8, u0), y1 = __builtin_bit_cast(f16x8, u1);
                if (s == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(OS_ABLATE & 8)) load_row(kb + 2 * OS_BK, R[P]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    const f16x8 w0 = read_frag(cur, fw[nt] + s * 8 * OS_ROWB);
                    const f16x8 w1 = read_frag(cur, fw[nt] + s * 8 * OS_ROWB + OS_IMG);
                    if (OS_ABLATE & 2) { asm volatile("" :: "v"(w0), "v"(w1), "v"(y0), "v"(y1)); continue; }
                    // smallest piece products first
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y0, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, y1, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, y0, acc[nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(OS_ABLATE & 4)) {
                store_w(kb + OS_BK, nxt, rw);    // (past the last block: zeros into the buffer nobody reads any more)
                load_w(kb + 2 * OS_BK, rw);
            }
            __syncthreads();
        };
        {
            int b = 0;
#pragma unroll 1
            for (; b + 1 < n_blk; b += 2) {
                block(std::integral_constant<int, 0>{}, b);
                block(std::integral_constant<int, 1>{}, b + 1);
            }
            if (b < n_blk) block(std::integral_constant<int, 0>{}, b);
        }
        out_scale = inv * i_w * p.alpha;
        if (pass == 1 || !p.guard) break;
        // guard: a non-zero row more than 2^16 below the tensor's largest magnitude keeps too few bits under the tensor's scale
        const float row_max = xor32_max(rmax);
        const float t_max = __uint_as_float(amax_a_bits);
        const bool small = valid_m && row_max > 0.f && row_max < t_max * (1.f / 65536.f) && ((amax_a_bits >> 23) & 0xffu) != 255u;
        if (small) os_word[1] = 1u;
        __syncthreads();
        if (os_word[1] == 0u) break;             // (workgroup-uniform)
        if (row_max > 0.f) scale_from_amax(__float_as_uint(row_max), sc, inv);
    }

    // ---- epilogue: lane = output row m, acc[nt][reg] = column 32 nt + (reg & 3) + 8 (reg >> 2) + 4 h
    float cmax = 0.f;
    if (p.resid) {
        // residual add + dropout (+ scatter through row_map): the residual quads of TWO column tiles are requested together, one pair
        // ahead of the pair being stored (loads issued inside the store loop would each wait for their own round trip: the stores
        // may alias them as far as the compiler knows)
        const int64_t orow = valid_m ? (p.row_map ? (int64_t)p.row_map[m] : (int64_t)m) : 0;
        const float* rrow = p.resid + orow * p.ldc;
        float* crow = p.C + orow * p.ldc;
        const DropoutRng rng(p.p_drop, p.seed);
        float4 rq[2][8];
        auto request = [&](int pr, float4 (&dst)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const float4*>(rrow + (2 * pr + (i >> 2)) * 32 + 8 * (i & 3) + 4 * h);
        };
        request(0, rq[0]);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (pr + 1 < 4) request(pr + 1, rq[(pr + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nt = 2 * pr + (i >> 2), g4 = i & 3;
                const int col = nt * 32 + 8 * g4 + 4 * h;
                float mu[4];
                rng.mult4((uint32_t)((orow * p.ldc + col) >> 2), mu);
                const float4 x = rq[pr & 1][i];
                const float4 t4 = make_float4(x.x + mu[0] * (acc[nt][4 * g4] * out_scale), x.y + mu[1] * (acc[nt][4 * g4 + 1] * out_scale),
                                              x.z + mu[2] * (acc[nt][4 * g4 + 2] * out_scale), x.w + mu[3] * (acc[nt][4 * g4 + 3] * out_scale));
                if (valid_m) *reinterpret_cast<float4*>(crow + col) = t4;
            }
        }
    } else if (valid_m && !(OS_ABLATE & 1)) {
        float* crow = p.C + (int64_t)m * p.ldc;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = nt * 32 + 8 * g4 + 4 * h;
                const float4 t4 = make_float4(acc[nt][4 * g4] * out_scale, acc[nt][4 * g4 + 1] * out_scale,
                                              acc[nt][4 * g4 + 2] * out_scale, acc[nt][4 * g4 + 3] * out_scale);
                *reinterpret_cast<float4*>(crow + col) = t4;
                if (p.amax_c != nullptr && col >= p.amax_c_col0)
                    cmax = fmaxf(fmaxf(fmaxf(cmax, fabsf(t4.x)), fabsf(t4.y)), fmaxf(fabsf(t4.z), fabsf(t4.w)));
            }
    } else if (OS_ABLATE & 1) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) asm volatile("" :: "v"(acc[nt][0]), "v"(acc[nt][15]));
    }
    if (p.amax_c) {
        uint32_t mw = __float_as_uint(cmax);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mw = max(mw, (uint32_t)__shfl_xor((int)mw, o2, 64));
        if (lane == 0 && mw) atomicMax(&os_word[0], mw);
        __syncthreads();
        if (tid == 0 && os_word[0]) amax_publish(os_word[0], p.amax_c, blockIdx.x);
    }
}

static std::atomic<long long> g_os_launches{0};

static inline bool gemm_os_enabled() {
    static EnvSwitch sw("GAMER_GEMM_OS");                 // (cached: gamer_reload_env() after a change inside the process)
    return sw.get(1) != 0;
}

// Does this descriptor take the output-stationary kernel?  (plain input gradient: A k-contiguous, B = W [K][256] row-contiguous with
// packed pieces, one group, no epilogue, enough rows to fill the chip)
// Does this Linear-FORWARD descriptor (both operands k-contiguous) take the kernel on W's transposed pieces?  256 output features, the
// plain store or the residual epilogue (+ row map, dropout), one group or the experts' row segments.
bool gemm_os_fwd_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc) {
    static EnvSwitch sw("GAMER_GEMM_OSF");                // (0: those launches stay on the 128 x 128 kernel - A/B runs)
    if (!gemm_os_enabled() || sw.get(1) == 0 || !a_kc || !b_kc || !d->b_planes_t || !d->amax_a || !d->amax_b) return false;
    if (d->group_mode != 0 || d->accumulate || d->rowdot_out || d->qk_q_rot || d->sw_gu || d->group_div > 1 || d->amax_c) return false;
    if (d->groups != 1 && (!d->group_offsets || d->strideC != 0)) return false;
    if (d->groups == 1 && d->group_offsets) return false;
    static EnvSwitch min_m("GAMER_GEMM_OS_MIN_M");
    if (d->N != OS_N || d->K < 4 || d->K % 4 != 0 || d->b_rs != d->K || d->M < min_m.get(16384)) return false;
    if (d->a_rs % 4 != 0 || d->ldc % 4 != 0 || !aligned16(d->C) || !aligned16(d->b_planes_t) || d->alpha != 1.f) return false;
    if (d->a_rs < d->K) return false;
    if (d->resid && (!aligned16(d->resid) || d->p_drop < 0.f || d->p_drop >= 1.f)) return false;
    if (!d->resid && d->row_map) return false;
    return true;
}

bool gemm_os_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc, const uint16_t* b_planes) {
    if (!gemm_os_enabled() || !a_kc || b_kc || !b_planes || !d->amax_a || !d->amax_b) return false;
    if (d->group_mode != 0 || d->accumulate || d->resid || d->rowdot_out || d->qk_q_rot || d->sw_gu || d->group_div > 1) return false;
    if (d->groups != 1 && (!d->group_offsets || d->strideC != 0 || d->amax_c)) return false;      // (grouped: the experts' gate|up at d_in = 256)
    if (d->groups == 1 && d->group_offsets) return false;      // (a one-group row window: stays on the tile kernel, which honours it)
    static EnvSwitch min_m("GAMER_GEMM_OS_MIN_M");
    if (d->N != OS_N || d->K < 4 || d->M < min_m.get(16384)) return false;
    if (d->a_rs % 4 != 0 || d->b_ks % 4 != 0 || d->ldc % 4 != 0 || !aligned16(d->C) || !aligned16(b_planes)) return false;
    if (d->a_rs < ((d->K + 3) & ~3)) return false;                   // (the clamped 16-byte loads stay inside a row of dY)
    if (d->amax_c && d->amax_c_col0 % 4 != 0) return false;
    return true;
}

int launch_gemm_os(const gamer_gemm_desc* d, const uint16_t* b_planes, int guard, hipStream_t st, bool fwd_t) {
    OsParams p;
    p.A = d->A; p.lda = d->a_rs;
    // (fwd_t: W's TRANSPOSED pieces [K][256], dense - the Linear forward; else W [K][256] row-contiguous itself - the input gradient)
    p.Wp = b_planes; p.ldw = fwd_t ? OS_N : d->b_ks;
    p.resid = fwd_t ? d->resid : nullptr; p.row_map = fwd_t ? d->row_map : nullptr; p.p_drop = d->p_drop; p.seed = d->seed;
    p.C = d->C; p.ldc = d->ldc;
    p.M = d->M; p.K = d->K;
    p.alpha = d->alpha;
    p.amax_a = d->amax_a; p.amax_w = d->amax_b;
    p.amax_c = d->amax_c; p.amax_c_col0 = d->amax_c_col0;
    p.guard = guard;
    p.groups = d->groups; p.group_offsets = d->groups > 1 ? d->group_offsets : nullptr; p.strideW = d->strideB;
    static bool attr_dev[MAX_DEVICES] = {};
    if (!attr_dev[current_device()]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_os_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS);
        if (e != hipSuccess) { set_error("gamer_gemm_f32_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_dev[current_device()] = true;
    }
    const dim3 grid((d->M + 32 * OS_WAVES - 1) / (32 * OS_WAVES) + (p.group_offsets ? d->groups : 0));
    hipLaunchKernelGGL(gemm_os_kernel, grid, dim3(OS_THREADS), OS_LDS, st, p);
    GAMER_CHECK_LAUNCH("gamer_gemm_f32_split/output-stationary input gradient");
    g_os_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

}  // namespace gamer

// Diagnostic (not in include/gamer_hip.h): launches of the output-stationary kernel by this process so far.
extern "C" long long gamer_debug_gemm_os_launches(void) { return gamer::g_os_launches.load(std::memory_order_relaxed); }
