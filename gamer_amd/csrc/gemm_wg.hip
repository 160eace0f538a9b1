// Weight gradient of an nn.Linear in the three-product fp16 form (gamer_gemm_f32_split, terms = 3, group_mode 1), large-tile kernel:
// dW[n][k] = sum over tokens m of dY[m][n] X[m][k]  (autograd of ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 145-149, 1001 and
// ref:SeqRec/models/generative/Qwen3Moe/FFN.py:25-27).  Same contract and the same BITS as the 128 x 128 kernel of gemm.hip in its
// deterministic form: a workgroup sums one chunk of tokens into partial 128 x 128 tiles of the chunk workspace (same layout, same order
// of the piece products per accumulator), wgrad_reduce_kernel adds the chunks in order.
//
// Why a second kernel.  In the 128 x 128 kernel a wave owns a 64 x 64 patch (2 x 2 MFMA tiles): per 16 tokens it reads 8 fragments of
// 512 bytes from LDS for 12 MFMAs, and every element of X is cut and stored once per 128 output rows: ~1 KB of LDS traffic per MFMA,
// which IS the LDS bandwidth (128 B / clock / CU against four matrix pipes at 32 clocks per MFMA) - 39 % MFMA-busy, 4.7 vector
// instructions per MFMA (profiles/r05a_mfma_busy.md).  Here a workgroup of FOUR waves (one per SIMD, 512 registers each) owns a
// 256 x 256 tile of dW, a wave a 128 x 128 patch (4 x 4 MFMA tiles, 256 accumulator registers): 16 fragments per 48 MFMAs
// (170 B / MFMA), each operand element cut once per 256 output rows / columns; two 64-KB LDS stages of 32 tokens, one barrier per stage,
// global loads two stages ahead in registers.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace gamer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef WG_ABLATE
#define WG_ABLATE 0       // timing-only builds: 1 no cut + LDS stores, 2 no MFMAs, 4 no global loads after the first, 8 no workspace stores
#endif
#ifndef WG_SETS_A
#define WG_SETS_A 2       // register sets (= stages in flight) of the dY loads
#endif
#ifndef WG_SETS_B
#define WG_SETS_B 2       // ... of the X loads
#endif
constexpr int WG_THREADS = 256;
constexpr int WG_T = 256;                       // rows and columns of dW per workgroup
constexpr int WG_BK = 32;                       // tokens per stage
constexpr int WG_ROWB = 2 * WG_T;               // bytes of one token row of a piece image
constexpr int WG_IMG = WG_BK * WG_ROWB;         // one piece image [32 tokens][256 rows] of 16-bit values: 16 KB
constexpr int WG_STAGE = 4 * WG_IMG;            // dY h0 | dY h1 | X h0 | X h1
constexpr int WG_LDS = 2 * WG_STAGE;            // 128 KB

struct WgParams {
    const float* A; int64_t a_ks;               // dY: A(row n, token m) at A[m a_ks + n]
    const float* B; int64_t b_ks;               // X:  B(col k, token m) at B[m b_ks + k]
    float* ws;                                  // [chunk][128-tile][128][128]
    int M, N, K;                                // dW is [M, N]; K tokens
    float alpha;
    int groups; const int32_t* group_offsets;
    int kchunk;
    int mt, nt;                                 // 256-tiles
    int mt128, nt128;                           // 128-tiles (the workspace's and the reduce kernel's tiling)
    const uint32_t* amax_a; const uint32_t* amax_b;
};

__device__ __forceinline__ int wg_xcd_remap(int id, int n) {      // consecutive logical ids on one XCD (as gemm.hip: xcd_remap)
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, idx = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// byte offset of (token k, row) in a piece image: 16-byte chunk (row >> 3) of the token's row XORed with (k & 3) << 2 - the 8-byte
// stores of a float4's pieces and the transposing reads both spread over all banks (gemm.hip: sp_rc_off, rows of 256 instead of 128)
__device__ __forceinline__ int wg_off(int k, int row) { return k * WG_ROWB + ((((row >> 3) ^ ((k & 3) << 2))) << 4) + ((row & 7) << 1); }

template <bool FULL>
__global__ void __launch_bounds__(WG_THREADS, 1)
gemm_wg_kernel(const WgParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform, and the compiler knows)
    const int wm = wid >> 1, wn = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int L = wg_xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.mt * p.nt;
    const int chunk = L / tiles, tile = L % tiles;
    const int r0 = (tile / p.nt) * WG_T, c0 = (tile % p.nt) * WG_T;
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
    (void)g;
    const int kbeg = seg_beg + (chunk - chunks_before) * p.kchunk;
    const int kend = min(seg_end, kbeg + p.kchunk);

    float scale_a, scale_b, ia, ib;
    scale_from_amax(amax_read(p.amax_a), scale_a, ia);
    scale_from_amax(amax_read(p.amax_b), scale_b, ib);
    const float alpha_eff = p.alpha * (ia * ib);

    // live 32-tiles of this wave's 128 x 128 patch (FULL: all sixteen)
    const int ni = FULL ? 4 : max(0, min(4, (p.M - (r0 + wm * 128) + 31) >> 5));
    const int nj = FULL ? 4 : max(0, min(4, (p.N - (c0 + wn * 128) + 31) >> 5));

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- global -> registers: float4 #j of a thread = token (wave + 4 j) of a stage, rows / columns 4 lane .. 4 lane + 3.  TWO
    // stages are in flight (128 registers = 128 KB per CU: with one the kernel ran at the latency of a load per stage).  Tokens past
    // the chunk's end (ragged expert segments, and the stages past the last one) and rows / columns past the edge of dW are zeros.
    float4 ra[WG_SETS_A][8], rb[WG_SETS_B][8];
    // Every load is UNCONDITIONAL: a token past the chunk's end reads the chunk's last token instead and a quad past the edge of dW
    // the last quad inside it (both zeroed at the cut).  A load inside a branch makes the compiler's wait counters pessimistic at the
    // merge point - it then waits for ALL loads in flight before each cut, and the kernel ran at the latency of a load per group.
    const int a_col = FULL ? r0 + 4 * lane : min(r0 + 4 * lane, (p.M - 1) & ~3);
    const int b_col = FULL ? c0 + 4 * lane : min(c0 + 4 * lane, (p.N - 1) & ~3);
    const int a_left = FULL ? 4 : p.M - (r0 + 4 * lane), b_left = FULL ? 4 : p.N - (c0 + 4 * lane);      // valid elements of this thread's quads
    int ka_load = kbeg + wid, kb_load = kbeg + wid;                                // token of float4 #0 of the stage that is loaded next
    int k_cut = kbeg + wid;                                                        // ... of the stage that is cut next
    auto load_q = [&](const float* base, int64_t ks, int col, int k) -> float4 {
        if ((WG_ABLATE & 4) && k >= kbeg + 3 * WG_BK) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int kc = min(k, kend - 1);                                           // (wave-uniform)
        return *reinterpret_cast<const float4*>(base + (int64_t)kc * ks + col);
    };
    auto load_a = [&](int set, int j) { ra[set][j] = load_q(p.A, p.a_ks, a_col, ka_load + 4 * j); };
    auto load_b = [&](int set, int j) { rb[set][j] = load_q(p.B, p.b_ks, b_col, kb_load + 4 * j); };
    auto next_a = [&]() { ka_load += WG_BK; };
    auto next_b = [&]() { kb_load += WG_BK; };
    // zeros for what lies outside: tokens past the end are cut with scale 0 (a scalar select; the values read instead are the
    // chunk's last token's - finite unless the operand holds Inf / NaN, and then dW does anyway), quads across the edge per element
    auto edge = [&](float4 v, int left) -> float4 {
        if (!FULL) {
            v.x = left > 0 ? v.x : 0.f; v.y = left > 1 ? v.y : 0.f; v.z = left > 2 ? v.z : 0.f; v.w = left > 3 ? v.w : 0.f;
        }
        return v;
    };
    // ---- registers -> cut -> LDS: token (wave + 4 j): k & 3 = wave; rows 4 lane ..: chunk lane >> 1, half lane & 1
    const int st_base = wid * WG_ROWB + ((((lane >> 1) ^ (wid << 2))) << 4) + ((lane & 1) << 3);
    // the cut in two halves of four instructions (common.h: cut2h_quad), so that each half sits in the shadow of one MFMA
    uint32_t c_a0 = 0, c_b0 = 0;
    float4 c_v = make_float4(0.f, 0.f, 0.f, 0.f);
    float c_sc = 0.f;
    auto cut_part1 = [&](const float4& raw, float sc, int j, int operand) {
        c_v = edge(raw, operand ? b_left : a_left);
        c_sc = (k_cut + 4 * j < kend) ? sc : 0.f;
        if (WG_ABLATE & 1) return;
        asm("v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
            "v_fma_mixlo_f16 %1, %4, %6, 0\n\t"
            "v_fma_mixhi_f16 %0, %3, %6, 0\n\t"
            "v_fma_mixhi_f16 %1, %5, %6, 0"
            : "=&v"(c_a0), "=&v"(c_b0) : "v"(c_v.x), "v"(c_v.y), "v"(c_v.z), "v"(c_v.w), "v"(c_sc));
    };
    auto cut_part2_store = [&](unsigned char* st, int j, int operand) {
        if (WG_ABLATE & 1) { asm volatile("" :: "v"(c_v.x), "v"(c_v.y), "v"(c_v.z), "v"(c_v.w)); return; }
        uint32_t a1, b1;
        asm("v_fma_mixlo_f16 %0, %4, %8, -%2 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %1, %6, %8, -%3 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %5, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %7, %8, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "s_nop 0"
            : "=&v"(a1), "=&v"(b1) : "v"(c_a0), "v"(c_b0), "v"(c_v.x), "v"(c_v.y), "v"(c_v.z), "v"(c_v.w), "v"(c_sc));
        unsigned char* d = st + operand * 2 * WG_IMG + st_base + j * 4 * WG_ROWB;
        *reinterpret_cast<uint2*>(d) = make_uint2(c_a0, c_b0);
        *reinterpret_cast<uint2*>(d + WG_IMG) = make_uint2(a1, b1);
    };
    auto cut_store = [&](unsigned char* st, const float4& raw, float sc, int j, int operand) {
        cut_part1(raw, sc, j, operand);
        cut_part2_store(st, j, operand);
    };
    // ---- fragments: transposing reads (ds_read_tr16_b64): lane 16 g + 4 q + pp supplies the address of (token 4-block row q, rows
    // 4 pp .. 4 pp + 3) of its group's 4 x 16 block and receives row (lane & 15) of the four tokens.  Token = 16 sub + 8 h + 4 c + q
    // (k & 3 = q), row = base + 32 i + 16 g + 4 pp: chunk = (base / 8 + 4 i + 2 g + (pp >> 1)) ^ (q << 2) = base / 8 + 4 (i ^ q) + ...
    const int q = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
    const int fr_lane = (8 * h + q) * WG_ROWB + ((2 * gsel + (pp >> 1)) << 4) + ((pp & 1) << 3);
    int fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = fr_lane + ((16 * wm + 4 * (i ^ q)) << 4);
        fb[i] = fr_lane + ((16 * wn + 4 * (i ^ q)) << 4) + 2 * WG_IMG;
    }
    auto read_frag = [&](const unsigned char* st, int off) -> f16x8 {
        bf16x8 out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(st + off + c * 4 * WG_ROWB));
            out[4 * c + 0] = v[0]; out[4 * c + 1] = v[1]; out[4 * c + 2] = v[2]; out[4 * c + 3] = v[3];
        }
        return __builtin_bit_cast(f16x8, out);
    };
    // ONE fragment set.  The three products of a 16-token step run in the order a1.b0, a0.b0, a0.b1 (gemm.hip's weight-gradient
    // layout uses the same order: same bits): each piece's registers are free one product before the next step needs them for the
    // same piece - a1 after the first product, b0 after the second - so the next step's fragments are read while this one multiplies.
    f16x8 af[4][2], bf[4][2];                        // [32-tile][piece]
    auto read_a = [&](const unsigned char* st, int sub, int pc, int half) {
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i) af[i][pc] = read_frag(st, fa[i] + pc * WG_IMG + sub * 16 * WG_ROWB);
    };
    auto read_b = [&](const unsigned char* st, int sub, int pc, int half) {
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i) bf[i][pc] = read_frag(st, fb[i] + pc * WG_IMG + sub * 16 * WG_ROWB);
    };
    // MFMA (gi, j), gi = 4 t + i: product t of row tile i with column tile j
    auto mfma1 = [&](int gi, int j) {
        if (WG_ABLATE & 2) return;
        const int t = gi >> 2, i = gi & 3;
        const int qa = t == 0 ? 1 : 0, qb = t == 2 ? 1 : 0;
        if (!FULL && (i >= ni || j >= nj)) return;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][qa], bf[j][qb], acc[i][j], 0, 0, 0);
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };      // (global loads stay in flight)

    // One stage of 32 tokens = two steps of twelve groups of four MFMAs.  Everything else is dealt out between the MFMAs, one slot
    // each (a wave hides ~six vector instructions in the 32 cycles of an MFMA, tools/ubench_fill.hip; one wave per SIMD: nobody else
    // fills a bubble): slot 0 first half of a cut, slot 1 second half + LDS store, slot 2 four fragment reads, slot 3 the global load
    // that re-requests the cut quad's registers for stage s + 1 + (sets).  At stage s the values of stage s + 1 (register set SET) are
    // cut and stored.  At entry a1 and b0 of (cur, step 0) are in registers.
    auto stage = [&](auto set_c, unsigned char* cur, unsigned char* nxt) {
        constexpr int SA = WG_SETS_A == 2 ? decltype(set_c)::value : 0, SB = WG_SETS_B == 2 ? decltype(set_c)::value : 0;
#define WG_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int gi = 0; gi < 12; ++gi) {                 // tokens 0 .. 15
            const int jq = gi < 8 ? gi : gi - 8, opnd = gi < 8 ? 0 : 1;
            mfma1(gi, 0);
            cut_part1(opnd ? rb[SB][jq] : ra[SA][jq], opnd ? scale_b : scale_a, jq, opnd);
            WG_SB();
            mfma1(gi, 1);
            cut_part2_store(nxt, jq, opnd);
            WG_SB();
            mfma1(gi, 2);
            if (gi < 2) read_a(cur, 0, 0, gi);            // this step's a0 (from the second product on), b1 (third)
            else if (gi < 4) read_b(cur, 0, 1, gi - 2);
            else if (gi < 6) read_a(cur, 1, 1, gi - 4);   // the next step's a1, b0
            else if (gi >= 8 && gi < 10) read_b(cur, 1, 0, gi - 8);
            WG_SB();
            mfma1(gi, 3);
            if (opnd) load_b(SB, jq); else load_a(SA, jq);
            WG_SB();
        }
#pragma unroll
        for (int gi = 0; gi < 12; ++gi) {                 // tokens 16 .. 31
            mfma1(gi, 0);
            if (gi < 4) cut_part1(rb[SB][4 + gi], scale_b, 4 + gi, 1);
            WG_SB();
            mfma1(gi, 1);
            if (gi < 4) cut_part2_store(nxt, 4 + gi, 1);
            WG_SB();
            mfma1(gi, 2);
            if (gi < 2) read_a(cur, 1, 0, gi);
            else if (gi < 4) read_b(cur, 1, 1, gi - 2);
            else if (gi < 6) read_a(nxt, 0, 1, gi - 4);   // (after the barrier below)
            else if (gi >= 8 && gi < 10) read_b(nxt, 0, 0, gi - 8);
            WG_SB();
            mfma1(gi, 3);
            if (gi < 4) load_b(SB, 4 + gi);
            if (gi == 3) lds_barrier();                   // every wave has read `cur` (into registers) and stored its part of `nxt`
            WG_SB();
        }
#undef WG_SB
        next_a(); next_b();
        k_cut += WG_BK;
    };

    const int n_st = (kend - kbeg + WG_BK - 1) / WG_BK;
    // stage 0 -> LDS; stages 1 (and 2, with two register sets) in flight
#pragma unroll
    for (int j = 0; j < 8; ++j) { load_a(0, j); load_b(0, j); }
    next_a(); next_b();
    if (WG_SETS_A == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) load_a(1, j);
        next_a();
    }
    if (WG_SETS_B == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) load_b(1, j);
        next_b();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { cut_store(wg_smem, ra[0][j], scale_a, j, 0); cut_store(wg_smem, rb[0][j], scale_b, j, 1); }
    k_cut += WG_BK;
#pragma unroll
    for (int j = 0; j < 8; ++j) { load_a(0, j); load_b(0, j); }
    next_a(); next_b();
    lds_barrier();
    read_a(wg_smem, 0, 1, 0); read_a(wg_smem, 0, 1, 1);
    read_b(wg_smem, 0, 0, 0); read_b(wg_smem, 0, 0, 1);
    {
        // (no branch around a stage inside the loop: a merge point there makes the register allocator move accumulators)
        int s = 0;
#pragma unroll 1
        for (; s + 1 < n_st; s += 2) {
            stage(std::integral_constant<int, 1>{}, wg_smem, wg_smem + WG_STAGE);
            stage(std::integral_constant<int, 0>{}, wg_smem + WG_STAGE, wg_smem);
        }
        if (s < n_st) stage(std::integral_constant<int, 1>{}, wg_smem, wg_smem + WG_STAGE);
    }

    // ---- the chunk's partial tiles -> workspace (a wave's 128 x 128 patch is one tile of the 128-tiling; rows / columns past the
    // edge of dW inside a live tile hold exact zeros: their operand rows were staged as zeros)
    if (WG_ABLATE & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][15]));
        return;
    }
    const int tr = r0 / 128 + wm, tc = c0 / 128 + wn;
    if (tr >= p.mt128 || tc >= p.nt128) return;
    float* blk = p.ws + ((int64_t)chunk * (p.mt128 * p.nt128) + (int64_t)tr * p.nt128 + tc) * (int64_t)(128 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lc = j * 32 + r32, lr0 = i * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) blk[(lr0 + (r & 3) + 8 * (r >> 2)) * 128 + lc] = alpha_eff * acc[i][j][r];
        }
}

static std::atomic<long long> g_wg_launches{0};

static inline int gemm_wg_mode() {
    static EnvSwitch sw("GAMER_GEMM_WG");                 // (cached: gamer_reload_env() after a change inside the process)
    return sw.get(1);                                     // 0 off, 1 the shapes it is faster on, 2 every shape it can compute
}

// Does this weight gradient take the large-tile kernel?  Three-product form, deterministic chunk workspace, both operands
// row-contiguous with 16-byte aligned rows; by default only when dW is whole 256 x 256 tiles (q|k|v, gate, down, gate|up at
// d_in = 256): a partly filled tile streams its operands at the full rate for a fraction of the MFMAs, and the 128 x 128 kernel
// is faster on o_proj (256 x 384), the head (1041 x 256) and the injecting layers' gate|up (1024 x 320).
bool gemm_wg_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc) {
    const int mode = gemm_wg_mode();
    if (!mode || a_kc || b_kc || d->group_mode != 1 || !d->wgrad_ws || !d->amax_a || !d->amax_b) return false;
    if (d->a_ks % 4 != 0 || d->b_ks % 4 != 0 || d->kchunk % WG_BK != 0 || d->M < 1 || d->N < 1) return false;
    return mode >= 2 || (d->M % WG_T == 0 && d->N % WG_T == 0);
}

int launch_gemm_wg(const gamer_gemm_desc* d, hipStream_t st) {
    WgParams p;
    p.A = d->A; p.a_ks = d->a_ks;
    p.B = d->B; p.b_ks = d->b_ks;
    p.ws = d->wgrad_ws;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.alpha = d->alpha;
    p.groups = d->groups; p.group_offsets = d->group_offsets;
    p.kchunk = d->kchunk;
    p.mt = (d->M + WG_T - 1) / WG_T; p.nt = (d->N + WG_T - 1) / WG_T;
    p.mt128 = (d->M + 127) / 128; p.nt128 = (d->N + 127) / 128;
    p.amax_a = d->amax_a; p.amax_b = d->amax_b;
    const int64_t chunks = (d->K + d->kchunk - 1) / d->kchunk + (d->group_offsets ? d->groups : 0);
    const int64_t blocks = chunks * p.mt * p.nt;
    if (blocks >= (1LL << 31)) { set_error("gamer_gemm_f32_split: weight-gradient grid too large"); return 1; }
    const bool full = d->M % WG_T == 0 && d->N % WG_T == 0;
#define GAMER_LAUNCH_WG(FULLV)                                                                                                \
    do {                                                                                                                      \
        static bool attr_dev[MAX_DEVICES] = {};                                                                               \
        if (!attr_dev[current_device()]) {                                                                                    \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wg_kernel<FULLV>),                    \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);                     \
            if (e != hipSuccess) { set_error("gamer_gemm_f32_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; } \
            attr_dev[current_device()] = true;                                                                                \
        }                                                                                                                     \
        hipLaunchKernelGGL(gemm_wg_kernel<FULLV>, dim3((int)blocks), dim3(WG_THREADS), WG_LDS, st, p);                        \
    } while (0)
    if (full) GAMER_LAUNCH_WG(true); else GAMER_LAUNCH_WG(false);
#undef GAMER_LAUNCH_WG
    GAMER_CHECK_LAUNCH("gamer_gemm_f32_split/weight gradient, 256 x 256 tiles");
    g_wg_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

}  // namespace gamer

// Diagnostic (not in include/gamer_hip.h): launches of the large-tile weight-gradient kernel by this process so far (its results are
// the bits of the 128 x 128 kernel's: a test cannot tell from them which of the two ran).
extern "C" long long gamer_debug_gemm_wg_launches(void) { return gamer::g_wg_launches.load(std::memory_order_relaxed); }
