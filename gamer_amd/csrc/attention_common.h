// Shared pieces of the attention kernels (attention.hip: fp32 MFMA; attention_split.hip: the same algorithm with its
// products formed on the bf16 pipe from exact three-way cuts): tile constants, row order / query span / work list
// structs, fp32 tile staging and key metadata.  See attention.hip for the algorithm.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace gamer {

constexpr int AT_THREADS = 256;
constexpr int KLD = 68;                 // floats per row of a [32][64] tile image read as MFMA fragments
constexpr int INT_BIG_A = 0x7fffffff;

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Optional query-row order (behaviour-level "cross" attention): rows with no allowed key ("empty", p = 1/S
// over all keys, no scores needed) are moved behind the others inside every sequence, so a 32-row wave tile
// is (almost always) either all normal rows -> causal key range only, or all empty rows -> no Q.K^T at all.
// Without it one empty row forces its whole wave through the full key range.  perm == nullptr: identity.
struct RowOrder {
    const int32_t* perm;         // [B,S] sorted slot -> position in the sequence
    const int32_t* tile_kind;    // [B,ceil(S/32)] bit0: has normal rows, bit1: has empty rows
    const int32_t* tile_maxpos;  // [B,ceil(S/32)] largest position of a normal row in the tile, -1 if none
    const int32_t* span;         // [B,S,4] optional per-query key span (QuerySpan); travels with the row metadata
};

// Session-wise masks (ref:SeqRec/models/generative/Qwen3SessionMulti/model.py:556-613, 676-728) as a per-query
// key span: key j is allowed iff j <= hi and not (hole_lo <= j < hole_hi) (and kl[j] < ql[i] as always), hi <= i.
// Kernels instantiated with SPAN = false use hi = i and no hole (Qwen3Multi) and carry none of this.
struct QuerySpan {
    int hi, hole_lo, hole_hi;
    template <bool SPAN>
    static __device__ __forceinline__ QuerySpan load(const int32_t* __restrict__ span, int64_t tok, int pos, bool valid, int S) {
        QuerySpan q;
        q.hi = valid ? pos : S;              // rows past the end: beyond every key (never stored)
        q.hole_lo = INT_BIG_A;
        q.hole_hi = 0;
        if (SPAN && valid) {
            const int4 t = reinterpret_cast<const int4*>(span)[tok];
            q.hi = t.x; q.hole_lo = t.y; q.hole_hi = t.z;
        }
        return q;
    }
};

// Static work partition for persistent workgroups.
//
// Measured on MI355X (tools/trace_attn.py): with one workgroup per (sequence, kv head, tile) the causal
// tiles differ 8x in duration and the in-order round-robin dispatcher (XCD -> SE -> CU) leaves CUs idle
// behind long workgroups: 1.3 of 2 possible workgroups resident per CU on average, per-CU busy time
// between 17 % and 98 %.  So the grid is sized to the chip (2 workgroups per CU) and every workgroup
// walks a list of equal-work items: item = (pair = b*nkv + kvh, u) -> the heavy tile and the matching
// light tile of that pair, i.e. the same number of inner iterations for every u.
// Workgroup ids are dealt round-robin over the 8 XCDs, so XCD x = id & 7 takes the pairs = x (mod 8)
// and its workgroups run all tiles of a pair at the same time: K/V (or Q/dO) of a pair stay in that
// XCD's L2.  Placement only affects speed, never results.
struct WorkList {
    int xcd, pair_slot, u, pairs_per_round, n_pairs, n_tiles;
    bool valid;
    __device__ __forceinline__ WorkList(int n_pairs_, int n_tiles_) : n_pairs(n_pairs_), n_tiles(n_tiles_) {
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
static inline int worklist_grid(int n_pairs, int n_tiles) {
    const int halves = (n_tiles + 1) / 2;
    int per_xcd = (n_pairs + 7) / 8;                     // pairs one XCD has to process
    int rounds_cap = 64 / halves;                        // 32 CUs x 2 workgroups per XCD
    if (rounds_cap < 1) rounds_cap = 1;
    const int ppr = per_xcd < rounds_cap ? per_xcd : rounds_cap;
    return 8 * ppr * halves;
}

// tile of 32 rows x 64 floats: thread -> 2 float4 (f = tid + 256*jj: row f>>4, quad f&15).
// Rows >= r_end are read from the last valid row (clamped address, no branch) and zeroed by store_tile32:
// a select placed on the loaded registers would make the wave wait for the load right where it was issued
// instead of one iteration later, in front of the LDS store.
__device__ __forceinline__ void load_tile32(const float* __restrict__ base, int64_t ld, int r0, int r_end, int tid,
                                            float4 (&rg)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int f = tid + AT_THREADS * jj;
        const int row = min(r0 + (f >> 4), r_end - 1);
        rg[jj] = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + ((f & 15) << 2));
    }
}
// same tile, rows gathered through a slot -> position map
__device__ __forceinline__ void load_tile32_rows(const float* __restrict__ base, int64_t ld, const int32_t* __restrict__ map,
                                                 int r0, int r_end, int tid, float4 (&rg)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int f = tid + AT_THREADS * jj;
        const int pos = map[min(r0 + (f >> 4), r_end - 1)];
        rg[jj] = *reinterpret_cast<const float4*>(base + (int64_t)pos * ld + ((f & 15) << 2));
    }
}

// n_valid = r_end - r0 of the matching load: rows at or past it are stored as zeros
template <int LD>
__device__ __forceinline__ void store_tile32(float* __restrict__ lds, int tid, const float4 (&rg)[2], int n_valid) {
    if (n_valid >= 32) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int f = tid + AT_THREADS * jj;
            *reinterpret_cast<float4*>(lds + (f >> 4) * LD + ((f & 15) << 2)) = rg[jj];
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int f = tid + AT_THREADS * jj;
            const bool ok = (f >> 4) < n_valid;
            float4 v = rg[jj];
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            *reinterpret_cast<float4*>(lds + (f >> 4) * LD + ((f & 15) << 2)) = v;
        }
    }
}

// Per staged key tile: key levels, dropout key words and the largest key level (a key past the end of the
// sequence counts as INT_MAX), produced by three different waves while the tile is loaded.
struct KeyMeta {
    int32_t kl[32];
    uint32_t kw[32];
    int32_t klmax;
    int32_t pad_[3];
};

template <bool DROP>
__device__ __forceinline__ int key_meta_load(const int32_t* __restrict__ klb, int j0, int S, int w, int lane,
                                             const AttnDropout& rng) {
    const int j = j0 + (lane & 31);
    int v = 0;
    if (w == 0 || w == 1) {
        v = j < S ? klb[j] : INT_BIG_A;      // wave 1 reduces at store time, when the load has landed
    } else if (w == 2) {
        if (DROP) v = (int)rng.key_word((uint32_t)j);
    }
    return v;
}
__device__ __forceinline__ void key_meta_store(KeyMeta& km, int w, int lane, int v) {
    if (w == 0) { if (lane < 32) km.kl[lane] = v; }
    else if (w == 1) { v = wave_max_i32(v); if (lane == 0) km.klmax = v; }
    else if (w == 2) { if (lane < 32) km.kw[lane] = (uint32_t)v; }
}

// The 16 score registers of a lane are keys rowmap(reg, h) = (reg & 3) + 8 * (reg >> 2) + 4 * h of the tile:
// four int4 LDS reads fetch their levels (or dropout words).
__device__ __forceinline__ void read_key_quads(const int32_t* __restrict__ base, int h, int (&out)[16]) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int4 t = *reinterpret_cast<const int4*>(base + 8 * g4 + 4 * h);
        out[4 * g4] = t.x; out[4 * g4 + 1] = t.y; out[4 * g4 + 2] = t.z; out[4 * g4 + 3] = t.w;
    }
}

// Online softmax with a lazily updated reference: p = exp2(s - m_ref) is exact for ANY reference (floating
// point is scale free), so m_ref only moves when a score exceeds it by more than 2^RESCALE_TAU (or nothing has
// been accumulated yet).  After the first key tile that almost never happens and the 32 accumulator
// multiplications + exp2 of the classic scheme disappear from the loop.
constexpr float RESCALE_TAU = 20.f;

}  // namespace gamer
