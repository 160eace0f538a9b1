// Shared device/host helpers for libgamer_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/gamer_hip.h"

namespace gamer {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define GAMER_CHECK_ARG(cond, ...)                       \
    do {                                                 \
        if (!(cond)) {                                   \
            gamer::set_error(__VA_ARGS__);               \
            return -1;                                   \
        }                                                \
    } while (0)

// Called right after a kernel launch: hipGetLastError catches bad launch configurations.
#define GAMER_CHECK_LAUNCH(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            gamer::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

// Per-device "done once" flags (hipFuncSetAttribute and friends are per device; a process may drive several GPUs).
constexpr int MAX_DEVICES = 16;
static inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
    return d;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- environment switches (A/B runs, tests) ---------------------------------------------------------------------------------
// Read ONCE per process, not per launch (a step issues hundreds of launches and the kernel choice must not change under a running
// step because another thread called setenv).  gamer_reload_env() starts a new epoch: every switch is read again at its next use -
// what tests and the A/B tools call after they change os.environ inside the process.
unsigned env_epoch();                                   // (prep.hip)
struct EnvSwitch {
    const char* name;
    unsigned seen = 0;                                   // epoch of the cached value (epochs start at 1)
    bool present = false;
    int val = 0;
    explicit EnvSwitch(const char* n) : name(n) {}
    void sync() {
        const unsigned e = env_epoch();
        if (seen == e) return;
        const char* s = getenv(name);
        present = s != nullptr;
        val = s ? atoi(s) : 0;
        seen = e;
    }
    bool is_set() { sync(); return present; }
    int get(int dflt) { sync(); return present ? val : dflt; }
};

// ---- device helpers -----------------------------------------------------------------------
constexpr int WAVE = 64;

// bf16 activations (the AMP variant): raw storage type of the C ABI is uint16_t (gamer_bf16), device code uses the
// compiler's __bf16 so that conversions are v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN).
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));

// four consecutive elements as a float4 (p aligned to 4 elements): 16-byte access for fp32, 8-byte for bf16
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
// EW_NT=1 (elementwise.hip only): nontemporal 16-byte loads / stores.  Measured (round 3, docs/DESIGN_rounds1-4.md section 5): in isolation SwiGLU
// forward / backward gain 5-11 % (6.3 TB/s), in the train step the GEMMs that read those tensors next lose what was gained.
#ifndef EW_NT
#define EW_NT 0
#endif
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) {
    if (EW_NT) { const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
    return *reinterpret_cast<const float4*>(p);
}
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = EW_NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p)) : *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float4 v) {
    if (EW_NT) { const f32x4v f = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(f, reinterpret_cast<f32x4v*>(p)); return; }
    *reinterpret_cast<float4*>(p) = v;
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float4 v) {
    const f32x4v f = {v.x, v.y, v.z, v.w};
    if (EW_NT) { __builtin_nontemporal_store(__builtin_convertvector(f, bf16x4), reinterpret_cast<bf16x4*>(p)); return; }
    *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(f, bf16x4);
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { *p = (T)v; }
// round a value through the activation type (identity for fp32): mirrors autocast's intermediate bf16 tensors
template <typename T> __device__ __forceinline__ float round_as(float v) { return (float)(T)v; }
template <typename T> static inline bool aligned_vec4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0; }

// The two fp16 pieces of four values times s (s a power of two): h0 = fp16(x s), h1 = fp16(x s - h0), both round-to-nearest-even,
// as packed pairs a0 = {x0's, x1's h0}, b0 = {x2's, x3's h0}, a1 / b1 = the h1 pairs.  Eight v_fma_mix instructions = two per
// value - the multiply, the exact residual in fp32 and the conversion in one instruction per piece (the C form: multiply,
// convert, convert back, subtract, convert = 3-3.5 per value).  ONE asm block, ordered so that no instruction reads a register
// the instruction before it wrote in part: on gfx940+ a VALU that reads a VGPR right after an op_sel (half-register) write of
// it needs a wait state, which the compiler inserts for its own instructions but not inside or around inline asm - a first
// form with mixlo / mixhi of one register back to back was right in the GEMM (where the scheduler happened to interleave
// pairs) and lost the h1 pieces in the dropout variants of the attention kernels.  Bit-equal to the C form except that x = -0
// gives (+0, -0) instead of (-0, +0) (tools/mix_cut_test.hip on MI355X: random pairs at five scales, Inf, NaN, the fp16
// overflow boundary, denormals).
__device__ __forceinline__ void cut2h_quad(float x0, float x1, float x2, float x3, float s,
                                           uint32_t& a0, uint32_t& a1, uint32_t& b0, uint32_t& b1) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %2, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s));
}
// the scale of a tensor from the bits of its largest magnitude (a non-negative float): max * s in [2^13, 2^14); 1 for a
// zero, denormal, infinite or NaN maximum (Inf / NaN then reach the MFMA as they are).  inv = 1 / s, exact.
__device__ __forceinline__ void scale_from_amax(uint32_t bits, float& s, float& inv) {
    const int e = (int)((bits >> 23) & 0xffu);
    int se = 127 + 13 - (e - 127);
    if (e == 0 || e == 255) se = 127;
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    s = __uint_as_float((uint32_t)se << 23);
    inv = __uint_as_float((uint32_t)(254 - se) << 23);
}
// ---- maxima of what a kernel stores (matmul = "split3": the consumer GEMM scales its operand by a power of two from it) ----
// gamer_amax_sink(out0, out1) arms the NEXT launch (on the calling host thread) of a kernel that supports it: that kernel folds
// the bits of max |value stored| of its first / second output into *out0 / *out1 (atomicMax; the words hold 0 or an earlier
// maximum).  The pointers travel as kernel arguments - stream-ordered like everything else - and are disarmed by the launch.
struct AmaxSink { uint32_t* out[3]; };          // (out[2]: gamer_amax_sink3 - the q/k-norm forward's third output, v + bias)
AmaxSink take_amax_sink();
void disarm_attn_amax();            // (attention_split.hip) drops what gamer_attn_split_amax armed; set_error calls both
// (float maxima: |x| is a free source modifier and max3 takes two values per instruction; a NaN is ignored here - it reaches the
// GEMM as a NaN piece whatever the scale is)
__device__ __forceinline__ uint32_t amax_f4(uint32_t m, const float4 v) {
    float f = __uint_as_float(m);
    f = fmaxf(fmaxf(f, fabsf(v.x)), fabsf(v.y));
    f = fmaxf(fmaxf(f, fabsf(v.z)), fabsf(v.w));
    return __float_as_uint(f);
}
// A maximum lives in AMAX_WAYS words, AMAX_STRIDE words apart (one 64-byte line each): a launch's workgroups publish into the
// word of their index modulo AMAX_WAYS - thousands of same-address atomics serialise at the memory side (20 us for the first
// resident wave of workgroups of a 25-us kernel at per-GPU batch 128) -, a consumer takes the maximum of the words.
constexpr int AMAX_WAYS = 16, AMAX_STRIDE = 16;
__device__ __forceinline__ uint32_t amax_read(const uint32_t* __restrict__ p) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < AMAX_WAYS; ++i) m = max(m, p[AMAX_STRIDE * i]);
    return m;
}
// (no atomic when the word already holds more; read at agent scope: the atomics execute at the memory side, an XCD's L2 may
// hold an older copy)
__device__ __forceinline__ void amax_publish(uint32_t m, uint32_t* __restrict__ out, uint32_t way) {
    uint32_t* w = out + AMAX_STRIDE * (way & (AMAX_WAYS - 1));
    if (m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, m);
}
// every thread of a 256-thread workgroup calls this once, at the end of the kernel (out is workgroup-uniform); at most one
// atomic per workgroup
__device__ __forceinline__ void amax_block_commit(uint32_t m, uint32_t* __restrict__ out, uint32_t* __restrict__ lds4) {
    if (!out) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(lds4[0], lds4[1]), max(lds4[2], lds4[3]));
        if (m) amax_publish(m, out, blockIdx.x);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout mask.  keep(seed, idx) is a pure function, so the backward kernels
// regenerate exactly the forward mask from the same (seed, element index).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct DropoutRng {
    uint32_t k0, k1, thr;
    float scale;
    bool on;
    __device__ __forceinline__ DropoutRng(float p, uint64_t seed) {
        on = p > 0.f;
        k0 = mix32((uint32_t)seed ^ 0x9e3779b9U);
        k1 = mix32((uint32_t)(seed >> 32) + 0x85ebca6bU);
        // p*2^32, clamped
        float t = p * 4294967296.f;
        thr = t >= 4294967040.f ? 0xffffffffU : (uint32_t)t;
        scale = on ? 1.f / (1.f - p) : 1.f;
    }
    // multipliers of the four elements 4*idx4 .. 4*idx4+3 (one float4): two hashes -> four 16-bit uniforms,
    // keep iff u16 >= p*65536.  This is the form every elementwise kernel and the GEMM epilogue use, so a
    // mask generated in a fused epilogue is regenerated bit-identically by the stand-alone backward kernel.
    __device__ __forceinline__ void mult4(uint32_t idx4, float (&m)[4]) const {
        if (!on) { m[0] = m[1] = m[2] = m[3] = 1.f; return; }
        const uint32_t thr16 = thr >> 16;
        const uint32_t h1 = mix32(idx4 * 0x9e3779b1U + k0);
        const uint32_t h2 = mix32(h1 ^ k1);
        m[0] = (h1 & 0xffffU) >= thr16 ? scale : 0.f;
        m[1] = (h1 >> 16) >= thr16 ? scale : 0.f;
        m[2] = (h2 & 0xffffU) >= thr16 ? scale : 0.f;
        m[3] = (h2 >> 16) >= thr16 ? scale : 0.f;
    }
    // multiplier for element idx: 0 (dropped) or 1/(1-p)   (kept for reference; unused by the kernels)
    __device__ __forceinline__ float mult(uint64_t idx) const {
        if (!on) return 1.f;
        uint32_t h = mix32((uint32_t)idx ^ k0);
        h = mix32(h + (uint32_t)(idx >> 32) * 0x9e3779b1U + k1);
        return h >= thr ? scale : 0.f;
    }
};

// Dropout on attention probabilities.  On gfx950 the fp32 MFMA runs on the same lanes as the fp32 VALU
// (tools/ubench_pipes.hip: an MFMA wave and a v_fma wave on one SIMD take the SUM of their times), so every
// VALU instruction in an attention inner loop is paid in full.  The mask is therefore a multiply-shift hash:
//     keep(i, j) = low32(a_i * b_j) >= p * 2^32
// with a 24-bit odd word a_i per (batch, head, query row) and a 24-bit word b_j per key, both from mix32.
// Per element that is one full-rate v_mul_u32_u24 + one compare; the words are hashed once per row / per key
// tile.  keep() is a pure function of (seed, row, key), so the three kernels regenerate the same mask in
// their own traversal order.  (Row/column keep rates, adjacent-element and 2x2 correlations of this mask are
// indistinguishable from numpy's generator at S = 505; checked offline.)
struct AttnDropout {
    uint32_t k0, k1, thr;
    float scale;
    __device__ __forceinline__ AttnDropout(float p, uint64_t seed) {
        k0 = mix32((uint32_t)seed ^ 0x9e3779b9U) ^ mix32((uint32_t)(seed >> 32) + 0x7f4a7c15U);
        k1 = mix32(k0 + 0x632be5abU);
        const float t = p * 4294967296.f;
        thr = t >= 4294967040.f ? 0xffffffffU : (uint32_t)t;
        scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    }
    // row = (b*nq + head)*S + i
    __device__ __forceinline__ uint32_t row_word(uint32_t row) const {
        return (mix32(row * 0x9e3779b1U + k0) & 0xffffffU) | 0x800001U;
    }
    __device__ __forceinline__ uint32_t key_word(uint32_t j) const { return mix32(j * 0x85ebca6bU + k1) & 0xffffffU; }
    __device__ __forceinline__ bool keep(uint32_t a, uint32_t b) const { return __umul24(a, b) >= thr; }
};

// value of the lane 32 positions away (lane ^ 32) without the LDS crossbar: v_permlane32_swap exchanges the upper half of one
// register with the lower half of another; with both = v the pair becomes (v[l & 31], v[32 + (l & 31)]) in every lane.  A
// ds_bpermute round trip (~100+ cycles) sat twice in the dependent chain of every attention tile (row maximum, row sum).
__device__ __forceinline__ float xor32_max(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// min / max over the 64 lanes without the LDS crossbar: four DPP steps reduce every row of 16 lanes (quad swaps, half-row and row
// mirrors: each lane ends with its row's result), four v_readlane fetch the rows, the rest is scalar.  A __shfl_xor chain is six
// dependent ds_bpermute round trips (~100+ cycles each): it sat in front of every row tile of the resident attention kernels.
// The result is wave-uniform (an SGPR).
#define GAMER_DPP_ROW_REDUCE(OP, v)                                                                   \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));  /* quad_perm [1,0,3,2] */    \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));  /* quad_perm [2,3,0,1] */    \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false)); /* row_half_mirror */        \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false)); /* row_mirror */
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
    GAMER_DPP_ROW_REDUCE(min, v)
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
    GAMER_DPP_ROW_REDUCE(max, v)
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// sum over each row of 16 lanes (every lane ends with its row's sum) in four DPP adds; the partners are those of the xor 1, 2, 4, 8
// butterfly (after the two quad steps a quad's lanes hold the same value, so "mirror" and "xor" pick equal operands): the bits are those of the
// __shfl_xor chain it replaces, without its four dependent ds_bpermute round trips
__device__ __forceinline__ float row16_sum_dpp(float v) {
#define GAMER_DPP_F(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
    v += GAMER_DPP_F(0xB1);
    v += GAMER_DPP_F(0x4E);
    v += GAMER_DPP_F(0x141);
    v += GAMER_DPP_F(0x140);
#undef GAMER_DPP_F
    return v;
}

// silu(x) = x sigmoid(x) with the hardware reciprocal (v_rcp_f32: 1 ulp) instead of an IEEE division (ten instructions): since round 6 the
// SwiGLU forward also runs in a GEMM epilogue (gemm_as.hip, EPI 5), where vector instructions are matrix time; every kernel uses this one
// definition, so a fused epilogue and the stand-alone kernel still produce the same bits
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = s*(1 + x*(1-s)), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
    const float s = sigmoid_f(x);
    return s * (1.f + x * (1.f - s));
}

// (gemm_as.hip) the activation-stationary Linear forward of gamer_gemm_f32_split(terms = 3): eligibility of a descriptor, launch
bool gemm_as_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc, const uint16_t* b_planes);
int launch_gemm_as(const gamer_gemm_desc* d, const uint16_t* b_planes, bool b_kc, hipStream_t st);
bool gemm_os_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc, const uint16_t* b_planes);      // (gemm_os.hip) input gradient, output-stationary
int launch_gemm_os(const gamer_gemm_desc* d, const uint16_t* b_planes, int guard, hipStream_t st, bool fwd_t = false);
bool gemm_os_fwd_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc);        // (gemm_os.hip) Linear forward with 256 outputs on W's transposed pieces
bool gemm_wg_eligible(const gamer_gemm_desc* d, bool a_kc, bool b_kc);      // (gemm_wg.hip) weight gradient, 256 x 256 tiles
int launch_gemm_wg(const gamer_gemm_desc* d, hipStream_t st);

// (gemm.hip) C (+)= the chunk partial tiles of an ordered weight gradient, in chunk order; returns hipGetLastError()
int launch_wgrad_reduce(const float* ws, float* C, int64_t ldc, int M, int N, int groups, const int32_t* group_offsets, int K,
                        int kchunk, int64_t strideC, hipStream_t st);

}  // namespace gamer
