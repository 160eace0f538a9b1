// Shared device/host helpers for libgamer_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -----------------------------------------------------------------------
constexpr int WAVE = 64;

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

// Dropout on attention probabilities: one 32-bit hash per (row, key pair) gives the two 16-bit
// uniform numbers of keys 2*jp and 2*jp+1, so the forward / dQ kernels (key pairs in adjacent
// accumulator registers of one lane) hash once per two elements.  keep iff u16 >= p*65536.
struct AttnDropout {
    uint32_t k0, thr16;
    float scale;
    bool on;
    __device__ __forceinline__ AttnDropout(float p, uint64_t seed) {
        on = p > 0.f;
        k0 = mix32((uint32_t)seed ^ 0x9e3779b9U) ^ mix32((uint32_t)(seed >> 32) + 0x7f4a7c15U);
        thr16 = (uint32_t)(p * 65536.f);
        scale = on ? 1.f / (1.f - p) : 1.f;
    }
    // row = (b*nq + head)*S + i
    __device__ __forceinline__ uint32_t row_base(uint32_t row) const { return mix32(row * 0x9e3779b1U + k0); }
    __device__ __forceinline__ uint32_t pair_bits(uint32_t rb, uint32_t jp) const { return mix32(rb + jp * 0x85ebca6bU); }
    __device__ __forceinline__ float mult_even(uint32_t bits) const { return (bits & 0xffffU) >= thr16 ? scale : 0.f; }
    __device__ __forceinline__ float mult_odd(uint32_t bits) const { return (bits >> 16) >= thr16 ? scale : 0.f; }
    __device__ __forceinline__ float mult(uint32_t rb, uint32_t j) const {
        const uint32_t bits = pair_bits(rb, j >> 1);
        return (j & 1U) ? mult_odd(bits) : mult_even(bits);
    }
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// d/dx silu(x) = s*(1 + x*(1-s)), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
    float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}

}  // namespace gamer
