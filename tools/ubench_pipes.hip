// Micro-benchmark: do fp32 MFMA and fp32 VALU overlap on a gfx950 SIMD, or do they share the pipe?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pipes.hip -o gpurun_out/ubench_pipes && gpurun_out/ubench_pipes
// Every workgroup has 8 waves = 2 per SIMD.  Roles per wave: M = fp32 MFMA loop, B = bf16 MFMA loop,
// V = v_fma_f32 loop, I = idle (exits).  Time for a fixed amount of per-wave work tells whether the second
// wave's work hides behind the first wave's or adds to it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Role { IDLE = 0, MFMA32 = 1, VALU = 2, MFMABF = 3, MIXED = 4, MIXEDBF = 5, CHAIN1 = 6, CHAIN2 = 7 };

__global__ void __launch_bounds__(512) pipes(int role_lo, int role_hi, int iters, int valu_per_mfma, float* out) {
    const int w = threadIdx.x >> 6;
    const int role = w < 4 ? role_lo : role_hi;
    const float x = (float)(threadIdx.x & 7) * 0.25f + 1.f;
    if (role == IDLE) return;
    if (role == CHAIN1 || role == CHAIN2) {
        // dependent accumulation chains: 4 MFMAs per iteration on one (CHAIN1) or two (CHAIN2) accumulators
        f32x16 a0 = {0}, a1 = {0};
        for (int i = 0; i < iters; ++i) {
            if (role == CHAIN1) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a1, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else if (role == MFMA32) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a3, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else if (role == MFMABF) {
        bf16x8 b;
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)x;
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, a3, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else if (role == VALU) {
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v4) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v5) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v6) : "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v7) : "v"(x));
            }
        }
        const float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        if (s == 12345.f) out[threadIdx.x] = s;
    } else {
        // one wave: 4 MFMAs and 4*valu_per_mfma independent v_fma_f32 per iteration, statically interleaved
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        bf16x8 b;
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)x;
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
        auto valu8 = [&]() {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v4) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v5) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v6) : "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v7) : "v"(x));
        };
        for (int i = 0; i < iters; ++i) {
            if (role == MIXED) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(a0) : "v"(x));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(a1) : "v"(x));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(a2) : "v"(x));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(a3) : "v"(x));
                if (valu_per_mfma >= 8) valu8();
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(a0) : "v"(b));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(a1) : "v"(b));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(a2) : "v"(b));
                if (valu_per_mfma >= 8) valu8();
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(a3) : "v"(b));
                if (valu_per_mfma >= 8) valu8();
            }
        }
        float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
        if (s == 12345.f) out[threadIdx.x] = s;
    }
}

static double run(int lo, int hi, int iters, int vpm, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) pipes<<<256, 512>>>(lo, hi, iters, vpm, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) pipes<<<256, 512>>>(lo, hi, iters, vpm, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10.0;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int it = 20000;
    struct Case { const char* name; int lo, hi, vpm; } cases[] = {
        {"fp32 MFMA x1 wave/SIMD (4*it MFMA)", MFMA32, IDLE, 0},
        {"fp32 MFMA x2 waves/SIMD", MFMA32, MFMA32, 0},
        {"VALU x1 wave/SIMD (32*it v_fma)", VALU, IDLE, 0},
        {"VALU x2 waves/SIMD", VALU, VALU, 0},
        {"fp32 MFMA wave + VALU wave", MFMA32, VALU, 0},
        {"fp32 MFMA 1 dependent chain, 1 wave/SIMD", CHAIN1, IDLE, 0},
        {"fp32 MFMA 1 dependent chain, 2 waves/SIMD", CHAIN1, CHAIN1, 0},
        {"fp32 MFMA 2 chains, 1 wave/SIMD", CHAIN2, IDLE, 0},
        {"fp32 MFMA 2 chains, 2 waves/SIMD", CHAIN2, CHAIN2, 0},
        {"bf16 MFMA x1 wave/SIMD", MFMABF, IDLE, 0},
        {"bf16 MFMA x2 waves/SIMD", MFMABF, MFMABF, 0},
        {"bf16 MFMA wave + VALU wave", MFMABF, VALU, 0},
        {"one wave: fp32 MFMA + 8 v_fma each", MIXED, IDLE, 8},
        {"one wave: fp32 MFMA + 0 v_fma", MIXED, IDLE, 0},
        {"two waves: fp32 MFMA + 8 v_fma each", MIXED, MIXED, 8},
        {"one wave: bf16 MFMA + 8 v_fma each", MIXEDBF, IDLE, 8},
        {"one wave: bf16 MFMA + 0 v_fma", MIXEDBF, IDLE, 0},
    };
    for (auto& c : cases) {
        const double ms = run(c.lo, c.hi, it, c.vpm, out);
        printf("%-44s %8.3f ms   (%.1f ns per iteration)\n", c.name, ms, ms * 1e6 / it);
    }
    return 0;
}
