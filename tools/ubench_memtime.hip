// What the matrix pipe sustains with the WHOLE chip busy, by operand data: constant operands vs random ones.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_memtime.hip -o tools/_ab/ubench_memtime && tools/_ab/ubench_memtime
// Grid = 256 CUs x (1 or 2) workgroups of 4 waves: one or two waves per SIMD, every wave issues 4 * iters MFMAs on four
// independent accumulators (no memory traffic, no LDS, no other vector work in the loop).  Time comes from HIP events.
// Result on MI355X (profiles/r03_mfma_sustained.txt): with random operand bits the chip sustains ~57 % of the nominal bf16
// rate and with constant operands ~75 % - the clock follows the switching activity of the multipliers (power limit), so a
// kernel's "% of 2.5 PFLOP/s" cannot exceed those figures no matter how its loop is scheduled.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>      // 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_32x32x2_f32
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ src, int iters, float* out) {
    const uint4 ua = src[threadIdx.x], ub = src[256 + threadIdx.x], uc = src[512 + threadIdx.x];
    float s = 0.f;
    if (KIND == 0) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub), c = __builtin_bit_cast(bf16x8, uc);
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, c, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, a, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c, a3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    } else if (KIND == 1) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub), c = __builtin_bit_cast(bf16x8, uc);
        f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, c, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, a, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, c, a3, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    } else {
        const float a = __uint_as_float(ua.x), b = __uint_as_float(ub.x), c = __uint_as_float(uc.x);
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, c, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, a, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, c, a3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    }
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name, double flop_per_mfma, double nominal_tf, uint4* src, float* out) {
    const int iters = 40000;
    std::vector<uint32_t> h(3 * 256 * 4);
    for (int data = 0; data < 3; ++data)
        for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
            for (auto& v : h) {
                if (KIND == 2) v = data == 0 ? 0x3f800000u : data == 1 ? (0x3f800000u | (rand() & 0x7fffff)) : ((rand() & 0x807fffff) | ((uint32_t)(0x70 + (rand() & 0x1f)) << 23));
                else { auto r = [&]() { return data == 0 ? 0x3f80u : data == 1 ? (0x3f80u | (rand() & 0x7f)) : ((rand() & 0x807f) | ((0x70u + (rand() & 0x1f)) << 7)); }; v = r() | (r() << 16); }
            }
            hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int blocks = 256 * wg_per_cu;
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, src, iters, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, src, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = 4.0 * iters * wg_per_cu;
            const double tf = per_simd * 1024 * flop_per_mfma / (ms * 1e-3) / 1e12;
            printf("%-28s %-34s %d wave(s)/SIMD: %8.3f ms  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s = %4.1f %% of the nominal %.0f\n", name,
                   data == 0 ? "operands 1.0" : data == 1 ? "random mantissas, exponent 0" : "random sign / mantissa / exponent",
                   wg_per_cu, ms, ms * 1e6 / per_simd, tf, 100 * tf / nominal_tf, nominal_tf);
        }
}
int main() {
    uint4* src; float* out;
    if (hipMalloc(&src, 3 * 256 * 16) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
    run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 2500, src, out);
    run<1>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 2500, src, out);
    run<2>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 157.3, src, out);
    return 0;
}
