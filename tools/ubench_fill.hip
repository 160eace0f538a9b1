// Micro-benchmark: how many independent vector instructions hide in the gap of a 16-bit MFMA inside ONE wave (one wave per SIMD),
// and what two such waves per SIMD do.  hipcc -O3 --offload-arch=gfx950 tools/ubench_fill.hip -o tools/ubench_fill.bin
// Per MFMA (v_mfma_f32_32x32x16_f16, four independent accumulators) N v_fma_f32 (eight independent registers) or N v_exp_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int N, bool EXP, bool DEP>
__global__ void __launch_bounds__(512) fill(int iters, float* out) {
    const float x = (float)(threadIdx.x & 7) * 0.25f + 1.f;
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)x;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x + i;
    int vi = 0;
    auto fillers = [&]() {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[vi & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[vi & 7]) : "v"(x));
            ++vi;
        }
    };
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a0, 0, 0, 0); fillers();
        if (DEP) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a0, 0, 0, 0); else a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a1, 0, 0, 0);
        fillers();
        if (DEP) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a0, 0, 0, 0); else a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a2, 0, 0, 0);
        fillers();
        if (DEP) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a0, 0, 0, 0); else a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, a3, 0, 0, 0);
        fillers();
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int N, bool EXP, bool DEP>
static void run(int threads, const char* what) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fill<N, EXP, DEP><<<256, threads>>>(100, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fill<N, EXP, DEP><<<256, threads>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d  N=%2d: %.2f ns per MFMA per wave\n", what, threads / 256, N, ms * 1e6 / (iters * 4.0));
    hipFree(out);
}

int main() {
    run<0, false, false>(256, "mfma only"); run<0, false, true>(256, "mfma only, ONE accumulator");
    run<2, false, false>(256, "v_fma fillers"); run<4, false, false>(256, "v_fma fillers"); run<5, false, false>(256, "v_fma fillers");
    run<6, false, false>(256, "v_fma fillers"); run<8, false, false>(256, "v_fma fillers"); run<10, false, false>(256, "v_fma fillers");
    run<12, false, false>(256, "v_fma fillers");
    run<8, false, true>(256, "v_fma fillers, ONE acc");
    run<1, true, false>(256, "v_exp fillers"); run<2, true, false>(256, "v_exp fillers"); run<4, true, false>(256, "v_exp fillers");
    run<0, false, false>(512, "mfma only"); run<4, false, false>(512, "v_fma fillers"); run<8, false, false>(512, "v_fma fillers");
    run<10, false, false>(512, "v_fma fillers");
    return 0;
}
