// The v_fma_mix form of the two-piece fp16 cut (gamer_amd/csrc/common.h: cut2h_quad) against its C form, bit for bit:
//   hipcc -O3 --offload-arch=gfx950 -Iinclude tools/mix_cut_test.hip -o tools/_ab/mix_cut_test && tools/_ab/mix_cut_test
#include "../gamer_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
using gamer::f16x2;
__device__ void ref_pair(float x0, float x1, float s, unsigned& p0, unsigned& p1) {
    x0 *= s; x1 *= s;
    const f16x2 a = {(_Float16)x0, (_Float16)x1};
    const f16x2 b = {(_Float16)(x0 - (float)a[0]), (_Float16)(x1 - (float)a[1])};
    p0 = __builtin_bit_cast(unsigned, a); p1 = __builtin_bit_cast(unsigned, b);
}
__device__ void mix_pair(float x0, float x1, float s, unsigned& h0, unsigned& h1) {
    unsigned b0, b1;
    gamer::cut2h_quad(x0, x1, x1, x0, s, h0, h1, b0, b1);
    if (b0 != ((h0 >> 16) | (h0 << 16)) || b1 != ((h1 >> 16) | (h1 << 16))) h0 = 0xdeadbeefu;      // the second pair = the first, swapped
}
__global__ void k(const float* x, float s, int n, unsigned* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a0, a1, b0, b1;
    ref_pair(x[2 * i], x[2 * i + 1], s, a0, a1);
    mix_pair(x[2 * i], x[2 * i + 1], s, b0, b1);
    out[4 * i] = a0; out[4 * i + 1] = a1; out[4 * i + 2] = b0; out[4 * i + 3] = b1;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    float* x; unsigned* o;
    hipMalloc(&x, n * 4); hipMalloc(&o, n * 8);
    for (float s : {1.f, 8192.f, 1.f / 1024.f, 3.0517578125e-05f, 1.2676506e30f}) {
        for (int i = 0; i < n; ++i) {
            const float m = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, rand() % 40 - 30);
            h[i] = (i % 97 == 0) ? 0.f : m / s * 4096.f;
        }
        h[1] = INFINITY; h[3] = NAN; h[5] = -0.f; h[7] = 65519.f / s; h[9] = 65520.f / s; h[11] = 1e-30f;
        hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, x, s, n, o);
        std::vector<unsigned> r(2 * n);
        hipMemcpy(r.data(), o, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n / 2; ++i)
            if (r[4 * i] != r[4 * i + 2] || r[4 * i + 1] != r[4 * i + 3]) {
                // NaN payloads may differ: compare as NaN-ness
                auto isn = [](unsigned v) { return ((v & 0x7c00) == 0x7c00 && (v & 0x3ff)) || ((v >> 16 & 0x7c00) == 0x7c00 && (v >> 16 & 0x3ff)); };
                if (isn(r[4 * i]) || isn(r[4 * i + 1])) continue;
                if (bad < 5) printf("  mismatch s=%g i=%d x=(%g,%g) ref %08x %08x mix %08x %08x\n", s, i, h[2 * i], h[2 * i + 1], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
                ++bad;
            }
        printf("scale %g: %ld mismatching pairs of %d\n", s, bad, n / 2);
    }
    return 0;
}
