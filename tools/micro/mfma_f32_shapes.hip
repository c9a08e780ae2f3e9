// Bare f32 MFMA loops on random register operands: does the chip hold a different clock for 16x16x4 than for 32x32x2?
// (MI355X_MICROARCH.md "DVFS give-back" (7) reports 1.15x for the bf16 16x16x32 vs 32x32x16 loops.)   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void loop(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = __sinf(seed * (lane + 1) * (i + 3)) * 0.5f;
    for (int i = 0; i < 8; ++i) b[i] = __cosf(seed * (lane + 7) * (i + 1)) * 0.5f;
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[t], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    } else {
        f32x4 acc[32];
        for (int t = 0; t < 32; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {          // same flops per iteration: 32 MFMAs of 2*32*32*2 = 64 of 2*16*16*4
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int t = 0; t < 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e + (t & 1) * 2], b[t & 7], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 32; ++t) for (int i = 0; i < 4; ++i) s += acc[t][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 512;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 3; ++w) { if (shape == 32) loop<32><<<blocks, 256>>>(out, iters, 0.37f); else loop<16><<<blocks, 256>>>(out, iters, 0.37f); }
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) { if (shape == 32) loop<32><<<blocks, 256>>>(out, iters, 0.37f); else loop<16><<<blocks, 256>>>(out, iters, 0.37f); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
            printf("shape %dx%d: %.3f ms  %.1f TF/s\n", shape, shape, ms, flops / ms / 1e9);
        }
    return 0;
}
