// How does v_mfma_f32_32x32x16_bf16 round?  One wave: D = C + sum_k A[m][k] * B[k][n], B = ones, so D[m][*] = C[m] + sum_k A[m][k].
// Row m of A holds a crafted pattern; results are compared with (i) the exactly rounded sum (long double), (ii) a sequential fp32
// chain C, +p0, +p1, ... in k order.  hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_bf16_rounding.hip -o /tmp/mfma_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void one_mfma(const float* A /*[32][16] exact-in-bf16 values*/, const float* Bv /*[16][32]*/, const float* C /*[32]*/, float* D /*[32][32]*/) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[r * 16 + 8 * h + j];
        b[j] = (__bf16)Bv[(8 * h + j) * 32 + r];
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[(i & 3) + 8 * (i >> 2) + 4 * h];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static float bf(float v) { unsigned u; memcpy(&u, &v, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; memcpy(&v, &u, 4); return v; }

int main() {
    float A[32 * 16] = {0}, B[16 * 32], C[32] = {0}, D[32 * 32];
    for (int i = 0; i < 16 * 32; ++i) B[i] = 1.0f;
    const float T24 = 16777216.0f;
    // row 0: [2^24, 1 x14, -2^24]  exact 14; sequential RNE 0
    A[0] = T24; for (int k = 1; k < 15; ++k) A[k] = 1.0f; A[15] = -T24;
    // row 1: C = 2^24, 16 x 1  -> exact 2^24+16
    C[1] = T24; for (int k = 0; k < 16; ++k) A[16 + k] = 1.0f;
    // row 2: C = 2^24, 16 x 0.75 -> exact 2^24+12
    C[2] = T24; for (int k = 0; k < 16; ++k) A[32 + k] = 0.75f;
    // row 3: C = 2^24, products [1 x 8 in k = 0..7 (h = 0), 1 x 8 in k = 8..15 (h = 1)] with alternating signs of 2^23 inside
    C[3] = 1.0f; A[48] = T24; A[49] = 0.5f; A[50] = 0.5f; A[51] = 0.5f; A[52] = 0.5f; A[56] = -T24; A[57] = 0.25f;
    // row 4: C = 1, products 2^-30 x 16 (tiny addends vs C): exact 1 + 2^-26 -> rounds to 1; (sanity)
    C[4] = 1.0f; for (int k = 0; k < 16; ++k) A[64 + k] = ldexpf(1.0f, -30);
    // row 5: C = 1, products 2^-25 x 16: exact 1 + 2^-21 (representable: ulp(1) = 2^-23); sequential RNE: 1 + 2^-25 -> 1 (tie to even) each time -> 1
    C[5] = 1.0f; for (int k = 0; k < 16; ++k) A[80 + k] = ldexpf(1.0f, -25);
    // row 6: C = 0, products 2^0, 2^-24 x 15: exact 1 + 15 * 2^-24 -> rounds to 1 + 2^-20 - ... (1 + 0.9375 * 2^-20)
    A[96] = 1.0f; for (int k = 1; k < 16; ++k) A[96 + k] = ldexpf(1.0f, -24);
    // rows 8..31: random bf16 values, random C
    srand(7);
    for (int m = 8; m < 32; ++m) {
        C[m] = (float)(rand() % 2001 - 1000) * ldexpf(1.0f, (rand() % 24) - 12);
        for (int k = 0; k < 16; ++k) A[m * 16 + k] = bf((float)(rand() % 2001 - 1000) * ldexpf(1.0f, (rand() % 24) - 16));
    }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, sizeof(C)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice); hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
    one_mfma<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int m = 0; m < 32; ++m) {
        long double ex = C[m], mag = fabsl((long double)C[m]);
        float seq = C[m];
        for (int k = 0; k < 16; ++k) { ex += (long double)A[m * 16 + k]; mag += fabsl((long double)A[m * 16 + k]); seq += A[m * 16 + k]; }
        const float exr = (float)ex;
        const double err = fabs((double)D[m * 32] - (double)ex) / (double)(mag > 0 ? mag : 1);
        if (m >= 8 && err > worst) worst = err;
        printf("row %2d: mfma %.9g  exact %.9Lg (rounded %.9g)  sequential-fp32 %.9g  |err|/magnitude-sum %.3g (2^%.1f)%s\n", m, D[m * 32], ex, exr, seq, err,
               err > 0 ? log2(err) : -99.0, D[m * 32] == exr ? "  == exactly rounded" : "");
    }
    printf("random rows: worst |err| / magnitude sum = %.3g = 2^%.2f\n", worst, worst > 0 ? log2(worst) : -99.0);
    return 0;
}
