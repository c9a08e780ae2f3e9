/* vq_chain.c -- CPU oracle for the VQ assignment, TEST INFRASTRUCTURE ONLY (never linked
 * into or called by the product; see oracle/torch_ref.py for the rule).
 *
 * Restates vector_quantizer/vq_img.py:167-168 of the reference,
 *     distance = torch.cdist(x, W, p=2);  idx = argmin(distance, -1)
 * with ATen's CPU formula for cdist when rows > 25 (aten/src/ATen/native/Distance.cpp,
 * _euclidean_dist; torch pinned 1.13.1 in requirements.txt, 2.10 here -- SURVEY 8c):
 *     d = sqrt(clamp_min(|x|^2 + |e|^2 - 2 x.e, 0))
 * in plain fp32 with every dot product an fmaf chain in a FIXED channel order, so that the
 * result is a pure function of the inputs (ATen's sgemm order is BLAS dependent):
 *   order 0 "natural": channels 0,1,2,...; |x|^2 one ascending chain
 *   order 1 "mfma8"  : per 8 channels 8j+{0,4,1,5,2,6,3,7}; |x|^2 = chain over channels with
 *                      (c&7)<4 plus chain over the others -- the order the gfx950 kernel's
 *                      v_mfma_f32_32x32x2_f32 sequence uses, which makes the kernel's distances
 *                      reproducible here bit for bit.
 * Pinned against the reference's own argmin on tests/golden/vq_*.npz (both orders).
 *
 * build: gcc -O2 -fPIC -shared -ffp-contract=off -mfma -fopenmp vq_chain.c -o _build/libvq_chain.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline int chan(int order, int i) {
    if (order == 0) return i;
    const int j = i & 7;                       /* position inside the block of 8 */
    return (i & ~7) + ((j & 1) ? 4 : 0) + (j >> 1);
}

/* x [N,C], W [K,C] -> idx [N] (first minimum), dmin [N] (may be NULL). returns 0 / -1 */
int vq_chain_assign(const float* x, const float* W, int64_t N, int C, int K, int order, int64_t* idx, float* dmin) {
    if (N <= 0 || C <= 0 || K <= 0 || (order == 1 && (C & 3))) return -1;
    const int Cp = (C + 7) & ~7;
    float* WT = (float*)calloc((size_t)Cp * K, sizeof(float));   /* WT[c][k], zero padded channels */
    float* en = (float*)malloc((size_t)K * sizeof(float));
    if (!WT || !en) return -1;
    for (int k = 0; k < K; ++k) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = W[(size_t)k * C + c];
            WT[(size_t)c * K + k] = v;
            s = fmaf(v, v, s);
        }
        en[k] = s;
    }
#pragma omp parallel
    {
        float* acc = (float*)malloc((size_t)K * sizeof(float));
        float* xr = (float*)calloc((size_t)Cp, sizeof(float));
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            for (int c = 0; c < C; ++c) xr[c] = x[(size_t)n * C + c];
            float xn;
            if (order == 0) {
                xn = 0.0f;
                for (int c = 0; c < C; ++c) xn = fmaf(xr[c], xr[c], xn);
            } else {
                float lo = 0.0f, hi = 0.0f;
                for (int c = 0; c < Cp; ++c) {
                    if ((c & 7) < 4) lo = fmaf(xr[c], xr[c], lo);
                    else hi = fmaf(xr[c], xr[c], hi);
                }
                xn = lo + hi;
            }
            for (int k = 0; k < K; ++k) acc[k] = 0.0f;
            for (int i = 0; i < Cp; ++i) {
                const int c = chan(order, i);
                const float a = xr[c];
                const float* w = WT + (size_t)c * K;
                for (int k = 0; k < K; ++k) acc[k] = fmaf(a, w[k], acc[k]);
            }
            float best = INFINITY;
            int64_t bi = 0;
            for (int k = 0; k < K; ++k) {
                float d = fmaf(-2.0f, acc[k], xn);
                d = d + en[k];
                d = d > 0.0f ? d : 0.0f;
                d = sqrtf(d);
                if (d < best) { best = d; bi = k; }
            }
            idx[n] = bi;
            if (dmin) dmin[n] = best;
        }
        free(acc);
        free(xr);
    }
    free(WT);
    free(en);
    return 0;
}
