// optim_kernels.hip -- the optimiser step of the training path as ONE launch per network.
//
// Reference: torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999)) created at train_vqreptunet1x1v2.py:106-107 and
// stepped at :200-201 (eps 1e-8, no weight decay, no amsgrad).  The arithmetic below is the single-tensor (non-fused) rule of
// torch/optim/adam.py in fp32, operation by operation:
//     m   = m + (1 - b1) * (g - m)                         exp_avg.lerp_(grad, 1 - beta1)            (one fma)
//     v   = fma((1 - b2) * g, g, v * b2)                   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//                                                          (ATen's CPU addcmul contracts the last product into an fma: measured bit-equal)
//     den = sqrt(v) / sqrt(1 - b2^t) + eps                 (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//     p   = p + (-(lr / (1 - b1^t)) * m) / den             param.addcdiv_(exp_avg, denom, value=-step_size)
// with the step-dependent scalars evaluated by the host in double, as torch does.  Divisions and the square root are the
// correctly rounded IEEE ones; ATen's vectorised CPU sqrt is not (its scalar and vector paths differ from each other by an ulp), so
// parameters agree with a CPU torch.optim.Adam to 1-2 ulp, the moments exactly (tests/test_optim_gpu.py).
//
// What the launch adds to a plain optimiser: the convolution kernels do not read nn.Conv2d.weight, they read bf16 images of it
// (conv_kernels.hip: forward [Cout][k][k][Cin^32], data-gradient [Cin][k][k flipped][Cout^32], split-3 [Cout][k][k][3 Cin]),
// which die with every step.  Rebuilding them cost a launch per layer and 4.9 ms per training step (r3); here the workgroup that
// updates a 32 x TCI-channel tile of a k x k weight keeps the NEW values in LDS and writes the three images in the same pass --
// the fp32 master is read once per step, not twice.
//
// Work decomposition: the host builds (once per parameter set) a table of work items (parameter, tile); plain parameters are cut
// into flat chunks of ADAM_CHUNK elements, k x k convolution weights into 32 (Cout) x TCI (Cin) tiles (TCI = 32 for k = 3, 128 for
// k = 1: contiguous runs of TCI * k * k floats per output channel).  HBM-bound: 28 B of fp32 state + <= 10 B of images per element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vqseg.h"

extern "C" int vqseg_set_error(int code, const char* msg);   // vqseg_abi.hip

namespace vqseg {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ADAM_CHUNK = VQSEG_ADAM_CHUNK;
constexpr int TILE_FLOATS = 32 * (32 * 9 + 1);              // LDS tile: 32 output channels x (run + 1 pad)

struct AdamScalars {
    float w1;        // 1 - beta1
    float b2;        // beta2
    float a2;        // 1 - beta2
    float bc2_sqrt;  // sqrt(1 - beta2^t)
    float eps;
    float neg_step;  // -(lr / (1 - beta1^t))
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& s) {
    m = __fmaf_rn(s.w1, __fsub_rn(g, m), m);
    v = __fmaf_rn(__fmul_rn(s.a2, g), g, __fmul_rn(v, s.b2));
    const float den = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), s.bc2_sqrt), s.eps);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(s.neg_step, m), den));
}

__device__ __forceinline__ unsigned int pack2_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned int)__builtin_bit_cast(unsigned short, x) | ((unsigned int)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_round_f(float a) { return (float)((__bf16)a); }

__device__ void adam_flat_chunk(const VqsegAdamParam& P, long chunk, const AdamScalars& s) {
    const long base = chunk * ADAM_CHUNK;
    long end = base + ADAM_CHUNK;
    if (end > P.numel) end = P.numel;
    const bool vec = (((uintptr_t)P.p | (uintptr_t)P.g | (uintptr_t)P.m | (uintptr_t)P.v) & 15u) == 0;
    if (vec) {
        const long end4 = base + ((end - base) & ~3L);
        for (long i = base + 4 * threadIdx.x; i < end4; i += 4 * 256) {
            f32x4 p = *reinterpret_cast<const f32x4*>(P.p + i), g = *reinterpret_cast<const f32x4*>(P.g + i);
            f32x4 m = *reinterpret_cast<const f32x4*>(P.m + i), v = *reinterpret_cast<const f32x4*>(P.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_one(pe, g[e], me, ve, s);
                p[e] = pe, m[e] = me, v[e] = ve;
            }
            *reinterpret_cast<f32x4*>(P.p + i) = p;
            *reinterpret_cast<f32x4*>(P.m + i) = m;
            *reinterpret_cast<f32x4*>(P.v + i) = v;
        }
        for (long i = end4 + threadIdx.x; i < end; i += 256) {
            float p = P.p[i], m = P.m[i], v = P.v[i];
            adam_one(p, P.g[i], m, v, s);
            P.p[i] = p, P.m[i] = m, P.v[i] = v;
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float p = P.p[i], m = P.m[i], v = P.v[i];
            adam_one(p, P.g[i], m, v, s);
            P.p[i] = p, P.m[i] = m, P.v[i] = v;
        }
    }
}

// One 32 (Cout) x TCI (Cin) x K x K tile of a convolution weight: update, then the images (layouts: conv_pack_all_kernel).
template <int K, int TCI>
__device__ void adam_conv_tile(const VqsegAdamParam& P, int tile_idx, const AdamScalars& s, float* tile) {
    constexpr int KK = K * K;
    constexpr int run = TCI * KK;                            // floats per output channel of the tile (contiguous in the weight)
    constexpr int LD = run + 1;
    const int Cout = P.cout, Cin = P.cin;
    const int tiles_ci = (Cin + TCI - 1) / TCI;
    const int cob = tile_idx / tiles_ci;
    const int ci0 = (tile_idx - cob * tiles_ci) * TCI, co0 = cob * 32;
    const int Cin_p = (Cin + 31) / 32 * 32, Cout_p = (Cout + 31) / 32 * 32;
    const int run_ok = (Cin - ci0 < TCI ? Cin - ci0 : TCI) * KK;      // valid floats of a run
#pragma unroll 4
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int cr = i / run, e = i - cr * run;           // e = ci_local * KK + tap
        const int co = co0 + cr;
        float pn = 0.0f;
        if (co < Cout && e < run_ok) {
            const long idx = ((long)co * Cin + ci0) * KK + e;
            float m = P.m[idx], v = P.v[idx];
            pn = P.p[idx];
            adam_one(pn, P.g[idx], m, v, s);
            P.p[idx] = pn, P.m[idx] = m, P.v[idx] = v;
        }
        tile[cr * LD + e] = pn;
    }
    if (!P.fwd && !P.tr && !P.s3) return;
    __syncthreads();
    unsigned short* fwd = static_cast<unsigned short*>(P.fwd);
    unsigned short* tr = static_cast<unsigned short*>(P.tr);
    unsigned short* s3 = static_cast<unsigned short*>(P.s3);
    const int C1 = P.c1;
    // forward / split-3 images: (co, tap) rows, 8 ci = one 16-byte store
    if (fwd || s3)
        for (int i = threadIdx.x; i < 32 * KK * (TCI / 8); i += 256) {
            const int c8 = i % (TCI / 8), rt = i / (TCI / 8);
            const int cr = rt / KK, tap = rt - cr * KK;
            const int co = co0 + cr, ci = ci0 + c8 * 8;
            if (co >= Cout || ci >= Cin_p) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[cr * LD + (c8 * 8 + e) * KK + tap];
            u32x4 h4, l4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h4[e] = pack2_bf16(v[2 * e], v[2 * e + 1]);
                l4[e] = pack2_bf16(v[2 * e] - bf16_round_f(v[2 * e]), v[2 * e + 1] - bf16_round_f(v[2 * e + 1]));
            }
            if (fwd) *reinterpret_cast<u32x4*>(fwd + ((long)co * KK + tap) * Cin_p + ci) = h4;
            if (s3 && ci < Cin) {                           // Cin % 32 == 0 (checked by the host): whole chunks
                const bool second = ci >= C1;
                const int cs = second ? Cin - C1 : C1, cloc = second ? ci - C1 : ci;
                unsigned short* row = s3 + ((long)co * KK + tap) * 3 * Cin + (second ? 3 * C1 : 0);
                *reinterpret_cast<u32x4*>(row + cloc) = h4;
                *reinterpret_cast<u32x4*>(row + cs + cloc) = h4;
                *reinterpret_cast<u32x4*>(row + 2 * cs + cloc) = l4;
            }
        }
    // data-gradient image: (ci, flipped tap) rows, 32 co contiguous
    if (tr)
        for (int i = threadIdx.x; i < TCI * KK * 4; i += 256) {
            const int c8 = i & 3, rt = i >> 2;
            const int cl = rt / KK, tap = rt - cl * KK;
            const int ci = ci0 + cl, co = co0 + c8 * 8;
            if (ci >= Cin || co >= Cout_p) continue;
            u32x4 h4;
#pragma unroll
            for (int e = 0; e < 4; ++e)                     // rows co >= Cout of the tile are zero
                h4[e] = pack2_bf16(tile[(c8 * 8 + 2 * e) * LD + cl * KK + tap], tile[(c8 * 8 + 2 * e + 1) * LD + cl * KK + tap]);
            *reinterpret_cast<u32x4*>(tr + ((long)ci * KK + (KK - 1 - tap)) * Cout_p + co) = h4;
        }
}

__global__ __launch_bounds__(256) void adam_step_kernel(const VqsegAdamParam* __restrict__ params, const int32_t* __restrict__ items,
                                                        const AdamScalars s) {
    __shared__ float tile[TILE_FLOATS];
    const int pi = items[2 * blockIdx.x], ti = items[2 * blockIdx.x + 1];
    const VqsegAdamParam P = params[pi];
    if (P.k == 3) adam_conv_tile<3, 32>(P, ti, s, tile);
    else if (P.k == 1) adam_conv_tile<1, 128>(P, ti, s, tile);
    else adam_flat_chunk(P, ti, s);
}

}  // namespace vqseg

extern "C" {

int64_t vqseg_adam_work_items(int64_t numel, int k, int cout, int cin) {
    if (numel <= 0) return 0;
    if (k == 3) return (int64_t)((cout + 31) / 32) * ((cin + 31) / 32);
    if (k == 1) return (int64_t)((cout + 31) / 32) * ((cin + 127) / 128);
    return (numel + vqseg::ADAM_CHUNK - 1) / vqseg::ADAM_CHUNK;
}

int vqseg_adam_step_f32(const VqsegAdamParam* params_dev, const int32_t* items_dev, int n_items, double lr, double beta1, double beta2,
                        double eps, int64_t step, void* stream) {
    if (!params_dev || !items_dev || n_items <= 0) return vqseg_set_error(VQSEG_EINVAL, "adam_step: null table or no work items");
    if (step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return vqseg_set_error(VQSEG_EINVAL, "adam_step: step >= 1, 0 <= beta < 1, eps >= 0 required");
    // torch/optim/adam.py (_single_tensor_adam): python floats = doubles, applied to fp32 tensors as fp32 scalars
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    vqseg::AdamScalars s;
    s.w1 = (float)(1.0 - beta1);
    s.b2 = (float)beta2;
    s.a2 = (float)(1.0 - beta2);
    s.bc2_sqrt = (float)sqrt(bc2);
    s.eps = (float)eps;
    s.neg_step = (float)(-(lr / bc1));
    hipLaunchKernelGGL(vqseg::adam_step_kernel, dim3((unsigned)n_items), dim3(256), 0, static_cast<hipStream_t>(stream), params_dev,
                       items_dev, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[160];
        snprintf(buf, sizeof(buf), "adam_step_kernel: %s", hipGetErrorString(e));
        return vqseg_set_error((int)e, buf);
    }
    return 0;
}

}  // extern "C"
