// conv_kernels.hip -- implicit-GEMM convolution on the bf16 matrix cores of gfx950 (MI355X).
//
// Reference ops replaced: nn.Conv2d inside conv_bn_relu (models/networks/unet/decoder.py:7-10) and inside
// the ResNet bottlenecks (models/encoders/resnet.py:117-190 on torchvision's Bottleneck): 3x3 / 1x1 / 7x7,
// stride 1 / 2, zero or reflect padding, no bias.  One kernel covers the forward convolution, the data
// gradient (same kernel on the output gradient with tap-flipped, channel-transposed weights and an
// up-sampled ("dilated") input grid for stride-2 layers) and -- through `x2` -- the channel concatenation
// of the decoder (decoder.py:35-37), so torch.cat never materialises.
//
// GEMM view:  Y[m, co] = sum_{tap, ci} A[m, (tap, ci)] * Wp[co, (tap, ci)],   m = (n, oh, ow) pixel rows, NHWC.
//   workgroup = 4 waves = 128 pixel rows x BN output channels, wave = 64 x (BN/2) via v_mfma_f32_32x32x16_bf16
//   K loop: taps outer, channel chunks of BK inner; A rows are gathered (zero / reflect / dilated), staged
//   global -> registers -> LDS (rows padded by 16 B => conflict-free ds_read_b128), weights likewise.
// Precision modes (template PRECISE):
//   fast    : activations and weights bf16, fp32 accumulate                       (BK = 64)
//   precise : activations fp32 in HBM, split on the fly into bf16 hi + lo; weights pre-split;
//             acc += a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  (3 MFMAs)  ~ 2^-16 relative per product (BK = 32).
//             This is the parity mode (logits within 1e-3 of the fp32 CPU reference) at 3/16 of the cost
//             of the fp32 MFMA.
// Optional epilogue: per-wave partial BatchNorm statistics (count, mean, M2 over the wave's 64 rows) for a
// deterministic two-level Welford merge (no float atomics).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>

#include "conv_kernels.h"

namespace vqseg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128;

__device__ __forceinline__ unsigned int pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned int)__builtin_bit_cast(unsigned short, x) | ((unsigned int)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ void unpack8(const u32x4 a, float (&v)[8]) {      // 8 bf16 -> fp32
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __builtin_bit_cast(float, a[e] << 16);
        v[2 * e + 1] = __builtin_bit_cast(float, a[e] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ float bf16_round(float a) { return (float)((__bf16)a); }

// pixel row m -> (image n, offset inside the image): 32-bit division whenever the row count allows (a 64-bit
// division is ~100 emulated instructions, which matters in layers whose whole K loop is one or two stages)
__device__ __forceinline__ void split_row(long m, int hw, long M, int& n, int& rem) {
    if (M <= 0x7fffffffL) {
        const unsigned int um = (unsigned int)m;
        n = (int)(um / (unsigned int)hw);
        rem = (int)(um - (unsigned int)n * (unsigned int)hw);
    } else {
        n = (int)(m / hw);
        rem = (int)(m - (long)n * hw);
    }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// Which source tensor holds contraction channels [ci, ci + chunk): the concat fusion ([0, C1) -> x, [C1, Cin) -> x2), and for
// split-3 INPUTS (p.s3_in) the fold of the logical [hi | lo | hi] segment (3 Cs channels) onto its stored [hi | lo] rows (2 Cs
// channels per pixel): the third part re-reads hi (from L2 -- the bytes were fetched a few stages ago).
struct ASource {
    const char* src;
    int csrc, cbase;                                        // channels per pixel row of the source, channel offset inside the row
};
__device__ __forceinline__ ASource a_source(const ConvArgs& p, int ci) {
    const bool second = ci >= p.C1;
    ASource a;
    a.src = reinterpret_cast<const char*>(second ? p.x2 : p.x);
    a.csrc = second ? (p.Cin - p.C1) : p.C1;
    a.cbase = second ? (ci - p.C1) : ci;
    if (p.s3_in) {
        const int cs2 = 2 * (second ? p.s3_cs2 : p.s3_cs1);
        a.csrc = cs2;
        if (a.cbase >= cs2) a.cbase -= cs2;
    }
    return a;
}

// Strided output placement (p.omap): GEMM row m = (n, oh, ow) of the Ho x Wo problem lands on pixel (n, 2 oh + ph, 2 ow + pw) of
// an OH x OW grid -- the parity classes of a stride-2 data gradient (launch_dgrad_s2).  Two 32-bit divisions per written row.
__device__ __forceinline__ long out_row(const ConvArgs& p, long m) {
    if (!p.omap) return m;
    const unsigned hw = (unsigned)(p.Ho * p.Wo), um = (unsigned)m;
    const unsigned n = um / hw, rem = um - n * hw;
    const unsigned oh = rem / (unsigned)p.Wo, ow = rem - oh * (unsigned)p.Wo;
    if (p.omap_fold) {                                       // (i, j) on the padded grid -> the unpadded gradient / ring / dump row
        const int i = 2 * (int)oh + p.omap_ph, j = 2 * (int)ow + p.omap_pw, H = p.omap_h, W = p.omap_w;
        const long body = (long)p.N * H * W;
        if (i >= 1 && j >= 1 && i <= H && j <= W) return ((long)n * H + i - 1) * W + j - 1;
        if (p.omap_fold == 1) {
            const int rl = W + 1 + H;
            if (i == 0 && j <= W) return body + (long)n * rl + j;
            if (j == 0 && i <= H) return body + (long)n * rl + W + i;          // i >= 1 here
            return body + (long)p.N * rl;
        }
        return body;
    }
    return ((long)n * p.omap_h + 2 * oh + p.omap_ph) * p.omap_w + 2 * ow + p.omap_pw;
}

#ifndef GLDS_ABL
#define GLDS_ABL 0                // debug builds only (results wrong): 1 no y stores, 2 no epilogue, 4 no BN partials, 8 no A-tile DMA
#endif
// ---- shared epilogue: per-wave BN partials from the accumulators, then Y through LDS as 16-byte row segments
struct LinearRows {                                          // tile row -> output pixel row (NHWC-flattened)
    long m0;
    __device__ __forceinline__ long operator()(int row) const { return m0 + row; }
};

// FAST: compile the whole-tile fast paths (the register-staged conv_igemm_kernel opts out: they cost it 24 VGPRs = half its occupancy)
template <int TBM, int BN, bool PRECISE, int MT, int NTT, int NT, int NTHR, typename RowMap = LinearRows, bool S3 = false, bool FAST = true>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MT][NTT], const ConvArgs& p, char* smem, long M, long m0, int co0,
                                              int wm, int wn, int r, int h, int tid, RowMap row_to_m = LinearRows{-1}) {
    if constexpr (__is_same(RowMap, LinearRows)) row_to_m.m0 = m0;
    const long wrow0 = m0 + (long)wm * MT * 32;
    if constexpr (S3) {
        // split-3 output: the fp32 result v = acc * scale + shift (+ residual, ReLU) leaves as [hi | lo] bf16, hi = bf16(v),
        // lo = bf16(v - hi): 2 * Cout channels per pixel row (the consumer's K loop reads hi twice: a_source).  Staged through LDS as two bf16 tiles so that the stores (and the
        // residual loads) are whole 16-byte row segments.  Host side guarantees Cout % 8 == 0 and a fused epilogue.
        static_assert(!PRECISE, "split-3 output belongs to the bf16 kernels");
        // (r4: a tile whose two staging tiles do not fit LDS -- the 256 x 256 tile: 270 KB -- leaves in two column halves)
        constexpr int HALVES = ((size_t)TBM * (BN + 8) * 4 > 160u * 1024u) ? 2 : 1;
        constexpr int BNH = BN / HALVES, OS = BNH + 8;
        __bf16* th = reinterpret_cast<__bf16*>(smem);
        __bf16* tl = th + TBM * OS;
        const bool ep_res = p.ep_res != nullptr;
        constexpr int CPR = BNH / 8;
        const long rs = 2L * p.Cout;                         // output (and residual) row stride in elements: [hi | lo]
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf) {
            if (hf) __syncthreads();                         // the previous half has been read out of LDS
#pragma unroll
            for (int b = 0; b < NTT; ++b) {
                const int col = (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
                if (HALVES > 1 && ((BN >= 64 ? (wn * NT + b) * 32 : 0) / BNH) != hf) continue;      // wave-uniform
                const int lc = col - hf * BNH;
                const bool cok = co0 + col < p.Cout;
                const float esc = cok ? p.ep_scale[co0 + col] : 1.0f, esh = cok ? p.ep_shift[co0 + col] : 0.0f;
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = (wm * MT + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        float v = __builtin_fmaf(acc[a][b][i], esc, esh);
                        if (p.ep_relu && !ep_res && !(v > 0.0f)) v = 0.0f;
                        const __bf16 vh = (__bf16)v;
                        th[row * OS + lc] = vh;
                        tl[row * OS + lc] = (__bf16)(v - (float)vh);
                    }
            }
            __syncthreads();
            for (int idx = tid; idx < TBM * CPR; idx += NTHR) {
                const int row = idx / CPR, ch = idx % CPR;
                const long m = row_to_m(row);
                const int co = co0 + hf * BNH + ch * 8;
                if (m < M && co < p.Cout) {
                    u32x4 vh = *reinterpret_cast<const u32x4*>(th + (size_t)row * OS + ch * 8);
                    u32x4 vl = *reinterpret_cast<const u32x4*>(tl + (size_t)row * OS + ch * 8);
                    if (ep_res) {
                        const unsigned short* rp = reinterpret_cast<const unsigned short*>(p.ep_res) + m * rs + co;
                        const u32x4 rh = *reinterpret_cast<const u32x4*>(rp);
                        const u32x4 rl = *reinterpret_cast<const u32x4*>(rp + p.Cout);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v0 = (__builtin_bit_cast(float, vh[e] << 16) + __builtin_bit_cast(float, vl[e] << 16)) +
                                       (__builtin_bit_cast(float, rh[e] << 16) + __builtin_bit_cast(float, rl[e] << 16));
                            float v1 = (__builtin_bit_cast(float, vh[e] & 0xFFFF0000u) + __builtin_bit_cast(float, vl[e] & 0xFFFF0000u)) +
                                       (__builtin_bit_cast(float, rh[e] & 0xFFFF0000u) + __builtin_bit_cast(float, rl[e] & 0xFFFF0000u));
                            if (p.ep_relu && !(v0 > 0.0f)) v0 = 0.0f;
                            if (p.ep_relu && !(v1 > 0.0f)) v1 = 0.0f;
                            vh[e] = pack2(v0, v1);
                            vl[e] = pack2(v0 - bf16_round(v0), v1 - bf16_round(v1));
                        }
                    }
                    unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + m * rs + co;
                    *reinterpret_cast<u32x4*>(yp) = vh;
                    *reinterpret_cast<u32x4*>(yp + p.Cout) = vl;
                }
            }
        }
        return;
    }
    // Whole tile inside the output (always for the 2-D pixel tiles of the patch kernel; every tile but the last one otherwise): the
    // per-element row tests (a 64-bit compare + select each, twice per accumulator in the statistics, once per store) drop out.
    // These epilogues are VALU-issue bound on the short-K layers (see the LIN note at conv_igemm_glds_kernel).
    const bool full = FAST && (!__is_same(RowMap, LinearRows) || m0 + TBM <= M);          // workgroup-uniform
    if (p.stat_partial && !(GLDS_ABL & 4)) {
        // one (mean, M2) partial per SLOT of RPS consecutive rows: 64 rows (two 32-row tiles) or 32 when the wave has one
        constexpr int TPS = (MT >= 2 && BN >= 64) ? 2 : 1, RPS = TPS * 32;   // vqseg_conv_stat_slots: 64 rows per slot from 64 output channels on, else 32
#pragma unroll
        for (int b = 0; b < NTT; ++b) {
            const int co = co0 + (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
            const bool cok = co < p.Cout;
#pragma unroll
            for (int g = 0; g < MT / TPS; ++g) {
                const long srow0 = wrow0 + g * RPS;
                float sum = 0.0f, mean, m2 = 0.0f;
                if (full) {
#pragma unroll
                    for (int a = g * TPS; a < (g + 1) * TPS; ++a)
#pragma unroll
                        for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
                    sum += __shfl_xor(sum, 32);
                    mean = sum / (float)RPS;
#pragma unroll
                    for (int a = g * TPS; a < (g + 1) * TPS; ++a)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float d = acc[a][b][i] - mean;
                            m2 = __builtin_fmaf(d, d, m2);
                        }
                } else {
#pragma unroll
                    for (int a = g * TPS; a < (g + 1) * TPS; ++a)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const long m = wrow0 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            if (m < M) sum += acc[a][b][i];
                        }
                    long cnt_l = M - srow0;
                    const float cnt = (float)(cnt_l < 0 ? 0 : (cnt_l > RPS ? RPS : cnt_l));
                    sum += __shfl_xor(sum, 32);
                    mean = cnt > 0.f ? sum / cnt : 0.f;
#pragma unroll
                    for (int a = g * TPS; a < (g + 1) * TPS; ++a)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const long m = wrow0 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            const float d = acc[a][b][i] - mean;
                            if (m < M) m2 = __builtin_fmaf(d, d, m2);
                        }
                }
                m2 += __shfl_xor(m2, 32);
                if (h == 0 && cok) {
                    const long slot = srow0 / RPS;                        // global slot index along M
                    p.stat_partial[(slot * 2 + 0) * p.Cout + co] = mean;
                    p.stat_partial[(slot * 2 + 1) * p.Cout + co] = m2;
                }
            }
        }
    }
    constexpr int O_EPC = PRECISE ? 4 : 8;                  // output elements per 16-byte store
    if (p.Cout % O_EPC == 0) {
        // the main loop's last barrier has passed: LDS is free.  Tile [BM][BN] of the output element type.
        constexpr int OS = BN + O_EPC;                      // row stride (elements) -- padded against bank conflicts
        char* ot = smem;
        const bool ep = p.ep_scale != nullptr, ep_res = ep && p.ep_res != nullptr;
#pragma unroll
        for (int b = 0; b < NTT; ++b) {
            const int col = (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
            const bool cok = co0 + col < p.Cout;
            const float esc = ep && cok ? p.ep_scale[co0 + col] : 1.0f, esh = ep && cok ? p.ep_shift[co0 + col] : 0.0f;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (wm * MT + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    float v = acc[a][b][i];
                    if (ep) {
                        v = __builtin_fmaf(v, esc, esh);
                        if (p.ep_relu && !ep_res && !(v > 0.0f)) v = 0.0f;
                    }
                    if (PRECISE) reinterpret_cast<float*>(ot)[row * OS + col] = v;
                    else reinterpret_cast<__bf16*>(ot)[row * OS + col] = (__bf16)v;
                }
        }
        __syncthreads();
        constexpr int CPR = BN / O_EPC;                     // 16-byte chunks per tile row
        if constexpr (FAST && __is_same(RowMap, LinearRows) && NTHR % CPR == 0 && (TBM * CPR) % NTHR == 0) {
            if (full && !ep_res && !p.omap) {
                // whole tile, plain rows: one base address per thread, then constant strides (the generic loop below spends ~25 VALU
                // instructions per 16-byte store on row tests and 64-bit address arithmetic)
                const int ch = tid % CPR, row0 = tid / CPR;
                if (co0 + ch * O_EPC < p.Cout) {
                    constexpr int ESZ = PRECISE ? 4 : 2;
                    char* gp = reinterpret_cast<char*>(p.y) + ((m0 + row0) * (long)p.Cout + co0 + ch * O_EPC) * ESZ;
                    const long gstep = (long)(NTHR / CPR) * p.Cout * ESZ;
                    const char* lp = ot + ((size_t)row0 * OS + ch * O_EPC) * ESZ;
#pragma unroll
                    for (int it = 0; it < TBM * CPR / NTHR; ++it) {
#if GLDS_ABL & 1
                        if (it == 12345)
#endif
                        *reinterpret_cast<u32x4*>(gp) = *reinterpret_cast<const u32x4*>(lp + (size_t)it * (NTHR / CPR) * OS * ESZ);
                        gp += gstep;
                    }
                }
                return;
            }
        }
        for (int idx = tid; idx < TBM * CPR; idx += NTHR) {
            const int row = idx / CPR, ch = idx % CPR;
            const long m = row_to_m(row);
            const int co = co0 + ch * O_EPC;
            if (m < M && co < p.Cout) {
                u32x4 v = *reinterpret_cast<const u32x4*>(ot + ((size_t)row * OS + ch * O_EPC) * (PRECISE ? 4 : 2));
                const long mo = out_row(p, m);
                if (ep_res) {                               // residual add (+ ReLU) on the way out
                    const long mr = p.ep_res_out ? mo : m;
                    u32x4 rv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.ep_res) + (mr * p.Cout + co) * (PRECISE ? 4 : 2));
                    if (!PRECISE && p.ep_res_bits) {            // masked shortcut gradient: one byte of mask bits per 16-byte chunk
                        const unsigned mb = p.ep_res_bits[(mr * p.Cout + co) >> 3];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            rv[e] &= ((mb >> (2 * e)) & 1u ? 0x0000FFFFu : 0u) | ((mb >> (2 * e + 1)) & 1u ? 0xFFFF0000u : 0u);
                    }
                    if (PRECISE) {
                        f32x4 a4 = __builtin_bit_cast(f32x4, v);
                        const f32x4 r4 = __builtin_bit_cast(f32x4, rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a4[e] += r4[e];
                            if (p.ep_relu && !(a4[e] > 0.0f)) a4[e] = 0.0f;
                        }
                        v = __builtin_bit_cast(u32x4, a4);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo = __builtin_bit_cast(float, v[e] << 16) + __builtin_bit_cast(float, rv[e] << 16);
                            float hi = __builtin_bit_cast(float, v[e] & 0xFFFF0000u) + __builtin_bit_cast(float, rv[e] & 0xFFFF0000u);
                            if (p.ep_relu && !(lo > 0.0f)) lo = 0.0f;
                            if (p.ep_relu && !(hi > 0.0f)) hi = 0.0f;
                            v[e] = pack2(lo, hi);
                        }
                    }
                }
#if GLDS_ABL & 1
                if (v[0] == 0x12345678u)
#endif
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.y) + (mo * p.Cout + co) * (PRECISE ? 4 : 2)) = v;
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < NTT; ++b) {
            const int co = co0 + (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long mv = wrow0 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const long m = row_to_m((int)(mv - m0));
                    if (mv < M && co < p.Cout) {
                        float v = acc[a][b][i];
                        const long mo = out_row(p, m);
                        if (p.ep_scale) {
                            v = __builtin_fmaf(v, p.ep_scale[co], p.ep_shift[co]);
                            if (p.ep_res) {
                                const long mr = p.ep_res_out ? mo : m;
                                if (!PRECISE) v = (float)((__bf16)v);       // same rounding points as the staged path
                                v += PRECISE ? reinterpret_cast<const float*>(p.ep_res)[mr * p.Cout + co]
                                             : (float)reinterpret_cast<const __bf16*>(p.ep_res)[mr * p.Cout + co];
                            }
                            if (p.ep_relu && !(v > 0.0f)) v = 0.0f;
                        }
                        if (PRECISE) reinterpret_cast<float*>(p.y)[mo * p.Cout + co] = v;
                        else reinterpret_cast<__bf16*>(p.y)[mo * p.Cout + co] = (__bf16)v;
                    }
                }
        }
    }
}

template <int BN, bool PRECISE, int BK, bool S3 = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int BKP = BK + 8;                          // bf16 elements per LDS row (16-byte pad)
    constexpr int NT = BN / 64;                          // 32-wide N tiles per wave (BN=128 -> 2; 64 -> 1)
    constexpr int WN = (BN >= 64) ? 2 : 1;               // waves along N
    constexpr int WM = 4 / WN;                           // waves along M
    constexpr int MT = BM / (WM * 32);                   // 32-tall M tiles per wave
    constexpr int NTT = (BN >= 64) ? NT : 1;
    constexpr int A_EPC = PRECISE ? 4 : 8;               // elements per 16-byte global chunk of A
    constexpr int A_CPR = BK / A_EPC;                    // 16-byte chunks per A row and stage (8, or 4 for fast BK=32)
    constexpr int A_RPP = 256 / A_CPR;                   // rows per pass
    constexpr int A_PASSES = BM / A_RPP;
    constexpr int B_CPR = BK / 8;                        // 16-byte chunks per weight row and stage
    constexpr int B_CHUNKS = BN * B_CPR * (PRECISE ? 2 : 1);
    constexpr int B_PER_THREAD = (B_CHUNKS + 255) / 256;

    constexpr int STAGE_ELEMS = (BM + BN) * BKP * (PRECISE ? 2 : 1);    // bf16 elements of one LDS stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As_hi = reinterpret_cast<__bf16*>(smem);                   // stage 0; stage 1 at + STAGE_ELEMS
    __bf16* As_lo = As_hi + (PRECISE ? BM * BKP : 0);
    __bf16* Bs_hi = As_lo + BM * BKP;
    __bf16* Bs_lo = Bs_hi + (PRECISE ? BN * BKP : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const long M = (long)p.N * p.Ho * p.Wo;
    const long m0 = (long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;

    // ---- per-thread A rows: A_PASSES passes of A_RPP rows, A_CPR chunks per row
    const int a_chunk = tid % A_CPR;
    const int a_row0 = tid / A_CPR;
    int a_n[A_PASSES], a_oh[A_PASSES], a_ow[A_PASSES];
    bool a_ok[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        long m = m0 + a_row0 + A_RPP * i;
        a_ok[i] = m < M;
        if (!a_ok[i]) m = M - 1;
        int n, rem;
        split_row(m, p.Ho * p.Wo, M, n, rem);
        a_n[i] = n;
        a_oh[i] = rem / p.Wo;
        a_ow[i] = rem - a_oh[i] * p.Wo;
    }

    f32x16 acc[MT][NTT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    const int n_taps = p.KH * p.KW;
    const int cin_p = (p.Cin + 31) / 32 * 32;            // packed weights pad Cin to a multiple of 32 with zeros
    const int chunks_per_tap = (cin_p + BK - 1) / BK;
    const int n_stage = n_taps * chunks_per_tap;
    const long w_row = (long)n_taps * cin_p;              // packed weight row length (bf16 elements)

    // ---- software pipeline: global loads run TWO stages ahead (two register sets), LDS is double buffered,
    //      one barrier per stage:   iteration s:  issue loads(s+2) | MFMAs on LDS[s&1] | regs(s+1) -> LDS[(s+1)&1] | barrier
    u32x4 a_r0[A_PASSES], a_r1[A_PASSES];
    u32x4 b_r0[B_PER_THREAD], b_r1[B_PER_THREAD];

    // pixel offset (in pixels, -1 = padding / out of range) of each of this thread's rows for the CURRENT tap;
    // recomputed only when the tap changes (every chunks_per_tap stages)
    long a_pix[A_PASSES];
    int cur_tap = -1;
    auto set_tap = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            int ih = a_oh[i] * p.stride - p.pad + kh;
            int iw = a_ow[i] * p.stride - p.pad_w + kw;
            bool ok = a_ok[i];
            if (p.reflect) {
                ih = reflect_idx(ih, p.H * p.up);
                iw = reflect_idx(iw, p.W * p.up);
            }
            if (p.up == 2) {                                 // dilated input grid (stride-2 data gradient)
                ok = ok && ih >= 0 && iw >= 0 && !((ih | iw) & 1);
                ih >>= 1;
                iw >>= 1;
            }
            ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            a_pix[i] = ok ? ((long)a_n[i] * p.H + ih) * p.W + iw : -1;
        }
        cur_tap = tap;
    };

    auto load_stage = [&](int s, u32x4 (&a_reg)[A_PASSES], u32x4 (&b_reg)[B_PER_THREAD]) {
        const int tap = s / chunks_per_tap;
        const int ci0 = (s - tap * chunks_per_tap) * BK;
        if (tap != cur_tap) set_tap(tap);
        // which source tensor holds this thread's 16-byte chunk (concat fusion): [0, C1) -> x, [C1, Cin) -> x2
        const int cg = ci0 + a_chunk * A_EPC;              // global input channel of the chunk
        const ASource as = a_source(p, cg);
        const char* src = as.src;
        const int csrc = as.csrc, cbase = as.cbase;
        const bool c_ok = cg < p.Cin;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c_ok && a_pix[i] >= 0) v = *reinterpret_cast<const u32x4*>(src + (a_pix[i] * csrc + cbase) * (PRECISE ? 4 : 2));
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            const int idx = tid + 256 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < B_CHUNKS) {
                const int arr = PRECISE ? idx / (BN * B_CPR) : 0;
                const int rem = PRECISE ? idx % (BN * B_CPR) : idx;
                const int row = rem / B_CPR, ch = rem % B_CPR;
                const unsigned short* wsrc = arr ? p.w_lo : p.w_hi;
                const int co = co0 + row;
                if (co < p.Cout && ci0 + ch * 8 < cin_p)
                    v = *reinterpret_cast<const u32x4*>(wsrc + (long)co * w_row + (long)tap * cin_p + ci0 + ch * 8);
            }
            b_reg[i] = v;
        }
    };

    auto store_stage = [&](int buf, const u32x4 (&a_reg)[A_PASSES], const u32x4 (&b_reg)[B_PER_THREAD]) {
        __bf16* Ah = As_hi + buf * STAGE_ELEMS;
        __bf16* Al = As_lo + buf * STAGE_ELEMS;
        __bf16* Bh = Bs_hi + buf * STAGE_ELEMS;
        __bf16* Bl = Bs_lo + buf * STAGE_ELEMS;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int row = a_row0 + A_RPP * i;
            if (PRECISE) {
                const f32x4 f = __builtin_bit_cast(f32x4, a_reg[i]);
                u32x2 hi, lo;
                hi[0] = pack2(f[0], f[1]);
                hi[1] = pack2(f[2], f[3]);
                lo[0] = pack2(f[0] - bf16_round(f[0]), f[1] - bf16_round(f[1]));
                lo[1] = pack2(f[2] - bf16_round(f[2]), f[3] - bf16_round(f[3]));
                *reinterpret_cast<u32x2*>(Ah + row * BKP + a_chunk * 4) = hi;
                *reinterpret_cast<u32x2*>(Al + row * BKP + a_chunk * 4) = lo;
            } else {
                *reinterpret_cast<u32x4*>(Ah + row * BKP + a_chunk * 8) = a_reg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < B_CHUNKS) {
                const int arr = PRECISE ? idx / (BN * B_CPR) : 0;
                const int rem = PRECISE ? idx % (BN * B_CPR) : idx;
                const int row = rem / B_CPR, ch = rem % B_CPR;
                __bf16* dst = arr ? Bl : Bh;
                *reinterpret_cast<u32x4*>(dst + row * BKP + ch * 8) = b_reg[i];
            }
        }
    };

    auto compute = [&](int buf) {
        const __bf16* Ah = As_hi + buf * STAGE_ELEMS;
        const __bf16* Al = As_lo + buf * STAGE_ELEMS;
        const __bf16* Bh = Bs_hi + buf * STAGE_ELEMS;
        const __bf16* Bl = Bs_lo + buf * STAGE_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a_hi[MT], a_lo[MT], b_hi[NTT], b_lo[NTT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = (wm * MT + a) * 32 + r;
                a_hi[a] = *reinterpret_cast<const bf16x8*>(Ah + row * BKP + kk * 16 + h * 8);
                if (PRECISE) a_lo[a] = *reinterpret_cast<const bf16x8*>(Al + row * BKP + kk * 16 + h * 8);
            }
#pragma unroll
            for (int b = 0; b < NTT; ++b) {
                const int row = (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
                b_hi[b] = *reinterpret_cast<const bf16x8*>(Bh + row * BKP + kk * 16 + h * 8);
                if (PRECISE) b_lo[b] = *reinterpret_cast<const bf16x8*>(Bl + row * BKP + kk * 16 + h * 8);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NTT; ++b) {
                    if (PRECISE) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[a], b_hi[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[a], b_lo[b], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[a], b_hi[b], acc[a][b], 0, 0, 0);
                }
        }
    };

    load_stage(0, a_r0, b_r0);
    if (n_stage > 1) load_stage(1, a_r1, b_r1);
    store_stage(0, a_r0, b_r0);
    __syncthreads();
    for (int s = 0; s < n_stage; s += 2) {
        // even stage s: registers set 0 is free (stored last iteration / prologue), set 1 holds stage s+1
        if (s + 2 < n_stage) load_stage(s + 2, a_r0, b_r0);
        compute(0);
        if (s + 1 < n_stage) store_stage(1, a_r1, b_r1);
        __syncthreads();
        if (s + 1 >= n_stage) break;
        // odd stage s+1
        if (s + 3 < n_stage) load_stage(s + 3, a_r1, b_r1);
        compute(1);
        if (s + 2 < n_stage) store_stage(0, a_r0, b_r0);
        __syncthreads();
    }

    conv_epilogue<BM, BN, PRECISE, MT, NTT, NT, 256, LinearRows, S3, false>(acc, p, smem, M, m0, co0, wm, wn, r, h, tid);
}

// =====================================================================================================
// Fast-mode (bf16) convolution with LDS-DMA staging.
//
// The register-staged kernel above is LDS-bound in bf16 mode: every stage pushes 32 KiB through the ds_write
// path (~79 B/clk/CU) on top of 64 KiB of fragment reads.  Here both operand tiles go global -> LDS directly
// (global_load_lds_dwordx4: no VGPR staging, no ds_write), which leaves only the fragment reads on the LDS.
//   * LDS image: [row][64 bf16] = 128-byte rows, UNPADDED (a wave-instruction writes 1 KiB = 8 whole rows,
//     lane l -> row l>>3, 16-byte slot l&7), XOR-swizzled through the SOURCE address: slot p of row R holds
//     channel chunk c = p ^ ((R >> 1) & 7); readers apply the same XOR.  With two rows per 256-byte bank row this
//     makes every 16-lane ds_read_b128 group hit 16 distinct slots (conflict-free).
//   * zero padding / out-of-range rows / channels read a 256-byte zero page in the code object.
//   * double-buffered stages, one barrier per stage, the next stage's DMA issued before the current MFMAs.
// Requires every tap's channel range to be whole 64-channel chunks (Cin % 64 == 0, concat split % 64 == 0).
// =====================================================================================================
__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];


__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LIN: 1x1 / stride 1 / no padding, output pixel grid == input pixel grid: GEMM row m IS input pixel m.  The generic prologue
// spends ~600 VALU instructions per wave on (image, row, column) splits and tap geometry that such a layer does not need -- and
// these launches are VALU-ISSUE bound, not memory bound (rocprofv3 SQ_INSTS_VALU: 1544 per wave for a tile whose K loop is 16 MFMAs;
// 16 resident waves per CU x 1544 x 4 cycles = 94 of the 111 us of the 64 -> 256 layer at 128^2; profiles/LEDGER.md, round 3).
template <int TBM, int BN, int NW, int NBUF, int MINW = 1, bool S3 = false, bool LIN = false, bool CLS = false>
__global__ __launch_bounds__(NW * 64, MINW) void conv_igemm_glds_kernel(const ConvArgs pk) {
    // CLS: the launch holds several convolution problems over the same input (pk.cls: the parity classes of a stride-2 data
    // gradient); this workgroup's class replaces the per-problem fields.  All of it is workgroup-uniform (scalar registers).
    ConvArgs pc;
    long cls_tile0 = 0;
    if constexpr (CLS) {
        pc = pk;
        int c = 0;
        while (c + 1 < pk.n_cls && blockIdx.x >= pk.cls[c].tile_end) ++c;
        cls_tile0 = c ? pk.cls[c - 1].tile_end : 0;
        const ConvArgs::Cls& k = pk.cls[c];
        pc.KH = k.KH, pc.KW = k.KW, pc.pad = k.pad, pc.pad_w = k.pad_w, pc.Ho = k.Ho, pc.Wo = k.Wo;
        pc.omap_ph = k.ph, pc.omap_pw = k.pw;
        pc.w_hi = pk.w_hi + k.w_off;
    }
    const ConvArgs& p = CLS ? pc : pk;
    constexpr int BK = 64;
    // wave grid WM x WN over the TBM x BN tile; wave tile (MT*32) x (NT*32)
    constexpr int WN = (BN >= 256) ? 4 : (BN >= 64 ? 2 : 1);
    constexpr int WM = NW / WN;
    constexpr int MT = TBM / (WM * 32);
    constexpr int NT = (BN >= 64) ? BN / (WN * 32) : 1;
    constexpr int NTT = NT;
    constexpr int STAGE_BYTES = (TBM + BN) * BK * 2;
    constexpr int A_INSTR = TBM / 8 / NW;                // wave-instructions per wave for the A tile (8 rows each)
    constexpr int B_ROWS_PER_WAVE = BN / NW;
    constexpr int B_INSTR = (B_ROWS_PER_WAVE + 7) / 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const long M = (long)p.N * p.Ho * p.Wo;
    // 1-D launch (pair_chunks > 0): the Cout chunks of an M tile sit 8 workgroup ids apart = on the same XCD at about the
    // same time, so the A rows come from HBM once and from that XCD's L2 for the other chunks
    long m_tile = blockIdx.x - cls_tile0;
    int co_chunk = blockIdx.y;
    if (p.pair_chunks > 0) {
        const unsigned group = 8u * (unsigned)p.pair_chunks, within = blockIdx.x % group;
        m_tile = (long)(blockIdx.x / group) * 8 + (within & 7u);
        co_chunk = (int)(within >> 3);
        if (m_tile >= p.pair_tiles) return;
    }
    const long m0 = m_tile * TBM;
    const int co0 = co_chunk * BN;

    // ---- this lane's A rows: instruction i of this wave covers tile rows 8*(A_INSTR*wave + i) .. +7; lane -> row l>>3
    const int slot = lane & 7;
    int a_n[A_INSTR], a_oh[A_INSTR], a_ow[A_INSTR], a_chunk[A_INSTR];
    bool a_ok[A_INSTR];
    long a_pix[A_INSTR];                                   // pixel offset of this lane's rows for the current tap, -1 = zero
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int trow = 8 * (A_INSTR * wave + i) + (lane >> 3);
        long m = m0 + trow;
        a_ok[i] = m < M;
        a_chunk[i] = slot ^ ((trow >> 1) & 7);           // channel chunk (8 bf16) this lane fetches for its slot
        if constexpr (LIN) {
            a_pix[i] = a_ok[i] ? m : -1;
            a_n[i] = a_oh[i] = a_ow[i] = 0;
        } else {
            if (!a_ok[i]) m = M - 1;
            int n, rem;
            split_row(m, p.Ho * p.Wo, M, n, rem);
            a_n[i] = n;
            a_oh[i] = rem / p.Wo;
            a_ow[i] = rem - a_oh[i] * p.Wo;
            if (p.ring) {                                    // rem = position on the border ring of a ring_h x ring_w grid
                const int q = rem, wp = p.ring_w, hp = p.ring_h;
                if (q < wp) a_oh[i] = 0, a_ow[i] = q;
                else if (q < 2 * wp) a_oh[i] = hp - 1, a_ow[i] = q - wp;
                else if (q < 2 * wp + hp - 2) a_oh[i] = 1 + (q - 2 * wp), a_ow[i] = 0;
                else a_oh[i] = 1 + (q - 2 * wp - (hp - 2)), a_ow[i] = wp - 1;
            }
        }
    }
    // ---- this lane's B rows (weights): wave w covers tile rows [w * BN/NW, (w+1) * BN/NW)
    // byte offset of (row, swizzled slot) inside the packed image, ~0u: no such row (zero page); the (tap, chunk) part of the
    // address is uniform and lives in scalar registers
    unsigned w_off[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int b_row = B_ROWS_PER_WAVE * wave + 8 * i + (lane >> 3);
        const int co = co0 + b_row;
        const bool ok = co < p.Cout && b_row < B_ROWS_PER_WAVE * (wave + 1);
        w_off[i] = ok ? ((unsigned)co * (unsigned)(p.KH * p.KW * ((p.Cin + 31) / 32 * 32)) + (unsigned)((slot ^ ((b_row >> 1) & 7)) << 3)) * 2u
                      : ~0u;
    }

    f32x16 acc[MT][NTT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    const int n_taps = p.KH * p.KW;
    const int cin_p = (p.Cin + 31) / 32 * 32;
    const int chunks_per_tap = p.Cin / BK;
    const int n_stage = n_taps * chunks_per_tap;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    int cur_tap = LIN ? 0 : -1;                            // LIN: one tap, a_pix set above
    auto set_tap = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            int ih = a_oh[i] * p.stride - p.pad + kh;
            int iw = a_ow[i] * p.stride - p.pad_w + kw;
            bool ok = a_ok[i];
            if (p.reflect) {
                ih = reflect_idx(ih, p.H * p.up);
                iw = reflect_idx(iw, p.W * p.up);
            }
            if (p.up == 2) {
                ok = ok && ih >= 0 && iw >= 0 && !((ih | iw) & 1);
                ih >>= 1;
                iw >>= 1;
            }
            ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            a_pix[i] = ok ? ((long)a_n[i] * p.H + ih) * p.W + iw : -1;
        }
        cur_tap = tap;
    };

    auto stage = [&](int s, int buf) {
        char* As = smem + buf * STAGE_BYTES;
        char* Bs = As + TBM * BK * 2;
        const int tap = s / chunks_per_tap;
        const int ci0 = (s - tap * chunks_per_tap) * BK;
        if (tap != cur_tap) set_tap(tap);
        const ASource as = a_source(p, ci0);               // whole stage comes from one source (split % 64 == 0)
        const char* src = as.src;
        const int csrc = as.csrc, cbase = as.cbase;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const long off = a_pix[i] * csrc + cbase + a_chunk[i] * 8;
            if (!(GLDS_ABL & 8)) glds16(a_pix[i] >= 0 ? src + off * 2 : zero, As + (A_INSTR * wave + i) * 1024);
        }
        const char* wbase = reinterpret_cast<const char*>(p.w_hi) + ((long)tap * cin_p + ci0) * 2;             // uniform
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            glds16(w_off[i] != ~0u ? wbase + w_off[i] : zero, Bs + (B_ROWS_PER_WAVE * wave + 8 * i) * 128);
    };

    auto compute = [&](int buf) {
        const char* As = smem + buf * STAGE_BYTES;
        const char* Bs = As + TBM * BK * 2;
#pragma unroll MT * NTT >= 8 ? 1 : 4
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[MT], bfr[NTT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = (wm * MT + a) * 32 + r;
                const int sl = (kk * 2 + h) ^ ((row >> 1) & 7);
                af[a] = *reinterpret_cast<const bf16x8*>(As + row * 128 + sl * 16);
            }
#pragma unroll
            for (int b = 0; b < NTT; ++b) {
                const int row = (BN >= 64 ? (wn * NT + b) * 32 : 0) + r;
                const int sl = (kk * 2 + h) ^ ((row >> 1) & 7);
                bfr[b] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + sl * 16);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NTT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    };

    // NBUF-deep ring, DMA NBUF-1 stages ahead, ONE raw barrier per stage and COUNTED vmcnt (never drained in the loop):
    //   iteration s:  wait until this wave's pieces of stage s have landed (later stages may still fly)  ->  s_barrier
    //   (everyone's pieces landed, everyone finished the MFMAs of stage s-1, so the slot of stage s-1 is free)  ->  issue
    //   the DMA of stage s+NBUF-1 into it  ->  MFMAs of stage s.  A slot is read only AFTER wait + barrier.
    constexpr int G = A_INSTR + B_INSTR;                   // DMA wave-instructions per wave and stage
    auto wait_in_flight = [&](int stages) {                // s_waitcnt vmcnt(stages * G), expcnt/lgkmcnt untouched
        if (stages >= 2 && NBUF >= 4) __builtin_amdgcn_s_waitcnt(((2 * G) & 0xF) | (((2 * G) >> 4) << 14) | 0x0F70);
        else if (stages >= 1 && NBUF >= 3) __builtin_amdgcn_s_waitcnt((G & 0xF) | ((G >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    if constexpr (NBUF == 1) {
        // one stage buffer: no overlap inside the workgroup -- the point is its small footprint (several workgroups per
        // CU cover each other's loads and epilogues), for layers whose whole K loop is a few stages
        for (int s = 0; s < n_stage; ++s) {
            stage(s, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            compute(0);
            __builtin_amdgcn_s_barrier();
        }
    } else {
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < n_stage) stage(i, i);
    int slot_c = 0, slot_i = NBUF - 1;                     // ring slots of stage s (compute) and s+NBUF-1 (issue)
    for (int s = 0; s < n_stage; ++s) {
        const int behind = n_stage - 1 - s;                // stages issued after stage s that may still be in flight
        wait_in_flight(behind < NBUF - 2 ? behind : NBUF - 2);
        __builtin_amdgcn_s_barrier();
        if (s + NBUF - 1 < n_stage) stage(s + NBUF - 1, slot_i);
        compute(slot_c);
        slot_c = slot_c == NBUF - 1 ? 0 : slot_c + 1;
        slot_i = slot_i == NBUF - 1 ? 0 : slot_i + 1;
    }
    }
    __syncthreads();                                       // all MFMAs done: LDS is free for the output tile
#if GLDS_ABL & 2
    {
        float sink = 0.0f;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NTT; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) sink += acc[a][b][i];
        if (sink == 12345.678f) reinterpret_cast<float*>(p.y)[0] = sink;
        return;
    }
#endif
    conv_epilogue<TBM, BN, false, MT, NTT, NT, NW * 64, LinearRows, S3>(acc, p, smem, M, m0, co0, wm, wn, r, h, tid);
}

// =====================================================================================================
// The stem (7x7 / stride 2 / pad 3 on the 3-channel fp32 image) WITHOUT its patch matrix (r4).
//
// As a 1x1 convolution over a materialised [pixels][160] patch matrix the stem moved 11.5 GB per step: the bf16 matrix written once
// per batch (0.67 GB) and read by both networks' forward and again by their weight gradients, the split-3 matrix of the pseudo-label
// forwards 1.6 GB written and read twice.  Here a workgroup owns 128 consecutive output pixels of one output row x all 64 output
// channels: it stages the 7 input rows x 261 input pixels x 3 channels those pixels read (fp32, padding applied while staging, as
// im2col_stem7_strip_kernel) and the weight image in LDS.  The contraction runs over k' = kh * 24 + (kw * 3 + ci) -- each kernel
// row's 21 taps padded to 24, 7 x 24 = 168 padded to 176 = eleven MFMA K steps -- so that the 8 operands of a fragment are 8
// CONSECUTIVE words of one staged input row (offset 6 * pixel + 8 * (k' / 8 % 3)): four ds_read_b64 and four conversions per
// fragment, no per-element address arithmetic (the first version kept the patch matrix's k order and spent ~1300 VALU instructions
// per lane on it: 313 us, slower than the 280 us it replaced).  The padded positions meet zero weights; what they read is the next
// pixel's first words (finite) -- the row tails are zero-filled.  Same v_mfma_f32_32x32x16_bf16, another summation grouping than the
// patch-matrix kernel: equal to rounding, not bit-identical.  Split-3: hi w_hi, lo w_hi, hi w_lo per K step.  Same epilogue (raw y +
// BatchNorm partials, or the fused affine epilogue).  Reads the image, writes y.
// Weight image: [64][176] bf16 (split-3: [64][2][176] = w_hi | w_lo), columns kh * 24 + kw * 3 + ci, zero elsewhere.
// =====================================================================================================
#ifndef STEM_ABL
#define STEM_ABL 0                // debug builds only (results wrong): 1 no image loads, 2 no epilogue
#endif
template <bool S3>
__global__ __launch_bounds__(256) void stem7_fused_kernel(const ConvArgs p) {
    constexpr int TP = 128, NCOL = (2 * (TP - 1) + 7) * 3, ROW = 790, KQ = 176, KPW = KQ + 8, NH = (ROW + 255) / 256, NK = KQ / 16;
    static_assert(ROW >= NCOL + 2 && ROW % 2 == 0, "strip row: room for the 2 words a padded fragment reads past the last pixel; 4-byte rows");
    // The strip is kept as bf16 (split-3: a hi plane and a lo plane): the fragments are what bounds this kernel -- every wave gathers
    // 22 x 2 fragments from LDS (fp32 words: 360 KB of LDS reads per workgroup, 200 us for the loop alone) -- and bf16 halves the bytes
    // and drops the conversions from the gather.
    constexpr int PLANE = 7 * ROW;                                          // bf16 elements of one plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* strip = reinterpret_cast<__bf16*>(smem);                        // [planes][7][ROW]
    __bf16* Bh = reinterpret_cast<__bf16*>(smem + (S3 ? 2 : 1) * ((PLANE * 2 + 15) / 16 * 16));     // [64][KPW] w_hi
    __bf16* Bl = Bh + 64 * KPW;                                             // [64][KPW] w_lo (split-3)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                                // wave tile 64 pixels x 32 channels: MT = 2, NTT = 1
    const int r = lane & 31, h = lane >> 5;
    const int strips = p.Wo / TP;
    const int sx = blockIdx.x % strips;
    const int oh = (blockIdx.x / strips) % p.Ho, n = blockIdx.x / (strips * p.Ho);
    const int H = p.H, W = p.W;
    const long M = (long)p.N * p.Ho * p.Wo;
    const long m0 = ((long)n * p.Ho + oh) * p.Wo + (long)sx * TP;
    const int iw0 = 2 * sx * TP - 3;
    // ---- stage the image strip (unconditional loads, clamped addresses: see im2col_stem7_strip_kernel) and the weights
    const float* xn = reinterpret_cast<const float*>(p.x) + (size_t)n * H * W * 3;
    float rv[NH][7];
#pragma unroll
    for (int q = 0; q < NH; ++q) {
        const int c = tid + 256 * q;
        const int px = c / 3, ci = c - 3 * px;
        int iw = iw0 + px;
        if (p.reflect) {
            if (iw < 0) iw = -iw;
            if (iw >= W) iw = 2 * W - 2 - iw;
        }
        const bool cok = c < NCOL && iw >= 0 && iw < W;
        const int coff = cok ? iw * 3 + ci : 0;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            int ih = 2 * oh - 3 + kh;
            if (p.reflect) {
                if (ih < 0) ih = -ih;
                if (ih >= H) ih = 2 * H - 2 - ih;
            }
            const bool ok = cok && ih >= 0 && ih < H;
#if STEM_ABL & 1
            const float v = (float)coff;
#else
            const float v = xn[(ok ? ih : 0) * W * 3 + coff];
#endif
            rv[q][kh] = ok ? v : 0.0f;
        }
    }
    {
        const int row_len = S3 ? 2 * KQ : KQ;
        for (int i = tid; i < 64 * (KQ / 8); i += 256) {
            const int co = i / (KQ / 8), ch = i - co * (KQ / 8);
            *reinterpret_cast<u32x4*>(Bh + co * KPW + ch * 8) = *reinterpret_cast<const u32x4*>(p.w_hi + (size_t)co * row_len + ch * 8);
            if constexpr (S3)
                *reinterpret_cast<u32x4*>(Bl + co * KPW + ch * 8) = *reinterpret_cast<const u32x4*>(p.w_hi + (size_t)co * row_len + KQ + ch * 8);
        }
    }
#pragma unroll
    for (int q = 0; q < NH; ++q)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
            if (tid + 256 * q < ROW) {                      // (columns >= NCOL: zeros)
                const __bf16 hi = (__bf16)rv[q][kh];
                strip[kh * ROW + tid + 256 * q] = hi;
                if constexpr (S3) strip[PLANE + kh * ROW + tid + 256 * q] = (__bf16)(rv[q][kh] - (float)hi);
            }
    __syncthreads();

    // ---- K loop.  A fragments: K step kk, lane half h -> 8-group g8 = 2 kk + h of k' = kernel row g8 / 3 (clamped: the last group is
    // padding), taps 8 (g8 % 3) .. + 7 of that row: 8 consecutive bf16 = four 4-byte LDS reads (a pixel is 6 elements = 12 bytes
    // further on).  Fragments are used as they arrive (few registers: four workgroups per CU cover each other's load / store phases);
    // split-3: the three products of a K step back to back.
    f32x16 acc[2][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][0][i] = 0.0f;
    const __bf16* brow = Bh + (wn * 32 + r) * KPW + h * 8;
    const __bf16* blrow = Bl + (wn * 32 + r) * KPW + h * 8;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const int g8 = 2 * kk + h;
        int kh = (g8 * 11) >> 5;                                            // g8 / 3 for g8 < 32
        const int j0 = 8 * (g8 - 3 * kh);
        if (kh > 6) kh = 6;
        const __bf16* src = strip + kh * ROW + j0;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + kk * 16);
        bf16x8 bl;
        if constexpr (S3) bl = *reinterpret_cast<const bf16x8*>(blrow + kk * 16);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned* sp = reinterpret_cast<const unsigned*>(src + 6 * ((wm * 2 + a) * 32 + r));
            const bf16x8 ah = __builtin_bit_cast(bf16x8, u32x4{sp[0], sp[1], sp[2], sp[3]});
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b, acc[a][0], 0, 0, 0);
            if constexpr (S3) {
                const unsigned* sl = sp + PLANE / 2;
                const bf16x8 al = __builtin_bit_cast(bf16x8, u32x4{sl[0], sl[1], sl[2], sl[3]});
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc[a][0], 0, 0, 0);
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[a][0], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                       // LDS is free for the output tile
#if STEM_ABL & 2
    if (acc[0][0][0] != 12345.678f) return;
#endif
    conv_epilogue<TP, 64, false, 2, 1, 1, 256, LinearRows, S3, true>(acc, p, smem, M, m0, 0, wm, wn, r, h, tid);
}

static int g_glds_pair = 8;                                 // layers with 2..this many Cout chunks: the chunks of an M tile share an XCD
                                                            // (1-D launch, see the kernel; alone: -3..-6 % at 2-4 chunks, +5 % at 8; in the two-stream step 8 wins: r4)

template <int TBM, int BN, int NW, int NBUF, int MINW = 1, bool LIN = false>
static void launch_glds_lin(const ConvArgs& a, hipStream_t st);

static int g_glds_lin = 1;                                  // 1x1 / stride 1 / pad 0 layers take the linear-pixel prologue (LIN)
template <int TBM, int BN, int NW, int NBUF, int MINW = 1>
static void launch_glds_t(const ConvArgs& a, hipStream_t st) {
    const bool lin = g_glds_lin && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.pad_w == 0 && a.up == 1 && !a.omap &&
                     a.Ho == a.H && a.Wo == a.W;
    if (lin) launch_glds_lin<TBM, BN, NW, NBUF, MINW, true>(a, st);
    else launch_glds_lin<TBM, BN, NW, NBUF, MINW, false>(a, st);
}

template <int TBM, int BN, int NW, int NBUF, int MINW, bool LIN>
static void launch_glds_lin(const ConvArgs& a, hipStream_t st) {
    const int n_stage = a.KH * a.KW * (a.Cin / 64);
    // ring slots actually used: a short K loop (1x1 layers with 64..128 input channels) then leaves LDS for more
    // resident workgroups, whose loads overlap each other's epilogues
    size_t lds = (size_t)(TBM + BN) * 64 * 2 * (n_stage < NBUF ? n_stage : NBUF);
    const size_t out_tile = (size_t)TBM * (BN + 8) * 2 * (a.out_s3 ? 2 : 1);
    if (out_tile > lds) lds = out_tile;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long m_tiles = (M + TBM - 1) / TBM;
    const int chunks = (a.Cout + BN - 1) / BN;
    if (chunks > 1 && chunks <= g_glds_pair && m_tiles < (1L << 30)) {
        ConvArgs b = a;
        b.pair_chunks = chunks;
        b.pair_tiles = (int)m_tiles;
        const dim3 grid1((unsigned)((m_tiles + 7) / 8 * 8 * chunks));
        if (a.out_s3) hipLaunchKernelGGL((conv_igemm_glds_kernel<TBM, BN, NW, NBUF, MINW, true, LIN>), grid1, dim3(NW * 64), lds, st, b);
        else hipLaunchKernelGGL((conv_igemm_glds_kernel<TBM, BN, NW, NBUF, MINW, false, LIN>), grid1, dim3(NW * 64), lds, st, b);
        return;
    }
    dim3 grid((unsigned)m_tiles, (unsigned)chunks);
    if (a.out_s3) hipLaunchKernelGGL((conv_igemm_glds_kernel<TBM, BN, NW, NBUF, MINW, true, LIN>), grid, dim3(NW * 64), lds, st, a);
    else hipLaunchKernelGGL((conv_igemm_glds_kernel<TBM, BN, NW, NBUF, MINW, false, LIN>), grid, dim3(NW * 64), lds, st, a);
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolution with PATCH REUSE (bf16; forward of every such layer and the data gradient
// of the zero-padded ones).
//
// The generic kernel above fetches a shifted copy of the input tile for each of the nine taps (A traffic 9x).
// Here a workgroup owns a 2-D tile of 256 output pixels (8 x 32, or 16 x 16 for narrow images) x BN output
// channels.  Per 64-channel chunk of the input it stages ONE haloed patch [(TH+2) x (TW+2) px][64 ci] and feeds
// all nine taps from shifted windows of it: the MFMA row (32 consecutive tile pixels) of tap (kh, kw) is the same
// LDS rows displaced by kh * (TW+2) + kw.  Only the weights [BN][64 ci] change per tap.
//   LDS: two patch buffers (the next chunk's patch arrives, one DMA instruction per wave and tap, under the taps
//   of the current chunk) + a ring of NBW weight stages; counted vmcnt, one raw barrier per tap.
// Rows are unpadded 128 B with the 16-byte chunks XOR-swizzled through the DMA source address.
// =====================================================================================================
#ifndef PATCH_ABL
#define PATCH_ABL 0               // debug builds only (results wrong): 1 no weight DMA after the prologue, 2 no patch DMA after chunk 0,
#endif                            // 4 no per-tap barrier, 8 no per-tap vmcnt wait  (tools/ab_conv.py with VQSEG_LIB: DESIGN 4.2, "what bounds it")
#ifndef CONV_SETPRIO
#define CONV_SETPRIO 0            // 1: s_setprio pair around every MFMA cluster of the patch kernel (measured: see DESIGN 4.2)
#endif
struct TileRows {                                           // tile row -> output pixel row of a 2-D pixel tile
    long base;                                              // (n * H + oh0) * W + ow0
    int tw_shift, W;
    __device__ __forceinline__ long operator()(int row) const {
        return base + (long)(row >> tw_shift) * W + (row & ((1 << tw_shift) - 1));
    }
};

// CK: input channels per chunk (64, or 32 for layers whose channel count is not a multiple of 64: rows of 64 bytes,
// four 16-byte chunks swizzled by (row >> 2) & 3, two K steps per tap)
// TBM: output pixels per workgroup (256, or 512 with 32-channel chunks: twice the pixels per staged weight tile -- half the weight
// DMA per MFMA -- and wave tiles twice as tall: a quarter fewer fragment reads per MFMA (a third for the 32-channel-wide tiles))
template <int BN, int NBW, bool UNROLL_TAPS, int CK = 64, bool S3 = false, int TBM = 256>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const ConvArgs p) {
    constexpr int NW = 8;
    static_assert(TBM == 256 || (TBM == 512 && CK == 32), "512-pixel tiles stage 32-channel chunks (two patch buffers must fit the LDS)");
    constexpr int PATCH_PX = TBM == 256 ? 344 : 616;       // haloed patch pixels, rounded up: 10 x 34 (18 x 18) / 18 x 34 (34 x 18)
    constexpr int NT = BN >= 128 ? 2 : 1;                  // 32-channel tiles per wave
    constexpr int WN = BN / (32 * NT), WM = NW / WN;       // wave grid; wave tile (MT*32) px x (NT*32) co
    constexpr int MT = TBM / (WM * 32);                    // BN 256/128/64/32 -> MT 4/2/2/1 (64/32-row BN stat slots)
    constexpr int ROWB = CK * 2;                           // bytes per LDS row (one pixel / one output channel)
    constexpr int RPI = 1024 / ROWB;                       // rows per 1 KB DMA instruction
    constexpr int KS = CK / 16;                            // MFMA K steps per tap
    constexpr int PI = (PATCH_PX + RPI * NW - 1) / (RPI * NW);   // patch DMA instructions per wave and chunk (6 or 3; 5 for 512 pixels)
    constexpr int PATCH_BYTES = PI * NW * 1024;
    constexpr int W_ROWS = (BN * ROWB >= NW * 1024) ? BN : NW * 1024 / ROWB;   // weight rows staged per tap (rows >= Cout: zero page)
    constexpr int W_BYTES = W_ROWS * ROWB;
    constexpr int WI = W_BYTES / 1024 / NW;                // weight DMA instructions per wave and tap
    static_assert(W_BYTES % (1024 * NW) == 0 && (CK == 64 || CK == 32), "weight tile must split evenly over the waves");
    // NBW == 0, "chunk stages": the weights of ALL nine taps of a chunk travel with its patch (one wait + one barrier per
    // chunk, none per tap).  For tiles whose taps are one or two K steps long a tap-by-tap ring only adds a DMA latency and a
    // barrier to every tap; the whole tap set of such a tile is small (9 x BN x CK x 2 B).
    constexpr bool CHUNK_STAGE = NBW == 0;
    constexpr int WT_BYTES = CHUNK_STAGE ? BN * ROWB : W_BYTES;             // LDS bytes per tap
    constexpr int WCI = 9 * BN * ROWB / 1024, WCW = (WCI + NW - 1) / NW;    // chunk-stage weight DMA instructions: all / per wave
    static_assert(!CHUNK_STAGE || (BN * ROWB) % 1024 == 0, "a tap's weight tile must be whole DMA instructions");
    auto swz = [](int row) { return CK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch0 = smem;
    char* const wring = smem + ((CHUNK_STAGE && p.Cin / CK == 1) ? 1 : 2) * PATCH_BYTES;     // single-chunk layers: one patch buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // ---- tile geometry (host guarantees H % TH == 0, W % TW == 0, Ho == H, Wo == W)
    const int tw_shift = (p.W & 31) == 0 ? 5 : 4;
    const int TW = 1 << tw_shift, TH = TBM >> tw_shift, PW = TW + 2, PH = TH + 2;
    const int tiles_x = p.W >> tw_shift, tiles_y = p.H / TH;
    // 1-D launch (pair_chunks > 0): the workgroups that share a pixel tile (one per Cout chunk) sit 8 ids apart, i.e. on the
    // same XCD at about the same time, so the patch is fetched from HBM once and then served by that XCD's L2
    int tile_id = blockIdx.x, co_chunk = blockIdx.y;
    if (p.pair_chunks > 0) {
        const unsigned group = 8u * (unsigned)p.pair_chunks, within = blockIdx.x % group;
        tile_id = (int)(blockIdx.x / group) * 8 + (int)(within & 7u);
        co_chunk = (int)(within >> 3);
        if (tile_id >= p.pair_tiles) return;
    }
    int t = tile_id;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    const int n = t / tiles_y;
    const int oh0 = tyi * TH, ow0 = txi * TW;
    const int co0 = co_chunk * BN;

    const int slot = lane & (ROWB / 16 - 1), lrow = lane / (ROWB / 16);     // 16-byte slot and row of this lane in a DMA instruction
    // ---- this lane's weight rows: byte offset of (row, swizzled 16-byte slot) inside the packed image; the (tap, chunk) part of
    // the address is uniform and stays in scalar registers (one 32-bit VGPR per DMA instruction instead of a 64-bit pointer
    // per instruction and tap).  Rows past Cout (tiles narrower than the DMA granule) re-read the last row: their outputs are
    // never stored.
    unsigned w_off[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = RPI * (wave * WI + i) + lrow;
        int co = co0 + row;
        co = co < p.Cout ? co : p.Cout - 1;
        w_off[i] = ((unsigned)co * (unsigned)(9 * ((p.Cin + 31) / 32 * 32)) + (unsigned)((slot ^ swz(row)) << 3)) * 2u;
    }
    // ---- MFMA A rows: tile row -> patch pixel (tap (0, 0))
    int a_pp[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int row = (wm * MT + a) * 32 + r;
        a_pp[a] = (row >> tw_shift) * PW + (row & (TW - 1));
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    const int cin_p = (p.Cin + 31) / 32 * 32;
    const int n_chunks = p.Cin / CK;
    const int n_stage = n_chunks * 9;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // patch rows: instruction i (0..PI-1) of this wave covers patch pixels RPI * (wave + NW * i) .. + RPI - 1.  The row's
    // input pixel is recomputed per instruction (a handful of VALU ops per tap) rather than kept in 18 registers.
    auto issue_patch = [&](int chunk, int i) {
        const int pp = RPI * (wave + NW * i) + lrow;
        const int pr = pp / PW, pc = pp - pr * PW;
        int ih = oh0 - 1 + pr, iw = ow0 - 1 + pc;
        bool ok = pr < PH;
        if (p.reflect && ok) {
            ih = reflect_idx(ih, p.H);
            iw = reflect_idx(iw, p.W);
        }
        ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const long pix = ((long)n * p.H + ih) * p.W + iw;
        const int ci0 = chunk * CK;
        const ASource as = a_source(p, ci0);
        const char* src = as.src;
        const int csrc = as.csrc, cbase = as.cbase;
        const long off = pix * csrc + cbase + ((slot ^ swz(pp)) << 3);
        glds16(ok ? src + off * 2 : zero, patch0 + (chunk & 1) * PATCH_BYTES + (wave + NW * i) * 1024);
    };
    auto issue_weights = [&](int chunk, int tap, int wslot) {
        char* Bs = wring + wslot * W_BYTES;
        const char* sbase = reinterpret_cast<const char*>(p.w_hi) + ((long)tap * cin_p + chunk * CK) * 2;      // uniform
#pragma unroll
        for (int i = 0; i < WI; ++i) glds16(sbase + w_off[i], Bs + (wave * WI + i) * 1024);
    };
    auto compute = [&](int chunk, int tapoff, const char* Bs) {
        const char* Ps = patch0 + (chunk & 1) * PATCH_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int pp = a_pp[a] + tapoff;
                af[a] = *reinterpret_cast<const bf16x8*>(Ps + pp * ROWB + (((kk * 2 + h) ^ swz(pp)) << 4));
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = (wn * NT + b) * 32 + r;
                bfr[b] = *reinterpret_cast<const bf16x8*>(Bs + row * ROWB + (((kk * 2 + h) ^ swz(row)) << 4));
            }
#if CONV_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
#if CONV_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
    };

    if constexpr (CHUNK_STAGE) {
        auto issue_chunk = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < PI; ++i) issue_patch(chunk, i);
            char* wb = wring + (chunk & 1) * (9 * WT_BYTES);
#pragma unroll
            for (int i = 0; i < WCW; ++i) {
                const int j = wave * WCW + i;                          // wave-uniform
                if (j < WCI) {
                    const int tap = j / (BN * ROWB / 1024), rg = j % (BN * ROWB / 1024);
                    const int row = RPI * rg + lrow;
                    int co = co0 + row;
                    co = co < p.Cout ? co : p.Cout - 1;                // outputs of such rows are never stored
                    const unsigned short* wp = p.w_hi + (long)co * (9L * cin_p) + (long)tap * cin_p + chunk * CK + ((slot ^ swz(row)) << 3);
                    glds16(reinterpret_cast<const char*>(wp), wb + j * 1024);
                }
            }
        };
        issue_chunk(0);
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): this chunk's patch and weights have landed
            __builtin_amdgcn_s_barrier();                              // ... for every wave; the other buffers are free
            if (chunk + 1 < n_chunks) issue_chunk(chunk + 1);
            const char* wb = wring + (chunk & 1) * (9 * WT_BYTES);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) compute(chunk, (tap / 3) * PW + tap % 3, wb + tap * WT_BYTES);
        }
        __syncthreads();
        const long M = (long)p.N * p.H * p.W;
        const TileRows rows{((long)n * p.H + oh0) * p.W + ow0, tw_shift, p.W};
        conv_epilogue<TBM, BN, false, MT, NT, NT, NW * 64, TileRows, S3>(acc, p, smem, M, (long)tile_id * TBM, co0, wm, wn, r, h, tid, rows);
        return;
    }
    // ---- prologue: whole patch of chunk 0, weights of the first NBW - 1 stages
#pragma unroll
    for (int i = 0; i < PI; ++i) issue_patch(0, i);
#pragma unroll
    for (int s = 0; s < NBW - 1; ++s)
        if (s < n_stage) issue_weights(s / 9, s % 9, s);

    // stage = (chunk, tap).  Issue order inside a stage: [patch piece of chunk + 1 (taps 0..PI-1)] [weights of stage
    // s + NBW - 1].  At the top of stage s the DMA instructions younger than the weights of stage s are those of the
    // NBW - 2 later weight stages plus the patch pieces issued with them; vmcnt completes in order.
    int wslot = 0;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const bool more = chunk + 1 < n_chunks;
#pragma unroll UNROLL_TAPS ? 9 : 1
        for (int tap = 0; tap < 9; ++tap) {
            const int s = chunk * 9 + tap;
            // DMA instructions younger than the weights of stage s (issued in stages s-NBW+2 .. s-1: the weights of
            // stages s+1 .. s+NBW-2 and the patch pieces that went with them) may still be in flight
            int young = 0;
#pragma unroll
            for (int d = 1; d <= NBW - 2; ++d) {
                if (s + d < n_stage) young += WI;
                const int t = s - d;                                   // stage that issued them
                if (t >= 0) {
                    const int tt = t % 9, tc = t / 9;
                    if (tt < PI && tc + 1 < n_chunks) young += 1;
                }
            }
            if (!(PATCH_ABL & 9)) switch (young) {
#define VM_CASE(N_) case N_: __builtin_amdgcn_s_waitcnt(((N_) & 0xF) | 0x0F70); break;
                VM_CASE(1) VM_CASE(2) VM_CASE(3) VM_CASE(4) VM_CASE(5) VM_CASE(6) VM_CASE(7) VM_CASE(8) VM_CASE(9) VM_CASE(10)
#undef VM_CASE
                default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
            }
            if (!(PATCH_ABL & 4)) __builtin_amdgcn_s_barrier();
            auto issue_next = [&]() {
                if (tap < PI && more && !(PATCH_ABL & 2)) issue_patch(chunk + 1, tap);
                const int s2 = s + NBW - 1;
                if (s2 < n_stage && !(PATCH_ABL & 1)) {
                    const int c2 = tap + NBW - 1 >= 9 ? chunk + 1 : chunk, t2 = tap + NBW - 1 >= 9 ? tap + NBW - 1 - 9 : tap + NBW - 1;
                    int ws2 = wslot + NBW - 1;
                    if (ws2 >= NBW) ws2 -= NBW;
                    issue_weights(c2, t2, ws2);
                }
            };
            // NBW == 4: the slot being refilled was last read a whole stage ago, so the DMA issue can follow this stage's
            // MFMAs (which then start right behind the barrier) instead of preceding them
            if (NBW < 4) issue_next();
            compute(chunk, (tap / 3) * PW + tap % 3, wring + wslot * W_BYTES);
            if (NBW >= 4) issue_next();
            wslot = wslot == NBW - 1 ? 0 : wslot + 1;
        }
    }
    __syncthreads();                                       // all MFMAs done: LDS is free for the output tile
    const long M = (long)p.N * p.H * p.W;
    const TileRows rows{((long)n * p.H + oh0) * p.W + ow0, tw_shift, p.W};
    conv_epilogue<TBM, BN, false, MT, NT, NT, NW * 64, TileRows, S3>(acc, p, smem, M, (long)tile_id * TBM, co0, wm, wn, r, h, tid, rows);
}

static int g_patch_min_wgs = 128;                          // (r4: 256 -> 128: in the two-stream step the other network's kernels fill what a 128-workgroup launch leaves idle)
static int g_patch_pair = 0;                                // 1: Cout chunks of a pixel tile share an XCD (1-D launch, see the kernel)
static int g_patch_chunk_stage = 1;                         // 32-channel chunks: chunk stages instead of the per-tap weight ring
static int g_patch_wide_s3 = 1;                             // ... also for the split-3 launches (their output tile leaves in two column halves)
static int g_patch_wide = 1;                                // 1 (r4): the 256-channel tile where it fills the chip -- 3-5 % slower than the unrolled 128 tile ALONE, but it reads the input rows once
                                                            // per 256 instead of per 128 output channels: in the two-stream step (fabric-bound as a whole) -0.9 ms; 0: r3
#ifndef SHORTK_MINW
#define SHORTK_MINW 4               // waves per SIMD the short-K tiles are compiled for (4: 128 registers, the epilogue spills ~18)
#endif
static int g_short_k_small = 1;                             // K loops of up to this many stages take the 128x128 tile at 4 waves/SIMD
static int g_short_k_single = 8;                            // K loops of up to this many stages: single-buffered 128x128 tile, 4 workgroups/CU
static int g_patch_tile512 = 2;                             // 512-pixel tiles for 32 / 64 output channels: 0 off, 2 on
static int g_patch_tile512_min_wgs = 512;
static int g_patch_tile512_launches = 0;                    // launches that took a 512-pixel tile (tests read it to see the dispatch)
static int g_patch_unroll = 1;                              // 128-channel tile: tap loop unrolled
static int g_wgrad1x1_narrow = 1;                           // 64-output-channel 1x1 weight gradients (stem patch matrix, 64 -> 64) on the LDS-DMA kernel
static int g_wgrad_round_pct = 100;                         // LDS-DMA weight gradients: workgroups aimed at, in percent of one resident round (fewer slabs = less partial traffic)
static int g_wgrad_xcd = 1;                                 // LDS-DMA weight gradients: a slab's (ci, co) tiles on one XCD (WgradArgs::xcd_slabs); 0: 3-D grid
static int g_wgrad3x3_fill = 1;                             // nine-tap weight gradients: slabs sized to one full round of resident workgroups (0: r3's split)
static int g_wgrad3x3_s2 = 1;                               // stride-2 3x3 weight gradients on the nine-tap kernel (0: the per-tap kernel, r3)

int conv_set_option(const char* key, int value) {
    if (key && !strcmp(key, "conv3x3_patch_min_workgroups")) {
        const int prev = g_patch_min_wgs;
        g_patch_min_wgs = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_wgrad1x1_narrow")) {
        const int prev = g_wgrad1x1_narrow;
        g_wgrad1x1_narrow = value;
        return prev;
    }
    if (key && !strcmp(key, "stem_fused")) {
        extern int stem_fused_option(int);
        return stem_fused_option(value);
    }
    if (key && !strcmp(key, "conv_wgrad_round_pct")) {
        const int prev = g_wgrad_round_pct;
        if (value >= 10 && value <= 400) g_wgrad_round_pct = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_wgrad_xcd")) {
        const int prev = g_wgrad_xcd;
        g_wgrad_xcd = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_dgrad_s2_merge")) {
        extern int dgrad_s2_merge_option(int);
        return dgrad_s2_merge_option(value);
    }
    if (key && !strcmp(key, "conv_wgrad3x3_fill")) {
        const int prev = g_wgrad3x3_fill;
        g_wgrad3x3_fill = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_wgrad3x3_stride2")) {
        const int prev = g_wgrad3x3_s2;
        g_wgrad3x3_s2 = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_short_k_small_tile")) {
        const int prev = g_short_k_small;
        g_short_k_small = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_short_k_single_buffer")) {
        const int prev = g_short_k_single;
        g_short_k_single = value;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_unroll")) {
        const int prev = g_patch_unroll;
        g_patch_unroll = value;
        return prev;
    }
    if (key && !strcmp(key, "conv_linear_prologue")) {
        const int prev = g_glds_lin;
        g_glds_lin = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "conv_xcd_pair")) {
        const int prev = g_glds_pair;
        g_glds_pair = value;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_xcd_pair")) {
        const int prev = g_patch_pair;
        g_patch_pair = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_chunk_stage")) {
        const int prev = g_patch_chunk_stage;
        g_patch_chunk_stage = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_tile512")) {
        const int prev = g_patch_tile512;
        g_patch_tile512 = value;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_tile512_launches")) {   // returns the counter, then sets it to `value`
        const int prev = g_patch_tile512_launches;
        g_patch_tile512_launches = value;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_tile512_min_workgroups")) {
        const int prev = g_patch_tile512_min_wgs;
        g_patch_tile512_min_wgs = value;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_wide_tile_s3")) {
        const int prev = g_patch_wide_s3;
        g_patch_wide_s3 = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "conv3x3_patch_wide_tile")) {
        const int prev = g_patch_wide;
        g_patch_wide = value ? 1 : 0;
        return prev;
    }
    return -1;
}

static bool conv3x3_patch_ok(const ConvArgs& a) {
    if (a.omap || a.pad_w != a.pad) return false;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.up != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    const int ck = a.Cin % 64 == 0 && (a.C1 == a.Cin || a.C1 % 64 == 0) ? 64 : 32;
    if (a.Cin % ck || (a.C1 != a.Cin && a.C1 % ck) || (a.Cout % 128 && a.Cout != 64 && a.Cout != 32)) return false;
    const int tw = (a.W % 32 == 0) ? 32 : 16, th = 256 / tw;
    if (a.W % tw || a.H % th) return false;
    const int bn = a.Cout % 128 == 0 ? 128 : a.Cout;
    return (long)a.N * (a.H / th) * (a.W / tw) * (a.Cout / bn) >= g_patch_min_wgs;   // at least one workgroup per CU
}

template <int BN, int NBW, bool UNROLL_TAPS, int CK = 64, int TBM = 256>
static void launch_patch_t(const ConvArgs& a, hipStream_t st) {
    constexpr int ROWB = CK * 2, RPI = 1024 / ROWB, PI = ((TBM == 256 ? 344 : 616) + RPI * 8 - 1) / (RPI * 8);
    constexpr int W_ROWS = (BN * ROWB >= 8 * 1024) ? BN : 8 * 1024 / ROWB;
    size_t lds = 2 * (size_t)PI * 8 * 1024 + (size_t)NBW * W_ROWS * ROWB;
    if (NBW == 0) {                                          // chunk stages: one or two (patch + nine taps of weights) buffers
        const int nb = a.Cin / CK > 1 ? 2 : 1;
        lds = (size_t)nb * ((size_t)PI * 8 * 1024 + 9 * (size_t)BN * ROWB);
    }
    size_t out_tile = (size_t)TBM * (BN + 8) * 2 * (a.out_s3 ? 2 : 1);
    if (a.out_s3 && out_tile > 160u * 1024u) out_tile = (size_t)TBM * (BN / 2 + 8) * 4;      // (conv_epilogue: split-3 tiles this large leave in two column halves)
    if (out_tile > lds) lds = out_tile;
    const int tw = (a.W % 32 == 0) ? 32 : 16, th = TBM / tw;
    const long tiles = (long)a.N * (a.H / th) * (a.W / tw);
    const int chunks = a.Cout / BN;
    if (g_patch_pair && chunks > 1) {
        ConvArgs b = a;
        b.pair_chunks = chunks;
        b.pair_tiles = (int)tiles;
        const dim3 grid1((unsigned)((tiles + 7) / 8 * 8 * chunks));
        if (a.out_s3) hipLaunchKernelGGL((conv3x3_patch_kernel<BN, NBW, UNROLL_TAPS, CK, true, TBM>), grid1, dim3(512), lds, st, b);
        else hipLaunchKernelGGL((conv3x3_patch_kernel<BN, NBW, UNROLL_TAPS, CK, false, TBM>), grid1, dim3(512), lds, st, b);
        return;
    }
    dim3 grid((unsigned)tiles, (unsigned)chunks);
    if (a.out_s3) hipLaunchKernelGGL((conv3x3_patch_kernel<BN, NBW, UNROLL_TAPS, CK, true, TBM>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<BN, NBW, UNROLL_TAPS, CK, false, TBM>), grid, dim3(512), lds, st, a);
}

// 512-pixel tiles: geometry (16 x 32 or 32 x 16 pixel tiles must divide the image), LDS of the split-3 output tile, enough workgroups
static bool conv3x3_tile512_ok(const ConvArgs& a, int bn) {
    if (a.Cin % 32 || (a.C1 != a.Cin && a.C1 % 32)) return false;
    const int tw = (a.W % 32 == 0) ? 32 : 16, th = 512 / tw;
    if (a.W % tw || a.H % th) return false;
    if (a.out_s3 && (size_t)512 * (bn + 8) * 2 * 2 > 160 * 1024) return false;
    return (long)a.N * (a.H / th) * (a.W / tw) * (a.Cout / bn) >= g_patch_tile512_min_wgs;
}

// ------------------------------------------------------------------------------------
// weight packing: nn.Conv2d weight [Cout][Cin][KH][KW] f32 -> [Cout][KH][KW][Cin] bf16 hi (+ lo)
//   transpose_flip: data-gradient form  [Cin][KH][KW][Cout] with taps flipped
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_pack_weights(const float* __restrict__ w, int Cout, int Cin, int KH, int KW,
                                                         int transpose_flip, unsigned short* __restrict__ hi,
                                                         unsigned short* __restrict__ lo) {
    // destination [R][KH][KW][Cc_p]: R rows, Cc contraction channels padded to a multiple of 32 with zeros
    const int R = transpose_flip ? Cin : Cout, Cc = transpose_flip ? Cout : Cin;
    const int Cp = (Cc + 31) / 32 * 32;
    const long total = (long)R * KH * KW * Cp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        long t = i / Cp;
        const int kw = (int)(t % KW);
        t /= KW;
        const int kh = (int)(t % KH);
        const int rr = (int)(t / KH);
        float v = 0.0f;
        if (c < Cc) {
            if (transpose_flip) v = w[(((long)c * Cin + rr) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
            else v = w[(((long)rr * Cin + c) * KH + kh) * KW + kw];
        }
        const __bf16 bh = (__bf16)v;
        hi[i] = __builtin_bit_cast(unsigned short, bh);
        if (lo) {
            const __bf16 bl = (__bf16)(v - (float)bh);
            lo[i] = __builtin_bit_cast(unsigned short, bl);
        }
    }
}

// split-3 image for activations stored as [hi | lo | hi] (per concat segment): [Cout][KH][KW][3 * Cin] bf16 with the channel
// order [w_hi(seg) | w_hi(seg) | w_lo(seg)] for seg = the first C1 channels, then the remaining Cin - C1, so that the plain bf16
// contraction over the 3 * Cin channels is  x_hi w_hi + x_lo w_hi + x_hi w_lo  (the three products of the precise mode)
__global__ __launch_bounds__(256) void conv_pack_weights_s3(const float* __restrict__ w, int Cout, int Cin, int C1, int KH, int KW,
                                                            unsigned short* __restrict__ out) {
    const int C3 = 3 * Cin;
    const long total = (long)Cout * KH * KW * C3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int j = (int)(i % C3);
        long t = i / C3;
        const int kw = (int)(t % KW);
        t /= KW;
        const int kh = (int)(t % KH);
        const int co = (int)(t / KH);
        const bool second = j >= 3 * C1;
        const int cs = second ? Cin - C1 : C1;
        if (second) j -= 3 * C1;
        const int part = j / cs, c = (second ? C1 : 0) + (j - part * cs);
        const float v = w[(((long)co * Cin + c) * KH + kh) * KW + kw];
        const __bf16 bh = (__bf16)v;
        const __bf16 o = part == 2 ? (__bf16)(v - (float)bh) : bh;
        out[i] = __builtin_bit_cast(unsigned short, o);
    }
}

// Data gradient of a STRIDE-2 convolution without the dilated grid.  Forward: y[oh, ow] = sum_{kh, kw} Xp[2 oh + kh, 2 ow + kw] w[kh, kw]
// (Xp: the padded input).  An input row ip receives only the taps with (ip - kh) even:
//   3 taps:  ip = 2 i   <- kh = 2 from output row i - 1, kh = 0 from row i      (a 2-tap window starting at i - 1: "pad" 1)
//            ip = 2 i + 1 <- kh = 1 from row i                                    (1 tap, pad 0)
//   1 tap (1x1 layers):  ip = 2 i <- row i;  odd rows receive nothing (the output is zero-filled first).
// So the gradient is 4 (or 1) small stride-1 convolutions over gy -- one per parity class (ph, pw) of the output pixel -- whose
// results land on pixels (2 i + ph, 2 j + pw) (ConvArgs::omap): 9 taps of MFMA work per 4 output pixels instead of 36.
// Sub-images, data-gradient form [Cin][KH'][KW'][Cout padded to 32] (contraction over Cout), in class order (0,0), (0,1), (1,0), (1,1):
// class parity 0 holds the forward taps (2, 0) in window order, parity 1 the tap (1).
__global__ __launch_bounds__(256) void conv_pack_weights_s2(const float* __restrict__ w, int Cout, int Cin, int K,
                                                            unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int Cp = (Cout + 31) / 32 * 32;
    const long per_tap = (long)Cp;                          // elements per (ci, tap)
    const int taps_total = K * K;                           // 9 (or 1): the classes partition the taps
    const long total = (long)Cin * taps_total * per_tap;
    // class (ph, pw): window sizes nh = (K == 3 ? (ph ? 1 : 2) : 1), same for nw; element order inside a class: [ci][th][tw][co]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        int ph = 0, pw = 0, nh = 1, nw = 1;
        long base = 0;
        if (K == 3) {
            for (int cls = 0; cls < 4; ++cls) {
                ph = cls >> 1;
                pw = cls & 1;
                nh = ph ? 1 : 2;
                nw = pw ? 1 : 2;
                const long sz = (long)Cin * nh * nw * per_tap;
                if (t < base + sz) break;
                base += sz;
            }
        }
        t -= base;
        const int co = (int)(t % Cp);
        t /= Cp;
        const int tw = (int)(t % nw);
        t /= nw;
        const int th = (int)(t % nh);
        const int ci = (int)(t / nh);
        const int kh = K == 3 ? (ph ? 1 : (th == 0 ? 2 : 0)) : 0;
        const int kw = K == 3 ? (pw ? 1 : (tw == 0 ? 2 : 0)) : 0;
        float v = 0.0f;
        if (co < Cout) v = w[(((long)co * Cin + ci) * K + kh) * K + kw];
        const __bf16 bh = (__bf16)v;
        hi[i] = __builtin_bit_cast(unsigned short, bh);
        if (lo) {
            const __bf16 bl = (__bf16)(v - (float)bh);
            lo[i] = __builtin_bit_cast(unsigned short, bl);
        }
    }
}

// ------------------------------------------------------------------------------------
// host-side launch
// ------------------------------------------------------------------------------------
template <int BN, bool PRECISE, int BK>
static void launch_t(const ConvArgs& a, hipStream_t st) {
    constexpr int BKP = BK + 8;
    size_t lds = (size_t)(BM + BN) * BKP * 2 * (PRECISE ? 2 : 1) * 2;                 // two stages
    const size_t out_tile = (size_t)BM * (BN + (PRECISE ? 4 : 8)) * (PRECISE ? 4 : 2);  // epilogue staging tile
    if (out_tile > lds) lds = out_tile;
    const long M = (long)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    if constexpr (!PRECISE) {
        if (a.out_s3) {
            const size_t s3_tile = (size_t)BM * (BN + 8) * 2 * 2;
            if (s3_tile > lds) lds = s3_tile;
            hipLaunchKernelGGL((conv_igemm_kernel<BN, PRECISE, BK, true>), grid, dim3(256), lds, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BN, PRECISE, BK>), grid, dim3(256), lds, st, a);
}

// ---- optional per-launch timing of the convolution kernels (bench.py's roofline_conv leg): event pairs on the launch stream
struct ConvProfile {
    bool enabled = false;
    int capacity = 0, count = 0;
    hipEvent_t* ev = nullptr;
    double* flops = nullptr;
    int* kind = nullptr;                                    // KH * 100 + precision tag (0 bf16, 1 precise, 2 split-3)
    int* shape = nullptr;                                   // [4]: output pixels / 1024, Cin (logical), Cout, stride * 10 + up
};
static ConvProfile g_cprof;

hipError_t conv_profile_begin(int capacity) {
    conv_profile_release();
    g_cprof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * (size_t)capacity);
    g_cprof.flops = (double*)malloc(sizeof(double) * (size_t)capacity);
    g_cprof.kind = (int*)malloc(sizeof(int) * (size_t)capacity);
    g_cprof.shape = (int*)malloc(sizeof(int) * 4 * (size_t)capacity);
    if (!g_cprof.ev || !g_cprof.flops || !g_cprof.kind || !g_cprof.shape) return hipErrorOutOfMemory;
    for (int i = 0; i < 2 * capacity; ++i) {
        hipError_t rc = hipEventCreate(&g_cprof.ev[i]);
        if (rc != hipSuccess) return rc;
    }
    g_cprof.capacity = capacity;
    g_cprof.count = 0;
    g_cprof.enabled = true;
    return hipSuccess;
}

void conv_profile_release() {
    if (g_cprof.ev)
        for (int i = 0; i < 2 * g_cprof.capacity; ++i) (void)hipEventDestroy(g_cprof.ev[i]);
    free(g_cprof.ev);
    free(g_cprof.flops);
    free(g_cprof.kind);
    free(g_cprof.shape);
    g_cprof = ConvProfile{};
}

int conv_profile_collect(int max_records, double* flops, int* kind, float* ms, int* shape) {
    g_cprof.enabled = false;
    int out = 0;
    for (int i = 0; i < g_cprof.count && out < max_records; ++i) {
        if (hipEventSynchronize(g_cprof.ev[2 * i + 1]) != hipSuccess) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_cprof.ev[2 * i], g_cprof.ev[2 * i + 1]) != hipSuccess) break;
        flops[out] = g_cprof.flops[i];
        kind[out] = g_cprof.kind[i];
        if (shape)
            for (int e = 0; e < 4; ++e) shape[4 * out + e] = g_cprof.shape[4 * i + e];
        ms[out] = t;
        ++out;
    }
    conv_profile_release();
    return out;
}

static hipError_t launch_conv_impl(const ConvArgs& a, int precise, hipStream_t st);

hipError_t launch_conv(const ConvArgs& a_in, int precise, hipStream_t st) {
    ConvArgs a = a_in;
    if (a.pad_w < 0) a.pad_w = a.pad;
    const bool rec = g_cprof.enabled && g_cprof.count < g_cprof.capacity;
    const int slot = g_cprof.count;
    if (rec) {
        // algorithmic flops: 2 * taps * Cin * Cout per output pixel (a split-3 launch contracts over 3 x the logical channels:
        // its algorithmic work is the logical convolution's; `up` launches are data gradients of strided layers: the
        // up-sampled grid's zero rows are not work, so count the forward layer's pixels = input pixels of this launch)
        const double cin = a.out_s3 ? a.Cin / 3.0 : (double)a.Cin;
        const double px = a.up == 2 ? (double)a.N * a.H * a.W : (double)a.N * a.Ho * a.Wo;
        g_cprof.flops[slot] = 2.0 * a.KH * a.KW * cin * a.Cout * px;
        g_cprof.kind[slot] = (a.prof_k ? a.prof_k : a.KH) * 100 + (a.out_s3 ? 2 : (precise ? 1 : 0));
        g_cprof.shape[4 * slot + 0] = (int)(((long)a.N * a.Ho * a.Wo) >> 10);
        g_cprof.shape[4 * slot + 1] = (int)cin;
        g_cprof.shape[4 * slot + 2] = a.Cout;
        g_cprof.shape[4 * slot + 3] = a.stride * 10 + a.up + (a.omap ? 2 : 0);          // ..3: a parity class of a stride-2 data gradient
        ++g_cprof.count;
        (void)hipEventRecord(g_cprof.ev[2 * slot], st);
    }
    const hipError_t e = launch_conv_impl(a, precise, st);
    if (rec) (void)hipEventRecord(g_cprof.ev[2 * slot + 1], st);
    return e;
}

static hipError_t launch_conv_impl(const ConvArgs& a, int precise, hipStream_t st) {
    const int bn = a.Cout >= 128 ? 128 : (a.Cout >= 64 ? 64 : 32);
    const bool k64 = !precise && a.Cin % 64 == 0 && (a.C1 == a.Cin || a.C1 % 64 == 0);
    if (a.ring) {                                            // border ring of a full correlation: few rows, generic LDS-DMA kernel
        if (!k64) return hipErrorInvalidValue;
        if (bn == 128) launch_glds_t<128, 128, 4, 2>(a, st);
        else if (bn == 64) launch_glds_t<128, 64, 4, 3>(a, st);
        else launch_glds_t<128, 32, 4, 3>(a, st);
        return hipGetLastError();
    }
    if (precise) {
        if (bn == 128) launch_t<128, true, 32>(a, st);
        else if (bn == 64) launch_t<64, true, 32>(a, st);
        else launch_t<32, true, 32>(a, st);
    } else if ((g_patch_tile512 & 2) && conv3x3_patch_ok(a) && (a.Cout == 32 || (a.Cout == 64 && a.Cin >= 64)) && conv3x3_tile512_ok(a, a.Cout)) {
        // 32 / 64 output channels: the 256-pixel tile gives every wave ONE 32-pixel row block (two LDS fragment reads per MFMA: the
        // LDS port is the limit); 512 pixels make it two (1.5 reads per MFMA) and halve the weight DMA per MFMA.  Measured (B = 32):
        // 192 -> 32 at 256^2 469 -> 337 us, 64 -> 64 at 128^2 83 -> 75 us, 32 -> 32 at 256^2 97 -> 91 us; single-chunk 32 -> 64 is
        // 14 % slower and stays on the 256-pixel tile; the 128-channel tile was 10-17 % slower at 512 pixels (and spilled in its epilogue):
        // not instantiated
        ++g_patch_tile512_launches;
        if (a.Cout == 64) launch_patch_t<64, 0, true, 32, 512>(a, st);
        else launch_patch_t<32, 0, true, 32, 512>(a, st);
    } else if (!k64 && conv3x3_patch_ok(a)) {                // 32-channel chunks (the last decoder level and its gradients)
        // taps of one or two K steps: all nine taps' weights ride with the patch (chunk stages) where that fits LDS
        if (g_patch_chunk_stage && a.Cout == 64) launch_patch_t<64, 0, true, 32>(a, st);
        else if (g_patch_chunk_stage && a.Cout == 32) launch_patch_t<32, 0, true, 32>(a, st);
        else if (g_patch_chunk_stage && a.Cin == 32) launch_patch_t<128, 0, true, 32>(a, st);
        else if (a.Cout == 64) launch_patch_t<64, 3, true, 32>(a, st);
        else if (a.Cout == 32) launch_patch_t<32, 3, true, 32>(a, st);
        else launch_patch_t<128, 3, true, 32>(a, st);
    } else if (k64 && conv3x3_patch_ok(a)) {
        const int tw = (a.W % 32 == 0) ? 32 : 16, th = 256 / tw;
        const long tiles = (long)a.N * (a.H / th) * (a.W / tw);
        // 256-wide channel tile (wave tile 128 px x 64 co: 25 % fewer LDS fragment reads per MFMA) when it still fills the chip
        if (a.Cout == 64) launch_patch_t<64, 3, true>(a, st);
        else if (a.Cout == 32) launch_patch_t<32, 3, true>(a, st);
        else if (a.Cout % 256 == 0 && g_patch_wide && (!a.out_s3 || g_patch_wide_s3) && tiles * (a.Cout / 256) >= g_patch_min_wgs)
            launch_patch_t<256, 2, false>(a, st);
        else if (g_patch_unroll == 2)
            launch_patch_t<128, 4, true>(a, st);
        else if (g_patch_unroll)
            launch_patch_t<128, 3, true>(a, st);
        else
            launch_patch_t<128, 3, false>(a, st);
    } else if (k64) {
        // L2 -> LDS operand traffic bounds this kernel (~35 B/clk/CU): prefer the largest tile that still yields
        // at least ~2 waves of workgroups over the 256 CUs
        const long M = (long)a.N * a.Ho * a.Wo;
        const long t256 = ((M + 255) / 256) * ((a.Cout + 255) / 256), t256x128 = ((M + 255) / 256) * ((a.Cout + 127) / 128);
        (void)t256;
        const int n_stage = a.KH * a.KW * (a.Cin / 64);
        if (n_stage <= g_short_k_small && bn == 128) launch_glds_t<128, 128, 4, 2, SHORTK_MINW>(a, st);
        else if (n_stage <= g_short_k_single && bn == 128) launch_glds_t<128, 128, 4, 1, SHORTK_MINW>(a, st);   // epilogue-bound: more, independent workgroups per CU
        else if (a.Cout % 128 == 0 && t256x128 >= 512) launch_glds_t<256, 128, 8, 3>(a, st);
        else if (bn == 128) launch_glds_t<128, 128, 4, 2>(a, st);
        else if (bn == 64) launch_glds_t<128, 64, 4, 3>(a, st);
        else launch_glds_t<128, 32, 4, 3>(a, st);
    } else {
        if (bn == 128) launch_t<128, false, 32>(a, st);
        else if (bn == 64) launch_t<64, false, 32>(a, st);
        else launch_t<32, false, 32>(a, st);
    }
    return hipGetLastError();
}

static int g_stem_fused = 1;                                // the stem without its patch matrix (0: callers keep the patch-matrix path)
int stem_fused_option(int value) {
    const int prev = g_stem_fused;
    if (value >= 0) g_stem_fused = value ? 1 : 0;
    return prev;
}
// x: image [N][H][W][3] f32; w_img: [64][176] bf16, columns kh * 24 + kw * 3 + ci (s3: [64][2][176] = w_hi | w_lo); y [N][Ho][Wo][64] bf16 (s3: [..][128] = hi | lo); either
// stat_partial (raw y + BatchNorm partials) or ep_scale / ep_shift (fused affine [+ ReLU]; required for s3)
hipError_t launch_stem7_fused(const float* x, const unsigned short* w_img, void* y, float* stat_partial, const float* ep_scale, const float* ep_shift,
                              int relu, int N, int H, int W, int reflect, int s3, hipStream_t st) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if (!g_stem_fused || Wo % 128 || H < 4 || W < 4 || (long)H * W * 3 >= (1L << 31) || (long)N * Ho * (Wo / 128) >= (1L << 31) || (s3 && !ep_scale))
        return hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.x2 = nullptr; a.C1 = 160; a.w_hi = w_img; a.w_lo = nullptr; a.y = y; a.stat_partial = stat_partial;
    a.N = N; a.H = H; a.W = W; a.Cin = 160; a.Ho = Ho; a.Wo = Wo; a.Cout = 64; a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0; a.reflect = reflect; a.up = 1;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_res = nullptr; a.ep_relu = relu; a.out_s3 = s3 ? 1 : 0;
    const bool rec = g_cprof.enabled && g_cprof.count < g_cprof.capacity;
    const int slot = g_cprof.count;
    if (rec) {
        g_cprof.flops[slot] = 2.0 * 160 * 64 * (double)N * Ho * Wo;
        g_cprof.kind[slot] = 100 + (s3 ? 2 : 0);
        g_cprof.shape[4 * slot + 0] = (int)(((long)N * Ho * Wo) >> 10);
        g_cprof.shape[4 * slot + 1] = 160;
        g_cprof.shape[4 * slot + 2] = 64;
        g_cprof.shape[4 * slot + 3] = 11;
        ++g_cprof.count;
        (void)hipEventRecord(g_cprof.ev[2 * slot], st);
    }
    const unsigned grid = (unsigned)((long)N * Ho * (Wo / 128));
    const size_t strip_b = (size_t)(s3 ? 2 : 1) * ((7 * 790 * 2 + 15) / 16 * 16), w_b = (size_t)64 * 184 * 2 * (s3 ? 2 : 1);
    size_t lds = strip_b + w_b;
    const size_t out_tile = (size_t)128 * (64 + 8) * 2 * (s3 ? 2 : 1);
    if (out_tile > lds) lds = out_tile;
    if (s3) hipLaunchKernelGGL((stem7_fused_kernel<true>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stem7_fused_kernel<false>), dim3(grid), dim3(256), lds, st, a);
    if (rec) (void)hipEventRecord(g_cprof.ev[2 * slot + 1], st);
    return hipGetLastError();
}

size_t packed_elems(int Cout, int Cin, int KH, int KW, int transpose_flip) {
    const int R = transpose_flip ? Cin : Cout, Cc = transpose_flip ? Cout : Cin;
    return (size_t)R * KH * KW * ((Cc + 31) / 32 * 32);
}

// All kernel-side images of one K x K (K = 1 or 3) weight in ONE launch: a workgroup transposes a 32 (co) x 32 (ci) x K^2 tile
// through LDS (the fp32 weight is read once, in whole contiguous runs) and writes 64-byte runs of
//   fwd [Cout][K][K][Cin_p]            (bf16 hi; the forward image of conv_pack_weights)
//   tr  [Cin][K][K][Cout_p], taps flipped (the data-gradient image)
//   s3  [Cout][K][K][3 Cin]  = [w_hi | w_hi | w_lo] per concat segment (split-3, conv_pack_weights_s3)
// -- each optional.  Replaces three launches per layer and step (the weights change with every optimiser step).
template <int K>                                             // compile-time tap count: the index arithmetic is divisions by K * K
__global__ __launch_bounds__(256) void conv_pack_all_kernel(const float* __restrict__ w, int Cout, int Cin, int C1,
                                                            unsigned short* __restrict__ fwd, unsigned short* __restrict__ tr,
                                                            unsigned short* __restrict__ s3) {
    constexpr int KK = K * K;
    __shared__ float tile[32][32 * KK + 1];
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int Cin_p = (Cin + 31) / 32 * 32, Cout_p = (Cout + 31) / 32 * 32;
    constexpr int run = 32 * KK;                            // floats per co row of the tile (contiguous in w when ci0 + 32 <= Cin)
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int cr = i / run, e = i - cr * run;           // e = ci_local * KK + tap
        const int co = co0 + cr, ci = ci0 + e / KK;
        tile[cr][e] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci0) * KK + e] : 0.0f;
    }
    __syncthreads();
    // forward / split-3 images: (co, tap) rows, 32 ci contiguous = four 16-byte stores (8 bf16) per row
    auto pack8 = [](const float (&v)[8], u32x4& hi, u32x4* lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack2(v[2 * e], v[2 * e + 1]);
            if (lo) (*lo)[e] = pack2(v[2 * e] - bf16_round(v[2 * e]), v[2 * e + 1] - bf16_round(v[2 * e + 1]));
        }
    };
    for (int i = threadIdx.x; i < 32 * KK * 4; i += 256) {
        const int c8 = i & 3, rt = i >> 2;                  // 8-channel chunk of the ci run, (co_local, tap)
        const int cr = rt / KK, tap = rt - cr * KK;
        const int co = co0 + cr, ci = ci0 + c8 * 8;
        if (co >= Cout || ci >= Cin_p) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[cr][(c8 * 8 + e) * KK + tap];
        u32x4 h4, l4;
        pack8(v, h4, &l4);
        if (fwd) *reinterpret_cast<u32x4*>(fwd + ((long)co * KK + tap) * Cin_p + ci) = h4;
        if (s3 && ci < Cin) {                               // Cin % 32 == 0 here: whole chunks
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
        for (int i = threadIdx.x; i < 32 * KK * 4; i += 256) {
            const int c8 = i & 3, rt = i >> 2;              // 8-channel chunk of the co run, (ci_local, tap)
            const int cl = rt / KK, tap = rt - cl * KK;
            const int ci = ci0 + cl, co = co0 + c8 * 8;
            if (ci >= Cin || co >= Cout_p) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[c8 * 8 + e][cl * KK + tap];     // rows co >= Cout of the tile are zero
            u32x4 h4;
            pack8(v, h4, nullptr);
            *reinterpret_cast<u32x4*>(tr + ((long)ci * KK + (KK - 1 - tap)) * Cout_p + co) = h4;
        }
}

hipError_t launch_pack_all(const float* w, int Cout, int Cin, int K, int C1, unsigned short* fwd, unsigned short* tr, unsigned short* s3,
                           hipStream_t st) {
    dim3 grid((unsigned)((Cin + 31) / 32), (unsigned)((Cout + 31) / 32));
    if (K == 3) hipLaunchKernelGGL(conv_pack_all_kernel<3>, grid, dim3(256), 0, st, w, Cout, Cin, C1, fwd, tr, s3);
    else hipLaunchKernelGGL(conv_pack_all_kernel<1>, grid, dim3(256), 0, st, w, Cout, Cin, C1, fwd, tr, s3);
    return hipGetLastError();
}

hipError_t launch_pack_weights_s2(const float* w, int Cout, int Cin, int K, unsigned short* hi, unsigned short* lo, hipStream_t st) {
    const long total = (long)Cin * K * K * ((Cout + 31) / 32 * 32);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_pack_weights_s2, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, K, hi, lo);
    return hipGetLastError();
}

// gx (n, OH, OW, Cin) = data gradient of a stride-2 K x K convolution (K = 3: padded-input grid OH = H + 2 with the reflect /
// zero fold left to the caller; K = 1: the input grid itself) from gy (n, Ho, Wo, Cout) and conv_pack_weights_s2's sub-images
// per-channel unit scale / zero shift for launches that use the fused epilogue only for its residual add
constexpr int UNIT_AFFINE_MAX = 4096;
struct UnitAffine {
    float one[UNIT_AFFINE_MAX];
    float zero[UNIT_AFFINE_MAX];
    constexpr UnitAffine() : one(), zero() {
        for (int i = 0; i < UNIT_AFFINE_MAX; ++i) one[i] = 1.0f;
    }
};
__device__ __attribute__((used)) UnitAffine g_unit_affine{};       // not `const`: a const namespace-scope object has internal linkage and is
                                                                    // not registered with the runtime (hipGetSymbolAddress aborts on it)

// `accumulate` (K == 1 only): gx already holds a gradient of the same tensor (the OTHER consumer's contribution to a fan-in);
// the data gradient is added to it in place at the pixels it touches -- no memset, no separate add pass.
hipError_t launch_dgrad_s2(const void* gy, const unsigned short* w_hi, const unsigned short* w_lo, void* gx, int N, int Ho, int Wo,
                           int Cout, int Cin, int K, int OH, int OW, int precise, int accumulate, hipStream_t st) {
    const int Cp = (Cout + 31) / 32 * 32;
    if (accumulate && (K != 1 || Cin > UNIT_AFFINE_MAX)) return hipErrorInvalidValue;
    static const UnitAffine* ua = nullptr;                   // one device per process (one process per GPU)
    if (accumulate && !ua) {
        hipError_t e = hipGetSymbolAddress((void**)&ua, HIP_SYMBOL(g_unit_affine));
        if (e != hipSuccess) return e;
    }
    if (K == 1 && !accumulate) {
        hipError_t e = hipMemsetAsync(gx, 0, (size_t)N * OH * OW * Cin * (precise ? 4 : 2), st);
        if (e != hipSuccess) return e;
    }
    size_t off = 0;
    for (int cls = 0; cls < (K == 3 ? 4 : 1); ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        const int nh = K == 3 ? (ph ? 1 : 2) : 1, nw = K == 3 ? (pw ? 1 : 2) : 1;
        ConvArgs a;
        a.x = gy; a.x2 = nullptr; a.C1 = Cout;
        a.w_hi = w_hi + off; a.w_lo = w_lo ? w_lo + off : nullptr;
        a.y = gx; a.stat_partial = nullptr;
        a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin; a.KH = nh; a.KW = nw;
        a.Ho = (OH - ph + 1) / 2; a.Wo = (OW - pw + 1) / 2;              // output pixels of this parity class
        a.stride = 1; a.pad = K == 3 ? (ph ? 0 : 1) : 0; a.pad_w = K == 3 ? (pw ? 0 : 1) : 0; a.reflect = 0; a.up = 1;
        a.ep_scale = nullptr; a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_relu = 0;
        if (accumulate) {
            a.ep_scale = ua->one; a.ep_shift = ua->zero; a.ep_res = gx; a.ep_res_out = 1;
        }
        a.omap = 1; a.omap_h = OH; a.omap_w = OW; a.omap_ph = ph; a.omap_pw = pw;
        a.prof_k = K;
        if ((long)a.N * a.Ho * a.Wo >= (1L << 31)) return hipErrorInvalidValue;
        hipError_t e = launch_conv(a, precise, st);
        if (e != hipSuccess) return e;
        off += (size_t)Cin * nh * nw * Cp;
    }
    return hipGetLastError();
}

static int g_dgrad_s2_merge = 1;                            // stride-2 3x3 data gradients: one launch, unpadded output (0: four launches + fold / crop)
int dgrad_s2_merge_option(int value) {
    const int prev = g_dgrad_s2_merge;
    if (value >= 0) g_dgrad_s2_merge = value ? 1 : 0;
    return prev;
}

long dgrad_s2_fold_rows(int N, int H, int W, int reflect) {
    return (long)N * H * W + (reflect ? (long)N * (W + 1 + H) : 0) + 1;
}

// x row 1 += padded row 0 (columns 1..W -> x columns 0..W-1), x column 1 += padded column 0 (rows 1..H -> x rows 0..H-1),
// x (1, 1) += the padded corner: the reflect-padding fold of a stride-2 layer (padded row H+1 / column W+1 carry no gradient:
// H, W even).  ring [N][W + 1 + H][C] as out_row writes it; fp32 sums, one rounding.  8 channels per thread.
__global__ __launch_bounds__(256) void reflect_s2_ring_add_kernel(const __bf16* __restrict__ ring, int N, int H, int W, int C, __bf16* __restrict__ gx) {
    const int cv = C / 8, rl = W + 1 + H;
    const long total = (long)N * (W + H - 1) * cv;         // row 1 (W pixels) + column 1 without (1, 1)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        long t = i / cv;
        const int q = (int)(t % (W + H - 1)), n = (int)(t / (W + H - 1));
        const __bf16* rn = ring + (long)n * rl * C;
        int xr, xc;
        float v[8];
        if (q < W) {                                        // x (1, q)
            xr = 1, xc = q;
            const u32x4 a = *reinterpret_cast<const u32x4*>(rn + (long)(q + 1) * C + c);
            unpack8(a, v);
            if (q == 1) {                                   // + padded (2, 0) + padded (0, 0)
                float w8[8];
                unpack8(*reinterpret_cast<const u32x4*>(rn + (long)(W + 2) * C + c), w8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += w8[e];
                unpack8(*reinterpret_cast<const u32x4*>(rn + c), w8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += w8[e];
            }
        } else {                                            // x (r, 1), r != 1: padded (r + 1, 0) = ring slot W + r + 1
            int r = q - W;
            if (r >= 1) ++r;
            xr = r, xc = 1;
            unpack8(*reinterpret_cast<const u32x4*>(rn + (long)(W + r + 1) * C + c), v);
        }
        __bf16* dst = gx + (((long)n * H + xr) * W + xc) * C + c;
        float g[8];
        unpack8(*reinterpret_cast<const u32x4*>(dst), g);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2(g[2 * e] + v[2 * e], g[2 * e + 1] + v[2 * e + 1]);
        *reinterpret_cast<u32x4*>(dst) = o;
    }
}

hipError_t launch_dgrad_s2_fold(const void* gy, const unsigned short* w_hi, void* gx, int N, int Ho, int Wo, int Cout, int Cin, int H, int W,
                                int reflect, hipStream_t st) {
    // (Cout: channels of gy = the contraction; Cin: channels of gx = the GEMM's output channels)
    if (!g_dgrad_s2_merge || H != 2 * Ho || W != 2 * Wo || Cout % 64 || Cin % 8 || Cin < 64 || H < 4 || W < 4) return hipErrorInvalidValue;
    const long body = dgrad_s2_fold_rows(N, H, W, reflect);
    if ((long)N * (Ho + 1) * (Wo + 1) >= (1L << 31) || body >= (1L << 31)) return hipErrorInvalidValue;
    const int Cp = (Cout + 31) / 32 * 32;
    ConvArgs a;
    a.x = gy; a.x2 = nullptr; a.C1 = Cout; a.w_hi = w_hi; a.w_lo = nullptr; a.y = gx; a.stat_partial = nullptr;
    a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin; a.stride = 1; a.reflect = 0; a.up = 1;
    a.ep_scale = nullptr; a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_relu = 0;
    a.omap = 1; a.omap_h = H; a.omap_w = W; a.omap_fold = reflect ? 1 : 2; a.prof_k = 3;
    const int bn = Cin >= 128 ? 128 : 64;
    size_t off = 0;
    unsigned tiles = 0;
    a.n_cls = 4;
    for (int cls = 0; cls < 4; ++cls) {                     // the padded grid's parity classes, as launch_dgrad_s2 (OH = H + 2)
        const int ph = cls >> 1, pw = cls & 1;
        ConvArgs::Cls& k = a.cls[cls];
        k.KH = ph ? 1 : 2, k.KW = pw ? 1 : 2, k.pad = ph ? 0 : 1, k.pad_w = pw ? 0 : 1, k.ph = ph, k.pw = pw;
        k.Ho = (H + 2 - ph + 1) / 2, k.Wo = (W + 2 - pw + 1) / 2;
        k.w_off = (unsigned)off;
        tiles += (unsigned)(((long)N * k.Ho * k.Wo + 127) / 128);
        k.tile_end = tiles;
        off += (size_t)Cin * k.KH * k.KW * Cp;
    }
    a.KH = 2, a.KW = 2, a.pad = 1, a.pad_w = 1, a.Ho = a.cls[0].Ho, a.Wo = a.cls[0].Wo;      // (class 0; the kernel substitutes)
    const bool rec = g_cprof.enabled && g_cprof.count < g_cprof.capacity;
    const int slot = g_cprof.count;
    if (rec) {
        g_cprof.flops[slot] = 2.0 * 9 * (double)Cout * Cin * (double)N * Ho * Wo;
        g_cprof.kind[slot] = 300;
        g_cprof.shape[4 * slot + 0] = (int)(((long)N * H * W) >> 10);
        g_cprof.shape[4 * slot + 1] = Cout;
        g_cprof.shape[4 * slot + 2] = Cin;
        g_cprof.shape[4 * slot + 3] = 13;
        ++g_cprof.count;
        (void)hipEventRecord(g_cprof.ev[2 * slot], st);
    }
    const dim3 grid(tiles, (unsigned)((Cin + bn - 1) / bn));
    if (bn == 128 && 4 * (Cout / 64) <= g_short_k_single) {   // K loops of a few stages: epilogue-bound -- single buffer, four workgroups per CU
        const size_t lds = (size_t)128 * (128 + 8) * 2;       // (the output tile is the larger of the two uses)
        hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 128, 4, 1, SHORTK_MINW, false, false, true>), grid, dim3(256), lds, st, a);
    } else if (bn == 128) {
        const size_t lds = (size_t)(128 + 128) * 64 * 2 * 2;
        hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 128, 4, 2, 1, false, false, true>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = (size_t)(128 + 64) * 64 * 2 * 3;
        hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 64, 4, 3, 1, false, false, true>), grid, dim3(256), lds, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && reflect) {
        const __bf16* ring = reinterpret_cast<const __bf16*>(gx) + (long)N * H * W * Cin;
        const long total = (long)N * (W + H - 1) * (Cin / 8);
        long blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(reflect_s2_ring_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ring, N, H, W, Cin, reinterpret_cast<__bf16*>(gx));
        e = hipGetLastError();
    }
    if (rec) (void)hipEventRecord(g_cprof.ev[2 * slot + 1], st);
    return e;
}

// ring [N][2 (W + 2) + 2 H][Cgx] = the full 3x3 correlation of gy [N][H][W][Cgy] with the tap-flipped transposed weights, evaluated
// on the border ring of the (H + 2) x (W + 2) grid only (bf16; Cgy % 64 == 0)
hipError_t launch_reflect_ring(const void* gy, const unsigned short* t_hi, void* ring, int N, int H, int W, int Cgy, int Cgx, hipStream_t st) {
    ConvArgs a;
    a.x = gy; a.x2 = nullptr; a.C1 = Cgy;
    a.w_hi = t_hi; a.w_lo = nullptr;
    a.y = ring; a.stat_partial = nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cgy; a.Cout = Cgx; a.KH = 3; a.KW = 3;
    a.ring = 1; a.ring_h = H + 2; a.ring_w = W + 2;
    a.Ho = 1; a.Wo = 2 * (W + 2) + 2 * H;
    a.stride = 1; a.pad = 2; a.pad_w = 2; a.reflect = 0; a.up = 1;
    a.ep_scale = nullptr; a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_relu = 0;
    a.prof_k = 3;
    if ((long)a.N * a.Wo >= (1L << 31)) return hipErrorInvalidValue;
    return launch_conv(a, 0, st);
}

hipError_t launch_pack_weights_s3(const float* w, int Cout, int Cin, int C1, int KH, int KW, unsigned short* out, hipStream_t st) {
    const long total = (long)Cout * KH * KW * 3 * Cin;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_pack_weights_s3, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, C1, KH, KW, out);
    return hipGetLastError();
}

hipError_t launch_pack_weights(const float* w, int Cout, int Cin, int KH, int KW, int transpose_flip, unsigned short* hi,
                               unsigned short* lo, hipStream_t st) {
    const long total = (long)packed_elems(Cout, Cin, KH, KW, transpose_flip);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_pack_weights, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, KH, KW, transpose_flip, hi,
                       lo);
    return hipGetLastError();
}


// =====================================================================================================
// Weight gradient:  dW[co][tap][ci] = sum_m GY[m][co] * A_tap[m][ci]      (m = pixel rows, the GEMM K dim)
//
// Both operands are pixel-major in HBM (NHWC), i.e. K is their SLOW dimension.  They are staged as
// [32 pixels][channels] bf16 tiles in LDS (rows padded to +64 B: conflict-free) and the MFMA fragments
// (8 consecutive pixels of one channel per lane) are fetched with the gfx950 transposing LDS read
// ds_read_b64_tr_b16 -- no explicit transpose pass.  The pixel range is split over gridDim.z slabs
// (deterministic: each slab writes its own fp32 partial, reduced in fixed order by wgrad_reduce_kernel,
// which also converts [Cout][taps][Cin] to nn.Conv2d's [Cout][Cin][KH][KW]).
// =====================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN>
struct WgradCfg {
    static constexpr int WGM = (TM >= 4 && TN == 1) ? 4 : (TM >= 2 ? 2 : 1);
    static constexpr int WGN = (TN >= 4 && TM == 1) ? 4 : ((TN >= 2 && WGM <= 2) ? 2 : 1);
    static constexpr int PM = TM / WGM, PN = TN / WGN;
};

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row_stride, int pix0, int col0, int lane) {
    // fragment for MFMA 32x32x16: lane (r = lane & 31, h = lane >> 5) gets column (col0 + r), rows pix0 + 8h .. +7
    const int g = lane >> 4;                      // 16-lane group: (g & 1) -> column block, (g >> 1) -> h
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const __bf16* a0 = tile + (pix0 + 8 * (g >> 1) + q) * row_stride + col0 + 16 * (g & 1) + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * row_stride));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int TM, int TN, bool PRECISE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    using Cfg = WgradCfg<TM, TN>;
    constexpr int BKM = 32;                               // pixels per stage
    constexpr int GS = TM * 32 + 32;                      // LDS row stride (bf16) of the GY tile (+64 B pad)
    constexpr int AS = TN * 32 + 32;
    constexpr int EPC = PRECISE ? 4 : 8;                  // elements per 16-byte chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* G_hi = reinterpret_cast<__bf16*>(smem);
    __bf16* G_lo = G_hi + (PRECISE ? BKM * GS : 0);
    __bf16* A_hi = G_lo + BKM * GS;
    __bf16* A_lo = A_hi + (PRECISE ? BKM * AS : 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
    const bool active = wave < Cfg::WGM * Cfg::WGN;

    const int ci_tiles = (p.Cin + TN * 32 - 1) / (TN * 32);
    const int tap = blockIdx.x / ci_tiles;
    const int ci0 = (blockIdx.x - tap * ci_tiles) * (TN * 32);
    const int co0 = blockIdx.y * (TM * 32);
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const long M = (long)p.N * p.Ho * p.Wo;
    const long Ma = (long)p.Na * p.Ho * p.Wo;               // rows of the first source (== M without a second one)
    long mz = (M + gridDim.z - 1) / gridDim.z;
    mz = (mz + BKM - 1) / BKM * BKM;
    const long m_begin = (long)blockIdx.z * mz;
    long m_end = m_begin + mz;
    if (m_end > M) m_end = M;

    f32x16 acc[Cfg::PM][Cfg::PN];
#pragma unroll
    for (int a = 0; a < Cfg::PM; ++a)
#pragma unroll
        for (int b = 0; b < Cfg::PN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    constexpr int G_CPR = TM * 32 / EPC, A_CPR = TN * 32 / EPC;     // 16-byte chunks per tile row
    constexpr int G_N = (BKM * G_CPR + 255) / 256, A_N = (BKM * A_CPR + 255) / 256;
    u32x4 g_reg[G_N], a_reg[A_N];

    // (n, oh, ow) of each A chunk's pixel row, advanced by BKM rows per stage without divisions
    int pn[A_N], poh[A_N], pow_[A_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int idx = tid + 256 * i;
        long m = m_begin + (idx < BKM * A_CPR ? idx / A_CPR : 0);
        if (m >= M) m = M - 1;
        pn[i] = (int)(m / ((long)p.Ho * p.Wo));
        const int rem = (int)(m - (long)pn[i] * p.Ho * p.Wo);
        poh[i] = rem / p.Wo;
        pow_[i] = rem - poh[i] * p.Wo;
    }
    auto advance_rows = [&]() {
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            pow_[i] += BKM;
            while (pow_[i] >= p.Wo) {
                pow_[i] -= p.Wo;
                if (++poh[i] >= p.Ho) {
                    poh[i] = 0;
                    ++pn[i];
                }
            }
        }
    };

    auto load_stage = [&](long mb) {
#pragma unroll
        for (int i = 0; i < G_N; ++i) {
            const int idx = tid + 256 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < BKM * G_CPR) {
                const int row = idx / G_CPR, ch = idx % G_CPR;
                const long m = mb + row;
                const int co = co0 + ch * EPC;
                if (m < m_end && co < p.Cout) {
                    const bool sb = m >= Ma;                         // second source of a two-use launch
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(sb ? p.gy_b : p.gy) + ((sb ? m - Ma : m) * p.Cout + co) * (PRECISE ? 4 : 2));
                }
            }
            g_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int idx = tid + 256 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < BKM * A_CPR) {
                const int row = idx / A_CPR, ch = idx % A_CPR;
                const long m = mb + row;
                const int cg = ci0 + ch * EPC;                       // global input channel of this chunk
                const bool second = cg >= p.C1;
                const bool sb = pn[i] >= p.Na;
                const char* xsrc = reinterpret_cast<const char*>(second ? (sb ? p.x2_b : p.x2) : (sb ? p.x_b : p.x));
                const int csrc = second ? (p.Cin - p.C1) : p.C1;
                const int cbase = second ? (cg - p.C1) : cg;
                if (m < m_end && cg < p.Cin) {
                    const int n = sb ? pn[i] - p.Na : pn[i], oh = poh[i], ow = pow_[i];    // tracked incrementally (advance_rows)
                    int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                    if (p.reflect) {
                        ih = reflect_idx(ih, p.H);
                        iw = reflect_idx(iw, p.W);
                    }
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
                        const long off = (((long)n * p.H + ih) * p.W + iw) * csrc + cbase;
                        v = *reinterpret_cast<const u32x4*>(xsrc + off * (PRECISE ? 4 : 2));
                    }
                }
            }
            a_reg[i] = v;
        }
    };

    auto store_tile = [&](const u32x4& v, __bf16* hi, __bf16* lo, int row, int ch, int stride_) {
        if (PRECISE) {
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            u32x2 h2, l2;
            h2[0] = pack2(f[0], f[1]);
            h2[1] = pack2(f[2], f[3]);
            l2[0] = pack2(f[0] - bf16_round(f[0]), f[1] - bf16_round(f[1]));
            l2[1] = pack2(f[2] - bf16_round(f[2]), f[3] - bf16_round(f[3]));
            *reinterpret_cast<u32x2*>(hi + row * stride_ + ch * 4) = h2;
            *reinterpret_cast<u32x2*>(lo + row * stride_ + ch * 4) = l2;
        } else {
            *reinterpret_cast<u32x4*>(hi + row * stride_ + ch * 8) = v;
        }
    };

    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < G_N; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BKM * G_CPR) store_tile(g_reg[i], G_hi, G_lo, idx / G_CPR, idx % G_CPR, GS);
        }
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BKM * A_CPR) store_tile(a_reg[i], A_hi, A_lo, idx / A_CPR, idx % A_CPR, AS);
        }
    };

    if (m_begin < m_end) {
        load_stage(m_begin);
        advance_rows();
    }
    for (long mb = m_begin; mb < m_end; mb += BKM) {
        store_stage();
        __syncthreads();
        if (mb + BKM < m_end) {
            load_stage(mb + BKM);
            advance_rows();
        }
        if (active) {
#pragma unroll
            for (int kk = 0; kk < BKM / 16; ++kk) {
                bf16x8 gh[Cfg::PM], gl[Cfg::PM], ah[Cfg::PN], al[Cfg::PN];
#pragma unroll
                for (int a = 0; a < Cfg::PM; ++a) {
                    const int col = (wm * Cfg::PM + a) * 32;
                    gh[a] = tr_frag(G_hi, GS, kk * 16, col, lane);
                    if (PRECISE) gl[a] = tr_frag(G_lo, GS, kk * 16, col, lane);
                }
#pragma unroll
                for (int b = 0; b < Cfg::PN; ++b) {
                    const int col = (wn * Cfg::PN + b) * 32;
                    ah[b] = tr_frag(A_hi, AS, kk * 16, col, lane);
                    if (PRECISE) al[b] = tr_frag(A_lo, AS, kk * 16, col, lane);
                }
#pragma unroll
                for (int a = 0; a < Cfg::PM; ++a)
#pragma unroll
                    for (int b = 0; b < Cfg::PN; ++b) {
                        if (PRECISE) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl[a], ah[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[a], al[b], acc[a][b], 0, 0, 0);
                        }
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[a], ah[b], acc[a][b], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: slab [z][Cout][taps][Cin] fp32 (row = co on registers, column = ci on lanes)
    if (active) {
        const int r = lane & 31, h = lane >> 5;
        const long taps = (long)p.KH * p.KW;
        float* slab = p.partial + (long)blockIdx.z * p.Cout * taps * p.Cin;
#pragma unroll
        for (int a = 0; a < Cfg::PM; ++a)
#pragma unroll
            for (int b = 0; b < Cfg::PN; ++b) {
                const int ci = ci0 + (wn * Cfg::PN + b) * 32 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co = co0 + (wm * Cfg::PM + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (co < p.Cout && ci < p.Cin) slab[((long)co * taps + tap) * p.Cin + ci] = acc[a][b][i];
                }
            }
    }
}

// sum the slabs in order; emit nn.Conv2d layout [Cout][Cin_out][KH][KW] (Cin_out <= Cin: the stem's padded
// im2col columns are dropped)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int slabs, int Cout, int Cin,
                                                           int Cin_out, int KH, int KW, int im2col, int accumulate,
                                                           float* __restrict__ gw) {
    // threads walk the SOURCE (packed) layout so the slabs are read coalesced; the transposing write happens once
    const long slab = (long)Cout * (im2col ? 1 : KH * KW) * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < slab; i += (long)gridDim.x * 256) {
        int co, kh, kw, ci;
        if (im2col) {                                       // source [Cout][Cin = padded (kh, kw, ci) columns]
            co = (int)(i / Cin);
            const int col = (int)(i % Cin);
            if (col >= KH * KW * Cin_out) continue;
            ci = col % Cin_out;
            kw = (col / Cin_out) % KW;
            kh = col / (Cin_out * KW);
        } else {                                            // source [Cout][KH][KW][Cin]
            ci = (int)(i % Cin);
            long t = i / Cin;
            kw = (int)(t % KW);
            t /= KW;
            kh = (int)(t % KH);
            co = (int)(t / KH);
            if (ci >= Cin_out) continue;
        }
        double s = 0.0;
        int z = 0;
        for (; z + 3 < slabs; z += 4) {                     // four slabs' values in flight (latency bound), added in slab order
            const float v0 = partial[(long)z * slab + i], v1 = partial[(long)(z + 1) * slab + i];
            const float v2 = partial[(long)(z + 2) * slab + i], v3 = partial[(long)(z + 3) * slab + i];
            s += (double)v0;
            s += (double)v1;
            s += (double)v2;
            s += (double)v3;
        }
        for (; z < slabs; ++z) s += (double)partial[(long)z * slab + i];
        float* dst = gw + (((long)co * Cin_out + ci) * KH + kh) * KW + kw;
        *dst = accumulate ? *dst + (float)s : (float)s;
    }
}

// =====================================================================================================
// Weight gradient of 3x3 / stride 1 / pad 1 layers, ALL NINE TAPS IN ONE WORKGROUP (bf16 activations).
//
// The per-tap kernel above re-reads the GY tile and a shifted copy of the input for each tap: 64 B of operand
// per MFMA clock and CU, far above what L2 -> LDS delivers.  Here a workgroup owns (32*COT output channels) x
// (32*CIT input channels) x 9 taps and walks 4 x 16 blocks of output pixels: per block it stages ONE GY tile
// [64 px][32*COT] and ONE input patch [6 x 18 px][32*CIT] (halo included) and feeds all nine taps from shifted
// windows of that patch -- the MFMA K dimension (16 pixels) is one block row, so a tap's window is 16
// consecutive patch pixels.  Both tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a 3-deep ring with
// counted vmcnt; rows are unpadded and 16-byte chunks are XOR-swizzled through the SOURCE address so the
// transposing fragment reads (ds_read_b64_tr_b16, 4 pixel rows x 64 B per half-wave) are conflict-free.
// Waves split (ci tile) x (co pair) x (tap range); a wave keeps its G fragments for all its taps.
// Output: the same per-slab fp32 partials [z][Cout][9][Cin] as the per-tap kernel (fixed-order reduction).
// =====================================================================================================
// S = 2 (r4): the stride-2 layers (first conv2 of a Bottleneck stage / first conv1 of a BasicBlock stage) ran on the per-tap kernel at
// 6-8 % MFMA busy and 3-4 x the algorithmic bytes (r04_conv_weak_layers_pmc.md).  Same machinery with a 2 x 16 output block and its
// 5 x 33 input patch; the DMA de-interleaves the patch columns into an even plane (17 pixels) and an odd plane (16) per patch row, so
// a tap's 16 input pixels (columns 2 ow + kw - 1) are again CONSECUTIVE patch pixels: kw = 0 / 2 read the even plane at ow / ow + 1,
// kw = 1 the odd plane at ow.
template <int COT, int CIT, int S = 1>
struct Wg3 {
    static constexpr int BH = S == 1 ? 4 : 2;               // output rows of a block (16 columns): one MFMA K step per row
    static constexpr int PR = S * BH + 3 - S, PC = S * 16 + 3 - S, NPATCH = PR * PC;   // patch rows / columns / pixels (halo included)
    static constexpr int PM = COT >= 2 ? 2 : 1;             // co tiles per wave
    static constexpr int WO = COT / PM, WC = CIT;            // wave groups over co pairs / ci tiles
    static constexpr int NW = (COT == 4 && CIT == 2) ? 8 : 4;
    static constexpr int WT = NW / (WO * WC);                // wave groups over taps
    static constexpr int NTM = (9 + WT - 1) / WT;            // taps per wave (upper bound)
    static constexpr int RBG = COT * 64, RBA = CIT * 64;     // row bytes of the GY tile / the patch
    static constexpr int G_INSTR = BH * 16 * RBG / 1024;     // 1 KB DMA wave-instructions for the GY tile
    static constexpr int G_PER = G_INSTR / NW;
    static constexpr int P_INSTR = (NPATCH * RBA + 1023) / 1024;
    static constexpr int P_PER = (P_INSTR + NW - 1) / NW;    // every wave issues the same count (counted vmcnt)
    static constexpr int G_BYTES = BH * 16 * RBG, P_BYTES = P_PER * NW * 1024;
    static constexpr int STAGE = G_BYTES + P_BYTES;
    static constexpr int NBUF = 3;
    static_assert(G_INSTR % NW == 0, "GY tile must split evenly over the waves");
};

template <int RB>
__device__ __forceinline__ int swz_chunk(int row) {
    return RB == 256 ? ((row & 3) << 2) : (RB == 128 ? (((row >> 1) & 1) << 2) : 0);
}

// transposing fragment read from an unpadded, chunk-swizzled [rows][RB bytes] bf16 tile:
// lane (r = lane & 31, h = lane >> 5) gets column col0 + r, rows pix0 + 8h .. +7
template <int RB>
__device__ __forceinline__ bf16x8 tr_frag_swz(const char* tile, int pix0, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row = pix0 + 8 * (g >> 1) + q;
    const int cbyte = 2 * col0 + 32 * (g & 1) + 8 * pp;
    const char* a0 = tile + row * RB + (((cbyte >> 4) ^ swz_chunk<RB>(row)) << 4) + (cbyte & 8);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * RB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

#ifndef WGRAD_WALK_DOWN
#define WGRAD_WALK_DOWN 1
#endif
template <int COT, int CIT, int S = 1>
__global__ __launch_bounds__((Wg3<COT, CIT, S>::NW * 64)) void conv_wgrad3x3_kernel(const WgradArgs p, int blocks_per_slab) {
    using C = Wg3<COT, CIT, S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % C::WC, wo = (wave / C::WC) % C::WO, wt = wave / (C::WC * C::WO);
    const int t0 = (9 * wt + C::WT - 1) / C::WT, t1 = (9 * (wt + 1) + C::WT - 1) / C::WT;

    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_slabs > 0) {                                  // 1-D grid, a slab's tiles on one XCD (WgradArgs::xcd_slabs)
        const int cit = p.Cin / (32 * CIT), t_all = cit * (p.Cout / (32 * COT));
        const int q = blockIdx.x >> 3, t = q % t_all;
        bz = (q / t_all) * 8 + (int)(blockIdx.x & 7u);
        if (bz >= p.xcd_slabs) return;
        bx = t % cit, by = t / cit;
    }
    const int ci0 = bx * (32 * CIT), co0 = by * (32 * COT);
    const int bw = p.Wo >> 4, bh = p.Ho / C::BH;
    const int n_blk = p.N * bh * bw;
    const int b_begin = bz * blocks_per_slab;
    int b_end = b_begin + blocks_per_slab;
    if (b_end > n_blk) b_end = n_blk;
    const int ns = b_end > b_begin ? b_end - b_begin : 0;

    const bool second = ci0 >= p.C1;
    const char* xsrc_a = reinterpret_cast<const char*>(second ? p.x2 : p.x);
    const char* xsrc_b = reinterpret_cast<const char*>(second ? p.x2_b : p.x_b);     // second source (two-use launch), images >= Na
    const int csrc = second ? (p.Cin - p.C1) : p.C1;
    const int cbase = second ? (ci0 - p.C1) : ci0;
    const char* gsrc_a = reinterpret_cast<const char*>(p.gy);
    const char* gsrc_b = reinterpret_cast<const char*>(p.gy_b);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // ---- stage-invariant lane roles of the DMA instructions
    int g_rel[C::G_PER];                                   // element offset of this lane's GY chunk inside a block
#pragma unroll
    for (int i = 0; i < C::G_PER; ++i) {
        const int j = wave + C::NW * i;
        const int bp = j * (1024 / C::RBG) + (lane * 16) / C::RBG;          // block pixel 0..16 BH - 1
        const int slot = lane & (C::RBG / 16 - 1);
        const int chunk = slot ^ swz_chunk<C::RBG>(bp);
        g_rel[i] = ((bp >> 4) * p.Wo + (bp & 15)) * p.Cout + co0 + chunk * 8;
    }
    int p_dr[C::P_PER], p_dc[C::P_PER], p_ch[C::P_PER];   // patch row / column (relative to the block) and channel
#pragma unroll
    for (int i = 0; i < C::P_PER; ++i) {
        const int j = wave + C::NW * i;
        const int pp = j * (1024 / C::RBA) + (lane * 16) / C::RBA;          // patch pixel 0..NPATCH - 1 (beyond: zero page)
        const int slot = lane & (C::RBA / 16 - 1);
        const int chunk = slot ^ swz_chunk<C::RBA>(pp);
        const int pr = pp / C::PC;
        const int q = pp - C::PC * pr;                                      // position inside the LDS patch row
        const int pc = S == 1 ? q : (q < 17 ? 2 * q : 2 * (q - 17) + 1);   // S = 2: even plane first, then the odd plane
        p_dr[i] = pp < C::NPATCH ? pr - 1 : -(1 << 20);
        p_dc[i] = pc - 1;
        p_ch[i] = cbase + chunk * 8;
    }

    int toff[C::NTM];                                      // patch pixel offset of each of this wave's taps
#pragma unroll
    for (int t = 0; t < C::NTM; ++t) {
        const int tap = t0 + t < 9 ? t0 + t : 8;
        const int kw = tap % 3;
        toff[t] = (tap / 3) * C::PC + (S == 1 ? kw : (kw == 1 ? 17 : kw >> 1));
    }

    f32x16 acc[C::PM][C::NTM];
#pragma unroll
    for (int a = 0; a < C::PM; ++a)
#pragma unroll
        for (int t = 0; t < C::NTM; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][t][i] = 0.0f;

    // block being issued: (image, block column, block row), advanced without divisions.  r4: the walk goes DOWN a column of blocks
    // first (WGRAD_WALK_DOWN): consecutive blocks then share the 2 halo rows of the same 18 / 33 patch columns, re-read by the very
    // next stage out of L2, instead of 16+ blocks later when a row-major walk comes back one block row further down (the vertical
    // halo was 1/3 of the patch traffic: counters 1.5 x the input bytes; the horizontal one, 2 of 18 columns, still comes around late).
    // Any bijection of the block index is a valid order: a slab is a range of it.
#if WGRAD_WALK_DOWN
    int in_ = b_begin / (bh * bw);
    int ibx = (b_begin - in_ * bh * bw) / bh;
    int iby = b_begin - (in_ * bw + ibx) * bh;
#else
    int in_ = b_begin / (bh * bw);
    int iby = (b_begin - in_ * bh * bw) / bw;
    int ibx = b_begin - (in_ * bh + iby) * bw;
#endif

    auto stage = [&](int buf) {
        char* Gs = smem + buf * C::STAGE;
        char* Ps = Gs + C::G_BYTES;
        const int oh0 = iby * C::BH, ow0 = ibx * 16;
        const bool sb = in_ >= p.Na;                        // wave-uniform: the block's image belongs to the second source
        const int img = sb ? in_ - p.Na : in_;
        const char* gsrc = sb ? gsrc_b : gsrc_a;
        const char* xsrc = sb ? xsrc_b : xsrc_a;
        const long gbase = (((long)img * p.Ho + oh0) * p.Wo + ow0) * p.Cout;
#pragma unroll
        for (int i = 0; i < C::G_PER; ++i) glds16(gsrc + (gbase + g_rel[i]) * 2, Gs + (wave + C::NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < C::P_PER; ++i) {
            int ih = S * oh0 + p_dr[i], iw = S * ow0 + p_dc[i];
            if (p.reflect) {
                if (ih > -(1 << 19)) ih = reflect_idx(ih, p.H);
                iw = reflect_idx(iw, p.W);
            }
            const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const long off = (((long)img * p.H + ih) * p.W + iw) * csrc + p_ch[i];
            glds16(ok ? xsrc + off * 2 : zero, Ps + (wave + C::NW * i) * 1024);
        }
#if WGRAD_WALK_DOWN
        if (++iby == bh) {
            iby = 0;
            if (++ibx == bw) {
                ibx = 0;
                ++in_;
            }
        }
#else
        if (++ibx == bw) {
            ibx = 0;
            if (++iby == bh) {
                iby = 0;
                ++in_;
            }
        }
#endif
    };

    auto compute = [&](int buf) {
        const char* Gs = smem + buf * C::STAGE;
        const char* Ps = Gs + C::G_BYTES;
#pragma unroll
        for (int kk = 0; kk < C::BH; ++kk) {               // block row kk = 16 pixels = one MFMA K step
            bf16x8 gf[C::PM];
#pragma unroll
            for (int a = 0; a < C::PM; ++a) gf[a] = tr_frag_swz<C::RBG>(Gs, kk * 16, (wo * C::PM + a) * 32, lane);
#pragma unroll
            for (int t = 0; t < C::NTM; ++t) {
                if (t0 + t < t1) {                          // wave-uniform
                    const bf16x8 af = tr_frag_swz<C::RBA>(Ps, kk * S * C::PC + toff[t], wc * 32, lane);
#pragma unroll
                    for (int a = 0; a < C::PM; ++a)
                        acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[a], af, acc[a][t], 0, 0, 0);
                }
            }
        }
    };

    // ring: see conv_igemm_glds_kernel
    constexpr int G = C::G_PER + C::P_PER;
    auto wait_in_flight = [&](int stages) {
        if (stages >= 1) __builtin_amdgcn_s_waitcnt((G & 0xF) | ((G >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
    };
#pragma unroll
    for (int i = 0; i < C::NBUF - 1; ++i)
        if (i < ns) stage(i);
    int slot_c = 0, slot_i = C::NBUF - 1;
    for (int s = 0; s < ns; ++s) {
        const int behind = ns - 1 - s;
        wait_in_flight(behind < C::NBUF - 2 ? behind : C::NBUF - 2);
        __builtin_amdgcn_s_barrier();
        if (s + C::NBUF - 1 < ns) stage(slot_i);
        compute(slot_c);
        slot_c = slot_c == C::NBUF - 1 ? 0 : slot_c + 1;
        slot_i = slot_i == C::NBUF - 1 ? 0 : slot_i + 1;
    }

    // ---- epilogue: slab [z][Cout][Cin][9] fp32 = nn.Conv2d's own layout, so the slab reduction is a plain vector
    // sum.  The accumulators (row = co on registers, column = ci on lanes, tap = register array) go through LDS
    // 16 output channels at a time as [co][ci][tap] and leave as contiguous 16-byte stores.
    const int r = lane & 31, h = lane >> 5;
    float* slab = p.partial + (long)bz * p.Cout * 9 * p.Cin;
    float* tile = reinterpret_cast<float*>(smem);
    constexpr int ROW = 32 * CIT * 9;                      // floats per output channel of this workgroup's tile
    __syncthreads();                                       // ring slots are free
#pragma unroll
    for (int ag = 0; ag < COT; ++ag) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (wo == ag / C::PM) {
#pragma unroll
                for (int t = 0; t < C::NTM; ++t) {
                    if (t0 + t < t1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                            tile[(row * (32 * CIT) + wc * 32 + r) * 9 + t0 + t] = acc[ag % C::PM][t][hf * 8 + j];
                        }
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < 16 * (ROW / 4); i += C::NW * 64) {
                const int row = i / (ROW / 4), q4 = i - row * (ROW / 4);
                const int co = co0 + ag * 32 + hf * 16 + row;
                *reinterpret_cast<f32x4*>(slab + ((long)co * p.Cin + ci0) * 9 + q4 * 4) = *reinterpret_cast<const f32x4*>(tile + row * ROW + q4 * 4);
            }
            __syncthreads();
        }
    }
}

// =====================================================================================================
// Weight gradient of 1x1 layers (bf16; stride 1 or 2, no padding): dW[co][ci] = sum_m GY[m][co] * X[m'][ci].
// Same machinery as the nine-tap kernel without the halo: 64 pixel rows per stage, GY tile [64][32*COT] and X tile
// [64][32*CIT] by LDS-DMA into a 3-deep ring (counted vmcnt), swizzled transposing fragment reads, 8 waves each owning
// PM x PN accumulator tiles.  The weight is small, the pixel count large: most workgroups are pixel slabs, so each
// operand row is fetched once or twice in total -- these layers are HBM-bound.  Partials [z][Cout][Cin] are already
// in nn.Conv2d's layout (fixed-order slab sum afterwards).
// =====================================================================================================
template <int COT, int CIT>
struct Wg1 {
    // 64-output-channel layers (r4): <2, 5> = the stem's patch matrix (160 columns: ALL of them in one workgroup, ten waves -- the
    // per-tap kernel read GY once per 32-column tile, 2 GB for a 0.94 GB problem), <2, 2> = 64 -> 64 (four waves)
    static constexpr int NW = (COT == 2 && CIT == 5) ? 10 : ((COT == 2 && CIT == 2) ? 4 : 8);
    static constexpr int WO = COT < 4 ? COT : 4, PM = COT / WO;   // wave groups over co, co tiles per wave
    static constexpr int WC = NW / WO, PN = CIT / WC;             // wave groups over ci, ci tiles per wave
    static constexpr int RBG = COT * 64, RBA = CIT * 64;         // row bytes
    static constexpr int G_BYTES = 64 * RBG, A_BYTES = 64 * RBA;
    // every wave issues the same number of 1 KB DMA instructions (counted vmcnt); instructions beyond a tile read the zero page into
    // the padding of its region
    static constexpr int G_PER = (G_BYTES / 1024 + NW - 1) / NW, A_PER = (A_BYTES / 1024 + NW - 1) / NW;
    static constexpr int G_REGION = G_PER * NW * 1024;
    static constexpr int STAGE = G_REGION + A_PER * NW * 1024;
    static constexpr int NBUF = 3;
    static_assert(WO * WC == NW && PN * WC == CIT && PM * WO == COT, "unsupported tile");
    static_assert(RBG <= 256 || RBG == 512, "GY rows: 64..256 B swizzle classes, or 512 B");
};

// chunk swizzle for 512-byte rows (32 chunks): as for 256-byte rows, rows q = 0..3 of a fragment read land in
// different 64-byte bank groups
template <int RB>
__device__ __forceinline__ int swz_chunk_w(int row) {
    // 320-byte rows (the stem's 160 patch columns) need none: consecutive rows start 16 banks apart, the four 64-byte row segments
    // of a half-wave's fragment read cover the 64 banks once
    return RB == 320 ? 0 : (RB >= 256 ? ((row & 3) << 2) : (RB == 128 ? (((row >> 1) & 1) << 2) : 0));
}
template <int RB>
__device__ __forceinline__ bf16x8 tr_frag_w(const char* tile, int pix0, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row = pix0 + 8 * (g >> 1) + q;
    const int cbyte = 2 * col0 + 32 * (g & 1) + 8 * pp;
    const char* a0 = tile + row * RB + (((cbyte >> 4) ^ swz_chunk_w<RB>(row)) << 4) + (cbyte & 8);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * RB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int COT, int CIT>
__global__ __launch_bounds__((Wg1<COT, CIT>::NW * 64)) void conv_wgrad1x1_kernel(const WgradArgs p, int stages_per_slab) {
    using C = Wg1<COT, CIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave % C::WO, wc = wave / C::WO;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_slabs > 0) {                                  // 1-D grid, a slab's tiles on one XCD (WgradArgs::xcd_slabs)
        const int cit = p.Cin / (32 * CIT), t_all = cit * (p.Cout / (32 * COT));
        const int q = blockIdx.x >> 3, t = q % t_all;
        bz = (q / t_all) * 8 + (int)(blockIdx.x & 7u);
        if (bz >= p.xcd_slabs) return;
        bx = t % cit, by = t / cit;
    }
    const int ci0 = bx * (32 * CIT), co0 = by * (32 * COT);
    const long M = (long)p.N * p.Ho * p.Wo;
    const long n_stage_all = (M + 63) / 64;
    const long s_begin = (long)bz * stages_per_slab;
    long s_end = s_begin + stages_per_slab;
    if (s_end > n_stage_all) s_end = n_stage_all;
    const int ns = s_end > s_begin ? (int)(s_end - s_begin) : 0;
    const char* gsrc_a = reinterpret_cast<const char*>(p.gy);
    const char* gsrc_b = reinterpret_cast<const char*>(p.gy_b);       // second source (two-use launch): rows >= Ma / images >= Na
    const char* xsrc_a = reinterpret_cast<const char*>(p.x);
    const char* xsrc_b = reinterpret_cast<const char*>(p.x_b);
    const long Ma = (long)p.Na * p.Ho * p.Wo;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // ---- lane roles of the DMA instructions: instruction j covers tile rows j * (1024 / RB) ...
    int g_row[C::G_PER], g_col[C::G_PER];
#pragma unroll
    for (int i = 0; i < C::G_PER; ++i) {
        const int j = wave + C::NW * i;
        g_row[i] = j * (1024 / C::RBG) + (lane * 16) / C::RBG;     // rows >= 64: instruction beyond the tile (zero page)
        const int slot = lane & (C::RBG / 16 - 1);
        g_col[i] = co0 + ((slot ^ swz_chunk_w<C::RBG>(g_row[i])) << 3);
    }
    int a_row[C::A_PER], a_col[C::A_PER];
    int a_n[C::A_PER], a_oh[C::A_PER], a_ow[C::A_PER];     // output pixel of the row in the stage being issued
#pragma unroll
    for (int i = 0; i < C::A_PER; ++i) {
        const int j = wave + C::NW * i;
        const int byte = j * 1024 + lane * 16;                   // (rows need not divide 1 KB: 320-byte rows)
        a_row[i] = byte / C::RBA;                                // rows >= 64: instruction beyond the tile (zero page)
        const int slot = (byte - a_row[i] * C::RBA) >> 4;
        a_col[i] = ci0 + ((slot ^ swz_chunk_w<C::RBA>(a_row[i])) << 3);
        long m = s_begin * 64 + a_row[i];
        if (m >= M) m = M - 1;
        int rem;
        split_row(m, p.Ho * p.Wo, M, a_n[i], rem);
        a_oh[i] = rem / p.Wo;
        a_ow[i] = rem - a_oh[i] * p.Wo;
    }

    f32x16 acc[C::PM][C::PN];
#pragma unroll
    for (int a = 0; a < C::PM; ++a)
#pragma unroll
        for (int b = 0; b < C::PN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    long issue_m0 = s_begin * 64;                          // first pixel row of the stage being issued
    auto stage = [&](int buf) {
        char* Gs = smem + buf * C::STAGE;
        char* As = Gs + C::G_REGION;
#pragma unroll
        for (int i = 0; i < C::G_PER; ++i) {
            const long m = issue_m0 + g_row[i];
            const bool sb = m >= Ma;
            glds16(g_row[i] < 64 && m < M ? (sb ? gsrc_b : gsrc_a) + ((sb ? m - Ma : m) * p.Cout + g_col[i]) * 2 : zero,
                   Gs + (wave + C::NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < C::A_PER; ++i) {
            const long m = issue_m0 + a_row[i];
            const bool ok = a_row[i] < 64 && m < M;
            const bool sb = a_n[i] >= p.Na;
            const long pix = ((long)(sb ? a_n[i] - p.Na : a_n[i]) * p.H + a_oh[i] * p.stride) * p.W + a_ow[i] * p.stride;
            glds16(ok ? (sb ? xsrc_b : xsrc_a) + (pix * p.Cin + a_col[i]) * 2 : zero, As + (wave + C::NW * i) * 1024);
            a_ow[i] += 64;                                  // this lane's row of the next stage
            while (a_ow[i] >= p.Wo) {
                a_ow[i] -= p.Wo;
                if (++a_oh[i] >= p.Ho) {
                    a_oh[i] = 0;
                    ++a_n[i];
                }
            }
        }
        issue_m0 += 64;
    };
    auto compute = [&](int buf) {
        const char* Gs = smem + buf * C::STAGE;
        const char* As = Gs + C::G_REGION;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 gf[C::PM], af[C::PN];
#pragma unroll
            for (int a = 0; a < C::PM; ++a) gf[a] = tr_frag_w<C::RBG>(Gs, kk * 16, (wo * C::PM + a) * 32, lane);
#pragma unroll
            for (int b = 0; b < C::PN; ++b) af[b] = tr_frag_w<C::RBA>(As, kk * 16, (wc * C::PN + b) * 32, lane);
#pragma unroll
            for (int a = 0; a < C::PM; ++a)
#pragma unroll
                for (int b = 0; b < C::PN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[a], af[b], acc[a][b], 0, 0, 0);
        }
    };

    constexpr int G = C::G_PER + C::A_PER;
    auto wait_in_flight = [&](int stages) {
        if (stages >= 1) __builtin_amdgcn_s_waitcnt((G & 0xF) | ((G >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
    };
#pragma unroll
    for (int i = 0; i < C::NBUF - 1; ++i)
        if (i < ns) stage(i);
    int slot_c = 0, slot_i = C::NBUF - 1;
    for (int s = 0; s < ns; ++s) {
        const int behind = ns - 1 - s;
        wait_in_flight(behind < C::NBUF - 2 ? behind : C::NBUF - 2);
        __builtin_amdgcn_s_barrier();
        if (s + C::NBUF - 1 < ns) stage(slot_i);
        compute(slot_c);
        slot_c = slot_c == C::NBUF - 1 ? 0 : slot_c + 1;
        slot_i = slot_i == C::NBUF - 1 ? 0 : slot_i + 1;
    }

    // ---- slab [z][Cout][Cin] fp32 (row = co on registers, column = ci on lanes: 128-byte runs)
    const int r = lane & 31, h = lane >> 5;
    float* slab = p.partial + (long)bz * p.Cout * p.Cin;
#pragma unroll
    for (int a = 0; a < C::PM; ++a)
#pragma unroll
        for (int b = 0; b < C::PN; ++b) {
            const int ci = ci0 + (wc * C::PN + b) * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + (wo * C::PM + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                slab[(long)co * p.Cin + ci] = acc[a][b][i];
            }
        }
}

struct Wg1Plan {
    int cot, cit, slabs, stages_per_slab;                  // cot == 0: not eligible
};

static Wg1Plan wgrad1x1_plan(const WgradArgs& a, int precise, bool shape_only) {
    Wg1Plan pl{0, 0, 0, 0};
    if (precise || a.KH != 1 || a.KW != 1 || (!shape_only && a.C1 != a.Cin)) return pl;
    if (!shape_only && (a.pad != 0 || (a.stride != 1 && a.stride != 2))) return pl;
    int cot, cit;
    if (a.per_tap_only) {                                   // the stem's patch matrix: 7 x 7 x 3 taps padded to 160 columns
        if (!g_wgrad1x1_narrow || a.Cout % 64 || a.Cin != 160) return pl;
        cot = 2, cit = 5;
    } else if (a.Cout % 256 == 0 && a.Cin % 128 == 0) cot = 8, cit = 4;
    else if (a.Cout % 256 == 0 && a.Cin % 64 == 0) cot = 8, cit = 2;
    else if (a.Cout % 128 == 0 && a.Cin % 128 == 0) cot = 4, cit = 4;
    else if (a.Cout % 128 == 0 && a.Cin % 64 == 0) cot = 4, cit = 2;
    else if (a.Cout % 64 == 0 && a.Cin % 128 == 0) cot = 2, cit = 4;
    else if (g_wgrad1x1_narrow && a.Cout % 64 == 0 && a.Cin % 64 == 0) cot = 2, cit = 2;
    else return pl;
    const long tiles = (long)(a.Cin / (32 * cit)) * (a.Cout / (32 * cot));
    const long n_stage = ((long)a.N * a.Ho * a.Wo + 63) / 64;
    if (n_stage > (1L << 30)) return pl;
    long s = ((cit == 2 && cot == 2 ? 768 : 256) * g_wgrad_round_pct / 100 + tiles - 1) / tiles;   // one resident round of workgroups (4-wave ones: three per CU)
    const long max_s = (n_stage + 3) / 4;                   // at least 4 stages (256 pixels) per slab
    if (s > max_s) s = max_s;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    const long sps = (n_stage + s - 1) / s;
    s = (n_stage + sps - 1) / sps;
    pl.cot = cot;
    pl.cit = cit;
    pl.slabs = (int)s;
    pl.stages_per_slab = (int)sps;
    return pl;
}

template <int COT, int CIT>
static void wgrad1x1_launch_t(const WgradArgs& a, const Wg1Plan& pl, hipStream_t st) {
    using C = Wg1<COT, CIT>;
    const unsigned tiles = (unsigned)(a.Cin / (32 * CIT)) * (unsigned)(a.Cout / (32 * COT));
    if (g_wgrad_xcd && tiles > 1 && pl.slabs >= 8) {
        WgradArgs b = a;
        b.xcd_slabs = pl.slabs;
        hipLaunchKernelGGL((conv_wgrad1x1_kernel<COT, CIT>), dim3(8u * tiles * (unsigned)((pl.slabs + 7) / 8)), dim3(C::NW * 64), (size_t)C::NBUF * C::STAGE, st,
                           b, pl.stages_per_slab);
        return;
    }
    dim3 grid((unsigned)(a.Cin / (32 * CIT)), (unsigned)(a.Cout / (32 * COT)), (unsigned)pl.slabs);
    hipLaunchKernelGGL((conv_wgrad1x1_kernel<COT, CIT>), grid, dim3(C::NW * 64), (size_t)C::NBUF * C::STAGE, st, a, pl.stages_per_slab);
}

struct Wg3Plan {
    int cot, cit, slabs, blocks_per_slab;                  // cot == 0: not eligible
    int stride;
};

static Wg3Plan wgrad3x3_plan(const WgradArgs& a, int precise, bool shape_only, int force_cit = 0) {
    Wg3Plan pl{0, 0, 0, 0, 1};
    const bool s2 = g_wgrad3x3_s2 && a.H == 2 * a.Ho && a.W == 2 * a.Wo;       // stride 2 (pad 1): Cout % 128 only (COT = 4)
    if (precise || a.per_tap_only || a.KH != 3 || a.KW != 3 || a.Cout % 32 || a.Cin % 32 || a.Wo % 16) return pl;
    if (s2 ? (a.Ho % 2 || a.Cout % 128) : (a.Ho != a.H || a.Wo != a.W || a.Ho % 4)) return pl;
    if (!shape_only && (a.stride != (s2 ? 2 : 1) || a.pad != 1)) return pl;
    int cit = (a.Cin % 64 == 0 && force_cit != 1) ? 2 : 1;
    if (!shape_only && a.C1 != a.Cin && a.C1 % (32 * cit)) {
        if (a.C1 % 32) return pl;
        cit = 1;
    }
    const int cot = a.Cout % 128 == 0 ? 4 : (a.Cout % 64 == 0 ? 2 : 1);
    const long tiles = (long)(a.Cin / (32 * cit)) * (a.Cout / (32 * cot));
    const long n_blk = (long)a.N * (a.Ho / (s2 ? 2 : 4)) * (a.Wo / 16);
    if (n_blk > (1L << 30)) return pl;
    // every workgroup writes its whole (co x ci x 9) fp32 tile once, so the partial volume is (#workgroups x tile
    // bytes) whatever the layer: one resident round of workgroups is the cheapest split
    long target = (cot == 4 && cit == 2) ? 256 : 512;
    long s = (target + tiles - 1) / tiles;
    long cap = 256;
    if (g_wgrad3x3_fill) {
        // r4: ONE FULL round, never one workgroup more.  192 -> 32 at 256^2 ran 3 x 171 = 513 workgroups on 512 slots (two 54 KB
        // workgroups per CU): the 513th ran alone after the others -- twice the time (647 us).  And the narrow tiles left the chip
        // mostly empty (32 -> 32: 256 four-wave workgroups = one per CU where four fit).  Slots = CUs x workgroups per CU by LDS and
        // wave slots; slabs = slots / tiles rounded DOWN, up to 1024 of them while the partials stay under 64 MB.
        const int bh = s2 ? 2 : 4, npatch = s2 ? 165 : 108, nw = (cot == 4 && cit == 2) ? 8 : 4;
        const long g_bytes = (long)bh * 16 * cot * 64;
        const long p_instr = ((long)npatch * cit * 64 + 1023) / 1024;
        const long stage = g_bytes + (p_instr + nw - 1) / nw * nw * 1024;
        long per_cu = (160 * 1024) / (3 * stage);
        if (per_cu > 32 / nw) per_cu = 32 / nw;
        if (per_cu < 1) per_cu = 1;
        target = 256 * per_cu * g_wgrad_round_pct / 100;
        s = target / tiles;
        cap = 1024;
        const long tile_bytes = (long)cot * 32 * cit * 32 * 9 * 4;
        while (cap > 256 && cap * tiles * tile_bytes > (64L << 20)) cap >>= 1;
    }
    const long max_s = s2 ? (n_blk + 15) / 16 : (n_blk + 7) / 8;   // at least 512 pixels per slab
    if (s > max_s) s = max_s;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    const long bps = (n_blk + s - 1) / s;
    s = (n_blk + bps - 1) / bps;                            // no empty slabs
    pl.stride = s2 ? 2 : 1;
    pl.cot = cot;
    pl.cit = cit;
    pl.slabs = (int)s;
    pl.blocks_per_slab = (int)bps;
    return pl;
}

template <int COT, int CIT, int S = 1>
static void wgrad3x3_launch_t(const WgradArgs& a, const Wg3Plan& pl, hipStream_t st) {
    using C = Wg3<COT, CIT, S>;
    const unsigned tiles = (unsigned)(a.Cin / (32 * CIT)) * (unsigned)(a.Cout / (32 * COT));
    if (g_wgrad_xcd && tiles > 1 && pl.slabs >= 8) {
        WgradArgs b = a;
        b.xcd_slabs = pl.slabs;
        hipLaunchKernelGGL((conv_wgrad3x3_kernel<COT, CIT, S>), dim3(8u * tiles * (unsigned)((pl.slabs + 7) / 8)), dim3(C::NW * 64), (size_t)C::NBUF * C::STAGE,
                           st, b, pl.blocks_per_slab);
        return;
    }
    dim3 grid((unsigned)(a.Cin / (32 * CIT)), (unsigned)(a.Cout / (32 * COT)), (unsigned)pl.slabs);
    hipLaunchKernelGGL((conv_wgrad3x3_kernel<COT, CIT, S>), grid, dim3(C::NW * 64), (size_t)C::NBUF * C::STAGE, st, a, pl.blocks_per_slab);
}

template <int TM, int TN, bool PRECISE>
static void wgrad_launch_t(const WgradArgs& a, int slabs, hipStream_t st) {
    constexpr int GS = TM * 32 + 32, AS = TN * 32 + 32;
    const size_t lds = (size_t)32 * (GS + AS) * 2 * (PRECISE ? 2 : 1);
    dim3 grid((unsigned)(a.KH * a.KW * ((a.Cin + TN * 32 - 1) / (TN * 32))), (unsigned)((a.Cout + TM * 32 - 1) / (TM * 32)),
              (unsigned)slabs);
    hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, PRECISE>), grid, dim3(256), lds, st, a);
}

static int wgrad_slabs_per_tap(const WgradArgs& a) {
    const int tm = a.Cout >= 128 ? 4 : (a.Cout >= 64 ? 2 : 1);
    const int tn = (a.Cin % 128 == 0) ? 4 : 1;
    const long tiles = (long)a.KH * a.KW * ((a.Cin + tn * 32 - 1) / (tn * 32)) * ((a.Cout + tm * 32 - 1) / (tm * 32));
    const long M = (long)a.N * a.Ho * a.Wo;
    long s = (1024 + tiles - 1) / tiles;                   // aim at >= ~1024 workgroups
    const long max_s = (M + 255) / 256;                     // at least 256 pixels per slab
    if (s > max_s) s = max_s;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
}

// slabs the launch will use (precise/stride/pad known) ...
int wgrad_slabs(const WgradArgs& a, int precise) {
    const Wg3Plan pl = wgrad3x3_plan(a, precise, false);
    if (pl.cot) return pl.slabs;
    const Wg1Plan p1 = wgrad1x1_plan(a, precise, false);
    return p1.cot ? p1.slabs : wgrad_slabs_per_tap(a);
}

// ... and an upper bound from the shape alone (workspace sizing)
int wgrad_slabs_max(const WgradArgs& a) {
    int m = wgrad_slabs_per_tap(a);
    for (int cit = 1; cit <= 2; ++cit) {                    // a channel split may force the narrower ci tile
        const Wg3Plan pl = wgrad3x3_plan(a, 0, true, cit);
        if (pl.cot && pl.slabs > m) m = pl.slabs;
    }
    const Wg1Plan p1 = wgrad1x1_plan(a, 0, true);
    if (p1.cot && p1.slabs > m) m = p1.slabs;
    WgradArgs ap = a;                                       // the same shape as a patch matrix (the stem)
    ap.per_tap_only = 1;
    const Wg1Plan p2 = wgrad1x1_plan(ap, 0, true);
    if (p2.cot && p2.slabs > m) m = p2.slabs;
    return m;
}

static hipError_t launch_wgrad_impl(const WgradArgs& a, int precise, int slabs, int* final_layout, hipStream_t st);

// the convolution profile files weight-gradient launches under kind KH * 100 + 50 + precise (the slab sums that follow are separate,
// unrecorded launches); flops = 2 * taps * Cin * Cout * output pixels, the same count as the forward layer
hipError_t launch_wgrad(const WgradArgs& a, int precise, int slabs, int* final_layout, hipStream_t st) {
    const bool rec = g_cprof.enabled && g_cprof.count < g_cprof.capacity;
    const int slot = g_cprof.count;
    if (rec) {
        const double px = (double)a.N * a.Ho * a.Wo;
        g_cprof.flops[slot] = 2.0 * a.KH * a.KW * (double)a.Cin * a.Cout * px;
        g_cprof.kind[slot] = a.KH * 100 + 50 + (precise ? 1 : 0);
        g_cprof.shape[4 * slot + 0] = (int)(((long)a.N * a.Ho * a.Wo) >> 10);
        g_cprof.shape[4 * slot + 1] = a.Cin;
        g_cprof.shape[4 * slot + 2] = a.Cout;
        g_cprof.shape[4 * slot + 3] = a.stride * 10 + 1;
        ++g_cprof.count;
        (void)hipEventRecord(g_cprof.ev[2 * slot], st);
    }
    const hipError_t e = launch_wgrad_impl(a, precise, slabs, final_layout, st);
    if (rec) (void)hipEventRecord(g_cprof.ev[2 * slot + 1], st);
    return e;
}

static hipError_t launch_wgrad_impl(const WgradArgs& a, int precise, int slabs, int* final_layout, hipStream_t st) {
    const Wg3Plan pl = wgrad3x3_plan(a, precise, false);
    *final_layout = pl.cot ? 1 : 0;                          // 1: slabs already are [Cout][Cin][KH][KW]
    if (pl.cot && pl.stride == 2) {
        if (pl.cit == 2) wgrad3x3_launch_t<4, 2, 2>(a, pl, st);
        else wgrad3x3_launch_t<4, 1, 2>(a, pl, st);
        return hipGetLastError();
    }
    if (pl.cot) {
#define WG3_CASE(COT_, CIT_) \
    if (pl.cot == COT_ && pl.cit == CIT_) wgrad3x3_launch_t<COT_, CIT_>(a, pl, st);
        WG3_CASE(4, 2) WG3_CASE(2, 2) WG3_CASE(1, 2) WG3_CASE(4, 1) WG3_CASE(2, 1) WG3_CASE(1, 1)
#undef WG3_CASE
        return hipGetLastError();
    }
    const Wg1Plan p1 = wgrad1x1_plan(a, precise, false);
    if (p1.cot) {
        *final_layout = a.per_tap_only ? 0 : 1;             // [Cout][Cin] == [Cout][Cin][1][1]; a patch matrix: [Cout][1][columns], as the per-tap kernel
#define WG1_CASE(COT_, CIT_) \
    if (p1.cot == COT_ && p1.cit == CIT_) wgrad1x1_launch_t<COT_, CIT_>(a, p1, st);
        WG1_CASE(8, 4) WG1_CASE(8, 2) WG1_CASE(4, 4) WG1_CASE(4, 2) WG1_CASE(2, 4) WG1_CASE(2, 5) WG1_CASE(2, 2)
#undef WG1_CASE
        return hipGetLastError();
    }
    const int tm = a.Cout >= 128 ? 4 : (a.Cout >= 64 ? 2 : 1);
    const int tn = (a.Cin % 128 == 0) ? 4 : 1;
#define WG_CASE(TM_, TN_)                                               \
    if (tm == TM_ && tn == TN_) {                                       \
        if (precise) wgrad_launch_t<TM_, TN_, true>(a, slabs, st);      \
        else wgrad_launch_t<TM_, TN_, false>(a, slabs, st);             \
    }
    WG_CASE(4, 4) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 1) WG_CASE(1, 4) WG_CASE(1, 1)
#undef WG_CASE
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce(const float* partial, int slabs, int Cout, int Cin, int Cin_out, int KH, int KW, int im2col,
                               int accumulate, float* gw, hipStream_t st) {
    const long total = (long)Cout * (im2col ? 1 : KH * KW) * Cin;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, slabs, Cout, Cin, Cin_out, KH, KW,
                       im2col, accumulate, gw);
    return hipGetLastError();
}

}  // namespace vqseg
