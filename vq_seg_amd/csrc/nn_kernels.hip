// nn_kernels.hip -- HBM-bound companions of the convolution kernel (gfx950): batch-norm statistics /
// apply / backward, ReLU + residual fusion, 3x3 stride-2 max-pool, bilinear resampling (both corner
// conventions), the 1x1 segmentation head, stem im2col, reflect-padding gradient fold, dtype casts.
//
// Reference ops: nn.BatchNorm2d + nn.ReLU inside conv_bn_relu (models/networks/unet/decoder.py:7-10) and
// the ResNet blocks; F.interpolate(mode='bilinear') (decoder.py:35, align_corners=False);
// nn.UpsamplingBilinear2d (modified_vqunet/net.py:1172, align_corners=True); nn.MaxPool2d(3,2,1)
// (models/encoders/resnet.py:167); the 1x1 head conv (net.py:1169).
// All tensors are NHWC "rows" (pixel-major, channels contiguous).  T = float (precise mode) or __bf16.
// Reductions are two-level with fixed order (no float atomics): results are run-to-run deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "nn_kernels.h"

namespace vqseg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float ld(const T* p, long i) { return (float)p[i]; }
template <typename T>
__device__ __forceinline__ void st(T* p, long i, float v) { p[i] = (T)v; }

// 16-byte vector access: 4 floats or 8 bf16 per thread (index i = first element, 16-byte aligned)
template <typename T>
struct VecN { static constexpr int N = 16 / sizeof(T); };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void ldv(const T* p, long i, float (&v)[VecN<T>::N]) {
    const u32x4 raw = *reinterpret_cast<const u32x4*>(p + i);
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(raw[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(raw[e] << 16);
            v[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
        }
    }
}
template <typename T>
__device__ __forceinline__ void stv(T* p, long i, const float (&v)[VecN<T>::N]) {
    u32x4 raw;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)v[2 * e], b = (__bf16)v[2 * e + 1];
            raw[e] = (unsigned int)__builtin_bit_cast(unsigned short, a) | ((unsigned int)__builtin_bit_cast(unsigned short, b) << 16);
        }
    }
    *reinterpret_cast<u32x4*>(p + i) = raw;
}

#ifndef BN_UNROLL
#define BN_UNROLL 1               // vectors in flight per thread in the streaming BatchNorm loops (A/B builds: make ab ABFLAGS=-DBN_UNROLL=2)
#endif
// ---------------------------------------------------------------------------------------------
// BatchNorm forward statistics
// ---------------------------------------------------------------------------------------------
// Merge the per-slot partials (mean_s, M2_s over rows_per_slot rows) written by the conv epilogue.  In DOUBLE
// the plain power sums are exact enough (the inputs are floats: 29 spare bits against the cancellation in
// S2 - M mean^2) and need no divisions:
//   S1 = sum n_s mean_s        S2 = sum (M2_s + n_s mean_s^2)        mean = S1 / M        M2 = S2 - M mean^2
// Level 1 (bn_stats_merge_kernel): grid (64-channel groups, slot groups); a block sums its slots (reads coalesced
// along the channels, fixed order) and writes (S1, S2) back IN PLACE into its first two slots as (hi, lo) float
// pairs, so the whole merge stays in double.  Level 2 (bn_finalize_kernel) sums the group results and derives
// scale = gamma * invstd, shift = beta - mean * scale and the running-stat update (nn.BatchNorm2d: biased
// variance to normalise, unbiased for running_var).
// device-coherent accesses (sc1: written through / read past this XCD's L2) for results one workgroup hands to another inside a launch
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct PowerSums {
    double s1, s2;
};

// sum slots s0, s0 + step, ... < s1 for 64 channels; `merged`: the slots hold level-1 (hi, lo) results
template <bool COHERENT = false>                           // COHERENT: the slots were written by OTHER workgroups of this launch (st_agent)
__device__ __forceinline__ PowerSums fold_slots(const float* partial, long s0, long s1, long step, bool merged,
                                                int rows_per_slot, long M, int C, int c, PowerSums* lds) {
    const int ty = threadIdx.x >> 6, cl = threadIdx.x & 63;
    PowerSums a{0.0, 0.0};
    auto rd = [&](long idx) { return COHERENT ? ld_agent(partial + idx) : partial[idx]; };
    long i_first = s0 + ty * step;
    if (c < C && merged) {
        // level-2 fold (bn_finalize): up to 32 dependent-free iterations of four loads each were issued one iteration at a time
        // (16 us for a kernel of C / 64 workgroups on the critical path conv -> statistics -> apply); four iterations' loads in
        // flight, added in the same order
        for (; i_first + 12 * step + 1 < s1; i_first += 16 * step) {
            float u[4], v[4], ul[4], vl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long i = i_first + 4 * step * q;
                u[q] = rd((i * 2 + 0) * C + c);
                v[q] = rd((i * 2 + 1) * C + c);
                ul[q] = rd((i * 2 + 2) * C + c);
                vl[q] = rd((i * 2 + 3) * C + c);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a.s1 += (double)u[q];
                a.s2 += (double)v[q];
                a.s1 += (double)ul[q];
                a.s2 += (double)vl[q];
            }
        }
    }
    if (c < C)
        for (long i = i_first; i < s1; i += 4 * step) {
            const double u = (double)rd((i * 2 + 0) * C + c), v = (double)rd((i * 2 + 1) * C + c);
            if (merged) {
                a.s1 += u;
                a.s2 += v;
                if (i + 1 < s1) {                           // low parts in the next slot (groups hold >= 2 slots)
                    a.s1 += (double)rd((i * 2 + 2) * C + c);
                    a.s2 += (double)rd((i * 2 + 3) * C + c);
                }
            } else {
                long n = M - i * rows_per_slot;
                n = n < 0 ? 0 : (n > rows_per_slot ? rows_per_slot : n);
                a.s1 += (double)n * u;
                a.s2 += v + (double)n * u * u;
            }
        }
    lds[ty * 64 + cl] = a;
    __syncthreads();
    PowerSums r = lds[cl];
    if (ty == 0)
        for (int t = 1; t < 4; ++t) {
            r.s1 += lds[t * 64 + cl].s1;
            r.s2 += lds[t * 64 + cl].s2;
        }
    return r;                                               // valid on ty == 0
}

__global__ __launch_bounds__(256) void bn_stats_merge_kernel(float* __restrict__ partial, long n_slots, long slots_per_group,
                                                             int rows_per_slot, long M, int C) {
    __shared__ PowerSums lds[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const long s0 = (long)blockIdx.y * slots_per_group;
    long s1 = s0 + slots_per_group;
    if (s1 > n_slots) s1 = n_slots;
    const PowerSums r = fold_slots(partial, s0, s1, 1, false, rows_per_slot, M, C, c, lds);
    if ((threadIdx.x >> 6) == 0 && c < C) {                 // every read of this block's slots happened before the barrier
        const float h1 = (float)r.s1, h2 = (float)r.s2;
        partial[(s0 * 2 + 0) * C + c] = h1;
        partial[(s0 * 2 + 1) * C + c] = h2;
        if (s0 + 1 < s1) {
            partial[(s0 * 2 + 2) * C + c] = (float)(r.s1 - (double)h1);
            partial[(s0 * 2 + 3) * C + c] = (float)(r.s2 - (double)h2);
        }
    }
}

// "Last workgroup done" hand-over (used by the fused merge + finalize kernels below): every workgroup of a channel group publishes
// its result (device-scope release fence), then takes a ticket from sync[group]; the one that draws the last ticket acquires and
// runs the group's final step alone, in a fixed order -> deterministic, and one launch instead of two.  sync[] holds zeros between
// launches (the finishing workgroup resets its counter); the caller owns it and never shares it between concurrent launches.
// The results handed over travel through st_agent / ld_agent: device-coherent accesses (sc1: written through / read past this XCD's
// L2).  NOT __threadfence(): on gfx950 an agent-scope release / acquire fence writes back and invalidates the XCD's whole L2
// (buffer_wbl2 / buffer_inv), per workgroup -- measured: the step 165 -> 225 ms, mostly in the OTHER stream's kernels.
// The ordering this relies on (sc1 write-through stores performed at vmcnt(0), then a relaxed agent-scope ticket) is what gfx942 /
// gfx950 hardware does, NOT a guarantee of the HIP memory model: the path is opt-in (py_bn_fused, default off), and the host side
// zeroes a module's counters whenever one of the fused entry points returns an error (nnf._check_fused_bn).

__device__ __forceinline__ bool last_workgroup_of(int* sync, int group, unsigned total) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's coherent stores have been performed
    __syncthreads();                                        // ... and every wave's of the workgroup
    if (threadIdx.x == 0) {
        const unsigned t = (unsigned)__hip_atomic_fetch_add(sync + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == total - 1);
        if (s_last) __hip_atomic_store(sync + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    return s_last != 0;
}

__device__ __forceinline__ void bn_finalize_channel(const PowerSums& r, long M, int c, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                    float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const double mean = r.s1 / (double)M;
    double m2 = r.s2 - (double)M * mean * mean;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / (double)M;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (run_mean) {
        const double unbiased = M > 1 ? m2 / (double)(M - 1) : var;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
}

// bn_stats_merge_kernel + bn_finalize_kernel in ONE launch: grid (64-channel groups, slot groups); the last workgroup of a channel
// group folds the group results (same order, same arithmetic as bn_finalize_kernel: bit-identical statistics).
__global__ __launch_bounds__(256) void bn_stats_merge_finalize_kernel(float* partial, long n_slots, long slots_per_group, int rows_per_slot,
                                                                      long M, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                                      float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                      float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                      long long* __restrict__ num_batches_tracked, int* sync) {
    __shared__ PowerSums lds[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const long s0 = (long)blockIdx.y * slots_per_group;
    long s1 = s0 + slots_per_group;
    if (s1 > n_slots) s1 = n_slots;
    const PowerSums r = fold_slots(partial, s0, s1, 1, false, rows_per_slot, M, C, c, lds);
    if ((threadIdx.x >> 6) == 0 && c < C) {
        const float h1 = (float)r.s1, h2 = (float)r.s2;
        st_agent(partial + (s0 * 2 + 0) * C + c, h1);
        st_agent(partial + (s0 * 2 + 1) * C + c, h2);
        if (s0 + 1 < s1) {
            st_agent(partial + (s0 * 2 + 2) * C + c, (float)(r.s1 - (double)h1));
            st_agent(partial + (s0 * 2 + 3) * C + c, (float)(r.s2 - (double)h2));
        }
    }
    if (!last_workgroup_of(sync, blockIdx.x, gridDim.y)) return;
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;   // nn.BatchNorm2d bookkeeping
    __syncthreads();                                        // lds is reused
    const PowerSums t = fold_slots<true>(partial, 0, n_slots, slots_per_group, true, rows_per_slot, M, C, c, lds);
    if ((threadIdx.x >> 6) == 0 && c < C)
        bn_finalize_channel(t, M, c, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, long n_slots, long slots_per_group,
                                                          int rows_per_slot, long M, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float momentum, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          long long* __restrict__ num_batches_tracked) {
    __shared__ PowerSums lds[256];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;   // nn.BatchNorm2d bookkeeping
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const PowerSums r = fold_slots(partial, 0, n_slots, slots_per_group, slots_per_group > 1, rows_per_slot, M, C, c, lds);
    if ((threadIdx.x >> 6) == 0 && c < C)
        bn_finalize_channel(r, M, c, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd);
}

// eval mode: scale/shift from the running statistics
__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                             float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(run_var[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - run_mean[c] * sc;
    save_mean[c] = run_mean[c];
    save_invstd[c] = invstd;
}

// (stored value > 0) of V values as a bit field: the test runs on the value ROUNDED to T, i.e. on what `out > 0` reads back later
template <typename T, int V>
__device__ inline unsigned char relu_bits(const float (&v)[V]) {
    unsigned b = 0;
#pragma unroll
    for (int e = 0; e < V; ++e) b |= ((float)(T)v[e] > 0.0f ? 1u : 0u) << e;
    return (unsigned char)b;
}

// out = [relu]( y * scale[c] + shift[c] [+ res] )      (16 bytes per thread when C allows, else scalar)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       long M, int C, int relu, T* __restrict__ out,
                                                       unsigned char* __restrict__ bits = nullptr) {
    // bits (nullable; 8-element vectors only, i.e. bf16 with C % 8 == 0): bit e of bits[i / 8] = (out[i + e] > 0), the ReLU mask
    // the backward of a residual layer reads instead of `out` (1/16 of its bytes)
    constexpr int V = VecN<T>::N;
    const long total = M * C;
    if (C % V == 0 && (256 * V) % C == 0) {
        // the per-thread channel offset is invariant under the grid stride: keep scale/shift in registers
        const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * V;
        const int c = (int)(i0 % C);
        float sc[V], sh[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            sc[e] = scale[c + e];
            sh[e] = shift[c + e];
        }
#pragma unroll BN_UNROLL
        for (long i = i0; i < total; i += (long)gridDim.x * 256 * V) {
            float v[V], r[V];
            ldv(y, i, v);
            if (res) ldv(res, i, r);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float t = __builtin_fmaf(v[e], sc[e], sh[e]);
                if (res) t += r[e];
                v[e] = (relu && !(t > 0.0f)) ? 0.0f : t;
            }
            stv(out, i, v);
            if (V == 8 && bits) bits[i >> 3] = relu_bits<T, V>(v);
        }
    } else if (C % V == 0) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * V; i < total; i += (long)gridDim.x * 256 * V) {
            const int c = (int)(i % C);
            float v[V], r[V];
            ldv(y, i, v);
            if (res) ldv(res, i, r);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float t = __builtin_fmaf(v[e], scale[c + e], shift[c + e]);
                if (res) t += r[e];
                v[e] = (relu && !(t > 0.0f)) ? 0.0f : t;
            }
            stv(out, i, v);
            if (V == 8 && bits) bits[i >> 3] = relu_bits<T, V>(v);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C);
            float t = ld(y, i) * scale[c] + shift[c];
            if (res) t += ld(res, i);
            st(out, i, (relu && !(t > 0.0f)) ? 0.0f : t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward (through the fused ReLU / residual):  gz = g_out * (out > 0)
//   partial[blk][0][c] = sum gz,   partial[blk][1][c] = sum gz * xhat,   xhat = (y - mean) * invstd
// ---------------------------------------------------------------------------------------------
constexpr int BNB_ROWS_MIN = 256;  // rows per block of the reduce kernel (more for tall tensors: at most ~512 row blocks)
constexpr int BNB_CG = 64;         // channels per block

// grid (row chunks, channel groups).  Each thread owns V consecutive channels (one 16-byte load) of every
// (256 / threads-per-row)-th row of the chunk; row lanes are folded through LDS in a fixed order.
// MASK: 0 no ReLU, 1 mask from `out`, 2 mask recomputed as fma(y, fsc, fsh) > 0 (the forward's expression; no residual),
//       3 mask from the forward's bit field (`out` then points at bytes: bit e of byte i/8 = out[i + e] > 0; 8-element vectors only)
// Finalize arguments of the fused reduce (sync != null): see bn_bwd_finalize_kernel
struct BnBwdFin {
    const float* gamma;
    int training, accumulate;
    float* dgamma;
    float* dbeta;
    float* coef;
    int* sync;                                             // [channel groups], zeros between launches; null: separate finalize launch
};

template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ g_out, const T* __restrict__ out,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ fsc,
                                                            const float* __restrict__ fsh, long M, int C, int relu,
                                                            float* partial, int rows_per_block, const BnBwdFin fin, T* __restrict__ gz_out = nullptr) {
    constexpr int V = VecN<T>::N;
    __shared__ float sh[2][256 * V];                       // [sum kind][row lane][channel in group]
    const int cg = C < BNB_CG ? C : BNB_CG;                // channels handled by this block
    const int c0 = blockIdx.y * BNB_CG;
    const int BNB_ROWS = rows_per_block;
    const long row0 = (long)blockIdx.x * BNB_ROWS;
    float s0[V], s1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s0[e] = s1[e] = 0.0f;
    int tpr, lanes, rl = 0, cv = 0;
    if (C % V == 0) {
        tpr = cg / V;                                      // threads per row
        lanes = 256 / tpr;
        rl = threadIdx.x / tpr;
        cv = (threadIdx.x % tpr) * V;
        if (rl < lanes && c0 + cv < C) {
            float mu[V], is[V], sc[V], sf[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                mu[e] = mean[c0 + cv + e];
                is[e] = invstd[c0 + cv + e];
                sc[e] = MASK == 2 ? fsc[c0 + cv + e] : 0.0f;
                sf[e] = MASK == 2 ? fsh[c0 + cv + e] : 0.0f;
            }
#pragma unroll BN_UNROLL
            for (int rr = rl; rr < BNB_ROWS; rr += lanes) {
                const long m = row0 + rr;
                if (m >= M) break;
                float g[V], o[V], yy[V];
                ldv(g_out, m * C + c0 + cv, g);
                ldv(y, m * C + c0 + cv, yy);
                if (MASK == 1) ldv(out, m * C + c0 + cv, o);
                if (MASK == 3) {
                    const unsigned mb = reinterpret_cast<const unsigned char*>(out)[(m * C + c0 + cv) >> 3];
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] = (mb >> (e & 7)) & 1u ? 1.0f : 0.0f;
                }
                if (MASK == 2) {
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] = __builtin_fmaf(yy[e], sc[e], sf[e]);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float gz = (MASK != 0 && !(o[e] > 0.0f)) ? 0.0f : g[e];
                    s0[e] += gz;
                    s1[e] = __builtin_fmaf(gz, (yy[e] - mu[e]) * is[e], s1[e]);
                    g[e] = gz;
                }
                // r4 (residual layers, MASK == 1): the masked gradient IS the residual branch's gradient -- written here, the apply pass
                // then reads it instead of (g_out, out): one tensor read less per backward of a residual layer
                if ((MASK == 1 || MASK == 3) && gz_out) stv(gz_out, m * C + c0 + cv, g);
            }
        }
    } else {                                               // scalar fallback (odd channel counts): one channel per thread
        tpr = cg;
        lanes = 256 / tpr;
        rl = threadIdx.x / tpr;
        cv = threadIdx.x % tpr;
        if (rl < lanes && c0 + cv < C) {
            const float mu = mean[c0 + cv], is = invstd[c0 + cv];
            for (int rr = rl; rr < BNB_ROWS; rr += lanes) {
                const long m = row0 + rr;
                if (m >= M) break;
                float gz = ld(g_out, m * C + c0 + cv);
                const float yv = ld(y, m * C + c0 + cv);
                if (MASK != 0 && !((MASK == 1 ? ld(out, m * C + c0 + cv) : yv * fsc[c0 + cv] + fsh[c0 + cv]) > 0.0f)) gz = 0.0f;
                if (MASK == 1 && gz_out) st(gz_out, m * C + c0 + cv, gz);
                s0[0] += gz;
                s1[0] = __builtin_fmaf(gz, (yv - mu) * is, s1[0]);
            }
        }
    }
    const int vv = (C % V == 0) ? V : 1;
    if (rl < lanes) {
        for (int e = 0; e < vv; ++e) {
            sh[0][rl * cg + cv + e] = s0[e];
            sh[1][rl * cg + cv + e] = s1[e];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < cg && c0 + (int)threadIdx.x < C) {
        float a = 0.0f, b = 0.0f;
        for (int l = 0; l < lanes; ++l) {
            a += sh[0][l * cg + threadIdx.x];
            b += sh[1][l * cg + threadIdx.x];
        }
        if (fin.sync) {
            st_agent(partial + ((long)blockIdx.x * 2 + 0) * C + c0 + threadIdx.x, a);
            st_agent(partial + ((long)blockIdx.x * 2 + 1) * C + c0 + threadIdx.x, b);
        } else {
            partial[((long)blockIdx.x * 2 + 0) * C + c0 + threadIdx.x] = a;
            partial[((long)blockIdx.x * 2 + 1) * C + c0 + threadIdx.x] = b;
        }
    }
    if (!fin.sync) return;
    // ---- fused finalize (was bn_bwd_finalize_kernel, a second launch): the last workgroup of this channel group sums the row
    // blocks' partials in block order (4 row lanes x 64 channels, coalesced along the channels, lanes folded in lane order:
    // fixed order -> deterministic) and writes dgamma / dbeta / the apply pass's coefficients for its channels.
    if (!last_workgroup_of(fin.sync, blockIdx.y, gridDim.x)) return;
    __shared__ double fr[2][256];
    const int cl = threadIdx.x & 63, ty = threadIdx.x >> 6, c = c0 + cl;
    double a = 0.0, b = 0.0;
    if (c < C && cl < cg) {
        // eight row blocks' partials in flight at a time (device-coherent loads are not batched by the compiler), added in block order
        const long nbk = (long)gridDim.x;
        for (long i0 = ty; i0 < nbk; i0 += 32) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long i = i0 + 4 * u;
                va[u] = i < nbk ? ld_agent(partial + (i * 2 + 0) * C + c) : 0.0f;
                vb[u] = i < nbk ? ld_agent(partial + (i * 2 + 1) * C + c) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += (double)va[u];
                b += (double)vb[u];
            }
        }
    }
    fr[0][threadIdx.x] = a;
    fr[1][threadIdx.x] = b;
    __syncthreads();
    if (ty == 0 && c < C && cl < cg) {
        for (int t = 1; t < 4; ++t) {
            a += fr[0][t * 64 + cl];
            b += fr[1][t * 64 + cl];
        }
        fin.dbeta[c] = fin.accumulate ? fin.dbeta[c] + (float)a : (float)a;
        fin.dgamma[c] = fin.accumulate ? fin.dgamma[c] + (float)b : (float)b;
        fin.coef[0 * C + c] = fin.gamma[c] * invstd[c];
        fin.coef[1 * C + c] = fin.training ? (float)(a / (double)M) : 0.0f;
        fin.coef[2 * C + c] = fin.training ? (float)(b / (double)M) : 0.0f;
    }
}

// dgamma = sum gz*xhat, dbeta = sum gz; coefficients of the apply pass:
//   train: g_y = k0[c] * (gz - k1[c] - xhat * k2[c]),  k0 = gamma*invstd, k1 = mean(gz), k2 = mean(gz*xhat)
//   eval : g_y = k0[c] * gz
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, long n_blocks, long M, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              int training, int accumulate, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x;
    __shared__ double r0[256], r1[256];
    double a = 0.0, b = 0.0;
    long i = threadIdx.x;
    for (; i + 768 < n_blocks; i += 1024) {                // four row blocks' partials in flight (latency bound), same order of adds
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = partial[((i + 256 * u) * 2 + 0) * C + c];
            vb[u] = partial[((i + 256 * u) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a += (double)va[u];
            b += (double)vb[u];
        }
    }
    for (; i < n_blocks; i += 256) {
        a += (double)partial[(i * 2 + 0) * C + c];
        b += (double)partial[(i * 2 + 1) * C + c];
    }
    r0[threadIdx.x] = a;
    r1[threadIdx.x] = b;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) {
            r0[threadIdx.x] += r0[threadIdx.x + m];
            r1[threadIdx.x] += r1[threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dbeta[c] = accumulate ? dbeta[c] + (float)r0[0] : (float)r0[0];
        dgamma[c] = accumulate ? dgamma[c] + (float)r1[0] : (float)r1[0];
        coef[0 * C + c] = gamma[c] * invstd[c];
        coef[1 * C + c] = training ? (float)(r0[0] / (double)M) : 0.0f;
        coef[2 * C + c] = training ? (float)(r1[0] / (double)M) : 0.0f;
    }
}

template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g_out, const T* __restrict__ out,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ coef,
                                                           const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                           long M, int C, int relu, T* __restrict__ g_y, T* __restrict__ g_res) {
    constexpr int V = VecN<T>::N;
    const long total = M * C;
    if (C % V == 0 && (256 * V) % C == 0) {
        const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * V;
        const int c = (int)(i0 % C);
        float mu[V], is[V], k0[V], k1[V], k2[V], sc[V], sf[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            mu[e] = mean[c + e];
            is[e] = invstd[c + e];
            k0[e] = coef[c + e];
            k1[e] = coef[C + c + e];
            k2[e] = coef[2 * C + c + e];
            sc[e] = MASK == 2 ? fsc[c + e] : 0.0f;
            sf[e] = MASK == 2 ? fsh[c + e] : 0.0f;
        }
#pragma unroll BN_UNROLL
        for (long i = i0; i < total; i += (long)gridDim.x * 256 * V) {
            float g[V], o[V], yy[V], r[V];
            ldv(g_out, i, g);
            ldv(y, i, yy);
            if (MASK == 1) ldv(out, i, o);
            if (MASK == 2) {
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = __builtin_fmaf(yy[e], sc[e], sf[e]);
            }
            if (MASK == 3) {                                // the forward's bit field (see bn_bwd_reduce_kernel)
                const unsigned mb = reinterpret_cast<const unsigned char*>(out)[i >> 3];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (mb >> (e & 7)) & 1u ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float gz = (MASK != 0 && !(o[e] > 0.0f)) ? 0.0f : g[e];
                const float xh = (yy[e] - mu[e]) * is[e];
                r[e] = gz;
                g[e] = k0[e] * (gz - k1[e] - xh * k2[e]);
            }
            stv(g_y, i, g);
            if (g_res) stv(g_res, i, r);
        }
    } else if (C % V == 0) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * V; i < total; i += (long)gridDim.x * 256 * V) {
            const int c = (int)(i % C);
            float g[V], o[V], yy[V], r[V];
            ldv(g_out, i, g);
            ldv(y, i, yy);
            if (MASK == 1) ldv(out, i, o);
            if (MASK == 2) {
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = __builtin_fmaf(yy[e], fsc[c + e], fsh[c + e]);
            }
            if (MASK == 3) {
                const unsigned mb = reinterpret_cast<const unsigned char*>(out)[i >> 3];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (mb >> (e & 7)) & 1u ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float gz = (MASK != 0 && !(o[e] > 0.0f)) ? 0.0f : g[e];
                const float xh = (yy[e] - mean[c + e]) * invstd[c + e];
                r[e] = gz;
                g[e] = coef[c + e] * (gz - coef[C + c + e] - xh * coef[2 * C + c + e]);
            }
            stv(g_y, i, g);
            if (g_res) stv(g_res, i, r);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C);
            float g = ld(g_out, i);
            const float yv = ld(y, i);
            if (MASK != 0 && !((MASK == 1 ? ld(out, i) : yv * fsc[c] + fsh[c]) > 0.0f)) g = 0.0f;
            const float xh = (yv - mean[c]) * invstd[c];
            st(g_y, i, coef[c] * (g - coef[C + c] - xh * coef[2 * C + c]));
            if (g_res) st(g_res, i, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// max-pool 3x3 stride 2 pad 1 (NHWC).  Backward recomputes the arg-max (first maximum in window scan
// order kh, kw -- the order ATen's max_pool2d uses) and gathers: deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
// Each thread handles VC consecutive channels of one pixel (VC = 16-byte vector when C allows, else 1).
template <typename T, int VC>
__device__ __forceinline__ void ldc(const T* p, long i, float (&v)[VecN<T>::N]) {
    if constexpr (VC == 1) v[0] = ld(p, i);
    else if constexpr (VC == 3) {                          // the 3-class logits: one pixel per thread, three scalar accesses
#pragma unroll
        for (int e = 0; e < 3; ++e) v[e] = ld(p, i + e);
    } else ldv(p, i, v);
}
template <typename T, int VC>
__device__ __forceinline__ void stc(T* p, long i, const float (&v)[VecN<T>::N]) {
    if constexpr (VC == 1) st(p, i, v[0]);
    else if constexpr (VC == 3) {
#pragma unroll
        for (int e = 0; e < 3; ++e) st(p, i + e, v[e]);
    } else stv(p, i, v);
}

// flat index -> (channel group, x, y, image): 32-bit divisions whenever the index fits (it does for every tensor of the
// model: 64-bit integer division costs more ALU time than these copy-like kernels spend on memory)
__device__ __forceinline__ void split_nhwc(long i, int cv, int W, int H, int& c, int& w, int& h, int& n) {
    if (i < (1L << 31)) {
        unsigned u = (unsigned)i;
        c = (int)(u % (unsigned)cv);
        u /= (unsigned)cv;
        w = (int)(u % (unsigned)W);
        u /= (unsigned)W;
        h = (int)(u % (unsigned)H);
        n = (int)(u / (unsigned)H);
    } else {
        c = (int)(i % cv);
        long t = i / cv;
        w = (int)(t % W);
        t /= W;
        h = (int)(t % H);
        n = (int)(t / H);
    }
}

template <typename T, int VC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                                          T* __restrict__ y, unsigned char* __restrict__ idx) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, ow, oh, n;
        split_nhwc(i, cv, Wo, Ho, c, ow, oh, n);
        c *= VC;
        float best[V];
        unsigned char bpos[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) {
            best[e] = -__builtin_inff();
            bpos[e] = 255;
        }
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                float v[V];
                ldc<T, VC>(x, (((long)n * H + ih) * W + iw) * C + c, v);
#pragma unroll
                for (int e = 0; e < VC; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bpos[e] = (unsigned char)(kh * 3 + kw);
                    }
            }
        }
        const long o = (((long)n * Ho + oh) * Wo + ow) * C + c;
        stc<T, VC>(y, o, best);
        if (idx) {                                          // window position (kh * 3 + kw) of the maximum, for the backward
#pragma unroll
            for (int e = 0; e < VC; ++e) idx[o + e] = bpos[e];
        }
    }
}

// backward from the saved window positions: an input pixel gathers g of the (at most four) windows whose maximum it is
template <typename T, int VC>
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, const T* __restrict__ g, int N,
                                                              int H, int W, int C, int Ho, int Wo, T* __restrict__ gx, const T* __restrict__ gx_add) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * H * W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, iw, ih, n;
        split_nhwc(i, cv, W, H, c, iw, ih, n);
        c *= VC;
        float acc[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) acc[e] = 0.0f;
        for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
            const int kh = ih - (oh * 2 - 1);
            if (kh < 0 || kh > 2) continue;
            for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
                const int kw = iw - (ow * 2 - 1);
                if (kw < 0 || kw > 2) continue;
                const long o = (((long)n * Ho + oh) * Wo + ow) * C + c;
                const unsigned char me = (unsigned char)(kh * 3 + kw);
                unsigned char pos[V];
                if constexpr (VC == 8) {
                    const unsigned long long raw = *reinterpret_cast<const unsigned long long*>(idx + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) pos[e] = (unsigned char)(raw >> (8 * e));
                } else if constexpr (VC == 4) {
                    const unsigned int raw = *reinterpret_cast<const unsigned int*>(idx + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pos[e] = (unsigned char)(raw >> (8 * e));
                } else {
                    pos[0] = idx[o];
                }
                bool any = false;
#pragma unroll
                for (int e = 0; e < VC; ++e) any = any || pos[e] == me;
                if (!any) continue;
                float gv[V];
                ldc<T, VC>(g, o, gv);
#pragma unroll
                for (int e = 0; e < VC; ++e)
                    if (pos[e] == me) acc[e] += gv[e];
            }
        }
        if (gx_add) {                                       // fan-in: the pooled tensor's other consumer (autograd's add: both rounded to T first)
            float ad[V];
            ldc<T, VC>(gx_add, (((long)n * H + ih) * W + iw) * C + c, ad);
#pragma unroll
            for (int e = 0; e < VC; ++e) acc[e] = (float)(T)acc[e] + ad[e];
        }
        stc<T, VC>(gx, (((long)n * H + ih) * W + iw) * C + c, acc);
    }
}

template <typename T, int VC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, int N, int H, int W,
                                                          int C, int Ho, int Wo, T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * H * W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, iw, ih, n;
        split_nhwc(i, cv, W, H, c, iw, ih, n);
        c *= VC;
        float acc[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) acc[e] = 0.0f;
        // output windows that contain (ih, iw): oh in [ih/2 .. (ih+1)/2]
        for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
            if (ih < oh * 2 - 1 || ih > oh * 2 + 1) continue;
            for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
                if (iw < ow * 2 - 1 || iw > ow * 2 + 1) continue;
                // arg-max of this window (first maximum in scan order), per channel
                float best[V];
                int bpos[V];
#pragma unroll
                for (int e = 0; e < VC; ++e) {
                    best[e] = -__builtin_inff();
                    bpos[e] = -1;
                }
                for (int kh = 0; kh < 3; ++kh) {
                    const int jh = oh * 2 - 1 + kh;
                    if (jh < 0 || jh >= H) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        const int jw = ow * 2 - 1 + kw;
                        if (jw < 0 || jw >= W) continue;
                        float v[V];
                        ldc<T, VC>(x, (((long)n * H + jh) * W + jw) * C + c, v);
#pragma unroll
                        for (int e = 0; e < VC; ++e)
                            if (v[e] > best[e] || v[e] != v[e]) {
                                best[e] = v[e];
                                bpos[e] = jh * W + jw;
                            }
                    }
                }
                float gv[V];
                ldc<T, VC>(g, (((long)n * Ho + oh) * Wo + ow) * C + c, gv);
#pragma unroll
                for (int e = 0; e < VC; ++e)
                    if (bpos[e] == ih * W + iw) acc[e] += gv[e];
            }
        }
        stc<T, VC>(gx, (((long)n * H + ih) * W + iw) * C + c, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// bilinear resize (NHWC), both corner conventions (ATen upsample_bilinear2d semantics)
//   align_corners: src = dst * (in-1)/(out-1);  else src = max((dst+0.5)*in/out - 0.5, 0)
// ---------------------------------------------------------------------------------------------
// (1 - lh) * ((1 - lw) * v00 + lw * v01) + lh * ((1 - lw) * v10 + lw * v11) with the multiply-adds spelled out: the generic and the
// exact-2x kernels (and their split-3 forms) must round identically, and the compiler's own choice of which product of a sum to fuse
// differs between kernels whose weights are run-time values and kernels whose weights are constants
__device__ __forceinline__ float bil_mix(float v00, float v01, float v10, float v11, float lh, float lw) {
    const float top = __builtin_fmaf(lw, v01, (1.0f - lw) * v00);
    const float bot = __builtin_fmaf(lw, v11, (1.0f - lw) * v10);
    return __builtin_fmaf(lh, bot, (1.0f - lh) * top);
}

__device__ __forceinline__ void bil_src(int d, int in, int out, int align, int& i0, int& i1, float& l1) {
    float s;
    if (align) s = out > 1 ? (float)d * ((float)(in - 1) / (float)(out - 1)) : 0.0f;
    else {
        s = ((float)d + 0.5f) * ((float)in / (float)out) - 0.5f;
        if (s < 0.0f) s = 0.0f;
    }
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

template <typename T, int VC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                                           int align, T* __restrict__ y) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, ow, oh, n;
        split_nhwc(i, cv, Wo, Ho, c, ow, oh, n);
        c *= VC;
        int h0, h1, w0, w1;
        float lh, lw;
        bil_src(oh, H, Ho, align, h0, h1, lh);
        bil_src(ow, W, Wo, align, w0, w1, lw);
        const long b = (long)n * H;
        float v00[V], v01[V], v10[V], v11[V], o[V];
        ldc<T, VC>(x, ((b + h0) * W + w0) * C + c, v00);
        ldc<T, VC>(x, ((b + h0) * W + w1) * C + c, v01);
        ldc<T, VC>(x, ((b + h1) * W + w0) * C + c, v10);
        ldc<T, VC>(x, ((b + h1) * W + w1) * C + c, v11);
#pragma unroll
        for (int e = 0; e < VC; ++e)
            o[e] = bil_mix(v00[e], v01[e], v10[e], v11[e], lh, lw);
        stc<T, VC>(y, (((long)n * Ho + oh) * Wo + ow) * C + c, o);
    }
}

// backward as a gather over the (few) output pixels whose footprint touches the input pixel
template <typename T, int VC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ g, int N, int H, int W, int C, int Ho, int Wo,
                                                           int align, T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * H * W * cv;
    const int rh = (Ho + H - 1) / H + 1, rw = (Wo + W - 1) / W + 1;       // search radius in output pixels
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, iw, ih, n;
        split_nhwc(i, cv, W, H, c, iw, ih, n);
        c *= VC;
        const int ohc = (int)(((unsigned)ih * (unsigned)Ho) / (unsigned)H), owc = (int)(((unsigned)iw * (unsigned)Wo) / (unsigned)W);
        float acc[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) acc[e] = 0.0f;
        for (int oh = ohc - rh; oh <= ohc + rh; ++oh) {
            if (oh < 0 || oh >= Ho) continue;
            int h0, h1;
            float lh;
            bil_src(oh, H, Ho, align, h0, h1, lh);
            float wh = 0.0f;
            if (h0 == ih) wh += 1.0f - lh;
            if (h1 == ih) wh += lh;
            if (wh == 0.0f) continue;
            for (int ow = owc - rw; ow <= owc + rw; ++ow) {
                if (ow < 0 || ow >= Wo) continue;
                int w0, w1;
                float lw;
                bil_src(ow, W, Wo, align, w0, w1, lw);
                float ww = 0.0f;
                if (w0 == iw) ww += 1.0f - lw;
                if (w1 == iw) ww += lw;
                if (ww == 0.0f) continue;
                float gv[V];
                ldc<T, VC>(g, (((long)n * Ho + oh) * Wo + ow) * C + c, gv);
#pragma unroll
                for (int e = 0; e < VC; ++e) acc[e] = __builtin_fmaf(wh * ww, gv[e], acc[e]);
            }
        }
        stc<T, VC>(gx, (((long)n * H + ih) * W + iw) * C + c, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// Exact 2x up-sampling with align_corners = false (the decoder's resize to the next skip, decoder.py:35, whenever the skip is twice
// the size -- every power-of-two input) on power-of-two extents.  The generic kernels above are VALU-ISSUE bound (rocprofv3
// SQ_INSTS_VALU: issue time / kernel time 0.95 forward, 1.07 backward -- profiles/r03_step_valu_issue.txt): per vector they
// spend three run-time integer divisions on the index split, and the backward searches 7 x 7 candidate output pixels through
// bil_src() for the (at most) 4 x 4 that touch an input pixel.  At scale 2 the source coordinate of output o is o / 2 - 1/4:
//   o = 2 i (> 0): taps (i - 1, i) with weights (1/4, 3/4);   o = 2 i + 1: taps (i, min(i + 1, in - 1)) with (3/4, 1/4);   o = 0: tap 0
// -- all exactly representable, so these kernels evaluate the SAME expressions on the SAME values in the SAME order as the generic
// ones (bit-identical; tests/test_nn_gpu.py compares them), with shifts for the index split.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int o, int in, int& i0, int& i1, float& l1) {
    const int i = o >> 1;
    if (o & 1) {
        i0 = i;
        i1 = i + (i < in - 1 ? 1 : 0);
        l1 = 0.25f;
    } else if (o == 0) {
        i0 = 0;
        i1 = in > 1 ? 1 : 0;
        l1 = 0.0f;
    } else {
        i0 = i - 1;
        i1 = i;
        l1 = 0.75f;
    }
}

template <typename T, int VC>
__global__ __launch_bounds__(256) void bilinear_up2_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int lcv, int lwo, int lho,
                                                               T* __restrict__ y) {
    constexpr int V = VecN<T>::N;
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = ((long)N << (lcv + lwo + lho));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = ((int)i & ((1 << lcv) - 1)) * VC;
        const int ow = (int)(i >> lcv) & (Wo - 1), oh = (int)(i >> (lcv + lwo)) & (Ho - 1), n = (int)(i >> (lcv + lwo + lho));
        int h0, h1, w0, w1;
        float lh, lw;
        up2_src(oh, H, h0, h1, lh);
        up2_src(ow, W, w0, w1, lw);
        const long b = (long)n * H;
        float v00[V], v01[V], v10[V], v11[V], o[V];
        ldc<T, VC>(x, ((b + h0) * W + w0) * C + c, v00);
        ldc<T, VC>(x, ((b + h0) * W + w1) * C + c, v01);
        ldc<T, VC>(x, ((b + h1) * W + w0) * C + c, v10);
        ldc<T, VC>(x, ((b + h1) * W + w1) * C + c, v11);
#pragma unroll
        for (int e = 0; e < VC; ++e)
            o[e] = bil_mix(v00[e], v01[e], v10[e], v11[e], lh, lw);
        stc<T, VC>(y, i * VC, o);                             // i enumerates (n, oh, ow, c / VC): the output's own order
    }
}

template <typename T, int VC>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const T* __restrict__ g, int N, int H, int W, int C, int lcv, int lw_, int lh_,
                                                               T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = ((long)N << (lcv + lw_ + lh_));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = ((int)i & ((1 << lcv) - 1)) * VC;
        const int iw = (int)(i >> lcv) & (W - 1), ih = (int)(i >> (lcv + lw_)) & (H - 1), n = (int)(i >> (lcv + lw_ + lh_));
        float acc[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) acc[e] = 0.0f;
        // output rows 2 ih - 1 .. 2 ih + 2 touch input row ih with weights 1/4, 3/4, 3/4, 1/4; at the borders the clamped neighbour folds
        // onto the same row: output 0 gives row 0 weight 1, output 2 H - 1 gives row H - 1 weight 3/4 + 1/4 (the generic kernel's sums)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oh = 2 * ih - 1 + a;
            if (oh < 0 || oh >= Ho) continue;
            float wh = (a == 0 || a == 3) ? 0.25f : 0.75f;
            if (a == 1 && ih == 0) wh = 1.0f;
            if (a == 2 && ih == H - 1) wh = 0.75f + 0.25f;
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                const int ow = 2 * iw - 1 + bq;
                if (ow < 0 || ow >= Wo) continue;
                float ww = (bq == 0 || bq == 3) ? 0.25f : 0.75f;
                if (bq == 1 && iw == 0) ww = 1.0f;
                if (bq == 2 && iw == W - 1) ww = 0.75f + 0.25f;
                float gv[V];
                ldc<T, VC>(g, (((long)n * Ho + oh) * Wo + ow) * C + c, gv);
#pragma unroll
                for (int e = 0; e < VC; ++e) acc[e] = __builtin_fmaf(wh * ww, gv[e], acc[e]);
            }
        }
        stc<T, VC>(gx, i * VC, acc);
    }
}

// r4: R input rows per thread.  The one-row form reads four gradient rows for two fresh ones and leaves the re-use of the shared pair to
// the L2 (measured: 1.7x the tensor fetched past the L2s); here a thread walks 2R + 2 rows for 2R fresh ones (R = 4: 1.25x at worst).
// Every output's own fmaf chain runs in the same order (row, then column, ascending) as above: bit-identical.
template <typename T, int VC, int R>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_rows_kernel(const T* __restrict__ g, int N, int H, int W, int C, int lcv, int lw_, int lhr,
                                                                    T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = ((long)N << (lcv + lw_ + lhr));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = ((int)i & ((1 << lcv) - 1)) * VC;
        const int iw = (int)(i >> lcv) & (W - 1), pr = (int)(i >> (lcv + lw_)) & ((H / R) - 1), n = (int)(i >> (lcv + lw_ + lhr));
        float acc[R][V];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < VC; ++e) acc[r][e] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2 * R + 2; ++t) {
            const int oh = 2 * R * pr - 1 + t;
            if (oh < 0 || oh >= Ho) continue;
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                const int ow = 2 * iw - 1 + bq;
                if (ow < 0 || ow >= Wo) continue;
                float ww = (bq == 0 || bq == 3) ? 0.25f : 0.75f;
                if (bq == 1 && iw == 0) ww = 1.0f;
                if (bq == 2 && iw == W - 1) ww = 0.75f + 0.25f;
                float gv[V];
                ldc<T, VC>(g, (((long)n * Ho + oh) * Wo + ow) * C + c, gv);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int a = t - 2 * r;                 // this gradient row's position in output row r's window
                    if (a < 0 || a > 3) continue;
                    const int ih = R * pr + r;
                    float wh = (a == 0 || a == 3) ? 0.25f : 0.75f;
                    if (a == 1 && ih == 0) wh = 1.0f;
                    if (a == 2 && ih == H - 1) wh = 0.75f + 0.25f;
#pragma unroll
                    for (int e = 0; e < VC; ++e) acc[r][e] = __builtin_fmaf(wh * ww, gv[e], acc[r][e]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) stc<T, VC>(gx, (((long)n * H + R * pr + r) * W + iw) * C + c, acc[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 segmentation head (Cin <= 64 -> Cout <= 4, no bias): forward, data gradient, weight gradient
// ---------------------------------------------------------------------------------------------
// 8 consecutive channels of row `row` (Cin per row) as floats: T = float / __bf16 rows, or split-3 rows (S3Row tag: [hi | lo])
struct S3Row { unsigned short raw; };
template <typename T>
__device__ __forceinline__ void ld8(const T* __restrict__ x, long row, int Cin, int c, float (&v)[8]) {
    if constexpr (__is_same(T, S3Row)) {
        const unsigned short* r = reinterpret_cast<const unsigned short*>(x) + row * 2 * Cin;
        const u32x4 h = *reinterpret_cast<const u32x4*>(r + c), l = *reinterpret_cast<const u32x4*>(r + Cin + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, h[e] << 16) + __builtin_bit_cast(float, l[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, h[e] & 0xFFFF0000u) + __builtin_bit_cast(float, l[e] & 0xFFFF0000u);
        }
    } else if constexpr (sizeof(T) == 2) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(x + row * Cin + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, h[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, h[e] & 0xFFFF0000u);
        }
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * Cin + c), b = *reinterpret_cast<const f32x4*>(x + row * Cin + c + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
}

// forward: one pixel row per thread, 16-byte loads, the weights in LDS; the channel order of the fmaf chain is ascending (as the
// scalar form it replaces: bit-identical results).  Cin % 8 == 0, Cin <= 64, Cout <= 4.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, long M, int Cin,
                                                       int Cout, float* __restrict__ y) {
    __shared__ float ws[4 * 64];
    for (int i = threadIdx.x; i < Cout * Cin; i += 256) ws[i] = w[i];
    __syncthreads();
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; c += 8) {
            float v[8];
            ld8<T>(x, m, Cin, c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < Cout) acc[o] = __builtin_fmaf(v[e], ws[o * Cin + c + e], acc[o]);
        }
        for (int o = 0; o < Cout; ++o) y[m * Cout + o] = acc[o];
    }
}

// data gradient: thread = (row, 8-channel chunk), one 16-byte (bf16) / two 16-byte (f32) stores
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w, long M,
                                                            int Cin, int Cout, T* __restrict__ gx, const T* __restrict__ gx_add) {
    __shared__ float ws[4 * 64];
    for (int i = threadIdx.x; i < Cout * Cin; i += 256) ws[i] = w[i];
    __syncthreads();
    const int cv = Cin / 8;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cv;
        const int c = (int)(i - m * cv) * 8;
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < Cout; ++o) gv[o] = g[m * Cout + o];
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.0f;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < Cout) a = __builtin_fmaf(gv[o], ws[o * Cin + c + e], a);
            out[e] = a;
        }
        if (gx_add) {
            // fan-in: the other consumer's gradient of the same tensor, added the way autograd would add the two (each rounded to T first)
            float ad[8];
            if constexpr (sizeof(T) == 2) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(gx_add + m * Cin + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ad[2 * e] = __builtin_bit_cast(float, r[e] << 16);
                    ad[2 * e + 1] = __builtin_bit_cast(float, r[e] & 0xFFFF0000u);
                }
            } else {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(gx_add + m * Cin + c), a1 = *reinterpret_cast<const f32x4*>(gx_add + m * Cin + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) ad[e] = a0[e], ad[4 + e] = a1[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (float)(T)out[e] + ad[e];
        }
        if constexpr (sizeof(T) == 2) {
            u32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 lo = (__bf16)out[2 * e], hi = (__bf16)out[2 * e + 1];
                r[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
            }
            *reinterpret_cast<u32x4*>(gx + m * Cin + c) = r;
        } else {
            *reinterpret_cast<f32x4*>(gx + m * Cin + c) = f32x4{out[0], out[1], out[2], out[3]};
            *reinterpret_cast<f32x4*>(gx + m * Cin + c + 4) = f32x4{out[4], out[5], out[6], out[7]};
        }
    }
}

constexpr int HEAD_ROWS = 2048;
// weight gradient: block -> HEAD_ROWS rows; thread -> (row slot, 8-channel chunk): one 16-byte x load feeds 8 x Cout accumulators;
// the 256 / (Cin / 8) row slots are folded through LDS in slot order (deterministic); partial[blk][Cout * Cin]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ g, long M, int Cin,
                                                              int Cout, float* __restrict__ partial) {
    extern __shared__ float sh[];                              // [slots][Cout * Cin]
    const int cv = Cin / 8, slots = 256 / cv;
    const int cc = threadIdx.x % cv, sl = threadIdx.x / cv;
    const long row0 = (long)blockIdx.x * HEAD_ROWS;
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.0f;
    if (sl < slots) {
        long end = row0 + HEAD_ROWS;
        if (end > M) end = M;
        for (long m = row0 + sl; m < end; m += slots) {
            float v[8];
            ld8<T>(x, m, Cin, cc * 8, v);
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < Cout) {
                    const float gv = g[m * Cout + o];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o][e] = __builtin_fmaf(gv, v[e], acc[o][e]);
                }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cout)
#pragma unroll
                for (int e = 0; e < 8; ++e) sh[(sl * Cout + o) * Cin + cc * 8 + e] = acc[o][e];
    }
    __syncthreads();
    if ((int)threadIdx.x < Cin * Cout) {
        float s = 0.0f;
        for (int l = 0; l < slots; ++l) s += sh[l * Cout * Cin + threadIdx.x];
        partial[(long)blockIdx.x * Cin * Cout + threadIdx.x] = s;
    }
}

// sum partial[blk][n] over blk -> out[n] (double accumulate, fixed order).  A block covers 64 x 4 consecutive
// elements (float4 per thread); its 4 thread rows take the slabs b = row, row + 4, ... (two independent chains
// each, so several loads are in flight) and are folded through LDS in row order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, long n_blocks, long n,
                                                              float* __restrict__ out, int accumulate) {
    __shared__ double sh[4][64][4];
    const int col = threadIdx.x & 63, rowl = threadIdx.x >> 6;
    const long i = ((long)blockIdx.x * 64 + col) * 4;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    const bool vec = (n & 3) == 0 && i + 3 < n;      // 16-byte aligned slab rows
    if (i < n) {
        long b = rowl;
        // the slab sums are latency bound (a thread's loads are its only work): four independent 16-byte loads in flight per
        // thread, added in the SAME order as the two-load loop below (s0: slabs b, b + 8, ...; s1: b + 4, b + 12, ...)
        if (vec)
            for (; b + 12 < n_blocks; b += 16) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(partial + b * n + i);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + (b + 4) * n + i);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(partial + (b + 8) * n + i);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(partial + (b + 12) * n + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[e] += (double)u0[e];
                    s1[e] += (double)v0[e];
                    s0[e] += (double)u1[e];
                    s1[e] += (double)v1[e];
                }
            }
        for (; b + 4 < n_blocks; b += 8) {
            if (vec) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(partial + b * n + i);
                const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (b + 4) * n + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[e] += (double)u[e];
                    s1[e] += (double)v[e];
                }
            } else {
                for (int e = 0; e < 4 && i + e < n; ++e) {
                    s0[e] += (double)partial[b * n + i + e];
                    s1[e] += (double)partial[(b + 4) * n + i + e];
                }
            }
        }
        if (b < n_blocks) {
            if (vec) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(partial + b * n + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) s0[e] += (double)u[e];
            } else {
                for (int e = 0; e < 4 && i + e < n; ++e) s0[e] += (double)partial[b * n + i + e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[rowl][col][e] = s0[e] + s1[e];
    __syncthreads();
    if (rowl == 0 && i < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e < n) {
                const double t = ((sh[0][col][e] + sh[1][col][e]) + sh[2][col][e]) + sh[3][col][e];
                out[i + e] = accumulate ? out[i + e] + (float)t : (float)t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stem im2col: x [N,H,W,3] -> rows [N*Ho*Wo][Kp] with the 7x7x3 patch (kh, kw, ci order), stride 2, pad 3
// (zero or reflect), columns >= 147 zero.  The stem then runs as a 1x1 convolution with Cin = Kp.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col_stem_kernel(const TI* __restrict__ x, int N, int H, int W, int Cin, int KH, int KW,
                                                          int stride, int pad, int reflect, int Ho, int Wo, int Kp,
                                                          TO* __restrict__ out) {
    const long total = (long)N * Ho * Wo * Kp;
    const int K = KH * KW * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        long t = i / Kp;
        const int ow = (int)(t % Wo);
        t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float v = 0.0f;
        if (k < K) {
            const int ci = k % Cin, kw = (k / Cin) % KW, kh = k / (Cin * KW);
            int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
            if (reflect) {
                if (ih < 0) ih = -ih;
                if (ih >= H) ih = 2 * H - 2 - ih;
                if (iw < 0) iw = -iw;
                if (iw >= W) iw = 2 * W - 2 - iw;
            }
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = (float)x[(((long)n * H + ih) * W + iw) * Cin + ci];
        }
        out[i] = (TO)v;
    }
}

struct S3Out { unsigned short raw; };                      // output tag of im2col_stem7_kernel: split-3 rows (see s3_* below)
__device__ __forceinline__ void s3_store8(unsigned short* row, int C, int c, const float (&v)[8]);

// the stem's own shape (7x7x3, compile-time divisors), 8 columns = one 16-byte (bf16) / two 16-byte (f32) stores per thread
template <typename TO>
__global__ __launch_bounds__(256) void im2col_stem7_kernel(const float* __restrict__ x, int N, int H, int W, int stride, int pad,
                                                           int reflect, int Ho, int Wo, int Kp, TO* __restrict__ out) {
    constexpr int KW = 7, CIN = 3, K = 147;
    const unsigned cpr = Kp / 8;                            // 8-column chunks per row
    const unsigned total = (unsigned)N * Ho * Wo * cpr;     // < 2^32 (checked by the launcher): 32-bit index arithmetic --
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {   // 64-bit divisions cost more than the copy
        const int ch = (int)(i % cpr);
        unsigned t = i / cpr;
        const int ow = (int)(t % (unsigned)Wo);
        t /= (unsigned)Wo;
        const int oh = (int)(t % (unsigned)Ho);
        const int n = (int)(t / (unsigned)Ho);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            v[e] = 0.0f;
            if (k < K) {
                const int ci = k % CIN, kw = (k / CIN) % KW, kh = k / (CIN * KW);
                int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
                if (reflect) {
                    if (ih < 0) ih = -ih;
                    if (ih >= H) ih = 2 * H - 2 - ih;
                    if (iw < 0) iw = -iw;
                    if (iw >= W) iw = 2 * W - 2 - iw;
                }
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) v[e] = x[(((long)n * H + ih) * W + iw) * CIN + ci];
            }
        }
        if constexpr (__is_same(TO, S3Out)) {              // split-3 rows [2 * Kp]: chunk ch of hi | lo
            const size_t row = (size_t)(i / cpr);
            s3_store8(reinterpret_cast<unsigned short*>(out) + row * 2 * Kp, Kp, ch * 8, v);
        } else {
        TO* dst = out + (size_t)i * 8;
        if constexpr (sizeof(TO) == 2) {
            u32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 lo = (__bf16)v[2 * e], hi = (__bf16)v[2 * e + 1];
                r[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
            }
            *reinterpret_cast<u32x4*>(dst) = r;
        } else {
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        }
    }
}

// r4: the same patch rows from an LDS-staged strip.  The gather kernel above spends its time on address arithmetic and 8 scattered
// 4-byte loads per 16-byte store (464 us for the bench's 671 MB bf16 patch matrix = 1.7 TB/s, and both networks of a CPS pair wait for
// it).  Here a workgroup owns 64 consecutive output pixels of one output row: the 7 input rows x (2 * 63 + 7 = 133) input pixels x 3
// channels they read (padding applied while staging: reflect or zero) go to LDS with coalesced loads -- 3.4 x fewer global loads --
// and a patch column k = (kh, kw, ci) of pixel p is LDS word kh * ROW + 6 p + (k - 21 kh): for a fixed kh the 21 columns are
// CONSECUTIVE input floats.  320 threads = 16 pixels x 20 column chunks: a thread keeps its chunk (its eight LDS offsets are computed
// once) and walks TP / 16 pixels; ROW = 22 mod 32 puts the 20 chunks of a pixel in different banks.  Stride 2 / pad 3 / 160 columns
// (the stem).  Values identical to the gather kernel's.  Measured at the bench's shape (tools/bench_im2col.py; gather kernel -> strips
// of 64 -> of 128 pixels): bf16 rows 475 -> 236 -> 194 us (4.0 TB/s), split-3 rows 604 -> 311 -> 319, fp32 rows 583 -> 359 -> 381:
// bf16 takes 128-pixel strips (one load round trip per 8 KB stored instead of per 4 KB), the 32-byte-per-element forms 64.
#ifndef IM2COL_ABL
#define IM2COL_ABL 0              // debug builds only (results wrong): 1 no global stores, 2 no global loads
#endif
template <typename TO, int TP, int KP>
__global__ __launch_bounds__(2 * KP) void im2col_stem7_strip_kernel(const float* __restrict__ x, int N, int H, int W, int reflect, int Ho, int Wo,
                                                                 TO* __restrict__ out) {
    constexpr int NCOL = (2 * (TP - 1) + 7) * 3, CPR = KP / 8, NT = 16 * CPR;   // threads: 16 pixels x CPR column chunks (KP = 160: 320, 192: 384)   // NCOL floats of an input row feed the strip
    constexpr int ROW = (NCOL + 31 - 22) / 32 * 32 + 22;    // >= NCOL, = 22 mod 32 (406 for 64-pixel strips)
    constexpr int NH = (NCOL + NT - 1) / NT;
    __shared__ float strip[7 * ROW];
    const int strips = (Wo + TP - 1) / TP;
    const int sx = blockIdx.x % strips;
    const int oh = (blockIdx.x / strips) % Ho, n = blockIdx.x / (strips * Ho);
    const int ow0 = sx * TP, iw0 = 2 * ow0 - 3;
    const int t = threadIdx.x;
    // staging: a thread owns NH columns of the strip in all seven rows.  The loads are UNCONDITIONAL (clamped address, value selected
    // afterwards) and all issued before the first LDS write: a load under a divergent branch is waited for right there -- fourteen
    // serialised round trips per workgroup (measured: 190 us of a 300 us launch).  One strip per workgroup: a loop over strips with
    // the next strip's loads in flight was SLOWER (285 vs 234 us) -- gfx950 counts loads and stores on one in-order counter, so waiting
    // for the prefetch drains the stores issued after it.
    const float* xn = x + (size_t)n * H * W * 3;
    float r[NH][7];
#pragma unroll
    for (int half = 0; half < NH; ++half) {
        const int c = t + NT * half;
        const int px = c / 3, ci = c - 3 * px;
        int iw = iw0 + px;
        if (reflect) {
            if (iw < 0) iw = -iw;
            if (iw >= W) iw = 2 * W - 2 - iw;
        }
        const bool cok = c < NCOL && iw >= 0 && iw < W;     // (beyond a ragged last strip: unused)
        const int coff = cok ? iw * 3 + ci : 0;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            int ih = 2 * oh - 3 + kh;                       // workgroup-uniform
            if (reflect) {
                if (ih < 0) ih = -ih;
                if (ih >= H) ih = 2 * H - 2 - ih;
            }
            const bool ok = cok && ih >= 0 && ih < H;
#if IM2COL_ABL == 2
            const float v = (float)coff;
#else
            const float v = xn[(ok ? ih : 0) * W * 3 + coff];
#endif
            r[half][kh] = ok ? v : 0.0f;
        }
    }
#pragma unroll
    for (int half = 0; half < NH; ++half)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
            if (t + NT * half < NCOL) strip[kh * ROW + t + NT * half] = r[half][kh];
    __syncthreads();
    const int ch = t % CPR, p0 = t / CPR;                   // this thread's column chunk; pixels p0, p0 + 16, ...
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = ch * 8 + e;
        const int kh = k / 21;
        off[e] = k < 147 ? kh * ROW + (k - 21 * kh) : -1;
    }
    const int npx = Wo - ow0 < TP ? Wo - ow0 : TP;
    const size_t row0 = ((size_t)n * Ho + oh) * Wo + ow0;
#pragma unroll
    for (int it = 0; it < TP / 16; ++it) {
        const int p = p0 + 16 * it;
        if (p < npx) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = off[e] >= 0 ? strip[off[e] + 6 * p] : 0.0f;
#if IM2COL_ABL == 1
            if (v[0] != 12345.678f) continue;
#endif
            if constexpr (__is_same(TO, S3Out)) {          // split-3 rows [2 * KP]: chunk ch of hi | lo
                s3_store8(reinterpret_cast<unsigned short*>(out) + (row0 + p) * 2 * KP, KP, ch * 8, v);
            } else {
                TO* dst = out + (row0 + p) * KP + ch * 8;
                if constexpr (sizeof(TO) == 2) {
                    u32x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 lo = (__bf16)v[2 * e], hi = (__bf16)v[2 * e + 1];
                        q[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                    }
                    *reinterpret_cast<u32x4*>(dst) = q;
                } else {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "split-3" activations (the fp32-precision eval forward on the bf16 kernels): a logical fp32 tensor [rows][C] is stored as
// [rows][2C] bf16 = [hi | lo], hi = bf16(v), lo = bf16(v - hi).  A bf16 convolution over the logical channels [hi | lo | hi]
// (the K loop reads hi twice) with weights [w_hi | w_hi | w_lo] computes x_hi w_hi + x_lo w_hi + x_hi w_lo -- the three
// products of the precise mode.
// Elementwise ops on such tensors work on v = hi + lo (exact in fp32) and re-split.  8 logical channels per thread.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void s3_load8(const unsigned short* row, int C, int c, float (&v)[8]) {
    const u32x4 h = *reinterpret_cast<const u32x4*>(row + c), l = *reinterpret_cast<const u32x4*>(row + C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __builtin_bit_cast(float, h[e] << 16) + __builtin_bit_cast(float, l[e] << 16);
        v[2 * e + 1] = __builtin_bit_cast(float, h[e] & 0xFFFF0000u) + __builtin_bit_cast(float, l[e] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ unsigned s3_pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ void s3_store8(unsigned short* row, int C, int c, const float (&v)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = s3_pack2(v[2 * e], v[2 * e + 1]);
        const float h0 = __builtin_bit_cast(float, h[e] << 16), h1 = __builtin_bit_cast(float, h[e] & 0xFFFF0000u);
        l[e] = s3_pack2(v[2 * e] - h0, v[2 * e + 1] - h1);
    }
    *reinterpret_cast<u32x4*>(row + c) = h;
    *reinterpret_cast<u32x4*>(row + C + c) = l;
}

__global__ __launch_bounds__(256) void s3_split_kernel(const float* __restrict__ x, long rows, int C, unsigned short* __restrict__ y) {
    const int cv = C / 8;
    const long total = rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cv;
        const int c = (int)(i - r * cv) * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * C + c), b = *reinterpret_cast<const f32x4*>(x + r * C + c + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        s3_store8(y + r * 2 * C, C, c, v);
    }
}

__global__ __launch_bounds__(256) void s3_merge_kernel(const unsigned short* __restrict__ x, long rows, int C, float* __restrict__ y) {
    const int cv = C / 8;
    const long total = rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cv;
        const int c = (int)(i - r * cv) * 8;
        float v[8];
        s3_load8(x + r * 2 * C, C, c, v);
        *reinterpret_cast<f32x4*>(y + r * C + c) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(y + r * C + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// nn.MaxPool2d(3, 2, 1) forward (no window positions: the split-3 path never runs a backward)
__global__ __launch_bounds__(256) void s3_maxpool_kernel(const unsigned short* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                                         unsigned short* __restrict__ y) {
    const int cv = C / 8;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, ow, oh, n;
        split_nhwc(i, cv, Wo, Ho, c, ow, oh, n);
        c *= 8;
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -__builtin_inff();
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                float v[8];
                s3_load8(x + (((long)n * H + ih) * W + iw) * 2 * C, C, c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) best[e] = v[e];
            }
        }
        s3_store8(y + (((long)n * Ho + oh) * Wo + ow) * 2 * C, C, c, best);
    }
}

__global__ __launch_bounds__(256) void s3_bilinear_kernel(const unsigned short* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                                          int align, unsigned short* __restrict__ y) {
    const int cv = C / 8;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, ow, oh, n;
        split_nhwc(i, cv, Wo, Ho, c, ow, oh, n);
        c *= 8;
        int h0, h1, w0, w1;
        float lh, lw;
        bil_src(oh, H, Ho, align, h0, h1, lh);
        bil_src(ow, W, Wo, align, w0, w1, lw);
        const long b = (long)n * H;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        s3_load8(x + ((b + h0) * W + w0) * 2 * C, C, c, v00);
        s3_load8(x + ((b + h0) * W + w1) * 2 * C, C, c, v01);
        s3_load8(x + ((b + h1) * W + w0) * 2 * C, C, c, v10);
        s3_load8(x + ((b + h1) * W + w1) * 2 * C, C, c, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = bil_mix(v00[e], v01[e], v10[e], v11[e], lh, lw);
        s3_store8(y + (((long)n * Ho + oh) * Wo + ow) * 2 * C, C, c, o);
    }
}

// exact 2x / align_corners = false / power-of-two extents: see bilinear_up2_fwd_kernel (same values, expressions and order)
__global__ __launch_bounds__(256) void s3_bilinear_up2_kernel(const unsigned short* __restrict__ x, int N, int H, int W, int C, int lcv, int lwo,
                                                              int lho, unsigned short* __restrict__ y) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = ((long)N << (lcv + lwo + lho));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = ((int)i & ((1 << lcv) - 1)) * 8;
        const int ow = (int)(i >> lcv) & (Wo - 1), oh = (int)(i >> (lcv + lwo)) & (Ho - 1), n = (int)(i >> (lcv + lwo + lho));
        int h0, h1, w0, w1;
        float lh, lw;
        up2_src(oh, H, h0, h1, lh);
        up2_src(ow, W, w0, w1, lw);
        const long b = (long)n * H;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        s3_load8(x + ((b + h0) * W + w0) * 2 * C, C, c, v00);
        s3_load8(x + ((b + h0) * W + w1) * 2 * C, C, c, v01);
        s3_load8(x + ((b + h1) * W + w0) * 2 * C, C, c, v10);
        s3_load8(x + ((b + h1) * W + w1) * 2 * C, C, c, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = bil_mix(v00[e], v01[e], v10[e], v11[e], lh, lw);
        s3_store8(y + (((long)n * Ho + oh) * Wo + ow) * 2 * C, C, c, o);
    }
}

// gradient of reflect padding (pad 1): gp [N, H+2, W+2, C] -> gx [N, H, W, C]
template <typename T, int VC>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const T* __restrict__ gp, int N, int H, int W, int C, T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const long total = (long)N * H * W * cv;
    const int Hp = H + 2, Wp = W + 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c, w, h, n;
        split_nhwc(i, cv, W, H, c, w, h, n);
        c *= VC;
        // padded rows mapping to h: h+1 always; 0 if h == 1; Hp-1 if h == H-2
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + 1;
        if (h == 1) hs[nh++] = 0;
        if (h == H - 2) hs[nh++] = Hp - 1;
        ws[nw++] = w + 1;
        if (w == 1) ws[nw++] = 0;
        if (w == W - 2) ws[nw++] = Wp - 1;
        float acc[V];
#pragma unroll
        for (int e = 0; e < VC; ++e) acc[e] = 0.0f;
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                float v[V];
                ldc<T, VC>(gp, (((long)n * Hp + hs[a]) * Wp + ws[b]) * C + c, v);
#pragma unroll
                for (int e = 0; e < VC; ++e) acc[e] += v[e];
            }
        stc<T, VC>(gx, (((long)n * H + h) * W + w) * C + c, acc);
    }
}

// gradient of reflect padding (pad 1) when only the border RING of the padded gradient was computed (conv ring mode): gx [N, H, W, C]
// already holds the padded gradient's interior (= the zero-padded data gradient); the pixels of rows 1 / H-2 and columns 1 / W-2
// gather the ring positions that reflect onto them.  ring [N][2 (W + 2) + 2 H][C]: top row, bottom row, left column, right column.
template <typename T, int VC>
__global__ __launch_bounds__(256) void reflect_ring_fold_kernel(const T* __restrict__ ring, int N, int H, int W, int C, T* __restrict__ gx) {
    constexpr int V = VecN<T>::N;
    const int cv = C / VC;
    const int Hp = H + 2, Wp = W + 2, rl = 2 * Wp + 2 * H, nb = 2 * W + 2 * (H - 2);      // border pixels of gx that receive something
    const long total = (long)N * nb * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * VC;
        long t = i / cv;
        const int b = (int)(t % nb);
        const int n = (int)(t / nb);
        int h, w;
        if (b < W) h = 1, w = b;
        else if (b < 2 * W) h = H - 2, w = b - W;
        else {                                              // columns 1 and W-2 of the rows other than 1 and H-2
            const int q = b - 2 * W, half = H - 2;
            int r = q < half ? q : q - half;                // index among the rows {0, 2, 3, ..., H-3, H-1}
            h = r == 0 ? 0 : (r == half - 1 ? H - 1 : r + 1);
            w = q < half ? 1 : W - 2;
        }
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + 1;
        if (h == 1) hs[nh++] = 0;
        if (h == H - 2) hs[nh++] = Hp - 1;
        ws[nw++] = w + 1;
        if (w == 1) ws[nw++] = 0;
        if (w == W - 2) ws[nw++] = Wp - 1;
        float acc[V];
        const long o = (((long)n * H + h) * W + w) * C + c;
        ldc<T, VC>(gx, o, acc);
        for (int a = 0; a < nh; ++a)
            for (int bb = 0; bb < nw; ++bb) {
                if (a == 0 && bb == 0) continue;           // the interior position itself: already in gx
                const int pa = hs[a], pb = ws[bb];
                const int q = pa == 0 ? pb : (pa == Hp - 1 ? Wp + pb : (pb == 0 ? 2 * Wp + pa - 1 : 2 * Wp + (Hp - 2) + pa - 1));
                float v[V];
                ldc<T, VC>(ring, ((long)n * rl + q) * C + c, v);
#pragma unroll
                for (int e = 0; e < VC; ++e) acc[e] += v[e];
            }
        stc<T, VC>(gx, o, acc);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, long n, TO* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (TO)(float)x[i];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static long g_grid_cap = 8192;                              // workgroups of the grid-stride elementwise kernels (option "nn_grid_cap")
static inline unsigned grid_for(long work, int per_block = 256, long cap = 0) {
    if (cap == 0) cap = g_grid_cap;
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// debug only (timing experiments, results garbage): bit 1 skip the forward statistics merge + finalize, 2 the backward finalize,
// 4 the weight-gradient slab sums -- "what do the ~950 tiny launches of a step cost on the wall clock?" (VQSEG_OPTS=debug_skip_small=7)
static int g_debug_skip_small = 0;
int nn_set_option(const char* key, int value) {
    if (key && !strcmp(key, "debug_skip_small")) {
        const int prev = g_debug_skip_small;
        g_debug_skip_small = value;
        return prev;
    }
    if (key && !strcmp(key, "bn_bwd_premask")) {
        extern int bn_bwd_premask_option(int);
        return bn_bwd_premask_option(value);
    }
    if (key && !strcmp(key, "nn_grid_cap") && value >= 256) {
        const int prev = (int)g_grid_cap;
        g_grid_cap = value;
        return prev;
    }
    if (key && !strcmp(key, "im2col_strip")) {
        extern int im2col_strip_option(int);
        return im2col_strip_option(value);
    }
    if (key && !strcmp(key, "bilinear_up2")) {
        extern int bilinear_up2_option(int);
        return bilinear_up2_option(value);
    }
    return -1;
}

hipError_t launch_bn_finalize(float* partial, long n_slots, int rows_per_slot, long M, int C, const float* gamma,
                              const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, long long* num_batches_tracked, int* sync,
                              hipStream_t st_) {
    long spg = 1;                                           // slots per group of level 1
    if (g_debug_skip_small & 1) return hipSuccess;
    if (n_slots > 128) {
        spg = (n_slots + 127) / 128;
        if (spg < 16) spg = 16;
        const long groups = (n_slots + spg - 1) / spg;
        if (sync && groups > 1) {                           // one launch: the last workgroup of every channel group finalizes
            hipLaunchKernelGGL(bn_stats_merge_finalize_kernel, dim3((C + 63) / 64, (unsigned)groups), dim3(256), 0, st_, partial, n_slots, spg,
                               rows_per_slot, M, C, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd,
                               num_batches_tracked, sync);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((C + 63) / 64, (unsigned)groups), dim3(256), 0, st_, partial, n_slots, spg,
                           rows_per_slot, M, C);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st_, partial, n_slots, spg, rows_per_slot, M, C, gamma,
                       beta, run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd, num_batches_tracked);
    return hipGetLastError();
}

hipError_t launch_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                                 float eps, float* scale, float* shift, float* save_mean, float* save_invstd, hipStream_t st_) {
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, st_, C, gamma, beta, run_mean, run_var, eps,
                       scale, shift, save_mean, save_invstd);
    return hipGetLastError();
}

template <typename T>
static hipError_t bn_apply_t(const void* y, const void* res, const float* scale, const float* shift, long M, int C, int relu,
                             void* out, hipStream_t st_, unsigned char* bits) {
    if (bits && (VecN<T>::N != 8 || C % 8 != 0)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(grid_for(M * C / (C % VecN<T>::N ? 1 : VecN<T>::N))), dim3(256), 0, st_, (const T*)y,
                       (const T*)res, scale, shift, M, C, relu, (T*)out, bits);
    return hipGetLastError();
}
hipError_t launch_bn_apply(int bf16, const void* y, const void* res, const float* scale, const float* shift, long M, int C,
                           int relu, void* out, hipStream_t st_, unsigned char* bits) {
    return bf16 ? bn_apply_t<__bf16>(y, res, scale, shift, M, C, relu, out, st_, bits)
                : bn_apply_t<float>(y, res, scale, shift, M, C, relu, out, st_, bits);
}

long bn_bwd_blocks(long M) { return (M + BNB_ROWS_MIN - 1) / BNB_ROWS_MIN; }       // upper bound (workspace sizing)
// rows per workgroup of the reduce: 256, more for tall tensors so that the final sum over the row blocks (done by ONE workgroup per
// channel group when the finalize is fused) stays short -- at most 512 row blocks
static int bn_bwd_rows_per_block(long M) {
    long r = BNB_ROWS_MIN;
    while ((M + r - 1) / r > 512) r *= 2;
    return (int)r;
}

static int g_bn_bwd_premask = 1;                            // residual layers: masked gradient written by the reduce pass (0: r3, both passes read g_out + out)
int bn_bwd_premask_option(int value) {
    const int prev = g_bn_bwd_premask;
    g_bn_bwd_premask = value ? 1 : 0;
    return prev;
}
template <typename T, int MASK>
static hipError_t bn_bwd_t(const void* g_out, const void* out, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* fsc, const float* fsh, long M, int C, int relu, int training,
                           int accumulate, float* partial, float* coef,
                           float* dgamma, float* dbeta, void* g_y, void* g_res, int* sync, hipStream_t st_) {
    const int rpb = sync ? bn_bwd_rows_per_block(M) : BNB_ROWS_MIN;
    const long nb = (M + rpb - 1) / rpb;
    const BnBwdFin fin{gamma, training, accumulate, dgamma, dbeta, coef, (g_debug_skip_small & 2) ? nullptr : sync};
    // residual layers (MASK == 1 with a g_res output): the reduce pass writes the masked gradient = g_res, the apply pass runs unmasked on it
    // bit-field form (MASK == 3): with a g_res output as above; without one (the residual branch's consumer masks g_out with the
    // bits itself, vqseg_conv2d_affine_bits_f) both passes read (g_out, bits) and no masked gradient is written at all
    const bool premask = (MASK == 1 || MASK == 3) && g_res != nullptr && (g_bn_bwd_premask || MASK == 3);
    if (MASK == 3 && (C % VecN<T>::N != 0 || VecN<T>::N != 8)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK>), dim3((unsigned)nb, (unsigned)((C + BNB_CG - 1) / BNB_CG)), dim3(256), 0, st_,
                       (const T*)g_out, (const T*)out, (const T*)y, mean, invstd, fsc, fsh, M, C, relu, partial, rpb, fin,
                       premask ? (T*)g_res : (T*)nullptr);
    if (!(g_debug_skip_small & 2) && !sync)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st_, partial, nb, M, C, gamma, invstd, training, accumulate, dgamma,
                       dbeta, coef);
    if (premask)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 0>), dim3(grid_for(M * C / (C % VecN<T>::N ? 1 : VecN<T>::N))), dim3(256), 0, st_,
                           (const T*)g_res, (const T*)nullptr, (const T*)y, mean, invstd, coef, fsc, fsh, M, C, 0, (T*)g_y, (T*)nullptr);
    else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK>), dim3(grid_for(M * C / (C % VecN<T>::N ? 1 : VecN<T>::N))), dim3(256), 0, st_,
                       (const T*)g_out, (const T*)out, (const T*)y, mean, invstd, coef, fsc, fsh, M, C, relu, (T*)g_y, (T*)g_res);
    return hipGetLastError();
}
hipError_t launch_bn_backward(int bf16, const void* g_out, const void* out, const void* y, const float* mean,
                              const float* invstd, const float* gamma, const float* fsc, const float* fsh, long M, int C, int relu,
                              int training, int accumulate, float* partial,
                              float* coef, float* dgamma, float* dbeta, void* g_y, void* g_res, int* sync, hipStream_t st_,
                              const unsigned char* bits) {
    const int mask = !relu ? 0 : (bits ? 3 : out ? 1 : 2);
    if (mask == 3) out = bits;
#define BN_BWD_CASE(T_, MASK_)                                                                                                   \
    if (mask == MASK_)                                                                                                            \
        return bn_bwd_t<T_, MASK_>(g_out, out, y, mean, invstd, gamma, fsc, fsh, M, C, relu, training, accumulate, partial, coef, dgamma, dbeta, \
                                   g_y, g_res, sync, st_);
    if (bf16) {
        BN_BWD_CASE(__bf16, 0) BN_BWD_CASE(__bf16, 1) BN_BWD_CASE(__bf16, 2) BN_BWD_CASE(__bf16, 3)
    } else {
        BN_BWD_CASE(float, 0) BN_BWD_CASE(float, 1) BN_BWD_CASE(float, 2)
    }
#undef BN_BWD_CASE
    return hipErrorInvalidValue;
}

#define DISPATCH_T(bf16, CALL_F, CALL_B) \
    do {                                 \
        if (bf16) { CALL_B; } else { CALL_F; } \
    } while (0)

template <typename T, int VC>
static void maxpool_t(int backward, const void* x, const void* g, int N, int H, int W, int C, int Ho, int Wo, void* out,
                      unsigned char* idx, hipStream_t st_) {
    if (!backward)
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, VC>), dim3(grid_for((long)N * Ho * Wo * (C / VC))), dim3(256), 0, st_, (const T*)x, N, H,
                           W, C, Ho, Wo, (T*)out, idx);
    else if (idx)
        hipLaunchKernelGGL((maxpool_bwd_idx_kernel<T, VC>), dim3(grid_for((long)N * H * W * (C / VC))), dim3(256), 0, st_, idx,
                           (const T*)g, N, H, W, C, Ho, Wo, (T*)out, (const T*)(backward == 2 ? x : nullptr));
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<T, VC>), dim3(grid_for((long)N * H * W * (C / VC))), dim3(256), 0, st_, (const T*)x,
                           (const T*)g, N, H, W, C, Ho, Wo, (T*)out);
}

hipError_t launch_maxpool(int bf16, int backward, const void* x, const void* g, int N, int H, int W, int C, void* out,
                          unsigned char* idx, hipStream_t st_) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (bf16) {
        if (C % 8 == 0) maxpool_t<__bf16, 8>(backward, x, g, N, H, W, C, Ho, Wo, out, idx, st_);
        else maxpool_t<__bf16, 1>(backward, x, g, N, H, W, C, Ho, Wo, out, idx, st_);
    } else {
        if (C % 4 == 0) maxpool_t<float, 4>(backward, x, g, N, H, W, C, Ho, Wo, out, idx, st_);
        else maxpool_t<float, 1>(backward, x, g, N, H, W, C, Ho, Wo, out, idx, st_);
    }
    return hipGetLastError();
}

static int g_bilinear_up2 = 1;                              // exact-2x fast kernels (bit-identical to the generic ones)
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int bilinear_up2_option(int value) {                        // 2: the fast kernels with one input row per thread in the backward (r3)
    const int prev = g_bilinear_up2;
    g_bilinear_up2 = value > 2 ? 1 : value;
    return prev;
}

template <typename T, int VC>
static void bilinear_t(int backward, const void* src, int N, int H, int W, int C, int Ho, int Wo, int align, void* dst,
                       hipStream_t st_) {
    const int cv = C / VC;
    if (g_bilinear_up2 && VC > 1 && !align && Ho == 2 * H && Wo == 2 * W && is_pow2(cv) && is_pow2(H) && is_pow2(W) &&
        (long)N * Ho * Wo * cv < (1L << 40)) {
        if (!backward)
            hipLaunchKernelGGL((bilinear_up2_fwd_kernel<T, VC>), dim3(grid_for((long)N * Ho * Wo * cv)), dim3(256), 0, st_, (const T*)src, N, H,
                               W, C, ilog2(cv), ilog2(Wo), ilog2(Ho), (T*)dst);
        else if (g_bilinear_up2 == 1 && H >= 4)
            hipLaunchKernelGGL((bilinear_up2_bwd_rows_kernel<T, VC, 4>), dim3(grid_for((long)N * (H / 4) * W * cv)), dim3(256), 0, st_,
                               (const T*)src, N, H, W, C, ilog2(cv), ilog2(W), ilog2(H / 4), (T*)dst);
        else
            hipLaunchKernelGGL((bilinear_up2_bwd_kernel<T, VC>), dim3(grid_for((long)N * H * W * cv)), dim3(256), 0, st_, (const T*)src, N, H, W,
                               C, ilog2(cv), ilog2(W), ilog2(H), (T*)dst);
        return;
    }
    if (!backward)
        hipLaunchKernelGGL((bilinear_fwd_kernel<T, VC>), dim3(grid_for((long)N * Ho * Wo * (C / VC))), dim3(256), 0, st_, (const T*)src, N,
                           H, W, C, Ho, Wo, align, (T*)dst);
    else
        hipLaunchKernelGGL((bilinear_bwd_kernel<T, VC>), dim3(grid_for((long)N * H * W * (C / VC))), dim3(256), 0, st_, (const T*)src, N, H,
                           W, C, Ho, Wo, align, (T*)dst);
}

hipError_t launch_bilinear(int bf16, int backward, const void* src, int N, int H, int W, int C, int Ho, int Wo, int align,
                           void* dst, hipStream_t st_) {
    // forward: src [N,H,W,C] -> dst [N,Ho,Wo,C];  backward: src = grad [N,Ho,Wo,C] -> dst = grad_x [N,H,W,C]
    if (bf16) {
        if (C % 8 == 0) bilinear_t<__bf16, 8>(backward, src, N, H, W, C, Ho, Wo, align, dst, st_);
        else bilinear_t<__bf16, 1>(backward, src, N, H, W, C, Ho, Wo, align, dst, st_);
    } else {
        if (C % 4 == 0) bilinear_t<float, 4>(backward, src, N, H, W, C, Ho, Wo, align, dst, st_);
        else if (C == 3) bilinear_t<float, 3>(backward, src, N, H, W, C, Ho, Wo, align, dst, st_);
        else bilinear_t<float, 1>(backward, src, N, H, W, C, Ho, Wo, align, dst, st_);
    }
    return hipGetLastError();
}

hipError_t launch_head_fwd(int bf16, const void* x, const float* w, long M, int Cin, int Cout, float* y, hipStream_t st_) {
    // bf16: 0 f32 rows, 1 bf16 rows, 2 split-3 rows
    const unsigned gr = grid_for(M);
    if (bf16 == 2) hipLaunchKernelGGL(head_fwd_kernel<S3Row>, dim3(gr), dim3(256), 0, st_, (const S3Row*)x, w, M, Cin, Cout, y);
    else if (bf16) hipLaunchKernelGGL(head_fwd_kernel<__bf16>, dim3(gr), dim3(256), 0, st_, (const __bf16*)x, w, M, Cin, Cout, y);
    else hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(gr), dim3(256), 0, st_, (const float*)x, w, M, Cin, Cout, y);
    return hipGetLastError();
}

long head_bwd_blocks(long M) { return (M + HEAD_ROWS - 1) / HEAD_ROWS; }

hipError_t launch_head_bwd(int bf16, const void* x, const float* w, const float* g, long M, int Cin, int Cout, void* gx,
                           float* gw, float* partial, const void* gx_add, hipStream_t st_) {
    const unsigned gr = grid_for(M * (Cin / 8));
    DISPATCH_T(bf16, hipLaunchKernelGGL(head_bwd_data_kernel<float>, dim3(gr), dim3(256), 0, st_, g, w, M, Cin, Cout, (float*)gx, (const float*)gx_add),
               hipLaunchKernelGGL(head_bwd_data_kernel<__bf16>, dim3(gr), dim3(256), 0, st_, g, w, M, Cin, Cout, (__bf16*)gx, (const __bf16*)gx_add));
    const long nb = head_bwd_blocks(M);
    const size_t lds = (size_t)(256 / (Cin / 8)) * Cout * Cin * sizeof(float);
    DISPATCH_T(bf16, hipLaunchKernelGGL(head_bwd_weight_kernel<float>, dim3((unsigned)nb), dim3(256), lds, st_, (const float*)x, g, M, Cin, Cout, partial),
               hipLaunchKernelGGL(head_bwd_weight_kernel<__bf16>, dim3((unsigned)nb), dim3(256), lds, st_, (const __bf16*)x, g, M, Cin, Cout, partial));
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, st_, partial, nb, (long)Cin * Cout, gw, 0);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const float* partial, long n_blocks, long n, float* out, int accumulate, hipStream_t st_) {
    if (!(g_debug_skip_small & 4))
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st_, partial, n_blocks, n, out, accumulate);
    return hipGetLastError();
}

hipError_t launch_s3_split(const float* x, long rows, int C, void* y, hipStream_t st_) {
    hipLaunchKernelGGL(s3_split_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, st_, x, rows, C, (unsigned short*)y);
    return hipGetLastError();
}
hipError_t launch_s3_merge(const void* x, long rows, int C, float* y, hipStream_t st_) {
    hipLaunchKernelGGL(s3_merge_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, st_, (const unsigned short*)x, rows, C, y);
    return hipGetLastError();
}
hipError_t launch_s3_maxpool(const void* x, int N, int H, int W, int C, void* y, hipStream_t st_) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(s3_maxpool_kernel, dim3(grid_for((long)N * Ho * Wo * (C / 8))), dim3(256), 0, st_, (const unsigned short*)x, N, H, W,
                       C, Ho, Wo, (unsigned short*)y);
    return hipGetLastError();
}
hipError_t launch_s3_bilinear(const void* x, int N, int H, int W, int C, int Ho, int Wo, int align, void* y, hipStream_t st_) {
    const int cv = C / 8;
    if (g_bilinear_up2 && !align && Ho == 2 * H && Wo == 2 * W && is_pow2(cv) && is_pow2(H) && is_pow2(W) && (long)N * Ho * Wo * cv < (1L << 40)) {
        hipLaunchKernelGGL(s3_bilinear_up2_kernel, dim3(grid_for((long)N * Ho * Wo * cv)), dim3(256), 0, st_, (const unsigned short*)x, N, H, W, C,
                           ilog2(cv), ilog2(Wo), ilog2(Ho), (unsigned short*)y);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(s3_bilinear_kernel, dim3(grid_for((long)N * Ho * Wo * (C / 8))), dim3(256), 0, st_, (const unsigned short*)x, N, H,
                       W, C, Ho, Wo, align, (unsigned short*)y);
    return hipGetLastError();
}

static int g_im2col_strip = 1;                              // the stem's patch matrix from LDS-staged strips (0: the gather kernel, r3)
int im2col_strip_option(int value) {
    const int prev = g_im2col_strip;
    g_im2col_strip = value ? 1 : 0;
    return prev;
}
hipError_t launch_im2col_stem(int out_bf16, const float* x, int N, int H, int W, int Cin, int KH, int KW, int stride, int pad,
                              int reflect, int Ho, int Wo, int Kp, void* out, hipStream_t st_) {
    if (g_im2col_strip && KH == 7 && KW == 7 && Cin == 3 && stride == 2 && pad == 3 && (Kp == 160 || (Kp == 192 && out_bf16 == 2)) &&
        (long)H * W * 3 < (1L << 31) && (long)N * Ho * ((Wo + 63) / 64) < (1L << 31)) {
        const unsigned g64 = (unsigned)((long)N * Ho * ((Wo + 63) / 64)), g128 = (unsigned)((long)N * Ho * ((Wo + 127) / 128));
        if (out_bf16 == 2 && Kp == 192)                     // (the split-3 forward pads its patch rows to 64 columns)
            hipLaunchKernelGGL((im2col_stem7_strip_kernel<S3Out, 64, 192>), dim3(g64), dim3(384), 0, st_, x, N, H, W, reflect, Ho, Wo, (S3Out*)out);
        else if (out_bf16 == 2)
            hipLaunchKernelGGL((im2col_stem7_strip_kernel<S3Out, 64, 160>), dim3(g64), dim3(320), 0, st_, x, N, H, W, reflect, Ho, Wo, (S3Out*)out);
        else if (out_bf16)
            hipLaunchKernelGGL((im2col_stem7_strip_kernel<__bf16, 128, 160>), dim3(g128), dim3(320), 0, st_, x, N, H, W, reflect, Ho, Wo, (__bf16*)out);
        else
            hipLaunchKernelGGL((im2col_stem7_strip_kernel<float, 64, 160>), dim3(g64), dim3(320), 0, st_, x, N, H, W, reflect, Ho, Wo, (float*)out);
        return hipGetLastError();
    }
    if (KH == 7 && KW == 7 && Cin == 3 && Kp % 8 == 0 && (long)N * Ho * Wo * (Kp / 8) < (1L << 31)) {
        const unsigned g8 = grid_for((long)N * Ho * Wo * (Kp / 8));
        if (out_bf16 == 2)
            hipLaunchKernelGGL((im2col_stem7_kernel<S3Out>), dim3(g8), dim3(256), 0, st_, x, N, H, W, stride, pad, reflect, Ho, Wo, Kp,
                               (S3Out*)out);
        else if (out_bf16)
            hipLaunchKernelGGL((im2col_stem7_kernel<__bf16>), dim3(g8), dim3(256), 0, st_, x, N, H, W, stride, pad, reflect, Ho, Wo, Kp,
                               (__bf16*)out);
        else
            hipLaunchKernelGGL((im2col_stem7_kernel<float>), dim3(g8), dim3(256), 0, st_, x, N, H, W, stride, pad, reflect, Ho, Wo, Kp,
                               (float*)out);
        return hipGetLastError();
    }
    if (out_bf16 == 2) return hipErrorInvalidValue;          // split-3 patches: the 7x7x3 stem shape only
    const unsigned gr = grid_for((long)N * Ho * Wo * Kp);
    if (out_bf16)
        hipLaunchKernelGGL((im2col_stem_kernel<float, __bf16>), dim3(gr), dim3(256), 0, st_, x, N, H, W, Cin, KH, KW, stride, pad,
                           reflect, Ho, Wo, Kp, (__bf16*)out);
    else
        hipLaunchKernelGGL((im2col_stem_kernel<float, float>), dim3(gr), dim3(256), 0, st_, x, N, H, W, Cin, KH, KW, stride, pad,
                           reflect, Ho, Wo, Kp, (float*)out);
    return hipGetLastError();
}

template <typename T, int VC>
static void reflect_fold_t(const void* gp, int N, int H, int W, int C, void* gx, hipStream_t st_) {
    hipLaunchKernelGGL((reflect_fold_kernel<T, VC>), dim3(grid_for((long)N * H * W * (C / VC))), dim3(256), 0, st_, (const T*)gp, N, H,
                       W, C, (T*)gx);
}

hipError_t launch_reflect_fold(int bf16, const void* gp, int N, int H, int W, int C, void* gx, hipStream_t st_) {
    if (bf16) {
        if (C % 8 == 0) reflect_fold_t<__bf16, 8>(gp, N, H, W, C, gx, st_);
        else reflect_fold_t<__bf16, 1>(gp, N, H, W, C, gx, st_);
    } else {
        if (C % 4 == 0) reflect_fold_t<float, 4>(gp, N, H, W, C, gx, st_);
        else reflect_fold_t<float, 1>(gp, N, H, W, C, gx, st_);
    }
    return hipGetLastError();
}

hipError_t launch_reflect_ring_fold(int bf16, const void* ring, int N, int H, int W, int C, void* gx, hipStream_t st_) {
    const long work = (long)N * (2 * W + 2 * (H - 2));
    if (bf16) {
        if (C % 8 == 0) hipLaunchKernelGGL((reflect_ring_fold_kernel<__bf16, 8>), dim3(grid_for(work * (C / 8))), dim3(256), 0, st_, (const __bf16*)ring, N, H, W, C, (__bf16*)gx);
        else hipLaunchKernelGGL((reflect_ring_fold_kernel<__bf16, 1>), dim3(grid_for(work * C)), dim3(256), 0, st_, (const __bf16*)ring, N, H, W, C, (__bf16*)gx);
    } else {
        if (C % 4 == 0) hipLaunchKernelGGL((reflect_ring_fold_kernel<float, 4>), dim3(grid_for(work * (C / 4))), dim3(256), 0, st_, (const float*)ring, N, H, W, C, (float*)gx);
        else hipLaunchKernelGGL((reflect_ring_fold_kernel<float, 1>), dim3(grid_for(work * C)), dim3(256), 0, st_, (const float*)ring, N, H, W, C, (float*)gx);
    }
    return hipGetLastError();
}

hipError_t launch_cast(int to_bf16, const void* x, long n, void* y, hipStream_t st_) {
    const unsigned gr = grid_for(n);
    if (to_bf16) hipLaunchKernelGGL((cast_kernel<float, __bf16>), dim3(gr), dim3(256), 0, st_, (const float*)x, n, (__bf16*)y);
    else hipLaunchKernelGGL((cast_kernel<__bf16, float>), dim3(gr), dim3(256), 0, st_, (const __bf16*)x, n, (float*)y);
    return hipGetLastError();
}

}  // namespace vqseg
