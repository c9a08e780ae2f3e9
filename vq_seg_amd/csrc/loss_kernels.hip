// Reliable prototype losses (reference: models/modules/prototype.py:500-613 v1, :778-888 v2) as two fused passes
// over the decoder features instead of ~25 element-wise ATen kernels per call:
//   forward : per pixel row  r = x / max(|x|, 1e-12),  cos_c = r . p_c,  z_c = f(cos_c) (ArcFace-style margin, scale),
//             ll = log( exp(S) / (sum_c exp(z_c) + 1e-7) + 1e-7 ),  loss = -mean(ll * w)
//             v1: S = sum_c z_c * (onehot_c + 1e-6) in double (the reference's float64 one-hot + eps), w = entropy keep mask
//             v2: S = z_t with z_t = scale * cos_t * phi(cos_t) (the reference multiplies), w = confidence mask or 1
//   backward: the same row recomputed, closed-form d loss / d x (and, v2, d loss / d prototypes: block partials).
// One thread per pixel row (C <= 64 channels stay in registers); HBM-bound: one read of x (+ one write of its gradient).
// Block sums are folded in fixed order (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "loss_kernels.h"

namespace vqseg {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int MAXC = 64, MAXK = 4;

template <typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ x, long row, int C, float (&u)[MAXC]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4* p = reinterpret_cast<const u32x4*>(x + row * C);
#pragma unroll
        for (int j = 0; j < MAXC / 8; ++j)
            if (j * 8 < C) {
                const u32x4 v = p[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[j * 8 + 2 * e] = __builtin_bit_cast(float, v[e] << 16);
                    u[j * 8 + 2 * e + 1] = __builtin_bit_cast(float, v[e] & 0xFFFF0000u);
                }
            }
    } else {
        const f32x4* p = reinterpret_cast<const f32x4*>(x + row * C);
#pragma unroll
        for (int j = 0; j < MAXC / 4; ++j)
            if (j * 4 < C) {
                const f32x4 v = p[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[j * 4 + e] = v[e];
            }
    }
}

__device__ __forceinline__ float phi_of(float c, const ProtoArgs& a) {
    float inner = 1.0f - c * c;
    inner = inner < 0.0f ? 0.0f : (inner > 1.0f ? 1.0f : inner);
    const float ph = c * a.cos_m - sqrtf(inner) * a.sin_m;
    return a.easy_margin ? (c > 0.0f ? ph : c) : (c > a.th ? ph : c - a.mm);
}
__device__ __forceinline__ float dphi_of(float c, const ProtoArgs& a) {
    const float inner = 1.0f - c * c;
    const float dsine = (inner > 0.0f && inner <= 1.0f) ? -c / sqrtf(inner) : 0.0f;   // clamp passes gradient inside [0, 1]
    const float d = a.cos_m - a.sin_m * dsine;
    return a.easy_margin ? (c > 0.0f ? d : 1.0f) : (c > a.th ? d : 1.0f);
}

struct RowTerms {
    double ll, w;
    double dcos[MAXK];          // d ll / d cos_c
    float cosv[MAXK];
    float inv_n;
};

// everything of one row except the sums over channels (n2, dots come in)
__device__ __forceinline__ RowTerms row_terms(const ProtoArgs& a, long row, float n2, const float (&dot)[MAXK], bool want_grad) {
    RowTerms t;
    const float n = fmaxf(sqrtf(n2), 1e-12f);             // F.normalize: x / max(|x|, eps)
    t.inv_n = 1.0f / n;
    const int tgt = (int)a.labels[row];
    double z[MAXK], dz[MAXK], oh[MAXK];
#pragma unroll
    for (int c = 0; c < MAXK; ++c)
        if (c < a.K) {
            const float cs = dot[c] * t.inv_n;
            t.cosv[c] = cs;
            if (a.variant == 1) {
                oh[c] = (c == tgt ? 1.0 : 0.0) + 1e-6;      // float64 one-hot + eps (utils/seg_tools.py onehot_1d)
                if (a.use_margin) {
                    z[c] = oh[c] * (double)phi_of(cs, a) + (1.0 - oh[c]) * (double)cs;
                    dz[c] = oh[c] * (double)dphi_of(cs, a) + (1.0 - oh[c]);
                } else {
                    z[c] = (double)cs;
                    dz[c] = 1.0;
                }
            } else {
                oh[c] = c == tgt ? 1.0 : 0.0;
                if (c == tgt) {                              // torch.where(hit, cosine * phi, cosine)
                    const float ph = phi_of(cs, a);
                    z[c] = (double)(cs * ph);
                    dz[c] = (double)ph + (double)cs * (double)dphi_of(cs, a);
                } else {
                    z[c] = (double)cs;
                    dz[c] = 1.0;
                }
            }
            z[c] *= (double)a.scale;
            dz[c] *= (double)a.scale;
        }
    double S = 0.0, total = 0.0;
#pragma unroll
    for (int c = 0; c < MAXK; ++c)
        if (c < a.K) {
            S += z[c] * oh[c];
            total += exp(z[c]);
        }
    const double positive = exp(S);
    const double den = total + 1e-7;
    const double q = positive / den + 1e-7;
    t.ll = log(q);
    t.w = a.variant == 1 ? ((a.keep == nullptr || a.keep[row]) ? 1.0 : 0.0) : (a.conf ? (double)a.conf[row] : 1.0);
    if (want_grad) {
#pragma unroll
        for (int c = 0; c < MAXK; ++c)
            if (c < a.K) {
                const double dll_dz = (positive * oh[c] / den - positive * exp(z[c]) / (den * den)) / q;
                t.dcos[c] = dll_dz * dz[c];
            }
    }
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void proto_fwd_kernel(const ProtoArgs a, double* __restrict__ partial) {
    __shared__ float ps[MAXK * MAXC];
    __shared__ double red[256];
    for (int i = threadIdx.x; i < a.K * a.C; i += 256) ps[i] = a.proto[i];
    __syncthreads();
    const long row = (long)blockIdx.x * PROTO_ROWS_PER_BLOCK + threadIdx.x;
    double v = 0.0;
    if (row < a.M) {
        float u[MAXC];
        load_row<T>(reinterpret_cast<const T*>(a.x), row, a.C, u);
        float n2 = 0.0f, dot[MAXK] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (i < a.C) {
                n2 = __builtin_fmaf(u[i], u[i], n2);
#pragma unroll
                for (int c = 0; c < MAXK; ++c)
                    if (c < a.K) dot[c] = __builtin_fmaf(u[i], ps[c * a.C + i], dot[c]);
            }
        const RowTerms t = row_terms(a, row, n2, dot, false);
        v = t.ll * t.w;
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void proto_loss_final_kernel(const double* __restrict__ partial, long n_blocks, long M,
                                                               double* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = threadIdx.x; i < n_blocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = -red[0] / (double)M;
}

template <typename T>
__global__ __launch_bounds__(256) void proto_bwd_kernel(const ProtoArgs a, const float* __restrict__ g_loss, T* __restrict__ gx,
                                                        float* __restrict__ gproto_partial, int row_tiles) {
    __shared__ float ps[MAXK * MAXC];
    extern __shared__ float tile[];                         // prototype gradient only: rs[256][C + 1], gcs[256][MAXK], part[256]
    for (int i = threadIdx.x; i < a.K * a.C; i += 256) ps[i] = a.proto[i];
    __syncthreads();
    // d loss / d p_c[i] = sum_rows gcos_c(row) * r_i(row), r = x / |x|: a (K x rows) . (rows x C) product.  Each tile of 256
    // rows leaves r and gcos in LDS; thread (group g, pair (c, i)) adds the tile's rows g, g + G, ... in row order.
    const int KC = a.K * a.C, G = 256 / KC;                 // K * C <= 256
    const int pair = threadIdx.x % KC, grp = threadIdx.x / KC;
    const int pc = pair / a.C, pi = pair % a.C;
    float* rs = tile;
    float* gcs = tile + 256 * (a.C + 1);
    float gacc = 0.0f;
    for (int rt = 0; rt < row_tiles; ++rt) {
    const long row = ((long)blockIdx.x * row_tiles + rt) * PROTO_ROWS_PER_BLOCK + threadIdx.x;
    float u[MAXC];
    float gc[MAXK] = {0.f, 0.f, 0.f, 0.f};
    float inv_n = 0.0f;
    float cosv[MAXK] = {0.f, 0.f, 0.f, 0.f};
    const bool live = row < a.M;
    if (live) {
        load_row<T>(reinterpret_cast<const T*>(a.x), row, a.C, u);
        float n2 = 0.0f, dot[MAXK] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (i < a.C) {
                n2 = __builtin_fmaf(u[i], u[i], n2);
#pragma unroll
                for (int c = 0; c < MAXK; ++c)
                    if (c < a.K) dot[c] = __builtin_fmaf(u[i], ps[c * a.C + i], dot[c]);
            }
        const RowTerms t = row_terms(a, row, n2, dot, true);
        const double k = -(double)g_loss[0] * t.w / (double)a.M;          // d loss / d ll of this row
        inv_n = t.inv_n;
#pragma unroll
        for (int c = 0; c < MAXK; ++c)
            if (c < a.K) {
                gc[c] = (float)(k * t.dcos[c]);
                cosv[c] = t.cosv[c];
            }
        // d cos_c / d x = (p_c - cos_c r) / n,  r = x / n
        float gs = 0.0f;                                                  // sum_c gcos_c cos_c
#pragma unroll
        for (int c = 0; c < MAXK; ++c)
            if (c < a.K) gs = __builtin_fmaf(gc[c], cosv[c], gs);
        T* grow = gx + row * a.C;
#pragma unroll
        for (int j = 0; j < MAXC / 8; ++j)
            if (j * 8 < a.C) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = j * 8 + e;
                    float acc = -gs * (u[i] * inv_n);
#pragma unroll
                    for (int c = 0; c < MAXK; ++c)
                        if (c < a.K) acc = __builtin_fmaf(gc[c], ps[c * a.C + i], acc);
                    o[e] = acc * inv_n;
                }
                if constexpr (sizeof(T) == 2) {
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 lo = (__bf16)o[2 * e], hi = (__bf16)o[2 * e + 1];
                        v[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                    }
                    *reinterpret_cast<u32x4*>(grow + j * 8) = v;
                } else {
                    *reinterpret_cast<f32x4*>(grow + j * 8) = f32x4{o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(grow + j * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
                }
            }
    }
    if (gproto_partial) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (i < a.C) rs[threadIdx.x * (a.C + 1) + i] = live ? u[i] * inv_n : 0.0f;
#pragma unroll
        for (int c = 0; c < MAXK; ++c) gcs[threadIdx.x * MAXK + c] = gc[c];
        __syncthreads();
        if (grp < G)
            for (int r = grp; r < PROTO_ROWS_PER_BLOCK; r += G) gacc = __builtin_fmaf(gcs[r * MAXK + pc], rs[r * (a.C + 1) + pi], gacc);
        __syncthreads();
    }
    }   // row tiles
    if (gproto_partial) {
        float* part = gcs + 256 * MAXK;
        part[threadIdx.x] = grp < G ? gacc : 0.0f;
        __syncthreads();
        if ((int)threadIdx.x < KC) {
            float s = 0.0f;
            for (int g = 0; g < G; ++g) s += part[g * KC + threadIdx.x];
            gproto_partial[(long)blockIdx.x * KC + threadIdx.x] = s;
        }
    }
}

// one workgroup per prototype entry: 256 strided double sums over the row-tile partials, folded in a fixed tree
__global__ __launch_bounds__(256) void proto_gproto_final_kernel(const float* __restrict__ partial, long n_blocks, int n,
                                                                 float* __restrict__ out) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (long b = threadIdx.x; b < n_blocks; b += 256) s += (double)partial[b * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = (float)red[0];
}

}  // namespace

long proto_blocks(long M) { return (M + PROTO_ROWS_PER_BLOCK - 1) / PROTO_ROWS_PER_BLOCK; }

hipError_t launch_proto_forward(const ProtoArgs& a, double* partial, double* loss, hipStream_t st) {
    const long nb = proto_blocks(a.M);
    if (a.bf16) hipLaunchKernelGGL(proto_fwd_kernel<__bf16>, dim3((unsigned)nb), dim3(256), 0, st, a, partial);
    else hipLaunchKernelGGL(proto_fwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, a, partial);
    hipLaunchKernelGGL(proto_loss_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, a.M, loss);
    return hipGetLastError();
}

hipError_t launch_proto_backward(const ProtoArgs& a, const float* g_loss, void* gx, float* gproto_partial, float* gproto,
                                 hipStream_t st) {
    // with a prototype gradient a workgroup walks several 256-row tiles (<= 2048 workgroups), so that the per-workgroup
    // fold of the (K x C) partial is paid once per few thousand rows
    const bool gp = gproto_partial && gproto;
    int row_tiles = 1;
    if (gp) {
        const long want = (proto_blocks(a.M) + 2047) / 2048;
        row_tiles = (int)(want < 1 ? 1 : want);
    }
    const long nb = (proto_blocks(a.M) + row_tiles - 1) / row_tiles;
    const size_t lds = gp ? (size_t)(256 * (a.C + 1) + 256 * MAXK + 256) * sizeof(float) : 0;
    float* gpp = gp ? gproto_partial : nullptr;
    if (a.bf16) hipLaunchKernelGGL(proto_bwd_kernel<__bf16>, dim3((unsigned)nb), dim3(256), lds, st, a, g_loss, (__bf16*)gx, gpp, row_tiles);
    else hipLaunchKernelGGL(proto_bwd_kernel<float>, dim3((unsigned)nb), dim3(256), lds, st, a, g_loss, (float*)gx, gpp, row_tiles);
    if (gp) hipLaunchKernelGGL(proto_gproto_final_kernel, dim3(a.K * a.C), dim3(256), 0, st, gproto_partial, nb, a.K * a.C, gproto);
    return hipGetLastError();
}

// =====================================================================================================
// Soft Dice sums.  Reference quirks kept (loss/dice_loss.py:12-18): ignored pixels get ZERO logits (softmax 1/C)
// and become class-0 targets.  One thread walks pixels with stride 256; block sums in double, fixed-order folds.
// =====================================================================================================
namespace {
constexpr int DMAXC = 4;

// nll: -log softmax(logits)[target] of a kept pixel (nn.CrossEntropyLoss(ignore_index) skips the others), 0 otherwise
__device__ __forceinline__ void dice_pixel(const DiceArgs& a, int b, long px, float (&p)[DMAXC], int& tgt, bool& keep, float& nll) {
    const long long t = a.target[(long)b * a.HW + px];
    keep = t != a.ignore_index;
    tgt = keep ? (int)t : 0;
    float z[DMAXC], mx = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            z[c] = keep ? a.logits[(long)b * a.sb + c * a.sc + px * a.sp] : 0.0f;
            mx = fmaxf(mx, z[c]);
        }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            p[c] = expf(z[c] - mx);
            sum += p[c];
        }
    const float inv = 1.0f / sum;
    float zt = 0.0f;
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            p[c] *= inv;
            if (c == tgt) zt = z[c];
        }
    nll = keep ? logf(sum) + mx - zt : 0.0f;
}

__global__ __launch_bounds__(256) void dice_fwd_kernel(const DiceArgs a, double* __restrict__ partial) {
    __shared__ double red[256];
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * DICE_PX_PER_BLOCK;
    double inter[DMAXC] = {0, 0, 0, 0}, sets[DMAXC] = {0, 0, 0, 0}, ce[2] = {0, 0};       // ce: sum of nll, kept pixels
    for (long px = p0 + threadIdx.x; px < p0 + DICE_PX_PER_BLOCK && px < a.HW; px += 256) {
        float p[DMAXC], nll;
        int tgt;
        bool keep;
        dice_pixel(a, b, px, p, tgt, keep, nll);
        ce[0] += (double)nll;
        ce[1] += keep ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < DMAXC; ++c)
            if (c < a.C) {
                const float oh = c == tgt ? 1.0f : 0.0f;
                inter[c] += (double)(p[c] * oh);
                sets[c] += (double)(p[c] + oh);
            }
    }
    const int Q = 2 * a.C + 2;
    double* out = partial + ((long)b * gridDim.x + blockIdx.x) * Q;
    double vals[2 * DMAXC + 2];                                 // [inter | sets | ce] at compile-time positions (registers, no scratch)
#pragma unroll
    for (int c = 0; c < DMAXC; ++c) {
        vals[c] = inter[c];
        vals[DMAXC + c] = sets[c];
    }
    vals[2 * DMAXC] = ce[0];
    vals[2 * DMAXC + 1] = ce[1];
#pragma unroll
    for (int qq = 0; qq < 2 * DMAXC + 2; ++qq) {
        // slot qq of the padded layout is output q of the dense one (classes >= C are skipped)
        const int cls = qq < DMAXC ? qq : qq < 2 * DMAXC ? qq - DMAXC : 0;
        if (qq < 2 * DMAXC && cls >= a.C) continue;
        const int q = qq < DMAXC ? qq : qq < 2 * DMAXC ? a.C + cls : 2 * a.C + (qq - 2 * DMAXC);
        red[threadIdx.x] = vals[qq];
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[q] = red[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void dice_final_kernel(const double* __restrict__ partial, long n_blocks, int C,
                                                        float* __restrict__ inter, float* __restrict__ sets, float* __restrict__ ce) {
    const int b = blockIdx.x, q = threadIdx.x, Q = 2 * C + 2;
    if (q >= Q) return;
    double s = 0.0;
    for (long i = 0; i < n_blocks; ++i) s += partial[((long)b * n_blocks + i) * Q + q];
    if (q < C) inter[b * C + q] = (float)s;
    else if (q < 2 * C) sets[b * C + q - C] = (float)s;
    else if (ce) ce[b * 2 + q - 2 * C] = (float)s;            // [b][0] = sum of nll over kept pixels, [b][1] = their number
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const DiceArgs a, const float* __restrict__ g_inter,
                                                       const float* __restrict__ g_sets, const float* __restrict__ g_ce,
                                                       float* __restrict__ g_logits) {
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * DICE_PX_PER_BLOCK;
    float gi[DMAXC], gs[DMAXC];
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            gi[c] = g_inter[b * a.C + c];
            gs[c] = g_sets[b * a.C + c];
        }
    const float gce = g_ce ? g_ce[b * 2] : 0.0f;               // d loss / d (sum of nll of image b)
    for (long px = p0 + threadIdx.x; px < p0 + DICE_PX_PER_BLOCK && px < a.HW; px += 256) {
        float p[DMAXC], nll;
        int tgt;
        bool keep;
        dice_pixel(a, b, px, p, tgt, keep, nll);
        float dp[DMAXC], dotp = 0.0f;
#pragma unroll
        for (int c = 0; c < DMAXC; ++c)
            if (c < a.C) {
                dp[c] = gs[c] + (c == tgt ? gi[c] : 0.0f);
                dotp = __builtin_fmaf(dp[c], p[c], dotp);
            }
#pragma unroll
        for (int c = 0; c < DMAXC; ++c)
            if (c < a.C)                                       // Dice: logits * keep;  CE: softmax - onehot on kept pixels
                g_logits[(long)b * a.sb + c * a.sc + px * a.sp] =
                    keep ? __builtin_fmaf(gce, p[c] - (c == tgt ? 1.0f : 0.0f), p[c] * (dp[c] - dotp)) : 0.0f;
    }
}
}  // namespace

namespace {
// Pseudo-label statistics of the CPS trainers (deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39 softmax -> argmax,
// entropy; train_vqreptunet1x1v2.py:43-46 softmax -> max) and of VQRePTUnet1x1.forward's entropy map (net.py:1199-1201)
__global__ __launch_bounds__(256) void softmax_stats_kernel(const DiceArgs a, long long* __restrict__ label,
                                                            float* __restrict__ entropy, float* __restrict__ top) {
    const int b = blockIdx.y;
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    if (px >= a.HW) return;
    float z[DMAXC], mx = -__builtin_inff();
    int arg = 0;
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            z[c] = a.logits[(long)b * a.sb + c * a.sc + px * a.sp];
            if (z[c] > mx) {                                    // first maximum, as torch.argmax
                mx = z[c];
                arg = c;
            }
        }
    float sum = 0.0f, p[DMAXC];
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            p[c] = expf(z[c] - mx);
            sum += p[c];
        }
    const float inv = 1.0f / sum;
    float ent = 0.0f;
#pragma unroll
    for (int c = 0; c < DMAXC; ++c)
        if (c < a.C) {
            const float q = p[c] * inv;
            ent -= q * logf(q + 1e-10f);
        }
    const long o = (long)b * a.HW + px;
    if (label) label[o] = arg;
    if (entropy) entropy[o] = ent;
    if (top) top[o] = inv;                                      // exp(0) / sum: the probability of the arg-max class
}
}  // namespace

namespace {
// Confusion counts of measurement.py:12-20 (Measurement._make_confusion_matrix: bincount of C * target + argmax(pred)) per
// image, rows = ground truth.  Integer counts (LDS histogram per workgroup, then integer atomics): order-independent.
__global__ __launch_bounds__(256) void confusion_kernel(const DiceArgs a, unsigned long long* __restrict__ out) {
    __shared__ unsigned cnt[DMAXC * DMAXC];
    if (threadIdx.x < DMAXC * DMAXC) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * DICE_PX_PER_BLOCK;
    for (long px = p0 + threadIdx.x; px < p0 + DICE_PX_PER_BLOCK && px < a.HW; px += 256) {
        const long long t = a.target[(long)b * a.HW + px];
        float mx = -__builtin_inff();
        int arg = 0;
#pragma unroll
        for (int c = 0; c < DMAXC; ++c)
            if (c < a.C) {
                const float z = a.logits[(long)b * a.sb + c * a.sc + px * a.sp];
                if (z > mx) {                                       // first maximum, as torch.argmax
                    mx = z;
                    arg = c;
                }
            }
        if (t >= 0 && t < a.C) atomicAdd(&cnt[(int)t * a.C + arg], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < a.C * a.C && cnt[threadIdx.x])
        atomicAdd(&out[(long)b * a.C * a.C + threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}
}  // namespace

hipError_t launch_confusion(const DiceArgs& a, long long* out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)a.B * a.C * a.C * sizeof(long long), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)dice_blocks(a.HW), (unsigned)a.B), dim3(256), 0, st, a,
                       reinterpret_cast<unsigned long long*>(out));
    return hipGetLastError();
}

hipError_t launch_softmax_stats(const DiceArgs& a, long long* label, float* entropy, float* top, hipStream_t st) {
    hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)((a.HW + 255) / 256), (unsigned)a.B), dim3(256), 0, st, a, label, entropy, top);
    return hipGetLastError();
}

namespace {
// ---- exact order statistics by radix select (the percentile of make_regularized_pseudo_label,
// deprecated/train_with_test_pt_pseudo_entropy_reg.py:35: np.percentile needs the two order statistics around the virtual
// index).  Three counting passes over the data (12 + 12 + 8 key bits); integer counts only, so the result does not depend
// on the order the workgroups run in.  Two target ranks (k, k + 1) are narrowed together.
constexpr int KTH_BINS = 4096;
struct KthState {
    unsigned prefix[2];
    unsigned long long rank[2];
};
__device__ __forceinline__ unsigned order_key(float v) {        // unsigned order == float order (-0 < +0; NaNs at the ends)
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ void kth_count(unsigned* hist, bool hit, unsigned digit) {
    // entropies cluster in a few exponent bins: when the whole wave hits one bin, one lane adds the population count
    const unsigned long long m = __ballot(hit);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    const unsigned first = __shfl(digit, leader);
    if (__ballot(hit && digit == first) == m) {
        if ((int)__lane_id() == leader) atomicAdd(&hist[first], (unsigned)__popcll(m));
    } else if (hit) {
        atomicAdd(&hist[digit], 1u);
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void kth_hist_kernel(const float* __restrict__ x, long n, const KthState* __restrict__ state,
                                                       unsigned* __restrict__ hist) {
    constexpr int SHIFT = PASS == 0 ? 20 : PASS == 1 ? 8 : 0;          // low end of this pass's digit
    constexpr int BITS = PASS == 2 ? 8 : 12;
    constexpr unsigned MASK = (1u << BITS) - 1;
    __shared__ unsigned lh[2][KTH_BINS];
    for (int i = threadIdx.x; i < 2 * KTH_BINS; i += 256) (&lh[0][0])[i] = 0;
    unsigned p0 = 0, p1 = 0;
    if (PASS > 0) { p0 = state->prefix[0]; p1 = state->prefix[1]; }
    const bool two = PASS > 0 && p0 != p1;
    __syncthreads();
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i <= n4; i += stride) {
        float v[4] = {0, 0, 0, 0};
        int cnt = 0;
        if (i < n4) {
            const float4 q = reinterpret_cast<const float4*>(x)[i];
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            cnt = 4;
        } else if (i == n4) {                                                            // the ragged tail (n % 4 values)
            cnt = (int)(n - n4 * 4);
            for (int j = 0; j < cnt; ++j) v[j] = x[n4 * 4 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned key = order_key(v[j]);
            const unsigned digit = (key >> SHIFT) & MASK;
            const unsigned hi = PASS == 0 ? 0u : key >> (SHIFT + BITS);
            kth_count(lh[0], j < cnt && hi == p0, digit);
            if (two) kth_count(lh[1], j < cnt && hi == p1, digit);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * KTH_BINS; i += 256) {
        const unsigned c = (&lh[0][0])[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// one workgroup: the bin holding each target rank becomes the next digit of its prefix; clears the histogram for the next pass
template <int PASS>
__global__ __launch_bounds__(256) void kth_scan_kernel(KthState* __restrict__ state, unsigned* __restrict__ hist, float* __restrict__ out) {
    constexpr int BITS = PASS == 2 ? 8 : 12;
    constexpr int NB = 1 << BITS, PER = KTH_BINS / 256;
    __shared__ unsigned long long part[256];
    __shared__ unsigned digit_s;
    __shared__ unsigned long long before_s;
    const int t = threadIdx.x;
    const bool same = PASS == 0 || state->prefix[0] == state->prefix[1];
    unsigned newp[2];
    unsigned long long newr[2];
    for (int w = 0; w < 2; ++w) {
        const unsigned* h = hist + ((w == 1 && !same) ? KTH_BINS : 0);
        const unsigned long long rank = state->rank[w];
        unsigned c[PER];
        unsigned long long s = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            c[j] = (t * PER + j < NB) ? h[t * PER + j] : 0u;
            s += c[j];
        }
        part[t] = s;
        __syncthreads();
        if (t == 0) {
            unsigned long long run = 0;
            for (int i = 0; i < 256; ++i) {
                const unsigned long long v = part[i];
                part[i] = run;                                   // exclusive prefix
                run += v;
            }
        }
        __syncthreads();
        unsigned long long run = part[t];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (rank >= run && rank < run + c[j]) {              // exactly one (thread, j) holds the rank (rank < n)
                digit_s = (unsigned)(t * PER + j);
                before_s = run;
            }
            run += c[j];
        }
        __syncthreads();
        newp[w] = (PASS == 0 ? 0u : state->prefix[w] << BITS) | digit_s;
        newr[w] = rank - before_s;
        __syncthreads();
    }
    for (int i = t; i < 2 * KTH_BINS; i += 256) hist[i] = 0;
    if (t == 0) {
        for (int w = 0; w < 2; ++w) {
            state->prefix[w] = newp[w];
            state->rank[w] = newr[w];
            if (PASS == 2) out[w] = key_value(newp[w]);
        }
    }
}

__global__ void kth_init_kernel(KthState* state, unsigned* hist, unsigned long long k0, unsigned long long k1) {
    for (int i = threadIdx.x; i < 2 * KTH_BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) {
        state->prefix[0] = state->prefix[1] = 0;
        state->rank[0] = k0;
        state->rank[1] = k1;
    }
}
}  // namespace

size_t order_stats_workspace_bytes() { return 64 + sizeof(unsigned) * 2 * KTH_BINS; }

hipError_t launch_order_stats(const float* x, long n, long k, void* workspace, float* out2, hipStream_t st) {
    KthState* state = static_cast<KthState*>(workspace);
    unsigned* hist = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + 64);
    const long k1 = k + 1 < n ? k + 1 : n - 1;
    long blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);              // >= 8 float4 per thread
    blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(kth_init_kernel, dim3(1), dim3(256), 0, st, state, hist, (unsigned long long)k, (unsigned long long)k1);
    hipLaunchKernelGGL(kth_hist_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, state, hist);
    hipLaunchKernelGGL(kth_scan_kernel<0>, dim3(1), dim3(256), 0, st, state, hist, out2);
    hipLaunchKernelGGL(kth_hist_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, state, hist);
    hipLaunchKernelGGL(kth_scan_kernel<1>, dim3(1), dim3(256), 0, st, state, hist, out2);
    hipLaunchKernelGGL(kth_hist_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, state, hist);
    hipLaunchKernelGGL(kth_scan_kernel<2>, dim3(1), dim3(256), 0, st, state, hist, out2);
    return hipGetLastError();
}

long dice_blocks(long HW) { return (HW + DICE_PX_PER_BLOCK - 1) / DICE_PX_PER_BLOCK; }

hipError_t launch_dice_forward(const DiceArgs& a, double* partial, float* inter, float* sets, float* ce, hipStream_t st) {
    const long nb = dice_blocks(a.HW);
    hipLaunchKernelGGL(dice_fwd_kernel, dim3((unsigned)nb, (unsigned)a.B), dim3(256), 0, st, a, partial);
    hipLaunchKernelGGL(dice_final_kernel, dim3((unsigned)a.B), dim3(64), 0, st, partial, nb, a.C, inter, sets, ce);
    return hipGetLastError();
}

hipError_t launch_dice_backward(const DiceArgs& a, const float* g_inter, const float* g_sets, const float* g_ce, float* g_logits,
                                hipStream_t st) {
    hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)dice_blocks(a.HW), (unsigned)a.B), dim3(256), 0, st, a, g_inter, g_sets, g_ce,
                       g_logits);
    return hipGetLastError();
}

namespace {
// one workgroup: the inputs are a few hundred numbers; what this replaces is ~45 scalar autograd nodes forward and ~70 tiny kernels
// backward, issued one by one with the chip idle (0.3 + 0.6 ms per step, tools/micro/scalar_graph.py)
__global__ __launch_bounds__(256) void loss_combine_kernel(const CombineArgs a) {
    __shared__ double red[256];
    __shared__ float term[4];
    const int tid = threadIdx.x, nt = a.n_sup + a.n_cps;
    for (int i = 0; i < nt; ++i) {
        const int b = a.b[i], n = b * a.c;
        const float w = (i < a.n_sup ? 1.0f : a.cps_weight) / (float)n;
        float mean = 0.0f;
        for (int cls = 0; cls < a.c; ++cls) {               // class by class: the per-class batch mean first, as dice_loss.py:27
            double acc = 0.0;
            for (int bi = tid; bi < b; bi += 256) {
                const int e = bi * a.c + cls;
                const float den = a.sets[i][e] + a.eps, num = 2.0f * a.inter[i][e];
                acc += (double)(num / den);
                a.g_inter[i][e] = -w * (2.0f / den);
                a.g_sets[i][e] = w * (num / (den * den));
            }
            red[tid] = acc;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] += red[tid + s];
                __syncthreads();
            }
            mean += (float)(red[0] / (double)b);
            __syncthreads();
        }
        float v = 1.0f - mean / (float)a.c;
        if (a.ce[i]) {
            double s0 = 0.0, s1 = 0.0;
            for (int bi = tid; bi < b; bi += 256) s0 += (double)a.ce[i][2 * bi], s1 += (double)a.ce[i][2 * bi + 1];
            red[tid] = s0;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] += red[tid + s];
                __syncthreads();
            }
            const float t0 = (float)red[0];
            __syncthreads();
            red[tid] = s1;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] += red[tid + s];
                __syncthreads();
            }
            const float t1 = (float)red[0];
            __syncthreads();
            v = a.ce_weight * (t0 / t1) + v;
            const float gce = (i < a.n_sup ? 1.0f : a.cps_weight) * a.ce_weight / t1;
            const float gcnt = -gce * (t0 / t1);                // (the pixel count's own derivative: no kernel upstream uses it)
            for (int bi = tid; bi < b; bi += 256) a.g_ce[i][2 * bi] = gce, a.g_ce[i][2 * bi + 1] = gcnt;
        }
        if (tid == 0) term[i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float sup = 0.0f, cps = 0.0f;
        for (int i = 0; i < a.n_sup; ++i) sup = i ? sup + term[i] : term[i];
        for (int i = 0; i < a.n_cps; ++i) cps = i ? cps + term[a.n_sup + i] : term[a.n_sup];
        float com = 0.0f;
        for (int l = 0; l < a.levels; ++l) {
            float v = 0.0f;
            for (int k = 0; k < a.n_commit; ++k) v = k ? v + a.commit[k][l] : a.commit[k][l];
            com += v * a.commit_weight;
        }
        double pd = 0.0;
        for (int k = 0; k < a.n_proto; ++k) pd = k ? pd + *a.proto[k] : *a.proto[k];
        const float pro = (float)(pd * (double)a.proto_weight);
        a.out[0] = ((sup + a.cps_weight * cps) + com) + pro;
        a.out[1] = com;
        a.out[2] = pro;
        a.out[3] = cps;
        for (int i = 0; i < nt; ++i) a.out[4 + i] = term[i];
    }
}
}  // namespace

hipError_t launch_loss_combine(const CombineArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace vqseg
