// nn_abi.hip -- extern "C" entry points for the convolution / batch-norm / resampling kernels
// (declared in include/vqseg.h).  Validation + launch only: no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vqseg.h"
#include "conv_kernels.h"
#include "vq_kernels.h"
#include "nn_kernels.h"
#include "loss_kernels.h"
#include <math.h>

extern "C" int vqseg_set_error(int code, const char* msg);   // vqseg_abi.hip

namespace {
int bad(const char* msg) { return vqseg_set_error(VQSEG_EINVAL, msg); }
int hipfail(hipError_t e, const char* where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    return vqseg_set_error((int)e, buf);
}
bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int conv_rows_per_slot(int cout) { return cout >= 64 ? 64 : 32; }
}  // namespace

extern "C" {

size_t vqseg_conv_packed_elems(int cout, int cin, int kh, int kw, int transpose_flip) {
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return 0;
    return vqseg::packed_elems(cout, cin, kh, kw, transpose_flip);
}

int vqseg_conv_pack_weights_f32(const float* w, int cout, int cin, int kh, int kw, int transpose_flip, void* hi, void* lo,
                                void* stream) {
    if (!w || !hi || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return bad("conv_pack_weights: bad argument");
    hipError_t e = vqseg::launch_pack_weights(w, cout, cin, kh, kw, transpose_flip, static_cast<unsigned short*>(hi),
                                              static_cast<unsigned short*>(lo), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv_pack_weights");
}

int64_t vqseg_conv_stat_slots(int64_t m_rows, int cout) {
    if (m_rows <= 0 || cout <= 0) return 0;
    return (m_rows + 255) / 256 * (256 / conv_rows_per_slot(cout));       // whole 256-row tiles, 64- or 32-row slots
}

static int conv2d_impl(const void* x, const void* x2, int c1, const void* w_hi, const void* w_lo, void* y, float* stat_partial,
                       const float* ep_scale, const float* ep_shift, const void* ep_res, int ep_relu,
                       int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int reflect, int up, int ho,
                       int wo, int precise, void* stream, const unsigned char* ep_res_bits = nullptr) {
    if (!x || !w_hi || !y || (precise && !w_lo)) return bad("conv2d: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ho <= 0 || wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || up <= 0)
        return bad("conv2d: non-positive dimension");
    const int epc = precise ? 4 : 8;                       // channels per 16-byte activation chunk
    if (cin % epc) return bad(precise ? "conv2d: Cin must be a multiple of 4" : "conv2d: Cin must be a multiple of 8 in bf16 mode");
    if (c1 <= 0 || c1 > cin || (c1 < cin && (!x2 || c1 % epc || (cin - c1) % epc))) return bad("conv2d: bad channel split");
    if (!a16(x) || !a16(x2) || !a16(w_hi) || !a16(w_lo) || !a16(y)) return bad("conv2d: pointers must be 16-byte aligned");
    if (reflect && (pad >= h * up || pad >= w * up)) return bad("conv2d: reflect padding needs pad < size");
    if (up != 1 && up != 2) return bad("conv2d: up (input dilation) must be 1 or 2");
    vqseg::ConvArgs a;
    a.x = x; a.x2 = x2; a.C1 = c1;
    a.w_hi = static_cast<const unsigned short*>(w_hi);
    a.w_lo = static_cast<const unsigned short*>(w_lo);
    a.y = y; a.stat_partial = stat_partial;
    a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kh; a.KW = kw;
    a.stride = stride; a.pad = pad; a.reflect = reflect; a.up = up;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_res = ep_res; a.ep_relu = ep_relu; a.ep_res_bits = ep_res_bits;
    hipError_t e = vqseg::launch_conv(a, precise, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv_igemm_kernel");
}

int vqseg_conv2d_f(const void* x, const void* x2, int c1, const void* w_hi, const void* w_lo, void* y, float* stat_partial,
                   int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int reflect, int up, int ho,
                   int wo, int precise, void* stream) {
    return conv2d_impl(x, x2, c1, w_hi, w_lo, y, stat_partial, nullptr, nullptr, nullptr, 0, n, h, w, cin, cout, kh, kw, stride, pad,
                       reflect, up, ho, wo, precise, stream);
}

int vqseg_conv2d_affine_f(const void* x, const void* x2, int c1, const void* w_hi, const void* w_lo, const float* scale,
                          const float* shift, const void* res, int relu, void* y, int n, int h, int w, int cin, int cout, int kh,
                          int kw, int stride, int pad, int reflect, int ho, int wo, int precise, void* stream) {
    if (!scale || !shift) return bad("conv2d_affine: null scale / shift");
    if (res && !a16(res)) return bad("conv2d_affine: pointers must be 16-byte aligned");
    if (precise == 2) {
        // split-3: x / x2 / res / y are [hi | lo] bf16 tensors of 2 * C channels, w_hi the [w_hi | w_hi | w_lo] image
        // (vqseg_conv_pack_weights_s3_f32); the bf16 kernels contract over 3 * cin channels, the third part re-reading hi
        if (cin % 32 || cout % 8 || c1 % 32 || c1 <= 0 || c1 > cin) return bad("conv2d_affine (split-3): needs Cin, C1 % 32 == 0 and Cout % 8 == 0");
        if (!x || !w_hi || !y || (c1 < cin && !x2)) return bad("conv2d: null pointer");
        if (n <= 0 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return bad("conv2d: non-positive dimension");
        if (!a16(x) || !a16(x2) || !a16(w_hi) || !a16(y)) return bad("conv2d: pointers must be 16-byte aligned");
        if (reflect && (pad >= h || pad >= w)) return bad("conv2d: reflect padding needs pad < size");
        vqseg::ConvArgs a;
        a.x = x; a.x2 = x2; a.C1 = 3 * c1;
        a.w_hi = static_cast<const unsigned short*>(w_hi); a.w_lo = nullptr;
        a.y = y; a.stat_partial = nullptr;
        a.N = n; a.H = h; a.W = w; a.Cin = 3 * cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kh; a.KW = kw;
        a.stride = stride; a.pad = pad; a.reflect = reflect; a.up = 1;
        a.ep_scale = scale; a.ep_shift = shift; a.ep_res = res; a.ep_relu = relu;
        a.out_s3 = 1;
        a.s3_in = 1; a.s3_cs1 = c1; a.s3_cs2 = cin - c1;
        hipError_t e = vqseg::launch_conv(a, 0, static_cast<hipStream_t>(stream));
        return e == hipSuccess ? 0 : hipfail(e, "conv kernel (split-3)");
    }
    return conv2d_impl(x, x2, c1, w_hi, w_lo, y, nullptr, scale, shift, res, relu, n, h, w, cin, cout, kh, kw, stride, pad, reflect, 1,
                       ho, wo, precise, stream);
}

int vqseg_conv2d_affine_bits_f(const void* x, const void* w_hi, const float* scale, const float* shift, const void* res,
                               const unsigned char* res_bits, void* y, int n, int h, int w, int cin, int cout, int kh, int kw, int pad,
                               int ho, int wo, void* stream) {
    if (!scale || !shift || !res || !res_bits) return bad("conv2d_affine_bits: null pointer");
    if (!a16(res)) return bad("conv2d_affine_bits: pointers must be 16-byte aligned");
    if (cout % 8) return bad("conv2d_affine_bits: Cout must be a multiple of 8");
    return conv2d_impl(x, nullptr, cin, w_hi, nullptr, y, nullptr, scale, shift, res, 0, n, h, w, cin, cout, kh, kw, 1, pad, 0, 1, ho, wo, 0,
                       stream, res_bits);
}

int vqseg_conv_profile_begin(int capacity) {
    if (capacity <= 0 || capacity > (1 << 20)) return bad("conv_profile_begin: capacity out of range");
    hipError_t e = vqseg::conv_profile_begin(capacity);
    return e == hipSuccess ? 0 : hipfail(e, "conv_profile_begin");
}

int vqseg_conv_profile_collect(int max_records, double* flops_host, int* kind_host, float* ms_host, int* shape_host) {
    if (max_records <= 0 || !flops_host || !kind_host || !ms_host) return bad("conv_profile_collect: bad argument");
    return vqseg::conv_profile_collect(max_records, flops_host, kind_host, ms_host, shape_host);
}

int vqseg_set_option(const char* key, int value) {
    if (!key || value < 0) return bad("set_option: null key or negative value");
    int prev = vqseg::conv_set_option(key, value);
    if (prev < 0) prev = vqseg::vq_set_option(key, value);
    if (prev < 0) prev = vqseg::nn_set_option(key, value);
    return prev < 0 ? bad("set_option: unknown key") : prev;
}

size_t vqseg_conv2d_wgrad_workspace_bytes(int n, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw) {
    if (n <= 0 || cin <= 0 || cout <= 0) return 0;
    vqseg::WgradArgs a{};
    a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kh; a.KW = kw;
    return (size_t)vqseg::wgrad_slabs_max(a) * cout * kh * kw * cin * sizeof(float);
}

int vqseg_conv2d_wgrad2_f(const void* gy, const void* x, const void* x2, int n, const void* gy_b, const void* x_b, const void* x2_b,
                          int n_b, int c1, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad,
                          int reflect, int precise, int cin_out, int im2col, int accumulate, void* workspace, size_t workspace_bytes,
                          float* gw, void* stream) {
    if (!gy || !x || !workspace || !gw) return bad("conv2d_wgrad: null pointer");
    if (n <= 0 || n_b < 0) return bad("conv2d_wgrad: bad image count");
    if (n_b > 0 && (!gy_b || !x_b)) return bad("conv2d_wgrad: second source missing");
    const int epc = precise ? 4 : 8;
    if (cin % epc || cout % epc) return bad("conv2d_wgrad: Cin and Cout must be multiples of 4 (f32) / 8 (bf16)");
    if (c1 <= 0 || c1 > cin || (c1 < cin && (!x2 || (n_b > 0 && !x2_b) || c1 % epc || (cin - c1) % epc))) return bad("conv2d_wgrad: bad channel split");
    vqseg::WgradArgs a;
    const int kkh = im2col ? 1 : kh, kkw = im2col ? 1 : kw;     // a patch matrix is convolved 1x1
    a.gy = gy; a.x = x; a.x2 = x2; a.C1 = c1; a.partial = static_cast<float*>(workspace);
    a.gy_b = gy_b; a.x_b = x_b; a.x2_b = x2_b; a.Na = n;
    a.N = n + n_b; a.H = h; a.W = w; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kkh; a.KW = kkw;
    a.stride = stride; a.pad = pad; a.reflect = reflect; a.per_tap_only = im2col;
    const int slabs = vqseg::wgrad_slabs(a, precise);
    if (workspace_bytes < (size_t)slabs * cout * kkh * kkw * cin * sizeof(float)) return vqseg_set_error(VQSEG_ENOSPC, "conv2d_wgrad: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int final_layout = 0;
    hipError_t e = vqseg::launch_wgrad(a, precise, slabs, &final_layout, st);
    if (e != hipSuccess) return hipfail(e, "conv_wgrad_kernel");
    if (final_layout && (cin_out != cin || im2col)) return bad("conv2d_wgrad: internal layout mismatch");
    if (final_layout) e = vqseg::launch_reduce_partials(a.partial, slabs, (long)cout * cin * kkh * kkw, gw, accumulate, st);
    else if (im2col) e = vqseg::launch_wgrad_reduce(a.partial, slabs, cout, cin, cin_out, kh, kw, 1, accumulate, gw, st);   // kh,kw = ORIGINAL taps here
    else e = vqseg::launch_wgrad_reduce(a.partial, slabs, cout, cin, cin_out, kh, kw, 0, accumulate, gw, st);
    return e == hipSuccess ? 0 : hipfail(e, "wgrad_reduce_kernel");
}

int vqseg_conv2d_wgrad_f(const void* gy, const void* x, const void* x2, int c1, int n, int h, int w, int cin, int ho, int wo,
                         int cout, int kh, int kw, int stride, int pad, int reflect, int precise, int cin_out, int im2col,
                         int accumulate, void* workspace, size_t workspace_bytes, float* gw, void* stream) {
    return vqseg_conv2d_wgrad2_f(gy, x, x2, n, nullptr, nullptr, nullptr, 0, c1, h, w, cin, ho, wo, cout, kh, kw, stride, pad, reflect,
                                 precise, cin_out, im2col, accumulate, workspace, workspace_bytes, gw, stream);
}

int vqseg_bn_sync_ints(int c) { return c > 0 ? (c + 63) / 64 : 0; }

int vqseg_bn_finalize_f(float* partial, int64_t m_rows, int c, const float* gamma, const float* beta, float* run_mean,
                        float* run_var, float momentum, float eps, int training, float* scale, float* shift, float* save_mean,
                        float* save_invstd, int64_t* num_batches_tracked, int* sync, void* stream) {
    if (!gamma || !beta || !scale || !shift || !save_mean || !save_invstd || c <= 0 || m_rows <= 0) return bad("bn_finalize: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (training) {
        if (!partial) return bad("bn_finalize: training mode needs the conv epilogue partials");
        e = vqseg::launch_bn_finalize(partial, (m_rows + conv_rows_per_slot(c) - 1) / conv_rows_per_slot(c), conv_rows_per_slot(c), m_rows, c, gamma, beta,
                                      run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd,
                                      reinterpret_cast<long long*>(num_batches_tracked), sync, st);
    } else {
        if (!run_mean || !run_var) return bad("bn_finalize: eval mode needs running statistics");
        e = vqseg::launch_bn_eval_coeffs(c, gamma, beta, run_mean, run_var, eps, scale, shift, save_mean, save_invstd, st);
    }
    return e == hipSuccess ? 0 : hipfail(e, "bn_finalize");
}

int vqseg_bn_apply_f(int bf16, const void* y, const void* res, const float* scale, const float* shift, int64_t m_rows, int c,
                     int relu, void* out, void* stream) {
    if (!y || !scale || !shift || !out || c <= 0 || m_rows <= 0) return bad("bn_apply: bad argument");
    hipError_t e = vqseg::launch_bn_apply(bf16, y, res, scale, shift, m_rows, c, relu, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "bn_apply_kernel");
}

int vqseg_bn_apply_bits_f(const void* y, const void* res, const float* scale, const float* shift, int64_t m_rows, int c, void* out,
                          unsigned char* bits, void* stream) {
    if (!y || !scale || !shift || !out || !bits || c <= 0 || m_rows <= 0) return bad("bn_apply_bits: bad argument");
    if (c % 8) return bad("bn_apply_bits: the channel count must be a multiple of 8");
    hipError_t e = vqseg::launch_bn_apply(1, y, res, scale, shift, m_rows, c, 1, out, static_cast<hipStream_t>(stream), bits);
    return e == hipSuccess ? 0 : hipfail(e, "bn_apply_kernel");
}

size_t vqseg_bn_backward_workspace_floats(int64_t m_rows, int c) {
    if (m_rows <= 0 || c <= 0) return 0;
    return (size_t)vqseg::bn_bwd_blocks(m_rows) * 2 * c + 3 * (size_t)c;
}

int vqseg_bn_backward_f(int bf16, const void* g_out, const void* out, const void* y, const float* mean, const float* invstd,
                        const float* gamma, const float* fwd_scale, const float* fwd_shift, int64_t m_rows, int c, int relu,
                        int training, int accumulate, float* workspace, float* dgamma, float* dbeta, void* g_y, void* g_res,
                        int* sync, void* stream) {
    if (!g_out || !y || !mean || !invstd || !gamma || !workspace || !dgamma || !dbeta || !g_y) return bad("bn_backward: null pointer");
    if (relu && !out && (!fwd_scale || !fwd_shift)) return bad("bn_backward: with ReLU pass either `out` or the forward scale/shift");
    if (relu && !out && g_res) return bad("bn_backward: a residual branch needs `out` (the mask depends on the residual)");
    if (c <= 0) return bad("bn_backward: unsupported channel count");
    float* partial = workspace;
    float* coef = workspace + (size_t)vqseg::bn_bwd_blocks(m_rows) * 2 * c;
    hipError_t e = vqseg::launch_bn_backward(bf16, g_out, out, y, mean, invstd, gamma, fwd_scale, fwd_shift, m_rows, c, relu, training, accumulate, partial, coef,
                                             dgamma, dbeta, g_y, g_res, sync, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "bn_backward");
}

int vqseg_bn_backward_bits_f(const void* g_out, const unsigned char* bits, const void* y, const float* mean, const float* invstd,
                             const float* gamma, int64_t m_rows, int c, int training, int accumulate, float* workspace, float* dgamma,
                             float* dbeta, void* g_y, void* g_res, int* sync, void* stream) {
    if (!g_out || !bits || !y || !mean || !invstd || !gamma || !workspace || !dgamma || !dbeta || !g_y)
        return bad("bn_backward_bits: null pointer");
    if (c <= 0 || c % 8) return bad("bn_backward_bits: the channel count must be a multiple of 8");
    float* partial = workspace;
    float* coef = workspace + (size_t)vqseg::bn_bwd_blocks(m_rows) * 2 * c;
    hipError_t e = vqseg::launch_bn_backward(1, g_out, nullptr, y, mean, invstd, gamma, nullptr, nullptr, m_rows, c, 1, training, accumulate, partial,
                                             coef, dgamma, dbeta, g_y, g_res, sync, static_cast<hipStream_t>(stream), bits);
    return e == hipSuccess ? 0 : hipfail(e, "bn_backward_bits");
}

int vqseg_maxpool3x3s2_f(int bf16, int backward, const void* x, const void* g, int n, int h, int w, int c, void* out,
                         unsigned char* idx, void* stream) {
    if (!out || (!x && !(backward && idx)) || (backward && !g)) return bad("maxpool: null pointer");
    hipError_t e = vqseg::launch_maxpool(bf16, backward, x, g, n, h, w, c, out, idx, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "maxpool");
}

int vqseg_bilinear_f(int bf16, int backward, const void* src, int n, int h, int w, int c, int ho, int wo, int align_corners,
                     void* dst, void* stream) {
    if (!src || !dst || n <= 0 || h <= 0 || w <= 0 || c <= 0 || ho <= 0 || wo <= 0) return bad("bilinear: bad argument");
    hipError_t e = vqseg::launch_bilinear(bf16, backward, src, n, h, w, c, ho, wo, align_corners, dst, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "bilinear");
}

int vqseg_head1x1_forward_f(int bf16, const void* x, const float* w, int64_t m_rows, int cin, int cout, float* y, void* stream) {
    if (!x || !w || !y || cout > 4 || cout <= 0 || cin <= 0 || cin % 8 || cin > 64 || !a16(x))
        return bad("head1x1: bad argument (Cout <= 4, Cin % 8 == 0, Cin <= 64, 16-byte aligned rows)");
    if (bf16 < 0 || bf16 > 2) return bad("head1x1: row type must be 0 (f32), 1 (bf16) or 2 (split-3)");
    hipError_t e = vqseg::launch_head_fwd(bf16, x, w, m_rows, cin, cout, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "head_fwd_kernel");
}

size_t vqseg_head1x1_backward_workspace_floats(int64_t m_rows, int cin, int cout) {
    return (size_t)vqseg::head_bwd_blocks(m_rows) * cin * cout;
}

int vqseg_head1x1_backward_f(int bf16, const void* x, const float* w, const float* g, int64_t m_rows, int cin, int cout, void* gx,
                             float* gw, float* workspace, void* stream) {
    if (!x || !w || !g || !gx || !gw || !workspace || cout > 4 || cout <= 0 || cin <= 0 || cin * cout > 256 || cin % 8 || cin > 64 || !a16(x) || !a16(gx))
        return bad("head1x1 backward: bad argument (Cout <= 4, Cin % 8 == 0, Cin <= 64, 16-byte aligned rows)");
    hipError_t e = vqseg::launch_head_bwd(bf16, x, w, g, m_rows, cin, cout, gx, gw, workspace, nullptr, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "head_bwd");
}

int vqseg_head1x1_backward_add_f(int bf16, const void* x, const float* w, const float* g, int64_t m_rows, int cin, int cout, void* gx,
                                 float* gw, float* workspace, const void* gx_add, void* stream) {
    if (!x || !w || !g || !gx || !gw || !workspace || cout > 4 || cout <= 0 || cin <= 0 || cin * cout > 256 || cin % 8 || cin > 64 || !a16(x) || !a16(gx) ||
        !a16(gx_add))
        return bad("head1x1 backward: bad argument (Cout <= 4, Cin % 8 == 0, Cin <= 64, 16-byte aligned rows)");
    hipError_t e = vqseg::launch_head_bwd(bf16, x, w, g, m_rows, cin, cout, gx, gw, workspace, gx_add, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "head_bwd");
}

int vqseg_maxpool3x3s2_backward_add_f(int bf16, const void* g, const unsigned char* idx, const void* gx_add, int n, int h, int w, int c,
                                      void* gx, void* stream) {
    if (!g || !idx || !gx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return bad("maxpool backward: null pointer or bad size");
    hipError_t e = vqseg::launch_maxpool(bf16, gx_add ? 2 : 1, gx_add, g, n, h, w, c, gx, const_cast<unsigned char*>(idx), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "maxpool");
}

int vqseg_im2col_f(int out_bf16, const float* x, int n, int h, int w, int cin, int kh, int kw, int stride, int pad, int reflect,
                   int ho, int wo, int kp, void* out, void* stream) {
    if (!x || !out || kp < kh * kw * cin) return bad("im2col: bad argument");
    hipError_t e = vqseg::launch_im2col_stem(out_bf16, x, n, h, w, cin, kh, kw, stride, pad, reflect, ho, wo, kp, out,
                                             static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "im2col_stem_kernel");
}

int vqseg_conv_pack_all_f32(const float* w, int cout, int cin, int k, int c1, void* fwd, void* tr, void* s3, void* stream) {
    if (!w || cout <= 0 || cin <= 0 || (k != 1 && k != 3) || c1 <= 0 || c1 > cin || (!fwd && !tr && !s3)) return bad("conv_pack_all: bad argument (k = 1 or 3)");
    if (s3 && (cin % 32 || c1 % 32)) return bad("conv_pack_all: the split-3 image needs Cin and the concat split to be multiples of 32");
    hipError_t e = vqseg::launch_pack_all(w, cout, cin, k, c1, static_cast<unsigned short*>(fwd), static_cast<unsigned short*>(tr),
                                          static_cast<unsigned short*>(s3), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv_pack_all_kernel");
}

size_t vqseg_conv_packed_s2_elems(int cout, int cin, int k) {
    if (cout <= 0 || cin <= 0 || (k != 1 && k != 3)) return 0;
    return (size_t)cin * k * k * ((cout + 31) / 32 * 32);
}

int vqseg_conv_pack_weights_s2_f32(const float* w, int cout, int cin, int k, void* hi, void* lo, void* stream) {
    if (!w || !hi || cout <= 0 || cin <= 0 || (k != 1 && k != 3)) return bad("conv_pack_weights_s2: bad argument (k = 1 or 3)");
    hipError_t e = vqseg::launch_pack_weights_s2(w, cout, cin, k, static_cast<unsigned short*>(hi), static_cast<unsigned short*>(lo),
                                                 static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv_pack_weights_s2");
}

int vqseg_conv2d_dgrad_s2_f(const void* gy, const void* w_hi, const void* w_lo, void* gx, int n, int ho, int wo, int cout, int cin, int k,
                            int oh, int ow, int precise, int accumulate, void* stream) {
    if (accumulate && k != 1) return bad("conv2d_dgrad_s2: accumulate is implemented for k == 1");
    if (!gy || !w_hi || !gx || (precise && !w_lo)) return bad("conv2d_dgrad_s2: null pointer");
    if (n <= 0 || ho <= 0 || wo <= 0 || cout <= 0 || cin <= 0 || oh <= 0 || ow <= 0 || (k != 1 && k != 3)) return bad("conv2d_dgrad_s2: bad dimension");
    const int epc = precise ? 4 : 8;
    if (cout % epc || cin % epc) return bad("conv2d_dgrad_s2: channels must be multiples of 4 (f32) / 8 (bf16)");
    if (!a16(gy) || !a16(w_hi) || !a16(w_lo) || !a16(gx)) return bad("conv2d_dgrad_s2: pointers must be 16-byte aligned");
    // the grid must be the one the forward layer's geometry implies: k = 3 -> padded input (2 ho + 1 or 2 ho + 2 rows), k = 1 -> 2 ho - 1 or 2 ho
    if (oh < 2 * ho - 1 || oh > 2 * ho + 2 || ow < 2 * wo - 1 || ow > 2 * wo + 2) return bad("conv2d_dgrad_s2: output grid does not match a stride-2 layer");
    hipError_t e = vqseg::launch_dgrad_s2(gy, static_cast<const unsigned short*>(w_hi), static_cast<const unsigned short*>(w_lo), gx, n, ho, wo,
                                          cout, cin, k, oh, ow, precise, accumulate, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv2d_dgrad_s2");
}

int vqseg_stem7_conv_f(int s3, const float* x, const void* w_img, void* y, float* stat_partial, const float* scale, const float* shift, int relu,
                       int n, int h, int w, int reflect, void* stream) {
    if (!x || !w_img || !y) return bad("stem7_conv: null pointer");
    if (n <= 0 || h < 4 || w < 4) return bad("stem7_conv: bad image size");
    if ((scale == nullptr) != (shift == nullptr)) return bad("stem7_conv: scale and shift come together");
    if (s3 && !scale) return bad("stem7_conv (split-3): needs the fused affine epilogue");
    if (!a16(x) || !a16(w_img) || !a16(y)) return bad("stem7_conv: pointers must be 16-byte aligned");
    hipError_t e = vqseg::launch_stem7_fused(x, static_cast<const unsigned short*>(w_img), y, stat_partial, scale, shift, relu, n, h, w, reflect, s3,
                                             static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return bad("stem7_conv: shape outside the fused path (output width % 128) or option stem_fused = 0");
    return e == hipSuccess ? 0 : hipfail(e, "stem7_fused_kernel");
}

int64_t vqseg_conv2d_dgrad_s2_fold_rows(int n, int h, int w, int reflect) {
    return (n > 0 && h > 0 && w > 0) ? (int64_t)vqseg::dgrad_s2_fold_rows(n, h, w, reflect) : 0;
}

int vqseg_conv2d_dgrad_s2_fold_f(const void* gy, const void* w_hi, void* gx, int n, int ho, int wo, int cout, int cin, int h, int w,
                                 int reflect, void* stream) {
    if (!gy || !w_hi || !gx) return bad("conv2d_dgrad_s2_fold: null pointer");
    if (n <= 0 || ho <= 0 || wo <= 0 || h != 2 * ho || w != 2 * wo || h < 4 || w < 4) return bad("conv2d_dgrad_s2_fold: needs H = 2 Ho, W = 2 Wo, H, W >= 4");
    if (cout <= 0 || cin < 64 || cout % 64 || cin % 8) return bad("conv2d_dgrad_s2_fold: needs Cout % 64 == 0, Cin % 8 == 0, Cin >= 64");
    if (!a16(gy) || !a16(w_hi) || !a16(gx)) return bad("conv2d_dgrad_s2_fold: pointers must be 16-byte aligned");
    hipError_t e = vqseg::launch_dgrad_s2_fold(gy, static_cast<const unsigned short*>(w_hi), gx, n, ho, wo, cout, cin, h, w, reflect,
                                               static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return bad("conv2d_dgrad_s2_fold: shape outside the merged path (or option conv_dgrad_s2_merge = 0)");
    return e == hipSuccess ? 0 : hipfail(e, "conv2d_dgrad_s2_fold");
}

int vqseg_reflect_ring_f(const void* gy, const void* t_hi, void* ring, void* gx, int n, int h, int w, int cgy, int cgx, void* stream) {
    if (!gy || !t_hi || !ring || !gx) return bad("reflect_ring: null pointer");
    if (n <= 0 || h < 4 || w < 4 || cgy <= 0 || cgx <= 0 || cgy % 64 || cgx % 8) return bad("reflect_ring: needs H, W >= 4, Cgy % 64 == 0, Cgx % 8 == 0");
    if (!a16(gy) || !a16(t_hi) || !a16(ring) || !a16(gx)) return bad("reflect_ring: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = vqseg::launch_reflect_ring(gy, static_cast<const unsigned short*>(t_hi), ring, n, h, w, cgy, cgx, st);
    if (e != hipSuccess) return hipfail(e, "reflect ring convolution");
    e = vqseg::launch_reflect_ring_fold(1, ring, n, h, w, cgx, gx, st);
    return e == hipSuccess ? 0 : hipfail(e, "reflect_ring_fold");
}

int vqseg_conv_pack_weights_s3_f32(const float* w, int cout, int cin, int c1, int kh, int kw, void* out, void* stream) {
    if (!w || !out || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || c1 <= 0 || c1 > cin || cin % 32 || c1 % 32)
        return bad("conv_pack_weights_s3: bad argument (Cin and the concat split must be multiples of 32)");
    hipError_t e = vqseg::launch_pack_weights_s3(w, cout, cin, c1, kh, kw, static_cast<unsigned short*>(out), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "conv_pack_weights_s3");
}

int vqseg_s3_split_f(const float* x, int64_t rows, int c, void* y, void* stream) {
    if (!x || !y || rows <= 0 || c <= 0 || c % 8 || !a16(x) || !a16(y)) return bad("s3_split: bad argument (channels % 8, 16-byte aligned)");
    hipError_t e = vqseg::launch_s3_split(x, rows, c, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "s3_split_kernel");
}

int vqseg_s3_merge_f(const void* x, int64_t rows, int c, float* y, void* stream) {
    if (!x || !y || rows <= 0 || c <= 0 || c % 8 || !a16(x) || !a16(y)) return bad("s3_merge: bad argument (channels % 8, 16-byte aligned)");
    hipError_t e = vqseg::launch_s3_merge(x, rows, c, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "s3_merge_kernel");
}

int vqseg_s3_maxpool3x3s2_f(const void* x, int n, int h, int w, int c, void* y, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 || !a16(x) || !a16(y)) return bad("s3_maxpool: bad argument");
    hipError_t e = vqseg::launch_s3_maxpool(x, n, h, w, c, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "s3_maxpool_kernel");
}

int vqseg_s3_bilinear_f(const void* x, int n, int h, int w, int c, int ho, int wo, int align_corners, void* y, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0 || c <= 0 || c % 8 || !a16(x) || !a16(y)) return bad("s3_bilinear: bad argument");
    hipError_t e = vqseg::launch_s3_bilinear(x, n, h, w, c, ho, wo, align_corners, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "s3_bilinear_kernel");
}

int vqseg_reflect_fold_f(int bf16, const void* gp, int n, int h, int w, int c, void* gx, void* stream) {
    if (!gp || !gx || h < 2 || w < 2) return bad("reflect_fold: bad argument");
    hipError_t e = vqseg::launch_reflect_fold(bf16, gp, n, h, w, c, gx, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "reflect_fold_kernel");
}

int vqseg_cast_f(int to_bf16, const void* x, int64_t n, void* y, void* stream) {
    if (!x || !y || n <= 0) return bad("cast: bad argument");
    hipError_t e = vqseg::launch_cast(to_bf16, x, n, y, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "cast_kernel");
}

static int proto_args(vqseg::ProtoArgs& a, int bf16, const void* x, const float* proto, const int64_t* labels, const unsigned char* keep,
                      const float* conf, int64_t m, int c, int k, int variant, float scale, float margin, int easy_margin) {
    if (!x || !proto || !labels || m <= 0) return bad("proto_loss: null pointer or empty input");
    if (c <= 0 || c > 64 || c % 8 || k <= 0 || k > 4) return bad("proto_loss: needs channels % 8 == 0, <= 64 and at most 4 classes");
    if (variant != 1 && variant != 2) return bad("proto_loss: variant must be 1 or 2");
    a.x = x; a.bf16 = bf16; a.proto = proto; a.labels = reinterpret_cast<const long long*>(labels); a.keep = keep; a.conf = conf;
    a.M = m; a.C = c; a.K = k; a.variant = variant; a.scale = scale;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mm = (float)(sin(M_PI - (double)margin) * (double)margin);
    a.easy_margin = easy_margin; a.use_margin = margin != 0.0f;
    return 0;
}

size_t vqseg_proto_loss_workspace_bytes(int64_t m, int c, int k) {
    if (m <= 0 || c <= 0 || k <= 0) return 0;
    const size_t nb = (size_t)vqseg::proto_blocks(m);
    return nb * sizeof(double) + nb * (size_t)k * c * sizeof(float);
}

int vqseg_proto_loss_forward_f(int bf16, const void* x, const float* proto, const int64_t* labels, const unsigned char* keep,
                               const float* conf, int64_t m, int c, int k, int variant, float scale, float margin, int easy_margin,
                               void* workspace, size_t workspace_bytes, double* loss, void* stream) {
    vqseg::ProtoArgs a;
    if (int rc = proto_args(a, bf16, x, proto, labels, keep, conf, m, c, k, variant, scale, margin, easy_margin)) return rc;
    if (!workspace || !loss) return bad("proto_loss: null pointer");
    if (workspace_bytes < vqseg_proto_loss_workspace_bytes(m, c, k)) return vqseg_set_error(VQSEG_ENOSPC, "proto_loss: workspace too small");
    hipError_t e = vqseg::launch_proto_forward(a, static_cast<double*>(workspace), loss, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "proto_fwd_kernel");
}

int vqseg_proto_loss_backward_f(int bf16, const void* x, const float* proto, const int64_t* labels, const unsigned char* keep,
                                const float* conf, int64_t m, int c, int k, int variant, float scale, float margin, int easy_margin,
                                const float* g_loss, void* gx, float* gproto, void* workspace, size_t workspace_bytes, void* stream) {
    vqseg::ProtoArgs a;
    if (int rc = proto_args(a, bf16, x, proto, labels, keep, conf, m, c, k, variant, scale, margin, easy_margin)) return rc;
    if (!workspace || !g_loss || !gx) return bad("proto_loss: null pointer");
    if (workspace_bytes < vqseg_proto_loss_workspace_bytes(m, c, k)) return vqseg_set_error(VQSEG_ENOSPC, "proto_loss: workspace too small");
    float* gpp = gproto ? reinterpret_cast<float*>(static_cast<char*>(workspace) + (size_t)vqseg::proto_blocks(m) * sizeof(double)) : nullptr;
    hipError_t e = vqseg::launch_proto_backward(a, g_loss, gx, gpp, gproto, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "proto_bwd_kernel");
}

static int dice_args(vqseg::DiceArgs& a, const float* logits, int64_t sb, int64_t sc, int64_t sp, const int64_t* target, int b, int c,
                     int64_t hw, int64_t ignore_index) {
    if (!logits || !target || b <= 0 || hw <= 0) return bad("dice: null pointer or empty input");
    if (c < 2 || c > 4) return bad("dice: 2..4 classes (the softmax form)");
    a.logits = logits; a.sb = sb; a.sc = sc; a.sp = sp; a.target = reinterpret_cast<const long long*>(target);
    a.B = b; a.C = c; a.HW = hw; a.ignore_index = ignore_index;
    return 0;
}

size_t vqseg_dice_workspace_bytes(int b, int c, int64_t hw) {
    if (b <= 0 || c <= 0 || hw <= 0) return 0;
    return (size_t)b * vqseg::dice_blocks(hw) * (2 * c + 2) * sizeof(double);
}

int vqseg_dice_ce_sums_forward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, const int64_t* target, int b,
                                 int c, int64_t hw, int64_t ignore_index, void* workspace, size_t workspace_bytes, float* inter,
                                 float* sets, float* ce, void* stream) {
    vqseg::DiceArgs a;
    if (int rc = dice_args(a, logits, stride_b, stride_c, stride_px, target, b, c, hw, ignore_index)) return rc;
    if (!workspace || !inter || !sets) return bad("dice: null pointer");
    if (workspace_bytes < vqseg_dice_workspace_bytes(b, c, hw)) return vqseg_set_error(VQSEG_ENOSPC, "dice: workspace too small");
    hipError_t e = vqseg::launch_dice_forward(a, static_cast<double*>(workspace), inter, sets, ce, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "dice_fwd_kernel");
}

int vqseg_dice_ce_sums_backward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, const int64_t* target, int b,
                                  int c, int64_t hw, int64_t ignore_index, const float* g_inter, const float* g_sets, const float* g_ce,
                                  float* g_logits, void* stream) {
    vqseg::DiceArgs a;
    if (int rc = dice_args(a, logits, stride_b, stride_c, stride_px, target, b, c, hw, ignore_index)) return rc;
    if (!g_inter || !g_sets || !g_logits) return bad("dice: null pointer");
    hipError_t e = vqseg::launch_dice_backward(a, g_inter, g_sets, g_ce, g_logits, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "dice_bwd_kernel");
}

int vqseg_dice_sums_forward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, const int64_t* target, int b,
                              int c, int64_t hw, int64_t ignore_index, void* workspace, size_t workspace_bytes, float* inter,
                              float* sets, void* stream) {
    return vqseg_dice_ce_sums_forward_f(logits, stride_b, stride_c, stride_px, target, b, c, hw, ignore_index, workspace, workspace_bytes,
                                        inter, sets, nullptr, stream);
}

int vqseg_dice_sums_backward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, const int64_t* target, int b,
                               int c, int64_t hw, int64_t ignore_index, const float* g_inter, const float* g_sets, float* g_logits,
                               void* stream) {
    return vqseg_dice_ce_sums_backward_f(logits, stride_b, stride_c, stride_px, target, b, c, hw, ignore_index, g_inter, g_sets, nullptr,
                                         g_logits, stream);
}

int vqseg_softmax_stats_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, int b, int c, int64_t hw,
                          int64_t* label, float* entropy, float* top, void* stream) {
    if (!logits || b <= 0 || hw <= 0 || c < 2 || c > 4) return bad("softmax_stats: bad argument (2..4 classes)");
    vqseg::DiceArgs a;
    a.logits = logits; a.sb = stride_b; a.sc = stride_c; a.sp = stride_px; a.target = nullptr; a.B = b; a.C = c; a.HW = hw; a.ignore_index = 0;
    hipError_t e = vqseg::launch_softmax_stats(a, reinterpret_cast<long long*>(label), entropy, top, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "softmax_stats_kernel");
}

int vqseg_confusion_counts_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, const int64_t* target, int b,
                             int c, int64_t hw, int64_t* counts, void* stream) {
    vqseg::DiceArgs a;
    if (int rc = dice_args(a, logits, stride_b, stride_c, stride_px, target, b, c, hw, 0)) return rc;
    if (!counts) return bad("confusion_counts: null pointer");
    hipError_t e = vqseg::launch_confusion(a, reinterpret_cast<long long*>(counts), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "confusion_kernel");
}

int vqseg_cps_loss_combine_f(int n_sup, int n_cps, int c, const float* const* inter, const float* const* sets, const float* const* ce,
                             const int* b, float cps_weight, float ce_weight, float eps, const float* const* commit, int n_commit,
                             int levels, float commit_weight, const double* const* proto, int n_proto, float proto_weight,
                             float* const* g_inter, float* const* g_sets, float* const* g_ce, float* out, void* stream) {
    const int nt = n_sup + n_cps;
    if (n_sup < 0 || n_cps < 0 || nt < 1 || nt > 4 || c < 1 || c > 8) return bad("cps_loss_combine: 1..4 terms, 1..8 classes");
    if (!inter || !sets || !b || !g_inter || !g_sets || !out) return bad("cps_loss_combine: null pointer");
    if (n_commit < 0 || n_commit > 4 || n_proto < 0 || n_proto > 4 || levels < 0 || (n_commit && !commit) || (n_proto && !proto))
        return bad("cps_loss_combine: at most 4 commitment vectors / prototype scalars");
    vqseg::CombineArgs a{};
    for (int i = 0; i < nt; ++i) {
        if (!inter[i] || !sets[i] || !g_inter[i] || !g_sets[i] || b[i] <= 0) return bad("cps_loss_combine: bad term");
        a.inter[i] = inter[i]; a.sets[i] = sets[i]; a.g_inter[i] = g_inter[i]; a.g_sets[i] = g_sets[i]; a.b[i] = b[i];
        a.ce[i] = ce ? ce[i] : nullptr;
        a.g_ce[i] = (ce && ce[i] && g_ce) ? g_ce[i] : nullptr;
        if (a.ce[i] && !a.g_ce[i]) return bad("cps_loss_combine: a cross-entropy term needs its gradient buffer");
    }
    for (int k = 0; k < n_commit; ++k) {
        if (!commit[k]) return bad("cps_loss_combine: null commitment vector");
        a.commit[k] = commit[k];
    }
    for (int k = 0; k < n_proto; ++k) {
        if (!proto[k]) return bad("cps_loss_combine: null prototype loss");
        a.proto[k] = proto[k];
    }
    a.n_sup = n_sup; a.n_cps = n_cps; a.c = c; a.cps_weight = cps_weight; a.ce_weight = ce_weight; a.eps = eps;
    a.n_commit = n_commit; a.levels = levels; a.commit_weight = commit_weight; a.n_proto = n_proto; a.proto_weight = proto_weight;
    a.out = out;
    hipError_t e = vqseg::launch_loss_combine(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "loss_combine_kernel");
}

size_t vqseg_order_stats_workspace_bytes(void) { return vqseg::order_stats_workspace_bytes(); }

int vqseg_order_stats_f(const float* x, int64_t n, int64_t k, void* workspace, size_t workspace_bytes, float* out2, void* stream) {
    if (!x || !workspace || !out2) return bad("order_stats: null pointer");
    if (n <= 0 || n >= (int64_t(1) << 32) || k < 0 || k >= n) return bad("order_stats: need 0 <= k < n < 2^32");
    if (workspace_bytes < vqseg::order_stats_workspace_bytes()) return bad("order_stats: workspace too small");
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return bad("order_stats: x must be 16-byte aligned");
    hipError_t e = vqseg::launch_order_stats(x, n, k, workspace, out2, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hipfail(e, "kth_hist_kernel");
}

}  // extern "C"
