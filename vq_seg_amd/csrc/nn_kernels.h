// Internal declarations of the HBM-bound NN kernels (nn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vqseg {

int nn_set_option(const char* key, int value);              // previous value, or -1 (unknown key)
hipError_t launch_bn_finalize(float* partial, long n_slots, int rows_per_slot, long M, int C, const float* gamma,
                              const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, long long* num_batches_tracked, int* sync,
                              hipStream_t st);
hipError_t launch_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                                 float eps, float* scale, float* shift, float* save_mean, float* save_invstd, hipStream_t st);
hipError_t launch_bn_apply(int bf16, const void* y, const void* res, const float* scale, const float* shift, long M, int C,
                           int relu, void* out, hipStream_t st, unsigned char* bits = nullptr);
long bn_bwd_blocks(long M);
hipError_t launch_bn_backward(int bf16, const void* g_out, const void* out, const void* y, const float* mean,
                              const float* invstd, const float* gamma, const float* fwd_scale, const float* fwd_shift, long M,
                              int C, int relu, int training, int accumulate, float* partial, float* coef, float* dgamma, float* dbeta, void* g_y,
                              void* g_res, int* sync, hipStream_t st, const unsigned char* bits = nullptr);
hipError_t launch_maxpool(int bf16, int backward, const void* x, const void* g, int N, int H, int W, int C, void* out,
                          unsigned char* idx, hipStream_t st);
hipError_t launch_bilinear(int bf16, int backward, const void* src, int N, int H, int W, int C, int Ho, int Wo, int align,
                           void* dst, hipStream_t st);
hipError_t launch_head_fwd(int bf16, const void* x, const float* w, long M, int Cin, int Cout, float* y, hipStream_t st);
long head_bwd_blocks(long M);
hipError_t launch_head_bwd(int bf16, const void* x, const float* w, const float* g, long M, int Cin, int Cout, void* gx,
                           float* gw, float* partial, const void* gx_add, hipStream_t st);
hipError_t launch_reduce_partials(const float* partial, long n_blocks, long n, float* out, int accumulate, hipStream_t st);
hipError_t launch_im2col_stem(int out_bf16, const float* x, int N, int H, int W, int Cin, int KH, int KW, int stride, int pad,
                              int reflect, int Ho, int Wo, int Kp, void* out, hipStream_t st);
hipError_t launch_s3_split(const float* x, long rows, int C, void* y, hipStream_t st);
hipError_t launch_s3_merge(const void* x, long rows, int C, float* y, hipStream_t st);
hipError_t launch_s3_maxpool(const void* x, int N, int H, int W, int C, void* y, hipStream_t st);
hipError_t launch_s3_bilinear(const void* x, int N, int H, int W, int C, int Ho, int Wo, int align, void* y, hipStream_t st);
hipError_t launch_reflect_fold(int bf16, const void* gp, int N, int H, int W, int C, void* gx, hipStream_t st);
hipError_t launch_reflect_ring_fold(int bf16, const void* ring, int N, int H, int W, int C, void* gx, hipStream_t st);
hipError_t launch_cast(int to_bf16, const void* x, long n, void* y, hipStream_t st);

}  // namespace vqseg
