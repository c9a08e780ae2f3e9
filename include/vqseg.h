/*
 * vqseg.h -- C ABI of libvqseg_hip.so: the MI355X (gfx950) hot path of the VQ-UNet
 * segmentation trainer (reference: chaeyeongyun/VQ_SEG, Python/PyTorch only).
 *
 * The reference has no FFI layer; its boundary for this path is the Python nn.Module
 * surface.  These entry points are what that surface binds: each one cites
 * the reference function (file:line, relative to the reference root) whose arithmetic
 * it replaces.  Plain pointers and sizes only -- no torch types, no C++ exceptions, no
 * global mutable state except the thread-local last-error string.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *     enqueued asynchronously on it, nothing synchronises, nothing allocates: the caller
 *     supplies a workspace of at least the size the matching *_workspace_bytes() returns;
 *   - return value: 0 = success, otherwise a negative VQSEG_E* code or a positive
 *     hipError_t; vqseg_last_error() returns a message for the calling thread;
 *   - activations are "rows": N = B*H*W pixel rows of C contiguous channels (NHWC /
 *     torch channels_last), which is the (B, HW, C) frame vq_img.py:232 rearranges into.
 */
#ifndef VQSEG_H
#define VQSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQSEG_ABI_VERSION 1

#define VQSEG_EINVAL  (-1)   /* bad argument (shape, alignment, null pointer)          */
#define VQSEG_ENOSPC  (-2)   /* workspace too small                                    */
#define VQSEG_ENODEV  (-3)   /* no gfx950 device / kernel image not loadable           */

int         vqseg_abi_version(void);
const char* vqseg_last_error(void);
/* Name of the dominant kernel symbol of an entry point (for rocprof matching). */
const char* vqseg_kernel_name(const char* entry_point);

/* Measurement aid (bench.py roofline leg): while enabled, every launch of the distance+argmin
 * kernel is bracketed by a hipEvent pair on its own stream.  vqseg_profile_collect waits for the
 * recorded launches, writes per-launch (n_rows, channels, n_codes, milliseconds) to HOST arrays,
 * disables recording and returns the number of records (>= 0).  Not thread safe. */
int vqseg_profile_begin(int capacity);
int vqseg_profile_collect(int max_records, int64_t* n_rows_host, int* channels_host,
                          int* n_codes_host, float* ms_host);

/* ---------------------------------------------------------------------------------- *
 * Vector quantiser forward.
 * Replaces EuclideanCodebook.forward (vector_quantizer/vq_img.py:160-177: cdist ->
 * argmin -> one_hot -> matmul -> bincount) and the straight-through / commitment part
 * of VectorQuantizer.forward (vq_img.py:228-244) in one call.
 *
 *   x         [N, C] f32   pixel rows (C % 4 == 0, 16-byte aligned)
 *   codebook  [K, C] f32   nn.Embedding weight (vq_img.py:152)
 *   quant     [N, C] f32   out: eval  -> codebook[idx]            (vq_img.py:170)
 *                               train -> x + (codebook[idx] - x)  (vq_img.py:236)
 *   idx       [N]    i64   out: argmin_k sqrt(max(|x|^2 + |e_k|^2 - 2 x.e_k, 0)), first
 *                               minimum wins (vq_img.py:167-168)
 *   loss      [1]    f32   out: training ? commitment_weight * mean((quant - x)^2) : 0
 *                               (vq_img.py:234-240)
 *   dead_pct  [1]    f32   out: 100 * (#codes never selected) / K  (vq_img.py:173-175)
 *   dmin      [N]    f32   optional out (may be NULL): the winning distance, for tests
 * ---------------------------------------------------------------------------------- */
size_t vqseg_vq_workspace_bytes(int64_t n_rows, int channels, int n_codes);
int vqseg_vq_forward_f32(const float* x, const float* codebook, const void* prepared,
                         int64_t n_rows, int channels, int n_codes, int training,
                         float commitment_weight, float* quant, int64_t* idx, float* loss,
                         float* dead_pct, float* dmin, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Assignment only (no gather): used by k-means and by tests. */
int vqseg_vq_assign_f32(const float* x, const float* codebook, const void* prepared,
                        int64_t n_rows, int channels, int n_codes, int64_t* idx, float* dmin,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Prepared codebook: the kernel-side image of an nn.Embedding weight (4-channel
 * interleaved, code-padded copy + |e_k|^2).  The reference's codebook never changes after
 * its k-means init (no gradient, no EMA -- vq_img.py:236-239), so callers prepare once
 * and pass the blob to every forward; `prepared == NULL` makes forward/assign prepare
 * into the workspace on every call instead. */
size_t vqseg_vq_prepared_bytes(int channels, int n_codes);
int vqseg_vq_prepare_f32(const float* codebook, int channels, int n_codes, void* prepared,
                         size_t prepared_bytes, void* stream);

/* ---------------------------------------------------------------------------------- *
 * Vector quantiser backward (analytic; autograd of vq_img.py:236-240):
 *   grad_x = grad_quant + grad_loss[0] * commitment_weight * 2 (x - quant) / (N*C)
 * `quant` is the training-mode forward output.  grad_loss may be NULL (treated as 0).
 * ---------------------------------------------------------------------------------- */
int vqseg_vq_backward_f32(const float* grad_quant, const float* grad_loss, const float* x,
                          const float* quant, int64_t n_rows, int channels,
                          float commitment_weight, float* grad_x, void* stream);

/* ---------------------------------------------------------------------------------- *
 * k-means codebook initialisation, Lloyd iterations GIVEN the initial means.
 * Replaces kmeans() (vq_img.py:29-63, euclidean branch) after its RNG draw (:10-17,
 * done by the host with torch.randperm):  per iteration  assign (argmax of -cdist ==
 * argmin of cdist) -> per-cluster counts -> per-cluster sums in ROW ORDER (the order
 * scatter_add_ uses on the CPU) -> divide -> clusters with no member keep their mean.
 *
 *   samples [N, C] f32, means [K, C] f32 in/out, bins [K] i64 out (last iteration).
 *
 * Data-parallel use: vqseg_kmeans_accumulate_f32 produces this rank's sums/counts so
 * the host can all-reduce them (RCCL) before vqseg_kmeans_finalize_f32.
 * ---------------------------------------------------------------------------------- */
size_t vqseg_kmeans_workspace_bytes(int64_t n_rows, int channels, int n_codes);
int vqseg_kmeans_f32(const float* samples, float* means, int64_t* bins, int64_t n_rows,
                     int channels, int n_codes, int iters, void* workspace,
                     size_t workspace_bytes, void* stream);
int vqseg_kmeans_accumulate_f32(const float* samples, const float* means, int64_t n_rows,
                                int channels, int n_codes, float* sums, int64_t* counts,
                                void* workspace, size_t workspace_bytes, void* stream);
int vqseg_kmeans_finalize_f32(const float* sums, const int64_t* counts, float* means,
                              int channels, int n_codes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQSEG_H */
