/*
 * vqseg.h -- C ABI of libvqseg_hip.so: the MI355X (gfx950) hot path of the VQ-UNet
 * segmentation trainer (reference: chaeyeongyun/VQ_SEG, Python/PyTorch only).
 *
 * The reference has no FFI layer; its boundary for this path is the Python nn.Module
 * surface.  These entry points are what that surface binds: each one cites
 * the reference function (file:line, relative to the reference root) whose arithmetic
 * it replaces.  Plain pointers and sizes only -- no torch types, no C++ exceptions.  Process-global
 * state: the thread-local last-error string, the dispatch options of vqseg_set_option and the two
 * measurement recorders (vqseg_profile_*, vqseg_conv_profile_*) -- none of them thread safe; launches
 * themselves keep no state between calls.
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

/* Dispatch tunables (tests use them to reach every kernel with small shapes).  Keys:
 *   "conv3x3_patch_min_workgroups"  minimum grid of the patch-reuse 3x3 kernel before the generic implicit-GEMM
 *                                   kernel is preferred, and of its 256-channel tile (default 128, r4; r3: 256 = one workgroup per CU)
 *   "conv3x3_patch_unroll", "conv3x3_patch_wide_tile"   variants of that kernel (tap loop unrolled; 256-channel tile: default ON from r4 -- half
 *                                   the input re-reads; slower on a layer alone, faster in the two-stream step)
 *   "vq_max_tiles_per_wave"         cap (8, 4, 2, 1) on the 32-code accumulator tiles a wave of the VQ distance kernel
 *                                   holds (default 8; 4 measured equal within 2 % on every benchmark shape)
 *   "vq_bf16_filter"                1 (default): bf16 rows of layers with K % 256 == 0 and C % 32 == 0 take the bf16-MFMA candidate
 *                                   filter + exact re-score (same indices and distances as the exact kernel); 0: exact kernel on every row.
 *                                   "vq_filter_force_all" = 1 (tests): the filter decides nothing, every row is re-scored;
 *                                   "vq_filter_launches": returns the number of launches that took the filter so far, sets the counter
 *   "conv_xcd_pair"                 implicit-GEMM layers with 2..value Cout chunks launch 1-D so that the chunks of an
 *                                   M tile run on the same XCD and share the input rows through its L2 (default 8, r4; r3: 4; 0: off)
 *   "conv3x3_patch_chunk_stage"     1 (default): 3x3 layers with 32-channel K chunks and <= 64 outputs (or 32 inputs) load all nine
 *                                   taps' weights with the patch -- one wait and barrier per chunk instead of per tap; 0: tap ring
 *   "conv3x3_patch_tile512"         512-pixel tiles of the patch kernel (32-channel chunks) for 32 / 64 output channels: 2 (default) on, 0
 *                                   off (192 -> 32 at 256^2 is 28 % faster with them; for 128 outputs they measured 10-17 % slower);
 *                                   "conv3x3_patch_tile512_min_workgroups" their minimum grid (default 512);
 *                                   "conv3x3_patch_tile512_launches" returns the number of launches that took such a tile so far and
 *                                   sets the counter to the value passed (tests)
 *   "conv3x3_patch_xcd_pair"        1: the Cout chunks of a pixel tile are dispatched onto the same XCD (default 0:
 *                                   measured equal; the kernel does not wait on HBM for its patches)
 *   "conv_short_k_small_tile", "conv_short_k_single_buffer"   K loops of up to that many 64-channel stages take the
 *                                   128x128 tile at 4 waves/SIMD, double- resp. single-buffered (defaults 1 and 8)
 *   "conv_wgrad3x3_stride2"         1 (default): stride-2 3x3 weight gradients (Cout % 128 == 0, Wo % 16 == 0) on the nine-tap LDS-DMA
 *                                   kernel (r4); 0: the per-tap kernel
 *   "conv_wgrad1x1_narrow"          1 (default): 64-output-channel 1x1 weight gradients -- the stem's 160-column patch matrix, 64 -> 64
 *                                   -- on the LDS-DMA kernel (r4); 0: the per-tap kernel
 *   "conv_dgrad_s2_merge"           1 (default): vqseg_conv2d_dgrad_s2_fold_f available (r4); 0: it returns VQSEG_EINVAL (callers take the
 *                                   four-launch path with the padded grid)
 *   "conv_wgrad3x3_fill"            1 (default): nine-tap weight gradients split into one FULL round of resident workgroups (r4);
 *                                   0: r3's split
 *   "conv3x3_patch_wide_tile_s3"    1 (default): the split-3 3x3 launches take the 256-channel tile too (r4); "conv_wgrad_round_pct": workgroups the
 *                                   LDS-DMA weight gradients aim at, in percent of one resident round (default 100; 10..400)
 *   "stem_fused"                    1 (default): vqseg_stem7_conv_f available (r4); 0: it returns VQSEG_EINVAL (patch-matrix path)
 *   "conv_wgrad_xcd"                1 (default): a pixel slab's (ci, co) tiles of the LDS-DMA weight-gradient kernels on one XCD (r4); 0: 3-D grid
 *   "im2col_strip"                  1 (default): the stem's patch matrix from LDS-staged strips (r4, bit-identical); 0: the gather kernel
 *   "bn_bwd_premask"                1 (default): vqseg_bn_backward_f with `out` and a g_res output (a residual layer): the reduce pass writes
 *                                   g_res, the apply pass reads it instead of (g_out, out) (r4, bit-identical); 0: r3's two passes over both
 *   "bilinear_up2"                  1 (default): exact-2x resizes on their own kernels (bit-identical to the generic ones), the backward with four
 *                                   input rows per thread (r4); 2: one row per thread (r3); 0: the generic kernels
 *   "vq_fine_split"                 1 (default, r4): the distance kernel takes 4 code tiles per wave instead of 8 when that brings a launch to
 *                                   >= 4096 workgroups (a shorter last round: K = 512 on 172 k rows 0.76 -> 0.81 of the fp32 MFMA peak); 0: r3's choice
 *   "nn_grid_cap"                   8192 (default): most workgroups a grid-stride elementwise kernel (BatchNorm passes, resizes, pools) is launched with
 * The environment variable VQSEG_OPTS="key=value,..." applies options when the Python binding loads the library.
 * Returns the previous value, or VQSEG_EINVAL for an unknown key / negative value.  Not thread safe. */
int vqseg_set_option(const char* key, int value);

/* Measurement aid (bench.py roofline leg): while enabled, every launch of the distance+argmin
 * kernel is bracketed by a hipEvent pair on its own stream.  vqseg_profile_collect waits for the
 * recorded launches, writes per-launch (n_rows, channels, n_codes, milliseconds) to HOST arrays,
 * disables recording and returns the number of records (>= 0).  Not thread safe. */
int vqseg_profile_begin(int capacity);
int vqseg_profile_collect(int max_records, int64_t* n_rows_host, int* channels_host,
                          int* n_codes_host, float* ms_host,
                          int* kind_host /* nullable; 0: exact fp32-MFMA kernel on f32 rows, 1: on bf16 rows, 2: bf16 candidate filter + exact re-score */);

/* The same for the convolution kernels (forward, data gradient, fused-epilogue and split-3 launches of vqseg_conv2d_*): per launch
 * the ALGORITHMIC flops 2 * KH * KW * Cin * Cout * output pixels (logical channels for split-3; forward-layer pixels for the
 * data gradient of a strided layer), kind = KH * 100 + {0 bf16, 1 precise, 2 split-3; 50 / 51: the weight-gradient kernel of
 * vqseg_conv2d_wgrad_f in bf16 / precise mode, without its slab sum}, the in-stream milliseconds and the shape. */
int vqseg_conv_profile_begin(int capacity);
int vqseg_conv_profile_collect(int max_records, double* flops_host, int* kind_host, float* ms_host,
                               int* shape_host /* optional [4] per record: output pixels / 1024, Cin, Cout, stride * 10 + up */);

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

/* The same forward for up to 4 independent layers ("levels") in ONE distance + argmin launch: the three VQ layers of a
 * VQ-UNet forward (net.py:1183-1191 calls them one after the other; they are independent).  Arrays of n_levels entries;
 * every level with its own workspace (vqseg_vq_workspace_bytes of ITS shape); one row type per call (bf16 != 0: bf16 rows /
 * quantised rows).  Results are bit-identical to n_levels single calls. */
int vqseg_vq_forward_group(int n_levels, int bf16, const void* const* x, const float* const* codebook,
                           const void* const* prepared, const int64_t* n_rows, const int* channels, const int* n_codes,
                           int training, const float* commitment_weight, void* const* quant, int64_t* const* idx,
                           float* const* loss, float* const* dead_pct, void* const* workspace,
                           const size_t* workspace_bytes, void* stream);

/* Diagnostics of the bf16 candidate filter (option "vq_bf16_filter"): byte offset, inside the workspace of a bf16 call of this shape,
 * of the 64 int32 counters (one per sub-list; their sum = the (row, code) candidate pairs the filter handed to the exact re-score;
 * valid after the call's stream work has finished; a 65th int is the overflow flag: the exact kernel served the level); 0 if the shape does not take the filter
 * (needs n_codes % 256 == 0 and channels % 32 == 0). */
size_t vqseg_vq_filter_counter_offset(int64_t n_rows, int channels, int n_codes);

/* Assignment only (no gather): used by k-means and by tests. */
int vqseg_vq_assign_f32(const float* x, const float* codebook, const void* prepared,
                        int64_t n_rows, int channels, int n_codes, int64_t* idx, float* dmin,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same for bf16 rows (autocast activations; every bf16 value is an exact float, the arithmetic is the fp32 one). */
int vqseg_vq_assign_bf16(const void* x, const float* codebook, const void* prepared,
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

/* The same layer on bf16 activations (the trainer's autocast mode): x and quant are bf16 rows, every bf16 value is an
 * exact float and all arithmetic (distances, argmin, straight-through value, commitment) is the fp32 arithmetic of the
 * f32 entry points, so the indices are identical to those of the up-cast rows; quant is rounded to bf16 on store.
 * Backward re-reads e = codebook[idx] in fp32 (the bf16 quant would cost the commitment gradient its accuracy):
 *   grad_x = grad_quant + (2 w grad_loss / (n c)) (x - e). */
int vqseg_vq_forward_bf16(const void* x, const float* codebook, const void* prepared, int64_t n_rows, int channels,
                          int n_codes, int training, float commitment_weight, void* quant, int64_t* idx, float* loss,
                          float* dead_pct, float* dmin, void* workspace, size_t workspace_bytes, void* stream);
int vqseg_vq_backward_bf16(const void* grad_quant, const float* grad_loss, const void* x, const int64_t* idx,
                           const float* codebook, int64_t n_rows, int channels, float commitment_weight,
                           void* grad_x, void* stream);

/* ---------------------------------------------------------------------------------- *
 * k-means codebook initialisation, Lloyd iterations GIVEN the initial means.
 * Replaces kmeans() (vq_img.py:29-63, euclidean branch) after its RNG draw (:10-17,
 * done by the host with torch.randperm):  per iteration  assign (argmax of -cdist ==
 * argmin of cdist) -> per-cluster counts -> per-cluster sums -> divide -> clusters with no
 * member keep their mean.  The sums are deterministic but not in scatter_add_'s single running order:
 * member lists are in row order, cut into segments of 128 members, four interleaved partial sums per
 * segment, a cluster's segments folded as four interleaved chains in fixed order (fp32; agreement with the CPU order is at rounding level, asserted
 * at 1e-5 in tests/test_vq_gpu.py).
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


/* ---------------------------------------------------------------------------------- *
 * EXTENSION -- EMA codebook update (opt-in; BASELINE.json north_star names it, the reference does not have it:
 * vq_img.py:72,83,140,151 accept and store `decay` / `eps` but the codebook is never written after the k-means init,
 * SURVEY 0.1 / q6).  The rule is the published one of the module the reference descends from (vector-quantize-pytorch,
 * EuclideanCodebook.forward: ema_inplace + laplace_smoothing); there is no reference output to pin it against, so its
 * tests compare with a tensor-op restatement of that rule ("parity unpinned").
 *   vqseg_vq_code_sums:      sums[k][c] = sum of the rows assigned to code k, counts[k] = how many -- from the idx a
 *                            forward returned (no second distance pass); rows f32 or bf16; deterministic (the k-means
 *                            member-list reduction).  workspace: vqseg_kmeans_workspace_bytes(n_rows, channels, n_codes).
 *                            Data-parallel use: all-reduce sums and counts (RCCL) before the update.
 *   vqseg_vq_ema_update_f32: cluster_size <- d cluster_size + (1-d) counts;  embed_avg <- d embed_avg + (1-d) sums;
 *                            codebook <- embed_avg / ((cluster_size + eps) / (S + K eps) * S),  S = sum cluster_size.
 *                            scratch: 1 float on the device.
 * ---------------------------------------------------------------------------------- */
int vqseg_vq_code_sums(int bf16, const void* x, const int64_t* idx, int64_t n_rows, int channels, int n_codes,
                       float* sums, int64_t* counts, void* workspace, size_t workspace_bytes, void* stream);
int vqseg_vq_ema_update_f32(float* cluster_size, float* embed_avg, float* codebook, const float* sums,
                            const int64_t* counts, int channels, int n_codes, float decay, float eps, float* scratch,
                            void* stream);

/* ================================================================================== *
 * Encoder / decoder blocks.  Tensors are NHWC rows; `precise` = 1: activations fp32, bf16x3
 * split MFMA (parity mode); 0: activations bf16 (fast mode).  `bf16` flags name the
 * element type of the activation tensors (0 = f32, 1 = bf16).
 * ================================================================================== */

/* nn.Conv2d weights [Cout][Cin][KH][KW] f32 -> MFMA-side image [Cout][KH][KW][Cin] bf16 (hi, and lo
 * when `lo` != NULL).  transpose_flip = 1 builds the data-gradient image [Cin][KH][KW][Cout] with the
 * taps flipped.  The contracted channel count is zero-padded to a multiple of 32; each array holds
 * vqseg_conv_packed_elems() 16-bit elements. */
size_t vqseg_conv_packed_elems(int cout, int cin, int kh, int kw, int transpose_flip);
int vqseg_conv_pack_weights_f32(const float* w, int cout, int cin, int kh, int kw, int transpose_flip,
                                void* hi, void* lo, void* stream);

/* Implicit-GEMM convolution, no bias.  Replaces nn.Conv2d in conv_bn_relu
 * (models/networks/unet/decoder.py:7-10) and in the ResNet blocks (models/encoders/resnet.py:117-190):
 *   y[n,oh,ow,co] = sum_{kh,kw,ci} in[n, (oh*stride - pad + kh)/up, (ow*stride - pad + kw)/up, ci] * w[co,kh,kw,ci]
 * with zero padding, or reflect padding (nn.Conv2d(padding_mode='reflect'), resnet.py:134-148) when
 * `reflect`; `up` > 1 reads the input on an up-sampled grid (only exact multiples exist): the data
 * gradient of a stride-`up` convolution.  Channels [0,c1) come from x, [c1,cin) from x2 -- the decoder's
 * torch.cat((upsampled, skip), 1) (decoder.py:35-37) fused into the loader (c1 == cin: no concat).
 * stat_partial (nullable): per-wave (mean, M2) BatchNorm partials, vqseg_conv_stat_slots() x 2 x cout floats. */
int64_t vqseg_conv_stat_slots(int64_t m_rows, int cout);
int vqseg_conv2d_f(const void* x, const void* x2, int c1, const void* w_hi, const void* w_lo, void* y,
                   float* stat_partial, int n, int h, int w, int cin, int cout, int kh, int kw, int stride,
                   int pad, int reflect, int up, int ho, int wo, int precise, void* stream);

/* Weight gradient of the same convolution: gw[co][ci][kh][kw] (nn.Conv2d layout, f32) =
 * sum_m gy[m][co] * in_tap[m][ci].  im2col = 1: x is a [M][cin] patch matrix of a kh x kw x cin_out
 * convolution (the 7x7 stem) and gw gets [cout][cin_out][kh][kw]. */
/* The same convolution with a fused per-channel affine epilogue: y = relu?( conv * scale[c] + shift[c] (+ res) ).
 * This is conv -> eval-mode nn.BatchNorm2d (scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean *
 * scale; vqseg_bn_finalize_f with training = 0 yields both) -> [+ residual] -> [ReLU] in one pass, for the
 * no-grad pseudo-label forwards (train_vqreptunet1x1v2.py eval passes). */
int vqseg_conv2d_affine_f(const void* x, const void* x2, int c1, const void* w_hi, const void* w_lo,
                          const float* scale, const float* shift, const void* res, int relu, void* y,
                          int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int reflect,
                          int ho, int wo, int precise, void* stream);
/* bf16, stride 1, zero padding, no concat: y = conv * scale[c] + shift[c] + res .* bits, with bits as written by
 * vqseg_bn_apply_bits_f over res's flat [n*ho*wo*cout] index (cout % 8 == 0).  This is the data gradient of a residual block's
 * first convolution meeting the block's shortcut gradient g_out .* (out > 0) (resnet.py:106-108) without that product ever
 * being stored. */
int vqseg_conv2d_affine_bits_f(const void* x, const void* w_hi, const float* scale, const float* shift, const void* res,
                               const unsigned char* res_bits, void* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                               int pad, int ho, int wo, void* stream);

/* `accumulate` (here and in vqseg_bn_backward_f): the parameter gradient is ADDED to gw / dgamma / dbeta (the
 * optimiser's gradient buffer) instead of overwriting it -- autograd's own accumulate kernel is then not needed. */
size_t vqseg_conv2d_wgrad_workspace_bytes(int n, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw);
int vqseg_conv2d_wgrad_f(const void* gy, const void* x, const void* x2, int c1, int n, int h, int w, int cin,
                         int ho, int wo, int cout, int kh, int kw, int stride, int pad, int reflect, int precise,
                         int cin_out, int im2col, int accumulate, void* workspace, size_t workspace_bytes, float* gw,
                         void* stream);
/* The same weight gradient summed over TWO uses of the layer in one launch: (gy, x, x2) with n images and (gy_b, x_b, x2_b) with
 * n_b images of the same geometry form a virtual batch of n + n_b images.  In a cross-pseudo-supervision step every weight is
 * used by two training forwards per network (labelled and unlabelled batch: train_vqreptunet1x1v2.py:153-156, one backward
 * :199): the caller queues the first use's (gy, x) and issues one launch over both -- twice the contraction length per
 * workgroup, half the slab sums.  n_b == 0: identical to vqseg_conv2d_wgrad_f.  Workspace: vqseg_conv2d_wgrad_workspace_bytes
 * with n + n_b images. */
int vqseg_conv2d_wgrad2_f(const void* gy, const void* x, const void* x2, int n, const void* gy_b, const void* x_b, const void* x2_b,
                          int n_b, int c1, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad,
                          int reflect, int precise, int cin_out, int im2col, int accumulate, void* workspace, size_t workspace_bytes,
                          float* gw, void* stream);

/* nn.BatchNorm2d (+ fused residual add and ReLU).  Training: batch statistics merged from the conv
 * epilogue partials (Welford/Chan, double, fixed order), running stats updated like nn.BatchNorm2d
 * (biased variance to normalise, unbiased into running_var).  Eval: running statistics.
 *   finalize -> scale[c] = gamma*invstd, shift[c] = beta - mean*scale, save_mean, save_invstd
 *               (`partial` is CONSUMED: the two-level merge folds group results back into it in place)
 *   apply    -> out = relu?( y*scale + shift (+ res) )
 *   backward -> gz = g_out * (out > 0); dgamma, dbeta; g_y (train: with the batch-statistics terms);
 *               g_res = gz when a residual branch exists.  Without a residual `out` may be NULL: the mask is then
 *               recomputed as fma(y, fwd_scale, fwd_shift) > 0, the forward's own expression (one tensor read less). */
/* `sync` (nullable; finalize and backward): vqseg_bn_sync_ints(c) ints that hold ZEROS between launches (zero them once; the
 * library leaves them zero) and that no two concurrent launches share -- e.g. one buffer per BatchNorm module and direction.
 * With it the statistics merge + finalize (forward) / the reduction + finalize (backward) run as ONE launch each: the last
 * workgroup of a 64-channel group does the final fold, in a fixed order (deterministic; r3: ~700 launches per training step
 * less).  NULL: the two-launch form. */
int vqseg_bn_sync_ints(int c);
int vqseg_bn_finalize_f(float* partial, int64_t m_rows, int c, const float* gamma, const float* beta,
                        float* run_mean, float* run_var, float momentum, float eps, int training,
                        float* scale, float* shift, float* save_mean, float* save_invstd,
                        int64_t* num_batches_tracked /* nullable: += 1 in training mode */, int* sync, void* stream);
int vqseg_bn_apply_f(int bf16, const void* y, const void* res, const float* scale, const float* shift,
                     int64_t m_rows, int c, int relu, void* out, void* stream);
size_t vqseg_bn_backward_workspace_floats(int64_t m_rows, int c);
int vqseg_bn_backward_f(int bf16, const void* g_out, const void* out, const void* y, const float* mean,
                        const float* invstd, const float* gamma, const float* fwd_scale, const float* fwd_shift,
                        int64_t m_rows, int c, int relu, int training, int accumulate,
                        float* workspace, float* dgamma, float* dbeta, void* g_y, void* g_res, int* sync, void* stream);
/* r4, residual layers in bf16 (out = relu(y*scale + shift + res), resnet.py:106-108 / 67-69; c % 8 == 0): the apply pass also
 * writes the ReLU mask as a bit field, bits[i / 8] bit (i % 8) = (out[i] > 0) over the flat [m_rows * c] index (m_rows * c / 8
 * bytes), and the backward reads that instead of `out` (1/16 of its bytes).  g_res (nullable) = g_out * mask, the residual
 * branch's gradient; NULL when its consumer applies the bits to g_out itself (vqseg_conv2d_affine_bits_f: the block's first
 * convolution adds the masked gradient in its data-gradient epilogue), which saves writing and re-reading that tensor.
 * Results are bit-identical to vqseg_bn_apply_f / vqseg_bn_backward_f with `out`. */
int vqseg_bn_apply_bits_f(const void* y, const void* res, const float* scale, const float* shift, int64_t m_rows, int c,
                          void* out, unsigned char* bits, void* stream);
int vqseg_bn_backward_bits_f(const void* g_out, const unsigned char* bits, const void* y, const float* mean,
                             const float* invstd, const float* gamma, int64_t m_rows, int c, int training, int accumulate,
                             float* workspace, float* dgamma, float* dbeta, void* g_y, void* g_res, int* sync, void* stream);

/* nn.MaxPool2d(3, 2, 1) (resnet.py:167) forward / backward (first maximum in scan order).
 * idx (nullable, uint8 [n, ho, wo, c]): forward writes the window position (kh*3+kw) of each maximum; backward
 * reads it instead of recomputing the arg-max from x (x may then be NULL). */
int vqseg_maxpool3x3s2_f(int bf16, int backward, const void* x, const void* g, int n, int h, int w, int c,
                         void* out, unsigned char* idx, void* stream);

/* Bilinear resize: F.interpolate(mode='bilinear') (decoder.py:35, align_corners = 0) and
 * nn.UpsamplingBilinear2d (modified_vqunet/net.py:1172, align_corners = 1).
 * forward: src [n,h,w,c] -> dst [n,ho,wo,c];  backward: src = grad [n,ho,wo,c] -> dst [n,h,w,c]. */
int vqseg_bilinear_f(int bf16, int backward, const void* src, int n, int h, int w, int c, int ho, int wo,
                     int align_corners, void* dst, void* stream);

/* 1x1 segmentation head, nn.Conv2d(32, num_classes, 1, bias=False) (net.py:1169): logits f32.  Cin % 8 == 0, Cin <= 64, Cout <= 4.
 * Row type `bf16`: 0 f32, 1 bf16, 2 (forward only) split-3 rows [2 * Cin] (see "Split-3" below). */
int vqseg_head1x1_forward_f(int bf16, const void* x, const float* w, int64_t m_rows, int cin, int cout,
                            float* y, void* stream);
size_t vqseg_head1x1_backward_workspace_floats(int64_t m_rows, int cin, int cout);
int vqseg_head1x1_backward_f(int bf16, const void* x, const float* w, const float* g, int64_t m_rows, int cin,
                             int cout, void* gx, float* gw, float* workspace, void* stream);
/* Fan-in forms (r4): the tensor whose gradient these produce has a second consumer whose gradient `gx_add` (same shape and type;
 * NULL: none) is added on the way out -- rounded as autograd's own add of the two tensors would round (each to the activation
 * type first).  head: the decoder output feeds the head AND the prototype loss (net.py:1193-1203); max-pool: the stem output feeds
 * the pool AND the last decoder block (resnet.py:164-171, decoder.py:35-37). */
int vqseg_head1x1_backward_add_f(int bf16, const void* x, const float* w, const float* g, int64_t m_rows, int cin,
                                 int cout, void* gx, float* gw, float* workspace, const void* gx_add, void* stream);
int vqseg_maxpool3x3s2_backward_add_f(int bf16, const void* g, const unsigned char* idx, const void* gx_add, int n, int h, int w,
                                      int c, void* gx, void* stream);

/* Stem support: patch matrix of the 7x7/2 convolution (resnet.py:122-125; zero or reflect padding),
 * columns (kh, kw, ci) padded with zeros to kp; gradient fold of reflect padding 1; f32 <-> bf16 cast. */
int vqseg_im2col_f(int out_bf16, const float* x, int n, int h, int w, int cin, int kh, int kw, int stride,
                   int pad, int reflect, int ho, int wo, int kp, void* out, void* stream);
int vqseg_reflect_fold_f(int bf16, const void* gp, int n, int h, int w, int c, void* gx, void* stream);
/* r4: the stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels: resnet.py:122-125) STRAIGHT from the fp32 image x [n][h][w][3], without
 * the patch matrix: a workgroup stages the input rows its 128 output pixels read and takes its MFMA operands from them.  The contraction
 * runs over k' = kh * 24 + (kw * 3 + ci) (each kernel row's 21 taps padded to 24; 176 columns = eleven K steps): a fragment is then 8
 * consecutive words of one staged row.  Equal to the 1x1 convolution over vqseg_im2col_f's matrix up to the summation grouping (bf16
 * outputs within one unit in the last place).  w_img: [64][176] bf16, column kh * 24 + kw * 3 + ci = bf16(w[co][ci][kh][kw]), zero
 * elsewhere; s3 = 1: [64][2][176] = the hi image | the lo image (bf16 of the remainder), y = split-3 rows [n][ho][wo][128] = hi | lo.
 * Either stat_partial (raw y + BatchNorm partials: vqseg_conv_stat_slots) or scale / shift (+ relu): the fused eval epilogue (required
 * for s3).  Output width (w + 6 - 7) / 2 + 1 must be a multiple of 128, else VQSEG_EINVAL (callers keep the patch-matrix path). */
int vqseg_stem7_conv_f(int s3, const float* x, const void* w_img, void* y, float* stat_partial, const float* scale, const float* shift, int relu,
                       int n, int h, int w, int reflect, void* stream);
/* The data gradient of a 3x3 / stride 1 / REFLECT-pad-1 convolution (resnet.py:134-148: every Bottleneck conv2) without the padded
 * gradient tensor (r4): gx [n][h][w][cgx] must already hold the ZERO-padded data gradient of gy (vqseg_conv2d_f with the transposed
 * image and pad 1: the padded gradient's interior, on the fast patch kernel); this call evaluates the full correlation only on the
 * border ring of the (h + 2) x (w + 2) grid -- ring [n][2 (w + 2) + 2 h][cgx], scratch -- and folds the ring onto rows 1 / h-2 and
 * columns 1 / w-2 of gx in place.  bf16 activations; cgy % 64 == 0 (gy's channels), cgx % 8 == 0; h, w >= 4. */
int vqseg_reflect_ring_f(const void* gy, const void* t_hi, void* ring, void* gx, int n, int h, int w, int cgy, int cgx, void* stream);

/* ----------------------------------------------------------------------------------
 * "Split-3" activations: the fp32-precision NO-GRAD EVAL forward (the trainers' pseudo-label passes, outside autocast:
 * train_vqreptunet1x1v2.py:143-149) on the bf16 MFMA kernels.  A logical fp32 tensor [rows][C] is kept as [rows][2C] bf16 =
 * [hi | lo] with hi = bf16(v), lo = bf16(v - hi) (v = hi + lo to ~2^-17).  A bf16 convolution whose K loop runs over the logical
 * channels [hi | lo | hi] (hi read twice) with the weight image [w_hi | w_hi | w_lo] computes x_hi w_hi + x_lo w_hi + x_hi w_lo:
 * the same three products, fp32 accumulation, as the "precise" kernels (vqseg_conv2d_f precise = 1), on the LDS-DMA / patch-reuse
 * kernels instead.
 *   vqseg_conv2d_affine_f(..., precise = 2): x / x2 / res / y are split-3 tensors, cin / c1 / cout LOGICAL channel counts
 *       (Cin, C1 % 32 == 0, Cout % 8 == 0), w_hi = vqseg_conv_pack_weights_s3_f32's image, w_lo unused.
 *   vqseg_im2col_f(out_bf16 = 2, ...): split-3 patch rows [2 * kp] (7x7x3 stem only).
 *   split / merge: fp32 rows <-> split-3 rows;  maxpool / bilinear: the forward ops of vqseg_maxpool3x3s2_f /
 *       vqseg_bilinear_f on split-3 tensors (values hi + lo, re-split on the way out).  channels % 8 == 0.
 * ---------------------------------------------------------------------------------- */
int vqseg_conv_pack_weights_s3_f32(const float* w, int cout, int cin, int c1, int kh, int kw, void* out, void* stream);

/* Data gradient of a STRIDE-2 convolution (encoder: the 3x3 conv2 of a stage's first bottleneck and its 1x1 projection shortcut,
 * resnet.py:117-190 on torchvision's Bottleneck) without the 4x wasted work of an up-sampled ("dilated") gradient grid: the
 * output pixels split into parity classes, each a small stride-1 convolution over gy (k = 3: 2x2, 2x1, 1x2, 1x1 taps; k = 1:
 * one class, the other pixels are zero).  gx is (n, oh, ow, cin): for k = 3 the PADDED input grid (oh = h + 2; fold reflect /
 * crop zero padding afterwards), for k = 1 the input grid.  Weights: vqseg_conv_pack_weights_s2_f32's sub-images
 * (vqseg_conv_packed_s2_elems elements each for hi / lo; lo only in precise mode). */
/* The forward (vqseg_conv_pack_weights_f32, transpose_flip = 0, hi), data-gradient (transpose_flip = 1, hi) and split-3
 * (vqseg_conv_pack_weights_s3_f32) images of one k x k weight (k = 1 or 3) in ONE launch; any of the three outputs may be NULL.
 * Bit-identical to the single-image entry points. */
int vqseg_conv_pack_all_f32(const float* w, int cout, int cin, int k, int c1, void* fwd, void* tr, void* s3, void* stream);
size_t vqseg_conv_packed_s2_elems(int cout, int cin, int k);
int vqseg_conv_pack_weights_s2_f32(const float* w, int cout, int cin, int k, void* hi, void* lo, void* stream);
/* `accumulate` (k == 1 only, r4): gx already holds the gradient the tensor received from its OTHER consumer (an encoder feature feeds
 * the next stage AND the decoder / VQ layer); the data gradient is added to it in place at the pixels it touches -- no memset and no
 * separate add pass (bit-identical to adding the two tensors). */
int vqseg_conv2d_dgrad_s2_f(const void* gy, const void* w_hi, const void* w_lo, void* gx, int n, int ho, int wo, int cout, int cin,
                            int k, int oh, int ow, int precise, int accumulate, void* stream);
/* r4: the 3x3 / stride 2 / pad 1 data gradient (bf16; the first conv2 of a Bottleneck stage, reflect-padded in the VQ-UNet's encoder:
 * resnet.py:134-148; the first conv1 of a BasicBlock stage, zero-padded) written STRAIGHT into the unpadded gx [n][h][w][cin]
 * (h = 2 ho, w = 2 wo): the four parity classes in ONE launch, no padded-grid tensor, no fold / crop pass over it.  gx must hold
 * vqseg_conv2d_dgrad_s2_fold_rows(n, h, w, reflect) rows of cin elements: behind the n * h * w pixel rows, reflect padding keeps
 * the gradients of the padded top row / left column (a ring of n * (w + 1 + h) rows, added onto x row 1 / column 1 by a second
 * small launch) and one dump row.  w_hi: vqseg_conv_pack_weights_s2_f32's image (k = 3).  VQSEG_EINVAL for shapes outside the
 * merged path (cout % 64, cin % 8, cin >= 64) or with option "conv_dgrad_s2_merge" = 0: use vqseg_conv2d_dgrad_s2_f then. */
int64_t vqseg_conv2d_dgrad_s2_fold_rows(int n, int h, int w, int reflect);
int vqseg_conv2d_dgrad_s2_fold_f(const void* gy, const void* w_hi, void* gx, int n, int ho, int wo, int cout, int cin, int h, int w,
                                 int reflect, void* stream);
int vqseg_s3_split_f(const float* x, int64_t rows, int channels, void* y, void* stream);
int vqseg_s3_merge_f(const void* x, int64_t rows, int channels, float* y, void* stream);
int vqseg_s3_maxpool3x3s2_f(const void* x, int n, int h, int w, int c, void* y, void* stream);
int vqseg_s3_bilinear_f(const void* x, int n, int h, int w, int c, int ho, int wo, int align_corners, void* y, void* stream);
int vqseg_cast_f(int to_bf16, const void* x, int64_t n, void* y, void* stream);

/* Reliable prototype losses (models/modules/prototype.py:500-613 ReliablePrototypeLoss = variant 1, :778-888
 * ReliablePrototypeLossv2 = variant 2) after the host-side label resize / entropy threshold: fused forward (scalar loss,
 * double) and backward (d loss / d x in the activation type; variant 2 also d loss / d prototypes [k][c], nullable).
 *   x [m][c] decoder features (f32 or bf16), proto [k][c] L2-normalised, labels [m] i64,
 *   keep [m] u8 (variant 1: entropy <= percentile; NULL = all), conf [m] f32 (variant 2 confidence mask; NULL = 1),
 *   margin / scale / easy_margin: the module's ArcFace parameters; g_loss: upstream gradient (device scalar). */
size_t vqseg_proto_loss_workspace_bytes(int64_t m_rows, int c, int k);
int vqseg_proto_loss_forward_f(int bf16, const void* x, const float* proto, const int64_t* labels,
                               const unsigned char* keep, const float* conf, int64_t m_rows, int c, int k,
                               int variant, float scale, float margin, int easy_margin,
                               void* workspace, size_t workspace_bytes, double* loss, void* stream);
int vqseg_proto_loss_backward_f(int bf16, const void* x, const float* proto, const int64_t* labels,
                                const unsigned char* keep, const float* conf, int64_t m_rows, int c, int k,
                                int variant, float scale, float margin, int easy_margin, const float* g_loss,
                                void* gx, float* gproto, void* workspace, size_t workspace_bytes, void* stream);

/* Soft Dice sums of loss/dice_loss.py:5-37 (softmax form, 2..4 classes; ignored pixels: zero logits, class-0 target):
 *   inter[b][c] = sum_px p_c 1[t == c],  sets[b][c] = sum_px (p_c + 1[t == c]);  the scalar formula on these (B x C)
 * sums stays with the host.  logits element (b, c, px) at b*stride_b + c*stride_c + px*stride_px (NCHW or NHWC);
 * backward writes d loss / d logits in the same layout from d loss / d inter, d loss / d sets. */
size_t vqseg_dice_workspace_bytes(int b, int c, int64_t hw);
int vqseg_dice_sums_forward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px,
                              const int64_t* target, int b, int c, int64_t hw, int64_t ignore_index,
                              void* workspace, size_t workspace_bytes, float* inter, float* sets, void* stream);
int vqseg_dice_sums_backward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px,
                               const int64_t* target, int b, int c, int64_t hw, int64_t ignore_index,
                               const float* g_inter, const float* g_sets, float* g_logits, void* stream);

/* The same pass with the cross-entropy term of the v2 recipe riding along (train_vqreptunet1x1v2.py:165-187:
 * 0.5 * nn.CrossEntropyLoss(ignore_index=255) + Dice on the same logits and targets):
 *   ce[b][0] = sum over pixels with target != ignore_index of -log softmax(logits)[target],  ce[b][1] = their number;
 * the mean CE is sum_b ce[b][0] / sum_b ce[b][1] on the host.  backward adds g_ce[b][0] * (softmax - onehot) on those pixels.
 * `ce` / `g_ce` may be NULL (= the two entry points above). */
int vqseg_dice_ce_sums_forward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px,
                                 const int64_t* target, int b, int c, int64_t hw, int64_t ignore_index,
                                 void* workspace, size_t workspace_bytes, float* inter, float* sets, float* ce,
                                 void* stream);
int vqseg_dice_ce_sums_backward_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px,
                                  const int64_t* target, int b, int c, int64_t hw, int64_t ignore_index,
                                  const float* g_inter, const float* g_sets, const float* g_ce, float* g_logits,
                                  void* stream);

/* Pseudo-label statistics in one pass over the logits (layout as for the Dice sums): per pixel the arg-max class (i64),
 * the entropy -sum p log(p + 1e-10) and the top probability of softmax(logits); any output may be NULL.  Replaces
 * softmax -> argmax / log / mul / sum / max of make_regularized_pseudo_label (deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39),
 * score_mask (train_vqreptunet1x1v2.py:43-46) and the entropy map of VQRePTUnet1x1.forward (net.py:1199-1201). */
int vqseg_softmax_stats_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px, int b, int c,
                          int64_t hw, int64_t* label, float* entropy, float* top, void* stream);

/* Confusion counts per image: counts[b][t][p] = number of pixels with ground truth t (in [0, c); others skipped) and
 * arg-max class p (first maximum).  Replaces Measurement._make_confusion_matrix (measurement.py:12-20: np.bincount of
 * c * target + pred on the host) without the host round trip torch.bincount needs.  logits layout as for the Dice
 * sums; 2..4 classes; counts [b][c][c] i64, overwritten. */
int vqseg_confusion_counts_f(const float* logits, int64_t stride_b, int64_t stride_c, int64_t stride_px,
                             const int64_t* target, int b, int c, int64_t hw, int64_t* counts, void* stream);

/* Exact order statistics of n floats by radix select: out2[0] = the k-th smallest (0-based), out2[1] = the (k+1)-th
 * (clamped to the last).  The two values bracket the virtual index of np.percentile in make_regularized_pseudo_label
 * (deprecated/train_with_test_pt_pseudo_entropy_reg.py:35); the host interpolates.  Replaces the full device sort of
 * torch.quantile (and its 2^24-element input limit).  0 <= k < n < 2^32. */
/* The loss combination of a CPS iteration in ONE launch (r4; train_vqreptunet1x1v2.py:165-192: sup_1 + sup_2 + cps_weight * (cps_1 +
 * cps_2) + commitment + prototype, each supervised / CPS term = ce_weight * CE + Dice loss from the sums of vqseg_dice[_ce]_sums_*):
 *   term_i = ce_weight * ce_i[:, 0].sum() / ce_i[:, 1].sum() + (1 - mean_c mean_b 2 inter_i / (sets_i + eps))     (dice_loss.py:27-37)
 *   commitment = sum_l (sum_k commit[k][l]) * commit_weight,  prototype = (sum_k *proto[k]) * proto_weight  (float64 inputs)
 *   out[0] = total, out[1] = commitment, out[2] = prototype, out[3] = cps_1 + cps_2, out[4 + i] = term_i (supervised terms first)
 * and the gradient of the total with respect to every inter / sets / ce entry (g_*: same shapes; the commitment / prototype inputs'
 * gradients are the constants commit_weight / proto_weight).  It replaces ~45 scalar autograd nodes forward and ~70 tiny kernels
 * backward that ran one by one with the GPU idle.  ce / g_ce (and their entries) nullable: no cross-entropy part.  Pointer arrays
 * are host arrays of device pointers. */
int vqseg_cps_loss_combine_f(int n_sup, int n_cps, int c, const float* const* inter, const float* const* sets, const float* const* ce,
                             const int* b, float cps_weight, float ce_weight, float eps, const float* const* commit, int n_commit,
                             int levels, float commit_weight, const double* const* proto, int n_proto, float proto_weight,
                             float* const* g_inter, float* const* g_sets, float* const* g_ce, float* out, void* stream);
size_t vqseg_order_stats_workspace_bytes(void);
int vqseg_order_stats_f(const float* x, int64_t n, int64_t k, void* workspace, size_t workspace_bytes, float* out2,
                        void* stream);

/* ---------------------------------------------------------------------------------- *
 * The optimiser step: torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999)) created at train_vqreptunet1x1v2.py:106-107
 * and stepped at :200-201 (eps 1e-8, no weight decay, no amsgrad), for ALL parameters of a network in ONE launch, with the
 * kernel-side bf16 images of the k x k convolution weights (the three images of vqseg_conv_pack_all_f32) rewritten in the same
 * pass from the freshly updated values.  fp32 arithmetic in the operation order of torch/optim/adam.py::_single_tensor_adam:
 *     m = fma(1 - b1, g - m, m);  v = fma((1 - b2) g, g, v b2);  p = p + (-(lr / (1 - b1^t)) m) / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * (`step` = t >= 1, the value AFTER this step's increment; the scalars are evaluated in double like torch's Python floats).
 *   params_dev [n_params] VqsegAdamParam records in DEVICE memory; items_dev [n_items][2] int32 (parameter index, tile index) in
 *   DEVICE memory: for every parameter its vqseg_adam_work_items() tiles 0 .. count-1, in any order.
 *   k = 0: plain parameter, tiles are flat chunks of VQSEG_ADAM_CHUNK elements;
 *   k = 1 / 3: nn.Conv2d weight [cout][cin][k][k], tiles of 32 output x 128 / 32 input channels; fwd / tr / s3 (each nullable):
 *   the images [cout][k][k][cin^32], [cin][k][k flipped][cout^32], [cout][k][k][3 cin] ([w_hi | w_hi | w_lo] per concat segment
 *   split at c1; needs cin % 32 == 0 and c1 % 32 == 0) -- bit-identical to vqseg_conv_pack_all_f32 of the updated weight.
 * ---------------------------------------------------------------------------------- */
#define VQSEG_ADAM_CHUNK 4096
typedef struct VqsegAdamParam {
    float* p;            /* parameter        [numel] f32, updated in place */
    const float* g;      /* gradient         [numel] f32 */
    float* m;            /* exp_avg          [numel] f32, updated in place */
    float* v;            /* exp_avg_sq       [numel] f32, updated in place */
    int64_t numel;
    int32_t k, cout, cin, c1;
    void* fwd;
    void* tr;
    void* s3;
} VqsegAdamParam;
int64_t vqseg_adam_work_items(int64_t numel, int k, int cout, int cin);
int vqseg_adam_step_f32(const VqsegAdamParam* params_dev, const int32_t* items_dev, int n_items, double lr, double beta1,
                        double beta2, double eps, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQSEG_H */
