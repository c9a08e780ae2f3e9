// Internal declarations for the convolution / batch-norm / resampling kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vqseg {

struct ConvArgs {
    const void* x;              // input rows [N, H, W, C1]      (f32 in precise mode, bf16 in fast mode)
    const void* x2;             // optional second input [N, H, W, Cin - C1] (channel concat), else unused
    int C1;                     // channels taken from x (== Cin without concat)
    const unsigned short* w_hi; // packed weights [Cout][KH][KW][Cin padded to 32] bf16 (hi part)
    const unsigned short* w_lo; // lo part (precise mode) or null
    void* y;                    // output rows [N, Ho, Wo, Cout]
    float* stat_partial;        // optional [ceil(M/128) * WM][2][Cout] per-wave (mean, M2), else null
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, reflect, up;
    // optional fused epilogue (eval-mode BatchNorm): y = relu?( conv * ep_scale[c] + ep_shift[c] (+ ep_res) )
    const float* ep_scale;      // null: plain convolution output
    const float* ep_shift;
    const void* ep_res;         // optional residual rows [N, Ho, Wo, Cout] of the activation type
    int ep_relu;
    int pair_chunks = 0, pair_tiles = 0;   // patch kernel, 1-D launch: Cout chunks per pixel tile / pixel tiles (0: 2-D grid)
    // "split-3" output (fast kernels + fused epilogue only): y and ep_res rows are [2 * Cout] bf16 = [hi | lo] of the fp32
    // value (see vqseg.h, vqseg_conv2d_affine_f precise == 2)
    int out_s3 = 0;
    // split-3 INPUT: Cin / C1 above are the LOGICAL contraction lengths 3 * (C1s + C2s) / 3 * C1s of [hi | lo | hi] per concat
    // segment, the tensors x / x2 store [hi | lo] (2 * C1s / 2 * C2s channels per pixel)
    int s3_in = 0, s3_cs1 = 0, s3_cs2 = 0;
    int pad_w = -1;             // column padding when it differs from `pad` (-1: same); set by launch_conv
    // strided output placement: GEMM row (n, oh, ow) -> pixel (n, 2 oh + omap_ph, 2 ow + omap_pw) of an omap_h x omap_w grid
    int omap = 0, omap_h = 0, omap_w = 0, omap_ph = 0, omap_pw = 0;
    int prof_k = 0;             // kernel size the launch is filed under by the convolution profile (0: KH)
    // ring mode (r4; generic LDS-DMA kernel only): the GEMM rows are the border ring of a ring_h x ring_w output grid (top row,
    // bottom row, left column, right column -- 2 ring_w + 2 (ring_h - 2) pixels per image; set Ho = 1, Wo = that length), written
    // densely as [N][ring length][Cout].  The data gradient of a REFLECT-padded 3x3 layer needs the full correlation only there:
    // its interior is the zero-padded data gradient (the fast patch kernel), see launch_reflect_ring / reflect_ring_fold
    int ring = 0, ring_h = 0, ring_w = 0;
    int ep_res_out = 0;         // 1: ep_res is indexed like the OUTPUT (through omap) instead of by GEMM row -- in-place accumulation
    const unsigned char* ep_res_bits = nullptr;   // bf16 only, Cout % 8 == 0: bit (i % 8) of byte i / 8 over ep_res's flat index = keep that element (else 0):
                                                 // the ReLU mask of the block's last BatchNorm applied to the shortcut gradient on the way in
                                // into a tensor of the output's geometry (launch_dgrad_s2 with `accumulate`)
    // r4: the four parity classes of a stride-2 3x3 data gradient in ONE launch (conv_igemm_glds_kernel<..., CLS = true>): a workgroup
    // picks its class from its tile index and takes that class's taps / padding / extent / weight image; and the rows land
    // straight in the UNPADDED input gradient (omap_fold; omap_h x omap_w is then the input's own H x W):
    //   1 (reflect padding): padded pixel (i, j), i, j >= 1 -> x pixel (i - 1, j - 1); the padded top row / left column (whose
    //     gradients reflect onto x row 1 / column 1) -> a ring region behind the N * H * W rows ([N][W + 1 + H]), added by
    //     reflect_s2_ring_add; pixels beyond (i > H or j > W: no gradient reaches them) -> one dump row behind the ring
    //   2 (zero padding): the border of the padded grid is dropped (dump row behind the N * H * W rows)
    struct Cls {
        int KH, KW, pad, pad_w, Ho, Wo, ph, pw;
        unsigned w_off;         // element offset of the class's weight image
        unsigned tile_end;      // exclusive end of the class's M tiles in blockIdx.x
    };
    int n_cls = 0;
    Cls cls[4];
    int omap_fold = 0;
};

struct WgradArgs {
    const void* gy;             // output gradient rows [N, Ho, Wo, Cout]
    const void* x;              // forward input rows [N, H, W, C1]
    const void* x2;             // optional second input (concat)
    int C1;
    float* partial;             // [slabs][Cout][KH*KW][Cin] fp32
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, reflect;
    int per_tap_only;           // 1: x is an im2col patch matrix whose padded columns are dropped by the reduction
    // Second source (r4, "two-use" weight gradients): images [Na, N) of the virtual batch come from gy_b / x_b / x2_b (their image
    // index n - Na), images [0, Na) from gy / x / x2.  A weight used by two forwards of a training step (the labelled and the
    // unlabelled batch) gets ONE launch over both uses: twice the K extent per workgroup, half the slab sums.  Na == N: one source.
    const void* gy_b = nullptr;
    const void* x_b = nullptr;
    const void* x2_b = nullptr;
    int Na = 0;
    // r4: XCD-aware workgroup order of the LDS-DMA weight-gradient kernels (1-D grid).  The (ci tile, co tile) workgroups of ONE pixel
    // slab all read that slab's rows (each x slice once per co tile, each g_y slice once per ci tile); dispatched as a 3-D grid they
    // land on different XCDs (consecutive ids go round-robin) and every XCD fetches its own copy through the fabric (counters: 2.3-2.4 x
    // the algorithmic bytes).  With xcd_slabs > 0 workgroup id -> XCD = id & 7, slab = 8 * (id / (8 T)) + XCD, tile = (id / 8) % T
    // (T = ci tiles x co tiles): a slab's tiles run back to back on one XCD and share its L2.  0: plain (ci, co, slab) grid.
    int xcd_slabs = 0;
};

hipError_t launch_conv(const ConvArgs& a, int precise, hipStream_t st);
// rows the input-gradient buffer of launch_dgrad_s2_fold must have (N * H * W, + the ring region for reflect padding, + a dump row)
long dgrad_s2_fold_rows(int N, int H, int W, int reflect);
// 3x3 / stride 2 / pad 1 data gradient (bf16) written straight into the unpadded gx [N][H][W][Cin] (H = 2 Ho, W = 2 Wo), one launch
// for the four parity classes (+ the ring add for reflect padding).  hipErrorInvalidValue: shape outside the merged path.
hipError_t launch_dgrad_s2_fold(const void* gy, const unsigned short* w_hi, void* gx, int N, int Ho, int Wo, int Cout, int Cin, int H, int W,
                                int reflect, hipStream_t st);
// the stem (7x7 / 2 / pad 3, 3 -> 64) straight from the fp32 image, no patch matrix (Wo % 128 == 0; hipErrorInvalidValue otherwise)
hipError_t launch_stem7_fused(const float* x, const unsigned short* w_img, void* y, float* stat_partial, const float* ep_scale, const float* ep_shift,
                              int relu, int N, int H, int W, int reflect, int s3, hipStream_t st);
hipError_t conv_profile_begin(int capacity);
void conv_profile_release();
int conv_profile_collect(int max_records, double* flops, int* kind, float* ms, int* shape);
int conv_set_option(const char* key, int value);   // previous value, or -1 (unknown key)
int wgrad_slabs(const WgradArgs& a, int precise);   // slabs launch_wgrad will write
int wgrad_slabs_max(const WgradArgs& a);           // upper bound from the shape alone (workspace sizing)
hipError_t launch_wgrad(const WgradArgs& a, int precise, int slabs, int* final_layout, hipStream_t st);
hipError_t launch_wgrad_reduce(const float* partial, int slabs, int Cout, int Cin, int Cin_out, int KH, int KW, int im2col, int accumulate,
                               float* gw, hipStream_t st);
size_t packed_elems(int Cout, int Cin, int KH, int KW, int transpose_flip);
hipError_t launch_pack_weights(const float* w, int Cout, int Cin, int KH, int KW, int transpose_flip, unsigned short* hi,
                               unsigned short* lo, hipStream_t st);
hipError_t launch_pack_all(const float* w, int Cout, int Cin, int K, int C1, unsigned short* fwd, unsigned short* tr, unsigned short* s3,
                           hipStream_t st);
hipError_t launch_pack_weights_s2(const float* w, int Cout, int Cin, int K, unsigned short* hi, unsigned short* lo, hipStream_t st);
hipError_t launch_dgrad_s2(const void* gy, const unsigned short* w_hi, const unsigned short* w_lo, void* gx, int N, int Ho, int Wo,
                           int Cout, int Cin, int K, int OH, int OW, int precise, int accumulate, hipStream_t st);
hipError_t launch_reflect_ring(const void* gy, const unsigned short* t_hi, void* ring, int N, int H, int W, int Cgy, int Cgx, hipStream_t st);
hipError_t launch_pack_weights_s3(const float* w, int Cout, int Cin, int C1, int KH, int KW, unsigned short* out, hipStream_t st);

}  // namespace vqseg
