// Host-side launch interface of the loss kernels (prototype losses).  See include/vqseg.h for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqseg {

struct ProtoArgs {
    const void* x;               // decoder features, rows [M][C] (f32 or bf16)
    int bf16;
    const float* proto;          // L2-normalised prototypes [K][C]
    const long long* labels;     // [M] class ids
    const unsigned char* keep;   // v1: [M] entropy filter (nullable = all kept)
    const float* conf;           // v2: [M] confidence weights (nullable = 1)
    long M;
    int C, K, variant;           // variant 1: ReliablePrototypeLoss, 2: ReliablePrototypeLossv2
    float scale, cos_m, sin_m, th, mm;
    int easy_margin, use_margin; // use_margin: v1 applies phi only when margin != 0
};

constexpr int PROTO_ROWS_PER_BLOCK = 256;
long proto_blocks(long M);
hipError_t launch_proto_forward(const ProtoArgs& a, double* partial, double* loss, hipStream_t st);
hipError_t launch_proto_backward(const ProtoArgs& a, const float* g_loss, void* gx, float* gproto_partial, float* gproto,
                                 hipStream_t st);

// ---- soft Dice sums (loss/dice_loss.py:5-37): inter[b][c] = sum_px p_c 1[t == c],  sets[b][c] = sum_px (p_c + 1[t == c])
struct DiceArgs {
    const float* logits;         // element (b, c, px) at b * sb + c * sc + px * sp
    long sb, sc, sp;
    const long long* target;     // [B][HW]
    int B, C;
    long HW;
    long long ignore_index;
};
// per pixel: argmax label, entropy -sum p log(p + 1e-10), top probability of softmax(logits)   (any output nullable)
hipError_t launch_softmax_stats(const DiceArgs& a, long long* label, float* entropy, float* top, hipStream_t st);
// out2[0], out2[1] = the k-th and (k+1)-th smallest (0-based; the second clamped to n-1) of x[0..n): exact radix select
size_t order_stats_workspace_bytes();
hipError_t launch_order_stats(const float* x, long n, long k, void* workspace, float* out2, hipStream_t st);
// out[b][t][p] += 1 per pixel with ground truth t in [0, C) and arg-max class p (zeroed here first)
hipError_t launch_confusion(const DiceArgs& a, long long* out, hipStream_t st);
constexpr int DICE_PX_PER_BLOCK = 4096;
long dice_blocks(long HW);
// ce (nullable): [B][2] = (sum of -log softmax[target] over kept pixels, number of kept pixels); g_ce (nullable): [B][2], [b][0] used
hipError_t launch_dice_forward(const DiceArgs& a, double* partial, float* inter, float* sets, float* ce, hipStream_t st);
hipError_t launch_dice_backward(const DiceArgs& a, const float* g_inter, const float* g_sets, const float* g_ce, float* g_logits,
                                hipStream_t st);

// ---- the CPS step's loss combination in one launch (r4): term_i = ce_weight * CE_i + (1 - mean_c mean_b 2 inter / (sets + eps)), i over
// n_sup supervised terms then n_cps CPS terms; total = ((sum of sup terms + cps_weight * sum of cps terms) + commitment) + prototype with
// commitment = sum_l (sum_k commit[k][l]) * commit_weight, prototype = (sum_k proto[k]) * proto_weight; and d total / d every input.
struct CombineArgs {
    const float* inter[4]; const float* sets[4]; const float* ce[4];       // ce[i] nullable: [b][2]
    float* g_inter[4]; float* g_sets[4]; float* g_ce[4];
    int b[4];
    int n_sup, n_cps, c;
    float cps_weight, ce_weight, eps;
    const float* commit[4]; int n_commit, levels; float commit_weight;
    const double* proto[4]; int n_proto; float proto_weight;
    float* out;                  // [4 + n_sup + n_cps]: total, commitment, prototype, cps sum (unweighted), term values
};
hipError_t launch_loss_combine(const CombineArgs& a, hipStream_t st);

}  // namespace vqseg
