// Internal launch-helper declarations shared by the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace vqseg {

struct VqPlan {
    int Cp, Kp;                 // channels; codes padded to a multiple of 32
    int T;                      // accumulator tiles (32 codes each) per wave chosen for this shape
    size_t off_prepared, off_keys, off_hist, off_partial, bytes;
    int gather_blocks;
};
int vq_set_option(const char* key, int value);              // returns the previous value, -1: unknown key / bad value
struct KmPlan {
    VqPlan vq;
    size_t off_idx, off_counts, off_offsets, off_members, off_sums, off_counts64, off_hist, off_segoff, off_partial, bytes;
    int row_blocks, max_segments;
};

struct ProfShape {
    int64_t n;
    int c, k;
    int slot, group;            // first record of the launch this level ran in (its event pair), number of levels in that launch
};

constexpr int VQ_MAX_LEVELS = 4;
struct VqLevel {
    const void* x;              // pixel rows [N][C] (f32 or bf16: one type per launch)
    const float* E4;            // prepared codebook
    const float* enorm;
    unsigned long long* keys;   // [N] (distance bits << 32 | code), pre-set to ~0
    long N;
    int C, Kp;
    unsigned wg_end;            // exclusive end of this level's workgroup ids in the launch
};
#ifndef VQ_TIMELINE
#define VQ_TIMELINE 0             // debug build only (`make timeline` -> libvqseg_hip_tl.so; tools/vq_timeline.py): per-workgroup clock stamps
#endif
struct VqGroup {
    VqLevel lv[VQ_MAX_LEVELS];
    int n;
#if VQ_TIMELINE
    unsigned long long* tl;     // [workgroups][16]: (s_memrealtime, s_memtime) at 6 points of wave 0 + HW_ID + XCC_ID, or null
#endif
};
struct Profile {
    bool enabled = false;
    int capacity = 0;
    std::vector<hipEvent_t> ev;
    std::vector<ProfShape> shape;
};
hipError_t profile_begin(int capacity);
void profile_release();
int profile_collect(int max_records, int64_t* n, int* c, int* k, float* ms);

VqPlan vq_plan(int64_t N, int C, int K);
KmPlan km_plan(int64_t N, int C, int K);

size_t prepared_bytes(int C, int K);
hipError_t launch_prepare(const float* W, int K, int C, void* prepared, hipStream_t st);
hipError_t launch_assign(const void* x, int x_bf16, int64_t N, int C, int K, const void* prepared, const VqPlan& p, char* ws,
                         int64_t* idx, float* dmin, hipStream_t st);
int vq_group_tiles(int n, const int64_t* N, const int* K);
hipError_t launch_assign_group(int n, const void* const* x, int x_bf16, const int64_t* N, const int* C, const int* K,
                               const void* const* prepared, const VqPlan* plans, char* const* ws, int64_t* const* idx,
                               float* const* dmin, int T, hipStream_t st);
hipError_t launch_gather(const void* x, int bf16, const float* W, const int64_t* idx, int64_t N, int C, int K, int training,
                         float cw, const VqPlan& p, char* ws, void* quant, float* loss, float* dead, hipStream_t st);
hipError_t launch_backward(const float* gq, const float* gloss, const float* x, const float* q, int64_t N, int C,
                           float cw, float* gx, hipStream_t st);
hipError_t launch_backward_idx(const void* gq, const float* gloss, const void* x, const int64_t* idx, const float* W, int64_t N,
                               int C, float cw, void* gx, hipStream_t st);
hipError_t launch_km_accumulate(const float* samples, const float* means, int64_t N, int C, int K, const KmPlan& p,
                                char* ws, float* sums, int64_t* counts64, hipStream_t st);
// per-code sums / counts of rows already assigned (idx from a forward): the k-means accumulation without the distance pass
hipError_t launch_code_sums(const void* x, int x_bf16, const int64_t* idx, int64_t N, int C, int K, const KmPlan& p, char* ws,
                            float* sums, int64_t* counts64, hipStream_t st);
hipError_t launch_ema_update(float* cluster_size, float* embed_avg, float* codebook, const float* sums, const int64_t* counts64,
                             int K, int C, float decay, float eps, float* total, hipStream_t st);
hipError_t launch_km_finalize(const float* sums, const int64_t* counts64, float* means, int C, int K, hipStream_t st);

}  // namespace vqseg
