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
    // bf16 candidate filter (r4): per (row, 128-code sub-chunk) summaries, per-row band / threshold, candidate pair list (its counter
    // first: off_amb), pair capacity
    size_t off_summary, off_brow, off_amb, off_pair_rc;
    int pair_cap;
};
// Layout of the prepared codebook blob: [E4: C * Kp f32][enorm: Kp f32][en_max: 64 f32 (first one used)][Eb: 2 * C * Kp bf16]
// (Eb = the bf16 hi / lo split of the codebook for the candidate filter: [hl][C / 8][Kp][8]); every part 256-byte aligned.
struct PreparedLayout {
    size_t off_e4, off_enorm, off_enmax, off_eb, bytes;
};
PreparedLayout prepared_layout(int C, int K);
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
    int kind;                   // 0: exact fp32 MFMA kernel on f32 rows, 1: on bf16 rows, 2: bf16 candidate filter + exact re-score
};

constexpr int VQ_MAX_LEVELS = 4;
constexpr int F_LISTS = 64;     // sub-lists of a level's candidate pair list (bf16 filter)
struct VqLevel {
    const void* x;              // pixel rows [N][C] (f32 or bf16: one type per launch)
    const float* E4;            // prepared codebook
    const float* enorm;
    unsigned long long* keys;   // [N] (distance bits << 32 | code), pre-set to ~0
    long N;
    int C, Kp;
    unsigned wg_end;            // exclusive end of this level's workgroup ids in the launch
    // gated mode (the bf16 filter's overflow fallback): the level runs, in full, only if the filter's candidate list overflowed
    // (*gate > gate_cap) -- degenerate codebooks; otherwise every workgroup exits at once
    const int* gate = nullptr;
    int gate_cap = 0;
};
// one level of the bf16 candidate filter launch (vq_filter_bf16_kernel)
struct VqFilterLevel {
    const void* x;              // bf16 pixel rows [N][C]
    const unsigned short* Eb;   // [2][C / 8][Kp][8] bf16: hi / lo split of the codebook
    const float* enorm;         // |e_k|^2 (the exact kernel's values)
    const float* en_max;        // max_k |e_k|^2
    unsigned long long* summary;// [N][Kp / 128]: (best code | candidate count << 16) << 32 | float bits of the sub-chunk's minimum score
    float* brow;                // [N] band of the row (score space)
    // candidate pairs (row << 32 | code) for the exact re-score: every code inside the band of its sub-chunk's minimum other than
    // that minimum's own code (stage 1), the surviving minima of open rows (stage 2); pair_count[0] counts every append
    // The list is cut into F_LISTS sub-lists (a workgroup appends to sub-list blockIdx % F_LISTS) with a counter each -- thousands of
    // waves adding to ONE address serialise in the L2 atomic unit (measured: +50 us on the filter kernel); pair_count[F_LISTS] is
    // the overflow flag (an append beyond a sub-list's capacity sets it: the gated exact launch then serves the level)
    unsigned long long* pair_rc;
    int* pair_count;
    int pair_cap;               // capacity of ONE sub-list
    unsigned long long* keys;   // [N] the exact kernel's keys (decided rows: written by stage 2; open rows: atomicMin of stage 3)
    const float* W;             // fp32 codebook [K][C] (nullable: the re-score then reads the prepared image E4)
    const float* E4;
    long N;
    int C, Kp;
    unsigned wg_end;            // stage 1 launch: exclusive end of this level's workgroup ids
    unsigned rblk_end;          // stage 2 launch: exclusive end of this level's 256-row blocks
    unsigned sblk_end;          // stage 3 launch: exclusive end of this level's workgroups (a multiple of F_LISTS per level)
};
struct VqFilterGroup {
    VqFilterLevel lv[VQ_MAX_LEVELS];
    int n;
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
int profile_collect(int max_records, int64_t* n, int* c, int* k, float* ms, int* kind);

VqPlan vq_plan(int64_t N, int C, int K);
KmPlan km_plan(int64_t N, int C, int K);

size_t prepared_bytes(int C, int K);
hipError_t launch_prepare(const float* W, int K, int C, void* prepared, hipStream_t st);
// `codebook(s)` (nullable): the fp32 [K][C] weight the prepared blob was built from -- the bf16 filter's re-score reads contiguous
// code rows from it instead of the strided prepared image
hipError_t launch_assign(const void* x, int x_bf16, int64_t N, int C, int K, const void* prepared, const VqPlan& p, char* ws,
                         int64_t* idx, float* dmin, hipStream_t st, const float* codebook = nullptr);
int vq_group_tiles(int n, const int64_t* N, const int* K);
hipError_t launch_assign_group(int n, const void* const* x, int x_bf16, const int64_t* N, const int* C, const int* K,
                               const void* const* prepared, const VqPlan* plans, char* const* ws, int64_t* const* idx,
                               float* const* dmin, int T, hipStream_t st, const float* const* codebooks = nullptr);
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
