// vq_kernels.hip -- vector-quantiser hot path for gfx950 (MI355X, CDNA4).
//
// Reference algorithm: vector_quantizer/vq_img.py:160-177 (EuclideanCodebook.forward) and
// :228-244 (VectorQuantizer.forward); k-means init :29-63.  Nothing here is translated
// from the reference (which is stock ATen calls); the design is MI355X-first:
//
//   * the (N, K) distance matrix is never materialised: a wave owns a 32-row strip and
//     keeps 32 x 256 distances in 128 accumulator registers (8 tiles of
//     v_mfma_f32_32x32x2_f32, exact fp32 == a k-ordered fmaf chain), reduces them to a
//     running (min, index) per row in registers, and walks the codebook in 256-code
//     chunks;
//   * the codebook is pre-transposed once per call to ET[c][k] so that the MFMA B
//     fragment is a conflict-free 128-byte ds_read_b32 per half-wave, streamed through
//     LDS with direct global->LDS loads (global_load_lds_dwordx4), double buffered;
//   * pixel rows (A fragments) go global -> registers in fragment shape (each lane one
//     16-byte load per 8 channels); they are read once per 256-code chunk;
//   * 2 workgroups (8 waves) per CU: one wave's VALU epilogue (|x|^2 + |e|^2 - 2x.e,
//     clamp, sqrt, compare) overlaps the partner wave's MFMAs on the same SIMD.
//
// Arithmetic contract (shared with oracle/vq_chain.c, order "mfma8"):
//   dot(x, e)  = fmaf chain over channels in the order, per block of 8 channels 8j..8j+7:
//                8j, 8j+4, 8j+1, 8j+5, 8j+2, 8j+6, 8j+3, 8j+7
//   |x|^2      = (chain over channels with (c & 7) < 4, ascending) + (chain over the rest)
//   |e|^2      = chain over channels ascending
//   d          = sqrt(max(fmaf(-2, dot, |x|^2) + |e|^2, 0))        (correctly rounded sqrt)
//   idx        = lowest k attaining min d
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

#include "vq_kernels.h"

namespace vqseg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// codebook preparation ("prepared codebook" blob, valid until the codebook changes):
//   E4[c/4][k][4]  (Cp/4 x Kp x 4, zero padded)  -- 4 channels of one code are one 16-byte
//                  LDS/HBM word, so one ds_read_b128 feeds four MFMAs
//   enorm[k]       |e_k|^2 as an ascending fmaf chain; +inf for k >= K (never selected)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_pack_codebook(const float* __restrict__ W, int K, int C,
                                                        float* __restrict__ E4, int Kp, int Cp) {
    // one thread per (c4, k): reads 16 B of row k, writes 16 B
    const long total = (long)(Cp / 4) * Kp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const int c = (int)(i / Kp) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K && c < C) v = *reinterpret_cast<const f32x4*>(W + (size_t)k * C + c);   // C % 4 == 0
        reinterpret_cast<f32x4*>(E4)[i] = v;
    }
}

__global__ __launch_bounds__(64) void vq_code_norms(const float* __restrict__ W, int K, int C, int Kp,
                                                    float* __restrict__ enorm) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= Kp) return;
    float s = __builtin_inff();
    if (k < K) {
        s = 0.0f;
        const f32x4* row = reinterpret_cast<const f32x4*>(W + (size_t)k * C);
        const int n4 = C >> 2;
        int i = 0;
        for (; i + 8 <= n4; i += 8) {                            // 8 independent 16-B loads in flight
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_fmaf(v[u][e], v[u][e], s);
        }
        for (; i < n4; ++i) {
            const f32x4 v = row[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_fmaf(v[e], v[e], s);
        }
    }
    enorm[k] = s;
}

// bf16 hi / lo split of the codebook for the candidate filter: Eb[hl][c / 8][k][8]  (hl = 0: bf16(e), 1: bf16(e - hi)); zero for
// k >= K.  One thread per (c8, k): two 16-byte loads, two 16-byte stores.
__global__ __launch_bounds__(256) void vq_pack_codebook_bf16(const float* __restrict__ W, int K, int C, int Kp, unsigned short* __restrict__ Eb) {
    const long total = (long)(C / 8) * Kp;
    const size_t half = (size_t)C * Kp;                          // elements per (hi | lo) image
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const int c = (int)(i / Kp) * 8;
        u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
        if (k < K) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(W + (size_t)k * C + c), v1 = *reinterpret_cast<const f32x4*>(W + (size_t)k * C + c + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 h0 = (__bf16)v[2 * e], h1 = (__bf16)v[2 * e + 1];
                const __bf16 l0 = (__bf16)(v[2 * e] - (float)h0), l1 = (__bf16)(v[2 * e + 1] - (float)h1);
                hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
        }
        *reinterpret_cast<u32x4*>(Eb + (size_t)i * 8) = hi;
        *reinterpret_cast<u32x4*>(Eb + half + (size_t)i * 8) = lo;
    }
}

// max_k |e_k|^2 over the real codes (one workgroup; the padding entries of enorm are +inf)
__global__ __launch_bounds__(256) void vq_enorm_max(const float* __restrict__ enorm, int K, float* __restrict__ out) {
    __shared__ float sh[256];
    float m = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) m = __builtin_fmaxf(m, enorm[k]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] = __builtin_fmaxf(sh[threadIdx.x], sh[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ------------------------------------------------------------------------------------
// fused distance + argmin
//
// Work item (one workgroup, 4 waves): 128 pixel rows x (32*T) codes, all channels.
//   wave w owns rows [32w, 32w+32): T accumulator tiles of 32x32 (16 VGPRs each);
//   the code slab E4[:, code0 : code0+32T, :] streams through LDS in stages of BK channels
//   (BK*T = 128, i.e. 16 KiB per stage, double buffered, filled by global_load_lds);
//   B fragments are fetched one 8-channel block ahead of the MFMAs that use them.
// Results of different code groups of the same row are merged with a 64-bit atomicMin on
//   key = (float_bits(d) << 32) | code   (d >= 0, so unsigned order == float order and the
//   low word breaks ties towards the lowest code) -- order independent, hence deterministic.
// ------------------------------------------------------------------------------------
constexpr int ROWS_PER_WAVE = 32;
constexpr int WAVES = 4;
constexpr int ROWS_PER_WG = ROWS_PER_WAVE * WAVES;
#ifndef VQ_BIG_STAGE
#define VQ_BIG_STAGE 0
#endif
#ifndef VQ_SHAPE16
#define VQ_SHAPE16 0              // 1: timing-only experiment build (wrong results), see mfma_block
#endif
// floats per LDS stage = BK * 32T: 16 KiB, or (VQ_BIG_STAGE, wide workgroups) 32 KiB = half as many barriers
constexpr int stage_floats(int T) { return (VQ_BIG_STAGE && T >= 4) ? 8192 : 4096; }

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    // one wave-instruction: 64 lanes x 16 B -> 1 KiB contiguous in LDS at lds_wave_base
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename TX> struct RawFrag { typedef f32x4 type; };
template <> struct RawFrag<__bf16> { typedef u32x2 type; };

// TX: storage type of the pixel rows (float, or __bf16 -- every bf16 value is an exact float, the arithmetic is the same)
// One launch serves up to VQ_MAX_LEVELS independent quantisation problems ("levels": the three VQ layers of one forward).
// Workgroup ids [wg_end of the previous level, wg_end) belong to a level; the host orders the levels longest workgroup
// first (most channels), so that the short workgroups of the last level fill the tail of the launch.
#if VQ_TIMELINE
// per-workgroup timeline (debug build): stamp k of wave 0 = (constant 100 MHz real-time counter, shader cycle counter)
#define TL_STAMP(k_)                                                          \
    if (g.tl && threadIdx.x == 0) {                                           \
        tl_rt[k_] = __builtin_amdgcn_s_memrealtime();                         \
        tl_ck[k_] = __builtin_amdgcn_s_memtime();                             \
    }
#else
#define TL_STAMP(k_)
#endif

template <int T, typename TX = float>
__global__ __launch_bounds__(256, 2) void vq_assign_f32_kernel(const VqGroup g) {
#if VQ_TIMELINE
    unsigned long long tl_rt[6] = {0, 0, 0, 0, 0, 0}, tl_ck[6] = {0, 0, 0, 0, 0, 0};
    TL_STAMP(0)
#endif
    int lvl = 0;
    while (lvl + 1 < g.n && blockIdx.x >= g.lv[lvl].wg_end) ++lvl;                 // uniform (scalar) search
    const unsigned bid = blockIdx.x - (lvl ? g.lv[lvl - 1].wg_end : 0u);           // level offsets are multiples of 8 (XCD pairing below)
    const TX* __restrict__ x = static_cast<const TX*>(g.lv[lvl].x);
    const float* __restrict__ E4 = g.lv[lvl].E4;
    const float* __restrict__ enorm = g.lv[lvl].enorm;
    unsigned long long* __restrict__ keys = g.lv[lvl].keys;
    if (g.lv[lvl].gate && *g.lv[lvl].gate <= g.lv[lvl].gate_cap) return;       // the bf16 filter's overflow fallback: not needed (the usual case)
    const long N = g.lv[lvl].N;
    const int C = g.lv[lvl].C, Cp = g.lv[lvl].C, Kp = g.lv[lvl].Kp;
    constexpr int CODES = 32 * T;                // codes per workgroup
    constexpr int STAGE_FLOATS = stage_floats(T);
    constexpr int BK = STAGE_FLOATS / (32 * T);  // channels per stage
    constexpr int JB = BK / 8;                   // 8-channel blocks per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Bs = reinterpret_cast<float*>(smem);                 // [2][BK/4][CODES][4]
    float* xn_s = Bs + 2 * STAGE_FLOATS + 256;                  // [WAVES][32]  (after the key scratch, which aliases Bs)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_stage = (Cp + BK - 1) / BK;
    // Workgroup id -> (row tile, code chunk).  The `chunks` workgroups that share a row tile sit 8 ids apart: the
    // dispatcher deals consecutive ids round-robin over the 8 XCDs, so they land on the SAME XCD at about the same time and
    // the second reader of the rows hits that XCD's L2 instead of HBM (HBM fetch 435 -> ~300 MB per launch at K = 512).
    const int chunks = Kp / CODES;
    const unsigned within = bid % (8u * chunks);
    const long row_tile = (long)(bid / (8u * chunks)) * 8 + (within & 7u);
    if (row_tile * ROWS_PER_WG >= N) return;                     // padding of the last group of 8 (whole workgroup exits)
    const int code0 = (int)(within >> 3) * CODES;
    const long row0 = row_tile * ROWS_PER_WG + wave * ROWS_PER_WAVE;
    long row = row0 + r;
    if (row > N - 1) row = N - 1;                               // clamp: loads stay in bounds
    const TX* xrow = x + row * (long)C + 4 * h;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float xn_part = 0.0f;

    // stage fill: the stage is (BK/4) slabs of CODES*4 floats; one wave-instruction moves 256 floats
    // (64 codes x 4 channels).  A stage is 16 wave-instructions, 4 per wave.
    auto fill = [&](int stage, int buf) {
        float* dst = Bs + buf * STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < STAGE_FLOATS / 1024; ++q) {
            const int piece = wave * (STAGE_FLOATS / 1024) + q; // 256 floats each
            const int fl = piece * 256 + lane * 4;              // float offset inside the stage
            const int c4 = fl / (CODES * 4);                    // slab (4-channel group) inside the stage
            const int code = (fl % (CODES * 4)) >> 2;
            int slab = stage * (BK / 4) + c4;
            if (slab > Cp / 4 - 1) slab = Cp / 4 - 1;           // past the last channel: A is zero there, any finite B works
            const float* src = E4 + ((size_t)slab * Kp + code0 + code) * 4;
            glds16(src, dst + piece * 256);
        }
    };
    // bf16 rows stay RAW (8 bytes = 4 bf16 per block) until their stage becomes current (widen_a): converting at the
    // load would put the wait for the global load in front of the current stage's MFMAs
    using RawA = typename RawFrag<TX>::type;
    auto load_a = [&](int stage, RawA (&a)[JB]) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int col = stage * BK + 8 * j + 4 * h;
            if (col >= C) {
                a[j] = RawA{};
            } else {
                a[j] = *reinterpret_cast<const RawA*>(xrow + stage * BK + 8 * j);
            }
        }
    };

    // Software pipeline (one barrier per stage, no exposed LDS latency):
    //   stage s lives in LDS buffer s&1.  B fragments are fetched into registers one 8-channel block AHEAD of the
    //   MFMAs that consume them, so by the time the stage's last block starts every fragment of the stage is in
    //   registers: the barrier sits right there, the freed buffer is refilled for stage s+2, and the first block of
    //   stage s+1 is fetched under the last block's MFMAs.
    f32x4 a_cur[JB];
    RawA a_nxt[JB];
    fill(0, 0);
    if (n_stage > 1) fill(1, 1);
    auto widen_a = [&](f32x4 (&dst)[JB], const RawA (&src)[JB]) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if constexpr (sizeof(TX) == 4) {
                dst[j] = src[j];
            } else {
                const unsigned lo = src[j][0], hi = src[j][1];
                dst[j] = f32x4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xFFFF0000u),
                               __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xFFFF0000u)};
            }
        }
    };
    load_a(0, a_nxt);
    widen_a(a_cur, a_nxt);
#pragma unroll
    for (int j = 0; j < JB; ++j) a_nxt[j] = RawA{};
    __syncthreads();                                           // drains vmcnt, then barrier
    TL_STAMP(1)                                                // prologue done: the first MFMA follows

    // B fragment of lane (r, h) for block j, tile t: Bs[buf][2j + h][32 t + r][0..3]
    const float* bs_lane = Bs + (h * CODES + r) * 4;
    f32x4 b_cur[T], b_nxt[T];
#pragma unroll
    for (int t = 0; t < T; ++t) b_cur[t] = *reinterpret_cast<const f32x4*>(bs_lane + t * 128);

    auto mfma_block = [&](int j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = a_cur[j][e];
            xn_part = __builtin_fmaf(a, a, xn_part);
#if VQ_SHAPE16
            // TIMING EXPERIMENT ONLY (results garbage; VERDICT r3 item 2 ii): the same flops as one 32x32x2 per (e, t) issued as TWO
            // v_mfma_f32_16x16x4_f32 on quarters of the accumulator tile, inside the real kernel -- does the chip hold another clock?
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 q0 = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]}, q1 = {acc[t][8], acc[t][9], acc[t][10], acc[t][11]};
                q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b_cur[t][e], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b_cur[t][e], q1, 0, 0, 0);
                acc[t][0] = q0[0], acc[t][1] = q0[1], acc[t][2] = q0[2], acc[t][3] = q0[3];
                acc[t][8] = q1[0], acc[t][9] = q1[1], acc[t][10] = q1[2], acc[t][11] = q1[3];
            }
#else
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b_cur[t][e], acc[t], 0, 0, 0);
#endif
        }
        // issue order: one LDS read per four MFMAs (the reads belong to the NEXT block and are never waited on here)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };

    for (int s = 0; s < n_stage; ++s) {
        const float* bsrc = bs_lane + (s & 1) * STAGE_FLOATS;
        if (s + 1 < n_stage) load_a(s + 1, a_nxt);
#pragma unroll
        for (int j = 0; j < JB - 1; ++j) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                b_nxt[t] = *reinterpret_cast<const f32x4*>(bsrc + (2 * (j + 1)) * CODES * 4 + t * 128);
            mfma_block(j);
#pragma unroll
            for (int t = 0; t < T; ++t) b_cur[t] = b_nxt[t];
        }
        __syncthreads();                                       // stage s+1 landed; nobody reads buffer s&1 any more
        if (s + 2 < n_stage) fill(s + 2, s & 1);
        if (s + 1 < n_stage) {
            const float* bnext = bs_lane + ((s + 1) & 1) * STAGE_FLOATS;
#pragma unroll
            for (int t = 0; t < T; ++t) b_nxt[t] = *reinterpret_cast<const f32x4*>(bnext + t * 128);
        }
        mfma_block(JB - 1);
#pragma unroll
        for (int t = 0; t < T; ++t) b_cur[t] = b_nxt[t];
        widen_a(a_cur, a_nxt);
    }

    // |e|^2 of this lane's code in every tile: ONE batch of loads, issued while the last MFMAs drain (the fragment registers of the
    // main loop are dead now; holding these through the loop spilled 67 registers)
    float en[T];
#pragma unroll
    for (int t = 0; t < T; ++t) en[t] = enorm[code0 + t * 32 + r];
    TL_STAMP(2)                                                // main loop issued (the last MFMAs are still draining)
    // ---- epilogue: distances -> (min, code) per row over this workgroup's codes
    {
        const float xn = xn_part + __shfl_xor(xn_part, 32);
        if (h == 0) xn_s[wave * 32 + r] = xn;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): xn_s visible within the wave
    __builtin_amdgcn_wave_barrier();
    float xnr[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xn_s + wave * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) xnr[4 * g + e] = v[e];
    }
    // Running per-lane minimum in SQUARED-distance space.
    float best2[16];
    int best_i[16];
    // Row by row (the 16 rows of a lane are independent; within a row the codes arrive in increasing index order).
    // FAST scan, branch-free: only "clearly better" candidates (below the holder by more than the collapse band 2^-20) move the
    // holder; a candidate inside the band raises `band`.  Without a band event anywhere in the wave the result IS the exact scan's
    // (every decision was a clear one).  With one (rare: two codes within 1e-6 relative of a row's running minimum) the wave redoes
    // that row on the EXACT path: the reference argmins over sqrt(d2), whose rounding can collapse nearly equal d2 into one float --
    // then the LOWER index wins -- so a candidate inside the band compares correctly-rounded square roots, and "keep the holder on
    // a tie" is the tie rule.  One wave-uniform branch per row instead of one per (row, tile); the tiles' |e|^2 come from en[] (they
    // used to be loaded tile by tile behind that branch: eight serialised L2 latencies, 19 us of a 141 us workgroup --
    // profiles/r03_vq_wg_timeline_before.md).
    const int code_r = code0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float b2 = __builtin_inff(), thr = __builtin_inff();
        int bi = 0x7fffffff;
        bool band = false;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float d2 = __builtin_fmaf(-2.0f, acc[t][i], xnr[i]);
            d2 = d2 + en[t];
            d2 = __builtin_fmaxf(d2, 0.0f);
            const bool clear = d2 < thr;                                       // better by more than the collapse band
            band = band || (!clear && d2 < b2);                                // inside the band
            bi = clear ? code_r + 32 * t : bi;
            b2 = clear ? d2 : b2;
            thr = clear ? d2 * 0.99999905f : thr;
        }
        if (__builtin_expect(__any(band), 0)) {
            b2 = __builtin_inff();
            thr = __builtin_inff();
            bi = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float d2 = __builtin_fmaf(-2.0f, acc[t][i], xnr[i]);
                d2 = d2 + en[t];
                d2 = __builtin_fmaxf(d2, 0.0f);
                const bool clear = d2 < thr;
                const bool near = !clear && d2 < b2;
                if (near) {
                    const bool better = __builtin_sqrtf(d2) < __builtin_sqrtf(b2);
                    bi = better ? code_r + 32 * t : bi;
                    b2 = d2;                                                   // same sqrt class or better: safe to lower
                    thr = d2 * 0.99999905f;
                }
                bi = clear ? code_r + 32 * t : bi;
                b2 = clear ? d2 : b2;
                thr = clear ? d2 * 0.99999905f : thr;
            }
        }
        best2[i] = b2;
        best_i[i] = bi;
    }
    TL_STAMP(3)                                                // running minima done (waited on every accumulator)
    // ---- merge the 32 code lanes of each half: keys (float_bits(sqrt d2) << 32 | code) through LDS, 2 lanes per row
    // (the B stage buffers are free: after the last stage barrier every fragment lives in registers)
    unsigned long long* ks = reinterpret_cast<unsigned long long*>(Bs) + wave * (32 * 33);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(__builtin_sqrtf(best2[i])) << 32) | (unsigned int)best_i[i];
        ks[(h * 16 + i) * 33 + r] = key;                                       // row (h, i), candidate lane r
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                        // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    {
        const int rw = lane >> 1, part = lane & 1;                             // rw = h' * 16 + i'
        unsigned long long m = ~0ull;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned long long v = ks[rw * 33 + part * 16 + j];
            m = v < m ? v : m;
        }
        const unsigned long long o = __shfl_xor(m, 1);
        m = o < m ? o : m;
        const int hh = rw >> 4, ii = rw & 15;
        const long orow = row0 + (ii & 3) + 8 * (ii >> 2) + 4 * hh;
        TL_STAMP(4)
        if (part == 0 && orow < N) atomicMin(keys + orow, m);
    }
#if VQ_TIMELINE
    TL_STAMP(5)
    if (g.tl && threadIdx.x == 0) {
        unsigned long long* o = g.tl + (size_t)blockIdx.x * 16;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            o[2 * k] = tl_rt[k];
            o[2 * k + 1] = tl_ck[k];
        }
        o[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        o[14] = (unsigned long long)lvl;
        o[15] = (unsigned long long)row_tile;
    }
#endif
}

// ------------------------------------------------------------------------------------
// bf16 candidate filter (r4): the distance + argmin of BF16 pixel rows at bf16-MFMA speed with the exact kernel's result.
//
// The rows are exact bf16 values (autocast activations); the fp32 codebook is split e = e_hi + e_lo + rho (bf16 each,
// |rho| <= 2^-18 |e|).  Stage 1 (this kernel) computes approximate scores
//       s~_k = |e_k|^2 - 2 (x . e_hi_k + x . e_lo_k)        on v_mfma_f32_32x32x16_bf16 (fp32 accumulation)
// for a 128-row x 256-code tile per workgroup and leaves, per row and 128-code sub-chunk, the minimum score, its code and the number
// of codes within the row's BAND of that minimum; every further code inside the band goes to a list of candidate pairs.  Stage 2
// (vq_resolve_kernel) takes the minimum over the sub-chunks: a row with exactly ONE code inside the band of the global minimum is
// decided (that code wins under the exact arithmetic too); for every other row the surviving sub-chunks' best codes join the pair
// list.  Stage 3 (vq_rescore_kernel) evaluates the EXACT kernel's fmaf chains for the listed (row, code) pairs on the vector ALU --
// one lane per pair -- and merges them with the exact kernel's own key rule: index AND distance bits are those of the exact chain.
// Whatever share of the rows is open (4 % on well separated codebooks, half of them on the 2048-channel level of an untrained
// network), only their few candidates are re-scored, not their K codes.  A pair list that overflows (degenerate codebooks: more
// than four candidates per row on average) switches the level to the exact kernel on every row (gated launch).
//
// The band.  Let D_k be the exact kernel's squared distance (fl(fl(|x|^2 - 2 dot_chain_k) + |e_k|^2)) and S = sum_i |x_i| |e_ki|
// <= sqrt(|x|^2 max_k |e_k|^2) =: S_max.  Errors against the real value |x|^2 + |e_k|^2 - 2 x.e_k:
//     exact chain of C fmaf's               |dot_chain - x.e| <= C 2^-24 S
//     split residual                         |x.rho|           <= 2^-18 S
//     C / 8 bf16 MFMAs, products exact, each sum of 17 addends aligned to the largest one and truncated below 2^-24 of it
//     (tools/micro/mfma_bf16_rounding.hip: errors up to 2^-23.3 of the magnitude sum on crafted rows, never more than
//      17 * 2^-24 of the largest addend)     |dot~ - x.(e_hi + e_lo)| <= (C / 8) 17 2^-24 S
//     the final roundings of both paths      <= 2^-22 (|x|^2 + |e|^2 + 2 S)
// so |(|x|^2 + s~_k) - D_k| <= eps := 2 (C 2^-24 + 2^-18 + (C / 8) 17 2^-24) S_max + 2^-22 (|x|^2 + E + 2 S_max), E = max |e|^2.
// The reference argmins over sqrt(D): codes whose D differ by less than 2^-20 relative can collapse into one float, the lower index
// wins.  Hence: if D_a <= D_b (1 + 2^-20) then s~_a <= s~_b + 2 eps + 2^-20 (|x|^2 + E + 2 S_max) =: BAND (x 1.25 for margin).
// Any code that can win or tie under the exact arithmetic lies within BAND of the approximate minimum.
//
// Roles are swapped against the exact kernel: codes are the MFMA's M dimension (LDS, staged by LDS-DMA), pixels the N dimension
// (registers, straight from global in fragment shape), so that a lane owns ONE pixel per fragment and the minimum / band count over
// its codes needs no cross-lane traffic (the two half-waves meet in one shuffle).
// Workgroup: 4 waves = 2 (64 pixels) x 2 (128 codes); wave tile = 2 pixel fragments x 4 code tiles = 8 accumulators.
// ------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int F_ROWS = 128, F_CODES = 256, F_SUB = 128;      // workgroup tile; codes per summary
constexpr int F_BK = 32;                                    // channels per LDS stage (two MFMA K steps)
constexpr int F_STAGE_BYTES = 2 * (F_BK / 8) * F_CODES * 16; // [hi | lo][4 slabs of 8 channels][256 codes][8 bf16] = 32 KiB

__device__ __forceinline__ void glds16b(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base,
                                     16, 0, 0);
}

// exclusive prefix sum of v over the 64 lanes; total = the wave's sum
__device__ __forceinline__ int wave_excl_scan(int v, int& total) {
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if ((int)(threadIdx.x & 63) >= d) inc += o;
    }
    total = __shfl(inc, 63);
    return inc - v;
}

struct FilterConsts {
    float c_s, c_r;             // BAND = c_s * sqrt(|x|^2 * E) + c_r * (|x|^2 + E)   (per level: depends on C)
};

__global__ __launch_bounds__(256, 2) void vq_filter_bf16_kernel(const VqFilterGroup g, const FilterConsts fc0, const FilterConsts fc1,
                                                                const FilterConsts fc2, const FilterConsts fc3) {
    int lvl = 0;
    while (lvl + 1 < g.n && blockIdx.x >= g.lv[lvl].wg_end) ++lvl;
    const unsigned bid = blockIdx.x - (lvl ? g.lv[lvl - 1].wg_end : 0u);
    const FilterConsts fc = lvl == 0 ? fc0 : (lvl == 1 ? fc1 : (lvl == 2 ? fc2 : fc3));
    const unsigned short* __restrict__ x = static_cast<const unsigned short*>(g.lv[lvl].x);
    const unsigned short* __restrict__ Eb = g.lv[lvl].Eb;
    const long N = g.lv[lvl].N;
    const int C = g.lv[lvl].C, Kp = g.lv[lvl].Kp;
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    char* Bs = fsm;                                             // [2 buffers][F_STAGE_BYTES]
    float* en_s = reinterpret_cast<float*>(fsm + 2 * F_STAGE_BYTES);   // [F_CODES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_stage = C / F_BK;
    // workgroup id -> (row tile, code chunk): the chunks of a row tile sit 8 ids apart = same XCD, about the same time (see the exact kernel)
    const int chunks = Kp / F_CODES;
    const unsigned within = bid % (8u * chunks);
    const long row_tile = (long)(bid / (8u * chunks)) * 8 + (within & 7u);
    if (row_tile * F_ROWS >= N) return;
    const int chunk = (int)(within >> 3);
    const int code0 = chunk * F_CODES;

    en_s[tid] = g.lv[lvl].enorm[code0 + tid];                   // 256 threads, 256 codes

    const unsigned short* xrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        long row = row_tile * F_ROWS + wp * 64 + p * 32 + r;
        if (row > N - 1) row = N - 1;
        xrow[p] = x + row * (long)C + 8 * h;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][t][i] = 0.0f;
    float xn_part[2] = {0.0f, 0.0f};

    // stage fill: 32 wave-instructions of 1 KiB (64 codes x 16 B of one (hi|lo, slab)), 8 per wave
    auto fill = [&](int stage, int buf) {
        char* dst = Bs + buf * F_STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int piece = wave * 8 + q;
            const int hlslab = piece >> 2, quarter = piece & 3;
            const int hl = hlslab >> 2, slab = hlslab & 3;
            const unsigned short* src = Eb + (((size_t)hl * (C / 8) + (size_t)stage * 4 + slab) * Kp + code0 + quarter * 64 + lane) * 8;
            glds16b(src, dst + (hlslab * F_CODES + quarter * 64) * 16);
        }
    };
    auto load_x = [&](int stage, u32x4 (&xf)[2][2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) xf[p][q] = *reinterpret_cast<const u32x4*>(xrow[p] + stage * F_BK + q * 16);
    };

    u32x4 xcur[2][2], xnxt[2][2];
    fill(0, 0);
    if (n_stage > 1) fill(1, 1);
    load_x(0, xcur);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) xnxt[p][q] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const char* a_lane = Bs + ((h * F_CODES) + wc * 128 + r) * 16;     // + buf * STAGE + ((hl * 4 + 2 q) * F_CODES + 32 t) * 16
    for (int s = 0; s < n_stage; ++s) {
        const char* abase = a_lane + (s & 1) * F_STAGE_BYTES;
        if (s + 1 < n_stage) load_x(s + 1, xnxt);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8*>(abase + ((0 * 4 + 2 * q) * F_CODES + 32 * t) * 16);
                al[t] = *reinterpret_cast<const bf16x8*>(abase + ((1 * 4 + 2 * q) * F_CODES + 32 * t) * 16);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, xcur[p][q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {                     // |x|^2 of this lane's 8 channels (fp32; only the band needs it)
                    const float lo = __builtin_bit_cast(float, xcur[p][q][e] << 16), hi = __builtin_bit_cast(float, xcur[p][q][e] & 0xFFFF0000u);
                    xn_part[p] = __builtin_fmaf(lo, lo, xn_part[p]);
                    xn_part[p] = __builtin_fmaf(hi, hi, xn_part[p]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], xb, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], xb, acc[p][t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                        // stage s + 1 landed (issued a whole stage ago); buffer s & 1 is free
        if (s + 2 < n_stage) fill(s + 2, s & 1);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) xcur[p][q] = xnxt[p][q];
    }

    // ---- epilogue: scores of this lane's 64 codes per pixel fragment, minimum, band count
    const float en_max = *g.lv[lvl].en_max;
    const int sub = chunk * 2 + wc, n_sub = Kp / F_SUB;
    const float* en_l = en_s + wc * 128 + 4 * h;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float m = __builtin_inff();
        int bi = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 en4 = *reinterpret_cast<const f32x4*>(en_l + 32 * t + 8 * gq);    // codes 32 t + 8 gq + 4 h + (0..3): same address in a half-wave
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = __builtin_fmaf(-2.0f, acc[p][t][4 * gq + e], en4[e]);
                    acc[p][t][4 * gq + e] = sc;
                    const bool better = sc < m;                   // codes arrive in increasing index order: the first minimum is kept
                    bi = better ? 32 * t + 8 * gq + e : bi;
                    m = better ? sc : m;
                }
            }
        bi += 4 * h;
        {   // the other half-wave holds the other 64 codes of this pixel
            const float mo = __shfl_xor(m, 32);
            const int bo = __shfl_xor(bi, 32);
            const bool take = mo < m || (mo == m && bo < bi);
            m = take ? mo : m;
            bi = take ? bo : bi;
        }
        const float xn = xn_part[p] + __shfl_xor(xn_part[p], 32);
        const float band = __builtin_fmaf(fc.c_s, __builtin_sqrtf(xn * en_max), fc.c_r * (xn + en_max));
        const float thr = m + band;
        // which of this half-wave's 64 codes lie inside the band: bit 16 t + i of (lo, hi) <-> acc[p][t][i]
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) lo |= acc[p][t][i] <= thr ? (1u << (16 * t + i)) : 0u;
#pragma unroll
        for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) hi |= acc[p][t][i] <= thr ? (1u << (16 * (t - 2) + i)) : 0u;
        const int mine = __popc(lo) + __popc(hi);
        const int cnt = mine + __shfl_xor(mine, 32);
        const long row = row_tile * F_ROWS + wp * 64 + p * 32 + r;
        const bool live = row < N;
        if (h == 0 && live) {
            const unsigned hiw = (unsigned)(code0 + wc * 128 + bi) | ((unsigned)(cnt > 0xffff ? 0xffff : cnt) << 16);
            g.lv[lvl].summary[row * n_sub + sub] = ((unsigned long long)hiw << 32) | __float_as_uint(m);
            if (sub == 0) g.lv[lvl].brow[row] = band;
        }
        // candidates beyond the best code -> the pair list (one reservation per wave; only lanes that have any walk their bits)
        if ((bi & 4) == 4 * h) {                                  // the best code sits in this half-wave: clear its bit
            const int bb = 16 * (bi >> 5) + 4 * ((bi >> 3) & 3) + (bi & 3);
            if (bb < 32) lo &= ~(1u << bb);
            else hi &= ~(1u << (bb - 32));
        }
        const int extra = live ? __popc(lo) + __popc(hi) : 0;
        if (__any(extra > 0)) {                                   // wave-uniform
            int total;
            const int before = wave_excl_scan(extra, total);
            const int list = blockIdx.x & (F_LISTS - 1), cap = g.lv[lvl].pair_cap;
            int base = 0;
            if (lane == 0) {
                base = atomicAdd(g.lv[lvl].pair_count + list, total);
                if (base + total > cap) g.lv[lvl].pair_count[F_LISTS] = 1;       // overflow: the gated exact launch serves the level
            }
            base = __shfl(base, 0);
            if (extra > 0) {
                int slot = base + before;
                unsigned long long* out = g.lv[lvl].pair_rc + (size_t)list * cap;
                unsigned long long bits = ((unsigned long long)hi << 32) | lo;
                while (bits) {
                    const int b = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    const int cl = 32 * (b >> 4) + 8 * ((b >> 2) & 3) + 4 * h + (b & 3);
                    if (slot < cap) out[slot] = ((unsigned long long)row << 32) | (unsigned)(code0 + wc * 128 + cl);
                    ++slot;
                }
            }
        }
    }
}

// Stage 2 (all levels of the launch): per row the minimum over its sub-chunk summaries; exactly one code inside the band -> decided
// (key written directly, the row's threshold word set to -inf: its listed candidates are skipped); otherwise the best code of every
// surviving sub-chunk joins the pair list and brow keeps +inf-free "open" (the threshold itself).
constexpr int F_MAX_SUB = 16;                               // sub-chunk summaries a row can have in registers (K <= 2048); more: re-read
__global__ __launch_bounds__(256) void vq_resolve_kernel(const VqFilterGroup g, int force_all) {
    int lvl = 0;
    while (lvl + 1 < g.n && blockIdx.x >= g.lv[lvl].rblk_end) ++lvl;
    const VqFilterLevel& L = g.lv[lvl];
    const long N = L.N;
    const int n_sub = L.Kp / F_SUB;
    long row = (long)(blockIdx.x - (lvl ? g.lv[lvl - 1].rblk_end : 0u)) * 256 + threadIdx.x;
    const bool valid = row < N;
    if (!valid) row = N - 1;                                   // keep the wave whole for the scan below
    const unsigned long long* sm = L.summary + row * n_sub;
    unsigned long long v[F_MAX_SUB];
#pragma unroll
    for (int j = 0; j < F_MAX_SUB; j += 2)
        if (j < n_sub) {                                        // n_sub is even (K % 256 == 0): 16-byte loads
            const u32x4 w = *reinterpret_cast<const u32x4*>(sm + j);
            v[j] = ((unsigned long long)w[1] << 32) | w[0];
            v[j + 1] = ((unsigned long long)w[3] << 32) | w[2];
        }
    float m = __builtin_inff();
    unsigned best = 0;
#pragma unroll
    for (int j = 0; j < F_MAX_SUB; ++j)
        if (j < n_sub) {
            const float mj = __uint_as_float((unsigned)(v[j] & 0xffffffffull));
            const unsigned cj = (unsigned)(v[j] >> 32) & 0xffffu;
            if (mj < m || (mj == m && cj < best)) m = mj, best = cj;
        }
    for (int j = F_MAX_SUB; j < n_sub; ++j) {                   // K > 2048
        const float mj = __uint_as_float((unsigned)(sm[j] & 0xffffffffull));
        const unsigned cj = (unsigned)(sm[j] >> 32) & 0xffffu;
        if (mj < m || (mj == m && cj < best)) m = mj, best = cj;
    }
    const float thr = m + L.brow[row];
    unsigned cnt = 0;
    int alive = 0;
#pragma unroll
    for (int j = 0; j < F_MAX_SUB; ++j)
        if (j < n_sub && __uint_as_float((unsigned)(v[j] & 0xffffffffull)) <= thr) cnt += (unsigned)(v[j] >> 48), ++alive;
    for (int j = F_MAX_SUB; j < n_sub; ++j)
        if (__uint_as_float((unsigned)(sm[j] & 0xffffffffull)) <= thr) cnt += (unsigned)(sm[j] >> 48), ++alive;
    const bool open = valid && !(cnt == 1 && !force_all);
    if (valid) {
        if (!open) L.keys[row] = (unsigned long long)best;    // distance word 0: decided rows carry no exact distance (see vq_exact_dist_kernel)
        L.brow[row] = open ? thr : -__builtin_inff();
    }
    if (!__any(open)) return;
    int total;
    const int before = wave_excl_scan(open ? alive : 0, total);
    const int list = blockIdx.x & (F_LISTS - 1), cap = L.pair_cap;
    int base = 0;
    if ((threadIdx.x & 63) == 0) {
        base = atomicAdd(L.pair_count + list, total);
        if (base + total > cap) L.pair_count[F_LISTS] = 1;
    }
    base = __shfl(base, 0);
    if (open) {
        int slot = base + before;
        unsigned long long* out = L.pair_rc + (size_t)list * cap;
        for (int j = 0; j < n_sub; ++j) {
            const unsigned long long vj = sm[j];
            if (__uint_as_float((unsigned)(vj & 0xffffffffull)) <= thr) {
                if (slot < cap) out[slot] = ((unsigned long long)row << 32) | ((unsigned)(vj >> 32) & 0xffffu);
                ++slot;
            }
        }
    }
}

// Stage 3 (all levels): the exact kernel's arithmetic for the listed (row, code) pairs of OPEN rows, one lane per pair: dot = the
// "mfma8" fmaf chain, |x|^2 its two half chains, |e|^2 from the prepared blob, d = sqrt(max(fmaf(-2, dot, |x|^2) + |e|^2, 0)),
// key = (bits(d) << 32) | code merged by atomicMin like the exact kernel's code groups (the lowest code wins a tie).
struct RescoreTrip {                                        // 32 channels of one (row, code) pair
    u32x4 xw[4];
    f32x4 e0[4], e1[4];
};

template <bool ROWMAJOR>
__device__ __forceinline__ void rescore_load(RescoreTrip& t, const unsigned short* xr, const float* er, int c, int Kp) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        t.xw[u] = *reinterpret_cast<const u32x4*>(xr + c + 8 * u);
        if (ROWMAJOR) {
            t.e0[u] = *reinterpret_cast<const f32x4*>(er + c + 8 * u);
            t.e1[u] = *reinterpret_cast<const f32x4*>(er + c + 8 * u + 4);
        } else {                                                // prepared image: 4-channel groups Kp * 16 bytes apart
            t.e0[u] = *reinterpret_cast<const f32x4*>(er + (size_t)((c + 8 * u) / 4) * Kp * 4);
            t.e1[u] = *reinterpret_cast<const f32x4*>(er + (size_t)((c + 8 * u) / 4 + 1) * Kp * 4);
        }
    }
}

__device__ __forceinline__ void rescore_fma(const RescoreTrip& t, float& dot, float& xa, float& xb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float xv[8] = {__builtin_bit_cast(float, t.xw[u][0] << 16), __builtin_bit_cast(float, t.xw[u][0] & 0xFFFF0000u),
                             __builtin_bit_cast(float, t.xw[u][1] << 16), __builtin_bit_cast(float, t.xw[u][1] & 0xFFFF0000u),
                             __builtin_bit_cast(float, t.xw[u][2] << 16), __builtin_bit_cast(float, t.xw[u][2] & 0xFFFF0000u),
                             __builtin_bit_cast(float, t.xw[u][3] << 16), __builtin_bit_cast(float, t.xw[u][3] & 0xFFFF0000u)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dot = __builtin_fmaf(xv[e], t.e0[u][e], dot);
            dot = __builtin_fmaf(xv[4 + e], t.e1[u][e], dot);
            xa = __builtin_fmaf(xv[e], xv[e], xa);
            xb = __builtin_fmaf(xv[4 + e], xv[4 + e], xb);
        }
    }
}

// one sub-list of one level: lane i takes pairs i, i + nthreads, ...; a chain is serial (C dependent fmaf's), so its loads run two
// 32-channel trips ahead of the arithmetic
template <bool ROWMAJOR>
__device__ __forceinline__ void rescore_pairs(const VqFilterLevel& L, int list, int tid, int nthreads) {
    int n = L.pair_count[list];
    if (n > L.pair_cap) n = L.pair_cap;                        // overflow: the gated exact launch serves the level anyway
    const int C = L.C, Kp = L.Kp;
    const unsigned short* __restrict__ x = static_cast<const unsigned short*>(L.x);
    const unsigned long long* pairs = L.pair_rc + (size_t)list * L.pair_cap;
    for (int i = tid; i < n; i += nthreads) {
        const unsigned long long rc = pairs[i];
        const long row = (long)(rc >> 32);
        const unsigned k = (unsigned)(rc & 0xffffffffull);
        if (L.brow[row] == -__builtin_inff()) continue;         // a decided row
        const unsigned short* xr = x + row * (long)C;
        const float* er = ROWMAJOR ? L.W + (size_t)k * C : L.E4 + (size_t)k * 4;
        float dot = 0.0f, xa = 0.0f, xb = 0.0f;
        RescoreTrip t0, t1, t2;                                  // C % 32 == 0 on this path
        rescore_load<ROWMAJOR>(t0, xr, er, 0, Kp);
        if (C > 32) rescore_load<ROWMAJOR>(t1, xr, er, 32, Kp);
        int c = 0;
        for (; c + 96 <= C; c += 96) {
            if (c + 64 < C) rescore_load<ROWMAJOR>(t2, xr, er, c + 64, Kp);
            rescore_fma(t0, dot, xa, xb);
            if (c + 96 < C) rescore_load<ROWMAJOR>(t0, xr, er, c + 96, Kp);
            rescore_fma(t1, dot, xa, xb);
            if (c + 128 < C) rescore_load<ROWMAJOR>(t1, xr, er, c + 128, Kp);
            rescore_fma(t2, dot, xa, xb);
        }
        if (c < C) {                                            // one or two trips left (already loaded into t0 / t1)
            rescore_fma(t0, dot, xa, xb);
            if (c + 32 < C) rescore_fma(t1, dot, xa, xb);
        }
        float d2 = __builtin_fmaf(-2.0f, dot, xa + xb) + L.enorm[k];
        d2 = __builtin_fmaxf(d2, 0.0f);
        atomicMin(L.keys + row, ((unsigned long long)__float_as_uint(__builtin_sqrtf(d2)) << 32) | k);
    }
}

// Stage 3 launch: every level gets its own workgroups (a multiple of F_LISTS), workgroup b of a level serves sub-list b % F_LISTS
__global__ __launch_bounds__(256) void vq_rescore_kernel(const VqFilterGroup g) {
    int lvl = 0;
    while (lvl + 1 < g.n && blockIdx.x >= g.lv[lvl].sblk_end) ++lvl;
    const unsigned first = lvl ? g.lv[lvl - 1].sblk_end : 0u;
    const unsigned b = blockIdx.x - first, per_list = (g.lv[lvl].sblk_end - first) / F_LISTS;
    const int list = (int)(b % F_LISTS), tid = (int)(b / F_LISTS) * 256 + threadIdx.x, nthreads = (int)per_list * 256;
    if (g.lv[lvl].W) rescore_pairs<true>(g.lv[lvl], list, tid, nthreads);
    else rescore_pairs<false>(g.lv[lvl], list, tid, nthreads);
}

// The exact kernel's distance of ONE given code per row on the vector ALU (tests ask for the winning distance; rows the filter
// decided never went through the exact kernel): the same fmaf chains in the same order (file header, "mfma8").
__global__ __launch_bounds__(256) void vq_exact_dist_kernel(const unsigned short* __restrict__ x, const float* __restrict__ E4, int Kp,
                                                            const float* __restrict__ enorm, const long long* __restrict__ idx, long N, int C,
                                                            float* __restrict__ dmin) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const unsigned short* xr = x + row * (long)C;
    const long long k = idx[row];
    float dot = 0.0f, xa = 0.0f, xb = 0.0f;
    for (int c = 0; c + 8 <= C; c += 8) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = __builtin_bit_cast(float, (unsigned)xr[c + e] << 16);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(E4 + ((size_t)(c / 4) * Kp + k) * 4);          // channels c .. c + 3 of code k
        const f32x4 e1 = *reinterpret_cast<const f32x4*>(E4 + ((size_t)(c / 4 + 1) * Kp + k) * 4);      // channels c + 4 .. c + 7
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dot = __builtin_fmaf(xv[e], e0[e], dot);
            dot = __builtin_fmaf(xv[4 + e], e1[e], dot);
            xa = __builtin_fmaf(xv[e], xv[e], xa);
            xb = __builtin_fmaf(xv[4 + e], xv[4 + e], xb);
        }
    }
    float d2 = __builtin_fmaf(-2.0f, dot, xa + xb) + enorm[k];
    d2 = __builtin_fmaxf(d2, 0.0f);
    dmin[row] = __builtin_sqrtf(d2);
}

// r4: everything a grouped launch wants pre-set, in ONE launch instead of up to three memsets per level (each a launch of its own, in a
// phase where the chip runs nothing else): keys = all ones (the atomicMin identity), the filter's sub-list counters + overflow flag = 0,
// the usage histogram = 0.
struct VqInitLevel {
    unsigned long long* keys;
    long n;
    int* amb;                   // 128 ints, or null
    int* hist;
    int kp;
    unsigned end;               // cumulative workgroup count
};
struct VqInitGroup {
    VqInitLevel lv[VQ_MAX_LEVELS];
    int n;
};
__global__ __launch_bounds__(256) void vq_group_init_kernel(const VqInitGroup g) {
    int q = 0;
    while (q + 1 < g.n && blockIdx.x >= g.lv[q].end) ++q;
    const VqInitLevel& L = g.lv[q];
    const unsigned first = q ? g.lv[q - 1].end : 0u, nb = L.end - first, b = blockIdx.x - first;
    for (long i = (long)b * 256 + threadIdx.x; i < L.n; i += (long)nb * 256) L.keys[i] = ~0ull;
    if (b == 0) {
        if (L.amb && threadIdx.x < 128) L.amb[threadIdx.x] = 0;
        for (int k = threadIdx.x; k < L.kp; k += 256) L.hist[k] = 0;
    }
}

// keys -> int64 code indices (+ the winning distance), and the code histogram of the dead-code statistic (vq_img.py:173-175):
// counted per workgroup in LDS (few codes win most rows -- thousands of rows per address: global atomics would serialise in the
// L2 atomic units; round 1 did that from the gather kernel and sat at ~1 TB/s), then one global add per NON-EMPTY bin and
// workgroup.  Integer adds: order independent, deterministic.
__global__ __launch_bounds__(256) void vq_unpack_keys(const unsigned long long* __restrict__ keys, long N,
                                                      long long* __restrict__ idx, float* __restrict__ dmin,
                                                      int* __restrict__ hist, int K, int lds_hist) {
    extern __shared__ int lh[];
    if (hist && lds_hist)
        for (int k = threadIdx.x; k < K; k += 256) lh[k] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        const unsigned long long k = keys[i];
        const unsigned code = (unsigned)(k & 0xffffffffull);
        idx[i] = (long long)code;
        if (dmin) dmin[i] = __uint_as_float((unsigned int)(k >> 32));
        if (hist && code < (unsigned)K) atomicAdd((lds_hist ? lh : hist) + code, 1);   // K beyond the LDS budget: global bins
    }
    if (hist && lds_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += 256) {
            const int c = lh[k];
            if (c) atomicAdd(hist + k, c);
        }
    }
}

// ------------------------------------------------------------------------------------
// gather + straight-through + commitment partial sums + code histogram (HBM-bound)
// ------------------------------------------------------------------------------------
constexpr int GATHER_BLOCKS_MAX = 2048;
constexpr int GATHER_ROWS_PER_BLOCK = 16;   // 4 waves x 4 rows in flight

template <typename TA>
__device__ __forceinline__ f32x4 ld4(const TA* p, long v) {                 // elements 4v .. 4v+3 as floats
    if constexpr (sizeof(TA) == 4) {
        return reinterpret_cast<const f32x4*>(p)[v];
    } else {
        const u32x2 r = reinterpret_cast<const u32x2*>(p)[v];
        return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xFFFF0000u),
                     __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xFFFF0000u)};
    }
}
template <typename TA>
__device__ __forceinline__ void st4(TA* p, long v, const f32x4& q) {
    if constexpr (sizeof(TA) == 4) {
        reinterpret_cast<f32x4*>(p)[v] = q;
    } else {
        u32x2 r;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const __bf16 lo = (__bf16)q[2 * e], hi = (__bf16)q[2 * e + 1];
            r[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        reinterpret_cast<u32x2*>(p)[v] = r;
    }
}

template <typename TA>
__global__ __launch_bounds__(256) void vq_gather_kernel(const TA* __restrict__ x, const float* __restrict__ W,
                                                        const long long* __restrict__ idx, long N, int C,
                                                        int training, TA* __restrict__ quant,
                                                        float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = C >> 2;
    float sq = 0.0f;
    // one wave per group of RG consecutive rows (their loads are all issued before the first use), grid-stride over groups.
    // (Per-lane squared-error partials accumulate in a fixed (group, chunk, row) order: deterministic.)
    constexpr int RG = 4;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RG; row0 < N; row0 += (long)gridDim.x * 4 * RG) {
        long long k[RG];
#pragma unroll
        for (int j = 0; j < RG; ++j) k[j] = row0 + j < N ? idx[row0 + j] : 0;
        for (int v = lane; v < c4; v += 64) {
            f32x4 e[RG], xv[RG];
#pragma unroll
            for (int j = 0; j < RG; ++j) {
                if (row0 + j < N) {
                    e[j] = reinterpret_cast<const f32x4*>(W + k[j] * (long)C)[v];
                    if (training) xv[j] = ld4<TA>(x + (row0 + j) * (long)C, v);
                }
            }
#pragma unroll
            for (int j = 0; j < RG; ++j) {
                if (row0 + j < N) {
                    TA* qr = quant + (row0 + j) * (long)C;
                    if (training) {
                        f32x4 q;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            q[i] = xv[j][i] + (e[j][i] - xv[j][i]);   // vq_img.py:236, fp32
                            const float dlt = q[i] - xv[j][i];
                            sq = __builtin_fmaf(dlt, dlt, sq);
                        }
                        st4<TA>(qr, v, q);
                    } else {
                        st4<TA>(qr, v, e[j]);
                    }
                }
            }
        }
    }
    // deterministic block reduction (fixed shuffle tree, fixed wave order)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    __shared__ float wsum[4];
    if (lane == 0) wsum[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ partial, int n_partial,
                                                          const int* __restrict__ hist, int K, int training,
                                                          float commitment_weight, double numel,
                                                          float* __restrict__ loss, float* __restrict__ dead_pct) {
    __shared__ double sd[256];
    __shared__ int sz[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += (double)partial[i];
    int z = 0;
    for (int k = threadIdx.x; k < K; k += 256) z += (hist[k] == 0);
    sd[threadIdx.x] = s;
    sz[threadIdx.x] = z;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) {
            sd[threadIdx.x] += sd[threadIdx.x + m];
            sz[threadIdx.x] += sz[threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float l = 0.0f;
        if (training && commitment_weight > 0.0f) l = (float)(sd[0] / numel) * commitment_weight;
        loss[0] = l;
        dead_pct[0] = 100.0f * ((float)sz[0] / (float)K);       // vq_img.py:174-175
    }
}

__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ gq, const float* __restrict__ gloss,
                                                          const float* __restrict__ x, const float* __restrict__ q,
                                                          long n4, float coef, float* __restrict__ gx) {
    const float k = gloss ? gloss[0] * coef : 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(gq)[i];
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 qv = reinterpret_cast<const f32x4*>(q)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(k, xv[j] - qv[j], g[j]);
        reinterpret_cast<f32x4*>(gx)[i] = o;
    }
}

// bf16 activations: the quantised rows are not kept for backward (their bf16 image would cost the commitment gradient
// its accuracy); the code index is, and e = W[idx] is re-read in fp32:  gx = g + k (x - e)
__global__ __launch_bounds__(256) void vq_backward_idx_kernel(const __bf16* __restrict__ gq, const float* __restrict__ gloss,
                                                              const __bf16* __restrict__ x, const long long* __restrict__ idx,
                                                              const float* __restrict__ W, long N, int C, float coef,
                                                              __bf16* __restrict__ gx) {
    const float k = gloss ? gloss[0] * coef : 0.0f;
    const int c4 = C >> 2;
    const long n4 = N * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / c4;
        const int v = (int)(i - row * c4);
        const f32x4 g = ld4<__bf16>(gq, i), xv = ld4<__bf16>(x, i);
        const f32x4 e = reinterpret_cast<const f32x4*>(W + idx[row] * (long)C)[v];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(k, xv[j] - e[j], g[j]);
        st4<__bf16>(gx, i, o);
    }
}

// ------------------------------------------------------------------------------------
// k-means update step: per-cluster sums of the member rows, deterministic and insensitive to skewed cluster sizes.
//   1. km_hist:    row blocks of KM_RB rows -> hist[block][k] (LDS histogram, no global atomics)
//   2. km_scan:    counts[k], member-list offsets[k], per-(block, k) write cursors, and segment offsets
//                  (a cluster's member list is cut into segments of KM_SEG members)
//   3. km_lists:   one wave per row block walks its rows in order and appends them to the member lists (stable:
//                  lists are in row order)
//   4. km_segsum:  block = (segment, 64-channel group): 4 waves stride the segment's members, combine in wave order
//   5. km_sums:    per cluster, its segment partials are added in segment order
// ------------------------------------------------------------------------------------
constexpr int KM_RB = 1024;    // rows per block of the histogram / list passes
constexpr int KM_SEG = 128;    // members per segment of the sum pass

__global__ __launch_bounds__(256) void km_hist_kernel(const long long* __restrict__ idx, long N, int K, int* __restrict__ hist) {
    extern __shared__ int lh[];
    for (int k = threadIdx.x; k < K; k += 256) lh[k] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * KM_RB;
    for (int i = threadIdx.x; i < KM_RB; i += 256)
        if (r0 + i < N) atomicAdd(&lh[idx[r0 + i]], 1);
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) hist[(long)blockIdx.x * K + k] = lh[k];
}

// single block: counts, exclusive scans (member offsets, segment offsets), per-block cursors (in place of hist)
__global__ __launch_bounds__(1024) void km_scan_kernel(int* __restrict__ hist, int n_blocks, int K, int* __restrict__ counts,
                                                       int* __restrict__ offsets, int* __restrict__ segoff) {
    __shared__ int part[1024], spart[1024];
    const int per = (K + 1023) / 1024;
    const int b = threadIdx.x * per;
    int s = 0, ss = 0;
    for (int i = 0; i < per; ++i)
        if (b + i < K) {
            int c = 0;
            for (int j = 0; j < n_blocks; ++j) c += hist[(long)j * K + b + i];
            counts[b + i] = c;
            s += c;
            ss += (c + KM_SEG - 1) / KM_SEG;
        }
    part[threadIdx.x] = s;
    spart[threadIdx.x] = ss;
    __syncthreads();
    for (int m = 1; m < 1024; m <<= 1) {
        const int v = (threadIdx.x >= m) ? part[threadIdx.x - m] : 0;
        const int w = (threadIdx.x >= m) ? spart[threadIdx.x - m] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        spart[threadIdx.x] += w;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s, srun = spart[threadIdx.x] - ss;
    for (int i = 0; i < per; ++i)
        if (b + i < K) {
            const int c = counts[b + i];
            offsets[b + i] = run;
            segoff[b + i] = srun;
            int cur = run;                                  // cursor of each row block inside this cluster's list
            for (int j = 0; j < n_blocks; ++j) {
                const int h = hist[(long)j * K + b + i];
                hist[(long)j * K + b + i] = cur;
                cur += h;
            }
            run += c;
            srun += (c + KM_SEG - 1) / KM_SEG;
        }
    if (threadIdx.x == 1023) {
        offsets[K] = part[1023];
        segoff[K] = spart[1023];
    }
}

__global__ __launch_bounds__(64) void km_lists_kernel(const long long* __restrict__ idx, long N, int K,
                                                      const int* __restrict__ cursors, int* __restrict__ members) {
    // one wave per row block; rows in order, 64 at a time; equal clusters inside a chunk are ranked by lane
    extern __shared__ int cur[];
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) cur[k] = cursors[(long)blockIdx.x * K + k];
    __syncthreads();
    const long r0 = (long)blockIdx.x * KM_RB;
    for (int base = 0; base < KM_RB && r0 + base < N; base += 64) {
        const long i = r0 + base + lane;
        const bool valid = i < N;
        const int k = valid ? (int)idx[i] : -1;
        bool todo = valid;
        while (__ballot(todo)) {
            const int lead = __ffsll((long long)__ballot(todo)) - 1;
            const int kk = __shfl(k, lead);
            const unsigned long long same = __ballot(todo && k == kk);
            if (todo && k == kk) {
                members[cur[kk] + __popcll(same & ((1ull << lane) - 1ull))] = (int)i;
                todo = false;
            }
            __syncthreads();                                // (one wave) order the cursor update after the reads
            if (lane == lead) cur[kk] += __popcll(same);
            __syncthreads();
        }
    }
}

template <typename TS>
__global__ __launch_bounds__(256) void km_segsum_kernel(const TS* __restrict__ samples, int C, int K,
                                                        const int* __restrict__ offsets, const int* __restrict__ segoff,
                                                        const int* __restrict__ members, float* __restrict__ partial) {
    const int seg = blockIdx.x;
    if (seg >= segoff[K]) return;
    int lo = 0, hi = K;                                     // cluster of this segment: last k with segoff[k] <= seg
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (segoff[mid] <= seg) lo = mid;
        else hi = mid;
    }
    const int k = lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int b = offsets[k] + (seg - segoff[k]) * KM_SEG;
    int e = b + KM_SEG;
    if (e > offsets[k + 1]) e = offsets[k + 1];
    float s = 0.0f;
    if (c < C)
        for (int m = b + wave; m < e; m += 4) s += (float)samples[(size_t)members[m] * C + c];
    __shared__ float sh[4][64];
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) partial[(size_t)seg * C + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}

__global__ __launch_bounds__(256) void km_sums_kernel(const float* __restrict__ partial, int C, const int* __restrict__ segoff,
                                                      float* __restrict__ sums) {
    // block = (cluster, 64-channel group); 4 thread rows stride the cluster's segments, folded in row order
    __shared__ float sh[4][64];
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, rowl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    float s = 0.0f;
    if (c < C)
        for (int g = segoff[k] + rowl; g < segoff[k + 1]; g += 4) s += partial[(size_t)g * C + c];
    sh[rowl][lane] = s;
    __syncthreads();
    if (rowl == 0 && c < C) sums[(size_t)k * C + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}

__global__ __launch_bounds__(256) void km_counts64_kernel(const int* __restrict__ counts, int K, long long* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) out[k] = counts[k];
}

__global__ __launch_bounds__(256) void km_finalize_kernel(const float* __restrict__ sums, const long long* __restrict__ counts,
                                                          float* __restrict__ means, int C) {
    const int k = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long n = counts[k];
    if (n > 0) means[(size_t)k * C + c] = sums[(size_t)k * C + c] / (float)n;      // vq_img.py:53; :58-61 keep if empty
}

// ---- opt-in EMA codebook update (an EXTENSION: the reference keeps `decay`/`eps` but never updates the codebook).
// The published rule the reference's module descends from (vector-quantize-pytorch EuclideanCodebook.forward):
//   cluster_size <- d cluster_size + (1-d) counts;  embed_avg <- d embed_avg + (1-d) sums;
//   codebook     <- embed_avg / ((cluster_size + eps) / (sum cluster_size + K eps) * sum cluster_size)
__global__ __launch_bounds__(256) void ema_counts_kernel(float* __restrict__ cluster_size, const long long* __restrict__ counts, int K,
                                                         float decay, float* __restrict__ total) {
    // one workgroup: update the K moving counts and leave their sum (fixed tree order) in *total
    __shared__ float sh[256];
    float s = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = __builtin_fmaf(cluster_size[k], decay, (1.0f - decay) * (float)counts[k]);
        cluster_size[k] = v;
        s += v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = sh[0];
}

__global__ __launch_bounds__(256) void ema_embed_kernel(float* __restrict__ embed_avg, const float* __restrict__ sums,
                                                        const float* __restrict__ cluster_size, const float* __restrict__ total,
                                                        float* __restrict__ codebook, int K, int C, float decay, float eps) {
    const int k = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const float n = *total;
    const float smoothed = (cluster_size[k] + eps) / (n + (float)K * eps) * n;
    const size_t o = (size_t)k * C + c;
    const float avg = __builtin_fmaf(embed_avg[o], decay, (1.0f - decay) * sums[o]);
    embed_avg[o] = avg;
    codebook[o] = avg / smoothed;
}

// ------------------------------------------------------------------------------------
// host-side launch helpers (called by the C ABI in vqseg_abi.cpp)
// ------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// the candidate filter serves bf16 rows of layers with whole 256-code chunks and 32-channel stages
static bool filter_shape_ok(int C, int K) { return K % 256 == 0 && C % 32 == 0 && K <= 65536; }

PreparedLayout prepared_layout(int C, int K) {
    const int Kp = round_up(K, 32);
    PreparedLayout l;
    l.off_e4 = 0;
    l.off_enorm = (size_t)C * Kp * sizeof(float);               // (kept adjacent to E4: the exact kernel's blob of rounds 1-3)
    l.off_enmax = up256(l.off_enorm + (size_t)Kp * sizeof(float));
    l.off_eb = l.off_enmax + 256;
    l.bytes = filter_shape_ok(C, K) ? up256(l.off_eb + (size_t)2 * C * Kp * sizeof(unsigned short)) : l.off_eb;
    return l;
}

size_t prepared_bytes(int C, int K) { return prepared_layout(C, K).bytes; }

static int g_vq_max_tiles = 8;                              // cap on T (option "vq_max_tiles_per_wave": 8, 4, 2 or 1)
static int g_vq_fine_split = 1;                             // option "vq_fine_split": 0 = r3's choice of T (see vq_group_tiles)
static int g_vq_bf16_filter = 1;                            // bf16 rows: candidate filter + exact re-score (0: the exact kernel on every row)
static int g_vq_filter_force_all = 0;                       // tests: 1 = every row is re-scored by the exact kernel (the filter decides nothing)
static int g_vq_filter_launches = 0;                        // launches that took the filter path (tests read and reset it)
int vq_set_option(const char* key, int value) {
    if (key && !strcmp(key, "vq_max_tiles_per_wave") && (value == 8 || value == 4 || value == 2 || value == 1)) {
        const int prev = g_vq_max_tiles;
        g_vq_max_tiles = value;
        return prev;
    }
    if (key && !strcmp(key, "vq_fine_split")) {
        const int prev = g_vq_fine_split;
        g_vq_fine_split = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "vq_bf16_filter")) {
        const int prev = g_vq_bf16_filter;
        g_vq_bf16_filter = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "vq_filter_force_all")) {
        const int prev = g_vq_filter_force_all;
        g_vq_filter_force_all = value ? 1 : 0;
        return prev;
    }
    if (key && !strcmp(key, "vq_filter_launches")) {
        const int prev = g_vq_filter_launches;
        g_vq_filter_launches = value;
        return prev;
    }
    return -1;
}

VqPlan vq_plan(int64_t N, int C, int K) {
    VqPlan p;
    p.Cp = C;
    p.Kp = round_up(K, 32);
    size_t off = 0;
    p.off_prepared = off;
    off += prepared_bytes(C, K);
    p.off_keys = off;
    off += ((size_t)N * sizeof(unsigned long long) + 255) & ~(size_t)255;
    p.off_hist = off;
    off += ((size_t)p.Kp * sizeof(int) + 255) & ~(size_t)255;
    p.off_partial = off;
    off += (size_t)GATHER_BLOCKS_MAX * sizeof(float);
    off = up256(off);
    p.off_summary = p.off_brow = p.off_amb = p.off_pair_rc = off;
    p.pair_cap = 0;
    if (filter_shape_ok(C, K)) {
        p.off_summary = off;
        off += up256((size_t)N * (p.Kp / F_SUB) * sizeof(unsigned long long));
        p.off_brow = off;
        off += up256((size_t)N * sizeof(float));
        p.off_amb = off;                                        // the pair sub-lists' counters [F_LISTS] + the overflow flag
        off += 512;
        const long total_cap = 4 * N < (1L << 30) ? 4 * N : (1L << 30);
        p.pair_cap = (int)((total_cap + F_LISTS - 1) / F_LISTS);     // per sub-list: four candidates per row on average
        if (p.pair_cap < 1024) p.pair_cap = 1024;
        p.off_pair_rc = off;
        off += up256((size_t)p.pair_cap * F_LISTS * sizeof(unsigned long long));
    }
    p.bytes = (off + 255) & ~(size_t)255;
    long blocks = (N + GATHER_ROWS_PER_BLOCK - 1) / GATHER_ROWS_PER_BLOCK;
    p.gather_blocks = (int)(blocks < GATHER_BLOCKS_MAX ? (blocks > 0 ? blocks : 1) : GATHER_BLOCKS_MAX);
    // tiles per wave: see vq_group_tiles
    const int k1 = K;
    p.T = vq_group_tiles(1, &N, &k1);
    return p;
}

hipError_t launch_prepare(const float* W, int K, int C, void* prepared, hipStream_t st) {
    const int Kp = round_up(K, 32);
    const PreparedLayout pl = prepared_layout(C, K);
    char* base = reinterpret_cast<char*>(prepared);
    float* E4 = reinterpret_cast<float*>(base + pl.off_e4);
    float* en = reinterpret_cast<float*>(base + pl.off_enorm);
    long total = (long)(C / 4) * Kp;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_pack_codebook, dim3((unsigned)blocks), dim3(256), 0, st, W, K, C, E4, Kp, C);
    hipLaunchKernelGGL(vq_code_norms, dim3((Kp + 63) / 64), dim3(64), 0, st, W, K, C, Kp, en);
    if (filter_shape_ok(C, K)) {                              // the candidate filter's image of the same codebook
        long b2 = ((long)(C / 8) * Kp + 255) / 256;
        if (b2 > 4096) b2 = 4096;
        hipLaunchKernelGGL(vq_pack_codebook_bf16, dim3((unsigned)b2), dim3(256), 0, st, W, K, C, Kp,
                           reinterpret_cast<unsigned short*>(base + pl.off_eb));
        hipLaunchKernelGGL(vq_enorm_max, dim3(1), dim3(256), 0, st, en, K, reinterpret_cast<float*>(base + pl.off_enmax));
    }
    return hipGetLastError();
}

// ---- optional per-launch timing of the assign kernel (bench.py's roofline leg): event pairs on the launch stream
static Profile g_prof;

hipError_t profile_begin(int capacity) {
    profile_release();
    g_prof.ev.resize((size_t)capacity * 2);
    for (auto& e : g_prof.ev) {
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return rc;
    }
    g_prof.shape.clear();
    g_prof.capacity = capacity;
    g_prof.enabled = true;
    return hipSuccess;
}

void profile_release() {
    for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.shape.clear();
    g_prof.enabled = false;
    g_prof.capacity = 0;
}

int profile_collect(int max_records, int64_t* n, int* c, int* k, float* ms, int* kind) {
    g_prof.enabled = false;
    const int cnt = (int)g_prof.shape.size();
    int out = 0;
    for (int i = 0; i < cnt && out < max_records; ++i) {
        const ProfShape& sh = g_prof.shape[i];
        if (hipEventSynchronize(g_prof.ev[2 * sh.slot + 1]) != hipSuccess) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * sh.slot], g_prof.ev[2 * sh.slot + 1]) != hipSuccess) break;
        double mine = 2.0 * (double)sh.n * sh.c * sh.k, all = 0.0;   // a grouped launch: this level's share of the flops
        for (int j = sh.slot; j < sh.slot + sh.group && j < cnt; ++j) all += 2.0 * (double)g_prof.shape[j].n * g_prof.shape[j].c * g_prof.shape[j].k;
        n[out] = sh.n;
        c[out] = sh.c;
        k[out] = sh.k;
        ms[out] = (float)(t * (all > 0 ? mine / all : 1.0));
        if (kind) kind[out] = sh.kind;
        ++out;
    }
    profile_release();
    return out;
}

#if VQ_TIMELINE
static unsigned long long* g_timeline = nullptr;
#endif

template <int T, typename TX>
static void launch_assign_t(const VqGroup& g, hipStream_t st) {
    constexpr int STAGE_FLOATS = stage_floats(T);
    static_assert((size_t)WAVES * 32 * 33 * sizeof(unsigned long long) <= (2 * STAGE_FLOATS + 256) * sizeof(float), "key scratch aliases the B stages");
    const size_t lds = (size_t)(2 * STAGE_FLOATS + 256 + WAVES * 32) * sizeof(float);
    hipLaunchKernelGGL((vq_assign_f32_kernel<T, TX>), dim3(g.lv[g.n - 1].wg_end), dim3(256), lds, st, g);
}

static unsigned assign_workgroups(int64_t N, int Kp, int T) {
    const long row_tiles = (N + ROWS_PER_WG - 1) / ROWS_PER_WG;
    return (unsigned)((row_tiles + 7) / 8 * 8 * (Kp / (32 * T)));               // see the id mapping in the kernel
}

// tiles per wave for a group of levels.  r4: first the largest T in {8,4} that divides every level's tile count and gives the launch
// >= VQ_FINE_WGS workgroups -- the workgroups of a launch differ 4x in length between the levels, and with fewer, longer ones the
// last round leaves the chip half empty (K = 512 on 172 k rows: T = 8 -> 2688 workgroups, 0.76 of the fp32 MFMA peak; T = 4 -> 5376,
// 0.81; K = 256 / K = 1024 / twice the rows: T = 8 stays best, T = 2 loses 9 % of the per-wave rate: LEDGER r4).  Else, as before, the
// largest T in {8,4,2,1} that still yields >= 2 workgroups per CU over the whole launch; if none does, the smallest.
constexpr long VQ_FINE_WGS = 4096;
static long vq_group_wgs(int n, const int64_t* N, const int* K, int t) {     // workgroups of the launch at T = t; -1: t does not divide
    long wgs = 0;
    for (int i = 0; i < n; ++i) {
        const int tiles = round_up(K[i], 32) / 32;
        if (tiles % t) return -1;
        wgs += ((N[i] + ROWS_PER_WG - 1) / ROWS_PER_WG) * (tiles / t);
    }
    return wgs;
}
int vq_group_tiles(int n, const int64_t* N, const int* K) {
    if (g_vq_fine_split)
        for (int t = g_vq_max_tiles; t >= 4; t >>= 1)
            if (vq_group_wgs(n, N, K, t) >= VQ_FINE_WGS) return t;
    for (int t = g_vq_max_tiles; t >= 1; t >>= 1) {
        bool ok = true;
        long wgs = 0;
        for (int i = 0; i < n; ++i) {
            const int tiles = round_up(K[i], 32) / 32;
            if (tiles % t) ok = false;
            else wgs += ((N[i] + ROWS_PER_WG - 1) / ROWS_PER_WG) * (tiles / t);
        }
        if (ok && (wgs >= 512 || t == 1)) return t;
    }
    return 1;
}

// the distance + argmin pass of n <= VQ_MAX_LEVELS levels in ONE launch (keys pre-set, unpack per level afterwards)
hipError_t launch_assign_group(int n, const void* const* x, int x_bf16, const int64_t* N, const int* C, const int* K,
                               const void* const* prepared, const VqPlan* plans, char* const* ws, int64_t* const* idx,
                               float* const* dmin, int T, hipStream_t st, const float* const* codebooks) {
    if (n < 1 || n > VQ_MAX_LEVELS) return hipErrorInvalidValue;
    int order[VQ_MAX_LEVELS];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)                                  // longest workgroups (most channels) first
        for (int j = i + 1; j < n; ++j)
            if (C[order[j]] > C[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    // bf16 rows of layers with whole 256-code chunks take the candidate filter: bf16 MFMA scores -> per-row decision -> the exact
    // kernel (indirect mode) on the rows the filter left open.  Same keys as the exact kernel on every row.
    bool filter = x_bf16 && g_vq_bf16_filter;
    for (int i = 0; i < n; ++i) filter = filter && filter_shape_ok(C[i], K[i]);
    VqGroup g;
    VqFilterGroup fg;
    FilterConsts fc[VQ_MAX_LEVELS] = {};
    g.n = fg.n = n;
#if VQ_TIMELINE
    g.tl = g_timeline;
#endif
    unsigned end = 0, fend = 0, rend = 0, send = 0, iend = 0;
    VqInitGroup ig;
    ig.n = n;
    for (int q = 0; q < n; ++q) {
        const int i = order[q];
        const PreparedLayout pl = prepared_layout(C[i], K[i]);
        const char* pb = reinterpret_cast<const char*>(prepared[i]);
        const float* E4 = reinterpret_cast<const float*>(pb + pl.off_e4);
        const float* enorm = reinterpret_cast<const float*>(pb + pl.off_enorm);
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_keys);
        {
            long ib = (N[i] + 4095) / 4096;                      // 16 keys per thread
            ib = ib < 1 ? 1 : (ib > 256 ? 256 : ib);
            iend += (unsigned)ib;
            ig.lv[q] = VqInitLevel{keys, (long)N[i], nullptr, reinterpret_cast<int*>(ws[i] + plans[i].off_hist), plans[i].Kp, iend};
        }
        end += assign_workgroups(N[i], plans[i].Kp, T);
        g.lv[q] = VqLevel{x[i], E4, enorm, keys, (long)N[i], C[i], plans[i].Kp, end};
        if (filter) {
            int* amb = reinterpret_cast<int*>(ws[i] + plans[i].off_amb);
            ig.lv[q].amb = amb;                                   // the sub-lists' counters and the overflow flag: zeroed by the init launch
            // a pair list that overflows (more than four candidates per row on average: degenerate codebooks) switches the level to the
            // exact kernel on every row -- the gated launch below
            g.lv[q].gate = amb + F_LISTS;                         // the overflow flag
            g.lv[q].gate_cap = 0;
            const long row_tiles = (N[i] + F_ROWS - 1) / F_ROWS;
            fend += (unsigned)((row_tiles + 7) / 8 * 8 * (plans[i].Kp / F_CODES));
            rend += (unsigned)((N[i] + 255) / 256);
            {   // stage 3: a thread per expected pair (about one per row at most), whole multiples of F_LISTS workgroups, 4 .. 16 per sub-list
                long per = (N[i] / 256 + F_LISTS - 1) / F_LISTS;
                if (per < 4) per = 4;
                if (per > 16) per = 16;
                send += (unsigned)(per * F_LISTS);
            }
            fg.lv[q] = VqFilterLevel{x[i], reinterpret_cast<const unsigned short*>(pb + pl.off_eb), enorm,
                                     reinterpret_cast<const float*>(pb + pl.off_enmax),
                                     reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_summary),
                                     reinterpret_cast<float*>(ws[i] + plans[i].off_brow),
                                     reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_pair_rc), amb, plans[i].pair_cap, keys,
                                     codebooks ? codebooks[i] : nullptr, E4, (long)N[i], C[i], plans[i].Kp, fend, rend, send};
            // BAND = 1.25 * [ 2 eps + 2^-20 (|x|^2 + E + 2 S) ],  eps = 2 (C 2^-24 + 2^-18 + (C / 8) 17 2^-24) S + 2^-22 (|x|^2 + E + 2 S),
            // S = sqrt(|x|^2 E)   (derivation at vq_filter_bf16_kernel)
            const double u = ldexp(1.0, -24), cc = (double)C[i];
            const double per_s = 2.0 * (cc * u + ldexp(1.0, -18) + (cc / 8.0) * 17.0 * u);
            const double tail = 2.0 * ldexp(1.0, -22) + ldexp(1.0, -20);
            fc[q].c_s = (float)(1.25 * (2.0 * per_s + 2.0 * tail));
            fc[q].c_r = (float)(1.25 * tail);
        }
    }
    static_assert(F_LISTS + 1 <= 128, "the init launch zeroes 128 ints of a level's counter block");
    hipLaunchKernelGGL(vq_group_init_kernel, dim3(iend), dim3(256), 0, st, ig);
    const bool rec = g_prof.enabled && (int)g_prof.shape.size() + n <= g_prof.capacity;
    const size_t slot = g_prof.shape.size();
    if (rec) {
        // one event pair for the launch; its time is apportioned to the levels by their flops (2 N K C) when collected
        for (int i = 0; i < n; ++i) g_prof.shape.push_back({N[i], C[i], K[i], (int)slot, n, filter ? 2 : (x_bf16 ? 1 : 0)});
        (void)hipEventRecord(g_prof.ev[2 * slot], st);
    }
    if (filter) {
        ++g_vq_filter_launches;
        hipLaunchKernelGGL(vq_filter_bf16_kernel, dim3(fend), dim3(256), (size_t)(2 * F_STAGE_BYTES + F_CODES * sizeof(float)), st, fg, fc[0],
                           fc[1], fc[2], fc[3]);
        hipLaunchKernelGGL(vq_resolve_kernel, dim3(rend), dim3(256), 0, st, fg, g_vq_filter_force_all);
        hipLaunchKernelGGL(vq_rescore_kernel, dim3(send), dim3(256), 0, st, fg);
    }
#define VQ_ASSIGN(T_)                                                 \
    if (x_bf16) launch_assign_t<T_, __bf16>(g, st);                   \
    else launch_assign_t<T_, float>(g, st)
    switch (T) {
        case 8: VQ_ASSIGN(8); break;
        case 4: VQ_ASSIGN(4); break;
        case 2: VQ_ASSIGN(2); break;
        default: VQ_ASSIGN(1); break;
    }
#undef VQ_ASSIGN
    if (rec) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int i = 0; i < n; ++i) {
        int* hist = reinterpret_cast<int*>(ws[i] + plans[i].off_hist);       // (zeroed by the init launch)
        long blocks = (N[i] + 1023) / 1024;                      // >= 1024 rows per workgroup: the LDS histogram pays
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        const int lds_hist = K[i] <= 8192;                       // 32 KB of LDS bins; larger codebooks count in global memory (rare
                                                                 // winners per address there, so the L2 atomics do not serialise)
        hipLaunchKernelGGL(vq_unpack_keys, dim3((unsigned)blocks), dim3(256), lds_hist ? (size_t)K[i] * sizeof(int) : 0, st,
                           reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_keys), (long)N[i],
                           reinterpret_cast<long long*>(idx[i]), dmin ? dmin[i] : nullptr, hist, K[i], lds_hist);
        if (filter && dmin && dmin[i]) {
            // rows the filter decided carry no exact distance: recompute the winner's on the vector ALU, in the exact kernel's order
            const PreparedLayout pl = prepared_layout(C[i], K[i]);
            const char* pb = reinterpret_cast<const char*>(prepared[i]);
            hipLaunchKernelGGL(vq_exact_dist_kernel, dim3((unsigned)((N[i] + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned short*>(x[i]),
                               reinterpret_cast<const float*>(pb + pl.off_e4), plans[i].Kp, reinterpret_cast<const float*>(pb + pl.off_enorm),
                               reinterpret_cast<const long long*>(idx[i]), (long)N[i], C[i], dmin[i]);
        }
    }
    return hipGetLastError();
}

hipError_t launch_assign(const void* x, int x_bf16, int64_t N, int C, int K, const void* prepared, const VqPlan& p, char* ws,
                         int64_t* idx, float* dmin, hipStream_t st, const float* codebook) {
    return launch_assign_group(1, &x, x_bf16, &N, &C, &K, &prepared, &p, &ws, &idx, &dmin, p.T, st, codebook ? &codebook : nullptr);
}

hipError_t launch_gather(const void* x, int bf16, const float* W, const int64_t* idx, int64_t N, int C, int K, int training,
                         float cw, const VqPlan& p, char* ws, void* quant, float* loss, float* dead, hipStream_t st) {
    int* hist = reinterpret_cast<int*>(ws + p.off_hist);
    float* partial = reinterpret_cast<float*>(ws + p.off_partial);
    // hist: filled by launch_assign (vq_unpack_keys), which every forward runs first on the same workspace
    if (bf16)
        hipLaunchKernelGGL(vq_gather_kernel<__bf16>, dim3(p.gather_blocks), dim3(256), 0, st, static_cast<const __bf16*>(x), W,
                           reinterpret_cast<const long long*>(idx), (long)N, C, training, static_cast<__bf16*>(quant), partial);
    else
        hipLaunchKernelGGL(vq_gather_kernel<float>, dim3(p.gather_blocks), dim3(256), 0, st, static_cast<const float*>(x), W,
                           reinterpret_cast<const long long*>(idx), (long)N, C, training, static_cast<float*>(quant), partial);
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partial, p.gather_blocks, hist, K, training, cw,
                       (double)N * (double)C, loss, dead);
    return hipGetLastError();
}

hipError_t launch_backward(const float* gq, const float* gloss, const float* x, const float* q, int64_t N, int C,
                           float cw, float* gx, hipStream_t st) {
    const long n4 = (long)N * C / 4;
    const float coef = (float)((double)cw * 2.0 / ((double)N * (double)C));
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vq_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gq, gloss, x, q, n4, coef, gx);
    return hipGetLastError();
}

KmPlan km_plan(int64_t N, int C, int K) {
    KmPlan p;
    p.vq = vq_plan(N, C, K);
    size_t off = p.vq.bytes;
    p.off_idx = off;
    off += (size_t)N * sizeof(int64_t);
    p.off_counts = off;
    off += (size_t)round_up(K + 1, 64) * sizeof(int);
    p.off_offsets = off;
    off += (size_t)round_up(K + 1, 64) * sizeof(int);
    p.off_members = off;
    off += (size_t)N * sizeof(int);
    p.off_sums = off;
    off += (size_t)K * C * sizeof(float);
    p.off_counts64 = off;
    off += (size_t)round_up(K, 32) * sizeof(int64_t);
    p.row_blocks = (int)((N + KM_RB - 1) / KM_RB);
    p.max_segments = (int)(N / KM_SEG + K);
    p.off_hist = off;
    off += (size_t)round_up(p.row_blocks * K, 64) * sizeof(int);
    p.off_segoff = off;
    off += (size_t)round_up(K + 1, 64) * sizeof(int);
    p.off_partial = off;
    off += (size_t)p.max_segments * C * sizeof(float);
    p.bytes = (off + 255) & ~(size_t)255;
    return p;
}

hipError_t launch_backward_idx(const void* gq, const float* gloss, const void* x, const int64_t* idx, const float* W, int64_t N,
                               int C, float cw, void* gx, hipStream_t st) {
    const long n4 = (long)N * C / 4;
    const float coef = (float)((double)cw * 2.0 / ((double)N * (double)C));
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vq_backward_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const __bf16*>(gq), gloss,
                       static_cast<const __bf16*>(x), reinterpret_cast<const long long*>(idx), W, (long)N, C, coef,
                       static_cast<__bf16*>(gx));
    return hipGetLastError();
}

template <typename TS>
static hipError_t code_sums_from_idx(const TS* samples, const int64_t* idx, int64_t N, int C, int K, const KmPlan& p, char* ws,
                                     float* sums, int64_t* counts64, hipStream_t st) {
    int* counts = reinterpret_cast<int*>(ws + p.off_counts);
    int* offsets = reinterpret_cast<int*>(ws + p.off_offsets);
    int* members = reinterpret_cast<int*>(ws + p.off_members);
    int* hist = reinterpret_cast<int*>(ws + p.off_hist);
    int* segoff = reinterpret_cast<int*>(ws + p.off_segoff);
    float* partial = reinterpret_cast<float*>(ws + p.off_partial);
    const long long* idx64 = reinterpret_cast<const long long*>(idx);
    hipLaunchKernelGGL(km_hist_kernel, dim3(p.row_blocks), dim3(256), (size_t)K * sizeof(int), st, idx64, (long)N, K, hist);
    hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(1024), 0, st, hist, p.row_blocks, K, counts, offsets, segoff);
    hipLaunchKernelGGL(km_lists_kernel, dim3(p.row_blocks), dim3(64), (size_t)K * sizeof(int), st, idx64, (long)N, K, hist, members);
    hipLaunchKernelGGL(km_segsum_kernel<TS>, dim3(p.max_segments, (C + 63) / 64), dim3(256), 0, st, samples, C, K, offsets, segoff,
                       members, partial);
    hipLaunchKernelGGL(km_sums_kernel, dim3(K, (C + 63) / 64), dim3(256), 0, st, partial, C, segoff, sums);
    hipLaunchKernelGGL(km_counts64_kernel, dim3((K + 255) / 256), dim3(256), 0, st, counts, K,
                       reinterpret_cast<long long*>(counts64));
    return hipGetLastError();
}

hipError_t launch_km_accumulate(const float* samples, const float* means, int64_t N, int C, int K, const KmPlan& p,
                                char* ws, float* sums, int64_t* counts64, hipStream_t st) {
    hipError_t e = launch_prepare(means, K, C, ws + p.vq.off_prepared, st);
    if (e != hipSuccess) return e;
    int64_t* idx = reinterpret_cast<int64_t*>(ws + p.off_idx);
    e = launch_assign(samples, 0, N, C, K, ws + p.vq.off_prepared, p.vq, ws, idx, nullptr, st);
    if (e != hipSuccess) return e;
    return code_sums_from_idx(samples, idx, N, C, K, p, ws, sums, counts64, st);
}

hipError_t launch_code_sums(const void* x, int x_bf16, const int64_t* idx, int64_t N, int C, int K, const KmPlan& p, char* ws,
                            float* sums, int64_t* counts64, hipStream_t st) {
    return x_bf16 ? code_sums_from_idx(static_cast<const __bf16*>(x), idx, N, C, K, p, ws, sums, counts64, st)
                  : code_sums_from_idx(static_cast<const float*>(x), idx, N, C, K, p, ws, sums, counts64, st);
}

hipError_t launch_ema_update(float* cluster_size, float* embed_avg, float* codebook, const float* sums, const int64_t* counts64,
                             int K, int C, float decay, float eps, float* total, hipStream_t st) {
    hipLaunchKernelGGL(ema_counts_kernel, dim3(1), dim3(256), 0, st, cluster_size, reinterpret_cast<const long long*>(counts64), K,
                       decay, total);
    hipLaunchKernelGGL(ema_embed_kernel, dim3(K, (C + 255) / 256), dim3(256), 0, st, embed_avg, sums, cluster_size, total, codebook,
                       K, C, decay, eps);
    return hipGetLastError();
}

hipError_t launch_km_finalize(const float* sums, const int64_t* counts64, float* means, int C, int K, hipStream_t st) {
    hipLaunchKernelGGL(km_finalize_kernel, dim3(K, (C + 255) / 256), dim3(256), 0, st, sums,
                       reinterpret_cast<const long long*>(counts64), means, C);
    return hipGetLastError();
}

}  // namespace vqseg

#if VQ_TIMELINE
// debug build only: device buffer [workgroups][16] u64 the next assign launches write their stamps to (null: off)
extern "C" int vqseg_debug_timeline(void* buf) {
    vqseg::g_timeline = static_cast<unsigned long long*>(buf);
    return 0;
}
#endif
