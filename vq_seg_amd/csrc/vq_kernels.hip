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
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b_cur[t][e], acc[t], 0, 0, 0);
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

size_t prepared_bytes(int C, int K) {
    const int Kp = round_up(K, 32);
    return (((size_t)C * Kp + Kp) * sizeof(float) + 255) & ~(size_t)255;
}

static int g_vq_max_tiles = 8;                              // cap on T (option "vq_max_tiles_per_wave": 8, 4, 2 or 1)
int vq_set_option(const char* key, int value) {
    if (key && !strcmp(key, "vq_max_tiles_per_wave") && (value == 8 || value == 4 || value == 2 || value == 1)) {
        const int prev = g_vq_max_tiles;
        g_vq_max_tiles = value;
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
    p.bytes = (off + 255) & ~(size_t)255;
    long blocks = (N + GATHER_ROWS_PER_BLOCK - 1) / GATHER_ROWS_PER_BLOCK;
    p.gather_blocks = (int)(blocks < GATHER_BLOCKS_MAX ? (blocks > 0 ? blocks : 1) : GATHER_BLOCKS_MAX);
    // tiles per wave: the largest T in {8,4,2,1} dividing Kp/32 that still yields >= 2 workgroups per CU;
    // if none does, the smallest (most workgroups).
    const long row_blocks = (N + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int tiles = p.Kp / 32;
    p.T = 1;
    for (int t = g_vq_max_tiles; t >= 1; t >>= 1) {
        if (tiles % t) continue;
        if (row_blocks * (tiles / t) >= 512 || t == 1) {
            p.T = t;
            break;
        }
    }
    return p;
}

hipError_t launch_prepare(const float* W, int K, int C, void* prepared, hipStream_t st) {
    const int Kp = round_up(K, 32);
    float* E4 = reinterpret_cast<float*>(prepared);
    float* en = E4 + (size_t)C * Kp;
    long total = (long)(C / 4) * Kp;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_pack_codebook, dim3((unsigned)blocks), dim3(256), 0, st, W, K, C, E4, Kp, C);
    hipLaunchKernelGGL(vq_code_norms, dim3((Kp + 63) / 64), dim3(64), 0, st, W, K, C, Kp, en);
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

int profile_collect(int max_records, int64_t* n, int* c, int* k, float* ms) {
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

// tiles per wave for a group of levels: the largest T in {8,4,2,1} that divides every level's tile count and still yields
// >= 2 workgroups per CU over the whole launch; if none does, the smallest.
int vq_group_tiles(int n, const int64_t* N, const int* K) {
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
                               float* const* dmin, int T, hipStream_t st) {
    if (n < 1 || n > VQ_MAX_LEVELS) return hipErrorInvalidValue;
    int order[VQ_MAX_LEVELS];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)                                  // longest workgroups (most channels) first
        for (int j = i + 1; j < n; ++j)
            if (C[order[j]] > C[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    VqGroup g;
    g.n = n;
#if VQ_TIMELINE
    g.tl = g_timeline;
#endif
    unsigned end = 0;
    for (int q = 0; q < n; ++q) {
        const int i = order[q];
        const float* E4 = reinterpret_cast<const float*>(prepared[i]);
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_keys);
        hipError_t e = hipMemsetAsync(keys, 0xff, (size_t)N[i] * sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
        end += assign_workgroups(N[i], plans[i].Kp, T);
        g.lv[q] = VqLevel{x[i], E4, E4 + (size_t)C[i] * plans[i].Kp, keys, (long)N[i], C[i], plans[i].Kp, end};
    }
    const bool rec = g_prof.enabled && (int)g_prof.shape.size() + n <= g_prof.capacity;
    const size_t slot = g_prof.shape.size();
    if (rec) {
        // one event pair for the launch; its time is apportioned to the levels by their flops (2 N K C) when collected
        for (int i = 0; i < n; ++i) g_prof.shape.push_back({N[i], C[i], K[i], (int)slot, n});
        (void)hipEventRecord(g_prof.ev[2 * slot], st);
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
        int* hist = reinterpret_cast<int*>(ws[i] + plans[i].off_hist);
        e = hipMemsetAsync(hist, 0, (size_t)plans[i].Kp * sizeof(int), st);
        if (e != hipSuccess) return e;
        long blocks = (N[i] + 1023) / 1024;                      // >= 1024 rows per workgroup: the LDS histogram pays
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        const int lds_hist = K[i] <= 8192;                       // 32 KB of LDS bins; larger codebooks count in global memory (rare
                                                                 // winners per address there, so the L2 atomics do not serialise)
        hipLaunchKernelGGL(vq_unpack_keys, dim3((unsigned)blocks), dim3(256), lds_hist ? (size_t)K[i] * sizeof(int) : 0, st,
                           reinterpret_cast<unsigned long long*>(ws[i] + plans[i].off_keys), (long)N[i],
                           reinterpret_cast<long long*>(idx[i]), dmin ? dmin[i] : nullptr, hist, K[i], lds_hist);
    }
    return hipGetLastError();
}

hipError_t launch_assign(const void* x, int x_bf16, int64_t N, int C, int K, const void* prepared, const VqPlan& p, char* ws,
                         int64_t* idx, float* dmin, hipStream_t st) {
    return launch_assign_group(1, &x, x_bf16, &N, &C, &K, &prepared, &p, &ws, &idx, &dmin, p.T, st);
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
