// vqseg_abi.hip -- extern "C" entry points declared in include/vqseg.h.
// Argument validation + workspace carving + kernel enqueue; never allocates, never syncs.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vqseg.h"
#include "vq_kernels.h"

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_shape(int64_t n, int c, int k) {
    if (n <= 0 || c <= 0 || k <= 0) return fail(VQSEG_EINVAL, "n_rows, channels and n_codes must be positive (got %lld, %d, %d)", (long long)n, c, k);
    if (c % 4) return fail(VQSEG_EINVAL, "channels must be a multiple of 4 (got %d)", c);
    if (n > (int64_t)1 << 31) return fail(VQSEG_EINVAL, "n_rows too large (%lld)", (long long)n);
    return 0;
}
}  // namespace

extern "C" {

int vqseg_set_error(int code, const char* msg) {      // shared with nn_abi.hip (not part of the public header)
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int vqseg_abi_version(void) { return VQSEG_ABI_VERSION; }
const char* vqseg_last_error(void) { return g_err; }

const char* vqseg_kernel_name(const char* entry) {
    if (!entry) return "";
    if (!strcmp(entry, "vqseg_vq_forward_f32") || !strcmp(entry, "vqseg_vq_assign_f32") ||
        !strcmp(entry, "vqseg_kmeans_f32") || !strcmp(entry, "vqseg_kmeans_accumulate_f32"))
        return "vq_assign_f32_kernel";
    if (!strcmp(entry, "vqseg_vq_backward_f32")) return "vq_backward_kernel";
    if (!strcmp(entry, "vqseg_kmeans_finalize_f32")) return "km_finalize_kernel";
    return "";
}

int vqseg_profile_begin(int capacity) {
    if (capacity <= 0 || capacity > (1 << 20)) return fail(VQSEG_EINVAL, "capacity out of range");
    hipError_t e = vqseg::profile_begin(capacity);
    if (e != hipSuccess) return hip_fail(e, "profile_begin");
    return 0;
}

int vqseg_profile_collect(int max_records, int64_t* n_rows_host, int* channels_host, int* n_codes_host, float* ms_host, int* kind_host) {
    if (max_records <= 0 || !n_rows_host || !channels_host || !n_codes_host || !ms_host)
        return fail(VQSEG_EINVAL, "bad argument");
    return vqseg::profile_collect(max_records, n_rows_host, channels_host, n_codes_host, ms_host, kind_host);
}

size_t vqseg_vq_workspace_bytes(int64_t n, int c, int k) {
    if (n <= 0 || c <= 0 || k <= 0) return 0;
    return vqseg::vq_plan(n, c, k).bytes;
}

size_t vqseg_vq_filter_counter_offset(int64_t n, int c, int k) {
    if (n <= 0 || c <= 0 || k <= 0) return 0;
    const vqseg::VqPlan p = vqseg::vq_plan(n, c, k);
    return p.off_amb == p.off_summary ? 0 : p.off_amb;          // 0: the shape does not take the candidate filter
}

size_t vqseg_vq_prepared_bytes(int c, int k) {
    if (c <= 0 || k <= 0) return 0;
    return vqseg::prepared_bytes(c, k);
}

int vqseg_vq_prepare_f32(const float* codebook, int c, int k, void* prepared, size_t prepared_bytes, void* stream) {
    if (int rc = check_shape(1, c, k)) return rc;
    if (!codebook || !prepared) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(codebook) || !aligned16(prepared)) return fail(VQSEG_EINVAL, "codebook and prepared must be 16-byte aligned");
    if (prepared_bytes < vqseg::prepared_bytes(c, k))
        return fail(VQSEG_ENOSPC, "prepared buffer %zu < %zu bytes", prepared_bytes, vqseg::prepared_bytes(c, k));
    hipError_t e = vqseg::launch_prepare(codebook, k, c, prepared, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "vq codebook prepare");
    return 0;
}

static int vq_assign_any(const void* x, int x_bf16, const float* codebook, const void* prepared, int64_t n, int c, int k,
                         int64_t* idx, float* dmin, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_shape(n, c, k)) return rc;
    if (!x || !idx || !ws || (!codebook && !prepared)) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(x) || !aligned16(ws) || !aligned16(prepared) || !aligned16(codebook))
        return fail(VQSEG_EINVAL, "x, codebook, prepared and workspace must be 16-byte aligned");
    const vqseg::VqPlan p = vqseg::vq_plan(n, c, k);
    if (ws_bytes < p.bytes) return fail(VQSEG_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, p.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(ws);
    if (!prepared) {
        hipError_t e = vqseg::launch_prepare(codebook, k, c, w + p.off_prepared, st);
        if (e != hipSuccess) return hip_fail(e, "vq codebook prepare");
        prepared = w + p.off_prepared;
    }
    hipError_t e = vqseg::launch_assign(x, x_bf16, n, c, k, prepared, p, w, idx, dmin, st, codebook);
    if (e != hipSuccess) return hip_fail(e, "vq_assign_f32_kernel");
    return 0;
}

int vqseg_vq_assign_f32(const float* x, const float* codebook, const void* prepared, int64_t n, int c, int k,
                        int64_t* idx, float* dmin, void* ws, size_t ws_bytes, void* stream) {
    return vq_assign_any(x, 0, codebook, prepared, n, c, k, idx, dmin, ws, ws_bytes, stream);
}

int vqseg_vq_assign_bf16(const void* x, const float* codebook, const void* prepared, int64_t n, int c, int k,
                         int64_t* idx, float* dmin, void* ws, size_t ws_bytes, void* stream) {
    if (c % 8) return fail(VQSEG_EINVAL, "bf16 rows need channels %% 8 == 0");
    return vq_assign_any(x, 1, codebook, prepared, n, c, k, idx, dmin, ws, ws_bytes, stream);
}

static int vq_forward_any(const void* x, int bf16, const float* codebook, const void* prepared, int64_t n, int c, int k,
                          int training, float cw, void* quant, int64_t* idx, float* loss, float* dead_pct, float* dmin,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!quant || !loss || !dead_pct || !codebook) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(quant)) return fail(VQSEG_EINVAL, "quant must be 16-byte aligned");
    if (int rc = vq_assign_any(x, bf16, codebook, prepared, n, c, k, idx, dmin, ws, ws_bytes, stream)) return rc;
    const vqseg::VqPlan p = vqseg::vq_plan(n, c, k);
    hipError_t e = vqseg::launch_gather(x, bf16, codebook, idx, n, c, k, training, cw, p, static_cast<char*>(ws), quant, loss,
                                        dead_pct, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "vq_gather_kernel");
    return 0;
}

int vqseg_vq_forward_f32(const float* x, const float* codebook, const void* prepared, int64_t n, int c, int k,
                         int training, float cw, float* quant, int64_t* idx, float* loss, float* dead_pct, float* dmin,
                         void* ws, size_t ws_bytes, void* stream) {
    return vq_forward_any(x, 0, codebook, prepared, n, c, k, training, cw, quant, idx, loss, dead_pct, dmin, ws, ws_bytes, stream);
}

int vqseg_vq_forward_bf16(const void* x, const float* codebook, const void* prepared, int64_t n, int c, int k,
                          int training, float cw, void* quant, int64_t* idx, float* loss, float* dead_pct, float* dmin,
                          void* ws, size_t ws_bytes, void* stream) {
    if (c % 8) return fail(VQSEG_EINVAL, "bf16 rows need channels %% 8 == 0");
    return vq_forward_any(x, 1, codebook, prepared, n, c, k, training, cw, quant, idx, loss, dead_pct, dmin, ws, ws_bytes, stream);
}

int vqseg_vq_forward_group(int n_levels, int bf16, const void* const* x, const float* const* codebook, const void* const* prepared,
                           const int64_t* n, const int* c, const int* k, int training, const float* cw, void* const* quant,
                           int64_t* const* idx, float* const* loss, float* const* dead_pct, void* const* ws, const size_t* ws_bytes,
                           void* stream) {
    if (n_levels < 1 || n_levels > vqseg::VQ_MAX_LEVELS) return fail(VQSEG_EINVAL, "1 .. %d levels per grouped launch", vqseg::VQ_MAX_LEVELS);
    if (!x || !codebook || !n || !c || !k || !cw || !quant || !idx || !loss || !dead_pct || !ws || !ws_bytes)
        return fail(VQSEG_EINVAL, "null pointer argument");
    vqseg::VqPlan plans[vqseg::VQ_MAX_LEVELS];
    const void* prep[vqseg::VQ_MAX_LEVELS];
    char* wsp[vqseg::VQ_MAX_LEVELS];
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n_levels; ++i) {
        if (int rc = check_shape(n[i], c[i], k[i])) return rc;
        if (bf16 && c[i] % 8) return fail(VQSEG_EINVAL, "bf16 rows need channels %% 8 == 0");
        if (!x[i] || !codebook[i] || !quant[i] || !idx[i] || !loss[i] || !dead_pct[i] || !ws[i]) return fail(VQSEG_EINVAL, "null pointer argument");
        if (!aligned16(x[i]) || !aligned16(ws[i]) || !aligned16(codebook[i]) || !aligned16(quant[i]) || (prepared && !aligned16(prepared[i])))
            return fail(VQSEG_EINVAL, "x, codebook, prepared, quant and workspace must be 16-byte aligned");
        plans[i] = vqseg::vq_plan(n[i], c[i], k[i]);
        if (ws_bytes[i] < plans[i].bytes) return fail(VQSEG_ENOSPC, "workspace %zu < %zu bytes", ws_bytes[i], plans[i].bytes);
        wsp[i] = static_cast<char*>(ws[i]);
        prep[i] = prepared ? prepared[i] : nullptr;
        if (!prep[i]) {
            hipError_t e = vqseg::launch_prepare(codebook[i], k[i], c[i], wsp[i] + plans[i].off_prepared, st);
            if (e != hipSuccess) return hip_fail(e, "vq codebook prepare");
            prep[i] = wsp[i] + plans[i].off_prepared;
        }
    }
    const int T = vqseg::vq_group_tiles(n_levels, n, k);
    hipError_t e = vqseg::launch_assign_group(n_levels, x, bf16, n, c, k, prep, plans, wsp, idx, nullptr, T, st, codebook);
    if (e != hipSuccess) return hip_fail(e, "vq_assign_f32_kernel (grouped)");
    for (int i = 0; i < n_levels; ++i) {
        e = vqseg::launch_gather(x[i], bf16, codebook[i], idx[i], n[i], c[i], k[i], training, cw[i], plans[i], wsp[i], quant[i], loss[i],
                                 dead_pct[i], st);
        if (e != hipSuccess) return hip_fail(e, "vq_gather_kernel");
    }
    return 0;
}

int vqseg_vq_backward_bf16(const void* gq, const float* gloss, const void* x, const int64_t* idx, const float* codebook,
                           int64_t n, int c, float cw, void* gx, void* stream) {
    if (int rc = check_shape(n, c, 1)) return rc;
    if (!gq || !x || !idx || !codebook || !gx) return fail(VQSEG_EINVAL, "null pointer argument");
    if (c % 8) return fail(VQSEG_EINVAL, "bf16 rows need channels %% 8 == 0");
    if (!aligned16(gq) || !aligned16(x) || !aligned16(codebook) || !aligned16(gx)) return fail(VQSEG_EINVAL, "tensors must be 16-byte aligned");
    hipError_t e = vqseg::launch_backward_idx(gq, gloss, x, idx, codebook, n, c, cw, gx, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "vq_backward_idx_kernel");
    return 0;
}

int vqseg_vq_backward_f32(const float* gq, const float* gloss, const float* x, const float* quant, int64_t n, int c,
                          float cw, float* gx, void* stream) {
    if (int rc = check_shape(n, c, 1)) return rc;
    if (!gq || !x || !quant || !gx) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(gq) || !aligned16(x) || !aligned16(quant) || !aligned16(gx))
        return fail(VQSEG_EINVAL, "tensors must be 16-byte aligned");
    hipError_t e = vqseg::launch_backward(gq, gloss, x, quant, n, c, cw, gx, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "vq_backward_kernel");
    return 0;
}

size_t vqseg_kmeans_workspace_bytes(int64_t n, int c, int k) {
    if (n <= 0 || c <= 0 || k <= 0) return 0;
    return vqseg::km_plan(n, c, k).bytes;
}

int vqseg_kmeans_accumulate_f32(const float* samples, const float* means, int64_t n, int c, int k, float* sums,
                                int64_t* counts, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_shape(n, c, k)) return rc;
    if (!samples || !means || !sums || !counts || !ws) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(samples) || !aligned16(ws)) return fail(VQSEG_EINVAL, "samples and workspace must be 16-byte aligned");
    const vqseg::KmPlan p = vqseg::km_plan(n, c, k);
    if (ws_bytes < p.bytes) return fail(VQSEG_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, p.bytes);
    hipError_t e = vqseg::launch_km_accumulate(samples, means, n, c, k, p, static_cast<char*>(ws), sums, counts,
                                               static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "kmeans accumulate");
    return 0;
}

int vqseg_vq_code_sums(int bf16, const void* x, const int64_t* idx, int64_t n, int c, int k, float* sums, int64_t* counts, void* ws,
                       size_t ws_bytes, void* stream) {
    if (int rc = check_shape(n, c, k)) return rc;
    if (!x || !idx || !sums || !counts || !ws) return fail(VQSEG_EINVAL, "null pointer argument");
    if (!aligned16(ws)) return fail(VQSEG_EINVAL, "workspace must be 16-byte aligned");
    const vqseg::KmPlan p = vqseg::km_plan(n, c, k);
    if (ws_bytes < p.bytes) return fail(VQSEG_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, p.bytes);
    hipError_t e = vqseg::launch_code_sums(x, bf16 != 0, idx, n, c, k, p, static_cast<char*>(ws), sums, counts,
                                           static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "code sums");
    return 0;
}

int vqseg_vq_ema_update_f32(float* cluster_size, float* embed_avg, float* codebook, const float* sums, const int64_t* counts, int c,
                            int k, float decay, float eps, float* scratch, void* stream) {
    if (c <= 0 || k <= 0 || !cluster_size || !embed_avg || !codebook || !sums || !counts || !scratch)
        return fail(VQSEG_EINVAL, "bad argument");
    if (!(decay >= 0.0f && decay <= 1.0f) || !(eps >= 0.0f)) return fail(VQSEG_EINVAL, "need 0 <= decay <= 1, eps >= 0");
    hipError_t e = vqseg::launch_ema_update(cluster_size, embed_avg, codebook, sums, counts, k, c, decay, eps, scratch,
                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "ema_embed_kernel");
    return 0;
}

int vqseg_kmeans_finalize_f32(const float* sums, const int64_t* counts, float* means, int c, int k, void* stream) {
    if (c <= 0 || k <= 0 || !sums || !counts || !means) return fail(VQSEG_EINVAL, "bad argument");
    hipError_t e = vqseg::launch_km_finalize(sums, counts, means, c, k, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "km_finalize_kernel");
    return 0;
}

int vqseg_kmeans_f32(const float* samples, float* means, int64_t* bins, int64_t n, int c, int k, int iters, void* ws,
                     size_t ws_bytes, void* stream) {
    if (int rc = check_shape(n, c, k)) return rc;
    if (!samples || !means || !bins || !ws) return fail(VQSEG_EINVAL, "null pointer argument");
    if (iters < 0) return fail(VQSEG_EINVAL, "iters must be >= 0");
    const vqseg::KmPlan p = vqseg::km_plan(n, c, k);
    if (ws_bytes < p.bytes) return fail(VQSEG_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, p.bytes);
    char* w = static_cast<char*>(ws);
    float* sums = reinterpret_cast<float*>(w + p.off_sums);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int it = 0; it < iters; ++it) {
        if (int rc = vqseg_kmeans_accumulate_f32(samples, means, n, c, k, sums, bins, ws, ws_bytes, stream)) return rc;
        hipError_t e = vqseg::launch_km_finalize(sums, bins, means, c, k, st);
        if (e != hipSuccess) return hip_fail(e, "km_finalize_kernel");
    }
    return 0;
}

}  // extern "C"
