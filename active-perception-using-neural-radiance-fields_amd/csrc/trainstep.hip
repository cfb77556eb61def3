// One training iteration's forward + loss + backward as ONE C call, and candidate-view scoring from poses as one C call.
//
//   mnf_train_step    scripts/pipeline.py:472-518 for one model: `render_image_with_occgrid_with_depth_guide`
//                     (perception/models/utils.py:63-219: occupancy sampling with the density pre-pass and the visibility
//                     filter of nerfacc/estimators/occ_grid.py:80-238, then `sem_rendering`, utils.py:362-461), the three-term
//                     loss (pipeline.py:506-511) and `loss.backward()` down to the three flat parameter-gradient vectors.
//                     The optimizer step and the NaN guard stay with the caller (pipeline.py:520-532), as does the occupancy
//                     refresh (mnf_update_occupancy).
//   mnf_score_poses   pipeline.py:674-781 for one trajectory: poses -> sub-sampled rays -> probabilistic renders of every
//                     ensemble member -> per-view predictive-information terms.
//
// Both are compositions of the library's own entry points plus the small kernels below (stratified near planes, visibility
// filter with compaction, loss + its gradient).  mnf_train_step never synchronises with the host: the marched and the
// surviving sample counts stay on the device (every launch is sized for the caller's upper bounds and reads the actual
// count from device memory), overflow of a bound is detected on the device (the step then yields zero gradients and a
// raised skip flag, which `mnf_adam_step_guarded` honours) and reported through `counts_dev`, which the caller reads
// whenever it wants to (the reference's boolean-mask indexing synchronises twice per step at these points).
//
// Build with -ffp-contract=off (the marcher's near planes feed bit-exact t values).
#include <cmath>
#include <cstring>
#include <vector>

#include "field.h"

// mnf_train_presample's handle: what was marched, for which rays / options, and the two events that order it
struct mnf_presample_s {
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    void *ws = nullptr;
    const float *rays_o = nullptr, *rays_d = nullptr;
    const uint8_t *binaries = nullptr;
    int32_t n_rays = 0, stratified = 0, n_levels = 0, res[3] = {0, 0, 0};
    uint64_t seed = 0;
    float near_plane = 0.f, far_plane = 0.f, step = 0.f, cone = 0.f, alpha_thre = 0.f;
    int64_t max_marched = 0;
    int valid = 0, launched = 0;
};

namespace mnf {
// train.hip: the next backward on this thread forms its per-sample output gradients from these factors (w[s] * g[ray[s]]) instead of reading them
struct FactoredGrad { const float *w; const int64_t *ray; const float *g_rgb, *g_sem; };
void set_factored_output_gradient(const FactoredGrad &fg);
// composite_train.hip: the public entry points with the logits class-major, sems[class * sem_stride + sample] (0: [sample][C])
int composite_train_forward_impl(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays, const float *t_starts, const float *t_ends, const float *sigmas,
                                 const float *rgbs, const float *sems, int64_t sem_stride, int32_t n_classes, int64_t n_samples, const float *bkgd, float *out_rgb,
                                 float *out_acc, float *out_depth, float *out_sem, float *weights, float *trans, float *alphas, mnf_stream_t stream);
int composite_train_backward_impl(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays, const float *t_starts, const float *t_ends, const float *sigmas,
                                  const float *rgbs, const float *sems, int64_t sem_stride, int32_t n_classes, int64_t n_samples, const float *bkgd, const float *weights,
                                  const float *trans, const float *out_acc, const float *out_depth, const float *g_rgb, const float *g_acc, const float *g_depth,
                                  const float *g_sem, float *d_sigmas, float *d_rgbs, float *d_sems, mnf_stream_t stream);
namespace {

constexpr float kEps = 1.1920928955078125e-07f;

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.x, p1 = (uint64_t)0xCD9E8D57u * ctr.z;
        ctr = {(uint32_t)(p1 >> 32) ^ ctr.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ ctr.w ^ k1, (uint32_t)p0};
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return ctr;
}

// occ_grid.py:181-189: near / far planes, stratified jitter near += U[0,1) * step (Philox counter (ray, 0, 7, 0), key = seed)
// Also the step's small zero-fills (loss terms, counters, skip flag, the two small gradient vectors): five separate memsets were five launches of ~5 us.
// The step's fills: what were seven memsets (losses, counters, skip flag, the three gradient vectors, the pre-pass's density array and its work counter) — launches of ~5 us
// each, the 50 MB one with a ~13 us bubble in front of it.
struct StepFills {
    float *g_base, *g_head, *g_sem, *sigma;
    int64_t n_base, n_head, n_sem, n_sigma;
    int32_t *ray_counter;
    const int64_t *adopt;      // a presampled step: the four counters its march's guard left (status bits != 0: the step is skipped), or NULL
};
__device__ __forceinline__ void fill_zero(float *p, int64_t n, int64_t tid, int64_t threads) {
    if (!p) return;
    const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;      // floats in front of the first 16-byte boundary
    const int64_t h = head < n ? head : n;
    for (int64_t i = tid; i < h; i += threads) p[i] = 0.f;
    float4 *q = reinterpret_cast<float4 *>(p + h);
    const int64_t nq = (n - h) / 4;
    for (int64_t i = tid; i < nq; i += threads) q[i] = float4{0.f, 0.f, 0.f, 0.f};
    for (int64_t i = h + nq * 4 + tid; i < n; i += threads) p[i] = 0.f;
}
__global__ void __launch_bounds__(256) planes_kernel(int32_t n, float near_plane, float far_plane, float step, int32_t stratified, uint32_t s0,
                                                     uint32_t s1, float *__restrict__ nearp, float *__restrict__ farp, float *__restrict__ losses,
                                                     int64_t *__restrict__ counts, int32_t *__restrict__ skip, const StepFills z) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (losses) {      // (NULL: a presample — the planes only; nearp NULL: a presampled step — the zero-fills only)
        if (r < 4) { losses[r] = 0.f; counts[r] = z.adopt ? z.adopt[r] : 0; }
        if (r == 0) { *skip = (z.adopt && z.adopt[3] != 0) ? 1 : 0; if (z.ray_counter) *z.ray_counter = 0; }
        const int64_t threads = (int64_t)gridDim.x * blockDim.x;
        fill_zero(z.g_base, z.n_base, r, threads); fill_zero(z.g_head, z.n_head, r, threads); fill_zero(z.g_sem, z.n_sem, r, threads);
        fill_zero(z.sigma, z.n_sigma, r, threads);
    }
    if (r >= n || !nearp) return;
    float v = near_plane;
    if (stratified) v += ((float)(philox4x32_10({(uint32_t)r, 0u, 7u, 0u}, s0, s1).x & 0xFFFFFFu) * 5.9604644775390625e-08f) * step;
    nearp[r] = v; farp[r] = far_plane;
}

// mean of occs (occ_grid.py:192: alpha_thre = min(alpha_thre, self.occs.mean())) in double: 128 partial sums, then one wave
__global__ void __launch_bounds__(256) mean_partial_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ part) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

__global__ void __launch_bounds__(64) mean_final_kernel(const double *__restrict__ part, int n_part, int64_t n, float alpha_thre, float *__restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 64) acc += part[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (threadIdx.x == 0) out[0] = fminf(alpha_thre, (float)(acc / (double)n));
}

__global__ void __launch_bounds__(1024) max_kernel(const int64_t *__restrict__ x, int64_t n, int64_t *__restrict__ out) {
    __shared__ long long s[1024];
    long long m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = x[i] > m ? x[i] : m;
    s[threadIdx.x] = m;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) { if ((int)threadIdx.x < d && s[threadIdx.x + d] > s[threadIdx.x]) s[threadIdx.x] = s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = s[0];
}

// render_visibility_from_density (volrend.py:424-483) on packed samples, one wave per ray: pass 0 counts the survivors of
// each ray, pass 1 (after the prefix sum of the counts) writes them compacted and grouped by ray.
template <bool WRITE>
__global__ void __launch_bounds__(64) visibility_kernel(int32_t n_rays, const int64_t *__restrict__ starts, const int64_t *__restrict__ cnts,
                                                        const float *__restrict__ ts, const float *__restrict__ te, const float *__restrict__ sig,
                                                        float early_stop_eps, const float *__restrict__ alpha_thre_dev,
                                                        int64_t *__restrict__ kept_cnts, const int64_t *__restrict__ kept_starts,
                                                        float *__restrict__ o_ts, float *__restrict__ o_te, int64_t *__restrict__ o_ray,
                                                        int64_t *__restrict__ o_src /* optional: the marched index of every survivor */) {
    const int lane = threadIdx.x;
    const float alpha_thre = alpha_thre_dev[0];
    for (int32_t r = blockIdx.x; r < n_rays; r += gridDim.x) {
        const int64_t s = starts[r];
        const int c = (int)cnts[r];
        float carry = 0.0f;
        int64_t out = WRITE ? kept_starts[r] : 0;
        int kept = 0;
        for (int base = 0; base < c; base += 64) {
            const bool valid = base + lane < c;
            const int64_t k = s + base + lane;
            const float a = valid ? ts[k] : 0.f, b = valid ? te[k] : 0.f;
            const float sdt = valid ? sig[k] * (b - a) : 0.0f;
            float incl = sdt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const float u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
            const float alpha = 1.0f - expf(-sdt);
            const float trans = expf(-((incl - sdt) + carry));
            carry += __shfl(incl, 63, 64);
            bool vis = valid && trans >= early_stop_eps;
            if (alpha_thre > 0.0f) vis = vis && alpha >= alpha_thre;
            const unsigned long long m = __ballot(vis);
            if (WRITE && vis) {
                const int64_t dst = out + __popcll(m & ((1ull << lane) - 1ull));
                o_ts[dst] = a; o_te[dst] = b; o_ray[dst] = r;
                if (o_src) o_src[dst] = k;
            }
            const int nk = __popcll(m);
            out += nk; kept += nk;
        }
        if (!WRITE && lane == 0) kept_cnts[r] = kept;
    }
}

// pipeline.py:506-511: 10 * smooth_l1(rgb, pixels) + smooth_l1(depth, dep[:, None]) / 5 + cross_entropy(sem, labels) / 2 (all
// "mean" reductions) and its gradient with respect to the rendered rgb / depth / semantic logits.  One lane per ray.
__device__ __forceinline__ void smooth_l1(float x, float &val, float &grad) {      // beta = 1
    const float ax = fabsf(x);
    val = ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
    grad = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);      // torch's branch order: a NaN difference gives a NaN gradient (the guard of pipeline.py:520-529 then drops the step)
}

__global__ void __launch_bounds__(256) loss_kernel(int32_t n_rays, int32_t C, const float *__restrict__ rgb, const float *__restrict__ depth,
                                                   const float *__restrict__ sem, const float *__restrict__ t_rgb, const float *__restrict__ t_dep,
                                                   const int64_t *__restrict__ t_sem, float *__restrict__ g_rgb, float *__restrict__ g_dep,
                                                   float *__restrict__ g_sem, float *__restrict__ losses /* [4]: total, rgb, depth, sem */,
                                                   int64_t *__restrict__ status, int32_t *__restrict__ skip) {
    __shared__ float s_acc[3][4];
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    float l_rgb = 0.f, l_dep = 0.f, l_sem = 0.f;
    if (r < n_rays) {
        const float inv_r = 1.0f / (float)n_rays;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v, g;
            smooth_l1(rgb[3 * r + k] - t_rgb[3 * r + k], v, g);
            l_rgb += v;
            g_rgb[3 * r + k] = g * (10.0f / 3.0f) * inv_r;
        }
        float v, g;
        smooth_l1(depth[r] - t_dep[r], v, g);
        l_dep = v;
        g_dep[r] = g * 0.2f * inv_r;
        const float *lg = sem + (int64_t)r * C;
        float mx = -INFINITY;
        for (int k = 0; k < C; ++k) mx = fmaxf(mx, lg[k]);
        float den = 0.f;
        for (int k = 0; k < C; ++k) den += expf(lg[k] - mx);
        const int64_t lab = t_sem[r];
        // F.cross_entropy raises a device assert for a class id outside [0, C); here the step is flagged (status bit 8, skip
        // raised: no optimizer update) and the ray contributes nothing, instead of reading past the logits row
        const bool lab_ok = lab >= 0 && lab < C;
        if (!lab_ok) { atomicOr(reinterpret_cast<unsigned long long *>(status), 8ull); atomicAdd(skip, 1); }
        l_sem = lab_ok ? logf(den) - (lg[lab] - mx) : 0.0f;
        for (int k = 0; k < C; ++k) g_sem[(int64_t)r * C + k] = lab_ok ? (expf(lg[k] - mx) / den - (k == lab ? 1.0f : 0.0f)) * 0.5f * inv_r : 0.0f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l_rgb += __shfl_xor(l_rgb, d, 64); l_dep += __shfl_xor(l_dep, d, 64); l_sem += __shfl_xor(l_sem, d, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_acc[0][wave] = l_rgb; s_acc[1][wave] = l_dep; s_acc[2][wave] = l_sem; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_r = 1.0f / (float)n_rays;
        const float a = (s_acc[0][0] + s_acc[0][1] + s_acc[0][2] + s_acc[0][3]) * inv_r / 3.0f;
        const float b = (s_acc[1][0] + s_acc[1][1] + s_acc[1][2] + s_acc[1][3]) * inv_r;
        const float c = (s_acc[2][0] + s_acc[2][1] + s_acc[2][2] + s_acc[2][3]) * inv_r;
        atomicAdd(&losses[1], a); atomicAdd(&losses[2], b); atomicAdd(&losses[3], c);
        atomicAdd(&losses[0], a * 10.0f + b / 5.0f + c / 2.0f);
    }
}

// Device-side bound checks (what the host did between its two syncs in rounds 1-2).  counts_dev: [marched, kept, longest ray, status];
// status bits: 1 marched > max_marched, 2 a ray longer than a scratch row, 4 kept > max_kept, 8 class id out of range, 16 no sample
// survived (the reference `continue`s: no backward, no optimizer step).  On overflow every per-ray count is zeroed, so the kernels
// behind the guard find nothing to do: the step yields zero gradients and `skip` > 0.
__global__ void __launch_bounds__(256) guard_marched_kernel(int32_t n_rays, int64_t *__restrict__ counts, int64_t *__restrict__ starts,
                                                            const int64_t *__restrict__ totals /* [0] marched, [2] longest */, int64_t max_marched,
                                                            int64_t row_cap, int64_t *__restrict__ eff /* [0] */, int64_t *__restrict__ counts_dev,
                                                            int32_t *__restrict__ skip) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = totals[0], longest = totals[2];
    const bool bad = tot > max_marched || longest > row_cap;
    if (bad && r < n_rays) { counts[r] = 0; starts[r] = 0; }
    if (r == 0) {
        eff[0] = bad ? 0 : tot;
        const unsigned long long bits = (unsigned long long)((tot > max_marched ? 1 : 0) | (longest > row_cap ? 2 : 0));
        if (!skip) {      // a presample: its own four counters (the step's planes_kernel adopts them: PreReport)
            counts_dev[0] = tot; counts_dev[1] = 0; counts_dev[2] = longest; counts_dev[3] = (int64_t)bits;
        } else {
            counts_dev[0] = tot; counts_dev[2] = longest;
            if (bad) { atomicOr(reinterpret_cast<unsigned long long *>(counts_dev + 3), bits); atomicAdd(skip, 1); }
        }
    }
}

__global__ void __launch_bounds__(256) guard_kept_kernel(int32_t n_rays, int64_t *__restrict__ kept_cnts, int64_t *__restrict__ kept_starts,
                                                         const int64_t *__restrict__ totals /* [1] kept */, int64_t max_kept,
                                                         int64_t *__restrict__ eff /* [1] */, int64_t *__restrict__ counts_dev, int32_t *__restrict__ skip) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t kept = totals[1];
    const bool bad = kept > max_kept;
    if (bad && r < n_rays) { kept_cnts[r] = 0; kept_starts[r] = 0; }
    if (r == 0) {
        counts_dev[1] = kept;
        eff[1] = bad ? 0 : kept;
        if (bad || kept == 0) {
            atomicOr(reinterpret_cast<unsigned long long *>(counts_dev + 3), bad ? 4ull : 16ull);
            atomicAdd(skip, 1);
        }
    }
}

__global__ void set3_kernel(float *dst, float a, float b, float c) { dst[0] = a; dst[1] = b; dst[2] = c; }

struct StepWs {
    float *nearp, *farp, *scratch_ts, *scratch_te, *alpha_thre;
    int64_t *counts, *starts, *kept_cnts, *kept_starts, *totals, *scan;
    float *ts, *te, *sigma;          // marched
    int64_t *ray;
    float *k_ts, *k_te, *k_rgb, *k_sigma, *k_sem, *k_pos, *k_w, *k_tr, *k_dsig, *k_drgb, *k_dsem;   // kept
    int64_t *k_ray;
    void *rows; int64_t *k_src;        // the pre-pass's feature rows [max_marched][64] x 16 bit and every surviving sample's row (field_rows_supported)
    float *o_rgb, *o_acc, *o_dep, *o_sem, *g_rgb, *g_dep, *g_sem;   // per ray
    void *field_ws;
    int64_t field_ws_bytes, bytes;
};

StepWs carve_step(char *base, mnf_field_t f, int64_t R, int32_t cap, int64_t max_marched, int64_t max_kept) {
    StepWs w;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += (b + 255) & ~(size_t)255; return p; };
    const int C = f->cfg.num_semantic_classes;
    w.nearp = (float *)take(R * 4); w.farp = (float *)take(R * 4); w.alpha_thre = (float *)take(256);
    w.counts = (int64_t *)take(R * 8); w.starts = (int64_t *)take(R * 8); w.kept_cnts = (int64_t *)take(R * 8); w.kept_starts = (int64_t *)take(R * 8);
    w.totals = (int64_t *)take(2048); w.scan = (int64_t *)take((size_t)mnf_scan_workspace_bytes(R));
    w.scratch_ts = (float *)take((size_t)R * cap * 4); w.scratch_te = (float *)take((size_t)R * cap * 4);
    w.ts = (float *)take(max_marched * 4); w.te = (float *)take(max_marched * 4); w.sigma = (float *)take(max_marched * 4); w.ray = (int64_t *)take(max_marched * 8);
    w.k_ts = (float *)take(max_kept * 4); w.k_te = (float *)take(max_kept * 4); w.k_ray = (int64_t *)take(max_kept * 8);
    w.k_rgb = (float *)take(max_kept * 12); w.k_sigma = (float *)take(max_kept * 4); w.k_sem = (float *)take((size_t)max_kept * C * 4);
    w.k_pos = (float *)take(max_kept * 12); w.k_w = (float *)take(max_kept * 4); w.k_tr = (float *)take(max_kept * 4);
    w.k_dsig = (float *)take(max_kept * 4); w.k_drgb = (float *)take(max_kept * 12); w.k_dsem = (float *)take((size_t)max_kept * C * 4);
    w.o_rgb = (float *)take(R * 12); w.o_acc = (float *)take(R * 4); w.o_dep = (float *)take(R * 4); w.o_sem = (float *)take((size_t)R * C * 4);
    w.g_rgb = (float *)take(R * 12); w.g_dep = (float *)take(R * 4); w.g_sem = (float *)take((size_t)R * C * 4);
    w.rows = nullptr; w.k_src = nullptr;
    if (field_rows_supported(f)) { w.rows = take((size_t)max_marched * 128); w.k_src = (int64_t *)take(max_kept * 8); }
    w.field_ws_bytes = mnf_field_train_workspace_bytes(f, max_kept);
    w.field_ws = take((size_t)w.field_ws_bytes);
    w.bytes = (int64_t)off;
    return w;
}

inline int32_t scratch_cap(int32_t n_rays) {
    const int64_t c = ((int64_t)1 << 27) / (n_rays > 0 ? n_rays : 1);
    return (int32_t)(c < 64 ? 64 : (c > 2048 ? 2048 : c));
}

// What the parameter-independent head of a step leaves behind (mnf_train_presample): the part of StepWs the march fills.
struct SampleWs {
    float *nearp, *farp, *alpha_thre, *scratch_ts, *scratch_te;
    int64_t *counts, *starts, *totals, *scan;
    int64_t bytes;
    float *ts, *te; int64_t *ray;      // a presample's packed samples (max_marched > 0)
};

SampleWs carve_sample(char *base, int64_t R, int32_t cap, int64_t max_marched = 0) {
    SampleWs w;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += (b + 255) & ~(size_t)255; return p; };
    w.nearp = (float *)take(R * 4); w.farp = (float *)take(R * 4); w.alpha_thre = (float *)take(256);
    w.counts = (int64_t *)take(R * 8); w.starts = (int64_t *)take(R * 8);
    w.totals = (int64_t *)take(2048); w.scan = (int64_t *)take((size_t)mnf_scan_workspace_bytes(R));
    w.scratch_ts = (float *)take((size_t)R * cap * 4); w.scratch_te = (float *)take((size_t)R * cap * 4);
    w.ts = w.te = nullptr; w.ray = nullptr;
    if (max_marched > 0) { w.ts = (float *)take(max_marched * 4); w.te = (float *)take(max_marched * 4); w.ray = (int64_t *)take(max_marched * 8); }
    w.bytes = (int64_t)off;
    return w;
}

constexpr int kPreReport = 160;      // int64 words into a presample's `totals` block (2048 bytes: counters 0-7, 128 doubles of the mean from word 8): its guard's four counters

inline int fill_blocks(int rblocks) { return rblocks > 2048 ? rblocks : 2048; }      // enough threads for the 50 MB gradient fill (16 bytes per thread and pass)

// near planes, alpha threshold, march, per-ray offsets, longest ray: everything of a step that reads the rays and the occupancy grid but not the parameters
int sample_stage(const SampleWs &w, const uint8_t *binaries, const uint32_t *bitgrid, const float *occs, int32_t res_x, int32_t res_y, int32_t res_z,
                 const float *aabb_host, const float *rays_o, const float *rays_d, int32_t n_rays, const mnf_train_opts *opts, float *losses,
                 int64_t *counts_dev, int32_t *skip_dev, const StepFills &fills, hipStream_t s) {
    const int64_t cells = (int64_t)res_x * res_y * res_z;
    const int rblocks = (n_rays + 255) / 256;
    const int32_t cap = scratch_cap(n_rays);
    hipLaunchKernelGGL(planes_kernel, dim3(losses ? fill_blocks(rblocks) : rblocks), dim3(256), 0, s, n_rays, opts->near_plane, opts->far_plane, opts->render_step_size,
                       opts->stratified, (uint32_t)opts->seed, (uint32_t)(opts->seed >> 32), w.nearp, w.farp, losses, counts_dev, skip_dev, fills);
    double *mean_part = reinterpret_cast<double *>(w.totals + 8);          // 128 doubles behind the counters
    const int n_levels = opts->n_levels > 1 ? opts->n_levels : 1;
    hipLaunchKernelGGL(mean_partial_kernel, dim3(128), dim3(256), 0, s, occs, cells * n_levels, mean_part);          // occ_grid.py:192: the mean over every level
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, s, (const double *)mean_part, 128, cells * n_levels, opts->alpha_thre, w.alpha_thre);
    int rc = mnf_sample_rays_levels(rays_o, rays_d, n_rays, binaries, n_levels, res_x, res_y, res_z, aabb_host, w.nearp, w.farp, opts->render_step_size,
                                    opts->cone_angle, cap, w.scratch_ts, w.scratch_te, w.counts, bitgrid, (mnf_stream_t)s);
    if (rc) return rc;
    rc = mnf_exclusive_scan_i64(w.counts, n_rays, w.starts, w.totals, w.scan, mnf_scan_workspace_bytes(n_rays), (mnf_stream_t)s);
    if (rc) return rc;
    hipLaunchKernelGGL(max_kernel, dim3(1), dim3(1024), 0, s, w.counts, (int64_t)n_rays, w.totals + 2);
    return launch_status("sample_stage");
}

}  // namespace
}  // namespace mnf

using namespace mnf;

extern "C" int mnf_visible_samples(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays, const float *t_starts, const float *t_ends,
                                   const float *sigmas, float early_stop_eps, const float *alpha_thre_dev, int64_t *kept_cnts, const int64_t *kept_starts,
                                   float *o_t_starts, float *o_t_ends, int64_t *o_ray_indices, mnf_stream_t stream) {
    if (n_rays <= 0) return MNF_OK;
    MNF_REQUIRE(chunk_starts && chunk_cnts && t_starts && t_ends && sigmas && alpha_thre_dev, "visible_samples: null pointer");
    const int vgrid = n_rays < 65535 ? n_rays : 65535;
    if (!o_t_starts) {
        MNF_REQUIRE(kept_cnts, "visible_samples: the count pass needs kept_cnts");
        hipLaunchKernelGGL(visibility_kernel<false>, dim3(vgrid), dim3(64), 0, as_stream(stream), n_rays, chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, early_stop_eps,
                           alpha_thre_dev, kept_cnts, (const int64_t *)nullptr, (float *)nullptr, (float *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr);
    } else {
        MNF_REQUIRE(kept_starts && o_t_ends && o_ray_indices, "visible_samples: the write pass needs kept_starts and the three outputs");
        hipLaunchKernelGGL(visibility_kernel<true>, dim3(vgrid), dim3(64), 0, as_stream(stream), n_rays, chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, early_stop_eps,
                           alpha_thre_dev, (int64_t *)nullptr, kept_starts, o_t_starts, o_t_ends, o_ray_indices, (int64_t *)nullptr);
    }
    return launch_status("visibility_kernel");
}

extern "C" int64_t mnf_train_step_workspace_bytes(mnf_field_t f, int32_t n_rays, int64_t max_marched, int64_t max_kept) {
    if (!f || n_rays <= 0 || max_marched <= 0 || max_kept <= 0) return -1;
    return carve_step(nullptr, f, n_rays, scratch_cap(n_rays), max_marched, max_kept).bytes;
}

extern "C" int mnf_presample_create(mnf_presample_t *out) {
    MNF_REQUIRE(out, "presample_create: null pointer");
    mnf_presample_s *p = new mnf_presample_s();
    if (hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming) != hipSuccess) {
        if (p->ev_ready) (void)hipEventDestroy(p->ev_ready);
        delete p;
        set_error("presample_create: hipEventCreate failed");
        return MNF_ERR_HIP;
    }
    *out = p;
    return MNF_OK;
}

extern "C" void mnf_presample_destroy(mnf_presample_t p) {
    if (!p) return;
    (void)hipEventDestroy(p->ev_ready);
    (void)hipEventDestroy(p->ev_done);
    delete p;
}

extern "C" int64_t mnf_train_presample_workspace_bytes(int32_t n_rays, int64_t max_marched) {
    if (n_rays <= 0 || max_marched <= 0) return -1;
    return carve_sample(nullptr, n_rays, scratch_cap(n_rays), max_marched).bytes;
}

extern "C" int mnf_presample_wait(mnf_presample_t p, mnf_stream_t stream) {
    MNF_REQUIRE(p, "presample_wait: null handle");
    if (p->launched) MNF_HIP(hipStreamWaitEvent(as_stream(stream), p->ev_done, 0));
    return MNF_OK;
}

extern "C" int mnf_train_presample(mnf_presample_t p, const uint8_t *binaries, const uint32_t *bitgrid, const float *occs, int32_t res_x, int32_t res_y,
                                   int32_t res_z, const float *aabb_host, const float *rays_o, const float *rays_d, int32_t n_rays,
                                   const mnf_train_opts *opts, int64_t max_marched, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    MNF_REQUIRE(p && binaries && occs && aabb_host && rays_o && rays_d && opts && workspace, "train_presample: null pointer");
    MNF_REQUIRE(opts->struct_size == sizeof(mnf_train_opts), "train_presample: opts->struct_size is %u, this library's mnf_train_opts has %zu bytes (MNF_INIT)",
                opts->struct_size, sizeof(mnf_train_opts));
    MNF_REQUIRE(n_rays > 0 && max_marched > 0 && opts->render_step_size > 0.f, "train_presample: bad sizes");
    const int n_levels = opts->n_levels > 1 ? opts->n_levels : 1;
    MNF_REQUIRE(n_levels <= 4, "train_presample: at most 4 occupancy levels");
    const int32_t cap = scratch_cap(n_rays);
    const SampleWs w = carve_sample((char *)workspace, n_rays, cap, max_marched);
    if (workspace_bytes < w.bytes) { set_error("train_presample: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)w.bytes); return MNF_ERR_WORKSPACE; }
    hipStream_t s = as_stream(stream), ss = shared_side_stream(2);
    if (!ss) return MNF_ERR_HIP;
    p->valid = 0;
    // fork: the side stream continues from the caller's stream as it is NOW — in front of whatever the caller enqueues next (the step this march is to hide behind)
    MNF_HIP(hipEventRecord(p->ev_ready, s));
    MNF_HIP(hipStreamWaitEvent(ss, p->ev_ready, 0));
    int rc = sample_stage(w, binaries, bitgrid, occs, res_x, res_y, res_z, aabb_host, rays_o, rays_d, n_rays, opts, nullptr, nullptr, nullptr, StepFills{}, ss);
    if (!rc) {      // the bound check of the step (its four counters go to the handle's own words: the adopting step copies them) and the packing of the scratch rows
        hipLaunchKernelGGL(guard_marched_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, ss, n_rays, w.counts, w.starts, (const int64_t *)w.totals, max_marched,
                           (int64_t)cap, w.totals + 4, w.totals + kPreReport, (int32_t *)nullptr);
        rc = mnf_compact_samples(w.scratch_ts, w.scratch_te, cap, w.starts, w.counts, n_rays, w.ts, w.te, w.ray, (mnf_stream_t)ss);
    }
    MNF_HIP(hipEventRecord(p->ev_done, ss));
    p->launched = 1;
    if (rc) return rc;
    p->ws = workspace; p->rays_o = rays_o; p->rays_d = rays_d; p->binaries = binaries; p->n_rays = n_rays; p->seed = opts->seed; p->stratified = opts->stratified;
    p->n_levels = n_levels; p->near_plane = opts->near_plane; p->far_plane = opts->far_plane; p->step = opts->render_step_size; p->cone = opts->cone_angle;
    p->alpha_thre = opts->alpha_thre; p->res[0] = res_x; p->res[1] = res_y; p->res[2] = res_z; p->max_marched = max_marched;
    p->valid = 1;
    return MNF_OK;
}

extern "C" int mnf_train_step(mnf_field_t f, const uint8_t *binaries, const uint32_t *bitgrid, const float *occs, int32_t res_x, int32_t res_y,
                              int32_t res_z, const float *aabb_host, const float *rays_o, const float *rays_d, int32_t n_rays,
                              const float *target_rgb, const float *target_depth, const int64_t *target_sem, const mnf_train_opts *opts,
                              float *g_base, float *g_head, float *g_sem, float *losses, int64_t *counts_dev, int32_t *skip_dev, int64_t max_marched,
                              int64_t max_kept, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    MNF_REQUIRE(f && f->params_loaded && opts && aabb_host && counts_dev && skip_dev, "train_step: bad handle or options");
    MNF_REQUIRE(opts->struct_size == sizeof(mnf_train_opts), "train_step: opts->struct_size is %u, this library's mnf_train_opts has %zu bytes (MNF_INIT)",
                opts->struct_size, sizeof(mnf_train_opts));
    MNF_REQUIRE(binaries && occs && rays_o && rays_d && target_rgb && target_depth && target_sem && g_base && g_head && g_sem && losses && workspace,
                "train_step: null pointer");
    MNF_REQUIRE(n_rays > 0 && max_marched > 0 && max_kept > 0 && opts->render_step_size > 0.f, "train_step: bad sizes");
    hipStream_t s = as_stream(stream);
    const int32_t cap = scratch_cap(n_rays);
    StepWs w = carve_step((char *)workspace, f, n_rays, cap, max_marched, max_kept);
    if (workspace_bytes < w.bytes) { set_error("train_step: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)w.bytes); return MNF_ERR_WORKSPACE; }
    const int C = f->cfg.num_semantic_classes;
    const int rblocks = (n_rays + 255) / 256;
    // (losses, counters, skip flag, the three gradient vectors, the pre-pass's density array and work counter: zeroed by planes_kernel)
    int64_t *eff = w.totals + 4;                                           // [0] marched, [1] kept samples the kernels behind the guards work on
    // ---- occupancy sampling (occ_grid.py:80-238): march, density pre-pass, visibility filter
    const int n_levels = opts->n_levels > 1 ? opts->n_levels : 1;
    MNF_REQUIRE(n_levels <= 4, "train_step: at most 4 occupancy levels");
    int rc = 0;
    StepFills fills = {g_base, g_head, g_sem, w.sigma, (int64_t)f->n_base, (int64_t)f->n_head, (int64_t)f->n_sem, max_marched, nullptr, nullptr};
    if (const mnf_presample_s *pre = opts->presampled) {
        // the march of THIS batch ran earlier, on a side stream beside the previous step (mnf_train_presample): adopt what it left, after its last kernel
        MNF_REQUIRE(pre->valid && pre->rays_o == rays_o && pre->rays_d == rays_d && pre->n_rays == n_rays && pre->binaries == binaries && pre->seed == opts->seed &&
                    pre->stratified == opts->stratified && pre->n_levels == n_levels && pre->near_plane == opts->near_plane && pre->far_plane == opts->far_plane &&
                    pre->step == opts->render_step_size && pre->cone == opts->cone_angle && pre->alpha_thre == opts->alpha_thre &&
                    pre->res[0] == res_x && pre->res[1] == res_y && pre->res[2] == res_z && pre->max_marched == max_marched,
                    "train_step: opts->presampled was made for other rays, options, sample bound or another grid");
        if (hipEventQuery(pre->ev_done) != hipSuccess) MNF_HIP(hipStreamWaitEvent(s, pre->ev_done, 0));      // (a wait on another queue costs the stream ~18 us even when the event is long done)
        const SampleWs sw = carve_sample((char *)pre->ws, n_rays, cap, max_marched);
        w.nearp = sw.nearp; w.farp = sw.farp; w.alpha_thre = sw.alpha_thre; w.counts = sw.counts; w.starts = sw.starts; w.totals = sw.totals;
        w.scratch_ts = sw.scratch_ts; w.scratch_te = sw.scratch_te; w.ts = sw.ts; w.te = sw.te; w.ray = sw.ray;
        eff = w.totals + 4;
        const_cast<mnf_presample_s *>(pre)->valid = 0;      // (the guards below may clear its counts: one use)
        fills.ray_counter = reinterpret_cast<int32_t *>(w.totals + 7);
        fills.adopt = w.totals + kPreReport;
        hipLaunchKernelGGL(planes_kernel, dim3(fill_blocks(rblocks)), dim3(256), 0, s, 0, 0.f, 0.f, 0.f, 0, 0u, 0u, (float *)nullptr, (float *)nullptr, losses, counts_dev,
                           skip_dev, fills);
    } else {
        const SampleWs sw = {w.nearp, w.farp, w.alpha_thre, w.scratch_ts, w.scratch_te, w.counts, w.starts, w.totals, w.scan, 0};
        fills.ray_counter = reinterpret_cast<int32_t *>(w.totals + 7);
        rc = sample_stage(sw, binaries, bitgrid, occs, res_x, res_y, res_z, aabb_host, rays_o, rays_d, n_rays, opts, losses, counts_dev, skip_dev, fills, s);
        if (rc) return rc;
    }
    if (!opts->presampled) {      // (a presample has done both)
        // a ray longer than its scratch row would have been truncated (the rows hold `cap` samples; the reference configurations stay far
        // below) and more marched samples than `max_marched` would not fit the packed arrays: both end the step here, on the device
        hipLaunchKernelGGL(guard_marched_kernel, dim3(rblocks), dim3(256), 0, s, n_rays, w.counts, w.starts, (const int64_t *)w.totals, max_marched,
                           (int64_t)cap, eff, counts_dev, skip_dev);
        rc = mnf_compact_samples(w.scratch_ts, w.scratch_te, cap, w.starts, w.counts, n_rays, w.ts, w.te, w.ray, stream);
        if (rc) return rc;
    }
    bool use_rows = false;
    static const bool prepass_flat = diag_env("MNF_PREPASS_FLAT") != nullptr;    // experiment: every marched sample, full lanes, no early termination
    if (prepass_flat) {
        FieldIO io = {};
        io.mode = 1; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = w.ray; io.t_starts = w.ts; io.t_ends = w.te;
        io.n = max_marched; io.n_dev64 = eff; io.density = w.sigma;
        rc = launch_field(f, io, true, s);
    } else {
        // mnf_field_density_rays without its two memsets (density array, work counter: planes_kernel's fills): ray-major, skips what lies behind T < eps / 2
        FieldIO io = {};
        io.mode = 3; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = w.ray; io.t_starts = w.ts; io.t_ends = w.te; io.n = max_marched;
        io.chunk_starts = w.starts; io.chunk_cnts = w.counts; io.n_rays = n_rays; io.ray_counter = fills.ray_counter;
        io.sdt_stop = opts->early_stop_eps > 0.0f ? -logf(opts->early_stop_eps) + 0.6931472f : INFINITY;
        io.density = w.sigma;
        static const bool no_rows = diag_env("MNF_NO_ROWS") != nullptr;      // A/B timing: both passes gather
        use_rows = w.rows != nullptr && !no_rows;
        io.rows_out = use_rows ? w.rows : nullptr;       // (NULL where the rows are not built: the forward then gathers for itself)
        rc = launch_field(f, io, true, s);
    }
    if (rc) return rc;
    const int vgrid = n_rays < 65535 ? n_rays : 65535;
    hipLaunchKernelGGL(visibility_kernel<false>, dim3(vgrid), dim3(64), 0, s, n_rays, w.starts, w.counts, w.ts, w.te, w.sigma, opts->early_stop_eps,
                       w.alpha_thre, w.kept_cnts, (const int64_t *)nullptr, (float *)nullptr, (float *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr);
    rc = mnf_exclusive_scan_i64(w.kept_cnts, n_rays, w.kept_starts, w.totals + 1, w.scan, mnf_scan_workspace_bytes(n_rays), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(guard_kept_kernel, dim3(rblocks), dim3(256), 0, s, n_rays, w.kept_cnts, w.kept_starts, (const int64_t *)w.totals, max_kept, eff,
                       counts_dev, skip_dev);
    hipLaunchKernelGGL(visibility_kernel<true>, dim3(vgrid), dim3(64), 0, s, n_rays, w.starts, w.counts, w.ts, w.te, w.sigma, opts->early_stop_eps,
                       w.alpha_thre, (int64_t *)nullptr, w.kept_starts, w.k_ts, w.k_te, w.k_ray, use_rows ? w.k_src : (int64_t *)nullptr);
    // ---- sem_rendering (utils.py:362-461): field with saved activations, compositing
    {
        FieldIO io = {};
        io.mode = 1; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = w.k_ray; io.t_starts = w.k_ts; io.t_ends = w.k_te;
        io.n = max_kept; io.n_dev64 = eff + 1;
        io.rgb = w.k_rgb; io.density = w.k_sigma; io.sem = w.k_sem; io.xn_out = w.k_pos;        // aabb-normalised: what the backward's scatter reads
        io.sem_stride = max_kept;                                                                // the step's own logit buffer is class-major: w.k_sem[class * max_kept + sample]
        if (use_rows) { io.rows_in = w.rows; io.rows_src = w.k_src; }                            // every survivor's features: the row the pre-pass left, not 128 gathers
        rc = forward_train(f, io, w.field_ws, w.field_ws_bytes, s, opts->deterministic != 0);
        if (rc) return rc;
    }
    const float *bk = opts->render_bkgd_dev;                               // the caller's device colour, or the by-value one
    if (!bk && (opts->render_bkgd[0] != 0.f || opts->render_bkgd[1] != 0.f || opts->render_bkgd[2] != 0.f)) {
        float *bkd = w.alpha_thre + 8;                                     // three floats of the small scalar block
        hipLaunchKernelGGL(set3_kernel, dim3(1), dim3(1), 0, s, bkd, opts->render_bkgd[0], opts->render_bkgd[1], opts->render_bkgd[2]);
        bk = bkd;
    }
    rc = composite_train_forward_impl(w.kept_starts, w.kept_cnts, n_rays, w.k_ts, w.k_te, w.k_sigma, w.k_rgb, w.k_sem, max_kept, C, max_kept, bk, w.o_rgb, w.o_acc,
                                      w.o_dep, w.o_sem, w.k_w, w.k_tr, nullptr, stream);
    if (rc) return rc;
    // ---- loss (pipeline.py:506-511) and backward (pipeline.py:518)
    hipLaunchKernelGGL(loss_kernel, dim3(rblocks), dim3(256), 0, s, n_rays, C, w.o_rgb, w.o_dep, w.o_sem, target_rgb, target_depth, target_sem,
                       w.g_rgb, w.g_dep, w.g_sem, losses, counts_dev + 3, skip_dev);
    rc = launch_status("loss_kernel");
    if (rc) return rc;
    // the rgb / semantic output gradients of a sample are its weight times its ray's loss gradient: the split backward forms them itself from (k_w, k_ray, g_rgb, g_sem)
    // — 12 bytes per sample and cached per-ray vectors instead of 128 bytes written here and read there; the fused backward (mode 2) reads the per-sample arrays
    const bool factored = C > 0;
    rc = composite_train_backward_impl(w.kept_starts, w.kept_cnts, n_rays, w.k_ts, w.k_te, w.k_sigma, w.k_rgb, w.k_sem, max_kept, C, max_kept, bk, w.k_w, w.k_tr, w.o_acc,
                                       w.o_dep, w.g_rgb, nullptr, w.g_dep, w.g_sem, w.k_dsig, factored ? nullptr : w.k_drgb, factored ? nullptr : w.k_dsem, stream);
    if (rc) return rc;
    if (factored) set_factored_output_gradient(FactoredGrad{w.k_w, w.k_ray, w.g_rgb, w.g_sem});
    return backward(f, w.k_pos, max_kept, eff + 1, w.k_drgb, w.k_dsig, w.k_dsem, w.k_rgb, w.k_sigma, w.field_ws, w.field_ws_bytes, opts->loss_scale, g_base,
                    g_head, g_sem, false, true, opts->deterministic != 0, s);
}

// view groups per member of a scoring call.  FOUR render jobs in flight (the caller's stream + the three shared side streams) is the measured optimum for
// these small views (profiles/r03_hw_queues.txt: 256 views x 2 members 101.1 ms with 4 jobs, 107.0 with 2; 32 views 21.3 against 22.6; more jobs than
// streams: worse): two members -> two halves of the pose list each, one member -> four quarters; one half's marcher runs beside another's field kernel.
// (profiles/r03_split_experiment.txt had two as the optimum: measured when every job still had a stream of its own, see common.h shared_side_stream.)
static inline int score_groups(int32_t n_views, int32_t n_members = 1) {
    int g = 4 / (n_members > 0 ? n_members : 1);
    if (g < 1) g = 1;
    return g > n_views ? (n_views > 0 ? n_views : 1) : g;
}
static inline int32_t group_lo(int32_t n_views, int g, int G) { return (int32_t)((int64_t)n_views * g / G); }

extern "C" int64_t mnf_score_poses_workspace_bytes(int32_t n_members, int32_t n_views, int32_t n_pix, int32_t n_classes) {
    if (n_members <= 0 || n_views <= 0 || n_pix <= 0 || n_classes <= 0) return -1;
    const int64_t R = (int64_t)n_views * n_pix;
    const int64_t per_member = R * (3 + 1 + 1 + n_classes + 3 + 1) * 4 + 8 * 256;
    const int G = score_groups(n_views, n_members);
    int64_t render = 0;
    for (int g = 0; g < G; ++g) render += mnf_render_workspace_bytes((int64_t)(group_lo(n_views, g + 1, G) - group_lo(n_views, g, G)) * n_pix, n_pix) + 256;
    return R * 24 + 512 + n_members * (per_member + render + 64 * G + 256) + 4096;
}

extern "C" int mnf_score_poses(const mnf_field_t *fields_host, const uint8_t *const *binaries_host, const uint32_t *const *bitgrids_host,
                               int32_t n_members, int32_t res_x, int32_t res_y, int32_t res_z, const float *aabb_host, const float *c2w,
                               int32_t n_views, int32_t width, int32_t height, float focal, const int64_t *pix_idx, int64_t n_pix,
                               const mnf_render_opts *opts, double *terms, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    MNF_REQUIRE(fields_host && binaries_host && aabb_host && c2w && opts && terms && workspace && pix_idx, "score_poses: null pointer");
    MNF_REQUIRE(opts->struct_size == sizeof(mnf_render_opts), "score_poses: opts->struct_size is %u, this library's mnf_render_opts has %zu bytes (MNF_INIT)",
                opts->struct_size, sizeof(mnf_render_opts));
    MNF_REQUIRE(n_members >= 1 && n_members <= 16 && n_views >= 1 && n_pix >= 1, "score_poses: bad sizes");
    const int C = fields_host[0]->cfg.num_semantic_classes;
    for (int m = 1; m < n_members; ++m) MNF_REQUIRE(fields_host[m]->cfg.num_semantic_classes == C, "score_poses: members disagree on the class count");
    const int64_t need = mnf_score_poses_workspace_bytes(n_members, n_views, (int32_t)n_pix, C);
    if (workspace_bytes < need) { set_error("score_poses: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)need); return MNF_ERR_WORKSPACE; }
    const int64_t R = (int64_t)n_views * n_pix;
    char *base = (char *)workspace;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += (b + 255) & ~(size_t)255; return p; };
    float *o = (float *)take(R * 12), *d = (float *)take(R * 12);
    // member-major stacks, exactly what mnf_score_views reads
    float *rgb_var = (float *)take((size_t)n_members * R * 12), *depth_var = (float *)take((size_t)n_members * R * 4);
    float *acc = (float *)take((size_t)n_members * R * 4), *sem = (float *)take((size_t)n_members * R * C * 4);
    int rc = mnf_generate_rays(c2w, n_views, width, height, focal, pix_idx, n_pix, o, d, stream);
    if (rc) return rc;
    mnf_render_opts ro = *opts;
    ro.probabilistic = 1; ro.rays_per_view = (int32_t)n_pix; ro.bitgrid = nullptr;      // (the caller's view_order, if any, applies to every view)
    // every (member, half of the pose list) is a render job of its own: they advance side by side (mnf_render_jobs)
    const int G = score_groups(n_views, n_members);
    std::vector<mnf_render_job> jobs;
    for (int m = 0; m < n_members; ++m) {
        float *rgb = (float *)take(R * 12), *depth = (float *)take(R * 4);          // outputs the scorer does not read
        for (int g = 0; g < G; ++g) {
            const int64_t r0 = (int64_t)group_lo(n_views, g, G) * n_pix, r1 = (int64_t)group_lo(n_views, g + 1, G) * n_pix;
            mnf_render_job j = {};
            j.struct_size = (uint32_t)sizeof(j);
            j.field = fields_host[m]; j.binaries = binaries_host[m]; j.bitgrid = bitgrids_host ? bitgrids_host[m] : nullptr;
            j.rays_o = o + 3 * r0; j.rays_d = d + 3 * r0; j.n_rays = r1 - r0;
            j.rgb = rgb + 3 * r0; j.depth = depth + r0;
            j.acc = acc + (size_t)m * R + r0; j.sem = sem + ((size_t)m * R + r0) * C;
            j.rgb_var = rgb_var + ((size_t)m * R + r0) * 3; j.depth_var = depth_var + (size_t)m * R + r0;
            j.total_samples = (int64_t *)take(64);
            j.workspace_bytes = mnf_render_workspace_bytes(r1 - r0, (int32_t)n_pix);
            j.workspace = take((size_t)j.workspace_bytes);
            jobs.push_back(j);
        }
    }
    MNF_REQUIRE((int64_t)off <= workspace_bytes, "score_poses: internal workspace accounting error");
    rc = mnf_render_jobs(jobs.data(), (int32_t)jobs.size(), res_x, res_y, res_z, aabb_host, &ro, stream);
    if (rc) return rc;
    return mnf_score_views(rgb_var, depth_var, acc, sem, n_members, n_views, (int32_t)n_pix, C, terms, stream);
}
