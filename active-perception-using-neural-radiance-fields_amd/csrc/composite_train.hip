// Train-mode semantic volume rendering on packed samples, forward and backward, one wave per ray.
//
// Replaces the chain the reference builds out of torch ops for every training iteration
// (perception/models/utils.py:362-461 `sem_rendering`): render_weight_from_density (volrend.py:213-267: exclusive_sum,
// exp, 1-exp), four accumulate_along_rays index_adds (volrend.py:27-66) and, in backward, their autograd twins.
//
//   s_i = sigma_i (te_i - ts_i)     T_i = exp(-sum_{j<i} s_j)     alpha_i = 1 - exp(-s_i)     w_i = T_i alpha_i
//   C = sum w_i c_i + bkgd (1 - A)   A = sum w_i   D = sum w_i (ts_i+te_i)/2 / max(A, eps)   S = sum w_i sem_i
//
// Backward, with g_i = dL/dw_i = gC.c_i + gA' + gDn m_i + gS.sem_i:
//   dL/ds_i = g_i T_i (1 - alpha_i) - sum_{k>i} g_k w_k          (one reverse sweep with a running suffix sum)
//
// A wave walks its ray in blocks of 64 samples (lane = sample); the [64 x C] semantic block is staged through LDS with
// coalesced row-major loads/stores so that neither the per-sample dot product nor the per-class sum strides global memory.
#include "common.h"

namespace mnf {
namespace {

constexpr int kMaxClasses = 64;
constexpr float kEps = 1.1920928955078125e-07f;     // torch.finfo(float32).eps, utils.py:447

__device__ __forceinline__ float wave_inclusive_scan(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Copy n_rows x C floats between row-major global memory and LDS rows of `stride` words (stride odd: conflict-free
// when lanes later walk one row each).
template <bool TO_LDS>
__device__ __forceinline__ void stage_rows(float *__restrict__ lds, float *g, int n_rows, int C, int stride, int lane) {
    const int n = n_rows * C;
    int row = lane / C, col = lane - row * C;
    const int drow = 64 / C, dcol = 64 - drow * C;
    for (int e = lane; e < n; e += 64) {
        if (TO_LDS) lds[row * stride + col] = g[e];
        else g[e] = lds[row * stride + col];
        row += drow; col += dcol;
        if (col >= C) { col -= C; ++row; }
    }
}

// The same copy in two halves with the block's elements held in registers in between (element e = lane + 64 j of the row-major block): the loads of the NEXT
// block are issued before the current block is worked on, so that a ray's blocks — one dependent chain per wave, and the longest ray sets the kernel's time
// (2000 rays: 49 us for 245 k samples) — do not each pay a full memory round trip.  CB: compile-time bound of C (registers).
template <int CB>
__device__ __forceinline__ void rows_load(float (&v)[CB], const float *g, int n_rows, int C, int lane) {
    const int n = n_rows * C;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int e = lane + 64 * j;
        v[j] = (j < C && e < n) ? g[e] : 0.0f;
    }
}
template <int CB>
__device__ __forceinline__ void rows_to_lds(const float (&v)[CB], float *__restrict__ lds, int n_rows, int C, int stride, int lane) {
    const int n = n_rows * C;
    int row = lane / C, col = lane - row * C;
    const int drow = 64 / C, dcol = 64 - drow * C;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        if (j < C && lane + 64 * j < n) lds[row * stride + col] = v[j];
        row += drow; col += dcol;
        if (col >= C) { col -= C; ++row; }
    }
}

// The same two halves for a class-major block, g[class * sstride + sample] (the train step's private logit buffer: FieldIO::sem_stride): lane = sample.
template <int CB>
__device__ __forceinline__ void rows_load_soa(float (&v)[CB], const float *g, int64_t sstride, int n_rows, int C, int lane) {
#pragma unroll
    for (int j = 0; j < CB; ++j) v[j] = (j < C && lane < n_rows) ? g[j * sstride + lane] : 0.0f;
}
template <int CB>
__device__ __forceinline__ void rows_to_lds_soa(const float (&v)[CB], float *__restrict__ lds, int n_rows, int C, int stride, int lane) {
#pragma unroll
    for (int j = 0; j < CB; ++j)
        if (j < C && lane < n_rows) lds[lane * stride + j] = v[j];
}

template <int CB>
__global__ void __launch_bounds__(64) composite_fwd_kernel(const int64_t *__restrict__ starts, const int64_t *__restrict__ cnts,
                                                           const float *__restrict__ ts, const float *__restrict__ te,
                                                           const float *__restrict__ sig, const float *__restrict__ rgb,
                                                           const float *__restrict__ sem, int C, const float *__restrict__ bkgd,
                                                           float *__restrict__ o_rgb, float *__restrict__ o_acc,
                                                           float *__restrict__ o_dep, float *__restrict__ o_sem,
                                                           float *__restrict__ w_out, float *__restrict__ t_out,
                                                           float *__restrict__ a_out, int64_t sstride) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int stride = C | 1;
    float *w_lds = lds + 64 * stride;
    const int64_t r = blockIdx.x;
    const int64_t s0 = starts[r];
    const int cnt = (int)cnts[r];

    float carry = 0.f, aR = 0.f, aG = 0.f, aB = 0.f, aA = 0.f, aD = 0.f, aS = 0.f;
    // one block ahead: the scalars and the semantic rows of the next block are requested before this block's arithmetic
    float na = 0.f, nb = 0.f, nsg = 0.f, nc0 = 0.f, nc1 = 0.f, nc2 = 0.f;
    float rows[CB];
    auto request = [&](int base) {
        const int nv = min(64, cnt - base);
        const int64_t k = s0 + base + lane;
        na = nb = nsg = nc0 = nc1 = nc2 = 0.f;
        if (lane < nv) {
            na = ts[k]; nb = te[k]; nsg = sig[k];
            nc0 = rgb[3 * k]; nc1 = rgb[3 * k + 1]; nc2 = rgb[3 * k + 2];
        }
        if (C > 0) { if (sstride) rows_load_soa<CB>(rows, sem + s0 + base, sstride, nv, C, lane); else rows_load<CB>(rows, sem + (s0 + base) * C, nv, C, lane); }
    };
    if (cnt > 0) request(0);
    for (int base = 0; base < cnt; base += 64) {
        const int nv = min(64, cnt - base);
        const bool valid = lane < nv;
        const int64_t k = s0 + base + lane;
        const float a = na, b = nb, sg = nsg, c0 = nc0, c1 = nc1, c2 = nc2;
        if (C > 0) { if (sstride) rows_to_lds_soa<CB>(rows, lds, nv, C, stride, lane); else rows_to_lds<CB>(rows, lds, nv, C, stride, lane); }
        if (base + 64 < cnt) request(base + 64);
        const float s = sg * (b - a);
        const float incl = wave_inclusive_scan(s, lane);
        const float T = __expf(-((incl - s) + carry));
        const float alpha = 1.f - __expf(-s);
        const float w = valid ? T * alpha : 0.f;
        carry += __shfl(incl, 63, 64);
        if (valid) {
            w_out[k] = w;
            if (t_out) t_out[k] = T;
            if (a_out) a_out[k] = alpha;
        }
        aR += w * c0; aG += w * c1; aB += w * c2; aA += w; aD += w * ((a + b) * 0.5f);
        w_lds[lane] = w;
        __syncthreads();
        if (lane < C) {
            for (int i = 0; i < nv; ++i) aS += w_lds[i] * lds[i * stride + lane];
        }
        __syncthreads();
    }
    aR = wave_sum(aR); aG = wave_sum(aG); aB = wave_sum(aB); aA = wave_sum(aA); aD = wave_sum(aD);
    if (lane == 0) {
        const float bk0 = bkgd ? bkgd[0] : 0.f, bk1 = bkgd ? bkgd[1] : 0.f, bk2 = bkgd ? bkgd[2] : 0.f;
        o_rgb[3 * r] = aR + bk0 * (1.f - aA);
        o_rgb[3 * r + 1] = aG + bk1 * (1.f - aA);
        o_rgb[3 * r + 2] = aB + bk2 * (1.f - aA);
        o_acc[r] = aA;
        o_dep[r] = aD / fmaxf(aA, kEps);
    }
    if (lane < C) o_sem[r * C + lane] = aS;
}

template <int CB>
__global__ void __launch_bounds__(64) composite_bwd_kernel(const int64_t *__restrict__ starts, const int64_t *__restrict__ cnts,
                                                           const float *__restrict__ ts, const float *__restrict__ te,
                                                           const float *__restrict__ sig, const float *__restrict__ rgb,
                                                           const float *__restrict__ sem, int C, const float *__restrict__ bkgd,
                                                           const float *__restrict__ w_in, const float *__restrict__ t_in,
                                                           const float *__restrict__ o_acc,
                                                           const float *__restrict__ o_dep, const float *__restrict__ g_rgb,
                                                           const float *__restrict__ g_acc, const float *__restrict__ g_dep,
                                                           const float *__restrict__ g_sem, float *__restrict__ d_sig,
                                                           float *__restrict__ d_rgb, float *__restrict__ d_sem, int64_t sstride) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int stride = C | 1;
    float *gs_lds = lds + 64 * stride;
    const int64_t r = blockIdx.x;
    const int64_t s0 = starts[r];
    const int cnt = (int)cnts[r];
    if (cnt == 0) return;

    const float gc0 = g_rgb ? g_rgb[3 * r] : 0.f, gc1 = g_rgb ? g_rgb[3 * r + 1] : 0.f, gc2 = g_rgb ? g_rgb[3 * r + 2] : 0.f;
    const float A = o_acc[r];
    const float gD = g_dep ? g_dep[r] : 0.f;
    const float gDn = gD / fmaxf(A, kEps);
    float gA = g_acc ? g_acc[r] : 0.f;
    if (bkgd) gA -= gc0 * bkgd[0] + gc1 * bkgd[1] + gc2 * bkgd[2];
    if (A >= kEps) gA -= gD * o_dep[r] / A;
    if (lane < C) gs_lds[lane] = g_sem ? g_sem[r * C + lane] : 0.f;
    __syncthreads();

    float suffix = 0.f;
    const int n_blocks = (cnt + 63) / 64;
    // one block ahead (the sweep runs back to front), as in the forward kernel
    float na = 0.f, ne = 0.f, nsg = 0.f, nc0 = 0.f, nc1 = 0.f, nc2 = 0.f, nw = 0.f, nT = 0.f;
    float rows[CB];
    auto request = [&](int base) {
        const int nv = min(64, cnt - base);
        const int64_t k = s0 + base + lane;
        na = ne = nsg = nc0 = nc1 = nc2 = nw = nT = 0.f;
        if (lane < nv) {
            na = ts[k]; ne = te[k]; nsg = sig[k]; nw = w_in[k]; nT = t_in[k];
            nc0 = rgb[3 * k]; nc1 = rgb[3 * k + 1]; nc2 = rgb[3 * k + 2];
        }
        if (C > 0) { if (sstride) rows_load_soa<CB>(rows, sem + s0 + base, sstride, nv, C, lane); else rows_load<CB>(rows, sem + (s0 + base) * C, nv, C, lane); }
    };
    request((n_blocks - 1) * 64);
    for (int b = n_blocks - 1; b >= 0; --b) {
        const int base = b * 64;
        const int nv = min(64, cnt - base);
        const bool valid = lane < nv;
        const int64_t k = s0 + base + lane;
        const float a = na, e = ne, sg = nsg, c0 = nc0, c1 = nc1, c2 = nc2, w = nw, T = nT;
        if (C > 0) { if (sstride) rows_to_lds_soa<CB>(rows, lds, nv, C, stride, lane); else rows_to_lds<CB>(rows, lds, nv, C, stride, lane); }
        if (b > 0) request(base - 64);
        __syncthreads();
        const float dt = e - a;
        const float s = sg * dt;
        const float one_minus_alpha = __expf(-s);
        float g = gc0 * c0 + gc1 * c1 + gc2 * c2 + gA + gDn * ((a + e) * 0.5f);
        for (int c = 0; c < C; ++c) g += gs_lds[c] * lds[lane * stride + c];
        const float gw = valid ? g * w : 0.f;
        const float gincl = wave_inclusive_scan(gw, lane);
        const float total = __shfl(gincl, 63, 64);
        const float ds = g * T * one_minus_alpha - ((total - gincl) + suffix);
        suffix += total;
        if (valid) {
            d_sig[k] = ds * dt;
            if (d_rgb) { d_rgb[3 * k] = w * gc0; d_rgb[3 * k + 1] = w * gc1; d_rgb[3 * k + 2] = w * gc2; }
        }
        __syncthreads();
        if (d_sem) {      // (NULL, with d_rgb: the caller forms w * g_rgb[ray] and w * g_sem[ray] itself — the train step's backward-data kernel)
            for (int c = 0; c < C; ++c) lds[lane * stride + c] = w * gs_lds[c];
            __syncthreads();
            if (C > 0) stage_rows<false>(lds, d_sem + (s0 + base) * C, nv, C, stride, lane);
            __syncthreads();
        }
    }
}

size_t lds_bytes(int C) { return (size_t)(64 * (C | 1) + kMaxClasses) * sizeof(float); }

}  // namespace
}  // namespace mnf

using namespace mnf;

namespace mnf {
// sem_stride != 0: `sems` is class-major, sems[class * sem_stride + sample] (the train step's private buffer); 0: [sample][C] as the public entry points take it
int composite_train_forward_impl(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                 const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                 const float *sems, int64_t sem_stride, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                 float *out_rgb, float *out_acc, float *out_depth, float *out_sem, float *weights,
                                 float *trans, float *alphas, mnf_stream_t stream) {
    if (n_rays == 0) return MNF_OK;
    MNF_REQUIRE(n_classes >= 0 && n_classes <= kMaxClasses, "composite_train_forward: n_classes %d not in [0, %d]", n_classes,
                kMaxClasses);
    MNF_REQUIRE(chunk_starts && chunk_cnts && out_rgb && out_acc && out_depth && (out_sem || n_classes == 0),
                "composite_train_forward: null pointer");
    MNF_REQUIRE(n_samples == 0 || (t_starts && t_ends && sigmas && rgbs && weights && (sems || n_classes == 0)),
                "composite_train_forward: null sample pointer");
    ProfScope ps("composite_train_forward", as_stream(stream));
    if (n_classes <= 32)
        hipLaunchKernelGGL(composite_fwd_kernel<32>, dim3(n_rays), dim3(64), lds_bytes(n_classes), as_stream(stream), chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, n_classes, bkgd, out_rgb, out_acc, out_depth, out_sem,
                           weights, trans, alphas, sem_stride);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<kMaxClasses>, dim3(n_rays), dim3(64), lds_bytes(n_classes), as_stream(stream), chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, n_classes, bkgd, out_rgb, out_acc, out_depth, out_sem,
                           weights, trans, alphas, sem_stride);
    return launch_status("composite_fwd_kernel");
}
}  // namespace mnf

extern "C" int mnf_composite_train_forward(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                           const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                           const float *sems, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                           float *out_rgb, float *out_acc, float *out_depth, float *out_sem, float *weights,
                                           float *trans, float *alphas, mnf_stream_t stream) {
    return composite_train_forward_impl(chunk_starts, chunk_cnts, n_rays, t_starts, t_ends, sigmas, rgbs, sems, 0, n_classes, n_samples, bkgd, out_rgb, out_acc,
                                        out_depth, out_sem, weights, trans, alphas, stream);
}

namespace mnf {
int composite_train_backward_impl(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                            const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                            const float *sems, int64_t sem_stride, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                            const float *weights, const float *trans, const float *out_acc,
                                            const float *out_depth,
                                            const float *g_rgb, const float *g_acc, const float *g_depth, const float *g_sem,
                                            float *d_sigmas, float *d_rgbs, float *d_sems, mnf_stream_t stream) {
    if (n_rays == 0 || n_samples == 0) return MNF_OK;
    MNF_REQUIRE(n_classes >= 0 && n_classes <= kMaxClasses, "composite_train_backward: n_classes %d not in [0, %d]", n_classes,
                kMaxClasses);
    MNF_REQUIRE(chunk_starts && chunk_cnts && t_starts && t_ends && sigmas && rgbs && weights && trans && out_acc && out_depth &&
                    d_sigmas && (sems || n_classes == 0) && ((d_rgbs == nullptr) == (d_sems == nullptr) || n_classes == 0),
                "composite_train_backward: null pointer");
    ProfScope ps("composite_train_backward", as_stream(stream));
    if (n_classes <= 32)
        hipLaunchKernelGGL(composite_bwd_kernel<32>, dim3(n_rays), dim3(64), lds_bytes(n_classes), as_stream(stream), chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, n_classes, bkgd, weights, trans, out_acc, out_depth, g_rgb, g_acc,
                           g_depth, g_sem, d_sigmas, d_rgbs, d_sems, sem_stride);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<kMaxClasses>, dim3(n_rays), dim3(64), lds_bytes(n_classes), as_stream(stream), chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, n_classes, bkgd, weights, trans, out_acc, out_depth, g_rgb, g_acc,
                           g_depth, g_sem, d_sigmas, d_rgbs, d_sems, sem_stride);
    return launch_status("composite_bwd_kernel");
}
}  // namespace mnf

extern "C" int mnf_composite_train_backward(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                            const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                            const float *sems, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                            const float *weights, const float *trans, const float *out_acc,
                                            const float *out_depth,
                                            const float *g_rgb, const float *g_acc, const float *g_depth, const float *g_sem,
                                            float *d_sigmas, float *d_rgbs, float *d_sems, mnf_stream_t stream) {
    return composite_train_backward_impl(chunk_starts, chunk_cnts, n_rays, t_starts, t_ends, sigmas, rgbs, sems, 0, n_classes, n_samples, bkgd, weights, trans,
                                         out_acc, out_depth, g_rgb, g_acc, g_depth, g_sem, d_sigmas, d_rgbs, d_sems, stream);
}
