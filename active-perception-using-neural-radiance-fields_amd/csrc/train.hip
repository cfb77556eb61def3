// Backward pass of the radiance field (the tiny-cuda-nn backward the reference reaches through
// `loss.backward()`, scripts/pipeline.py:518; forward call sites perception/models/radiance_fields/ngp.py:171-238):
//
//   dgrad_kernel   dL/d(rgb, sigma, sem) per sample -> pre-activation gradients of every layer (kept in MFMA
//                  registers from layer to layer exactly like the forward chain, with the TRANSPOSED weight
//                  fragments as the A operand), stored feature-major for the weight-gradient GEMMs, and
//                  dL/d(hash features) per sample.
//   wgrad_kernel   dW[n][k] = sum_samples dOut^T[n][c] * In^T[k][c]: the contraction runs over samples, and with
//                  feature-major fp16 activations both MFMA operands are plain 16-byte loads; split over sample
//                  chunks, fp32 atomics into the flat parameter-gradient vectors.
//   hash_bwd_walk_kernel  trilinear scatter of dL/d(features) into the dense fp32 table gradient (float atomics).
//
// Gradients of activations travel in fp16 scaled by `loss_scale` (tcnn does the same with its default scale of
// 128); parameter gradients are accumulated and returned in fp32, un-scaled.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "field_dev.h"

namespace mnf {
// The train step's output gradients in factored form: d_rgb[s] = w[s] * g_rgb[ray[s]], d_sem[s][c] = w[s] * g_sem[ray[s]][c] — what sem_rendering's backward
// (composite_train.hip) would write per sample, 128 bytes each at 29 classes, and the backward-data kernel would read back at a 116-byte lane stride.  Given the
// factors the kernel forms the products itself (same fp32 multiplications): 12 bytes per sample plus per-ray vectors that stay in cache.
struct FactoredGrad { const float *w; const int64_t *ray; const float *g_rgb, *g_sem; };
// set by trainstep.hip for the NEXT backward on this thread; taken (and cleared) by backward_impl
void set_factored_output_gradient(const FactoredGrad &fg);
bool take_factored_output_gradient(FactoredGrad &fg);
}  // namespace mnf

MNF_DT_BEGIN

// ------------------------------------------------------------------ transposed-fragment bookkeeping
template <int W, int NH>
struct LayoutT {
    static constexpr int Wh = W / 2, RT = W / 32, RTh = Wh / 32, KSW = W / 16, KSh = Wh / 16;
    static constexpr int o_r3 = 0;                         // rgb head out^T : RTh x 1
    static constexpr int o_r2 = o_r3 + RTh * 1;            // rgb head hid^T : RTh x KSh
    static constexpr int o_r1 = o_r2 + RTh * KSh;          // rgb head in^T (geo rows) : 1 x KSh
    static constexpr int o_s3 = o_r1 + KSh;                // sem head out^T : RTh x 2
    static constexpr int o_s2 = o_s3 + RTh * 2;
    static constexpr int o_s1 = o_s2 + RTh * KSh;
    static constexpr int o_bo = o_s1 + KSh;                // base out^T : RT x 1
    static constexpr int o_bh = o_bo + RT * 1;             // base hidden^T l : RT x KSW each, l = 0..NH-2
    static constexpr int o_b1 = o_bh + (NH - 1) * RT * KSW;  // base in^T : 2 x KSW
    static constexpr int blocks = o_b1 + 2 * KSW;
};

// The number of samples of a training step may live on the device (mnf_train_step never brings it to the host): `n` is then the
// upper bound the launch was sized for and `n_dev` the actual count, read with one scalar load.
__device__ __forceinline__ int64_t count_here(int64_t n_cap, const int64_t *n_dev) {
    if (!n_dev) return n_cap;
    typedef const int64_t __attribute__((address_space(4))) *CntPtr;
    const int64_t nd = *(CntPtr)(uintptr_t)n_dev;
    return nd < n_cap ? nd : n_cap;
}

// The backward's kernels can be issued for one of `n_chunks` contiguous ranges of 64-sample tiles (whole tiles: dgrad's unit), so that the
// hash-table scatter of a range starts behind the backward-data launch of THAT range while the next range is still in dgrad.  The ranges are
// fractions of the actual (device-side) count.
__device__ __forceinline__ void chunk_tiles(int64_t n, int chunk, int n_chunks, int64_t &t0, int64_t &t1) {
    const int64_t n_tiles = (n + kWaveSamples - 1) / kWaveSamples;
    t0 = n_chunks > 1 ? n_tiles * chunk / n_chunks : 0;
    t1 = n_chunks > 1 ? n_tiles * (chunk + 1) / n_chunks : n_tiles;
}

struct BwdArgs {
    const half8 *fragsT;
    const float *d_rgb, *d_sigma, *d_sem;   // [N,3], [N], [N,C]
    const float *rgb, *sigma;               // forward outputs [N,3], [N]
    float *dX;                              // [16 levels][Np][4] fp32, un-scaled (level-major for hash_bwd)
    int64_t n;
    const int64_t *n_dev;
    int C;
    float loss_scale;
    int chunk, n_chunks;
    TrainBuf train;
    FactoredGrad fg;                        // fg.w != NULL: d_rgb / d_sem are not read, the products are formed here
};

// fp32 -> fp16 with saturation at the largest finite half: a loss-scaled gradient that leaves the fp16 range is clipped
// instead of becoming Inf (which would poison every weight gradient of the step)
#ifdef MNF_BF16
__device__ __forceinline__ half_t sat_half(float v) { return (half_t)v; }   // bf16 has fp32's range
#else
__device__ __forceinline__ half_t sat_half(float v) { return (half_t)(v != v ? v : fminf(fmaxf(v, -65504.0f), 65504.0f)); }   // NaN stays NaN
#endif

// acc[ct] = sum_ks A(rt, ks) * b[ct][ks] for one 32-row tile
template <int KS>
__device__ __forceinline__ void dense_tile(const half8 *__restrict__ w_lds, int lane, const half8 (&b)[CT][KS], f32x16 (&acc)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const half8 a = w_lds[ks * 64 + lane];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma(a, b[ct][ks], acc[ct]);
    }
}

// One backward layer through a ReLU: dz[ct][2rt+s] = pack(W^T(rt,:) * dOut[ct]) masked by the saved activation,
// stored feature-major at row0 for the weight gradient.
template <int RT_OUT, int KS, int NP>
__device__ __forceinline__ void dense_mask(const half8 *__restrict__ w_lds, int lane, int h, const half8 (&b)[CT][KS],
                                           const u32x4 (&mrec)[NP], int mblock0, const TrainBuf &tb, int64_t tile, int row0,
                                           half_t *stage, half8 (&o)[CT][RT_OUT * 2]) {
#pragma unroll
    for (int rt = 0; rt < RT_OUT; ++rt) {
        f32x16 acc[CT];
        dense_tile<KS>(w_lds + rt * KS * 64, lane, b, acc);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                // byte (mblock0 + rt * 2 + s) * CT + ct of the lane's mask record (indices are constants once the loops are unrolled)
                const int mb = (mblock0 + rt * 2 + s) * CT + ct;
                const uint32_t m = mrec[mb >> 4][(mb >> 2) & 3] >> (8 * (mb & 3));
                // pack two values per conversion first, then clear the halves whose ReLU was inactive with ONE and: element 2i is
                // bit i of the mask byte, element 2i+1 bit 4+i (frag_mask_bit); a sign-extended 1-bit field extract is an
                // all-ones / all-zeros word.  (Selecting per element before the conversion cost 4.5 instructions per value.)
                u32x4 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const half2 p = {(half_t)acc[ct][8 * s + 2 * i], (half_t)acc[ct][8 * s + 2 * i + 1]};
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m, i, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)m, 4 + i, 1);
                    w[i] = __builtin_bit_cast(uint32_t, p) & ((lo & 0x0000FFFFu) | (hi & 0xFFFF0000u));
                }
                o[ct][rt * 2 + s] = __builtin_bit_cast(half8, w);
            }
            save_pair<true>(tb, tile, row0 + 16 * (rt * 2 + s), lane, stage, o[0][rt * 2 + s], o[1][rt * 2 + s]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the row-tile iterations from interleaving (register pressure)
    }
}

template <int W, int NH>
__global__ void __launch_bounds__(kThreads, 2) dgrad_kernel(const BwdArgs args) {
    using L = LayoutT<W, NH>;
    using T = TrainLayout<W, NH>;
    __shared__ half8 s_w[L::blocks * 64];
    __shared__ half_t s_stage[kWavesPerBlock * kStageHalves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    half_t *stage = s_stage + wave * kStageHalves;
    const int64_t n = count_here(args.n, args.n_dev);
    int64_t tile0, n_tiles;
    chunk_tiles(n, args.chunk, args.n_chunks, tile0, n_tiles);
    if (tile0 + (int64_t)blockIdx.x * kWavesPerBlock >= n_tiles) return;
    for (int i = threadIdx.x; i < L::blocks * 64; i += kThreads) s_w[i] = args.fragsT[i];
    __syncthreads();
    const float ls = args.loss_scale;

    for (int64_t tile = tile0 + (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t fcol0 = tile * kWaveSamples + c;
        // the lane's ReLU-mask record of the tile: mask_bytes / 16 loads here instead of a byte load per fragment where it is used
        constexpr int kMaskPieces = T::mask_bytes / 16;
        u32x4 mrec[kMaskPieces];
        {
            const u32x4 *mp = reinterpret_cast<const u32x4 *>(args.train.masks + (tile * 64 + lane) * T::mask_bytes);
#pragma unroll
            for (int i = 0; i < kMaskPieces; ++i) mrec[i] = mp[i];
        }
        // ---- output-layer gradients, built directly as natural-order B fragments ----
        half8 dyr[CT][1], dys[CT][2];
        float dlogit[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int64_t col = fcol0 + 32 * ct;
            const bool ok = col < n;
#pragma unroll
            for (int j = 0; j < 8; ++j) dyr[ct][0][j] = (half_t)0.0f;
            if (args.fg.w) {      // (wave-uniform) factored output gradients: one weight per sample, the ray's vectors from cache
                const float wv = ok ? args.fg.w[col] : 0.0f;
                const int64_t ry = ok ? args.fg.ray[col] : 0;
                if (ok && h == 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float y = args.rgb[3 * col + k];
                        dyr[ct][0][k] = sat_half(wv * args.fg.g_rgb[3 * ry + k] * y * (1.0f - y) * ls);   // sigmoid'
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = 16 * s + 8 * h + j;
                        dys[ct][s][j] = (ok && row < args.C) ? sat_half(wv * args.fg.g_sem[ry * args.C + row] * ls) : (half_t)0.0f;
                    }
            } else {
            if (ok && h == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float y = args.rgb[3 * col + k];
                    dyr[ct][0][k] = sat_half(args.d_rgb[3 * col + k] * y * (1.0f - y) * ls);   // sigmoid'
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = 16 * s + 8 * h + j;
                    dys[ct][s][j] = (ok && row < args.C) ? sat_half(args.d_sem[col * args.C + row] * ls) : (half_t)0.0f;
                }
            }
            // trunc_exp backward (ngp.py:34-39): g * exp(min(x, 15)) with exp(x) = sigma (0 outside the aabb)
            dlogit[ct] = ok ? args.d_sigma[col] * fminf(args.sigma[col], 3269017.3724721107f) * ls : 0.0f;
        }
        save_pair<false>(args.train, tile, T::rdYr, lane, stage, dyr[0][0], dyr[1][0]);
        save_pair<false>(args.train, tile, T::rdYs, lane, stage, dys[0][0], dys[1][0]);
        save_pair<false>(args.train, tile, T::rdYs + 16, lane, stage, dys[0][1], dys[1][1]);
        // ---- heads ----
        half8 dz2[CT][L::KSh], dz1[CT][L::KSh];
        f32x16 dgeo_r[CT], dgeo_s[CT];
        dense_mask<L::RTh, 1>(s_w + L::o_r3 * 64, lane, h, dyr, mrec, T::mHH2, args.train, tile, T::rdZr2, stage, dz2);
        dense_mask<L::RTh, L::KSh>(s_w + L::o_r2 * 64, lane, h, dz2, mrec, T::mHH1, args.train, tile, T::rdZr1, stage, dz1);
        dense_tile<L::KSh>(s_w + L::o_r1 * 64, lane, dz1, dgeo_r);
        dense_mask<L::RTh, 2>(s_w + L::o_s3 * 64, lane, h, dys, mrec, T::mHS2, args.train, tile, T::rdZs2, stage, dz2);
        dense_mask<L::RTh, L::KSh>(s_w + L::o_s2 * 64, lane, h, dz2, mrec, T::mHS1, args.train, tile, T::rdZs1, stage, dz1);
        dense_tile<L::KSh>(s_w + L::o_s1 * 64, lane, dz1, dgeo_s);
        // ---- base output gradient: geo rows from both heads, row 0 = density logit ----
        half8 dbo[CT][1];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dbo[ct][0][j] = (half_t)(dgeo_r[ct][j] + dgeo_s[ct][j]);
            if (h == 0) dbo[ct][0][0] = sat_half(dlogit[ct]);
        }
        save_pair<true>(args.train, tile, T::rdBO, lane, stage, dbo[0][0], dbo[1][0]);
        // ---- base MLP ----
        half8 dz[CT][L::KSW];
        dense_mask<L::RT, 1>(s_w + L::o_bo * 64, lane, h, dbo, mrec, T::mH0 + (NH - 1) * L::KSW, args.train, tile,
                             T::rdZ0 + (NH - 1) * W, stage, dz);
#pragma unroll
        for (int l = NH - 2; l >= 0; --l) {
            half8 dn[CT][L::KSW];
            dense_mask<L::RT, L::KSW>(s_w + (L::o_bh + l * L::RT * L::KSW) * 64, lane, h, dz, mrec, T::mH0 + l * L::KSW,
                                      args.train, tile, T::rdZ0 + l * W, stage, dn);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int k = 0; k < L::KSW; ++k) dz[ct][k] = dn[ct][k];
        }
        // ---- gradient of the 64 hash features: rows 32rt + 8g + 4h + i == level (8rt + 2g + h), feature i ----
        const float inv = 1.0f / ls;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 acc[CT];
            dense_tile<L::KSW>(s_w + (L::o_b1 + rt * L::KSW) * 64, lane, dz, acc);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float4 *dst = reinterpret_cast<float4 *>(args.dX) + fcol0 + 32 * ct;   // 32 lanes x 16 B contiguous per level
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = {acc[ct][4 * g] * inv, acc[ct][4 * g + 1] * inv, acc[ct][4 * g + 2] * inv, acc[ct][4 * g + 3] * inv};
                    dst[(int64_t)(8 * rt + 2 * g + h) * args.train.Np] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ weight gradients
struct WgradJob {
    int32_t dout_row0, in_row0;   // rows of ACT
    int32_t buf;                  // 0 base, 1 head, 2 sem
    int32_t param_off, stride;    // dW[n][k] lands at param_off + n*stride + colmap[k]
    int32_t n_valid;              // rows of this 32-row tile that exist in the parameter matrix
    int32_t n0;
    int32_t colmap[32];           // parameter column of in-row k of this tile, -1 = none
};

// A group = up to 2 x 2 jobs of one matrix that share their dOut row tiles and their In row tiles.
struct WgradGroup {
    int32_t no, ni;       // dOut tiles and In tiles in the group (1 or 2 each)
    int32_t job[2][2];    // job[o][i]
};

// one wave = one (group, chunk of 64-sample tiles): D[o][i][32][32] += dOut_o^T[32][16 samples] * In_i^T[16 samples][32],
// four MFMA steps per tile and (o, i); the 32 rows of each operand are one contiguous 4 KB block of the tile-major
// activation matrix.  Register blocking 2 x 2 reads each activation row once per two output tiles: the kernel is bound
// by re-reading the activations (44 jobs x 64 rows -> 14 groups, -43 % bytes), not by the MFMAs.
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradJob *__restrict__ jobs, const WgradGroup *__restrict__ groups, int n_groups,
                                                    int split, const half_t *__restrict__ act, int64_t n_cap, const int64_t *n_dev, int rows,
                                                    float inv_scale, float *g0, float *g1, float *g2, float *__restrict__ partials) {
    const int lane = threadIdx.x & 63;
    // a workgroup = four consecutive groups on the SAME range of tiles: groups of one matrix share operand tiles (each dOut / In tile of a hidden matrix is read by
    // two of its four groups), and read by waves of one workgroup at about the same time the second read is a cache hit instead of a second trip to HBM
    const int grp = (blockIdx.x / split) * 4 + (threadIdx.x >> 6);
    const int part = blockIdx.x % split;
    if (grp >= n_groups) return;
    const int wid = grp * split + part;                       // slot of this wave's partial sums (deterministic mode)
    const int64_t n_tiles = (count_here(n_cap, n_dev) + 63) / 64;
    const WgradGroup gp = groups[grp];
    const int64_t t0 = n_tiles * part / split, t1 = n_tiles * (part + 1) / split;
    const int r = lane & 31, h = lane >> 5;
    const bool o2 = gp.no > 1, i2 = gp.ni > 1;   // wave-uniform
    const int dout0 = jobs[gp.job[0][0]].dout_row0, dout1 = o2 ? jobs[gp.job[1][0]].dout_row0 : dout0;
    const int in0 = jobs[gp.job[0][0]].in_row0, in1 = i2 ? jobs[gp.job[0][1]].in_row0 : in0;
    f32x16 acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[o][i][k] = 0.0f;
    for (int64_t t = t0; t < t1; ++t) {
        const half_t *base = act + (t * rows + r) * 64 + 8 * h;
        half8 a[2][4], b[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[0][q] = *reinterpret_cast<const half8 *>(base + dout0 * 64 + 16 * q);
            b[0][q] = *reinterpret_cast<const half8 *>(base + in0 * 64 + 16 * q);
        }
        if (o2)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[1][q] = *reinterpret_cast<const half8 *>(base + dout1 * 64 + 16 * q);
        if (i2)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[1][q] = *reinterpret_cast<const half8 *>(base + in1 * 64 + 16 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][0] = mfma(a[0][q], b[0][q], acc[0][0]);
        if (i2)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[0][1] = mfma(a[0][q], b[1][q], acc[0][1]);
        if (o2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[1][0] = mfma(a[1][q], b[0][q], acc[1][0]);
            if (i2)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[1][1] = mfma(a[1][q], b[1][q], acc[1][1]);
        }
    }
    if (partials) {   // deterministic mode: this wave's partial sums go to its own slot; wgrad_reduce_kernel adds the slots in a fixed order
        float *dst = partials + ((size_t)wid * 4) * 1024 + lane * 16;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (o >= gp.no || i >= gp.ni) continue;
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[(o * 2 + i) * 1024 + k] = acc[o][i][k];
            }
        return;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (o >= gp.no || i >= gp.ni) continue;
            const WgradJob &jb = jobs[gp.job[o][i]];
            float *g = jb.buf == 0 ? g0 : (jb.buf == 1 ? g1 : g2);
            const int col = jb.colmap[r];
            if (col < 0) continue;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
                if (row < jb.n_valid) atomicAdd(g + jb.param_off + (int64_t)(jb.n0 + row) * jb.stride + col, acc[o][i][k] * inv_scale);
            }
        }
}

// deterministic mode: one thread per element of a (group, o, i) output tile sums the `split` partial slots front to back
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgradJob *__restrict__ jobs, const WgradGroup *__restrict__ groups, int n_groups, int split,
                                                           const float *__restrict__ partials, float inv_scale, float *g0, float *g1, float *g2) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // ((group * 4 + oi) * 64 + lane) * 16 + k
    if (e >= (int64_t)n_groups * 4 * 1024) return;
    const int k = (int)(e & 15), lane = (int)((e >> 4) & 63), oi = (int)((e >> 10) & 3), grp = (int)(e >> 12);
    const WgradGroup gp = groups[grp];
    const int o = oi >> 1, i = oi & 1;
    if (o >= gp.no || i >= gp.ni) return;
    const WgradJob &jb = jobs[gp.job[o][i]];
    const int r = lane & 31, h = lane >> 5;
    const int col = jb.colmap[r];
    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
    if (col < 0 || row >= jb.n_valid) return;
    float sum = 0.0f;
    for (int part = 0; part < split; ++part) sum += partials[(((size_t)grp * split + part) * 4 + oi) * 1024 + lane * 16 + k];
    float *g = jb.buf == 0 ? g0 : (jb.buf == 1 ? g1 : g2);
    g[jb.param_off + (int64_t)(jb.n0 + row) * jb.stride + col] = sum * inv_scale;   // every weight belongs to exactly one tile
}

// ------------------------------------------------------------------ hash-grid gradient scatter
// Deterministic mode (mnf_train_opts.deterministic): the table gradient is accumulated in 64-bit fixed point (value * 2^56, integer
// atomics: associative, so the result does not depend on the order in which the adds arrive), then converted to fp32 once.  The
// memory-side atomic unit takes 8-byte integer adds at the request rate of 4-byte float adds (tools/atomic_bench.hip), so the mode
// costs the wider zero-fill and the conversion pass, not scatter time.  Range +-128 (saturating), resolution 1.4e-17.
constexpr float kFixScale = 72057594037927936.0f;        // 2^56
constexpr double kFixInv = 1.0 / 72057594037927936.0;
__device__ __forceinline__ unsigned long long to_fixed(float v) {
    const float c = fminf(fmaxf(v, -127.99999f), 127.99999f);
    return (unsigned long long)__float2ll_rn(c * kFixScale);              // NaN -> 0 (llrint of NaN is unspecified: guard below)
}

struct HashBwdArgs {
    const float *positions;   // aabb-normalised positions xn = (x - aabb_min) / (aabb_max - aabb_min) [N,3] (normalize_kernel): the
                              // three divisions are done once per sample, not once per sample, level and lane
    const float *dX;      // [16][Np][4]
    int64_t Np;
    int chunk, n_chunks;  // this launch covers tile range `chunk` of `n_chunks` (chunk_tiles)
    int level0;           // first level of this launch (experiments; 0 in production)
    int n_walk_levels;    // levels level0 .. level0 + n_walk_levels - 1
    float *repl;          // [kReplicas][repl_floats]: private copies of the coarsest levels' gradient (see below)
    uint32_t repl_floats;
    int repl_levels;
    float *g_table;       // fp32 [entries][4]
    int64_t n;
    const int64_t *n_dev;
    unsigned long long *flush_count;   // diagnostic build (MNF_SCATTER_COUNT=1): quad atomics issued per level, else NULL
    unsigned long long *q_table, *q_repl;   // deterministic mode: fixed-point table gradient [entries][4] and replicas, else NULL
    unsigned long long *q_bad;              // deterministic mode: number of non-finite contributions (they cannot be represented in fixed point)
    float aabb[6];
    LevelMeta levels[16];
};

// ngp.py:177-178 once per sample for the scatter below (the same expression as the forward kernel's fetch_sample)
__global__ void __launch_bounds__(256) normalize_kernel(const float *__restrict__ pos, int64_t n_cap, const int64_t *n_dev, float a0, float a1, float a2,
                                                        float a3, float a4, float a5, float *__restrict__ xn) {
    const float lo[3] = {a0, a1, a2}, hi[3] = {a3, a4, a5};
    const int64_t n = count_here(n_cap, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n; i += (int64_t)blockDim.x * gridDim.x) {
        const int d = (int)(i % 3);
        xn[i] = (pos[i] - lo[d]) / (hi[d] - lo[d]);
    }
}

// Samples arrive sorted by ray and by distance along the ray, so consecutive samples usually sit in the same grid cell
// of a level (always at the coarse levels, often at the fine ones).  A scattered float atomic is the slow operation
// here (MI355X: ~17x below the contiguous rate) and the kernel is bound by the rate of memory-side read-modify-writes,
// so everything below is about issuing fewer of them.  (History: one atomic per lane and corner 94 ms; runs of lanes in
// one cell summed by a segmented lane scan, tails only 21 ms; LDS transpose to 16-byte quad atomics 3.7 ms; the walk
// below with face sharing 3.4 ms.)
// The coarsest dense levels have a few thousand entries that every ray of a camera crosses near its origin: their
// atomics pile up on the same addresses (level 0 alone cost as much as the finest level).  Workgroups therefore add
// into one of kReplicas private copies of those levels, which a small kernel folds into the gradient afterwards.
constexpr int kReplicas = 16;
constexpr uint32_t kReplMaxEntries = 131072;

__global__ void __launch_bounds__(256) fold_replicas_fixed_kernel(const unsigned long long *__restrict__ repl, uint32_t repl_floats,
                                                                  unsigned long long *__restrict__ q_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= repl_floats) return;
    unsigned long long acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += repl[(size_t)r * repl_floats + i];
    if (acc) q_table[i] += acc;
}

// fixed point -> fp32 gradient (overwrites the table part of the gradient vector)
__global__ void __launch_bounds__(256) fixed_to_float_kernel(const long long *__restrict__ q, float *__restrict__ g, int64_t n,
                                                             const unsigned long long *__restrict__ bad) {
    const bool poisoned = *bad != 0;     // a non-finite contribution was met: the gradient says so (the NaN guard of pipeline.py:520-529 sees it)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)
        g[i] = poisoned && i == 0 ? __builtin_nanf("") : (float)((double)q[i] * kFixInv);
}

__global__ void __launch_bounds__(256) fold_replicas_kernel(const float *__restrict__ repl, uint32_t repl_floats, float *__restrict__ g_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= repl_floats) return;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < kReplicas; ++r) acc += repl[(size_t)r * repl_floats + i];
    if (acc != 0.f) g_table[i] += acc;
}

// The scatter as a WALK along the sample order.  A half-wave owns a chunk of kWalkChunk consecutive samples of one level;
// its 32 lanes are the 8 corners x 4 features of a grid cell.  Samples are visited in order, every lane accumulating
// w_corner * g_feature for the open cell in a register; when the cell changes the sums go out as quad-atomics (the four
// features of a corner are 16 contiguous bytes).  What the lane-scan form cannot do: when the ray steps into a FACE
// neighbour, the four corners on the shared face stay in registers (they move to the partner lanes, lane ^ 4 / 8 / 16)
// and only the four corners left behind are flushed; edge and vertex neighbours are reached by two or three such
// crossings (6 or 7 flushed corners instead of 8).  The kernel is bound by the rate of memory-side read-modify-writes.  Ray structure is not needed: runs are found by comparing
// consecutive cells, and a chunk boundary only costs one extra flush.
constexpr int kWalkChunk = 128;
#ifndef MNF_EXP_SCATTER
#define MNF_EXP_SCATTER 0
#endif

template <bool DET>
__global__ void __launch_bounds__(256) hash_bwd_walk_kernel(const HashBwdArgs args) {
    const int sub = threadIdx.x & 31;                 // lane inside the half-wave
    const int corner = sub >> 2, feat = sub & 3;
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
    const int64_t n_cnt = count_here(args.n, args.n_dev);
    int64_t rt0, rt1;
    chunk_tiles(n_cnt, args.chunk, args.n_chunks, rt0, rt1);
    const int64_t s_lo = rt0 * kWaveSamples, n_all = rt1 * kWaveSamples < n_cnt ? rt1 * kWaveSamples : n_cnt;       // samples [s_lo, n_all) of this launch
    // A bounded number of workgroups walks all (level, chunk) pairs (level by level, so that the workgroups of the moment add into the same few
    // megabytes): the kernel is bound by the atomic unit, not by waves in flight, and the wave slots it leaves are what lets the binned passes
    // and the weight gradients run beside it instead of behind it.
    const int64_t chunks = n_all > s_lo ? (n_all - s_lo + kWalkChunk - 1) / kWalkChunk : 0;
    const int64_t items = chunks * args.n_walk_levels;
    for (int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; item < items; item += ((int64_t)gridDim.x * blockDim.x) >> 5) {
    const int64_t chunk = item % chunks;
    const int l = (int)(item / chunks) + args.level0;
    const int64_t i0 = s_lo + chunk * kWalkChunk;
    const int64_t i1 = i0 + kWalkChunk < n_all ? i0 + kWalkChunk : n_all;
    const LevelMeta m = args.levels[l];
    const uint32_t replica = (uint32_t)((chunk >> 3) % kReplicas);
    float *const g_dst = l < args.repl_levels ? args.repl + (size_t)replica * args.repl_floats : args.g_table;
    unsigned long long *const q_dst = !DET ? nullptr : (l < args.repl_levels ? args.q_repl + (size_t)replica * args.repl_floats : args.q_table);
    const float *gl = args.dX + ((int64_t)l * args.Np) * 4 + feat;

    float acc = 0.0f;
    bool open = false;
    int cx = 0, cy = 0, cz = 0;                       // cell of the open run
    auto flush = [&]() {                              // this lane's corner of the open cell
        if (acc != 0.0f) {
            const uint32_t px = (uint32_t)cx + bx, py = (uint32_t)cy + by, pz = (uint32_t)cz + bz;
            uint32_t idx;
            if (m.hashed) {
                idx = (px ^ (py * 2654435761u) ^ (pz * 805459861u)) & (m.size - 1u);
            } else {
                idx = px + py * m.res + pz * m.res * m.res;
                uint32_t q = __umulhi(m.div_magic, idx);
                q = (((idx - q) >> 1) + q) >> m.div_shift;
                idx -= q * m.size;
            }
#ifdef MNF_DIAG
            if (args.flush_count && feat == 0) atomicAdd(&args.flush_count[l], 1ull);
#endif
#if MNF_EXP_SCATTER != 1      /* timing experiment: 1 = the walk without its atomics */
            if (DET) {
                // a contribution outside the fixed-point range would saturate silently into a wrong finite gradient: it poisons the
                // step like a non-finite one
                if (fabsf(acc) < 127.99999f) atomicAdd(q_dst + ((size_t)(m.offset + idx) << 2) + feat, to_fixed(acc));
                else atomicAdd(args.q_bad, 1ull);          // too large / NaN / Inf: reported through the converted gradient (fixed_to_float_kernel)
            }
            else atomicAdd(g_dst + ((size_t)(m.offset + idx) << 2) + feat, acc);
#else
            if (acc == 123.456f) g_dst[0] = acc;
#endif
        }
        acc = 0.0f;
    };
    constexpr int B = 8;                              // samples fetched ahead of the sequential part
    for (int64_t i = i0; i < i1; i += B) {
        float sx[B], sy[B], sz[B], g[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int64_t k = i + b < i1 ? i + b : i1 - 1;
            sx[b] = args.positions[3 * k]; sy[b] = args.positions[3 * k + 1]; sz[b] = args.positions[3 * k + 2];
            g[b] = gl[k * 4];
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (i + b >= i1) break;
            // position -> cell and fractions, exactly as hash_prep
            const float x0 = __builtin_fmaf(m.scale, sx[b], 0.5f);
            const float x1 = __builtin_fmaf(m.scale, sy[b], 0.5f);
            const float x2 = __builtin_fmaf(m.scale, sz[b], 0.5f);
            const float f0 = floorf(x0), f1 = floorf(x1), f2 = floorf(x2);
            const int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
            const int dx = c0 - cx, dy = c1 - cy, dz = c2 - cz;
            if (!open) {
                open = true;
                cx = c0; cy = c1; cz = c2;
            } else if ((dx | dy | dz) != 0) {         // half-wave uniform
                const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
                if ((ax | ay | az) > 1) {
                    flush();                           // a jump (new ray, or a step longer than a cell)
                    cx = c0; cy = c1; cz = c2;
                } else {
                    // one face crossing per changed axis, through the intermediate cells: the corners on the far side of
                    // the cell being left are the near side of the next one and stay in registers (partner lane)
                    auto face_move = [&](int bit, int dir, int mask) {
                        const bool stays_behind = dir > 0 ? bit == 0 : bit == 1;
                        if (stays_behind) flush();
                        const float moved = __shfl_xor(acc, mask, 64);
                        acc = stays_behind ? moved : 0.0f;
                    };
                    if (dx) { face_move(bx, dx, 4); cx = c0; }
                    if (dy) { face_move(by, dy, 8); cy = c1; }
                    if (dz) { face_move(bz, dz, 16); cz = c2; }
                }
            }
            const float fx = x0 - f0, fy = x1 - f1, fz = x2 - f2;
            const float wx = bx ? fx : 1.0f - fx, wy = by ? fy : 1.0f - fy, wz = bz ? fz : 1.0f - fz;
            acc += (((1.0f * wx) * wy) * wz) * g[b];
        }
    }
    flush();
    }
}

// ---- the large hashed levels: binned scatter -------------------------------------------------------------------------------------
// The walk above is bound by the memory-side atomic unit (~21 G 64-byte requests/s whatever they carry, profiles/r03_atomic_microbench.txt),
// and on the fine levels nearly every corner of every sample is a request of its own (hashed neighbours are not neighbours in memory).
// Those levels go through memory instead: a level's table (2^k entries) is cut into bins of kBinEntries consecutive entries whose fp32
// gradient fits in LDS.  Pass A (bin_items_kernel) writes one 16-byte item per (sample, corner) — the four feature gradients with the entry's
// index inside its bin kept in their lowest mantissa bits — to the bin's list (slots reserved per workgroup: one returning atomic per bin
// and workgroup).  Pass B (bin_accumulate_kernel), one workgroup per (level, bin), streams its list, adds into LDS and adds
// the bin's sums to the gradient with plain 16-byte read-modify-writes: it is the only writer of those entries at that time.  The LDS sums
// are DOUBLES: ds_add_f32 retires about one lane every 2.5 clocks per CU (4 x slower than the loads feed it), ds_add_f64 and the integer
// adds run at the rate of the loads (tools/bin_bench.hip, profiles/r03_bin_bench.txt).  A list that
// is full sends its items to the gradient with float atomics as before, so the bins only need to be sized for a uniform hash.
#ifndef MNF_BIN_LOG2
#define MNF_BIN_LOG2 12
#endif
constexpr int kBinEntriesLog2 = MNF_BIN_LOG2;
constexpr uint32_t kBinEntries = 1u << kBinEntriesLog2;     // x 4 features x 8 bytes = 128 KB of LDS per workgroup of pass B
constexpr int kMaxBins = 512;                                // per level: tables up to 2^21 entries
constexpr int kBinThreadsB = MNF_BIN_LOG2 >= 12 ? 1024 : 512;

struct BinArgs {
    const float *positions;   // normalised positions [N,3]
    const float *dX;          // [16][Np][4]
    int64_t Np, n;
    const int64_t *n_dev;
    int chunk, n_chunks;      // pass A of this launch covers tile range `chunk` of `n_chunks` (chunk_tiles); pass B runs once, behind the last pass A
    int level0, n_levels;     // levels level0 .. level0 + n_levels - 1, all hashed with size a multiple of kBinEntries
    float4 *items;            // [n_levels][bins of that level][cap]: level k's lists start at item_base[k]
    uint32_t *cursors;        // [n_levels][kMaxBins] items handed out per list (may exceed cap: the excess went to the atomics)
    uint32_t cap;             // items per list
    float *g_table;           // fp32 [entries][4]
    LevelMeta levels[16];
};

// index bits into / out of the lowest 3 mantissa bits of each of the four values (round to nearest on the remaining 20 bits; a non-finite
// value stays non-finite, so the NaN guard of the step still sees it)
__device__ __forceinline__ float4 pack_item(const float v[4], uint32_t idx) {
    float4 o;
    float *po = &o.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t b = __builtin_bit_cast(uint32_t, v[k]);
        if ((b & 0x7f800000u) != 0x7f800000u) b += 4u;          // (Inf / NaN: no carry out of the exponent)
        b = (b & ~7u) | ((idx >> (3 * k)) & 7u);
        po[k] = __builtin_bit_cast(float, b);
    }
    return o;
}

static_assert(kBinEntriesLog2 <= 12, "pack_item keeps 3 index bits per value");

#ifndef MNF_BIN_CHUNK
#define MNF_BIN_CHUNK 2
#endif
constexpr int kBinChunk = MNF_BIN_CHUNK;                     // sub-chunks of 256 samples per workgroup of pass A (experiment builds: -DMNF_BIN_CHUNK=1 / 4)
constexpr int kBinStage = kBinChunk * 256 * 8;               // items staged in LDS per workgroup (64 KB)

__global__ void __launch_bounds__(256) bin_items_kernel(const BinArgs args) {
    __shared__ uint32_t s_cnt[kMaxBins], s_off[kMaxBins];
    __shared__ float4 s_stage[kBinStage];
    __shared__ uint16_t s_binof[kBinStage];
    __shared__ uint32_t s_total;
    // level fastest: neighbouring workgroups reserve in different levels' cursors
    const int lk = (int)(blockIdx.x % (unsigned)args.n_levels), l = lk + args.level0;
    const int64_t chunk = blockIdx.x / (unsigned)args.n_levels;
    const LevelMeta m = args.levels[l];
    const uint32_t nb = m.size >> kBinEntriesLog2;
    const int64_t n_cnt = count_here(args.n, args.n_dev);
    int64_t rt0, rt1;
    chunk_tiles(n_cnt, args.chunk, args.n_chunks, rt0, rt1);
    const int64_t n_all = rt1 * kWaveSamples < n_cnt ? rt1 * kWaveSamples : n_cnt;
    const int64_t first = rt0 * kWaveSamples + chunk * (kBinChunk * 256);
    if (first >= n_all) return;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_cnt[b] = 0;
    __syncthreads();
    // 1. this thread's samples: entries, blend weights, gradient; rank of every item among the workgroup's items of its list
    uint32_t idx[kBinChunk][8], rank[kBinChunk][8];
    float w[kBinChunk][8];
    float4 g[kBinChunk];
#pragma unroll
    for (int sc = 0; sc < kBinChunk; ++sc) {
        const int64_t i = first + sc * 256 + threadIdx.x;
        if (i < n_all) {
            float xn[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) xn[d] = args.positions[3 * i + d];
            LevelPrep p;
            hash_prep(m, xn, p);
            g[sc] = reinterpret_cast<const float4 *>(args.dX)[(int64_t)l * args.Np + i];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                idx[sc][c] = p.off[c] >> 3;                          // entry inside the level
                w[sc][c] = ((c & 1) ? p.wxy[(c >> 1) & 1].y : p.wxy[(c >> 1) & 1].x) * p.wz[c >> 2];
                rank[sc][c] = atomicAdd(&s_cnt[idx[sc][c] >> kBinEntriesLog2], 1u);
            }
        }
    }
    __syncthreads();
    // 2. where each list's run starts in the staging buffer (exclusive scan, one wave) ...
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
            const uint32_t b = b0 + threadIdx.x;
            const uint32_t c = b < nb ? s_cnt[b] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if ((int)threadIdx.x >= d) incl += t;
            }
            if (b < nb) s_off[b] = run + incl - c;
            run += __shfl(incl, 63, 64);
        }
        if (threadIdx.x == 0) s_total = run;
    }
    __syncthreads();
    // ... and in the list itself: one returning add per list and workgroup; s_cnt becomes (slot in the list) - (position in the staging buffer)
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t c = s_cnt[b];
        const uint32_t base = c ? atomicAdd(&args.cursors[(size_t)lk * kMaxBins + b], c) : 0u;
        s_cnt[b] = base - s_off[b];
    }
    // 3. items into the staging buffer, list by list
#pragma unroll
    for (int sc = 0; sc < kBinChunk; ++sc) {
        const int64_t i = first + sc * 256 + threadIdx.x;
        if (i < n_all) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t b = idx[sc][c] >> kBinEntriesLog2, pos = s_off[b] + rank[sc][c];
                const float v[4] = {w[sc][c] * g[sc].x, w[sc][c] * g[sc].y, w[sc][c] * g[sc].z, w[sc][c] * g[sc].w};
                s_stage[pos] = pack_item(v, idx[sc][c] & (kBinEntries - 1u));
                s_binof[pos] = (uint16_t)b;
            }
        }
    }
    __syncthreads();
    // 4. out: consecutive lanes write consecutive slots of a list
    float4 *const lists = args.items + (size_t)lk * nb * args.cap;      // every binned level has the same size (2^k, checked by the host)
    const uint32_t total = s_total;
    for (uint32_t p = threadIdx.x; p < total; p += 256) {
        const uint32_t b = s_binof[p], slot = s_cnt[b] + p;
        const float4 it = s_stage[p];
        if (slot < args.cap) {
            lists[(size_t)b * args.cap + slot] = it;
        } else {                                                 // the list is full: straight to the gradient
            const uint32_t bits[4] = {__float_as_uint(it.x), __float_as_uint(it.y), __float_as_uint(it.z), __float_as_uint(it.w)};
            const uint32_t e = (bits[0] & 7u) | ((bits[1] & 7u) << 3) | ((bits[2] & 7u) << 6) | ((bits[3] & 7u) << 9);
            float *dst = args.g_table + ((size_t)(m.offset + b * kBinEntries + e) << 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = __uint_as_float(bits[k] & ~7u);
                if (v != 0.f) atomicAdd(dst + k, v);
            }
        }
    }
}

__global__ void __launch_bounds__(kBinThreadsB) bin_accumulate_kernel(const BinArgs args) {
    __shared__ double s_sum[kBinEntries * 4];
    const int lk = blockIdx.y, l = lk + args.level0;
    const LevelMeta m = args.levels[l];
    const uint32_t nb = m.size >> kBinEntriesLog2, b = blockIdx.x;
    if (b >= nb) return;
    uint32_t count = args.cursors[(size_t)lk * kMaxBins + b];
    if (count == 0) return;
    if (count > args.cap) count = args.cap;
    for (uint32_t e = threadIdx.x; e < kBinEntries * 4; e += kBinThreadsB) s_sum[e] = 0.0;
    __syncthreads();
    const float4 *list = args.items + ((size_t)lk * nb + b) * args.cap;
    constexpr int U = 4;                                         // items in flight per lane
    for (uint32_t i0 = threadIdx.x; i0 < count; i0 += kBinThreadsB * U) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f it[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * kBinThreadsB;
            it[u] = reinterpret_cast<const v4f *>(list)[i < count ? i : count - 1u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * kBinThreadsB >= count) break;
            const float f0 = it[u][0], f1 = it[u][1], f2 = it[u][2], f3 = it[u][3];     // (__builtin_bit_cast straight from a vector element reads element 0)
            const uint32_t b0 = __float_as_uint(f0), b1 = __float_as_uint(f1), b2 = __float_as_uint(f2), b3 = __float_as_uint(f3);
            const uint32_t e = (b0 & 7u) | ((b1 & 7u) << 3) | ((b2 & 7u) << 6) | ((b3 & 7u) << 9);
            double *dst = s_sum + e * 4;
            atomicAdd(dst, (double)__uint_as_float(b0 & ~7u));
            atomicAdd(dst + 1, (double)__uint_as_float(b1 & ~7u));
            atomicAdd(dst + 2, (double)__uint_as_float(b2 & ~7u));
            atomicAdd(dst + 3, (double)__uint_as_float(b3 & ~7u));
        }
    }
    __syncthreads();
    // this bin's sums into the gradient: plain read-modify-writes (nothing else touches these entries at this time)
    float *const out = args.g_table + (((size_t)m.offset + (size_t)b * kBinEntries) << 2);
    for (uint32_t i = threadIdx.x; i < kBinEntries * 4; i += kBinThreadsB) {
        const float a = (float)s_sum[i];
        if (a != 0.f) {
            out[i] += a;
        }
    }
}

// Reference form of the scatter (MNF_HASH_BWD_SIMPLE=1, debugging and A/B timing): one lane per sample, one atomic per
// corner and feature, no run merging.
__global__ void __launch_bounds__(256) hash_bwd_simple_kernel(const HashBwdArgs args) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count_here(args.n, args.n_dev)) return;
    float xn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) xn[d] = args.positions[3 * i + d];
    const int l = blockIdx.y + args.level0;
    const LevelMeta m = args.levels[l];
    float *const g_dst = l < args.repl_levels ? args.repl + (size_t)(blockIdx.x % kReplicas) * args.repl_floats : args.g_table;
    LevelPrep p;
    hash_prep(m, xn, p);
    const float4 g = reinterpret_cast<const float4 *>(args.dX)[(int64_t)l * args.Np + i];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float w = ((corner & 1) ? p.wxy[(corner >> 1) & 1].y : p.wxy[(corner >> 1) & 1].x) * p.wz[corner >> 2];
        float *dst = g_dst + (size_t)p.base * 4 + (p.off[corner] >> 1);   // byte offset of the fp16 entry / 2 == float index inside the level
        const float v[4] = {w * g.x, w * g.y, w * g.z, w * g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (v[k] != 0.f) atomicAdd(dst + k, v[k]);
    }
}

// ------------------------------------------------------------------ host side: transposed fragments and job table
enum KMapT { T_NATURAL, T_ACC };
enum ColMap { C_IDENT, C_GEO_RGB, C_GEO_SEM };

static int colmap(ColMap m, int i, int n_in_real) {
    if (m == C_IDENT) return i < n_in_real ? i : -1;
    if (i >= 16) return -1;
    if (m == C_GEO_RGB) return i == 0 ? 31 : 15 + i;
    return i == 0 ? 15 : i - 1;
}

// blocks of W^T for one matrix W[n_out][stride]: rows = input features (row_tiles x 32), K = output index n
static void append_matrix_T(std::vector<int32_t> &t, int buf, int64_t off, int n_out_real, int stride, int n_in_real,
                            int row_tiles, ColMap cm, KMapT km, int n_ksteps) {
    for (int rt = 0; rt < row_tiles; ++rt)
        for (int ks = 0; ks < n_ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r = lane & 31, h = lane >> 5;
                    const int col = colmap(cm, 32 * rt + r, n_in_real);
                    const int n = 16 * ks + (km == T_NATURAL ? 8 * h + j : 8 * (j >> 2) + 4 * h + (j & 3));
                    int32_t v = -1;
                    if (col >= 0 && n < n_out_real) v = (int32_t)((buf << 28) | (int32_t)(off + (int64_t)n * stride + col));
                    t.push_back(v);
                }
}

struct TrainTables {
    std::vector<int32_t> fragT;
    std::vector<WgradJob> jobs;
    std::vector<WgradGroup> groups;
    int rows, mask_blocks, mask_bytes;
};

static void add_jobs(std::vector<WgradJob> &jobs, int buf, int64_t off, int n_out_real, int stride, int n_in_real,
                     int dout_row0, int n_out_tiles, int in_row0, int n_in_tiles, ColMap cm) {
    for (int ot = 0; ot < n_out_tiles; ++ot)
        for (int it = 0; it < n_in_tiles; ++it) {
            WgradJob j;
            j.dout_row0 = dout_row0 + 32 * ot; j.in_row0 = in_row0 + 32 * it; j.buf = buf; j.param_off = (int32_t)off; j.stride = stride;
            j.n0 = 32 * ot;
            j.n_valid = n_out_real - 32 * ot < 32 ? n_out_real - 32 * ot : 32;
            if (j.n_valid < 0) j.n_valid = 0;
            for (int k = 0; k < 32; ++k) j.colmap[k] = colmap(cm, 32 * it + k, n_in_real);
            jobs.push_back(j);
        }
}

// 2 x 2 groups over the (n_out_tiles x n_in_tiles) jobs of one matrix, which add_jobs appended row-major at `first`
static void add_groups(std::vector<WgradGroup> &groups, int first, int n_out_tiles, int n_in_tiles) {
    for (int ot = 0; ot < n_out_tiles; ot += 2)
        for (int it = 0; it < n_in_tiles; it += 2) {
            WgradGroup g;
            g.no = n_out_tiles - ot < 2 ? 1 : 2; g.ni = n_in_tiles - it < 2 ? 1 : 2;
            for (int o = 0; o < 2; ++o)
                for (int i = 0; i < 2; ++i)
                    g.job[o][i] = first + (ot + (o < g.no ? o : 0)) * n_in_tiles + it + (i < g.ni ? i : 0);
            groups.push_back(g);
        }
}

template <int W, int NH>
static TrainTables build_tables(int C) {
    using T = TrainLayout<W, NH>;
    constexpr int Wh = W / 2;
    const int sem_pad = ((C + 15) / 16) * 16;
    TrainTables tt;
    tt.rows = T::rows; tt.mask_blocks = T::mask_blocks; tt.mask_bytes = T::mask_bytes;
    // parameter offsets (reference state_dict layout)
    int64_t b_in = 0, b_hid = (int64_t)W * 64, b_out = b_hid + (int64_t)(NH - 1) * W * W;
    int64_t h_in = 0, h_hid = (int64_t)Wh * 32, h_out = h_hid + (int64_t)Wh * Wh;
    int64_t s_in = 0, s_hid = (int64_t)Wh * 16, s_out = s_hid + (int64_t)Wh * Wh;
    auto &t = tt.fragT;
    append_matrix_T(t, 1, h_out, 16, Wh, Wh, Wh / 32, C_IDENT, T_NATURAL, 1);          // o_r3
    append_matrix_T(t, 1, h_hid, Wh, Wh, Wh, Wh / 32, C_IDENT, T_ACC, Wh / 16);         // o_r2
    append_matrix_T(t, 1, h_in, Wh, 32, 32, 1, C_GEO_RGB, T_ACC, Wh / 16);              // o_r1
    append_matrix_T(t, 2, s_out, sem_pad, Wh, Wh, Wh / 32, C_IDENT, T_NATURAL, 2);     // o_s3
    append_matrix_T(t, 2, s_hid, Wh, Wh, Wh, Wh / 32, C_IDENT, T_ACC, Wh / 16);         // o_s2
    append_matrix_T(t, 2, s_in, Wh, 16, 16, 1, C_GEO_SEM, T_ACC, Wh / 16);              // o_s1
    append_matrix_T(t, 0, b_out, 16, W, W, W / 32, C_IDENT, T_ACC, 1);                  // o_bo
    for (int l = 0; l < NH - 1; ++l) append_matrix_T(t, 0, b_hid + (int64_t)l * W * W, W, W, W, W / 32, C_IDENT, T_ACC, W / 16);   // o_bh
    append_matrix_T(t, 0, b_in, W, 64, 64, 2, C_IDENT, T_ACC, W / 16);                  // o_b1
    // weight-gradient jobs: (dOut rows, In rows), grouped 2 x 2 per matrix
    auto &j = tt.jobs;
    auto matrix = [&](int buf, int64_t off, int n_out_real, int stride, int n_in_real, int dout_row0, int n_out_tiles, int in_row0,
                      int n_in_tiles, ColMap cm) {
        const int first = (int)j.size();
        add_jobs(j, buf, off, n_out_real, stride, n_in_real, dout_row0, n_out_tiles, in_row0, n_in_tiles, cm);
        add_groups(tt.groups, first, n_out_tiles, n_in_tiles);
        return first;
    };
    matrix(0, b_in, W, 64, 64, T::rdZ0, W / 32, T::rX, 2, C_IDENT);
    const size_t hid_g0 = tt.groups.size();
    for (int l = 0; l < NH - 1; ++l)
        matrix(0, b_hid + (int64_t)l * W * W, W, W, W, T::rdZ0 + (l + 1) * W, W / 32, T::rH0 + l * W, W / 32, C_IDENT);
    const size_t hid_g1 = tt.groups.size();
    matrix(0, b_out, 16, W, W, T::rdBO, 1, T::rH0 + (NH - 1) * W, W / 32, C_IDENT);
    // rgb head: in = [SH(16) | geo fragment(16)] (contiguous rows rS..rS+31): SH rows map to columns 0..15
    {
        const int first = matrix(1, h_in, Wh, 32, 32, T::rdZr1, Wh / 32, T::rS, 1, C_IDENT);
        for (size_t q = first; q < j.size(); ++q)
            for (int k = 0; k < 16; ++k) { j[q].colmap[k] = k; j[q].colmap[16 + k] = colmap(C_GEO_RGB, k, 32); }
    }
    matrix(1, h_hid, Wh, Wh, Wh, T::rdZr2, Wh / 32, T::rHH1, Wh / 32, C_IDENT);
    matrix(1, h_out, 16, Wh, Wh, T::rdYr, 1, T::rHH2, Wh / 32, C_IDENT);
    matrix(2, s_in, Wh, 16, 16, T::rdZs1, Wh / 32, T::rG, 1, C_GEO_SEM);
    matrix(2, s_hid, Wh, Wh, Wh, T::rdZs2, Wh / 32, T::rHS1, Wh / 32, C_IDENT);
    matrix(2, s_out, sem_pad, Wh, Wh, T::rdYs, 1, T::rHS2, Wh / 32, C_IDENT);
    // wgrad_kernel runs four consecutive groups in one workgroup on the same tiles: the groups of a hidden matrix (W = 128: 2 x 2 groups that share every operand tile
    // pairwise) go first, so that a set of four is one hidden matrix; behind them base-in (two groups, same In tiles) + base-out (two groups, same dOut tile), then the heads
    std::rotate(tt.groups.begin(), tt.groups.begin() + hid_g0, tt.groups.begin() + hid_g1);
    return tt;
}

static bool tables_for(int W, int NH, int C, TrainTables &tt) {
#define MNF_CASE(w, nh) if (W == w && NH == nh) { tt = build_tables<w, nh>(C); return true; }
#ifdef MNF_DEV_ONLY_128x2
    MNF_CASE(128, 2)
#else
    MNF_CASE(128, 1) MNF_CASE(128, 2) MNF_CASE(128, 3) MNF_CASE(128, 4)
    MNF_CASE(64, 1) MNF_CASE(64, 2) MNF_CASE(64, 3) MNF_CASE(64, 4)
#endif
#undef MNF_CASE
    return false;
}

// Transposed fragments are a re-ordering of the forward fragments the handle already holds in fp16 (every weight sits in
// exactly one forward slot): `src_slot[i]` = forward slot of transposed element i, -1 = structural zero.  No pointer into
// the caller's fp32 vectors is kept between calls.
// Also clears the list cursors of the binned scatter: as a memset at the head of the bins' stream (round 3) the 2.5 KB fill was dispatched
// at the fork, beside the first workgroups of wgrad and the walk, and took 335 us to get through (profiles/r03_train_timeline.txt) — on the
// critical path of the step's tail.  Here it rides on a launch that runs before the fork.
__global__ void __launch_bounds__(256) gather_fragsT_kernel(const int32_t *__restrict__ src_slot, const half_t *__restrict__ frags,
                                                            half_t *__restrict__ dst, int64_t n, uint32_t *__restrict__ cursors, int n_cursors) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cursors) cursors[i] = 0u;
    if (i >= n) return;
    const int32_t s = src_slot[i];
    dst[i] = s >= 0 ? frags[s] : (half_t)0.0f;
}

struct TrainState {
    TrainTables tt;
    int32_t *d_fragT_src = nullptr;
    half_t *d_fragT = nullptr;
    WgradJob *d_jobs = nullptr;
    WgradGroup *d_groups = nullptr;
    // the hash-table scatter (float atomics, memory-side) runs beside the weight-gradient GEMMs (matrix cores + streaming loads)
    // on a second stream of the handle: both only depend on the backward-data kernel
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_entry = nullptr;
    hipEvent_t ev_chunk[4] = {nullptr, nullptr, nullptr, nullptr};      // behind each backward-data launch of the chunked pipeline
    // ... and the binned part of the scatter (streams through HBM and LDS) beside the walk (memory-side atomics) on a third
    hipStream_t side2 = nullptr;
    hipEvent_t ev_join2 = nullptr;
    // deterministic mode only (allocated on first use): fixed-point table gradient + replicas, weight-gradient partial slots
    unsigned long long *d_qtable = nullptr;
    float *d_partials = nullptr;
    size_t partials_floats = 0;
    // binned scatter of the large hashed levels (allocated on first use, grows with the sample bound): item lists + list cursors
    uint32_t *d_bin_cursors = nullptr;
};

static int ensure_train_state(mnf_field_t f) {
    if (f->train_state) return MNF_OK;
    TrainState *ts = new TrainState();
    if (!tables_for(f->cfg.neurons, f->cfg.layers, f->cfg.num_semantic_classes, ts->tt)) {
        delete ts;
        set_error("train: unsupported neurons=%d layers=%d", f->cfg.neurons, f->cfg.layers);
        return MNF_ERR_UNSUPPORTED;
    }
    {   // parameter id -> forward fragment slot
        std::vector<int32_t> slot_of[3];
        slot_of[0].assign((size_t)f->n_base_mlp, -1); slot_of[1].assign((size_t)f->n_head, -1); slot_of[2].assign((size_t)f->n_sem, -1);
        for (size_t i = 0; i < f->frag_src_host.size(); ++i) {
            const int32_t s = f->frag_src_host[i];
            if (s >= 0) slot_of[s >> 28][s & 0x0FFFFFFF] = (int32_t)i;
        }
        for (auto &s : ts->tt.fragT)
            if (s >= 0) s = slot_of[s >> 28][s & 0x0FFFFFFF];
    }
    hipError_t e = hipMalloc((void **)&ts->d_fragT_src, ts->tt.fragT.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ts->d_fragT, ts->tt.fragT.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ts->d_jobs, ts->tt.jobs.size() * sizeof(WgradJob));
    if (e == hipSuccess) e = hipMalloc((void **)&ts->d_groups, ts->tt.groups.size() * sizeof(WgradGroup));
    if (e == hipSuccess) e = hipMemcpy(ts->d_groups, ts->tt.groups.data(), ts->tt.groups.size() * sizeof(WgradGroup), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ts->d_fragT_src, ts->tt.fragT.data(), ts->tt.fragT.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ts->d_jobs, ts->tt.jobs.data(), ts->tt.jobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&ts->d_bin_cursors, (size_t)16 * kMaxBins * sizeof(uint32_t));
    // side streams: the process-wide ones (common.h shared_side_stream), not a pair per train state
    ts->side = shared_side_stream(0);
    ts->side2 = shared_side_stream(1);
    if (!ts->side || !ts->side2) { delete ts; return MNF_ERR_HIP; }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ts->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ts->ev_join, hipEventDisableTiming);
    for (int c = 0; c < 4 && e == hipSuccess; ++c) e = hipEventCreateWithFlags(&ts->ev_chunk[c], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ts->ev_entry, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ts->ev_join2, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("train: %s", hipGetErrorString(e));
        delete ts;
        return MNF_ERR_HIP;
    }
    f->train_state = ts;
    return MNF_OK;
}

void free_train_state_impl(mnf_field_t f) {
    TrainState *ts = reinterpret_cast<TrainState *>(f->train_state);
    if (!ts) return;
    if (ts->d_fragT_src) (void)hipFree(ts->d_fragT_src);
    if (ts->d_fragT) (void)hipFree(ts->d_fragT);
    if (ts->d_jobs) (void)hipFree(ts->d_jobs);
    if (ts->d_groups) (void)hipFree(ts->d_groups);
    if (ts->d_qtable) (void)hipFree(ts->d_qtable);
    if (ts->d_partials) (void)hipFree(ts->d_partials);
    if (ts->d_bin_cursors) (void)hipFree(ts->d_bin_cursors);
    if (ts->ev_fork) (void)hipEventDestroy(ts->ev_fork);
    if (ts->ev_join) (void)hipEventDestroy(ts->ev_join);
    for (int c = 0; c < 4; ++c) if (ts->ev_chunk[c]) (void)hipEventDestroy(ts->ev_chunk[c]);
    if (ts->ev_entry) (void)hipEventDestroy(ts->ev_entry);
    if (ts->ev_join2) (void)hipEventDestroy(ts->ev_join2);
    delete ts;
    f->train_state = nullptr;
}

struct WsView {
    half_t *act; uint8_t *masks; float *dX, *xn, *repl;
    float4 *bin_items;     // item lists of the binned scatter (levels kBinLevel0Default .. 15): part of the caller's workspace since round 4 (ADVICE r03:
    size_t bin_items_n;    // they were ~720 B per sample of raw hipMalloc per train state, invisible to the caller's allocator and grown mid-step)
    int64_t Np, bytes;
};

// the binned scatter's plan for an upper bound of n samples: levels [first, 16) (all hashed, one size, whole bins), lists per level, items per list
constexpr int kBinLevel0Default = 12;      // measured: round 3 (tools/r03_bins_step.sh, profiles/r03_bins_step_*.txt) the step time was flat from 10 to 12, worse below and above (11 then);
                                           // round 4, with wgrad's lighter traffic, 12 is 60 us ahead of 11 at 1.0 M samples (3.393 -> 3.327 ms, four alternations on one box) and equal at 0.25 M
static void bin_plan(const mnf_field_s *f, int64_t n, int want, int &first_binned, uint32_t &nb, uint32_t &cap) {
    first_binned = 16; nb = 0; cap = 0;
    if (n < 8192) return;                    // small batches: the walk alone (pass B costs ~20 us whatever it is given)
    for (int l = 15; l >= want && l >= 0; --l) {
        const LevelMeta &lm = f->levels[l];
        if (!lm.hashed || lm.size < kBinEntries || (lm.size & (lm.size - 1u)) || lm.size != f->levels[15].size || (lm.size >> kBinEntriesLog2) > (uint32_t)kMaxBins) break;
        first_binned = l;
    }
    if (first_binned == 16) return;
    nb = f->levels[15].size >> kBinEntriesLog2;
    const uint64_t per_list = (uint64_t)n * 8 / nb;
    cap = (uint32_t)(per_list + per_list / 8 + 2048);      // a uniform hash fills the lists evenly; the excess of a full one goes to the atomics
}

static WsView carve_train(const TrainTables &tt, const mnf_field_s *f, void *base, int64_t n) {
    WsView v;
    v.Np = ceil_div(n, 64) * 64;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? (char *)base + off : nullptr; off += (b + 255) & ~(size_t)255; return p; };
    v.act = (half_t *)take((size_t)tt.rows * v.Np * 2);
    v.masks = (uint8_t *)take((size_t)v.Np * tt.mask_bytes);      // [tile][lane] records
    v.dX = (float *)take((size_t)v.Np * 64 * 4);
    v.xn = (float *)take((size_t)v.Np * 3 * 4);
    v.repl = (float *)take((size_t)kReplicas * kReplMaxEntries * 4 * sizeof(float));   // private copies of the coarsest levels' gradient (scatter)
    {
        int first; uint32_t nb, cap;
        bin_plan(f, n, kBinLevel0Default, first, nb, cap);
        v.bin_items_n = (size_t)(16 - first) * nb * cap;
        v.bin_items = (float4 *)take(v.bin_items_n * sizeof(float4));
    }
    v.bytes = (int64_t)off;
    return v;
}

template <int W, int NH>
static void launch_dgrad(const BwdArgs &a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((dgrad_kernel<W, NH>), dim3(grid), dim3(kThreads), 0, s, a);
}

int64_t train_workspace_bytes_impl(mnf_field_t f, int64_t n) {
    if (!f || n < 0) return -1;
    TrainTables tt;
    if (!tables_for(f->cfg.neurons, f->cfg.layers, f->cfg.num_semantic_classes, tt)) return -1;
    return carve_train(tt, f, nullptr, n).bytes;
}

int forward_train_impl(mnf_field_t f, const FieldIO &io, void *workspace, int64_t workspace_bytes, hipStream_t stream, bool deterministic) {
    MNF_REQUIRE(f && f->params_loaded, "field_forward_train: parameters not loaded");
    MNF_REQUIRE(io.n >= 0, "field_forward_train: negative n");
    if (io.n == 0) return MNF_OK;
    int rc = ensure_train_state(f);
    if (rc) return rc;
    TrainState *ts = reinterpret_cast<TrainState *>(f->train_state);
    WsView v = carve_train(ts->tt, f, workspace, io.n);
    if (!workspace || workspace_bytes < v.bytes) {
        set_error("field_forward_train: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)v.bytes);
        return MNF_ERR_WORKSPACE;
    }
    TrainBuf tb = {v.act, v.masks, v.Np, ts->tt.rows};
    return launch_field_impl(f, io, false, stream, &tb);
}

int backward_impl(mnf_field_t f, const float *positions, int64_t n, const int64_t *n_dev, const float *d_rgb, const float *d_density,
                  const float *d_sem, const float *rgb, const float *density, void *workspace, int64_t workspace_bytes, float loss_scale,
                  float *g_base, float *g_head, float *g_sem, bool zero_grads, bool positions_normalized, bool deterministic, hipStream_t s) {
    FactoredGrad fg = {nullptr, nullptr, nullptr, nullptr};
    const bool factored = take_factored_output_gradient(fg);      // first of all: taken (and cleared) on every path out of this call
    MNF_REQUIRE(f && f->params_loaded, "field_backward: parameters not loaded");
    MNF_REQUIRE(n >= 0 && loss_scale > 0.f, "field_backward: bad arguments");
    MNF_REQUIRE(g_base && g_head && g_sem, "field_backward: null gradient buffer");
    if (zero_grads) {
        MNF_HIP(hipMemsetAsync(g_base, 0, (size_t)f->n_base * 4, s));
        MNF_HIP(hipMemsetAsync(g_head, 0, (size_t)f->n_head * 4, s));
        MNF_HIP(hipMemsetAsync(g_sem, 0, (size_t)f->n_sem * 4, s));
    }
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(positions && d_density && rgb && density && (factored || (d_rgb && d_sem)), "field_backward: null pointer");
    int rc = ensure_train_state(f);
    if (rc) return rc;
    TrainState *ts = reinterpret_cast<TrainState *>(f->train_state);
    WsView v = carve_train(ts->tt, f, workspace, n);
    if (!workspace || workspace_bytes < v.bytes) {
        set_error("field_backward: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)v.bytes);
        return MNF_ERR_WORKSPACE;
    }
    // The scatter's replica buffers are zeroed on the side stream while the backward-data kernel runs (behind the caller's earlier work:
    // the workspace may have served another call on this stream a moment ago), not between that kernel and the scatter.
    uint32_t repl_entries = 0;
    int repl_levels = 0;
    for (int l = 0; l < 16; ++l) {   // the leading dense levels while they stay small
        if (f->levels[l].hashed || f->levels[l].offset != repl_entries || repl_entries + f->levels[l].size > kReplMaxEntries) break;
        repl_entries += f->levels[l].size;
        repl_levels = l + 1;
    }
    MNF_HIP(hipEventRecord(ts->ev_entry, s));
    MNF_HIP(hipStreamWaitEvent(ts->side, ts->ev_entry, 0));
    if (repl_levels && !deterministic) MNF_HIP(hipMemsetAsync(v.repl, 0, (size_t)kReplicas * repl_entries * 4 * sizeof(float), ts->side));
    // transposed fp16 weight fragments from the handle's forward fragments (the parameters of the last set_params)
    const int64_t n_frag = (int64_t)ts->tt.fragT.size();
    hipLaunchKernelGGL(gather_fragsT_kernel, dim3((unsigned)ceil_div(n_frag > 16 * kMaxBins ? n_frag : (int64_t)16 * kMaxBins, 256)), dim3(256), 0, s, ts->d_fragT_src,
                       reinterpret_cast<const half_t *>(f->d_frags), ts->d_fragT, n_frag, ts->d_bin_cursors, 16 * kMaxBins);
    BwdArgs a;
    a.fg = fg;
    a.fragsT = reinterpret_cast<const half8 *>(ts->d_fragT);
    a.d_rgb = d_rgb; a.d_sigma = d_density; a.d_sem = d_sem; a.rgb = rgb; a.sigma = density;
    a.dX = v.dX; a.n = n; a.n_dev = n_dev; a.C = f->cfg.num_semantic_classes; a.loss_scale = loss_scale;
    a.train = {v.act, v.masks, v.Np, ts->tt.rows};
    // Chunked pipeline (float-atomic mode, large batches): the backward-data kernel is launched for four contiguous ranges of tiles and the scatter of a
    // range (walk on the side stream, pass A of the bins on the third) starts behind ITS launch, beside the next range's dgrad: dgrad writes activations
    // through the matrix cores and HBM stores, the scatter is bound by the memory-side atomic unit — the two overlap, and only the last range's scatter
    // (and the weight gradients, which read every range) is left behind the last dgrad.  One range = the old schedule (deterministic mode: its grouping
    // of partial sums must not change).
    static const int chunks_env = diag_env("MNF_BWD_CHUNKS") ? atoi(diag_env("MNF_BWD_CHUNKS")) : 0;
    // measured (profiles/r04_bwd_chunks.txt): 5.77 / 5.52 / 5.60 ms per step with 1 / 2 / 4 ranges.
    int n_chunks = (!deterministic && n >= ((int64_t)1 << 18)) ? 2 : 1;
    if (chunks_env >= 1 && chunks_env <= 4 && !deterministic) n_chunks = chunks_env;
    if (!positions_normalized) {   // (the train step's forward hands over normalised positions already: FieldIO::xn_out)
        const float *ab = f->cfg.aabb;
        const int64_t blocks = ceil_div(3 * n, 256);
        hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, positions, n, n_dev, ab[0], ab[1], ab[2],
                           ab[3], ab[4], ab[5], v.xn);
    }
    int grid = 256;
    const int64_t chunk_tiles_max = ceil_div(ceil_div(n, kWaveSamples), n_chunks) + 1;      // tiles of one range, at most
    const int64_t wgs = ceil_div(chunk_tiles_max, kWavesPerBlock);
    if (wgs < grid) grid = (int)wgs;
    const int W = f->cfg.neurons, NH = f->cfg.layers;
    bool ok = false;
    const int prof_dgrad = prof_start("dgrad", s);
    for (int c = 0; c < n_chunks; ++c) {
        a.chunk = c; a.n_chunks = n_chunks;
#define MNF_CASE(w, nh) if (W == w && NH == nh) { launch_dgrad<w, nh>(a, grid, s); ok = true; }
#ifdef MNF_DEV_ONLY_128x2
        MNF_CASE(128, 2)
#else
        MNF_CASE(128, 1) MNF_CASE(128, 2) MNF_CASE(128, 3) MNF_CASE(128, 4)
        MNF_CASE(64, 1) MNF_CASE(64, 2) MNF_CASE(64, 3) MNF_CASE(64, 4)
#endif
#undef MNF_CASE
        MNF_HIP(hipEventRecord(ts->ev_chunk[c], s));      // ---- fork point of range c: its scatter may start (the weight gradients stay on the caller's stream)
    }
    prof_stop(prof_dgrad, s);
    MNF_REQUIRE(ok, "field_backward: unsupported shape");
    rc = launch_status("dgrad_kernel");
    if (rc) return rc;
    hipStream_t ss = ts->side;
    // weight gradients
    const int n_groups = (int)ts->tt.groups.size();
    // enough sample chunks to fill the chip with waves (the loop is load-latency bound; 2 x 2 blocking leaves room for
    // ~3 waves per SIMD), but at least 16 tiles per wave so the final atomics stay negligible (sized on the upper bound `n`)
    const int64_t n_tiles = v.Np / 64;
    // (12 waves per CU and 14 groups: 219 tile ranges; the deterministic mode's grouping of partial sums depends on this value and keeps it.  With float atomics
    //  18 per CU — 329 ranges — is 20-40 us ahead at 1.0 M samples: 3.211 / 3.184 / 3.187 / 3.104 ms per step with 219 / 330 / 440 / 660, MNF_WGRAD_SPLIT)
    int split = (int)(256 * (deterministic ? 12 : 18) / n_groups);
    if (!deterministic) {
        static const int split_env = diag_env("MNF_WGRAD_SPLIT") ? atoi(diag_env("MNF_WGRAD_SPLIT")) : 0;
        if (split_env > 0) split = split_env;
    }
    if (split > n_tiles / 16) split = (int)(n_tiles / 16);
    if (split < 1) split = 1;
    float *partials = nullptr;
    if (deterministic) {
        const size_t need = (size_t)n_groups * split * 4 * 1024;
        if (ts->partials_floats < need) {
            if (ts->d_partials) (void)hipFree(ts->d_partials);
            ts->d_partials = nullptr; ts->partials_floats = 0;
            MNF_HIP(hipMalloc((void **)&ts->d_partials, need * sizeof(float)));
            ts->partials_floats = need;
        }
        partials = ts->d_partials;
    }
    static const bool no_wgrad = diag_env("MNF_NO_WGRAD") != nullptr;    // timing experiments: the scatter alone on the chip
    if (!no_wgrad) {
        ProfScope ps("wgrad", s);
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(ceil_div(n_groups, 4) * split)), dim3(256), 0, s, ts->d_jobs, ts->d_groups, n_groups,
                           split, v.act, n, n_dev, ts->tt.rows, 1.0f / loss_scale, g_base, g_head, g_sem, partials);
        if (partials)
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div((int64_t)n_groups * 4 * 1024, 256)), dim3(256), 0, s, ts->d_jobs, ts->d_groups,
                               n_groups, split, (const float *)partials, 1.0f / loss_scale, g_base, g_head, g_sem);
    }
    rc = launch_status("wgrad_kernel");
    if (rc) return rc;
    // hash-table gradient (side stream)
    HashBwdArgs hb;
    hb.positions = positions_normalized ? positions : v.xn; hb.dX = v.dX; hb.Np = v.Np; hb.g_table = g_base + f->n_base_mlp; hb.n = n; hb.n_dev = n_dev;
    std::memcpy(hb.aabb, f->cfg.aabb, sizeof(hb.aabb));
    std::memcpy(hb.levels, f->levels, sizeof(hb.levels));
    int n_levels = 16;
    hb.level0 = 0; hb.chunk = 0; hb.n_chunks = 1;
    if (const char *e = diag_env("MNF_HASH_BWD_LEVELS")) { int lo = 0, hi = 16; if (sscanf(e, "%d,%d", &lo, &hi) == 2) { hb.level0 = lo; n_levels = hi - lo; } }
    // replicated coarse levels (their own region of the workspace, zeroed above)
    hb.repl_levels = repl_levels;
    hb.repl_floats = repl_entries * 4;
    hb.repl = v.repl;
    const size_t repl_bytes = (size_t)kReplicas * hb.repl_floats * sizeof(float);
    hb.flush_count = nullptr;
    hb.q_table = nullptr; hb.q_repl = nullptr; hb.q_bad = nullptr;
    const size_t q_table_words = (size_t)f->table_entries * 4, q_repl_words = (size_t)kReplicas * kReplMaxEntries * 4;
    if (deterministic) {
        if (!ts->d_qtable) MNF_HIP(hipMalloc((void **)&ts->d_qtable, (q_table_words + q_repl_words + 8) * sizeof(unsigned long long)));
        hb.q_table = ts->d_qtable; hb.q_repl = ts->d_qtable + q_table_words; hb.q_bad = hb.q_repl + q_repl_words;
    }
#ifdef MNF_DIAG
    static unsigned long long *d_flush = nullptr;
    if (diag_env("MNF_SCATTER_COUNT")) {
        if (!d_flush) MNF_HIP(hipMalloc((void **)&d_flush, 16 * sizeof(unsigned long long)));
        MNF_HIP(hipMemsetAsync(d_flush, 0, 16 * sizeof(unsigned long long), ss));
        hb.flush_count = d_flush;
    }
#endif
    static const bool simple = diag_env("MNF_HASH_BWD_SIMPLE") != nullptr;   // debugging aid: one atomic per lane, corner and feature
    // levels [first_binned, 16): through the bins (all of them hashed tables of one size that whole bins tile); the rest: the walk
    int first_binned = 16;
    uint32_t nb_plan = 0, cap = 0;
    if (!deterministic && !simple && n_levels == 16) {
        int want = kBinLevel0Default;
        if (const char *e = diag_env("MNF_BIN_LEVEL0")) want = atoi(e);
        bin_plan(f, n, want, first_binned, nb_plan, cap);
        if (const char *e = diag_env("MNF_BIN_CAP")) cap = (uint32_t)atoi(e);     // tests: force the full-list path
        while (first_binned < 16 && (size_t)(16 - first_binned) * nb_plan * cap > v.bin_items_n) ++first_binned;      // (a diagnostic plan larger than the workspace's)
    }
    BinArgs ba;
    const int n_binned = 16 - first_binned;
    if (n_binned > 0) {
        ba.chunk = 0; ba.n_chunks = 1;
        ba.positions = hb.positions; ba.dX = hb.dX; ba.Np = hb.Np; ba.n = n; ba.n_dev = n_dev; ba.level0 = first_binned; ba.n_levels = n_binned;
        ba.items = v.bin_items; ba.cursors = ts->d_bin_cursors; ba.cap = cap; ba.g_table = hb.g_table;
        std::memcpy(ba.levels, f->levels, sizeof(ba.levels));
        n_levels = first_binned;
    }
    MNF_HIP(hipStreamWaitEvent(ss, ts->ev_chunk[0], 0));
    const int prof_scatter = prof_start("hash_scatter", ss);
    hb.n_walk_levels = n_levels;
    int walk_wgs = 8;                                                                   // workgroups per CU of the 256 (gfx950 only: as the field kernel)
    if (const char *e = diag_env("MNF_WALK_WGS")) walk_wgs = atoi(e);
    const int64_t range_samples = chunk_tiles_max * kWaveSamples;                        // samples of one range, at most
    const int64_t walk_full = ceil_div(ceil_div(range_samples, kWalkChunk) * 32, 256) * (n_levels > 0 ? n_levels : 1);
    const dim3 walk_grid((unsigned)(walk_wgs > 0 && walk_full > (int64_t)256 * walk_wgs ? (int64_t)256 * walk_wgs : (walk_full > 0 ? walk_full : 1)));
    if (deterministic) {
        MNF_HIP(hipStreamWaitEvent(ss, ts->ev_chunk[0], 0));
        MNF_HIP(hipMemsetAsync(hb.q_table, 0, q_table_words * sizeof(unsigned long long), ss));
        if (hb.repl_levels) MNF_HIP(hipMemsetAsync(hb.q_repl, 0, (size_t)kReplicas * hb.repl_floats * sizeof(unsigned long long), ss));
        MNF_HIP(hipMemsetAsync(hb.q_bad, 0, sizeof(unsigned long long), ss));
        hipLaunchKernelGGL(hash_bwd_walk_kernel<true>, walk_grid, dim3(256), 0, ss, hb);
        if (hb.repl_levels)
            hipLaunchKernelGGL(fold_replicas_fixed_kernel, dim3((hb.repl_floats + 255) / 256), dim3(256), 0, ss, (const unsigned long long *)hb.q_repl,
                               hb.repl_floats, hb.q_table);
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(4096), dim3(256), 0, ss, reinterpret_cast<const long long *>(hb.q_table), hb.g_table,
                           (int64_t)q_table_words, (const unsigned long long *)hb.q_bad);
    } else {
        if (n_binned > 0) {
            const uint32_t nb = f->levels[15].size >> kBinEntriesLog2;
            static const bool same_stream = diag_env("MNF_BIN_SAME_STREAM") != nullptr;     // timing experiments: bins in front of the walk on one stream
            hipStream_t s2 = same_stream ? ss : ts->side2;
            MNF_HIP(hipStreamWaitEvent(s2, ts->ev_chunk[0], 0));
            const int prof_bins = prof_start("hash_scatter_bins", s2);
            // (list cursors: cleared by gather_fragsT_kernel, in front of the fork)
            // (Measured and dropped, tools/r03_walk_wgs.sh: pass B level by level or half by half on a fourth stream under the next pass A — 5.8 and 5.6 ms
            // per step against 5.55-5.6 for one launch each: walk, bins and wgrad together are bound by HBM, the order inside does not matter.)
            for (int c = 0; c < n_chunks; ++c) {          // pass A range by range, each behind its dgrad; pass B once, behind the last
                if (c) MNF_HIP(hipStreamWaitEvent(s2, ts->ev_chunk[c], 0));
                ba.chunk = c; ba.n_chunks = n_chunks;
                hipLaunchKernelGGL(bin_items_kernel, dim3((unsigned)(ceil_div(range_samples, kBinChunk * 256) * n_binned)), dim3(256), 0, s2, ba);
            }
            hipLaunchKernelGGL(bin_accumulate_kernel, dim3(nb, n_binned), dim3(kBinThreadsB), 0, s2, ba);
            prof_stop(prof_bins, s2);
            MNF_HIP(hipEventRecord(ts->ev_join2, s2));
            MNF_HIP(hipStreamWaitEvent(s, ts->ev_join2, 0));
        }
        if (simple) {
            MNF_HIP(hipStreamWaitEvent(ss, ts->ev_chunk[n_chunks - 1], 0));
            hipLaunchKernelGGL(hash_bwd_simple_kernel, dim3((unsigned)ceil_div(n, 256), n_levels), dim3(256), 0, ss, hb);
        } else {
            for (int c = 0; c < n_chunks; ++c) {
                MNF_HIP(hipStreamWaitEvent(ss, ts->ev_chunk[c], 0));
                hb.chunk = c; hb.n_chunks = n_chunks;
                if (n_levels > 0) hipLaunchKernelGGL(hash_bwd_walk_kernel<false>, walk_grid, dim3(256), 0, ss, hb);
            }
        }
        if (hb.repl_levels)
            hipLaunchKernelGGL(fold_replicas_kernel, dim3((hb.repl_floats + 255) / 256), dim3(256), 0, ss, hb.repl, hb.repl_floats, hb.g_table);
    }
    rc = launch_status("hash_bwd_walk_kernel");
    if (rc) return rc;
    prof_stop(prof_scatter, ss);
    rc = launch_status("fold_replicas_kernel");
#ifdef MNF_DIAG
    if (hb.flush_count) {      // quad atomics per level of this backward (synchronises: diagnostic only)
        unsigned long long h[16];
        MNF_HIP(hipMemcpyAsync(h, hb.flush_count, sizeof(h), hipMemcpyDeviceToHost, ss));
        MNF_HIP(hipStreamSynchronize(ss));
        unsigned long long tot = 0;
        fprintf(stderr, "[mnf scatter] quad atomics per level:");
        for (int l = 0; l < 16; ++l) { fprintf(stderr, " %llu", h[l]); tot += h[l]; }
        fprintf(stderr, "  total %llu (upper bound n = %lld samples)\n", tot, (long long)n);
    }
#endif
    // ---- join
    MNF_HIP(hipEventRecord(ts->ev_join, ss));
    MNF_HIP(hipStreamWaitEvent(s, ts->ev_join, 0));
    return rc;
}

MNF_DT_END

#ifndef MNF_BF16   // ---- operand-type independent: compiled once
namespace mnf {
static thread_local FactoredGrad t_factored = {nullptr, nullptr, nullptr, nullptr};
void set_factored_output_gradient(const FactoredGrad &fg) { t_factored = fg; }
bool take_factored_output_gradient(FactoredGrad &fg) {
    fg = t_factored;
    t_factored = FactoredGrad{nullptr, nullptr, nullptr, nullptr};
    return fg.w != nullptr;
}
// ------------------------------------------------------------------ optimizer step (pipeline.py:173-178, :531)
// torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) on one flat parameter vector in a single pass
// (torch's foreach implementation is six passes over the 25 M table entries): lerp of the first moment, addcmul of the
// second, bias-corrected step.  `any_nan` counts NaN gradients (the reference skips the whole iteration then).
__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float beta1, float beta2, float eps,
                                                   float step_size, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

// The same update for a training loop that never synchronises with the host: the step count lives on the device and the whole update is
// skipped when `skip` is non-zero (non-finite gradients, pipeline.py:520-529; a step that overflowed its sample bounds or rendered
// no sample, csrc/trainstep.hip).  adam_prepare_kernel advances the count and derives the bias corrections (in double, as the host
// version does); adam_guarded_kernel optionally mirrors the new parameters from `half_from` on into the field handle's fp16 hash
// table, so the per-step fp32 -> fp16 conversion pass over the 25 M entries disappears.
__global__ void adam_prepare_kernel(float *__restrict__ step, const int32_t *__restrict__ skip, float lr, float beta1, float beta2, float *__restrict__ hyper) {
    const bool sk = skip && *skip != 0;
    float st = *step;
    if (!sk) { st += 1.0f; *step = st; }
    const double bc1 = 1.0 - pow((double)beta1, (double)st), bc2 = 1.0 - pow((double)beta2, (double)st);
    hyper[0] = (float)((double)lr / bc1); hyper[1] = (float)sqrt(bc2); hyper[2] = sk ? 1.0f : 0.0f;
}

__global__ void __launch_bounds__(256) adam_guarded_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                           float *__restrict__ v, int64_t n, float beta1, float beta2, float eps,
                                                           const float *__restrict__ hyper, _Float16 *__restrict__ half_out, int64_t half_from) {
    const float step_size = hyper[0], bc2_sqrt = hyper[1];
    if (hyper[2] != 0.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pn = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        p[i] = pn;
        if (half_out && i >= half_from) half_out[i - half_from] = (_Float16)pn;
    }
}

// The three parameter vectors of a field in ONE launch each (guard, step counts, update): mnf_field_optimizer_step issued 3 + 3 + 3 launches, most of them ~5 us of nothing.
struct Adam3 {
    float *p[3]; const float *g[3]; float *m[3], *v[3]; float *step[3];
    int64_t n[3];
};
// Both passes over the parameter vectors move 16 bytes per lane and access (one thread = one aligned quad of one vector; the up-to-three elements behind the last
// quad of a vector one by one): with 4-byte accesses the update of the 12.6 M table parameters ran at 2.3 TB/s (167 us of the reference-yaml step's 1.29 ms).
__device__ __forceinline__ bool quad_of(const Adam3 &a, int64_t q, int &k, int64_t &i0) {
    const int64_t q0 = (a.n[0] + 3) >> 2, q1 = (a.n[1] + 3) >> 2, q2 = (a.n[2] + 3) >> 2;
    if (q < q0) { k = 0; i0 = q << 2; return true; }
    if (q < q0 + q1) { k = 1; i0 = (q - q0) << 2; return true; }
    if (q < q0 + q1 + q2) { k = 2; i0 = (q - q0 - q1) << 2; return true; }
    return false;
}
__device__ __forceinline__ bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__global__ void __launch_bounds__(256) count_nan3_kernel(const Adam3 a, int32_t *__restrict__ count) {
    int local = 0;
    const int64_t quads = ((a.n[0] + 3) >> 2) + ((a.n[1] + 3) >> 2) + ((a.n[2] + 3) >> 2);
    const int64_t stride = (int64_t)blockDim.x * gridDim.x;
    // four quads per trip, their loads issued together (one 16-byte load in flight per thread read the 50 MB of gradients at 2.3 TB/s)
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < quads; q0 += 4 * stride) {
        float4 x[4];
        bool fast[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = q0 + u * stride;
            int k = 0; int64_t i0 = 0;
            const bool in = q < quads && quad_of(a, q, k, i0);
            const float *g = a.g[k] + i0;
            fast[u] = in && i0 + 4 <= a.n[k] && aligned16(g);
            x[u] = fast[u] ? *reinterpret_cast<const float4 *>(g) : float4{0.f, 0.f, 0.f, 0.f};
            if (in && !fast[u])
                for (int64_t i = i0; i < a.n[k] && i < i0 + 4; ++i) local += !(fabsf(a.g[k][i]) <= 3.4028234664e38f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            local += !(fabsf(x[u].x) <= 3.4028234664e38f) + !(fabsf(x[u].y) <= 3.4028234664e38f) + !(fabsf(x[u].z) <= 3.4028234664e38f) + !(fabsf(x[u].w) <= 3.4028234664e38f);
    }
    if (__ballot(local != 0) != 0ull && local) atomicAdd(count, local);
}
__global__ void adam_prepare3_kernel(const Adam3 a, const int32_t *__restrict__ skip, float lr, float beta1, float beta2, float *__restrict__ hyper,
                                     const int64_t *__restrict__ counts, long long *__restrict__ report) {
    const int k = threadIdx.x;
    // the step's report to the host, written straight into pinned host memory: the train step's four counters and its FINAL skip flag (this launch runs behind
    // the non-finite count).  Two device-to-host copies did this before; each cost the stream ~5 us of copy plus ~18 us of bubble around it.
    if (report && k >= 3 && k < 8) {
        report[k - 3] = k < 7 ? (counts ? (long long)counts[k - 3] : 0ll) : (long long)(skip ? *skip : 0);
        __threadfence_system();
    }
    if (k >= 3) return;
    const bool sk = skip && *skip != 0;
    float st = *a.step[k];
    if (!sk) { st += 1.0f; *a.step[k] = st; }
    const double bc1 = 1.0 - pow((double)beta1, (double)st), bc2 = 1.0 - pow((double)beta2, (double)st);
    hyper[4 * k] = (float)((double)lr / bc1); hyper[4 * k + 1] = (float)sqrt(bc2); hyper[4 * k + 2] = sk ? 1.0f : 0.0f;
}
__global__ void __launch_bounds__(256) adam_guarded3_kernel(const Adam3 a, float beta1, float beta2, float eps, const float *__restrict__ hyper,
                                                            _Float16 *__restrict__ half_out, int64_t half_from) {
    if (hyper[2] != 0.0f) return;
    const int64_t quads = ((a.n[0] + 3) >> 2) + ((a.n[1] + 3) >> 2) + ((a.n[2] + 3) >> 2);
    auto one = [&](float gi, float &mi, float &vi, float &pi, float step_size, float bc2_sqrt) {      // (the scalar kernel's arithmetic, operation for operation)
        mi = mi + (gi - mi) * (1.0f - beta1);
        vi = vi * beta2 + (1.0f - beta2) * gi * gi;
        pi = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    };
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)blockDim.x * gridDim.x) {
        int k; int64_t i0;
        quad_of(a, q, k, i0);
        const float step_size = hyper[4 * k], bc2_sqrt = hyper[4 * k + 1];
        const bool mirror = k == 0 && half_out && i0 >= half_from;
        if (i0 + 4 <= a.n[k] && aligned16(a.g[k] + i0) && aligned16(a.m[k] + i0) && aligned16(a.v[k] + i0) && aligned16(a.p[k] + i0)) {
            const float4 g = *reinterpret_cast<const float4 *>(a.g[k] + i0);
            float4 m = *reinterpret_cast<const float4 *>(a.m[k] + i0), v = *reinterpret_cast<const float4 *>(a.v[k] + i0), p = *reinterpret_cast<const float4 *>(a.p[k] + i0);
            one(g.x, m.x, v.x, p.x, step_size, bc2_sqrt); one(g.y, m.y, v.y, p.y, step_size, bc2_sqrt);
            one(g.z, m.z, v.z, p.z, step_size, bc2_sqrt); one(g.w, m.w, v.w, p.w, step_size, bc2_sqrt);
            *reinterpret_cast<float4 *>(a.m[k] + i0) = m; *reinterpret_cast<float4 *>(a.v[k] + i0) = v; *reinterpret_cast<float4 *>(a.p[k] + i0) = p;
            if (k == 0 && half_out && i0 < half_from && i0 + 4 > half_from) {      // the quad the mirror starts inside
                const float pe[4] = {p.x, p.y, p.z, p.w};
                for (int e = 0; e < 4; ++e) if (i0 + e >= half_from) half_out[i0 + e - half_from] = (_Float16)pe[e];
            }
            if (mirror) {
                _Float16 *h = half_out + (i0 - half_from);
                if ((reinterpret_cast<uintptr_t>(h) & 7) == 0) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 o; o[0] = (_Float16)p.x; o[1] = (_Float16)p.y; o[2] = (_Float16)p.z; o[3] = (_Float16)p.w;
                    *reinterpret_cast<h4 *>(h) = o;
                } else { h[0] = (_Float16)p.x; h[1] = (_Float16)p.y; h[2] = (_Float16)p.z; h[3] = (_Float16)p.w; }
            }
        } else {
            for (int64_t i = i0; i < a.n[k] && i < i0 + 4; ++i) {
                float mi = a.m[k][i], vi = a.v[k][i], pi = a.p[k][i];
                one(a.g[k][i], mi, vi, pi, step_size, bc2_sqrt);
                a.m[k][i] = mi; a.v[k][i] = vi; a.p[k][i] = pi;
                if (k == 0 && half_out && i >= half_from) half_out[i - half_from] = (_Float16)pi;
            }
        }
    }
}

__global__ void __launch_bounds__(256) count_nan_kernel(const float *__restrict__ g, int64_t n, int32_t *__restrict__ count) {
    int local = 0;
    // NaN as the reference's guard (pipeline.py:520-529), and +-Inf as well: an overflowed fp16 activation gradient shows up
    // as Inf, which Adam would turn into NaN parameters one step later
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) local += !(fabsf(g[i]) <= 3.4028234664e38f);
    if (__ballot(local != 0) != 0ull && local) atomicAdd(count, local);
}


void free_train_state(mnf_field_t f) {
    if (f->cfg.mfma_bf16) bf16::free_train_state_impl(f); else f16::free_train_state_impl(f);
}
int forward_train(mnf_field_t f, const FieldIO &io, void *workspace, int64_t workspace_bytes, hipStream_t stream, bool deterministic) {
    MNF_REQUIRE(f, "field_forward_train: null handle");
    return f->cfg.mfma_bf16 ? bf16::forward_train_impl(f, io, workspace, workspace_bytes, stream, deterministic)
                            : f16::forward_train_impl(f, io, workspace, workspace_bytes, stream, deterministic);
}
int backward(mnf_field_t f, const float *positions, int64_t n, const int64_t *n_dev, const float *d_rgb, const float *d_density, const float *d_sem,
             const float *rgb, const float *density, void *workspace, int64_t workspace_bytes, float loss_scale, float *g_base, float *g_head,
             float *g_sem, bool zero_grads, bool positions_normalized, bool deterministic, hipStream_t stream) {
    MNF_REQUIRE(f, "field_backward: null handle");
    return f->cfg.mfma_bf16 ? bf16::backward_impl(f, positions, n, n_dev, d_rgb, d_density, d_sem, rgb, density, workspace, workspace_bytes, loss_scale,
                                                  g_base, g_head, g_sem, zero_grads, positions_normalized, deterministic, stream)
                            : f16::backward_impl(f, positions, n, n_dev, d_rgb, d_density, d_sem, rgb, density, workspace, workspace_bytes, loss_scale,
                                                 g_base, g_head, g_sem, zero_grads, positions_normalized, deterministic, stream);
}
}  // namespace mnf

using namespace mnf;

extern "C" int64_t mnf_field_train_workspace_bytes(mnf_field_t f, int64_t n) {
    if (!f) return -1;
    return f->cfg.mfma_bf16 ? bf16::train_workspace_bytes_impl(f, n) : f16::train_workspace_bytes_impl(f, n);
}

extern "C" int mnf_field_forward_train(mnf_field_t f, const float *positions, const float *directions, int64_t n,
                                       float *rgb, float *density, float *sem, void *workspace, int64_t workspace_bytes,
                                       mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_forward_train: null handle");
    MNF_REQUIRE(n == 0 || (positions && directions && rgb && density && sem), "field_forward_train: null pointer");
    FieldIO io = {};
    io.mode = 0; io.positions = positions; io.directions = directions; io.n = n;
    io.rgb = rgb; io.density = density; io.sem = sem;
    return f->cfg.mfma_bf16 ? bf16::forward_train_impl(f, io, workspace, workspace_bytes, as_stream(stream), false)
                            : f16::forward_train_impl(f, io, workspace, workspace_bytes, as_stream(stream), false);
}

extern "C" int mnf_field_forward_train_samples(mnf_field_t f, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                               const float *t_starts, const float *t_ends, int64_t n, float *rgb, float *density,
                                               float *sem, float *positions_out, void *workspace, int64_t workspace_bytes,
                                               mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_forward_train_samples: null handle");
    MNF_REQUIRE(n == 0 || (rays_o && rays_d && ray_indices && t_starts && t_ends && rgb && density && sem && positions_out),
                "field_forward_train_samples: null pointer");
    FieldIO io = {};
    io.mode = 1; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = ray_indices; io.t_starts = t_starts; io.t_ends = t_ends; io.n = n;
    io.rgb = rgb; io.density = density; io.sem = sem; io.positions_out = positions_out;
    return f->cfg.mfma_bf16 ? bf16::forward_train_impl(f, io, workspace, workspace_bytes, as_stream(stream), false)
                            : f16::forward_train_impl(f, io, workspace, workspace_bytes, as_stream(stream), false);
}

extern "C" int mnf_field_backward(mnf_field_t f, const float *positions, int64_t n,
                                  const float *d_rgb, const float *d_density, const float *d_sem,
                                  const float *rgb, const float *density,
                                  void *workspace, int64_t workspace_bytes, float loss_scale,
                                  float *g_base, float *g_head, float *g_sem, mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_backward: null handle");
    return backward(f, positions, n, nullptr, d_rgb, d_density, d_sem, rgb, density, workspace, workspace_bytes, loss_scale, g_base, g_head, g_sem, true,
                    false, false, as_stream(stream));
}

extern "C" int mnf_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t step, mnf_stream_t stream) {
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(params && grads && exp_avg && exp_avg_sq && step >= 1, "adam_step: bad arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const int64_t blocks = ceil_div(n, 256 * 4);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 8192 ? (blocks < 1 ? 1 : blocks) : 8192)), dim3(256), 0, as_stream(stream), params,
                       grads, exp_avg, exp_avg_sq, n, beta1, beta2, eps, (float)((double)lr / bc1), (float)std::sqrt(bc2));
    return launch_status("adam_kernel");
}

extern "C" int mnf_adam_step_guarded(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                                     float beta2, float eps, float *step_dev, const int32_t *skip_dev, float *hyper_dev, void *half_out,
                                     int64_t half_from, mnf_stream_t stream) {
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(params && grads && exp_avg && exp_avg_sq && step_dev && hyper_dev, "adam_step_guarded: bad arguments");
    MNF_REQUIRE(!half_out || (half_from >= 0 && half_from <= n), "adam_step_guarded: bad fp16 mirror range");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, s, step_dev, skip_dev, lr, beta1, beta2, hyper_dev);
    const int64_t blocks = ceil_div(n, 256 * 4);
    hipLaunchKernelGGL(adam_guarded_kernel, dim3((unsigned)(blocks < 8192 ? (blocks < 1 ? 1 : blocks) : 8192)), dim3(256), 0, s, params, grads, exp_avg,
                       exp_avg_sq, n, beta1, beta2, eps, (const float *)hyper_dev, reinterpret_cast<_Float16 *>(half_out), half_from);
    return launch_status("adam_guarded_kernel");
}

// pipeline.py:520-532 for one model as ONE call: the non-finite-gradient guard over the three parameter vectors, the three Adam updates
// (skipped on the device when the flag is raised) with the fp16 table mirror, and the handle's MLP fragments re-derived.
static int optimizer_step_impl(mnf_field_t f, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                               float *const *exp_avg_sq_host, float *const *step_dev_host, float lr, float beta1, float beta2, float eps,
                               int32_t *skip_dev, int32_t count_nonfinite, float *hyper_dev, const int64_t *counts_dev, int64_t *report_host, mnf_stream_t stream) {
    MNF_REQUIRE(f && params_host && grads_host && exp_avg_host && exp_avg_sq_host && step_dev_host && hyper_dev, "field_optimizer_step: null argument");
    MNF_REQUIRE(f->params_loaded, "field_optimizer_step: the handle holds no parameters yet (mnf_field_set_params)");
    const int64_t n[3] = {f->n_base, f->n_head, f->n_sem};
    Adam3 a;
    for (int k = 0; k < 3; ++k) {
        MNF_REQUIRE(n[k] == 0 || (params_host[k] && grads_host[k] && exp_avg_host[k] && exp_avg_sq_host[k] && step_dev_host[k]), "field_optimizer_step: null pointer");
        a.p[k] = params_host[k]; a.g[k] = grads_host[k]; a.m[k] = exp_avg_host[k]; a.v[k] = exp_avg_sq_host[k]; a.step[k] = step_dev_host[k]; a.n[k] = n[k];
    }
    MNF_REQUIRE(n[0] > 0 && n[1] > 0 && n[2] > 0, "field_optimizer_step: empty parameter vector");
    hipStream_t s = as_stream(stream);
    const int64_t total = n[0] + n[1] + n[2];
    if (count_nonfinite) {
        MNF_REQUIRE(skip_dev, "field_optimizer_step: counting non-finite gradients needs the skip flag");
        const int64_t blocks = ceil_div(total, 256 * 4 * 4);
        hipLaunchKernelGGL(count_nan3_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, a, skip_dev);
    }
    hipLaunchKernelGGL(adam_prepare3_kernel, dim3(1), dim3(64), 0, s, a, (const int32_t *)skip_dev, lr, beta1, beta2, hyper_dev, counts_dev,
                       reinterpret_cast<long long *>(report_host));
    const int64_t blocks = ceil_div(total, 256 * 4 * 2);
    hipLaunchKernelGGL(adam_guarded3_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, a, beta1, beta2, eps, (const float *)hyper_dev,
                       reinterpret_cast<_Float16 *>(f->d_table), (int64_t)f->n_base_mlp);
    int rc = launch_status("adam_guarded3_kernel");
    if (rc) return rc;
    return mnf_field_refresh_weights(f, params_host[0], params_host[1], params_host[2], stream);
}

extern "C" int mnf_field_optimizer_step(mnf_field_t f, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                                        float *const *exp_avg_sq_host, float *const *step_dev_host, float lr, float beta1, float beta2, float eps,
                                        int32_t *skip_dev, int32_t count_nonfinite, float *hyper_dev, mnf_stream_t stream) {
    return optimizer_step_impl(f, params_host, grads_host, exp_avg_host, exp_avg_sq_host, step_dev_host, lr, beta1, beta2, eps, skip_dev, count_nonfinite, hyper_dev,
                               nullptr, nullptr, stream);
}

extern "C" int mnf_field_optimizer_step_report(mnf_field_t f, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                                               float *const *exp_avg_sq_host, float *const *step_dev_host, float lr, float beta1, float beta2, float eps,
                                               int32_t *skip_dev, int32_t count_nonfinite, float *hyper_dev, const int64_t *counts_dev, int64_t *report_host,
                                               mnf_stream_t stream) {
    MNF_REQUIRE(report_host, "field_optimizer_step_report: null report buffer");
    return optimizer_step_impl(f, params_host, grads_host, exp_avg_host, exp_avg_sq_host, step_dev_host, lr, beta1, beta2, eps, skip_dev, count_nonfinite, hyper_dev,
                               counts_dev, report_host, stream);
}

extern "C" int mnf_count_nan(const float *values, int64_t n, int32_t *count, mnf_stream_t stream) {
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(values && count, "count_nan: null pointer");
    const int64_t blocks = ceil_div(n, 256 * 8);
    hipLaunchKernelGGL(count_nan_kernel, dim3((unsigned)(blocks < 4096 ? (blocks < 1 ? 1 : blocks) : 4096)), dim3(256), 0, as_stream(stream),
                       values, n, count);
    return launch_status("count_nan_kernel");
}
#endif  // MNF_BF16
