// nerfacc_cuda replacements: ray/AABB test, occupancy-grid traversal, packed scans, and the
// pinhole ray generator.  Build with -ffp-contract=off (see march_dev.h).
#include "common.h"
#include "march_dev.h"

namespace mnf {

// ------------------------------------------------------------------ ray_aabb_intersect
// grid.cu:284-313 — one (ray, aabb) pair per lane.
__global__ void __launch_bounds__(256) ray_aabb_kernel(int64_t numel, int32_t n_aabbs,
                                                       const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                       float near_plane, float far_plane,
                                                       const float *__restrict__ aabbs, float miss,
                                                       float *__restrict__ t_mins, float *__restrict__ t_maxs,
                                                       uint8_t *__restrict__ hits) {
    for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < numel; tid += (int64_t)blockDim.x * gridDim.x) {
        const int64_t r = tid / n_aabbs, a = tid % n_aabbs;
        const F3 o = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const F3 inv = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        float t0, t1;
        const bool hit = ray_aabb(o, inv, near_plane, far_plane, aabbs + 6 * a, t0, t1);
        t_mins[tid] = hit ? t0 : miss;
        t_maxs[tid] = hit ? t1 : miss;
        hits[tid] = hit ? 1 : 0;
    }
}

// ------------------------------------------------------------------ traverse_grids
struct SegOut {
    float *vals; int64_t *ray_indices; uint8_t *is_left; uint8_t *is_right; uint8_t *is_valid;
    const int64_t *chunk_starts; int64_t *chunk_cnts;
};

// Sink reproducing the interval / sample bookkeeping of grid.cu:219-257.
struct PackedSink {
    SegOut iv, sm;
    bool first_pass;
    int64_t iv_start, sm_start, tid;
    int64_t n_intervals;
    __device__ __forceinline__ void sample(float t_last, float t_next, bool continuous, int32_t n_samples) {
        if (iv.chunk_cnts) {
            if (!continuous) {
                if (!first_pass) {
                    const int64_t i0 = iv_start + n_intervals;
                    iv.vals[i0] = t_last; iv.ray_indices[i0] = tid; iv.is_left[i0] = 1;
                    iv.vals[i0 + 1] = t_next; iv.ray_indices[i0 + 1] = tid; iv.is_right[i0 + 1] = 1;
                }
                n_intervals += 2;
            } else {
                if (!first_pass) {
                    const int64_t i0 = iv_start + n_intervals;
                    iv.vals[i0] = t_next; iv.ray_indices[i0] = tid; iv.is_left[i0 - 1] = 1; iv.is_right[i0] = 1;
                }
                n_intervals += 1;
            }
        }
        if (sm.chunk_cnts && !first_pass) {
            const int64_t i0 = sm_start + n_samples;
            sm.vals[i0] = (t_next + t_last) * 0.5f;
            sm.ray_indices[i0] = tid;
            if (sm.is_valid) sm.is_valid[i0] = 1;
        }
    }
};

__global__ void __launch_bounds__(256) traverse_kernel(int32_t n_rays, const float *__restrict__ rays_o,
                                                       const float *__restrict__ rays_d, const uint8_t *__restrict__ rays_mask,
                                                       int32_t n_grids, I3 res, const uint8_t *__restrict__ binaries,
                                                       const float *__restrict__ aabbs, const uint8_t *__restrict__ hits,
                                                       const float *__restrict__ t_sorted, const int64_t *__restrict__ t_indices,
                                                       const float *__restrict__ near_planes, const float *__restrict__ far_planes,
                                                       float step_size, float cone_angle, int32_t limit, bool first_pass,
                                                       SegOut iv, SegOut sm, float *__restrict__ terminate_planes) {
    const int64_t cells = (int64_t)res.x * res.y * res.z;
    for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < n_rays; tid += (int64_t)blockDim.x * gridDim.x) {
        if (rays_mask && !rays_mask[tid]) continue;
        if (iv.chunk_cnts && !first_pass && iv.chunk_cnts[tid] == 0) continue;
        if (sm.chunk_cnts && !first_pass && sm.chunk_cnts[tid] == 0) continue;
        PackedSink sink;
        sink.iv = iv; sink.sm = sm; sink.first_pass = first_pass; sink.tid = tid; sink.n_intervals = 0;
        sink.iv_start = (!first_pass && iv.chunk_cnts) ? iv.chunk_starts[tid] : 0;
        sink.sm_start = (!first_pass && sm.chunk_cnts) ? sm.chunk_starts[tid] : 0;
        const float near_plane = near_planes[tid], far_plane = far_planes[tid];
        const F3 org = {rays_o[3 * tid], rays_o[3 * tid + 1], rays_o[3 * tid + 2]};
        const F3 dir = {rays_d[3 * tid], rays_d[3 * tid + 1], rays_d[3 * tid + 2]};
        const F3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
        const int64_t base_hits = tid * n_grids, base_t = tid * n_grids * 2;
        MarchState st = {near_plane, false, 0};
        for (int64_t i = base_t; i < base_t + n_grids * 2 - 1; ++i) {   // grid.cu:125-151
            const bool is_entering = t_indices[i] < n_grids;
            int64_t level = t_indices[i] % n_grids;
            if (!hits[base_hits + level]) continue;
            if (!is_entering) {
                if (t_indices[i + 1] < n_grids) continue;
                level = t_indices[i + 1] % n_grids;
                if (!hits[base_hits + level]) continue;
            }
            const float this_tmin = fmaxf(t_sorted[i], near_plane);
            const float this_tmax = fminf(t_sorted[i + 1], far_plane);
            if (this_tmin >= this_tmax) continue;
            march_segment(org, dir, inv, this_tmin, this_tmax, aabbs + level * 6, res, ByteGrid{binaries + level * cells},
                          step_size, cone_angle, limit, st, sink);
        }
        if (terminate_planes) terminate_planes[tid] = st.t_last;
        if (iv.chunk_cnts) iv.chunk_cnts[tid] = sink.n_intervals;
        if (sm.chunk_cnts) sm.chunk_cnts[tid] = st.n_samples;
    }
}

// ------------------------------------------------------------------ single-pass sampler (one grid level)
// `OccGridEstimator.sampling` (occ_grid.py:80-238) only needs (t_start, t_end, ray) per sample, and a ray is one long
// sequential chain (~1 ms for the longest training ray), so marching every ray twice (count pass, fill pass:
// grid.cu:320-474) doubles the latency.  Here every ray is marched once into its own row of a caller-provided scratch
// [n_rays][cap]; rows are then packed by `compact_samples_kernel`.  A ray with more than `cap` samples only counts
// (the host mirror then falls back to the two-pass traversal).  The occupancy grid is bit-packed into LDS by each
// workgroup (a dependent global load per visited cell otherwise).  Same t values as traverse_kernel, bit for bit.
constexpr int kSamplerGridWords = 16384;   // 64 KB of LDS = 524 288 cells

#ifndef MNF_SAMPLER_THREADS
#define MNF_SAMPLER_THREADS 1024     /* threads per workgroup of sample_rays_kernel: sixteen waves share one LDS copy of the grid, two workgroups per compute unit (57 VGPRs).
                                        The kernel is one dependent chain per ray and a wave pays for the union of its lanes' chains: with full waves its time did not depend on
                                        the ray count or the workgroup size (197 us for 8192 and for 2000 rays with 256 threads, 184 / 187 us with 64, profiles/r04_exp_sampler.txt);
                                        with ONE ray per wave (round 6, `rpw` below) 8192 rays take 0.179 instead of 0.234 ms, 2000 rays 0.145 instead of 0.242 */
#endif
#ifndef MNF_SAMPLER_EXP
#define MNF_SAMPLER_EXP 0            /* timing experiments only: 1 = no sample stores (results invalid) */
#endif
struct ScratchSink {
    float *ts, *te;
    int32_t cap;
    __device__ __forceinline__ void sample(float t_last, float t_next, bool, int32_t k) {
#if MNF_SAMPLER_EXP == 1
        if (k == 0x7fffffff) { ts[0] = t_last; te[0] = t_next; }
#elif MNF_SAMPLER_EXP == 2
        if (k < cap) *reinterpret_cast<float2 *>(ts + 2 * (k & 0x3ff)) = float2{t_last, t_next};     /* one 8-byte store per sample (layout experiment: results invalid) */
#else
        if (k < cap) { ts[k] = t_last; te[k] = t_next; }
#endif
    }
};

// `rpw` rays per wave (16 / 32 / 64): a wave pays for the UNION of its lanes' paths — empty-space stepping, sampling and cell steps have different trip counts in
// every lane — so a batch that cannot fill the chip with full waves is marched with fewer rays per wave on more waves (lanes >= rpw idle).  Per-ray arithmetic is
// unchanged: same samples, bit for bit.
template <bool LDS_GRID, bool MULTI>
__global__ void __launch_bounds__(MNF_SAMPLER_THREADS) sample_rays_kernel(int32_t n_rays, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                          I3 res, const uint8_t *__restrict__ binaries, const LevelBoxes boxes, const float *__restrict__ near_planes,
                                                          const float *__restrict__ far_planes, float step_size, float cone_angle,
                                                          int32_t cap, float *__restrict__ scratch_ts, float *__restrict__ scratch_te,
                                                          int64_t *__restrict__ counts, const uint32_t *__restrict__ bitgrid, int32_t rpw) {
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[LDS_GRID ? kSamplerGridWords : 1];
    const int64_t cells = (int64_t)res.x * res.y * res.z;
    if (LDS_GRID) {   // every level's bits: level l at word l * words_per_level (the layout of mnf_pack_bitgrid)
        const int wpl = boxes.words_per_level, n_words = wpl * boxes.n;
        if (bitgrid) {   // the estimator's bit-packed grid (kept current by mnf_occ_binarize): 1/8 of the bytes, no packing here
            // 16-byte loads, all of a thread's requests in flight before its first LDS write (the tail by single words)
            const int n4 = (reinterpret_cast<uintptr_t>(bitgrid) & 15) == 0 ? n_words >> 2 : 0;
            const uint4 *src4 = reinterpret_cast<const uint4 *>(bitgrid);
            uint4 *dst4 = reinterpret_cast<uint4 *>(s_bits);
            for (int w = threadIdx.x; w < n4; w += blockDim.x) dst4[w] = src4[w];
            for (int w = 4 * n4 + threadIdx.x; w < n_words; w += blockDim.x) s_bits[w] = bitgrid[w];
        } else {
            for (int w = threadIdx.x; w < n_words; w += blockDim.x) {
                const int lvl = w / wpl, wl = w - lvl * wpl;
                uint32_t v = 0;
                for (int b = 0; b < 32; ++b) {
                    const int64_t c = (int64_t)wl * 32 + b;
                    if (c < cells && binaries[(int64_t)lvl * cells + c]) v |= 1u << b;
                }
                s_bits[w] = v;
            }
        }
        __syncthreads();
    }
    const float *ab = boxes.ab[0];
#if MNF_SAMPLER_EXP == 3
    if (n_rays > 0) { for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)blockDim.x * gridDim.x) counts[r] = s_bits[r & 1023] & 1; return; }   /* staging only */
#endif
    const int lane_ = threadIdx.x & 63;
    if (lane_ >= rpw) return;
    const int64_t wave_ = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves_ = ((int64_t)blockDim.x * gridDim.x) >> 6;
    for (int64_t r = wave_ * rpw + lane_; r < n_rays; r += n_waves_ * rpw) {
        const F3 org = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const F3 dir = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        const F3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
        const float near_plane = near_planes[r], far_plane = far_planes[r];
        MarchState st = {near_plane, false, 0};
        ScratchSink sink = {scratch_ts + (int64_t)r * cap, scratch_te + (int64_t)r * cap, cap};
        float t0, t1;
        if (MULTI) {
            if (LDS_GRID) {
                const uint32_t *bits = s_bits; const int wpl = boxes.words_per_level;
                march_levels(org, dir, inv, near_plane, far_plane, boxes, res, [=](int level) { return BitGrid{bits + level * wpl}; }, step_size, cone_angle, 0, st, sink);
            } else {
                march_levels(org, dir, inv, near_plane, far_plane, boxes, res, [=](int level) { return ByteGrid{binaries + level * cells}; }, step_size, cone_angle, 0, st, sink);
            }
        } else if (ray_aabb(org, inv, -INFINITY, INFINITY, ab, t0, t1)) {   // grid.py:150-160 default planes, then grid.cu:125-151
            const float this_tmin = fmaxf(t0, near_plane), this_tmax = fminf(t1, far_plane);
            if (this_tmin < this_tmax) {
                if (LDS_GRID) march_segment(org, dir, inv, this_tmin, this_tmax, ab, res, BitGrid{s_bits}, step_size, cone_angle, 0, st, sink);
                else march_segment(org, dir, inv, this_tmin, this_tmax, ab, res, ByteGrid{binaries}, step_size, cone_angle, 0, st, sink);
            }
        }
        counts[r] = st.n_samples;
    }
}

// one wave per ray: scratch row -> packed position
__global__ void __launch_bounds__(64) compact_samples_kernel(const float *__restrict__ scratch_ts, const float *__restrict__ scratch_te,
                                                             int32_t cap, const int64_t *__restrict__ chunk_starts,
                                                             const int64_t *__restrict__ counts, int32_t n_rays,
                                                             float *__restrict__ t_starts, float *__restrict__ t_ends,
                                                             int64_t *__restrict__ ray_indices) {
    for (int32_t r = blockIdx.x; r < n_rays; r += gridDim.x) {
        const int64_t s = chunk_starts[r];
        const int c = (int)counts[r];
        const float *ts = scratch_ts + (int64_t)r * cap, *te = scratch_te + (int64_t)r * cap;
        for (int k = threadIdx.x; k < c; k += 64) { t_starts[s + k] = ts[k]; t_ends[s + k] = te[k]; ray_indices[s + k] = r; }
    }
}

// ------------------------------------------------------------------ packed scans
// One lane per ray: chunks are short (<= a few hundred samples) and consecutive lanes own
// consecutive chunks, so the wave streams a contiguous window.  Sequential fp32 order == oracle.
__global__ void __launch_bounds__(256) exclusive_sum_kernel(int32_t n_rays, const int64_t *__restrict__ starts,
                                                            const int64_t *__restrict__ cnts, const float *__restrict__ in,
                                                            float *__restrict__ out, bool backward) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)blockDim.x * gridDim.x) {
        const int64_t s = starts[r], c = cnts[r];
        float acc = 0.0f;
        if (!backward) for (int64_t k = 0; k < c; ++k) { out[s + k] = acc; acc += in[s + k]; }
        else           for (int64_t k = c - 1; k >= 0; --k) { out[s + k] = acc; acc += in[s + k]; }
    }
}

// volrend.py:258-267 + :361-365 fused: sigma*dt -> alpha, T = exp(-excl_sum) * prefix, w = T*alpha.
// One wave per ray, 64 samples per pass with a wave scan (a thread per ray serialises ~250 dependent loads per ray
// on the 8192-ray training batches).
__global__ void __launch_bounds__(64) weight_from_density_kernel(int32_t n_rays, const int64_t *__restrict__ starts,
                                                                 const int64_t *__restrict__ cnts,
                                                                 const float *__restrict__ ts, const float *__restrict__ te,
                                                                 const float *__restrict__ sig, const float *__restrict__ prefix,
                                                                 float *__restrict__ w, float *__restrict__ tr, float *__restrict__ al) {
    const int lane = threadIdx.x;
    for (int32_t r = blockIdx.x; r < n_rays; r += gridDim.x) {
        const int64_t s = starts[r];
        const int c = (int)cnts[r];
        float carry = 0.0f;
        for (int base = 0; base < c; base += 64) {
            const bool valid = base + lane < c;
            const int64_t k = s + base + lane;
            const float sdt = valid ? sig[k] * (te[k] - ts[k]) : 0.0f;
            float incl = sdt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float u = __shfl_up(incl, d, 64);
                if (lane >= d) incl += u;
            }
            const float alpha = 1.0f - expf(-sdt);
            float trans = expf(-((incl - sdt) + carry));
            carry += __shfl(incl, 63, 64);
            if (valid) {
                if (prefix) trans *= prefix[k];
                if (w) w[k] = trans * alpha;
                if (tr) tr[k] = trans;
                if (al) al[k] = alpha;
            }
        }
    }
}

// pack_info (perception/nerfacc/nerfacc/pack.py:10-38) for ray indices that are already grouped by ray (what the
// marcher emits): run boundaries instead of one atomic per sample.  first/last must be zero-filled.
__global__ void __launch_bounds__(256) run_bounds_kernel(const int64_t *__restrict__ ray_indices, int64_t n,
                                                         int64_t *__restrict__ first, int64_t *__restrict__ last) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t r = ray_indices[i];
        if (i == 0 || ray_indices[i - 1] != r) first[r] = i;
        if (i == n - 1 || ray_indices[i + 1] != r) last[r] = i + 1;
    }
}

// ------------------------------------------------------------------ int64 exclusive prefix sum (chunk starts from chunk counts)
// `RaySegmentsSpec::memalloc_data_from_chunk` (include/data_spec.hpp:86-96) and `pack_info` (pack.py:10-38) turn per-ray
// counts into chunk starts.  Three short launches: tile sums, a one-workgroup scan of the tile sums, tile-local scans.
constexpr int kScanTile = 2048;   // 256 threads x 8 values

__device__ __forceinline__ int64_t block_exclusive_scan_256(int64_t v, int64_t *s_wave, int64_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int64_t off = 0;
    for (int k = 0; k < wave; ++k) off += s_wave[k];
    block_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return off + incl - v;
}

__global__ void __launch_bounds__(256) scan_tile_sums_kernel(const int64_t *__restrict__ in, int64_t n, int64_t *__restrict__ tile_sums) {
    __shared__ int64_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
    int64_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) v += in[base + k];
    int64_t total;
    block_exclusive_scan_256(v, s_wave, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) scan_spine_kernel(int64_t *__restrict__ tile_sums, int64_t n_tiles, int64_t *__restrict__ total_out) {
    __shared__ int64_t s_wave[4];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n_tiles ? tile_sums[i] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan_256(v, s_wave, total);
        if (i < n_tiles) tile_sums[i] = s_carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = s_carry;
}

__global__ void __launch_bounds__(256) scan_apply_kernel(const int64_t *__restrict__ in, int64_t n, const int64_t *__restrict__ tile_offsets,
                                                         int64_t *__restrict__ out) {
    __shared__ int64_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
    int64_t x[8], v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = base + k < n ? in[base + k] : 0; v += x[k]; }
    int64_t total;
    int64_t run = tile_offsets[blockIdx.x] + block_exclusive_scan_256(v, s_wave, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += x[k]; }
}

// pack.py:10-38 for ray indices in ANY order: per-ray sample counts (the reference uses index_add_ of ones)
__global__ void __launch_bounds__(256) count_per_ray_kernel(const int64_t *__restrict__ ray_indices, int64_t n, unsigned long long *__restrict__ cnts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)
        atomicAdd(&cnts[ray_indices[i]], 1ull);
}

// [n_rays] starts + cnts -> interleaved [n_rays, 2] packed_info
__global__ void __launch_bounds__(256) interleave2_kernel(const int64_t *__restrict__ a, const int64_t *__restrict__ b, int64_t n,
                                                          int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        out[2 * i] = a[i]; out[2 * i + 1] = b[i];
    }
}

// ------------------------------------------------------------------ accumulate_along_rays (volrend.py:486-576), packed branch
// outputs[ray] += w * v for samples in any order (float atomics, like the reference's index_add_), and its adjoint.
__global__ void __launch_bounds__(256) accumulate_fwd_kernel(const float *__restrict__ w, const float *__restrict__ v, const int64_t *__restrict__ ri,
                                                             int64_t n, int32_t D, float *__restrict__ out) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t k = i / D;
        const int d = (int)(i - k * D);
        atomicAdd(&out[ri[k] * D + d], v ? w[k] * v[i] : w[k]);
    }
}

__global__ void __launch_bounds__(256) accumulate_bwd_kernel(const float *__restrict__ w, const float *__restrict__ v, const int64_t *__restrict__ ri,
                                                             int64_t n, int32_t D, const float *__restrict__ g_out, float *__restrict__ g_w,
                                                             float *__restrict__ g_v) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)blockDim.x * gridDim.x) {
        const float *g = g_out + ri[k] * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
            acc += v ? g[d] * v[k * D + d] : g[d];
            if (g_v) g_v[k * D + d] = w[k] * g[d];
        }
        if (g_w) g_w[k] = acc;
    }
}

// ------------------------------------------------------------------ ray generation
// habitat_to_data.py:274-301; arithmetic order pinned by tests/golden/raygen.npz (oracle/render.py).
__global__ void __launch_bounds__(256) raygen_kernel(const float *__restrict__ c2w, int32_t n_views, int32_t width, int32_t height,
                                                     float focal, const int64_t *__restrict__ pix_idx, int64_t n_pix,
                                                     float *__restrict__ origins, float *__restrict__ viewdirs) {
    const int64_t total = (int64_t)n_views * n_pix;
    const float cx = (float)width * 0.5f, cy = (float)height * 0.5f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t v = i / n_pix, p = i % n_pix;
        const int64_t pix = pix_idx ? pix_idx[p] : p;
        const float x = (float)(pix % width), y = (float)(pix / width);
        const float *m = c2w + v * 12;
        const float cam0 = (x - cx + 0.5f) / focal;
        const float cam1 = (y - cy + 0.5f) / focal * -1.0f;
        const float cam2 = -1.0f;
        const float dx = (cam0 * m[0] + cam1 * m[1]) + cam2 * m[2];
        const float dy = (cam0 * m[4] + cam1 * m[5]) + cam2 * m[6];
        const float dz = (cam0 * m[8] + cam1 * m[9]) + cam2 * m[10];
        const float n = sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
        origins[3 * i] = m[3]; origins[3 * i + 1] = m[7]; origins[3 * i + 2] = m[11];
        viewdirs[3 * i] = dx / n; viewdirs[3 * i + 1] = dy / n; viewdirs[3 * i + 2] = dz / n;
    }
}

// ------------------------------------------------------------------ dataset ingest: pixel gather
// habitat_to_data.py:229-232: rgb = images[id, y, x] / 255.0, dep = depths[id, y, x], sem = semantics[id, y, x] for the pixels
// of one image, from the reference's storage dtypes (u8 / f32 / i64) or the packed on-device layout (u8 / f16 / u8).
__global__ void __launch_bounds__(256) gather_pixels_kernel(const uint8_t *__restrict__ images, const void *__restrict__ depths, int depth_f16,
                                                            const void *__restrict__ sems, int sem_u8, int64_t pixels_per_image,
                                                            const int64_t *__restrict__ image_id, const int64_t *__restrict__ pix, int64_t n,
                                                            float *__restrict__ rgb, float *__restrict__ dep, int64_t *__restrict__ sem) {
    const int64_t base = image_id[0] * pixels_per_image;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t p = base + pix[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = (float)images[3 * p + k] / 255.0f;
        dep[i] = depth_f16 ? (float)reinterpret_cast<const _Float16 *>(depths)[p] : reinterpret_cast<const float *>(depths)[p];
        sem[i] = sem_u8 ? (int64_t)reinterpret_cast<const uint8_t *>(sems)[p] : reinterpret_cast<const int64_t *>(sems)[p];
    }
}

// ------------------------------------------------------------------ planner hand-off
// scripts/pipeline.py:1043-1049 + planning/planning_funcs.py:243-261: slice the occupancy grids of the ensemble at
// height index `y_slice`, merge (any member occupied), dilate with a 3x3 box ("symm" boundary == clamped neighbours)
__global__ void __launch_bounds__(256) planner_map_kernel(const uint8_t *__restrict__ binaries, int n_members, int X, int Y, int Z,
                                                          int y_slice, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)X * Z; i += (int64_t)blockDim.x * gridDim.x) {
        const int x = (int)(i / Z), z = (int)(i % Z);
        int hit = 0;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dz = -1; dz <= 1; ++dz) {
                const int xx = min(max(x + dx, 0), X - 1), zz = min(max(z + dz, 0), Z - 1);
                for (int m = 0; m < n_members; ++m) hit |= binaries[(((int64_t)m * X + xx) * Y + y_slice) * Z + zz];
            }
        out[i] = hit ? 1 : 0;
    }
}

static inline int grid_for(int64_t n, int threads) {
    int64_t b = ceil_div(n, threads);
    return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

}  // namespace mnf

using namespace mnf;

extern "C" int mnf_ray_aabb_intersect(const float *rays_o, const float *rays_d, int32_t n_rays, const float *aabbs,
                                      int32_t n_aabbs, float near_plane, float far_plane, float miss_value,
                                      float *t_mins, float *t_maxs, uint8_t *hits, mnf_stream_t stream) {
    MNF_REQUIRE(n_rays >= 0 && n_aabbs > 0, "ray_aabb_intersect: bad sizes (%d rays, %d aabbs)", n_rays, n_aabbs);
    const int64_t numel = (int64_t)n_rays * n_aabbs;
    if (numel == 0) return MNF_OK;
    MNF_REQUIRE(rays_o && rays_d && aabbs && t_mins && t_maxs && hits, "ray_aabb_intersect: null pointer");
    hipLaunchKernelGGL(ray_aabb_kernel, dim3(grid_for(numel, 256)), dim3(256), 0, as_stream(stream), numel, n_aabbs,
                       rays_o, rays_d, near_plane, far_plane, aabbs, miss_value, t_mins, t_maxs, hits);
    return launch_status("ray_aabb_kernel");
}

extern "C" int mnf_traverse_grids(const float *rays_o, const float *rays_d, const uint8_t *rays_mask, int32_t n_rays,
                                  const uint8_t *binaries, const float *aabbs, int32_t n_grids,
                                  int32_t res_x, int32_t res_y, int32_t res_z,
                                  const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
                                  const float *near_planes, const float *far_planes,
                                  float step_size, float cone_angle, int32_t traverse_steps_limit, int32_t first_pass,
                                  float *iv_vals, int64_t *iv_ray_indices, uint8_t *iv_is_left, uint8_t *iv_is_right,
                                  const int64_t *iv_chunk_starts, int64_t *iv_chunk_cnts,
                                  float *sm_vals, int64_t *sm_ray_indices, uint8_t *sm_is_valid,
                                  const int64_t *sm_chunk_starts, int64_t *sm_chunk_cnts,
                                  float *terminate_planes, mnf_stream_t stream) {
    MNF_REQUIRE(n_rays >= 0 && n_grids > 0 && n_grids <= 8, "traverse_grids: bad sizes");
    if (n_rays == 0) return MNF_OK;
    MNF_REQUIRE(rays_o && rays_d && binaries && aabbs && hits && t_sorted && t_indices && near_planes && far_planes,
                "traverse_grids: null input pointer");
    if (!first_pass) {
        // value buffers may legitimately be empty (NULL) when no ray produced a sample: the fill pass skips such rays
        MNF_REQUIRE(!iv_chunk_cnts || iv_chunk_starts, "traverse_grids: fill pass needs interval chunk_starts");
        MNF_REQUIRE(!sm_chunk_cnts || sm_chunk_starts, "traverse_grids: fill pass needs sample chunk_starts");
    }
    SegOut iv = {iv_vals, iv_ray_indices, iv_is_left, iv_is_right, nullptr, iv_chunk_starts, iv_chunk_cnts};
    SegOut sm = {sm_vals, sm_ray_indices, nullptr, nullptr, sm_is_valid, sm_chunk_starts, sm_chunk_cnts};
    const I3 res = {res_x, res_y, res_z};
    hipLaunchKernelGGL(traverse_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, as_stream(stream), n_rays, rays_o, rays_d,
                       rays_mask, n_grids, res, binaries, aabbs, hits, t_sorted, t_indices, near_planes, far_planes,
                       step_size, cone_angle, traverse_steps_limit, first_pass != 0, iv, sm, terminate_planes);
    return launch_status("traverse_kernel");
}

extern "C" int mnf_sample_rays_levels(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries, int32_t n_levels, int32_t res_x,
                                      int32_t res_y, int32_t res_z, const float *aabb_host, const float *near_planes, const float *far_planes,
                                      float step_size, float cone_angle, int32_t cap, float *scratch_ts, float *scratch_te, int64_t *counts,
                                      const uint32_t *bitgrid, mnf_stream_t stream) {
    if (n_rays == 0) return MNF_OK;
    MNF_REQUIRE(rays_o && rays_d && binaries && aabb_host && near_planes && far_planes && scratch_ts && scratch_te && counts,
                "sample_rays: null pointer");
    MNF_REQUIRE(res_x > 0 && res_y > 0 && res_z > 0 && cap > 0 && step_size > 0.f, "sample_rays: bad sizes");
    MNF_REQUIRE(n_levels >= 1 && n_levels <= 4, "sample_rays: 1 to 4 occupancy levels");
    const I3 res = {res_x, res_y, res_z};
    LevelBoxes boxes;
    LevelBoxes boxes0 = {}; boxes = boxes0;
    boxes.n = n_levels;
    boxes.words_per_level = (int32_t)(((int64_t)res_x * res_y * res_z + 31) / 32);
    for (int l = 0; l < n_levels; ++l)
        for (int k = 0; k < 6; ++k) boxes.ab[l][k] = aabb_host[6 * l + k];
    ProfScope ps("sample_rays", as_stream(stream));
    const bool lds = (int64_t)boxes.words_per_level * n_levels <= (int64_t)kSamplerGridWords;
    // rays per wave: as few as keep the launch within the waves the chip holds at once (256 compute units x 2 workgroups x 16 waves = 8192): one ray per wave up to
    // 8192 rays, two up to 16 384 ... (profiles/r06_sampler_rpw.txt: 0.179 / 0.188 / 0.222 / 0.249 / 0.234 ms for 8192 rays with 1 / 2 / 4 / 8 / 64 rays per wave)
#ifndef MNF_SAMPLER_RPW
#define MNF_SAMPLER_RPW 0            /* 0 = by batch size (below); 1 .. 64: fixed (A/B builds) */
#endif
    int rpw = MNF_SAMPLER_RPW ? MNF_SAMPLER_RPW : 1;
    while (!MNF_SAMPLER_RPW && rpw < 64 && (int64_t)rpw * 8192 < n_rays) rpw <<= 1;
    const int grid = grid_for((int64_t)ceil_div(n_rays, rpw) * 64, MNF_SAMPLER_THREADS);
#define MNF_SAMPLE(LDS, ML) hipLaunchKernelGGL((sample_rays_kernel<LDS, ML>), dim3(grid), dim3(MNF_SAMPLER_THREADS), 0, as_stream(stream), n_rays, rays_o, rays_d, res, binaries, boxes, \
                                               near_planes, far_planes, step_size, cone_angle, cap, scratch_ts, scratch_te, counts, bitgrid, rpw)
    if (n_levels > 1) { if (lds) MNF_SAMPLE(true, true); else MNF_SAMPLE(false, true); }
    else { if (lds) MNF_SAMPLE(true, false); else MNF_SAMPLE(false, false); }
#undef MNF_SAMPLE
    return launch_status("sample_rays_kernel");
}

extern "C" int mnf_sample_rays(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries, int32_t res_x,
                               int32_t res_y, int32_t res_z, const float *aabb_host, const float *near_planes, const float *far_planes,
                               float step_size, float cone_angle, int32_t cap, float *scratch_ts, float *scratch_te, int64_t *counts,
                               const uint32_t *bitgrid, mnf_stream_t stream) {
    return mnf_sample_rays_levels(rays_o, rays_d, n_rays, binaries, 1, res_x, res_y, res_z, aabb_host, near_planes, far_planes, step_size, cone_angle, cap,
                                  scratch_ts, scratch_te, counts, bitgrid, stream);
}

extern "C" int mnf_compact_samples(const float *scratch_ts, const float *scratch_te, int32_t cap, const int64_t *chunk_starts,
                                   const int64_t *counts, int32_t n_rays, float *t_starts, float *t_ends, int64_t *ray_indices,
                                   mnf_stream_t stream) {
    if (n_rays == 0) return MNF_OK;
    MNF_REQUIRE(scratch_ts && scratch_te && chunk_starts && counts && cap > 0, "compact_samples: null pointer");
    hipLaunchKernelGGL(compact_samples_kernel, dim3(n_rays < 65535 ? n_rays : 65535), dim3(64), 0, as_stream(stream), scratch_ts, scratch_te,
                       cap, chunk_starts, counts, n_rays, t_starts, t_ends, ray_indices);
    return launch_status("compact_samples_kernel");
}

extern "C" int mnf_exclusive_sum(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                 const float *inputs, float *outputs, int64_t n_edges, int32_t backward, mnf_stream_t stream) {
    if (n_rays == 0 || n_edges == 0) return MNF_OK;
    MNF_REQUIRE(chunk_starts && chunk_cnts && inputs && outputs, "exclusive_sum: null pointer");
    hipLaunchKernelGGL(exclusive_sum_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, as_stream(stream), n_rays,
                       chunk_starts, chunk_cnts, inputs, outputs, backward != 0);
    return launch_status("exclusive_sum_kernel");
}

extern "C" int mnf_render_weight_from_density(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                              const float *t_starts, const float *t_ends, const float *sigmas,
                                              const float *prefix_trans, int64_t n_samples,
                                              float *weights, float *trans, float *alphas, mnf_stream_t stream) {
    if (n_rays == 0 || n_samples == 0) return MNF_OK;
    MNF_REQUIRE(chunk_starts && chunk_cnts && t_starts && t_ends && sigmas, "render_weight_from_density: null pointer");
    hipLaunchKernelGGL(weight_from_density_kernel, dim3(n_rays < (1 << 20) ? n_rays : (1 << 20)), dim3(64), 0, as_stream(stream), n_rays,
                       chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas);
    return launch_status("weight_from_density_kernel");
}

extern "C" int mnf_run_bounds(const int64_t *ray_indices, int64_t n_samples, int64_t *first, int64_t *last, mnf_stream_t stream) {
    if (n_samples == 0) return MNF_OK;
    MNF_REQUIRE(ray_indices && first && last, "run_bounds: null pointer");
    hipLaunchKernelGGL(run_bounds_kernel, dim3(grid_for(n_samples, 256)), dim3(256), 0, as_stream(stream), ray_indices, n_samples,
                       first, last);
    return launch_status("run_bounds_kernel");
}

extern "C" int64_t mnf_scan_workspace_bytes(int64_t n) { return n < 0 ? -1 : (ceil_div(n > 0 ? n : 1, kScanTile) + 1) * (int64_t)sizeof(int64_t); }

// short inputs (the per-ray counts of a train batch): one workgroup, one launch instead of three (each small launch is ~5 us of the stream's time)
constexpr int kScanSmall = 256 * 64;
// Each of the four waves owns a contiguous quarter and walks it in runs of 64 consecutive values (one coalesced 512-byte load, one wave scan): a thread per `per`
// consecutive values read at a 256-byte lane stride and took 24 us for the 8192 rays of a config-5 batch.
__device__ __forceinline__ int64_t wave_inclusive_scan_i64(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}
template <int kRuns>      // runs of 64 values per wave: 256 * kRuns >= n
__global__ void __launch_bounds__(256) scan_small_kernel(const int64_t *__restrict__ in, int64_t n, int64_t *__restrict__ out, int64_t *__restrict__ total_out) {
    __shared__ int64_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t quarter = ((n + 255) / 256) * 64;              // a multiple of 64: runs never straddle two waves
    const int64_t q0 = wave * quarter, q1 = q0 + quarter < n ? q0 + quarter : n;
    // the quarter's values in registers first (every load in flight at once; a load per run in turn was 32 memory round trips at 8192 values)
    int64_t x[kRuns];
    int64_t sum = 0;
#pragma unroll
    for (int r = 0; r < kRuns; ++r) {
        const int64_t i = q0 + r * 64 + lane;
        x[r] = i < q1 ? in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < kRuns; ++r) sum += x[r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    int64_t carry = 0, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) carry += s_wave[w]; total += s_wave[w]; }
#pragma unroll
    for (int r = 0; r < kRuns; ++r) {
        if (q0 + r * 64 >= q1) break;                            // (wave-uniform)
        const int64_t i = q0 + r * 64 + lane;
        const int64_t incl = wave_inclusive_scan_i64(x[r], lane);
        if (i < q1) out[i] = carry + incl - x[r];
        carry += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

static int exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, int64_t *total, int64_t *tiles, hipStream_t s) {
    if (n <= kScanSmall && in != out) {
        if (n <= 256 * 8) hipLaunchKernelGGL(scan_small_kernel<8>, dim3(1), dim3(256), 0, s, in, n, out, total);
        else if (n <= 256 * 16) hipLaunchKernelGGL(scan_small_kernel<16>, dim3(1), dim3(256), 0, s, in, n, out, total);
        else if (n <= 256 * 32) hipLaunchKernelGGL(scan_small_kernel<32>, dim3(1), dim3(256), 0, s, in, n, out, total);
        else hipLaunchKernelGGL(scan_small_kernel<kScanSmall / 256>, dim3(1), dim3(256), 0, s, in, n, out, total);
        return launch_status("scan_small_kernel");
    }
    const int64_t n_tiles = ceil_div(n, kScanTile);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, in, n, tiles);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(256), 0, s, tiles, n_tiles, total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, in, n, tiles, out);
    return launch_status("scan_apply_kernel");
}

extern "C" int mnf_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, int64_t *total_out, void *workspace, int64_t workspace_bytes,
                                      mnf_stream_t stream) {
    if (n == 0) {
        if (total_out) MNF_HIP(hipMemsetAsync(total_out, 0, sizeof(int64_t), as_stream(stream)));
        return MNF_OK;
    }
    MNF_REQUIRE(in && out && workspace && n > 0, "exclusive_scan_i64: bad arguments");
    MNF_REQUIRE(workspace_bytes >= mnf_scan_workspace_bytes(n), "exclusive_scan_i64: workspace too small");
    return exclusive_scan_i64(in, n, out, total_out, (int64_t *)workspace, as_stream(stream));
}

extern "C" int mnf_pack_info(const int64_t *ray_indices, int64_t n_samples, int64_t n_rays, int64_t *packed_info, void *workspace,
                             int64_t workspace_bytes, mnf_stream_t stream) {
    if (n_rays == 0) return MNF_OK;
    MNF_REQUIRE(packed_info && workspace && n_rays > 0 && n_samples >= 0 && (ray_indices || n_samples == 0), "pack_info: bad arguments");
    const int64_t need = 2 * n_rays * (int64_t)sizeof(int64_t) + mnf_scan_workspace_bytes(n_rays);
    MNF_REQUIRE(workspace_bytes >= need, "pack_info: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
    hipStream_t s = as_stream(stream);
    int64_t *cnts = (int64_t *)workspace, *starts = cnts + n_rays, *tiles = starts + n_rays;
    MNF_HIP(hipMemsetAsync(cnts, 0, (size_t)n_rays * sizeof(int64_t), s));
    if (n_samples) hipLaunchKernelGGL(count_per_ray_kernel, dim3(grid_for(n_samples, 256)), dim3(256), 0, s, ray_indices, n_samples, (unsigned long long *)cnts);
    int rc = exclusive_scan_i64(cnts, n_rays, starts, nullptr, tiles, s);
    if (rc) return rc;
    hipLaunchKernelGGL(interleave2_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, starts, cnts, n_rays, packed_info);
    return launch_status("interleave2_kernel");
}

extern "C" int mnf_accumulate_along_rays(const float *weights, const float *values, const int64_t *ray_indices, int64_t n_samples, int32_t dim,
                                         float *outputs, mnf_stream_t stream) {
    if (n_samples == 0) return MNF_OK;
    MNF_REQUIRE(weights && ray_indices && outputs && dim >= 1, "accumulate_along_rays: bad arguments");
    hipLaunchKernelGGL(accumulate_fwd_kernel, dim3(grid_for(n_samples * dim, 256)), dim3(256), 0, as_stream(stream), weights, values, ray_indices,
                       n_samples, dim, outputs);
    return launch_status("accumulate_fwd_kernel");
}

extern "C" int mnf_accumulate_along_rays_backward(const float *weights, const float *values, const int64_t *ray_indices, int64_t n_samples,
                                                  int32_t dim, const float *grad_outputs, float *grad_weights, float *grad_values,
                                                  mnf_stream_t stream) {
    if (n_samples == 0) return MNF_OK;
    MNF_REQUIRE(weights && ray_indices && grad_outputs && dim >= 1 && (values || !grad_values), "accumulate_along_rays_backward: bad arguments");
    hipLaunchKernelGGL(accumulate_bwd_kernel, dim3(grid_for(n_samples, 256)), dim3(256), 0, as_stream(stream), weights, values, ray_indices,
                       n_samples, dim, grad_outputs, grad_weights, grad_values);
    return launch_status("accumulate_bwd_kernel");
}

extern "C" int mnf_generate_rays(const float *c2w, int32_t n_views, int32_t width, int32_t height, float focal,
                                 const int64_t *pix_idx, int64_t n_pix, float *origins, float *viewdirs, mnf_stream_t stream) {
    MNF_REQUIRE(n_views >= 0 && width > 0 && height > 0 && focal > 0, "generate_rays: bad arguments");
    if (!pix_idx) n_pix = (int64_t)width * height;
    if (n_views == 0 || n_pix == 0) return MNF_OK;
    MNF_REQUIRE(c2w && origins && viewdirs, "generate_rays: null pointer");
    hipLaunchKernelGGL(raygen_kernel, dim3(grid_for((int64_t)n_views * n_pix, 256)), dim3(256), 0, as_stream(stream), c2w,
                       n_views, width, height, focal, pix_idx, n_pix, origins, viewdirs);
    return launch_status("raygen_kernel");
}

extern "C" int mnf_gather_pixels(const uint8_t *images, const void *depths, int32_t depth_is_f16, const void *semantics, int32_t sem_is_u8,
                                 int64_t pixels_per_image, const int64_t *image_id, const int64_t *pix_idx, int64_t n_pix, float *rgb,
                                 float *dep, int64_t *sem, mnf_stream_t stream) {
    if (n_pix == 0) return MNF_OK;
    MNF_REQUIRE(images && depths && semantics && image_id && pix_idx && rgb && dep && sem && pixels_per_image > 0, "gather_pixels: bad arguments");
    hipLaunchKernelGGL(gather_pixels_kernel, dim3(grid_for(n_pix, 256)), dim3(256), 0, as_stream(stream), images, depths, depth_is_f16, semantics,
                       sem_is_u8, pixels_per_image, image_id, pix_idx, n_pix, rgb, dep, sem);
    return launch_status("gather_pixels_kernel");
}

extern "C" int mnf_planner_map(const uint8_t *binaries, int32_t n_members, int32_t res_x, int32_t res_y, int32_t res_z,
                               int32_t y_slice, int32_t *out_map, mnf_stream_t stream) {
    MNF_REQUIRE(binaries && out_map, "planner_map: null pointer");
    MNF_REQUIRE(n_members > 0 && res_x > 0 && res_y > 0 && res_z > 0 && y_slice >= 0 && y_slice < res_y, "planner_map: bad sizes");
    hipLaunchKernelGGL(planner_map_kernel, dim3(grid_for((int64_t)res_x * res_z, 256)), dim3(256), 0, as_stream(stream), binaries,
                       n_members, res_x, res_y, res_z, y_slice, out_map);
    return launch_status("planner_map_kernel");
}
