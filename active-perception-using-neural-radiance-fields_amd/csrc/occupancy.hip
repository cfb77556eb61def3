// Occupancy-grid refresh on the device: `OccGridEstimator._update` of the reference
// (perception/nerfacc/nerfacc/estimators/occ_grid.py:345-437) as a short chain of kernels with no host round trip.
//
//   sample   which cells are re-evaluated and where inside them (occ_grid.py:345-375, :395-401):
//            warm-up (step < warmup_steps): every cell with occs >= 0; afterwards N = cells/4 uniformly drawn cells (kept if
//            occs >= 0) plus the occupied cells (all of them, or N drawn with replacement when there are more than N);
//            point = aabb_min + ((cell_xyz + U[0,1)^3) / resolution) * (aabb_max - aabb_min)
//   (the caller evaluates the field's density at the points: `occ_eval_fn`, scripts/pipeline.py:376-378)
//   apply    occs[cell] = max(occs[cell] * ema_decay, occ) with the fork's NaN roll-back (occ_grid.py:403-434)
//   binarize thre = min(mean(occs[occs >= 0]), occ_thre); binaries = occs > thre (occ_grid.py:436-437), written both as the
//            [L,X,Y,Z] byte grid the planner and the reference's callers read and as the bit-packed grid the marchers
//            stage in LDS.
//
// Shapes are static (no boolean-mask compaction, hence no host sync): the sample list has a fixed capacity and unused
// slots carry cell index -1.  Random draws come from Philox4x32-10 (documented below) or, for tests against the
// reference's recorded draws, from caller-provided arrays.  Duplicate cells in one update resolve as "the last list
// element wins" (what torch's index assignment does on the CPU; on CUDA the reference's order is undefined).
//
// Build with -ffp-contract=off: the sample point is a chain of separately rounded fp32 operations in the reference.
#include <cmath>
#include <cstring>

#include "field.h"

namespace mnf {
namespace {

// ------------------------------------------------------------------ Philox4x32-10 (Salmon et al., SC'11)
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

// Draw stream of one update: counter = (element, 0, kind, step), key = (seed low, seed high);
// kind 0 = uniform half, 1 = occupied half, 2 = warm-up.  Word 0 picks the cell (uniform: floor(w0 * cells / 2^32);
// occupied: floor(w0 * n_occupied / 2^32)), words 1..3 are the in-cell offsets (w & 0xFFFFFF) * 2^-24 in [0, 1).
__device__ __forceinline__ float unit_float(uint32_t w) { return (float)(w & 0xFFFFFFu) * 5.9604644775390625e-08f; }

struct SampleArgs {
    const float *occs;         // [cells] of this level
    const uint32_t *bits;      // bit-packed binaries of this level (occupied half)
    const int32_t *prefix;     // inclusive prefix popcount per word (occupied half)
    int64_t cells;
    int32_t n_words;
    int32_t res[3];
    float aabb[6];
    int32_t warm;              // 1: warm-up list (every cell)
    int64_t n_quarter;         // N = cells / 4
    uint32_t seed_lo, seed_hi;
    int32_t step;
    const int64_t *idx_in;     // optional explicit cell list + offsets (tests): n_in entries, all used
    const float *jitter_in;
    int64_t n_in;
    int64_t *idx_out;          // [cap]: cell or -1
    float *pts_out;            // [cap,3]
    int32_t *owner;            // [cells], -1 filled: highest list position that names the cell
    int64_t cap;
};

__device__ __forceinline__ int64_t select_set_bit(const SampleArgs &a, int64_t j) {
    // smallest word w with prefix[w] > j, then the (j - prefix[w-1])-th set bit of that word (ascending cell order ==
    // torch.nonzero of the flattened grid)
    int lo = 0, hi = a.n_words - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)a.prefix[mid] > j) hi = mid; else lo = mid + 1;
    }
    uint32_t word = a.bits[lo];
    int r = (int)(j - ((int64_t)a.prefix[lo] - __popc(word)));
    for (; r > 0; --r) word &= word - 1u;
    return (int64_t)lo * 32 + (__ffs((int)word) - 1);
}

__global__ void __launch_bounds__(256) occ_sample_kernel(const SampleArgs a) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.cap; e += (int64_t)blockDim.x * gridDim.x) {
        int64_t c = -1;
        float u[3] = {0.f, 0.f, 0.f};
        if (a.idx_in) {
            if (e < a.n_in) { c = a.idx_in[e]; u[0] = a.jitter_in[3 * e]; u[1] = a.jitter_in[3 * e + 1]; u[2] = a.jitter_in[3 * e + 2]; }
        } else {
            const int kind = a.warm ? 2 : (e < a.n_quarter ? 0 : 1);
            const int64_t k = kind == 1 ? e - a.n_quarter : e;
            const U4 rnd = philox4x32_10({(uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint32_t)kind, (uint32_t)a.step}, a.seed_lo, a.seed_hi);
            if (kind == 2) {
                c = e < a.cells ? e : -1;
            } else if (kind == 0) {
                c = (int64_t)(((uint64_t)rnd.x * (uint64_t)a.cells) >> 32);
            } else {
                const int64_t n_occ = a.n_words ? a.prefix[a.n_words - 1] : 0;
                if (k < (n_occ < a.n_quarter ? n_occ : a.n_quarter))
                    c = select_set_bit(a, n_occ > a.n_quarter ? (int64_t)(((uint64_t)rnd.x * (uint64_t)n_occ) >> 32) : k);
            }
            if (c >= 0 && kind != 1 && !(a.occs[c] >= 0.0f)) c = -1;     // occ_grid.py:352, :364 (occupied cells are not filtered)
            u[0] = unit_float(rnd.y); u[1] = unit_float(rnd.z); u[2] = unit_float(rnd.w);
        }
        a.idx_out[e] = c;
        float p[3];
        if (c >= 0) {
            const int64_t yz = (int64_t)a.res[1] * a.res[2];
            const int coord[3] = {(int)(c / yz), (int)((c / a.res[2]) % a.res[1]), (int)(c % a.res[2])};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float x = ((float)coord[d] + u[d]) / (float)a.res[d];           // occ_grid.py:396-398
                p[d] = a.aabb[d] + x * (a.aabb[3 + d] - a.aabb[d]);                     // occ_grid.py:400-401
            }
            atomicMax(&a.owner[c], (int32_t)e);
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) p[d] = (a.aabb[d] + a.aabb[3 + d]) * 0.5f;     // unused slot: any point inside the box
        }
        a.pts_out[3 * e] = p[0]; a.pts_out[3 * e + 1] = p[1]; a.pts_out[3 * e + 2] = p[2];
    }
}

// inclusive prefix popcount over the words of one level: one workgroup, chunks of 1024 words with a running carry
__global__ void __launch_bounds__(1024) occ_prefix_kernel(const uint32_t *__restrict__ bits, int n_words, int32_t *__restrict__ prefix) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_words; base += 1024) {
        const int w = base + threadIdx.x;
        int v = w < n_words ? __popc(bits[w]) : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        int off = s_carry;
        for (int k = 0; k < wave; ++k) off += s_wave[k];
        if (w < n_words) prefix[w] = v + off;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = v + off;
        __syncthreads();
    }
}

// occ_grid.py:403-434: the list element that owns a cell writes max(old * decay, occ); a NaN candidate (or a NaN old
// value) leaves the cell as it was, which is what the roll-back from `occs_backup` amounts to.
__global__ void __launch_bounds__(256) occ_apply_kernel(float *__restrict__ occs, const int64_t *__restrict__ idx, const float *__restrict__ values,
                                                        float value_scale, int64_t n, float ema_decay, const int32_t *__restrict__ owner) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)blockDim.x * gridDim.x) {
        const int64_t c = idx[e];
        if (c < 0 || owner[c] != (int32_t)e) continue;
        const float old = occs[c];
        const float occ = values[e] * value_scale;
        if (occ != occ || old != old) continue;
        occs[c] = fmaxf(old * ema_decay, occ);
    }
}

constexpr int kReduceBlocks = 256;

// fixed-order partial sums of occs[occs >= 0] in double: block b owns a contiguous slice
__global__ void __launch_bounds__(256) occ_reduce_kernel(const float *__restrict__ occs, int64_t total, double *__restrict__ part_sum,
                                                         int64_t *__restrict__ part_cnt) {
    __shared__ double s_sum[256];
    __shared__ int64_t s_cnt[256];
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < total ? lo + per : total;
    double acc = 0.0;
    int64_t cnt = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = occs[i];
        if (v >= 0.0f) { acc += (double)v; ++cnt; }
    }
    s_sum[threadIdx.x] = acc; s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { s_sum[threadIdx.x] += s_sum[threadIdx.x + d]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_sum[blockIdx.x] = s_sum[0]; part_cnt[blockIdx.x] = s_cnt[0]; }
}

// threshold + both grid forms.  One wave covers 2048 consecutive cells of a level (64 words): lane l reads cell
// base + 64 k + l for k = 0..31 (coalesced) and the ballots of those rounds are re-assembled into 32-cell words.
__global__ void __launch_bounds__(256) occ_binarize_kernel(const float *__restrict__ occs, int64_t cells_per_lvl, int levels, float occ_thre,
                                                           const double *__restrict__ part_sum, const int64_t *__restrict__ part_cnt,
                                                           uint8_t *__restrict__ binaries, uint32_t *__restrict__ bits, int words_per_lvl,
                                                           float *__restrict__ thre_out) {
    __shared__ float s_thre;
    if (threadIdx.x == 0) {
        double s = 0.0;
        int64_t n = 0;
        for (int b = 0; b < kReduceBlocks; ++b) { s += part_sum[b]; n += part_cnt[b]; }
        float thre = n > 0 ? (float)(s / (double)n) : NAN;           // mean of an empty selection is NaN in torch: nothing is occupied
        if (thre == thre) thre = fminf(thre, occ_thre);              // torch.clamp(mean, max=occ_thre)
        s_thre = thre;
        if (blockIdx.x == 0 && blockIdx.y == 0 && thre_out) *thre_out = thre;
    }
    __syncthreads();
    const float thre = s_thre;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lvl = blockIdx.y;
    const float *src = occs + (int64_t)lvl * cells_per_lvl;
    uint8_t *dst = binaries + (int64_t)lvl * cells_per_lvl;
    uint32_t *wdst = bits + (int64_t)lvl * words_per_lvl;
    const int64_t n_groups = (cells_per_lvl + 2047) / 2048;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t base = grp * 2048;
        // cell base + 32 w + b sits in round k = (32 w + b) / 64 at lane (32 w + b) % 64: word w = half (w & 1) of round w >> 1
        uint32_t my_word = 0;
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
            const int64_t c = base + 64 * k + lane;
            const bool on = c < cells_per_lvl && src[c] > thre;
            if (c < cells_per_lvl) dst[c] = on ? 1 : 0;
            const unsigned long long m = __ballot(on);
            if ((lane >> 1) == k) my_word = (lane & 1) ? (uint32_t)(m >> 32) : (uint32_t)m;
        }
        const int64_t w = base / 32 + lane;
        if (w < words_per_lvl) wdst[w] = my_word;
    }
}

__global__ void __launch_bounds__(256) pack_bits_kernel(const uint8_t *__restrict__ binaries, int64_t cells_per_lvl, int words_per_lvl,
                                                        uint32_t *__restrict__ bits) {
    const int lvl = blockIdx.y;
    const uint8_t *src = binaries + (int64_t)lvl * cells_per_lvl;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words_per_lvl; w += (int64_t)blockDim.x * gridDim.x) {
        uint32_t v = 0;
        for (int b = 0; b < 32; ++b) {
            const int64_t c = w * 32 + b;
            if (c < cells_per_lvl && src[c]) v |= 1u << b;
        }
        bits[(int64_t)lvl * words_per_lvl + w] = v;
    }
}

__global__ void __launch_bounds__(256) fill_i32_kernel(int32_t *__restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) p[i] = v;
}

struct OccWs {
    int32_t *owner, *prefix;
    double *part_sum;
    int64_t *part_cnt;
    int64_t *idx;      // fused path
    float *pts, *vals;
    int64_t bytes;
};

inline int64_t list_capacity(int64_t cells, int32_t step, int32_t warmup_steps) { return step < warmup_steps ? cells : 2 * (cells / 4); }

OccWs carve_occ(char *base, int64_t cells, bool fused) {
    OccWs w;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += (b + 255) & ~(size_t)255; return p; };
    w.owner = (int32_t *)take((size_t)cells * 4);
    w.prefix = (int32_t *)take((size_t)((cells + 31) / 32) * 4);
    w.part_sum = (double *)take(kReduceBlocks * 8);
    w.part_cnt = (int64_t *)take(kReduceBlocks * 8);
    w.idx = nullptr; w.pts = nullptr; w.vals = nullptr;
    if (fused) {
        w.idx = (int64_t *)take((size_t)cells * 8);
        w.pts = (float *)take((size_t)cells * 12);
        w.vals = (float *)take((size_t)cells * 4);
    }
    w.bytes = (int64_t)off;
    return w;
}

inline int blocks_for(int64_t n, int threads, int cap = 4096) {
    const int64_t b = (n + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int sample_cells(const float *occs, const uint32_t *bitgrid, int32_t rx, int32_t ry, int32_t rz, const float *aabb_host, int32_t step,
                 int32_t warmup_steps, uint64_t seed, const int64_t *idx_in, const float *jitter_in, int64_t n_in, int64_t *idx_out,
                 float *pts_out, int64_t cap, const OccWs &w, hipStream_t s) {
    const int64_t cells = (int64_t)rx * ry * rz;
    SampleArgs a;
    a.occs = occs; a.bits = bitgrid; a.prefix = w.prefix; a.cells = cells; a.n_words = (int)((cells + 31) / 32);
    a.res[0] = rx; a.res[1] = ry; a.res[2] = rz;
    std::memcpy(a.aabb, aabb_host, sizeof(a.aabb));
    a.warm = step < warmup_steps ? 1 : 0;
    a.n_quarter = cells / 4;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.step = step;
    a.idx_in = idx_in; a.jitter_in = jitter_in; a.n_in = n_in;
    a.idx_out = idx_out; a.pts_out = pts_out; a.owner = w.owner; a.cap = cap;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s, w.owner, cells, -1);
    if (!idx_in && !a.warm) hipLaunchKernelGGL(occ_prefix_kernel, dim3(1), dim3(1024), 0, s, bitgrid, a.n_words, w.prefix);
    hipLaunchKernelGGL(occ_sample_kernel, dim3(blocks_for(cap, 256)), dim3(256), 0, s, a);
    return launch_status("occ_sample_kernel");
}

}  // namespace
}  // namespace mnf

using namespace mnf;

extern "C" int64_t mnf_occ_workspace_bytes(int64_t cells_per_level, int32_t fused) {
    if (cells_per_level <= 0) return -1;
    return carve_occ(nullptr, cells_per_level, fused != 0).bytes;
}

extern "C" int64_t mnf_occ_list_capacity(int64_t cells_per_level, int32_t step, int32_t warmup_steps) {
    return cells_per_level <= 0 ? -1 : list_capacity(cells_per_level, step, warmup_steps);
}

extern "C" int mnf_pack_bitgrid(const uint8_t *binaries, int64_t cells_per_level, int32_t levels, uint32_t *bitgrid, mnf_stream_t stream) {
    MNF_REQUIRE(binaries && bitgrid && cells_per_level > 0 && levels > 0, "pack_bitgrid: bad arguments");
    const int words = (int)((cells_per_level + 31) / 32);
    hipLaunchKernelGGL(pack_bits_kernel, dim3(blocks_for(words, 256), levels), dim3(256), 0, as_stream(stream), binaries, cells_per_level, words, bitgrid);
    return launch_status("pack_bits_kernel");
}

extern "C" int mnf_occ_sample_cells(const float *occs, const uint32_t *bitgrid, int32_t res_x, int32_t res_y, int32_t res_z,
                                    const float *aabb_host, int32_t step, int32_t warmup_steps, uint64_t seed,
                                    const int64_t *indices_in, const float *jitter_in, int64_t n_in,
                                    int64_t *cell_idx, float *points, int64_t capacity, void *workspace, int64_t workspace_bytes,
                                    mnf_stream_t stream) {
    MNF_REQUIRE(occs && aabb_host && cell_idx && points && workspace, "occ_sample_cells: null pointer");
    MNF_REQUIRE(res_x > 0 && res_y > 0 && res_z > 0, "occ_sample_cells: bad resolution");
    const int64_t cells = (int64_t)res_x * res_y * res_z;
    MNF_REQUIRE(cells < ((int64_t)1 << 31), "occ_sample_cells: grid too large");
    const OccWs w = carve_occ((char *)workspace, cells, false);
    if (workspace_bytes < w.bytes) { set_error("occ_sample_cells: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)w.bytes); return MNF_ERR_WORKSPACE; }
    if (indices_in) {
        MNF_REQUIRE(jitter_in && n_in >= 0 && n_in <= capacity, "occ_sample_cells: explicit list needs offsets and must fit the capacity");
    } else {
        MNF_REQUIRE(capacity >= list_capacity(cells, step, warmup_steps), "occ_sample_cells: capacity below mnf_occ_list_capacity()");
        MNF_REQUIRE(step < warmup_steps || bitgrid, "occ_sample_cells: the occupied half needs the bit-packed grid");
    }
    if (capacity == 0) return MNF_OK;
    return sample_cells(occs, bitgrid, res_x, res_y, res_z, aabb_host, step, warmup_steps, seed, indices_in, jitter_in, n_in, cell_idx, points,
                        capacity, w, as_stream(stream));
}

extern "C" int mnf_occ_apply(float *occs, const int64_t *cell_idx, const float *values, float value_scale, int64_t n, int64_t cells_per_level,
                             float ema_decay, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(occs && cell_idx && values && workspace && n > 0 && cells_per_level > 0, "occ_apply: bad arguments");
    const OccWs w = carve_occ((char *)workspace, cells_per_level, false);
    if (workspace_bytes < w.bytes) { set_error("occ_apply: workspace too small"); return MNF_ERR_WORKSPACE; }
    hipLaunchKernelGGL(occ_apply_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), occs, cell_idx, values, value_scale, n, ema_decay,
                       w.owner);
    return launch_status("occ_apply_kernel");
}

extern "C" int mnf_occ_binarize(const float *occs, int64_t cells_per_level, int32_t levels, float occ_thre, uint8_t *binaries,
                                uint32_t *bitgrid, float *threshold_out, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    MNF_REQUIRE(occs && binaries && bitgrid && workspace && cells_per_level > 0 && levels > 0, "occ_binarize: bad arguments");
    const OccWs w = carve_occ((char *)workspace, cells_per_level, false);
    if (workspace_bytes < w.bytes) { set_error("occ_binarize: workspace too small"); return MNF_ERR_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(occ_reduce_kernel, dim3(kReduceBlocks), dim3(256), 0, s, occs, cells_per_level * levels, w.part_sum, w.part_cnt);
    const int words = (int)((cells_per_level + 31) / 32);
    hipLaunchKernelGGL(occ_binarize_kernel, dim3(blocks_for((cells_per_level + 2047) / 2048, 4, 1024), levels), dim3(256), 0, s, occs,
                       cells_per_level, levels, occ_thre, w.part_sum, w.part_cnt, binaries, bitgrid, words, threshold_out);
    return launch_status("occ_binarize_kernel");
}

extern "C" int mnf_update_occupancy(mnf_field_t f, float *occs, uint8_t *binaries, uint32_t *bitgrid, int32_t res_x, int32_t res_y,
                                    int32_t res_z, const float *aabb_host, int32_t step, int32_t warmup_steps, float occ_thre,
                                    float ema_decay, float density_scale, uint64_t seed, void *workspace, int64_t workspace_bytes,
                                    mnf_stream_t stream) {
    MNF_REQUIRE(f && f->params_loaded, "update_occupancy: field parameters not loaded");
    MNF_REQUIRE(occs && binaries && bitgrid && aabb_host && workspace, "update_occupancy: null pointer");
    MNF_REQUIRE(res_x > 0 && res_y > 0 && res_z > 0, "update_occupancy: bad resolution");
    const int64_t cells = (int64_t)res_x * res_y * res_z;
    MNF_REQUIRE(cells < ((int64_t)1 << 31), "update_occupancy: grid too large");
    const OccWs w = carve_occ((char *)workspace, cells, true);
    if (workspace_bytes < w.bytes) { set_error("update_occupancy: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)w.bytes); return MNF_ERR_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    const int64_t cap = list_capacity(cells, step, warmup_steps);
    int rc = sample_cells(occs, bitgrid, res_x, res_y, res_z, aabb_host, step, warmup_steps, seed, nullptr, nullptr, 0, w.idx, w.pts, cap, w, s);
    if (rc) return rc;
    FieldIO io = {};
    io.mode = 0; io.positions = w.pts; io.n = cap; io.density = w.vals;
    rc = launch_field(f, io, true, s);                              // occ_eval_fn: query_density(x) * render_step_size (pipeline.py:376-378)
    if (rc) return rc;
    hipLaunchKernelGGL(occ_apply_kernel, dim3(blocks_for(cap, 256)), dim3(256), 0, s, occs, w.idx, w.vals, density_scale, cap, ema_decay, w.owner);
    rc = launch_status("occ_apply_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(occ_reduce_kernel, dim3(kReduceBlocks), dim3(256), 0, s, occs, cells, w.part_sum, w.part_cnt);
    const int words = (int)((cells + 31) / 32);
    hipLaunchKernelGGL(occ_binarize_kernel, dim3(blocks_for((cells + 2047) / 2048, 4, 1024), 1), dim3(256), 0, s, occs, cells, 1, occ_thre,
                       w.part_sum, w.part_cnt, binaries, bitgrid, words, (float *)nullptr);
    return launch_status("occ_binarize_kernel");
}
