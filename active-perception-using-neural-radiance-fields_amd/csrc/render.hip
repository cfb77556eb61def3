// Fused test-mode renderers and the predictive-information scorer.
//
//   mnf_render_test  <- perception/models/utils.py:555-779 (render_image_with_occgrid_test) and
//                       utils.py:782-1032 (render_probablistic_image_with_occgrid_test)
//   mnf_score_views  <- scripts/pipeline.py:727-781
//
// The reference runs <=256 host-synchronised rounds per call (utils.py:666-757): count alive rays
// (`.item()`), march <= n_samples new samples per alive ray, query the field, composite with the
// carried transmittance, retire rays.  Here a round is four dependent launches with every
// data-dependent quantity (alive counts, per-view n_samples, number of sample columns) kept on the
// device, and many independent reference calls ("views" of rays_per_view rays) advance through the
// same launches, each with its own round schedule.
//
// Build with -ffp-contract=off (march_dev.h).
#include <chrono>
#include <vector>

#include "field.h"
#include "march_dev.h"

namespace mnf {

constexpr int kRayThreads = 256;
#ifndef MNF_MARCH_THREADS
#define MNF_MARCH_THREADS 1024
#endif
constexpr int kMarchThreads = MNF_MARCH_THREADS;   // one bit-packed occupancy grid in LDS shared by 16 waves (A/B builds: 256 / 512)
constexpr int kMaxGridWords = 16384;           // 64 KB of LDS = 524 288 cells (largest reference grid: 396 900)

struct RenderWs {
    // per ray
    float *near_plane, *t_min, *t_max;
    int32_t *col0, *cnt;
    uint8_t *alive, *hit;
    // per view
    int32_t *alive_count, *n_samples, *iter_samples, *active;   // alive_count, iter_samples: two copies [2][views], read at the round's parity, written at the other
                                                                // (round k's marcher resets the copy that round k's compositing counts into)
    // global
    int32_t *n_cols, *any_active, *overflow;   // n_cols[2], any_active[2] by round parity; overflow: set if a round ever asked for more columns than col_cap (never, by construction)
    uint32_t *bitgrid;   // bit-packed copy of the occupancy grid (built once per call)
    uint32_t *tickets;   // eight tile counters of the round's field launch (FieldIO::tickets), zeroed by the round's marcher
    // per column
    int32_t *col_ray;
    float *col_ts, *col_te;
    int32_t *tile_hdr;   // per 64-column tile: stride (= the view's budget this round) | view << 8
    void *enc;           // hash features of the round's columns in MLP fragment order (two-launch field path)
    int64_t col_cap;
};

static bool split_field() {
    static const bool on = diag_env("MNF_FIELD_SPLIT") != nullptr;
    return on;
}

static hipEvent_t *log_events() {
    static hipEvent_t ev[4];
    static bool made = false;
    if (!made) { for (auto &e : ev) (void)hipEventCreate(&e); made = true; }
    return ev;
}
static bool round_log() { static const bool on = diag_env("MNF_ROUND_LOG") != nullptr; return on; }

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// A view's rays are marched by its own ceil(rays_per_view / kMarchThreads) workgroups (the last one partly idle when the
// size is not a multiple): every marching ray of a workgroup then has the same per-ray budget n_samples[view], which is what
// bounds the columns a round can allocate (below).
static inline int64_t march_blocks_per_view(int32_t rays_per_view) { return (rays_per_view + kMarchThreads - 1) / kMarchThreads; }

static int64_t carve(RenderWs *ws, char *base, int64_t n_rays, int32_t rays_per_view) {
    const int64_t n_views = n_rays / rays_per_view;
    // columns of one round: a tile of 64 holds floor(64/stride) rays of `stride` columns, so it is at least half
    // used; n_alive*stride <= max(R, 4*n_alive) <= 4R (utils.py:670) -> <= 8R, plus one partial tile per workgroup
    // (a march workgroup never mixes views: march_blocks_per_view() workgroups per view, so `stride` is the view's budget)
    int64_t col_cap = 8 * n_rays + 64 * (n_views * march_blocks_per_view(rays_per_view) + 2);
    if (const char *e = diag_env("MNF_MIN_SAMPLES")) col_cap = (int64_t)(2 * atoi(e) > 8 ? 2 * atoi(e) : 8) * n_rays + 64 * (n_views * march_blocks_per_view(rays_per_view) + 2);   // diagnostic schedule
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    char *p;
    p = take(n_rays * 4); if (ws) ws->near_plane = (float *)p;
    p = take(n_rays * 4); if (ws) ws->t_min = (float *)p;
    p = take(n_rays * 4); if (ws) ws->t_max = (float *)p;
    p = take(n_rays * 4); if (ws) ws->col0 = (int32_t *)p;
    p = take(n_rays * 4); if (ws) ws->cnt = (int32_t *)p;
    p = take(n_rays); if (ws) ws->alive = (uint8_t *)p;
    p = take(n_rays); if (ws) ws->hit = (uint8_t *)p;
    p = take(2 * n_views * 4); if (ws) ws->alive_count = (int32_t *)p;
    p = take(n_views * 4); if (ws) ws->n_samples = (int32_t *)p;
    p = take(2 * n_views * 4); if (ws) ws->iter_samples = (int32_t *)p;
    p = take(n_views * 4); if (ws) ws->active = (int32_t *)p;
    p = take(256); if (ws) { ws->n_cols = (int32_t *)p; ws->any_active = (int32_t *)p + 2; ws->overflow = (int32_t *)p + 4; }   // n_cols[2] | any_active[2] | overflow
    p = take(512); if (ws) ws->tickets = (uint32_t *)p;
    p = take(kMaxGridWords * 4); if (ws) ws->bitgrid = (uint32_t *)p;
    p = take(col_cap * 4); if (ws) ws->col_ray = (int32_t *)p;
    p = take(col_cap * 4); if (ws) ws->col_ts = (float *)p;
    p = take(col_cap * 4); if (ws) ws->col_te = (float *)p;
    p = take(col_cap / 64 * 4); if (ws) ws->tile_hdr = (int32_t *)p;
    if (split_field()) { p = take((col_cap / 64 + 1) * 8192); if (ws) ws->enc = p; }   // two-launch path: 128 B per column
    else if (ws) ws->enc = nullptr;
    if (ws) ws->col_cap = col_cap;
    return (int64_t)off;
}

struct RenderOut {
    float *rgb, *acc, *depth, *sem, *rgb_var, *depth_var;
    int64_t *total_samples;
};

// ------------------------------------------------------------------ kernels
// utils.py:640-664: zero the accumulators, all rays alive, one ray/AABB test per call
__global__ void __launch_bounds__(kRayThreads) init_kernel(int64_t n_rays, int32_t rays_per_view, int32_t C,
                                                           const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                           float a0, float a1, float a2, float a3, float a4, float a5,
                                                           float near_plane, RenderWs ws, RenderOut out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) { out.total_samples[0] = 0; out.total_samples[1] = 0; *ws.overflow = 0; ws.n_cols[0] = 0; ws.n_cols[1] = 0; ws.any_active[0] = 0; ws.any_active[1] = 0; }
    if (r < n_rays / rays_per_view) { ws.alive_count[r] = rays_per_view; ws.iter_samples[r] = 0; }      // round 0 reads the copies of parity 0
    if (r >= n_rays) return;
    const float ab[6] = {a0, a1, a2, a3, a4, a5};
    const F3 o = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const F3 inv = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
    float t0, t1;
    const bool hit = ray_aabb(o, inv, -INFINITY, INFINITY, ab, t0, t1);   // utils.py:658 (default near/far)
    ws.t_min[r] = hit ? t0 : INFINITY;
    ws.t_max[r] = hit ? t1 : INFINITY;
    ws.hit[r] = hit;
    ws.alive[r] = 1;
    ws.near_plane[r] = near_plane;
    out.rgb[3 * r] = 0.f; out.rgb[3 * r + 1] = 0.f; out.rgb[3 * r + 2] = 0.f;
    out.acc[r] = 0.f; out.depth[r] = 0.f;
    if (out.rgb_var) { out.rgb_var[3 * r] = 0.f; out.rgb_var[3 * r + 1] = 0.f; out.rgb_var[3 * r + 2] = 0.f; }
    if (out.depth_var) out.depth_var[r] = 0.f;
}

struct RoundSink {
    float *col_ts, *col_te;
    int32_t col0;
    __device__ __forceinline__ void sample(float t_last, float t_next, bool, int32_t k) {
        // unsigned 32-bit byte offset from the (wave-uniform) array base: a scalar-base store, no 64-bit address math per sample
        const uint32_t off = (uint32_t)(col0 + k) * 4u;
        *reinterpret_cast<float *>(reinterpret_cast<char *>(col_ts) + off) = t_last;
        *reinterpret_cast<float *>(reinterpret_cast<char *>(col_te) + off) = t_next;
    }
};

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// bit-pack the [X,Y,Z] byte grid once per call: word w holds cells 32w .. 32w+31
// (per level: n_words = levels * words_per_level, level l's cells start at binaries + l * cells and at word l * words_per_level)
__global__ void __launch_bounds__(256) pack_grid_kernel(const uint8_t *__restrict__ binaries, int64_t cells, uint32_t *__restrict__ bits, int n_words, int words_per_level) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const int lvl = w / words_per_level, wl = w - lvl * words_per_level;
    uint32_t v = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t c = (int64_t)wl * 32 + b;
        if (c < cells && binaries[(int64_t)lvl * cells + c]) v |= 1u << b;
    }
    bits[w] = v;
}

// utils.py:674-696: one traversal of <= n_samples steps per alive ray (over-allocated mode of grid.cu:364-404).
// The occupancy grid is read from a bit-packed copy staged in LDS (<= 64 KB) once per workgroup.
#ifndef MNF_MARCH_WAVES
#define MNF_MARCH_WAVES 8      /* waves per SIMD the single-level LDS variant is compiled for: 8 = two 16-wave workgroups per compute unit (42 VGPRs, 69 KB of LDS each) */
#endif
template <bool LDS_GRID, bool MULTI>
__global__ void __launch_bounds__(kMarchThreads, (LDS_GRID && !MULTI) ? MNF_MARCH_WAVES : 4) round_march_kernel(int64_t n_rays, int32_t rays_per_view,
                                                                    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                    const uint8_t *__restrict__ binaries, I3 res, int n_words,
                                                                    float a0, float a1, float a2, float a3, float a4, float a5,
                                                                    float far_plane, float step_size, float cone_angle, RenderWs ws,
                                                                    const int32_t *__restrict__ view_order, int32_t blocks_per_view, LevelBoxes boxes,
                                                                    int32_t n_views, int32_t parity, int32_t max_samples, int32_t min_samples) {
    __shared__ int s_wave_tot[kMarchThreads / 64];
    __shared__ int s_base;
    __shared__ int s_list[kMarchThreads];       // the workgroup's marching rays (index inside the view), in thread order
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[LDS_GRID ? kMaxGridWords : 1];
    // workgroup -> (view, slice of the view): a workgroup never mixes views, so all its marching rays share one budget.
    // thread -> ray inside the view: identity, or the caller's order (neighbouring rays into the same tile)
    const int v = (int)(blockIdx.x / blocks_per_view);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The round's budget of this view (utils.py:666-672: n_alive -> n_samples, until max_samples are spent), worked out by every workgroup of the view from the same two
    // words — what a one-workgroup launch in front of every marcher did until round 5 (5 us of kernel, more of queue).  The per-view words the round changes have two copies:
    // this round reads the copy of its parity; the view's first workgroup writes the other (samples spent so far; zero survivors, which this round's compositing counts up).
    const int n_alive = ws.alive_count[parity * n_views + v], spent = ws.iter_samples[parity * n_views + v];
    const bool act = spent < max_samples && n_alive > 0;
    const int stride = act ? max(min(rays_per_view / n_alive, 64), min_samples) : 0;
    if (threadIdx.x == 0 && blockIdx.x == (unsigned)v * blocks_per_view) {
        if (act) { ws.n_samples[v] = stride; ws.any_active[parity] = 1; }
        ws.active[v] = act ? 1 : 0;
        ws.iter_samples[(parity ^ 1) * n_views + v] = spent + stride;
        ws.alive_count[(parity ^ 1) * n_views + v] = 0;        // rays of inactive views are never marched again
    }
    if (blockIdx.x == 0) {      // the next round's column counter and flag, this round's tile tickets
        if (threadIdx.x < 8) ws.tickets[16 * threadIdx.x] = 0u;
        if (threadIdx.x == 8) ws.n_cols[parity ^ 1] = 0;
        if (threadIdx.x == 9) ws.any_active[parity ^ 1] = 0;
    }
    if (!act) return;                           // uniform: the view has no round to run
    int rv = 0;
    bool go = false;
    {
        const int in_view = (int)(blockIdx.x - (int64_t)v * blocks_per_view) * kMarchThreads + (int)threadIdx.x;
        if (in_view < rays_per_view) {
            rv = view_order ? view_order[in_view] : in_view;
            go = ws.alive[(int64_t)v * rays_per_view + rv] != 0;
        }
    }
    // The workgroup's marching rays are packed to the front of the workgroup (rank k among them, in thread order -> thread k): late in a render one ray in ten is
    // still alive, and a wave with six busy lanes costs what a full one does.  Waves behind the last marching ray leave.
    const int incl = wave_inclusive_scan(go ? 1 : 0, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    int s_total = 0, before = 0;      // every thread adds the sixteen wave totals up itself (broadcast reads)
#pragma unroll
    for (int w = 0; w < kMarchThreads / 64; ++w) {
        const int t = s_wave_tot[w];
        before += w < wave ? t : 0;
        s_total += t;
    }
    if (s_total == 0) return;   // uniform: no ray of this workgroup marches this round — and nothing of the occupancy grid was touched
    if (go) s_list[incl - 1 + before] = rv;
    // Column allocation.  All marching rays of the workgroup get `stride` = the view's budget of this round; a 64-column tile holds cap = 64/stride rays, so no
    // ray straddles a tile and the field kernel can composite a ray inside one wave.  Ray with rank k among the
    // workgroup's marching rays -> tile k / cap, columns (k % cap) * stride ...
    if (LDS_GRID) {
        // The occupancy bits go from L2 straight into LDS (global_load_lds, 16 bytes per lane, no registers in between) while thread 0 reserves the columns.  Through
        // registers the 64 KB were 64 VGPRs per thread, requested by workgroups that march nothing as well.
        constexpr int kStageVec = kMaxGridWords / 4 / kMarchThreads;
#pragma unroll
        for (int j = 0; j < kStageVec; ++j) {
            const int q = (int)threadIdx.x + j * kMarchThreads;
            if (4 * q + 3 < n_words)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4 *>(ws.bitgrid) + q,
                                                 (__attribute__((address_space(3))) uint32_t *)(s_bits) + 4 * (j * kMarchThreads + wave * 64), 16, 0, 0);
        }
        for (int i = (n_words & ~3) + (int)threadIdx.x; i < n_words; i += kMarchThreads) s_bits[i] = ws.bitgrid[i];
    }
    if (threadIdx.x == 0) {
        const int cap = 64 / stride;
        const int need = ((s_total + cap - 1) / cap) * 64;
        int base = atomicAdd(ws.n_cols + parity, need);
        if ((int64_t)base + need > ws.col_cap) {   // cannot happen with one view per workgroup (carve()); never write past the workspace
            atomicExch(ws.overflow, 1);
            base = -1;
        }
        s_base = base;
    }
    // The LDS-DMA loads above are global loads whose data lands in LDS: they count on vmcnt, and nothing in a workgroup barrier or an LDS fence has to wait for
    // them (gfx950's s_barrier does not drain counters by itself).  ROCm 7.2's hipcc happens to emit s_waitcnt vmcnt(0) here; the wait is written out so that every
    // wave's share of the grid is in LDS before any wave passes the barrier whatever the compiler does.
    if (LDS_GRID) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s_base < 0) return;     // uniform: the workspace guard fired
    // Rank k among the workgroup's marching rays -> lane.  A wave pays for the UNION of its lanes' paths (empty-space stepping, sampling, cell steps: different trip
    // counts in every lane; measured on the sampler: 0.179 / 0.188 / 0.222 / 0.249 ms with 1 / 2 / 4 / 8 rays per wave, profiles/r06_sampler_rpw.txt), and late in a
    // render a workgroup marches a few dozen rays: they are spread over the workgroup's sixteen waves, as few to a wave as that allows, instead of filling one
    // wave completely (lanes >= rpw idle).  Which lane marches a ray changes nothing for the ray.
#ifndef MNF_MARCH_RPW
#define MNF_MARCH_RPW 0              /* 0 = by the number of marching rays; 64 = full waves (rounds 1-5; A/B builds) */
#endif
    // Only in launches the chip holds at once (<= 512 workgroups = two per compute unit: the scorer's jobs, small pose lists): there the marcher is latency-bound and
    // the extra waves are free; in the large launches of full-resolution renders it is throughput-bound and sixteen waves for sixteen rays cost issue slots
    // (800 x 800 x 4: 63.7 -> 64.0 ms per step with the spreading everywhere; scoring shard of 32 views 21.3 -> 20.7 ms, profiles/r06_rpw_ab.txt).
    int rpw = MNF_MARCH_RPW ? MNF_MARCH_RPW : (gridDim.x <= 512 ? 1 : 64);       // the smallest power of two with sixteen waves x rpw >= s_total
    while (!MNF_MARCH_RPW && rpw * (kMarchThreads / 64) < s_total) rpw <<= 1;
    if (lane >= rpw) return;
    const int k = wave * rpw + lane;
    if (k >= s_total) return;
    const int64_t r = (int64_t)v * rays_per_view + s_list[k];
    const int ns = stride, cap = 64 / stride;
    float ro[3], rd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { ro[d] = rays_o[3 * r + d]; rd[d] = rays_d[3 * r + d]; }
    const float ray_near = ws.near_plane[r], ray_tmin = ws.t_min[r], ray_tmax = ws.t_max[r];
    const bool ray_hit = ws.hit[r];
    const int tile_local = k / cap, slot = k - tile_local * cap;
    const int col0 = s_base + tile_local * 64 + slot * stride;
    if (slot == 0) {   // the first ray of a tile also describes the tile and blanks the columns no ray owns
        const int nslots = min(cap, s_total - tile_local * cap);
        ws.tile_hdr[(s_base >> 6) + tile_local] = stride | (v << 8);   // a workgroup marches one view: the tile's view and budget are uniform
        for (int k = nslots * stride; k < 64; ++k) ws.col_ray[s_base + tile_local * 64 + k] = -1;
    }

    const float ab[6] = {a0, a1, a2, a3, a4, a5};
    const F3 org = {ro[0], ro[1], ro[2]};
    const F3 dir = {rd[0], rd[1], rd[2]};
    const F3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    const float near_plane = ray_near;
    MarchState st = {near_plane, false, 0};
    RoundSink sink = {ws.col_ts, ws.col_te, col0};
    if (MULTI) {   // several occupancy levels: march_dev.h march_levels
        const int64_t cells = (int64_t)res.x * res.y * res.z;
        if (LDS_GRID) {
            const uint32_t *bits = s_bits; const int wpl = boxes.words_per_level;
            march_levels(org, dir, inv, near_plane, far_plane, boxes, res, [=](int level) { return BitGrid{bits + level * wpl}; }, step_size, cone_angle, ns, st, sink);
        } else {
            march_levels(org, dir, inv, near_plane, far_plane, boxes, res, [=](int level) { return ByteGrid{binaries + level * cells}; }, step_size, cone_angle, ns, st, sink);
        }
    } else if (ray_hit) {   // single grid level: the only interval is [t_min, t_max] (grid.cu:125-151 with n_grids == 1)
        const float this_tmin = fmaxf(ray_tmin, near_plane);
        const float this_tmax = fminf(ray_tmax, far_plane);
        if (this_tmin < this_tmax) {
            if (LDS_GRID) march_segment(org, dir, inv, this_tmin, this_tmax, ab, res, BitGrid{s_bits}, step_size, cone_angle, ns, st, sink);
            else march_segment(org, dir, inv, this_tmin, this_tmax, ab, res, ByteGrid{binaries}, step_size, cone_angle, ns, st, sink);
        }
    }
    // per-column ray id: the run of this ray's valid samples, -1 for the rest of its slot
    for (int k = 0; k < stride; ++k) ws.col_ray[col0 + k] = k < st.n_samples ? (int32_t)r : -1;
    if (st.n_samples == 0) ws.alive[r] = 0;   // left the grid: retired here, the compositing pass never sees it (utils.py:751-756)
    ws.col0[r] = col0;
    ws.cnt[r] = st.n_samples;
    ws.near_plane[r] = st.t_last;   // utils.py:749 near_planes = termination_planes
}

// utils.py:759-760
__global__ void __launch_bounds__(kRayThreads) finalize_kernel(int64_t n_rays, float b0, float b1, float b2, RenderOut out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float op = out.acc[r];
    out.rgb[3 * r] = out.rgb[3 * r] + b0 * (1.0f - op);
    out.rgb[3 * r + 1] = out.rgb[3 * r + 1] + b1 * (1.0f - op);
    out.rgb[3 * r + 2] = out.rgb[3 * r + 2] + b2 * (1.0f - op);
    out.depth[r] = out.depth[r] / fmaxf(op, 1.1920928955078125e-07f);   // torch.finfo(float32).eps
}

// ------------------------------------------------------------------ scorer (pipeline.py:727-781), one block per view
__device__ __forceinline__ double block_sum(double v, double *s_buf) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_buf[w];
    return t;
}

constexpr int kScoreMaxM = 4;             // ensemble sizes the array-free form of the semantic term is unrolled for (the reference's ensemble has two members)
constexpr int kScoreThreads = 1024;      // 16 waves per view: the 4096 pixels of a view are ~220 double-precision exp / log each; four waves per view left three of four issue slots to their latencies (1.07 ms per 256 views)
template <bool SMALL_M>      // SMALL_M: M <= kScoreMaxM (checked by the host)
__global__ void __launch_bounds__(kScoreThreads) score_kernel(const float *__restrict__ rgb_var, const float *__restrict__ depth_var,
                                                    const float *__restrict__ acc, const float *__restrict__ sem,
                                                    int M, int V, int P, int C, double *__restrict__ terms) {
    __shared__ double s_buf[kScoreThreads / 64];
    const int v = blockIdx.x;
    const double k2pie = 2.0 * 3.14159265358979323846 * 2.71828182845904523536;
    double s_rgb = 0.0, s_dep = 0.0, s_sem = 0.0, s_occ = 0.0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        // rgb / depth: entropy of the summed variance minus mean member entropy (pipeline.py:727-746)
        for (int ch = 0; ch < 3; ++ch) {
            double sum_var = 0.0, mean_ce = 0.0;
            for (int m = 0; m < M; ++m) {
                const double x = rgb_var[(((int64_t)m * V + v) * P + p) * 3 + ch];
                sum_var += x;
                mean_ce += log(k2pie * x + 1e-4) / 2.0;
            }
            s_rgb += log(k2pie * (sum_var / 2.0) + 1e-4) / 2.0 - mean_ce / M;
        }
        {
            double sum_var = 0.0, mean_ce = 0.0;
            for (int m = 0; m < M; ++m) {
                const double x = depth_var[((int64_t)m * V + v) * P + p];
                sum_var += x;
                mean_ce += log(k2pie * x + 1e-4) / 2.0;
            }
            s_dep += log(k2pie * (sum_var / 2.0) + 1e-4) / 2.0 - mean_ce / M;
        }
        // semantics (pipeline.py:748-760)
        if constexpr (SMALL_M) {
            // class by class with the members' max / denominator worked out first: the same sums in the same order as the general form below, without its
            // per-class array (dynamically indexed -> 272 bytes of scratch per lane, which was most of this kernel's time)
            double mx[kScoreMaxM], den[kScoreMaxM], ce[kScoreMaxM];
#pragma unroll
            for (int m = 0; m < kScoreMaxM; ++m) {
                mx[m] = -1e300; den[m] = 0.0; ce[m] = 0.0;
                if (m < M) {
                    const float *lg = sem + (((int64_t)m * V + v) * P + p) * C;
                    for (int k = 0; k < C; ++k) mx[m] = fmax(mx[m], (double)lg[k]);
                    for (int k = 0; k < C; ++k) den[m] += exp((double)lg[k] - mx[m]);
                }
            }
            double ent = 0.0;
            for (int k = 0; k < C; ++k) {
                double pe = 0.0;
#pragma unroll
                for (int m = 0; m < kScoreMaxM; ++m)
                    if (m < M) {
                        const double pk = exp((double)sem[(((int64_t)m * V + v) * P + p) * C + k] - mx[m]) / den[m];
                        pe += pk;
                        ce[m] -= (pk + 1e-4) * log(pk + 1e-4);
                    }
                pe /= M;
                ent -= (pe + 1e-4) * log(pe + 1e-4);
            }
            double mean_ce = 0.0;
#pragma unroll
            for (int m = 0; m < kScoreMaxM; ++m) if (m < M) mean_ce += ce[m];
            s_sem += ent - mean_ce / M;
        } else {
            double p_ens[32];
            for (int k = 0; k < C; ++k) p_ens[k] = 0.0;
            double mean_ce = 0.0;
            for (int m = 0; m < M; ++m) {
                const float *lg = sem + (((int64_t)m * V + v) * P + p) * C;
                double mx = -1e300;
                for (int k = 0; k < C; ++k) mx = fmax(mx, (double)lg[k]);
                double den = 0.0;
                for (int k = 0; k < C; ++k) den += exp((double)lg[k] - mx);
                double ce = 0.0;
                for (int k = 0; k < C; ++k) {
                    const double pk = exp((double)lg[k] - mx) / den;
                    p_ens[k] += pk;
                    ce -= (pk + 1e-4) * log(pk + 1e-4);
                }
                mean_ce += ce;
            }
            double ent = 0.0;
            for (int k = 0; k < C; ++k) { const double pk = p_ens[k] / M; ent -= (pk + 1e-4) * log(pk + 1e-4); }
            s_sem += ent - mean_ce / M;
        }
        // occupancy (pipeline.py:762-773)
        {
            double a_ens = 0.0, mean_ce = 0.0;
            for (int m = 0; m < M; ++m) {
                const double a = acc[((int64_t)m * V + v) * P + p];
                a_ens += a;
                mean_ce += -(a + 1e-4) * log(a + 1e-4) - (1.0 - a + 1e-4) * log(1.0 - a + 1e-4);
            }
            a_ens /= M;
            s_occ += -(a_ens + 1e-4) * log(a_ens + 1e-4) - (1.0 - a_ens + 1e-4) * log(1.0 - a_ens + 1e-4) - mean_ce / M;
        }
    }
    const double t0 = block_sum(s_rgb, s_buf), t1 = block_sum(s_dep, s_buf), t2 = block_sum(s_sem, s_buf), t3 = block_sum(s_occ, s_buf);
    if (threadIdx.x == 0) {
        terms[4 * v + 0] = t0 / (3.0 * P);
        terms[4 * v + 1] = t1 / P;
        terms[4 * v + 2] = t2 / P;
        terms[4 * v + 3] = t3 / P;
    }
}

}  // namespace mnf

using namespace mnf;

extern "C" int64_t mnf_render_workspace_bytes(int64_t n_rays, int32_t rays_per_view) {
    if (n_rays <= 0 || rays_per_view <= 0 || n_rays % rays_per_view) return -1;
    return carve(nullptr, nullptr, n_rays, rays_per_view);
}

// ------------------------------------------------------------------ render jobs
// One job = one mnf_render_test call (a batch of views of one field).  Several independent jobs (the members of an ensemble,
// groups of views of one call) advance side by side on separate streams: every job is a chain prep -> march -> field per round,
// and while one job's short kernels (prep, march, the tail of a field launch) leave compute units idle the other jobs' launches
// fill them.  A view's result does not depend on which job it is in (its round schedule and its tiles are its own).
// The host enqueues `sync_every` rounds per job at a time and learns one block LATE whether a job has finished (the flag word is
// copied to pinned memory behind the first prep of a block and waited for after the block is enqueued): the queues never run dry,
// at the price of at most one block of empty rounds per job.
namespace mnf {
namespace {

struct JobRes {          // per-(thread, device) pool: pinned flag words, events, side streams
    int32_t *flags = nullptr;
    hipEvent_t ev_flags = nullptr, ev_join = nullptr;
    hipStream_t side = nullptr;
};
struct JobPool {
    std::vector<JobRes> res;
    hipEvent_t ev_fork = nullptr;
    int ensure(size_t n) {
        if (!ev_fork) MNF_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        while (res.size() < n) {
            JobRes r;
            MNF_HIP(hipHostMalloc((void **)&r.flags, 64, hipHostMallocDefault));
            MNF_HIP(hipEventCreateWithFlags(&r.ev_flags, hipEventDisableTiming));
            MNF_HIP(hipEventCreateWithFlags(&r.ev_join, hipEventDisableTiming));
            r.side = shared_side_stream((int)(res.size() % kSharedSideStreams));     // (job 0 runs on the caller's stream; more than 3 side jobs share)
            if (!r.side) return MNF_ERR_HIP;
            res.push_back(r);
        }
        return MNF_OK;
    }
};
// One pool per host thread AND device: events and the shared side streams belong to the device that was current when they
// were created, so a later call for a field on another GPU must not record device-0 events on device-1 streams.
JobPool *job_pool() {
    static thread_local std::vector<JobPool> pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) { set_error("render: no current device"); return nullptr; }
    if ((int)pools.size() <= dev) pools.resize(dev + 1);
    return &pools[dev];
}

struct RenderJob {
    mnf_field_t f;
    const uint8_t *binaries;
    const float *rays_o, *rays_d;
    int64_t n_rays;
    mnf_render_opts opts;
    RenderWs ws;
    RenderOut out;
    FieldIO io;
    hipStream_t s;
    JobRes *res;
    I3 grid;
    float ab[6];
    LevelBoxes boxes;
    int n_words, max_rounds, round;
    bool lds_grid, done, flags_pending;
    int flags_parity;
    int32_t n_views, bpv, min_samples, C;
};

int job_begin(RenderJob &j, mnf_field_t f, const uint8_t *binaries, int32_t res_x, int32_t res_y, int32_t res_z, const float *aabb_host,
              const float *rays_o, const float *rays_d, int64_t n_rays, const mnf_render_opts *opts, float *rgb, float *acc, float *depth,
              float *sem, float *rgb_var, float *depth_var, int64_t *total_samples, void *workspace, int64_t workspace_bytes, hipStream_t s,
              JobRes *res) {
    MNF_REQUIRE(f && opts && aabb_host, "render_test: null argument");
    MNF_REQUIRE(opts->struct_size == sizeof(mnf_render_opts), "render_test: opts->struct_size is %u, this library's mnf_render_opts has %zu bytes (MNF_INIT)",
                opts->struct_size, sizeof(mnf_render_opts));
    MNF_REQUIRE(f->params_loaded, "render_test: field parameters not loaded");
    MNF_REQUIRE(n_rays > 0, "render_test: a job needs rays");
    MNF_REQUIRE(opts->rays_per_view > 0 && n_rays % opts->rays_per_view == 0,
                "render_test: n_rays (%lld) must be a multiple of rays_per_view (%d)", (long long)n_rays, opts->rays_per_view);
    MNF_REQUIRE(n_rays <= (int64_t)500 * 1000 * 1000, "render_test: too many rays for 32-bit column indices");
    MNF_REQUIRE(binaries && rays_o && rays_d && rgb && acc && depth && sem && total_samples, "render_test: null buffer");
    MNF_REQUIRE(!opts->probabilistic || (rgb_var && depth_var), "render_test: probabilistic needs rgb_var and depth_var");
    MNF_REQUIRE(opts->max_samples > 0 && opts->render_step_size > 0.f, "render_test: max_samples and render_step_size must be > 0");
    const int64_t need = carve(nullptr, nullptr, n_rays, opts->rays_per_view);
    if (!workspace || workspace_bytes < need) {
        set_error("render_test: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)need);
        return MNF_ERR_WORKSPACE;
    }
    j.f = f; j.binaries = binaries; j.rays_o = rays_o; j.rays_d = rays_d; j.n_rays = n_rays; j.opts = *opts; j.s = s; j.res = res;
    carve(&j.ws, (char *)workspace, n_rays, opts->rays_per_view);
    j.out = {rgb, acc, depth, sem, opts->probabilistic ? rgb_var : nullptr, opts->probabilistic ? depth_var : nullptr, total_samples};
    j.n_views = (int32_t)(n_rays / opts->rays_per_view);
    j.C = f->cfg.num_semantic_classes;
    const int ray_blocks = (int)ceil_div(n_rays > j.n_views ? n_rays : j.n_views, kRayThreads);
    const int n_levels = opts->n_levels > 1 ? opts->n_levels : 1;
    MNF_REQUIRE(n_levels <= 4, "render_test: at most 4 occupancy levels (got %d)", n_levels);
    for (int k = 0; k < 6; ++k) j.ab[k] = aabb_host[k];
    for (int l = 0; l < n_levels; ++l)
        for (int k = 0; k < 6; ++k) j.boxes.ab[l][k] = aabb_host[6 * l + k];
    j.boxes.n = n_levels;
    const float *ab = j.ab;
    j.grid = {res_x, res_y, res_z};
    j.min_samples = opts->cone_angle == 0.f ? 1 : 4;                                  // utils.py:645
    if (const char *e = diag_env("MNF_MIN_SAMPLES")) j.min_samples = atoi(e);   // diagnostic only (locality experiments): NOT the reference's schedule
    const float opc_thre = 1.0f - opts->early_stop_eps;                                // utils.py:664
    const int64_t cells = (int64_t)res_x * res_y * res_z;
    j.boxes.words_per_level = (int)ceil_div(cells, 32);
    j.n_words = j.boxes.words_per_level * n_levels;          // every level's bits side by side in LDS
#ifdef MNF_NO_LDS_GRID
    j.lds_grid = false;                                      // A/B builds: occupancy bytes straight from global memory
#else
    j.lds_grid = j.n_words <= kMaxGridWords;
#endif
    if (j.lds_grid) {
        if (opts->bitgrid) j.ws.bitgrid = const_cast<uint32_t *>(opts->bitgrid);   // the estimator's own packed grid: nothing to build
        else hipLaunchKernelGGL(pack_grid_kernel, dim3((int)ceil_div(j.n_words, 256)), dim3(256), 0, s, binaries, cells, j.ws.bitgrid, j.n_words, j.boxes.words_per_level);
    }
    MNF_HIP(hipMemsetAsync(sem, 0, (size_t)n_rays * j.C * sizeof(float), s));   // [R,C] accumulators: one streaming fill
    hipLaunchKernelGGL(init_kernel, dim3(ray_blocks), dim3(kRayThreads), 0, s, n_rays, opts->rays_per_view, j.C, rays_o, rays_d,
                       ab[0], ab[1], ab[2], ab[3], ab[4], ab[5], opts->near_plane, j.ws, j.out);
    int rc = launch_status("init_kernel");
    if (rc) return rc;
    FieldIO io = {};
    io.mode = 2; io.rays_o = rays_o; io.rays_d = rays_d; io.col_ray = j.ws.col_ray; io.t_starts = j.ws.col_ts; io.t_ends = j.ws.col_te;
    io.n_dev = j.ws.n_cols; io.n_cap = j.ws.col_cap;
#ifndef MNF_STATIC_TILES
    // tiles in arrival order (field.hip, ticket_take).  -DMNF_STATIC_TILES, or MNF_FIELD_STATIC_TILES in the diagnostic library: the fixed stride of rounds 1-4
    // (A/B builds; tests/diag_tile_order.py: the two orders render the same bits)
    if (!diag_env("MNF_FIELD_STATIC_TILES")) io.tickets = j.ws.tickets;
#endif
    io.enc = split_field() ? j.ws.enc : nullptr;   // MNF_FIELD_SPLIT (diagnostic: gather and MLP as two launches on one stream)
    io.fr.tile_hdr = j.ws.tile_hdr; io.fr.alive = j.ws.alive; io.fr.alive_count = j.ws.alive_count;
    io.fr.n_samples = j.ws.n_samples; io.fr.rgb = rgb; io.fr.acc = acc; io.fr.depth = depth; io.fr.sem = sem;
    io.fr.rgb_var = j.out.rgb_var; io.fr.depth_var = j.out.depth_var;
    io.fr.totals = reinterpret_cast<unsigned long long *>(total_samples);
    io.fr.rays_per_view = opts->rays_per_view; io.fr.probabilistic = opts->probabilistic;
    io.fr.general_only = diag_env("MNF_COMPOSITE_GENERAL") != nullptr;   // tests compare the two compositing paths with it
    io.fr.alpha_thre = opts->alpha_thre; io.fr.opc_thre = opc_thre;
    j.io = io;
    j.max_rounds = (int)ceil_div(opts->max_samples, j.min_samples);
    j.bpv = (int32_t)march_blocks_per_view(opts->rays_per_view);
    j.round = 0; j.done = false; j.flags_pending = false; j.flags_parity = 0;
    return MNF_OK;
}

// enqueue the next `block` rounds of a job; behind the first prep of every block but the first, the flag words go to pinned memory
int job_enqueue_block(RenderJob &j, int block) {
    hipStream_t s = j.s;
    const float *ab = j.ab;
    const mnf_render_opts *opts = &j.opts;
    const int march_grid = (int)((int64_t)j.n_views * j.bpv);
    const int last = j.round + block < j.max_rounds ? j.round + block : j.max_rounds;
    for (int k = 0; j.round < last; ++j.round, ++k) {
        const int round = j.round;
        const int parity = round & 1;
        if (round_log()) MNF_HIP(hipEventRecord(log_events()[2], s));
#define MNF_MARCH(LDS, ML) hipLaunchKernelGGL((round_march_kernel<LDS, ML>), dim3(march_grid), dim3(kMarchThreads), 0, s, j.n_rays, opts->rays_per_view, j.rays_o, \
                                             j.rays_d, j.binaries, j.grid, j.n_words, ab[0], ab[1], ab[2], ab[3], ab[4], ab[5], opts->far_plane,                  \
                                             opts->render_step_size, opts->cone_angle, j.ws, opts->view_order, j.bpv, j.boxes, j.n_views, parity, opts->max_samples, j.min_samples)
        if (j.boxes.n > 1) { if (j.lds_grid) MNF_MARCH(true, true); else MNF_MARCH(false, true); }
        else { if (j.lds_grid) MNF_MARCH(true, false); else MNF_MARCH(false, false); }
#undef MNF_MARCH
        if (k == 0 && round > 0) {
            // the marcher just enqueued decided whether any view still has a round to run: any_active[2], overflow (adjacent words)
            MNF_HIP(hipMemcpyAsync(j.res->flags, j.ws.any_active, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            MNF_HIP(hipEventRecord(j.res->ev_flags, s));
            j.flags_pending = true; j.flags_parity = parity;
        }
        if (round_log()) MNF_HIP(hipEventRecord(log_events()[3], s));
        j.io.n_dev = j.ws.n_cols + parity;
        j.io.fr.alive_count = j.ws.alive_count + (parity ^ 1) * j.n_views;
        int rc;
        {
            ProfScope ps("field_render", s);
            if (round_log()) MNF_HIP(hipEventRecord(log_events()[0], s));
            rc = launch_field(j.f, j.io, false, s);   // field evaluation + compositing + ray retirement of this round
            if (round_log()) MNF_HIP(hipEventRecord(log_events()[1], s));
        }
        if (rc) return rc;
        if (round_log()) {   // MNF_ROUND_LOG (diagnostic, synchronises every round): columns and per-view budgets of the round
            const int n_views = j.n_views;
            int32_t n_cols = 0, ns[8] = {0}, act[8] = {0};
            std::vector<int32_t> act_all(n_views), alive_all(n_views);
            MNF_HIP(hipMemcpyAsync(&n_cols, j.ws.n_cols + parity, 4, hipMemcpyDeviceToHost, s));
            MNF_HIP(hipMemcpyAsync(ns, j.ws.n_samples, 4 * (n_views < 8 ? n_views : 8), hipMemcpyDeviceToHost, s));
            MNF_HIP(hipMemcpyAsync(act, j.ws.active, 4 * (n_views < 8 ? n_views : 8), hipMemcpyDeviceToHost, s));
            MNF_HIP(hipMemcpyAsync(act_all.data(), j.ws.active, 4 * (size_t)n_views, hipMemcpyDeviceToHost, s));
            MNF_HIP(hipMemcpyAsync(alive_all.data(), j.ws.alive_count + (parity ^ 1) * n_views, 4 * (size_t)n_views, hipMemcpyDeviceToHost, s));
            MNF_HIP(hipStreamSynchronize(s));
            int n_act = 0; long long n_alive_after = 0;
            for (int v = 0; v < n_views; ++v) { n_act += act_all[v] != 0; n_alive_after += alive_all[v]; }
            float ms = 0.f, ms_march = 0.f;
            (void)hipEventElapsedTime(&ms, log_events()[0], log_events()[1]);
            (void)hipEventElapsedTime(&ms_march, log_events()[2], log_events()[3]);
            fprintf(stderr, "[mnf round %d] cols %d  field %.4f ms (%.3f ns/col)  march %.4f ms  active_views %d  alive_after %lld  budgets", round, n_cols, ms, n_cols ? ms * 1e6 / n_cols : 0.0, ms_march, n_act, n_alive_after);
            for (int v = 0; v < n_views && v < 8; ++v) fprintf(stderr, " %d", act[v] ? ns[v] : 0);
            fprintf(stderr, "\n");
        }
    }
    return MNF_OK;
}

// wait for the flag words of the block just enqueued (they were written at its START: the device is still a block behind)
int job_check(RenderJob &j) {
    if (j.flags_pending) {
        MNF_HIP(hipEventSynchronize(j.res->ev_flags));
        j.flags_pending = false;
        if (j.res->flags[2]) {
            set_error("render_test: a round needed more sample columns than the workspace holds");
            return MNF_ERR_WORKSPACE;
        }
        if (!j.res->flags[j.flags_parity]) j.done = true;        // the rounds of this block find nothing to do
    }
    if (j.round >= j.max_rounds) j.done = true;
    return MNF_OK;
}

int job_finish(RenderJob &j) {
    hipLaunchKernelGGL(finalize_kernel, dim3((int)ceil_div(j.n_rays, kRayThreads)), dim3(kRayThreads), 0, j.s, j.n_rays,
                       j.opts.render_bkgd[0], j.opts.render_bkgd[1], j.opts.render_bkgd[2], j.out);
    return launch_status("finalize_kernel");
}

int run_jobs(std::vector<RenderJob> &jobs) {
    int block = jobs[0].opts.sync_every > 0 ? jobs[0].opts.sync_every : 1 << 30;
    if (round_log()) block = 1 << 30;
    bool any = true;
    // Negative result kept as a diagnostic (MNF_JOB_SHARE=1, profiles/r03_split_experiment.txt): giving every job's field launch 256 / (jobs running)
    // workgroups, so that the jobs' field kernels run BESIDE each other, is slower everywhere (800x800 x4: 61.2 -> 65.8 ms with two jobs, 111 ms with four;
    // scoring 106.9 -> 114.1 ms; 32 views 22.7 -> 23.7 ms): the half-sized launches take twice as long and do not overlap accordingly.
    auto share = [&]() {
        static const bool on = diag_env("MNF_JOB_SHARE") != nullptr;
        if (!on) return;
        int running = 0;
        for (auto &j : jobs) running += !j.done;
        int g = running > 1 ? 256 / running : 0;
        g = g ? (g / 8) * 8 : 0;
        for (auto &j : jobs) j.io.grid_limit = g && g < 32 ? 32 : g;
    };
#ifdef MNF_DIAG
    const bool host_log = diag_env("MNF_HOST_LOG") != nullptr;     // host time spent enqueuing vs waiting (diagnostic)
    double t_enq = 0.0, t_wait = 0.0; int n_rounds = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
#endif
    while (any) {
#ifdef MNF_DIAG
        const double t0 = now();
        for (auto &j : jobs) n_rounds -= j.round;
#endif
        share();
        for (auto &j : jobs) if (!j.done) { int rc = job_enqueue_block(j, block); if (rc) return rc; }
#ifdef MNF_DIAG
        const double t1 = now();
        for (auto &j : jobs) n_rounds += j.round;
#endif
        any = false;
        for (auto &j : jobs) if (!j.done) { int rc = job_check(j); if (rc) return rc; any = any || !j.done; }
#ifdef MNF_DIAG
        t_enq += t1 - t0; t_wait += now() - t1;
#endif
    }
#ifdef MNF_DIAG
    if (host_log) fprintf(stderr, "[mnf jobs] %zu jobs, %d job-rounds: host enqueue %.3f ms (%.2f us per job-round), host wait %.3f ms\n", jobs.size(), n_rounds,
                          1e3 * t_enq, n_rounds ? 1e6 * t_enq / n_rounds : 0.0, 1e3 * t_wait);
#endif
    for (auto &j : jobs) { int rc = job_finish(j); if (rc) return rc; }
    return MNF_OK;
}

}  // namespace
}  // namespace mnf

extern "C" int mnf_render_test(mnf_field_t f, const uint8_t *binaries, int32_t res_x, int32_t res_y, int32_t res_z,
                               const float *aabb_host, const float *rays_o, const float *rays_d, int64_t n_rays,
                               const mnf_render_opts *opts,
                               float *rgb, float *acc, float *depth, float *sem, float *rgb_var, float *depth_var,
                               int64_t *total_samples, void *workspace, int64_t workspace_bytes, mnf_stream_t stream) {
    MNF_REQUIRE(n_rays >= 0, "render_test: negative n_rays");
    if (n_rays == 0) return MNF_OK;
    JobPool *pp = job_pool();
    if (!pp) return MNF_ERR_HIP;
    JobPool &pool = *pp;
    int rc = pool.ensure(1);
    if (rc) return rc;
    std::vector<RenderJob> jobs(1);
    rc = job_begin(jobs[0], f, binaries, res_x, res_y, res_z, aabb_host, rays_o, rays_d, n_rays, opts, rgb, acc, depth, sem, rgb_var, depth_var,
                   total_samples, workspace, workspace_bytes, as_stream(stream), &pool.res[0]);
    if (rc) return rc;
    return run_jobs(jobs);
}

extern "C" int mnf_render_jobs(const mnf_render_job *jobs_host, int32_t n_jobs, int32_t res_x, int32_t res_y, int32_t res_z,
                               const float *aabb_host, const mnf_render_opts *opts, mnf_stream_t stream) {
    MNF_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= 64 && opts, "render_jobs: bad arguments");
    MNF_REQUIRE(opts->struct_size == sizeof(mnf_render_opts), "render_jobs: opts->struct_size is %u, this library's mnf_render_opts has %zu bytes (MNF_INIT)",
                opts->struct_size, sizeof(mnf_render_opts));
    for (int k = 0; k < n_jobs; ++k)      // before anything is enqueued or forked
        MNF_REQUIRE(jobs_host[k].struct_size == sizeof(mnf_render_job), "render_jobs: jobs[%d].struct_size is %u, this library's mnf_render_job has %zu bytes (MNF_INIT)", k,
                    jobs_host[k].struct_size, sizeof(mnf_render_job));
    JobPool *pp = job_pool();
    if (!pp) return MNF_ERR_HIP;
    JobPool &pool = *pp;
    int rc = pool.ensure((size_t)n_jobs);
    if (rc) return rc;
    hipStream_t s0 = as_stream(stream);
    std::vector<RenderJob> jobs;
    jobs.reserve(n_jobs);
    MNF_HIP(hipEventRecord(pool.ev_fork, s0));            // whatever produced the inputs on the caller's stream comes first
    for (int k = 0; k < n_jobs; ++k) {
        const mnf_render_job &d = jobs_host[k];
        if (d.n_rays == 0) continue;
        hipStream_t s = k == 0 ? s0 : pool.res[k].side;
        if (k) MNF_HIP(hipStreamWaitEvent(s, pool.ev_fork, 0));
        mnf_render_opts o = *opts;
        o.bitgrid = d.bitgrid;
        jobs.emplace_back();
        rc = job_begin(jobs.back(), d.field, d.binaries, res_x, res_y, res_z, aabb_host, d.rays_o, d.rays_d, d.n_rays, &o, d.rgb, d.acc, d.depth,
                       d.sem, d.rgb_var, d.depth_var, d.total_samples, d.workspace, d.workspace_bytes, s, &pool.res[k]);
        if (rc) return rc;
    }
    if (jobs.empty()) return MNF_OK;
    rc = run_jobs(jobs);
    // join: the caller's stream continues after every job (also on an error path, so that no side stream is left racing the caller)
    for (auto &j : jobs)
        if (j.s != s0) { (void)hipEventRecord(j.res->ev_join, j.s); (void)hipStreamWaitEvent(s0, j.res->ev_join, 0); }
    return rc;
}

extern "C" int mnf_score_views(const float *rgb_var, const float *depth_var, const float *acc, const float *sem,
                               int32_t n_members, int32_t n_views, int32_t n_pix, int32_t n_classes, double *terms, mnf_stream_t stream) {
    MNF_REQUIRE(n_members >= 1 && n_views >= 0 && n_pix > 0 && n_classes >= 1 && n_classes <= 32, "score_views: bad sizes");
    if (n_views == 0) return MNF_OK;
    MNF_REQUIRE(rgb_var && depth_var && acc && sem && terms, "score_views: null pointer");
    if (n_members <= kScoreMaxM)
        hipLaunchKernelGGL(score_kernel<true>, dim3(n_views), dim3(kScoreThreads), 0, as_stream(stream), rgb_var, depth_var, acc, sem, n_members, n_views,
                           n_pix, n_classes, terms);
    else
        hipLaunchKernelGGL(score_kernel<false>, dim3(n_views), dim3(kScoreThreads), 0, as_stream(stream), rgb_var, depth_var, acc, sem, n_members, n_views,
                           n_pix, n_classes, terms);
    return launch_status("score_kernel");
}
