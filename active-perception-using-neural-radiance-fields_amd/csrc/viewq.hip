// View-queue renderer: the reference's test-mode render loop (perception/models/utils.py:555-779, :782-1032) for batches of SMALL views
// (the predictive-information scorer's 64 x 64 sub-sampled candidate views, scripts/pipeline.py:697-711) as ONE persistent launch.
//
// The per-round form (render.hip) spends three dependent launches per round — budget, march, field + compositing — and a view of a few
// thousand rays lives through ~150 rounds: at the tail a round is ~100 us of launch and pipeline latency around a handful of tiles, and a
// rank's share of an 8-GPU scoring pass (32 views) ran at 1.7x its share of the single-GPU time.  Here nothing but the data dependence of the
// reference's loop orders the work:
//
//   * every WAVE is a worker of its own.  A work ITEM is (view, round, <= 64 of the view's still-alive rays): the wave marches them against the
//     occupancy bits in LDS (lane = ray; march_dev.h, the bit-exact restatement of grid.cu:68-282), packs the samples into 64-column tiles in
//     its private scratch (a ray never straddles a tile), evaluates the field on each tile with the same register-resident gather -> MLP -> heads
//     chain as field.hip (lane = sample), composites into the rays' accumulators (composite_dev.h) and appends the rays that stay alive to the
//     view's list for the next round.  The two waves of a SIMD are in different phases of different items, so one wave's march (dependent
//     scalar-ish arithmetic) runs under the other's matrix instructions.
//   * only items whose inputs exist are ever in a queue: round 0 of every view at launch; the wave that finishes the LAST item of a view's
//     round turns the length of the next list into the next per-ray budget n_samples = max(min(R // n_alive, 64), min_samples) (utils.py:667-672:
//     the reference's `.item()` round trip, taken on the device) and pushes the next round's items.  Views advance independently of each other.
//   * no wave ever waits for a particular other wave (a popper waits for pushes, and pushes come from waves that are running), so the launch needs no
//     co-residency and cannot deadlock against other processes' kernels; every wait is bounded (4 s, then an error code).  Hand-offs — a ray's
//     accumulators written in round k by one CU, read in round k + 1 by another — follow the agent-scope release / acquire protocol (producer: the
//     wave drains its stores, release fence + drain, then the atomic on the view's arrival counter; consumer: relaxed poll of its queue slot, acquire
//     fence + drain, plain loads).
//   * a workgroup serves ONE (field, occupancy grid) pair for the whole launch — an ensemble's members are separate groups with their own
//     workgroups — so the weights (86 KB at 128 x 2) and the occupancy bits (<= 64 KB) are staged into LDS once.
//
// Per-ray results do not depend on what else is in the batch, on the order rays sit in a list or on which rays share a tile: a view's budget
// schedule is its own, a matrix product's columns are independent, and a ray's sums run over its own samples (tests: bitwise repeatable, a view
// alone == the view in a batch, == the per-round renderer to fp32 rounding).
#include "composite_dev.h"
#include "viewq.h"

MNF_DT_BEGIN

static_assert(kVQWaves == kWavesPerBlock, "viewq.h and field_dev.h disagree on the waves of a workgroup");

typedef const VQJob __attribute__((address_space(4))) *JobPtr;
typedef const VQArgs __attribute__((address_space(4))) *ArgPtr;

// Pointers that come out of the job record (device memory) are generic to the compiler: derived from an address-space-1 pointer they become global again,
// so that every access through them is a global_* instruction (flat_* ones count on both wait counters and would undo the gather's counted waits).
template <class T>
__device__ __forceinline__ T *as_global(T *p) { return (T *)(T __attribute__((address_space(1))) *)(uintptr_t)p; }

__device__ __forceinline__ int aload(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int aadd(int32_t *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct VQSink {
    float *ts, *te;      // the wave's scratch, already offset to the ray's first column
    __device__ __forceinline__ void sample(float t_last, float t_next, bool, int32_t k) { ts[k] = t_last; te[k] = t_next; }
};

// The compositing epilogue's view of a job, rebuilt from the job record where it is used (scalar loads; see field.hip fr_of_kernarg for why)
__device__ __forceinline__ FusedRender fr_of_job(JobPtr jp, ArgPtr ap) {
    asm volatile("" : "+s"(jp));
    asm volatile("" : "+s"(ap));
    FusedRender fr;
    fr.tile_hdr = nullptr; fr.alive = as_global(jp->alive); fr.alive_count = as_global(jp->alive_sink); fr.n_samples = as_global(jp->n_samples);
    fr.rgb = as_global(jp->rgb); fr.acc = as_global(jp->acc); fr.depth = as_global(jp->depth); fr.sem = as_global(jp->sem);
    fr.rgb_var = as_global(jp->rgb_var); fr.depth_var = as_global(jp->depth_var);
    fr.totals = as_global(jp->totals); fr.rays_per_view = ap->rays_per_view; fr.probabilistic = ap->probabilistic; fr.general_only = 0;
    fr.alpha_thre = ap->alpha_thre; fr.opc_thre = ap->opc_thre;
    return fr;
}

template <int W, int NH>
struct VQFits {
    static constexpr bool value = (Layout<W, NH>::blocks * 1024 + kVQGridWords * 4 + 1024) <= 160 * 1024;
};

#ifdef MNF_DIAG
#define VQ_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define VQ_ADD(k, a, b) st_acc[k] += (b) - (a)
#else
#define VQ_STAMP(var)
#define VQ_ADD(k, a, b)
#endif

constexpr unsigned long long kVQTimeout = 400000000ull;      // 4 s of the 100 MHz clock: no wait inside a healthy launch comes near it

template <int W, int NH>
__global__ void __launch_bounds__(kThreads, 2) viewq_kernel(const VQArgs args) {
    using L = Layout<W, NH>;
    constexpr int kBlocks = L::blocks;
    __shared__ half8 s_w[kBlocks * 64];
    __shared__ uint32_t s_bits[kVQGridWords];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const ArgPtr ap = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();

    // ------------------------------------------------------------------ this workgroup's group: its weights and occupancy bits -> LDS, once
    int grp = 0;
    while (grp + 1 < args.n_groups && (int)blockIdx.x >= args.group_wg_end[grp]) ++grp;
    const int j_lo = grp ? args.group_job_end[grp - 1] : 0, j_hi = args.group_job_end[grp];
    {
        const JobPtr j0 = (JobPtr)(uintptr_t)(args.jobs + j_lo);
        const half8 *src = reinterpret_cast<const half8 *>(as_global(j0->frags));
        constexpr int kPer = (kBlocks * 64 + kThreads - 1) / kThreads;
        half8 tmp[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x + j * kThreads;
            if (i < kBlocks * 64) tmp[j] = src[i];
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x + j * kThreads;
            if (i < kBlocks * 64) s_w[i] = tmp[j];
        }
        const uint32_t *bits = as_global(j0->bitgrid);
        const int nw = args.n_words;
        for (int q = threadIdx.x; 4 * q + 3 < nw; q += kThreads) reinterpret_cast<uint4 *>(s_bits)[q] = reinterpret_cast<const uint4 *>(bits)[q];
        for (int i = (nw & ~3) + (int)threadIdx.x; i < nw; i += kThreads) s_bits[i] = bits[i];
    }
    __syncthreads();

    const int n_gjobs = j_hi - j_lo;
    const int pref = j_lo + (int)((blockIdx.x * kWavesPerBlock + wave) % (unsigned)n_gjobs);      // the queue this wave tries first
    const int64_t wbase = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kVQWaveCols;            // this wave's column scratch
    int32_t *const c_ray = args.col_ray + wbase;
    float *const c_ts = args.col_ts + wbase, *const c_te = args.col_te + wbase;
    const int rpv = args.rays_per_view;

#ifdef MNF_DIAG
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        // ------------------------------------------------------------------ pop an item (lane 0), broadcast it
        VQ_STAMP(s0);
        int got = 0, jq = 0;
        unsigned item = 0;
        if (lane == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
            int stop = 0, naps = 0;
            while (!got && !stop) {
                bool all_done = true;
                for (int i = 0; i < n_gjobs && !got && !stop; ++i) {
                    const int q = pref + i < j_hi ? pref + i : pref + i - n_gjobs;
                    int32_t *ctrl = as_global(args.jobs[q].ctrl);
                    const int jd = aload(ctrl + kVQJobDone), tl = aload(ctrl + kVQTail), hd = aload(ctrl + kVQHead);
                    if (jd) continue;
                    all_done = false;
                    if (tl - hd <= 0) continue;
                    const unsigned hq = (unsigned)aadd(ctrl + kVQHead, 1);        // a ticket: the hq-th item this queue ever holds
                    const unsigned long long *slot = as_global(args.jobs[q].ring) + (hq & (unsigned)args.jobs[q].ring_mask);
                    for (;;) {
                        const unsigned long long e = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(e >> 32) == hq + 1u) { item = (unsigned)e; got = 1; jq = q; break; }
                        if (aload(ctrl + kVQJobDone)) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > kVQTimeout) { stop = 1; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (all_done || got) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > kVQTimeout) { stop = 1; break; }
                // nothing to take: back off (256 cycles .. ~8 k cycles) so that idle waves do not hammer the control words the busy ones update
                __builtin_amdgcn_s_sleep(4);
                if (naps > 1) __builtin_amdgcn_s_sleep(16);
                if (naps > 4) __builtin_amdgcn_s_sleep(48);
                if (naps > 16) __builtin_amdgcn_s_sleep(64);
                ++naps;
            }
            if (stop) {      // never hang the device: flag the error, close every queue
                astore(args.error, 1);
                for (int q = 0; q < args.n_jobs; ++q) astore(as_global(args.jobs[q].ctrl) + kVQJobDone, 1);
                got = 0;
            }
            if (got) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // what other waves wrote for this item is visible to this CU's loads from here on
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (!got) break;
        VQ_STAMP(s1); VQ_ADD(0, s0, s1);
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
        jq = __builtin_amdgcn_readfirstlane(jq);
        const JobPtr jp = (JobPtr)(uintptr_t)(args.jobs + jq);
        const int v = (int)(item >> kVQItemBits), it_i = (int)(item & ((1u << kVQItemBits) - 1u));

        // ------------------------------------------------------------------ this item's rays: a run of the view's list of alive rays
        const int ns = __builtin_amdgcn_readfirstlane(aload(as_global(jp->n_samples) + v));      // the view's per-ray budget of this round
        const int rpi = __builtin_amdgcn_readfirstlane(aload(as_global(jp->rpi) + v));
        const int n_alive = __builtin_amdgcn_readfirstlane(aload(as_global(jp->cnt) + v));
        const int lstride = rpv + 64;
        const int idx = it_i * rpi + lane;
        const bool go = lane < rpi && idx < n_alive;
        const int32_t *const lst = as_global(jp->list) + (int64_t)v * lstride;
        const int64_t r = go ? lst[idx] : 0;
        uint8_t *const j_alive = as_global(jp->alive);
        float *const j_near = as_global(jp->near_plane);
        const unsigned long long gm = __ballot(go);
        const int n_go = __popcll(gm);
        WaveCounters wc;
        if (n_go) {                                        // wave-uniform
            const int cap = 64 / ns;                       // rays per 64-column tile; the item's rays fill at most kVQWaveTiles tiles (vq_rays_per_item)
            const int tile_local = lane / cap, slot = lane - tile_local * cap;
            const int col0 = tile_local * 64 + slot * ns;
            if (go) {
                float ro[3], rd[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { ro[d] = as_global(jp->rays_o)[3 * r + d]; rd[d] = as_global(jp->rays_d)[3 * r + d]; }
                const float ray_near = j_near[r], ray_tmin = as_global(jp->t_min)[r], ray_tmax = as_global(jp->t_max)[r];
                const bool ray_hit = as_global(jp->hit)[r] != 0;
                if (slot == 0) {                           // the first ray of a tile blanks the columns no ray of the tile owns
                    const int nslots = min(cap, n_go - tile_local * cap);
                    for (int c = nslots * ns; c < 64; ++c) c_ray[tile_local * 64 + c] = -1;
                }
                const F3 org = {ro[0], ro[1], ro[2]};
                const F3 dir = {rd[0], rd[1], rd[2]};
                const F3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
                MarchState st = {ray_near, false, 0};
                VQSink sink = {c_ts + col0, c_te + col0};
                if (ray_hit) {                             // one grid level: the only interval is [t_min, t_max] (grid.cu:125-151 with n_grids == 1)
                    const float this_tmin = fmaxf(ray_tmin, ray_near);
                    const float this_tmax = fminf(ray_tmax, args.far_plane);
                    if (this_tmin < this_tmax) {
                        const float ab[6] = {args.occ_aabb[0], args.occ_aabb[1], args.occ_aabb[2], args.occ_aabb[3], args.occ_aabb[4], args.occ_aabb[5]};
                        march_segment(org, dir, inv, this_tmin, this_tmax, ab, args.res, BitGrid{s_bits}, args.step_size, args.cone_angle, ns, st, sink);
                    }
                }
                for (int c = 0; c < ns; ++c) c_ray[col0 + c] = c < st.n_samples ? (int32_t)r : -1;
                if (st.n_samples == 0) j_alive[r] = 0;     // left the grid: retired here, the compositing never sees it (utils.py:751-756)
                j_near[r] = st.t_last;                     // utils.py:749 near_planes = termination_planes
            }
            // the wave's own stores, then its own loads of the same addresses from other lanes (L2-served: a lane's store does not refresh another lane's L1 hit)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int n_tiles = (n_go + cap - 1) / cap;
            VQ_STAMP(s2); VQ_ADD(1, s1, s2);
#ifdef MNF_DIAG
            st_acc[6] += (unsigned long long)n_tiles;
#endif

            for (int t = 0; t < n_tiles; ++t) {
                // ---- this lane's sample (lane = column) ----
                ArgPtr lp = ap;
                asm volatile("" : "+s"(lp));
                JobPtr jl = jp;
                asm volatile("" : "+s"(jl));
                const int col = t * 64 + lane;
                TileSample tsm = {-1, ns, v, false, 0.f, 0.f, 0.f};
                tsm.ray = __hip_atomic_load(c_ray + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool valid = tsm.ray >= 0;
                tsm.valid = valid;
                float pos[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 1.f};
                if (valid) {
                    tsm.ts = __hip_atomic_load(c_ts + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tsm.te = __hip_atomic_load(c_te + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t ray = tsm.ray;                               // (ray ids are the job's)
                    tsm.opac0 = as_global(jl->acc)[ray];
                    const float tsum = tsm.ts + tsm.te;
                    const float *const j_o = as_global(jl->rays_o), *const j_d = as_global(jl->rays_d);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        dir[d] = j_d[3 * ray + d];
                        pos[d] = j_o[3 * ray + d] + (dir[d] * tsum) / 2.0f;                  // utils.py:614
                    }
                }
                float xn[3];
                bool selector = valid;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    xn[d] = (pos[d] - jl->aabb[d]) / (jl->aabb[3 + d] - jl->aabb[d]);      // ngp.py:177-178
                    selector = selector && (xn[d] > 0.0f) && (xn[d] < 1.0f);               // ngp.py:179
                }
                if (!valid) { xn[0] = 0.5f; xn[1] = 0.5f; xn[2] = 0.5f; }
                const LevelsPtr lv = levels_here(jl->levels);
                const tab4 *table = as_global(reinterpret_cast<const tab4 *>(jl->table));
                const bool in_box = __ballot(valid && !selector) == 0ull;

                // ---- hash encode (field.hip: 16 levels per lane, four batches of four, double-buffered), halves traded with lane ^ 32 ----
                half8 bfeat[CT][4];
                {
                    LevelPrep prep[2][4];
                    tab4 tv[2][4][8];
                    __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        hash_prep(level_meta(lv, q), xn, prep[0][q], in_box);
                        hash_load(table, prep[0][q], tv[0][q]);
                    }
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const int cur = kb & 1, nxt = cur ^ 1;
                        if (kb < 3) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                hash_prep(level_meta(lv, 4 * (kb + 1) + q), xn, prep[nxt][q], in_box);
                                hash_load(table, prep[nxt][q], tv[nxt][q]);
                            }
                        }
                        float f[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) hash_blend(prep[cur][q], tv[cur][q], f + 4 * q);
                        half8 lo, hi;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { lo[j] = (half_t)f[j]; hi[j] = (half_t)f[8 + j]; }
                        exchange_halves(lo, hi);
                        bfeat[0][kb] = lo; bfeat[1][kb] = hi;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_s_setprio(1);
                }

                // ---- base MLP ----
                half8 hb[CT][L::KSW];
                dense_relu<L::RT, 4>(s_w + L::o_b_in * 64, lane, bfeat, hb);
#pragma unroll
                for (int l = 0; l < NH - 1; ++l) {
                    half8 hn[CT][L::KSW];
                    dense_relu<L::RT, L::KSW>(s_w + (L::o_b_hid + l * L::RT * L::KSW) * 64, lane, hb, hn);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int kq = 0; kq < L::KSW; ++kq) hb[ct][kq] = hn[ct][kq];
                }
                f32x16 bo[CT];
                dense_out<L::KSW>(s_w + L::o_b_out * 64, lane, hb, bo);
                const int out16 = jl->out_fp16;
                if (out16) round_outputs_fp16(bo);
                const float logit_t0 = __shfl(bo[0][0], lane & 31, 64);
                const float logit_t1 = __shfl(bo[1][0], lane & 31, 64);
                const float sigma = selector ? expf((h ? logit_t1 : logit_t0) - 1.0f) : 0.0f;   // ngp.py:79, :193-195

                // ---- heads ----
                half8 bgeo[CT][1];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bgeo[ct][0][j] = (half_t)bo[ct][j];
                    if (h == 0) bgeo[ct][0][0] = (half_t)1.0f;
                }
                half8 hin[CT][2];
                {
                    half8 lo, hi;
                    sh4(dir, lo, hi);
                    exchange_halves(lo, hi);
                    hin[0][0] = lo; hin[1][0] = hi;
                    hin[0][1] = bgeo[0][0]; hin[1][1] = bgeo[1][0];
                }
                half8 h1[CT][L::KSh], h2[CT][L::KSh];
                f32x16 out_rgb[CT], out_sem[CT];
                dense_relu<L::RTh, 2>(s_w + L::o_h_in * 64, lane, hin, h1);
                dense_relu<L::RTh, L::KSh>(s_w + L::o_h_hid * 64, lane, h1, h2);
                dense_out<L::KSh>(s_w + L::o_h_out * 64, lane, h2, out_rgb);
                if (out16) round_outputs_fp16(out_rgb);
                dense_relu<L::RTh, 1>(s_w + L::o_s_in * 64, lane, bgeo, h1);
                dense_relu<L::RTh, L::KSh>(s_w + L::o_s_hid * 64, lane, h1, h2);
                dense_out<L::KSh>(s_w + L::o_s_out * 64, lane, h2, out_sem);
                if (out16) round_outputs_fp16(out_sem);
                float rgb[3];
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const float t0 = __shfl(out_rgb[0][c3], lane & 31, 64);
                    const float t1 = __shfl(out_rgb[1][c3], lane & 31, 64);
                    rgb[c3] = 1.0f / (1.0f + expf(-(h ? t1 : t0)));   // ngp.py:211-212
                }
                int C = jl->C;
                asm volatile("" : "+s"(C));
                int lane_v = lane;
                asm volatile("" : "+v"(lane_v));
                fused_composite(fr_of_job(jl, lp), C, lane_v, tsm, sigma, rgb, out_sem, wc);
            }
        }
        // ------------------------------------------------------------------ the rays that stay alive: this item's segment of the view's next list
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VQ_STAMP(s3);
        {
            const bool still = go && __hip_atomic_load(j_alive + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;     // (written by the compositing's owner lane of this wave)
            const unsigned long long sm = __ballot(still);
            if (still) as_global(jp->next)[(int64_t)v * lstride + it_i * rpi + __popcll(sm & ((1ull << lane) - 1ull))] = (int32_t)r;
            if (lane == 0) as_global(jp->segcnt)[(int64_t)v * jp->seg_stride + it_i] = __popcll(sm);
        }
        flush_counters(fr_of_job(jp, ap), wc, lane);

        // ------------------------------------------------------------------ publish; the last item of the view's round opens the next round
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int arrived = aadd(as_global(jp->done) + v, 1);
            last = arrived == aload(as_global(jp->n_items) + v) - 1;
            if (last) {      // every item of the view's round has finished: their survivor segments are visible from here on
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        VQ_STAMP(s4); VQ_ADD(2, s1, s3); VQ_ADD(3, s3, s4);
#ifdef MNF_DIAG
        st_acc[5] += 1;
#endif
        if (!last) continue;
        // compact the segments into the view's list, in item order (the whole wave; a view's list keeps its march order through every round)
        int n_next = 0;
        {
            const int n_it = __builtin_amdgcn_readfirstlane(aload(as_global(jp->n_items) + v));
            const int32_t *const seg = as_global(jp->segcnt) + (int64_t)v * jp->seg_stride;
            const int32_t *const nxt = as_global(jp->next) + (int64_t)v * lstride;
            int32_t *const dst = as_global(jp->list) + (int64_t)v * lstride;
            const int n_e = n_it * rpi;
            // four 64-entry chunks per trip: all their loads are issued before the first is consumed (one dependent round trip per trip, not per chunk:
            // this loop is on the critical path of every round of the view)
            for (int e0 = 0; e0 < n_e; e0 += 256) {
                int32_t cnt4[4], val4[4];
                int k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + 64 * u + lane;
                    const int it_e = e / rpi;
                    k4[u] = e - it_e * rpi;
                    const bool in = e < n_e;
                    cnt4[u] = in ? seg[it_e] : 0;
                    val4[u] = in ? nxt[e] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = k4[u] < cnt4[u];
                    const unsigned long long m = __ballot(ok);
                    if (ok) dst[n_next + __popcll(m & ((1ull << lane) - 1ull))] = val4[u];
                    n_next += __popcll(m);
                }
            }
        }
        int n_push = 0, t_push = 0;
        if (lane == 0) {
            int32_t *const j_ctrl = as_global(jp->ctrl);
            const int it = aload(as_global(jp->iter_samples) + v);
            if (it < args.max_samples && n_next > 0) {                               // utils.py:666-672
                const int nxt_ns = max(min(rpv / n_next, 64), args.min_samples);
                const int rpi_n = vq_rays_per_item(nxt_ns, n_next, args.waves_per_view);
                n_push = (n_next + rpi_n - 1) / rpi_n;
                astore(as_global(jp->n_samples) + v, nxt_ns);
                astore(as_global(jp->iter_samples) + v, it + nxt_ns);
                astore(as_global(jp->rpi) + v, rpi_n);
                astore(as_global(jp->n_items) + v, n_push);
                astore(as_global(jp->cnt) + v, n_next);
                astore(as_global(jp->done) + v, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the list the compaction wrote, every lane's part of it)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t_push = aadd(j_ctrl + kVQTail, n_push);
            } else if (aadd(j_ctrl + kVQViewsLeft, -1) == 1) {
                astore(j_ctrl + kVQJobDone, 1);
            }
        }
        n_push = __builtin_amdgcn_readfirstlane(n_push);
        VQ_STAMP(s5); VQ_ADD(4, s4, s5);
#ifdef MNF_DIAG
        st_acc[7] += 1;
#endif
        if (n_push) {                                      // the whole wave writes the slots
            t_push = __builtin_amdgcn_readfirstlane(t_push);
            unsigned long long *const ring = as_global(jp->ring);
            const unsigned mask = (unsigned)jp->ring_mask;
            for (int i = lane; i < n_push; i += 64) {
                const unsigned tk = (unsigned)t_push + (unsigned)i;
                __hip_atomic_store(ring + (tk & mask), ((unsigned long long)(tk + 1u) << 32) | ((unsigned)v << kVQItemBits) | (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef MNF_DIAG
    if (args.stats && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(args.stats + k, st_acc[k]);
        atomicAdd(args.stats + 8, 1ull);
    }
#endif
}

bool viewq_supported(int W, int NH) {
#define MNF_VQ_CASE(w, nh) if (W == w && NH == nh) return VQFits<w, nh>::value
    MNF_VQ_CASE(128, 1); MNF_VQ_CASE(128, 2); MNF_VQ_CASE(128, 3); MNF_VQ_CASE(128, 4);
    MNF_VQ_CASE(64, 1); MNF_VQ_CASE(64, 2); MNF_VQ_CASE(64, 3); MNF_VQ_CASE(64, 4);
#undef MNF_VQ_CASE
    return false;
}

template <int W, int NH>
static int launch_one(const VQArgs &a, int grid, hipStream_t s) {
    if constexpr (VQFits<W, NH>::value) {
        hipLaunchKernelGGL((viewq_kernel<W, NH>), dim3(grid), dim3(kThreads), 0, s, a);
        return launch_status("viewq_kernel");
    } else {
        set_error("viewq: the weights of a %d x %d field and the occupancy bits do not fit the LDS together", W, NH);
        return MNF_ERR_UNSUPPORTED;
    }
}

int launch_viewq_impl(const VQArgs &a, int W, int NH, int grid, hipStream_t s) {
#define MNF_VQ_CASE(w, nh) if (W == w && NH == nh) return launch_one<w, nh>(a, grid, s)
    MNF_VQ_CASE(128, 1); MNF_VQ_CASE(128, 2); MNF_VQ_CASE(128, 3); MNF_VQ_CASE(128, 4);
    MNF_VQ_CASE(64, 1); MNF_VQ_CASE(64, 2); MNF_VQ_CASE(64, 3); MNF_VQ_CASE(64, 4);
#undef MNF_VQ_CASE
    set_error("viewq: unsupported neurons=%d layers=%d", W, NH);
    return MNF_ERR_UNSUPPORTED;
}

MNF_DT_END
