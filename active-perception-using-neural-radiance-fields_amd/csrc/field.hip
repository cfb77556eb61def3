// Fused radiance-field forward for gfx950: multiresolution hash-grid gather -> base MLP ->
// {trunc_exp density, SH + rgb head, semantic head} (-> per-ray compositing in renderer mode), one wave per 64 samples.
//
// Replaces the tiny-cuda-nn modules the reference builds at
// perception/models/radiance_fields/ngp.py:108-169 and the Python glue of ngp.py:171-238.
//
// Data flow per wave (64 lanes, 64 samples = two 32-column tiles "ct"):
//   * lane = sample for the gather: each lane encodes all 16 hash levels of its own sample (level metadata is
//     wave-uniform and read with scalar loads); one v_permlane32_swap per dword between lanes l and l+32 then turns
//     "lane = sample" into the fp16 B-operand fragments of v_mfma_f32_32x32x16_f16 for the first layer (feature
//     k = 16*ks + 8*h + j lives in element j of k-step ks of lane (c, h)), so encoded features never touch LDS or HBM;
//   * every layer computes H_out^T[n][c] = sum_k W[n][k] * H_in^T[k][c]: the weights are the A
//     operand (read from LDS, pre-permuted on the host side of the handle into fragment order,
//     one conflict-free ds_read_b128 per lane), the activations are the B operand.  The 32x32
//     accumulator of one layer (feature on the register index, sample on the lane) is, after
//     ReLU + cvt to fp16, directly the B fragment of the next layer (k order
//     32*nt + 16*s + 8*(j>>2) + 4*h + (j&3), matched by the weight permutation), so the whole
//     MLP chain stays in registers;
//   * the 16 base outputs sit in accumulator registers 0..7 of both lane halves: register 0 of
//     half 0 is the density logit, the other 15 are the geo features; replacing the logit by the
//     constant 1.0 (tcnn's input padding value) makes that fragment the input k-step of both heads.
//
// Precision: fp16 parameters, fp16 features/activations at every matrix-product input, fp32
// accumulation, fp32 outputs (oracle/field.py states the same model).
#include "composite_dev.h"

#include <cmath>
#include <cstring>



MNF_DT_BEGIN

// ------------------------------------------------------------------ sample fetch (shared by the kernels below)
// Position / direction of column `col`: mode 0 explicit arrays, mode 1 packed samples with int64 ray ids, mode 2 renderer
// columns.  xn = position normalised to the aabb (ngp.py:177-178), selector = inside the open unit box (ngp.py:179).
// Renderer columns of one tile (mode 2).  The tile header (budget | view << 8, written by the marcher of the same round) is
// wave-uniform and read with a scalar load.
struct ColData {
    int ray, stride, view;
    float ts, te;
};

template <class A>
__device__ __forceinline__ ColData load_cols(const A &args, int64_t tile, int lane) {
    const int64_t col = tile * kWaveSamples + lane;
    ColData c;
    typedef const int32_t __attribute__((address_space(4))) *HdrPtr;
    const int hdr = ((HdrPtr)(uintptr_t)args.io.fr.tile_hdr)[tile];
    c.stride = hdr & 0xff; c.view = hdr >> 8;
    c.ray = args.io.col_ray[col];
    c.ts = args.io.t_starts[col]; c.te = args.io.t_ends[col];
    return c;
}

template <int MODE, bool WANT_DIR, class A>
__device__ __forceinline__ void fetch_sample(const A &args, const ColData &cd, int64_t col, int64_t n, float (&xn)[3],
                                             float (&dir)[3], TileSample &tsm, bool &valid, bool &selector) {
    valid = col < n;
    float pos[3] = {0.f, 0.f, 0.f};
    dir[0] = 0.f; dir[1] = 0.f; dir[2] = 1.f;
    tsm = {-1, 64, 0, false, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        if (valid) {
#pragma unroll
            for (int d = 0; d < 3; ++d) pos[d] = args.io.positions[3 * col + d];
            if (WANT_DIR)
#pragma unroll
                for (int d = 0; d < 3; ++d) dir[d] = args.io.directions[3 * col + d];
        }
    } else if (MODE == 1 || MODE == 3) {
        int64_t ray = -1;
        if (valid) ray = args.io.ray_idx64[col];
        valid = ray >= 0;
        if (valid) {
            tsm.ts = args.io.t_starts[col]; tsm.te = args.io.t_ends[col];
            const float tsum = tsm.ts + tsm.te;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                dir[d] = args.io.rays_d[3 * ray + d];
                // utils.py:92 / :614: origins + dirs * (t_starts + t_ends) / 2.0
                pos[d] = args.io.rays_o[3 * ray + d] + (dir[d] * tsum) / 2.0f;
            }
            if (args.io.positions_out) {
#pragma unroll
                for (int d = 0; d < 3; ++d) args.io.positions_out[3 * col + d] = pos[d];
            }
        }
    } else {
        // renderer tile: column -> ray id (-1: unused); the runs of equal ids are the rays of this tile
        tsm.stride = cd.stride; tsm.view = cd.view; tsm.ray = cd.ray; tsm.ts = cd.ts; tsm.te = cd.te;
        valid = tsm.ray >= 0;
        tsm.valid = valid;
        if (valid) {
            if (WANT_DIR) tsm.opac0 = args.io.fr.acc[tsm.ray];
            const float tsum = tsm.ts + tsm.te;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                dir[d] = args.io.rays_d[3 * (int64_t)tsm.ray + d];
                pos[d] = args.io.rays_o[3 * (int64_t)tsm.ray + d] + (dir[d] * tsum) / 2.0f;   // utils.py:614
            }
        }
    }
    selector = valid;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        xn[d] = (pos[d] - args.aabb[d]) / (args.aabb[3 + d] - args.aabb[d]);   // ngp.py:177-178
        selector = selector && (xn[d] > 0.0f) && (xn[d] < 1.0f);               // ngp.py:179
    }
    if (MODE == 1 && args.io.xn_out && valid) {
#pragma unroll
        for (int d = 0; d < 3; ++d) args.io.xn_out[3 * col + d] = xn[d];
    }
    if (!valid) { xn[0] = 0.5f; xn[1] = 0.5f; xn[2] = 0.5f; }
}

// XCD-aware tile-group order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an L2), and
// neighbouring tiles hold neighbouring rays that touch the same hash-table lines, so each XCD walks one contiguous
// eighth of the tile groups.  A speed choice only: any placement is correct.  Grids that are not a multiple of 8 use
// the plain grid-stride order.
__device__ __forceinline__ void group_range(int64_t n_groups, int64_t &g_first, int64_t &g_end, int64_t &g_step) {
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7;
        g_first = n_groups * xcd / 8 + (blockIdx.x >> 3);
        g_end = n_groups * (xcd + 1) / 8;
        g_step = gridDim.x >> 3;
    } else {
        g_first = blockIdx.x; g_end = n_groups; g_step = gridDim.x;
    }
}

// Dynamic tile order (FieldIO::tickets, the render rounds): a wave takes the next tile of its XCD's eighth with one atomic per tile and, when that eighth is used up,
// goes on with the following eighths.  Consecutive tickets are consecutive tiles, so an XCD still walks neighbouring rays together; what changes is that no wave
// owns a fixed share: the workgroups of a launch that starts while another job's kernel drains (or whose tiles hit the caches less) no longer finish last.
constexpr int kTicketStride = 16;            // counters 64 bytes apart
__device__ __forceinline__ int ticket_tile(uint32_t t, int x, int64_t n_tiles) {
    const int64_t lo = n_tiles * x / 8, hi = n_tiles * (x + 1) / 8;
    return lo + (int64_t)t < hi ? (int)(lo + (int64_t)t) : -1;
}
__device__ __forceinline__ int ticket_take(uint32_t *tk, int &x, int &seen, int64_t n_tiles, int lane) {      // synchronous; -1 once all eight ranges are used up
    while (seen < 8) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(tk + kTicketStride * x, 1u);
        const int tile = ticket_tile(__builtin_amdgcn_readfirstlane(t), x, n_tiles);
        if (tile >= 0) return tile;
        x = (x + 1) & 7; ++seen;
    }
    return -1;
}

// ------------------------------------------------------------------ hash-grid encode as its own launch (diagnostic path)
// MNF_FIELD_SPLIT=1 runs the multiresolution gather and the MLP chain as two launches, which separates their costs:
// on the 800x800 workload the gather alone takes 64 % of the fused kernel's time and is insensitive to occupancy
// (3..8 waves per SIMD) and to halving the L1 tag lookups (paired 16-byte loads), i.e. it is bound by the miss path
// (random 64-byte fetches), while the MLP + compositing launch alone takes 38 %.  The fused kernel overlaps the two
// and stays the default: split is ~7 % slower end to end and needs 128 B of scratch per column.
// Features leave in the B-fragment order the MLP kernel consumes: half8 block ((tile*4 + ks)*2 + ct)*64 + (h*32 + c)
// holds levels 4ks+2h, 4ks+2h+1 of sample 32ct + c  (k = 16ks + 8h + j), so the consumer's load is one coalesced b128.
constexpr int kEncodeThreads = 256;

template <int MODE>
__global__ void __launch_bounds__(kEncodeThreads, 6) encode_kernel(const KernelArgs args, half8 *__restrict__ enc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int kWaves = kEncodeThreads / 64;
    constexpr int LPB = 2;   // levels per batch of gathers (16 loads in flight per wave, <= 80 VGPRs, 6 waves per SIMD)
    int64_t n = args.io.n;
    if (MODE == 2) { n = *args.io.n_dev; if (n > args.io.n_cap) n = args.io.n_cap; }
    const int64_t n_tiles_all = (n + kWaveSamples - 1) / kWaveSamples;
    int64_t tile0 = 0, n_tiles = n_tiles_all;
    if (args.io.n_chunks > 1) { tile0 = n_tiles_all * args.io.chunk / args.io.n_chunks; n_tiles = n_tiles_all * (args.io.chunk + 1) / args.io.n_chunks; }
    const int64_t n_groups = (n_tiles - tile0 + kWaves - 1) / kWaves;
    int64_t g_first, g_end, g_step;
    group_range(n_groups, g_first, g_end, g_step);
    for (int64_t grp = g_first; grp < g_end; grp += g_step) {
        const int64_t tile = tile0 + grp * kWaves + wave;
        if (tile >= n_tiles) break;
        const int64_t col = tile * kWaveSamples + lane;
        ColData cd = {-1, 64, 0, 0.f, 0.f};
        if (MODE == 2) cd = load_cols(args, tile, lane);
        float xn[3], dir[3];
        TileSample tsm;
        bool valid, selector;
        fetch_sample<MODE, false>(args, cd, col, n, xn, dir, tsm, valid, selector);
        const LevelsPtr lv = levels_here(args.levels);
        const bool in_box = __ballot(valid && !selector) == 0ull;   // wave-uniform: the cheap dense-level wrap applies
        half8 *dst = enc + (tile * 8 + (lane >> 5)) * 64 + (lane & 31);
#pragma unroll
        for (int l0 = 0; l0 < 16; l0 += LPB) {
            LevelPrep prep[LPB];
            tab4 v[LPB][8];
#pragma unroll
            for (int q = 0; q < LPB; ++q) {
                hash_prep(level_meta(lv, l0 + q), xn, prep[q], in_box);
                hash_load(args.table, prep[q], v[q]);
            }
#pragma unroll
            for (int q = 0; q < LPB; q += 2) {
                float f[8];
                hash_blend(prep[q], v[q], f);
                hash_blend(prep[q + 1], v[q + 1], f + 4);
                half8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)f[j];
                const int p = (l0 + q) >> 1;                      // level pair: ks = p >> 1, h = p & 1
                dst[((p >> 1) * 2) * 64 + (p & 1) * 32] = o;      // block (tile*4 + ks)*2 + ct, slot h*32 + c
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------ the fused kernel
// The compositing epilogue's arguments (eleven pointers and a few scalars) re-read from the kernel-argument segment where the epilogue runs, instead of living in
// SGPRs for the whole launch: the kernel sits at the SGPR limit (106) and the compiler had moved these loop-invariant pointers to VGPRs and from there to SCRATCH
// (72 bytes per lane, ~27 scratch loads per tile).  Scalar loads from the kernarg segment hit the scalar cache; the empty asm keeps them inside the tile loop.
// (the same for a scalar the epilogue compares per-lane row indices with: hoisted out of the tile loop, the 32 `row < C` lane masks alone took 64 SGPRs)
// by-value copy of a small struct that lives in the constant address space (the kernel-argument segment): scalar loads
template <class T>
__device__ __forceinline__ T from_constant(const T __attribute__((address_space(4))) *p) {
    T out;
    __builtin_memcpy(&out, (const void *)p, sizeof(T));
    return out;
}
__device__ __forceinline__ int in_loop(int x) { asm volatile("" : "+s"(x)); return x; }
// ... and for the lane index: the epilogue's lane-role predicates (first / last lane, DPP row, half) and row offsets are one v_cmp / v_or each; kept across the
// loop they were ~15 SGPR pairs and a dozen VGPRs, spilled to VGPR lanes (v_readlane + wait states at every use) and to scratch
__device__ __forceinline__ int in_loop_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ FusedRender fr_of_kernarg() {
    typedef const FusedRender __attribute__((address_space(4))) *FrPtr;
    typedef const char __attribute__((address_space(4))) *BytePtr;
    FrPtr p = (FrPtr)((BytePtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernelArgs, io) + offsetof(FieldIO, fr));
    asm volatile("" : "+s"(p));
    return from_constant(p);
}

// B16: the hash levels' 8-corner blend in tiny-cuda-nn's fp16 arithmetic (mnf_field_config.blend_fp16).  A template parameter, not a run-time branch: the
// kernel sits at its register limits, and a wave-uniform `if` around the two blends cost the DEFAULT path 3 % (0.7116 -> 0.7325 ms per render launch).
template <int W, int NH, int MODE, bool DENSITY_ONLY, int SAVEK = 0, int ENC = 0, bool B16 = false>
__global__ void __launch_bounds__(kThreads, 2) field_kernel(const KernelArgs args) {
    using L = Layout<W, NH>;
    using T = TrainLayout<W, NH>;
    // ENC: 0 the kernel encodes; 1 the features come from encode_kernel's scratch in fragment order (two-launch diagnostic path); 2 they come from the rows the
    // density pre-pass left (FieldIO::rows_in / rows_src: one 128-byte line per sample instead of 128 gathers); 3 the kernel encodes AND leaves those rows (rows_out)
    // SAVEK: 0 inference; 1 training with the activation dump the backward kernels (dgrad + wgrad) read
    constexpr bool SAVE = SAVEK == 1;
    constexpr int kBlocks = DENSITY_ONLY ? L::o_h_in : L::blocks;
    __shared__ half8 s_w[kBlocks * 64];
    __shared__ half_t s_stage[SAVE ? kWavesPerBlock * kStageHalves : 1];   // training: per-wave transpose tile of the activation dump
    // ENC 3: per-wave image of the tile's 64 feature rows (8 KB), so that they leave as whole lines — where LDS has the 64 KB (not beside the weights of four hidden layers)
    constexpr bool ROWS_LDS = ENC == 3 && (kBlocks + kWavesPerBlock * 8) * 1024 <= 160 * 1024;
    __shared__ half8 s_rows[ROWS_LDS ? kWavesPerBlock * 512 : 1];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    half_t *stage = s_stage + (SAVE ? wave * kStageHalves : 0);

    int64_t n = args.io.n;
    if (MODE == 2) { n = *args.io.n_dev; if (n > args.io.n_cap) n = args.io.n_cap; }
    if ((MODE == 0 || MODE == 1) && args.io.n_dev64) {   // the count lives on the device (train step): one scalar load
        typedef const int64_t __attribute__((address_space(4))) *CntPtr;
        const int64_t nd = *(CntPtr)(uintptr_t)args.io.n_dev64;
        n = nd < n ? nd : n;
    }
    const int64_t n_tiles_all = (n + kWaveSamples - 1) / kWaveSamples;
    if (n_tiles_all == 0) return;
    int64_t tile0 = 0, n_tiles = n_tiles_all;
    if (ENC == 1 && args.io.n_chunks > 1) { tile0 = n_tiles_all * args.io.chunk / args.io.n_chunks; n_tiles = n_tiles_all * (args.io.chunk + 1) / args.io.n_chunks; }
    const int wpb = MODE == 3 ? kWavesPerBlock : args.active_waves;
    const int64_t n_groups = (n_tiles - tile0 + wpb - 1) / wpb;
    int64_t g_first = 0, g_end = 1, g_step = 1;
    if (MODE != 3) group_range(n_groups, g_first, g_end, g_step);
    if (g_first >= g_end) return;   // uniform per block: nothing to do

    {   // weight fragments -> LDS: every load of a lane is issued before the first LDS write (a plain copy loop waited for each
        // 16-byte load in turn: ~11 serial L2 round trips per launch, which matters in the late render rounds of few tiles)
        constexpr int kPer = (kBlocks * 64 + kThreads - 1) / kThreads;
        half8 tmp[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x + j * kThreads;
            if (i < kBlocks * 64) tmp[j] = args.frags[i];
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x + j * kThreads;
            if (i < kBlocks * 64) s_w[i] = tmp[j];
        }
    }
    __syncthreads();
    if (wave >= wpb) return;

    typedef const KernelArgs __attribute__((address_space(4))) KArgs;
    KArgs *const kp = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    WaveCounters wc;
    ColData cd_next = {-1, 64, 0, 0.f, 0.f};
    // MODE 3 (ray-major density pass, `mnf_field_density_rays`): a wave takes whole rays (dynamically, one atomic per ray)
    // and walks a ray's samples front to back in 64-sample steps; once the optical depth accumulated so far makes every
    // later sample invisible (transmittance below early_stop_eps / 2) the rest of the ray is skipped — its densities stay
    // at the zeros the entry point filled in, which leaves the visibility mask (volrend.py:424-483) unchanged.
    int64_t ray_start = 0, ray_cnt = 0, ray_base = 0;
    float ray_sdt = 0.0f;
    // (the render rounds only: on the train step's forward — uniform tiles, two per wave at the reference's batch size — the per-tile ticket cost 0.110 -> 0.129 ms)
#ifndef MNF_TICKET_MIN_TILES
#define MNF_TICKET_MIN_TILES 16384
#endif
    // launches of fewer than eight tiles per wave keep the fixed stride: little to balance, and every wave's eight refused tickets at the end of a launch (one atomic
    // per range, the same eight addresses for all waves) weigh more the shorter the launch is.  Measured (256-view scoring pass / 32 views / 800x800 x4 / one 800x800 view):
    // always tickets 99.8 / 22.4 / 64.7 / 19.7 ms; from 4096 tiles 98.2 / 21.3 / 64.6 / 19.7; from 16384 96.4 / 21.4 / 64.0 / 19.3; from 32768 99.0 / 21.3 / 63.4 / 19.6; from 65536 99.4 / 21.4 / 64.9 / 19.8
    const bool dyn = MODE == 2 && ENC == 0 && args.io.tickets != nullptr && n_tiles >= MNF_TICKET_MIN_TILES;
    int dyn_x = blockIdx.x & 7, dyn_seen = 0, dyn_next = -1;
    if (dyn) dyn_next = ticket_take(args.io.tickets, dyn_x, dyn_seen, n_tiles, lane);
    for (int64_t grp = g_first; MODE == 3 || dyn || grp < g_end; grp += g_step) {
        // Everything the tile reads from the kernel arguments is re-read from the kernel-argument segment (scalar loads, scalar cache) where it is used: kept in SGPRs for
        // the whole launch these ~60 dwords pushed the kernel over the 102 SGPRs (spills to VGPR lanes: ~220 v_readlane with their wait states per tile, and to scratch).
        KArgs *lp = kp;
        asm volatile("" : "+s"(lp));
        KArgs &la = *lp;
        int64_t tile = tile0 + grp * wpb + wave, col, n_eff = n;
        if (MODE == 3) {
            if (ray_base >= ray_cnt || ray_sdt > la.io.sdt_stop) {
                int r = 0;
                if (lane == 0) r = atomicAdd(la.io.ray_counter, 1);
                r = __builtin_amdgcn_readfirstlane(r);
                if (r >= la.io.n_rays) break;
                ray_start = la.io.chunk_starts[r]; ray_cnt = la.io.chunk_cnts[r]; ray_base = 0; ray_sdt = 0.0f;
                if (ray_cnt == 0) continue;
            }
            col = ray_start + ray_base + lane; n_eff = ray_start + ray_cnt; ray_base += kWaveSamples;
            tile = 0;
        } else {
            if (MODE == 2 && dyn) tile = dyn_next;
            if (tile < 0 || tile >= n_tiles) break;
            col = tile * kWaveSamples + lane;
        }
        uint32_t dyn_pf = 0;
        if (MODE == 2 && dyn && lane == 0) dyn_pf = atomicAdd(la.io.tickets + kTicketStride * dyn_x, 1u);   // the ticket of the tile after this one: back long before the gathers are
        // ---- this lane's sample ----
        ColData cd = {-1, 64, 0, 0.f, 0.f};
        if (MODE == 2) {
            cd = load_cols(args, tile, lane);
        }
        float xn[3], dir[3];
        TileSample tsm;
        bool valid, selector;
        fetch_sample<MODE, !DENSITY_ONLY>(args, cd, col, n_eff, xn, dir, tsm, valid, selector);
        const LevelsPtr lv = levels_here(la.levels);
        const bool in_box = __ballot(valid && !selector) == 0ull;   // wave-uniform: the cheap dense-level wrap applies

        // ---- hash encode: all 16 levels of this lane's sample (one k-step = 4 levels = 32 gathers in flight),
        //      then trade halves with lane^32 ----
        half8 bfeat[CT][4];
        if (ENC == 2) {
            // this lane's sample: its row of 64 features as the pre-pass encoded them, eight 16-byte pieces of one line; then the usual trade with lane ^ 32
            const bool have = valid;
            const half8 *src = reinterpret_cast<const half8 *>(la.io.rows_in) + (have ? la.io.rows_src[col] : 0) * 8;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                half8 lo = src[2 * kb], hi = src[2 * kb + 1];
                if (!have) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)0.0f; hi[j] = (half_t)0.0f; }
                }
                exchange_halves(lo, hi);
                bfeat[0][kb] = lo; bfeat[1][kb] = hi;
            }
        } else if (ENC == 1) {
            // features were produced by encode_kernel, already in fragment order
            const half8 *src = reinterpret_cast<const half8 *>(la.io.enc) + tile * 512 + lane;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bfeat[ct][ks] = src[(ks * 2 + ct) * 64];
        }
        else {
            // double-buffered: the loads of batch kb+1 are issued before batch kb is blended
            LevelPrep prep[2][4];
            tab4 v[2][4][8];
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hash_prep(level_meta(lv, q), xn, prep[0][q], in_box);
                hash_load(la.table, prep[0][q], v[0][q]);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int cur = kb & 1, nxt = cur ^ 1;
                if (kb < 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        hash_prep(level_meta(lv, 4 * (kb + 1) + q), xn, prep[nxt][q], in_box);
                        hash_load(la.table, prep[nxt][q], v[nxt][q]);
                    }
                }
                half8 lo, hi;
                if (B16) {              // tcnn's fp16 blend: the sums come out as the packed 16-bit features themselves
                    u32x2 r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = hash_blend16(prep[cur][q], v[cur][q]);
#ifdef MNF_BF16
                    float f[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const h16x2 p0 = __builtin_bit_cast(h16x2, (uint32_t)r[q].x), p1 = __builtin_bit_cast(h16x2, (uint32_t)r[q].y);
                        f[4 * q] = (float)p0.x; f[4 * q + 1] = (float)p0.y; f[4 * q + 2] = (float)p1.x; f[4 * q + 3] = (float)p1.y;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)f[j]; hi[j] = (half_t)f[8 + j]; }
#else
                    lo = __builtin_bit_cast(half8, u32x4{r[0].x, r[0].y, r[1].x, r[1].y});
                    hi = __builtin_bit_cast(half8, u32x4{r[2].x, r[2].y, r[3].x, r[3].y});
#endif
                } else {
                    float f[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) hash_blend(prep[cur][q], v[cur][q], f + 4 * q);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)f[j]; hi[j] = (half_t)f[8 + j]; }
                }
                if (ENC == 3) {      // the sample's features 16 kb .. 16 kb + 15, as the MLP is about to see them: pieces 2 kb, 2 kb + 1 of its row (swizzled: a row is 128 bytes)
                    if (ROWS_LDS) {
                        half8 *img = s_rows + wave * 512 + lane * 8;
                        img[(2 * kb) ^ (lane & 7)] = lo; img[(2 * kb + 1) ^ (lane & 7)] = hi;
                    } else if (valid) {
                        half8 *dst = reinterpret_cast<half8 *>(la.io.rows_out) + col * 8 + 2 * kb;
                        dst[0] = lo; dst[1] = hi;
                    }
                }
                exchange_halves(lo, hi);
                bfeat[0][kb] = lo; bfeat[1][kb] = hi;
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(1);
            if (ROWS_LDS) {
                // the tile's rows are consecutive in rows_out (the lanes' samples are consecutive): eight stores of 1 KB each instead of 16-byte pieces at a 128-byte lane
                // stride (the pre-pass took 453 us with those, 399 without rows)
                half8 *dst = reinterpret_cast<half8 *>(la.io.rows_out) + (col - lane) * 8;
                const half8 *img = s_rows + wave * 512;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int piece = i * 64 + lane, row = piece >> 3;
                    const half8 v = img[row * 8 + ((piece & 7) ^ (row & 7))];
                    if (col - lane + row < n_eff) dst[piece] = v;
                }
            }
        }

        if (MODE == 2 && dyn) {      // the next tile: the ticket asked for at the top of this one has arrived behind the tile's feature loads
            dyn_next = ticket_tile(__builtin_amdgcn_readfirstlane(dyn_pf), dyn_x, n_tiles);
            if (dyn_next < 0) { dyn_x = (dyn_x + 1) & 7; ++dyn_seen; dyn_next = ticket_take(la.io.tickets, dyn_x, dyn_seen, n_tiles, lane); }
        }

        // the mask-dump base of this tile
        uint8_t *mdump = SAVE ? la.train.masks + (tile * 64 + lane) * T::mask_bytes : nullptr;      // this lane's mask record of the tile
        u32x4 mpiece = {0u, 0u, 0u, 0u};
        constexpr int kMaskUsed = T::mask_blocks * CT;
        if (SAVE) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) save_pair<false>(from_constant(&la.train), tile, T::rX + 16 * ks, lane, stage, bfeat[0][ks], bfeat[1][ks]);
        }

        // ---- base MLP ----
        half8 hb[CT][L::KSW];
        dense_relu<L::RT, 4>(s_w + L::o_b_in * 64, lane, bfeat, hb);
        if (SAVE) {
#pragma unroll
            for (int k = 0; k < L::KSW; ++k) {
                save_pair<true>(from_constant(&la.train), tile, T::rH0 + 16 * k, lane, stage, hb[0][k], hb[1][k]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) mask_put<kMaskUsed>(mpiece, mdump, (T::mH0 + k) * CT + ct, frag_mask(hb[ct][k]));
            }
        }
#pragma unroll
        for (int l = 0; l < NH - 1; ++l) {
            half8 hn[CT][L::KSW];
            dense_relu<L::RT, L::KSW>(s_w + (L::o_b_hid + l * L::RT * L::KSW) * 64, lane, hb, hn);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int k = 0; k < L::KSW; ++k) hb[ct][k] = hn[ct][k];
            if (SAVE) {
#pragma unroll
                for (int k = 0; k < L::KSW; ++k) {
                    save_pair<true>(from_constant(&la.train), tile, T::rH0 + (l + 1) * W + 16 * k, lane, stage, hb[0][k], hb[1][k]);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) mask_put<kMaskUsed>(mpiece, mdump, (T::mH0 + (l + 1) * L::KSW + k) * CT + ct, frag_mask(hb[ct][k]));
                }
            }
        }
        f32x16 bo[CT];
        dense_out<L::KSW>(s_w + L::o_b_out * 64, lane, hb, bo);
        if (la.out_fp16) round_outputs_fp16(bo);     // tcnn hands its network outputs over in fp16 (ngp.py:181-200 casts them back)

        // Results come back in MFMA layout: lane (c, h) holds rows {8g + 4h + i} of column c of tile ct.
        // The density logit (row 0) of this lane's OWN sample sits in lane (lane&31) register 0 of tile h.
        const float logit_t0 = __shfl(bo[0][0], lane & 31, 64);
        const float logit_t1 = __shfl(bo[1][0], lane & 31, 64);
        const float sigma = selector ? expf((h ? logit_t1 : logit_t0) - 1.0f) : 0.0f;   // ngp.py:79, :193-195

        if (DENSITY_ONLY) {
            if (col < n_eff && la.io.density) la.io.density[col] = sigma;
            if (MODE == 3) {
                float sdt = valid ? sigma * (tsm.te - tsm.ts) : 0.0f;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) sdt += __shfl_xor(sdt, d, 64);
                ray_sdt += sdt;
            }
            continue;
        }

        // ---- heads ----
        half8 bgeo[CT][1];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bgeo[ct][0][j] = (half_t)bo[ct][j];
            if (h == 0) bgeo[ct][0][0] = (half_t)1.0f;   // tcnn pads MLP inputs with 1.0 (row 0 = density logit slot)
        }
        half8 hin[CT][2];
        {
            half8 lo, hi;
            sh4(dir, lo, hi);
            exchange_halves(lo, hi);
            hin[0][0] = lo; hin[1][0] = hi;
            hin[0][1] = bgeo[0][0]; hin[1][1] = bgeo[1][0];
        }
        if (SAVE) {
            save_pair<false>(from_constant(&la.train), tile, T::rS, lane, stage, hin[0][0], hin[1][0]);
            save_pair<true>(from_constant(&la.train), tile, T::rG, lane, stage, hin[0][1], hin[1][1]);
        }
        half8 h1[CT][L::KSh], h2[CT][L::KSh];
        f32x16 out_rgb[CT], out_sem[CT];
        auto save_hidden = [&](const half8 (&a)[CT][L::KSh], int row0, int mblk) {
#pragma unroll
            for (int k = 0; k < L::KSh; ++k) {
                save_pair<true>(from_constant(&la.train), tile, row0 + 16 * k, lane, stage, a[0][k], a[1][k]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) mask_put<kMaskUsed>(mpiece, mdump, (mblk + k) * CT + ct, frag_mask(a[ct][k]));
            }
        };
        // rgb head (ngp.py:143-156, :202-213)
        dense_relu<L::RTh, 2>(s_w + L::o_h_in * 64, lane, hin, h1);
        if (SAVE) save_hidden(h1, T::rHH1, T::mHH1);
        dense_relu<L::RTh, L::KSh>(s_w + L::o_h_hid * 64, lane, h1, h2);
        if (SAVE) save_hidden(h2, T::rHH2, T::mHH2);
        dense_out<L::KSh>(s_w + L::o_h_out * 64, lane, h2, out_rgb);
        if (la.out_fp16) round_outputs_fp16(out_rgb);
        // semantic head (ngp.py:158-169, :215-220)
        dense_relu<L::RTh, 1>(s_w + L::o_s_in * 64, lane, bgeo, h1);
        if (SAVE) save_hidden(h1, T::rHS1, T::mHS1);
        dense_relu<L::RTh, L::KSh>(s_w + L::o_s_hid * 64, lane, h1, h2);
        if (SAVE) save_hidden(h2, T::rHS2, T::mHS2);
        dense_out<L::KSh>(s_w + L::o_s_out * 64, lane, h2, out_sem);
        if (la.out_fp16) round_outputs_fp16(out_sem);

        // ---- write out ----
        // rgb rows 0..2 of this lane's own sample: lane (lane&31), registers 0..2 of tile h
        float rgb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t0 = __shfl(out_rgb[0][k], lane & 31, 64);
            const float t1 = __shfl(out_rgb[1][k], lane & 31, 64);
            rgb[k] = 1.0f / (1.0f + expf(-(h ? t1 : t0)));   // ngp.py:211-212
        }
        if (MODE == 2) {
            fused_composite(fr_of_kernarg(), in_loop(la.C), in_loop_v(lane), tsm, sigma, rgb, out_sem, wc);
            continue;
        }
        if (col < n) {
            if (la.io.density) la.io.density[col] = sigma;
            if (la.io.rgb) { la.io.rgb[3 * col] = rgb[0]; la.io.rgb[3 * col + 1] = rgb[1]; la.io.rgb[3 * col + 2] = rgb[2]; }
        }
        // semantic logits: lane (c, h) writes rows 8g + 4h + i of column c for both tiles
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int64_t scol = tile * kWaveSamples + ct * 32 + (lane & 31);
            if (la.io.sem && scol < n) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 8 * g + 4 * h + i;
                        if (row < la.C) la.io.sem[la.io.sem_stride ? row * la.io.sem_stride + scol : scol * la.C + row] = out_sem[ct][4 * g + i];
                    }
            }
        }
    }
    if (MODE == 2) flush_counters(fr_of_kernarg(), wc, lane);
}

// ------------------------------------------------------------------ parameter preparation kernels
__global__ void __launch_bounds__(256) table_to_half_kernel(const float *__restrict__ src, tab_t *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) dst[i] = (tab_t)src[i];
}

__global__ void __launch_bounds__(256) gather_frags_kernel(const int32_t *__restrict__ src_idx, const float *__restrict__ p0,
                                                           const float *__restrict__ p1, const float *__restrict__ p2,
                                                           half_t *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = src_idx[i];
    float v = 0.0f;
    if (s >= 0) {
        const int buf = s >> 28;
        const int idx = s & 0x0FFFFFFF;
        v = buf == 0 ? p0[idx] : (buf == 1 ? p1[idx] : p2[idx]);
    }
    dst[i] = (half_t)v;
}

// ------------------------------------------------------------------ host side
static int grid_levels(const mnf_field_config &cfg, LevelMeta *levels, int64_t *total) {
    const double pls = std::exp((std::log((double)cfg.max_resolution) - std::log((double)cfg.base_resolution)) / (cfg.n_levels - 1));
    const double log2_pls = std::log2(pls);
    int64_t offset = 0;
    for (int l = 0; l < cfg.n_levels; ++l) {
        const float scale = (float)(std::exp2(l * log2_pls) * cfg.base_resolution - 1.0);
        const uint32_t res = (uint32_t)std::ceil((double)scale) + 1u;
        const uint64_t dense = (uint64_t)res * res * res;
        uint64_t n = ((dense + 7) / 8) * 8;
        const uint64_t cap = 1ull << cfg.log2_hashmap_size;
        if (n > cap) n = cap;
        levels[l].scale = scale;
        levels[l].res = res;
        levels[l].size = (uint32_t)n;
        levels[l].offset = (uint32_t)offset;
        levels[l].hashed = dense > n ? 1u : 0u;
        {   // round-up magic number for an exact, branch-free unsigned division by n (33-bit multiplier form)
            const uint32_t d = (uint32_t)n;
            uint32_t fl = 31; while (!((d >> fl) & 1u)) --fl;
            if ((d & (d - 1)) == 0) { levels[l].div_magic = 0; levels[l].div_shift = fl - 1; }
            else {
                const uint64_t num = 1ull << (32 + fl);
                uint32_t pm = (uint32_t)(num / d);
                const uint64_t rem = num % d;
                pm += pm;
                const uint64_t twice = rem + rem;
                if (twice >= d) pm += 1;
                levels[l].div_magic = pm + 1; levels[l].div_shift = fl;
            }
        }
        offset += (int64_t)n;
    }
    *total = offset;
    return MNF_OK;
}

enum KMap { K_NATURAL, K_ACC, K_GEO };

// Append the fragment blocks of one [n_out_real x n_in_real] matrix (row-major [out][in], at `off` in buffer `buf`).
static void append_matrix(std::vector<int32_t> &t, int buf, int64_t off, int n_out_real, int n_in_real, int row_tiles,
                          const std::vector<std::pair<KMap, int>> &ksteps, int geo_base, int pad_col) {
    for (int rt = 0; rt < row_tiles; ++rt)
        for (size_t ks = 0; ks < ksteps.size(); ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r = lane & 31, h = lane >> 5;
                    const int n = 32 * rt + r;
                    int k = -1;
                    const KMap km = ksteps[ks].first;
                    const int kb = ksteps[ks].second;
                    if (km == K_NATURAL) k = kb + 8 * h + j;
                    else if (km == K_ACC) k = kb + 8 * (j >> 2) + 4 * h + (j & 3);
                    else {
                        const int row = 8 * (j >> 2) + 4 * h + (j & 3);
                        k = row == 0 ? pad_col : geo_base + row - 1;
                    }
                    int32_t v = -1;
                    if (n < n_out_real && k >= 0 && k < n_in_real) v = (int32_t)((buf << 28) | (int32_t)(off + (int64_t)n * n_in_real + k));
                    t.push_back(v);
                }
}

static std::vector<std::pair<KMap, int>> acc_ksteps(int width) {
    std::vector<std::pair<KMap, int>> v;
    for (int k = 0; k < width; k += 16) v.push_back({K_ACC, (k / 32) * 32 + ((k / 16) & 1) * 16});
    return v;
}

static std::vector<int32_t> build_frag_table(const mnf_field_config &cfg) {
    const int W = cfg.neurons, Wh = W / 2, NH = cfg.layers;
    const int sem_pad = ((cfg.num_semantic_classes + 15) / 16) * 16;
    std::vector<int32_t> t;
    int64_t off = 0;
    // mlp_base: [W][64], (NH-1) x [W][W], [16][W]
    std::vector<std::pair<KMap, int>> nat64;
    for (int k = 0; k < 64; k += 16) nat64.push_back({K_NATURAL, k});
    append_matrix(t, 0, off, W, 64, W / 32, nat64, 0, 0); off += (int64_t)W * 64;
    for (int l = 0; l < NH - 1; ++l) { append_matrix(t, 0, off, W, W, W / 32, acc_ksteps(W), 0, 0); off += (int64_t)W * W; }
    append_matrix(t, 0, off, 16, W, 1, acc_ksteps(W), 0, 0); off += 16 * (int64_t)W;
    // mlp_head: [Wh][32] (cols 0..15 SH, 16..30 geo, 31 pad), [Wh][Wh], [16][Wh]
    off = 0;
    append_matrix(t, 1, off, Wh, 32, Wh / 32, {{K_NATURAL, 0}, {K_GEO, 0}}, 16, 31); off += (int64_t)Wh * 32;
    append_matrix(t, 1, off, Wh, Wh, Wh / 32, acc_ksteps(Wh), 0, 0); off += (int64_t)Wh * Wh;
    append_matrix(t, 1, off, 16, Wh, 1, acc_ksteps(Wh), 0, 0);
    // mlp_sem: [Wh][16] (cols 0..14 geo, 15 pad), [Wh][Wh], [sem_pad][Wh]
    off = 0;
    append_matrix(t, 2, off, Wh, 16, Wh / 32, {{K_GEO, 0}}, 0, 15); off += (int64_t)Wh * 16;
    append_matrix(t, 2, off, Wh, Wh, Wh / 32, acc_ksteps(Wh), 0, 0); off += (int64_t)Wh * Wh;
    append_matrix(t, 2, off, sem_pad, Wh, 1, acc_ksteps(Wh), 0, 0);
    return t;
}

template <int W, int NH>
static int launch_variant(mnf_field_t f, const FieldIO &io, bool density_only, int grid, hipStream_t stream, const TrainBuf *train) {
    KernelArgs a;
    a.train = train ? *train : TrainBuf{nullptr, nullptr, 0, 0};
    a.table = reinterpret_cast<const tab4 *>(f->d_table);
    a.frags = reinterpret_cast<const half8 *>(f->d_frags);
    std::memcpy(a.aabb, f->cfg.aabb, sizeof(a.aabb));
    a.C = f->cfg.num_semantic_classes;
    a.out_fp16 = f->cfg.output_fp16 ? 1 : 0;
    a.blend16 = f->cfg.blend_fp16 ? 1 : 0;
    static const int active_waves = diag_env("MNF_FIELD_ACTIVE_WAVES") ? atoi(diag_env("MNF_FIELD_ACTIVE_WAVES")) : kWavesPerBlock;
    a.active_waves = active_waves >= 1 && active_waves <= kWavesPerBlock ? active_waves : kWavesPerBlock;
    a.levels = reinterpret_cast<const LevelMeta *>(reinterpret_cast<const char *>(f->d_frags) + (size_t)f->shape.blocks_total * 1024);
    a.io = io;
    // the fp16-blend instantiations exist for neurons = 128 (the reference yamls) only: they double the kernels of a shape
    const bool b16 = f->cfg.blend_fp16 != 0;
    if (b16 && (W != 128 || io.enc)) { set_error("field: blend_fp16 is built for neurons = 128 (and not for the two-launch diagnostic path)"); return MNF_ERR_UNSUPPORTED; }
#define MNF_LAUNCH_S(MODE, DO, SAVEK)                                                                                                                  \
    do {                                                                                                                                               \
        if constexpr (W == 128) {                                                                                                                      \
            if (b16) { hipLaunchKernelGGL((field_kernel<W, NH, MODE, DO, SAVEK, 0, true>), dim3(grid), dim3(kThreads), 0, stream, a); break; }   \
        }                                                                                                                                              \
        hipLaunchKernelGGL((field_kernel<W, NH, MODE, DO, SAVEK, 0, false>), dim3(grid), dim3(kThreads), 0, stream, a);                           \
    } while (0)
#define MNF_LAUNCH(MODE, DO) MNF_LAUNCH_S(MODE, DO, 0)
    // the feature rows of FieldIO (pre-pass writes, training forward reads): instantiated for neurons = 128 with the fp32 blend (field_rows_supported)
    const bool rows_ok = W == 128 && !b16 && !io.enc;
    if ((io.rows_in || io.rows_out) && !rows_ok) { set_error("field: feature rows are built for neurons = 128 with the fp32 blend"); return MNF_ERR_UNSUPPORTED; }
    if (train && io.rows_in && io.mode == 1) {
        if constexpr (W == 128) hipLaunchKernelGGL((field_kernel<W, NH, 1, false, 1, 2, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    } else if (density_only && io.mode == 3 && io.rows_out) {
        if constexpr (W == 128) hipLaunchKernelGGL((field_kernel<W, NH, 3, true, 0, 3, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    } else if (train) {
        if (io.mode == 1) MNF_LAUNCH_S(1, false, 1); else MNF_LAUNCH_S(0, false, 1);
    } else if (io.enc) {
        // two launches: gather at high occupancy, then the register-heavy MLP chain on ready-made fragments
        half8 *enc = reinterpret_cast<half8 *>(const_cast<void *>(io.enc));
        const int egrid = io.gather_grid > 0 ? io.gather_grid : 2048;
        if (io.phase != 2) {
            if (io.mode == 0) hipLaunchKernelGGL((encode_kernel<0>), dim3(egrid), dim3(kEncodeThreads), 0, stream, a, enc);
            else if (io.mode == 1) hipLaunchKernelGGL((encode_kernel<1>), dim3(egrid), dim3(kEncodeThreads), 0, stream, a, enc);
            else hipLaunchKernelGGL((encode_kernel<2>), dim3(egrid), dim3(kEncodeThreads), 0, stream, a, enc);
        }
        if (io.mlp_waves > 0 && io.mlp_waves <= kWavesPerBlock) a.active_waves = io.mlp_waves;
        if (io.phase == 1) {
        } else if (density_only) {
            if (io.mode == 0) hipLaunchKernelGGL((field_kernel<W, NH, 0, true, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
            else if (io.mode == 1) hipLaunchKernelGGL((field_kernel<W, NH, 1, true, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
            else hipLaunchKernelGGL((field_kernel<W, NH, 2, true, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
        } else {
            if (io.mode == 0) hipLaunchKernelGGL((field_kernel<W, NH, 0, false, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
            else if (io.mode == 1) hipLaunchKernelGGL((field_kernel<W, NH, 1, false, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
            else hipLaunchKernelGGL((field_kernel<W, NH, 2, false, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
        }
    } else if (density_only) {
        if (io.mode == 0) MNF_LAUNCH(0, true); else if (io.mode == 1) MNF_LAUNCH(1, true); else if (io.mode == 3) MNF_LAUNCH(3, true); else MNF_LAUNCH(2, true);
    } else {
        if (io.mode == 0) MNF_LAUNCH(0, false); else if (io.mode == 1) MNF_LAUNCH(1, false); else MNF_LAUNCH(2, false);
    }
#undef MNF_LAUNCH
#undef MNF_LAUNCH_S
    return launch_status("field_kernel");
}

int launch_field_impl(mnf_field_t f, const FieldIO &io, bool density_only, hipStream_t stream, const TrainBuf *train) {
    MNF_REQUIRE(f && f->params_loaded, "field: parameters not loaded (call mnf_field_set_params first)");
    int grid = 256;  // one persistent workgroup per CU (LDS-limited), grid-stride over 64-sample tiles
    if (io.mode == 2 && io.grid_limit > 0 && io.grid_limit < grid) grid = io.grid_limit;
    if (io.mode != 2 && io.mode != 3) {
        const int64_t tiles = ceil_div(io.n, kWaveSamples);
        if (tiles == 0) return MNF_OK;
        const int64_t wgs = ceil_div(tiles, kWavesPerBlock);
        if (wgs < grid) grid = (int)wgs;
    }
    const int W = f->cfg.neurons, NH = f->cfg.layers;
    ProfScope ps(io.mode == 2 ? nullptr : (train ? "field_train_forward" : (density_only ? "field_density" : "field_forward")), stream);   // mode 2: timed by its caller
#define MNF_CASE(w, nh) if (W == w && NH == nh) return launch_variant<w, nh>(f, io, density_only, grid, stream, train)
#ifdef MNF_DEV_ONLY_128x2
    MNF_CASE(128, 2);
#else
    MNF_CASE(128, 1); MNF_CASE(128, 2); MNF_CASE(128, 3); MNF_CASE(128, 4);
    MNF_CASE(64, 1); MNF_CASE(64, 2); MNF_CASE(64, 3); MNF_CASE(64, 4);
#endif
#undef MNF_CASE
    set_error("field: unsupported neurons=%d layers=%d (supported: 64|128 x 1..4)", W, NH);
    return MNF_ERR_UNSUPPORTED;
}

int set_params_impl(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem, bool table_current, hipStream_t s) {
    const int64_t n_tab = f->table_entries * 4;
    int rc = MNF_OK;
    if (!table_current) {      // the optimizer kernel may already have written the fp16 table (mnf_adam_step_guarded's mirror)
        hipLaunchKernelGGL(table_to_half_kernel, dim3(2048), dim3(256), 0, s, mlp_base + f->n_base_mlp, (tab_t *)f->d_table, n_tab);
        rc = launch_status("table_to_half_kernel");
        if (rc) return rc;
    }
    const int64_t n_frag = (int64_t)f->shape.blocks_total * 512;
    hipLaunchKernelGGL(gather_frags_kernel, dim3((unsigned)ceil_div(n_frag, 256)), dim3(256), 0, s, f->d_frag_src, mlp_base, mlp_head,
                       mlp_sem, (half_t *)f->d_frags, n_frag);
    rc = launch_status("gather_frags_kernel");
    if (rc) return rc;
    f->params_loaded = true;   // the handle keeps no pointer into the caller's vectors: everything it needs later is in d_table / d_frags
    return MNF_OK;
}

MNF_DT_END

#ifndef MNF_BF16   // ---- everything below is operand-type independent and compiled once
namespace mnf {
int launch_field(mnf_field_t f, const FieldIO &io, bool density_only, hipStream_t stream, const TrainBuf *train) {
    MNF_REQUIRE(f, "field: null handle");
    return f->cfg.mfma_bf16 ? bf16::launch_field_impl(f, io, density_only, stream, train) : f16::launch_field_impl(f, io, density_only, stream, train);
}
}  // namespace mnf

using namespace mnf;
using namespace mnf::f16;    // host-side table builders (identical in both translation units)

extern "C" int mnf_field_set_params(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem, mnf_stream_t stream) {
    MNF_REQUIRE(f && mlp_base && mlp_head && mlp_sem, "field_set_params: null argument");
    return f->cfg.mfma_bf16 ? bf16::set_params_impl(f, mlp_base, mlp_head, mlp_sem, false, as_stream(stream))
                            : f16::set_params_impl(f, mlp_base, mlp_head, mlp_sem, false, as_stream(stream));
}

extern "C" int mnf_field_refresh_weights(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem, mnf_stream_t stream) {
    MNF_REQUIRE(f && mlp_base && mlp_head && mlp_sem, "field_refresh_weights: null argument");
    MNF_REQUIRE(f->params_loaded, "field_refresh_weights: parameters were never loaded (the fp16 table is not current)");
    return f->cfg.mfma_bf16 ? bf16::set_params_impl(f, mlp_base, mlp_head, mlp_sem, true, as_stream(stream))
                            : f16::set_params_impl(f, mlp_base, mlp_head, mlp_sem, true, as_stream(stream));
}

extern "C" void *mnf_field_table_mirror(mnf_field_t f, int64_t *first_param_host) {
    if (!f) return nullptr;
    if (first_param_host) *first_param_host = f->n_base_mlp;
    return f->d_table;
}

extern "C" int mnf_field_create(const mnf_field_config *cfg, mnf_field_t *out) {
    MNF_REQUIRE(cfg && out, "field_create: null argument");
    MNF_REQUIRE(cfg->struct_size == sizeof(mnf_field_config), "field_create: cfg->struct_size is %u, this library's mnf_field_config has %zu bytes (MNF_INIT)",
                cfg->struct_size, sizeof(mnf_field_config));
    MNF_REQUIRE(cfg->neurons == 64 || cfg->neurons == 128, "field_create: neurons must be 64 or 128 (got %d)", cfg->neurons);
    MNF_REQUIRE(cfg->layers >= 1 && cfg->layers <= 4, "field_create: layers must be 1..4 (got %d)", cfg->layers);
    MNF_REQUIRE(cfg->num_semantic_classes >= 1 && cfg->num_semantic_classes <= 32,
                "field_create: num_semantic_classes must be 1..32 (got %d)", cfg->num_semantic_classes);
    MNF_REQUIRE(cfg->n_levels == 16 && cfg->n_features == 4, "field_create: only 16 levels x 4 features are supported");
    MNF_REQUIRE(cfg->log2_hashmap_size >= 8 && cfg->log2_hashmap_size <= 24, "field_create: bad log2_hashmap_size");
    MNF_REQUIRE(!cfg->blend_fp16 || cfg->neurons == 128, "field_create: blend_fp16 is built for neurons = 128 only");
    mnf_field_s *f = new mnf_field_s();
    f->cfg = *cfg;
    grid_levels(*cfg, f->levels, &f->table_entries);
    const int W = cfg->neurons, Wh = W / 2, NH = cfg->layers;
    const int sem_pad = ((cfg->num_semantic_classes + 15) / 16) * 16;
    f->n_base_mlp = (int64_t)W * 64 + (int64_t)(NH - 1) * W * W + 16 * (int64_t)W;
    f->n_base = f->n_base_mlp + f->table_entries * 4;
    f->n_head = (int64_t)Wh * 32 + (int64_t)Wh * Wh + 16 * (int64_t)Wh;
    f->n_sem = (int64_t)Wh * 16 + (int64_t)Wh * Wh + (int64_t)sem_pad * Wh;
    std::vector<int32_t> table = build_frag_table(*cfg);
    f->shape = {W, NH, Wh, cfg->num_semantic_classes, (int)(table.size() / 512)};
    f->frag_src_host = table;
    f->d_table = nullptr; f->d_frags = nullptr; f->d_frag_src = nullptr; f->params_loaded = false; f->d_counter = nullptr;
    f->train_state = nullptr;
    hipError_t e = hipMalloc(&f->d_table, (size_t)f->table_entries * 4 * sizeof(uint16_t) + 64);   // (+ slack: a paired 16-byte gather may start at a level's last entry)
    if (e == hipSuccess) e = hipMalloc(&f->d_frags, table.size() * sizeof(uint16_t) + sizeof(LevelMeta) * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_frag_src, table.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(f->d_frag_src, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy((char *)f->d_frags + table.size() * sizeof(uint16_t), f->levels, sizeof(LevelMeta) * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("field_create: %s", hipGetErrorString(e));
        mnf_field_destroy(f);
        return MNF_ERR_HIP;
    }
    *out = f;
    return MNF_OK;
}

extern "C" int mnf_field_destroy(mnf_field_t f) {
    if (!f) return MNF_OK;
    free_train_state(f);
    if (f->d_table) (void)hipFree(f->d_table);
    if (f->d_frags) (void)hipFree(f->d_frags);
    if (f->d_frag_src) (void)hipFree(f->d_frag_src);
    if (f->d_counter) (void)hipFree(f->d_counter);
    delete f;
    return MNF_OK;
}

extern "C" int64_t mnf_field_param_count(mnf_field_t f, int32_t which) {
    if (!f) return -1;
    return which == 0 ? f->n_base : (which == 1 ? f->n_head : (which == 2 ? f->n_sem : -1));
}

extern "C" int mnf_field_grid_meta_host(mnf_field_t f, float *scale_host, int32_t *res_host, int32_t *size_host,
                                        int64_t *offset_host, int32_t *hashed_host) {
    MNF_REQUIRE(f, "grid_meta: null handle");
    for (int l = 0; l < f->cfg.n_levels; ++l) {
        if (scale_host) scale_host[l] = f->levels[l].scale;
        if (res_host) res_host[l] = (int32_t)f->levels[l].res;
        if (size_host) size_host[l] = (int32_t)f->levels[l].size;
        if (offset_host) offset_host[l] = (int64_t)f->levels[l].offset;
        if (hashed_host) hashed_host[l] = (int32_t)f->levels[l].hashed;
    }
    return MNF_OK;
}

extern "C" int mnf_field_forward(mnf_field_t f, const float *positions, const float *directions, int64_t n,
                                 float *rgb, float *density, float *sem, mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_forward: null handle");
    MNF_REQUIRE(n >= 0, "field_forward: negative n");
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(positions && directions, "field_forward: null positions/directions");
    FieldIO io = {};
    io.mode = 0; io.positions = positions; io.directions = directions; io.n = n;
    io.rgb = rgb; io.density = density; io.sem = sem;
    return launch_field(f, io, false, as_stream(stream));
}

extern "C" int mnf_field_density(mnf_field_t f, const float *positions, int64_t n, float *density, mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_density: null handle");
    MNF_REQUIRE(n >= 0, "field_density: negative n");
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(positions && density, "field_density: null pointer");
    FieldIO io = {};
    io.mode = 0; io.positions = positions; io.n = n; io.density = density;
    return launch_field(f, io, true, as_stream(stream));
}

extern "C" int mnf_field_forward_samples(mnf_field_t f, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                         const float *t_starts, const float *t_ends, int64_t n,
                                         float *rgb, float *density, float *sem, mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_forward_samples: null handle");
    MNF_REQUIRE(n >= 0, "field_forward_samples: negative n");
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends, "field_forward_samples: null pointer");
    FieldIO io = {};
    io.mode = 1; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = ray_indices; io.t_starts = t_starts; io.t_ends = t_ends; io.n = n;
    io.rgb = rgb; io.density = density; io.sem = sem;
    const bool density_only = (rgb == nullptr && sem == nullptr);
    return launch_field(f, io, density_only, as_stream(stream));
}

extern "C" int mnf_field_density_rays(mnf_field_t f, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                      const float *t_starts, const float *t_ends, const int64_t *chunk_starts, const int64_t *chunk_cnts,
                                      int32_t n_rays, int64_t n_samples, float early_stop_eps, float *density, mnf_stream_t stream) {
    MNF_REQUIRE(f, "field_density_rays: null handle");
    MNF_REQUIRE(n_rays >= 0 && n_samples >= 0, "field_density_rays: negative size");
    if (n_samples == 0 || n_rays == 0) return MNF_OK;
    MNF_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && chunk_starts && chunk_cnts && density, "field_density_rays: null pointer");
    hipStream_t s = as_stream(stream);
    if (!f->d_counter) MNF_HIP(hipMalloc((void **)&f->d_counter, 256));
    MNF_HIP(hipMemsetAsync(f->d_counter, 0, sizeof(int32_t), s));
    MNF_HIP(hipMemsetAsync(density, 0, (size_t)n_samples * sizeof(float), s));
    FieldIO io = {};
    io.mode = 3; io.rays_o = rays_o; io.rays_d = rays_d; io.ray_idx64 = ray_indices; io.t_starts = t_starts; io.t_ends = t_ends; io.n = n_samples;
    io.chunk_starts = chunk_starts; io.chunk_cnts = chunk_cnts; io.n_rays = n_rays; io.ray_counter = f->d_counter;
    // skip what lies behind T < eps / 2: the factor 2 keeps the decision clear of the rounding of any other summation order
    io.sdt_stop = early_stop_eps > 0.0f ? -logf(early_stop_eps) + 0.6931472f : INFINITY;
    io.density = density;
    return launch_field(f, io, true, s);
}
#endif  // MNF_BF16
