// View-queue renderer (csrc/viewq.hip): what the host side (render.hip) and the kernel share.
#pragma once
#include "field.h"
#include "march_dev.h"

namespace mnf {

constexpr int kVQWaves = 8;              // waves of a workgroup (= field_dev.h kWavesPerBlock: two per SIMD at 256 registers); every wave is a worker of its own
constexpr int kVQWaveTiles = 16;         // tiles (of 64 columns) one work item may fill: the wave's private column scratch
constexpr int kVQWaveCols = kVQWaveTiles * 64;
constexpr int kVQMaxGrid = 256;          // workgroups of a launch (one per CU: the weights of a field and the occupancy bits fill the LDS)
constexpr int kVQGridWords = 16384;      // occupancy bits staged in LDS: 64 KB = 524 288 cells
constexpr int kVQMaxRaysPerView = 16384; // views up to 128 x 128 rays go through the queue; larger ones through the per-round launches (render.hip)
constexpr int kVQMaxGroups = 8;          // distinct (field, occupancy grid) pairs of a call: an ensemble's members
constexpr int kVQItemBits = 11;          // item = view << 11 | index of the item inside the view's round (a round has at most rays_per_view / 16 + 1 <= 1025 items)
// control words of a job's queue (one 64-byte line each: they are the targets of device-scope atomics from every workgroup)
constexpr int kVQHead = 0, kVQTail = 16, kVQViewsLeft = 32, kVQJobDone = 48, kVQError = 64, kVQCtrlWords = 80;

// rays of one work item for a per-ray budget `ns` and `n_alive` rays of the view: whole tiles (64 / ns rays each); at most what one wave has lanes for and what
// fills its column scratch; and no more than spreads the view's alive rays over its share of the launch's waves (`waves_per_view`: with few views in the batch a
// round's latency is one item's, so items shrink to a single tile and every wave of the chip takes one)
__host__ __device__ inline int vq_rays_per_item(int ns, int n_alive, int waves_per_view) {
    const int cap = 64 / ns;
    int most = kVQWaveTiles * cap;
    if (most > 64) most = (64 / cap) * cap;
    int want = (n_alive + waves_per_view - 1) / waves_per_view;
    want = ((want + cap - 1) / cap) * cap;
    return want < cap ? cap : (want > most ? most : want);
}

// One render job (a field, its occupancy bits, its rays, its outputs) as the kernel sees it; lives in device memory (job 0's workspace), read with scalar loads.
struct VQJob {
    const void *table, *frags;           // fp16 hash table, fragment-ordered MLP weights (mnf_field_s::d_table / d_frags)
    const LevelMeta *levels;
    float aabb[6];                       // the field's box (ngp.py:177-178)
    int32_t C, out_fp16;
    const uint32_t *bitgrid;             // bit-packed occupancy grid of the job's estimator
    const float *rays_o, *rays_d;
    uint8_t *alive; const uint8_t *hit;
    float *near_plane; const float *t_min, *t_max;
    // per view: this round's per-ray budget, samples offered so far (utils.py:669-670), items of this round that have finished / that exist, rays per item,
    // the length of the view's list of alive rays; the list itself ([view][rays_per_view + 64] ray ids of the job, in the view's march order) and where
    // this round's items leave their survivors: item i at entries [i * rpi, i * rpi + segcnt[i]) of `next` (same shape), compacted into `list` by the
    // wave that finishes the round — in item order, so a view's lists (hence its tiles) never depend on timing or on what else is in the batch
    int32_t *n_samples, *iter_samples, *done, *n_items, *rpi, *cnt, *list, *next, *segcnt;
    int32_t seg_stride;                  // entries of segcnt per view (rays_per_view / 16 + 2: a round's items hold at least 16 rays each, the last one fewer)
    int32_t *alive_sink;                 // [views] where composite_dev.h's survivor counter goes (the lists' lengths are what the schedule reads)
    float *rgb, *acc, *depth, *sem, *rgb_var, *depth_var;
    unsigned long long *totals;
    unsigned long long *ring; int32_t ring_mask, n_views;       // the job's queue: slot (t & mask) holds (t + 1) << 32 | item for the t-th item ever pushed
    int32_t *ctrl;                                              // kVQCtrlWords control words
};

struct VQArgs {
    const VQJob *jobs; int32_t n_jobs, n_groups;
    int32_t group_job_end[kVQMaxGroups];  // jobs of group g (same weights, same occupancy bits): [group_job_end[g - 1], group_job_end[g])
    int32_t group_wg_end[kVQMaxGroups];   // its workgroups: [group_wg_end[g - 1], group_wg_end[g])
    int32_t rays_per_view, waves_per_view;   // waves_per_view: waves of the launch / views of the call (>= 1)
    int32_t max_samples, min_samples, probabilistic;
    float far_plane, step_size, cone_angle, alpha_thre, opc_thre;
    I3 res; int32_t n_words;
    float occ_aabb[6];                   // the occupancy level's box (estimator.aabbs[0])
    int32_t *col_ray; float *col_ts, *col_te;     // [workgroup][wave][kVQWaveCols] column scratch
    int32_t *error;                      // = jobs[0].ctrl + kVQError
    unsigned long long *stats;           // diag build (MNF_VQ_STATS=1): 16 counters summed over all waves — cycles popping / marching / in tiles / publishing / compacting, items, tiles
};

inline int64_t vq_scratch_bytes() { return (int64_t)kVQMaxGrid * kVQWaves * kVQWaveCols * 12; }

// dispatchers on the fields' operand type (viewq.hip, once per type)
namespace f16 { bool viewq_supported(int W, int NH); int launch_viewq_impl(const VQArgs &a, int W, int NH, int grid, hipStream_t s); }
namespace bf16 { bool viewq_supported(int W, int NH); int launch_viewq_impl(const VQArgs &a, int W, int NH, int grid, hipStream_t s); }

}  // namespace mnf
