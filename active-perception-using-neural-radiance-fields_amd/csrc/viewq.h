// View-queue renderer (csrc/viewq.hip): what the host side (render.hip) and the kernel share.
#pragma once
#include "field.h"
#include "march_dev.h"

namespace mnf {

constexpr int kVQWaves = 8;              // waves of a workgroup (= field_dev.h kWavesPerBlock: two per SIMD at 256 registers)
constexpr int kVQSlice = kVQWaves * 64;  // rays of one work item = threads of a workgroup (lane = ray while marching)
constexpr int kVQWaveTiles = 16;         // a wave marches as many of its rays at a time as fill this many 64-column tiles ...
constexpr int kVQWaveCols = kVQWaveTiles * 64;   // ... into its private column scratch
constexpr int kVQMaxGrid = 256;          // workgroups of a launch (one per CU: the weights of a field and the occupancy bits fill the LDS)
constexpr int kVQGridWords = 16384;      // occupancy bits staged in LDS: 64 KB = 524 288 cells
constexpr int kVQMaxRaysPerView = 16384; // views up to 128 x 128 rays go through the queue; larger ones through the per-round launches (render.hip)
// control words of a job's queue (one 64-byte line each: they are the targets of device-scope atomics from every workgroup)
constexpr int kVQHead = 0, kVQTail = 16, kVQViewsLeft = 32, kVQJobDone = 48, kVQError = 64, kVQCtrlWords = 80;

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
    int32_t *alive_count, *n_samples, *iter_samples, *done;      // per view
    float *rgb, *acc, *depth, *sem, *rgb_var, *depth_var;
    unsigned long long *totals;
    uint32_t *slots; int32_t slots_cap; int32_t n_views;          // the job's queue: slot h holds (unit + 1) of the h-th item pushed, 0 = not pushed yet
    int32_t *ctrl;                                                // kVQCtrlWords control words
};

struct VQArgs {
    const VQJob *jobs; int32_t n_jobs;
    int32_t rays_per_view, spv;          // spv: slices (work items per round) of a view = ceil(rays_per_view / kVQSlice)
    int32_t max_samples, min_samples, probabilistic;
    float far_plane, step_size, cone_angle, alpha_thre, opc_thre;
    I3 res; int32_t n_words;
    float occ_aabb[6];                   // the occupancy level's box (estimator.aabbs[0])
    const int32_t *view_order;
    int32_t *col_ray; float *col_ts, *col_te;     // [workgroup][wave][kVQWaveCols] column scratch
    int32_t *error;                      // = jobs[0].ctrl + kVQError
};

inline int64_t vq_scratch_bytes() { return (int64_t)kVQMaxGrid * kVQWaves * kVQWaveCols * 12; }

// dispatchers on the fields' operand type (viewq.hip, once per type)
namespace f16 { bool viewq_supported(int W, int NH); int launch_viewq_impl(const VQArgs &a, int W, int NH, int grid, hipStream_t s); }
namespace bf16 { bool viewq_supported(int W, int NH); int launch_viewq_impl(const VQArgs &a, int W, int NH, int grid, hipStream_t s); }

}  // namespace mnf
