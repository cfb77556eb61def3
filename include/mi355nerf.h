/*
 * mi355nerf.h — C ABI of libmi355nerf.so, the MI355X (gfx950) implementation of the
 * reference's perception hot path.
 *
 * The reference has no FFI of its own: its native layer is (1) the pybind11 module
 * `nerfacc_cuda` (perception/nerfacc/nerfacc/cuda/csrc/nerfacc.cpp:100-128) and (2) the
 * un-vendored `tinycudann` torch extension (perception/models/radiance_fields/ngp.py:108-169).
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *     buffers are caller-owned, the library allocates nothing per call except inside handles;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     enqueues work unless documented otherwise;
 *   - return value 0 = ok, negative = error; text via mnf_last_error() (thread-local);
 *   - handles are thread-compatible, not thread-safe;
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef MI355NERF_H
#define MI355NERF_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNF_OK 0
#define MNF_ERR_INVALID (-1)
#define MNF_ERR_HIP (-2)
#define MNF_ERR_UNSUPPORTED (-3)
#define MNF_ERR_WORKSPACE (-4)

typedef void *mnf_stream_t;
typedef struct mnf_field_s *mnf_field_t;

const char *mnf_last_error(void);
int mnf_version(void);
/* number of visible HIP devices (0 when there is none; never initialises a context) */
int mnf_device_count(void);

/* ---------------------------------------------------------------- nerfacc_cuda replacements */

/* nerfacc.cpp:110 `ray_aabb_intersect` (grid.cu:477-519, kernel :284-313).
 * t_mins,t_maxs [n_rays,n_aabbs] f32; hits [n_rays,n_aabbs] u8 (torch.bool layout). */
int mnf_ray_aabb_intersect(const float *rays_o, const float *rays_d, int32_t n_rays,
                           const float *aabbs, int32_t n_aabbs,
                           float near_plane, float far_plane, float miss_value,
                           float *t_mins, float *t_maxs, uint8_t *hits, mnf_stream_t stream);

/* One launch of the reference's `traverse_grids_kernel` (grid.cu:68-282); the host-side
 * count / cumsum / fill protocol of grid.cu:320-474 is driven by the caller exactly as the
 * reference's C++ does (first_pass=1 writes chunk_cnts only; first_pass=0 fills at
 * chunk_starts and rewrites chunk_cnts with the actual counts).  Pointers of a disabled output
 * are NULL (iv_chunk_cnts==NULL disables intervals, sm_chunk_cnts==NULL disables samples).
 * binaries [n_grids,X,Y,Z] u8; t_indices/ray_indices/chunk_* are int64 as in the reference. */
int mnf_traverse_grids(const float *rays_o, const float *rays_d, const uint8_t *rays_mask, int32_t n_rays,
                       const uint8_t *binaries, const float *aabbs, int32_t n_grids,
                       int32_t res_x, int32_t res_y, int32_t res_z,
                       const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
                       const float *near_planes, const float *far_planes,
                       float step_size, float cone_angle, int32_t traverse_steps_limit, int32_t first_pass,
                       float *iv_vals, int64_t *iv_ray_indices, uint8_t *iv_is_left, uint8_t *iv_is_right,
                       const int64_t *iv_chunk_starts, int64_t *iv_chunk_cnts,
                       float *sm_vals, int64_t *sm_ray_indices, uint8_t *sm_is_valid,
                       const int64_t *sm_chunk_starts, int64_t *sm_chunk_cnts,
                       float *terminate_planes, mnf_stream_t stream);

/* Single-pass form of the sampling traversal for one grid level (what `OccGridEstimator.sampling`,
 * occ_grid.py:80-238, needs from `traverse_grids`): every ray is marched ONCE; its samples (t_start, t_end) go to row r of
 * the caller's scratch [n_rays][cap] and counts[r] receives the ray's full sample count (rows are truncated at cap:
 * if any count exceeds cap the caller falls back to mnf_traverse_grids).  aabb_host: 6 host floats.  bitgrid: optional
 * bit-packed form of `binaries` (mnf_pack_bitgrid / mnf_occ_binarize; NULL = packed from the bytes by every workgroup).
 * The t values are those of mnf_traverse_grids, bit for bit. */
int mnf_sample_rays(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries, int32_t res_x,
                    int32_t res_y, int32_t res_z, const float *aabb_host, const float *near_planes, const float *far_planes,
                    float step_size, float cone_angle, int32_t cap, float *scratch_ts, float *scratch_te, int64_t *counts,
                    const uint32_t *bitgrid, mnf_stream_t stream);
/* The same over n_levels (1..4) occupancy levels (occ_grid.py:37-55; grid.cu:125-151 takes a ray's segments level by level):
 * binaries [n_levels,X,Y,Z], aabb_host n_levels x 6 floats (estimator.aabbs), bitgrid [n_levels][ceil(cells / 32)] words. */
int mnf_sample_rays_levels(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries, int32_t n_levels,
                           int32_t res_x, int32_t res_y, int32_t res_z, const float *aabb_host, const float *near_planes,
                           const float *far_planes, float step_size, float cone_angle, int32_t cap, float *scratch_ts, float *scratch_te,
                           int64_t *counts, const uint32_t *bitgrid, mnf_stream_t stream);

/* Pack the scratch rows: samples of ray r go to [chunk_starts[r], chunk_starts[r] + counts[r]) of t_starts / t_ends /
 * ray_indices (chunk_starts = exclusive prefix of counts, `RaySegmentsSpec::memalloc_data_from_chunk`, data_spec.hpp:86-96). */
int mnf_compact_samples(const float *scratch_ts, const float *scratch_te, int32_t cap, const int64_t *chunk_starts,
                        const int64_t *counts, int32_t n_rays, float *t_starts, float *t_ends, int64_t *ray_indices,
                        mnf_stream_t stream);

/* nerfacc.cpp:104 `exclusive_sum` (scan.cu:68-125); backward=1 is the reverse-direction scan
 * used by the autograd rule (scan.py:226-228). */
int mnf_exclusive_sum(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                      const float *inputs, float *outputs, int64_t n_edges, int32_t backward,
                      mnf_stream_t stream);

/* Fused render_weight_from_density (volrend.py:315-365) on packed samples:
 * weights = T*alpha, trans, alphas with an optional per-sample prefix transmittance (may be NULL). */
int mnf_render_weight_from_density(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                   const float *t_starts, const float *t_ends, const float *sigmas,
                                   const float *prefix_trans, int64_t n_samples,
                                   float *weights, float *trans, float *alphas, mnf_stream_t stream);

/* The tail of `OccGridEstimator.sampling` (occ_grid.py:209-236): render_visibility_from_density (volrend.py:424-483: T >= early_stop_eps and
 * alpha >= alpha_thre) and the three boolean-mask selections after it, on packed samples, in two passes around one prefix sum:
 *   count pass (o_t_starts NULL): kept_cnts[r] = survivors of ray r;
 *   write pass: kept_starts = exclusive scan of kept_cnts; survivors go to o_*[kept_starts[r] ...], grouped by ray in marching order.
 * alpha_thre_dev: ONE float on the device — min(alpha_thre, occs.mean()) of occ_grid.py:211 without the host round trip. */
int mnf_visible_samples(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays, const float *t_starts, const float *t_ends,
                        const float *sigmas, float early_stop_eps, const float *alpha_thre_dev, int64_t *kept_cnts, const int64_t *kept_starts,
                        float *o_t_starts, float *o_t_ends, int64_t *o_ray_indices, mnf_stream_t stream);

/* Run boundaries of ray indices that are grouped by ray (the marcher's output order): first[r] = index of the
 * first sample of ray r, last[r] = one past its last sample; both must be zero-filled by the caller, rays without
 * samples stay (0, 0).  cnts = last - first is `pack_info`'s second column (nerfacc/pack.py:10-38). */
int mnf_run_bounds(const int64_t *ray_indices, int64_t n_samples, int64_t *first, int64_t *last, mnf_stream_t stream);

/* Train-mode semantic volume rendering on packed samples (perception/models/utils.py:362-461 `sem_rendering`):
 * render_weight_from_density (volrend.py:213-267) + the four accumulate_along_rays (volrend.py:27-66) + background
 * blend + depth normalisation in one launch.  Outputs: rgb [R,3], acc [R], depth [R], sem [R,C]; per-sample weights and
 * trans [N] are saved for backward, alphas is optional.  bkgd (3 floats, device) may be NULL.  n_classes <= 64. */
int mnf_composite_train_forward(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                const float *sems, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                float *out_rgb, float *out_acc, float *out_depth, float *out_sem, float *weights,
                                float *trans, float *alphas, mnf_stream_t stream);

/* Its adjoint (what torch autograd derives for the reference's op chain): gradients of the four per-ray outputs
 * (any of g_* may be NULL = zero) -> d_sigmas [N], d_rgbs [N,3], d_sems [N,C].  d_rgbs and d_sems may BOTH be NULL: they are
 * weights[s] * g_rgb[ray] and weights[s] * g_sem[ray], which a caller that has the weights can form itself (mnf_train_step's
 * backward-data kernel does: 128 bytes per sample less to write and to read back). */
int mnf_composite_train_backward(const int64_t *chunk_starts, const int64_t *chunk_cnts, int32_t n_rays,
                                 const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                                 const float *sems, int32_t n_classes, int64_t n_samples, const float *bkgd,
                                 const float *weights, const float *trans, const float *out_acc,
                                 const float *out_depth, const float *g_rgb, const float *g_acc, const float *g_depth,
                                 const float *g_sem, float *d_sigmas, float *d_rgbs, float *d_sems, mnf_stream_t stream);

/* One `torch.optim.Adam` step (weight_decay 0, amsgrad off: scripts/pipeline.py:173-178, :531) on a flat fp32
 * parameter vector in a single pass; `step` counts from 1.  State buffers are the optimizer's exp_avg / exp_avg_sq. */
int mnf_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int32_t step, mnf_stream_t stream);

/* The same step for a training loop that never synchronises with the host (scripts/pipeline.py:520-532 decided on the device):
 * step_dev (device float, torch's `state["step"]`) is advanced and the update applied only when *skip_dev == 0 (skip_dev: device
 * int32 or NULL; raised by mnf_count_nan for non-finite gradients and by mnf_train_step for a step without samples or beyond its
 * bounds).  hyper_dev: 4 device floats of scratch.  half_out / half_from: optional fp16 mirror — parameters half_from.. are also
 * written, rounded to fp16, to half_out[0..] (the field handle's hash table, mnf_field_table_mirror: no per-step conversion pass). */
int mnf_adam_step_guarded(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                          float beta2, float eps, float *step_dev, const int32_t *skip_dev, float *hyper_dev, void *half_out,
                          int64_t half_from, mnf_stream_t stream);

/* scripts/pipeline.py:520-532 for one model as ONE call: non-finite-gradient guard over the three parameter vectors of a field (added to
 * *skip_dev when count_nonfinite != 0), the three mnf_adam_step_guarded updates (mlp_base / mlp_head / mlp_sem, in that order: host arrays
 * of 3 device pointers each; the hash table's fp16 mirror in the handle is written by the update), then the handle's MLP weight fragments
 * are re-derived: after the call the handle is current for the new parameters.  hyper_dev: 12 device floats of scratch. */
int mnf_field_optimizer_step(mnf_field_t f, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                             float *const *exp_avg_sq_host, float *const *step_dev_host, float lr, float beta1, float beta2, float eps,
                             int32_t *skip_dev, int32_t count_nonfinite, float *hyper_dev, mnf_stream_t stream);
/* The same call with the step's report to the host: counts_dev (mnf_train_step's 4 device counters, or NULL) and the final value of *skip_dev are written by the
 * step-count kernel straight into report_host — 5 x int64 of PINNED host memory (device-accessible) — behind the non-finite count: an asynchronous training loop
 * learns sample counts and skipped steps without a device-to-host copy on the stream (each such copy cost the stream ~25 us).  Readable once an event recorded
 * behind the call has completed. */
int mnf_field_optimizer_step_report(mnf_field_t f, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                                    float *const *exp_avg_sq_host, float *const *step_dev_host, float lr, float beta1, float beta2, float eps,
                                    int32_t *skip_dev, int32_t count_nonfinite, float *hyper_dev, const int64_t *counts_dev,
                                    int64_t *report_host, mnf_stream_t stream);

/* Adds the number of NaN / Inf entries of `values` to *count (device int32): the gradient guard of pipeline.py:520-529. */
int mnf_count_nan(const float *values, int64_t n, int32_t *count, mnf_stream_t stream);

/* `pack_info` (nerfacc/pack.py:10-38) for ray indices in any order: packed_info [n_rays,2] = (chunk start, chunk count).
 * workspace: 2*n_rays*8 + mnf_scan_workspace_bytes(n_rays) bytes. */
int64_t mnf_scan_workspace_bytes(int64_t n);
int mnf_pack_info(const int64_t *ray_indices, int64_t n_samples, int64_t n_rays, int64_t *packed_info, void *workspace,
                  int64_t workspace_bytes, mnf_stream_t stream);
/* Exclusive prefix sum of int64 counts (chunk starts of `RaySegmentsSpec::memalloc_data_from_chunk`,
 * include/data_spec.hpp:86-96); total_out (device, optional) receives the grand total. */
int mnf_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, int64_t *total_out, void *workspace, int64_t workspace_bytes,
                           mnf_stream_t stream);

/* `accumulate_along_rays` / `accumulate_along_rays_` with ray_indices (nerfacc/volrend.py:486-576): outputs [n_rays,dim]
 * (caller-initialised) += weights[k] * values[k,:] at row ray_indices[k]; values NULL = weights alone (dim 1).
 * The backward gives what autograd derives for the reference's index_add_: grad_weights [n] and grad_values [n,dim]
 * (either may be NULL) from grad_outputs [n_rays,dim]. */
int mnf_accumulate_along_rays(const float *weights, const float *values, const int64_t *ray_indices, int64_t n_samples, int32_t dim,
                              float *outputs, mnf_stream_t stream);
int mnf_accumulate_along_rays_backward(const float *weights, const float *values, const int64_t *ray_indices, int64_t n_samples,
                                       int32_t dim, const float *grad_outputs, float *grad_weights, float *grad_values,
                                       mnf_stream_t stream);

/* ---------------------------------------------------------------- occupancy-grid refresh
 * `OccGridEstimator._update` (nerfacc/estimators/occ_grid.py:345-437) without host round trips.  A refresh of one level is
 *   mnf_occ_sample_cells -> (density of the points: occ_eval_fn) -> mnf_occ_apply, then mnf_occ_binarize over all levels;
 * mnf_update_occupancy does the whole chain for one level with the field's own density kernel as occ_eval_fn
 * (scripts/pipeline.py:376-378: query_density(x) * render_step_size).
 * The sample list has a fixed capacity (mnf_occ_list_capacity: every cell during warm-up, 2 * (cells / 4) afterwards);
 * unused slots hold cell index -1 and a point inside the box.  Draws: Philox4x32-10, counter (element, 0, kind, step),
 * key = seed, kind 0 uniform / 1 occupied / 2 warm-up; word 0 -> cell, words 1..3 -> in-cell offsets (24-bit).  Tests
 * pass the reference's recorded draws instead (indices_in, jitter_in [n_in,3]).  Duplicate cells: the last list element
 * wins.  workspace: mnf_occ_workspace_bytes(cells, fused), the same buffer for sample and apply of one refresh. */
int64_t mnf_occ_workspace_bytes(int64_t cells_per_level, int32_t fused);
int64_t mnf_occ_list_capacity(int64_t cells_per_level, int32_t step, int32_t warmup_steps);
/* binaries [levels, cells] u8 -> bitgrid [levels, ceil(cells/32)] (bit c & 31 of word c >> 5) */
int mnf_pack_bitgrid(const uint8_t *binaries, int64_t cells_per_level, int32_t levels, uint32_t *bitgrid, mnf_stream_t stream);
int mnf_occ_sample_cells(const float *occs, const uint32_t *bitgrid, int32_t res_x, int32_t res_y, int32_t res_z,
                         const float *aabb_host, int32_t step, int32_t warmup_steps, uint64_t seed,
                         const int64_t *indices_in, const float *jitter_in, int64_t n_in,
                         int64_t *cell_idx, float *points, int64_t capacity, void *workspace, int64_t workspace_bytes,
                         mnf_stream_t stream);
/* occs[cell] = max(occs[cell] * ema_decay, values[e] * value_scale), NaN candidates leave the cell unchanged (occ_grid.py:403-434) */
int mnf_occ_apply(float *occs, const int64_t *cell_idx, const float *values, float value_scale, int64_t n, int64_t cells_per_level,
                  float ema_decay, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);
/* thre = min(mean(occs[occs >= 0]), occ_thre); binaries = occs > thre as bytes and as bits (occ_grid.py:436-437);
 * threshold_out (device float, optional) receives thre */
int mnf_occ_binarize(const float *occs, int64_t cells_per_level, int32_t levels, float occ_thre, uint8_t *binaries,
                     uint32_t *bitgrid, float *threshold_out, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);
int mnf_update_occupancy(mnf_field_t f, float *occs, uint8_t *binaries, uint32_t *bitgrid, int32_t res_x, int32_t res_y,
                         int32_t res_z, const float *aabb_host, int32_t step, int32_t warmup_steps, float occ_thre,
                         float ema_decay, float density_scale, uint64_t seed, void *workspace, int64_t workspace_bytes,
                         mnf_stream_t stream);

/* ---------------------------------------------------------------- ray generation */

/* Dataset.generate_image_rays (perception/data_proc/habitat_to_data.py:274-301) for n_views poses,
 * restricted to the flat pixel indices pix_idx[n_pix] (int64; the np.round(np.linspace(...))
 * sub-sampler of :462-467 is evaluated by the caller on the host in float64, as the reference does;
 * NULL = all width*height pixels).  c2w [n_views,3,4] row-major f32.
 * origins, viewdirs [n_views, n_pix, 3] f32. */
int mnf_generate_rays(const float *c2w, int32_t n_views, int32_t width, int32_t height, float focal,
                      const int64_t *pix_idx, int64_t n_pix, float *origins, float *viewdirs,
                      mnf_stream_t stream);

/* Dataset.fetch_data's pixel gather (habitat_to_data.py:229-232) for ONE image: rgb [n,3] f32 = images[id, pix] / 255.0,
 * dep [n] f32, sem [n] i64 at the flat pixel indices pix_idx (y * W + x).  images [N,H,W,3] u8; depths [N,H,W] f32
 * (reference storage) or f16 (packed layout, depth_is_f16 = 1); semantics [N,H,W] i64 or u8 (sem_is_u8 = 1); image_id:
 * ONE int64 on the device (the reference draws it with torch.randint on the device). */
int mnf_gather_pixels(const uint8_t *images, const void *depths, int32_t depth_is_f16, const void *semantics, int32_t sem_is_u8,
                      int64_t pixels_per_image, const int64_t *image_id, const int64_t *pix_idx, int64_t n_pix, float *rgb,
                      float *dep, int64_t *sem, mnf_stream_t stream);

/* ---------------------------------------------------------------- radiance field (tinycudann replacement) */

/* Structs that the library takes BY POINTER (mnf_field_config, mnf_vanilla_config, mnf_train_opts, mnf_render_opts, and every element of a mnf_render_job
 * array) start with `struct_size`: set it to sizeof(the struct) (MNF_INIT below zero-fills and sets it).
 * The library refuses any other value (MNF_ERR_INVALID), so a caller compiled against an older header — whose struct is shorter than what this
 * library reads — is stopped at the boundary instead of having the tail of its struct read from whatever follows it.  Every field added after a
 * struct's first release reads 0 / NULL as "the previous behaviour". */
#define MNF_INIT(var) do { memset(&(var), 0, sizeof(var)); (var).struct_size = (uint32_t)sizeof(var); } while (0)

typedef struct {
    uint32_t struct_size;          /* sizeof(mnf_field_config) */
    float aabb[6];                 /* ngp.py:91 */
    int32_t neurons;               /* ngp.py:77  (64 or 128) */
    int32_t layers;                /* ngp.py:78  == tcnn n_hidden_layers of the base MLP (>=1) */
    int32_t num_semantic_classes;  /* ngp.py:87  (1..32) */
    int32_t n_levels;              /* ngp.py:84  (must be 16) */
    int32_t n_features;            /* ngp.py:127 (must be 4) */
    int32_t log2_hashmap_size;     /* ngp.py:85 */
    int32_t base_resolution;       /* ngp.py:81 */
    int32_t max_resolution;        /* ngp.py:82 */
    int32_t output_fp16;           /* 0 (default): network outputs stay fp32.  1: every output of the three networks is rounded
                                      to fp16 before it is used, as tiny-cuda-nn hands them over (ngp.py:181-200, :210-220 cast
                                      tcnn's fp16 outputs back with `.to(x)`) — the tcnn-faithful mode of DESIGN.md §2 */
    int32_t mfma_bf16;             /* 0 (default): fp16 matrix-core operands (weights, MLP inputs, activations and their gradients), the
                                      reference's tiny-cuda-nn arithmetic.  1: bf16 operands (BASELINE config 5); the hash table stays
                                      fp16, accumulation fp32 */
    int32_t blend_fp16;            /* 0 (default): the 8-corner blend of a hash level runs in fp32 (fp32 weight x widened fp16 entry, fp32 sum, ONE rounding
                                      to the 16-bit MLP input).  1: tiny-cuda-nn's own arithmetic as published (grid encoding instantiated with T = __half
                                      behind a half-precision network): the corner weight is rounded to fp16 and the sum runs as fp16 fused multiply-adds,
                                      `result = fma((T)weight, entry, result)`, corners in index order (x fastest) — packed v_pk_fma_f16 on the two
                                      feature pairs.  Changes features by ~1e-3 relative; which one the reference's un-pinned tinycudann build computes
                                      cannot be verified in this container. */
} mnf_field_config;

/* tcnn.NetworkWithInputEncoding / tcnn.Network / tcnn.Encoding construction, ngp.py:108-169 */
int mnf_field_create(const mnf_field_config *cfg, mnf_field_t *out);
int mnf_field_destroy(mnf_field_t f);
/* number of fp32 parameters of module `which` (0 = mlp_base incl. hash table, 1 = mlp_head, 2 = mlp_sem),
 * i.e. the length of the reference state_dict entry `<module>.params` */
int64_t mnf_field_param_count(mnf_field_t f, int32_t which);
/* per-level metadata for tests: out arrays of n_levels entries (HOST pointers) */
int mnf_field_grid_meta_host(mnf_field_t f, float *scale_host, int32_t *res_host, int32_t *size_host,
                             int64_t *offset_host, int32_t *hashed_host);
/* Load fp32 master parameters (device pointers, reference state_dict layout) and build the fp16
 * hash table and MFMA-fragment-ordered fp16 weights held by the handle. */
int mnf_field_set_params(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem,
                         mnf_stream_t stream);
/* The handle's fp16 hash table as an optimizer mirror: device pointer to [entries][4] fp16; *first_param_host receives the index of
 * the first table entry inside `mlp_base.params` (the base-MLP weights precede it, ngp.py:123-141).  An optimizer that writes the
 * rounded new table values there (mnf_adam_step_guarded) calls mnf_field_refresh_weights afterwards instead of
 * mnf_field_set_params: only the few-KB MLP fragments are rebuilt. */
void *mnf_field_table_mirror(mnf_field_t f, int64_t *first_param_host);
int mnf_field_refresh_weights(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem, mnf_stream_t stream);

/* NGPRadianceField.forward (ngp.py:222-238): positions,directions [n,3] f32 ->
 * rgb [n,3], density [n,1], sem [n,C] f32 (any output may be NULL). */
int mnf_field_forward(mnf_field_t f, const float *positions, const float *directions, int64_t n,
                      float *rgb, float *density, float *sem, mnf_stream_t stream);
/* NGPRadianceField.query_density (ngp.py:171-200): density [n,1] */
int mnf_field_density(mnf_field_t f, const float *positions, int64_t n, float *density, mnf_stream_t stream);
/* the closures sigma_fn / rgb_sigma_sem_fn of utils.py:89-137 fused with the field:
 * positions = o[ray] + d[ray]*(t_start+t_end)/2, directions = d[ray]; ray_indices int64 */
int mnf_field_forward_samples(mnf_field_t f, const float *rays_o, const float *rays_d,
                              const int64_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                              float *rgb, float *density, float *sem, mnf_stream_t stream);

/* The density pre-pass of `OccGridEstimator.sampling` (occ_grid.py:196-233: sigma_fn on every marched sample, then
 * render_visibility_from_density) with the work behind opaque surfaces left out: samples are packed by ray
 * (chunk_starts / chunk_cnts [n_rays] int64, as traverse_grids returns them); a ray is evaluated front to back and the
 * rest of it is skipped once its transmittance has fallen below early_stop_eps / 2 — those samples fail the visibility
 * test whatever their density, so `density` (zero there) yields the same mask.  early_stop_eps <= 0 evaluates everything. */
int mnf_field_density_rays(mnf_field_t f, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                           const float *t_starts, const float *t_ends, const int64_t *chunk_starts, const int64_t *chunk_cnts,
                           int32_t n_rays, int64_t n_samples, float early_stop_eps, float *density, mnf_stream_t stream);

/* ---------------------------------------------------------------- frequency-encoded MLP field (BASELINE config 1)
 * perception/models/radiance_fields/mlp.py: `VanillaNeRFRadianceField` (:206-245) = `SinusoidalEncoder` (:168-203, 10 degrees
 * on positions, 4 on directions, identity included) + `NerfMLP` (:113-165) of biased Linear + ReLU layers (:14-101), in
 * exact fp32 on the matrix cores.  Parameters cross the boundary as ONE flat fp32 vector in the order of the reference's
 * `named_parameters()`: per Linear its weight [out][in] then its bias — base hidden layers, sigma layer, bottleneck layer,
 * rgb hidden layers, rgb output layer (mnf_vanilla_param_layout_host lists the tensors). */
typedef struct mnf_vanilla_s *mnf_vanilla_t;
typedef struct {
    uint32_t struct_size;         /* sizeof(mnf_vanilla_config), see MNF_INIT */
    int32_t net_depth;            /* mlp.py:209 */
    int32_t net_width;            /* mlp.py:210 (multiple of 32) */
    int32_t skip_layer;           /* mlp.py:211; <= 0 = None */
    int32_t net_depth_condition;  /* mlp.py:212 (>= 1) */
    int32_t net_width_condition;  /* mlp.py:213 (multiple of 32) */
} mnf_vanilla_config;
int mnf_vanilla_create(const mnf_vanilla_config *cfg, mnf_vanilla_t *out);
int mnf_vanilla_destroy(mnf_vanilla_t v);
int64_t mnf_vanilla_param_count(mnf_vanilla_t v);
/* offsets / rows / cols of the parameter tensors inside the flat vector, in named_parameters() order (host arrays of
 * max_tensors entries; *n_tensors_host receives the count) */
int mnf_vanilla_param_layout_host(mnf_vanilla_t v, int32_t max_tensors, int32_t *n_tensors_host, int64_t *offsets_host,
                                  int32_t *rows_host, int32_t *cols_host);
int mnf_vanilla_set_params(mnf_vanilla_t v, const float *flat_params, mnf_stream_t stream);
int64_t mnf_vanilla_train_workspace_bytes(mnf_vanilla_t v, int64_t n);
/* VanillaNeRFRadianceField.forward (mlp.py:238-243): positions [n,3]; directions [ceil(n / samples_per_direction),3] (the
 * reference broadcasts a per-ray condition over the ray's samples, mlp.py:154-160; 1 = one direction per sample);
 * rgb [n,3] = sigmoid, sigma [n] = relu.  workspace non-NULL (mnf_vanilla_train_workspace_bytes) keeps the activations for
 * mnf_vanilla_backward. */
int mnf_vanilla_forward(mnf_vanilla_t v, const float *positions, const float *directions, int64_t n,
                        int32_t samples_per_direction, float *rgb, float *sigma, void *workspace, int64_t workspace_bytes,
                        mnf_stream_t stream);
/* VanillaNeRFRadianceField.query_density (mlp.py:233-236) */
int mnf_vanilla_density(mnf_vanilla_t v, const float *positions, int64_t n, float *sigma, mnf_stream_t stream);
/* gradients of all parameters (flat, overwritten) from dL/d(rgb) [n,3] and dL/d(sigma) [n]; rgb / sigma are the forward
 * outputs, workspace the one the forward filled */
int mnf_vanilla_backward(mnf_vanilla_t v, const float *d_rgb, const float *d_sigma, const float *rgb, const float *sigma,
                         int64_t n, void *workspace, int64_t workspace_bytes, float *grad_flat, mnf_stream_t stream);

/* ---------------------------------------------------------------- training: differentiable field
 * What `loss.backward()` reaches inside tiny-cuda-nn in the reference (scripts/pipeline.py:518): the forward
 * that keeps activations and the backward to the flat parameter vectors.  Compositing, the loss and the optimizer
 * stay with the caller exactly as in the reference (perception/models/utils.py:362-461, pipeline.py:506-535). */

/* The backward behind the train forward is tiny-cuda-nn's own decomposition (ngp.py:123-169 / pipeline.py:518): a fused backward-data kernel (dgrad) and
 * weight-gradient GEMMs (wgrad) over the 2.4 KB per sample of 16-bit activations the train forward saves.  (A single-kernel form without the dump was built in
 * round 4 and measured slower: tools/experiments/fused_backward.patch, DESIGN.md "negative results".) */
/* bytes of workspace the train forward/backward pair needs for n samples (activations, masks, feature gradients) */
int64_t mnf_field_train_workspace_bytes(mnf_field_t f, int64_t n);
/* NGPRadianceField.forward with activations saved into `workspace` for the following backward */
int mnf_field_forward_train(mnf_field_t f, const float *positions, const float *directions, int64_t n,
                            float *rgb, float *density, float *sem, void *workspace, int64_t workspace_bytes,
                            mnf_stream_t stream);
/* Backward of the forward above: d_rgb [n,3], d_density [n], d_sem [n,C] are dL/d(outputs); rgb/density are the
 * forward outputs; `workspace` is the one the forward filled.  g_base / g_head / g_sem receive dL/d(params) in the
 * state_dict layout (fp32, overwritten).  Activation gradients are carried in fp16 scaled by loss_scale (tcnn's
 * default is 128); the returned gradients are un-scaled.  The parameters are the ones last passed to
 * mnf_field_set_params (the handle holds its own fp16 copies; the caller's vectors need not be alive). */
int mnf_field_backward(mnf_field_t f, const float *positions, int64_t n,
                       const float *d_rgb, const float *d_density, const float *d_sem,
                       const float *rgb, const float *density,
                       void *workspace, int64_t workspace_bytes, float loss_scale,
                       float *g_base, float *g_head, float *g_sem, mnf_stream_t stream);

/* mnf_field_forward_train for packed samples given as (ray, t_start, t_end): positions are formed in the kernel as the closure of
 * utils.py:122-137 does (origins + dirs * (t_starts + t_ends) / 2) and also written to positions_out [n,3], which
 * mnf_field_backward takes as `positions`. */
int mnf_field_forward_train_samples(mnf_field_t f, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                    const float *t_starts, const float *t_ends, int64_t n, float *rgb, float *density,
                                    float *sem, float *positions_out, void *workspace, int64_t workspace_bytes,
                                    mnf_stream_t stream);

/* ---------------------------------------------------------------- one training iteration's forward + loss + backward
 * scripts/pipeline.py:472-518 for one model as ONE call: the train render `render_image_with_occgrid_with_depth_guide`
 * (perception/models/utils.py:63-219: occupancy sampling with stratified near planes, density pre-pass and visibility filter,
 * occ_grid.py:80-238; then sem_rendering, utils.py:362-461), the loss 10 smooth_l1(rgb) + smooth_l1(depth) / 5 + CE(sem) / 2
 * (pipeline.py:506-511) and its backward to the three flat parameter-gradient vectors g_base / g_head / g_sem (overwritten).
 * Not included, as in the reference they belong to the caller: occupancy refresh (mnf_update_occupancy), NaN guard
 * (mnf_count_nan), optimizer (mnf_adam_step / mnf_adam_step_guarded).  losses (device, 4 floats): total, rgb, depth, semantic
 * terms (un-weighted means).
 * NO stream synchronisation: the sample counts stay on the device.  counts_dev (DEVICE, 4 x int64): [0] marched samples, [1] surviving
 * samples (= the reference's n_rendering_samples), [2] samples of the longest ray, [3] status bits: 1 marched > max_marched, 2 a ray
 * longer than a scratch row (use the two-pass sampler), 4 surviving > max_kept, 8 a class id outside [0, C) (F.cross_entropy's device
 * assert), 16 no sample survived (the reference `continue`s, pipeline.py:491).  skip_dev (DEVICE int32): set to 0, then raised for
 * every status bit: with it non-zero the gradients are zero / must not be applied (mnf_adam_step_guarded reads it).  The caller
 * bounds the sample counts (max_marched, max_kept), provides mnf_train_step_workspace_bytes(), reads counts_dev when it wants
 * to and retries with larger bounds after bits 1 / 4.
 * Stratified near planes: near + U[0,1) * render_step_size per ray from Philox4x32-10 (counter (ray, 0, 7, 0), key = seed). */
typedef struct {
    uint32_t struct_size;     /* sizeof(mnf_train_opts) */
    float near_plane, far_plane, render_step_size, cone_angle, alpha_thre, early_stop_eps;   /* utils.py:63-76 / occ_grid.py:80-96 */
    float render_bkgd[3];
    float loss_scale;         /* fp16 activation-gradient scale of the backward (tcnn: 128) */
    int32_t stratified;       /* occ_grid.py:187-189 (radiance_field.training) */
    uint64_t seed;
    const float *render_bkgd_dev;   /* optional: 3 device floats used instead of render_bkgd (pipeline.py:437 draws the colour on
                                       the GPU: no host copy of it is needed) */
    int32_t deterministic;          /* 0 (default): gradients accumulate with float atomics (the reference's index_add_ / tcnn atomics: the
                                       result depends on arrival order in the last bits).  1: bitwise reproducible — the hash-table gradient
                                       accumulates in 64-bit fixed point (integer atomics), the weight gradients' partial sums are added in
                                       a fixed order.  Same kernels otherwise; used for stand-in scenes that must come out the same on
                                       every box (apnrf_amd.standin) and for regression tests. */
    int32_t n_levels;               /* occupancy levels (0 or 1: one): binaries [n_levels,X,Y,Z], occs [n_levels * cells], aabb_host n_levels x 6 floats,
                                       bitgrid [n_levels][ceil(cells / 32)]; the field's aabb is the largest level's (pipeline.py:167-172) */
    struct mnf_presample_s *presampled;   /* optional (NULL: march inside the step): a mnf_train_presample of exactly these rays, options and grid, see below */
} mnf_train_opts;
/* The parameter-independent head of a training iteration, ahead of time.  The march of a batch (occ_grid.py:181-208: stratified near planes, the alpha
 * threshold from the grid's mean occupancy, traverse_grids, per-ray offsets) reads the rays and the occupancy grid but not the model: the reference runs it inside
 * every iteration, where one lane per ray leaves most of the chip idle for 0.2 ms.  mnf_train_presample runs it for the NEXT batch on a library-owned
 * side stream, forked from `stream` as it is at the call — call it BEFORE enqueuing the current iteration and it marches beside that iteration's kernels —
 * and mnf_train_step adopts the result when opts->presampled names the handle (it waits for the march on its own stream; same rays / options / seed / grid
 * pointers required, MNF_ERR_INVALID otherwise; one use).  Results are bit-identical to marching inside the step.  The caller keeps rays, grid and `workspace`
 * (mnf_train_presample_workspace_bytes(n_rays, max_marched) device bytes; max_marched: the adopting step's bound, checked there — the march's guard and the packing of its samples run in the presample too) untouched until the adopting step has been enqueued, and must not refresh the occupancy grid in
 * between without mnf_presample_wait (then simply do not pass the handle: the step marches itself).
 * Replaces nothing in the reference's call list: a scheduling entry point (the data loader's "next batch" prefetch applied to the sampler). */
typedef struct mnf_presample_s *mnf_presample_t;
int mnf_presample_create(mnf_presample_t *out);
void mnf_presample_destroy(mnf_presample_t p);
int64_t mnf_train_presample_workspace_bytes(int32_t n_rays, int64_t max_marched);
int mnf_train_presample(mnf_presample_t p, const uint8_t *binaries, const uint32_t *bitgrid, const float *occs, int32_t res_x, int32_t res_y,
                        int32_t res_z, const float *aabb_host, const float *rays_o, const float *rays_d, int32_t n_rays,
                        const mnf_train_opts *opts, int64_t max_marched, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);
/* make `stream` wait for the handle's march (no-op if none was launched) */
int mnf_presample_wait(mnf_presample_t p, mnf_stream_t stream);

int64_t mnf_train_step_workspace_bytes(mnf_field_t f, int32_t n_rays, int64_t max_marched, int64_t max_kept);
int mnf_train_step(mnf_field_t f, const uint8_t *binaries, const uint32_t *bitgrid, const float *occs, int32_t res_x, int32_t res_y,
                   int32_t res_z, const float *aabb_host, const float *rays_o, const float *rays_d, int32_t n_rays,
                   const float *target_rgb, const float *target_depth, const int64_t *target_sem, const mnf_train_opts *opts,
                   float *g_base, float *g_head, float *g_sem, float *losses, int64_t *counts_dev, int32_t *skip_dev, int64_t max_marched,
                   int64_t max_kept, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);

/* ---------------------------------------------------------------- fused test-mode renderers */

typedef struct {
    uint32_t struct_size;    /* sizeof(mnf_render_opts), see MNF_INIT */
    float near_plane;        /* utils.py:563 */
    float far_plane;         /* utils.py:564 */
    float render_step_size;  /* utils.py:565 */
    float cone_angle;        /* utils.py:567 */
    float alpha_thre;        /* utils.py:568 */
    float early_stop_eps;    /* utils.py:569 */
    float render_bkgd[3];    /* utils.py:566 */
    int32_t max_samples;     /* utils.py:557 */
    int32_t probabilistic;   /* 0: render_image_with_occgrid_test, 1: render_probablistic_image_with_occgrid_test */
    int32_t rays_per_view;   /* rays of one reference call; n_rays must be a multiple of it.  Each group of
                                rays_per_view consecutive rays is rendered exactly as one call of the reference
                                function (its own n_alive / n_samples round schedule, utils.py:667-672). */
    int32_t sync_every;      /* host checks the device "all views finished" flag every this many rounds
                                (stream sync); 0 = never (all ceil(max_samples/min_samples) rounds are enqueued) */
    const int32_t *view_order; /* optional (NULL = identity): device permutation of 0..rays_per_view-1, the order in which the
                                rays of every view are marched and packed into the field kernel's 64-column tiles.  Results
                                are per ray and do not depend on it; rays that are neighbours in the image should be
                                neighbours in this order (e.g. 8x8 pixel blocks instead of one-pixel-high strips) so that
                                a tile's samples share hash-table lines. */
    const uint32_t *bitgrid; /* optional (NULL = packed from `binaries` at the start of the call): the bit-packed form of
                                `binaries` that mnf_occ_binarize / mnf_pack_bitgrid maintain (bit c & 31 of word c >> 5;
                                [n_levels][ceil(cells / 32)] words) */
    int32_t n_levels;        /* occupancy levels (occ_grid.py:37-55: level l covers the roi enlarged 2^l times); 0 or 1 = one level.
                                With n_levels > 1 `binaries` is [n_levels,X,Y,Z], `aabb_host` holds n_levels x 6 floats
                                (estimator.aabbs) and a ray's segments are taken level by level as grid.cu:125-151 does (<= 4 levels) */
} mnf_render_opts;

/* bytes of workspace mnf_render_test needs for n_rays rays */
int64_t mnf_render_workspace_bytes(int64_t n_rays, int32_t rays_per_view);

/* render_image_with_occgrid_test (perception/models/utils.py:555-779) and
 * render_probablistic_image_with_occgrid_test (utils.py:782-1032).
 * binaries [L,X,Y,Z] u8, aabb_host[6 L] = estimator.aabbs, L = opts->n_levels (0 or 1: one level; <= 4).
 * Outputs: rgb [n,3], acc [n,1], depth [n,1], sem [n,C]; rgb_var [n,3], depth_var [n,1] (probabilistic only,
 * may be NULL otherwise); total_samples: TWO int64 (device): [0] = the reference's total_samples (samples kept
 * after the alpha threshold, utils.py:757), [1] = samples evaluated by the field (all marched samples).
 * Synchronises the stream only as sync_every asks. */
int mnf_render_test(mnf_field_t f, const uint8_t *binaries, int32_t res_x, int32_t res_y, int32_t res_z,
                    const float *aabb_host, const float *rays_o, const float *rays_d, int64_t n_rays,
                    const mnf_render_opts *opts,
                    float *rgb, float *acc, float *depth, float *sem, float *rgb_var, float *depth_var,
                    int64_t *total_samples, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);

/* Several independent render calls advancing side by side (the members of an ensemble rendering the same candidate views,
 * pipeline.py:697-711 called once per member; or groups of views of one pose list, habitat_to_data.py:304-549): every job is what
 * one mnf_render_test call is, with its own field / grid / rays / outputs / workspace; job 0 runs on `stream`, the others on
 * streams of the library that fork from and join `stream`.  While one job's short kernels leave compute units idle the others'
 * launches use them.  Results are those of separate mnf_render_test calls, bit for bit.  `opts->bitgrid` is ignored (per job). */
typedef struct {
    uint32_t struct_size;         /* sizeof(mnf_render_job) in EVERY element of the array, see MNF_INIT */
    mnf_field_t field;
    const uint8_t *binaries;      /* [L,X,Y,Z] u8, L = opts->n_levels */
    const uint32_t *bitgrid;      /* optional packed form of `binaries` (see mnf_render_opts.bitgrid) */
    const float *rays_o, *rays_d; /* [n_rays,3] */
    int64_t n_rays;               /* a multiple of opts->rays_per_view; 0 = nothing to do */
    float *rgb, *acc, *depth, *sem, *rgb_var, *depth_var;
    int64_t *total_samples;       /* 2 x int64 (device) */
    void *workspace;              /* mnf_render_workspace_bytes(n_rays, rays_per_view), not shared between jobs */
    int64_t workspace_bytes;
} mnf_render_job;
int mnf_render_jobs(const mnf_render_job *jobs_host, int32_t n_jobs, int32_t res_x, int32_t res_y, int32_t res_z,
                    const float *aabb_host, const mnf_render_opts *opts, mnf_stream_t stream);

/* ---------------------------------------------------------------- planner hand-off
 * The only perception artefact the planner reads (scripts/pipeline.py:1043-1049, planning/planning_funcs.py:243-261):
 * binaries [n_members,X,Y,Z] u8 (the ensemble's `estimator.binaries[0]`), sliced at height index y_slice (8 in the
 * reference), merged over members and dilated by the 3x3 box with symmetric boundary -> out_map [X,Z] int32 (1 = blocked).
 * The caller clears the cells around the vehicle (planning_funcs.py:262-266) on the host. */
int mnf_planner_map(const uint8_t *binaries, int32_t n_members, int32_t res_x, int32_t res_y, int32_t res_z,
                    int32_t y_slice, int32_t *out_map, mnf_stream_t stream);

/* Optional in-library kernel timing for bench.py's roofline figures: between begin and end the library brackets its main
 * launches with hipEvent pairs on the launch stream, grouped by label: "field_render" (the fused field kernel of
 * mnf_render_test), "field_density", "field_forward", "field_train_forward", "dgrad", "wgrad", "hash_scatter",
 * "composite_train_forward", "composite_train_backward", "sample_rays".  mnf_profile_end synchronises the events, sums
 * the milliseconds per label and returns the "field_render" totals; mnf_profile_query reads any label afterwards.
 * Process-wide (backward passes run on torch's autograd thread).  Not part of the reference surface. */
int mnf_profile_begin(void);
int mnf_profile_end(double *field_ms_host, int64_t *launches_host);
int mnf_profile_query(const char *label_host, double *ms_host, int64_t *launches_host);

/* ---------------------------------------------------------------- predictive-information scorer */

/* scripts/pipeline.py:727-781 on device.  Inputs are the renders of n_members ensemble members for the
 * same n_views views of n_pix pixels: member-major arrays rgb_var [M,V,P,3], depth_var [M,V,P], acc [M,V,P],
 * sem [M,V,P,C] f32.  Output terms [V,4] f64 = per-view means of the rgb / depth / semantic / occupancy
 * predictive-information terms (un-weighted); the trajectory score is
 * mean_v(t0 + t1 + 3*t2 + 2*t3) (pipeline.py:775-781). */
int mnf_score_views(const float *rgb_var, const float *depth_var, const float *acc, const float *sem,
                    int32_t n_members, int32_t n_views, int32_t n_pix, int32_t n_classes,
                    double *terms, mnf_stream_t stream);

/* scripts/pipeline.py:674-781 for one trajectory as ONE call: candidate poses (c2w [n_views,3,4] f32, device) -> the
 * sub-sampled rays of every view (pix_idx [n_pix] int64, device: the np.round(np.linspace) pixels of habitat_to_data.py:462-467)
 * -> probabilistic renders by every ensemble member (fields_host / binaries_host / bitgrids_host: HOST arrays of n_members
 * handles / device pointers; bitgrids_host or its entries may be NULL) -> per-view terms [n_views,4] f64 as mnf_score_views. */
int64_t mnf_score_poses_workspace_bytes(int32_t n_members, int32_t n_views, int32_t n_pix, int32_t n_classes);
int mnf_score_poses(const mnf_field_t *fields_host, const uint8_t *const *binaries_host, const uint32_t *const *bitgrids_host,
                    int32_t n_members, int32_t res_x, int32_t res_y, int32_t res_z, const float *aabb_host, const float *c2w,
                    int32_t n_views, int32_t width, int32_t height, float focal, const int64_t *pix_idx, int64_t n_pix,
                    const mnf_render_opts *opts, double *terms, void *workspace, int64_t workspace_bytes, mnf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
