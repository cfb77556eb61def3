// Radiance-field handle shared between field.hip (kernels) and render.hip (fused renderer).
#pragma once
#include <vector>

#include "common.h"

namespace mnf {

struct LevelMeta {
    float scale;
    uint32_t res;
    uint32_t size;    // entries in this level (tcnn params_in_level)
    uint32_t offset;  // first entry of this level in the table
    uint32_t hashed;  // 1: spatial hash, 0: dense
    // idx / size for any uint32 idx without a branch (dense levels wrap the index as tcnn does):
    // q = mulhi(div_magic, idx); q = (((idx - q) >> 1) + q) >> div_shift
    uint32_t div_magic, div_shift;
};

struct FieldShape {
    int W, NH, Wh, C;
    int blocks_total;  // 1 KiB fragment blocks of all nine weight matrices
};

}  // namespace mnf

struct mnf_field_s {
    mnf_field_config cfg;
    mnf::FieldShape shape;
    mnf::LevelMeta levels[16];
    int64_t table_entries;
    int64_t n_base_mlp, n_base, n_head, n_sem;  // fp32 parameter counts
    // device buffers owned by the handle
    void *d_table;       // fp16 [table_entries][4]
    void *d_frags;       // fp16 fragment-ordered MLP weights, blocks_total KiB
    int32_t *d_frag_src; // gather table: (buffer << 28) | index, or -1 for a structural zero
    int32_t *d_counter;  // work counter of the ray-major density pass (lazily allocated)
    bool params_loaded;
    std::vector<int32_t> frag_src_host;   // host copy of the gather table (train.hip inverts it: parameter -> fragment slot)
    void *train_state;        // lazily built by train.hip (transposed fragments, weight-gradient job table)
};
inline bool field_rows_supported(const mnf_field_s *f) { return f->cfg.neurons == 128 && !f->cfg.blend_fp16; }

namespace mnf {

// Per-round volumetric compositing fused into the field kernel's epilogue (mode 2): the renderer's tiles hold
// `nslots` rays x `stride` columns (a ray never straddles a 64-column tile), described by one header word per tile.
struct FusedRender {
    const int32_t *tile_hdr;      // [tiles]: columns per ray slot (= the budget of the tile's view this round) | view << 8
    uint8_t *alive;               // [rays]
    int32_t *alive_count;         // [views] survivors of this round
    const int32_t *n_samples;     // [views] this round's per-ray sample budget
    float *rgb, *acc, *depth, *sem, *rgb_var, *depth_var;   // running accumulators (the call's outputs)
    unsigned long long *totals;   // [2]: kept samples, evaluated samples
    int32_t rays_per_view, probabilistic;
    int32_t general_only;         // diagnostic (MNF_COMPOSITE_GENERAL=1): composite every tile with the general segmented-scan path
    float alpha_thre, opc_thre;
};

// What the fused kernel reads / writes.  mode 0: explicit positions+directions; mode 1: packed samples
// with int64 ray indices; mode 2: renderer columns (int32 ray id, -1 = unused column); mode 3: mode 1 walked ray-major.
struct FieldIO {
    int mode;
    const float *positions, *directions;     // mode 0
    const float *rays_o, *rays_d;            // mode 1, 2
    const int64_t *ray_idx64;                // mode 1
    const int32_t *col_ray;                  // mode 2
    const float *t_starts, *t_ends;          // mode 1, 2
    int64_t n;                               // modes 0, 1
    const int32_t *n_dev;                    // mode 2: number of columns (device)
    int64_t n_cap;                           // mode 2: capacity of the column arrays (the device count is clamped to it)
    int32_t grid_limit;                      // mode 2: workgroups of this launch (0 = one per CU).  Render jobs that advance side by side share the chip:
                                             // each job's launch takes its share of the CUs, so the jobs' field kernels run BESIDE each other
    uint32_t *tickets;                       // mode 2, optional: eight zeroed counters 64 bytes apart.  The waves then take their tiles in the order they get to
                                             // them (one ticket per tile, range of XCD x first, then the next ranges) instead of a fixed stride
    const int64_t *n_dev64;                  // modes 0, 1, optional: the sample count lives on the device (`n` is then its upper
                                             // bound and sizes the launch): the train step never brings a count to the host
    // mode 3: packed samples walked ray by ray (ray_idx64 / t_starts / t_ends as mode 1) with early termination
    const int64_t *chunk_starts, *chunk_cnts;
    int32_t n_rays;
    int32_t *ray_counter;
    float sdt_stop;                          // stop a ray once its accumulated sigma * dt exceeds this
    // two-launch forms of the field evaluation (enc != NULL): phase 0 = gather launch then MLP launch on one stream (diagnostic
    // MNF_FIELD_SPLIT), 1 = gather launch only, 2 = MLP launch only; both restricted to tile chunk `chunk` of `n_chunks`
    // (0 / 0 = all tiles).  The renderer's pipelined mode (MNF_FIELD_PIPE) runs phase 1 of chunk i+1 beside phase 2 of chunk i.
    int32_t phase, chunk, n_chunks, mlp_waves, gather_grid;
    const void *enc;                         // optional [ceil(n/64)][8][64] x 16 B feature scratch: non-null selects the
                                             // two-launch path (encode_kernel, then the MLP kernel on its output)
    // outputs: user layout (modes 0,1) ...
    float *rgb, *density, *sem;
    float *positions_out;                    // mode 1, optional: the sample positions [n,3] the kernel formed (the backward's scatter reads them)
    float *xn_out;                           // mode 1, optional: the aabb-normalised positions [n,3] (ngp.py:177-178) instead: the scatter's own input
    // ... or, in mode 2, composites straight into the renderer's per-ray accumulators
    FusedRender fr;
    // The train step's two passes over the hash table made one (neurons = 128, fp32 blend: field_rows_supported): the density pre-pass (mode 3) leaves every sample's
    // 64 encoded features behind as one 128-byte row, rows_out[sample] — exactly the 16-bit values its own MLP consumed — and the training forward (mode 1) reads
    // row rows_src[k] of rows_in for its sample k (one cache line) instead of gathering 128 table entries again.
    void *rows_out;
    const void *rows_in;
    const int64_t *rows_src;
    // modes 0, 1: `sem` class-major, sem[class * sem_stride + sample], instead of [sample][C] (0): a wave then writes 128-byte runs instead of 4-byte pieces at a
    // C * 4-byte lane stride (the train step's private logit buffer: 75 us of the 650 us training forward at 1.0 M samples were those stores)
    int64_t sem_stride;
};

// Training-time activation storage handed to the forward kernel (layout: field_dev.h, TrainLayout)
struct TrainBuf {
    void *act;        // [tiles][rows][64] 16-bit elements
    uint8_t *masks;   // [tiles][64 lanes][mask_bytes]: a record per lane (field_dev.h TrainLayout)
    int64_t Np;
    int32_t rows;
};

// dispatchers on the handle's operand type (defined once, in the fp16 translation units)
void free_train_state(mnf_field_t f);
int launch_field(mnf_field_t f, const FieldIO &io, bool density_only, hipStream_t stream, const TrainBuf *train = nullptr);
// training forward / backward with the sample count optionally on the device (io.n_dev64 / n_dev; n = upper bound)
int forward_train(mnf_field_t f, const FieldIO &io, void *workspace, int64_t workspace_bytes, hipStream_t stream, bool deterministic = false);
int backward(mnf_field_t f, const float *positions, int64_t n, const int64_t *n_dev, const float *d_rgb, const float *d_density,
             const float *d_sem, const float *rgb, const float *density, void *workspace, int64_t workspace_bytes, float loss_scale,
             float *g_base, float *g_head, float *g_sem, bool zero_grads, bool positions_normalized, bool deterministic, hipStream_t stream);

#define MNF_DECLARE_DT_IMPL(ns)                                                                                                      \
    namespace ns {                                                                                                                   \
    int launch_field_impl(mnf_field_t f, const FieldIO &io, bool density_only, hipStream_t stream, const TrainBuf *train);           \
    int set_params_impl(mnf_field_t f, const float *mlp_base, const float *mlp_head, const float *mlp_sem, bool table_current,       \
                        hipStream_t stream);                                                                                         \
    void free_train_state_impl(mnf_field_t f);                                                                                       \
    int64_t train_workspace_bytes_impl(mnf_field_t f, int64_t n);                                                                    \
    int forward_train_impl(mnf_field_t f, const FieldIO &io, void *workspace, int64_t workspace_bytes, hipStream_t stream,          \
                           bool deterministic);                                                                                      \
    int backward_impl(mnf_field_t f, const float *positions, int64_t n, const int64_t *n_dev, const float *d_rgb,                  \
                      const float *d_density, const float *d_sem, const float *rgb, const float *density, void *workspace,          \
                      int64_t workspace_bytes, float loss_scale, float *g_base, float *g_head, float *g_sem, bool zero_grads,       \
                      bool positions_normalized, bool deterministic, hipStream_t stream);                                           \
    }
MNF_DECLARE_DT_IMPL(f16)
MNF_DECLARE_DT_IMPL(bf16)

}  // namespace mnf
