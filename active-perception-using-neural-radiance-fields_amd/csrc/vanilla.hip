// Frequency positional encoding + biased ReLU MLP radiance field of BASELINE config 1, forward and backward, in exact fp32
// on the matrix cores (v_mfma_f32_32x32x2_f32: an fp32 fma chain per output element, bit-comparable with a CPU fp32 GEMM
// up to summation order).
//
// Replaces perception/models/radiance_fields/mlp.py: `SinusoidalEncoder` (:168-203), `MLP` (:14-101), `NerfMLP` (:113-165)
// and `VanillaNeRFRadianceField` (:206-245).  The network is a chain
//     E = enc(x, 10 deg) -> base hidden 0..D-1 (ReLU, optional skip concat of E) -> head = [bottleneck (identity) ; raw sigma]
//       -> [bottleneck, enc(dir, 4 deg)] -> rgb hidden 0..Dc-1 (ReLU) -> raw rgb;   sigma = relu, rgb = sigmoid
// (the sigma and bottleneck `DenseLayer`s read the same input and are evaluated as one (W+1)-row matrix).
//
// One wave owns 32 samples.  The activations of the layer being evaluated live in LDS as feature-major blocks
// [feature][32 samples] (the per-sample feature blocks of the north star), which makes the MFMA B operand (two adjacent
// feature rows of one sample column) a conflict-free ds_read_b32; the weights are the A operand, pre-permuted into
// fragment order so that a lane fetches four k-steps with one 16-byte load.  Every layer is H_out^T = W . H_in^T + b.
// Training saves each layer's input rows tile-major in HBM (ACT[tile][row][32]); the backward-data kernel walks the
// chain in reverse with W^T fragments and stores the pre-activation gradients next to them; the weight-gradient kernel
// contracts gradient rows with input rows over the samples (again on the matrix cores) and adds bias gradients on the way.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace mnf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxLayers = 16;
constexpr int kEncRows = 64;    // 63 position-encoding rows + 1 zero row
constexpr int kCondRows = 32;   // 27 direction-encoding rows + 5 zero rows
constexpr int kPosDeg = 10, kDirDeg = 4;

enum Kind { kHidden = 0, kHead = 1, kRgbOut = 2 };
enum Buf { bufE = 0, bufC = 1, bufP0 = 2, bufP1 = 3 };

struct VLayer {
    int32_t kind, relu;
    int32_t n_out, n_out_tiles, n_out_pad8;
    int32_t n_seg, seg_buf[2], seg_pad[2], seg_real[2];   // input = concat of <= 2 LDS regions; padded (multiple of 8) and real row counts
    int32_t k8;                                           // (seg_pad[0] + seg_pad[1]) / 8
    int32_t out_buf;
    int32_t w_off, b_off, in_real;                        // flat parameter offsets, real input width (row stride of the weight)
    int32_t tail_w_off, tail_b_off;                       // head only: the raw-sigma row (its last row) is the sigma layer's own weight / bias
    int32_t frag_off, fragT_off, t_tiles, t_k8;           // fragment buffers (floats); W^T: t_tiles row tiles over the first segment, t_k8 = n_out_pad8 / 8
    int32_t act_seg_row[2], act_out_row, dz_row;          // ACT rows: inputs, saved post-activation output (-1: none), dZ
};

struct VNet {
    VLayer layer[kMaxLayers];
    int32_t n_layers, head_index, W, Wc, buf_rows, act_rows;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// mlp.py:184-203: [x, sin(2^i x_d) (deg-major), sin(2^i x_d + pi/2)]; rows past the real width are zero
__device__ __forceinline__ float enc_row(const float x[3], int row, int deg) {
    if (row < 3) return x[row];
    const int q = row - 3;
    if (q >= 2 * 3 * deg) return 0.0f;
    const int shifted = q >= 3 * deg;
    const int k = shifted ? q - 3 * deg : q;
    const float xb = x[k % 3] * (float)(1 << (k / 3));
    return sinf(shifted ? xb + 1.57079637050628662109375f : xb);      // float32(0.5 * math.pi)
}

struct FwdArgs {
    const float *params, *frags;
    const float *pos, *dirs;
    int64_t n;
    int32_t samples_per_dir;
    float *rgb, *sigma;
    float *act;            // NULL: inference
    int32_t density_only, waves;
};

__device__ __forceinline__ float *lds_region(float *wave_base, int buf, int buf_rows) {
    return buf == bufE ? wave_base : (buf == bufC ? wave_base + kEncRows * 32 : wave_base + (kEncRows + kCondRows) * 32 + (buf - bufP0) * buf_rows * 32);
}

__global__ void __launch_bounds__(256) vanilla_forward_kernel(const VNet net, const FwdArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    if (wave >= a.waves) return;
    float *base = lds + (size_t)wave * (kEncRows + kCondRows + 2 * net.buf_rows) * 32;
    const int64_t n_tiles = (a.n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * a.waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * a.waves) {
        const int64_t s = tile * 32 + c;
        const bool valid = s < a.n;
        float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = a.pos[3 * s + k];
            if (!a.density_only)
#pragma unroll
                for (int k = 0; k < 3; ++k) d[k] = a.dirs[3 * (s / a.samples_per_dir) + k];
        }
        float *E = base, *C = base + kEncRows * 32;
        float *act_tile = a.act ? a.act + (size_t)tile * net.act_rows * 32 : nullptr;
        for (int r = h; r < kEncRows; r += 2) {
            const float v = enc_row(x, r, kPosDeg);
            E[r * 32 + c] = v;
            if (act_tile) act_tile[r * 32 + c] = v;
        }
        if (!a.density_only)
            for (int r = h; r < kCondRows; r += 2) {
                const float v = enc_row(d, r, kDirDeg);
                C[r * 32 + c] = v;
                if (act_tile) act_tile[(kEncRows + r) * 32 + c] = v;
            }
        __builtin_amdgcn_wave_barrier();
        const int n_layers = a.density_only ? net.head_index + 1 : net.n_layers;
        for (int li = 0; li < n_layers; ++li) {
            const VLayer &L = net.layer[li];
            const float *in0 = lds_region(base, L.seg_buf[0], net.buf_rows);
            const float *in1 = L.n_seg > 1 ? lds_region(base, L.seg_buf[1], net.buf_rows) : in0;
            float *out = L.out_buf >= 0 ? lds_region(base, L.out_buf, net.buf_rows) : nullptr;
            const float4 *frag = reinterpret_cast<const float4 *>(a.frags + L.frag_off) + lane;
            // density only: of the head, only the row tile that holds the raw-sigma row
            const int rt0 = (a.density_only && L.kind == kHead) ? L.n_out_tiles - 1 : 0;
            for (int rt = rt0; rt < L.n_out_tiles; ++rt) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                const float4 *fr = frag + (size_t)rt * L.k8 * 64;
                const int k8_0 = L.seg_pad[0] >> 3;
                for (int kq = 0; kq < L.k8; ++kq) {
                    const float4 w = fr[(size_t)kq * 64];
                    const float *src = kq < k8_0 ? in0 + (kq * 8 + h) * 32 + c : in1 + ((kq - k8_0) * 8 + h) * 32 + c;
                    const float b0 = src[0], b1 = src[64], b2 = src[128], b3 = src[192];
                    acc = mfma32(w.x, b0, acc); acc = mfma32(w.y, b1, acc); acc = mfma32(w.z, b2, acc); acc = mfma32(w.w, b3, acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 32 * rt + 8 * g + 4 * h + i;
                        if (row >= L.n_out) continue;
                        const bool sigma_row = L.kind == kHead && row == L.n_out - 1;
                        float v = acc[4 * g + i] + a.params[sigma_row ? L.tail_b_off : L.b_off + row];
                        if (L.relu) v = fmaxf(v, 0.0f);
                        if (sigma_row) {                                          // raw sigma (mlp.py:150-152), relu at :235 / :243
                            if (valid && a.sigma) a.sigma[s] = fmaxf(v, 0.0f);
                        } else if (L.kind == kRgbOut) {                           // mlp.py:243 sigmoid
                            if (valid && a.rgb) a.rgb[3 * s + row] = 1.0f / (1.0f + expf(-v));
                        } else {
                            out[row * 32 + c] = v;
                            if (act_tile && L.act_out_row >= 0) act_tile[(L.act_out_row + row) * 32 + c] = v;
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

struct BwdArgs {
    const float *fragsT;
    const float *d_rgb, *d_sigma, *rgb, *sigma;
    int64_t n;
    float *act;
    int32_t waves;
};

__global__ void __launch_bounds__(256) vanilla_backward_kernel(const VNet net, const BwdArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    if (wave >= a.waves) return;
    float *base = lds + (size_t)wave * 2 * net.buf_rows * 32;
    const int64_t n_tiles = (a.n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * a.waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * a.waves) {
        const int64_t s = tile * 32 + c;
        const bool valid = s < a.n;
        float *act_tile = a.act + (size_t)tile * net.act_rows * 32;
        float *cur = base, *nxt = base + net.buf_rows * 32;
        // dZ of the rgb output layer: d(sigmoid) (rows 0..2), rows 3..7 zero
        for (int r = h; r < 8; r += 2) {
            float v = 0.0f;
            if (valid && r < 3) { const float y = a.rgb[3 * s + r]; v = a.d_rgb[3 * s + r] * y * (1.0f - y); }
            cur[r * 32 + c] = v;
        }
        __builtin_amdgcn_wave_barrier();
        for (int li = net.n_layers - 1; li >= 0; --li) {
            const VLayer &L = net.layer[li];
            for (int r = h; r < L.n_out; r += 2) act_tile[(L.dz_row + r) * 32 + c] = cur[r * 32 + c];     // for the weight gradients
            if (li == 0) break;
            const VLayer &P = net.layer[li - 1];
            // dZ_prev = mask( W_L^T[:, rows of the first input segment] . dZ_L )
            const float4 *frag = reinterpret_cast<const float4 *>(a.fragsT + L.fragT_off) + lane;
            const int n_rows = L.seg_real[0];
            for (int rt = 0; rt < L.t_tiles; ++rt) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                const float4 *fr = frag + (size_t)rt * L.t_k8 * 64;
                for (int kq = 0; kq < L.t_k8; ++kq) {
                    const float4 w = fr[(size_t)kq * 64];
                    const float *src = cur + (kq * 8 + h) * 32 + c;
                    acc = mfma32(w.x, src[0], acc); acc = mfma32(w.y, src[64], acc); acc = mfma32(w.z, src[128], acc); acc = mfma32(w.w, src[192], acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 32 * rt + 8 * g + 4 * h + i;
                        if (row >= n_rows) continue;
                        float v = acc[4 * g + i];
                        if (P.relu && !(act_tile[(P.act_out_row + row) * 32 + c] > 0.0f)) v = 0.0f;
                        nxt[row * 32 + c] = v;
                    }
            }
            // rows up to the next multiple of 8 feed the k loop of the next step: zero, except the raw-sigma row of the head
            for (int r = n_rows + h; r < P.n_out_pad8; r += 2) {
                float v = 0.0f;
                if (P.kind == kHead && r == P.n_out - 1 && valid) v = a.sigma[s] > 0.0f ? a.d_sigma[s] : 0.0f;      // relu' (mlp.py:243)
                nxt[r * 32 + c] = v;
            }
            __builtin_amdgcn_wave_barrier();
            float *t = cur; cur = nxt; nxt = t;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// dW[o][i] += sum_samples dZ[o][s] * In[i][s], db[o] += sum_samples dZ[o][s].  One wave = one (layer, 32-row out tile, 32-row
// in tile, range of sample tiles).  Both operands are rows of the tile-major ACT buffer; a lane loads four float4 per
// operand and tile: its row's samples 8t + 4h .. 8t + 4h + 3 (the k-slot -> sample assignment is the same for A and B,
// which is all a contraction needs).
struct WJob {
    int32_t dz_row, n_out, o0;        // A rows: dz_row + o0 + r; rows o0 .. n_out - 1 of this tile exist
    int32_t row_shift;                // parameter row = o - row_shift (the head's last row is row 0 of the sigma layer)
    int32_t in_row[32];               // B row of lane r (clamped to a valid row), per job
    int32_t col[32];                  // weight column of lane r, -1: padding
    int32_t w_off, in_real, b_off;    // b_off >= 0: this job also accumulates the bias gradient
};

__global__ void __launch_bounds__(256) vanilla_wgrad_kernel(const WJob *__restrict__ jobs, int n_jobs, int split, const float *__restrict__ act,
                                                            int64_t n_tiles, int act_rows, float *__restrict__ grad) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= n_jobs * split) return;
    const WJob &jb = jobs[wid / split];
    const int part = wid % split;
    const int64_t t0 = n_tiles * part / split, t1 = n_tiles * (part + 1) / split;
    int arow = jb.dz_row + jb.o0 + r;
    if (arow >= act_rows) arow = act_rows - 1;
    const int brow = jb.in_row[r];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float bsum = 0.0f;
    for (int64_t t = t0; t < t1; ++t) {
        const float *tb = act + (size_t)t * act_rows * 32;
        const float4 *pa = reinterpret_cast<const float4 *>(tb + arow * 32) + h;
        const float4 *pb = reinterpret_cast<const float4 *>(tb + brow * 32) + h;
        float4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = pa[2 * q]; bv[q] = pb[2 * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = mfma32(av[q].x, bv[q].x, acc); acc = mfma32(av[q].y, bv[q].y, acc);
            acc = mfma32(av[q].z, bv[q].z, acc); acc = mfma32(av[q].w, bv[q].w, acc);
            bsum += (av[q].x + av[q].y) + (av[q].z + av[q].w);
        }
    }
    const int col = jb.col[r];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = jb.o0 + 8 * g + 4 * h + i;
            if (o < jb.n_out && col >= 0) atomicAdd(grad + jb.w_off + (int64_t)(o - jb.row_shift) * jb.in_real + col, acc[4 * g + i]);
        }
    if (jb.b_off >= 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (h == 0 && jb.o0 + r < jb.n_out) atomicAdd(grad + jb.b_off + (jb.o0 + r - jb.row_shift), bsum);
    }
}

__global__ void __launch_bounds__(256) gather_f32_kernel(const int32_t *__restrict__ src_idx, const float *__restrict__ params, float *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = src_idx[i];
    dst[i] = s >= 0 ? params[s] : 0.0f;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace
}  // namespace mnf

using namespace mnf;

struct mnf_vanilla_s {
    mnf_vanilla_config cfg;
    VNet net;
    int64_t n_params;
    std::vector<int64_t> tensor_off;              // weight, bias per layer in named_parameters() order (head split: sigma, bottleneck)
    std::vector<int32_t> tensor_rows, tensor_cols;
    std::vector<int32_t> frag_src, fragT_src;     // host gather tables
    std::vector<WJob> jobs;
    float *d_params = nullptr, *d_frags = nullptr, *d_fragsT = nullptr;
    int32_t *d_frag_src = nullptr, *d_fragT_src = nullptr;
    WJob *d_jobs = nullptr;
    bool loaded = false;
};

namespace {

// Builds the layer chain, the flat parameter layout (reference `named_parameters()` order: base hidden layers, sigma layer,
// bottleneck layer, rgb hidden layers, rgb output; weight [out][in] then bias per layer), ACT row assignments, fragment
// gather tables and weight-gradient jobs.
int build_net(mnf_vanilla_s *v) {
    const int D = v->cfg.net_depth, W = v->cfg.net_width, Dc = v->cfg.net_depth_condition, Wc = v->cfg.net_width_condition;
    const int skip = v->cfg.skip_layer;
    VNet &net = v->net;
    std::memset(&net, 0, sizeof(net));
    net.W = W; net.Wc = Wc;
    int64_t off = 0;
    int row = kEncRows + kCondRows;      // ACT rows: E, C, then per-layer outputs, then dZ rows
    int n = 0;
    auto add_tensor = [&](int rows, int cols) { v->tensor_off.push_back(off); v->tensor_rows.push_back(rows); v->tensor_cols.push_back(cols); off += (int64_t)rows * cols; };
    // ---- base hidden layers (mlp.py:45-60, :86-96)
    int prev_buf = bufE, prev_real = 63, prev_pad = kEncRows, prev_act_row = 0;
    bool concat_E = false;
    for (int l = 0; l < D; ++l) {
        VLayer &L = net.layer[n++];
        L.kind = kHidden; L.relu = 1; L.n_out = W;
        L.n_seg = concat_E ? 2 : 1;
        L.seg_buf[0] = prev_buf; L.seg_pad[0] = prev_pad; L.seg_real[0] = prev_real; L.act_seg_row[0] = prev_act_row;
        if (concat_E) { L.seg_buf[1] = bufE; L.seg_pad[1] = kEncRows; L.seg_real[1] = 63; L.act_seg_row[1] = 0; }
        L.out_buf = (l & 1) ? bufP1 : bufP0;
        L.in_real = L.seg_real[0] + (concat_E ? 63 : 0);
        L.w_off = (int32_t)off; add_tensor(W, L.in_real);
        L.b_off = (int32_t)off; add_tensor(W, 1);
        L.act_out_row = row; row += W;
        prev_buf = L.out_buf; prev_real = W; prev_pad = W; prev_act_row = L.act_out_row;
        concat_E = skip > 0 && (l % skip == 0) && l > 0;                       // mlp.py:52-57: the NEXT layer sees [x, inputs]
    }
    // ---- head: rows 0..W-1 bottleneck (mlp.py:143), row W raw sigma (mlp.py:140); parameters: sigma first, then bottleneck
    {
        VLayer &L = net.layer[n];
        net.head_index = n++;
        L.kind = kHead; L.relu = 0; L.n_out = W + 1;
        L.n_seg = concat_E ? 2 : 1;
        L.seg_buf[0] = prev_buf; L.seg_pad[0] = prev_pad; L.seg_real[0] = prev_real; L.act_seg_row[0] = prev_act_row;
        if (concat_E) { L.seg_buf[1] = bufE; L.seg_pad[1] = kEncRows; L.seg_real[1] = 63; L.act_seg_row[1] = 0; }
        L.out_buf = prev_buf == bufP0 ? bufP1 : bufP0;
        L.in_real = L.seg_real[0] + (concat_E ? 63 : 0);
        L.tail_w_off = (int32_t)off; add_tensor(1, L.in_real);
        L.tail_b_off = (int32_t)off; add_tensor(1, 1);
        L.w_off = (int32_t)off; add_tensor(W, L.in_real);          // bottleneck rows
        L.b_off = (int32_t)off; add_tensor(W, 1);
        L.act_out_row = row; row += W;                              // bottleneck output = first input segment of rgb hidden 0
        prev_buf = L.out_buf; prev_act_row = L.act_out_row;
    }
    // ---- rgb hidden layers (input [bottleneck, dir encoding], mlp.py:144-151, :163-164)
    for (int l = 0; l < Dc; ++l) {
        VLayer &L = net.layer[n++];
        L.kind = kHidden; L.relu = 1; L.n_out = Wc;
        if (l == 0) {
            L.n_seg = 2;
            L.seg_buf[0] = prev_buf; L.seg_pad[0] = W; L.seg_real[0] = W; L.act_seg_row[0] = prev_act_row;
            L.seg_buf[1] = bufC; L.seg_pad[1] = kCondRows; L.seg_real[1] = 27; L.act_seg_row[1] = kEncRows;
            L.in_real = W + 27;
        } else {
            L.n_seg = 1;
            L.seg_buf[0] = prev_buf; L.seg_pad[0] = Wc; L.seg_real[0] = Wc; L.act_seg_row[0] = prev_act_row;
            L.in_real = Wc;
        }
        L.out_buf = prev_buf == bufP0 ? bufP1 : bufP0;
        L.w_off = (int32_t)off; add_tensor(Wc, L.in_real);
        L.b_off = (int32_t)off; add_tensor(Wc, 1);
        L.act_out_row = row; row += Wc;
        prev_buf = L.out_buf; prev_act_row = L.act_out_row;
    }
    {
        VLayer &L = net.layer[n++];
        L.kind = kRgbOut; L.relu = 0; L.n_out = 3; L.n_seg = 1;
        L.seg_buf[0] = prev_buf; L.seg_pad[0] = Wc; L.seg_real[0] = Wc; L.act_seg_row[0] = prev_act_row;
        L.out_buf = -1; L.in_real = Wc;
        L.w_off = (int32_t)off; add_tensor(3, Wc);
        L.b_off = (int32_t)off; add_tensor(3, 1);
        L.act_out_row = -1;
    }
    net.n_layers = n;
    v->n_params = off;
    int buf_rows = 8;
    for (int i = 0; i < n; ++i) {
        VLayer &L = net.layer[i];
        L.n_out_tiles = (L.n_out + 31) / 32;
        L.n_out_pad8 = round_up(L.n_out, 8);
        L.k8 = (L.seg_pad[0] + (L.n_seg > 1 ? L.seg_pad[1] : 0)) / 8;
        L.dz_row = row; row += L.n_out_pad8;
        L.t_tiles = (L.seg_real[0] + 31) / 32;
        L.t_k8 = L.n_out_pad8 / 8;
        if (L.out_buf >= 0 && L.n_out_pad8 > buf_rows) buf_rows = L.n_out_pad8;
    }
    net.buf_rows = buf_rows;
    net.act_rows = row;
    const VLayer &H = net.layer[net.head_index];
    const int64_t sig_w_off = H.tail_w_off, sig_b_off = H.tail_b_off;
    // parameter index of weight (o, i) / bias o of layer li (the head's last row is the sigma layer)
    auto w_index = [&](int li, int o, int i) -> int32_t {
        const VLayer &L = net.layer[li];
        if (li == net.head_index && o == H.n_out - 1) return (int32_t)(sig_w_off + i);
        return (int32_t)(L.w_off + (int64_t)o * L.in_real + i);
    };
    // padded concatenated input position -> real input column (-1: padding)
    auto in_col = [&](const VLayer &L, int kp) -> int {
        if (kp < L.seg_pad[0]) return kp < L.seg_real[0] ? kp : -1;
        const int q = kp - L.seg_pad[0];
        return (L.n_seg > 1 && q < L.seg_real[1]) ? L.seg_real[0] + q : -1;
    };
    // ---- forward fragments: [rt][kq][lane] float4, element j = W[32 rt + r][8 kq + 2 j + h]
    for (int li = 0; li < n; ++li) {
        VLayer &L = net.layer[li];
        L.frag_off = (int32_t)v->frag_src.size();
        for (int rt = 0; rt < L.n_out_tiles; ++rt)
            for (int kq = 0; kq < L.k8; ++kq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int o = 32 * rt + (lane & 31), col = in_col(L, 8 * kq + 2 * j + (lane >> 5));
                        v->frag_src.push_back(o < L.n_out && col >= 0 ? w_index(li, o, col) : -1);
                    }
        // ---- transposed fragments: rows = first-segment inputs, k = outputs: element j = W[8 kq + 2 j + h][32 rt + r]
        L.fragT_off = (int32_t)v->fragT_src.size();
        for (int rt = 0; rt < L.t_tiles; ++rt)
            for (int kq = 0; kq < L.t_k8; ++kq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int i = 32 * rt + (lane & 31), o = 8 * kq + 2 * j + (lane >> 5);
                        v->fragT_src.push_back(o < L.n_out && i < L.seg_real[0] ? w_index(li, o, i) : -1);
                    }
    }
    // ---- weight-gradient jobs
    for (int li = 0; li < n; ++li) {
        const VLayer &L = net.layer[li];
        const int in_pad = L.seg_pad[0] + (L.n_seg > 1 ? L.seg_pad[1] : 0);
        for (int ot = 0; ot < L.n_out_tiles; ++ot)
            for (int it = 0; it < (in_pad + 31) / 32; ++it) {
                // the head's sigma row has its own weight vector: give it a job of its own (one row) per in tile
                const bool head_tail = li == net.head_index && ot == L.n_out_tiles - 1;
                WJob jb;
                jb.dz_row = L.dz_row; jb.o0 = 32 * ot; jb.row_shift = 0;
                jb.n_out = L.n_out;
                jb.w_off = L.w_off; jb.in_real = L.in_real; jb.b_off = it == 0 ? L.b_off : -1;
                for (int r = 0; r < 32; ++r) {
                    const int kp = 32 * it + r;
                    const int col = kp < in_pad ? in_col(L, kp) : -1;
                    jb.col[r] = col;
                    int arow = 0;
                    if (kp < L.seg_pad[0]) arow = L.act_seg_row[0] + kp;
                    else if (kp < in_pad) arow = L.act_seg_row[1] + (kp - L.seg_pad[0]);
                    jb.in_row[r] = arow;
                }
                if (head_tail) {   // the tile's only row (W) is row 0 of the sigma layer
                    jb.row_shift = 32 * ot;
                    jb.w_off = (int32_t)sig_w_off;
                    jb.b_off = it == 0 ? (int32_t)sig_b_off : -1;
                }
                v->jobs.push_back(jb);
            }
    }
    return MNF_OK;
}

size_t fwd_lds_bytes(const VNet &net, int waves) { return (size_t)waves * (kEncRows + kCondRows + 2 * net.buf_rows) * 32 * sizeof(float); }
size_t bwd_lds_bytes(const VNet &net, int waves) { return (size_t)waves * 2 * net.buf_rows * 32 * sizeof(float); }

int pick_waves(size_t per_wave_bytes) {
    int w = (int)((size_t)150 * 1024 / per_wave_bytes);
    return w > 4 ? 4 : w;
}

}  // namespace

extern "C" int mnf_vanilla_destroy(mnf_vanilla_t v) {
    if (!v) return MNF_OK;
    for (void *p : {(void *)v->d_params, (void *)v->d_frags, (void *)v->d_fragsT, (void *)v->d_frag_src, (void *)v->d_fragT_src, (void *)v->d_jobs})
        if (p) (void)hipFree(p);
    delete v;
    return MNF_OK;
}

extern "C" int mnf_vanilla_create(const mnf_vanilla_config *cfg, mnf_vanilla_t *out) {
    MNF_REQUIRE(cfg && out, "vanilla_create: null argument");
    MNF_REQUIRE(cfg->struct_size == sizeof(mnf_vanilla_config), "vanilla_create: cfg->struct_size is %u, this library's mnf_vanilla_config has %zu bytes (MNF_INIT)",
                cfg->struct_size, sizeof(mnf_vanilla_config));
    MNF_REQUIRE(cfg->net_depth >= 1 && cfg->net_depth_condition >= 1, "vanilla_create: net_depth and net_depth_condition must be >= 1");
    MNF_REQUIRE(cfg->net_depth + cfg->net_depth_condition + 2 <= kMaxLayers, "vanilla_create: too many layers (max %d)", kMaxLayers - 2);
    MNF_REQUIRE(cfg->net_width >= 32 && cfg->net_width % 32 == 0 && cfg->net_width <= 512 && cfg->net_width_condition >= 32 &&
                    cfg->net_width_condition % 32 == 0 && cfg->net_width_condition <= 512,
                "vanilla_create: widths must be multiples of 32 in [32, 512] (got %d, %d)", cfg->net_width, cfg->net_width_condition);
    mnf_vanilla_s *v = new mnf_vanilla_s();
    v->cfg = *cfg;
    build_net(v);
    const size_t per_wave = fwd_lds_bytes(v->net, 1);
    if (pick_waves(per_wave) < 1) { delete v; set_error("vanilla_create: network too wide for the LDS staging"); return MNF_ERR_UNSUPPORTED; }
    hipError_t e = hipMalloc((void **)&v->d_params, (size_t)v->n_params * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_frags, v->frag_src.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_fragsT, v->fragT_src.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_frag_src, v->frag_src.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_fragT_src, v->fragT_src.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_jobs, v->jobs.size() * sizeof(WJob));
    if (e == hipSuccess) e = hipMemcpy(v->d_frag_src, v->frag_src.data(), v->frag_src.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_fragT_src, v->fragT_src.data(), v->fragT_src.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_jobs, v->jobs.data(), v->jobs.size() * sizeof(WJob), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)vanilla_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)vanilla_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        set_error("vanilla_create: %s", hipGetErrorString(e));
        mnf_vanilla_destroy(v);
        return MNF_ERR_HIP;
    }
    *out = v;
    return MNF_OK;
}

extern "C" int64_t mnf_vanilla_param_count(mnf_vanilla_t v) { return v ? v->n_params : -1; }

extern "C" int mnf_vanilla_param_layout_host(mnf_vanilla_t v, int32_t max_tensors, int32_t *n_tensors_host, int64_t *offsets_host,
                                             int32_t *rows_host, int32_t *cols_host) {
    MNF_REQUIRE(v && n_tensors_host, "vanilla_param_layout: null argument");
    const int n = (int)v->tensor_off.size();
    *n_tensors_host = n;
    for (int i = 0; i < n && i < max_tensors; ++i) {
        if (offsets_host) offsets_host[i] = v->tensor_off[i];
        if (rows_host) rows_host[i] = v->tensor_rows[i];
        if (cols_host) cols_host[i] = v->tensor_cols[i];
    }
    return MNF_OK;
}

extern "C" int mnf_vanilla_set_params(mnf_vanilla_t v, const float *flat_params, mnf_stream_t stream) {
    MNF_REQUIRE(v && flat_params, "vanilla_set_params: null argument");
    hipStream_t s = as_stream(stream);
    MNF_HIP(hipMemcpyAsync(v->d_params, flat_params, (size_t)v->n_params * 4, hipMemcpyDeviceToDevice, s));
    const int64_t nf = (int64_t)v->frag_src.size(), nt = (int64_t)v->fragT_src.size();
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)ceil_div(nf, 256)), dim3(256), 0, s, v->d_frag_src, v->d_params, v->d_frags, nf);
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)ceil_div(nt, 256)), dim3(256), 0, s, v->d_fragT_src, v->d_params, v->d_fragsT, nt);
    v->loaded = true;
    return launch_status("gather_f32_kernel");
}

extern "C" int64_t mnf_vanilla_train_workspace_bytes(mnf_vanilla_t v, int64_t n) {
    if (!v || n < 0) return -1;
    return ceil_div(n > 0 ? n : 1, 32) * (int64_t)v->net.act_rows * 32 * (int64_t)sizeof(float);
}

static int vanilla_forward(mnf_vanilla_t v, const float *positions, const float *directions, int64_t n, int32_t samples_per_direction,
                           float *rgb, float *sigma, void *workspace, int64_t workspace_bytes, bool density_only, hipStream_t s) {
    MNF_REQUIRE(v && v->loaded, "vanilla_forward: parameters not loaded");
    MNF_REQUIRE(n >= 0, "vanilla_forward: negative n");
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(positions && (density_only || (directions && samples_per_direction >= 1)), "vanilla_forward: null pointer");
    if (workspace) {
        const int64_t need = mnf_vanilla_train_workspace_bytes(v, n);
        if (workspace_bytes < need) { set_error("vanilla_forward: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need); return MNF_ERR_WORKSPACE; }
    }
    FwdArgs a;
    a.params = v->d_params; a.frags = v->d_frags; a.pos = positions; a.dirs = directions; a.n = n;
    a.samples_per_dir = samples_per_direction > 0 ? samples_per_direction : 1;
    a.rgb = rgb; a.sigma = sigma; a.act = (float *)workspace; a.density_only = density_only ? 1 : 0;
    a.waves = pick_waves(fwd_lds_bytes(v->net, 1));
    const int64_t tiles = ceil_div(n, 32);
    int64_t grid = ceil_div(tiles, a.waves);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(vanilla_forward_kernel, dim3((unsigned)grid), dim3(256), fwd_lds_bytes(v->net, a.waves), s, v->net, a);
    return launch_status("vanilla_forward_kernel");
}

extern "C" int mnf_vanilla_forward(mnf_vanilla_t v, const float *positions, const float *directions, int64_t n,
                                   int32_t samples_per_direction, float *rgb, float *sigma, void *workspace, int64_t workspace_bytes,
                                   mnf_stream_t stream) {
    return vanilla_forward(v, positions, directions, n, samples_per_direction, rgb, sigma, workspace, workspace_bytes, false, as_stream(stream));
}

extern "C" int mnf_vanilla_density(mnf_vanilla_t v, const float *positions, int64_t n, float *sigma, mnf_stream_t stream) {
    return vanilla_forward(v, positions, nullptr, n, 1, nullptr, sigma, nullptr, 0, true, as_stream(stream));
}

extern "C" int mnf_vanilla_backward(mnf_vanilla_t v, const float *d_rgb, const float *d_sigma, const float *rgb, const float *sigma,
                                    int64_t n, void *workspace, int64_t workspace_bytes, float *grad_flat, mnf_stream_t stream) {
    MNF_REQUIRE(v && v->loaded && grad_flat, "vanilla_backward: bad arguments");
    hipStream_t s = as_stream(stream);
    MNF_HIP(hipMemsetAsync(grad_flat, 0, (size_t)v->n_params * 4, s));
    if (n == 0) return MNF_OK;
    MNF_REQUIRE(d_rgb && d_sigma && rgb && sigma && workspace && n > 0, "vanilla_backward: null pointer");
    const int64_t need = mnf_vanilla_train_workspace_bytes(v, n);
    if (workspace_bytes < need) { set_error("vanilla_backward: workspace too small"); return MNF_ERR_WORKSPACE; }
    BwdArgs a;
    a.fragsT = v->d_fragsT; a.d_rgb = d_rgb; a.d_sigma = d_sigma; a.rgb = rgb; a.sigma = sigma; a.n = n; a.act = (float *)workspace;
    a.waves = pick_waves(bwd_lds_bytes(v->net, 1));
    const int64_t tiles = ceil_div(n, 32);
    int64_t grid = ceil_div(tiles, a.waves);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(vanilla_backward_kernel, dim3((unsigned)grid), dim3(256), bwd_lds_bytes(v->net, a.waves), s, v->net, a);
    int rc = launch_status("vanilla_backward_kernel");
    if (rc) return rc;
    // the tail tile (n % 32 != 0) must not contribute: its dZ columns are zero (gradients of invalid samples are zero) by construction
    const int n_jobs = (int)v->jobs.size();
    int split = 256 * 8 / n_jobs;
    if (split > tiles / 8) split = (int)(tiles / 8);
    if (split < 1) split = 1;
    hipLaunchKernelGGL(vanilla_wgrad_kernel, dim3((unsigned)ceil_div((int64_t)n_jobs * split, 4)), dim3(256), 0, s, v->d_jobs, n_jobs, split,
                       (const float *)workspace, tiles, v->net.act_rows, grad_flat);
    return launch_status("vanilla_wgrad_kernel");
}
