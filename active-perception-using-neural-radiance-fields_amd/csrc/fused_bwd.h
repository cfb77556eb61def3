// The fused backward of the radiance field (W = 128, NH <= 2): forward recompute + backward-data + weight gradients of one 64-sample
// tile in ONE kernel, no activation dump.  Replaces, for the shapes it supports, the pair dgrad_kernel / wgrad_kernel of train.hip and the
// ~2.4 KB per sample of 16-bit activations they exchange through HBM (the forward then only leaves its encoded inputs, 160 B per sample).
// Reference: the tiny-cuda-nn backward behind `loss.backward()` (scripts/pipeline.py:518; modules built at
// perception/models/radiance_fields/ngp.py:108-169).
//
// Why a different decomposition than the inference kernel.  There a wave owns a 64-sample tile and keeps the whole MLP chain in its
// registers.  Weight gradients need dW[n][k] = sum over samples of dZ[n][c] * In[k][c] for all nine matrices: 44 accumulator tiles of
// 32 x 32 fp32 = 704 VGPRs, which no wave can hold, and LDS (160 KB) holds neither them (176 KB) nor a tile's activations beside
// the weights.  Here a WORKGROUP OF FOUR WAVES (one per SIMD, up to 512 registers each) owns a tile and every wave owns one 32-row
// tile of every layer ("row-tile owner"):
//   * chain orientation F (feature on the register index, sample on the lane): wave q computes rows 32q .. 32q+31 of a layer,
//     D = W(q) * In, and publishes them as B fragments in LDS for the next layer (lane-linear 16-byte slots, no transposes);
//   * sample-major orientation S (sample on the register index, feature on the lane): THE SAME two fragments with the operand roles
//     swapped, D' = In^T * W(q)^T, give the tile transposed.  A packed S tile is directly an MFMA operand of a product that contracts
//     over SAMPLES, which is what a weight gradient is: dW(nt, kt) = sum over ct, s of mfma(a = pack(dZ_S(ct, nt), s),
//     b = pack(In_S(ct, kt), s)).  So no LDS transpose and no transposed reads anywhere; activations that come from elsewhere (hash
//     features, SH, output-layer gradients) are turned into S tiles by one MFMA against an identity fragment (exact);
//   * forward S tiles of the wave's own feature tile stay in its registers until the backward reaches that layer (64 VGPRs), backward S
//     tiles are published once to LDS (16 KB) for the other column owners; the 11 accumulator tiles a wave owns (176 VGPRs, AGPRs in
//     practice) live across the whole persistent tile loop and leave with float atomics once per workgroup;
//   * forward weights (A fragments, and B fragments of the S products) come from the handle's fragment table in LDS (78 KB: the heads'
//     output layers are not recomputed); the transposed fragments of the wave's own row tiles are read from the table in L2 where they are used
//     (15 fragments per tile), the two transposed matrices several waves share (base input, heads' input) from LDS (24 KB).  158 KB of LDS in all.
// Everything is MFMA + lane-linear LDS traffic; nine workgroup barriers per tile.
#pragma once
// (included by train.hip INSIDE namespace mnf::f16 / mnf::bf16, after LayoutT, count_here, sat_half and WgradJob)

constexpr int kFusedThreads = 256;

// Phase stamps of workgroup 0 (experiment builds with -DMNF_FUSED_STAMPS only: tools/exp_fused_stamps.py): shader clock at tile start and on both sides of every barrier
#ifdef MNF_FUSED_STAMPS
constexpr int kStampTiles = 24, kStampSlots = 24;
__device__ unsigned long long g_fused_stamps[kStampTiles * 4 * kStampSlots];
#define MNF_STAMP(k) do { if (blockIdx.x == 0 && lane == 0 && stamp_it < kStampTiles) g_fused_stamps[(stamp_it * 4 + q) * kStampSlots + (k)] = __builtin_readcyclecounter(); } while (0)
#define MNF_SYNC(k) do { MNF_STAMP(2 * (k) - 1); __syncthreads(); MNF_STAMP(2 * (k)); } while (0)
#else
#define MNF_STAMP(k) do { } while (0)
#define MNF_SYNC(k) __syncthreads()
#endif

struct FusedBwdArgs {
    const half8 *frags;      // forward fragment table (Layout<128, NH>)
    const half8 *fragsT;     // transposed fragment table (LayoutT<128, NH>)
    const half8 *enc;        // [tiles][kEncBlocks][64]: hash features + SH fragment written by the forward (field_kernel SAVEK = 2)
    const WgradJob *jobs;
    const float *d_rgb, *d_sigma, *d_sem;   // [N,3], [N], [N,C]
    const float *rgb, *sigma;               // forward outputs [N,3], [N]
    float *dX;                              // [16 levels][Np][4] fp32, un-scaled
    float *g0, *g1, *g2;                    // flat parameter gradients (base, rgb head, semantic head)
    int64_t n, Np;
    const int64_t *n_dev;
    int C, out_fp16;
    float loss_scale;
    int chunk, n_chunks;     // this launch covers tile range `chunk` of `n_chunks` (chunk_tiles)
};

__device__ __forceinline__ half8 pack8(const f32x16 &acc, int s) {
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (half_t)acc[8 * s + j];
    return r;
}

// clear the 16-bit elements of `v` whose mask bit (frag_mask_bit order) is 0
__device__ __forceinline__ half8 mask_by_bits(half8 v, uint32_t m) {
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m, i, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)m, 4 + i, 1);
        w[i] &= (lo & 0x0000FFFFu) | (hi & 0xFFFF0000u);
    }
    return __builtin_bit_cast(half8, w);
}

// clear the elements of `v` where the post-ReLU value `ref` (same fragment position) is zero
__device__ __forceinline__ half8 mask_by_nonzero(half8 v, half8 ref) {
    u32x4 w = __builtin_bit_cast(u32x4, v);
    const u32x4 r = __builtin_bit_cast(u32x4, ref);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = ((r[i] & 0x7FFF7FFFu) + 0x7FFF7FFFu) & 0x80008000u;     // bit 15 / 31: the half is non-zero
        w[i] &= (t >> 15) * 0xFFFFu;
    }
    return __builtin_bit_cast(half8, w);
}

// identity B fragments: lane (n = r, h), element j = 1 where n equals the feature that element (h, j) of the OTHER operand carries
__device__ __forceinline__ half8 ident_nat(int r, int h, int base) {      // natural k order: element (h, j) = feature base + 8h + j
    half8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (half_t)(r == base + 8 * h + j ? 1.0f : 0.0f);
    return f;
}
__device__ __forceinline__ half8 ident_acc(int r, int h, int base) {      // accumulator k order: element (h, j) = row base + 8 (j >> 2) + 4h + (j & 3)
    half8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (half_t)(r == base + 8 * (j >> 2) + 4 * h + (j & 3) ? 1.0f : 0.0f);
    return f;
}

// one 1 KiB fragment block of a table in global memory: wave-uniform base and block (scalar address arithmetic) + the lane's 16-byte slot.
// Written this way the load is `global_load_dwordx4 v, v_lane_offset, s[base]`; as `table[block * 64 + lane]` every block got a 64-bit
// VGPR address of its own, hoisted out of the tile loop and spilled.
__device__ __forceinline__ half8 ldg_block(const half8 *table, int block, int lane) {
    const char *p = reinterpret_cast<const char *>(table) + (size_t)block * 1024;
    return *reinterpret_cast<const half8 *>(p + (uint32_t)lane * 16u);
}

__device__ __forceinline__ void zero2(f32x16 (&a)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[ct][i] = 0.0f;
}

// one accumulator tile += sum over the 64 samples of the tile: A = packed dOut S tile, B = packed In S tile
__device__ __forceinline__ void wgrad_acc(f32x16 &acc, const half8 (&a)[CT][2], const half8 (&b)[CT][2]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = mfma(a[ct][s], b[ct][s], acc);
}

__device__ __forceinline__ void flush_tile(const FusedBwdArgs &args, const f32x16 &acc, int job, int r, int h, float inv_scale) {
    const WgradJob &jb = args.jobs[job];
    float *g = jb.buf == 0 ? args.g0 : (jb.buf == 1 ? args.g1 : args.g2);
    const int col = jb.colmap[r];
    if (col < 0) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
        const float v = acc[k] * inv_scale;
        if (row < jb.n_valid && v != 0.0f) atomicAdd(g + jb.param_off + (int64_t)(jb.n0 + row) * jb.stride + col, v);
    }
}

template <int NH>
__global__ void __launch_bounds__(kFusedThreads, 1) fused_bwd_kernel(const FusedBwdArgs args) {
    constexpr int W = 128;
    using L = Layout<W, NH>;
    using LT = LayoutT<W, NH>;
    constexpr int NHH = NH - 1;                         // hidden-to-hidden matrices of the base network
    // forward fragments without the two heads' output layers (the recompute stops at their inputs): blocks [0, o_h_out) and [o_s_in, o_s_out)
    constexpr int kHeadOut = L::KSh;                    // blocks of the rgb head's output layer, cut out of the LDS copy
    constexpr int kFwdBlocks = L::o_s_out - kHeadOut;
    constexpr int oS_in = L::o_s_in - kHeadOut, oS_hid = L::o_s_hid - kHeadOut;
    __shared__ half8 s_w[kFwdBlocks * 64];
    __shared__ half8 s_t[24 * 64];                      // transposed fragments shared by several waves: base input^T (2 row tiles x 8), heads' input^T geo rows (2 x 4)
    __shared__ half8 s_e[2][16 * 64];                   // chain (F) exchange, two buffers: block ks * 2 + ct (base), head * 8 + ks * 2 + ct (heads)
    __shared__ half8 s_sd[16 * 64];                     // packed backward S tiles: block nt * 4 + ct * 2 + s (base), (head * 2 + nt) * 4 + ct * 2 + s (heads)
    __shared__ float s_g[2 * 2 * 8 * 64];               // geo-feature gradients of the two heads: ((head * 2 + ct) * 8 + i) * 64 + lane
    __shared__ half_t s_dyr[64 * 4];                    // the tile's rgb output gradient (through the sigmoid, loss-scaled): [sample][4], slot 3 stays 0
    __shared__ float s_dl[64];                          // ... and its density-logit gradient
    half_t *s_dys = reinterpret_cast<half_t *>(s_g);    // ... and its semantic output gradient [sample][32 padded classes]: lives in s_g's first 4 KB between barrier 1 and barrier 4

    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform, and the compiler knows it)
    const int r = lane & 31, h = lane >> 5;
    const int head = q >> 1, e = q & 1;                 // waves 0, 1: rgb head; 2, 3: semantic head; e: the 32-feature tile of the 64 head neurons
    const int64_t n = count_here(args.n, args.n_dev);
    int64_t tile0, n_tiles;                             // [tile0, n_tiles): this launch's range of 64-sample tiles
    chunk_tiles(n, args.chunk, args.n_chunks, tile0, n_tiles);
    if (tile0 + (int64_t)blockIdx.x >= n_tiles) return;
    for (int i = threadIdx.x; i < kFwdBlocks * 64; i += kFusedThreads) s_w[i] = args.frags[i < L::o_h_out * 64 ? i : i + kHeadOut * 64];
    for (int i = threadIdx.x; i < 24 * 64; i += kFusedThreads) {
        const int b = i >> 6;
        const int src = b < 16 ? LT::o_b1 + b : (b < 20 ? LT::o_r1 + (b - 16) : LT::o_s1 + (b - 20));
        s_t[i] = args.fragsT[src * 64 + (i & 63)];
    }

    // transposed fragments of this wave's own row tiles: read from the table (82 KB, L2 / L1 resident) where they are used.  Held in registers
    // for the whole launch (15 fragments) the compiler spilled them to scratch in the prologue: with 176 accumulator registers and the forward
    // S tiles alive, the tile loop has no 60 registers to spare.
    const int oTo = head == 0 ? LT::o_r3 + e : LT::o_s3 + e * 2;          // head output^T (rgb: 1 k-step, semantic: 2)
    const int oTh = (head == 0 ? LT::o_r2 : LT::o_s2) + e * 4;             // head hidden^T, row tile e
    // weight-gradient accumulators of this wave (jobs: build_tables): base-in (q, kt) x 2, hidden (nt, q) x 4 per matrix, base-out (0, q),
    // head-in (e, 0), head-hidden (nt, e) x 2, head-out (0, e)
    f32x16 a_in[2], a_hid[NHH > 0 ? NHH : 1][4], a_bo, a_hi, a_hh[2], a_ho;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a_in[0][i] = 0.f; a_in[1][i] = 0.f; a_bo[i] = 0.f; a_hi[i] = 0.f; a_hh[0][i] = 0.f; a_hh[1][i] = 0.f; a_ho[i] = 0.f;
#pragma unroll
        for (int l = 0; l < (NHH > 0 ? NHH : 1); ++l)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) a_hid[l][nt][i] = 0.f;
    }
    s_dyr[threadIdx.x] = (half_t)0.0f;
    __syncthreads();
    const float ls = args.loss_scale;
    const int C = args.C;

#ifdef MNF_FUSED_STAMPS
    int stamp_it = -1;
#endif
    for (int64_t tile = tile0 + blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef MNF_FUSED_STAMPS
        ++stamp_it;
#endif
        MNF_STAMP(0);
        // The tile's output gradients: coalesced loads by the whole workgroup here (consecutive lanes = consecutive floats of [N,C] / [N,3] / [N]), staged to
        // LDS behind the first barrier.  (Loaded where they are used, fragment-shaped — 16 predicated 4-byte loads per lane at a 116-byte lane stride, by both
        // waves of the semantic head — this was 31 % of the tile: tools/exp_fused_stamps.py.)
        float pre_sem[8], pre_o;
        {
            const int t = threadIdx.x;
            const int64_t s0 = tile * kWaveSamples;
            const int cls = t & 31;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t smp = s0 + (t >> 5) + 8 * k;
                const bool ok = cls < C && smp < n;
                pre_sem[k] = 0.0f;
                if (C > 0) { const float v = args.d_sem[ok ? smp * C + cls : 0]; pre_sem[k] = ok ? v : 0.0f; }
            }
            if (t < 192) {
                const int64_t gi = s0 * 3 + t;
                const bool ok = gi < 3 * n;
                const float y = args.rgb[ok ? gi : 0], d = args.d_rgb[ok ? gi : 0];
                pre_o = ok ? d * y * (1.0f - y) * ls : 0.0f;                                      // sigmoid'
            } else {
                // trunc_exp backward (ngp.py:34-39): g * exp(min(x, 15)) with exp(x) = sigma (0 outside the aabb)
                const int64_t col = s0 + (t - 192);
                const bool ok = col < n;
                const float sg = args.sigma[ok ? col : 0], d = args.d_sigma[ok ? col : 0];
                pre_o = ok ? d * fminf(sg, 3269017.3724721107f) * ls : 0.0f;
            }
        }
        const half8 *enct = args.enc + tile * (kEncBlocks * 64);      // this tile's encoded inputs (uniform)
        const half8 *ft = args.fragsT;
        asm volatile("" : "+s"(ft));                                   // keep the table's address arithmetic inside the loop (scalar, cheap)
        // ... and everything derived from the lane index: hoisted out of the tile loop, the identity fragments, the per-lane row predicates and the
        // 64-bit addresses of the gradient loads were computed once, spilled to scratch and re-read in every tile
        int rl = r, hl = h;
        asm volatile("" : "+v"(rl), "+v"(hl));
        // =============================================================== forward recompute
        half8 hS[NH][CT][2];                  // S tiles (own feature tile q) of the base hidden activations, post-ReLU, packed
        uint32_t mF[NH];                      // ReLU masks of the F tiles: byte ct * 2 + s
        {   // base layer 0
            half8 xb[CT][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xb[ct][ks] = ldg_block(enct, ks * 2 + ct, lane);
            half8 wa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[ks] = s_w[(L::o_b_in + q * 4 + ks) * 64 + lane];
            uint32_t m = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x16 aF, aS;
#pragma unroll
                for (int i = 0; i < 16; ++i) { aF[i] = 0.0f; aS[i] = 0.0f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { aF = mfma(wa[ks], xb[ct][ks], aF); aS = mfma(xb[ct][ks], wa[ks], aS); }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 f = relu_pack8(aF, s);
                    s_e[0][((2 * q + s) * 2 + ct) * 64 + lane] = f;
                    m |= (uint32_t)frag_mask(f) << (8 * (ct * 2 + s));
                    hS[0][ct][s] = relu_pack8(aS, s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mF[0] = m;
        }
        MNF_SYNC(1);
        {   // (every wave is past the previous tile's last read of s_g)
            const int t = threadIdx.x;
#pragma unroll
            for (int k = 0; k < 8; ++k) s_dys[((t >> 5) + 8 * k) * 32 + (t & 31)] = sat_half(pre_sem[k] * ls);
            if (t < 192) s_dyr[(t / 3) * 4 + (t % 3)] = sat_half(pre_o);
            else s_dl[t - 192] = pre_o;
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {          // hidden layers: read buffer (l - 1) & 1, write l & 1
            uint32_t m = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {       // one 32-column tile at a time: two accumulator tiles live instead of four (register pressure)
                f32x16 aF, aS;
#pragma unroll
                for (int i = 0; i < 16; ++i) { aF[i] = 0.0f; aS[i] = 0.0f; }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const half8 a = s_w[(L::o_b_hid + (l - 1) * 32 + q * 8 + ks) * 64 + lane];
                    const half8 b = s_e[(l - 1) & 1][(ks * 2 + ct) * 64 + lane];
                    aF = mfma(a, b, aF); aS = mfma(b, a, aS);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 f = relu_pack8(aF, s);
                    s_e[l & 1][((2 * q + s) * 2 + ct) * 64 + lane] = f;
                    m |= (uint32_t)frag_mask(f) << (8 * (ct * 2 + s));
                    hS[l][ct][s] = relu_pack8(aS, s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mF[l] = m;
            MNF_SYNC(2);
        }
        constexpr int LB = (NH - 1) & 1;        // buffer that holds the last base activation
        // base output (every wave: 16 rows): geo fragment with tcnn's 1.0 pad in the density slot
        half8 geoF[CT];
        {
            f32x16 bo[CT];
            zero2(bo);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const half8 a = s_w[(L::o_b_out + ks) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bo[ct] = mfma(a, s_e[LB][(ks * 2 + ct) * 64 + lane], bo[ct]);
            }
            if (args.out_fp16) round_outputs_fp16(bo);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int j = 0; j < 8; ++j) geoF[ct][j] = (half_t)bo[ct][j];
                if (hl == 0) geoF[ct][0] = (half_t)1.0f;
            }
        }
        // head layer 1 (rgb: SH + geo, semantic: geo), published to the partner wave through s_e[LB ^ 1]; head layer 2: masks and S tiles only
        half8 h1S[CT][2], h2S[CT][2];
        uint32_t mH1, mH2;
        {
            f32x16 aF[CT], aS[CT];
            zero2(aF); zero2(aS);
            if (head == 0) {
                const half8 a0 = s_w[(L::o_h_in + e * 2) * 64 + lane], a1 = s_w[(L::o_h_in + e * 2 + 1) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const half8 sh = ldg_block(enct, 8 + ct, lane);
                    aF[ct] = mfma(a0, sh, aF[ct]); aF[ct] = mfma(a1, geoF[ct], aF[ct]);
                    aS[ct] = mfma(sh, a0, aS[ct]); aS[ct] = mfma(geoF[ct], a1, aS[ct]);
                }
            } else {
                const half8 a0 = s_w[(oS_in + e) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) { aF[ct] = mfma(a0, geoF[ct], aF[ct]); aS[ct] = mfma(geoF[ct], a0, aS[ct]); }
            }
            uint32_t m = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 f = relu_pack8(aF[ct], s);
                    s_e[LB ^ 1][(head * 8 + (2 * e + s) * 2 + ct) * 64 + lane] = f;
                    m |= (uint32_t)frag_mask(f) << (8 * (ct * 2 + s));
                    h1S[ct][s] = relu_pack8(aS[ct], s);
                }
            mH1 = m;
        }
        MNF_SYNC(3);
        {
            const int wbase = (head == 0 ? L::o_h_hid : oS_hid) + e * 4;
            half8 wa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[ks] = s_w[(wbase + ks) * 64 + lane];
            uint32_t m = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x16 aF, aS;
#pragma unroll
                for (int i = 0; i < 16; ++i) { aF[i] = 0.0f; aS[i] = 0.0f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const half8 b = s_e[LB ^ 1][(head * 8 + ks * 2 + ct) * 64 + lane];
                    aF = mfma(wa[ks], b, aF); aS = mfma(b, wa[ks], aS);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    m |= (uint32_t)frag_mask(relu_pack8(aF, s)) << (8 * (ct * 2 + s));
                    h2S[ct][s] = relu_pack8(aS, s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mH2 = m;
        }

        // =============================================================== backward
        const int64_t fcol0 = tile * kWaveSamples + rl;
        // output-layer gradients as natural-order F fragments (rgb pair: 1 k-step, semantic pair: 2), from the staged copies
        half8 dyF[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) dyF[ct][s][j] = (half_t)0.0f;
            if (head == 0) {
                if (hl == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) dyF[ct][0][k] = s_dyr[(rl + 32 * ct) * 4 + k];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) dyF[ct][s] = *reinterpret_cast<const half8 *>(s_dys + (rl + 32 * ct) * 32 + 16 * s + 8 * hl);
            }
        }
        const int nks_o = head == 0 ? 1 : 2;            // k-steps of the head's output layer (16 / 32 padded rows)
        {   // head output layer: weight gradient (dY_S x h2_S), then dZ2 in both orientations
            half8 dyP[CT][2];
            f32x16 aF[CT], aS[CT];
            zero2(aF); zero2(aS);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x16 t;
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = 0.0f;
                t = mfma(dyF[ct][0], ident_nat(rl, hl, 0), t);
                if (head) t = mfma(dyF[ct][1], ident_nat(rl, hl, 16), t);
                dyP[ct][0] = pack8(t, 0); dyP[ct][1] = pack8(t, 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks < nks_o) { const half8 w = ldg_block(ft, oTo + ks, lane); aF[ct] = mfma(w, dyF[ct][ks], aF[ct]); aS[ct] = mfma(dyF[ct][ks], w, aS[ct]); }
                }
            }
            wgrad_acc(a_ho, dyP, h2S);
            half8 dzP[CT][2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    s_e[LB][(head * 8 + (2 * e + s) * 2 + ct) * 64 + lane] = mask_by_bits(pack8(aF[ct], s), mH2 >> (8 * (ct * 2 + s)));
                    dzP[ct][s] = mask_by_nonzero(pack8(aS[ct], s), h2S[ct][s]);
                    s_sd[((head * 2 + e) * 4 + ct * 2 + s) * 64 + lane] = dzP[ct][s];
                }
            MNF_SYNC(4);
            // head hidden layer: weight gradient tiles (nt, e), nt = 0, 1
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                half8 a[CT][2];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int s = 0; s < 2; ++s) a[ct][s] = s_sd[((head * 2 + nt) * 4 + ct * 2 + s) * 64 + lane];
                wgrad_acc(a_hh[nt], a, h1S);
            }
        }
        {   // dZ1 of the head in both orientations; F published for the geo gradient, S stays here (row owner of the head's input matrix)
            half8 tw[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) tw[ks] = ldg_block(ft, oTh + ks, lane);
            half8 dzP[CT][2], inP[CT][2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x16 aF, aS;
#pragma unroll
                for (int i = 0; i < 16; ++i) { aF[i] = 0.0f; aS[i] = 0.0f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const half8 b = s_e[LB][(head * 8 + ks * 2 + ct) * 64 + lane];
                    aF = mfma(tw[ks], b, aF); aS = mfma(b, tw[ks], aS);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    s_e[LB ^ 1][(head * 8 + (2 * e + s) * 2 + ct) * 64 + lane] = mask_by_bits(pack8(aF, s), mH1 >> (8 * (ct * 2 + s)));
                    dzP[ct][s] = mask_by_nonzero(pack8(aS, s), h1S[ct][s]);
                }
                // the head's input as an S tile: rgb columns 0..15 = SH, 16..31 = geo fragment rows; semantic columns 0..15 = geo fragment rows
                f32x16 t;
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = 0.0f;
                if (head == 0) {
                    t = mfma(ldg_block(enct, 8 + ct, lane), ident_nat(rl, hl, 0), t);
                    t = mfma(geoF[ct], ident_acc(rl, hl, 16), t);
                } else {
                    t = mfma(geoF[ct], ident_acc(rl, hl, 0), t);
                }
                inP[ct][0] = pack8(t, 0); inP[ct][1] = pack8(t, 1);
            }
            wgrad_acc(a_hi, dzP, inP);
        }
        MNF_SYNC(5);
        {   // geo-feature gradient of this head: wave e takes column tile ct = e; exchanged in fp32
            f32x16 g;
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) g = mfma(s_t[(16 + head * 4 + ks) * 64 + lane], s_e[LB ^ 1][(head * 8 + ks * 2 + e) * 64 + lane], g);
#pragma unroll
            for (int i = 0; i < 8; ++i) s_g[((head * 2 + e) * 8 + i) * 64 + lane] = g[i];
        }
        MNF_SYNC(6);
        half8 dboF[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dboF[ct][j] = (half_t)(s_g[((0 * 2 + ct) * 8 + j) * 64 + lane] + s_g[((1 * 2 + ct) * 8 + j) * 64 + lane]);
            if (hl == 0) dboF[ct][0] = sat_half(s_dl[rl + 32 * ct]);
        }
        half8 dzS[CT][2];                       // packed S tile (own rows q) of the current base pre-activation gradient
        {   // base output layer: weight gradient (0, q), then dZ(NH-1) in both orientations
            half8 dboP[CT][2];
            f32x16 aF[CT], aS[CT];
            zero2(aF); zero2(aS);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x16 t;
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = 0.0f;
                t = mfma(dboF[ct], ident_acc(rl, hl, 0), t);
                dboP[ct][0] = pack8(t, 0); dboP[ct][1] = pack8(t, 1);
                const half8 w = ldg_block(ft, LT::o_bo + q, lane);
                aF[ct] = mfma(w, dboF[ct], aF[ct]); aS[ct] = mfma(dboF[ct], w, aS[ct]);
            }
            wgrad_acc(a_bo, dboP, hS[NH - 1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    s_e[LB][((2 * q + s) * 2 + ct) * 64 + lane] = mask_by_bits(pack8(aF[ct], s), mF[NH - 1] >> (8 * (ct * 2 + s)));
                    dzS[ct][s] = mask_by_nonzero(pack8(aS[ct], s), hS[NH - 1][ct][s]);
                    if (NH > 1) s_sd[(q * 4 + ct * 2 + s) * 64 + lane] = dzS[ct][s];
                }
        }
        MNF_SYNC(7);
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {     // hidden matrix l - 1 -> l: weight gradient tiles (nt, q), then dZ(l - 1); F buffers alternate from LB
            const int cur = (LB + (NH - 1 - l)) & 1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                half8 a[CT][2];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int s = 0; s < 2; ++s) a[ct][s] = s_sd[(nt * 4 + ct * 2 + s) * 64 + lane];
                wgrad_acc(a_hid[l - 1][nt], a, hS[l - 1]);
            }
            half8 tw[8];                          // this wave's transposed fragments of the matrix (one fetch, both column tiles)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) tw[ks] = ldg_block(ft, LT::o_bh + (l - 1) * 32 + q * 8 + ks, lane);
            static_assert(NH <= 2, "more hidden layers: s_sd is rewritten below while other waves may still read it (add a barrier)");
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {       // one 32-column tile at a time (register pressure)
                f32x16 aF, aS;
#pragma unroll
                for (int i = 0; i < 16; ++i) { aF[i] = 0.0f; aS[i] = 0.0f; }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const half8 b = s_e[cur][(ks * 2 + ct) * 64 + lane];
                    aF = mfma(tw[ks], b, aF); aS = mfma(b, tw[ks], aS);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    s_e[cur ^ 1][((2 * q + s) * 2 + ct) * 64 + lane] = mask_by_bits(pack8(aF, s), mF[l - 1] >> (8 * (ct * 2 + s)));
                    dzS[ct][s] = mask_by_nonzero(pack8(aS, s), hS[l - 1][ct][s]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MNF_SYNC(8);
        }
        constexpr int DB = (LB + (NH - 1)) & 1;   // buffer that holds dZ(0)
        {   // gradient of the 64 hash features: wave q takes row tile q >> 1, column tile q & 1; rows 8g + 4h + i == level 8rt + 2g + h, feature i
            const int rt = q >> 1, ct = q & 1;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma(s_t[(rt * 8 + ks) * 64 + lane], s_e[DB][(ks * 2 + ct) * 64 + lane], acc);
            const float inv = 1.0f / ls;
            float4 *dst = reinterpret_cast<float4 *>(args.dX) + fcol0 + 32 * ct;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = {acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv};
                dst[(int64_t)(8 * rt + 2 * g + hl) * args.Np] = v;
            }
        }
        {   // base input layer: weight gradient tiles (q, kt): the hash features as S tiles (identity products of the fragments re-read from L2)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                half8 xP[CT][2];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x16 t;
#pragma unroll
                    for (int i = 0; i < 16; ++i) t[i] = 0.0f;
                    t = mfma(ldg_block(enct, (2 * kt) * 2 + ct, lane), ident_nat(rl, hl, 0), t);
                    t = mfma(ldg_block(enct, (2 * kt + 1) * 2 + ct, lane), ident_nat(rl, hl, 16), t);
                    xP[ct][0] = pack8(t, 0); xP[ct][1] = pack8(t, 1);
                }
                wgrad_acc(a_in[kt], dzS, xP);
            }
        }
        MNF_SYNC(9);        // the next tile's first layer rewrites s_e[0], s_sd and s_g
    }

    // =============================================================== weight gradients out: once per workgroup
    const float inv_scale = 1.0f / ls;
    // job numbering of build_tables: base-in 4 x 2, hidden 4 x 4 each, base-out 1 x 4, rgb in 2 x 1, rgb hidden 2 x 2, rgb out 1 x 2, semantic likewise
    constexpr int j_hid = 8, j_bo = j_hid + 16 * NHH, j_ri = j_bo + 4, j_rh = j_ri + 2, j_ro = j_rh + 4, j_si = j_ro + 2, j_sh = j_si + 2, j_so = j_sh + 4;
    flush_tile(args, a_in[0], 2 * q, r, h, inv_scale);
    flush_tile(args, a_in[1], 2 * q + 1, r, h, inv_scale);
#pragma unroll
    for (int l = 0; l < NHH; ++l)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) flush_tile(args, a_hid[l][nt], j_hid + 16 * l + nt * 4 + q, r, h, inv_scale);
    flush_tile(args, a_bo, j_bo + q, r, h, inv_scale);
    flush_tile(args, a_hi, (head == 0 ? j_ri : j_si) + e, r, h, inv_scale);
    flush_tile(args, a_hh[0], (head == 0 ? j_rh : j_sh) + e, r, h, inv_scale);
    flush_tile(args, a_hh[1], (head == 0 ? j_rh : j_sh) + 2 + e, r, h, inv_scale);
    flush_tile(args, a_ho, (head == 0 ? j_ro : j_so) + e, r, h, inv_scale);
}

