// Device-side building blocks of the fused field kernels (inference: field.hip, training: train.hip):
// fp16 vector types, fragment-block bookkeeping, MFMA layer helpers, the lane<->lane+32 half exchange,
// hash-level index math and degree-4 spherical harmonics.  See the header of field.hip for the data flow.
#pragma once
#include "field.h"

MNF_DT_BEGIN

// `half_t` = the 16-bit type of the matrix-core operands (weights, hash features at the MLP input, activations and their
// gradients): fp16 as tiny-cuda-nn, or bf16 (-DMNF_BF16).  The hash TABLE is fp16 in both builds (`tab_t`).
typedef _Float16 tab_t;
typedef _Float16 tab4 __attribute__((ext_vector_type(4)));
#ifdef MNF_BF16
typedef __bf16 half_t;
typedef __bf16 half2 __attribute__((ext_vector_type(2)));
typedef __bf16 half8 __attribute__((ext_vector_type(8)));
#else
typedef _Float16 half_t;
typedef _Float16 half2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CT = 2;                            // 32-sample column tiles per wave (lane = sample, 64 samples per wave)
constexpr int kWavesPerBlock = 8;                // 512 samples per workgroup pass, two waves per SIMD
constexpr int kThreads = kWavesPerBlock * 64;
constexpr int kWaveSamples = 32 * CT;

// ------------------------------------------------------------------ fragment block bookkeeping
template <int W, int NH>
struct Layout {
    static constexpr int Wh = W / 2;
    static constexpr int RT = W / 32;     // row tiles of a base hidden layer
    static constexpr int RTh = Wh / 32;   // row tiles of a head hidden layer
    static constexpr int KSW = W / 16;    // k-steps over a W-wide activation
    static constexpr int KSh = Wh / 16;
    static constexpr int o_b_in = 0;
    static constexpr int o_b_hid = o_b_in + RT * 4;
    static constexpr int o_b_out = o_b_hid + (NH - 1) * RT * KSW;
    static constexpr int o_h_in = o_b_out + KSW;
    static constexpr int o_h_hid = o_h_in + RTh * 2;
    static constexpr int o_h_out = o_h_hid + RTh * KSh;
    static constexpr int o_s_in = o_h_out + KSh;
    static constexpr int o_s_hid = o_s_in + RTh * 1;
    static constexpr int o_s_out = o_s_hid + RTh * KSh;
    static constexpr int blocks = o_s_out + KSh;
};

// Training-time activation storage.  Two views of the same values:
//  * `act`: feature-major fp16 tiles ACT[tile][row][64 samples] (Np = samples rounded up to 64) — every row is one
//    feature over the 64 samples of a tile, so the weight-gradient GEMM reads its MFMA operands (8 consecutive
//    samples of one feature) as plain 16-byte loads, and the 32 rows of an operand tile are one contiguous 4 KB block;
//  * `masks`: one byte per post-ReLU hidden B fragment and lane, one RECORD per lane ([tile][lane][block * CT + ct], padded to mask_bytes): bit
//    frag_mask_bit(j) says whether element j is non-zero; fragments line up register-for-register with the accumulators of the backward chain.
//    (Until round 4 the layout was [tile][block][ct][lane]: a byte store / load per fragment, 64 bytes per wave instruction — the backward-data
//    kernel ran 34 % faster with its ~60 mask loads per tile knocked out.  A record is written in 16-byte pieces as its layers finish and read with
//    mask_bytes / 16 loads at the head of the tile.)
template <int W, int NH>
struct TrainLayout {
    static constexpr int Wh = W / 2;
    static constexpr int KSW = W / 16, KSh = Wh / 16;
    // forward rows
    static constexpr int rX = 0;
    static constexpr int rH0 = 64;                    // H(l) = rH0 + l*W, l = 0..NH-1
    static constexpr int rS = rH0 + NH * W;           // SH (16) then geo fragment (16): contiguous = head input
    static constexpr int rG = rS + 16;
    static constexpr int rHH1 = rG + 16, rHH2 = rHH1 + Wh, rHS1 = rHH2 + Wh, rHS2 = rHS1 + Wh;
    static constexpr int fwd_rows = rHS2 + Wh;
    // backward rows (written by the dgrad kernel)
    static constexpr int rdYr = fwd_rows;             // 16
    static constexpr int rdZr2 = rdYr + 16, rdZr1 = rdZr2 + Wh;
    static constexpr int rdYs = rdZr1 + Wh;           // 32
    static constexpr int rdZs2 = rdYs + 32, rdZs1 = rdZs2 + Wh;
    static constexpr int rdBO = rdZs1 + Wh;           // 16
    static constexpr int rdZ0 = rdBO + 16;            // dZ(l) = rdZ0 + l*W, l = 0..NH-1
    static constexpr int rows = rdZ0 + NH * W;
    // mask blocks per tile
    static constexpr int mH0 = 0;                     // H(l): mH0 + l*KSW
    static constexpr int mHH1 = NH * KSW, mHH2 = mHH1 + KSh, mHS1 = mHH2 + KSh, mHS2 = mHS1 + KSh;
    static constexpr int mask_blocks = mHS2 + KSh;
    static constexpr int mask_bytes = (mask_blocks * CT + 15) / 16 * 16;     // a lane's record
};

// byte `b` of a lane's mask record under construction: `piece` holds the 16-byte piece b / 16; a finished piece (or the record's last byte) is stored
template <int MASK_BYTES_USED>
__device__ __forceinline__ void mask_put(u32x4 &piece, uint8_t *record, int b, uint32_t m) {
    const int d = (b >> 2) & 3, sh = 8 * (b & 3);
    piece[d] = ((b & 3) == 0 ? 0u : piece[d]) | (m << sh);
    if ((b & 15) == 15 || b == MASK_BYTES_USED - 1) *reinterpret_cast<u32x4 *>(record + (b & ~15)) = piece;
}

// ReLU mask of a post-ReLU fp16 fragment as one byte: element j -> bit frag_mask_bit(j).  For non-negative halves
// "non-zero" is "integer value >= 1": adding 0x7FFF carries into bit 15 of each half without crossing into the other.
__device__ __forceinline__ constexpr int frag_mask_bit(int j) { return (j & 1) * 4 + (j >> 1); }
__device__ __forceinline__ uint8_t frag_mask(const half8 &f) {
    const u32x4 d = __builtin_bit_cast(u32x4, f);
    uint32_t u = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = ((d[i] & 0x7FFF7FFFu) + 0x7FFF7FFFu) & 0x80008000u;   // bit 15: element 2i, bit 31: element 2i+1
        u |= t >> (15 - i);
    }
    return (uint8_t)((u & 0xFu) | ((u >> 12) & 0xF0u));
}

struct KernelArgs {
    const tab4 *table;
    const half8 *frags;
    float aabb[6];
    int C;
    int out_fp16;           // mnf_field_config.output_fp16: round every network output to fp16 (tcnn's hand-over precision)
    int active_waves;       // experiment knob (MNF_FIELD_ACTIVE_WAVES): waves per workgroup that take tiles, default 8
    int blend16;            // mnf_field_config.blend_fp16: the hash levels' 8-corner blend as fp16 fused multiply-adds (tcnn's T = __half arithmetic)
    const LevelMeta *levels;   // [16] in device memory, wave-uniform: read with scalar loads where a batch needs them
                               // (as a by-value kernarg array the 112 dwords were all loaded up front and lived in
                               // spilled SGPRs: ~1000 v_readlane per tile)
    FieldIO io;
    TrainBuf train;
};

// feature index of element j of lane half h inside a 16-wide k-step, by fragment kind
__device__ __forceinline__ int feat_natural(int h, int j) { return 8 * h + j; }
__device__ __forceinline__ int feat_acc(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

// Store the two column-tile fragments of one 16-row group (16 features x 64 samples) into the activation matrix.
// Layout (tile-major): ACT[tile][row][64 samples] fp16, i.e. for one 64-sample tile all `rows` feature rows are contiguous
// 128-byte lines — the weight-gradient GEMM then reads the 32 rows of an operand tile as one contiguous 4 KB block.
// A lane holds 8 FEATURES of one sample, the matrix wants 8 SAMPLES of one feature next to each other, so the group is
// transposed on the way out: neighbouring lanes (samples c, c+1) first trade one half of every register (one DPP move and
// one byte permute), so that each lane owns a (row, 2 adjacent samples) dword; those go into this wave's 2 KB LDS tile
// [16 rows][64 samples], which is read back as 16-byte pieces and leaves as two `global_store_dwordx4` per lane (2 KB
// contiguous per group).  Before: 16 `global_store_short` per group — the stores, not the bytes, cost 0.6 ms of the 1.3 ms
// training forward and 0.5 ms of the backward-data kernel (round-2 experiments; tools/experiments/field_variants.patch).
constexpr int kStageHalves = 8 * 64;     // per-wave staging tile: 8 rows x 64 samples (the 16-row group goes out in two passes)

template <bool ACC_ORDER>
__device__ __forceinline__ void save_pair(const TrainBuf &tb, int64_t tile, int row0, int lane, half_t *stage, const half8 &f0, const half8 &f1) {
    const int c = lane & 31, h = lane >> 5, odd = c & 1;
    uint32_t *st32 = reinterpret_cast<uint32_t *>(stage);
    const uint32_t sel = odd ? 0x03020706u : 0x05040100u;   // even lane: {own.lo, partner.lo}; odd lane: {partner.hi, own.hi}
    const u32x4 *src = reinterpret_cast<const u32x4 *>(stage);
    char *gbase = reinterpret_cast<char *>(tb.act) + ((tile * tb.rows + row0) * 64) * 2;     // uniform: scalar base + lane offset
#pragma unroll
    for (int q = 0; q < 2; ++q) {                           // rows 8q .. 8q+7 of the group
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const u32x4 d = __builtin_bit_cast(u32x4, ct ? f1 : f0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // accumulator order: register i holds rows 8 (i >> 1) + 4h + {2 (i & 1), 2 (i & 1) + 1}; natural order: rows 8h + {2i, 2i + 1}
                if (ACC_ORDER && (i >> 1) != q) continue;
                const uint32_t own = d[i];
                const uint32_t partner = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                const uint32_t packed = __builtin_amdgcn_perm(partner, own, sel);
                const int j = 2 * i + odd;                                // the element whose row this lane now owns
                const int rowl = ACC_ORDER ? 4 * h + (j & 3) : j;         // row inside this pass
                if (ACC_ORDER || h == q) st32[(rowl * 64 + 32 * ct + (c & ~1)) >> 1] = packed;
            }
        }
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4 *>(gbase + (q * 64 + lane) * 16) = src[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) {
#ifdef MNF_BF16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// relu(round_to_fp16(x)) == round_to_fp16(relu(x)) (rounding is monotonic and sign-preserving), so pack first
// (v_cvt_pk_f16_f32, two values per instruction) and clamp in packed fp16 (v_pk_max_f16).
__device__ __forceinline__ half8 relu_pack8(const f32x16 &acc, int s) {
    half8 r;
#ifdef MNF_BF16
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (half_t)fmaxf(acc[8 * s + j], 0.0f);     // no packed bf16 max: clamp in fp32, then v_cvt_pk_bf16_f32
#else
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        half2 p = {(half_t)acc[8 * s + j], (half_t)acc[8 * s + j + 1]};
        const half2 z = {(half_t)0.0f, (half_t)0.0f};
        p = __builtin_elementwise_max(p, z);
        r[j] = p[0]; r[j + 1] = p[1];
    }
#endif
    return r;
}

// One hidden layer, fused with ReLU + fp16 packing, one 32-row output tile at a time so that only
// CT accumulator tiles are live: o[ct][2*rt + s] <- relu(W(rt,:) * b[ct])
template <int RT_OUT, int KS>
__device__ __forceinline__ void dense_relu(const half8 *__restrict__ w_lds, int lane, const half8 (&b)[CT][KS],
                                           half8 (&o)[CT][RT_OUT * 2]) {
#pragma unroll
    for (int rt = 0; rt < RT_OUT; ++rt) {
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 a = w_lds[(rt * KS + ks) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma(a, b[ct][ks], acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) o[ct][rt * 2 + s] = relu_pack8(acc[ct], s);
    }
}

// Output layer (one 32-row tile, no activation)
template <int KS>
__device__ __forceinline__ void dense_out(const half8 *__restrict__ w_lds, int lane, const half8 (&b)[CT][KS], f32x16 (&o)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[ct][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const half8 a = w_lds[ks * 64 + lane];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) o[ct] = mfma(a, b[ct][ks], o[ct]);
    }
}

// tcnn-faithful hand-over: the three networks' outputs rounded to fp16 and widened again (`.to(x)` in ngp.py:181-220)
__device__ __forceinline__ void round_outputs_fp16(f32x16 (&o)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[ct][i] = (float)(half_t)o[ct][i];
}

// Lane l (sample A = column l of tile 0) and lane l+32 (sample B = column l of tile 1) each hold all 16
// features of a k-step of THEIR sample as lo = features 0..7, hi = features 8..15.  The MFMA B operand wants,
// for tile t, lane half h to hold features 8h..8h+7 of the tile-t sample.  One v_permlane32_swap per dword
// (lo's upper 32 lanes <-> hi's lower 32 lanes) produces exactly that: lo -> tile-0 fragment, hi -> tile-1 fragment.
__device__ __forceinline__ void exchange_halves(half8 &lo, half8 &hi) {
    u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
        a[i] = r[0]; b[i] = r[1];
    }
    lo = __builtin_bit_cast(half8, a); hi = __builtin_bit_cast(half8, b);
}

// The level table pointer made opaque once per tile, so that the scalar loads of a level's metadata stay next to their
// use instead of being hoisted out of the tile loop (where they would occupy, and spill, a hundred SGPRs).  The pointer is
// typed CONSTANT address space: only then does the compiler read the metadata with s_load into SGPRs.  Through a plain
// global pointer (round 1 and the first half of round 2) it issued vector loads with a uniform address, whose
// `s_waitcnt vmcnt(0)` also waited for every hash gather still in flight, and it treated `hashed` as a per-lane value.
typedef const LevelMeta __attribute__((address_space(4))) *LevelsPtr;
__device__ __forceinline__ LevelsPtr levels_here(const LevelMeta *p) {
    LevelsPtr q = (LevelsPtr)(uintptr_t)p;
    asm volatile("" : "+s"(q));
    return q;
}
__device__ __forceinline__ LevelMeta level_meta(LevelsPtr lv, int l) {
    LevelMeta m;
    m.scale = lv[l].scale; m.res = lv[l].res; m.size = lv[l].size; m.offset = lv[l].offset; m.hashed = lv[l].hashed;
    m.div_magic = lv[l].div_magic; m.div_shift = lv[l].div_shift;
    return m;
}

// One hash level (wave-uniform metadata) for the lane's sample, split in two so that the gathers of several
// levels can be in flight together: hash_prep computes the 8 byte offsets and the separable trilinear weights,
// hash_blend consumes the 8 loaded entries.  The blend weight of corner (bx,by,bz) is ((wx*wy)*wz), the same
// association as the oracle's running product.
typedef float f32x2 __attribute__((ext_vector_type(2)));   // arithmetic on it is v_pk_{mul,add,fma}_f32: two fp32 results per issue slot

typedef _Float16 tab8 __attribute__((ext_vector_type(8), aligned(8)));      // two adjacent entries: a 16-byte load that is only 8-byte aligned

struct LevelPrep {
    uint32_t paired; // wave-uniform: this level's corners are fetched as four x-neighbour pairs (dense level, every pair contiguous)
    uint32_t base;   // first entry of the level (wave-uniform): folded into the scalar base address of the gathers, not into every offset
    uint32_t off[8]; // byte offset of each corner's entry inside the level
    f32x2 wxy[2];    // {wx0*wy0, wx1*wy0}, {wx0*wy1, wx1*wy1}
    float wz[2];
};

__device__ __forceinline__ void hash_prep(const LevelMeta m, const float xn[3], LevelPrep &o, bool all_in_box = false) {
    // x and y as one packed pair (same operations, same rounding as the scalar form), z alone
    const float px = __builtin_fmaf(m.scale, xn[0], 0.5f), py = __builtin_fmaf(m.scale, xn[1], 0.5f), pz = __builtin_fmaf(m.scale, xn[2], 0.5f);
    const f32x2 fxy = {floorf(px), floorf(py)};
    const float fz = floorf(pz);
    const f32x2 frxy = {px - fxy.x, py - fxy.y};
    const float frz = pz - fz;
    const uint32_t cell[3] = {(uint32_t)(int32_t)fxy.x, (uint32_t)(int32_t)fxy.y, (uint32_t)(int32_t)fz};
    const f32x2 one_m = {1.0f - frxy.x, 1.0f - frxy.y};
    o.wz[0] = 1.0f - frz; o.wz[1] = frz;
    const f32x2 wx = {one_m.x, frxy.x};                    // {wx0, wx1}
    o.wxy[0] = wx * f32x2{one_m.y, one_m.y};
    o.wxy[1] = wx * f32x2{frxy.y, frxy.y};
    // ONE wave-uniform branch per level (hashed levels: size is 2^k; dense levels wrap the index as tcnn does).  Written as
    // two corner loops: with the test inside a single loop the compiler kept a scalar branch per corner.
    o.base = m.offset;
    o.paired = 0;
    if (m.hashed) {
        const uint32_t ty0 = cell[1] * 2654435761u, ty1 = ty0 + 2654435761u;   // per-axis terms, shared by the four corners that use them
        const uint32_t tz0 = cell[2] * 805459861u, tz1 = tz0 + 805459861u;
        const uint32_t mask = m.size - 1u;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const uint32_t px = cell[0] + (uint32_t)(corner & 1);
            const uint32_t idx = (px ^ ((corner >> 1) & 1 ? ty1 : ty0) ^ ((corner >> 2) ? tz1 : tz0)) & mask;
            o.off[corner] = idx * 8u;   // 32-bit byte offset from the level's first entry (SGPR base + VGPR offset)
        }
    } else {
        const uint32_t r2 = m.res * m.res;
        const uint32_t ty0 = cell[1] * m.res, ty1 = ty0 + m.res;
        const uint32_t tz0 = cell[2] * r2, tz1 = tz0 + r2;
        // idx %= m.size (only out-of-box positions / the far corner actually wrap).  `all_in_box` is wave-uniform: with every
        // position inside the unit box the corner coordinates are <= res, so idx < 2*size and one conditional subtraction is
        // the modulo (unsigned min of idx and idx - size).  Otherwise: exact for every uint32 and branch-free (multiply-high by
        // a precomputed reciprocal) -- the compiler's generic modulo put a rarely taken division loop behind every corner.
        if (all_in_box) {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const uint32_t idx = cell[0] + (uint32_t)(corner & 1) + ((corner >> 1) & 1 ? ty1 : ty0) + ((corner >> 2) ? tz1 : tz0);
                o.off[corner] = min(idx, idx - m.size) * 8u;
            }
        } else {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                uint32_t idx = cell[0] + (uint32_t)(corner & 1) + ((corner >> 1) & 1 ? ty1 : ty0) + ((corner >> 2) ? tz1 : tz0);
                uint32_t q = __umulhi(m.div_magic, idx);
                q = (((idx - q) >> 1) + q) >> m.div_shift;
                idx -= q * m.size;
                o.off[corner] = idx * 8u;
            }
        }
    }
}

__device__ __forceinline__ void hash_load(const tab4 *__restrict__ table, const LevelPrep &p, tab4 (&v)[8]) {
    const char *level = reinterpret_cast<const char *>(table + p.base);      // uniform: scalar address arithmetic
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
        v[corner] = *reinterpret_cast<const tab4 *>(level + p.off[corner]);
}

// acc = fma((float)half, w, acc) in ONE instruction: v_fma_mix_f32 reads the fp16 operand straight from one half of a
// 32-bit register (exact widening, single rounding: the same value as cvt + fma).  The compiler's own choice was two
// v_cvt_f32_f16 per dword plus a packed fp32 fma: 6 instructions per corner instead of 4.
template <int HI>
__device__ __forceinline__ float fma_mix_half(uint32_t packed, float w, float acc) {
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}

__device__ __forceinline__ void hash_blend(const LevelPrep &p, const tab4 (&v)[8], float *f) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    // corner (bx,by,bz) = bit 0, 1, 2 of its index; weight ((wx*wy)*wz); two corners' weights per packed multiply
    const f32x2 w4[4] = {p.wxy[0] * f32x2{p.wz[0], p.wz[0]}, p.wxy[1] * f32x2{p.wz[0], p.wz[0]},
                         p.wxy[0] * f32x2{p.wz[1], p.wz[1]}, p.wxy[1] * f32x2{p.wz[1], p.wz[1]}};
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float w = (corner & 1) ? w4[corner >> 1].y : w4[corner >> 1].x;
        // fma(fpext(half), w, acc): selected as v_fma_mix_f32 (one instruction per feature; no SLP pairing in this file).
        // Left to the compiler rather than inline asm so that its hazard recogniser sees producer and consumer: the asm form
        // right behind a v_pk_mul_f32 gave wrong values in lanes 48..63 of occasional tiles.
        a0 = __builtin_fmaf((float)v[corner][0], w, a0); a1 = __builtin_fmaf((float)v[corner][1], w, a1);
        a2 = __builtin_fmaf((float)v[corner][2], w, a2); a3 = __builtin_fmaf((float)v[corner][3], w, a3);
    }
    f[0] = a0; f[1] = a1; f[2] = a2; f[3] = a3;
}

// The same blend in tiny-cuda-nn's half-precision arithmetic (mnf_field_config.blend_fp16): `result = fma((T)weight, entry, result)` with T = __half,
// corners in index order (x fastest) — the weight rounded to fp16, the running sum an fp16 fused multiply-add per corner.  The two feature pairs of an
// entry are the two dwords it was loaded as, so a corner is one conversion and two v_pk_fma_f16; the result is already the 16-bit MLP input.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 hash_blend16(const LevelPrep &p, const tab4 (&v)[8]) {
    h16x2 a01 = {(_Float16)0.0f, (_Float16)0.0f}, a23 = a01;
    const f32x2 w4[4] = {p.wxy[0] * f32x2{p.wz[0], p.wz[0]}, p.wxy[1] * f32x2{p.wz[0], p.wz[0]},
                         p.wxy[0] * f32x2{p.wz[1], p.wz[1]}, p.wxy[1] * f32x2{p.wz[1], p.wz[1]}};
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const _Float16 wh = (_Float16)((corner & 1) ? w4[corner >> 1].y : w4[corner >> 1].x);
        const h16x2 w2 = {wh, wh};
        const u32x2 d = __builtin_bit_cast(u32x2, v[corner]);
        const uint32_t d0 = d.x, d1 = d.y;
        a01 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h16x2, d0), a01);
        a23 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h16x2, d1), a23);
    }
    return u32x2{__builtin_bit_cast(uint32_t, a01), __builtin_bit_cast(uint32_t, a23)};
}

// tcnn SphericalHarmonics degree 4 on 2u-1, u = (d+1)/2 (ngp.py:205): all 16 values of the lane's sample
__device__ __forceinline__ void sh4(const float d[3], half8 &lo, half8 &hi) {
    const float x = ((d[0] + 1.0f) / 2.0f) * 2.0f - 1.0f;
    const float y = ((d[1] + 1.0f) / 2.0f) * 2.0f - 1.0f;
    const float z = ((d[2] + 1.0f) / 2.0f) * 2.0f - 1.0f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    lo[0] = (half_t)0.28209479177387814f;
    lo[1] = (half_t)(-0.48860251190291987f * y);
    lo[2] = (half_t)(0.48860251190291987f * z);
    lo[3] = (half_t)(-0.48860251190291987f * x);
    lo[4] = (half_t)(1.0925484305920792f * xy);
    lo[5] = (half_t)(-1.0925484305920792f * yz);
    lo[6] = (half_t)(0.94617469575755997f * z2 - 0.31539156525251999f);
    lo[7] = (half_t)(-1.0925484305920792f * xz);
    hi[0] = (half_t)(0.54627421529603959f * x2 - 0.54627421529603959f * y2);
    hi[1] = (half_t)(0.59004358992664352f * y * (-3.0f * x2 + y2));
    hi[2] = (half_t)(2.8906114426405538f * xy * z);
    hi[3] = (half_t)(0.45704579946446572f * y * (1.0f - 5.0f * z2));
    hi[4] = (half_t)(0.3731763325901154f * z * (5.0f * z2 - 3.0f));
    hi[5] = (half_t)(0.45704579946446572f * x * (1.0f - 5.0f * z2));
    hi[6] = (half_t)(1.4453057213202769f * z * (x2 - y2));
    hi[7] = (half_t)(0.59004358992664352f * x * (-x2 + 3.0f * y2));
}

MNF_DT_END
