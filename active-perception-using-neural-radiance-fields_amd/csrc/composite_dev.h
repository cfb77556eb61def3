// Volumetric compositing of one render round, fused into the field kernel's epilogue (mode 2).
//
// Reference semantics: perception/models/utils.py:704-757 (+ :984-999 for the probabilistic variant) on top of
// perception/nerfacc/nerfacc/volrend.py:258-267, :361-365 — per ray: transmittance from the exclusive sum of
// sigma*dt carried by `1 - opacity` of the previous rounds, alpha threshold, weighted sums of rgb / depth / the 29
// raw semantic logits, variance terms against the post-round running means, ray retirement.
//
// The tile layout guarantees that a ray's samples are consecutive lanes of ONE wave (lane = sample), so every
// per-ray sum is a segmented scan over lanes (the reference's packed exclusive_sum / index_add_ pair), and the
// per-sample rgb / sigma / logits never leave registers.
#pragma once
#include "field_dev.h"

MNF_DT_BEGIN

// inclusive segmented scan over the 64 lanes of a wave; `head` marks the first lane of every segment
// (`maxlen` = wave-uniform upper bound of the segment length: only ceil(log2(maxlen)) steps are needed)
template <int N>
__device__ __forceinline__ void seg_scan64(float (&v)[N], bool head, int lane, int maxlen) {
    bool f = head;
#pragma unroll 1
    for (int d = 1; d < maxlen; d <<= 1) {
        const int fu = __shfl_up((int)f, d, 64);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float t = __shfl_up(v[k], d, 64);
            if (lane >= d && !f) v[k] += t;
        }
        if (lane >= d) f = f || (fu != 0);
    }
}

// The same scan on the VALU's data-parallel-primitive path instead of ds_bpermute (the LDS pipe is shared by the whole
// CU and 192 permutes per level made the hash-gradient scatter LDS-bound): Hillis-Steele inside each 16-lane row with
// row_shr:1/2/4/8, then the row totals are carried across the three row boundaries with row_bcast:15.  `heads` is the
// wave's ballot of segment heads (lanes outside every segment count as heads); HALVES = true scans lanes 0..31 and
// 32..63 independently (no carry across lane 32).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}

template <int N, int D>
__device__ __forceinline__ void seg_step_row(float (&v)[N], unsigned long long &window, int lane) {
    // lane i may add lane i-D iff no head in (i-D, i]; `window` holds the OR of the head mask over (i-D', i] for the
    // previous distance D' = D/2 and is widened here
    if (D > 1) window |= window << (D / 2);
    const bool add = !((window >> lane) & 1ull);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float t = dpp_f32<0x110 + D, 0xf>(v[k]);      // row_shr:D, lanes without an in-row source read 0
        v[k] += add ? t : 0.0f;
    }
}

template <int N, bool HALVES = false>
__device__ __forceinline__ void seg_scan_dpp(float (&v)[N], unsigned long long heads, int lane, int maxlen) {
    unsigned long long window = heads;
    seg_step_row<N, 1>(v, window, lane);
    if (maxlen > 2) seg_step_row<N, 2>(v, window, lane);
    if (maxlen > 4) seg_step_row<N, 4>(v, window, lane);
    if (maxlen > 8) seg_step_row<N, 8>(v, window, lane);
    // row totals across the row boundaries, for lanes whose segment began before their row (no head in [row start, lane])
    const int row = lane >> 4, in_row = lane & 15;
    const unsigned long long row_heads = (heads >> (row * 16)) & 0xFFFFull;
    const bool carries = (row_heads & ((2ull << in_row) - 1ull)) == 0ull;
    const unsigned long long crossing = ~heads & (HALVES ? 0x0001000000010000ull : 0x0001000100010000ull);
    if (crossing == 0ull) return;                          // wave-uniform: no segment spans a row boundary
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float t = dpp_f32<0x142, 0x2>(v[k]);          // row_bcast:15 into row 1
        v[k] += (carries && row == 1) ? t : 0.0f;
    }
    if (!HALVES) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float t = dpp_f32<0x142, 0x4>(v[k]);      // row 1's (updated) total into row 2
            v[k] += (carries && row == 2) ? t : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float t = dpp_f32<0x142, 0x8>(v[k]);          // row 2's total into row 3
        v[k] += (carries && row == 3) ? t : 0.0f;
    }
}

// the same over each 32-lane half independently (MFMA layout: column c = lane & 31)
__device__ __forceinline__ void seg_scan32(f32x16 &v, bool head, int c, int maxlen) {
    bool f = head;
    const int lim = maxlen < 32 ? maxlen : 32;
#pragma unroll 1
    for (int d = 1; d < lim; d <<= 1) {
        const int fu = __shfl_up((int)f, d, 32);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float t = __shfl_up(v[k], d, 32);
            if (c >= d && !f) v[k] += t;
        }
        if (c >= d) f = f || (fu != 0);
    }
}

// ---- aligned power-of-two runs (the renderer's tiles at budget 4, 8 or 16: 90 % of all columns) ----
// Every slot of `stride` columns starts at a multiple of `stride`, so a per-ray sum is a butterfly over the slot's lanes: one
// DPP add per step and value, no segment flags.  quad_perm [1,0,3,2] / [2,3,0,1] inside a quad, then row_half_mirror (8 lanes)
// and row_mirror (16 lanes): after each step every lane of the slot holds the same partial sum.
template <int N>
__device__ __forceinline__ void slot_totals(float (&v)[N], int stride) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_f32<0xB1, 0xf>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_f32<0x4E, 0xf>(v[k]);
    if (stride >= 8) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += dpp_f32<0x141, 0xf>(v[k]);
    }
    if (stride >= 16) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += dpp_f32<0x140, 0xf>(v[k]);
    }
}

// Four values per lane summed over the four lanes of a quad, lane j of the quad keeping the total of value j (a
// reduce-scatter: 9 instructions instead of 8 for four all-lane totals, and each lane then owns ONE value to write).
__device__ __forceinline__ float quad_reduce_scatter(float x0, float x1, float x2, float x3, bool odd, bool upper) {
    const float keep01 = odd ? x1 : x0, send01 = odd ? x0 : x1;
    const float keep23 = odd ? x3 : x2, send23 = odd ? x2 : x3;
    const float a = keep01 + dpp_f32<0xB1, 0xf>(send01);     // lanes {j, j^1}: value (j & 1)
    const float b = keep23 + dpp_f32<0xB1, 0xf>(send23);     //                 value 2 + (j & 1)
    const float keep = upper ? b : a, send = upper ? a : b;
    return keep + dpp_f32<0x4E, 0xf>(send);                  // all four lanes: value j
}

// Wave-uniform counters carried across the tiles a wave processes and flushed with ONE atomic each at the end
// (same-address atomics serialise at ~12 ns apiece: one per tile would cost more than the compositing itself).
struct WaveCounters {
    int view = -1, alive = 0;
    float kept = 0.f, marched = 0.f;
};

__device__ __forceinline__ void flush_alive(const FusedRender &fr, WaveCounters &wc, int lane) {
    if (lane == 0 && wc.view >= 0 && wc.alive > 0) atomicAdd(&fr.alive_count[wc.view], wc.alive);
    wc.alive = 0;
}

__device__ __forceinline__ void flush_counters(const FusedRender &fr, WaveCounters &wc, int lane) {
    flush_alive(fr, wc, lane);
    if (lane == 0) {
        if (wc.kept > 0.f) atomicAdd(fr.totals, (unsigned long long)(wc.kept + 0.5f));
        if (wc.marched > 0.f) atomicAdd(fr.totals + 1, (unsigned long long)(wc.marched + 0.5f));
    }
    wc.kept = 0.f; wc.marched = 0.f;
}

struct TileSample {     // this lane's sample (lane = sample)
    int ray;            // -1: unused column
    int stride;         // wave-uniform: columns per ray slot in this tile = the per-ray budget of the tile's view this round
    int view;           // wave-uniform: the view all rays of the tile belong to (a march workgroup never mixes views)
    bool valid;
    float ts, te;
    float opac0;        // opacity of the ray before this round (prefetched)
};

// ---- per-ray bookkeeping by the owner lane (both compositing paths) ----
__device__ __forceinline__ void finish_rays(const FusedRender &fr, int lane, const TileSample &sm, WaveCounters &wc, bool owner, int view,
                                            int budget, int cnt, const float (&c_new)[3], float d_new, float o_new,
                                            const float (&v_prev)[4], const float (&vt)[4], int tile_kept, int tile_marched) {
    bool still_alive = false;
    if (owner) {
        if (fr.probabilistic) {
            fr.rgb_var[3 * sm.ray] = v_prev[0] + vt[0]; fr.rgb_var[3 * sm.ray + 1] = v_prev[1] + vt[1];
            fr.rgb_var[3 * sm.ray + 2] = v_prev[2] + vt[2];
            fr.depth_var[sm.ray] = v_prev[3] + vt[3];
        }
        fr.rgb[3 * sm.ray] = c_new[0]; fr.rgb[3 * sm.ray + 1] = c_new[1]; fr.rgb[3 * sm.ray + 2] = c_new[2];
        fr.acc[sm.ray] = o_new; fr.depth[sm.ray] = d_new;
        still_alive = (o_new <= fr.opc_thre) && (cnt == budget);       // utils.py:751-756
        fr.alive[sm.ray] = still_alive;
    }
    // survivors per view and sample totals: accumulated per wave, flushed when the view changes / after the last tile
    // survivors of the tile's view and sample totals: accumulated per wave, flushed when the view changes / after the last tile
    if (__ballot(owner)) {
        if (wc.view != sm.view) { flush_alive(fr, wc, lane); wc.view = sm.view; }
        wc.alive += __popcll(__ballot(still_alive));
        wc.kept += (float)tile_kept; wc.marched += (float)tile_marched;
    }
}

// Compositing of a tile whose slots are aligned runs of 4, 8 or 16 columns (see slot_totals): same arithmetic per sample as
// the general path below, the per-ray sums as butterflies, and the 29 semantic sums reduce-scattered so that every lane of
// a slot's first quad adds ONE row per register group into the ray's accumulator (8 read-modify-writes per lane and tile
// instead of 32 by the slot's last lane).  Summation order differs from the general path (pairwise instead of front to back).
__device__ __forceinline__ void fused_composite_slots(const FusedRender &fr, int C, int lane, const TileSample &sm, float sigma,
                                                      const float (&rgb)[3], const f32x16 (&sem)[CT], WaveCounters &wc, bool owner,
                                                      int view, int budget, const float (&c_prev)[3], float d_prev,
                                                      const float (&v_prev)[4]) {
    const int c = lane & 31, h = lane >> 5, stride = sm.stride;
    const int j = c & 3;
    // ---- the semantic accumulators this lane will update: ray of the slot that column 32 ct + c belongs to, rows j + 4h + 8g ----
    float s_prev[CT][4];
    float *s_ptr[CT];
    bool s_store[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int slot_ray = __shfl(sm.ray, (32 * ct + c) & ~(stride - 1), 64);     // a slot's valid columns come first
        s_store[ct] = slot_ray >= 0 && (c & (stride - 1)) < 4;
        s_ptr[ct] = fr.sem + ((int64_t)(slot_ray < 0 ? 0 : slot_ray) * C + j + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) s_prev[ct][g] = (s_store[ct] && j + 4 * h + 8 * g < C) ? s_ptr[ct][8 * g] : 0.f;
    }
    // ---- weights (volrend.py:258-267, :361-365; utils.py:712-725): exclusive prefix of sigma*dt inside the slot ----
    const float sdt = sm.valid ? sigma * (sm.te - sm.ts) : 0.0f;
    const int in_slot = lane & (stride - 1);
    float incl = sdt;
    {
        float t = dpp_f32<0x111, 0xf>(incl); incl += in_slot >= 1 ? t : 0.0f;      // row_shr:1
        t = dpp_f32<0x112, 0xf>(incl); incl += in_slot >= 2 ? t : 0.0f;            // row_shr:2
        if (stride >= 8) { t = dpp_f32<0x114, 0xf>(incl); incl += in_slot >= 4 ? t : 0.0f; }
        if (stride >= 16) { t = dpp_f32<0x118, 0xf>(incl); incl += in_slot >= 8 ? t : 0.0f; }
    }
    const float excl = incl - sdt;
    const float alpha = 1.0f - expf(-sdt);
    const float opac0 = sm.opac0;
    const float w = expf(-excl) * (1.0f - opac0) * alpha;
    const bool keep = sm.valid && !(fr.alpha_thre > 0.f && !(alpha >= fr.alpha_thre));
    const float wk = keep ? w : 0.0f;
    const float tmid = sm.valid ? (sm.ts + sm.te) / 2.0f : 0.0f;   // an unused column holds whatever the workspace held before: 0 * NaN would poison the depth of its ray
    float tot[6] = {wk, wk * rgb[0], wk * rgb[1], wk * rgb[2], wk * tmid, sm.valid ? 1.0f : 0.0f};
    slot_totals<6>(tot, stride);
    const int cnt = (int)(tot[5] + 0.5f);
    const float c_new[3] = {c_prev[0] + tot[1], c_prev[1] + tot[2], c_prev[2] + tot[3]};
    const float d_new = d_prev + tot[4];
    const float o_new = opac0 + tot[0];
    float vt[4] = {0.f, 0.f, 0.f, 0.f};
    if (fr.probabilistic) {                                                     // utils.py:984-999
        const float e0 = rgb[0] - c_new[0], e1 = rgb[1] - c_new[1], e2 = rgb[2] - c_new[2], ed = tmid - d_new;
        vt[0] = wk * (e0 * e0); vt[1] = wk * (e1 * e1); vt[2] = wk * (e2 * e2); vt[3] = wk * (ed * ed);
        slot_totals<4>(vt, stride);
    }
    // ---- semantic logits (MFMA layout: lane (c, h) holds rows 8g + 4h + i of column 32 ct + c in register 4g + i) ----
    const bool odd = (j & 1) != 0, upper = (j & 2) != 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float wm = __shfl(wk, 32 * ct + c, 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float z = quad_reduce_scatter(sem[ct][4 * g] * wm, sem[ct][4 * g + 1] * wm, sem[ct][4 * g + 2] * wm, sem[ct][4 * g + 3] * wm, odd, upper);
            if (stride >= 8) z += dpp_f32<0x104, 0xf>(z);                        // row_shl:4: the slot's first quad collects the second ...
            if (stride >= 16) z += dpp_f32<0x108, 0xf>(z);                       // row_shl:8: ... and the third and fourth
            if (s_store[ct] && j + 4 * h + 8 * g < C) s_ptr[ct][8 * g] = s_prev[ct][g] + z;
        }
    }
    finish_rays(fr, lane, sm, wc, owner, view, budget, cnt, c_new, d_new, o_new, v_prev, vt, __popcll(__ballot(keep)), __popcll(__ballot(sm.valid)));
}

__device__ __forceinline__ void fused_composite(const FusedRender &fr, int C, int lane, const TileSample &sm,
                                                float sigma, const float (&rgb)[3], const f32x16 (&sem)[CT], WaveCounters &wc) {
    const int c = lane & 31, h = lane >> 5;
    // run structure from the per-column ray ids: valid columns of a ray are consecutive lanes
    const int prev_ray = __shfl_up(sm.ray, 1, 64);
    const bool head = lane == 0 || prev_ray != sm.ray || sm.ray < 0;
    const unsigned long long heads = __ballot(head);
    const bool tail = sm.ray >= 0 && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
    const bool owner = sm.ray >= 0 && head;                            // one bookkeeping lane per ray
    // lane of this run's tail: first head bit above this lane, minus one
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int tail_lane = above ? lane + (__ffsll(above) - 1) : 63;
    const int maxlen = sm.stride;
    // ---- every load of the epilogue is issued here, before any arithmetic or store: they depend only on the run
    //      structure, and issued one read-modify-write at a time they cost four to five serial memory round trips
    //      per tile at two waves per SIMD ----
    float c_prev[3] = {0.f, 0.f, 0.f}, d_prev = 0.f, v_prev[4] = {0.f, 0.f, 0.f, 0.f};
    int view = -1, budget = 0;
    if (sm.ray >= 0 && (owner || fr.probabilistic)) {
        c_prev[0] = fr.rgb[3 * sm.ray]; c_prev[1] = fr.rgb[3 * sm.ray + 1]; c_prev[2] = fr.rgb[3 * sm.ray + 2];
        d_prev = fr.depth[sm.ray];
    }
    if (owner) {
        view = sm.view; budget = sm.stride;
        if (fr.probabilistic) {
            v_prev[0] = fr.rgb_var[3 * sm.ray]; v_prev[1] = fr.rgb_var[3 * sm.ray + 1]; v_prev[2] = fr.rgb_var[3 * sm.ray + 2];
            v_prev[3] = fr.depth_var[sm.ray];
        }
    }
    const bool fast = (maxlen == 4 || maxlen == 8 || maxlen == 16) && !fr.general_only;   // wave-uniform: aligned power-of-two slots
    if (fast) {
        fused_composite_slots(fr, C, lane, sm, sigma, rgb, sem, wc, owner, view, budget, c_prev, d_prev, v_prev);
        return;
    }
    f32x16 s_prev[CT];
    int raym[CT];
    bool hm[CT], tm[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int src = 32 * ct + c;
        raym[ct] = __shfl(sm.ray, src, 64);
        hm[ct] = __shfl((int)head, src, 64) != 0;
        tm[ct] = __shfl((int)tail, src, 64) != 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
            s_prev[ct][k] = (tm[ct] && row < C) ? fr.sem[(int64_t)raym[ct] * C + row] : 0.f;
        }
    }
    // ---- weights: w = exp(-excl_sum(sigma*dt)) * (1 - opacity_before) * alpha ----
    const float sdt = sm.valid ? sigma * (sm.te - sm.ts) : 0.0f;
    float sc[1] = {sdt};
    seg_scan_dpp<1>(sc, heads, lane, maxlen);
    const float excl = sc[0] - sdt;
    const float alpha = 1.0f - expf(-sdt);
    const float opac0 = sm.opac0;
    const float w = expf(-excl) * (1.0f - opac0) * alpha;               // volrend.py:258-267, :361-365; utils.py:712
    const bool keep = sm.valid && !(fr.alpha_thre > 0.f && !(alpha >= fr.alpha_thre));   // utils.py:714-725
    const float wk = keep ? w : 0.0f;
    const float tmid = sm.valid ? (sm.ts + sm.te) / 2.0f : 0.0f;   // an unused column holds whatever the workspace held before: 0 * NaN would poison the depth of its ray
    // ---- per-ray sums of the lane=sample quantities ----
    float acc5[7] = {wk, wk * rgb[0], wk * rgb[1], wk * rgb[2], wk * tmid, keep ? 1.0f : 0.0f, sm.valid ? 1.0f : 0.0f};
    seg_scan_dpp<7>(acc5, heads, lane, maxlen);
    float tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tot[k] = __shfl(acc5[k], tail_lane, 64);
    const int cnt = (int)(tot[6] + 0.5f);                              // samples of this ray in this round
    const float c_new[3] = {c_prev[0] + tot[1], c_prev[1] + tot[2], c_prev[2] + tot[3]};
    const float d_new = d_prev + tot[4];
    const float o_new = opac0 + tot[0];
    // ---- variance against the post-round running means (utils.py:984-999) ----
    float vt[4] = {0.f, 0.f, 0.f, 0.f};
    if (fr.probabilistic) {
        const float e0 = rgb[0] - c_new[0], e1 = rgb[1] - c_new[1], e2 = rgb[2] - c_new[2], ed = tmid - d_new;
        float var4[4] = {wk * (e0 * e0), wk * (e1 * e1), wk * (e2 * e2), wk * (ed * ed)};
        seg_scan_dpp<4>(var4, heads, lane, maxlen);
#pragma unroll
        for (int k = 0; k < 4; ++k) vt[k] = __shfl(var4[k], tail_lane, 64);
    }
    // ---- semantic logits: weights into MFMA layout (column c of tile ct <-> sample lane 32 ct + c) ----
    f32x16 x0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float wm = __shfl(wk, 32 * ct + c, 64);
        f32x16 x;
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sem[ct][k] * wm;
        if (ct == 1) {
            // a run that started in columns 0..31 continues into column 32: carry its partial sums across
            const bool cont = !((heads >> 32) & 1ull);        // wave-uniform: lane 32 continues the run of lane 31
            if (cont) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float cv = __shfl(x0[k], 31 + 32 * h, 64);
                    if (c == 0) x[k] += cv;
                }
            }
        }
        {
            float xv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) xv[k] = x[k];
            seg_scan_dpp<16, true>(xv, __ballot(hm[ct]), lane, maxlen < 32 ? maxlen : 32);
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = xv[k];
        }
        if (ct == 0) x0 = x;
        if (tm[ct]) {                                        // the run's last column owns the totals
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
                if (row < C) fr.sem[(int64_t)raym[ct] * C + row] = s_prev[ct][k] + x[k];
            }
        }
    }
    finish_rays(fr, lane, sm, wc, owner, view, budget, cnt, c_new, d_new, o_new, v_prev, vt, __popcll(__ballot(keep)), __popcll(__ballot(sm.valid)));
}

MNF_DT_END
