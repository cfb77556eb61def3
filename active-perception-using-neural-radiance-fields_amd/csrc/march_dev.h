// Device-side occupancy-grid ray marcher (one lane = one ray).
//
// Semantics follow the reference kernel perception/nerfacc/nerfacc/cuda/csrc/grid.cu:68-282 with
// its helpers include/utils_grid.cuh:10-142; results are bit-identical to oracle/nerfacc_grid.c
// (both must be built with -ffp-contract=off: every t value is a chain of separately rounded
// fp32 operations).  The traversal is written once, parameterised by a `Sink` that receives each
// emitted sample (t_last, t_next, continuous), so the same code serves the count pass, the fill
// pass and the fused per-round marcher of the test-mode renderer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnf {

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ __forceinline__ int clampi(int f, int a, int b) { return max(a, min(f, b)); }

// utils_grid.cuh:10-55
__device__ __forceinline__ bool ray_aabb(const F3 o, const F3 inv, float ray_tmin, float ray_tmax,
                                         const float *__restrict__ aabb, float &tmin, float &tmax) {
    float tmin_t, tmax_t;
    if (inv.x >= 0) { tmin = (aabb[0] - o.x) * inv.x; tmax = (aabb[3] - o.x) * inv.x; }
    else            { tmin = (aabb[3] - o.x) * inv.x; tmax = (aabb[0] - o.x) * inv.x; }
    if (inv.y >= 0) { tmin_t = (aabb[1] - o.y) * inv.y; tmax_t = (aabb[4] - o.y) * inv.y; }
    else            { tmin_t = (aabb[4] - o.y) * inv.y; tmax_t = (aabb[1] - o.y) * inv.y; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (inv.z >= 0) { tmin_t = (aabb[2] - o.z) * inv.z; tmax_t = (aabb[5] - o.z) * inv.z; }
    else            { tmin_t = (aabb[5] - o.z) * inv.z; tmax_t = (aabb[2] - o.z) * inv.z; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return false;
    tmin = fmaxf(tmin, ray_tmin);
    tmax = fminf(tmax, ray_tmax);
    return true;
}

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max) {
    return clampf(t * cone_angle, dt_min, dt_max);  // grid.cu:23-28
}

// grid.cu:158-161 / :199-203: advance t_last in dt steps until the step's midpoint passes `target`.
// The extra `!(nt > t_last)` exit only triggers where the reference would spin forever
// (dt below half an ulp of t_last, or NaN) — it protects the GPU from a hang.
__device__ __forceinline__ void skip_to(float &t_last, float dt, float target) {
    const float hd = dt * 0.5f;
#if defined(MNF_SAMPLER_EXP) && MNF_SAMPLER_EXP == 4
    t_last = target - hd; return;          /* timing experiment: no empty-space stepping (results invalid) */
#endif
    if (t_last >= 0.0f && target + dt > target) {
        // dt still moves `target`, so it moves every smaller non-negative t as well: the hang guard cannot fire inside this skip and
        // the loop is one add, one compare and one exit per step (same t sequence, same exit test)
        // Eight steps at a time while the eighth still falls short: t only grows (dt > 0) and fp32 addition is monotone, so
        // `t7 + hd < target` implies the same for the seven steps before it -- the one-by-one loop would have taken all eight
        // and arrived at the same t8 through the same additions.  One compare and one loop branch per eight additions in
        // empty space, where a ray crosses a 20 cm cell in 25-200 steps of `dt`.
        for (;;) {
            const float t1 = t_last + dt, t2 = t1 + dt, t3 = t2 + dt, t4 = t3 + dt;
            const float t5 = t4 + dt, t6 = t5 + dt, t7 = t6 + dt;
            if (t7 + hd >= target) break;
            t_last = t7 + dt;
        }
        while (!(t_last + hd >= target)) t_last += dt;
        return;
    }
    for (;;) {
        if (t_last + hd >= target) break;
        const float nt = t_last + dt;
        if (!(nt > t_last)) break;
        t_last = nt;
    }
}

struct MarchState {
    float t_last;
    bool continuous;
    int32_t n_samples;
};

// Occupancy accessors: the byte grid exactly as `estimator.binaries` ([X,Y,Z] bools), or a bit-packed copy
// (bit `cell & 31` of word `cell >> 5`), e.g. staged in LDS.
struct ByteGrid {
    const uint8_t *__restrict__ p;
    __device__ __forceinline__ bool operator()(int64_t cell) const { return p[cell] != 0; }
};
struct BitGrid {
    const uint32_t *p;
    __device__ __forceinline__ bool operator()(int64_t cell) const { return (p[cell >> 5] >> (cell & 31)) & 1u; }
};

// March one [this_tmin, this_tmax] segment of one grid level.  `occupied(cell)` answers for that level's
// [X,Y,Z] grid.  Returns through `st`; `sink.sample(t_last, t_next, continuous)` per sample.
template <class Sink, class Occ>
__device__ __forceinline__ void march_segment(const F3 org, const F3 dir, const F3 inv,
                                              float this_tmin, float this_tmax,
                                              const float *__restrict__ ab, const I3 res,
                                              const Occ occupied,
                                              float step_size, float cone_angle, int32_t limit,
                                              MarchState &st, Sink &sink) {
    const float eps = 1e-6f;
    if (!st.continuous) {
        if (step_size <= 0.0f) {
            st.t_last = this_tmin;
        } else {
            float dt = calc_dt(st.t_last, cone_angle, step_size, 1e10f);
            skip_to(st.t_last, dt, this_tmin);
        }
    }
    // setup_traversal, utils_grid.cuh:58-114
    const F3 amin = {ab[0], ab[1], ab[2]}, amax = {ab[3], ab[4], ab[5]};
    const F3 resf = {(float)res.x, (float)res.y, (float)res.z};
    const F3 vox = {(amax.x - amin.x) / resf.x, (amax.y - amin.y) / resf.y, (amax.z - amin.z) / resf.z};
    const float ts = this_tmin + eps, te = this_tmax - eps;
    const F3 rs = {org.x + dir.x * ts, org.y + dir.y * ts, org.z + dir.z * ts};
    const F3 re = {org.x + dir.x * te, org.y + dir.y * te, org.z + dir.z * te};
    I3 cur = {(int)(((rs.x - amin.x) / (amax.x - amin.x)) * resf.x),
              (int)(((rs.y - amin.y) / (amax.y - amin.y)) * resf.y),
              (int)(((rs.z - amin.z) / (amax.z - amin.z)) * resf.z)};
    cur.x = clampi(cur.x, 0, res.x - 1); cur.y = clampi(cur.y, 0, res.y - 1); cur.z = clampi(cur.z, 0, res.z - 1);
    I3 fin = {(int)(((re.x - amin.x) / (amax.x - amin.x)) * resf.x),
              (int)(((re.y - amin.y) / (amax.y - amin.y)) * resf.y),
              (int)(((re.z - amin.z) / (amax.z - amin.z)) * resf.z)};
    fin.x = clampi(fin.x, 0, res.x - 1); fin.y = clampi(fin.y, 0, res.y - 1); fin.z = clampi(fin.z, 0, res.z - 1);
    const I3 start = {cur.x + (dir.x > 0 ? 1 : 0), cur.y + (dir.y > 0 ? 1 : 0), cur.z + (dir.z > 0 ? 1 : 0)};
    const F3 tmx = {((amin.x + (((float)start.x * vox.x) - rs.x)) * inv.x) + this_tmin,
                    ((amin.y + (((float)start.y * vox.y) - rs.y)) * inv.y) + this_tmin,
                    ((amin.z + (((float)start.z * vox.z) - rs.z)) * inv.z) + this_tmin};
    F3 tdist = {dir.x == 0.0f ? this_tmax : tmx.x, dir.y == 0.0f ? this_tmax : tmx.y, dir.z == 0.0f ? this_tmax : tmx.z};
    const F3 stepf = {dir.x == 0.0f ? 0.0f : (dir.x > 0.0f ? 1.0f : -1.0f),
                      dir.y == 0.0f ? 0.0f : (dir.y > 0.0f ? 1.0f : -1.0f),
                      dir.z == 0.0f ? 0.0f : (dir.z > 0.0f ? 1.0f : -1.0f)};
    const I3 step = {(int)stepf.x, (int)stepf.y, (int)stepf.z};
    const F3 dtmp = {vox.x * inv.x * stepf.x, vox.y * inv.y * stepf.y, vox.z * inv.z * stepf.z};
    const F3 delta = {dir.x == 0.0f ? this_tmax : dtmp.x, dir.y == 0.0f ? this_tmax : dtmp.y, dir.z == 0.0f ? this_tmax : dtmp.z};
    const I3 overflow = {fin.x + step.x, fin.y + step.y, fin.z + step.z};
    // linear cell index, updated incrementally (the grids of this path have < 2^31 cells: checked by the host wrappers)
    const int32_t stride_x = res.y * res.z * step.x, stride_y = res.z * step.y, stride_z = step.z;
    int32_t cell = (cur.x * res.y + cur.y) * res.z + cur.z;
    // With dt >= step_size > 0 and t < 2*this_tmax, `t + dt > t` holds whenever step_size still moves 2*this_tmax: the hang
    // guard of the sampling loop cannot fire and is compiled out of the common path (skip_to() makes its own check).
    const bool guard_free = step_size > 0.0f && st.t_last >= 0.0f && (this_tmax * 2.0f + step_size > this_tmax * 2.0f);

    while (limit <= 0 || st.n_samples < limit) {
        float t_traverse = fminf(tdist.x, fminf(tdist.y, tdist.z));
        t_traverse = fminf(t_traverse, this_tmax);
        if (!occupied((int64_t)cell)) {
            if (step_size <= 0.0f) {
                st.t_last = t_traverse;
            } else {
                float dt = calc_dt(st.t_last, cone_angle, step_size, 1e10f);
                skip_to(st.t_last, dt, t_traverse);
            }
            st.continuous = false;
        } else if (guard_free) {
            while (limit <= 0 || st.n_samples < limit) {
                const float dt = calc_dt(st.t_last, cone_angle, step_size, 1e10f);
                if (st.t_last + dt * 0.5f >= t_traverse) break;
                const float t_next = st.t_last + dt;
                sink.sample(st.t_last, t_next, st.continuous, st.n_samples);
                st.n_samples++;
                st.continuous = true;
                st.t_last = t_next;
                if (t_next >= t_traverse) break;
            }
        } else {
            while (limit <= 0 || st.n_samples < limit) {
                float t_next;
                if (step_size <= 0.0f) {
                    t_next = t_traverse;
                } else {
                    float dt = calc_dt(st.t_last, cone_angle, step_size, 1e10f);
                    if (st.t_last + dt * 0.5f >= t_traverse) break;
                    t_next = st.t_last + dt;
                    if (!(t_next > st.t_last)) return;  // hang guard, see skip_to()
                }
                sink.sample(st.t_last, t_next, st.continuous, st.n_samples);
                st.n_samples++;
                st.continuous = true;
                st.t_last = t_next;
                if (t_next >= t_traverse) break;
            }
        }
        // single_traversal, utils_grid.cuh:116-142, without divergent branches: exactly one axis advances
        const bool ax = tdist.x < tdist.y && tdist.x < tdist.z;
        const bool ay = !ax && tdist.y < tdist.z;
        const bool az = !ax && !ay;
        cur.x += ax ? step.x : 0; cur.y += ay ? step.y : 0; cur.z += az ? step.z : 0;
        cell += ax ? stride_x : (ay ? stride_y : stride_z);
        tdist.x = ax ? tdist.x + delta.x : tdist.x;
        tdist.y = ay ? tdist.y + delta.y : tdist.y;
        tdist.z = az ? tdist.z + delta.z : tdist.z;
        if ((ax && cur.x == overflow.x) || (ay && cur.y == overflow.y) || (az && cur.z == overflow.z)) break;
    }
}

// aabbs of the occupancy levels, by value (<= 4 levels); words_per_level: 32-bit words of one level in the bit-packed grid
struct LevelBoxes { float ab[4][6]; int32_t n, words_per_level; };

// Several occupancy levels (grid.cu:125-151): the 2L entry / exit distances of the level boxes (default near / far, utils.py:658; a miss is
// +inf twice), sorted (stable: ties keep the order [t_min of level 0.., t_max of level 0..], as the reference's argsort), cut the ray into
// segments; a segment that begins where a level is entered is marched on that level, one that begins where a level is left is marched on the
// level left next — if the ray is inside it.  `grid_of(level)` returns the occupancy accessor of that level.
template <class Sink, class GridOf>
__device__ __forceinline__ void march_levels(const F3 org, const F3 dir, const F3 inv, float near_plane, float far_plane, const LevelBoxes &boxes,
                                             const I3 res, const GridOf grid_of, float step_size, float cone_angle, int32_t limit,
                                             MarchState &st, Sink &sink) {
    const int L = boxes.n;
    float tv[8]; int ti[8]; bool lhit[4];
    for (int l = 0; l < L; ++l) {
        float t0, t1;
        lhit[l] = ray_aabb(org, inv, -INFINITY, INFINITY, boxes.ab[l], t0, t1);
        tv[l] = lhit[l] ? t0 : INFINITY; tv[L + l] = lhit[l] ? t1 : INFINITY;
        ti[l] = l; ti[L + l] = L + l;
    }
    for (int a = 1; a < 2 * L; ++a) {   // stable insertion sort of <= 8 values
        const float v = tv[a]; const int id = ti[a];
        int b = a - 1;
        while (b >= 0 && tv[b] > v) { tv[b + 1] = tv[b]; ti[b + 1] = ti[b]; --b; }
        tv[b + 1] = v; ti[b + 1] = id;
    }
    for (int i = 0; i < 2 * L - 1; ++i) {
        const bool is_entering = ti[i] < L;
        int level = ti[i] % L;
        if (!lhit[level]) continue;
        if (!is_entering) {
            if (ti[i + 1] < L) continue;
            level = ti[i + 1] % L;
            if (!lhit[level]) continue;
        }
        const float this_tmin = fmaxf(tv[i], near_plane);
        const float this_tmax = fminf(tv[i + 1], far_plane);
        if (this_tmin >= this_tmax) continue;
        march_segment(org, dir, inv, this_tmin, this_tmax, boxes.ab[level], res, grid_of(level), step_size, cone_angle, limit, st, sink);
    }
}

}  // namespace mnf
