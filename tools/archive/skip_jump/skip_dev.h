// Jump form of the marcher's empty-space skip (grid.cu:158-161 / :199-203):
//     while (!(fl(t + hd) >= target)) t = fl(t + dt);          hd = dt / 2, t >= 0, dt > 0, fl = round to nearest fp32
// gives the SAME t, bit for bit, without walking the steps.  Inside one binade [2^e, 2^(e+1)) every t is M * u with u = 2^(e-23) and an
// integer M in [2^23, 2^24); as long as a sum stays inside the binade and c / u is not exactly half-way between two integers,
// fl(M u + c) = (M + rn(c / u)) u: the loop adds the same integer I = rn(dt / u) to M every time and its exit test is
// M + rn(hd / u) >= target / u.  So the number of steps to the exit (or to the end of the binade) is one integer division.  The few cases
// the argument does not cover (a tie, a step that crosses into the next binade, zero / denormal t) take single reference steps.
// Host + device: tests/test_skip_cpu.py compiles this header with g++ and compares it with the loop on millions of random cases.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#ifdef __HIPCC__
#define MNF_SKIP_FN __host__ __device__ __forceinline__
#else
#define MNF_SKIP_FN static inline
#endif

namespace mnf {

MNF_SKIP_FN uint32_t skip_f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
MNF_SKIP_FN float skip_u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// One reference step; returns false when the loop would have stopped (exit test true, or the hang guard of csrc/march_dev.h)
MNF_SKIP_FN bool skip_one(float &t, float dt, float hd, float target) {
    if (t + hd >= target) return false;
    const float nt = t + dt;
    if (!(nt > t)) return false;
    t = nt;
    return true;
}

MNF_SKIP_FN void skip_jump(float &t_last, float dt, float target) {
    const float hd = dt * 0.5f;
    float t = t_last;
    for (;;) {
        const uint32_t tb = skip_f2u(t);
        const uint32_t e = (tb >> 23) & 0xffu;
        bool jumped = false;
        if (e > 24u && e < 255u && !(tb >> 31)) {
            const float inv_u = skip_u2f((277u - e) << 23);                 // 2^(23 - (e - 127)) = 1 / u   (biased exponent 127 + 150 - e)
            const float u = skip_u2f((e - 23u) << 23);                      // 2^(e - 127 - 23)
            const float dq = dt * inv_u, hq = hd * inv_u, tq = target * inv_u;   // exact scalings by a power of two (or overflow to inf: handled by the range tests)
            if (dq >= 1.0f && dq < 8388608.0f && hq < 8388608.0f && tq < 2147483648.0f) {
                const float dfl = floorf(dq), hfl = floorf(hq);
                if (dq - dfl != 0.5f && hq - hfl != 0.5f) {                  // no round-to-even tie
                    const int64_t I = (int64_t)rintf(dq), Ih = (int64_t)rintf(hq);
                    const int64_t M = (int64_t)((tb & 0x7fffffu) | 0x800000u);
                    const int64_t Tc = (int64_t)ceilf(tq);                   // M_k + Ih >= target / u  <=>  M_k + Ih >= ceil(target / u)
                    const int64_t big = I > Ih ? I : Ih;
                    const int64_t L = 16777216 - 1 - big;                   // largest M for which both sums stay inside the binade
                    if (I >= 1 && M <= L) {
                        const int64_t need = Tc - Ih - M;                    // steps until the exit test holds
                        const int64_t k_exit = need <= 0 ? 0 : (need + I - 1) / I;
                        const int64_t k_lim = (L - M) / I;                   // the test and the step of k = 0 .. k_lim are covered
                        if (k_exit <= k_lim) {
                            t_last = (float)(M + k_exit * I) * u;            // <= 2^24: exact
                            return;
                        }
                        t = (float)(M + (k_lim + 1) * I) * u;                // k_lim + 1 steps taken, none of them was the exit
                        jumped = true;
                    }
                }
            }
        }
        if (!jumped && !skip_one(t, dt, hd, target)) { t_last = t; return; }
    }
}

}  // namespace mnf
