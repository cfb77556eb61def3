// Host harness for csrc/skip_dev.h (tests/test_skip_cpu.py): the jump form of the marcher's empty-space skip against the step-by-step loop, bit for bit.
// build: g++ -O2 -ffp-contract=off -o skip_harness skip_harness.cpp -I<csrc>     run: skip_harness <cases> <seed>  -> prints "ok <cases> <steps walked by the loop>" or the first mismatch
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include "skip_dev.h"

static uint64_t s;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double uni() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }

static uint64_t walked = 0;
static void skip_loop(float &t, float dt, float target) {        // csrc/march_dev.h skip_to as it was: one step at a time
    const float hd = dt * 0.5f;
    for (;;) {
        if (t + hd >= target) break;
        const float nt = t + dt;
        if (!(nt > t)) break;
        t = nt; ++walked;
    }
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 1000000;
    s = argc > 2 ? strtoull(argv[2], nullptr, 10) * 2654435761ull + 88172645463325252ull : 88172645463325252ull;
    for (long c = 0; c < cases; ++c) {
        const int kind = (int)(rnd() % 8);
        float t = (float)(exp(uni() * 12.0 - 5.0));                     // 0.0067 .. 1100
        if (kind == 0) t = (float)(uni() * 0.3);                         // small t, also exactly 0 sometimes
        if (kind == 1 && (rnd() & 3) == 0) t = 0.0f;
        float dt;
        const float cone = (rnd() & 1) ? 0.004f : (float)(uni() * 0.02);
        const float step = (float)(exp(uni() * 5.0 - 7.5));              // 5.5e-4 .. 0.08
        dt = fmaxf(step, fminf(t * cone, 1e10f));                        // calc_dt
        if (kind == 2) dt = (float)(exp(uni() * 14.0 - 12.0));          // anything from 6e-6 to 7
        if (kind == 3) {                                                 // a tie on purpose: dt = (k + 0.5) ulp(t)
            const float u = ldexpf(1.0f, ilogbf(t > 0 ? t : 1.0f) - 23);
            dt = ((float)(rnd() % 4096) + 0.5f) * u;
            if (rnd() & 1) dt *= 2.0f;                                   // ... or hd = (k + 0.5) u
        }
        if (kind == 4) {                                                 // start close below a power of two: the skip crosses binades
            const int e = (int)(rnd() % 12) - 4;
            t = ldexpf(1.0f, e) * (float)(1.0 - uni() * 1e-3);
            dt = fmaxf(step, t * cone);
        }
        float span = (float)(uni() * uni() * 6.0);                      // up to 6 m of empty space, mostly short
        if (kind == 5) span = (float)(uni() * 400.0);
        if (kind == 6) span = dt * (float)(uni() * 3.0);                 // a step or two
        float target = t + span;
        if (kind == 7 && (rnd() & 7) == 0) target = t * (float)uni();   // target behind t: no step at all
        if (!(dt > 0.0f) || !(target + dt > target)) continue;           // (the caller's precondition)
        if ((double)span / dt > 3e6) continue;                           // keep the reference loop affordable
        float a = t, b = t;
        skip_loop(a, dt, target);
        mnf::skip_jump(b, dt, target);
        if (mnf::skip_f2u(a) != mnf::skip_f2u(b)) {
            printf("MISMATCH case %ld kind %d: t %.9g (0x%08x) dt %.9g (0x%08x) target %.9g (0x%08x): loop %.9g (0x%08x) jump %.9g (0x%08x)\n", c, kind, t, mnf::skip_f2u(t), dt,
                   mnf::skip_f2u(dt), target, mnf::skip_f2u(target), a, mnf::skip_f2u(a), b, mnf::skip_f2u(b));
            return 1;
        }
    }
    printf("ok %ld %llu\n", cases, (unsigned long long)walked);
    return 0;
}
