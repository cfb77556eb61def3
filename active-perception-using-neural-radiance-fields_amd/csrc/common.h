// Shared host-side helpers for libmi355nerf.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mi355nerf.h"

// The fused field kernels exist twice, once per matrix-core operand type: field.hip / train.hip are compiled as two
// translation units (plain: fp16 operands, the reference's tcnn arithmetic; -DMNF_BF16: bf16 operands, BASELINE config 5) and
// everything type-dependent lives in mnf::f16 / mnf::bf16.  The C entry points are compiled once and dispatch on the handle.
#ifdef MNF_BF16
#define MNF_DT bf16
#else
#define MNF_DT f16
#endif
#define MNF_DT_BEGIN namespace mnf { namespace MNF_DT {
#define MNF_DT_END }}

namespace mnf {

void set_error(const char *fmt, ...);

#define MNF_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            mnf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MNF_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define MNF_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            mnf::set_error(__VA_ARGS__);      \
            return MNF_ERR_INVALID;           \
        }                                     \
    } while (0)

inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return MNF_ERR_HIP;
    }
    return MNF_OK;
}

inline hipStream_t as_stream(mnf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Diagnostic knobs (tools/README.md) exist only in the -DMNF_DIAG build (libmi355nerf_diag.so): the product library never
// reads the environment, so no stray variable can change its results, its round schedule or its timings.
#ifdef MNF_DIAG
inline const char *diag_env(const char *name) { return getenv(name); }
#else
inline const char *diag_env(const char *) { return nullptr; }
#endif

// The library's side streams (api.cpp): ONE small set per device for the whole process, shared by every user (the train step's scatter streams, the
// render jobs) and never destroyed.  Measured on MI355X / ROCm 7.2 (tools/exp_after_training3.py, profiles/r03_stream_count.txt): with one more pair of
// streams alive in the process (a second field that has trained: round 3 first gave every train state its own pair) each SMALL kernel of a train step
// took ~45 us instead of ~5 and the step 5.2 ms instead of 3.9 — for the rest of the process, also for fields created later.  Users fork from and join
// the caller's stream with events, so sharing a stream only serialises work that was not meant to overlap anyway.
hipStream_t shared_side_stream(int slot);          // slot 0 .. kSharedSideStreams - 1 of the current device; NULL on failure (error set)
constexpr int kSharedSideStreams = 3;

// per-kernel timing for bench.py (api.cpp): a no-op unless mnf_profile_begin() was called on this thread
bool prof_on();
int prof_start(const char *label, hipStream_t s);
void prof_stop(int id, hipStream_t s);
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(const char *label, hipStream_t stream) : id(prof_start(label, stream)), s(stream) {}
    ~ProfScope() { prof_stop(id, s); }
};

}  // namespace mnf
