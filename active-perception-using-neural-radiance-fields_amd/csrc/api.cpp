// Error reporting and library-level queries of libmi355nerf.so.
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace mnf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mnf

// ------------------------------------------------------------------ optional per-kernel timing (bench.py's roofline figures)
// Between mnf_profile_begin and mnf_profile_end every launch site wrapped in a ProfScope is bracketed by a hipEvent pair
// on its launch stream; mnf_profile_end synchronises the events and sums the milliseconds per label.
namespace mnf {
namespace {
struct ProfRec { const char *label; hipEvent_t start, stop; };
struct ProfState {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    std::map<std::string, std::pair<double, int64_t>> totals;
};
// process-wide (not thread-local): torch runs backward passes on its own autograd thread, and those launches belong to the
// same measurement; the mutex only guards the bookkeeping
ProfState g_prof;
std::mutex g_prof_mutex;
hipEvent_t prof_event() {
    if (!g_prof.pool.empty()) { hipEvent_t e = g_prof.pool.back(); g_prof.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_on() { return g_prof.on; }
// ---- the process-wide side streams (common.h)
hipStream_t shared_side_stream(int slot) {
    static std::mutex mu;
    static std::vector<std::vector<hipStream_t>> per_device;
    if (slot < 0 || slot >= kSharedSideStreams) { set_error("shared_side_stream: bad slot %d", slot); return nullptr; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("shared_side_stream: no current device"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if ((int)per_device.size() <= dev) per_device.resize(dev + 1);
    std::vector<hipStream_t> &v = per_device[dev];
    if (v.empty()) {
        v.assign(kSharedSideStreams, nullptr);
        for (int k = 0; k < kSharedSideStreams; ++k) {
            const hipError_t e = hipStreamCreateWithFlags(&v[k], hipStreamNonBlocking);
            if (e != hipSuccess) { set_error("shared_side_stream: %s", hipGetErrorString(e)); v.clear(); return nullptr; }
        }
    }
    return v[slot];
}

int prof_start(const char *label, hipStream_t s) {
    if (!g_prof.on || !label) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ProfRec r = {label, prof_event(), prof_event()};
    if (!r.start || !r.stop) return -1;
    (void)hipEventRecord(r.start, s);
    g_prof.recs.push_back(r);
    return (int)g_prof.recs.size() - 1;
}
void prof_stop(int id, hipStream_t s) {
    if (id < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (id < (int)g_prof.recs.size()) (void)hipEventRecord(g_prof.recs[id].stop, s);
}
}  // namespace mnf

extern "C" int mnf_profile_begin(void) {
    mnf::g_prof.on = true;
    mnf::g_prof.totals.clear();
    return MNF_OK;
}

extern "C" int mnf_profile_end(double *field_ms_host, int64_t *launches_host) {
    using namespace mnf;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof.on = false;
    for (auto &r : g_prof.recs) {
        float t = 0.f;
        MNF_HIP(hipEventSynchronize(r.stop));
        MNF_HIP(hipEventElapsedTime(&t, r.start, r.stop));
        auto &acc = g_prof.totals[r.label];
        acc.first += t; acc.second += 1;
        g_prof.pool.push_back(r.start); g_prof.pool.push_back(r.stop);
    }
    g_prof.recs.clear();
    const auto it = g_prof.totals.find("field_render");
    if (field_ms_host) *field_ms_host = it == g_prof.totals.end() ? 0.0 : it->second.first;
    if (launches_host) *launches_host = it == g_prof.totals.end() ? 0 : it->second.second;
    return MNF_OK;
}

extern "C" int mnf_profile_query(const char *label_host, double *ms_host, int64_t *launches_host) {
    MNF_REQUIRE(label_host, "profile_query: null label");
    const auto it = mnf::g_prof.totals.find(label_host);
    if (ms_host) *ms_host = it == mnf::g_prof.totals.end() ? 0.0 : it->second.first;
    if (launches_host) *launches_host = it == mnf::g_prof.totals.end() ? 0 : it->second.second;
    return MNF_OK;
}

extern "C" const char *mnf_last_error(void) { return mnf::g_err; }
extern "C" int mnf_version(void) { return 1; }
extern "C" int mnf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
