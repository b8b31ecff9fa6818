// Event-based launch profiler behind bpm_prof_* (include/bpmult_hip.h).
#include <mutex>
#include <vector>

#include "../../include/bpmult_hip.h"
#include "bpm_prof.h"

unsigned g_bpm_prof_mask = 0;

namespace {
struct Rec { hipEvent_t a, b; int kind; double work, bytes; };
std::mutex g_mu;                 // forward runs on the Python thread, backward on autograd's
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[BPM_K_COUNT];

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

void bpm_prof_open(int kind, hipStream_t s, double work, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r;
    r.a = get_event(); r.b = get_event(); r.kind = kind; r.work = work; r.bytes = bytes;
    hipEventRecord(r.a, s);
    g_open[kind] = r.b;
    g_recs.push_back(r);
}

void bpm_prof_close(int kind, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipEventRecord(g_open[kind], s);
}

extern "C" int bpm_prof_enable(unsigned kind_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_bpm_prof_mask = kind_mask;
    return 0;
}

// Sums (and clears) the records of `kind`: elapsed ms between each launch's two
// events, algorithmic work, launch count.  Synchronises on the recorded events.
extern "C" int bpm_prof_collect2(int kind, double* total_ms, double* total_work, double* total_bytes, int* launches) {
    if (kind < 0 || kind >= BPM_K_COUNT || !total_ms || !total_work || !launches) return BPM_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0, work = 0, bytes = 0;
    int n = 0;
    std::vector<Rec> keep;
    for (const Rec& r : g_recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        ms += t; work += r.work; bytes += r.bytes; ++n;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.swap(keep);
    *total_ms = ms; *total_work = work; *launches = n;
    if (total_bytes) *total_bytes = bytes;
    return 0;
}

extern "C" int bpm_prof_collect(int kind, double* total_ms, double* total_work, int* launches) {
    return bpm_prof_collect2(kind, total_ms, total_work, nullptr, launches);
}

// A HIP stream at the lowest (low_priority != 0) or default priority the device offers, for the engine's
// off-critical-path work: the dispatcher then fills CUs from the main stream's kernels first.
extern "C" int bpm_stream_create(int low_priority, void** out) {
    if (!out) return BPM_ERR_ARG;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    hipStream_t s;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, low_priority ? least : 0);
    if (e != hipSuccess) return (int)e;
    *out = (void*)s;
    return 0;
}

extern "C" int bpm_stream_priority_range(int* least, int* greatest) {
    if (!least || !greatest) return BPM_ERR_ARG;
    return (int)hipDeviceGetStreamPriorityRange(least, greatest);
}
