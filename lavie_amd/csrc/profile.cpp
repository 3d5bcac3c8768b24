#include "profile.h"

#include <vector>

namespace lavie {

namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; };
unsigned g_mask = 0;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;
}  // namespace

bool profile_enabled(int cls) { return (g_mask >> cls) & 1u; }

int profile_begin(unsigned mask, int max_events) {
    g_recs.clear();
    g_next = 0;
    if (max_events < 2) max_events = 2;
    while ((int)g_pool.size() < max_events) {
        hipEvent_t e;
        LAVIE_HIP(hipEventCreate(&e));
        g_pool.push_back(e);
    }
    g_recs.reserve(max_events / 2);
    g_mask = mask;
    return 0;
}

ProfileScope::ProfileScope(int cls, hipStream_t s, double flops, double bytes, bool kernel_events_)
    : slot(-1), stream(s), kernel_events(kernel_events_) {
    if (!profile_enabled(cls) || g_next + 2 > g_pool.size()) return;
    Rec r{g_pool[g_next], g_pool[g_next + 1], cls, flops, bytes};
    g_next += 2;
    if (!kernel_events && hipEventRecord(r.a, s) != hipSuccess) return;
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}

ProfileScope::~ProfileScope() {
    if (slot >= 0 && !kernel_events) (void)hipEventRecord(g_recs[slot].b, stream);
}

hipEvent_t ProfileScope::start() const { return g_recs[slot].a; }
hipEvent_t ProfileScope::stop() const { return g_recs[slot].b; }

int profile_end(hipStream_t stream, long long* launches, double* ms, double* flops, double* bytes) {
    g_mask = 0;
    LAVIE_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < KC_COUNT; ++i) { launches[i] = 0; ms[i] = 0; flops[i] = 0; bytes[i] = 0; }
    for (const Rec& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        launches[r.cls] += 1;
        ms[r.cls] += t;
        flops[r.cls] += r.flops;
        bytes[r.cls] += r.bytes;
    }
    g_recs.clear();
    g_next = 0;
    return 0;
}

}  // namespace lavie
