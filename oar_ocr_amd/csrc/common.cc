#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace oar {

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }

Profiler& Profiler::get() {
    static Profiler p;
    static bool init = [] { const char* e = getenv("OAR_PROF_DETAIL"); p.detail = e && e[0] == '1'; return true; }();
    (void)init;
    return p;
}
int Profiler::cls(const char* name) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = index.find(name);
    if (it != index.end()) return it->second;
    int id = (int)names.size();
    names.push_back(name);
    index[name] = id;
    oar_prof_entry e;
    memset(&e, 0, sizeof e);
    snprintf(e.name, sizeof e.name, "%s", name);
    totals.push_back(e);
    return id;
}
hipEvent_t Profiler::ev() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    OAR_HIP(hipEventCreate(&e));
    return e;
}
void Profiler::begin(hipStream_t s, int c, double bytes, double flops) {
    std::lock_guard<std::mutex> lk(mu);
    Pending p;
    p.a = ev();
    p.b = ev();
    p.cls = c;
    p.bytes = bytes;
    p.flops = flops;
    OAR_HIP(hipEventRecord(p.a, s));
    pending.push_back(p);
}
void Profiler::end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(mu);
    if (pending.empty()) return;
    // the most recent pending entry on this thread's stream
    (void)hipEventRecord(pending.back().b, s);
}
void Profiler::flush() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& p : pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& t = totals[p.cls];
            t.launches += 1;
            t.total_ms += ms;
            t.alg_bytes += p.bytes;
            t.alg_flops += p.flops;
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
}
void Profiler::reset() {
    flush();
    std::lock_guard<std::mutex> lk(mu);
    for (auto& t : totals) {
        t.launches = 0;
        t.total_ms = t.alg_bytes = t.alg_flops = 0;
    }
}

}  // namespace oar
