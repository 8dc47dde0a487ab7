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
thread_local Profiler::GraphEvents* Profiler::capturing = nullptr;

size_t Profiler::begin(hipStream_t s, int c, double bytes, double flops) {
    std::lock_guard<std::mutex> lk(mu);
    Pending p;
    p.a = ev();
    p.b = ev();
    p.cls = c;
    p.bytes = bytes;
    p.flops = flops;
    p.b_recorded = false;
    if (capturing) {   // external event-record node: re-recorded by every replay, readable from the host afterwards
        OAR_HIP(hipEventRecordWithFlags(p.a, s, hipEventRecordExternal));
        capturing->ev.push_back(p);
        return capturing->ev.size() - 1;
    }
    OAR_HIP(hipEventRecord(p.a, s));
    pending.push_back(p);
    return pending_base + pending.size() - 1;
}
bool Profiler::begin_ext(int c, double bytes, double flops, hipEvent_t& a, hipEvent_t& b) {
    std::lock_guard<std::mutex> lk(mu);
    if (capturing) return false;
    Pending p;
    p.a = a = ev();
    p.b = b = ev();
    p.cls = c;
    p.bytes = bytes;
    p.flops = flops;
    p.b_recorded = true;   // bound to the dispatch itself by hipExtLaunchKernelGGL
    pending.push_back(p);
    return true;
}
void Profiler::end(hipStream_t s, size_t handle) {
    std::lock_guard<std::mutex> lk(mu);
    if (capturing) {
        if (handle < capturing->ev.size()) { (void)hipEventRecordWithFlags(capturing->ev[handle].b, s, hipEventRecordExternal); capturing->ev[handle].b_recorded = true; }
        return;
    }
    // handles count entries since the profiler was created; a flush() in between (only legal with idle streams, i.e. no
    // open scope) retires everything below pending_base
    if (handle < pending_base || handle - pending_base >= pending.size()) return;
    Pending& p = pending[handle - pending_base];
    if (hipEventRecord(p.b, s) == hipSuccess) p.b_recorded = true;
}
static void harvest_locked(Profiler& P, Profiler::GraphEvents& g) {
    if (!g.launched) return;
    for (auto& p : g.ev) {
        float ms = 0.f;
        if (p.b_recorded && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& t = P.totals[p.cls];
            t.launches += 1;
            t.total_ms += ms;
            t.alg_bytes += p.bytes;
            t.alg_flops += p.flops;
        }
    }
    g.launched = false;
}
void Profiler::harvest(GraphEvents& g) {
    std::lock_guard<std::mutex> lk(mu);
    harvest_locked(*this, g);
}
void Profiler::release(GraphEvents& g) {
    std::lock_guard<std::mutex> lk(mu);
    harvest_locked(*this, g);
    for (auto& p : g.ev) { pool.push_back(p.a); pool.push_back(p.b); }
    g.ev.clear();
}
void Profiler::flush() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto it = graphs.begin(); it != graphs.end();) {
        if (auto g = it->lock()) { harvest_locked(*this, *g); ++it; }
        else it = graphs.erase(it);
    }
    for (auto& p : pending) {
        float ms = 0.f;
        if (p.b_recorded && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& t = totals[p.cls];
            t.launches += 1;
            t.total_ms += ms;
            t.alg_bytes += p.bytes;
            t.alg_flops += p.flops;
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending_base += pending.size();
    pending.clear();
}
void Profiler::drop_events() {
    flush();
    std::lock_guard<std::mutex> lk(mu);
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    pool.clear();
    (void)hipGetLastError();   // (results deliberately ignored above: do not leave their error state behind)
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
