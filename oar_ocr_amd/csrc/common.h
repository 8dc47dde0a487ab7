// common.h -- error plumbing, HIP checks, device buffers, per-kernel profiler.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/oar_mi355x.h"

namespace oar {

struct Error : std::runtime_error {
    oar_status code;
    Error(oar_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(oar_status c, const std::string& m) { throw Error(c, m); }

#define OAR_HIP(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            ::oar::fail(_e == hipErrorOutOfMemory ? OAR_OOM : OAR_DEVICE,                                   \
                        std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" +         \
                            std::to_string(__LINE__) + ")");                                                \
    } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute belongs to the (function, DEVICE) pair, so it is set once per device the calling thread
// has current -- a process-wide once-flag would leave an Engine created later on another device_id launching with > 64 KB and no opt-in.
// One flag word per call site (per template instantiation inside a function template), one bit per device ordinal.
#define OAR_MAX_LDS_ONCE(kernel_expr, bytes)                                                                                     \
    do {                                                                                                                          \
        static std::atomic<unsigned long long> oar_lds_done_{0};                                                                  \
        int oar_lds_dev_ = 0;                                                                                                     \
        (void)hipGetDevice(&oar_lds_dev_);                                                                                        \
        const unsigned long long oar_lds_bit_ = 1ull << (oar_lds_dev_ & 63);                                                      \
        if (!(oar_lds_done_.load(std::memory_order_acquire) & oar_lds_bit_)) {                                                    \
            OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_expr), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            oar_lds_done_.fetch_or(oar_lds_bit_, std::memory_order_release);                                                      \
        }                                                                                                                         \
    } while (0)


#define OAR_CHECK(cond, code, msg)                \
    do {                                          \
        if (!(cond)) ::oar::fail((code), (msg));  \
    } while (0)

void set_last_error(const std::string& m);

// Growable device buffer (never shrinks).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) OAR_HIP(hipFree(p));
        p = nullptr;
        size_t want = bytes + (bytes >> 3) + 256;
        OAR_HIP(hipMalloc(&p, want));
        cap = want;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
    ~DevBuf() {
        if (p) { (void)hipFree(p); (void)hipGetLastError(); }   // (an ignored result must not stay behind as the thread's last error)
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Pinned host buffer.
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) OAR_HIP(hipHostFree(p));
        p = nullptr;
        size_t want = bytes + (bytes >> 3) + 256;
        OAR_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
};

// ---------------------------------------------------------------------------------- profiler
// When enabled, launch sites bracket kernels with hipEvents on the launching stream; flush() (after a
// stream sync) turns them into per-class totals.
struct Profiler {
    struct Pending {
        hipEvent_t a, b;
        int cls;
        double bytes, flops;
        bool b_recorded = false;   // an entry whose stop event was never recorded is skipped (hipEventElapsedTime would fail)
    };
    // Events recorded by the event-record nodes of ONE captured hipGraph (Engine keeps one per graph).  Every replay
    // re-records the same events, so they are harvested into the totals before the graph is launched again and at flush().
    struct GraphEvents {
        std::vector<Pending> ev;
        bool launched = false;   // replayed since the last harvest
    };
    std::mutex mu;
    int epoch = 0;        // bumped when enabled / filter change: captured graphs embed the instrumentation of their epoch
    static thread_local GraphEvents* capturing;   // set by Engine while it captures a plan on this thread
    std::vector<std::weak_ptr<GraphEvents>> graphs;
    void harvest(GraphEvents& g);    // the graph's last launch must have been submitted; synchronises on its events
    void release(GraphEvents& g);    // harvest + give the events back to the pool
    bool enabled = false;
    // launch sampling (oar_prof_sampling): launch i since the last call is timed iff i % stride == phase
    std::atomic<int> sample_stride{1}, sample_phase{0};
    std::atomic<long> sample_counter{0};
    bool detail = false;  // OAR_PROF_DETAIL=1: split conv classes by shape
    std::string filter;  // when non-empty only these kernel classes are instrumented: one name, or several separated by commas
    bool filter_match(const char* name) const {
        if (filter.empty()) return true;
        const size_t n = strlen(name);
        for (size_t pos = 0; pos <= filter.size();) {
            size_t e = filter.find(',', pos);
            if (e == std::string::npos) e = filter.size();
            if (e - pos == n && filter.compare(pos, n, name) == 0) return true;
            pos = e + 1;
        }
        return false;
    }
    std::vector<std::string> names;
    std::map<std::string, int> index;
    std::vector<oar_prof_entry> totals;
    std::vector<Pending> pending;
    size_t pending_base = 0;   // handle of pending[0]
    std::vector<hipEvent_t> pool;

    static Profiler& get();
    int cls(const char* name);
    hipEvent_t ev();
    // Returns the handle end() needs: the index of the entry begin() created (in `pending`, or in the capturing graph's
    // list).  Scopes from several host threads / handles interleave in the process-global list, so "the last entry" is not
    // necessarily this scope's.
    size_t begin(hipStream_t s, int cls, double bytes, double flops);
    // Events for ONE kernel launched with hipExtLaunchKernelGGL(..., a, b, 0, ...): they are bound to the dispatch's own
    // completion signal (its begin / end timestamps), so the stream carries no extra barrier packets -- two
    // hipEventRecord calls around a kernel cost ~6 us of idle queue on either side of it (rocprofv3 kernel trace).
    // Returns false while a graph is being captured (use begin / end there).
    bool begin_ext(int cls, double bytes, double flops, hipEvent_t& a, hipEvent_t& b);
    void end(hipStream_t s, size_t handle);
    void flush();  // requires the streams to be idle
    // A stream is about to be destroyed: harvest what is pending and DESTROY every pooled event.  A hipEvent_t remembers the stream it was last
    // recorded on; re-recording a pooled event after that stream is gone made the runtime consult the dead stream (sporadic
    // hipErrorStreamCaptureUnsupported from whatever call came next: seen as a 1-in-8 flake of the profiler-using engine tests).
    void drop_events();
    void reset();
};

struct ProfScope {
    hipStream_t s;
    bool on;
    bool ext = false;
    hipEvent_t a = nullptr, b = nullptr;
    size_t handle = 0;
    // single_launch: the scope covers exactly one kernel and the launch site passes start() / stop() to
    // hipExtLaunchKernelGGL (null events = a plain launch).
    ProfScope(hipStream_t s_, const char* name, double bytes, double flops, bool single_launch = false) : s(s_) {
        Profiler& p = Profiler::get();
        on = p.enabled && p.filter_match(name);
        if (on && !Profiler::capturing) {   // (a captured graph keeps every event node: sampling cannot vary per replay)
            const int stride = p.sample_stride.load(std::memory_order_relaxed), phase = p.sample_phase.load(std::memory_order_relaxed);
            if (phase < 0) on = false;
            else if (stride > 1) on = (p.sample_counter.fetch_add(1, std::memory_order_relaxed) % stride) == phase;
        }
        if (!on) return;
        const int c = p.cls(name);
        if (single_launch && p.begin_ext(c, bytes, flops, a, b)) ext = true;
        else handle = p.begin(s, c, bytes, flops);
    }
    ~ProfScope() {
        if (on && !ext) Profiler::get().end(s, handle);
    }
    hipEvent_t start() const { return ext ? a : nullptr; }
    hipEvent_t stop() const { return ext ? b : nullptr; }
};

}  // namespace oar
