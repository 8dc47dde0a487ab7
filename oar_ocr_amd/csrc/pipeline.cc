#include "pipeline.h"

#include <algorithm>
#include <map>
#include <string>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <sched.h>

namespace oar {

namespace {
struct PhaseTimer {  // OAR_TIMING=1: prints host-side phase times of each predict() to stderr
    bool on;
    std::chrono::steady_clock::time_point t0;
    std::vector<std::pair<const char*, double>> marks;
    PhaseTimer() {
        static const bool en = [] { const char* e = getenv("OAR_TIMING"); return e && (e[0] == '1' || e[0] == '2'); }();
        on = en;
        t0 = std::chrono::steady_clock::now();
    }
    void mark(const char* name) {
        if (!on) return;
        auto t = std::chrono::steady_clock::now();
        marks.push_back({name, std::chrono::duration<double, std::milli>(t - t0).count()});
        t0 = t;
    }
    void dump(const char* title) {
        if (!on) return;
        static const bool seq = [] { const char* e = getenv("OAR_TIMING"); return e && e[0] == '2'; }();   // 2: the marks in order, not summed by name
        if (seq) {
            fprintf(stderr, "[timing] %s (sequence):", title);
            for (auto& m : marks) fprintf(stderr, " %s=%.2f", m.first, m.second);
            fprintf(stderr, "\n");
        }
        std::map<std::string, double> agg;
        std::vector<std::string> order;
        for (auto& m : marks) { if (!agg.count(m.first)) order.push_back(m.first); agg[m.first] += m.second; }
        fprintf(stderr, "[timing] %s:", title);
        for (auto& k : order) fprintf(stderr, " %s=%.2fms", k.c_str(), agg[k]);
        fprintf(stderr, "\n");
    }
};
thread_local PhaseTimer* g_timer = nullptr;
inline void tmark(const char* n) { if (g_timer) g_timer->mark(n); }
}  // namespace


// ================================================================================================= thread pool
int ThreadPool::available_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = n > 0 ? std::min(n, c) : c; }
    // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = unlimited)
    double quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
        fclose(f);
    } else {
        if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lf", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lf", &period) != 1) period = 0; fclose(fp); }
    }
    if (quota > 0 && period > 0) { const int c = (int)(quota / period); if (c >= 1) n = n > 0 ? std::min(n, c) : c; }
    return n > 0 ? n : 1;
}

ThreadPool::ThreadPool(int n) {
    if (n <= 0) { const char* e = getenv("OAR_HOST_THREADS"); n = e ? atoi(e) : 0; }
    // default: the CPUs this process may use (affinity mask and container quota), at most 16 -- and at most the affinity mask minus two when
    // that mask has six or more CPUs: a predict() also runs an uploader and an enqueuer thread next to the caller, and on a rank PINNED to
    // its slice of the host, polling workers on every core of the slice starve them.  Measured with taskset on the bench workload
    // (profiles/r4/host_cores_sweep.txt): 8 cores 1600-1750 images/s with a pool of 8, 2165-2177 with 6; 16 cores 1650-1990 with 16, 2167-2310
    // with 14; 4 cores 1757-1778 with 4 = 1752-1763 with 3; 2 cores 1359-1417 with 2, 1075-1093 with 1.  (A CPU-time quota alone -- 16 CPUs
    // of a 256-thread host on the 1-GPU bench box -- does not have the problem: threads float, 16 measured >= 14 there.)
    if (n <= 0) {
        n = std::min<int>(available_cpus(), 16);
        cpu_set_t set;
        CPU_ZERO(&set);
        const int aff = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 0;
        if (aff >= 6 && n > aff - 2) n = aff - 2;
    }
    if (n <= 0) n = 1;
    if (n > 64) n = 64;
    for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this] { loop(); });
}
ThreadPool::~ThreadPool() {
    stop_.store(true, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(mu_);
        cv_.notify_all();
    }
    for (auto& t : workers_) t.join();
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
// Wait for another THREAD of this call (uploader / enqueuer): poll briefly, then give the core away on every round -- on a rank with two
// or four cores the thread waited for may not be running at all while this one spins (measured with taskset: device-resident throughput
// fell by 30 % at 2 cores when the waits only polled).
template <typename Pred>
static inline void wait_for_thread(Pred done) {
    for (int spins = 0; !done(); ++spins) {
        if (spins < 256) cpu_relax();
        else std::this_thread::yield();
    }
}
void ThreadPool::loop() {
    // OAR_POOL_SPIN_MS: how long an idle worker keeps polling before it parks (default 25 ms ~ one predict() of the
    // bench workload, so workers stay hot across the recognition phase of continuous serving)
    static const int spin_ms = [] { const char* e = getenv("OAR_POOL_SPIN_MS"); int v = e ? atoi(e) : 25; return v < 0 ? 0 : v; }();
    static const bool pool_yield = [] { const char* e = getenv("OAR_POOL_YIELD"); return !(e && e[0] == '0'); }();   // A/B knob
    int seen = 0;
    while (!stop_.load(std::memory_order_acquire)) {
        // wait for a new generation: poll while the pool is active (and for at most spin_ms), else park
        int g = gen_.load(std::memory_order_acquire);
        if (g == seen) {
            auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while ((g = gen_.load(std::memory_order_acquire)) == seen && !stop_.load(std::memory_order_acquire)) {
                // poll; after ~50 us without work give the core away on every round (a no-op while nobody else wants it): on a rank with two or
                // four cores the uploader / enqueuer threads of the call are what a polling worker would otherwise be running instead of
                if (spins < 2048 || !pool_yield) cpu_relax();
                else std::this_thread::yield();
                const bool idle = active_.load(std::memory_order_acquire) == 0;
                if (idle || ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(spin_ms))) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return stop_.load() || gen_.load() != seen || (idle && active_.load() != 0); });
                    t0 = std::chrono::steady_clock::now();
                }
            }
            if (stop_.load(std::memory_order_acquire)) return;
        }
        seen = g;
        if (selftest_worker_delay_) selftest_worker_delay_();   // oar_host_pool_selftest: a worker that is late with its descriptor read
        const std::function<void(int)>* fn = fn_.load(std::memory_order_acquire);
        const int count = count_.load(std::memory_order_acquire);
        int i;
        while (claim(g, count, i)) {
            try {
                (*fn)(i);
            } catch (...) {
                std::lock_guard<std::mutex> g2(err_mu_);
                if (!err_) err_ = std::current_exception();
            }
            done_.fetch_add(1, std::memory_order_acq_rel);
        }
    }
}
void ThreadPool::set_active(bool on) {
    if (on) {
        if (active_.fetch_add(1, std::memory_order_acq_rel) == 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();   // parked workers start polling: the first parallel_for of the phase finds them awake
        }
    } else {
        active_.fetch_sub(1, std::memory_order_acq_rel);
    }
}
bool ThreadPool::claim(int gen, int count, int& index) {
    uint64_t v = state_.load(std::memory_order_acquire);
    for (;;) {
        if ((int)(v >> 32) != gen) return false;            // the job this descriptor belongs to is over
        const uint32_t i = (uint32_t)(v & 0xffffffffu);     // kClosed (0xffffffff) while the next job is being published
        if (i >= (uint32_t)count) return false;
        if (state_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel, std::memory_order_acquire)) { index = (int)i; return true; }
    }
}
void ThreadPool::parallel_for(int count, const std::function<void(int)>& fn) {
    if (count <= 0) return;
    if (workers_.empty() || count == 1) {
        for (int i = 0; i < count; ++i) fn(i);
        return;
    }
    // Publish the job.  Order matters: a worker that saw the PREVIOUS generation late (after that job was finished by
    // the others) reads fn_ / count_ with no re-validation, so it may pick up THIS job's larger count while the claim word
    // still says (previous generation, previous count) -- and would then claim index `previous count` of a finished
    // job.  The claim word is therefore closed first (new generation, index = kClosed): from here on every claim made
    // with an older generation fails, whatever count it was made with.  count_ is a release store, so a worker that
    // read the new count also sees the closed word.  Only then the descriptor, the open word, and the generation.
    constexpr uint64_t kClosed = 0xffffffffull;
    const int g = gen_.load(std::memory_order_relaxed) + 1;
    state_.store(((uint64_t)(uint32_t)g << 32) | kClosed, std::memory_order_seq_cst);
    fn_.store(&fn, std::memory_order_release);
    done_.store(0, std::memory_order_release);
    count_.store(count, std::memory_order_release);
    if (selftest_publish_delay_) selftest_publish_delay_();   // oar_host_pool_selftest: widen the window between descriptor and claim word
    state_.store((uint64_t)(uint32_t)g << 32, std::memory_order_seq_cst);
    gen_.store(g, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(mu_);
        cv_.notify_all();
    }
    int i;
    while (claim(g, count, i)) {  // the caller works too
        try {
            fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> g2(err_mu_);
            if (!err_) err_ = std::current_exception();
        }
        done_.fetch_add(1, std::memory_order_acq_rel);
    }
    while (done_.load(std::memory_order_acquire) < count) cpu_relax();
    // every claimed index has finished; a late worker still holding this job's descriptor fails its next claim (generation)
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> g2(err_mu_);
        e = err_;
        err_ = nullptr;
    }
    if (e) std::rethrow_exception(e);
}

// ================================================================================================= detector
static const int kDbSrc[3] = {2, 1, 0};  // ColorOrder::BGR (models/detection/db.rs:409-415)

Detector::Detector(const uint8_t* onnx, size_t len, const oar_det_cfg& cfg) : cfg_(cfg) {
    if (cfg_.limit_side_len == 0) cfg_.limit_side_len = 960;
    if (cfg_.max_side_limit == 0) cfg_.max_side_limit = 4000;
    if (cfg_.max_candidates == 0) cfg_.max_candidates = 1000;
    OAR_CHECK(cfg_.box_type == 0 || cfg_.box_type == 1, OAR_INVALID_INPUT, "box_type must be 0 (Quad) or 1 (Poly)");
    OAR_CHECK(cfg_.score_mode == 0 || cfg_.score_mode == 1, OAR_INVALID_INPUT, "score_mode must be 0 (fast) or 1 (slow)");
    eng_.reset(new Engine(onnx, len, cfg_.device_id));
    pool_.reset(new ThreadPool(cfg_.host_threads));
    OAR_HIP(hipSetDevice(eng_->device()));
    // OAR_CONTOUR_CUS=n (experiment): the streams that carry the border follower / unclip / box scores are confined to n CUs spread over
    // the XCDs (hipExtStreamCreateWithCUMask).  The network's stream keeps all CUs -- its persistent kernels are sized for 256 -- so this reserves nothing;
    // measured in profiles/r4/gpu_contours_cu_mask.txt
    static const int contour_cus = [] { const char* e = getenv("OAR_CONTOUR_CUS"); return e ? atoi(e) : 0; }();
    if (contour_cus > 0 && contour_cus <= 256) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < contour_cus; ++i) { const int cu = (int)((long)i * 256 / contour_cus); mask[cu >> 5] |= 1u << (cu & 31); }
        OAR_HIP(hipExtStreamCreateWithCUMask(&copy_stream_, 8, mask));
        OAR_HIP(hipExtStreamCreateWithCUMask(&score_stream_, 8, mask));
    } else {
        OAR_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
        OAR_HIP(hipStreamCreateWithFlags(&score_stream_, hipStreamNonBlocking));
    }
    OAR_HIP(hipStreamCreateWithFlags(&upload_stream_, hipStreamNonBlocking));
    OAR_HIP(hipStreamCreateWithFlags(&upload_stream2_, hipStreamNonBlocking));
    OAR_HIP(hipEventCreateWithFlags(&stage_free_, hipEventDisableTiming));
}
Detector::~Detector() {
    Profiler::get().drop_events();   // (see Profiler::drop_events: pooled events must not outlive the streams they were recorded on)
    if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); (void)hipStreamDestroy(copy_stream_); }
    if (score_stream_) { (void)hipStreamSynchronize(score_stream_); (void)hipStreamDestroy(score_stream_); }
    if (upload_stream_) { (void)hipStreamSynchronize(upload_stream_); (void)hipStreamDestroy(upload_stream_); }
    if (upload_stream2_) { (void)hipStreamSynchronize(upload_stream2_); (void)hipStreamDestroy(upload_stream2_); }
    if (stage_free_) (void)hipEventDestroy(stage_free_);
    for (hipEvent_t e : upload_done_) (void)hipEventDestroy(e);
    for (hipEvent_t e : upload_done2_) (void)hipEventDestroy(e);
    for (hipEvent_t e : sub_events_) (void)hipEventDestroy(e);
    for (hipEvent_t e : mask_ready_) (void)hipEventDestroy(e);
    for (hipEvent_t e : score_done_) (void)hipEventDestroy(e);
}

namespace {
struct ContourSizes { size_t rows, band_y, n_bands, lists, scratch, ctrl, table_dev, ctrl_host, table, packed, packed_words_per_page; };
ContourSizes contour_sizes(int pages, int sub_pages, int H, int W, int n_sub) {
    ContourSizes z;
    const size_t maxb = (size_t)pp::trace_max_bands(H);
    z.packed_words_per_page = std::max<size_t>((size_t)H * W / 4, 4096);   // a page's border chains: 1 word per 4 pixels, else host fallback
    z.rows = (size_t)sub_pages * H;
    z.band_y = (size_t)sub_pages * maxb * 2 * sizeof(int32_t);
    z.n_bands = (size_t)sub_pages * sizeof(int32_t);
    z.lists = (size_t)sub_pages * ContourBufs::kSegsPerPage * 2 * sizeof(uint32_t);
    z.scratch = (size_t)sub_pages * H * W * 2 * sizeof(uint32_t);
    z.ctrl = (size_t)std::max(n_sub, 1) * pp::kTraceCtlWords * sizeof(uint32_t);
    z.table_dev = (size_t)sub_pages * ContourBufs::kSegsPerPage * sizeof(pp::SegRec);
    z.ctrl_host = z.ctrl;
    z.table = (size_t)pages * ContourBufs::kSegsPerPage * sizeof(pp::SegRec);
    z.packed = (size_t)pages * z.packed_words_per_page * sizeof(uint32_t);
    return z;
}
}  // namespace

bool ContourBufs::fits(int pages, int sub_pages, int H, int W, int n_sub) const {
    const ContourSizes z = contour_sizes(pages, sub_pages, H, W, n_sub);
    return z.rows <= rows.cap && z.band_y <= band_y.cap && z.n_bands <= n_bands.cap && z.lists <= lists.cap && z.scratch <= scratch.cap && z.ctrl <= ctrl.cap &&
           z.table_dev <= table_dev.cap && z.ctrl_host <= ctrl_host.cap && z.table <= table.cap && z.packed <= packed.cap;
}

void ContourBufs::reserve(int pages, int sub_pages, int H, int W, int n_sub) {
    const ContourSizes z = contour_sizes(pages, sub_pages, H, W, n_sub);
    packed_words_per_page = z.packed_words_per_page;
    rows.reserve(z.rows); band_y.reserve(z.band_y); n_bands.reserve(z.n_bands); lists.reserve(z.lists); scratch.reserve(z.scratch); ctrl.reserve(z.ctrl);
    table_dev.reserve(z.table_dev); ctrl_host.reserve(z.ctrl_host); table.reserve(z.table); packed.reserve(z.packed);
}

namespace {
struct Candidate { float pts[8]; std::vector<host::Pt> contour; /* kCandSlow: the border chain; kCandPoly: the approximated polygon */ };
// what a contour becomes: the mini box alone (Quad + ScoreMode::Fast), the mini box + the chain it is scored on (Quad +
// ScoreMode::Slow), or the Douglas-Peucker polygon (BoxType::Poly, db_bitmap.rs:37-47)
enum CandKind { kCandFast = 0, kCandSlow = 1, kCandPoly = 2 };

// a9: one contour -> candidate (false when rejected)
bool contour_candidate(const host::Contour& c, Candidate& cd, int keep_contour = kCandFast) {
    if (keep_contour == kCandPoly) {   // polygons_from_bitmap: >= 4 border points, approx_poly_dp(0.002 * perimeter), >= 4 vertices
        if (c.pts.size() < 4) return false;
        const float epsilon = 0.002f * host::perimeter(c.pts);
        cd.contour = host::approx_poly_dp(c.pts, epsilon);
        return cd.contour.size() >= 4;
    }
    host::Pt mb[4];
    float min_side = 0.f;
    if (!host::contour_mini_box(c, mb, min_side)) return false;
    if (min_side < 3.0f) return false;  // DBPostProcess::min_size (db_postprocess.rs:83)
    for (int i = 0; i < 4; ++i) { cd.pts[i * 2] = mb[i].x; cd.pts[i * 2 + 1] = mb[i].y; }
    if (keep_contour) cd.contour = c.pts;
    return true;
}

void page_candidates(const uint8_t* mask, int H, int W, uint32_t max_candidates, std::vector<Candidate>& out, int keep_contour = kCandFast) {
    out.clear();
    std::vector<host::Contour> cs = host::find_contours(mask, W, H, max_candidates);
    for (auto& c : cs) {
        Candidate cd;
        if (contour_candidate(c, cd, keep_contour)) out.push_back(std::move(cd));
    }
}

// stage 2 of a sub-batch: per-contour geometry (simplify -> hull -> min-area rect -> mini box) spread over the pool in chunks of
// 8 contours; discovery order is preserved
void contours_to_candidates(ThreadPool& pool, std::vector<std::vector<host::Contour>>& cs, int nb, std::vector<Candidate>* out, int keep_contour) {
    struct Chunk { int page; size_t c0, c1; };
    std::vector<Chunk> chunks;
    const size_t step = 8;
    for (int k = 0; k < nb; ++k)
        for (size_t c0 = 0; c0 < cs[k].size(); c0 += step) chunks.push_back({k, c0, std::min(cs[k].size(), c0 + step)});
    std::vector<std::vector<Candidate>> res(nb);
    std::vector<std::vector<uint8_t>> ok(nb);
    for (int k = 0; k < nb; ++k) { res[k].resize(cs[k].size()); ok[k].assign(cs[k].size(), 0); }
    pool.parallel_for((int)chunks.size(), [&](int i) {
        const Chunk& ch = chunks[i];
        for (size_t c = ch.c0; c < ch.c1; ++c) ok[ch.page][c] = contour_candidate(cs[ch.page][c], res[ch.page][c], keep_contour) ? 1 : 0;
    });
    for (int k = 0; k < nb; ++k) {
        out[k].clear();
        for (size_t c = 0; c < cs[k].size(); ++c) if (ok[k][c]) out[k].push_back(std::move(res[k][c]));
    }
}

// Two-stage variant for a sub-batch: contour tracing is inherently serial per page (one worker per page), the
// per-contour geometry is then spread over the whole pool; discovery order is preserved.
// `masks`: nb bit planes of H rows x ceil(W / 8) bytes (pp::pack_mask_bits), hw = bytes per plane
void subbatch_candidates(ThreadPool& pool, const uint8_t* masks, size_t hw, int H, int W, int nb, uint32_t max_candidates,
                         std::vector<Candidate>* out /* [nb] */, int keep_contour = kCandFast) {
    const int row_bytes = (W + 7) / 8;
    const auto t_entry = std::chrono::steady_clock::now();
    // stage 1: contour tracing, parallel over (page, row band) -- bands are cut at fully-blank rows
    struct Band { int page, y0, y1; };
    std::vector<Band> bands;
    static const int bands_mult = [] { const char* e = getenv("OAR_BANDS_MULT"); int v = e ? atoi(e) : 2; return v > 0 ? v : 2; }();
    const int bands_per_page = std::max(1, std::min(12, (pool.size() + 1) * bands_mult / std::max(nb, 1)));
    auto t_setup = std::chrono::steady_clock::now();
    std::vector<std::vector<int>> cuts(nb);
    pool.parallel_for(nb, [&](int k) { cuts[k] = host::blank_row_bands_bits(masks + (size_t)k * hw, row_bytes, H, bands_per_page); });
    for (int k = 0; k < nb; ++k)
        for (size_t i = 0; i + 1 < cuts[k].size(); ++i) bands.push_back({k, cuts[k][i], cuts[k][i + 1]});
    std::vector<std::vector<host::Contour>> band_cs(bands.size());
    std::vector<double> tpage(bands.size(), 0.0);
    auto ta = std::chrono::steady_clock::now();
    // tallest bands first: the pool hands out indices in order, and a tall band claimed last would be the whole tail of the stage
    std::vector<int> by_rows(bands.size());
    for (size_t i = 0; i < bands.size(); ++i) by_rows[i] = (int)i;
    std::stable_sort(by_rows.begin(), by_rows.end(), [&](int a, int b) { return bands[a].y1 - bands[a].y0 > bands[b].y1 - bands[b].y0; });
    if (keep_contour == kCandFast) {
        // Quad boxes, fast score (the hot path): a band's worker turns its contours into candidates right away -- the mini box of a contour
        // costs about what following it does since round 5, so the bands still balance -- and the few hundred border chains of a band are
        // released by the thread that allocated them: handing 2 600 small vectors per sub-batch back to the caller cost it 0.3 ms of
        // cross-thread frees after the two stages had taken 0.3 ms together.  `take(max_candidates)` counts CONTOURS in discovery order, so
        // every contour keeps its slot (ok = 0: no candidate) until the bands of a page are concatenated.
        struct BandOut { std::vector<Candidate> cand; std::vector<uint8_t> ok; };
        std::vector<BandOut> bo(bands.size());
        pool.parallel_for((int)bands.size(), [&](int slot) {
            auto t0 = std::chrono::steady_clock::now();
            const int i = by_rows[slot];
            const Band& bd = bands[i];
            std::vector<host::Contour> cs = host::find_contours_band_bits(masks + (size_t)bd.page * hw, row_bytes, W, bd.y0, bd.y1, max_candidates, true);
            BandOut& o = bo[i];
            o.ok.assign(cs.size(), 0);
            o.cand.reserve(cs.size() / 4 + 4);
            for (size_t c = 0; c < cs.size(); ++c) {
                Candidate cd;
                if (contour_candidate(cs[c], cd, kCandFast)) { o.ok[c] = 1; o.cand.push_back(std::move(cd)); }
            }
            tpage[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
        std::vector<size_t> taken(nb, 0);
        for (int k = 0; k < nb; ++k) out[k].clear();
        for (size_t i = 0; i < bands.size(); ++i) {   // band order == raster discovery order
            const int k = bands[i].page;
            size_t next = 0;
            for (size_t c = 0; c < bo[i].ok.size() && taken[k] < max_candidates; ++c, ++taken[k])
                if (bo[i].ok[c]) out[k].push_back(std::move(bo[i].cand[next++]));
        }
        if (g_timer && g_timer->on) {
            double mx = 0, sum = 0;
            for (double t : tpage) { mx = std::max(mx, t); sum += t; }
            fprintf(stderr, "[timing]   subbatch nb=%d bands=%zu (contours + mini boxes per band: max %.2f avg %.2f ms) total=%.2fms\n", nb, bands.size(), mx,
                    sum / std::max<size_t>(tpage.size(), 1), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
        }
        return;
    }
    pool.parallel_for((int)bands.size(), [&](int slot) {
        auto t0 = std::chrono::steady_clock::now();
        const int i = by_rows[slot];
        const Band& bd = bands[i];
        band_cs[i] = host::find_contours_band_bits(masks + (size_t)bd.page * hw, row_bytes, W, bd.y0, bd.y1, max_candidates, false);
        tpage[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
    std::vector<std::vector<host::Contour>> cs(nb);
    for (size_t i = 0; i < bands.size(); ++i) {   // band order == raster discovery order; `take(max_candidates)` afterwards
        auto& dst = cs[bands[i].page];
        for (auto& c : band_cs[i]) { if (dst.size() >= max_candidates) break; dst.push_back(std::move(c)); }
    }
    auto tb = std::chrono::steady_clock::now();
    contours_to_candidates(pool, cs, nb, out, keep_contour);
    if (g_timer && g_timer->on) {
        auto tc = std::chrono::steady_clock::now();
        double mx = 0, sum = 0;
        for (double t : tpage) { mx = std::max(mx, t); sum += t; }
        fprintf(stderr, "[timing]   subbatch nb=%d setup=%.2fms stage1=%.2fms (per-band max %.2f avg %.2f, %zu bands) stage2=%.2fms total=%.2fms\n", nb,
                std::chrono::duration<double, std::milli>(ta - t_setup).count(), std::chrono::duration<double, std::milli>(tb - ta).count(), mx,
                sum / std::max<size_t>(tpage.size(), 1), tpage.size(), std::chrono::duration<double, std::milli>(tc - tb).count(),
                std::chrono::duration<double, std::milli>(tc - t_entry).count());
    }
}

// Border chains of one traced record (a word stream of pp::trace_contours) -> (start key, contour) items
void parse_traced_record(const pp::SegRec& r, const uint32_t* packed, std::vector<std::pair<uint32_t, host::Contour>>& items) {
    const uint32_t* w = packed + r.off;
    uint32_t at = 0;
    for (uint32_t c = 0; c < r.n_contours; ++c) {
        OAR_CHECK(at < r.used, OAR_INTERNAL, "contour tracer: corrupt stream");
        const uint32_t hdr = w[at++];
        const uint32_t cnt = hdr & 0x7fffffffu;
        OAR_CHECK(cnt >= 1 && at + cnt <= r.used, OAR_INTERNAL, "contour tracer: corrupt stream");
        host::Contour ct;
        ct.hole = (hdr >> 31) != 0;
        ct.pts.resize(cnt);
        for (uint32_t i = 0; i < cnt; ++i) { const uint32_t v = w[at + i]; ct.pts[i] = host::Pt{(float)(v & 0xffffu), (float)(v >> 16)}; }
        const uint32_t first = w[at];
        at += cnt;
        items.emplace_back((first >> 16) << 16 | (first & 0xffffu), std::move(ct));   // key = y << 16 | x of the start pixel: raster order
    }
}

// A record the kernel flagged: follow rows [y0, y1) x columns [x0, x1) of the page on the host (db_host.cc), from a copy of the
// mask with everything outside the rectangle blanked -- the rectangle is bounded by background, so this changes nothing inside it.
void trace_record_on_host(const pp::SegRec& r, const uint8_t* mask, int H, int W, size_t max_contours, std::vector<std::pair<uint32_t, host::Contour>>& items) {
    std::vector<uint8_t> tmp((size_t)r.y1 * W, 0);
    for (int y = r.y0; y < r.y1; ++y) std::memcpy(tmp.data() + (size_t)y * W + r.x0, mask + (size_t)y * W + r.x0, (size_t)(r.x1 - r.x0));
    std::vector<int32_t> plane((size_t)r.y1 * W);
    (void)H;
    for (auto& c : host::find_contours_band(tmp.data(), W, r.y1, r.y0, r.y1, max_contours, plane.data())) {
        const uint32_t key = ((uint32_t)c.pts[0].y << 16) | (uint32_t)c.pts[0].x;
        items.emplace_back(key, std::move(c));
    }
}

// One page's contours in find_contours order from its records: all borders sorted by start pixel, then take(max_candidates)
void merge_traced_page(std::vector<std::pair<uint32_t, host::Contour>>& items, size_t max_candidates, std::vector<host::Contour>& dst) {
    std::sort(items.begin(), items.end(), [](const std::pair<uint32_t, host::Contour>& a, const std::pair<uint32_t, host::Contour>& b) { return a.first < b.first; });
    dst.clear();
    for (auto& it : items) { if (dst.size() >= max_candidates) break; dst.push_back(std::move(it.second)); }
}

// The sub-batch variant fed by pp::trace_contours (contours.hip): the border chains of every segment arrive as word streams in
// pinned host memory; what the kernel flagged (too large for LDS, table / stream overflow) is followed here from the mask,
// which `fetch_mask` brings over on demand.
void subbatch_candidates_traced(ThreadPool& pool, const uint32_t* ctrl, const pp::SegRec* table, uint32_t table_cap, const uint32_t* packed, int H, int W, int nb,
                                uint32_t max_candidates, const std::function<const uint8_t*(int)>& fetch_mask, std::vector<Candidate>* out /* [nb] */,
                                int keep_contour = kCandFast) {
    std::vector<std::vector<host::Contour>> cs(nb);
    const uint32_t n_rec = ctrl[pp::kTraceCtlSegments];
    int fallbacks = 0;
    if (ctrl[pp::kTraceCtlOverflow] || n_rec > table_cap) {   // table too small for this sub-batch: every page on the host
        std::vector<const uint8_t*> masks(nb);
        for (int k = 0; k < nb; ++k) masks[k] = fetch_mask(k);
        pool.parallel_for(nb, [&](int k) { cs[k] = host::find_contours(masks[k], W, H, max_candidates); });
        fallbacks = nb;
    } else {
        std::vector<std::vector<uint32_t>> by_page(nb);
        std::vector<const uint8_t*> masks(nb, nullptr);
        for (uint32_t i = 0; i < n_rec; ++i) {
            const pp::SegRec& r = table[i];
            OAR_CHECK(r.page >= 0 && r.page < nb && r.y0 >= 0 && r.y0 < r.y1 && r.y1 <= H && r.x0 >= 0 && r.x0 < r.x1 && r.x1 <= W, OAR_INTERNAL,
                      "contour tracer: corrupt segment record");
            by_page[r.page].push_back(i);
            if (r.flags) { if (!masks[r.page]) masks[r.page] = fetch_mask(r.page); ++fallbacks; }   // fetched before the workers start
        }
        pool.parallel_for(nb, [&](int k) {
            std::vector<std::pair<uint32_t, host::Contour>> items;
            for (uint32_t i : by_page[k]) {
                const pp::SegRec& r = table[i];
                if (r.flags) trace_record_on_host(r, masks[k], H, W, max_candidates, items);
                else parse_traced_record(r, packed, items);
            }
            merge_traced_page(items, max_candidates, cs[k]);
        });
    }
    auto tb = std::chrono::steady_clock::now();
    contours_to_candidates(pool, cs, nb, out, keep_contour);
    if (g_timer && g_timer->on)
        fprintf(stderr, "[timing]   subbatch (gpu-traced) nb=%d records=%u host-fallback=%d stage2=%.2fms\n", nb, n_rec, fallbacks,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count());
}

// `gpu_unclip` (optional): the candidates' unclipped polygons as pp::unclip_quads left them (n_pts -1: not handled there)
// a11-a12 for ONE candidate: score gate -> unclip -> second mini box -> min side -> scale / round / clamp (db_bitmap.rs:255-277); false = dropped
bool finish_one_box(const Candidate& cd, float score, float box_thresh, float unclip_ratio, float wscale, float hscale, float dwf, float dhf,
                    const pp::UnclipOut* gpu_unclip, float* pts8) {
    if (score < box_thresh) return false;
    std::vector<host::Pt> un;
    if (gpu_unclip && gpu_unclip->n_pts >= 0) {
        const pp::UnclipOut& u = *gpu_unclip;
        un.resize((size_t)u.n_pts);
        for (int k = 0; k < u.n_pts; ++k) un[k] = {u.pts[k * 2], u.pts[k * 2 + 1]};
    } else {
        host::Pt mb[4];
        for (int k = 0; k < 4; ++k) mb[k] = {cd.pts[k * 2], cd.pts[k * 2 + 1]};
        un = host::unclip(mb, unclip_ratio);
    }
    if (un.empty()) return false;
    host::Pt bp[4];
    float sside = 0.f;
    if (!host::mini_box(un, bp, sside)) return false;
    if (sside < 3.0f + 2.0f) return false;
    for (int k = 0; k < 4; ++k) {
        float x = std::round(bp[k].x * wscale), y = std::round(bp[k].y * hscale);
        x = x < 0.0f ? 0.0f : (x > dwf ? dwf : x);
        y = y < 0.0f ? 0.0f : (y > dhf ? dhf : y);
        pts8[k * 2] = x; pts8[k * 2 + 1] = y;
    }
    return true;
}

void finish_boxes(const std::vector<Candidate>& cands, const float* scores, int H, int W, uint32_t src_w, uint32_t src_h, float box_thresh,
                  float unclip_ratio, DetBoxes& out, const pp::UnclipOut* gpu_unclip = nullptr) {
    out.pts.clear(); out.scores.clear();
    const float wscale = (float)src_w / (float)W, hscale = (float)src_h / (float)H;
    const float dwf = (float)src_w, dhf = (float)src_h;
    for (size_t i = 0; i < cands.size(); ++i) {
        float p8[8];
        if (!finish_one_box(cands[i], scores[i], box_thresh, unclip_ratio, wscale, hscale, dwf, dhf, gpu_unclip ? gpu_unclip + i : nullptr, p8)) continue;
        out.pts.insert(out.pts.end(), p8, p8 + 8);
        out.scores.push_back(scores[i]);
    }
}

// BoxType::Poly tail of polygons_from_bitmap (db_bitmap.rs:49-78): score gate -> unclip -> min side of the unclipped polygon's
// mini box -> every vertex scaled, rounded and clamped
void finish_polys(const std::vector<Candidate>& cands, const float* scores, int H, int W, uint32_t src_w, uint32_t src_h, float box_thresh,
                  float unclip_ratio, DetBoxes& out) {
    out.pts.clear(); out.scores.clear(); out.counts.clear();
    const float wscale = (float)src_w / (float)W, hscale = (float)src_h / (float)H;
    const float dwf = (float)src_w, dhf = (float)src_h;
    for (size_t i = 0; i < cands.size(); ++i) {
        const float score = scores[i];
        if (score < box_thresh) continue;
        std::vector<host::Pt> un = host::unclip_poly(cands[i].contour, unclip_ratio);
        if (un.empty()) continue;
        host::Pt bp[4];
        float sside = 0.f;
        if (!host::mini_box(un, bp, sside)) continue;
        if (sside < 3.0f + 2.0f) continue;
        for (const host::Pt& p : un) {
            float x = std::round(p.x * wscale), y = std::round(p.y * hscale);
            x = x < 0.0f ? 0.0f : (x > dwf ? dwf : x);
            y = y < 0.0f ? 0.0f : (y > dhf ? dhf : y);
            out.pts.push_back(x); out.pts.push_back(y);
        }
        out.counts.push_back((uint32_t)un.size());
        out.scores.push_back(score);
    }
}
}  // namespace

std::atomic<int> g_inject_batched_det_failures{0};   // oar_debug_inject_failure("batched_detection", n): the next n multi-page runs throw

void Detector::run(const std::vector<PageRef>& pages, float thresh, float box_thresh, float unclip, std::vector<DetBoxes>& out,
                   std::vector<const uint8_t*>* dev_pages_out, const ReadyFn& on_ready) {
    std::lock_guard<std::mutex> lk(mu_);
    ThreadPool::ActiveScope hot(*pool_);   // the geometry workers poll for work during a detector call only
    const int n = (int)pages.size();
    if (n > 1 && g_inject_batched_det_failures.load(std::memory_order_relaxed) > 0 && g_inject_batched_det_failures.fetch_sub(1) > 0)
        fail(OAR_DEVICE, "injected failure of a batched detection (oar_debug_inject_failure)");
    out.assign(n, DetBoxes());
    if (n == 0) return;
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    // Host pages get their slot in the HBM staging area now; the copies themselves are issued sub-batch by sub-batch on
    // upload_stream_ (run_group), so the H2D of sub-batch k+1 runs while the network of sub-batch k does: from pageable
    // memory a hipMemcpyAsync occupies the CALLING thread for the duration (~0.7 ms per 8 pages of 960^2 at ~35 GB/s), which
    // is exactly the time the GPU needs no new work from it.  Only the first sub-batch's upload is exposed.
    size_t total = 0;
    for (auto& p : pages) {
        OAR_CHECK(p.w > 0 && p.h > 0 && (p.host || p.dev), OAR_INVALID_INPUT, "detector: empty page");
        if (!p.dev) total += ((size_t)p.w * p.h * 3 + 255) & ~(size_t)255;
    }
    if (total > pages_dev_.cap) { OAR_HIP(hipStreamSynchronize(s)); OAR_HIP(hipStreamSynchronize(upload_stream_)); OAR_HIP(hipStreamSynchronize(upload_stream2_)); pages_dev_.reserve(total); }
    // whatever still reads the staging area on the engine stream (the previous call's crop kernels) must be done before
    // this call's uploads overwrite it
    OAR_HIP(hipEventRecord(stage_free_, s));
    OAR_HIP(hipStreamWaitEvent(upload_stream_, stage_free_, 0));
    OAR_HIP(hipStreamWaitEvent(upload_stream2_, stage_free_, 0));
    page_ptrs_.assign(n, nullptr);
    upload_src_.assign(n, nullptr);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (pages[i].dev) { page_ptrs_[i] = pages[i].dev; continue; }
        uint8_t* d = pages_dev_.as<uint8_t>() + off;
        page_ptrs_[i] = d;
        upload_src_[i] = pages[i].host;   // pending until run_group reaches the page's sub-batch
        off += ((size_t)pages[i].w * pages[i].h * 3 + 255) & ~(size_t)255;
    }
    if (dev_pages_out) *dev_pages_out = page_ptrs_;
    // group by resized shape, first-appearance order (models/detection/db.rs:297-309)
    struct Group { uint32_t rh, rw; std::vector<int> idx; };
    std::vector<Group> groups;
    // small pages (h + w < 64) are padded with black to at least 32 x 32, origin at (0, 0); box coordinates are still
    // scaled with the ORIGINAL size (ImageScaleInfo keeps the pre-padding src_h / src_w, resize_detection.rs:170,191)
    det_src_.assign(n, nullptr); det_w_.assign(n, 0); det_h_.assign(n, 0);
    size_t pad_total = 0;
    for (int i = 0; i < n; ++i)
        if (pages[i].h + pages[i].w < 64) pad_total += ((size_t)std::max(pages[i].w, 32u) * std::max(pages[i].h, 32u) * 3 + 255) & ~(size_t)255;
    if (pad_total > padded_dev_.cap) { OAR_HIP(hipStreamSynchronize(s)); padded_dev_.reserve(pad_total); }
    size_t pad_off = 0;
    for (int i = 0; i < n; ++i) {
        det_src_[i] = page_ptrs_[i]; det_w_[i] = pages[i].w; det_h_[i] = pages[i].h;
        if (pages[i].h + pages[i].w >= 64) continue;
        const uint32_t pw = std::max(pages[i].w, 32u), ph = std::max(pages[i].h, 32u);
        if (pw == pages[i].w && ph == pages[i].h) continue;
        if (upload_src_[i]) {   // the padding copy below reads the page on the engine stream: upload it there, now
            OAR_HIP(hipMemcpyAsync(const_cast<uint8_t*>(page_ptrs_[i]), upload_src_[i], (size_t)pages[i].w * pages[i].h * 3, hipMemcpyHostToDevice, s));
            upload_src_[i] = nullptr;
        }
        uint8_t* d = padded_dev_.as<uint8_t>() + pad_off;
        OAR_HIP(hipMemsetAsync(d, 0, (size_t)pw * ph * 3, s));
        OAR_HIP(hipMemcpy2DAsync(d, (size_t)pw * 3, page_ptrs_[i], (size_t)pages[i].w * 3, (size_t)pages[i].w * 3, pages[i].h, hipMemcpyDeviceToDevice, s));
        det_src_[i] = d; det_w_[i] = pw; det_h_[i] = ph;
        pad_off += ((size_t)pw * ph * 3 + 255) & ~(size_t)255;
    }
    for (int i = 0; i < n; ++i) {
        uint32_t w = det_w_[i], h = det_h_[i];
        uint32_t rh, rw;
        host::det_resize_dims(w, h, cfg_.limit_side_len, cfg_.limit_type, cfg_.max_side_limit, rh, rw);
        bool placed = false;
        for (auto& g : groups) if (g.rh == rh && g.rw == rw) { g.idx.push_back(i); placed = true; break; }
        if (!placed) groups.push_back({rh, rw, {i}});
    }
    // one shape group (the common case): pages become final sub-batch by sub-batch, in page order.  Several groups: the
    // groups interleave page indices, so the callback fires once at the end to keep the caller's page order.
    const bool stream_out = on_ready && groups.size() == 1;
    for (auto& g : groups) run_group(g.idx, pages, g.rh, g.rw, thresh, box_thresh, unclip, out, stream_out ? on_ready : ReadyFn());
    if (on_ready && !stream_out) on_ready(0, n);
}

void Detector::run_group(const std::vector<int>& idx, const std::vector<PageRef>& pages, uint32_t rh, uint32_t rw, float thresh,
                         float box_thresh, float unclip, std::vector<DetBoxes>& out, const ReadyFn& on_ready) {
    // The group is cut into sub-batches whose GPU work (normalize -> network -> threshold -> mask D2H) is enqueued
    // back to back; the host traces the contours of sub-batch i while the GPU is already on sub-batch i+1.
    // Pages are independent in detection, so the result is identical to one big batch.
    hipStream_t s = eng_->stream();
    const int B = (int)idx.size();
    const size_t plane = (size_t)rh * rw;
    // DB normalisation constants (processors/normalization.rs:142-143, f32): alpha = scale/std, beta = -mean/std
    const float scale = 1.0f / 255.0f;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float alpha[3], beta[3];
    for (int c = 0; c < 3; ++c) { alpha[c] = scale / stdv[c]; beta[c] = -mean[c] / stdv[c]; }
    // pages per sub-batch.  Round 5: 16 (8 / 16 / 8 for a 32-page call) now that the host stages of a sub-batch cost a third of what they did --
    // 96 instead of 160 detector launches per call, each with its ~5 us boundary: +3.9 % and +4.7 % images/s over three alternating pairs each
    // (8: 2350 / 2409 / 2406, 16: 2527 / 2457 / 2462; profiles/r5/det_sub_ab.txt), 4 pinned cores +3 %.  A rank with two cores is host-bound in this
    // phase and prefers the finer grain (8: 1988, 16: 1927): a pool of <= 2 threads keeps 8.
    static const int sub_env = [] { const char* e = getenv("OAR_DET_SUB"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    // Round 6: 12 (6 / 10 / 10 / 6) on the real-size detector, whose sub-batch takes 1.5x as long on the GPU: +2 % over five alternating pairs
    // (16: 2178 / 2214 / 2246 / 2283 / 2242, 12: 2278 / 2297 / 2293 / 2320 / 2288; profiles/r6/det_sub_ab.txt)
    const int kSub = sub_env > 0 ? sub_env : (pool_->size() <= 1 ? 8 : 12);   // size() = workers next to the calling thread
    // sub-batch sizes: the LAST one is half-size (its contour tracing is the only host work the GPU cannot overlap); when the
    // pages still have to be uploaded the FIRST one is half-size too (the GPU idles until its pages have crossed PCIe:
    // 0.9 ms for ten 960^2 pages, OAR_DET_FIRST=0 disables); the others share the rest evenly
    static const int first_half = [] { const char* e = getenv("OAR_DET_FIRST"); return e ? atoi(e) : 1; }();
    bool host_pages = false;
    for (int b = 0; b < B; ++b) host_pages = host_pages || upload_src_[idx[b]] != nullptr;
    std::vector<int> sb_off{0};
    {
        const int base = std::min(B, kSub);
        static const int last_env = [] { const char* e = getenv("OAR_DET_LAST"); return e ? atoi(e) : 0; }();   // (A/B knob: pages in the last sub-batch)
        const int last = B > base ? (last_env > 0 ? std::min(last_env, base) : std::max(1, base / 2)) : 0;
        const int first = (first_half && host_pages && B - last > base) ? (first_half >= 2 ? std::min(first_half, base) : std::max(1, base / 2)) : 0;   // (OAR_DET_FIRST >= 2: that many pages)
        int rest = B - last - first;
        if (first) sb_off.push_back(first);
        int left = (rest + base - 1) / base;
        for (; left > 0; --left) { const int take = (rest + left - 1) / left; sb_off.push_back(sb_off.back() + take); rest -= take; }
        if (last) sb_off.push_back(B);
    }
    const int nsub = (int)sb_off.size() - 1;
    int SB = 0;
    for (int i = 0; i < nsub; ++i) SB = std::max(SB, sb_off[i + 1] - sb_off[i]);

    // output geometry from the plan of the largest sub-batch (shape inference only)
    // normalisation folded into the stem convolution when the graph allows it (OAR_FUSE_STEM=0: separate normalize kernel)
    static const bool fuse_stem = [] { const char* e = getenv("OAR_FUSE_STEM"); return !(e && e[0] == '0'); }();
    const bool stem = fuse_stem && SB <= 32 && eng_->stem_fusable();
    const Plan& plan0 = eng_->plan_for({SB, 3, (int64_t)rh, (int64_t)rw}, true, false, nullptr, stem);
    OAR_CHECK(!plan0.outputs.empty(), OAR_INTERNAL, "DB: no output returned from inference");
    OAR_CHECK(plan0.outputs[0].dims.size() == 4, OAR_SHAPE_MISMATCH, "DB: expected a 4-D [batch,1,H,W] output");
    const int C = (int)plan0.outputs[0].dims[1], H = (int)plan0.outputs[0].dims[2], W = (int)plan0.outputs[0].dims[3];
    const size_t hw = (size_t)H * W;

    size_t need_in = (size_t)SB * plane * 3 * sizeof(float);
    size_t need_rs = 0;
    for (int b = 0; b < B; ++b) if (det_w_[idx[b]] != rw || det_h_[idx[b]] != rh) need_rs += (plane * 3 + 255) & ~(size_t)255;
    if (need_in > input_f32_.cap || need_rs > resized_dev_.cap || (size_t)B * hw > mask_dev_.cap || (size_t)B * hw * 4 > probs_keep_.cap ||
        (cfg_.use_dilation && (size_t)B * hw > mask_dil_.cap)) {
        OAR_HIP(hipStreamSynchronize(s));
        input_f32_.reserve(need_in); resized_dev_.reserve(need_rs); mask_dev_.reserve((size_t)B * hw); probs_keep_.reserve((size_t)B * hw * 4);
        if (cfg_.use_dilation) mask_dil_.reserve((size_t)B * hw);
    }
    mask_host_.reserve((size_t)B * hw);
    const size_t hbits = (size_t)H * ((W + 7) / 8);   // bytes of one mask as a bit plane (what the host border follower reads back)
    if ((size_t)B * hbits > mask_bits_.cap) { OAR_HIP(hipStreamSynchronize(s)); OAR_HIP(hipStreamSynchronize(copy_stream_)); mask_bits_.reserve((size_t)B * hbits); }
    // a8 on the host pool (default) or on the GPU: oar_det_cfg.gpu_contours, overridden by OAR_GPU_CONTOURS=0|1
    static const int gpu_contours_env = [] { const char* e = getenv("OAR_GPU_CONTOURS"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
    const bool gpu_contours = gpu_contours_env >= 0 ? gpu_contours_env == 1 : cfg_.gpu_contours != 0;
    if (gpu_contours) {
        if (!trace_.fits(B, SB, H, W, nsub)) { OAR_HIP(hipStreamSynchronize(s)); OAR_HIP(hipStreamSynchronize(copy_stream_)); }
        trace_.reserve(B, SB, H, W, nsub);
    }
    while ((int)sub_events_.size() < nsub) { hipEvent_t e; OAR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); sub_events_.push_back(e); }
    while ((int)mask_ready_.size() < nsub) { hipEvent_t e; OAR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); mask_ready_.push_back(e); }

    float* probs = probs_keep_.as<float>();   // [B][H][W] channel-0 planes, kept for the score kernel
    while ((int)upload_done_.size() < nsub) { hipEvent_t e; OAR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); upload_done_.push_back(e); }

    // ---- uploader: the sub-batches' host pages -> HBM, in order, on upload_stream_.  From pageable memory a
    // hipMemcpyAsync occupies its CALLING thread for the whole copy (~0.7 ms per 8 pages of 960^2), so the copies are issued
    // from a helper thread while this one keeps the kernel queue full; `issued` counts the sub-batches whose upload (and
    // the event that follows it) has been submitted -- an event must be recorded before a stream can wait on it.
    // OAR_UPLOAD_THREADS (1 or 2, default 1): with 2, two helper threads on two streams take alternate pages of every sub-batch (one thread's
    // staged copy runs at ~31 GB/s, barely ahead of the detector -- 11 pages / ms against 8 -- and the first sub-batch's upload is exposed).
    // Measured on the bench workload, five alternations on one box: 2198 images/s against 2221 with one thread -- no gain, stays off.
    static const int n_up = [] { const char* e = getenv("OAR_UPLOAD_THREADS"); const int v = e ? atoi(e) : 1; return v >= 2 ? 2 : 1; }();
    while ((int)upload_done2_.size() < nsub) { hipEvent_t e; OAR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); upload_done2_.push_back(e); }
    struct Uploader {
        std::thread th[2];
        std::atomic<int> issued[2];
        std::atomic<bool> failed{false};
        std::mutex mu;
        std::string error;
        Uploader() { issued[0].store(0); issued[1].store(0); }
        ~Uploader() { for (auto& t : th) if (t.joinable()) t.join(); }
    } up;
    bool any_upload = false;
    for (int b = 0; b < B; ++b) any_upload = any_upload || upload_src_[idx[b]] != nullptr;
    if (any_upload) {
        const int device = eng_->device();
        for (int t = 0; t < n_up; ++t)
            up.th[t] = std::thread([&, device, t] {
                try {
                    OAR_HIP(hipSetDevice(device));
                    hipStream_t us = t == 0 ? upload_stream_ : upload_stream2_;
                    for (int sb = 0; sb < nsub; ++sb) {
                        for (int k = sb_off[sb] + t; k < sb_off[sb + 1]; k += n_up) {
                            const int pi = idx[k];
                            if (!upload_src_[pi]) continue;
                            OAR_HIP(hipMemcpyAsync(const_cast<uint8_t*>(page_ptrs_[pi]), upload_src_[pi], (size_t)pages[pi].w * pages[pi].h * 3, hipMemcpyHostToDevice, us));
                            upload_src_[pi] = nullptr;
                        }
                        OAR_HIP(hipEventRecord(t == 0 ? upload_done_[sb] : upload_done2_[sb], us));
                        up.issued[t].store(sb + 1, std::memory_order_release);
                    }
                } catch (const std::exception& e) {
                    { std::lock_guard<std::mutex> lk(up.mu); up.error = e.what(); }
                    up.failed.store(true, std::memory_order_release);
                }
            });
    }

    size_t rs_off = 0;
    // GPU work of one sub-batch: (wait for its pages) -> normalize -> network -> threshold -> mask D2H on the copy stream
    auto enqueue = [&](int sb) {
        const int b0 = sb_off[sb], nb = sb_off[sb + 1] - b0;
        const uint8_t* srcs[32];
        if (any_upload) {
            for (int t = 0; t < n_up; ++t)
                wait_for_thread([&] { return up.issued[t].load(std::memory_order_acquire) > sb || up.failed.load(std::memory_order_acquire); });
            if (up.failed.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lk(up.mu); fail(OAR_DEVICE, "page upload failed: " + up.error); }
            OAR_HIP(hipStreamWaitEvent(s, upload_done_[sb], 0));
            if (n_up > 1) OAR_HIP(hipStreamWaitEvent(s, upload_done2_[sb], 0));
        }
        for (int k = 0; k < nb; ++k) {
            const int pi = idx[b0 + k];
            const uint8_t* src = det_src_[pi];
            if (det_w_[pi] != rw || det_h_[pi] != rh) {
                uint8_t* dst = resized_dev_.as<uint8_t>() + rs_off;
                pp::resize_triangle(s, src, (int)det_w_[pi], (int)det_h_[pi], dst, (int)rw, (int)rh);
                src = dst;
                rs_off += (plane * 3 + 255) & ~(size_t)255;
            }
            if (nb <= 32) srcs[k] = src;
            else pp::normalize(s, src, input_f32_.as<float>() + (size_t)k * plane * 3, 1, (int64_t)plane, kDbSrc, alpha, beta, 1);
        }
        k::StemU8 st{};
        if (stem) {
            for (int k = 0; k < nb; ++k) st.pages[k] = srcs[k];
            for (int c = 0; c < 3; ++c) { st.src[c] = kDbSrc[c]; st.alpha[c] = alpha[c]; st.beta[c] = beta[c]; }
        } else if (nb <= 32) {
            pp::normalize_pages(s, srcs, nb, input_f32_.as<float>(), (int64_t)plane, kDbSrc, alpha, beta, 1);   // one launch per sub-batch
        }
        const Plan& plan = stem ? eng_->run_stem(st, {nb, 3, (int64_t)rh, (int64_t)rw}) : eng_->run(input_f32_.as<float>(), {nb, 3, (int64_t)rh, (int64_t)rw}, true);
        const PlanOutput& po = plan.outputs[0];
        OAR_CHECK(po.dims.size() == 4 && po.dims[0] == nb && po.dims[1] == C && po.dims[2] == H && po.dims[3] == W, OAR_SHAPE_MISMATCH,
                  "DB: inconsistent output shape across sub-batches");
        // only channel 0 is used (processors/db_postprocess.rs:122-123); keep it past the next sub-batch's arena reuse
        // OAR_DB_FINISH_FUSED=0 restores copy2d + threshold (+ pack_mask_bits on the copy stream)
        static const bool fused_finish_env = [] { const char* e = getenv("OAR_DB_FINISH_FUSED"); return !(e && e[0] == '0'); }();
        const bool fused_finish = fused_finish_env && !gpu_contours && !cfg_.use_dilation;   // nothing else reads the byte mask then
        const uint8_t* traced = mask_dev_.as<uint8_t>() + (size_t)b0 * hw;
        if (fused_finish) {
            pp::db_keep_and_pack(s, eng_->out_ptr(po.loc), (int64_t)(hw * C), probs + (size_t)b0 * hw, mask_bits_.as<uint8_t>() + (size_t)b0 * hbits, nb, H, W, thresh);
        } else {
            k::copy2d(s, eng_->out_ptr(po.loc), probs + (size_t)b0 * hw, nb, (int)hw, (int)(hw * C), (int)hw);
            pp::threshold(s, probs + (size_t)b0 * hw, mask_dev_.as<uint8_t>() + (size_t)b0 * hw, (int64_t)nb * hw, thresh);
        }
        if (cfg_.use_dilation) {   // db_postprocess.rs:163-168: the contours are traced on the dilated mask, the scores still read pred
            pp::dilate3x3(s, traced, mask_dil_.as<uint8_t>() + (size_t)b0 * hw, nb, H, W);
            traced = mask_dil_.as<uint8_t>() + (size_t)b0 * hw;
        }
        OAR_HIP(hipEventRecord(mask_ready_[sb], s));
        // OAR_GPU_CONTOURS_INLINE=1 (experiment, round 5; default 0): the border follower ON the detector's stream, between this sub-batch's network and the
        // next one's.  Measured (profiles/r5/gpu_follower_inline.txt): 35.6 ms per step against 31.3 on its own stream and 14-25 with the host tracer -- the
        // follower's launch time is not queueing behind the network but its own serial work: one wave per mask segment, ~0.5 us per border pixel, so a launch
        // lasts as long as its longest segment (a 900-pixel text line: ~1.5 ms), alone on the GPU as well.
        static const bool trace_inline = [] { const char* e = getenv("OAR_GPU_CONTOURS_INLINE"); return e && atoi(e) != 0; }();
        hipStream_t ts = (gpu_contours && trace_inline) ? s : copy_stream_;
        if (ts != s) OAR_HIP(hipStreamWaitEvent(copy_stream_, mask_ready_[sb], 0));
        if (gpu_contours) {
            // a8 on the GPU: only the border chains cross PCIe
            const uint32_t tcap = (uint32_t)nb * ContourBufs::kSegsPerPage;
            pp::trace_contours(ts, traced, nb, H, W, trace_.rows.as<uint8_t>(), trace_.band_y.as<int32_t>(), trace_.n_bands.as<int32_t>(),
                               trace_.lists.as<uint32_t>(), tcap, trace_.scratch.as<uint32_t>(), trace_.packed.as<uint32_t>() + (size_t)b0 * trace_.packed_words_per_page,
                               (uint32_t)std::min<size_t>((size_t)nb * trace_.packed_words_per_page, 0xffffffffu),
                               trace_.ctrl.as<uint32_t>() + (size_t)sb * pp::kTraceCtlWords, trace_.ctrl_host.as<uint32_t>() + (size_t)sb * pp::kTraceCtlWords,
                               trace_.table_dev.as<pp::SegRec>(), trace_.table.as<pp::SegRec>() + (size_t)b0 * ContourBufs::kSegsPerPage, tcap);
        } else {
            // the mask crosses PCIe as a bit plane: 8x less traffic for the blit kernel that shares the GPU with the next network
            if (!fused_finish) pp::pack_mask_bits(copy_stream_, traced, mask_bits_.as<uint8_t>() + (size_t)b0 * hbits, nb, H, W);
            OAR_HIP(hipMemcpyAsync(mask_host_.as<uint8_t>() + (size_t)b0 * hbits, mask_bits_.as<uint8_t>() + (size_t)b0 * hbits, (size_t)nb * hbits,
                                   hipMemcpyDeviceToHost, copy_stream_));
        }
        OAR_HIP(hipEventRecord(sub_events_[sb], ts));
        tmark("det_enqueue");
    };

    // Software pipeline over the sub-batches: the GPU always has the NEXT sub-batch queued while the host works on this one
    //   enqueue(0), enqueue(1), host(0), enqueue(2), host(1), ... , host(last)
    //   host(sb): wait mask(sb) -> contours(sb) -> enqueue the box-score kernel of sb on the score stream (it shares the GPU
    //   with a later sub-batch's network and is slow there, ~0.7 ms, but nobody waits for it: its result is read one
    //   contour pass later) -> finish(sb-1): box scores back -> unclip / scale (host) -> on_ready.
    // Only the last, half-size sub-batch's contours + scores + unclip are exposed.
    std::vector<std::vector<Candidate>> cands(B);
    const uint8_t* mh = mask_host_.as<uint8_t>();
    const uint32_t maxc = cfg_.max_candidates;
    while ((int)score_slots_.size() < nsub) score_slots_.emplace_back(new ScoreSlot());
    while ((int)score_done_.size() < nsub) { hipEvent_t e; OAR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); score_done_.push_back(e); }
    static const bool enq_thread = [] { const char* e = getenv("OAR_DET_ENQ_THREAD"); return !e || atoi(e) != 0; }();
    // (with hipGraph replay the engine CAPTURES on the detector stream: the calling thread's crop launches / copies / synchronisations on that
    // stream would be recorded into -- or invalidate -- the helper's capture, so the helper thread stays off and enqueue / host stages interleave
    // on one thread as before)
    const bool helper_enqueues = enq_thread && nsub >= 3 && !eng_->graphs_enabled();
    auto finish = [&](int sb) {
        const int b0 = sb_off[sb], nb = sb_off[sb + 1] - b0;
        ScoreSlot& sl = *score_slots_[sb];
        if (sl.total) OAR_HIP(hipEventSynchronize(score_done_[sb]));
        tmark("box_scores_wait");
        const float* sc = sl.scores_host.as<float>();
        static const bool chunked = [] { const char* e = getenv("OAR_FINISH_CHUNKS"); return !e || atoi(e) != 0; }();   // 0: one task per page (A/B)
        if (cfg_.box_type == 1 || !chunked) {
            pool_->parallel_for(nb, [&](int k) {
                const PageRef& pg = pages[idx[b0 + k]];
                if (cfg_.box_type == 1) finish_polys(cands[b0 + k], sc + sl.base[k], H, W, pg.w, pg.h, box_thresh, unclip, out[idx[b0 + k]]);
                else finish_boxes(cands[b0 + k], sc + sl.base[k], H, W, pg.w, pg.h, box_thresh, unclip, out[idx[b0 + k]],
                                  sl.unclipped ? sl.unclip_host.as<pp::UnclipOut>() + sl.base[k] : nullptr);
            });
        } else {
            // quads: chunks of 8 candidates over the whole pool (a sub-batch is 4-8 pages: one task per page left most workers idle while the
            // GPU waited for the detection phase's host side); kept candidates are gathered in discovery order afterwards
            struct Chunk { int k; size_t c0, c1; };
            std::vector<Chunk> chunks;
            for (int k = 0; k < nb; ++k)
                for (size_t c0 = 0; c0 < cands[b0 + k].size(); c0 += 8) chunks.push_back({k, c0, std::min(cands[b0 + k].size(), c0 + 8)});
            finish_pts_.resize(sl.total * 8); finish_ok_.assign(sl.total, 0);
            const pp::UnclipOut* gu = sl.unclipped ? sl.unclip_host.as<pp::UnclipOut>() : nullptr;
            pool_->parallel_for((int)chunks.size(), [&](int ci) {
                const Chunk& ch = chunks[ci];
                const PageRef& pg = pages[idx[b0 + ch.k]];
                const float wscale = (float)pg.w / (float)W, hscale = (float)pg.h / (float)H;
                for (size_t c = ch.c0; c < ch.c1; ++c) {
                    const size_t g = sl.base[ch.k] + c;
                    finish_ok_[g] = finish_one_box(cands[b0 + ch.k][c], sc[g], box_thresh, unclip, wscale, hscale, (float)pg.w, (float)pg.h, gu ? gu + g : nullptr,
                                                   finish_pts_.data() + g * 8) ? 1 : 0;
                }
            });
            for (int k = 0; k < nb; ++k) {
                DetBoxes& o = out[idx[b0 + k]];
                o.pts.clear(); o.scores.clear();
                for (size_t c = 0; c < cands[b0 + k].size(); ++c) {
                    const size_t g = sl.base[k] + c;
                    if (!finish_ok_[g]) continue;
                    o.pts.insert(o.pts.end(), finish_pts_.data() + g * 8, finish_pts_.data() + g * 8 + 8);
                    o.scores.push_back(sc[g]);
                }
            }
        }
        tmark("host_unclip");
        if (on_ready) { on_ready(idx[b0], nb); tmark("crop_plan+warp"); }
    };
    auto host_stage = [&](int sb) {
        const int b0 = sb_off[sb], nb = sb_off[sb + 1] - b0;
        // finish(sb - 1) -- scores back, unclip, crop planning of the previous sub-batch -- goes FIRST when the GPU is still working on this
        // sub-batch's network.  With the helper thread enqueueing (below) the host stages never hold the GPU up, so the order only matters
        // after the LAST sub-batch, where everything this thread still has to do is exposed between detector and recognizer (1.0-1.4 ms of
        // idle queue per step in the rocprofv3 kernel trace): whatever can be done before that network ends should be.  On by default with the
        // helper thread (+0.9 % over four A/B pairs, profiles/r4/host_finish_ab.txt); without it -- the calling thread then also has to
        // enqueue -- it measured +3 / +3 / -5 / -5 % and stays off.  OAR_DET_FINISH_EARLY=0|1 overrides.
        static const int finish_early_env = [] { const char* e = getenv("OAR_DET_FINISH_EARLY"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        const bool finish_early = finish_early_env >= 0 ? finish_early_env == 1 : helper_enqueues;
        bool finished_prev = false;
        if (sb > 0 && finish_early) {
            const hipError_t q = hipEventQuery(sub_events_[sb]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); finish(sb - 1); finished_prev = true; }   // ("not ready" must not stay behind as the thread's last error)
            else if (q != hipSuccess) OAR_HIP(q);
        }
        OAR_HIP(hipEventSynchronize(sub_events_[sb]));
        tmark("det_gpu_wait");
        // BoxType::Poly scores the approximated polygon with box_score_fast whatever score_mode says (db_bitmap.rs:49)
        const bool poly = cfg_.box_type == 1;
        const int slow = poly ? kCandPoly : cfg_.score_mode == 1 ? kCandSlow : kCandFast;
        if (gpu_contours) {
            const uint8_t* traced_dev = (cfg_.use_dilation ? mask_dil_.as<uint8_t>() : mask_dev_.as<uint8_t>()) + (size_t)b0 * hw;
            auto fetch_mask = [&](int k) -> const uint8_t* {   // rare: a band the kernel could not hold in LDS
                uint8_t* dst = mask_host_.as<uint8_t>() + (size_t)(b0 + k) * hw;
                OAR_HIP(hipMemcpyAsync(dst, traced_dev + (size_t)k * hw, hw, hipMemcpyDeviceToHost, score_stream_));
                OAR_HIP(hipStreamSynchronize(score_stream_));
                return dst;
            };
            subbatch_candidates_traced(*pool_, trace_.ctrl_host.as<uint32_t>() + (size_t)sb * pp::kTraceCtlWords,
                                       trace_.table.as<pp::SegRec>() + (size_t)b0 * ContourBufs::kSegsPerPage, (uint32_t)nb * ContourBufs::kSegsPerPage,
                                       trace_.packed.as<uint32_t>() + (size_t)b0 * trace_.packed_words_per_page, H, W, nb, maxc, fetch_mask, &cands[b0], slow);
        } else {
            subbatch_candidates(*pool_, mh + (size_t)b0 * hbits, hbits, H, W, nb, maxc, &cands[b0], slow);
        }
        tmark("host_contours");
        ScoreSlot& sl = *score_slots_[sb];
        sl.base.assign(nb + 1, 0);
        size_t total = 0;
        for (int k = 0; k < nb; ++k) { sl.base[k] = total; total += cands[b0 + k].size(); }
        sl.base[nb] = total; sl.total = total;
        sl.unclipped = false;
        if (total && slow) {   // ScoreMode::Slow: the contour is the polygon (db_score.rs:139-181); BoxType::Poly: its approximation
            size_t npts = 0;
            for (int k = 0; k < nb; ++k) for (auto& cd : cands[b0 + k]) npts += cd.contour.size();
            sl.poly_pts_host.reserve(npts * 8 + 8); sl.poly_desc_host.reserve(total * sizeof(pp::PolyDesc)); sl.scores_host.reserve(total * sizeof(float));
            sl.poly_pts_dev.reserve(npts * 8 + 8); sl.poly_desc_dev.reserve(total * sizeof(pp::PolyDesc)); sl.scores_dev.reserve(total * sizeof(float));
            float* pp_ = sl.poly_pts_host.as<float>();
            pp::PolyDesc* pdesc = sl.poly_desc_host.as<pp::PolyDesc>();
            size_t at = 0, q = 0;
            for (int k = 0; k < nb; ++k)
                for (auto& cd : cands[b0 + k]) {
                    pdesc[q++] = pp::PolyDesc{(int32_t)at, (int32_t)cd.contour.size(), b0 + k, 0};
                    for (auto& p : cd.contour) { pp_[at * 2] = p.x; pp_[at * 2 + 1] = p.y; ++at; }
                    if (!poly) std::vector<host::Pt>().swap(cd.contour);   // the polygon itself is unclipped later
                }
            OAR_HIP(hipMemcpyAsync(sl.poly_pts_dev.p, pp_, npts * 8, hipMemcpyHostToDevice, score_stream_));
            OAR_HIP(hipMemcpyAsync(sl.poly_desc_dev.p, pdesc, total * sizeof(pp::PolyDesc), hipMemcpyHostToDevice, score_stream_));
            pp::poly_scores(score_stream_, probs, H, W, sl.poly_pts_dev.as<float>(), sl.poly_desc_dev.as<pp::PolyDesc>(), (int)total, sl.scores_dev.as<float>());
            OAR_HIP(hipMemcpyAsync(sl.scores_host.p, sl.scores_dev.p, total * sizeof(float), hipMemcpyDeviceToHost, score_stream_));
            OAR_HIP(hipEventRecord(score_done_[sb], score_stream_));
        } else if (total) {
            sl.boxes_host.reserve(total * sizeof(pp::ScoreBox)); sl.scores_host.reserve(total * sizeof(float));
            sl.boxes_dev.reserve(total * sizeof(pp::ScoreBox)); sl.scores_dev.reserve(total * sizeof(float));
            pp::ScoreBox* sbx = sl.boxes_host.as<pp::ScoreBox>();
            for (int k = 0; k < nb; ++k)
                for (size_t i = 0; i < cands[b0 + k].size(); ++i) {
                    pp::ScoreBox& x = sbx[sl.base[k] + i];
                    std::memcpy(x.pts, cands[b0 + k][i].pts, sizeof x.pts);
                    x.image = b0 + k; x.pad = 0;
                }
            OAR_HIP(hipMemcpyAsync(sl.boxes_dev.p, sbx, total * sizeof(pp::ScoreBox), hipMemcpyHostToDevice, score_stream_));
            pp::box_scores(score_stream_, probs, H, W, sl.boxes_dev.as<pp::ScoreBox>(), (int)total, sl.scores_dev.as<float>());
            OAR_HIP(hipMemcpyAsync(sl.scores_host.p, sl.scores_dev.p, total * sizeof(float), hipMemcpyDeviceToHost, score_stream_));
            // a11: the mini boxes are on the device already -- unclip them there, in the same round trip.  Goes with the GPU border
            // follower (oar_det_cfg.gpu_contours: the deployment whose host cores are scarce); on a host with idle cores the pool
            // does it faster than the extra 0.7 MB read-back (bench -1.5 %).  OAR_GPU_UNCLIP=0|1 overrides.
            static const int gpu_unclip_env = [] { const char* e = getenv("OAR_GPU_UNCLIP"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
            const bool gpu_unclip = gpu_unclip_env >= 0 ? gpu_unclip_env == 1 : gpu_contours;
            sl.unclipped = gpu_unclip;
            if (gpu_unclip) {
                sl.unclip_dev.reserve(total * sizeof(pp::UnclipOut)); sl.unclip_host.reserve(total * sizeof(pp::UnclipOut));
                pp::unclip_quads(score_stream_, sl.boxes_dev.as<pp::ScoreBox>(), (int)total, unclip, sl.unclip_dev.as<pp::UnclipOut>());
                OAR_HIP(hipMemcpyAsync(sl.unclip_host.p, sl.unclip_dev.p, total * sizeof(pp::UnclipOut), hipMemcpyDeviceToHost, score_stream_));
            }
            OAR_HIP(hipEventRecord(score_done_[sb], score_stream_));
        }
        tmark("box_scores_enqueue");
        if (sb > 0 && !finished_prev) finish(sb - 1);   // its scores were enqueued one contour pass ago
    };
    // `depth` sub-batches are queued ahead of the one the host works on (OAR_DET_DEPTH, default 1; every buffer a sub-batch's GPU
    // work touches is either per page or used in stream order, so any depth is safe)
    static const int depth = [] { const char* e = getenv("OAR_DET_DEPTH"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 8 ? 8 : v; }();
    // OAR_DET_ENQ_THREAD (default 1): with three or more sub-batches their GPU work is enqueued by a helper thread, all of it as soon as the
    // uploads allow (nothing a sub-batch's launches touch depends on the host's results for an earlier one), and this thread only does the
    // host stages.  The ~100 launches of a sub-batch cost the launching thread 0.13 ms -- 0.5 ms for the first two, which wait for their
    // pages -- and OAR_TIMING=2 showed the detection phase bounded by this thread (det_gpu_wait = 0 from the third sub-batch on), not by the GPU.
    if (helper_enqueues) {
        struct Enqueuer {   // declared last: joined before anything its thread refers to goes out of scope
            std::thread th;
            std::atomic<int> done{0};
            std::atomic<bool> failed{false};
            std::exception_ptr err;
            ~Enqueuer() { if (th.joinable()) th.join(); }
        } enq;
        const int device = eng_->device();
        enq.th = std::thread([&, device] {
            try {
                OAR_HIP(hipSetDevice(device));
                for (int q = 0; q < nsub; ++q) { enqueue(q); enq.done.store(q + 1, std::memory_order_release); }
            } catch (...) {
                enq.err = std::current_exception();
                enq.failed.store(true, std::memory_order_release);
            }
        });
        for (int sb = 0; sb < nsub; ++sb) {
            wait_for_thread([&] { return enq.done.load(std::memory_order_acquire) > sb || enq.failed.load(std::memory_order_acquire); });   // (an event must be recorded before it is waited for)
            if (enq.failed.load(std::memory_order_acquire)) { enq.th.join(); std::rethrow_exception(enq.err); }
            host_stage(sb);
        }
        finish(nsub - 1);
        enq.th.join();
        if (enq.failed.load(std::memory_order_acquire)) std::rethrow_exception(enq.err);
    } else {
        int queued = 0;
        for (; queued < std::min(depth, nsub); ++queued) enqueue(queued);
        for (int sb = 0; sb < nsub; ++sb) {
            if (queued < nsub) enqueue(queued++);
            host_stage(sb);
        }
        finish(nsub - 1);
    }
    if (Profiler::get().enabled) Profiler::get().flush();
}

// One device mask through pp::trace_contours (the same kernels the detector uses), flagged bands followed on the host.
std::vector<host::Contour> Detector::trace_device_mask(const uint8_t* d_mask, int H, int W, uint32_t max_contours, bool gpu_contours) {
    const size_t hw = (size_t)H * W;
    std::vector<uint8_t> mask;
    auto fetch = [&]() -> const uint8_t* {
        if (mask.empty()) { mask.resize(hw); OAR_HIP(hipMemcpy(mask.data(), d_mask, hw, hipMemcpyDeviceToHost)); }
        return mask.data();
    };
    if (!gpu_contours) return host::find_contours(fetch(), W, H, max_contours);
    ContourBufs tb;
    tb.reserve(1, 1, H, W, 1);
    pp::trace_contours(nullptr, d_mask, 1, H, W, tb.rows.as<uint8_t>(), tb.band_y.as<int32_t>(), tb.n_bands.as<int32_t>(), tb.lists.as<uint32_t>(),
                       ContourBufs::kSegsPerPage, tb.scratch.as<uint32_t>(), tb.packed.as<uint32_t>(), (uint32_t)std::min<size_t>(tb.packed_words_per_page, 0xffffffffu),
                       tb.ctrl.as<uint32_t>(), tb.ctrl_host.as<uint32_t>(), tb.table_dev.as<pp::SegRec>(), tb.table.as<pp::SegRec>(), ContourBufs::kSegsPerPage);
    OAR_HIP(hipStreamSynchronize(nullptr));
    const uint32_t* ctrl = tb.ctrl_host.as<uint32_t>();
    const uint32_t n_rec = ctrl[pp::kTraceCtlSegments];
    if (ctrl[pp::kTraceCtlOverflow] || n_rec > ContourBufs::kSegsPerPage) return host::find_contours(fetch(), W, H, max_contours);
    std::vector<std::pair<uint32_t, host::Contour>> items;
    for (uint32_t i = 0; i < n_rec; ++i) {
        const pp::SegRec& r = tb.table.as<pp::SegRec>()[i];
        OAR_CHECK(r.page == 0 && r.y0 >= 0 && r.y0 < r.y1 && r.y1 <= H && r.x0 >= 0 && r.x0 < r.x1 && r.x1 <= W, OAR_INTERNAL, "contour tracer: corrupt segment record");
        if (r.flags) trace_record_on_host(r, fetch(), H, W, max_contours, items);
        else parse_traced_record(r, tb.packed.as<uint32_t>(), items);
    }
    std::vector<host::Contour> out;
    merge_traced_page(items, max_contours, out);
    return out;
}

void Detector::postprocess_host(const float* pred, int H, int W, uint32_t src_w, uint32_t src_h, float thresh, float box_thresh,
                                float unclip, uint32_t max_candidates, DetBoxes& out, int score_mode, int use_dilation, int box_type) {
    // Stand-alone DB post-processing on a host probability map (parity hook for a7..a12): same kernels, tiny batch.
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(OAR_DEVICE, "no HIP device visible: libOarMi355x has no CPU fallback");
    const size_t hw = (size_t)H * W;
    DevBuf dpred, dmask, dmask2, dboxes, dscores, dpts;
    dpred.reserve(hw * 4); dmask.reserve(hw);
    OAR_HIP(hipMemcpy(dpred.p, pred, hw * 4, hipMemcpyHostToDevice));
    pp::threshold(nullptr, dpred.as<float>(), dmask.as<uint8_t>(), (int64_t)hw, thresh);
    const uint8_t* traced = dmask.as<uint8_t>();
    if (use_dilation) { dmask2.reserve(hw); pp::dilate3x3(nullptr, traced, dmask2.as<uint8_t>(), 1, H, W); traced = dmask2.as<uint8_t>(); }
    std::vector<Candidate> cands;
    const int kind = box_type == 1 ? kCandPoly : score_mode == 1 ? kCandSlow : kCandFast;
    {
        std::vector<host::Contour> cs = Detector::trace_device_mask(traced, H, W, max_candidates ? max_candidates : 1000);
        for (auto& c : cs) {
            Candidate cd;
            if (contour_candidate(c, cd, kind)) cands.push_back(std::move(cd));
        }
    }
    std::vector<float> scores(cands.size(), 0.f);
    if (!cands.empty() && kind != kCandFast) {
        std::vector<float> pts;
        std::vector<pp::PolyDesc> pd;
        for (auto& c : cands) {
            pd.push_back(pp::PolyDesc{(int32_t)(pts.size() / 2), (int32_t)c.contour.size(), 0, 0});
            for (auto& p : c.contour) { pts.push_back(p.x); pts.push_back(p.y); }
        }
        dpts.reserve(pts.size() * 4 + 8); dboxes.reserve(pd.size() * sizeof(pp::PolyDesc)); dscores.reserve(pd.size() * 4);
        OAR_HIP(hipMemcpy(dpts.p, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dboxes.p, pd.data(), pd.size() * sizeof(pp::PolyDesc), hipMemcpyHostToDevice));
        pp::poly_scores(nullptr, dpred.as<float>(), H, W, dpts.as<float>(), dboxes.as<pp::PolyDesc>(), (int)pd.size(), dscores.as<float>());
        OAR_HIP(hipMemcpy(scores.data(), dscores.p, pd.size() * 4, hipMemcpyDeviceToHost));
    } else if (!cands.empty()) {
        std::vector<pp::ScoreBox> sb(cands.size());
        for (size_t i = 0; i < cands.size(); ++i) { std::memcpy(sb[i].pts, cands[i].pts, sizeof sb[i].pts); sb[i].image = 0; sb[i].pad = 0; }
        dboxes.reserve(sb.size() * sizeof(pp::ScoreBox)); dscores.reserve(sb.size() * 4);
        OAR_HIP(hipMemcpy(dboxes.p, sb.data(), sb.size() * sizeof(pp::ScoreBox), hipMemcpyHostToDevice));
        pp::box_scores(nullptr, dpred.as<float>(), H, W, dboxes.as<pp::ScoreBox>(), (int)sb.size(), dscores.as<float>());
        OAR_HIP(hipMemcpy(scores.data(), dscores.p, sb.size() * 4, hipMemcpyDeviceToHost));
    }
    if (box_type == 1) finish_polys(cands, scores.data(), H, W, src_w, src_h, box_thresh, unclip, out);
    else finish_boxes(cands, scores.data(), H, W, src_w, src_h, box_thresh, unclip, out);
}

// ================================================================================================= recognizer
Recognizer::Recognizer(const uint8_t* onnx, size_t len, const oar_rec_cfg& cfg) : cfg_(cfg) {
    if (cfg_.rec_image_shape[0] == 0) { cfg_.rec_image_shape[0] = 3; cfg_.rec_image_shape[1] = 48; cfg_.rec_image_shape[2] = 320; }
    if (cfg_.max_img_w == 0) cfg_.max_img_w = 3200;
    eng_.reset(new Engine(onnx, len, cfg_.device_id));
    // OAR_REC_LANES (default 1): with 2 lanes bench.py gains 4-5 % (1447-1500 vs 1393-1421 images/s), but the kernels of
    // the two streams then share the GPU and their individual durations (the roofline accounting, the rocprof averages)
    // stop being comparable with the isolated numbers -- opt-in until the profiler separates lanes
    const int n_lanes = [] { const char* e = getenv("OAR_REC_LANES"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 4 ? 4 : v; }();   // read per handle: bench.py times both
    for (int i = 1; i < n_lanes; ++i) {
        lanes_.emplace_back(new Engine(onnx, len, cfg_.device_id));
        lane_in_.emplace_back(new DevBuf());
        hipEvent_t ev;
        OAR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        lane_done_.push_back(ev);
    }
}

const float* Recognizer::pack(const std::vector<Crop>& crops, int& Wt, bool nchw, size_t desc_slot, size_t stage_slot, int lane) {
    // desc_slot / stage_slot: element / byte offsets into the pinned + device staging areas, so that several batches
    // can be enqueued without the host overwriting a descriptor block an earlier async copy has not consumed yet.
    hipStream_t s = lane_engine(lane).stream();
    DevBuf& in_buf = lane == 0 ? input_f32_ : *lane_in_[lane - 1];
    const int n = (int)crops.size();
    const int img_h = (int)cfg_.rec_image_shape[1], img_w = (int)cfg_.rec_image_shape[2];
    std::vector<uint32_t> ws(n), hs(n);
    size_t stage = 0;
    for (int i = 0; i < n; ++i) {
        OAR_CHECK(crops[i].w > 0 && crops[i].h > 0 && (crops[i].host || crops[i].dev), OAR_INVALID_INPUT, "recognizer: empty crop");
        ws[i] = crops[i].w; hs[i] = crops[i].h;
        if (!crops[i].dev) stage += ((size_t)crops[i].w * crops[i].h * 3 + 63) & ~(size_t)63;
    }
    std::vector<int32_t> rws;
    Wt = host::rec_tensor_width(ws, hs, img_h, img_w, (int)cfg_.max_img_w, rws);
    size_t need_in = (size_t)n * 3 * img_h * Wt * sizeof(float);
    size_t need_desc = (desc_slot + n) * sizeof(pp::CropDesc), need_stage = stage_slot + stage;
    if (need_stage > crops_dev_.cap || need_in > in_buf.cap || need_desc > descs_dev_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        OAR_CHECK(desc_slot == 0 && stage_slot == 0, OAR_INTERNAL, "recognizer staging buffers must be pre-sized for multi-batch runs");
        crops_dev_.reserve(need_stage); in_buf.reserve(need_in); descs_dev_.reserve(need_desc);
    }
    if (need_desc > descs_host_.cap || need_stage > stage_host_.cap) {
        OAR_CHECK(desc_slot == 0 && stage_slot == 0, OAR_INTERNAL, "recognizer pinned buffers must be pre-sized for multi-batch runs");
        descs_host_.reserve(need_desc); stage_host_.reserve(need_stage);
    }
    pp::CropDesc* dh = descs_host_.as<pp::CropDesc>() + desc_slot;
    pp::CropDesc* dd = descs_dev_.as<pp::CropDesc>() + desc_slot;
    size_t off = stage_slot;
    for (int i = 0; i < n; ++i) {
        const uint8_t* d = crops[i].dev;
        if (!d) {
            size_t bytes = (size_t)crops[i].w * crops[i].h * 3;
            std::memcpy(stage_host_.as<uint8_t>() + off, crops[i].host, bytes);
            d = crops_dev_.as<uint8_t>() + off;
            off += (bytes + 63) & ~(size_t)63;
        }
        dh[i].src = d; dh[i].w = (int)crops[i].w; dh[i].h = (int)crops[i].h; dh[i].rw = rws[i]; dh[i].flip = crops[i].flip ? 1 : 0;
    }
    if (stage) OAR_HIP(hipMemcpyAsync(crops_dev_.as<uint8_t>() + stage_slot, stage_host_.as<uint8_t>() + stage_slot, stage, hipMemcpyHostToDevice, s));
    OAR_HIP(hipMemcpyAsync(dd, dh, (size_t)n * sizeof(pp::CropDesc), hipMemcpyHostToDevice, s));
    pp::rec_pack(s, dd, n, img_h, Wt, in_buf.as<float>(), nchw ? 1 : 0);
    return in_buf.as<float>();
}

static_assert(sizeof(pp::ResizedImg) == sizeof(k::StemImg) && sizeof(pp::ResizedImg) == 16, "one table for pp::rec_resize_u8 and k::conv_smallcin_u8");

// Fused-stem packing: the crops are resized to their own width in u8 (into the lane's input buffer, a quarter of what the f32
// tensor would take); normalisation and zero padding to Wt happen inside the stem convolution (k::StemU8::dev).
const pp::ResizedImg* Recognizer::pack_u8(const std::vector<Crop>& crops, int& Wt, size_t desc_slot, size_t stage_slot, int lane) {
    hipStream_t s = lane_engine(lane).stream();
    DevBuf& in_buf = lane == 0 ? input_f32_ : *lane_in_[lane - 1];
    const int n = (int)crops.size();
    const int img_h = (int)cfg_.rec_image_shape[1], img_w = (int)cfg_.rec_image_shape[2];
    std::vector<uint32_t> ws(n), hs(n);
    size_t stage = 0;
    for (int i = 0; i < n; ++i) {
        OAR_CHECK(crops[i].w > 0 && crops[i].h > 0 && (crops[i].host || crops[i].dev), OAR_INVALID_INPUT, "recognizer: empty crop");
        ws[i] = crops[i].w; hs[i] = crops[i].h;
        if (!crops[i].dev) stage += ((size_t)crops[i].w * crops[i].h * 3 + 63) & ~(size_t)63;
    }
    std::vector<int32_t> rws;
    Wt = host::rec_tensor_width(ws, hs, img_h, img_w, (int)cfg_.max_img_w, rws);
    size_t need_in = 0;
    int max_rw = 0;
    for (int i = 0; i < n; ++i) { need_in += ((size_t)img_h * rws[i] * 3 + 63) & ~(size_t)63; max_rw = std::max(max_rw, (int)rws[i]); }
    const size_t need_desc = (desc_slot + n) * sizeof(pp::CropDesc), need_stage = stage_slot + stage, need_imgs = (desc_slot + n) * sizeof(pp::ResizedImg);
    OAR_CHECK(need_stage <= crops_dev_.cap && need_in <= in_buf.cap && need_desc <= descs_dev_.cap && need_imgs <= imgs_dev_.cap && need_desc <= descs_host_.cap &&
                  need_stage <= stage_host_.cap && need_imgs <= imgs_host_.cap, OAR_INTERNAL, "recognizer staging buffers must be pre-sized (run_batches)");
    pp::CropDesc* dh = descs_host_.as<pp::CropDesc>() + desc_slot;
    pp::CropDesc* dd = descs_dev_.as<pp::CropDesc>() + desc_slot;
    pp::ResizedImg* ih = imgs_host_.as<pp::ResizedImg>() + desc_slot;
    pp::ResizedImg* id = imgs_dev_.as<pp::ResizedImg>() + desc_slot;
    size_t off = stage_slot, roff = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t* d = crops[i].dev;
        if (!d) {
            const size_t bytes = (size_t)crops[i].w * crops[i].h * 3;
            std::memcpy(stage_host_.as<uint8_t>() + off, crops[i].host, bytes);
            d = crops_dev_.as<uint8_t>() + off;
            off += (bytes + 63) & ~(size_t)63;
        }
        dh[i].src = d; dh[i].w = (int)crops[i].w; dh[i].h = (int)crops[i].h; dh[i].rw = rws[i]; dh[i].flip = crops[i].flip ? 1 : 0;
        ih[i].ptr = in_buf.as<uint8_t>() + roff; ih[i].w = rws[i]; ih[i].pad = 0;
        roff += ((size_t)img_h * rws[i] * 3 + 63) & ~(size_t)63;
    }
    if (stage) OAR_HIP(hipMemcpyAsync(crops_dev_.as<uint8_t>() + stage_slot, stage_host_.as<uint8_t>() + stage_slot, stage, hipMemcpyHostToDevice, s));
    OAR_HIP(hipMemcpyAsync(dd, dh, (size_t)n * sizeof(pp::CropDesc), hipMemcpyHostToDevice, s));
    OAR_HIP(hipMemcpyAsync(id, ih, (size_t)n * sizeof(pp::ResizedImg), hipMemcpyHostToDevice, s));
    pp::rec_resize_u8(s, dd, id, n, img_h, max_rw);
    return id;
}

void Recognizer::pack_only(const std::vector<Crop>& crops, std::vector<float>& nchw, uint32_t& Wt_out) {
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    nchw.clear(); Wt_out = 0;
    if (crops.empty()) return;
    int Wt = 0;
    const float* d = pack(crops, Wt, true);
    size_t cnt = crops.size() * 3 * (size_t)cfg_.rec_image_shape[1] * Wt;
    nchw.resize(cnt);
    OAR_HIP(hipMemcpyAsync(nchw.data(), d, cnt * 4, hipMemcpyDeviceToHost, eng_->stream()));
    OAR_HIP(hipStreamSynchronize(eng_->stream()));
    Wt_out = (uint32_t)Wt;
}

void Recognizer::run(const std::vector<Crop>& crops, RecOut& out) {
    std::vector<std::vector<Crop>> b(1, crops);
    std::vector<RecOut> o;
    run_batches(b, o);
    out = std::move(o[0]);
}

void Recognizer::run_batches(const std::vector<std::vector<Crop>>& batches, std::vector<RecOut>& outs) {
    std::lock_guard<std::mutex> lk(mu_);
    outs.assign(batches.size(), RecOut());
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    const int img_h = (int)cfg_.rec_image_shape[1], img_w = (int)cfg_.rec_image_shape[2];
    // pre-size every staging area for the whole run so nothing is reallocated (or overwritten) mid-flight
    size_t total_crops = 0, total_stage = 0, max_in = 0;
    for (auto& b : batches) {
        std::vector<uint32_t> ws, hs;
        for (auto& c : b) {
            ws.push_back(c.w); hs.push_back(c.h);
            OAR_CHECK(c.w > 0 && c.h > 0 && (c.host || c.dev), OAR_INVALID_INPUT, "recognizer: empty crop");
            if (!c.dev) total_stage += ((size_t)c.w * c.h * 3 + 63) & ~(size_t)63;
        }
        if (b.empty()) continue;
        std::vector<int32_t> rws;
        int Wt = host::rec_tensor_width(ws, hs, img_h, img_w, (int)cfg_.max_img_w, rws);
        max_in = std::max(max_in, (size_t)b.size() * 3 * img_h * Wt * sizeof(float));
        total_crops += b.size();
    }
    if (total_crops == 0) return;
    if (total_stage > crops_dev_.cap || max_in > input_f32_.cap || total_crops * sizeof(pp::CropDesc) > descs_dev_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        crops_dev_.reserve(total_stage); input_f32_.reserve(max_in); descs_dev_.reserve(total_crops * sizeof(pp::CropDesc));
    }
    for (auto& lb : lane_in_)
        if (max_in > lb->cap) { for (auto& le : lanes_) OAR_HIP(hipStreamSynchronize(le->stream())); lb->reserve(max_in); }
    descs_host_.reserve(total_crops * sizeof(pp::CropDesc));
    stage_host_.reserve(total_stage);
    // the recognizer's stem fused with normalisation + padding (k::StemU8::dev) when the graph starts with an RGB convolution;
    // OAR_REC_FUSE_STEM=0: the f32 input tensor is materialised by pp::rec_pack as before
    const char* fuse_env = getenv("OAR_REC_FUSE_STEM");
    const bool fuse_stem = !(fuse_env && fuse_env[0] == '0') && eng_->stem_fusable();
    if (fuse_stem) {
        if (total_crops * sizeof(pp::ResizedImg) > imgs_dev_.cap) { OAR_HIP(hipStreamSynchronize(s)); for (auto& le : lanes_) OAR_HIP(hipStreamSynchronize(le->stream())); imgs_dev_.reserve(total_crops * sizeof(pp::ResizedImg)); }
        imgs_host_.reserve(total_crops * sizeof(pp::ResizedImg));
    }

    struct Pending { size_t row0, rows; };
    std::vector<Pending> pend(batches.size(), Pending{0, 0});
    size_t desc_slot = 0, stage_slot = 0, row_total = 0;
    // first pass: plans (shape inference only) to size the result buffers
    std::vector<int> Wts(batches.size(), 0);
    std::vector<char> fuse_tail(batches.size(), 0);
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        const auto& b = batches[bi];
        if (b.empty()) continue;
        std::vector<uint32_t> ws, hs;
        for (auto& c : b) { ws.push_back(c.w); hs.push_back(c.h); }
        std::vector<int32_t> rws;
        Wts[bi] = host::rec_tensor_width(ws, hs, img_h, img_w, (int)cfg_.max_img_w, rws);
        // the fused tail mirrors the workgroup-per-row softmax kernel (vocab > 1024); smaller vocabularies keep the
        // unfused path so both seams stay bit-identical
        const Plan& probe = eng_->plan_for({(int64_t)b.size(), 3, img_h, Wts[bi]}, true, false, nullptr, fuse_stem);
        OAR_CHECK(!probe.outputs.empty(), OAR_INTERNAL, "CRNN: no output returned from inference");
        fuse_tail[bi] = probe.outputs[0].dims.size() == 3 && probe.outputs[0].dims[2] > 1024;
        const Plan& plan = fuse_tail[bi] ? eng_->plan_for({(int64_t)b.size(), 3, img_h, Wts[bi]}, true, true, nullptr, fuse_stem) : probe;
        const PlanOutput& po = plan.outputs[0];
        OAR_CHECK(po.dims.size() == 3, OAR_SHAPE_MISMATCH, "CRNN: expected 3D output (batch, time, vocab)");  // crnn.rs:273-279
        OAR_CHECK(po.dims[0] == (int64_t)b.size(), OAR_SHAPE_MISMATCH, "CRNN: output batch differs from input batch");
        outs[bi].T = (uint32_t)po.dims[1]; outs[bi].V = (uint32_t)po.dims[2]; outs[bi].Wt = (uint32_t)Wts[bi];
        pend[bi].row0 = row_total; pend[bi].rows = (size_t)b.size() * po.dims[1];
        if (po.dims[2] == 0) pend[bi].rows = 0;
        row_total += pend[bi].rows;
    }
    if (row_total * 8 > idx_dev_.cap || row_total * 4 > prob_dev_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        idx_dev_.reserve(row_total * 8); prob_dev_.reserve(row_total * 4);
    }
    idx_host_.reserve(row_total * 8); prob_host_.reserve(row_total * 4);
    const int n_lanes = 1 + (int)lanes_.size();
    std::vector<char> lane_used(n_lanes, 0);
    int next_lane = 0;
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        const auto& b = batches[bi];
        if (b.empty()) continue;
        const int lane = next_lane;
        next_lane = (next_lane + 1) % n_lanes;
        lane_used[lane] = 1;
        Engine& eng = lane_engine(lane);
        hipStream_t sl = eng.stream();
        int Wt = 0;
        const float* in = nullptr;
        const pp::ResizedImg* imgs = nullptr;
        if (fuse_stem) imgs = pack_u8(b, Wt, desc_slot, stage_slot, lane);
        else in = pack(b, Wt, false, desc_slot, stage_slot, lane);
        desc_slot += b.size();
        for (auto& c : b) if (!c.dev) stage_slot += ((size_t)c.w * c.h * 3 + 63) & ~(size_t)63;
        k::StemU8 st{};
        if (fuse_stem) {   // tensor channel c = source channel 2 - c (BGR), v -> (v / 255 - 0.5) / 0.5 (crnn.rs:98-121)
            st.src[0] = 2; st.src[1] = 1; st.src[2] = 0;
            st.dev = reinterpret_cast<const k::StemImg*>(imgs);
        }
        const Plan& plan = fuse_stem ? eng.run_stem(st, {(int64_t)b.size(), 3, img_h, Wt}, fuse_tail[bi] != 0)
                                     : eng.run(in, {(int64_t)b.size(), 3, img_h, Wt}, true, fuse_tail[bi] != 0);
        if (pend[bi].rows == 0) continue;
        const PlanOutput& po = plan.outputs[0];
        if (plan.skipped_softmax && plan.ctc_part.kind != Loc::NONE)   // not even the logits hit HBM: merge the per-tile partials
            k::ctc_combine(sl, eng.out_ptr(plan.ctc_part), (int64_t)pend[bi].rows, plan.ctc_tiles, idx_dev_.as<int64_t>() + pend[bi].row0,
                           prob_dev_.as<float>() + pend[bi].row0);
        else if (plan.skipped_softmax)   // output[0] holds logits: softmax + argmax in one pass, probabilities never hit HBM
            k::softmax_argmax(sl, eng.out_ptr(po.loc), (int64_t)pend[bi].rows, plan.logits_valid > 0 ? plan.logits_valid : (int)po.dims[2], (int)po.dims[2],
                              idx_dev_.as<int64_t>() + pend[bi].row0, prob_dev_.as<float>() + pend[bi].row0);
        else
            pp::ctc_argmax(sl, eng.out_ptr(po.loc), (int64_t)pend[bi].rows, (int)po.dims[2], idx_dev_.as<int64_t>() + pend[bi].row0,
                           prob_dev_.as<float>() + pend[bi].row0);
    }
    for (int lane = 1; lane < n_lanes; ++lane) {   // the result copies below run on lane 0's stream after every lane is done
        if (!lane_used[lane]) continue;
        OAR_HIP(hipEventRecord(lane_done_[lane - 1], lanes_[lane - 1]->stream()));
        OAR_HIP(hipStreamWaitEvent(s, lane_done_[lane - 1], 0));
    }
    if (row_total) {
        OAR_HIP(hipMemcpyAsync(idx_host_.p, idx_dev_.p, row_total * 8, hipMemcpyDeviceToHost, s));
        OAR_HIP(hipMemcpyAsync(prob_host_.p, prob_dev_.p, row_total * 4, hipMemcpyDeviceToHost, s));
    }
    OAR_HIP(hipStreamSynchronize(s));
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        if (pend[bi].rows == 0) continue;
        outs[bi].idx.assign(idx_host_.as<int64_t>() + pend[bi].row0, idx_host_.as<int64_t>() + pend[bi].row0 + pend[bi].rows);
        outs[bi].prob.assign(prob_host_.as<float>() + pend[bi].row0, prob_host_.as<float>() + pend[bi].row0 + pend[bi].rows);
    }
    if (Profiler::get().enabled) Profiler::get().flush();
}

// ================================================================================================= classifier (a22)
Classifier::Classifier(const uint8_t* onnx, size_t len, const ClsCfg& cfg) : cfg_(cfg) {
    if (cfg_.input_h == 0 || cfg_.input_w == 0) { cfg_.input_h = 224; cfg_.input_w = 224; }
    if (cfg_.topk == 0) cfg_.topk = 1;
    if (cfg_.batch == 0) cfg_.batch = 64;
    eng_.reset(new Engine(onnx, len, cfg_.device_id));
}

const float* Classifier::pack(const std::vector<Image>& images, size_t i0, size_t n, bool nchw) {
    hipStream_t s = eng_->stream();
    const int ch = (int)cfg_.input_h, cw = (int)cfg_.input_w;
    size_t stage = 0;
    for (size_t i = 0; i < n; ++i) {
        const Image& im = images[i0 + i];
        OAR_CHECK(im.w > 0 && im.h > 0 && (im.host || im.dev), OAR_INVALID_INPUT, "classifier: empty image");
        if (!im.dev) stage += ((size_t)im.w * im.h * 3 + 255) & ~(size_t)255;
    }
    const size_t need_in = n * 3 * (size_t)ch * cw * sizeof(float), need_desc = n * sizeof(pp::ClsDesc);
    if (stage > stage_dev_.cap || need_in > input_f32_.cap || need_desc > descs_dev_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        stage_dev_.reserve(stage); input_f32_.reserve(need_in); descs_dev_.reserve(need_desc);
    } else {
        OAR_HIP(hipStreamSynchronize(s));   // the pinned staging / descriptor blocks are reused by every batch
    }
    stage_host_.reserve(stage); descs_host_.reserve(need_desc);
    pp::ClsDesc* dh = descs_host_.as<pp::ClsDesc>();
    size_t off = 0;
    for (size_t i = 0; i < n; ++i) {
        const Image& im = images[i0 + i];
        const uint8_t* d = im.dev;
        if (!d) {
            const size_t bytes = (size_t)im.w * im.h * 3;
            std::memcpy(stage_host_.as<uint8_t>() + off, im.host, bytes);
            d = stage_dev_.as<uint8_t>() + off;
            off += (bytes + 255) & ~(size_t)255;
        }
        pp::ClsDesc& c = dh[i];
        c.src = d; c.w = (int)im.w; c.h = (int)im.h; c.pad = 0;
        if (cfg_.resize_short == 0) {      // direct resize (pp_lcnet.rs:172-189)
            c.nw = cw; c.nh = ch; c.x1 = 0; c.y1 = 0;
        } else {                           // short edge -> resize_short, centre crop (pp_lcnet.rs:147-170)
            const float shortf = (float)std::min(im.w, im.h);
            const float scale = (float)cfg_.resize_short / shortf;
            const uint32_t nw = (uint32_t)std::max(std::round((float)im.w * scale), (float)cw);
            const uint32_t nh = (uint32_t)std::max(std::round((float)im.h * scale), (float)ch);
            c.nw = (int)nw; c.nh = (int)nh;
            c.x1 = (int)((nw > (uint32_t)cw ? nw - (uint32_t)cw : 0u) / 2);
            c.y1 = (int)((nh > (uint32_t)ch ? nh - (uint32_t)ch : 0u) / 2);
        }
    }
    if (stage) OAR_HIP(hipMemcpyAsync(stage_dev_.p, stage_host_.p, stage, hipMemcpyHostToDevice, s));
    OAR_HIP(hipMemcpyAsync(descs_dev_.p, descs_host_.p, need_desc, hipMemcpyHostToDevice, s));
    // ImageNet constants, RGB order (pp_lcnet.rs:48-50, 400-412): alpha = scale/std, beta = -mean/std in f32
    const float scale = 1.0f / 255.0f;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float alpha[3], beta[3];
    for (int c = 0; c < 3; ++c) { alpha[c] = scale / stdv[c]; beta[c] = -mean[c] / stdv[c]; }
    pp::cls_pack(s, descs_dev_.as<pp::ClsDesc>(), (int)n, ch, cw, alpha, beta, input_f32_.as<float>(), nchw ? 1 : 0);
    return input_f32_.as<float>();
}

void Classifier::pack_only(const std::vector<Image>& images, std::vector<float>& nchw) {
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    nchw.clear();
    if (images.empty()) return;
    const float* d = pack(images, 0, images.size(), true);
    nchw.resize(images.size() * 3 * (size_t)cfg_.input_h * cfg_.input_w);
    OAR_HIP(hipMemcpyAsync(nchw.data(), d, nchw.size() * 4, hipMemcpyDeviceToHost, eng_->stream()));
    OAR_HIP(hipStreamSynchronize(eng_->stream()));
}

void Classifier::run(const std::vector<Image>& images, ClsOut& out) {
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    out = ClsOut();
    out.topk = cfg_.topk;
    for (size_t i0 = 0; i0 < images.size(); i0 += cfg_.batch) {
        const size_t n = std::min<size_t>(cfg_.batch, images.size() - i0);
        const float* in = pack(images, i0, n, false);
        const Plan& plan = eng_->run(in, {(int64_t)n, 3, (int64_t)cfg_.input_h, (int64_t)cfg_.input_w}, true);
        OAR_CHECK(!plan.outputs.empty(), OAR_INTERNAL, "PP-LCNet: no output returned from inference");        // pp_lcnet.rs:226-231
        const PlanOutput& po = plan.outputs[0];
        OAR_CHECK(po.dims.size() == 2 && po.dims[0] == (int64_t)n, OAR_SHAPE_MISMATCH, "PP-LCNet: failed to convert output to 2D array");
        const int nc = (int)po.dims[1];
        out.n_classes = (uint32_t)nc;
        probs_host_.reserve(n * nc * sizeof(float));
        OAR_HIP(hipMemcpyAsync(probs_host_.p, eng_->out_ptr(po.loc), n * nc * sizeof(float), hipMemcpyDeviceToHost, s));
        OAR_HIP(hipStreamSynchronize(s));
        const float* pr = probs_host_.as<float>();
        const int k = std::min<int>((int)cfg_.topk, nc);
        for (size_t i = 0; i < n; ++i) {
            // Topk (utils/topk.rs:181-199): stable sort by score descending => the first index wins ties
            std::vector<int> order(nc);
            for (int c = 0; c < nc; ++c) order[c] = c;
            const float* row = pr + i * nc;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return row[a] > row[b]; });
            for (int t = 0; t < (int)cfg_.topk; ++t) {
                out.ids.push_back(t < k ? order[t] : -1);
                out.scores.push_back(t < k ? row[order[t]] : 0.0f);
            }
        }
    }
    if (Profiler::get().enabled) Profiler::get().flush();
}

// ================================================================================================= rectifier (a23)
Rectifier::Rectifier(const uint8_t* onnx, size_t len, const RectCfg& cfg) : cfg_(cfg) {
    eng_.reset(new Engine(onnx, len, cfg_.device_id));
}

void Rectifier::run_device(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst) {
    std::lock_guard<std::mutex> lk(mu_);
    run_device_locked(src, w, h, dst);
}

void Rectifier::run_device_locked(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst) {
    OAR_CHECK(w > 0 && h > 0 && src && dst, OAR_INVALID_INPUT, "rectifier: empty image");
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    const uint32_t th = cfg_.target_h, tw = cfg_.target_w;
    const bool resize = th > 0 && tw > 0 && (w != tw || h != th);          // uvdoc.rs:88-101
    const uint32_t iw = resize ? tw : w, ih = resize ? th : h;
    const size_t plane = (size_t)iw * ih;
    if (plane * 3 > resized_dev_.cap || plane * 12 > input_f32_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        resized_dev_.reserve(plane * 3); input_f32_.reserve(plane * 12);
    }
    const uint8_t* in_u8 = src;
    if (resize) { pp::resize_triangle(s, src, (int)w, (int)h, resized_dev_.as<uint8_t>(), (int)iw, (int)ih); in_u8 = resized_dev_.as<uint8_t>(); }
    // v / 255 in BGR plane order, no mean shift (uvdoc.rs:296-303)
    const int srcc[3] = {2, 1, 0};
    const float alpha[3] = {1.0f / 255.0f, 1.0f / 255.0f, 1.0f / 255.0f}, beta[3] = {-0.0f, -0.0f, -0.0f};
    pp::normalize(s, in_u8, input_f32_.as<float>(), 1, (int64_t)plane, srcc, alpha, beta, 1);
    const Plan& plan = eng_->run(input_f32_.as<float>(), {1, 3, (int64_t)ih, (int64_t)iw}, true);
    OAR_CHECK(!plan.outputs.empty(), OAR_INTERNAL, "UVDoc: no output returned from inference");
    const PlanOutput& po = plan.outputs[0];
    OAR_CHECK(po.dims.size() == 4 && po.dims[0] == 1 && po.dims[1] == 3, OAR_SHAPE_MISMATCH, "UVDoc: expected a [n,3,h,w] output");
    const uint32_t oh = (uint32_t)po.dims[2], ow = (uint32_t)po.dims[3];
    const size_t oplane = (size_t)oh * ow;
    const bool back = ow != w || oh != h;                                   // uvdoc.rs:188-203
    if (back && oplane * 3 > out_u8_.cap) { OAR_HIP(hipStreamSynchronize(s)); out_u8_.reserve(oplane * 3); }
    uint8_t* o8 = back ? out_u8_.as<uint8_t>() : dst;
    pp::bgr_planes_to_rgb(s, eng_->out_ptr(po.loc), (int64_t)oplane, 255.0f, o8);
    if (back) pp::resize_triangle(s, o8, (int)ow, (int)oh, dst, (int)w, (int)h);
}

// All pages of a call in sub-batches of `kSub`: per page only the two Triangle resizes and the byte conversion are separate launches;
// normalisation and the network run once per sub-batch ([n,3,th,tw]).  The reference rectifies page by page
// (src/oarocr/preprocess.rs:59-97 inside the serial loop of ocr.rs:544-548); UVDoc has no cross-image op, so the batched graph
// computes the same per-page values.  Pages whose size cannot be batched (no fixed target size) take run_device_locked.
void Rectifier::run_device_batch(const std::vector<Page>& pages) {
    std::lock_guard<std::mutex> lk(mu_);
    const uint32_t th = cfg_.target_h, tw = cfg_.target_w;
    if (th == 0 || tw == 0 || pages.size() < 2) {
        for (const Page& p : pages) run_device_locked(p.src, p.w, p.h, p.dst);
        return;
    }
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    constexpr size_t kSub = 8;
    const size_t plane = (size_t)tw * th;
    const size_t nsub = std::min(kSub, pages.size());
    if (plane * 3 * nsub > resized_dev_.cap || plane * 12 * nsub > input_f32_.cap) {
        OAR_HIP(hipStreamSynchronize(s));
        resized_dev_.reserve(plane * 3 * nsub); input_f32_.reserve(plane * 12 * nsub);
    }
    const int srcc[3] = {2, 1, 0};
    const float alpha[3] = {1.0f / 255.0f, 1.0f / 255.0f, 1.0f / 255.0f}, beta[3] = {-0.0f, -0.0f, -0.0f};
    for (size_t i0 = 0; i0 < pages.size(); i0 += kSub) {
        const size_t n = std::min(kSub, pages.size() - i0);
        for (size_t i = 0; i < n; ++i) {
            const Page& p = pages[i0 + i];
            OAR_CHECK(p.w > 0 && p.h > 0 && p.src && p.dst, OAR_INVALID_INPUT, "rectifier: empty image");
            uint8_t* r = resized_dev_.as<uint8_t>() + i * plane * 3;
            if (p.w != tw || p.h != th) pp::resize_triangle(s, p.src, (int)p.w, (int)p.h, r, (int)tw, (int)th);     // uvdoc.rs:88-101
            else OAR_HIP(hipMemcpyAsync(r, p.src, plane * 3, hipMemcpyDeviceToDevice, s));
        }
        pp::normalize(s, resized_dev_.as<uint8_t>(), input_f32_.as<float>(), (int64_t)n, (int64_t)plane, srcc, alpha, beta, 1);
        const Plan& plan = eng_->run(input_f32_.as<float>(), {(int64_t)n, 3, (int64_t)th, (int64_t)tw}, true);
        OAR_CHECK(!plan.outputs.empty(), OAR_INTERNAL, "UVDoc: no output returned from inference");
        const PlanOutput& po = plan.outputs[0];
        OAR_CHECK(po.dims.size() == 4 && po.dims[0] == (int64_t)n && po.dims[1] == 3, OAR_SHAPE_MISMATCH, "UVDoc: expected a [n,3,h,w] output");
        const uint32_t oh = (uint32_t)po.dims[2], ow = (uint32_t)po.dims[3];
        const size_t oplane = (size_t)oh * ow;
        if (oplane * 3 * n > out_u8_.cap) { OAR_HIP(hipStreamSynchronize(s)); out_u8_.reserve(oplane * 3 * nsub); }
        for (size_t i = 0; i < n; ++i) {
            const Page& p = pages[i0 + i];
            const bool back = ow != p.w || oh != p.h;                               // uvdoc.rs:188-203
            uint8_t* o8 = back ? out_u8_.as<uint8_t>() + i * oplane * 3 : p.dst;
            pp::bgr_planes_to_rgb(s, eng_->out_ptr(po.loc) + i * 3 * oplane, (int64_t)oplane, 255.0f, o8);
            if (back) pp::resize_triangle(s, o8, (int)ow, (int)oh, p.dst, (int)p.w, (int)p.h);
        }
    }
}

void Rectifier::run_host(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst) {
    // ONE lock across upload, run and read-back: io_dev_ is shared staging, a second caller (or its reserve()) must not
    // get in between
    const size_t bytes = (size_t)w * h * 3;
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    if (bytes * 2 > io_dev_.cap) { OAR_HIP(hipStreamSynchronize(eng_->stream())); io_dev_.reserve(bytes * 2); }
    OAR_HIP(hipMemcpyAsync(io_dev_.p, src, bytes, hipMemcpyHostToDevice, eng_->stream()));
    run_device_locked(io_dev_.as<uint8_t>(), w, h, io_dev_.as<uint8_t>() + bytes);
    OAR_HIP(hipMemcpyAsync(dst, io_dev_.as<uint8_t>() + bytes, bytes, hipMemcpyDeviceToHost, eng_->stream()));
    OAR_HIP(hipStreamSynchronize(eng_->stream()));
}

// ================================================================================================= OCR pipeline
Ocr::Ocr(const uint8_t* det, size_t det_len, const uint8_t* rec, size_t rec_len, const oar_ocr_cfg& cfg) : cfg_(cfg) {
    if (cfg_.image_batch_size == 0) cfg_.image_batch_size = 8;     // text_detection_adapter.rs:85-87
    // reference adapter: 64 (text_recognition_adapter.rs:117-127). This backend recommends 256: the tiny recognizer is
    // launch/latency-bound at 64 crops on 256 CUs; a drop-in adapter may report a larger recommended_batch_size.
    if (cfg_.region_batch_size == 0) cfg_.region_batch_size = 256;
    if (cfg_.max_pooled_crops == 0) cfg_.max_pooled_crops = 4096;  // src/oarocr/ocr.rs:603
    OAR_CHECK(cfg_.box_sort >= 0 && cfg_.box_sort <= 2, OAR_INVALID_INPUT, "box_sort must be 0 (by box type), 1 (sort_quad_boxes) or 2 (sort_poly_boxes)");
    OAR_CHECK(cfg_.image_batch_size <= 4096 && cfg_.region_batch_size <= 4096, OAR_INVALID_INPUT,
              "batch sizes must be in 1..=4096");                  // src/oarocr/ocr.rs:250-255,419-430
    det_.reset(new Detector(det, det_len, cfg_.det));
    rec_.reset(new Recognizer(rec, rec_len, cfg_.rec));
}

// DocumentPreprocessor::preprocess (src/oarocr/preprocess.rs:59-97) for every page: optional orientation class ->
// rotate (class 1 -> rotate270, 2 -> rotate180, 3 -> rotate90, :128-133), optional UVDoc rectification.  The corrected
// pages live in pre_pages_ (device); meta_ records what was done so that boxes can be mapped back.
void Ocr::preprocess_pages(const std::vector<PageRef>& pages, std::vector<PageRef>& cur) {
    const int n = (int)pages.size();
    hipStream_t s = det_->engine().stream();
    cur = pages;
    meta_.assign(n, PageMeta());
    // device copies of host pages (the rotate / rectify kernels read device memory)
    size_t up = 0;
    for (auto& p : pages) {
        OAR_CHECK(p.w > 0 && p.h > 0 && (p.host || p.dev), OAR_INVALID_INPUT, "OCR Pipeline: empty page");
        if (!p.dev) up += ((size_t)p.w * p.h * 3 + 255) & ~(size_t)255;
    }
    if (up > upload_pages_.cap) { OAR_HIP(hipStreamSynchronize(s)); upload_pages_.reserve(up); }
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (cur[i].dev) continue;
        uint8_t* d = upload_pages_.as<uint8_t>() + off;
        const size_t bytes = (size_t)pages[i].w * pages[i].h * 3;
        OAR_HIP(hipMemcpyAsync(d, pages[i].host, bytes, hipMemcpyHostToDevice, s));
        cur[i].dev = d; cur[i].host = nullptr;
        off += (bytes + 255) & ~(size_t)255;
    }
    OAR_HIP(hipStreamSynchronize(s));
    // room for one rotated and one rectified copy of every page
    size_t need = 0;
    for (auto& p : pages) need += 2 * (((size_t)p.w * p.h * 3 + 255) & ~(size_t)255);
    if (need > pre_pages_.cap) pre_pages_.reserve(need);
    size_t poff = 0;
    std::vector<int32_t> cls(n, -1);
    if (doc_cls_) {
        std::vector<Classifier::Image> imgs(n);
        for (int i = 0; i < n; ++i) { imgs[i].dev = cur[i].dev; imgs[i].w = cur[i].w; imgs[i].h = cur[i].h; }
        ClsOut co;
        doc_cls_->run(imgs, co);   // the pipeline only reads classifications[0] (preprocess.rs:155-159)
        for (int i = 0; i < n; ++i) cls[i] = co.ids[(size_t)i * co.topk];
    }
    for (int i = 0; i < n; ++i) {
        if (doc_cls_ && cls[i] >= 0) {
            const int quarter = cls[i] == 1 ? 3 : cls[i] == 2 ? 2 : cls[i] == 3 ? 1 : 0;
            if (quarter) {
                uint8_t* d = pre_pages_.as<uint8_t>() + poff;
                pp::rotate_rgb(s, cur[i].dev, (int)cur[i].w, (int)cur[i].h, quarter, d);
                poff += ((size_t)cur[i].w * cur[i].h * 3 + 255) & ~(size_t)255;
                cur[i].dev = d;
                if (quarter != 2) std::swap(cur[i].w, cur[i].h);
            }
            meta_[i].angle = (float)cls[i] * 90.0f;
            meta_[i].rotated_w = cur[i].w; meta_[i].rotated_h = cur[i].h;
        }
    }
    OAR_HIP(hipStreamSynchronize(s));
    if (rect_) {
        std::vector<Rectifier::Page> rp(n);
        for (int i = 0; i < n; ++i) {
            uint8_t* d = pre_pages_.as<uint8_t>() + poff;
            rp[i].src = cur[i].dev; rp[i].w = cur[i].w; rp[i].h = cur[i].h; rp[i].dst = d;
            poff += ((size_t)cur[i].w * cur[i].h * 3 + 255) & ~(size_t)255;
            cur[i].dev = d;
            meta_[i].rectified = true;
        }
        rect_->run_device_batch(rp);
        OAR_HIP(hipStreamSynchronize(rect_->engine().stream()));
    }
}

void Ocr::predict(const std::vector<PageRef>& pages, std::vector<std::vector<OcrRegion>>& out, std::vector<PageMeta>* meta_out) {
    std::lock_guard<std::mutex> lk(mu_);
    struct MetaCopy {   // the caller's copy is taken under the lock (another thread's predict() overwrites meta_)
        Ocr& o; std::vector<PageMeta>* dst;
        ~MetaCopy() { if (dst) *dst = o.meta_; }
    } meta_copy{*this, meta_out};
    OAR_CHECK(!pages.empty(), OAR_INVALID_INPUT, "OCR Pipeline: images must be a non-empty slice");  // ocr.rs:525-532
    OAR_HIP(hipSetDevice(det_->engine().device()));
    meta_.assign(pages.size(), PageMeta());
    if (!doc_cls_ && !rect_) { predict_core(pages, out); return; }
    std::vector<PageRef> cur;
    preprocess_pages(pages, cur);
    predict_core(cur, out);
    // boxes are mapped back only when the page was rotated and NOT rectified (preprocess.rs:84-89, ocr.rs:644-646,
    // BoundingBox::rotate_back_to_original geometry.rs:848-889)
    for (size_t i = 0; i < out.size(); ++i) {
        const PageMeta& m = meta_[i];
        if (m.rectified || m.angle < 0.0f) continue;
        const int a = (int)m.angle;
        auto back = [&](float* q, size_t npts) {
            for (size_t k = 0; k < npts; ++k) {
                const float x = q[2 * k], y = q[2 * k + 1];
                if (a == 90) { q[2 * k] = (float)m.rotated_h - y; q[2 * k + 1] = x; }
                else if (a == 180) { q[2 * k] = (float)m.rotated_w - x; q[2 * k + 1] = (float)m.rotated_h - y; }
                else if (a == 270) { q[2 * k] = y; q[2 * k + 1] = (float)m.rotated_w - x; }
            }
        };
        for (auto& r : out[i]) {
            back(r.pts, 4);
            if (!r.poly.empty()) back(r.poly.data(), r.poly.size() / 2);   // every point of a polygon (geometry.rs:848-889 maps self.points)
        }
    }
}

void Ocr::predict_core(const std::vector<PageRef>& pages, std::vector<std::vector<OcrRegion>>& out) {
    const int n = (int)pages.size();
    OAR_HIP(hipSetDevice(det_->engine().device()));
    hipStream_t s = det_->engine().stream();
    PhaseTimer timer;
    g_timer = &timer;
    struct TimerReset { ~TimerReset() { g_timer = nullptr; } } timer_reset;
    struct PoolItem { int img; int det_index; uint32_t w, h; float wh_ratio; size_t off; bool flip = false; };
    struct Slot { bool filled = false; OcrRegion r; };
    std::vector<std::vector<Slot>> per_image(n);
    std::vector<PoolItem> pool;
    size_t pool_bytes = 0;
    const float base_ratio = (float)cfg_.rec.rec_image_shape[2] > 0 ? (float)cfg_.rec.rec_image_shape[2] / (float)cfg_.rec.rec_image_shape[1] : 320.0f / 48.0f;

    auto flush = [&]() {
        if (pool.empty()) return;
        OAR_HIP(hipStreamSynchronize(s));  // crops complete
        std::vector<float> line_angle(pool.size(), -1.0f);
        if (line_cls_) {
            // classify_line_orientations (src/oarocr/ocr.rs:757-790): class 1 => the crop is rotated by 180 degrees
            std::vector<Classifier::Image> imgs(pool.size());
            for (size_t i = 0; i < pool.size(); ++i) { imgs[i].dev = crop_pool_.as<uint8_t>() + pool[i].off; imgs[i].w = pool[i].w; imgs[i].h = pool[i].h; }
            ClsOut co;
            line_cls_->run(imgs, co);
            // class 1 => the recognizer reads the crop as its rotate180 (CropDesc::flip): the resize walks the stored crop backwards,
            // tap for tap what resizing a rotated copy would read, so no rotated crop is ever written (round 2 launched one
            // rotate kernel per such crop: 526 launches per step on BASELINE config 5)
            for (size_t i = 0; i < pool.size(); ++i) {
                const int c = co.ids[i * co.topk];
                if (c < 0) continue;
                line_angle[i] = (float)c * 180.0f;
                per_image[pool[i].img][pool[i].det_index].r.line_angle = line_angle[i];
                pool[i].flip = c == 1;
            }
        }
        std::vector<PoolItem> sorted = pool;
        std::stable_sort(sorted.begin(), sorted.end(), [](const PoolItem& a, const PoolItem& b) { return a.wh_ratio < b.wh_ratio; });
        const size_t bs = cfg_.region_batch_size;
        std::vector<std::vector<Recognizer::Crop>> batches;
        std::vector<float> chunk_max;
        for (size_t c0 = 0; c0 < sorted.size(); c0 += bs) {
            size_t c1 = std::min(sorted.size(), c0 + bs);
            std::vector<Recognizer::Crop> crops;
            float cm = base_ratio;
            for (size_t i = c0; i < c1; ++i) {
                Recognizer::Crop c;
                c.dev = crop_pool_.as<uint8_t>() + sorted[i].off; c.w = sorted[i].w; c.h = sorted[i].h; c.flip = sorted[i].flip;
                crops.push_back(c);
                if (sorted[i].wh_ratio > cm) cm = sorted[i].wh_ratio;
            }
            batches.push_back(std::move(crops));
            chunk_max.push_back(cm);
        }
        std::vector<RecOut> ros;
        tmark("rec_sort_batching");
        rec_->run_batches(batches, ros);
        tmark("rec_all_batches");
        for (size_t bi = 0; bi < batches.size(); ++bi) {
            const RecOut& ro = ros[bi];
            const size_t c0 = bi * bs;
            for (size_t k = 0; k < batches[bi].size(); ++k) {
                Slot& sl = per_image[sorted[c0 + k].img][sorted[c0 + k].det_index];
                sl.filled = true;
                sl.r.T = ro.T; sl.r.max_wh_ratio = chunk_max[bi];
                if (!ro.idx.empty()) {
                    sl.r.idx.assign(ro.idx.begin() + k * ro.T, ro.idx.begin() + (k + 1) * ro.T);
                    sl.r.prob.assign(ro.prob.begin() + k * ro.T, ro.prob.begin() + (k + 1) * ro.T);
                }
            }
        }
        pool.clear();
        pool_bytes = 0;
    };

    int flush_count = 0;
    bool planning_failed = false;   // an error from the crop-planning callback (not from the detector): no per-image retry for those
    auto run_chunk = [&](int start, int end) {
        std::vector<PageRef> chunk(pages.begin() + start, pages.begin() + end);
        std::vector<DetBoxes> boxes;
        std::vector<const uint8_t*> dev_pages;
        size_t desc_used = 0;   // warp descriptors of this chunk already handed to the GPU (each launch gets its own slots)
        // Called by the detector as soon as the boxes of pages [first, first + count) of the chunk are final (the GPU is
        // still on later pages): sort + plan the crops on the host, one warp launch for those pages.
        auto plan_pages = [&](int first, int count) {
            struct Planned { int img; int det_index; host::CropPlan plan; };
            std::vector<Planned> planned;
            for (int li = first; li < first + count; ++li) {
                const int img = start + li;
                const bool poly = cfg_.det.box_type == 1;   // BoxType::Poly (seal text): variable-size polygons
                // sort_detection_boxes (ocr.rs:699-716) keys on text_type == "seal", the detector's box type is a separate setting:
                // cfg_.box_sort 1 / 2 force sort_quad_boxes / sort_poly_boxes, 0 follows the box type (what text_type sets together)
                const bool poly_sort = cfg_.box_sort == 2 || (cfg_.box_sort == 0 && poly);
                std::vector<uint32_t> poly_off;
                std::vector<int> order;
                if (poly) {
                    poly_off.assign(boxes[li].counts.size() + 1, 0);
                    for (size_t b = 0; b < boxes[li].counts.size(); ++b) poly_off[b + 1] = poly_off[b] + boxes[li].counts[b];
                } else if (poly_sort) {
                    const size_t nb = boxes[li].pts.size() / 8;
                    poly_off.resize(nb + 1);
                    for (size_t b = 0; b <= nb; ++b) poly_off[b] = (uint32_t)(4 * b);
                }
                if (poly_sort) {
                    order = host::sort_poly_boxes(boxes[li].pts, poly_off);
                } else if (!poly) {
                    order = host::sort_quad_boxes(boxes[li].pts);
                } else {
                    // sort_quad_boxes on polygons: the reference's routine only reads each box's y_min / x_min (sorting.rs:35-84); feed it the
                    // polygons' axis-aligned corner boxes
                    std::vector<float> aabb(boxes[li].counts.size() * 8);
                    for (size_t b = 0; b < boxes[li].counts.size(); ++b) {
                        float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
                        for (uint32_t q = poly_off[b]; q < poly_off[b + 1]; ++q) {
                            const float x = boxes[li].pts[2 * q], y = boxes[li].pts[2 * q + 1];
                            x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
                        }
                        const float q8[8] = {x0, y0, x1, y0, x1, y1, x0, y1};
                        std::memcpy(aabb.data() + b * 8, q8, sizeof q8);
                    }
                    order = host::sort_quad_boxes(aabb);
                }
                per_image[img].resize(order.size());
                for (size_t k = 0; k < order.size(); ++k) {
                    Slot& sl = per_image[img][k];
                    sl.r.det_score = boxes[li].scores[order[k]];
                    host::CropPlan pl;
                    if (poly) {
                        const float* pp_ = boxes[li].pts.data() + (size_t)poly_off[order[k]] * 2;
                        const uint32_t np_ = boxes[li].counts[order[k]];
                        sl.r.poly.assign(pp_, pp_ + (size_t)np_ * 2);
                        std::memset(sl.r.pts, 0, sizeof sl.r.pts);
                        if (np_ == 4) {   // TextCroppingProcessor::crop_single: exactly four points take the rotated crop (processors.rs:96-102)
                            std::memcpy(sl.r.pts, pp_, sizeof sl.r.pts);
                            pl = host::plan_crop((int)pages[img].w, (int)pages[img].h, sl.r.pts);
                        } else {
                            pl = host::plan_bbox_crop((int)pages[img].w, (int)pages[img].h, pp_, (int)np_);
                        }
                    } else {
                        std::memcpy(sl.r.pts, boxes[li].pts.data() + (size_t)order[k] * 8, sizeof sl.r.pts);
                        pl = host::plan_crop((int)pages[img].w, (int)pages[img].h, sl.r.pts);
                    }
                    if (pl.mode == 0) continue;  // crop failure => region dropped (ocr.rs:736-738)
                    sl.r.crop_w = (uint32_t)pl.out_w(); sl.r.crop_h = (uint32_t)pl.out_h();
                    planned.push_back({img, (int)k, pl});
                }
            }
            size_t pi = 0;
            while (pi < planned.size()) {
                // respect the pool cap exactly like the reference's per-crop flush check
                size_t room = cfg_.max_pooled_crops - pool.size();
                size_t take = std::min(room, planned.size() - pi);
                size_t add_bytes = 0;
                int max_px = 0;
                for (size_t q = pi; q < pi + take; ++q) {
                    const host::CropPlan& pl = planned[q].plan;
                    add_bytes += ((size_t)pl.out_w() * pl.out_h() * 3 + 63) & ~(size_t)63;
                    max_px = std::max(max_px, pl.out_w() * pl.out_h());
                }
                if (pool_bytes + add_bytes > crop_pool_.cap) {
                    // grow while preserving existing crops
                    OAR_HIP(hipStreamSynchronize(s));
                    DevBuf bigger;
                    bigger.reserve((pool_bytes + add_bytes) * 2);
                    if (pool_bytes) OAR_HIP(hipMemcpy(bigger.p, crop_pool_.p, pool_bytes, hipMemcpyDeviceToDevice));
                    std::swap(bigger.p, crop_pool_.p); std::swap(bigger.cap, crop_pool_.cap);
                }
                const size_t need_desc = (desc_used + take) * sizeof(pp::WarpDesc);
                if (need_desc > warp_descs_dev_.cap || need_desc > warp_descs_host_.cap) {
                    OAR_HIP(hipStreamSynchronize(s));   // earlier launches may still read the buffers that are about to move
                    const size_t want = std::max<size_t>(2 * take * sizeof(pp::WarpDesc), 8192 * sizeof(pp::WarpDesc));
                    warp_descs_host_.reserve(want); warp_descs_dev_.reserve(want);
                    desc_used = 0;
                }
                pp::WarpDesc* wd = warp_descs_host_.as<pp::WarpDesc>() + desc_used;
                pp::WarpDesc* wd_dev = warp_descs_dev_.as<pp::WarpDesc>() + desc_used;
                for (size_t q = 0; q < take; ++q) {
                    const Planned& P = planned[pi + q];
                    const host::CropPlan& pl = P.plan;
                    pp::WarpDesc& d = wd[q];
                    d.page = dev_pages[P.img - start]; d.page_w = (int)pages[P.img].w; d.page_h = (int)pages[P.img].h;
                    d.left = pl.left; d.top = pl.top; d.cw = pl.cw; d.ch = pl.ch; d.ow = pl.ow; d.oh = pl.oh; d.rot = pl.rot; d.mode = pl.mode;
                    std::memcpy(d.inv, pl.inv, sizeof d.inv);
                    d.out_off = (int64_t)pool_bytes;
                    PoolItem it;
                    it.img = P.img; it.det_index = P.det_index; it.w = (uint32_t)pl.out_w(); it.h = (uint32_t)pl.out_h();
                    it.wh_ratio = (float)it.w / (float)std::max<uint32_t>(it.h, 1);  // ocr.rs:739
                    it.off = pool_bytes;
                    pool.push_back(it);
                    pool_bytes += ((size_t)it.w * it.h * 3 + 63) & ~(size_t)63;
                }
                OAR_HIP(hipMemcpyAsync(wd_dev, wd, take * sizeof(pp::WarpDesc), hipMemcpyHostToDevice, s));
                pp::rotate_crops(s, wd_dev, (int)take, crop_pool_.as<uint8_t>(), max_px);
                desc_used += take;
                pi += take;
                if (pool.size() >= cfg_.max_pooled_crops) { flush(); ++flush_count; desc_used = 0; }
            }
        };
        auto guarded_plan = [&](int first, int count) {
            try { plan_pages(first, count); } catch (...) { planning_failed = true; throw; }
        };
        det_->run(chunk, cfg_.det_thresh, cfg_.det_box_thresh, cfg_.det_unclip_ratio, boxes, &dev_pages, guarded_plan);
        // the detector's page staging buffer is reused by the next chunk: crops must be done first
        OAR_HIP(hipStreamSynchronize(s));
        tmark("crop_sync");
    };
    for (int start = 0; start < n; start += (int)cfg_.image_batch_size) {
        const int end = std::min(n, start + (int)cfg_.image_batch_size);
        const size_t pool_mark = pool.size(), bytes_mark = pool_bytes;
        const int flush_mark = flush_count;
        try {
            run_chunk(start, end);
        } catch (const Error& err) {
            // "Batched text detection failed; falling back to per-image detection" (src/oarocr/ocr.rs:576-588): the chunk is
            // redone page by page; a page that fails on its own fails the call, as the reference's `?` does.
            // only a DETECTOR failure is retried page by page; an error raised while planning / warping / recognising crops (a flush may
            // already have mixed this chunk's crops into batches with earlier pages) fails the call, as any error after detection does
            // in the reference (ocr.rs:590-640: `?` on everything behind the detection fallback)
            if (end - start <= 1 || planning_failed) throw;
            fprintf(stderr, "[oar] batched text detection failed (%s); falling back to per-image detection for pages %d..%d\n", err.what(), start, end);
            (void)hipStreamSynchronize(s);
            (void)hipGetLastError();
            // forget what the failed attempt planned for this chunk's pages (a flush in between consumed everything older)
            if (flush_count == flush_mark) { pool.resize(pool_mark); pool_bytes = bytes_mark; }
            else {
                pool.erase(std::remove_if(pool.begin(), pool.end(), [&](const PoolItem& it) { return it.img >= start; }), pool.end());
                pool_bytes = pool.empty() ? 0 : pool.back().off + ((((size_t)pool.back().w * pool.back().h * 3) + 63) & ~(size_t)63);
            }
            for (int i = start; i < end; ++i) per_image[i].clear();
            for (int i = start; i < end; ++i) run_chunk(i, i + 1);
        }
    }
    flush();
    out.assign(n, {});
    for (int i = 0; i < n; ++i)
        for (auto& sl : per_image[i])
            if (sl.filled) out[i].push_back(std::move(sl.r));
    tmark("assemble");
    timer.dump("ocr.predict");
    if (Profiler::get().enabled) Profiler::get().flush();
}

}  // namespace oar
