// contours.hip -- border following (imageproc 0.27 `find_contours`, the Suzuki-Abe variant called at
// processors/db_bitmap.rs:100) on the GPU, so that a detector mask never crosses PCIe: the host receives the border
// chains (a few KB per page) instead of the H x W mask (0.9 MB per 960^2 page).
//
// Border following is sequential by nature -- which pixel starts the next border, and whether a pixel can still start one,
// depends on the marks earlier traces left -- and its output ORDER is part of the contract (raster discovery order, then
// `take(max_candidates)`).  What IS independent:
//   * a border never crosses a fully blank ROW, so the image falls into bands (maximal runs of non-blank rows);
//   * inside a band, a border never crosses a fully blank COLUMN of that band, so the band falls into segments.
// A segment is a rectangle [y0, y1) x [x0, x1) with background all around it (or the image edge).  Nothing outside it can
// influence the walk inside it -- marks only ever touch pixels of the component being followed, neighbour reads across its
// edge see background either way, and the algorithm's two references to absolute position (`x > 0` for an outer start,
// `x + 1 == width` for the right-edge mark) are kept by working in image coordinates.  So every segment is followed
// independently (one wavefront each) and yields exactly the borders the whole-image scan finds in it; the whole-image
// DISCOVERY ORDER is then the borders sorted by their start pixel (y, x) -- a pixel starts at most one border -- which the
// host does when it merges the segments of a page.  A 960^2 text mask gives a few hundred segments of one word each.
//
// One wavefront per segment:
//   * the segment's state lives in LDS at 2 bits per pixel: 00 background, 01 foreground not yet on a followed border,
//     11 marked positive, 10 marked negative -- the only distinctions the algorithm ever makes (`== 1`, `> 0`, `!= 0`; the
//     border numbers themselves only feed the hierarchy, which DB post-processing never reads).  With these codes a mark is
//     an OR (positive: |= 10, which leaves a negative pixel negative) or an OR + AND (negative: |= 10, &= ~01): LDS atomics
//     without a return value, so marking never stalls the walk -- and since a mark never turns a pixel into background or
//     back, the walk itself (which pixel comes next) only ever reads bits that do not change;
//   * raster scan for border starts, 64 pixels per step (a ballot over the start conditions; marks can only REMOVE
//     candidates, so after each followed border the remaining lanes are simply re-evaluated);
//   * following a border: the 8 neighbours of the current pixel are read by 8 lanes at once (the state is framed by
//     background, so no bounds checks), the ballot gives the next pixel and the "right neighbour examined" flag in a handful
//     of scalar ops -- one LDS round trip per border pixel.
// Output per segment: a word stream [header = hole << 31 | n_points][x | y << 16]... in the segment's discovery order,
// appended to a packed buffer in pinned host memory (one atomicAdd per segment for the offset; the segment table records where).
// A segment that does not fit LDS, a band with more segments than the table holds, or a stream that overflows its scratch
// region / the packed buffer is flagged and followed on the host from the mask (pipeline.cc fetches the mask lazily in that
// case) -- correctness never depends on a capacity.
#include "prepost.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace oar {
namespace pp {

namespace {

constexpr int kTraceLdsSmall = 8 * 1024;    // segments up to this much state (a word of a text line needs ~1 KB)
constexpr int kTraceLdsLarge = 64 * 1024;   // larger ones; beyond this the segment is flagged for the host

// ---- row occupancy: one wave per (page, row)
__global__ __launch_bounds__(256) void row_any_kernel(const uint8_t* __restrict__ masks, int n_rows_total, int W, uint8_t* __restrict__ rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows_total) return;
    const int lane = threadIdx.x & 63;
    const uint8_t* m = masks + (size_t)row * W;
    bool any = false;
    if ((W & 15) == 0 && ((size_t)m & 15) == 0) {
        const uint4* m4 = reinterpret_cast<const uint4*>(m);
        for (int i = lane; i < W / 16; i += 64) { const uint4 v = m4[i]; any = any || (v.x | v.y | v.z | v.w) != 0; }
    } else {
        for (int i = lane; i < W; i += 64) any = any || m[i] != 0;
    }
    const unsigned long long b = __ballot(any);
    if (lane == 0) rows[row] = b != 0 ? 1 : 0;
}

// LDS layout of a segment of `ws` columns: 2 bits per pixel, one background row above and below and 16 background pixels left
// and right, so that the 8 neighbours of any pixel can be read without a bounds check: local pixel xl of segment row ry sits
// at bit 2 * (xl & 15) of dword (ry + 1) * WDp + ((xl + 16) >> 4)
__host__ __device__ inline int trace_wdp(int ws) { return ((ws + 15) >> 4) + 2; }
__host__ __device__ inline size_t trace_lds_need(int rows, int ws) { return (size_t)(rows + 2) * trace_wdp(ws) * 4; }

// ---- bands = maximal runs of non-blank rows; one wave per page
__global__ __launch_bounds__(64) void band_scan_kernel(const uint8_t* __restrict__ rows, int H, int maxb, int32_t* __restrict__ band_y, int32_t* __restrict__ n_bands) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const uint8_t* r = rows + (size_t)p * H;
    int32_t* by = band_y + (size_t)p * maxb * 2;
    int n_start = 0, n_end = 0;
    unsigned long long carry = 0;
    for (int b = 0; b < H; b += 64) {
        const int y = b + lane;
        const unsigned long long m = __ballot(y < H && r[y] != 0);
        const unsigned long long prev = (m << 1) | carry;
        const unsigned long long starts = m & ~prev, ends = ~m & prev;   // ends: first blank row after a run (rows >= H read as blank)
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if ((starts >> lane) & 1) by[(n_start + __popcll(starts & below)) * 2] = y;
        if ((ends >> lane) & 1) by[(n_end + __popcll(ends & below)) * 2 + 1] = y < H ? y : H;
        n_start += __popcll(starts);
        n_end += __popcll(ends);
        carry = m >> 63;
    }
    if (lane == 0) {
        if (n_end < n_start) by[n_end * 2 + 1] = H;   // the last run touches the bottom row
        n_bands[p] = n_start;
    }
}

// ---- segments = maximal runs of non-blank columns of a band; one wave per (band, page).  Each segment gets a table entry and
// goes to the work list of its LDS class; what cannot be handled here is flagged (flags = 1, used = 0) for the host.
__global__ __launch_bounds__(64) void segment_scan_kernel(const uint8_t* __restrict__ masks, int H, int W, int maxb, const int32_t* __restrict__ band_y,
                                                          const int32_t* __restrict__ n_bands, uint32_t* __restrict__ small_list, uint32_t* __restrict__ large_list,
                                                          uint32_t* ctrl, SegRec* __restrict__ table_dev, SegRec* __restrict__ table, uint32_t table_cap) {
    const int p = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_bands[p]) return;
    const int y0 = band_y[((size_t)p * maxb + j) * 2], y1 = band_y[((size_t)p * maxb + j) * 2 + 1];
    const uint8_t* m = masks + ((size_t)p * H + y0) * W;
    const int rows = y1 - y0;
    // pass 1: count the runs
    uint32_t n_runs = 0;
    unsigned long long carry = 0;
    for (int b = 0; b < W; b += 64) {
        const int x = b + lane;
        bool any = false;
        if (x < W) for (int r = 0; r < rows; ++r) any = any || m[(size_t)r * W + x] != 0;
        const unsigned long long mm = __ballot(any);
        n_runs += __popcll(mm & ~((mm << 1) | carry));
        carry = mm >> 63;
    }
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctrl[kTraceCtlSegments], n_runs == 0 ? 1u : n_runs);
    base = __shfl(base, 0, 64);
    if (n_runs == 0 || (unsigned long long)base + n_runs > table_cap) {
        // more segments than the table holds: one entry for the whole band (if even that fits), flagged for the host
        if (lane == 0 && base < table_cap) {
            SegRec* rec = &table[base];
            rec->page = p; rec->y0 = y0; rec->y1 = y1; rec->x0 = 0; rec->x1 = W; rec->off = 0; rec->used = 0; rec->n_contours = 0; rec->flags = 1;
        }
        if (lane == 0 && base >= table_cap) atomicOr(&ctrl[kTraceCtlOverflow], 1u);
        return;
    }
    // pass 2: write the segments (the column occupancy is simply recomputed)
    uint32_t n_start = 0, n_end = 0;
    carry = 0;
    for (int b = 0; b < W + 64; b += 64) {   // one extra group so that a run touching the right edge gets its end
        const int x = b + lane;
        bool any = false;
        if (x < W) for (int r = 0; r < rows; ++r) any = any || m[(size_t)r * W + x] != 0;
        const unsigned long long mm = __ballot(any);
        const unsigned long long prev = (mm << 1) | carry;
        const unsigned long long starts = mm & ~prev, ends = ~mm & prev;
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if ((starts >> lane) & 1) table_dev[base + n_start + __popcll(starts & below)].x0 = x;
        if ((ends >> lane) & 1) table_dev[base + n_end + __popcll(ends & below)].x1 = x < W ? x : W;
        n_start += __popcll(starts);
        n_end += __popcll(ends);
        carry = mm >> 63;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t i = lane; i < n_runs; i += 64) {
        SegRec* rec = &table_dev[base + i];   // geometry for the tracer (device memory); the host table entry is written once, complete
        rec->page = p; rec->y0 = y0; rec->y1 = y1; rec->off = 0; rec->used = 0; rec->n_contours = 0; rec->flags = 0;
        const size_t need = trace_lds_need(rows, rec->x1 - rec->x0);
        if (need <= (size_t)kTraceLdsSmall) small_list[atomicAdd(&ctrl[kTraceCtlNSmall], 1u)] = base + i;
        else if (need <= (size_t)kTraceLdsLarge) large_list[atomicAdd(&ctrl[kTraceCtlNLarge], 1u)] = base + i;
        else { SegRec h = *rec; h.flags = 1; table[base + i] = h; }
    }
}

// direction table of imageproc's border follower: index -> (dx, dy), clockwise starting at "left"
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x1A90u >> (2 * d)) & 3u) - 1; }   // {-1,-1,0,1,1,1,0,-1} + 1 = {0,0,1,2,2,2,1,0}
__device__ __forceinline__ int dir_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }   // { 0,-1,-1,-1,0,1,1,1} + 1 = {1,0,0,0,1,2,2,2}

struct SegState {
    uint32_t* st;   // LDS, (rows + 2) x WDp dwords
    int WDp;
    __device__ __forceinline__ int word(int xl, int ry) const { return (ry + 1) * WDp + ((xl + 16) >> 4); }
    __device__ __forceinline__ uint32_t get(int xl, int ry) const { return (st[word(xl, ry)] >> ((xl & 15) * 2)) & 3u; }   // xl in [-16, ws + 16), ry in [-1, rows]
    // marks (called by ONE lane; the results are unused, so these are fire-and-forget ds_or / ds_and)
    __device__ __forceinline__ void mark(int xl, int ry, bool negative) const {
        uint32_t* w = &st[word(xl, ry)];
        const int sh = (xl & 15) * 2;
        __hip_atomic_fetch_or(w, 2u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (negative) __hip_atomic_fetch_and(w, ~(1u << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
};

// Persistent single-wave workgroups: each takes segments from `list` through a cursor until the list is exhausted.
__global__ __launch_bounds__(64) void trace_segments_kernel(const uint8_t* __restrict__ masks, int H, int W, const uint32_t* __restrict__ list, const uint32_t* n_list,
                                                            uint32_t* cursor, uint32_t* __restrict__ scratch, uint32_t* __restrict__ packed, uint32_t packed_cap,
                                                            uint32_t* total, const SegRec* __restrict__ table_dev, SegRec* __restrict__ table, int prio) {
    extern __shared__ uint32_t lds_state[];
    const int lane = threadIdx.x;
    if (prio) __builtin_amdgcn_s_setprio(3);   // a latency-bound single wave next to the network's throughput waves: issue it first
    const uint32_t n_items = *n_list;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(cursor, 1u);
        item = __shfl(item, 0, 64);
        if (item >= n_items) return;
        const uint32_t seg = list[item];
        const SegRec* geo = &table_dev[seg];
        const int p = geo->page, y0 = geo->y0, y1 = geo->y1, x0 = geo->x0, x1 = geo->x1;
        const int rows = y1 - y0, ws = x1 - x0;
        SegState S;
        S.st = lds_state; S.WDp = trace_wdp(ws);
        // ---- load the segment with its background frame: 16 mask bytes -> one state dword
        const uint8_t* m = masks + ((size_t)p * H + y0) * W + x0;
        const int n_dw = (rows + 2) * S.WDp;
        for (int i = lane; i < n_dw; i += 64) {
            const int pr = i / S.WDp, xd = i - pr * S.WDp;   // padded row / padded dword column
            const int ry = pr - 1, xl0 = (xd - 1) * 16;
            uint32_t v = 0;
            if (ry >= 0 && ry < rows && xl0 >= 0 && xl0 < ws) {
                const uint8_t* src = m + (size_t)ry * W + xl0;
                const int cnt = ws - xl0 < 16 ? ws - xl0 : 16;
                for (int k = 0; k < cnt; ++k) if (src[k]) v |= 1u << (2 * k);
            }
            lds_state[i] = v;
        }
        __syncthreads();

        // this segment's private scratch region: 2 words per pixel of its area, inside its band's region
        uint32_t* out = scratch + ((size_t)p * H + y0) * W * 2 + (size_t)rows * x0 * 2;
        const uint32_t cap = (uint32_t)((size_t)rows * ws * 2 > 0x7fffffffu ? 0x7fffffffu : (size_t)rows * ws * 2);
        uint32_t n = 0, nc = 0;   // words written, contours found (both wave-uniform)

        for (int ry = 0; ry < rows && n <= cap; ++ry) {
            for (int xb = 0; xb < ws && n <= cap; xb += 64) {
                // nothing can start in 64 pixels that are all background
                const bool some = lds_state[(ry + 1) * S.WDp + 1 + (xb >> 4) + (lane & 3)] != 0;   // the frame makes the over-read harmless
                if (__ballot(some) == 0) continue;
                const int xl = xb + lane, x = x0 + xl;   // local / image column of this lane
                int cursor_lane = 0;
                while (n <= cap) {
                    const uint32_t s = xl < ws ? S.get(xl, ry) : 0u;
                    const bool outer = s == 1u && x > 0 && S.get(xl - 1, ry) == 0u;
                    const bool hole = !outer && (s & 1u) && x + 1 < W && S.get(xl + 1, ry) == 0u;   // "> 0": unmarked or marked positive
                    const unsigned long long om = __ballot(outer && lane >= cursor_lane), hm = __ballot(hole && lane >= cursor_lane);
                    const unsigned long long cm = om | hm;
                    if (cm == 0) break;
                    const int L = __builtin_ctzll(cm);
                    const bool is_outer = (om >> L) & 1ull;
                    cursor_lane = L + 1;
                    // ------------------------------------------------------------ follow one border from (sx, sy), local coordinates
                    const int sx = xb + L, sy = ry;
                    const int start = is_outer ? 0 : 4;   // the background neighbour that triggered the start: left / right
                    const uint32_t hdr = n;
                    n += 1;
                    uint32_t count = 0;
                    const int d0 = (start + lane) & 7;
                    const unsigned long long m1 = __ballot(S.get(sx + dir_dx(d0), sy + dir_dy(d0)) != 0u) & 0xffull;
                    if (m1 == 0) {   // isolated pixel
                        if (lane == 0) { if (n < cap) out[n] = (uint32_t)(sx + x0) | ((uint32_t)(sy + y0) << 16); S.mark(sx, sy, true); }
                        n += 1; count = 1;
                    } else {
                        const int d1 = (start + __builtin_ctzll(m1)) & 7;   // first non-zero neighbour clockwise from the trigger
                        const int p1x = sx + dir_dx(d1), p1y = sy + dir_dy(d1);
                        int p3x = sx, p3y = sy, front = d1;
                        for (;;) {
                            // the 8 neighbours, one per lane (lanes 8..63 repeat them); a mark never changes "non-zero", so this
                            // read does not depend on the marks still in flight
                            const int d = (front + lane) & 7;
                            const unsigned long long mm = __ballot(S.get(p3x + dir_dx(d), p3y + dir_dy(d)) != 0u) & 0xffull;
                            // counter-clockwise search = k from 7 down to 0; the pixel we came from (k = 0) is non-zero, so mm != 0
                            const int k4 = 63 - __builtin_clzll(mm | 1ull);
                            const int d4 = (front + k4) & 7;
                            const int kr = (4 - front) & 7;   // at which k the right neighbour (+1, 0) is examined
                            const bool negative = p3x + x0 + 1 == W || kr > k4;
                            if (lane == 0) {
                                if (n < cap) out[n] = (uint32_t)(p3x + x0) | ((uint32_t)(p3y + y0) << 16);
                                S.mark(p3x, p3y, negative);
                            }
                            n += 1; count += 1;
                            const int p4x = p3x + dir_dx(d4), p4y = p3y + dir_dy(d4);
                            if (p4x == sx && p4y == sy && p3x == p1x && p3y == p1y) break;
                            front = (d4 + 4) & 7;   // direction from the new current pixel back to the old one
                            p3x = p4x; p3y = p4y;
                            if (n > cap + 64u) break;   // hopeless overflow: stop producing, the segment is flagged below
                        }
                    }
                    if (lane == 0 && hdr < cap) out[hdr] = (is_outer ? 0u : 0x80000000u) | count;
                    nc += 1;
                }
            }
        }
        // ---- publish: reserve space in the packed (host-visible) buffer, copy, fill the table entry
        uint32_t off = 0, flags = n > cap ? 1u : 0u;
        if (!flags) {
            if (lane == 0) off = atomicAdd(total, n);
            off = __shfl(off, 0, 64);
            if ((unsigned long long)off + n > packed_cap) flags = 1u;
        }
        if (!flags) {
            __threadfence_block();
            for (uint32_t i = lane; i < n; i += 64) packed[off + i] = out[i];
        }
        if (lane == 0) {
            SegRec h;
            h.page = p; h.y0 = y0; h.y1 = y1; h.x0 = x0; h.x1 = x1; h.off = off; h.used = flags ? 0u : n; h.n_contours = flags ? 0u : nc; h.flags = flags;
            table[seg] = h;
        }
        __syncthreads();   // the next segment reuses the LDS state
    }
}

__global__ void publish_ctrl_kernel(const uint32_t* ctrl, uint32_t* ctrl_host) {
    if (threadIdx.x < kTraceCtlWords) ctrl_host[threadIdx.x] = ctrl[threadIdx.x];
}

}  // namespace

void trace_contours(hipStream_t s, const uint8_t* masks, int n_pages, int H, int W, uint8_t* rows, int32_t* band_y, int32_t* n_bands, uint32_t* lists,
                    uint32_t list_cap, uint32_t* scratch, uint32_t* packed, uint32_t packed_cap_words, uint32_t* ctrl, uint32_t* ctrl_host, SegRec* table_dev,
                    SegRec* table, uint32_t table_cap) {
    if (n_pages <= 0 || H <= 0 || W <= 0) return;
    OAR_CHECK(W < 65536 && H < 65536, OAR_INVALID_INPUT, "trace_contours: mask larger than 65535 pixels on a side");
    OAR_CHECK(list_cap >= table_cap, OAR_INTERNAL, "trace_contours: work lists smaller than the segment table");
    ProfScope ps(s, "contours", (double)n_pages * H * W, 0.0);
    OAR_HIP(hipMemsetAsync(ctrl, 0, kTraceCtlWords * sizeof(uint32_t), s));
    const int n_rows = n_pages * H, maxb = trace_max_bands(H);
    uint32_t* small_list = lists;
    uint32_t* large_list = lists + list_cap;
    hipLaunchKernelGGL(row_any_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, masks, n_rows, W, rows);
    hipLaunchKernelGGL(band_scan_kernel, dim3((unsigned)n_pages), dim3(64), 0, s, (const uint8_t*)rows, H, maxb, band_y, n_bands);
    hipLaunchKernelGGL(segment_scan_kernel, dim3((unsigned)maxb, (unsigned)n_pages), dim3(64), 0, s, masks, H, W, maxb, (const int32_t*)band_y, (const int32_t*)n_bands,
                       small_list, large_list, ctrl, table_dev, table, table_cap);
    // persistent workgroups, one launch per LDS class; a launch whose list is empty costs a few microseconds
    static const int prio = [] { const char* e = getenv("OAR_TRACE_PRIO"); return e ? atoi(e) : 1; }();
    const unsigned g_small = (unsigned)std::min<size_t>((size_t)n_pages * 64, 2048), g_large = (unsigned)std::min<size_t>((size_t)n_pages * 2, 32);
    hipLaunchKernelGGL(trace_segments_kernel, dim3(g_small), dim3(64), kTraceLdsSmall, s, masks, H, W, (const uint32_t*)small_list,
                       (const uint32_t*)(ctrl + kTraceCtlNSmall), ctrl + kTraceCtlCurSmall, scratch, packed, packed_cap_words, ctrl + kTraceCtlTotal, (const SegRec*)table_dev, table, prio);
    hipLaunchKernelGGL(trace_segments_kernel, dim3(g_large), dim3(64), kTraceLdsLarge, s, masks, H, W, (const uint32_t*)large_list,
                       (const uint32_t*)(ctrl + kTraceCtlNLarge), ctrl + kTraceCtlCurLarge, scratch, packed, packed_cap_words, ctrl + kTraceCtlTotal, (const SegRec*)table_dev, table, prio);
    hipLaunchKernelGGL(publish_ctrl_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)ctrl, ctrl_host);
    OAR_HIP(hipGetLastError());
}

}  // namespace pp
}  // namespace oar
