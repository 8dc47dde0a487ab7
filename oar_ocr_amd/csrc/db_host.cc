#include "db_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace oar {
namespace host {

namespace {
constexpr float kEps = 1.1920929e-7f;          // f32::EPSILON
constexpr double kEpsD = 2.220446049250313e-16; // f64::EPSILON
constexpr float kPi = 3.14159265358979323846f;
constexpr double kPiD = 3.14159265358979323846;

inline uint32_t sat_u32(float v) {  // Rust `as u32`
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
inline int32_t total_order_key(float f) {  // f32::total_cmp
    int32_t i;
    std::memcpy(&i, &f, 4);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
// OAR_HOST_FAST=0: the round-4 host route (byte state plane, every point through simplify_chain and the sort) for A/B timing; same results
inline bool host_fast() { static const bool v = [] { const char* e = getenv("OAR_HOST_FAST"); return !(e && e[0] == '0'); }(); return v; }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

// ------------------------------------------------------------------------------------------ contours
std::vector<Contour> find_contours(const uint8_t* mask, int width, int height, size_t max_contours) {
    return find_contours_band(mask, width, height, 0, height, max_contours, nullptr);
}

std::vector<int> blank_row_bands(const uint8_t* mask, int width, int height, int max_bands) {
    std::vector<uint8_t> occupied(height, 0);
    int fg_rows = 0;
    for (int y = 0; y < height; ++y) {
        const uint8_t* r = mask + (size_t)y * width;
        int x = 0;
        bool any = false;
        for (; x + 8 <= width; x += 8) { uint64_t w8; std::memcpy(&w8, r + x, 8); if (w8) { any = true; break; } }
        if (!any) for (; x < width; ++x) if (r[x]) { any = true; break; }
        occupied[y] = any;
        fg_rows += any;
    }
    std::vector<int> cuts{0};
    if (max_bands > 1 && fg_rows > 0) {
        const int target = (fg_rows + max_bands - 1) / max_bands;
        int acc = 0;
        for (int y = 0; y < height; ++y) {
            if (occupied[y]) { ++acc; continue; }
            if (acc >= target && y > cuts.back()) { cuts.push_back(y); acc = 0; }   // y is blank: safe cut
        }
    }
    cuts.push_back(height);
    return cuts;
}

namespace {
// Border following over rows [band_y0, band_y1): `fill_row(r, dst)` writes the 0 / 1 foreground flags of band row r into dst[0 .. width)
template <typename FillRow>
std::vector<Contour> follow_band(int width, int band_y0, int band_y1, size_t max_contours, FillRow fill_row) {
    // imageproc's border labels only ever matter through three predicates -- `== 1` (foreground not yet on a followed border),
    // `> 0` (that, or marked with a positive label) and `!= 0` -- plus the hierarchy (parent links), which DB post-processing
    // never reads.  The state is therefore one byte per pixel: 0 background, 1 unmarked, 2 marked positive, 3 marked negative,
    // in a plane framed by one background pixel on every side (the band may be cut only at blank rows, so its neighbours above
    // and below ARE background), which removes every bounds check from the walk.
    const int rows = band_y1 - band_y0;
    std::vector<Contour> out;
    if (rows <= 0 || width <= 0) return out;
    const int stride = width + 2;
    static thread_local std::vector<uint8_t> plane;
    plane.resize((size_t)(rows + 2) * stride);
    uint8_t* st = plane.data();
    std::memset(st, 0, (size_t)stride);
    std::memset(st + (size_t)(rows + 1) * stride, 0, (size_t)stride);
    for (int r = 0; r < rows; ++r) {
        uint8_t* d = st + (size_t)(r + 1) * stride;
        d[0] = 0;
        fill_row(r, d + 1);
        d[width + 1] = 0;
    }
    // neighbour offsets in the framed plane: w, nw, n, ne, e, se, s, sw (clockwise on screen), twice so that (front + k) needs no mask
    const int off8[16] = {-1, -stride - 1, -stride, -stride + 1, 1, stride + 1, stride, stride - 1,
                          -1, -stride - 1, -stride, -stride + 1, 1, stride + 1, stride, stride - 1};
    bool full = false;

    // One border from start pixel (x, y) [band-local row y]; `first` = direction index of the background neighbour that triggered it
    auto follow = [&](int x, int y, int first, bool hole) {
        Contour c;
        c.hole = hole;
        c.pts.reserve(256);
        uint8_t* const p0 = st + (size_t)(y + 1) * stride + (x + 1);
        int d1 = -1;
        for (int k = 0; k < 8; ++k) if (p0[off8[first + k]]) { d1 = (first + k) & 7; break; }   // clockwise from the trigger
        if (d1 < 0) {
            c.pts.push_back({(float)x, (float)(y + band_y0)});
            *p0 = 3;
        } else {
            uint8_t* const p1 = p0 + off8[d1];
            uint8_t* p3 = p0;
            int px = x, py = y, front = d1;   // front: direction from the current pixel to the previous one
            static const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
            static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
            for (;;) {
                c.pts.push_back({(float)px, (float)(py + band_y0)});
                // counter-clockwise search starting next to the previous pixel; the reference's second loop ("was the east
                // neighbour examined before the next pixel was found") is folded into the same pass
                int k4 = 0;
                bool right_edge = false;
                for (int k = 7; k >= 0; --k) {
                    if (p3[off8[front + k]]) { k4 = k; break; }
                    if (((front + k) & 7) == 4) right_edge = true;
                }
                const int d4 = (front + k4) & 7;
                if (px + 1 == width || right_edge) *p3 = 3;
                else if (*p3 == 1) *p3 = 2;
                uint8_t* const p4 = p3 + off8[d4];
                if (p4 == p0 && p3 == p1) break;
                p3 = p4; px += DX[d4]; py += DY[d4];
                front = (d4 + 4) & 7;
            }
        }
        out.push_back(std::move(c));
        if (out.size() >= max_contours) full = true;
    };
    // imageproc's loop body for a pixel that can start a border (first / last pixel of a foreground run)
    auto visit = [&](int x, int y) {
        const uint8_t* p = st + (size_t)(y + 1) * stride + (x + 1);
        const uint8_t v = *p;
        if (v == 1 && x > 0 && p[-1] == 0) follow(x, y, 0, false);
        else if ((v == 1 || v == 2) && x + 1 < width && p[1] == 0) follow(x, y, 4, true);
    };

    for (int y = 0; y < rows && !full; ++y) {
        const uint8_t* mrow = st + (size_t)(y + 1) * stride + 1;   // zero-ness of the state plane == zero-ness of the mask, whatever marks it carries
        int x = 0;
        while (x < width && !full) {
            // background never changes state (zero-ness is invariant under border labelling): skip 8 bytes at a time
            if (mrow[x] == 0) {
                while (x + 8 <= width) {
                    uint64_t w8;
                    std::memcpy(&w8, mrow + x, 8);
                    if (w8 != 0) break;
                    x += 8;
                }
                while (x < width && mrow[x] == 0) ++x;
                if (x >= width) break;
            }
            // foreground run [x, xe): interior pixels cannot start a border (both horizontal neighbours are foreground)
            const void* z = std::memchr(mrow + x, 0, (size_t)(width - x));
            const int xe = z ? (int)((const uint8_t*)z - mrow) : width;
            visit(x, y);                                  // may start an outer (or, for a 1-px run, hole) border
            if (full) break;
            if (xe - x > 1) visit(xe - 1, y);             // may start a hole border
            x = xe;
        }
    }
    return out;
}
}  // namespace

std::vector<Contour> find_contours_band(const uint8_t* mask, int width, int height, int band_y0, int band_y1, size_t max_contours, int32_t* /*unused*/) {
    (void)height;
    return follow_band(width, band_y0, band_y1, max_contours, [&](int r, uint8_t* dst) {
        const uint8_t* m = mask + (size_t)(band_y0 + r) * width;
        for (int x = 0; x < width; ++x) dst[x] = m[x] != 0;
    });
}

// ---- the same border following straight from a bit-packed mask (pp::pack_mask_bits: pixel x of row y is bit x & 7 of byte
// bits[y * row_bytes + (x >> 3)]), without ever expanding it to a byte per pixel.
// What the byte version spends its time on is not the walk: it is writing the 0.9 MB state plane of a 960^2 page and reading it back
// to find the run ends.  Here the foreground stays a bit plane (framed by one background bit on every side, so the walk has no
// bounds checks), the state "0 / unmarked / marked positive / marked negative" becomes two more bit planes that start out zero
// (`marked`, `neg`: the three predicates of follow_band read exactly these), run starts / ends come out of 64-bit word arithmetic
// (w & ~(w << 1), w & ~(w >> 1)), and a step of the walk reads the 3 x 3 neighbourhood with three 16-bit loads: a 512-entry table turns it
// into the 8 neighbour flags in direction order, a rotate + count-leading-zeros replaces the search loop.
// `corners_only`: DBPostProcess (Quad boxes, fast score) only ever looks at a contour through simplify_chain; whether a point survives that
// (the step into it differs from the step out of it, db_bitmap.rs:207-239) is known while walking, so only those points are stored and the
// contour is marked `simplified`.  A chain with fewer than three such points is kept whole, as simplify_chain does.
namespace {
struct StepTable {
    uint8_t nb[512];        // 3 x 3 window -> neighbour flags in direction order w, nw, n, ne, e, se, s, sw
    uint8_t step[8 * 512];  // (front, window) -> d4 | right_edge << 3 | (dx + 1) << 4 | (dy + 1) << 6
    StepTable() {
        static const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
        static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
        // window index: bits 0..2 = row above (x - 1, x, x + 1), 3..5 = the pixel's row, 6..8 = row below
        for (int i = 0; i < 512; ++i) {
            const int a = i & 7, m = (i >> 3) & 7, b = i >> 6;
            nb[i] = (uint8_t)(((m >> 0) & 1) | ((a & 1) << 1) | (((a >> 1) & 1) << 2) | (((a >> 2) & 1) << 3) | (((m >> 2) & 1) << 4) |
                              (((b >> 2) & 1) << 5) | (((b >> 1) & 1) << 6) | ((b & 1) << 7));
        }
        for (int front = 0; front < 8; ++front)
            for (int i = 0; i < 512; ++i) {
                // counter-clockwise search starting next to the previous pixel: bit k of r is direction (front + k) & 7, the search runs
                // k = 7 .. 0 and stops at the first foreground one (bit 0, the previous pixel, always is inside a walk); east was examined on
                // the way iff its k is larger
                const unsigned n = nb[i], r = ((n >> front) | (n << (8 - front))) & 0xffu;
                if (!r) { step[front * 512 + i] = 0; continue; }
                const int k4 = 31 - __builtin_clz(r);
                const int d4 = (front + k4) & 7;
                const int right_edge = ((4 - front) & 7) > k4;
                step[front * 512 + i] = (uint8_t)(d4 | (right_edge << 3) | ((DX[d4] + 1) << 4) | ((DY[d4] + 1) << 6));
            }
    }
};
const StepTable kStep;

struct BitFollower {
    int width, band_y0, stride;            // stride: bytes per framed row (a multiple of 8; 8 spare bytes on either side of the pixels for the unaligned loads)
    uint8_t *fg, *marked, *neg;            // framed planes (pointing past a row's left spare bytes): pixel x of band row r is bit (x + 1) of row (r + 1)
    std::vector<Contour>* out;
    size_t max_contours;
    bool corners_only, full = false;
    std::vector<Pt> pts;

    inline unsigned window(int X, int Y) const {   // framed coordinates
        const uint8_t* r = fg + (size_t)Y * stride + ((X - 1) >> 3);
        const int sh = (X - 1) & 7;
        uint16_t a, m, b;
        std::memcpy(&a, r - stride, 2); std::memcpy(&m, r, 2); std::memcpy(&b, r + stride, 2);
        return ((a >> sh) & 7) | (((m >> sh) & 7) << 3) | (((b >> sh) & 7) << 6);
    }

    // one border from (x, y) [band-local row]; `first`: direction of the background neighbour that triggered it.  keep_all: every point,
    // otherwise only the points simplify_chain keeps (returns false, with nothing stored, when those are fewer than three)
    template <bool kKeepAll>
    bool walk(int x, int y, int first) {
        static const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
        static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
        pts.clear();
        const int X0 = x + 1, Y0 = y + 1;
        const unsigned n0 = kStep.nb[window(X0, Y0)];
        if (n0 == 0) {
            pts.push_back({(float)x, (float)(y + band_y0)});
            const size_t o = (size_t)Y0 * stride + (X0 >> 3);
            marked[o] |= (uint8_t)(1u << (X0 & 7)); neg[o] |= (uint8_t)(1u << (X0 & 7));
            return true;
        }
        const int d1 = (first + __builtin_ctz(((n0 >> first) | (n0 << (8 - first))) & 0xffu)) & 7;   // clockwise from the trigger
        const int X1 = X0 + DX[d1], Y1 = Y0 + DY[d1];
        int X = X0, Y = Y0, front = d1;   // front: direction from the current pixel to the previous one
        size_t n_pts = 0;
        for (;;) {
            const unsigned e = kStep.step[front * 512 + window(X, Y)];
            const int d4 = e & 7;
            const size_t o = (size_t)Y * stride + (X >> 3);
            const uint8_t bit = (uint8_t)(1u << (X & 7));
            marked[o] |= bit;
            if (X == width || (e & 8)) neg[o] |= bit;
            ++n_pts;
            if (kKeepAll || (front ^ 4) != d4) pts.push_back({(float)(X - 1), (float)(Y - 1 + band_y0)});   // the step in differs from the step out
            const int X4 = X + (int)((e >> 4) & 3) - 1, Y4 = Y + (int)(e >> 6) - 1;
            if (X4 == X0 && Y4 == Y0 && X == X1 && Y == Y1) break;
            X = X4; Y = Y4; front = d4 ^ 4;
            // Four steps in five run straight along a horizontal edge (a text line's top and bottom): a pixel entered from the west leaves
            // to the east iff its east neighbour is foreground and sw, s, se are background -- then east was not examined (no negative mark)
            // and the step in equals the step out (not a simplify_chain point); mirrored (ne, n, nw) for a pixel entered from the east.
            // Whole runs of such pixels are taken from 64-bit windows of the two rows involved: one ctz / clz, one OR into `marked`.
            if (d4 == 4) {
                for (;;) {
                    const size_t ob = (size_t)Y * stride + ((X - 1) >> 3);
                    const int sh = (X - 1) & 7;
                    uint64_t c, b;
                    std::memcpy(&c, fg + ob, 8); std::memcpy(&b, fg + ob + stride, 8);
                    c >>= sh; b >>= sh;   // bit i = pixel X - 1 + i; 64 - sh >= 57 bits are real
                    const uint64_t go = (c >> 2) & ~b & ~(b >> 1) & ~(b >> 2);   // bit i: pixel X + i continues east
                    int L = __builtin_ctzll(~go | (1ull << 54));
                    if (Y == Y1 && X1 >= X && X1 < X + L) L = X1 - X;   // the walk may end at p1 -> p0: p1 takes an ordinary step
                    if (L == 0) break;
                    uint64_t mk;
                    std::memcpy(&mk, marked + ob, 8);
                    mk |= ((1ull << L) - 1) << (sh + 1);
                    std::memcpy(marked + ob, &mk, 8);
                    n_pts += (size_t)L;
                    if (kKeepAll) for (int i = 0; i < L; ++i) pts.push_back({(float)(X - 1 + i), (float)(Y - 1 + band_y0)});
                    X += L;
                    if (L < 54) break;
                }
            } else if (d4 == 0) {
                for (;;) {
                    const int top = X + 1;                                   // the word's last pixel
                    const ptrdiff_t ob = (ptrdiff_t)Y * stride + (top >> 3) - 7;   // may reach into the row's left spare bytes (zero)
                    const int up = 7 - (top & 7);
                    uint64_t c, a;
                    std::memcpy(&c, fg + ob, 8); std::memcpy(&a, fg + ob - stride, 8);
                    c <<= up; a <<= up;   // bit 63 - i = pixel X + 1 - i; 64 - up >= 57 bits are real
                    const uint64_t go = (c << 2) & ~a & ~(a << 1) & ~(a << 2);   // bit 63 - i: pixel X - i continues west
                    int L = __builtin_clzll(~go | (1ull << 9));
                    if (Y == Y1 && X1 <= X && X1 > X - L) L = X - X1;
                    if (L == 0) break;
                    uint64_t mk;
                    std::memcpy(&mk, marked + ob, 8);
                    mk |= (((1ull << L) - 1) << (63 - L)) >> up;   // pixels X - L + 1 .. X: bits 63 - L .. 62 before the shift back
                    std::memcpy(marked + ob, &mk, 8);
                    n_pts += (size_t)L;
                    if (kKeepAll) for (int i = 0; i < L; ++i) pts.push_back({(float)(X - 1 - i), (float)(Y - 1 + band_y0)});
                    X -= L;
                    if (L < 54) break;
                }
            }
        }
        return kKeepAll || (n_pts > 2 && pts.size() >= 3);
    }

    void follow(int x, int y, int first, bool hole) {
        Contour c;
        c.hole = hole;
        if (corners_only && walk<false>(x, y, first)) c.simplified = pts.size() >= 3;   // a lone pixel comes back as itself
        else walk<true>(x, y, first);   // (again: the marks it leaves are the same) a chain simplify_chain would hand back whole
        c.pts.assign(pts.begin(), pts.end());
        out->push_back(std::move(c));
        if (out->size() >= max_contours) full = true;
    }
};
}  // namespace

std::vector<Contour> find_contours_band_bits(const uint8_t* bits, int row_bytes, int width, int band_y0, int band_y1, size_t max_contours, bool corners_only) {
    if (!host_fast())
        return follow_band(width, band_y0, band_y1, max_contours, [&](int r, uint8_t* dst) {
            const uint8_t* b = bits + (size_t)(band_y0 + r) * row_bytes;
            int x = 0;
            for (; x + 8 <= width; x += 8) {   // one mask byte -> eight 0 / 1 flag bytes (bits 0..6 by a carry-free multiply, bit 7 apart)
                const uint64_t v = b[x >> 3];
                uint64_t flags = ((v & 0x7full) * 0x0002040810204081ull) & 0x0001010101010101ull;
                flags |= (uint64_t)((v >> 7) & 1u) << 56;
                std::memcpy(dst + x, &flags, 8);
            }
            for (; x < width; ++x) dst[x] = (b[x >> 3] >> (x & 7)) & 1u;
        });
    std::vector<Contour> out;
    const int rows = band_y1 - band_y0;
    if (rows <= 0 || width <= 0) return out;
    const int words = (width + 2 + 63) / 64;
    const int stride = words * 8 + 16;
    static thread_local std::vector<uint8_t> planes;
    const size_t plane_bytes = (size_t)(rows + 2) * stride;
    planes.assign(plane_bytes * 3 + 8, 0);
    BitFollower bf;
    bf.width = width; bf.band_y0 = band_y0; bf.stride = stride;
    bf.fg = planes.data() + 8; bf.marked = bf.fg + plane_bytes; bf.neg = bf.marked + plane_bytes;
    bf.out = &out; bf.max_contours = max_contours; bf.corners_only = corners_only;
    // framed foreground: the source row shifted up by one bit
    const int src_words = (row_bytes + 7) / 8;
    for (int r = 0; r < rows; ++r) {
        const uint8_t* src = bits + (size_t)(band_y0 + r) * row_bytes;
        uint8_t* dst = bf.fg + (size_t)(r + 1) * stride;
        uint64_t carry = 0;
        for (int j = 0; j < words; ++j) {
            uint64_t w = 0;
            if (j < src_words) std::memcpy(&w, src + (size_t)j * 8, (size_t)std::min(8, row_bytes - j * 8));
            const uint64_t o = (w << 1) | carry;
            carry = w >> 63;
            std::memcpy(dst + (size_t)j * 8, &o, 8);
        }
    }
    // raster scan: only the first / last pixel of a foreground run can start a border (imageproc's loop body, see follow_band::visit): an
    // outer one where the run starts (x > 0, pixel not yet on a followed border), else a hole border where it ends (x + 1 < width, pixel not
    // marked negative).  A followed border only ever ADDS marks, so the words are simply re-read after each one.
    const int last_word = (width >> 6), last_bit = width & 63;   // framed position of pixel width - 1
    for (int y = 0; y < rows && !bf.full; ++y) {
        const size_t ro = (size_t)(y + 1) * stride;
        uint64_t prev_msb = 0, w;
        std::memcpy(&w, bf.fg + ro, 8);
        for (int j = 0; j < words && !bf.full; ++j) {
            uint64_t next = 0;
            if (j + 1 < words) std::memcpy(&next, bf.fg + ro + (size_t)(j + 1) * 8, 8);
            if (w) {
                uint64_t starts = w & ~((w << 1) | prev_msb);
                uint64_t ends = w & ~((w >> 1) | (next << 63));
                if (j == 0) starts &= ~2ull;                                   // x == 0 never starts an outer border
                if (j == last_word) ends &= ~(1ull << last_bit);               // x == width - 1 never starts a hole border
                uint64_t done = 0;                                             // positions already visited in this word
                for (;;) {
                    uint64_t mk, ng;
                    std::memcpy(&mk, bf.marked + ro + (size_t)j * 8, 8);
                    std::memcpy(&ng, bf.neg + ro + (size_t)j * 8, 8);
                    const uint64_t outer = starts & ~mk, hole = ends & ~ng & ~outer;
                    const uint64_t cand = (outer | hole) & ~done;
                    if (!cand) break;
                    const int b = __builtin_ctzll(cand);
                    done |= (2ull << b) - 1;                                   // (b == 63: 2 << 63 wraps to 0, - 1 = all ones)
                    bf.follow(j * 64 + b - 1, y, ((outer >> b) & 1) ? 0 : 4, !((outer >> b) & 1));
                    if (bf.full) break;
                }
            }
            prev_msb = w >> 63;
            w = next;
        }
    }
    return out;
}

// blank_row_bands for a bit-packed mask
std::vector<int> blank_row_bands_bits(const uint8_t* bits, int row_bytes, int height, int max_bands) {
    std::vector<uint8_t> occupied(height, 0);
    int fg_rows = 0;
    for (int y = 0; y < height; ++y) {
        const uint8_t* r = bits + (size_t)y * row_bytes;
        int x = 0;
        bool any = false;
        for (; x + 8 <= row_bytes; x += 8) { uint64_t w8; std::memcpy(&w8, r + x, 8); if (w8) { any = true; break; } }
        if (!any) for (; x < row_bytes; ++x) if (r[x]) { any = true; break; }
        occupied[y] = any;
        fg_rows += any;
    }
    std::vector<int> cuts{0};
    if (max_bands > 1 && fg_rows > 0) {
        const int target = (fg_rows + max_bands - 1) / max_bands;
        int acc = 0;
        for (int y = 0; y < height; ++y) {
            if (occupied[y]) { ++acc; continue; }
            if (acc >= target && y > cuts.back()) { cuts.push_back(y); acc = 0; }   // y is blank: safe cut
        }
    }
    cuts.push_back(height);
    return cuts;
}

// ------------------------------------------------------------------------------------------ hull / min-area rect
std::vector<Pt> convex_hull(const std::vector<Pt>& src) {
    if (src.size() < 3) return src;
    const size_t n = src.size();
    size_t si = 0;
    float mnx = src[0].x, mxx = src[0].x, mny = src[0].y, mxy = src[0].y;
    bool integral = true;
    for (size_t i = 0; i < n; ++i) {
        const Pt& q = src[i];
        if (q.y < src[si].y || (q.y == src[si].y && q.x < src[si].x)) si = i;
        mnx = std::min(mnx, q.x); mxx = std::max(mxx, q.x); mny = std::min(mny, q.y); mxy = std::max(mxy, q.y);
        integral = integral && q.x == std::floor(q.x) && q.y == std::floor(q.y);
    }
    const Pt s = src[si];
    // The points that enter the sort.  The reference sorts ALL of them (geometry.rs:226-271: Graham scan, keys atan2f then squared
    // distance, stable).  For border pixels -- integer coordinates -- inside a box whose diagonal is at most 1000 px every quantity of
    // the scan is exact: two directions from s that differ at all differ by >= 1 / (|p| |q|) >= 1e-6 rad, four ulps of an angle in
    // (0, pi], so the atan2f keys (glibc: < 1 ulp, and a function of the float quotient y / x, hence equal along a ray) order the
    // points exactly by angle; the squared distances and the cross products are integers below 2^24.  An exact Graham scan returns the
    // strict hull vertices counter-clockwise from s whatever else was in its input, and a strict hull vertex is the leftmost or the
    // rightmost point of its row -- so only those (at most two per row) are sorted: a text line's few hundred corner points become
    // a few dozen, and the atan2f calls with them.  Anything else (non-integer points: unclipped polygons; larger boxes) takes the
    // reference's route point for point.
    static thread_local std::vector<Pt> cand;
    cand.clear();
    cand.push_back(s);
    const float ex = mxx - mnx, ey = mxy - mny;
    if (host_fast() && integral && n > 12 && ex * ex + ey * ey <= 1.0e6f && std::isfinite(ex) && std::isfinite(ey)) {
        const int R = (int)ey + 1, y0 = (int)mny;
        static thread_local std::vector<float> lo, hi;
        lo.assign((size_t)R, INFINITY); hi.assign((size_t)R, -INFINITY);
        for (const Pt& q : src) {
            const int r = (int)q.y - y0;
            lo[r] = std::min(lo[r], q.x); hi[r] = std::max(hi[r], q.x);
        }
        for (int r = 0; r < R; ++r) {
            if (lo[r] > hi[r]) continue;   // a row without points
            const float y = (float)(y0 + r);
            if (!(r == 0 && lo[r] == s.x)) cand.push_back({lo[r], y});
            if (hi[r] != lo[r]) cand.push_back({hi[r], y});
        }
    } else {
        for (size_t i = 0; i < n; ++i) if (i != si) cand.push_back(src[i]);
    }
    // the comparator of the reference (atan2 total_cmp, then squared distance) is a pure function of each point: evaluate it once
    // per point instead of once per comparison; the index as the last key makes std::sort the reference's stable sort
    struct Keyed { int32_t ang, dist; uint32_t idx; };
    static thread_local std::vector<Keyed> keyed;
    keyed.resize(cand.size() - 1);
    for (size_t i = 1; i < cand.size(); ++i) {
        const Pt& a = cand[i];
        float d = (a.x - s.x) * (a.x - s.x) + (a.y - s.y) * (a.y - s.y);
        keyed[i - 1] = {total_order_key(std::atan2(a.y - s.y, a.x - s.x)), total_order_key(d), (uint32_t)i};
    }
    std::sort(keyed.begin(), keyed.end(), [](const Keyed& a, const Keyed& b) {
        if (a.ang != b.ang) return a.ang < b.ang;
        if (a.dist != b.dist) return a.dist < b.dist;
        return a.idx < b.idx;
    });
    std::vector<Pt> hull;
    hull.reserve(std::min<size_t>(cand.size(), 64));
    auto feed = [&](const Pt& p) {
        while (hull.size() > 1) {
            const Pt &a = hull[hull.size() - 2], &b = hull[hull.size() - 1];
            float cr = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
            if (cr <= 0.0f) hull.pop_back();
            else break;
        }
        hull.push_back(p);
    };
    feed(s);
    for (const Keyed& k : keyed) feed(cand[k.idx]);
    return hull;
}

// The inner loop of the reference's min-area rectangle (geometry.rs:381-409: every hull edge against every hull vertex, h^2 projections --
// 3 000 for the 56-gon an unclipped text line is): pn = nx dx + ny dy, pp = -ny dx + nx dy with their running minima / maxima.  Each lane
// of the vector forms performs the scalar statement sequence (separate multiplies and adds, `v < m ? v : m` as min, `v > m ? v : m` as max),
// and a minimum / maximum does not depend on the order it is taken in, so the four numbers are the scalar loop's (up to the sign of a
// zero, which nothing downstream can see: the extents enter as differences and sums).
namespace {
using ExtentsFn = void (*)(const float*, const float*, size_t, float, float, float, float, float*);
void extents_scalar(const float* xs, const float* ys, size_t n, float hix, float hiy, float nx, float ny, float* ext) {
    const float px = -ny, py = nx;
    float mnn = std::numeric_limits<float>::max(), mxn = std::numeric_limits<float>::lowest();
    float mnp = mnn, mxp = mxn;
    for (size_t k = 0; k < n; ++k) {
        float dx = xs[k] - hix, dy = ys[k] - hiy;
        float pn = nx * dx + ny * dy, pp = px * dx + py * dy;
        if (pn < mnn) mnn = pn;
        if (pn > mxn) mxn = pn;
        if (pp < mnp) mnp = pp;
        if (pp > mxp) mxp = pp;
    }
    ext[0] = mnn; ext[1] = mxn; ext[2] = mnp; ext[3] = mxp;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
void extents_sse2(const float* xs, const float* ys, size_t n, float hix, float hiy, float nx, float ny, float* ext) {   // n % 4 == 0
    const __m128 vhx = _mm_set1_ps(hix), vhy = _mm_set1_ps(hiy), vnx = _mm_set1_ps(nx), vny = _mm_set1_ps(ny), vpx = _mm_set1_ps(-ny);
    __m128 mnn = _mm_set1_ps(std::numeric_limits<float>::max()), mxn = _mm_set1_ps(std::numeric_limits<float>::lowest()), mnp = mnn, mxp = mxn;
    for (size_t k = 0; k < n; k += 4) {
        const __m128 dx = _mm_sub_ps(_mm_loadu_ps(xs + k), vhx), dy = _mm_sub_ps(_mm_loadu_ps(ys + k), vhy);
        const __m128 pn = _mm_add_ps(_mm_mul_ps(vnx, dx), _mm_mul_ps(vny, dy)), pp = _mm_add_ps(_mm_mul_ps(vpx, dx), _mm_mul_ps(vnx, dy));
        mnn = _mm_min_ps(pn, mnn); mxn = _mm_max_ps(pn, mxn); mnp = _mm_min_ps(pp, mnp); mxp = _mm_max_ps(pp, mxp);
    }
    float a[4], b[4], c[4], d[4];
    _mm_storeu_ps(a, mnn); _mm_storeu_ps(b, mxn); _mm_storeu_ps(c, mnp); _mm_storeu_ps(d, mxp);
    ext[0] = std::min(std::min(a[0], a[1]), std::min(a[2], a[3])); ext[1] = std::max(std::max(b[0], b[1]), std::max(b[2], b[3]));
    ext[2] = std::min(std::min(c[0], c[1]), std::min(c[2], c[3])); ext[3] = std::max(std::max(d[0], d[1]), std::max(d[2], d[3]));
}
__attribute__((target("avx"))) void extents_avx(const float* xs, const float* ys, size_t n, float hix, float hiy, float nx, float ny, float* ext) {   // n % 8 == 0
    const __m256 vhx = _mm256_set1_ps(hix), vhy = _mm256_set1_ps(hiy), vnx = _mm256_set1_ps(nx), vny = _mm256_set1_ps(ny), vpx = _mm256_set1_ps(-ny);
    __m256 mnn = _mm256_set1_ps(std::numeric_limits<float>::max()), mxn = _mm256_set1_ps(std::numeric_limits<float>::lowest()), mnp = mnn, mxp = mxn;
    for (size_t k = 0; k < n; k += 8) {
        const __m256 dx = _mm256_sub_ps(_mm256_loadu_ps(xs + k), vhx), dy = _mm256_sub_ps(_mm256_loadu_ps(ys + k), vhy);
        const __m256 pn = _mm256_add_ps(_mm256_mul_ps(vnx, dx), _mm256_mul_ps(vny, dy)), pp = _mm256_add_ps(_mm256_mul_ps(vpx, dx), _mm256_mul_ps(vnx, dy));
        mnn = _mm256_min_ps(pn, mnn); mxn = _mm256_max_ps(pn, mxn); mnp = _mm256_min_ps(pp, mnp); mxp = _mm256_max_ps(pp, mxp);
    }
    // lanes -> one number each, in registers (eight lanes of four accumulators through memory cost as much as the loop of a 56-gon)
    __m128 a = _mm_min_ps(_mm256_castps256_ps128(mnn), _mm256_extractf128_ps(mnn, 1)), b = _mm_max_ps(_mm256_castps256_ps128(mxn), _mm256_extractf128_ps(mxn, 1));
    __m128 c = _mm_min_ps(_mm256_castps256_ps128(mnp), _mm256_extractf128_ps(mnp, 1)), d = _mm_max_ps(_mm256_castps256_ps128(mxp), _mm256_extractf128_ps(mxp, 1));
    a = _mm_min_ps(a, _mm_movehl_ps(a, a)); b = _mm_max_ps(b, _mm_movehl_ps(b, b)); c = _mm_min_ps(c, _mm_movehl_ps(c, c)); d = _mm_max_ps(d, _mm_movehl_ps(d, d));
    a = _mm_min_ss(a, _mm_shuffle_ps(a, a, 1)); b = _mm_max_ss(b, _mm_shuffle_ps(b, b, 1)); c = _mm_min_ss(c, _mm_shuffle_ps(c, c, 1)); d = _mm_max_ss(d, _mm_shuffle_ps(d, d, 1));
    ext[0] = _mm_cvtss_f32(a); ext[1] = _mm_cvtss_f32(b); ext[2] = _mm_cvtss_f32(c); ext[3] = _mm_cvtss_f32(d);
}
#endif
ExtentsFn pick_extents() {
    if (!host_fast()) return nullptr;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx") ? extents_avx : extents_sse2;
#else
    return extents_scalar;
#endif
}
const ExtentsFn g_extents = pick_extents();
}  // namespace

MinAreaRect min_area_rect(const std::vector<Pt>& src) {
    MinAreaRect zero{0, 0, 0, 0, 0};
    if (src.size() < 3) return zero;
    std::vector<Pt> hp = convex_hull(src);
    if (hp.size() < 3) {
        float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
        for (const Pt& p : src) {
            if (p.x < mnx) mnx = p.x;
            if (p.x > mxx) mxx = p.x;
            if (p.y < mny) mny = p.y;
            if (p.y > mxy) mxy = p.y;
        }
        if (!std::isfinite(mnx)) return zero;
        return {(mnx + mxx) * 0.5f, (mny + mxy) * 0.5f, mxx - mnx, mxy - mny, 0.0f};
    }
    float min_area = std::numeric_limits<float>::max();
    MinAreaRect best = zero;
    const size_t n = hp.size();
    // the hull as x / y arrays, padded to a multiple of 8 with copies of the last vertex (a duplicate changes no minimum / maximum)
    static thread_local std::vector<float> xs, ys;
    const size_t np = (n + 7) & ~(size_t)7;
    xs.resize(np); ys.resize(np);
    for (size_t i = 0; i < np; ++i) { const Pt& q = hp[i < n ? i : n - 1]; xs[i] = q.x; ys[i] = q.y; }
    for (size_t i = 0; i < n; ++i) {
        size_t j = (i + 1) % n;
        float ex = hp[j].x - hp[i].x, ey = hp[j].y - hp[i].y;
        float el2 = ex * ex + ey * ey;
        if (el2 < kEps) continue;
        float inv = 1.0f / std::sqrt(el2);
        float nx = ex * inv, ny = ey * inv, px = -ny, py = nx;
        float hix = hp[i].x, hiy = hp[i].y;
        float ext[4];   // min / max of the projections on the edge direction, min / max on its normal
        if (g_extents) g_extents(xs.data(), ys.data(), np, hix, hiy, nx, ny, ext);
        else {   // OAR_HOST_FAST=0: the loop as round 4 had it
            float mnn = std::numeric_limits<float>::max(), mxn = std::numeric_limits<float>::lowest();
            float mnp = mnn, mxp = mxn;
            for (const Pt& q : hp) {
                float dx = q.x - hix, dy = q.y - hiy;
                float pn = nx * dx + ny * dy, pp = px * dx + py * dy;
                if (pn < mnn) mnn = pn;
                if (pn > mxn) mxn = pn;
                if (pp < mnp) mnp = pp;
                if (pp > mxp) mxp = pp;
            }
            ext[0] = mnn; ext[1] = mxn; ext[2] = mnp; ext[3] = mxp;
        }
        const float mnn = ext[0], mxn = ext[1], mnp = ext[2], mxp = ext[3];
        float w = mxn - mnn, h = mxp - mnp, area = w * h;
        if (area < min_area) {
            min_area = area;
            float cn = (mnn + mxn) * 0.5f, cp = (mnp + mxp) * 0.5f;
            best.cx = hix + cn * nx + cp * px;
            best.cy = hiy + cn * ny + cp * py;
            best.w = w; best.h = h;
            best.angle = std::atan2(ny, nx) * 180.0f / kPi;
        }
    }
    return best;
}

std::vector<Pt> simplify_chain(const std::vector<Pt>& p) {
    const size_t n = p.size();
    if (n <= 2) return p;
    auto sgn = [](float v) { return v > 0.0f ? 1 : (v < 0.0f ? -1 : 0); };
    std::vector<Pt> out;
    out.reserve(n);
    const Pt* prev = &p[n - 1];
    for (size_t i = 0; i < n; ++i) {
        const Pt& cur = p[i];
        const Pt& next = p[i + 1 == n ? 0 : i + 1];
        if (sgn(cur.x - prev->x) != sgn(next.x - cur.x) || sgn(cur.y - prev->y) != sgn(next.y - cur.y)) out.push_back(cur);
        prev = &cur;
    }
    if (out.size() < 3) return p;
    return out;
}

bool mini_box(const std::vector<Pt>& pts, Pt out[4], float& min_side) {
    if (pts.size() < 3) return false;
    MinAreaRect r = min_area_rect(pts);
    float ms = r.w < r.h ? r.w : r.h;
    if (!std::isfinite(ms) || ms <= 0.0f) return false;
    float ca = std::cos(r.angle * kPi / 180.0f), sa = std::sin(r.angle * kPi / 180.0f);
    float w2 = r.w / 2.0f, h2 = r.h / 2.0f;
    const float cs[4][2] = {{-w2, -h2}, {w2, -h2}, {w2, h2}, {-w2, h2}};
    Pt raw[4];
    for (int i = 0; i < 4; ++i) {
        raw[i].x = cs[i][0] * ca - cs[i][1] * sa + r.cx;
        raw[i].y = cs[i][0] * sa + cs[i][1] * ca + r.cy;
    }
    std::stable_sort(raw, raw + 4, [](const Pt& a, const Pt& b) { return a.x < b.x; });
    int i1, i4, i2, i3;
    if (raw[1].y > raw[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
    if (raw[3].y > raw[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
    out[0] = raw[i1]; out[1] = raw[i2]; out[2] = raw[i3]; out[3] = raw[i4];
    min_side = ms;
    return true;
}

bool contour_mini_box(const Contour& c, Pt out[4], float& min_side) {
    if (c.simplified) return mini_box(c.pts, out, min_side);   // the tracer already kept simplify_chain's points (>= 3 of them)
    std::vector<Pt> simp = simplify_chain(c.pts);
    return simp.size() >= 3 ? mini_box(simp, out, min_side) : mini_box(c.pts, out, min_side);
}

// ------------------------------------------------------------------------------------------ unclip (Clipper2 offset)
// DBPostProcess::unclip (processors/db_bitmap.rs:279-368) for the one shape the hot path feeds it: the 4 corners of a
// mini box.  Clipper2's InflatePaths(Round, Polygon, precision 2) on a ring of at most 4 vertices, organised around
// the ring's EDGES: each edge contributes its offset vector (unit normal x signed delta, on the 1/100 px integer
// grid); walking the ring, corner i is the pivot of an arc that swings the previous edge's offset vector onto the next
// edge's.  The arithmetic inside an arc (the incremental rotation, the rounding of every emitted vertex to the grid)
// is Clipper2's own, because the vertices -- and through them the hull / min-area rectangle / integer box corners --
// must match the reference to the bit.  What is checked independently of any restatement: tests/test_third_party_pins_cpu.py
// (analytic offsets of axis-aligned and rotated rectangles, arc radius / sagitta / vertex-count bounds).
namespace {
struct Ring4 {
    int n = 0;
    int64_t x[4], y[4];
    double twice_area() const {   // Clipper2 Area(): sum (y_prev + y_cur) * (x_prev - x_cur)
        double s = 0.0;
        for (int i = 0, p = n - 1; i < n; p = i++) s += (double)(y[p] + y[i]) * (double)(x[p] - x[i]);
        return s;
    }
};
struct ArcStepper {   // constant-angle rotation, sized from the offset radius (ClipperOffset::DoGroupOffset)
    double cs, sn, per_rad;
    ArcStepper(double signed_radius) {
        const double r = std::fabs(signed_radius), tol = r * 0.002;   // arc_tolerance 0.0 => radius / 500
        const double per_turn = std::min(kPiD / std::acos(1.0 - tol / r), r * kPiD);
        sn = std::sin(2.0 * kPiD / per_turn);
        cs = std::cos(2.0 * kPiD / per_turn);
        if (signed_radius < 0.0) sn = -sn;
        per_rad = per_turn / (2.0 * kPiD);
    }
};
}  // namespace

std::vector<Pt> unclip(const Pt box[4], float ratio) {
    constexpr double kGrid = 100.0;   // precision 2
    // offset distance from the f64 polygon: area * ratio / perimeter (db_bitmap.rs:297-323)
    double qx[4], qy[4];
    for (int i = 0; i < 4; ++i) { qx[i] = (double)box[i].x; qy[i] = (double)box[i].y; }
    double shoelace = 0.0, perimeter = 0.0;
    for (int i = 0, p = 3; i < 4; p = i++) shoelace += (qy[p] + qy[i]) * (qx[p] - qx[i]);
    const double area = std::fabs(shoelace * 0.5);
    for (int i = 1; i < 4; ++i) perimeter += std::hypot(qx[i] - qx[i - 1], qy[i] - qy[i - 1]);
    perimeter += std::hypot(qx[0] - qx[3], qy[0] - qy[3]);
    if (area <= kEpsD || perimeter <= kEpsD) return {};
    const double delta = area * (double)ratio / perimeter;
    if (std::fabs(delta) <= kEpsD) return {};

    // onto the integer grid, consecutive duplicates (and a duplicated closing vertex) removed
    Ring4 ring;
    for (int i = 0; i < 4; ++i) {
        const int64_t gx = (int64_t)std::round(qx[i] * kGrid), gy = (int64_t)std::round(qy[i] * kGrid);
        if (ring.n && ring.x[ring.n - 1] == gx && ring.y[ring.n - 1] == gy) continue;
        ring.x[ring.n] = gx; ring.y[ring.n] = gy; ++ring.n;
    }
    while (ring.n > 1 && ring.x[ring.n - 1] == ring.x[0] && ring.y[ring.n - 1] == ring.y[0]) --ring.n;
    if (ring.n < 3) return {};

    std::vector<Pt> out;
    out.reserve(96);
    auto emit = [&](double gx, double gy) {   // Point64(double, double) rounds; back to pixels in f64, then f32
        out.push_back({(float)((double)(int64_t)std::round(gx) / kGrid), (float)((double)(int64_t)std::round(gy) / kGrid)});
    };
    const double grid_delta = delta * kGrid;
    if (std::fabs(grid_delta) < 0.5) {   // an offset below half a grid step leaves the ring as it is
        for (int i = 0; i < ring.n; ++i) emit((double)ring.x[i], (double)ring.y[i]);
    } else {
        // a clockwise ring is grown by a negative radius
        const double radius = ring.twice_area() * 0.5 < 0 ? -grid_delta : grid_delta;
        const ArcStepper arc(radius);
        // offset vector of edge e (vertex e -> e + 1): unit normal (dy, -dx) / |edge|, times the radius
        double ux[4], uy[4];
        for (int e = 0; e < ring.n; ++e) {
            const int f = e + 1 == ring.n ? 0 : e + 1;
            double dx = (double)(ring.x[f] - ring.x[e]), dy = (double)(ring.y[f] - ring.y[e]);
            if (dx == 0.0 && dy == 0.0) { ux[e] = uy[e] = 0.0; continue; }
            const double inv_len = 1.0 / std::sqrt(dx * dx + dy * dy);
            dx *= inv_len; dy *= inv_len;
            ux[e] = dy; uy[e] = -dx;
        }
        for (int v = 0, in_e = ring.n - 1; v < ring.n; in_e = v++) {   // corner v sits between edge in_e and edge v
            const double cx = (double)ring.x[v], cy = (double)ring.y[v];
            double turn_sin = uy[v] * ux[in_e] - uy[in_e] * ux[v];
            const double turn_cos = ux[v] * ux[in_e] + uy[v] * uy[in_e];
            turn_sin = turn_sin > 1.0 ? 1.0 : turn_sin < -1.0 ? -1.0 : turn_sin;
            double sx = ux[in_e] * radius, sy = uy[in_e] * radius;   // where the arc starts (relative to the corner)
            const double ex = cx + ux[v] * radius, ey = cy + uy[v] * radius;   // where it ends
            if (turn_cos > -0.999 && turn_sin * radius < 0) {   // reflex corner (never for a mini box): spike through the corner
                emit(cx + sx, cy + sy); emit(cx, cy); emit(ex, ey);
                continue;
            }
            emit(cx + sx, cy + sy);
            const int hops = (int)std::ceil(arc.per_rad * std::fabs(std::atan2(turn_sin, turn_cos)));
            for (int h = 1; h < hops; ++h) {
                const double rx = sx * arc.cs - arc.sn * sy, ry = sx * arc.sn + sy * arc.cs;
                sx = rx; sy = ry;
                emit(cx + sx, cy + sy);
            }
            emit(ex, ey);
        }
    }
    // db_bitmap.rs:355-361: the closing vertex is dropped when it repeats the first
    if (out.size() > 1 && std::fabs(out.front().x - out.back().x) < kEps && std::fabs(out.front().y - out.back().y) < kEps) out.pop_back();
    if (out.size() < 3) out.clear();
    return out;
}

// ------------------------------------------------------------------------------------------ sorting
std::vector<int> sort_quad_boxes(const std::vector<float>& b8) {
    const int n = (int)(b8.size() / 8);
    auto ymin = [&](int i) { float m = INFINITY; for (int k = 0; k < 4; ++k) m = b8[i * 8 + k * 2 + 1] < m ? b8[i * 8 + k * 2 + 1] : m; return m; };
    auto xmin = [&](int i) { float m = INFINITY; for (int k = 0; k < 4; ++k) m = b8[i * 8 + k * 2] < m ? b8[i * 8 + k * 2] : m; return m; };
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        float ay = ymin(a), by = ymin(b);
        if (ay < by) return true;
        if (ay > by) return false;
        if (ay == by) return xmin(a) < xmin(b);
        return false;
    });
    for (int i = 0; i + 1 < n; ++i) {
        for (int j = i; j >= 0; --j) {
            if (j + 1 >= n) break;
            int c = order[j], nx = order[j + 1];
            if (std::fabs(ymin(nx) - ymin(c)) < 10.0f && xmin(nx) < xmin(c)) std::swap(order[j], order[j + 1]);
            else break;
        }
    }
    return order;
}

// ------------------------------------------------------------------------------------------ crop planning
namespace {
// The homography of a crop: 8 unknowns from 4 point pairs, solved the way `nalgebra` 0.35 does it for the reference
// (utils/transform.rs:266-267: `a.lu().solve(&b)`, then Matrix3::try_inverse :312-316).  The pixel each bicubic tap lands on
// depends on the last bit of these nine numbers, so the elimination ORDER is nalgebra's (column-major axpy updates,
// partial pivoting by |value|, unit-diagonal forward substitution, then back substitution); the storage and the driver
// are this file's own: one flat column-major array with the right-hand side as a ninth column, so a row exchange swaps
// one row of the whole tableau, and the pivots are remembered as a permutation instead of a swap log.
// Independent check: tests/test_third_party_pins_cpu.py compares against numpy float64 solve / inv.
struct Tableau8 {
    float v[9][8];   // v[col][row]; column 8 = right-hand side
    float& at(int r, int c) { return v[c][r]; }
};
bool solve_homography8(Tableau8& t, float sol[8]) {
    int row_of[8];   // row_of[i]: which ORIGINAL rhs entry sits in row i after the exchanges
    for (int i = 0; i < 8; ++i) row_of[i] = i;
    float rhs0[8];
    for (int i = 0; i < 8; ++i) rhs0[i] = t.at(i, 8);
    for (int c = 0; c < 8; ++c) {
        int best = c;
        float best_abs = std::fabs(t.at(c, c));
        for (int r = c + 1; r < 8; ++r) { const float a = std::fabs(t.at(r, c)); if (a > best_abs) { best_abs = a; best = r; } }
        const float pivot = t.at(best, c);
        if (pivot == 0.0f) continue;            // an all-zero column below the diagonal: nothing to eliminate
        if (best != c) {
            std::swap(row_of[c], row_of[best]);
            for (int k = 0; k < 8; ++k) std::swap(t.at(c, k), t.at(best, k));
        }
        const float inv_pivot = 1.0f / pivot;
        float* col = t.v[c];
        for (int r = c + 1; r < 8; ++r) col[r] *= inv_pivot;          // multipliers stay in place (the L factor)
        for (int k = c + 1; k < 8; ++k) {                             // column k -= U[c][k] * multipliers
            float* dst = t.v[k];
            const float f = -dst[c];
            for (int r = c + 1; r < 8; ++r) dst[r] = f * col[r] + dst[r];
        }
    }
    for (int i = 0; i < 8; ++i) sol[i] = rhs0[row_of[i]];              // P b
    for (int c = 0; c < 7; ++c) {                                      // L y = P b, unit diagonal
        const float f = -(sol[c] / 1.0f);
        for (int r = c + 1; r < 8; ++r) sol[r] = f * t.at(r, c) + sol[r];
    }
    for (int c = 7; c >= 0; --c) {                                     // U x = y
        const float d = t.at(c, c);
        if (d == 0.0f) return false;
        const float x = sol[c] / d;
        sol[c] = x;
        const float f = -x;
        for (int r = 0; r < c; ++r) sol[r] = f * t.at(r, c) + sol[r];
    }
    return true;
}
// adjugate / determinant inverse of a row-major 3x3, nalgebra's cofactor grouping (Matrix3::try_inverse)
bool invert3(const float m[9], float out[9]) {
    const float c00 = m[4] * m[8] - m[7] * m[5], c01 = m[3] * m[8] - m[6] * m[5], c02 = m[3] * m[7] - m[6] * m[4];
    const float det = m[0] * c00 - m[1] * c01 + m[2] * c02;
    if (det == 0.0f) return false;
    const float adj[9] = {c00, m[2] * m[7] - m[8] * m[1], m[1] * m[5] - m[4] * m[2],
                          -c01, m[0] * m[8] - m[6] * m[2], m[2] * m[3] - m[5] * m[0],
                          c02, m[1] * m[6] - m[7] * m[0], m[0] * m[4] - m[3] * m[1]};
    for (int i = 0; i < 9; ++i) out[i] = adj[i] / det;
    return true;
}
}  // namespace

CropPlan plan_crop(int img_w, int img_h, const float box8[8]) {
    CropPlan pl;
    float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int i = 0; i < 4; ++i) {
        mnx = std::fmin(mnx, box8[i * 2]); mxx = std::fmax(mxx, box8[i * 2]);
        mny = std::fmin(mny, box8[i * 2 + 1]); mxy = std::fmax(mxy, box8[i * 2 + 1]);
    }
    uint32_t left = sat_u32(std::fmax(mnx, 0.0f)), top = sat_u32(std::fmax(mny, 0.0f));
    uint32_t right = sat_u32(std::fmin(mxx, (float)img_w)), bottom = sat_u32(std::fmin(mxy, (float)img_h));
    if (right <= left || bottom <= top) return pl;
    uint32_t cw = right - left, ch = bottom - top;
    Pt s[4];
    for (int i = 0; i < 4; ++i) s[i] = {box8[i * 2] - (float)left, box8[i * 2 + 1] - (float)top};
    std::stable_sort(s, s + 4, [](const Pt& a, const Pt& b) { return a.x < b.x; });
    int ia = 0, id = 1, ib = 2, ic = 3;
    if (s[1].y < s[0].y) { ia = 1; id = 0; }
    if (s[3].y < s[2].y) { ib = 3; ic = 2; }
    Pt o[4] = {s[ia], s[ib], s[ic], s[id]};
    pl.left = (int)left; pl.top = (int)top; pl.cw = (int)cw; pl.ch = (int)ch;
    float fw = (float)cw, fh = (float)ch;
    if (o[0].x == 0.0f && o[0].y == 0.0f && o[1].x == fw && o[1].y == 0.0f && o[2].x == fw && o[2].y == fh && o[3].x == 0.0f && o[3].y == fh) {
        pl.mode = 1; pl.ow = (int)cw; pl.oh = (int)ch;
        pl.rot = (float)ch >= (float)cw * 1.5f ? 1 : 0;
        return pl;
    }
    float w1 = std::hypot(o[0].x - o[1].x, o[0].y - o[1].y), w2 = std::hypot(o[2].x - o[3].x, o[2].y - o[3].y);
    uint32_t ow = sat_u32(std::round(std::fmax(w1, w2)));
    float h1 = std::hypot(o[0].x - o[3].x, o[0].y - o[3].y), h2 = std::hypot(o[1].x - o[2].x, o[1].y - o[2].y);
    uint32_t oh = sat_u32(std::round(std::fmax(h1, h2)));
    if (ow == 0 || oh == 0) return pl;
    const Pt dst[4] = {{0.0f, 0.0f}, {(float)ow, 0.0f}, {(float)ow, (float)oh}, {0.0f, (float)oh}};
    // two equations per correspondence (utils/transform.rs:230-262): rows 2i / 2i+1 of the 8 x 8 system
    Tableau8 t;
    for (int i = 0; i < 4; ++i) {
        const float sx = o[i].x, sy = o[i].y, dx = dst[i].x, dy = dst[i].y;
        const float eq_x[9] = {sx, sy, 1.0f, 0.0f, 0.0f, 0.0f, -sx * dx, -sy * dx, dx};
        const float eq_y[9] = {0.0f, 0.0f, 0.0f, sx, sy, 1.0f, -sx * dy, -sy * dy, dy};
        for (int c = 0; c < 9; ++c) { t.at(2 * i, c) = eq_x[c]; t.at(2 * i + 1, c) = eq_y[c]; }
    }
    float hcoef[8];
    if (!solve_homography8(t, hcoef)) return pl;
    const float m[9] = {hcoef[0], hcoef[1], hcoef[2], hcoef[3], hcoef[4], hcoef[5], hcoef[6], hcoef[7], 1.0f};
    if (!invert3(m, pl.inv)) return pl;
    pl.mode = 2; pl.ow = (int)ow; pl.oh = (int)oh;
    pl.rot = (float)oh >= (float)ow * 1.5f ? 1 : 0;
    return pl;
}

bool det_resize_dims(uint32_t w, uint32_t h, uint32_t limit_side_len, int limit_type, uint32_t max_side_limit, uint32_t& rh, uint32_t& rw) {
    uint32_t mx = std::max(h, w), mn = std::min(h, w);
    float ratio;
    if (limit_type == 0) ratio = mx > limit_side_len ? (float)limit_side_len / (float)mx : 1.0f;
    else if (limit_type == 1) ratio = mn < limit_side_len ? (float)limit_side_len / (float)mn : 1.0f;
    else ratio = (float)limit_side_len / (float)mx;
    rh = sat_u32((float)h * ratio); rw = sat_u32((float)w * ratio);
    uint32_t rmx = std::max(rh, rw);
    if (rmx > max_side_limit) {
        float lr = (float)max_side_limit / (float)rmx;
        rh = sat_u32((float)rh * lr); rw = sat_u32((float)rw * lr);
    }
    rh = std::max((rh + 16) / 32 * 32, 32u);
    rw = std::max((rw + 16) / 32 * 32, 32u);
    return !(rh == h && rw == w);
}

int rec_tensor_width(const std::vector<uint32_t>& ws, const std::vector<uint32_t>& hs, int img_h, int img_w, int max_img_w, std::vector<int32_t>& resized_w) {
    float max_wh = (float)img_w / (float)std::max(img_h, 1);
    for (size_t i = 0; i < ws.size(); ++i) {
        float r = (float)ws[i] / (float)std::max<uint32_t>(hs[i], 1);
        if (r > max_wh) max_wh = r;
    }
    uint32_t tw = std::min<uint32_t>(sat_u32((float)img_h * max_wh), (uint32_t)max_img_w);
    resized_w.resize(ws.size());
    for (size_t i = 0; i < ws.size(); ++i) {
        float ratio = (float)ws[i] / (float)hs[i];
        uint32_t rw = std::min<uint32_t>(sat_u32(std::ceil((float)img_h * ratio)), tw);
        resized_w[i] = (int32_t)rw;
    }
    return (int)tw;
}

}  // namespace host
}  // namespace oar
