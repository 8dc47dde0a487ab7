// poly_host.cc -- host geometry of the POLYGON (seal text) branch of DB post-processing:
//   approx_poly_dp   processors/geometry.rs:453-561   (Douglas-Peucker on the open contour chain)
//   unclip_poly      processors/db_bitmap.rs:279-368  (Clipper2 inflate_paths_d, Round join, precision 2) for ANY polygon
//   sort_poly_boxes  processors/sorting.rs:100-118
//
// unclip_poly = Clipper2's raw round-join offset ring (the same arithmetic as the mini-box `unclip` in db_host.cc, for n
// vertices, reflex corners included) followed by what Clipper2's closing Union(Positive) does to that ring: keep the outline
// of the area the ring winds around at least once.  Clipper2 gets there with a Vatti sweep; this file gets there with the
// ring's own crossings:
//   1. all proper crossings between the ring's segments, exactly (integer orientation tests, rational parameters);
//   2. winding numbers by PROPAGATION: the side windings of the segment leaving the lexicographically lowest vertex are known
//      (one side is the unbounded face); walking the ring, they change by +-1 at every crossing, by the crossing's handedness;
//   3. the outline follows the ring and SWITCHES to the other segment at every crossing it meets (with winding 0 | 1 on the two
//      sides of the current piece, the continuation of the same segment is never on the outline, the crossing one always is);
//   4. crossing points are rounded to the 1/100 px grid, repeated and collinear vertices dropped (Clipper2 does not preserve
//      collinear vertices in an offset), and the loop is rotated to end at the vertex where Clipper2's sweep closes the polygon.
// More than one outline loop (a hole, or islands) means inflate_paths_d would return != 1 paths: the reference drops the box,
// so does this.  Exact touches (a vertex on another segment, overlapping collinear pieces, three segments through a point) are
// resolved by re-running on a 4x finer grid with a fixed +-1 jitter per vertex and rounding back.
//
// Parity statement: every vertex is Clipper2's (raw ring arithmetic restated, crossings to the nearest grid
// point); the cyclic order is the ring's; the START vertex is derived from reading Clipper2's sweep, not from running it
// (no Rust toolchain here) -- consumers use the polygon's bounding box and min y only (bbox_crop.rs:26-72, sorting.rs:100).
#include "db_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace oar {
namespace host {

namespace {
constexpr double kPiD = 3.14159265358979323846;
constexpr double kEpsD = 2.220446049250313e-16;
constexpr float kEps = 1.1920929e-7f;

// ------------------------------------------------------------------------------------------ Douglas-Peucker
inline float line_distance(const Pt& p, const Pt& s, const Pt& e) {   // geometry.rs:550-561
    const float a = e.y - s.y, b = s.x - e.x, c = e.x * s.y - s.x * e.y;
    const float den = std::sqrt(a * a + b * b);
    if (den == 0.0f) return 0.0f;
    return std::fabs(a * p.x + b * p.y + c) / den;
}
}  // namespace

float perimeter(const std::vector<Pt>& pts) {   // geometry.rs:161-171 (closed ring, f32 accumulation)
    float per = 0.0f;
    const size_t n = pts.size();
    for (size_t i = 0; i < n; ++i) {
        const size_t j = i + 1 == n ? 0 : i + 1;
        const float dx = pts[j].x - pts[i].x, dy = pts[j].y - pts[i].y;
        per += std::sqrt(dx * dx + dy * dy);
    }
    return per;
}

std::vector<Pt> approx_poly_dp(const std::vector<Pt>& pts, float epsilon) {
    const size_t n = pts.size();
    if (n <= 2) return pts;
    std::vector<uint8_t> keep(n, 0);
    keep[0] = keep[n - 1] = 1;
    std::vector<std::pair<size_t, size_t>> stack;
    stack.push_back({0, n - 1});
    size_t iterations = 0;
    while (!stack.empty()) {
        const auto [start, end] = stack.back();
        stack.pop_back();
        if (++iterations > 10000) {   // the reference's guard: keep the whole span it was about to look at, then stop
            for (size_t i = start; i <= end; ++i) keep[i] = 1;
            break;
        }
        if (end - start <= 1) continue;
        float max_dist = 0.0f;
        size_t max_index = start;
        for (size_t i = start + 1; i < end; ++i) {
            const float d = line_distance(pts[i], pts[start], pts[end]);
            if (d > max_dist) { max_dist = d; max_index = i; }
        }
        if (max_dist > epsilon) {
            keep[max_index] = 1;
            if (max_index - start > 1) stack.push_back({start, max_index});
            if (end - max_index > 1) stack.push_back({max_index, end});
        }
    }
    std::vector<Pt> out;
    for (size_t i = 0; i < n; ++i) if (keep[i]) out.push_back(pts[i]);
    return out;
}

std::vector<int> sort_poly_boxes(const std::vector<float>& pts_xy, const std::vector<uint32_t>& offsets) {
    const int n = offsets.empty() ? 0 : (int)offsets.size() - 1;
    std::vector<float> ymin(n);
    for (int i = 0; i < n; ++i) {
        float m = INFINITY;
        if (offsets[i] == offsets[i + 1]) m = 0.0f;   // BoundingBox::y_min of an empty box (geometry.rs:198-209)
        for (uint32_t k = offsets[i]; k < offsets[i + 1]; ++k) if (pts_xy[k * 2 + 1] < m) m = pts_xy[k * 2 + 1];
        ymin[i] = m;
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ymin[a] < ymin[b]; });   // sort_by is stable
    return order;
}

// ------------------------------------------------------------------------------------------ outline of a self-crossing ring
namespace {
struct I2 { int64_t x, y; };
inline bool operator==(const I2& a, const I2& b) { return a.x == b.x && a.y == b.y; }
inline int64_t cross3(const I2& a, const I2& b, const I2& c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }
inline int sign64(int64_t v) { return (v > 0) - (v < 0); }
typedef __int128 i128;

struct Crossing {
    i128 num, den;   // parameter along the own segment, den > 0, 0 < num < den
    int other;       // the crossing segment
    int twin;        // index of the same crossing in the other segment's list
    int step;        // change of the right-hand winding when passing it: +1 if the other segment runs right-to-left ... see below
};

enum OutlineStatus { kOutlineOne, kOutlineNotOne, kOutlineDegenerate };

// nearest integer of a / b (b > 0), ties to even
inline int64_t round_div(i128 a, i128 b) {
    i128 q = a / b, r = a % b;
    if (r < 0) { r += b; q -= 1; }            // floor division
    const i128 twice = r * 2;
    if (twice > b || (twice == b && (q & 1))) q += 1;
    return (int64_t)q;
}

// ring: closed, no two consecutive vertices equal, positive (counter-clockwise in x/y axes) sense = the sense whose inside
// has winding +1.  out: the outline of {winding >= 1}, oriented like the ring, rotated to END at the closing vertex.
OutlineStatus outline_positive(const std::vector<I2>& ring, std::vector<I2>& out) {
    const int n = (int)ring.size();
    out.clear();
    if (n < 3) return kOutlineNotOne;
    auto nxt = [&](int i) { return i + 1 == n ? 0 : i + 1; };
    // neighbours must not fold back onto each other
    for (int i = 0; i < n; ++i) {
        const I2 &a = ring[i], &b = ring[nxt(i)], &c = ring[nxt(nxt(i))];
        if (cross3(a, b, c) == 0 && ((b.x - a.x) * (c.x - b.x) + (b.y - a.y) * (c.y - b.y)) < 0) return kOutlineDegenerate;
    }
    std::vector<std::vector<Crossing>> cr(n);
    std::vector<int64_t> lox(n), hix(n), loy(n), hiy(n);
    for (int i = 0; i < n; ++i) {
        const I2 &a = ring[i], &b = ring[nxt(i)];
        lox[i] = std::min(a.x, b.x); hix[i] = std::max(a.x, b.x); loy[i] = std::min(a.y, b.y); hiy[i] = std::max(a.y, b.y);
    }
    for (int i = 0; i < n; ++i) {
        const I2 &a = ring[i], &b = ring[nxt(i)];
        for (int j = i + 1; j < n; ++j) {
            if (hix[j] < lox[i] || hix[i] < lox[j] || hiy[j] < loy[i] || hiy[i] < loy[j]) continue;
            const bool adjacent = j == i + 1 || (i == 0 && j == n - 1);
            if (adjacent) continue;   // share exactly their common vertex (fold-backs were rejected above)
            const I2 &c = ring[j], &d = ring[nxt(j)];
            const int o1 = sign64(cross3(a, b, c)), o2 = sign64(cross3(a, b, d)), o3 = sign64(cross3(c, d, a)), o4 = sign64(cross3(c, d, b));
            if (o1 * o2 > 0 || o3 * o4 > 0) continue;                 // one segment entirely on one side of the other's line
            if (o1 * o2 < 0 && o3 * o4 < 0) {                          // proper crossing
                const int64_t dix = b.x - a.x, diy = b.y - a.y, djx = d.x - c.x, djy = d.y - c.y;
                i128 den = (i128)dix * djy - (i128)diy * djx;          // d_i x d_j
                i128 ti = (i128)(c.x - a.x) * djy - (i128)(c.y - a.y) * djx;
                i128 tj = (i128)(c.x - a.x) * diy - (i128)(c.y - a.y) * dix;
                // moving along i across j: the right-hand winding grows by one when i passes from j's right to j's left,
                // i.e. when d_j x d_i > 0
                const int step_i = den < 0 ? +1 : -1;
                if (den < 0) { den = -den; ti = -ti; tj = -tj; }
                Crossing ci{ti, den, j, (int)cr[j].size(), step_i};
                Crossing cj{tj, den, i, (int)cr[i].size(), -step_i};
                cr[i].push_back(ci); cr[j].push_back(cj);
                continue;
            }
            // some orientation is zero and the segments are not separated by a line: they touch or overlap unless the collinear
            // pieces are disjoint (the box test above already failed to separate them)
            if (o1 == 0 && o2 == 0 && o3 == 0 && o4 == 0) return kOutlineDegenerate;   // collinear with overlapping boxes
            return kOutlineDegenerate;                                                    // an end point lies on the other segment
        }
    }
    // order the crossings of every segment; `twin` indices follow the permutation
    for (int i = 0; i < n; ++i) {
        auto& v = cr[i];
        if (v.size() < 2) continue;
        std::vector<int> perm(v.size());
        for (size_t k = 0; k < v.size(); ++k) perm[k] = (int)k;
        std::sort(perm.begin(), perm.end(), [&](int p, int q) { return v[p].num * v[q].den < v[q].num * v[p].den; });
        for (size_t k = 0; k + 1 < perm.size(); ++k)
            if (v[perm[k]].num * v[perm[k + 1]].den == v[perm[k + 1]].num * v[perm[k]].den) return kOutlineDegenerate;   // three segments, one point
        std::vector<Crossing> sorted(v.size());
        for (size_t k = 0; k < perm.size(); ++k) { sorted[k] = v[perm[k]]; cr[sorted[k].other][sorted[k].twin].twin = (int)k; }
        v.swap(sorted);
    }
    // pieces: segment i has cr[i].size() + 1 pieces; piece (i, k) runs from event k - 1 (the vertex for k = 0) to event k (the next
    // vertex for k = size).  Right-hand windings by propagation from the lexicographically lowest vertex.
    int low = 0;
    for (int i = 1; i < n; ++i) if (ring[i].y < ring[low].y || (ring[i].y == ring[low].y && ring[i].x < ring[low].x)) low = i;
    const int before = low == 0 ? n - 1 : low - 1;
    const int64_t turn = cross3(ring[before], ring[low], ring[nxt(low)]);
    if (turn == 0) return kOutlineDegenerate;
    std::vector<int> first_piece(n + 1, 0);
    for (int i = 0; i < n; ++i) first_piece[i + 1] = first_piece[i] + (int)cr[i].size() + 1;
    std::vector<int> wright(first_piece[n], 0);
    {
        int w = turn > 0 ? 0 : -1;   // a left turn at the lowest vertex: outside (winding 0) on the right; a right turn: outside on the left
        for (int s = 0, i = low; s < n; ++s, i = nxt(i)) {
            wright[first_piece[i]] = w;
            for (size_t k = 0; k < cr[i].size(); ++k) { w += cr[i][k].step; wright[first_piece[i] + (int)k + 1] = w; }
        }
        if (w != (turn > 0 ? 0 : -1)) return kOutlineDegenerate;   // cannot happen for a closed ring with consistent crossings
    }
    // outline pieces: winding 0 on the right (and 1 on the left).  Follow them, switching segments at every crossing.
    std::vector<uint8_t> seen(first_piece[n], 0);
    int loops = 0;
    std::vector<I2> loop;
    for (int i0 = 0; i0 < n; ++i0) {
        for (int k0 = 0; k0 <= (int)cr[i0].size(); ++k0) {
            if (wright[first_piece[i0] + k0] != 0 || seen[first_piece[i0] + k0]) continue;
            if (++loops > 1) return kOutlineNotOne;
            int i = i0, k = k0;
            for (;;) {
                const int id = first_piece[i] + k;
                if (seen[id]) break;
                if (wright[id] != 0) return kOutlineDegenerate;   // the switching rule left the outline: inconsistent input
                seen[id] = 1;
                if (k == (int)cr[i].size()) {           // the piece ends at the segment's end vertex
                    i = nxt(i); k = 0;
                    loop.push_back(ring[i]);
                } else {                                 // the piece ends at a crossing: emit it, continue on the other segment
                    const Crossing& c = cr[i][k];
                    const I2 &a = ring[i], &b = ring[nxt(i)];
                    loop.push_back({round_div((i128)a.x * c.den + c.num * (b.x - a.x), c.den), round_div((i128)a.y * c.den + c.num * (b.y - a.y), c.den)});
                    const int j = c.other, kj = c.twin;
                    i = j; k = kj + 1;
                }
            }
        }
    }
    if (loops != 1) return kOutlineNotOne;
    out.swap(loop);
    return kOutlineOne;
}

// Clipper2 ClipperBase::CleanCollinear + BuildPath64 on the finished loop: drop repeated and collinear vertices (the mark moves
// to the predecessor when its vertex goes), `mark` = index of the vertex the sweep closed the polygon at.
bool clean_loop(std::vector<I2>& loop, int& mark) {
    // linked ring over the vertex array
    int n = (int)loop.size();
    if (n < 3) return false;
    std::vector<int> nx(n), pv(n);
    for (int i = 0; i < n; ++i) { nx[i] = i + 1 == n ? 0 : i + 1; pv[i] = i == 0 ? n - 1 : i - 1; }
    int alive = n, cur = mark, start = mark;
    for (;;) {
        const I2 &p = loop[pv[cur]], &c = loop[cur], &q = loop[nx[cur]];
        if (cross3(p, c, q) == 0) {   // repeated points are collinear with anything
            if (cur == mark) mark = pv[cur];
            const int after = nx[cur];
            nx[pv[cur]] = after; pv[after] = pv[cur];
            if (--alive < 3) return false;
            cur = after; start = cur;
            continue;
        }
        cur = nx[cur];
        if (cur == start) break;
    }
    std::vector<I2> kept;
    kept.reserve(alive);
    int m2 = 0;
    for (int i = mark, c = 0; c < alive; ++c, i = nx[i]) kept.push_back(loop[i]);
    loop.swap(kept);
    mark = m2;   // the mark is now vertex 0
    return true;
}

// The vertex Clipper2's bottom-up sweep finishes the outer polygon at: on the top-most row (smallest y), the last vertex of the
// run in the direction the INPUT ring runs (AddPathsToVertexList flags the vertex after which the ring first descends as the
// local maximum); of several such runs, the right-most one.  `loop` is positive; `negative`: the input ran the other way.
int closing_vertex(const std::vector<I2>& loop, bool negative) {
    const int n = (int)loop.size();
    int64_t top = loop[0].y;
    for (const I2& p : loop) top = std::min(top, p.y);
    int best = -1;
    for (int i = 0; i < n; ++i) {
        if (loop[i].y != top) continue;
        const I2& after = negative ? loop[i == 0 ? n - 1 : i - 1] : loop[i + 1 == n ? 0 : i + 1];
        if (after.y == top) continue;   // the run goes on
        if (best < 0 || loop[i].x > loop[best].x) best = i;
    }
    if (best < 0) best = 0;   // every vertex on one row: not a polygon, the caller's area test removes it
    return best;
}

// Clipper2 raw offset ring of a closed polygon on the integer grid (ClipperOffset::{BuildNormals, OffsetPoint, DoRound}, Round
// join, arc tolerance = radius / 500): same arithmetic as the mini-box version in db_host.cc, any number of vertices.
void offset_ring(const std::vector<I2>& ring, double radius, std::vector<I2>& out) {
    const int n = (int)ring.size();
    const double r = std::fabs(radius), tol = r * 0.002;
    const double per_turn = std::min(kPiD / std::acos(1.0 - tol / r), r * kPiD);
    double sn = std::sin(2.0 * kPiD / per_turn);
    const double cs = std::cos(2.0 * kPiD / per_turn);
    if (radius < 0.0) sn = -sn;
    const double per_rad = per_turn / (2.0 * kPiD);
    std::vector<double> ux(n), uy(n);
    for (int e = 0; e < n; ++e) {
        const int f = e + 1 == n ? 0 : e + 1;
        double dx = (double)(ring[f].x - ring[e].x), dy = (double)(ring[f].y - ring[e].y);
        if (dx == 0.0 && dy == 0.0) { ux[e] = uy[e] = 0.0; continue; }
        const double inv_len = 1.0 / std::sqrt(dx * dx + dy * dy);
        dx *= inv_len; dy *= inv_len;
        ux[e] = dy; uy[e] = -dx;
    }
    auto emit = [&](double gx, double gy) { out.push_back({(int64_t)std::round(gx), (int64_t)std::round(gy)}); };
    for (int v = 0, in_e = n - 1; v < n; in_e = v++) {
        const double cx = (double)ring[v].x, cy = (double)ring[v].y;
        double turn_sin = uy[v] * ux[in_e] - uy[in_e] * ux[v];
        const double turn_cos = ux[v] * ux[in_e] + uy[v] * uy[in_e];
        turn_sin = turn_sin > 1.0 ? 1.0 : turn_sin < -1.0 ? -1.0 : turn_sin;
        double sx = ux[in_e] * radius, sy = uy[in_e] * radius;
        const double ex = cx + ux[v] * radius, ey = cy + uy[v] * radius;
        if (turn_cos > -0.999 && turn_sin * radius < 0) {   // reflex corner: out along the old normal, through the corner, out along the new one
            emit(cx + sx, cy + sy); emit(cx, cy); emit(ex, ey);
            continue;
        }
        emit(cx + sx, cy + sy);
        const int hops = (int)std::ceil(per_rad * std::fabs(std::atan2(turn_sin, turn_cos)));
        for (int h = 1; h < hops; ++h) {
            const double rx = sx * cs - sn * sy, ry = sx * sn + sy * cs;
            sx = rx; sy = ry;
            emit(cx + sx, cy + sy);
        }
        emit(ex, ey);
    }
}

void strip_repeats(std::vector<I2>& ring) {
    std::vector<I2> r;
    for (const I2& p : ring) if (r.empty() || !(r.back() == p)) r.push_back(p);
    while (r.size() > 1 && r.back() == r.front()) r.pop_back();
    ring.swap(r);
}

double twice_area(const std::vector<I2>& ring) {   // Clipper2 Area()
    double s = 0.0;
    const int n = (int)ring.size();
    for (int i = 0, p = n - 1; i < n; p = i++) s += (double)(ring[p].y + ring[i].y) * (double)(ring[p].x - ring[i].x);
    return s;
}

// outline of the raw ring as Clipper2's Union(Positive / Negative) leaves it; `negative`: the input polygon ran clockwise
OutlineStatus ring_outline(std::vector<I2> raw, bool negative, std::vector<I2>& out) {
    strip_repeats(raw);
    if (negative) std::reverse(raw.begin(), raw.end());   // the same area as a positive ring
    std::vector<I2> loop;
    OutlineStatus st = outline_positive(raw, loop);
    for (int attempt = 0; st == kOutlineDegenerate && attempt < 3; ++attempt) {
        // exact touches: 4x finer grid, a fixed jitter of one fine step per vertex, outline there, round back
        static const int mul[3][2] = {{7, 5}, {11, 13}, {17, 19}};
        std::vector<I2> fine(raw.size());
        for (size_t i = 0; i < raw.size(); ++i)
            fine[i] = {raw[i].x * 4 + (int64_t)((i * mul[attempt][0] + 3) % 3) - 1, raw[i].y * 4 + (int64_t)((i * mul[attempt][1] + 1) % 3) - 1};
        std::vector<I2> fl;
        st = outline_positive(fine, fl);
        if (st == kOutlineOne) {
            loop.clear();
            for (const I2& p : fl) loop.push_back({round_div(p.x, 4), round_div(p.y, 4)});
        }
    }
    if (st != kOutlineOne) return st == kOutlineDegenerate ? kOutlineNotOne : st;
    int mark = closing_vertex(loop, negative);
    if (!clean_loop(loop, mark)) return kOutlineNotOne;
    // loop[0] is the closing vertex.  Positive input: the path starts right after it and ends with it; negative input
    // (ReverseSolution): the path starts with it and runs the other way round.
    out.clear();
    const int m = (int)loop.size();
    if (!negative) { for (int i = 1; i <= m; ++i) out.push_back(loop[i % m]); }
    else { out.push_back(loop[0]); for (int i = m - 1; i >= 1; --i) out.push_back(loop[i]); }
    return kOutlineOne;
}
}  // namespace

std::vector<Pt> unclip_poly(const std::vector<Pt>& poly, float ratio) {
    constexpr double kGrid = 100.0;   // precision 2
    const int np = (int)poly.size();
    if (np < 3) return poly;           // db_bitmap.rs:280-282
    std::vector<double> qx(np), qy(np);
    for (int i = 0; i < np; ++i) { qx[i] = (double)poly[i].x; qy[i] = (double)poly[i].y; }
    double shoelace = 0.0, per = 0.0;
    for (int i = 0, p = np - 1; i < np; p = i++) shoelace += (qy[p] + qy[i]) * (qx[p] - qx[i]);
    const double area = std::fabs(shoelace * 0.5);
    if (area <= kEpsD) return {};
    for (int i = 1; i < np; ++i) per += std::hypot(qx[i] - qx[i - 1], qy[i] - qy[i - 1]);
    per += std::hypot(qx[0] - qx[np - 1], qy[0] - qy[np - 1]);
    if (per <= kEpsD) return {};
    const double delta = area * (double)ratio / per;
    if (std::fabs(delta) <= kEpsD) return {};

    std::vector<I2> ring;
    ring.reserve(np);
    for (int i = 0; i < np; ++i) ring.push_back({(int64_t)std::round(qx[i] * kGrid), (int64_t)std::round(qy[i] * kGrid)});
    strip_repeats(ring);
    if (ring.size() < 3) return {};
    std::vector<Pt> out;
    auto to_pixels = [&](const std::vector<I2>& g) {
        for (const I2& p : g) out.push_back({(float)((double)p.x / kGrid), (float)((double)p.y / kGrid)});
    };
    const double grid_delta = delta * kGrid;
    if (std::fabs(grid_delta) < 0.5) {
        to_pixels(ring);               // ClipperOffset::Execute copies the paths for a sub-half-step offset
    } else {
        const bool negative = twice_area(ring) * 0.5 < 0;
        std::vector<I2> raw, outline;
        offset_ring(ring, negative ? -grid_delta : grid_delta, raw);
        if (ring_outline(std::move(raw), negative, outline) != kOutlineOne) return {};   // offset_paths.len() != 1 (db_bitmap.rs:341)
        to_pixels(outline);
    }
    if (out.size() > 1 && std::fabs(out.front().x - out.back().x) < kEps && std::fabs(out.front().y - out.back().y) < kEps) out.pop_back();
    if (out.size() < 3) out.clear();
    return out;
}

CropPlan plan_bbox_crop(int img_w, int img_h, const float* pts_xy, int n_points) {
    CropPlan pl;
    if (n_points <= 0 || img_w <= 0 || img_h <= 0) return pl;   // "Empty bounding box"
    // fold(INFINITY, f32::min) / fold(NEG_INFINITY, f32::max): NaN coordinates are skipped by min / max
    float min_x = INFINITY, max_x = -INFINITY, min_y = INFINITY, max_y = -INFINITY;
    for (int i = 0; i < n_points; ++i) {
        min_x = std::fmin(min_x, pts_xy[i * 2]); max_x = std::fmax(max_x, pts_xy[i * 2]);
        min_y = std::fmin(min_y, pts_xy[i * 2 + 1]); max_y = std::fmax(max_y, pts_xy[i * 2 + 1]);
    }
    min_x = std::fmax(min_x, 0.0f); min_y = std::fmax(min_y, 0.0f);
    auto as_u32 = [](float v) -> uint32_t {   // Rust `as u32`: truncates, saturates, NaN -> 0
        if (!(v > 0.0f)) return 0u;
        if (v >= 4294967296.0f) return 0xFFFFFFFFu;
        return (uint32_t)v;
    };
    const uint32_t W = (uint32_t)img_w, H = (uint32_t)img_h;
    const uint32_t x1 = std::min(as_u32(min_x), W - 1), y1 = std::min(as_u32(min_y), H - 1);
    const uint32_t x2 = std::min(as_u32(max_x), W), y2 = std::min(as_u32(max_y), H);
    if (x2 <= x1 || y2 <= y1) return pl;   // "Invalid crop region"
    pl.mode = 1; pl.left = (int)x1; pl.top = (int)y1; pl.cw = (int)(x2 - x1); pl.ch = (int)(y2 - y1); pl.ow = pl.cw; pl.oh = pl.ch; pl.rot = 0;
    return pl;
}

// test hook: the outline step alone, on grid coordinates (x0, y0, x1, y1, ...); returns the number of vertices written, 0 for
// "not exactly one loop", -1 when out_cap is too small
int ring_outline_for_tests(const int64_t* xy, int n, int negative, int64_t* out_xy, int out_cap) {
    std::vector<I2> raw(n), outl;
    for (int i = 0; i < n; ++i) raw[i] = {xy[i * 2], xy[i * 2 + 1]};
    if (ring_outline(std::move(raw), negative != 0, outl) != kOutlineOne) return 0;
    if ((int)outl.size() > out_cap) return -1;
    for (size_t i = 0; i < outl.size(); ++i) { out_xy[i * 2] = outl[i].x; out_xy[i * 2 + 1] = outl[i].y; }
    return (int)outl.size();
}

// test hook: the raw offset ring on grid coordinates
int offset_ring_for_tests(const int64_t* xy, int n, double radius, int64_t* out_xy, int out_cap) {
    std::vector<I2> ring(n), raw;
    for (int i = 0; i < n; ++i) ring[i] = {xy[i * 2], xy[i * 2 + 1]};
    offset_ring(ring, radius, raw);
    if ((int)raw.size() > out_cap) return -1;
    for (size_t i = 0; i < raw.size(); ++i) { out_xy[i * 2] = raw[i].x; out_xy[i * 2 + 1] = raw[i].y; }
    return (int)raw.size();
}

}  // namespace host
}  // namespace oar
