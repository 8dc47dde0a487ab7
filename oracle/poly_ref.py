"""Oracle for the POLYGON (seal text) branch of DB post-processing -- SURVEY 8f-2.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under oar_ocr_amd/).  A CPU restatement of

    DBPostProcess::polygons_from_bitmap   oar-ocr-core/src/processors/db_bitmap.rs:16-82
    BoundingBox::approx_poly_dp           oar-ocr-core/src/processors/geometry.rs:453-561
    BoundingBox::perimeter                oar-ocr-core/src/processors/geometry.rs:161-171
    DBPostProcess::unclip                 oar-ocr-core/src/processors/db_bitmap.rs:279-368
    sort_poly_boxes                       oar-ocr-core/src/processors/sorting.rs:100-118
    BBoxCrop::crop_bounding_box           oar-ocr-core/src/utils/bbox_crop.rs:26-72

`unclip` calls clipper2-rust 1.0.3 `inflate_paths_d(Round, Polygon, miter 2, precision 2, arc tolerance 0)`, a dependency that
is not vendored in the reference tree.  Its published algorithm is restated here in two steps:

  * the raw offset ring (ClipperOffset::{BuildNormals, OffsetPoint, DoRound}) in Python floats (= C doubles, libm calls);
  * the closing Union(Positive): the outline of the area the raw ring winds around at least once.  Restated BY DEFINITION, not
    by any sweep: every segment is split at its exact (rational) crossings, the winding number on both sides of every piece is
    counted by exact ray casting against all other segments, the pieces with winding 0 | 1 are chained by their exact end
    points.  (The product, oar_ocr_amd/csrc/poly_host.cc, propagates windings along the ring and switches segments at crossings:
    a different algorithm for the same set.)

PARITY IS UNPINNED for the outline step against Clipper2 itself: no golden vectors for concave offsets exist in the reference's
tests and there is no Rust toolchain to run it.  What pins it instead is analytic (tests/test_poly_cpu.py): every outline vertex
lies at the offset distance from the input polygon, no input point is nearer than that to the outline, the area matches the
Minkowski sum.  The START vertex of the returned path follows a reading of Clipper2's sweep (closing vertex = last vertex of the
top-most run in input direction); the reference's consumers are independent of it (bounding box and min y only).
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from . import cpu_ref

F32 = np.float32


# ------------------------------------------------------------------------------------------ geometry.rs
def perimeter(pts: np.ndarray) -> np.float32:
    """geometry.rs:161-171: closed ring, f32 accumulation in vertex order."""
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    per = F32(0.0)
    n = len(pts)
    for i in range(n):
        j = (i + 1) % n
        dx = F32(pts[j, 0] - pts[i, 0])
        dy = F32(pts[j, 1] - pts[i, 1])
        per = F32(per + np.sqrt(F32(F32(dx * dx) + F32(dy * dy))))
    return per


def _line_distances(p: np.ndarray, s: np.ndarray, e: np.ndarray) -> np.ndarray:
    """geometry.rs:550-561 for an array of points (every operation rounds to f32, none is fused)."""
    a = F32(e[1] - s[1])
    b = F32(s[0] - e[0])
    c = F32(F32(e[0] * s[1]) - F32(s[0] * e[1]))
    den = np.sqrt(F32(F32(a * a) + F32(b * b)))
    if den == 0:
        return np.zeros(len(p), np.float32)
    t = (a * p[:, 0]).astype(np.float32) + (b * p[:, 1]).astype(np.float32)
    t = (t + c).astype(np.float32)
    return (np.abs(t) / den).astype(np.float32)


def approx_poly_dp(pts: np.ndarray, epsilon) -> np.ndarray:
    """geometry.rs:453-537: Douglas-Peucker over the OPEN chain pts[0] .. pts[n-1], explicit stack, 10 000-pop guard."""
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    if n <= 2:
        return pts.copy()
    epsilon = F32(epsilon)
    keep = np.zeros(n, bool)
    keep[0] = keep[-1] = True
    stack = [(0, n - 1)]
    pops = 0
    while stack:
        start, end = stack.pop()
        pops += 1
        if pops > 10000:
            keep[start:end + 1] = True
            break
        if end - start <= 1:
            continue
        d = _line_distances(pts[start + 1:end], pts[start], pts[end])
        k = int(np.argmax(d))                       # first of the maxima, like `dist > max_dist`
        if d[k] > 0 and d[k] > epsilon:
            mi = start + 1 + k
            keep[mi] = True
            if mi - start > 1:
                stack.append((start, mi))
            if end - mi > 1:
                stack.append((mi, end))
    return pts[keep].copy()


# ------------------------------------------------------------------------------------------ Clipper2 raw offset ring
def _round_half_away(v: float) -> int:
    """Point64(double, double): std::round / f64::round."""
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def strip_repeats(ring):
    out = []
    for p in ring:
        if not out or out[-1] != p:
            out.append(p)
    while len(out) > 1 and out[-1] == out[0]:
        out.pop()
    return out


def twice_area(ring) -> float:
    """Clipper2 Area(): sum (y_prev + y_cur) * (x_prev - x_cur), in doubles."""
    s = 0.0
    n = len(ring)
    for i in range(n):
        p = ring[i - 1]
        c = ring[i]
        s += float(p[1] + c[1]) * float(p[0] - c[0])
    return s


def offset_ring(ring, radius: float):
    """ClipperOffset::OffsetPolygon with JoinType::Round on grid coordinates; radius < 0 for a clockwise ring."""
    n = len(ring)
    r = abs(radius)
    tol = r * 0.002
    per_turn = min(math.pi / math.acos(1.0 - tol / r), r * math.pi)
    sn = math.sin(2.0 * math.pi / per_turn)
    cs = math.cos(2.0 * math.pi / per_turn)
    if radius < 0:
        sn = -sn
    per_rad = per_turn / (2.0 * math.pi)
    normals = []
    for e in range(n):
        a, b = ring[e], ring[(e + 1) % n]
        dx, dy = float(b[0] - a[0]), float(b[1] - a[1])
        if dx == 0 and dy == 0:
            normals.append((0.0, 0.0))
            continue
        inv = 1.0 / math.sqrt(dx * dx + dy * dy)
        dx *= inv
        dy *= inv
        normals.append((dy, -dx))
    out = []
    emit = lambda x, y: out.append((_round_half_away(x), _round_half_away(y)))
    for v in range(n):
        k = v - 1 if v else n - 1
        nk, nv = normals[k], normals[v]
        cx, cy = float(ring[v][0]), float(ring[v][1])
        sin_a = nv[1] * nk[0] - nk[1] * nv[0]
        cos_a = nv[0] * nk[0] + nv[1] * nk[1]
        sin_a = 1.0 if sin_a > 1.0 else -1.0 if sin_a < -1.0 else sin_a
        ox, oy = nk[0] * radius, nk[1] * radius
        if cos_a > -0.999 and sin_a * radius < 0:      # concave: three points, the middle one is the vertex itself
            emit(cx + ox, cy + oy)
            emit(cx, cy)
            emit(cx + nv[0] * radius, cy + nv[1] * radius)
            continue
        emit(cx + ox, cy + oy)
        steps = int(math.ceil(per_rad * abs(math.atan2(sin_a, cos_a))))
        for _ in range(1, steps):
            ox, oy = ox * cs - sn * oy, ox * sn + oy * cs
            emit(cx + ox, cy + oy)
        emit(cx + nv[0] * radius, cy + nv[1] * radius)
    return out


# ------------------------------------------------------------------------------------------ the union, by definition
class Degenerate(Exception):
    """two segments touch without crossing, overlap, or three pass through one point"""


def _orient(a, b, c) -> int:
    v = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    return (v > 0) - (v < 0)


def _on_segment(a, b, p) -> bool:
    """p collinear with a-b: inside the closed segment?"""
    return min(a[0], b[0]) <= p[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= p[1] <= max(a[1], b[1])


def _share_a_point(a, b, c, d) -> bool:
    o1, o2, o3, o4 = _orient(a, b, c), _orient(a, b, d), _orient(c, d, a), _orient(c, d, b)
    if o1 * o2 < 0 and o3 * o4 < 0:
        return True
    return (o1 == 0 and _on_segment(a, b, c)) or (o2 == 0 and _on_segment(a, b, d)) or (o3 == 0 and _on_segment(c, d, a)) or (o4 == 0 and _on_segment(c, d, b))


def outline_positive(ring):
    """All loops of the boundary of {winding >= 1} of a counter-clockwise-positive closed integer ring; each loop a list of exact
    (Fraction, Fraction) vertices in ring direction, starting anywhere.  Raises Degenerate on exact touches."""
    n = len(ring)
    if n < 3:
        return []
    seg = [(ring[i], ring[(i + 1) % n]) for i in range(n)]
    for i in range(n):                                   # neighbours folding back onto each other
        a, b = seg[i]
        c = seg[(i + 1) % n][1]
        if _orient(a, b, c) == 0 and (b[0] - a[0]) * (c[0] - b[0]) + (b[1] - a[1]) * (c[1] - b[1]) < 0:
            raise Degenerate("fold-back")
    cuts = [[] for _ in range(n)]                        # parameters of the crossings on each segment
    for i in range(n):
        a, b = seg[i]
        for j in range(i + 1, n):
            if j == i + 1 or (i == 0 and j == n - 1):
                continue
            c, d = seg[j]
            if not _share_a_point(a, b, c, d):
                continue
            if not (_orient(a, b, c) * _orient(a, b, d) < 0 and _orient(c, d, a) * _orient(c, d, b) < 0):
                raise Degenerate("touch")
            den = (b[0] - a[0]) * (d[1] - c[1]) - (b[1] - a[1]) * (d[0] - c[0])
            ti = Fraction((c[0] - a[0]) * (d[1] - c[1]) - (c[1] - a[1]) * (d[0] - c[0]), den)
            tj = Fraction((c[0] - a[0]) * (b[1] - a[1]) - (c[1] - a[1]) * (b[0] - a[0]), den)
            cuts[i].append(ti)
            cuts[j].append(tj)
    pieces = []                                          # (segment, start point, end point)
    for i in range(n):
        ts = sorted(cuts[i])
        if any(ts[k] == ts[k + 1] for k in range(len(ts) - 1)):
            raise Degenerate("three segments through one point")
        a, b = seg[i]
        at = lambda t: (a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]))
        stops = [Fraction(0)] + ts + [Fraction(1)]
        for k in range(len(stops) - 1):
            pieces.append((i, at(stops[k]), at(stops[k + 1])))

    def others(i, mx, my):                               # winding of every segment but i around (mx, my + tiny), ray to +x
        w = 0
        for k in range(n):
            if k == i:
                continue
            a, b = seg[k]
            if a[1] <= my < b[1]:
                if a[0] + Fraction((my - a[1]) * (b[0] - a[0]), (b[1] - a[1])) > mx:
                    w += 1
            elif b[1] <= my < a[1]:
                if a[0] + Fraction((my - a[1]) * (b[0] - a[0]), (b[1] - a[1])) > mx:
                    w -= 1
        return w

    on_outline = {}
    for (i, p, q) in pieces:
        mx, my = (p[0] + q[0]) / 2, (p[1] + q[1]) / 2
        a, b = seg[i]
        dx, dy = b[0] - a[0], b[1] - a[1]
        base = others(i, mx, my)
        if dy > 0:        # the ray from the left side meets the segment itself (going up: +1); the right side is at larger x
            left, right = base + 1, base
        elif dy < 0:      # the right side is at smaller x and meets it going down (-1)
            left, right = base, base - 1
        elif dx > 0:      # horizontal: `base` was counted just above the segment; above = left when it points to +x
            left, right = base, base - 1
        else:
            left, right = base + 1, base
        assert left == right + 1
        if right == 0:
            if p in on_outline:
                raise Degenerate("two outline pieces leave one point")
            on_outline[p] = q
    loops = []
    while on_outline:
        start = next(iter(on_outline))
        loop, p = [], start
        while p in on_outline:
            q = on_outline.pop(p)
            loop.append(q)
            p = q
        if p != start:
            raise Degenerate("open outline")
        loops.append(loop)
    return loops


def _jitter(ring, attempt):
    mul = ((7, 5), (11, 13), (17, 19))[attempt]
    return [(x * 4 + (i * mul[0] + 3) % 3 - 1, y * 4 + (i * mul[1] + 1) % 3 - 1) for i, (x, y) in enumerate(ring)]


def _closing_vertex(loop, negative):
    n = len(loop)
    top = min(p[1] for p in loop)
    best = -1
    for i in range(n):
        if loop[i][1] != top:
            continue
        after = loop[i - 1] if negative else loop[(i + 1) % n]
        if after[1] == top:
            continue
        if best < 0 or loop[i][0] > loop[best][0]:
            best = i
    return max(best, 0)


def _clean_loop(loop, mark):
    """Clipper2 CleanCollinear with preserve_collinear = false; returns the loop rotated so that the (possibly moved) mark is first."""
    n = len(loop)
    nx = [(i + 1) % n for i in range(n)]
    pv = [(i - 1) % n for i in range(n)]
    alive, cur, start = n, mark, mark
    while True:
        p, c, q = loop[pv[cur]], loop[cur], loop[nx[cur]]
        if (c[0] - p[0]) * (q[1] - p[1]) - (c[1] - p[1]) * (q[0] - p[0]) == 0:
            if cur == mark:
                mark = pv[cur]
            after = nx[cur]
            nx[pv[cur]] = after
            pv[after] = pv[cur]
            alive -= 1
            if alive < 3:
                return None
            cur = start = after
            continue
        cur = nx[cur]
        if cur == start:
            break
    out, i = [], mark
    for _ in range(alive):
        out.append(loop[i])
        i = nx[i]
    return out


def ring_outline(raw, negative: bool):
    """What Clipper2's closing Union leaves of the raw offset ring: one path on the grid, or None when it is not exactly one."""
    raw = strip_repeats(list(raw))
    if negative:
        raw = raw[::-1]
    loop = None
    try:
        loops = outline_positive(raw)
        if len(loops) != 1:
            return None
        loop = [(round(x), round(y)) for x, y in loops[0]]     # Fraction -> nearest int, ties to even
    except Degenerate:
        for attempt in range(3):
            try:
                loops = outline_positive(_jitter(raw, attempt))
            except Degenerate:
                continue
            if len(loops) != 1:
                return None
            loop = [(round(Fraction(round(x), 4)), round(Fraction(round(y), 4))) for x, y in loops[0]]
            break
        if loop is None:
            return None
    loop = _clean_loop(loop, _closing_vertex(loop, negative))
    if loop is None:
        return None
    m = len(loop)
    if not negative:
        return [loop[i % m] for i in range(1, m + 1)]
    return [loop[0]] + [loop[i] for i in range(m - 1, 0, -1)]


# ------------------------------------------------------------------------------------------ db_bitmap.rs
def unclip_poly(poly: np.ndarray, ratio: float) -> np.ndarray:
    """db_bitmap.rs:279-368 for any polygon; an empty array where the reference returns an empty box."""
    poly = np.asarray(poly, np.float32).reshape(-1, 2)
    n = len(poly)
    if n < 3:
        return poly.copy()
    q = [(float(x), float(y)) for x, y in poly]
    shoelace = 0.0
    for i in range(n):
        p, c = q[i - 1], q[i]
        shoelace += (p[1] + c[1]) * (p[0] - c[0])
    area = abs(shoelace * 0.5)
    eps = 2.220446049250313e-16
    if area <= eps:
        return np.zeros((0, 2), np.float32)
    per = 0.0
    for i in range(1, n):
        per += math.hypot(q[i][0] - q[i - 1][0], q[i][1] - q[i - 1][1])
    per += math.hypot(q[0][0] - q[-1][0], q[0][1] - q[-1][1])
    if per <= eps:
        return np.zeros((0, 2), np.float32)
    delta = area * float(np.float32(ratio)) / per
    if abs(delta) <= eps:
        return np.zeros((0, 2), np.float32)
    ring = strip_repeats([(_round_half_away(x * 100.0), _round_half_away(y * 100.0)) for x, y in q])
    if len(ring) < 3:
        return np.zeros((0, 2), np.float32)
    gd = delta * 100.0
    if abs(gd) < 0.5:
        path = ring
    else:
        negative = twice_area(ring) * 0.5 < 0
        path = ring_outline(offset_ring(ring, -gd if negative else gd), negative)
        if path is None:
            return np.zeros((0, 2), np.float32)
    out = np.array([[np.float32(x / 100.0), np.float32(y / 100.0)] for x, y in path], np.float32).reshape(-1, 2)
    f32eps = np.float32(1.1920929e-7)
    if len(out) > 1 and abs(out[0, 0] - out[-1, 0]) < f32eps and abs(out[0, 1] - out[-1, 1]) < f32eps:
        out = out[:-1]
    if len(out) < 3:
        return np.zeros((0, 2), np.float32)
    return out


def polygons_from_bitmap(pred, mask, dest_w, dest_h, box_thresh=0.6, unclip_ratio=0.5, max_candidates=1000, min_size=3.0):
    """db_bitmap.rs:16-82.  Returns (list of [n_i, 2] f32 polygons in destination coordinates, list of scores)."""
    pred = np.ascontiguousarray(pred, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    wscale = F32(F32(dest_w) / F32(w))
    hscale = F32(F32(dest_h) / F32(h))
    polys, scores = [], []
    for pts, _btype, _parent in cpu_ref.find_contours(mask)[:max_candidates]:
        if len(pts) < 4:
            continue
        chain = pts.astype(np.float32)
        epsilon = F32(F32(0.002) * perimeter(chain))
        approx = approx_poly_dp(chain, epsilon)
        if len(approx) < 4:
            continue
        score = np.float32(cpu_ref.box_score_fast(pred, approx))
        if score < F32(box_thresh):
            continue
        un = unclip_poly(approx, unclip_ratio)
        if len(un) == 0:
            continue
        mb = cpu_ref.mini_box(un)
        if mb is None:
            continue
        if F32(mb[1]) < F32(F32(min_size) + F32(2.0)):
            continue
        x = np.clip(_round_half_away_f32((un[:, 0] * wscale).astype(np.float32)), F32(0), F32(dest_w))
        y = np.clip(_round_half_away_f32((un[:, 1] * hscale).astype(np.float32)), F32(0), F32(dest_h))
        polys.append(np.stack([x, y], 1).astype(np.float32))
        scores.append(float(score))
    return polys, scores


def _round_half_away_f32(v: np.ndarray) -> np.ndarray:
    """f32::round"""
    v = np.asarray(v, np.float32)
    return (np.sign(v) * np.floor(np.abs(v) + F32(0.5))).astype(np.float32)


def db_postprocess_poly(pred, src_h, src_w, thresh=0.2, box_thresh=0.6, unclip_ratio=0.5, max_candidates=1000, use_dilation=False):
    """processors/db_postprocess.rs:134-183 with BoxType::Poly."""
    mask = cpu_ref.threshold_mask(pred, thresh)
    if use_dilation:
        mask = cpu_ref.dilate3x3(mask)
    return polygons_from_bitmap(pred, mask, int(src_w), int(src_h), box_thresh, unclip_ratio, max_candidates)


def sort_poly_boxes(polys):
    """sorting.rs:100-118: stable sort by min y; returns the permutation."""
    ymin = [float(np.min(p[:, 1])) if len(p) else 0.0 for p in polys]
    return sorted(range(len(polys)), key=lambda i: ymin[i])


def crop_bounding_box(img: np.ndarray, poly: np.ndarray):
    """utils/bbox_crop.rs:26-72: axis-aligned crop of a polygon's bounding rectangle; None where the reference errs."""
    poly = np.asarray(poly, np.float32).reshape(-1, 2)
    if len(poly) == 0:
        return None
    h, w = img.shape[:2]
    min_x = max(float(poly[:, 0].min()), 0.0)
    max_x = float(poly[:, 0].max())
    min_y = max(float(poly[:, 1].min()), 0.0)
    max_y = float(poly[:, 1].max())
    as_u32 = lambda v: 0 if v != v or v <= 0 else min(int(v), 0xFFFFFFFF)     # `as u32` saturates, truncates
    x1 = min(as_u32(min_x), max(w - 1, 0))
    y1 = min(as_u32(min_y), max(h - 1, 0))
    x2 = min(as_u32(max_x), w)
    y2 = min(as_u32(max_y), h)
    if x2 <= x1 or y2 <= y1:
        return None
    return img[y1:y2, x1:x2].copy()
