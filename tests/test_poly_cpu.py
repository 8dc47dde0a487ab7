"""SURVEY 8f-2 -- polygon (seal text) branch of DB post-processing, host half (no GPU):

  approx_poly_dp / perimeter   geometry.rs:453-561, 161-171    product (poly_host.cc) == oracle (oracle/poly_ref.py), bit for bit
  unclip of any polygon        db_bitmap.rs:279-368            product == oracle (two different algorithms for Clipper2's closing
                                                               union) AND analytic pins that depend on neither
  sort_poly_boxes              sorting.rs:100-118              incl. the reference's own test vector (sorting.rs:786-799)

The oracle's outline step is parity-unpinned against Clipper2 itself (see its header); the analytic tests below are what pins it.
"""
import math

import numpy as np
import pytest

from oar_ocr_amd import api
from oracle import poly_ref


# ------------------------------------------------------------------------------------------------ shapes
def arc_band(cx, cy, r0, r1, a0, a1, n):
    outer = [(cx + r1 * math.cos(a), cy + r1 * math.sin(a)) for a in np.linspace(a0, a1, n)]
    inner = [(cx + r0 * math.cos(a), cy + r0 * math.sin(a)) for a in np.linspace(a1, a0, n)]
    return np.array(outer + inner, np.float32)


SHAPES = {
    "quad": np.array([[10, 10], [200, 10], [200, 60], [10, 60]], np.float32),
    "rotated": np.array([[50.3, 20.7], [180.2, 60.1], [160.9, 120.4], [30.5, 80.8]], np.float32),
    "L": np.array([[10, 10], [200, 10], [200, 60], [80, 60], [80, 180], [10, 180]], np.float32),
    "U": np.array([[10, 10], [60, 10], [60, 150], [140, 150], [140, 10], [190, 10], [190, 200], [10, 200]], np.float32),
    "seal_arc": arc_band(300, 300, 150, 200, 0.2, 2.6, 14),
    "star": np.array([[100 + (80 if i % 2 == 0 else 30) * math.cos(i * math.pi / 5), 100 + (80 if i % 2 == 0 else 30) * math.sin(i * math.pi / 5)]
                      for i in range(10)], np.float32),
}


def seg_dist(p, a, b):
    ab = b - a
    t = np.clip(np.dot(p - a, ab) / max(np.dot(ab, ab), 1e-30), 0.0, 1.0)
    return float(np.linalg.norm(p - (a + t * ab)))


def poly_dist(p, poly):
    return min(seg_dist(p, poly[i], poly[(i + 1) % len(poly)]) for i in range(len(poly)))


def shoelace(poly):
    x, y = poly[:, 0].astype(np.float64), poly[:, 1].astype(np.float64)
    return 0.5 * (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def offset_distance(poly, ratio):
    p = poly.astype(np.float64)
    per = sum(math.hypot(*(p[(i + 1) % len(p)] - p[i])) for i in range(len(p)))
    return abs(shoelace(poly)) * ratio / per


def cyclic_equal(a, b):
    a, b = [tuple(p) for p in a], [tuple(p) for p in b]
    if len(a) != len(b):
        return False
    return any(a == b[k:] + b[:k] for k in range(len(b))) if a else True


# ------------------------------------------------------------------------------------------------ Douglas-Peucker
def test_perimeter_matches_the_oracle_and_closed_form():
    sq = np.array([[0, 0], [3, 0], [3, 4], [0, 4]], np.float32)
    assert api.host_perimeter(sq) == 14.0
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = rng.uniform(0, 500, (int(rng.integers(3, 60)), 2)).astype(np.float32)
        assert np.float32(api.host_perimeter(p)) == poly_ref.perimeter(p)


def test_approx_poly_dp_known_answers():
    # a straight chain collapses to its ends; a corner farther than epsilon survives (the chain is OPEN: ends are always kept)
    line = np.array([[x, 0] for x in range(11)], np.float32)
    assert api.host_approx_poly_dp(line, 0.5).tolist() == [[0, 0], [10, 0]]
    bent = np.array([[0, 0], [5, 0.4], [10, 0], [10, 5], [10, 10]], np.float32)
    assert api.host_approx_poly_dp(bent, 0.5).tolist() == [[0, 0], [10, 0], [10, 10]]
    assert api.host_approx_poly_dp(bent, 0.3).tolist() == [[0, 0], [5, 0.4000000059604645], [10, 0], [10, 10]]
    two = np.array([[1, 2], [3, 4]], np.float32)
    assert api.host_approx_poly_dp(two, 1.0).tolist() == two.tolist()       # <= 2 points: returned as they are


def test_approx_poly_dp_matches_the_oracle_on_contour_like_chains():
    rng = np.random.default_rng(1)
    for it in range(40):
        n = int(rng.integers(4, 1500))
        a = np.linspace(0, 2 * math.pi, n, endpoint=False)
        r = 100 + 30 * np.sin(a * rng.integers(1, 6)) + rng.normal(0, 0.7, n)
        chain = np.round(np.stack([300 + r * np.cos(a), 300 + r * np.sin(a)], 1)).astype(np.float32)   # integer pixels, like a border chain
        eps = np.float32(0.002) * poly_ref.perimeter(chain) if it % 4 else np.float32(rng.uniform(0.2, 3.0))
        got = api.host_approx_poly_dp(chain, float(eps))
        ref = poly_ref.approx_poly_dp(chain, eps)
        assert got.shape == ref.shape and np.array_equal(got, ref), it


# ------------------------------------------------------------------------------------------------ unclip: product == oracle
@pytest.mark.parametrize("name", sorted(SHAPES))
@pytest.mark.parametrize("reverse", [False, True])
def test_unclip_poly_matches_the_oracle(name, reverse):
    poly = SHAPES[name][::-1].copy() if reverse else SHAPES[name]
    for ratio in (0.5, 1.5, 3.0):
        ref = poly_ref.unclip_poly(poly, ratio)
        got = api.host_unclip_poly(poly, ratio)
        assert len(ref) >= 3
        assert got.shape == ref.shape and np.array_equal(got, ref), (name, reverse, ratio)


def test_unclip_poly_matches_the_oracle_on_random_star_polygons():
    rng = np.random.default_rng(5)
    for it in range(60):
        k = int(rng.integers(5, 40))
        ang = np.sort(rng.uniform(0, 2 * math.pi, k))
        rad = rng.uniform(20, 120, k)
        pts = np.stack([200 + rad * np.cos(ang), 200 + rad * np.sin(ang)], 1)
        if it % 3 == 0:
            pts = pts.round()              # integer vertices: exact touches on the grid become likely
        pts = pts.astype(np.float32)
        if it % 2:
            pts = pts[::-1].copy()
        ratio = float(rng.choice([0.5, 1.0, 1.5, 2.5]))
        ref = poly_ref.unclip_poly(pts, ratio)
        got = api.host_unclip_poly(pts, ratio)
        assert got.shape == ref.shape and np.array_equal(got, ref), it


def test_unclip_of_a_quad_agrees_with_the_mini_box_path():
    """For a convex quad the union changes nothing but the start vertex and the collinear vertices: the vertex SET of the polygon
    path contains no point the quad path (oar_host_unclip, pinned in test_third_party_pins_cpu.py) does not have."""
    for name in ("quad", "rotated"):
        q = SHAPES[name]
        a = {tuple(p) for p in api.host_unclip(q.reshape(8), 1.5).tolist()}
        b = [tuple(p) for p in api.host_unclip_poly(q, 1.5).tolist()]
        assert len(b) >= 20 and set(b) <= a
        assert len(a) - len(set(b)) <= 8      # at most the collinear middle vertices of the four straight sides go


# ------------------------------------------------------------------------------------------------ unclip: analytic pins
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_unclip_poly_vertices_lie_at_the_offset_distance(name):
    """Minkowski sum with a disc: every outline vertex is exactly `delta` from the input polygon (arc chords and crossing points
    to the 1/100 px grid), the input lies strictly inside, and the area is A + P * delta + (pi - reflex correction) * delta^2."""
    poly = SHAPES[name]
    for ratio in (0.5, 1.5):
        d = offset_distance(poly, ratio)
        out = api.host_unclip_poly(poly, ratio).astype(np.float64)
        assert len(out) >= 8
        dist = np.array([poly_dist(p, poly.astype(np.float64)) for p in out])
        assert np.abs(dist - d).max() <= 0.002 * d + 0.02, (name, ratio, float(np.abs(dist - d).max()))
        mids = 0.5 * (out + np.roll(out, -1, 0))
        dm = np.array([poly_dist(p, poly.astype(np.float64)) for p in mids])
        assert dm.min() >= d * (1 - 0.0021) - 0.02           # chords sag by at most the arc tolerance delta / 500
        assert dm.max() <= d + 0.02
        # same orientation as the input, and an area between the polygon's and the disc-sum bound for a convex shape
        assert np.sign(shoelace(out)) == np.sign(shoelace(poly))
        per = sum(math.hypot(*(poly.astype(np.float64)[(i + 1) % len(poly)] - poly.astype(np.float64)[i])) for i in range(len(poly)))
        assert abs(shoelace(poly)) < abs(shoelace(out)) <= abs(shoelace(poly)) + per * d + math.pi * d * d + 0.02 * (per + 2 * math.pi * d)   # grid rounding: 0.01 px along the outline


def test_unclip_of_an_axis_aligned_L_has_the_reflex_corner_where_geometry_puts_it():
    """The reflex corner (80, 60) of the L moves to (80 + d, 60 + d): the crossing of the two offset edges, which only the
    closing union produces (the raw ring passes through (80, 60) itself there)."""
    poly = SHAPES["L"]
    d = offset_distance(poly, 1.0)
    out = api.host_unclip_poly(poly, 1.0)
    near = np.abs(out - np.array([80 + d, 60 + d], np.float32)).max(1)
    assert near.min() <= 0.011, near.min()
    assert not any((p == [80.0, 60.0]).all() for p in out)      # the spike vertex of the raw ring is gone
    # convex corners are arcs: about 1/4 of the 50-per-turn steps each
    assert 60 <= len(out) <= 90


def test_unclip_with_a_hole_is_dropped():
    """A ring with a 4 px slit: the offset closes the slit and leaves a hole -> Clipper2 returns two paths -> db_bitmap.rs:341
    returns an empty box."""
    ring = np.array([[0, 0], [120, 0], [120, 120], [0, 120], [0, 64], [100, 64], [100, 20], [20, 20], [20, 100], [100, 100], [100, 68], [0, 68]],
                    np.float32) + 10
    for ratio in (0.2, 0.5, 1.5):
        assert len(api.host_unclip_poly(ring, ratio)) == 0
        assert len(poly_ref.unclip_poly(ring, ratio)) == 0


def test_degenerate_polygons_are_dropped_like_the_reference():
    assert len(api.host_unclip_poly(np.array([[0, 0], [10, 0], [20, 0]], np.float32), 1.5)) == 0      # zero area
    two = np.array([[0, 0], [10, 0]], np.float32)
    assert api.host_unclip_poly(two, 1.5).tolist() == two.tolist()                                      # < 3 points: returned as is


# ------------------------------------------------------------------------------------------------ the outline step alone
def test_ring_outline_handles_exact_touches():
    # a vertex exactly on another segment, and a figure whose crossing falls on a vertex: both go through the jitter path
    raw = [(0, 0), (1000, 0), (1000, 1000), (500, 1000), (500, 0), (400, -300), (0, -300)]
    got = api.host_ring_outline(np.array(raw, np.int64))
    ref = poly_ref.ring_outline(raw, False)
    assert got is not None and [tuple(p) for p in got.tolist()] == ref
    raw2 = [(0, 0), (400, 0), (400, 400), (200, 400), (200, 200), (600, 200), (600, 600), (0, 600)]
    got2 = api.host_ring_outline(np.array(raw2, np.int64))
    assert [tuple(p) for p in got2.tolist()] == poly_ref.ring_outline(raw2, False)
    assert cyclic_equal(got2.tolist(), [(400, 200), (600, 200), (600, 600), (0, 600), (0, 0), (400, 0)])


def test_ring_outline_of_a_figure_eight_keeps_the_positive_lobe_only():
    # lobe A (0..100) runs counter-clockwise (winding +1), lobe B runs clockwise (winding -1): Union(Positive) keeps A
    raw = [(0, 0), (100, 0), (200, 100), (200, 0), (100, 100), (0, 100)]
    got = api.host_ring_outline(np.array(raw, np.int64))
    assert got is not None
    assert [tuple(p) for p in got.tolist()] == poly_ref.ring_outline(raw, False)
    assert cyclic_equal(got.tolist(), [(0, 0), (100, 0), (150, 50), (100, 100), (0, 100)])


def test_offset_ring_matches_the_oracle():
    ring = [(int(x * 100), int(y * 100)) for x, y in SHAPES["U"]]
    for radius in (350.0, -350.0, 1234.5):
        got = api.host_offset_ring(np.array(ring, np.int64), radius)
        assert [tuple(p) for p in got.tolist()] == poly_ref.offset_ring(ring, radius)


# ------------------------------------------------------------------------------------------------ sorting
def test_sort_poly_boxes_reference_vector_and_stability():
    # sorting.rs:786-799: three boxes at y = 50, 10, 30 -> order by min y
    def box(x, y):
        return np.array([[x, y], [x + 10, y], [x + 10, y + 10], [x, y + 10]], np.float32)
    polys = [box(10, 50), box(10, 10), box(10, 30)]
    assert api.host_sort_poly_boxes(polys).tolist() == [1, 2, 0]
    # equal keys keep their input order (sort_by is stable); polygons of different sizes
    polys = [np.array([[0, 5], [9, 7], [3, 30]], np.float32), SHAPES["seal_arc"], np.array([[50, 5], [60, 5], [60, 9], [50, 9], [49, 7]], np.float32)]
    assert api.host_sort_poly_boxes(polys).tolist() == poly_ref.sort_poly_boxes(polys) == [0, 2, 1]
    assert api.host_sort_poly_boxes([]).tolist() == []
