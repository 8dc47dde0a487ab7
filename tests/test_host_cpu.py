"""CPU parity of the host-side geometry in libOarMi355x (csrc/db_host.cc) against the oracle -- no GPU involved.
Bit-exact: both sides are float code over the same glibc libm."""
import numpy as np
import pytest
from scipy.ndimage import uniform_filter

from oar_ocr_amd import api, build
from oar_ocr_amd.synth import pages
from oracle import cpu_ref as R


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_lib()


def blob_mask(seed, size=(480, 640), lines=10):
    pg = pages.make_page(seed, size, lines)
    d = 1.0 - pg[:, :, 0].astype(np.float32) / 255.0
    b = uniform_filter(uniform_filter(d, 9), 9)
    rng = np.random.default_rng(seed)
    m = b > 0.25
    m ^= rng.random(m.shape) < 0.002           # speckle: isolated pixels, pinholes (hole borders)
    return (m * 255).astype(np.uint8)


def oracle_candidates(mask, max_candidates=1000):
    out = []
    for pts, _, _ in R.find_contours(mask)[:max_candidates]:
        p = pts.astype(np.float32)
        s = R.simplify_chain(p)
        mb = R.mini_box(s if len(s) >= 3 else p)
        if mb is None or mb[1] < 3.0:
            continue
        out.append(mb[0])
    return np.stack(out) if out else np.zeros((0, 4, 2), np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("bands", [1, 4, 8])
def test_candidates_match_oracle_and_bands_are_exact(seed, bands):
    mask = blob_mask(seed)
    got = api.host_candidates(mask, 1000, bands)
    ref = oracle_candidates(mask)
    assert got.shape == ref.shape and len(ref) > 5
    assert np.array_equal(got, ref)


def test_max_candidates_truncates_in_discovery_order():
    mask = blob_mask(3)
    full = R.find_contours(mask)
    k = max(3, len(full) // 3)
    assert np.array_equal(api.host_candidates(mask, k, 8), oracle_candidates(mask, k))


def test_edge_masks():
    z = np.zeros((64, 96), np.uint8)
    assert len(api.host_candidates(z, 100, 4)) == 0
    f = np.full((64, 96), 255, np.uint8)                      # all foreground: one border touching every edge
    assert np.array_equal(api.host_candidates(f, 100, 4), oracle_candidates(f))
    one = z.copy(); one[10, 10] = 255                          # single pixel: 1-point contour, rejected (< 3 points)
    assert len(api.host_candidates(one, 100, 1)) == 0


def test_unclip_minibox_sort_crop_plan_match_oracle():
    rng = np.random.default_rng(0)
    boxes = []
    for _ in range(60):
        cx, cy = rng.uniform(50, 900), rng.uniform(50, 900)
        w, h, a = rng.uniform(6, 400), rng.uniform(4, 60), rng.uniform(-0.3, 0.3)
        c, s = np.cos(a), np.sin(a)
        pts = (np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]).astype(np.float32)
        boxes.append(pts)
    for b in boxes:
        for ratio in (1.4, 1.5, 2.0):
            u_got, u_ref = api.host_unclip(b, ratio), R.unclip(b, ratio)
            assert np.array_equal(u_got, u_ref)
            if len(u_ref) >= 3:
                g, r = api.host_mini_box(u_got), R.mini_box(u_ref)
                assert (g is None) == (r is None)
                if r is not None:
                    assert np.array_equal(g[0], r[0]) and g[1] == r[1]
    rb = np.round(np.stack(boxes))
    assert np.array_equal(api.host_sort_quad_boxes(rb), R.sort_quad_boxes(rb))
    for b in rb:
        plan, inv = api.host_plan_crop(960, 960, b)
        oplan = np.zeros(7, np.int32); oinv = np.zeros(9, np.float32)
        mode = R.lib().orc_crop_plan(960, 960, R._p(np.ascontiguousarray(b, np.float32)), R._p(oplan), R._p(oinv))
        assert plan[0] == mode
        if mode:
            assert plan[1:5].tolist() == oplan[0:4].tolist() and plan[5:8].tolist() == oplan[4:7].tolist()
            if mode == 2:
                assert np.array_equal(inv, oinv)
    # degenerate boxes are dropped, not crashed on
    assert len(api.host_unclip(np.zeros((4, 2), np.float32), 1.5)) == 0
    assert api.host_plan_crop(100, 100, np.array([[5, 5]] * 4, np.float32))[0][0] == 0


def test_thread_pool_back_to_back_jobs():
    """Every index of every loop runs exactly once, in its own loop -- also with more workers than cores, where a worker
    is regularly descheduled between reading a job's descriptor and claiming its first index."""
    import ctypes as C
    L = api.lib()
    L.oar_host_pool_selftest.argtypes = [C.c_int32, C.c_int32]
    L.oar_host_pool_selftest.restype = C.c_int32
    for threads in (2, 7, 16, 48):
        assert L.oar_host_pool_selftest(threads, 20000) == 0


# ---------------------------------------------------------------------------------------------- round 5: bit-plane border follower
def _stress_masks():
    rng = np.random.default_rng(55)
    out = []
    for w in (1, 2, 7, 8, 9, 62, 63, 64, 65, 66, 126, 127, 128, 129, 190, 200):       # word / byte boundaries of the framed bit plane
        for dens in (0.15, 0.5, 0.85):
            out.append(((rng.random((1 + int(rng.integers(1, 40)), w)) < dens) * 255).astype(np.uint8))
    for _ in range(12):                                                                  # blobs with long straight edges (run skipping)
        h, w = int(rng.integers(20, 80)), int(rng.integers(100, 400))
        m = np.zeros((h, w), np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            y0, x0 = int(rng.integers(0, h - 3)), int(rng.integers(0, w - 10))
            m[y0:y0 + int(rng.integers(1, 12)), x0:x0 + int(rng.integers(3, 300))] = 255
        m ^= ((rng.random(m.shape) < 0.01) * 255).astype(np.uint8)
        out.append(m)
    for _ in range(8):                                                                   # one-pixel strokes in all 8 directions
        h, w = 60, 130
        m = np.zeros((h, w), np.uint8)
        for _ in range(10):
            x, y, dx, dy = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(-1, 2)), int(rng.integers(-1, 2))
            for i in range(int(rng.integers(2, 60))):
                xx, yy = x + i * dx, y + i * dy
                if 0 <= xx < w and 0 <= yy < h:
                    m[yy, xx] = 255
        out.append(m)
    return out


def test_bit_plane_follower_matches_oracle_chain_for_chain():
    """find_contours_band_bits (word-level run ends, table-driven steps, whole horizontal runs per ctz / clz) against the oracle's
    restatement of imageproc's find_contours: same borders, same order, same points -- and the candidates that come out of its
    corner-only mode (the detector's route) against simplify_chain + mini_box of the oracle."""
    n_c = 0
    for m in _stress_masks():
        ref = R.find_contours(m)
        for bands in (1, 3):
            got = api.host_contours(m, 100000, bands, bits=True)
            assert len(got) == len(ref)
            for (gp, gt), (rp, rt, _) in zip(got, ref):
                assert gt == rt and np.array_equal(gp, rp)
        n_c += len(ref)
        assert np.array_equal(api.host_candidates(m, 100000, 2), oracle_candidates(m, 100000))
        k = max(1, len(ref) // 2)                                                        # take(max_candidates) mid-band
        assert np.array_equal(api.host_candidates(m, k, 1), oracle_candidates(m, k))
    assert n_c > 3000


def test_host_fast_route_equals_the_round4_route():
    """OAR_HOST_FAST=0 keeps the round-4 host route (byte state plane, every point through simplify_chain and the Graham sort, scalar
    min-area-rectangle loop) for A/B timing: both routes must give the same candidates, unclipped polygons and second mini boxes."""
    import os, subprocess, sys
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from oar_ocr_amd import api
sys.path.insert(0, %r)
from test_host_cpu import blob_mask, _stress_masks
h = hashlib.sha256()
n = 0
for m in [blob_mask(s, (960, 960), 40) for s in (11, 12)] + _stress_masks():
    c = api.host_candidates(m, 1000, 4)
    h.update(c.tobytes())
    for b in c[:40]:
        u = api.host_unclip(b, 1.5)
        h.update(u.tobytes())
        if len(u) >= 3:
            mb = api.host_mini_box(u)
            h.update(repr(None if mb is None else (mb[0].tobytes(), float(mb[1]))).encode())
            n += 1
print(n, h.hexdigest())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OAR_HOST_FAST=v), capture_output=True, text=True, timeout=600) for v in ("0", "1")]
    assert all(o.returncode == 0 for o in outs), outs[0].stderr[-2000:] + outs[1].stderr[-2000:]
    assert outs[0].stdout == outs[1].stdout and int(outs[0].stdout.split()[0]) > 100


def test_convex_hull_fast_path_equals_an_exact_integer_graham_scan():
    """db_host.cc's convex_hull keeps only each row's leftmost / rightmost point before the Graham sort when the points are integer border
    pixels inside a <= 1000 px diagonal, on the argument that every float quantity of the scan is then exact.  Checked here against a scan in exact
    integer arithmetic (Python ints: angle order by cross-product sign, ties by squared distance, pop on cross <= 0 -- geometry.rs:226-271 with no
    rounding anywhere) instead of only against the library's own slow route: blobs, thin diagonals (many collinear points), duplicates, full rows."""
    from functools import cmp_to_key
    rng = np.random.default_rng(11)

    def exact_hull(pts):
        pts = [(int(x), int(y)) for x, y in pts]
        si = min(range(len(pts)), key=lambda i: (pts[i][1], pts[i][0]))      # first of the lowest-then-leftmost points (strict < in the scan for the start)
        s = pts[si]
        rest = [(i, p) for i, p in enumerate(pts) if i != si]

        def cmp(a, b):
            (ia, pa), (ib, pb) = a, b
            ax, ay, bx, by = pa[0] - s[0], pa[1] - s[1], pb[0] - s[0], pb[1] - s[1]
            # all points have y >= s.y: angles lie in [0, pi]; (0, 0) has atan2 = 0 like the +x ray
            cr = ax * by - ay * bx
            za, zb = (ax == 0 and ay == 0), (bx == 0 and by == 0)
            if za or zb:
                ka = 0 if (za or (ay == 0 and ax > 0)) else 1
                kb = 0 if (zb or (by == 0 and bx > 0)) else 1
                if ka != kb:
                    return -1 if ka < kb else 1
                if ka == 1:
                    return 0 if ia == ib else (-1 if ia < ib else 1)
            elif cr != 0:
                return -1 if cr > 0 else 1
            elif ax * bx + ay * by < 0:                                      # opposite rays on the start row: angle 0 before angle pi
                return -1 if ax > 0 else 1
            da, db = ax * ax + ay * ay, bx * bx + by * by
            if da != db:
                return -1 if da < db else 1
            return -1 if ia < ib else (1 if ia > ib else 0)
        rest.sort(key=cmp_to_key(cmp))
        hull = []
        for p in [s] + [p for _, p in rest]:
            while len(hull) > 1:
                a, b = hull[-2], hull[-1]
                if (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) <= 0:
                    hull.pop()
                else:
                    break
            hull.append(p)
        return np.asarray(hull, np.float32)

    n_cases = 0
    for case in range(400):
        kind = case % 4
        if kind == 0:      # a filled blob's pixels
            w, h = int(rng.integers(8, 300)), int(rng.integers(4, 60))
            n = int(rng.integers(13, 400))
            pts = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1)
        elif kind == 1:    # a thin diagonal band: long collinear runs
            n = int(rng.integers(13, 200))
            t = rng.integers(0, 300, n)
            pts = np.stack([t * 2 + rng.integers(0, 2, n), t + rng.integers(0, 2, n)], 1)
        elif kind == 2:    # a rectangle border with duplicates
            w, h = int(rng.integers(5, 400)), int(rng.integers(3, 50))
            xs = np.concatenate([np.arange(w), np.full(h, w - 1), np.arange(w)[::-1], np.zeros(h, int)])
            ys = np.concatenate([np.zeros(w, int), np.arange(h), np.full(w, h - 1), np.arange(h)[::-1]])
            pts = np.stack([xs, ys], 1)
            pts = np.concatenate([pts, pts[rng.integers(0, len(pts), 10)]])
        else:              # a rotated text-line outline
            ang = rng.uniform(-0.5, 0.5)
            L, H = rng.uniform(40, 600), rng.uniform(8, 40)
            u = rng.uniform(0, 1, (200, 2)) * [L, H]
            pts = np.round(u @ np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]]) + 300).astype(int)
        pts = pts + np.array([int(rng.integers(0, 2000)), int(rng.integers(0, 2000))])
        if np.hypot(np.ptp(pts[:, 0]), np.ptp(pts[:, 1])) > 1000 or len(pts) <= 12:
            continue
        got = api.host_convex_hull(pts.astype(np.float32))
        want = exact_hull(pts)
        assert got.shape == want.shape and np.array_equal(got, want), (case, kind, got, want)
        n_cases += 1
    assert n_cases > 300
