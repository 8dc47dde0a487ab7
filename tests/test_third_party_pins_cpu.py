"""Independent pins for the four third-party algorithms the reference calls but does not ship or test
(VERDICT r1 "parity unpinned exactly where boxes bit-exact is decided"):

  image 0.25.6       Triangle resize          processors/resize_detection.rs:314, models/recognition/crnn.rs:104-109
  imageproc 0.27     find_contours            processors/db_bitmap.rs:100
  clipper2-rust 1.0.3 inflate_paths_d (Round) processors/db_bitmap.rs:332-340
  nalgebra 0.35      LU solve + 3x3 inverse   utils/transform.rs:266-267,312-316

None of these can be run here (Rust crates, no cargo), so the restatements -- the oracle's (oracle/oar_oracle.c) AND the
product's host code (oar_ocr_amd/csrc/db_host.cc, through the oar_host_* hooks) -- are checked against implementations
that share no code with either: torch / PIL antialiased bilinear resampling, scipy.ndimage connected-component labelling,
closed-form polygon offsets, numpy float64 linear algebra.  The product's geometry used to be a statement-for-statement
twin of the oracle's; bit-equality between the two proved nothing, these vectors do."""
import ctypes as C
import math

import numpy as np
import pytest
import scipy.ndimage as ndi

from oar_ocr_amd import api, build
from oracle import cpu_ref


@pytest.fixture(scope="module", autouse=True)
def _libs():
    build.build_lib()
    cpu_ref.build()


def _smooth(rng, h, w):
    a = ndi.gaussian_filter(rng.random((h, w, 3)) * 255.0, (2, 2, 0))
    return ((a - a.min()) / (a.max() - a.min()) * 255.0).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ Triangle resize
TRI_CASES = [(100, 160, 48, 77, "smooth"), (37, 233, 48, 301, "smooth"), (480, 640, 352, 480, "smooth"), (64, 64, 224, 224, "smooth"),
             (31, 57, 48, 89, "noise"), (200, 300, 48, 72, "noise"), (960, 720, 512, 512, "smooth"), (20, 333, 48, 800, "noise"),
             (48, 320, 48, 320, "noise")]


@pytest.mark.parametrize("h,w,nh,nw,kind", TRI_CASES)
def test_triangle_resize_matches_antialiased_bilinear(h, w, nh, nw, kind):
    """`image`'s Triangle filter is the separable tent of support max(ratio, 1): the same resampling torch calls
    bilinear + antialias and PIL calls BILINEAR (reducing).  f32 torch agrees to <= 1 grey level on <= 2 % of the bytes
    (different summation order); PIL rounds to u8 between its two passes, so only the 1-level bound holds there."""
    import torch
    import torch.nn.functional as F
    from PIL import Image
    rng = np.random.default_rng(h * 1000 + w)
    img = _smooth(rng, h, w) if kind == "smooth" else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = cpu_ref.resize_triangle(img, nw, nh).astype(np.int32)
    t = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    t = t[0].permute(1, 2, 0).clamp(0, 255).round().numpy().astype(np.int32)
    assert np.abs(got - t).max() <= 1
    assert (got != t).mean() <= 0.02
    pil = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)).astype(np.int32)
    assert np.abs(got - pil).max() <= 1
    if (h, w) == (nh, nw):
        assert np.array_equal(got, img)   # same size short-circuits to a copy (imageops::resize)


def test_triangle_resize_of_constant_and_ramp_images():
    """closed forms: a constant image stays constant (weights are normalised); a horizontal ramp downscaled by an integer
    factor k gives the mean of each k-block pair weighted by the tent, which for a linear ramp is the ramp value at the
    sample centre."""
    const = np.full((40, 60, 3), 137, np.uint8)
    assert (cpu_ref.resize_triangle(const, 23, 17) == 137).all()
    assert (cpu_ref.resize_triangle(const, 111, 93) == 137).all()
    w = 240
    ramp = np.tile((np.arange(w, dtype=np.float64) / 2.0)[None, :, None], (8, 1, 3)).astype(np.uint8)   # 0, 0, 1, 1, 2, 2, ...
    out = cpu_ref.resize_triangle(ramp, w // 4, 8)[4, :, 0].astype(np.float64)
    centre = ((np.arange(w // 4) + 0.5) * 4 - 0.5) / 2.0 - 0.25   # ramp value (incl. the floor's mean offset) at the output centre
    inner = slice(2, w // 4 - 2)
    assert np.abs(out[inner] - centre[inner]).max() <= 1.0


# ------------------------------------------------------------------------------------------------ contours
S8, S4 = np.ones((3, 3), int), ndi.generate_binary_structure(2, 1)


def _border_sets(mask):
    """Ground truth from connected-component labelling alone: one border per 4-adjacent pair (8-connected foreground
    component C, 4-connected background component B); its pixels are the pixels of C that touch B.  imageproc starts a
    border only at a pixel with a background pixel INSIDE the image to its left (x > 0) or right (x + 1 < width), so a
    border without such a pixel -- the outline of a band that spans the full image width -- is never found (the reference
    shares this).  Returns {frozenset(points): ('outer' | 'hole', C)} for the borders that can be found."""
    fg, _ = ndi.label(mask > 0, structure=S8)
    bg, _ = ndi.label(~np.pad(mask > 0, 1), structure=S4)   # the zero frame joins everything that reaches the image edge
    fgp = np.pad(fg, 1)
    pairs = {}
    for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        nb = np.roll(bg, (-dy, -dx), axis=(0, 1))
        ys, xs = np.nonzero((fgp > 0) & (nb > 0))
        for y, x in zip(ys, xs):
            pairs.setdefault((int(fgp[y, x]), int(nb[y, x])), set()).add((int(x) - 1, int(y) - 1))
    h, w = mask.shape
    startable = set()
    for dx in (1, -1):   # horizontal neighbours that are real pixels
        nb = np.roll(bg, -dx, axis=1)
        inside = np.zeros_like(bg, bool)
        inside[1:h + 1, (1 if dx > 0 else 2):(w if dx > 0 else w + 1)] = True
        ys, xs = np.nonzero((fgp > 0) & (nb > 0) & inside)
        startable.update(zip(fgp[ys, xs].tolist(), nb[ys, xs].tolist()))
    frame = int(bg[0, 0])
    kind, seen_b, seen_c, todo = {}, {frame}, set(), [("B", frame)]
    while todo:   # alternate background / foreground levels of the containment tree
        t, i = todo.pop()
        for (c, b) in pairs:
            if t == "B" and b == i and c not in seen_c:
                kind[(c, b)] = "outer"; seen_c.add(c); todo.append(("C", c))
            if t == "C" and c == i and b not in seen_b:
                kind[(c, b)] = "hole"; seen_b.add(b); todo.append(("B", b))
    return {frozenset(v): (kind[k], k[0]) for k, v in pairs.items() if k in startable}, fg


def _random_masks(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        h, w = int(rng.integers(4, 64)), int(rng.integers(4, 64))
        sigma = float(rng.uniform(0.5, 2.2))
        m = (ndi.gaussian_filter(rng.random((h, w)), sigma) > 0.5).astype(np.uint8) * 255
        if i % 5 == 0:
            m[:, 0] = 255 * (rng.random(h) > 0.4)        # plenty of components on the x == 0 column (imageproc's `x > 0` quirk)
        if i % 7 == 0:
            m[rng.integers(0, h)] = 255                   # a full-width 1-pixel line
        yield m


@pytest.mark.parametrize("impl", ["oracle", "product", "product_banded"])
def test_contours_are_the_component_borders(impl):
    """find_contours (Suzuki-Abe) must yield exactly one closed 8-connected chain per (component, adjacent background
    region) pair, visiting exactly that border's pixels, in raster order of the chain's first pixel; topological holes are
    typed Hole; a chain typed Outer starts at its border's raster-first pixel.  (imageproc only starts an Outer border at
    x > 0, so a component whose border begins on the x == 0 column starts later, or comes out typed Hole -- only then.)"""
    total = 0
    for mask in _random_masks(150, 11):
        truth, fg = _border_sets(mask)
        if impl == "oracle":
            cs = [(p, t) for p, t, _ in cpu_ref.find_contours(mask)]
        else:
            cs = api.host_contours(mask, max_bands=1 if impl == "product" else 6)
        assert len(cs) == len(truth)
        seen = set()
        starts = []
        for pts, btype in cs:
            key = frozenset(map(tuple, pts.tolist()))
            assert key in truth and key not in seen
            seen.add(key)
            kind, comp = truth[key]
            first = min(key, key=lambda p: (p[1], p[0]))
            if kind == "hole":
                assert btype == 1
            elif btype == 0:
                sx, sy = int(pts[0][0]), int(pts[0][1])
                assert sx > 0 and mask[sy, sx - 1] == 0          # an Outer border starts right of a background pixel
                if first[0] > 0:
                    assert (sx, sy) == first                      # ... at the border's raster-first pixel
                else:
                    assert (fg[:, 0] == comp).any()
            else:
                assert (fg[:, 0] == comp).any(), "an outer border typed Hole must belong to a component on the x == 0 column"
            if len(pts) > 1:
                step = np.abs(np.diff(np.vstack([pts, pts[:1]]), axis=0)).max(axis=1)
                assert step.max() <= 1 and step.min() >= 0
            starts.append((int(pts[0][1]), int(pts[0][0])))
            total += 1
        assert starts == sorted(starts)
    assert total > 1000


def test_product_contours_equal_the_oracle_chain_for_chain():
    for mask in _random_masks(60, 23):
        a = cpu_ref.find_contours(mask)
        for bands, bits in ((1, False), (5, False), (1, True), (4, True)):      # bits: the detector's bit-plane read-back format
            b = api.host_contours(mask, max_bands=bands, bits=bits)
            assert len(a) == len(b)
            for (pa, ta, _), (pb, tb) in zip(a, b):
                assert ta == tb and np.array_equal(pa, pb)


def test_contour_counts_follow_from_component_counts():
    """#contours = #8-components + #enclosed 4-background regions (each background region that does not reach the image
    edge is a hole of exactly one component) -- on masks without a full-width band, whose outline imageproc cannot start."""
    checked = 0
    for mask in _random_masks(120, 31):
        if (mask > 0).all(axis=1).any():
            continue
        n8 = ndi.label(mask > 0, structure=S8)[1]
        bg, nb = ndi.label(~np.pad(mask > 0, 1), structure=S4)
        assert len(cpu_ref.find_contours(mask)) == n8 + (nb - 1)
        assert len(api.host_contours(mask)) == n8 + (nb - 1)
        checked += 1
    assert checked > 60


def test_full_width_band_has_no_outline():
    """the quirk itself: rows that are foreground from x = 0 to x = width - 1 offer no start pixel"""
    mask = np.zeros((12, 9), np.uint8)
    mask[3:6] = 255
    assert cpu_ref.find_contours(mask) == [] and api.host_contours(mask) == []
    mask[4, 4] = 0                                # a hole inside the band IS found (its left neighbour has a right background pixel)
    for cs in ([(p, t) for p, t, _ in cpu_ref.find_contours(mask)], api.host_contours(mask)):
        assert len(cs) == 1 and cs[0][1] == 1 and set(map(tuple, cs[0][0].tolist())) == {(3, 4), (4, 3), (5, 4), (4, 5)}


# ------------------------------------------------------------------------------------------------ unclip (polygon offset)
def _rect(cx, cy, w, h, deg):
    a = math.radians(deg)
    r = np.array([[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]])
    return (np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ r.T + [cx, cy])


def _dist_to_rect(p, cx, cy, w, h, deg):
    a = math.radians(-deg)
    r = np.array([[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]])
    q = np.abs((p - [cx, cy]) @ r.T)
    d = np.maximum(q - [w / 2, h / 2], 0.0)
    return np.hypot(d[:, 0], d[:, 1])


UNCLIP = {"oracle": (cpu_ref.unclip, cpu_ref.mini_box), "product": (api.host_unclip, api.host_mini_box)}
RECTS = [(50, 30, 100, 20, 0, 1.5), (400, 300, 321, 37, 0, 2.0), (200, 200, 180, 24, 7.5, 1.5), (300, 310, 90, 40, -31, 2.0),
         (500, 500, 640, 48, 2.25, 1.5), (150, 150, 60, 60, 45, 0.5), (256, 128, 33, 9, 88, 1.6), (100, 100, 12, 4, 13, 3.0)]


@pytest.mark.parametrize("impl", ["oracle", "product"])
@pytest.mark.parametrize("cx,cy,w,h,deg,ratio", RECTS)
def test_unclip_is_the_round_offset_of_the_rectangle(impl, cx, cy, w, h, deg, ratio):
    """Minkowski sum of a w x h rectangle with a disc of radius d = area * ratio / perimeter: every vertex of the result
    lies at distance d from the rectangle (to the 0.01 px output grid, arcs approximated from inside within the arc
    tolerance d / 500), its area is w h + 2 d (w + h) + pi d^2 minus the arcs' sagitta slivers, its bounding rectangle is
    the rectangle grown by d on every side -- and that is the box DB post-processing keeps."""
    unclip, mini_box = UNCLIP[impl]
    box = _rect(cx, cy, w, h, deg).astype(np.float32)
    d = (w * h) * ratio / (2 * (w + h))
    out = unclip(box, ratio).astype(np.float64)
    assert len(out) >= 8
    dist = _dist_to_rect(out, cx, cy, w, h, deg)
    grid = 0.01 * math.sqrt(2) / 2 + 0.012          # output grid + the input corners' own snap to the grid (+ f32 box)
    assert dist.max() <= d + grid and dist.min() >= d - d * 0.002 - grid
    x, y = out[:, 0], out[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    full = w * h + 2 * d * (w + h) + math.pi * d * d
    assert full - 4 * d * d * 0.05 - 0.05 * (w + h) <= area <= full + 0.05 * (w + h)
    # vertex count: four quarter circles of ceil(steps_per_rad * pi / 2) segments, steps/360 = min(pi / acos(1 - 1/500), 100 d pi)
    per_turn = min(math.pi / math.acos(1 - 0.002), d * 100 * math.pi)
    seg = math.ceil(per_turn / (2 * math.pi) * (math.pi / 2) - 1e-9)
    assert abs(len(out) - 4 * (seg + 1)) <= 4
    got, min_side = mini_box(out.astype(np.float32))
    want = _rect(cx, cy, w + 2 * d, h + 2 * d, deg)
    err = min(np.abs(got - np.roll(want, s, axis=0)).max() for s in range(4))
    assert err <= (0.011 if deg == 0 else 0.03), (got, want)
    assert abs(min_side - (min(w, h) + 2 * d)) <= 0.03


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_unclip_orientation_and_degenerate_inputs(impl):
    unclip, mini_box = UNCLIP[impl]
    box = _rect(120, 80, 140, 22, 11.0).astype(np.float32)
    a, b = mini_box(unclip(box, 1.5)), mini_box(unclip(box[::-1].copy(), 1.5))   # clockwise and counter-clockwise input
    assert np.abs(a[0] - b[0]).max() <= 0.02
    assert len(unclip(np.array([[5, 5], [5, 5], [5, 5], [5, 5]], np.float32), 1.5)) == 0          # zero area => dropped
    assert len(unclip(np.array([[0, 0], [10, 0], [20, 0], [30, 0]], np.float32), 1.5)) == 0        # collinear => dropped
    tiny = unclip(np.array([[0, 0], [4, 0], [4, 0.004], [0, 0.004]], np.float32), 1.0)              # offset < half a grid step
    assert len(tiny) in (0, 2, 3, 4)


# ------------------------------------------------------------------------------------------------ homography solve
def _ordered(q):
    """transform.rs:126-143: sort by x (stable), pair by y"""
    s = q[np.argsort(q[:, 0], kind="stable")]
    a, d = (1, 0) if s[1, 1] < s[0, 1] else (0, 1)
    b, c = (3, 2) if s[3, 1] < s[2, 1] else (2, 3)
    return s[[a, b, c, d]]


def _h64(src, dst):
    rows, rhs = [], []
    for (sx, sy), (dx, dy) in zip(src, dst):
        rows += [[sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx], [0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy]]
        rhs += [dx, dy]
    hm = np.append(np.linalg.solve(np.array(rows, np.float64), np.array(rhs, np.float64)), 1.0).reshape(3, 3)
    return hm, np.linalg.inv(hm)


def test_homography_solve_and_inverse_match_float64_linear_algebra():
    """8 x 8 LU solve + 3 x 3 inverse in f32 (nalgebra's elimination order) vs numpy float64: coefficients within 1e-5
    relative, back-projected crop pixels within 2e-3 px -- for the oracle (orc_perspective_transform / orc_inverse3) and the
    product (oar_host_plan_crop)."""
    L = cpu_ref.lib()
    rng = np.random.default_rng(3)
    checked = 0
    for _ in range(400):
        w, h, ang = rng.uniform(20, 900), rng.uniform(8, 80), rng.uniform(-0.5, 0.5)
        c = np.array([rng.uniform(150, 850), rng.uniform(150, 850)])
        r = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        q = (np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ r.T + c + rng.normal(0, 1.5, (4, 2))).astype(np.float32)
        plan, inv = api.host_plan_crop(2000, 2000, q)
        if plan[0] != 2:
            continue
        ow, oh = (plan[6], plan[5]) if plan[7] else (plan[5], plan[6])
        src = _ordered(q - np.array([plan[1], plan[2]], np.float32))
        dst = np.array([[0, 0], [ow, 0], [ow, oh], [0, oh]], np.float32)
        h64, hi64 = _h64(src.astype(np.float64), dst.astype(np.float64))
        m9, io = np.zeros(9, np.float32), np.zeros(9, np.float32)
        assert L.orc_perspective_transform(src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), m9.ctypes.data_as(C.c_void_p)) == 1
        assert L.orc_inverse3(m9.ctypes.data_as(C.c_void_p), io.ctypes.data_as(C.c_void_p)) == 1
        assert np.abs(m9.reshape(3, 3) - h64).max() <= 1e-5 * np.abs(h64).max()
        grid = np.array([[x, y, 1.0] for x in (0, ow / 2, ow) for y in (0, oh / 2, oh)])
        want = grid @ hi64.T
        want = want[:, :2] / want[:, 2:]
        for name, mat in (("oracle", io), ("product", inv)):
            m = mat.reshape(3, 3).astype(np.float64)
            assert np.abs(m - hi64).max() <= 1e-5 * np.abs(hi64).max(), name
            got = grid @ m.T
            assert np.abs(got[:, :2] / got[:, 2:] - want).max() <= 2e-3, name
        assert np.abs((np.c_[dst, np.ones(4)] @ inv.reshape(3, 3).astype(np.float64).T)[:, :2] / (np.c_[dst, np.ones(4)] @ inv.reshape(3, 3).astype(np.float64).T)[:, 2:] - src).max() <= 2e-3
        checked += 1
    assert checked > 300
