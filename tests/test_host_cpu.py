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
