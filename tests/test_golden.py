"""tests/golden/third_party.npz (made by tests/golden/make_golden.py) pins the restatements of the four third-party algorithms of the
path -- image's Triangle resize, imageproc's find_contours, Clipper2's round-join offset (quads and concave polygons), the
homography / bicubic crop -- to ONE committed set of numbers:
  * CPU: the oracle still reproduces them (a change to oracle/ that moves any of these bytes must regenerate the file on purpose);
    the product's host routines (db_host.cc, poly_host.cc through the C ABI) produce the same contours / offsets;
  * GPU (-m gpu): the HIP kernels produce the same resized images, contours, unclipped polygons and crops.
The vectors are the oracle's, not the real crates' (no Rust toolchain here); third_party.json says which call each array stands for,
so that they can be diffed against the crates wherever cargo exists."""
from pathlib import Path

import numpy as np
import pytest

from oar_ocr_amd import api
from oracle import cpu_ref as R
from oracle import poly_ref

G = np.load(Path(__file__).parent / "golden" / "third_party.npz")
N_TRI = sum(1 for k in G.files if k.startswith("tri") and k.endswith("_in"))
N_CNT = sum(1 for k in G.files if k.startswith("cnt") and k.endswith("_mask"))
N_CROP = sum(1 for k in G.files if k.startswith("crop") and k.endswith("_box"))
RATIOS = (1.5, 2.0, 0.5)
POLYS = sorted(k[5:-3] for k in G.files if k.startswith("poly_") and k.endswith("_in"))


def _split(points, offsets):
    return [points[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


# ------------------------------------------------------------------------------------------------ CPU: oracle + host routines
def test_oracle_reproduces_the_committed_vectors():
    for i in range(N_TRI):
        o = G[f"tri{i}_out"]
        assert np.array_equal(R.resize_triangle(G[f"tri{i}_in"], o.shape[1], o.shape[0]), o)
    for i in range(N_CNT):
        cs = R.find_contours(G[f"cnt{i}_mask"])
        want = _split(G[f"cnt{i}_points"], G[f"cnt{i}_offsets"])
        assert len(cs) == len(want)
        for (pts, bt, _), w, hole in zip(cs, want, G[f"cnt{i}_is_hole"]):
            assert np.array_equal(pts, w) and bt == hole
    for r in RATIOS:
        want = _split(G[f"unclip_r{r}_points"], G[f"unclip_r{r}_offsets"])
        for q, w in zip(G["unclip_quads"], want):
            assert np.array_equal(R.unclip(q, r), w)
    for name in POLYS:
        for r in (0.5, 1.5):
            got = poly_ref.unclip_poly(G[f"poly_{name}_in"], r)
            got = np.asarray(got, np.float32).reshape(-1, 2) if got is not None else np.zeros((0, 2), np.float32)
            assert np.array_equal(got, G[f"poly_{name}_r{r}"]), (name, r)
    for i in range(N_CROP):
        c = R.rotate_crop(G["crop_page"], G[f"crop{i}_box"])
        assert np.array_equal(c, G[f"crop{i}_out"])


def test_host_routines_match_the_committed_vectors():
    """db_host.cc / poly_host.cc (what the pipeline runs on the host pool by default) through the C ABI -- no GPU involved"""
    for i in range(N_CNT):
        cs = api.host_contours(G[f"cnt{i}_mask"])
        want = _split(G[f"cnt{i}_points"], G[f"cnt{i}_offsets"])
        assert len(cs) == len(want)
        for (pts, hole), w, h in zip(cs, want, G[f"cnt{i}_is_hole"]):
            assert np.array_equal(pts, w) and int(hole) == h
    for r in RATIOS:
        want = _split(G[f"unclip_r{r}_points"], G[f"unclip_r{r}_offsets"])
        for q, w in zip(G["unclip_quads"], want):
            assert np.array_equal(api.host_unclip(q, r), w)
    for name in POLYS:
        for r in (0.5, 1.5):
            got = api.host_unclip_poly(G[f"poly_{name}_in"], r)
            assert np.array_equal(np.asarray(got, np.float32).reshape(-1, 2), G[f"poly_{name}_r{r}"]), (name, r)


# ------------------------------------------------------------------------------------------------ GPU: the HIP kernels
@pytest.mark.gpu
def test_hip_kernels_match_the_committed_vectors():
    for i in range(N_TRI):
        o = G[f"tri{i}_out"]
        assert np.array_equal(api.k_resize_triangle(G[f"tri{i}_in"], o.shape[1], o.shape[0]), o)
    for i in range(N_CNT):
        cs = api.k_contours(G[f"cnt{i}_mask"])
        want = _split(G[f"cnt{i}_points"], G[f"cnt{i}_offsets"])
        assert len(cs) == len(want)
        for (pts, hole), w, h in zip(cs, want, G[f"cnt{i}_is_hole"]):
            assert np.array_equal(pts, w) and int(hole) == h
    for r in RATIOS:
        want = _split(G[f"unclip_r{r}_points"], G[f"unclip_r{r}_offsets"])
        got = api.k_unclip(G["unclip_quads"], r)
        for g, w in zip(got, want):
            assert g is not None and np.array_equal(g, w)
    for i in range(N_CROP):
        c = api.k_rotate_crop(G["crop_page"], G[f"crop{i}_box"])
        assert np.array_equal(c, G[f"crop{i}_out"])
