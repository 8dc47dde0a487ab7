"""Generates tests/golden/third_party.npz: input / output vectors of the four THIRD-PARTY algorithms the hot path depends on and the
reference holds no test vector for: image 0.25.6 `imageops::resize(Triangle)`, imageproc 0.27 `find_contours`,
clipper2-rust 1.0.3 `inflate_paths_d` (round join, as DBPostProcess::unclip calls it, db_bitmap.rs:279-368) for quads and for
concave / touching polygons (seal mode), nalgebra 0.35's 8x8 solve behind `get_rotate_crop_image` (transform.rs:212-283).

The outputs are those of THIS repository's oracle (oracle/oar_oracle.c, oracle/poly_ref.py) -- the reference is a Rust workspace
that cannot be built in this image, so these are NOT outputs of the real crates.  They exist so that (a) the oracle, the host
routines and the HIP kernels are pinned to one committed set of numbers (tests/test_golden.py), and (b) anyone with cargo can diff
the real crates against this file: each array is documented in `third_party.json` with the exact call it stands for.

usage: python tests/golden/make_golden.py    (from the repository root; rewrites third_party.npz / third_party.json)"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import cpu_ref as R   # noqa: E402
from oracle import poly_ref       # noqa: E402

rng = np.random.default_rng(20260928)
out, doc = {}, {}

# ---- Triangle resize: image::imageops::resize(&img, nw, nh, FilterType::Triangle) on RGB8
tri_cases = [(37, 23, 64, 48), (160, 48, 107, 48), (91, 33, 200, 48), (200, 60, 50, 15), (17, 96, 9, 48), (240, 32, 120, 16)]
for i, (w, h, nw, nh) in enumerate(tri_cases):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out[f"tri{i}_in"] = img
    out[f"tri{i}_out"] = R.resize_triangle(img, nw, nh)
    doc[f"tri{i}"] = f"imageops::resize(RgbImage {w}x{h}, {nw}, {nh}, FilterType::Triangle) -> tri{i}_out [h, w, 3] u8"

# ---- find_contours: imageproc::contours::find_contours::<u32>(&GrayImage) (foreground = non-zero); points, border type, discovery order
masks = []
m = np.zeros((40, 56), np.uint8); m[5:20, 6:30] = 255; m[9:15, 10:20] = 0; m[11:13, 12:16] = 255; m[25:38, 40:55] = 255; m[0, 0] = 255; m[39, 55] = 255
masks.append(m)
masks.append((rng.random((48, 64)) < 0.35).astype(np.uint8) * 255)
m = np.zeros((30, 30), np.uint8); m[:, 0] = 255; m[0, :] = 255; m[10:20, 29] = 255; m[15, 10:25] = 255
masks.append(m)
for i, m in enumerate(masks):
    cs = R.find_contours(m)
    out[f"cnt{i}_mask"] = m
    out[f"cnt{i}_offsets"] = np.cumsum([0] + [len(c[0]) for c in cs]).astype(np.int64)
    out[f"cnt{i}_points"] = np.concatenate([c[0] for c in cs]).astype(np.int32) if cs else np.zeros((0, 2), np.int32)
    out[f"cnt{i}_is_hole"] = np.array([int(c[1]) for c in cs], np.int32)
    doc[f"cnt{i}"] = "find_contours::<u32>(GrayImage from cnt_mask): contour k = points[offsets[k]:offsets[k+1]] (x, y), BorderType::Hole == is_hole[k]"

# ---- unclip of quads: DBPostProcess::unclip(box, ratio) -> inflate_paths_d(.., delta = area * ratio / perimeter, JoinType::Round, EndType::Polygon, 2.0, 2, 0.0)
quads = []
for _ in range(24):
    cx, cy, w, h, a = rng.uniform(50, 400), rng.uniform(50, 400), rng.uniform(8, 300), rng.uniform(6, 60), rng.uniform(-np.pi, np.pi)
    c, s = np.cos(a), np.sin(a)
    q = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
    quads.append(q.astype(np.float32))
quads[0] = np.round(quads[0]); quads[1] = quads[1][::-1].copy()
out["unclip_quads"] = np.stack(quads)
ratios = [1.5, 2.0, 0.5]
for r in ratios:
    res = [R.unclip(q, r) for q in quads]
    out[f"unclip_r{r}_offsets"] = np.cumsum([0] + [len(p) for p in res]).astype(np.int64)
    out[f"unclip_r{r}_points"] = np.concatenate(res).astype(np.float32)
doc["unclip"] = "for ratio in (1.5, 2.0, 0.5): DBPostProcess::unclip(unclip_quads[k], ratio) -> f32 points of the single offset path (closing duplicate dropped)"

# ---- unclip of concave / touching / self-crossing polygons (seal mode): the raw round-join ring and the closing Union(Positive) outline
polys = {
    "L": [(10, 10), (110, 10), (110, 40), (40, 40), (40, 120), (10, 120)],
    "U": [(0, 0), (30, 0), (30, 60), (70, 60), (70, 0), (100, 0), (100, 90), (0, 90)],
    "arc": [(100 + 80 * np.cos(t), 100 + 80 * np.sin(t)) for t in np.linspace(0.2, 2.9, 14)] + [(100 + 55 * np.cos(t), 100 + 55 * np.sin(t)) for t in np.linspace(2.9, 0.2, 14)],
    "star": [((30 if k % 2 else 90) * np.cos(k * np.pi / 7) + 120, (30 if k % 2 else 90) * np.sin(k * np.pi / 7) + 120) for k in range(14)],
    "slit": [(0, 0), (100, 0), (100, 100), (52, 100), (52, 20), (48, 20), (48, 100), (0, 100)],
}
for name, pts in polys.items():
    p = np.asarray(pts, np.float32)
    out[f"poly_{name}_in"] = p
    for r in (0.5, 1.5):
        res = poly_ref.unclip_poly(p, r)
        out[f"poly_{name}_r{r}"] = np.asarray(res, np.float32).reshape(-1, 2) if res is not None else np.zeros((0, 2), np.float32)
doc["poly"] = ("for ratio in (0.5, 1.5): DBPostProcess::unclip(poly_<name>_in, ratio) as polygons_from_bitmap calls it (db_bitmap.rs:16-82); an EMPTY array = "
               "inflate_paths_d returned != 1 path (box dropped, db_bitmap.rs:341).  Start vertex and vertex order are the part unverified against Clipper2.")

# ---- homography of get_rotate_crop_image: the rectified crop itself (bicubic, transform.rs:76-191) pins the 8x8 solve + inverse end to end
page = rng.integers(0, 256, (110, 160, 3), dtype=np.uint8)
out["crop_page"] = page
crop_boxes = [np.array([[20.3, 30.1], [140.7, 22.4], [143.2, 58.9], [22.8, 66.0]], np.float32), np.array([[30, 20], [90, 20], [90, 50], [30, 50]], np.float32),
              np.array([[100.5, 10.2], [120.1, 12.0], [112.3, 100.8], [92.9, 98.7]], np.float32)]
for i, b in enumerate(crop_boxes):
    out[f"crop{i}_box"] = b
    c = R.rotate_crop(page, b)
    out[f"crop{i}_out"] = c if c is not None else np.zeros((0, 0, 3), np.uint8)
doc["crop"] = "get_rotate_crop_image(crop_page, crop<i>_box) -> crop<i>_out (includes the rotate270 of tall crops, transform.rs:40-51)"

np.savez_compressed(Path(__file__).with_name("third_party.npz"), **out)
Path(__file__).with_name("third_party.json").write_text(json.dumps({"_note": "see make_golden.py; outputs are the repository oracle's, not the real crates'", **doc}, indent=1) + "\n")
print("wrote", len(out), "arrays")
