"""Seeded synthetic pages: white RGB page, K text-like lines made of dark glyph-ish
blocks, heights 18-30 px, widths 200-800 px; a fraction is rotated by up to +-3 degrees so the bicubic
perspective-warp path (utils/transform.rs:155-191) is exercised, the rest stay exactly axis aligned
(fast path, utils/transform.rs:150-152).  Pure numpy, deterministic per (seed, size, lines)."""
from __future__ import annotations

import numpy as np


def _draw_line(page, rng, x0, y0, w, h, angle_deg):
    H, W, _ = page.shape
    # glyph pattern in a local (h x w) canvas: dark character cells separated by thin gaps
    canvas = np.full((h, w), 255, np.uint8)
    x = 0
    while x < w:
        cw = int(rng.integers(max(h // 2, 6), max(h, 8)))
        ink = int(rng.integers(10, 70))
        canvas[:, x:min(x + cw, w)] = ink
        # carve a few light strokes so the block is glyph-like rather than solid
        for _ in range(2):
            sx = x + int(rng.integers(1, max(cw - 2, 2)))
            if sx < w:
                canvas[int(rng.integers(0, h // 2)):int(rng.integers(h // 2, h)), sx:sx + 2] = 230
        x += cw + int(rng.integers(2, 4))
    if angle_deg == 0.0:
        page[y0:y0 + h, x0:x0 + w] = canvas[:, :, None]
        return
    # rotate about the line centre with nearest sampling (inverse map)
    a = np.deg2rad(angle_deg)
    ca, sa = np.cos(a), np.sin(a)
    pad = int(abs(sa) * w / 2) + 3
    ys, xs = np.mgrid[y0 - pad:y0 + h + pad, x0 - pad:x0 + w + pad]
    cx, cy = x0 + w / 2.0, y0 + h / 2.0
    u = (xs - cx) * ca + (ys - cy) * sa + w / 2.0
    v = -(xs - cx) * sa + (ys - cy) * ca + h / 2.0
    ui, vi = np.floor(u).astype(int), np.floor(v).astype(int)
    ok = (ui >= 0) & (ui < w) & (vi >= 0) & (vi < h) & (ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)
    page[ys[ok], xs[ok]] = canvas[vi[ok], ui[ok]][:, None]


def make_page(seed: int, size=(960, 960), lines: int = 40, rotated_fraction: float = 0.25) -> np.ndarray:
    """Returns [H,W,3] u8."""
    H, W = size
    rng = np.random.default_rng(seed)
    page = np.full((H, W, 3), 255, np.uint8)
    # rows: place lines on a jittered grid so they never overlap
    margin = 24
    slot = 44 if lines >= 18 else 60
    rows = (H - 2 * margin) // slot
    if rows <= 0 or W - 2 * margin < 120 or lines <= 0:   # too small for a text line: a blank page
        return page
    cols = max(1, int(np.ceil(lines / rows)))
    col_w = (W - 2 * margin) // cols
    if col_w < 96:
        cols = max(1, (W - 2 * margin) // 96)
        col_w = (W - 2 * margin) // cols
    k = 0
    for r in range(rows):
        for c in range(cols):
            if k >= lines:
                break
            h = int(rng.integers(18, min(31, slot - 12)))
            max_w = max(min(800, col_w - 24), 40)
            w = int(rng.integers(min(200, max_w - 1), max_w))
            x0 = margin + c * col_w + int(rng.integers(0, max(col_w - 24 - w, 1)))
            y0 = margin + r * slot + int(rng.integers(0, slot - 4 - h))
            ang = 0.0
            if rng.random() < rotated_fraction:
                lim = min(3.0, np.rad2deg(np.arctan2((slot - h) / 2.0 - 1.0, w / 2.0)))
                ang = float(rng.uniform(-lim, lim))
            _draw_line(page, rng, x0, y0, w, h, ang)
            k += 1
    return page


def make_seal_page(seed: int, size=(640, 640), arcs: int = 3, straight: int = 1) -> np.ndarray:
    """A stamp-like page for the seal / polygon branch: `arcs` curved text bands on concentric circles (glyph cells
    laid out along the arc, so the detector's blob is a bent band whose contour approximates to a concave polygon) and `straight`
    ordinary lines through the middle.  Returns [H,W,3] u8; deterministic per (seed, size, arcs, straight)."""
    H, W = size
    rng = np.random.default_rng(seed)
    page = np.full((H, W, 3), 255, np.uint8)
    cx, cy = W / 2.0 + float(rng.uniform(-8, 8)), H / 2.0 + float(rng.uniform(-8, 8))
    ys, xs = np.mgrid[0:H, 0:W]
    rr = np.hypot(xs - cx, ys - cy)
    th = np.arctan2(ys - cy, xs - cx)
    r_out = min(H, W) / 2.0 - 28.0
    for a in range(arcs):
        band = float(rng.integers(20, 30))
        r1 = r_out - a * (band + 26.0)
        r0 = r1 - band
        if r0 < 40:
            break
        t0 = float(rng.uniform(-np.pi, np.pi))
        span = float(rng.uniform(1.6, 3.6))                      # radians of arc covered by the band
        rel = np.mod(th - t0, 2 * np.pi)
        inside = (rr >= r0) & (rr < r1) & (rel < span)
        rm = 0.5 * (r0 + r1)
        s_along = rel * rm                                       # arc length along the band's mid circle
        cell = float(rng.integers(14, 24))
        gap = 3.0
        in_cell = np.mod(s_along, cell + gap) < cell
        ink = (20 + 40 * ((np.floor(s_along / (cell + gap)).astype(np.int64) * 2654435761 + seed) % 7) / 7.0).astype(np.uint8)
        m = inside & in_cell
        page[m] = ink[m][:, None]
    for k in range(straight):
        h = int(rng.integers(18, 28))
        w = int(min(W * 0.45, 2 * (r_out - arcs * 52.0) - 20)) if arcs else int(W * 0.5)
        if w >= 60:
            _draw_line(page, rng, int(cx - w / 2), int(cy - h / 2 + k * (h + 14)), w, h, 0.0)
    return page


def make_pages(n: int, size=(960, 960), lines: int = 40, seed0: int = 0):
    return [make_page(seed0 + i, size, lines) for i in range(n)]


def make_crop(seed: int, w: int = 320, h: int = 48) -> np.ndarray:
    """A single text-line crop (BASELINE config 1: one 48x320 line)."""
    rng = np.random.default_rng(seed)
    page = np.full((h, w, 3), 255, np.uint8)
    _draw_line(page, rng, 4, 6, w - 8, h - 12, 0.0)
    return page
