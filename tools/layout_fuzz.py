"""Random sweep of the layout path's kernels against the oracle (round 6): LayoutPostProcess (all three model types, every row format, random row counts / limits / thresholds,
exact score ties, NaN / inf scores, normalised and degenerate boxes) and the filtered resizes (Triangle / CatmullRom / Lanczos3 at random sizes) -- bit for bit.
usage: python tools/layout_fuzz.py [n_cases] [seed]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from oar_ocr_amd import api
from oracle import cpu_ref as R
from test_gpu_layout import _random_predictions

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
combos = [("picodet", 9, "scores"), ("picodet", 6, "csb"), ("picodet", 6, "bsc"), ("picodet", 7, "scb"), ("picodet", 12, "scores"), ("rtdetr", 6, "csb"), ("rtdetr", 7, "csb"),
          ("pp-doclayout", 6, "csb"), ("pp-doclayout", 7, "csb"), ("pp-doclayout", 8, "csb"), ("pp-doclayout", 4, "csb")]
for case in range(n_cases):
    if case % 4 == 3:
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        nw, nh = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        filt = str(rng.choice(["triangle", "catmullrom", "lanczos3"]))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ok = np.array_equal(api.k_resize_filter(a, nw, nh, filt), R.resize_filter(a, nw, nh, filt))
        label = f"resize {filt} {w}x{h} -> {nw}x{nh}"
    else:
        model_type, feat, fmt = combos[int(rng.integers(0, len(combos)))]
        ncls = int(rng.integers(1, 12))
        n, rows = int(rng.integers(1, 5)), int(rng.choice([1, 2, 17, 100, 300, 1000, 3000]))
        max_det, nms, thr = int(rng.choice([1, 5, 40, 100, 300])), float(rng.choice([0.1, 0.3, 0.5, 0.9])), float(rng.choice([0.0, 0.18, 0.5, 0.9]))
        pred = _random_predictions(rng, n, rows, feat, ncls, fmt, nan_ok=model_type != "pp-doclayout")
        wh = np.stack([rng.integers(30, 900, n), rng.integers(30, 1200, n)], -1).astype(np.float32)
        got = api.k_layout_postprocess(pred, wh, ncls, thr, nms, max_det, model_type)
        ok = True
        for i in range(n):
            rb, rc, rs = R.layout_postprocess(pred[i], wh[i, 0], wh[i, 1], ncls, thr, nms, max_det, model_type)
            gb, gc, gs = got[i]
            ok = ok and np.array_equal(gc, rc) and np.array_equal(gs, rs, equal_nan=True) and np.array_equal(gb, rb)
        label = f"{model_type} feat {feat} {fmt} classes {ncls} {n}x{rows} max {max_det} nms {nms} thr {thr}"
    if not ok:
        bad += 1
        print(f"FAIL case {case} [{label}]", flush=True)
print(f"{n_cases - bad}/{n_cases} layout cases bit-identical to the oracle in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
