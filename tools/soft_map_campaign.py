"""Detection parity in the SOFT-map regime, as a campaign: random pages through `build_det(..., soft=True)` detectors -- shallow final gain,
40x the weight on the random channels: thousands of pixels within 0.05 of the threshold, box scores straddling box_thresh -- HIP path against the oracle.
The rule is tests/test_gpu_pipeline.py::test_soft_probability_maps_boxes_within_the_float_budget's: a page is `identical` (boxes bit-exact, scores <= 1e-3) or
`within budget` (every oracle box has a GPU twin within 2 px and 2e-3 of score, unpaired boxes only with a score within 2e-3 of box_thresh) or a FAILURE.
usage: python tools/soft_map_campaign.py [pages_per_graph] [seed]     (graphs: tiny, tiny_full, hgnet_small; random sizes, thresholds, unclip ratios)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import pipeline_ref

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def classify(gb, gs, rb, rs, bt):
    if gb.shape == rb.shape and np.array_equal(gb, rb):
        return "identical" if np.allclose(gs, rs, atol=1e-3) else "FAIL(score)"
    used = set()
    for b, sc in zip(rb, rs):
        d = [np.abs(gb[j] - b).max() if j not in used else 1e9 for j in range(len(gb))]
        j = int(np.argmin(d)) if d else -1
        if j >= 0 and d[j] <= 2.0:
            used.add(j)
            if abs(float(gs[j]) - float(sc)) > 2e-3:
                return "FAIL(twin score)"
        elif abs(float(sc) - bt) > 2e-3:
            return "FAIL(oracle box without twin)"
    for j in range(len(gb)):
        if j not in used and abs(float(gs[j]) - bt) > 2e-3:
            return "FAIL(extra box)"
    return "within budget"


total = {"identical": 0, "within budget": 0}
fails = []
t0 = time.time()
for gi, graph in enumerate(("tiny", "tiny_full", "hgnet_small")):
    rng = np.random.default_rng(seed * 101 + gi)
    det, _ = models.build_det(graph, seed=int(rng.integers(0, 4)), soft=True)
    oracle = pipeline_ref.OracleDetector(det)
    pred = api.TextDetectionPredictor(det)
    done = boxes = marginal = 0
    tally = {"identical": 0, "within budget": 0}
    while done < n_pages:
        n_img = int(min(n_pages - done, rng.integers(1, 5)))
        same = rng.random() < 0.5
        h0, w0 = int(rng.integers(96, 700)), int(rng.integers(96, 700))
        imgs = []
        for _ in range(n_img):
            h, w = (h0, w0) if same else (int(rng.integers(96, 700)), int(rng.integers(96, 700)))
            imgs.append(pages.make_page(int(rng.integers(0, 1 << 30)), (h, w), int(rng.integers(1, 16))))
        thr, bt, un = float(rng.choice([0.2, 0.3, 0.4])), float(rng.choice([0.4, 0.5, 0.6])), float(rng.choice([1.5, 1.8, 2.0]))
        got = pred.predict(imgs, api.TextDetectionConfig(thr, bt, un))
        for k, (g, (rb, rs, prob)) in enumerate(zip(got, oracle.detect(imgs, thr, bt, un))):
            gb = np.stack([d.bbox for d in g]) if g else np.zeros((0, 4, 2), np.float32)
            gs = np.array([d.score for d in g], np.float32)
            boxes += len(rb)
            marginal += int((np.abs(prob - thr) < 1e-4).sum())
            c = classify(gb, gs, rb, rs, bt)
            if c in tally:
                tally[c] += 1
            else:
                fails.append((graph, done + k, c, imgs[k].shape, thr, bt, un))
        done += n_img
    for k in tally:
        total[k] += tally[k]
    print(f"== {graph}: {done} pages, {tally['identical']} identical, {tally['within budget']} within the float budget, {boxes} oracle boxes, "
          f"{marginal} pixels within 1e-4 of the threshold", flush=True)
n = sum(total.values()) + len(fails)
print(f"{n} soft-map pages: {total['identical']} identical, {total['within budget']} within the float budget, {len(fails)} FAIL in {time.time() - t0:.0f} s")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
