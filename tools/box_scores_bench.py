"""box_scores_kernel alone on the bench's boxes: 9 pages of 960 x 960, the detector's probability maps, the host tracer's candidate boxes.
usage: [OAR_BOX_SCORES_LDS=0|1] python tools/box_scores_bench.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from oar_ocr_amd import api
rng = np.random.default_rng(0)
H = W = 960
pred = rng.random((H, W), dtype=np.float32)
boxes = []
for i in range(360):
    x0, y0 = rng.uniform(20, 200), 20 + (i % 40) * 23
    w, h = rng.uniform(300, 720), rng.uniform(14, 34)
    t = rng.uniform(-0.01, 0.01)
    c, s = np.cos(t), np.sin(t)
    q = np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float32) @ np.array([[c, s], [-s, c]], np.float32) + [x0, y0]
    boxes.append(q.reshape(8))
boxes = np.stack(boxes).astype(np.float32)
api.k_box_scores(pred, boxes)
api.prof_enable(True); api.prof_reset()
for _ in range(20):
    r = api.k_box_scores(pred, boxes)
for e in api.prof_snapshot():
    if e["launches"]:
        print(f"{e['name']:30s} n={e['launches']} us/launch={e['total_ms'] * 1e3 / e['launches']:.1f}")
print("checksum", float(r.sum()))
