"""One warm recognizer batch (256 crops of 48 x 320) and one warm detector sub-batch (8 pages of 960^2) under the in-library profiler with
per-shape classes: the launch list of one graph run, in order of total time.  usage: python tools/rec_trace.py [rec|det]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
what = sys.argv[1] if len(sys.argv) > 1 else "rec"
if what == "rec":
    m, _ = models.build_rec("tiny", vocab=6906, seed=1)
    import os
    x = np.random.default_rng(0).standard_normal((int(os.environ.get("REC_N", "256")), 3, 48, 320)).astype(np.float32)
else:
    m, _ = models.build_det(__import__("os").environ.get("DET_SIZE", "tiny"), seed=0)
    x = np.random.default_rng(0).standard_normal((8, 3, 960, 960)).astype(np.float32)
eng = api.OrtInfer(m, profile=True)
eng.infer(x)
api.prof_enable(True); api.prof_reset()
eng.infer(x)
snap = api.prof_snapshot()
tot = sum(e["total_ms"] for e in snap)
print(what, "kernel ms", round(tot, 3), "launches", sum(e["launches"] for e in snap))
for e in snap:
    ms = e["total_ms"]
    print(f"{e['name']:64s} n={e['launches']:4d} ms={ms:8.3f} us/launch={ms * 1e3 / max(e['launches'], 1):8.1f} GB/s={e['alg_bytes'] / ms / 1e6 if ms else 0:8.1f}")
