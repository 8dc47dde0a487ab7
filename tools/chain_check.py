import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models
m, _ = models.build_rec("tiny", vocab=6906, seed=1)
for (n, W) in [(256, 320), (3, 184), (17, 160), (1, 8), (64, 640)]:
    x = np.random.default_rng(n).standard_normal((n, 3, 48, W)).astype(np.float32)
    os.environ["OAR_FUSE_CHAIN"] = "0"
    e0 = api.OrtInfer(m); y0 = e0.infer(x)[0][1]
    os.environ["OAR_FUSE_CHAIN"] = "1"
    e1 = api.OrtInfer(m); y1 = e1.infer(x)[0][1]
    d = np.abs(y0 - y1).max()
    am = (y0.argmax(-1) != y1.argmax(-1)).sum()
    ts = []
    for e in (e0, e1):
        e.infer(x); t = time.perf_counter()
        for _ in range(10): e.infer(x)
        ts.append((time.perf_counter() - t) / 10 * 1e3)
    print(f"n={n} W={W} max|diff|={d:.3e} argmax mismatches={am} of {y0.shape[0]*y0.shape[1]}  ms unfused={ts[0]:.3f} chain={ts[1]:.3f}", flush=True)
