"""Masks for the host-side microbenchmarks (tools/host_bench.cc), written under gpurun_out/.
  python tools/make_mask.py            -> mask.bin: one 960 x 960 stand-in (blurred page, no network)
  python tools/make_mask.py --oracle N -> masksN.bin (+ probsN.bin): the oracle detector's thresholded probability maps of bench pages 0..N-1
                                          (torch CPU, ~1 s per page): what the detector's host stage really sees (37 % foreground, ~330 contours per page)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oar_ocr_amd.synth import pages

if len(sys.argv) > 2 and sys.argv[1] == "--oracle":
    from oar_ocr_amd.synth import models
    from oracle import pipeline_ref
    n = int(sys.argv[2])
    det, _ = models.build_det("tiny", seed=0)
    pm = pipeline_ref.OracleDetector(det).prob_maps([pages.make_page(i, (960, 960), 40) for i in range(n)])
    masks = np.stack([(p > 0.3).astype(np.uint8) * 255 for p, _ in pm])
    print(masks.shape, "fg fraction", (masks > 0).mean())
    masks.tofile(f"gpurun_out/masks{n}.bin")
    np.stack([p for p, _ in pm]).astype(np.float32).tofile(f"gpurun_out/probs{n}.bin")
else:
    from scipy.ndimage import uniform_filter
    pg = pages.make_page(0, (960, 960), 40)
    d = 1.0 - pg[:, :, 0].astype(np.float32) / 255.0
    b = uniform_filter(uniform_filter(d, 9), 9)
    m = ((b > 0.25) * 255).astype(np.uint8)
    print("fg fraction", (m > 0).mean())
    m.tofile("gpurun_out/mask.bin")
