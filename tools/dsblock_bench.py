"""Times the fused depthwise-separable block kernel (csrc/dsblock.inc) on the bench's layer shapes through Seam A.
OAR_DSB_TIMING=1 adds per-phase shader-clock counters of workgroup 0 (stderr).  usage: python tools/dsblock_bench.py [case ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models

CASES = {  # name: (C, Cout, k, stride, N, H, W)
    "rec192": (192, 192, 5, (1, 1), 256, 12, 80), "rec96": (96, 96, 3, (1, 1), 256, 12, 160), "rec96s": (96, 192, 3, (1, 2), 256, 12, 160),
    "rec48": (48, 48, 3, (1, 1), 256, 24, 160), "rec48s": (48, 96, 3, (2, 1), 256, 24, 160), "rec24": (24, 48, 3, (1, 1), 256, 24, 160),
    "det16": (16, 24, 3, (1, 1), 9, 480, 480), "det48": (48, 48, 3, (1, 1), 9, 240, 240), "det32": (32, 32, 3, (1, 1), 9, 240, 240), "det64": (64, 64, 3, (1, 1), 9, 120, 120), "det24s": (24, 32, 3, (2, 2), 9, 480, 480), "det128": (128, 128, 5, (1, 1), 9, 60, 60), "srv128": (128, 128, 3, (1, 1), 64, 24, 400), "srv256": (256, 256, 3, (1, 1), 64, 12, 200), "srv128s": (128, 256, 3, (2, 2), 64, 24, 400),
}
for name in (sys.argv[1:] or list(CASES)):
    C, Cout, k, stride, N, H, W = CASES[name]
    net = models._Net("ds", seed=1, decomposed_hswish=False)
    g = net.g
    g.add_input("x", ["N", C, "H", "W"])
    y = net.conv("x", C, C, k, stride, groups=C, act="hswish")
    z = net.conv(y, C, Cout, 1, 1, act="hswish")
    g.nodes.append(models.node("Identity", [z], ["out"]))
    g.add_output("out", ["N", Cout, "H", "W"])
    eng = api.OrtInfer(g.model(), profile=True)
    x = np.random.default_rng(0).standard_normal((N, C, H, W)).astype(np.float32)
    eng.infer(x)
    api.prof_enable(True); api.prof_reset()
    for _ in range(3):
        eng.infer(x)
    for e in api.prof_snapshot():
        if e["launches"] and ("dsblock" in e["name"] or "conv" in e["name"]):
            ms = e["total_ms"] / e["launches"]
            print(f"{name:8s} {e['name']:20s} us/launch={ms * 1e3:8.1f} GB/s={e['alg_bytes'] / e['launches'] / ms / 1e6:8.1f} TF={e['alg_flops'] / e['launches'] / ms / 1e9:6.2f}", flush=True)
    api.prof_enable(False)
    eng.close()
