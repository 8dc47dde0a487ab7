"""Depthwise-conv micro-benchmark at the bench's layer shapes, through the engine (profiler timing per launch)."""
import sys, numpy as np
sys.path.insert(0, ".")
from oar_ocr_amd import api
from oar_ocr_amd.synth.onnx_writer import GraphBuilder


def dw_graph(c, k, strides):
    g = GraphBuilder("dw")
    rng = np.random.default_rng(0)
    g.add_input("x", ["N", c, "H", "W"])
    w = rng.standard_normal((c, 1, k, k)).astype(np.float32) * 0.1
    y = g.op("Conv", ["x", g.init(w), g.init(np.zeros(c, np.float32))], kernel_shape=[k, k], strides=list(strides), pads=[k // 2] * 4, group=c, dilations=[1, 1])
    y = g.op("HardSwish", [y])
    g.add_output(y, ["N", c, "H", "W"])
    return g.model()


#          n    c   h    w   k  strides       (recognizer batch 256 / detector sub-batch 9 layers of the bench graphs)
shapes = [(256, 192, 6, 160, 5, (1, 1)), (256, 256, 3, 160, 5, (1, 1)), (256, 192, 6, 160, 5, (2, 1)), (256, 96, 12, 160, 3, (1, 1)),
          (256, 48, 24, 160, 3, (1, 1)), (9, 16, 480, 480, 3, (1, 1)), (9, 24, 480, 480, 3, (2, 2)), (9, 128, 60, 60, 5, (1, 1))]
api.prof_enable(True)
for (n, c, h, w, k, st) in shapes:
    eng = api.OrtInfer(dw_graph(c, k, st))
    x = np.random.default_rng(1).standard_normal((n, c, h, w)).astype(np.float32)
    for _ in range(2):
        eng.infer(x)
    api.prof_reset()
    for _ in range(6):
        eng.infer(x)
    e = [e for e in api.prof_snapshot() if e["name"].startswith("conv_dw")][0]
    us = e["total_ms"] * 1e3 / e["launches"]
    print(f"n={n:3d} C={c:3d} {h:3d}x{w:3d} k{k} s{st}: {us:8.1f} us  {e['alg_bytes']/e['launches']/us/1e3:7.1f} GB/s", flush=True)
    eng.close()
