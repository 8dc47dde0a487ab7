"""Writes the outputs of single-conv graphs (bench-sized shapes) to an .npz so two builds / env settings can be diffed."""
import sys, os, numpy as np
sys.path.insert(0, ".")
from oar_ocr_amd import api
from oar_ocr_amd.synth.onnx_writer import GraphBuilder

def graph(kind, cin, cout, k, stride=1):
    g = GraphBuilder("b"); rng = np.random.default_rng(0)
    g.add_input("x", ["N", cin, "H", "W"])
    if kind == "conv":
        w = rng.standard_normal((cout, cin, k, k)).astype(np.float32) * 0.1
        y = g.op("Conv", ["x", g.init(w), g.init(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[k, k], strides=[stride, stride], pads=[k // 2] * 4, group=1, dilations=[1, 1])
        y = g.op("Relu", [y])
    else:
        w = rng.standard_normal((cin, cout, 2, 2)).astype(np.float32) * 0.1
        y = g.op("ConvTranspose", ["x", g.init(w), g.init(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[2, 2], strides=[2, 2])
    g.add_output(y, ["N", cout, "H2", "W2"])
    return g.model()

shapes = [("conv", 8, 64, 16, 240, 240, 3), ("conv", 8, 32, 64, 240, 240, 1), ("conv", 8, 24, 32, 240, 240, 1), ("conv", 8, 64, 64, 120, 120, 1),
          ("convt", 8, 16, 64, 240, 240, 2), ("conv", 8, 48, 48, 24, 160, 1), ("conv", 37, 192, 192, 12, 80, 1), ("conv", 5, 64, 16, 30, 30, 3), ("conv", 3, 20, 36, 17, 19, 3)]
out = {}
for i, (kind, n, cin, cout, h, w, k) in enumerate(shapes):
    eng = api.OrtInfer(graph(kind, cin, cout, k))
    x = np.random.default_rng(i).standard_normal((n, cin, h, w)).astype(np.float32)
    y = eng.infer(x)[0][1]
    out[f"s{i}"] = y
    print(i, kind, n, cin, cout, h, w, k, y.shape, float(np.abs(y).sum()), flush=True)
    eng.close()
np.savez(sys.argv[1], **out)
