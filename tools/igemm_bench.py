"""Single-shape conv micro-benchmark through the engine (Seam A path, device-resident via the profiler)."""
import sys, os, numpy as np
sys.path.insert(0, ".")
from oar_ocr_amd import api
from oar_ocr_amd.synth.onnx_writer import GraphBuilder

def conv_graph(cin, cout, k=1):
    g = GraphBuilder("b")
    rng = np.random.default_rng(0)
    g.add_input("x", ["N", cin, "H", "W"])
    w = rng.standard_normal((cout, cin, k, k)).astype(np.float32) * 0.1
    y = g.op("Conv", ["x", g.init(w), g.init(np.zeros(cout, np.float32))], kernel_shape=[k, k], strides=[1, 1], pads=[k // 2] * 4, group=1, dilations=[1, 1])
    g.add_output(y, ["N", cout, "H", "W"])
    return g.model()

shapes = [(256, 192, 192, 12, 80, 1), (256, 256, 256, 6, 80, 1), (256, 48, 48, 24, 160, 1), (256, 96, 96, 12, 160, 1), (8, 64, 16, 240, 240, 3)]
if os.environ.get("SWEEP_K"):
    shapes = [(256, k, 192, 12, 80, 1) for k in (64, 128, 192, 256, 384, 512, 768)]
if os.environ.get("SWEEP_N"):
    shapes = [(256, 192, n, 12, 80, 1) for n in (64, 128, 192, 256, 384, 512)]
api.prof_enable(True)
for (n, cin, cout, h, w, k) in shapes:
    eng = api.OrtInfer(conv_graph(cin, cout, k))
    x = np.random.default_rng(1).standard_normal((n, cin, h, w)).astype(np.float32)
    for _ in range(2): eng.infer(x)
    api.prof_reset()
    for _ in range(5): eng.infer(x)
    snap = [e for e in api.prof_snapshot() if e["name"].startswith("conv_igemm")]
    e = snap[0]
    us = e["total_ms"] * 1e3 / e["launches"]
    print(f"M={n*h*w:8d} K={cin*k*k:4d} N={cout:4d}: {us:8.1f} us  {e['alg_flops']/e['launches']/us/1e6:6.1f} TF  {e['alg_bytes']/e['launches']/us/1e3:7.1f} GB/s", flush=True)
    eng.close()
