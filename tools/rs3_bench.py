"""conv3x3_n16_x6_kernel (igemm_rs3_x6.hip) alone at the DB head's shape: 8 images of 240 x 210, 64 -> 16.  usage: [OAR_RS3_DBG=mask] python tools/rs3_bench.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth.onnx_writer import GraphBuilder
g = GraphBuilder("conv")
rng = np.random.default_rng(0)
g.add_input("x", ["N", 64, "H", "W"])
w = (rng.standard_normal((16, 64, 3, 3)) / 24).astype(np.float32)
y = g.op("Conv", ["x", g.init(w), g.init(rng.standard_normal(16).astype(np.float32))], kernel_shape=[3, 3], strides=[1, 1], pads=[1, 1, 1, 1], group=1, dilations=[1, 1])
y = g.op("Relu", [y])
g.add_output(y, ["N", 16, "H", "W"])
eng = api.OrtInfer(g.model(), profile=True)
x = rng.standard_normal((8, 64, 240, 210)).astype(np.float32)
eng.infer(x)
api.prof_enable(True); api.prof_reset()
for _ in range(5):
    eng.infer(x)
for e in api.prof_snapshot():
    if e["launches"] and "conv" in e["name"]:
        print(f"{e['name']:20s} us/launch={e['total_ms'] * 1e3 / e['launches']:8.1f}")
