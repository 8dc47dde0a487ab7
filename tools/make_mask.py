import sys, numpy as np
sys.path.insert(0, ".")
from scipy.ndimage import uniform_filter
from oar_ocr_amd.synth import pages
pg = pages.make_page(0, (960, 960), 40)
d = 1.0 - pg[:, :, 0].astype(np.float32) / 255.0
b = uniform_filter(uniform_filter(d, 9), 9)
m = ((b > 0.25) * 255).astype(np.uint8)
print("fg fraction", (m > 0).mean())
m.tofile("gpurun_out/mask.bin")
