"""Small valid files of every format the host decoders take, as seeds for tools/fuzz/fuzz_image_decoders.cc.
usage: python tools/fuzz/make_seeds.py <dir>"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

out = Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
rng = np.random.default_rng(0)
a = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
g = rng.integers(0, 256, (13, 17), dtype=np.uint8)
big = rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)
blocky = np.kron(rng.integers(0, 256, (8, 10, 3), dtype=np.uint8), np.ones((4, 4, 1), np.uint8))


def save(img, name, **kw):
    img.save(out / name, **kw)


rgb, grey, rgba = Image.fromarray(a), Image.fromarray(g), Image.fromarray(np.dstack([a, g]))
save(rgb, "rgb.bmp"); save(grey, "l.bmp"); save(rgb.convert("P"), "p.bmp"); save(Image.fromarray(g > 128), "1.bmp"); save(rgba, "rgba.bmp")
save(rgb, "rgb.ppm"); save(grey, "l.pgm"); save(Image.fromarray(g > 128), "1.pbm")
(out / "p3.ppm").write_text("P3\n3 2\n255\n" + " ".join(str(int(x)) for x in a[:2, :3].ravel()) + "\n")
(out / "p2.pgm").write_text("P2\n# c\n3 2\n65535\n" + " ".join(str(int(x) * 200) for x in g[:2, :3].ravel()) + "\n")
for comp in (None, "tiff_lzw", "packbits", "tiff_adobe_deflate"):
    kw = {} if comp is None else {"compression": comp}
    save(rgb, f"rgb_{comp}.tif", **kw); save(grey, f"l_{comp}.tif", **kw)
save(rgba, "rgba.tif", compression="tiff_lzw"); save(Image.fromarray(g.astype(np.uint16) * 257), "l16.tif")
save(Image.fromarray(big), "big_lzw.tif", compression="tiff_lzw")
save(rgb.convert("P"), "p.gif"); save(grey, "l.gif"); save(Image.fromarray(big).convert("P"), "big.gif", interlace=True)
rgb.convert("P").save(out / "anim.gif", save_all=True, append_images=[Image.fromarray(a[::-1]).convert("P")], transparency=0, disposal=2)
save(rgb, "rgb.png"); save(grey, "l.png"); save(rgb.convert("P"), "p.png"); save(rgba, "rgba.png"); save(Image.fromarray(g > 128), "1.png")
save(Image.fromarray(g.astype(np.uint16) * 257), "l16.png")
b = Image.fromarray(blocky)
save(b, "444.jpg", quality=90, subsampling=0); save(b, "420.jpg", quality=75, subsampling=2); save(b, "422.jpg", quality=75, subsampling=1)
save(b, "prog.jpg", quality=80, progressive=True); save(grey, "l.jpg", quality=85); save(b, "opt.jpg", quality=60, optimize=True)
save(b.convert("CMYK"), "cmyk.jpg", quality=80)
