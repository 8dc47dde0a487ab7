"""Random JPEG files through oar_image_decode (host) and oar_image_decode_device (Huffman on the host, IDCT + upsampling + colour in jpeg.hip) against PIL / libjpeg-turbo,
byte for byte (round 6): random sizes down to 1 x 1, noise / gradient / page content, quality 1 ... 100, all chroma subsamplings PIL writes, grey files, baseline and
progressive, optimised tables, restart intervals.   usage: python tools/jpeg_fuzz.py [n_cases] [seed]"""
import io, sys, time
sys.path.insert(0, ".")
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 25   # (PIL's encoder buffer: quality 100 + optimize on a noise page overruns the default)
from oar_ocr_amd import api
from oar_ocr_amd.synth import pages

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def download(buf, w, h):
    out = np.empty((h, w, 3), np.uint8)
    api._check(api.lib().oar_dev_download(api._p(out), buf.ptr, out.nbytes))
    return out


bad = 0
t0 = time.time()
for case in range(n_cases):
    small = [1, 2, 7, 8, 9, 15, 16, 17, 31, 33]
    h = int(rng.integers(1, 700)) if rng.random() < 0.6 else int(rng.choice(small))
    w = int(rng.integers(1, 900)) if rng.random() < 0.6 else int(rng.choice(small))
    kind = str(rng.choice(["noise", "gradient", "page", "flat"]))
    if kind == "noise":
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == "gradient":
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(x * 3 + y) % 256, (x + y * 2) % 256, (x * y // 7) % 256], -1).astype(np.uint8)
    elif kind == "page":
        a = pages.make_page(int(rng.integers(0, 1 << 30)), (max(h, 48), max(w, 48)), int(rng.integers(0, 12)))[:h, :w]
    else:
        a = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
    grey = bool(rng.random() < 0.2)
    kw = dict(quality=int(rng.choice([1, 10, 50, 75, 85, 95, 100])), progressive=bool(rng.random() < 0.4), optimize=bool(rng.random() < 0.3))
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))   # 4:4:4, 4:2:2, 4:2:0 (what this Pillow writes)
    if rng.random() < 0.25:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    src = np.ascontiguousarray(a[:, :, 1]) if grey else np.ascontiguousarray(a)
    bio = io.BytesIO()
    try:
        Image.fromarray(src).save(bio, "JPEG", **kw)
    except Exception as e:      # a PIL build that does not write this variant
        kw.pop("restart_marker_blocks", None)
        if kw.get("subsampling") == "4:1:1":
            kw["subsampling"] = 2
        bio = io.BytesIO()
        Image.fromarray(src).save(bio, "JPEG", **kw)
    data = bio.getvalue()
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    try:
        host = api.load_image_from_memory(data)
        buf, dw, dh = api.load_image_to_device(data)
        dev = download(buf, dw, dh)
        buf.free()
        ok = np.array_equal(host, want) and (dw, dh) == (w, h) and np.array_equal(dev, want)
        msg = "" if ok else f"host equal {np.array_equal(host, want)} device equal {np.array_equal(dev, want)}"
    except Exception as e:
        ok, msg = False, f"{type(e).__name__}: {str(e)[:160]}"
    if not ok:
        bad += 1
        print(f"FAIL case {case} [{kind} {w}x{h} grey {grey} {kw}] {msg}", flush=True)
print(f"{n_cases - bad}/{n_cases} JPEG files decode to PIL's bytes (host and GPU paths) in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
