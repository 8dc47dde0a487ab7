"""Random PNG / BMP / PNM / GIF / TIFF files through oar_image_decode (host code: runs without a GPU) against PIL (round 6): PNG colour types x bit depths x interlace x
compression levels, palettes with and without tRNS, 16-bit samples, alpha; the other containers in the variants the decoder documents.  usage: python tools/png_fuzz.py [n] [seed]"""
import io, sys, time
sys.path.insert(0, ".")
import numpy as np
from PIL import Image
from oar_ocr_amd import api

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
compared, refused, unwritable = {}, {}, 0
t0 = time.time()
for case in range(n_cases):
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    mode = str(rng.choice(["RGB", "RGBA", "L", "LA", "P", "1", "I;16", "RGB", "L"]))
    fmt = str(rng.choice(["PNG", "PNG", "PNG", "BMP", "PPM", "GIF", "TIFF"]))
    if mode == "I;16":
        im = Image.fromarray(rng.integers(0, 65536, (h, w), dtype=np.uint16))
    elif mode == "1":
        im = Image.fromarray((rng.random((h, w)) < 0.5).astype(np.uint8) * 255).convert("1")
    elif mode == "P":
        im = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).quantize(int(rng.choice([2, 4, 16, 200, 256])))
    else:
        ch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        im = Image.fromarray(a[:, :, 0] if ch == 1 else a, mode)
    kw = {}
    if fmt == "PNG":
        kw = dict(compress_level=int(rng.integers(0, 10)), optimize=bool(rng.random() < 0.2))
        if mode == "P" and rng.random() < 0.4:
            kw["transparency"] = int(rng.integers(0, 2))
    elif fmt in ("BMP", "PPM", "GIF", "TIFF") and mode in ("RGBA", "LA", "I;16", "1") and fmt != "TIFF":
        im = im.convert("RGB") if mode != "1" or fmt == "GIF" else im
    if fmt == "GIF" and im.mode not in ("P", "L"):
        im = im.convert("P")
    if fmt == "PPM" and im.mode not in ("RGB", "L", "1"):
        im = im.convert("RGB")
    if fmt == "BMP" and im.mode not in ("RGB", "L", "P", "1"):
        im = im.convert("RGB")
    if fmt == "TIFF" and im.mode not in ("RGB", "L"):
        im = im.convert("RGB")
    bio = io.BytesIO()
    try:
        im.save(bio, fmt, **kw)
    except Exception as e:
        unwritable += 1
        continue
    data = bio.getvalue()
    src = Image.open(io.BytesIO(data))
    label = f"{fmt} {src.mode} {w}x{h} {kw}"
    try:
        got = api.load_image_from_memory(data)
    except api.OCRError as e:
        # documented refusals only (README: PNM maxval != 255, tiled / palette / 1-bit / fax TIFF, ...)
        if any(t in str(e) for t in ("not supported", "unsupported", "Unsupported")):
            refused[f"{fmt} {src.mode}"] = refused.get(f"{fmt} {src.mode}", 0) + 1
            continue
        bad += 1
        print(f"FAIL case {case} [{label}] {str(e)[:160]}", flush=True)
        continue
    if src.mode == "I;16":          # image crate: 16 -> 8 bit as (v + 128) / 257
        v = np.asarray(src).astype(np.uint32)
        want = np.repeat((((v + 128) // 257).astype(np.uint8))[:, :, None], 3, 2)
    else:
        want = np.asarray(src.convert("RGBA").convert("RGB") if False else src.convert("RGB"))
        if src.mode in ("RGBA", "LA", "P") and ("transparency" in src.info or src.mode in ("RGBA", "LA")):
            want = np.asarray(src.convert("RGBA"))[:, :, :3]      # alpha dropped, colour kept (image crate's to_rgb8)
    ok = got.shape == want.shape and np.array_equal(got, want)
    compared[f"{fmt} {src.mode}"] = compared.get(f"{fmt} {src.mode}", 0) + 1
    if not ok:
        bad += 1
        print(f"FAIL case {case} [{label}] shape {got.shape} vs {want.shape}, {int((got != want).sum()) if got.shape == want.shape else -1} bytes differ", flush=True)
print(f"{sum(compared.values()) - bad}/{sum(compared.values())} files compared byte for byte and equal ({n_cases} drawn; {unwritable} variants this Pillow cannot write; documented refusals: {refused}) in {time.time() - t0:.0f} s")
print("compared:", dict(sorted(compared.items())))
sys.exit(1 if bad else 0)
