"""SURVEY 8f-3 -- load_image_from_memory / load_image(s) (oar-ocr-core/src/utils/image.rs:65-92, 299-345) for PNG input.

The files are written HERE by a small encoder (every colour type, bit depth, filter type, Adam7, split IDAT, ancillary chunks), so
the expected RGB8 pixels are known from the source arrays and the three conversion rules of the image crate; the product decoder
(csrc/image_decode.cc), the oracle (oracle/png_ref.py: zlib + numpy) and PIL (an independent implementation, where it implements
the same conversion) must all agree with them.  No GPU involved."""
import io
import struct
import zlib

import numpy as np
import pytest

from oar_ocr_amd import api
from oracle import png_ref

ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def chunk(kind: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def pack_row(samples: np.ndarray, depth: int) -> bytes:
    """samples: [n] ints of one scanline (all channels interleaved) -> the scanline's bytes"""
    if depth == 8:
        return samples.astype(np.uint8).tobytes()
    if depth == 16:
        return samples.astype(">u2").tobytes()
    bits = ((samples[:, None].astype(np.uint32) >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(-1)
    return np.packbits(bits).tobytes()


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_row(ft, cur: bytes, prev: bytes, bpp: int) -> bytes:
    out = bytearray(len(cur))
    for i in range(len(cur)):
        a = cur[i - bpp] if i >= bpp else 0
        b = prev[i] if prev else 0
        c = prev[i - bpp] if (prev and i >= bpp) else 0
        pred = (0, a, b, (a + b) // 2, paeth(a, b, c))[ft]
        out[i] = (cur[i] - pred) & 255
    return bytes(out)


def encode_png(samples: np.ndarray, color: int, depth: int, interlace=False, rng=None, plte=None, extra=(), idat_split=1):
    """samples: [H, W, channels] integer samples at `depth` bits (palette: indices).  Filters are chosen at random per scanline."""
    rng = rng or np.random.default_rng(0)
    h, w, ch = samples.shape
    assert ch == CHANNELS[color]
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    for x0, y0, dx, dy in (ADAM7 if interlace else ((0, 0, 1, 1),)):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for r in range(sub.shape[0]):
            cur = pack_row(sub[r].reshape(-1), depth)
            ft = int(rng.integers(0, 5))
            raw.append(ft)
            raw += filter_row(ft, cur, prev, bpp)
            prev = cur
    z = zlib.compress(bytes(raw), 6)
    cuts = [len(z) * k // idat_split for k in range(idat_split + 1)]
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    for kind, body in extra:
        out += chunk(kind, body)
    if plte is not None:
        out += chunk(b"PLTE", plte.astype(np.uint8).tobytes())
    for k in range(idat_split):
        out += chunk(b"IDAT", z[cuts[k]:cuts[k + 1]])
    return out + chunk(b"IEND", b"")


def expected_rgb8(samples, color, depth, plte=None):
    """the conversion rules of png (EXPAND) + image (to_rgb8), applied to the source samples"""
    s = samples.astype(np.uint32)
    if color == 3:
        return plte[s[:, :, 0]].astype(np.uint8)
    v = ((s + 128) // 257) if depth == 16 else s if depth == 8 else s * (255 // ((1 << depth) - 1))
    v = v.astype(np.uint8)
    return np.repeat(v[:, :, :1], 3, 2) if color in (0, 4) else v[:, :, :3]


CASES = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("color,depth", CASES)
@pytest.mark.parametrize("interlace", [False, True])
def test_png_decodes_to_the_image_crates_rgb8(color, depth, interlace):
    rng = np.random.default_rng(color * 100 + depth + (7 if interlace else 0))
    for (h, w) in ((1, 1), (5, 3), (9, 17), (33, 40)):
        ch = CHANNELS[color]
        plte = rng.integers(0, 256, (min(256, 1 << depth), 3)).astype(np.uint8) if color == 3 else None
        samples = rng.integers(0, 1 << depth, (h, w, ch))
        extra = ((b"gAMA", struct.pack(">I", 45455)), (b"tRNS", bytes(rng.integers(0, 256, len(plte) if color == 3 else 2 * (1 if color == 0 else 3)).astype(np.uint8))))
        if color in (4, 6):
            extra = extra[:1]                       # tRNS is not allowed next to a real alpha channel
        data = encode_png(samples, color, depth, interlace, rng, plte, extra, idat_split=int(rng.integers(1, 4)))
        want = expected_rgb8(samples, color, depth, plte)
        got = api.load_image_from_memory(data)
        assert got.shape == (h, w, 3) and got.dtype == np.uint8
        assert np.array_equal(got, want), (color, depth, interlace, h, w)
        assert np.array_equal(png_ref.decode_png_rgb8(data), want)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
def test_png_files_written_by_pil_and_pil_as_an_independent_decoder(mode):
    from PIL import Image
    rng = np.random.default_rng(len(mode))
    rgb = rng.integers(0, 256, (47, 61, 3)).astype(np.uint8)
    im = Image.fromarray(rgb).convert(mode)
    for opts in (dict(), dict(optimize=True), dict(compress_level=1)):
        buf = io.BytesIO()
        im.save(buf, format="PNG", **opts)
        data = buf.getvalue()
        got = api.load_image_from_memory(data)
        assert np.array_equal(got, png_ref.decode_png_rgb8(data))
        # PIL's own conversion to RGB agrees wherever alpha is not involved in it (PIL drops alpha like to_rgb8 does)
        assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def test_damaged_files_are_image_load_errors_not_pixels():
    rng = np.random.default_rng(3)
    good = encode_png(rng.integers(0, 256, (12, 12, 3)), 2, 8)
    assert api.load_image_from_memory(good).shape == (12, 12, 3)
    bad_crc = bytearray(good); bad_crc[40] ^= 1
    truncated = good[:len(good) - 20]
    no_idat = good[:33] + chunk(b"IEND", b"")
    short_stream = encode_png(rng.integers(0, 256, (12, 12, 3)), 2, 8)
    short_stream = short_stream[:16] + struct.pack(">II", 12, 13) + short_stream[24:29]     # IHDR says 13 rows, the data holds 12
    short_stream = short_stream[:12] + short_stream[12:29] + struct.pack(">I", zlib.crc32(short_stream[12:29])) + good[33:]
    for blob in (bytes(bad_crc), truncated, no_idat, short_stream, good[:8], b"", b"not an image at all"):
        with pytest.raises(api.OCRError) as e:
            api.load_image_from_memory(blob)
        assert e.value.code == api.OAR_INVALID_INPUT, blob[:16]


def test_formats_of_the_image_crate_that_are_not_decoded_here_say_so():
    for name, head in (("WebP", b"RIFF\0\0\0\0WEBPVP8 "), ("PNM", b"P7\nWIDTH 1\nHEIGHT 1\nDEPTH 3\nMAXVAL 255\nENDHDR\n\0\0\0"),
                       ("PNM", b"P6\n1 1\n1023\n\0\0\0\0\0\0"), ("BMP", b"BM" + struct.pack("<IHHI", 70, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 2, 2, 1, 24, 4, 16, 0, 0, 0, 0) + b"\0" * 16)):
        with pytest.raises(api.OCRError) as e:
            api.load_image_from_memory(head)
        assert e.value.code == api.OAR_UNSUPPORTED_OP and name in e.value.message


def test_load_images_sequential_and_parallel(tmp_path):
    rng = np.random.default_rng(9)
    arrays, paths = [], []
    for i in range(9):
        a = rng.integers(0, 256, (20 + i, 30 + 2 * i, 3))
        arrays.append(a.astype(np.uint8))
        p = tmp_path / f"page_{i}.png"
        p.write_bytes(encode_png(a, 2, 8, interlace=bool(i % 2), rng=rng))
        paths.append(p)
    assert np.array_equal(api.load_image(paths[0]), arrays[0])
    for thr in (None, 100, 0):          # above the default threshold of 4 -> parallel; forced sequential; forced parallel
        got = api.load_images(paths, parallel_threshold=thr)
        assert len(got) == 9 and all(np.array_equal(g, a) for g, a in zip(got, arrays))     # order preserved
    with pytest.raises(FileNotFoundError):
        api.load_images(paths + [tmp_path / "missing.png"])
    (tmp_path / "broken.png").write_bytes(paths[0].read_bytes()[:50])
    with pytest.raises(api.OCRError):
        api.load_images(paths + [tmp_path / "broken.png"])


def test_decoded_page_goes_straight_into_the_pipeline_types():
    """what load_image returns is what the predictors take: HxWx3 uint8, C-contiguous"""
    a = np.random.default_rng(1).integers(0, 256, (8, 8, 3))
    img = api.load_image_from_memory(encode_png(a, 2, 8))
    assert img.flags["C_CONTIGUOUS"] and img.dtype == np.uint8 and img.shape == (8, 8, 3)


def test_forged_header_is_rejected_before_any_allocation():
    """a 33-byte IHDR claiming 32768 x 32768 x RGBA16 (8 GiB of filtered data) used to make the decoder reserve the
    buffers before it had looked at IDAT.  Now it is refused by the 512 MiB allocation budget of image::Limits::default(), quickly."""
    import struct, time, zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    ihdr = struct.pack(">IIBBBBB", 32768, 32768, 16, 6, 0, 0, 0)
    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    t0 = time.time()
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(blob)
    assert time.time() - t0 < 1.0
    assert "512 MiB allocation limit" in str(e.value)


def test_header_larger_than_the_stream_is_corrupt_not_oom():
    """IHDR within the budget but an IDAT stream that ends early: `raw` grows with the inflated bytes only, the error is 'corrupt'."""
    import struct, zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    ihdr = struct.pack(">IIBBBBB", 8000, 8000, 8, 2, 0, 0, 0)     # 192 MB of RGB
    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 4096)) + chunk(b"IEND", b"")
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(blob)
    assert "corrupt or truncated" in str(e.value)


# ------------------------------------------------------------------------------------------------ JPEG (round 3)
def _jpeg_bytes(arr, **kw):
    import io
    from PIL import Image
    bio = io.BytesIO()
    Image.fromarray(arr).save(bio, "JPEG", **kw)
    return bio.getvalue()


def _pil_rgb(data):
    import io
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _test_image(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x * 3 + y) % 256, (x + y * 2) % 256, (x * y // 7) % 256], -1).astype(np.uint8)
    a[h // 4:h // 2, w // 4:w // 2] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 4, 3))
    return a


JPEG_SIZES = [(16, 16), (17, 23), (64, 48), (100, 133), (7, 5), (1, 1), (33, 129), (2, 300)]


@pytest.mark.parametrize("progressive", [False, True], ids=["baseline", "progressive"])
@pytest.mark.parametrize("subsampling", [0, 1, 2, "4:1:1", "4:4:0"], ids=["444", "422", "420", "411", "440"])
def test_jpeg_equals_libjpeg_turbo_bit_for_bit(subsampling, progressive):
    """JPEG decoding is not bit-specified; this decoder restates libjpeg's default path (islow IDCT, fancy h2v1 / h2v2 / h1v2 upsampling,
    replication elsewhere, ycc_rgb tables).  PIL decodes through libjpeg-turbo with those defaults: EXACT equality on every size /
    subsampling / mode / quality here, including partial MCUs, 1-pixel images and components too narrow for the fancy filters."""
    for (h, w) in JPEG_SIZES:
        for q in (35, 90, 100):
            try:
                data = _jpeg_bytes(_test_image(h, w, h * w + q), quality=q, subsampling=subsampling, progressive=progressive, optimize=(q == 90))
            except (ValueError, TypeError, KeyError):
                pytest.skip("this Pillow cannot write the subsampling")
            got = api.load_image_from_memory(data)
            assert np.array_equal(got, _pil_rgb(data)), (h, w, q)


def test_jpeg_grey_restart_intervals_and_large_image():
    from PIL import Image
    for (h, w) in JPEG_SIZES:
        g = _test_image(h, w, 3)[:, :, 0]
        for prog in (False, True):
            data = _jpeg_bytes(g, quality=80, progressive=prog)
            assert np.array_equal(api.load_image_from_memory(data), _pil_rgb(data))
    a = _test_image(240, 333, 5)
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1), dict(restart_marker_blocks=1, progressive=True)):
        try:
            data = _jpeg_bytes(a, quality=75, subsampling=2, **kw)
        except TypeError:
            continue
        assert b"\xff\xdd" in data                    # a DRI segment was really written
        assert np.array_equal(api.load_image_from_memory(data), _pil_rgb(data)), kw
    big = _test_image(960, 1280, 11)
    data = _jpeg_bytes(big, quality=85, subsampling=2)
    got = api.load_image_from_memory(data)
    assert np.array_equal(got, _pil_rgb(data))
    assert np.abs(got.astype(np.int32) - big).mean() < 40    # and it IS the picture (the test pattern is deliberately busy)


def test_jpeg_damage_and_undecoded_processes():
    a = _test_image(64, 80, 1)
    good = _jpeg_bytes(a, quality=80)
    for blob in (good[:200], good[:2], good[:-2][:len(good) // 2]):
        with pytest.raises(api.OCRError) as e:
            api.load_image_from_memory(blob)
        assert e.value.code == api.OAR_INVALID_INPUT
    from PIL import Image
    import io
    bio = io.BytesIO()
    Image.fromarray(a).convert("CMYK").save(bio, "JPEG")
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(bio.getvalue())
    assert e.value.code == api.OAR_UNSUPPORTED_OP and "CMYK" in e.value.message
    sof = good.index(b"\xff\xc0")
    arith = good[:sof] + b"\xff\xc9" + good[sof + 2:]          # SOF9: arithmetic coding
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(arith)
    assert e.value.code == api.OAR_UNSUPPORTED_OP and "arithmetic" in e.value.message
    huge = good[:sof + 5] + b"\xff\xff\xff\xff" + good[sof + 9:]     # 65535 x 65535: refused by the allocation budget, not attempted
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(huge)
    assert "512 MiB allocation limit" in e.value.message


def test_jpeg_scan_bomb_and_subset_scan_are_refused_quickly():
    """untrusted progressive files -- every extra SOS walks all blocks again and a reader that feeds zeros past the data never
    stops.  A file whose scans are repeated hundreds of times is refused by the scan cap / block budget in bounded time; data that ends
    before its scan does is a truncation error; an interleaved scan over a subset of the components (own MCU geometry, T.81 A.2.3) is named
    as unsupported rather than decoded with the frame's geometry."""
    import time
    a = _test_image(96, 128, 3)
    prog = _jpeg_bytes(a, quality=85, progressive=True)
    assert np.array_equal(api.load_image_from_memory(prog), _pil_rgb(prog))
    # the scans of the file: (offset of FF DA, Ss, Ah)
    scans, pos = [], 0
    while True:
        pos = prog.find(b"\xff\xda", pos)
        if pos < 0:
            break
        ns = prog[pos + 4]
        scans.append((pos, prog[pos + 5 + 2 * ns], prog[pos + 7 + 2 * ns] >> 4))
        pos += 2
    first = scans[0][0]
    eoi = prog.rindex(b"\xff\xd9")
    k = next(i for i, (_, ss, ah) in enumerate(scans) if ss == 0 and ah > 0)          # a DC refinement scan: legal to repeat (it ORs one bit)
    end = scans[k + 1][0] if k + 1 < len(scans) else eoi
    one_scan = prog[scans[k][0]:end]
    bomb = prog[:eoi] + one_scan * 400 + b"\xff\xd9"
    t0 = time.time()
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(bomb)
    assert e.value.code == api.OAR_INVALID_INPUT and ("scans" in e.value.message or "budget" in e.value.message), e.value.message
    assert time.time() - t0 < 5.0
    assert np.array_equal(api.load_image_from_memory(prog[:eoi] + one_scan * 3 + b"\xff\xd9"), _pil_rgb(prog))   # a few repeats are just a longer file
    # entropy-coded data cut inside the first scan, EOI appended: the scan would run on zeros
    cut = prog[:first + 40] + b"\xff\xd9"
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(cut)
    assert e.value.code == api.OAR_INVALID_INPUT
    base = _jpeg_bytes(a, quality=85)
    sos = base.index(b"\xff\xda")
    assert base[sos + 4] == 3
    subset = base[:sos + 2] + struct.pack(">H", 10) + bytes([2]) + base[sos + 5:sos + 9] + base[sos + 11:]
    with pytest.raises(api.OCRError) as e:
        api.load_image_from_memory(subset)
    assert e.value.code == api.OAR_UNSUPPORTED_OP and "subset" in e.value.message


def test_load_images_mixed_png_and_jpeg(tmp_path):
    """load_images (utils/image.rs:299-345) over a directory of both formats, sequential and parallel: same pages either way"""
    from PIL import Image
    want, paths = [], []
    for i in range(12):
        a = _test_image(40 + 3 * i, 60 + i, i)
        p = tmp_path / (f"p{i}.png" if i % 2 else f"p{i}.jpg")
        Image.fromarray(a).save(p, **({} if i % 2 else dict(quality=90, subsampling=i % 3, progressive=bool(i % 4))))
        paths.append(str(p))
        want.append(a if i % 2 else np.asarray(Image.open(p).convert("RGB")))
    for thr in (100, 2):
        got = api.load_images(paths, parallel_threshold=thr)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))


def _pil():
    return pytest.importorskip("PIL.Image")


def _pil_has_libtiff():
    from PIL import features
    return bool(features.check("libtiff"))


def test_bmp_pnm_gif_equal_pil_on_files_pil_wrote():
    """The small formats of image_misc_decode.cc (round 4) against PIL's decoder on files PIL encoded: BMP 1 / 8 (palette) / 24 / 32 bits, binary PNM
    (P4 / P5 / P6), GIF with global palettes (interlaced too) -- every pixel equal to Image.open(...).convert("RGB")."""
    import io
    Image = _pil()
    rng = np.random.default_rng(5)
    cases = []
    for (h, w) in ((1, 1), (7, 13), (33, 64), (50, 31)):
        rgbx = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
        pal = Image.fromarray(rgbx).quantize(colors=min(256, max(2, h * w)))
        bits = Image.fromarray((grey > 127).astype(np.uint8) * 255).convert("1")
        rgba = Image.fromarray(np.concatenate([rgbx, grey[..., None]], -1), "RGBA")
        cases += [("BMP", Image.fromarray(rgbx), {}), ("BMP", Image.fromarray(grey), {}), ("BMP", pal, {}), ("BMP", bits, {}), ("BMP", rgba, {}),
                  ("PPM", Image.fromarray(rgbx), {}), ("PPM", Image.fromarray(grey), {}), ("PPM", bits, {}),
                  ("GIF", pal, {}), ("GIF", Image.fromarray(grey), {}), ("GIF", pal, {"interlace": True}), ("GIF", bits, {})]
    for fmt, im, kw in cases:
        buf = io.BytesIO()
        im.save(buf, format=fmt, **kw)
        data = buf.getvalue()
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = api.load_image_from_memory(data)
        assert got.shape == want.shape and np.array_equal(got, want), (fmt, im.mode, im.size, kw)


def test_bmp_variants_pil_cannot_write():
    """Hand-built BMP files: 16-bit 5-5-5 and 5-6-5 (BI_BITFIELDS), top-down 24-bit, 4-bit palette, OS/2 core header, RLE8 with every escape --
    against PIL's decoder where PIL reads the variant, else against the pixels the file was built from."""
    import io
    Image = _pil()
    rng = np.random.default_rng(6)
    h, w = 5, 7

    def bmp(info, body, palette=b""):
        off = 14 + len(info) + len(palette)
        return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + info + palette + body

    def rows(arr, bpp_bytes):   # bottom-up rows padded to 4 bytes
        out = b""
        for y in range(arr.shape[0] - 1, -1, -1):
            r = arr[y].tobytes()
            out += r + b"\0" * (-len(r) % 4)
        return out
    px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # 5-5-5 (BI_RGB) and 5-6-5 (BI_BITFIELDS)
    for masks, comp in (((0x7C00, 0x03E0, 0x001F), 0), ((0xF800, 0x07E0, 0x001F), 3)):
        bits = [bin(m).count("1") for m in masks]
        shifts = [(m & -m).bit_length() - 1 for m in masks]
        q = [(px[..., c].astype(np.uint32) >> (8 - bits[c])) for c in range(3)]
        v = ((q[0] << shifts[0]) | (q[1] << shifts[1]) | (q[2] << shifts[2])).astype("<u2")
        info = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 16, comp, 0, 0, 0, 0, 0) + (struct.pack("<III", *masks) if comp == 3 else b"")
        data = bmp(info, rows(v, 2))
        want = np.stack([np.round(q[c] * 255.0 / ((1 << bits[c]) - 1)).astype(np.uint8) for c in range(3)], -1)
        got = api.load_image_from_memory(data)
        assert np.array_equal(got, want), masks
        assert np.abs(got.astype(int) - np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)).max() <= 1   # (PIL widens by bit replication)
    # top-down 24-bit
    info = struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 24, 0, 0, 0, 0, 0, 0)
    body = b"".join(px[y, :, ::-1].tobytes() + b"\0" * (-3 * w % 4) for y in range(h))
    data = bmp(info, body)
    assert np.array_equal(api.load_image_from_memory(data), px) and np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), px)
    # 4-bit palette, OS/2 core header (3-byte palette entries)
    palette = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    packed = b""
    for y in range(h - 1, -1, -1):
        r = bytearray((w + 1) // 2)
        for x in range(w):
            r[x // 2] |= int(idx[y, x]) << (4 if x % 2 == 0 else 0)
        packed += bytes(r) + b"\0" * (-len(r) % 4)
    for core in (False, True):
        info = struct.pack("<IHHHH", 12, w, h, 1, 4) if core else struct.pack("<IiiHHIIiiII", 40, w, h, 1, 4, 0, 0, 0, 0, 16, 0)
        pal_bytes = b"".join(bytes(c[::-1]) + (b"" if core else b"\0") for c in palette)
        data = bmp(info, packed, pal_bytes)
        want = palette[idx]
        assert np.array_equal(api.load_image_from_memory(data), want), core
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want)
    # RLE8: encoded runs, an absolute run (odd length: padded), a delta, end of line, end of bitmap before the last rows (left at palette entry 0 ... black)
    palette = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    palette[0] = 0
    want_idx = np.zeros((4, 8), np.uint8)
    stream = bytearray()
    stream += bytes([3, 9, 0, 3, 5, 6, 7, 0, 2, 11, 0, 0]); want_idx[3] = [9, 9, 9, 5, 6, 7, 11, 11]     # file row 0 = image row 3
    stream += bytes([0, 2, 2, 1, 4, 200, 0, 0]); want_idx[1, 2:6] = 200                                     # delta (+2, +1) from the start of file row 1 -> row 2
    stream += bytes([8, 77, 0, 1]); want_idx[0] = 77
    info = struct.pack("<IiiHHIIiiII", 40, 8, 4, 1, 8, 1, len(stream), 0, 0, 256, 0)
    data = bmp(info, bytes(stream), b"".join(bytes(c[::-1]) + b"\0" for c in palette))
    got = api.load_image_from_memory(data)
    assert np.array_equal(got, palette[want_idx])
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), got)
    # a SPARSE RLE8 page: 600 x 400 pixels in 18 bytes -- one run, end-of-line escapes, a delta and end-of-bitmap; every skipped pixel
    # stays at palette entry 0.  (Round 5's pre-check wanted two bytes per 255 pixels and refused such files as truncated.)
    sparse = bytes([5, 9, 0, 0, 0, 0, 0, 2, 10, 3, 2, 200, 0, 0, 0, 0, 0, 1])
    want_idx = np.zeros((400, 600), np.uint8)
    want_idx[399, 0:5] = 9
    want_idx[399 - 2 - 3, 10:12] = 200
    info = struct.pack("<IiiHHIIiiII", 40, 600, 400, 1, 8, 1, len(sparse), 0, 0, 256, 0)
    data = bmp(info, sparse, b"".join(bytes(c[::-1]) + b"\0" for c in palette))
    got = api.load_image_from_memory(data)
    assert np.array_equal(got, palette[want_idx])                    # (PIL's own RLE decoder refuses this file as "not enough image data"; the image crate zero-fills)
    with pytest.raises(api.OCRError):                                # a stream too short to even hold the end-of-bitmap marker
        api.load_image_from_memory(bmp(struct.pack("<IiiHHIIiiII", 40, 600, 400, 1, 8, 1, 1, 0, 0, 256, 0), b"\0", b"".join(bytes(c[::-1]) + b"\0" for c in palette)))


def test_ascii_pnm_and_gif_frame_inside_the_screen():
    import io
    Image = _pil()
    assert np.array_equal(api.load_image_from_memory(b"P3\n# a comment\n2 2\n255\n255 0 0  0 255 0\n0 0 255  9 8 7\n"), np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [9, 8, 7]]], np.uint8))
    assert np.array_equal(api.load_image_from_memory(b"P2 3 1 255 0 128 255"), np.array([[[0] * 3, [128] * 3, [255] * 3]], np.uint8))
    assert np.array_equal(api.load_image_from_memory(b"P1\n4 1\n1001"), np.array([[[0] * 3, [255] * 3, [255] * 3, [0] * 3]], np.uint8))
    # a GIF whose first frame covers only part of the logical screen: the rest is the transparent canvas -> black after to_rgb8
    frame = Image.fromarray(np.random.default_rng(2).integers(0, 256, (3, 4, 3), dtype=np.uint8)).quantize(colors=8)
    buf = io.BytesIO(); frame.save(buf, format="GIF"); g = bytearray(buf.getvalue())
    assert struct.unpack("<HH", g[6:10]) == (4, 3)
    g[6:10] = struct.pack("<HH", 9, 8)                               # logical screen 9 x 8
    at = g.index(b"\x2c")                                            # image descriptor
    g[at + 1:at + 5] = struct.pack("<HH", 2, 1)                      # frame at (2, 1)
    got = api.load_image_from_memory(bytes(g))
    want = np.zeros((8, 9, 3), np.uint8)
    want[1:4, 2:6] = np.asarray(frame.convert("RGB"))
    assert np.array_equal(got, want)
    for bad in (bytes(g[:at + 12]), b"BM" + struct.pack("<IHHI", 70, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 4, 4, 1, 24, 0, 0, 0, 0, 0, 0) + b"\0" * 16, b"P6\n2 2\n255\n\0", b"GIF89a" + b"\0" * 20):
        with pytest.raises(api.OCRError) as e:
            api.load_image_from_memory(bad)
        assert e.value.code == api.OAR_INVALID_INPUT, bad[:8]


def test_tiff_equals_pil_and_the_16_bit_rule():
    """Baseline TIFF (image_misc_decode.cc): strips, compression none / LZW / PackBits / Deflate, the horizontal predictor, grey / RGB / RGBA, 8 and
    16 bits, WhiteIsZero, big-endian -- against PIL where PIL produces 8-bit RGB, and against (v + 128) / 257 (DynamicImage::to_rgb8) for 16 bits."""
    import io
    Image = _pil()
    rng = np.random.default_rng(8)
    n = 0
    for (h, w) in ((1, 1), (9, 14), (67, 45)):
        rgbx = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        rgbx[h // 2:] = rgbx[h // 2][None]                       # runs, so the coders have something to do
        grey = rgbx[..., 0].copy()
        for im in (Image.fromarray(rgbx), Image.fromarray(grey), Image.fromarray(np.concatenate([rgbx, grey[..., None]], -1), "RGBA")):
            for kw in (dict(), dict(compression="tiff_lzw"), dict(compression="packbits"), dict(compression="tiff_adobe_deflate"),
                       dict(compression="tiff_lzw", tiffinfo={317: 2}), dict(compression="tiff_adobe_deflate", tiffinfo={317: 2})):
                buf = io.BytesIO()
                try:
                    im.save(buf, format="TIFF", **kw)
                except Exception:          # a PIL build without libtiff writes only uncompressed files
                    continue
                data = buf.getvalue()
                want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
                assert np.array_equal(api.load_image_from_memory(data), want), (im.mode, (h, w), kw)
                n += 1
    assert n >= 9 * (6 if _pil_has_libtiff() else 1)
    # 16-bit grey, little- and big-endian, hand-built (one strip, uncompressed); WhiteIsZero inverts
    v = rng.integers(0, 65536, (4, 5), dtype=np.uint16)
    for be in (False, True):
        for photo in (1, 0):
            e = ">" if be else "<"
            pix = v.astype(e + "u2").tobytes()
            tags = [(256, 3, 1, 5), (257, 3, 1, 4), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, photo), (273, 4, 1, 8), (277, 3, 1, 1), (278, 3, 1, 4), (279, 4, 1, len(pix))]
            ifd_at = 8 + len(pix)
            ifd = struct.pack(e + "H", len(tags))
            for tag, typ, cnt, val in tags:
                ifd += struct.pack(e + "HHI", tag, typ, cnt) + (struct.pack(e + "HH", val, 0) if typ == 3 else struct.pack(e + "I", val))
            data = (b"MM\0*" if be else b"II*\0") + struct.pack(e + "I", ifd_at) + pix + ifd + struct.pack(e + "I", 0)
            g8 = ((v.astype(np.uint32) + 128) // 257).astype(np.uint8)
            if photo == 0:
                g8 = 255 - g8
            assert np.array_equal(api.load_image_from_memory(data), np.repeat(g8[..., None], 3, -1)), (be, photo)
    # refused, by name: 1-bit, palette, tiles
    one_bit = io.BytesIO(); Image.fromarray((rng.random((8, 8)) > 0.5).astype(np.uint8) * 255).convert("1").save(one_bit, format="TIFF")
    pal = io.BytesIO(); Image.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).quantize(16).save(pal, format="TIFF")
    for blob in (one_bit.getvalue(), pal.getvalue()):
        with pytest.raises(api.OCRError) as ex:
            api.load_image_from_memory(blob)
        assert ex.value.code == api.OAR_UNSUPPORTED_OP and "TIFF" in ex.value.message
    with pytest.raises(api.OCRError) as ex:
        api.load_image_from_memory(b"II*\0" + b"\0" * 20)
    assert ex.value.code == api.OAR_INVALID_INPUT
