"""Oracle for SURVEY 8f-3 -- load_image_from_memory (oar-ocr-core/src/utils/image.rs:65-68) on PNG input.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by oar_ocr_amd/).  The reference is `image::load_from_memory(bytes)` followed
by `DynamicImage::to_rgb8()`: the `png` crate with Transformations::EXPAND (image 0.25.6, codecs/png.rs) and image's colour
conversions.  Neither crate is vendored in the reference tree; what they compute is fixed by the PNG specification (lossless) plus
three documented conversion rules, restated here in numpy on top of Python's zlib:

    palette            -> RGB through PLTE (tRNS becomes alpha, which to_rgb8 drops)
    grey 1 / 2 / 4 bit -> 8 bit by v * (255 / (2^depth - 1))                         (png: expand_gray_u8)
    16 bit -> 8 bit    -> (v + 128) / 257                                            (image: FromPrimitive<u16> for u8)
    grey -> RGB by replication, alpha dropped without pre-multiplication             (image: FromColor)

Pinned in tests/test_image_decode_cpu.py against PIL (an independent decoder) wherever PIL implements the same conversion, and
against the source pixels of the generated files everywhere.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))   # x0, y0, dx, dy
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


class PngError(ValueError):
    pass


def _chunks(data: bytes):
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise PngError("signature")
    pos = 8
    while pos + 12 <= len(data):
        (n,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n or pos + 12 + n > len(data):
            raise PngError("truncated chunk")
        if zlib.crc32(kind + body) != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise PngError("crc")
        yield kind, body
        pos += 12 + n
        if kind == b"IEND":
            return
    raise PngError("no IEND")


def _unfilter(ft, cur, prev, bpp):
    cur = cur.astype(np.int32)
    n = len(cur)
    prev = np.zeros(n, np.int32) if prev is None else prev.astype(np.int32)
    out = np.zeros(n, np.int32)
    for i in range(n):          # the filters are defined byte by byte on already reconstructed neighbours
        a = out[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) // 2
        elif ft == 4:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        else:
            raise PngError("filter type")
        out[i] = (cur[i] + pred) & 255
    return out.astype(np.uint8)


def _samples(row: np.ndarray, count: int, depth: int) -> np.ndarray:
    if depth == 8:
        return row[:count].astype(np.uint32)
    if depth == 16:
        return (row[:count * 2:2].astype(np.uint32) << 8) | row[1:count * 2:2]
    bits = np.unpackbits(row)                                   # MSB first
    return bits[:count * depth].reshape(count, depth).dot(1 << np.arange(depth - 1, -1, -1)).astype(np.uint32)


def decode_png_rgb8(data: bytes) -> np.ndarray:
    """[H, W, 3] u8, or PngError."""
    hdr, plte, idat = None, None, []
    for kind, body in _chunks(data):
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
    if hdr is None or not idat:
        raise PngError("missing chunks")
    w, h, depth, color, _comp, _flt, interlace = hdr
    ch = _CHANNELS[color]
    try:
        raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    except zlib.error as e:
        raise PngError(str(e))
    bpp = max(1, ch * depth // 8)
    out = np.zeros((h, w, 3), np.uint8)
    passes = _ADAM7 if interlace else ((0, 0, 1, 1),)
    at = 0
    for x0, y0, dx, dy in passes:
        pw = (w - x0 + dx - 1) // dx if w > x0 else 0
        ph = (h - y0 + dy - 1) // dy if h > y0 else 0
        if pw == 0 or ph == 0:
            continue
        rb = (pw * ch * depth + 7) // 8
        prev = None
        for r in range(ph):
            if at + 1 + rb > len(raw):
                raise PngError("short image data")
            line = _unfilter(int(raw[at]), raw[at + 1:at + 1 + rb], prev, bpp)
            prev = line
            at += 1 + rb
            s = _samples(line, pw * ch, depth).reshape(pw, ch)
            if color == 3:
                if plte is None or s.max() >= len(plte):
                    raise PngError("palette")
                rgb = plte[s[:, 0]]
            else:
                if depth == 16:
                    v = ((s + 128) // 257).astype(np.uint8)
                elif depth == 8:
                    v = s.astype(np.uint8)
                else:
                    v = (s * (255 // ((1 << depth) - 1))).astype(np.uint8)
                rgb = np.repeat(v[:, :1], 3, 1) if color in (0, 4) else v[:, :3]
            out[y0 + r * dy, x0::dx][:pw] = rgb
    if at != len(raw):
        raise PngError("excess image data")
    return out
