"""The GPU half of JPEG decoding (jpeg.hip: dequantise + islow IDCT + fancy upsampling + ycc_rgb) against the host half of the same
library (== PIL, tests/test_image_decode_cpu.py) and against PIL directly: the page decoded INTO HBM is bit-identical, and feeds
oar_ocr_predict_device without a host copy of the pixels."""
import io

import numpy as np
import pytest
from PIL import Image

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages

pytestmark = pytest.mark.gpu


def _download(buf, w, h):
    out = np.empty((h, w, 3), np.uint8)
    api._check(api.lib().oar_dev_download(api._p(out), buf.ptr, out.nbytes))
    return out


def _jpeg(arr, **kw):
    bio = io.BytesIO()
    Image.fromarray(arr).save(bio, "JPEG", **kw)
    return bio.getvalue()


@pytest.mark.parametrize("progressive", [False, True])
def test_device_decode_equals_host_decode_and_pil(progressive):
    rng = np.random.default_rng(0)
    for (h, w) in [(16, 16), (17, 23), (100, 133), (1, 1), (2, 300), (481, 640)]:
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(x * 3 + y) % 256, (x + y * 2) % 256, (x * y // 7) % 256], -1).astype(np.uint8)
        a[h // 4:h // 2, w // 4:w // 2] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 4, 3))
        for sub in (0, 1, 2, "4:4:0"):
            for src in (a, a[:, :, 1]):
                kw = dict(quality=85, progressive=progressive)
                if src.ndim == 3:
                    kw["subsampling"] = sub
                try:
                    data = _jpeg(src, **kw)
                except (TypeError, ValueError, KeyError):
                    continue   # this Pillow cannot write the subsampling
                buf, dw, dh = api.load_image_to_device(data)
                got = _download(buf, dw, dh)
                buf.free()
                assert (dh, dw) == (h, w)
                assert np.array_equal(got, api.load_image_from_memory(data))
                assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def test_png_goes_through_the_same_entry():
    a = np.random.default_rng(1).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    bio = io.BytesIO()
    Image.fromarray(a).save(bio, "PNG")
    buf, w, h = api.load_image_to_device(bio.getvalue())
    assert np.array_equal(_download(buf, w, h), a)
    buf.free()


def test_jpeg_pages_decoded_into_hbm_feed_the_pipeline():
    """encoded pages -> oar_image_decode_device -> oar_ocr_predict_device: same regions as decoding on the host and calling predict"""
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    blobs = [_jpeg(pages.make_page(60 + i, (320, 480), lines=6), quality=92, subsampling=2) for i in range(3)]
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(4).region_batch_size(16).build()
    host = ocr.predict([api.load_image_from_memory(b) for b in blobs])
    bufs = [api.load_image_to_device(b) for b in blobs]
    dev = ocr.predict_device([int(b[0].ptr.value) for b in bufs], [b[1] for b in bufs], [b[2] for b in bufs])
    assert sum(len(r.text_regions) for r in host) > 10
    for a, b in zip(host, dev):
        assert len(a.text_regions) == len(b.text_regions)
        for ta, tb in zip(a.text_regions, b.text_regions):
            assert np.array_equal(ta.bounding_box, tb.bounding_box) and ta.text == tb.text and ta.confidence == tb.confidence
    for b in bufs:
        b[0].free()
    ocr.close()
