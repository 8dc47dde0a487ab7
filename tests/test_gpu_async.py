"""oar_ocr_predict_async / oar_ocr_wait (oar_ocr_cfg.lanes): calls in flight on one handle return exactly what the synchronous
oar_ocr_predict returns for the same pages -- a call is still one OAROCR::predict (crops pooled over ITS pages, ocr.rs:594-634) on
one lane; tickets can be waited out of order; errors travel with their ticket."""
import numpy as np
import pytest

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets():
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    return det, rec, api.read_dict(models.synth_dict(6904))


def _same(a, b):
    assert len(a) == len(b)
    for pa, pb in zip(a, b):
        assert len(pa.text_regions) == len(pb.text_regions)
        for ta, tb in zip(pa.text_regions, pb.text_regions):
            assert np.array_equal(ta.bounding_box, tb.bounding_box) and ta.text == tb.text and ta.confidence == tb.confidence
            assert ta.rec_max_wh_ratio == tb.rec_max_wh_ratio and ta.rec_seq_len == tb.rec_seq_len


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_calls_in_flight_equal_synchronous_calls(nets, lanes):
    det, rec, chars = nets
    calls = [[pages.make_page(200 + 10 * c + i, (320 + 32 * (c % 3), 480), lines=4 + c) for i in range(1 + c % 4)] for c in range(7)]
    b = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(4).region_batch_size(16)
    sync = b.build()
    want = [sync.predict(c) for c in calls]
    sync.close()
    ocr = b.lanes(lanes).build()
    tickets = [ocr.submit(c) for c in calls]                 # everything queued before anything is collected
    got = {t: ocr.wait(t) for t in reversed(tickets)}        # collected out of order
    for t, w in zip(tickets, want):
        _same(got[t], w)
    _same(ocr.predict(calls[0]), want[0])                    # the synchronous entry still works on the same handle
    assert sum(len(p.text_regions) for w in want for p in w) > 60
    with pytest.raises(api.OCRError):
        ocr.wait(tickets[0])                                 # a ticket is collected exactly once
    ocr.close()


def test_an_error_travels_with_its_ticket(nets):
    det, rec, chars = nets
    ocr = api.OAROCRBuilder(det, rec, chars).lanes(2).build()
    good = [pages.make_page(300, (320, 480), lines=5)]
    imgs, ptrs, ws, hs = api._img_arrays(good)
    ws_bad = (type(ws))(*[0])                                # a zero-width page: "empty page" inside the worker
    t_bad = ocr.submit_packed(ptrs, ws_bad, hs, 1)
    t_ok = ocr.submit_packed(ptrs, ws, hs, 1)
    ok = ocr.wait_packed(t_ok, 1)
    assert len(ok.scores) > 0
    with pytest.raises(api.OCRError) as e:
        ocr.wait_packed(t_bad, 1)
    assert e.value.code == api.OAR_INVALID_INPUT
    ocr.close()
