"""Degenerate and extreme pages through oar_ocr_predict against the oracle pipeline (round 6): no pages, a 1 x 1 page, slivers three pixels wide or tall, all-black / all-white /
noise pages, a page larger than max_side_limit, one text line across a whole wide page, a page dense with short lines (many crops, several recognition batches), pages of very
different sizes in one call, in every batch policy.  usage: python tools/edge_pages.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import pipeline_ref

det, _ = models.build_det("tiny_full", seed=0)
rec, _ = models.build_rec("tiny_full", vocab=6906, seed=1)
chars = api.read_dict(models.synth_dict(6904))
rng = np.random.default_rng(5)
white = lambda h, w: np.full((h, w, 3), 255, np.uint8)
cases = {
    "no pages": [],
    "1 x 1 page": [white(1, 1)],
    "slivers": [white(3, 700), white(900, 3), pages.make_page(3, (40, 1500), 1)],
    "black / white / noise": [np.zeros((200, 300, 3), np.uint8), white(200, 300), rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)],
    "beyond max_side_limit": [pages.make_page(7, (4300, 900), 30)],
    "one line across a wide page": [pages.make_page(9, (120, 3000), 1)],
    "dense page": [pages.make_page(11, (1400, 1000), 60)],
    "mixed sizes in one call": [pages.make_page(13, (64, 64), 1), pages.make_page(14, (1200, 1600), 25), white(5, 5), pages.make_page(15, (300, 200), 6)],
}
bad = 0
t0 = time.time()
for name, imgs in cases.items():
    for ibs, rbs in ((1, 3), (8, 64), (32, 256)):
        try:
            ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(ibs).region_batch_size(rbs).build()
            got = ocr.predict(imgs)
            ocr.close()
            ref = pipeline_ref.OracleOCR(det, rec, chars, 0.3, 0.6, 1.5, image_batch_size=ibs, region_batch_size=rbs).predict(imgs)
            ok = len(got) == len(ref)
            nreg = 0
            for g, r in zip(got, ref):
                rep = pipeline_ref.compare_results(g, r)
                ok = ok and rep["ok"]
                nreg += len(r)
                if not rep["ok"]:
                    print("  MISMATCH", rep)
            msg = f"{len(imgs)} pages, {nreg} regions"
        except Exception as e:
            # the oracle and the product must then fail alike: run the oracle alone to see
            try:
                pipeline_ref.OracleOCR(det, rec, chars, 0.3, 0.6, 1.5, image_batch_size=ibs, region_batch_size=rbs).predict(imgs)
                ok, msg = False, f"product raised {type(e).__name__}: {str(e)[:160]} -- the oracle did not"
            except Exception as e2:
                ok, msg = True, f"both refuse: product {type(e).__name__}: {str(e)[:100]} | oracle {type(e2).__name__}: {str(e2)[:100]}"
        print(f"{'ok  ' if ok else 'FAIL'} [{name}] batches {ibs}/{rbs}: {msg}", flush=True)
        bad += 0 if ok else 1
print(f"{3 * len(cases) - bad}/{3 * len(cases)} edge cases agree with the oracle in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
