"""Prints a digest of three predicts over 24 pages of 640x640 (three or more detector sub-batches: the configuration in which the helper
enqueue thread is on by default).  tests/test_gpu_pipeline.py runs it with and without OAR_HIP_GRAPH=1 and compares the digests: with graph
replay the engine captures on the detector stream, so the helper thread must stay off (pipeline.cc helper_enqueues)."""
import hashlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages


def main():
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    imgs = [pages.make_page(300 + i, (640, 640), 24) for i in range(24)]
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).build()
    for rep in range(3):   # plans are captured on their second run and replayed from the third
        got = ocr.predict(imgs)
        h = hashlib.sha256()
        n = 0
        for g in got:
            for t in g.text_regions:
                h.update(np.asarray(t.bounding_box, np.float32).tobytes()); h.update(t.text.encode()); h.update(np.float32(t.confidence).tobytes())
                n += 1
        print(f"DIGEST {rep} {n} {h.hexdigest()}", flush=True)
    ocr.close()


if __name__ == "__main__":
    main()
