"""End-to-end parity at the sizes BASELINE.json quotes ("no -m gpu test runs BASELINE C2 at its real size";
C3 had only Seam-A graph checks at 256x320).

C2  PP-OCRv6-tiny-class det+rec, ONE predict of 32 pages of 960x960 (image batch 32, region batch 256: the bench's
    workload, with its 8/9-page detector sub-batches, the M >= 100 k kernel selections and ~1100 pooled crops);
    8 pages -- at least one from each detector sub-batch -- are checked against the oracle (round 5: 4 -> 8; C3 3 -> 5; C4 rank 4 -> 6).
C3  PP-OCRv5-server-class det + SVTR rec (V = 18710) through OAROCR.predict, 64 pages of 1280x1280 in one call with limit_side_len = 1280
    (the reference needs that setting to really run 1280^2, src/oarocr/ocr.rs:351-363), 5 pages checked.
C4  rank 0 of 8's shard of the 1024-page list (128 pages, 2 host threads) through predict_packed -> oar_ocr_pack -> oar_packed_merge.

Bar: boxes bit-exact, region order identical, recognition scores within 1e-3, texts equal unless the oracle's own top-2
probabilities tie within 1e-5 at some time step."""
import numpy as np
import pytest

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import cpu_ref as R
from oracle import pipeline_ref

pytestmark = pytest.mark.gpu


def _check_pages_against_oracle(got, imgs, check, det, rec, chars, det_kw, thresholds):
    """got: OAROCRResult list of ALL pages (crops pooled over all of them); check: page indices verified with the oracle.
    Detection / sorting / cropping of a page do not depend on the other pages.  Recognition does, through the padding
    width of its batch only (crnn.rs:80-87): every sample of the network is independent, so the oracle recognises the
    checked crops padded to the width of the batch the pipeline put them in (TextRegion.rec_max_wh_ratio)."""
    od = pipeline_ref.OracleDetector(det, **det_kw)
    orec = pipeline_ref.OracleRecognizer(rec, chars)
    n_regions = n_ties = 0
    for pi in check:
        (boxes, scores, prob), = od.detect([imgs[pi]], *thresholds)
        order = R.sort_quad_boxes(boxes)
        regs = got[pi].text_regions
        slots = []
        for o in order:
            crop = R.rotate_crop(imgs[pi], boxes[o])
            if crop is not None:
                slots.append((boxes[o], float(scores[o]), crop))
        marginal = int((np.abs(prob - thresholds[0]) < 1e-4).sum())
        assert len(regs) == len(slots), (pi, len(regs), len(slots), marginal)
        by_width = {}
        for k, (box, sc, crop) in enumerate(slots):
            g = regs[k]
            assert np.array_equal(np.asarray(g.bounding_box, np.float32), box), (pi, k, g.bounding_box, box, marginal)
            assert abs(g.det_score - sc) <= 1e-3
            assert g.crop_wh == (crop.shape[1], crop.shape[0])
            by_width.setdefault(np.float32(g.rec_max_wh_ratio).item(), []).append(k)
        for mwh, ks in by_width.items():
            r = orec.recognize([slots[k][2] for k in ks], batch_max_wh_ratio=mwh)
            for j, k in enumerate(ks):
                g = regs[k]
                assert g.rec_seq_len == r["idx"].shape[1]
                assert abs(g.confidence - r["scores"][j]) <= 1e-3, (pi, k, g.confidence, r["scores"][j])
                if g.text != r["texts"][j]:   # only where the oracle's own top-2 tie at some time step
                    top2 = np.sort(r["probs_full"][j], axis=1)[:, -2:]
                    assert (top2[:, 1] - top2[:, 0]).min() < 1e-5, (pi, k, g.text, r["texts"][j])
                    n_ties += 1
                n_regions += 1
    return n_regions, n_ties


def test_c2_gpu_border_follower_gives_the_same_pages():
    """BASELINE C2 size with oar_det_cfg.gpu_contours = 1: every box, text and score equal to the default (host-traced) run."""
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    imgs = [pages.make_page(100 + i, (960, 960), 40) for i in range(12)]
    runs = []
    for gpu in (False, True):
        cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5, gpu_contours=gpu)
        ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).build()
        runs.append(ocr.predict(imgs))
        ocr.close()
    assert sum(len(g.text_regions) for g in runs[0]) > 300
    for a, b in zip(*runs):
        assert len(a.text_regions) == len(b.text_regions)
        for ta, tb in zip(a.text_regions, b.text_regions):
            assert np.array_equal(ta.bounding_box, tb.bounding_box) and ta.text == tb.text and ta.confidence == tb.confidence


def test_c2_32_pages_of_960x960_in_one_predict():
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    imgs = [pages.make_page(i, (960, 960), 40) for i in range(32)]
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).build()
    got = ocr.predict(imgs)
    assert len(got) == 32
    total = sum(len(g.text_regions) for g in got)
    assert total > 900                                            # ~1090 regions: several 256-crop recognition batches
    assert len({round(t.rec_max_wh_ratio, 4) for g in got for t in g.text_regions}) >= 3
    n, ties = _check_pages_against_oracle(got, imgs, [0, 3, 9, 14, 20, 25, 28, 31], det, rec, chars, {}, (0.3, 0.6, 1.5))
    assert n > 200 and ties <= 3
    # the packed metric path of bench.py returns the same boxes / texts / scores
    _, ptrs, ws, hs = api._img_arrays(imgs)
    packed = ocr.predict_packed(ptrs, ws, hs, 32)
    assert packed.region_offsets.tolist() == np.concatenate([[0], np.cumsum([len(g.text_regions) for g in got])]).tolist()
    k = 0
    for g in got:
        for t in g.text_regions:
            assert np.array_equal(packed.points[k], t.bounding_box) and packed.text(k) == t.text and packed.scores[k] == np.float32(t.confidence)
            k += 1
    ocr.close()


def test_c2_32_pages_on_graphs_of_the_files_sizes():
    """BASELINE C2 on the graphs bench.py times by default since round 6: detector 447 089 parameters (pp-ocrv6_tiny_det.onnx:
    1 780 590 bytes), recognizer 1 135 700 (pp-ocrv6_tiny_rec.onnx: 4 462 639 bytes; CTC head K = 96 on ctc_head_x6_kernel<3>) -- 32 pages of 960 x 960 in one predict, eight against the oracle."""
    det, di = models.build_det("tiny_full", seed=0)
    rec, ri = models.build_rec("tiny_full", vocab=6906, seed=1)
    assert abs(di["params"] * 4 / 1780590 - 1) < 0.02 and abs(ri["params"] * 4 / 4462639 - 1) < 0.02
    chars = api.read_dict(models.synth_dict(6904))
    imgs = [pages.make_page(i, (960, 960), 40) for i in range(32)]
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).build()
    got = ocr.predict(imgs)
    assert len(got) == 32 and sum(len(g.text_regions) for g in got) > 900
    n, ties = _check_pages_against_oracle(got, imgs, [0, 3, 9, 14, 20, 25, 28, 31], det, rec, chars, {}, (0.3, 0.6, 1.5))
    assert n > 200 and ties <= 3
    ocr.close()


def test_c3_server_graphs_on_1280x1280_pages():
    det, _ = models.build_det("server", seed=0)
    rec, _ = models.build_rec("server", vocab=18710, seed=1)
    chars = api.read_dict(models.synth_dict(18708))
    imgs = [pages.make_page(100 + i, (1280, 1280), 40) for i in range(64)]     # BASELINE C3 as stated: batch = 64 pages in ONE predict
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5, limit_side_len=1280)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(64).region_batch_size(64).build()
    got = ocr.predict(imgs)
    assert len(got) == 64 and sum(len(g.text_regions) for g in got) > 1500
    n, ties = _check_pages_against_oracle(got, imgs, [0, 17, 30, 46, 63], det, rec, chars, dict(limit_side_len=1280), (0.3, 0.6, 1.5))
    assert n > 60 and ties <= 2
    ocr.close()


def test_c3_named_graphs_on_1280x1280_pages():
    """BASELINE C3 on graphs of the size and kind it names (VERDICT r5 next #1a): PP-HGNetV2 / LK-PAN detector (21.7 M parameters) at 1280 x 1280
    (limit_side_len = 1280, src/oarocr/ocr.rs:351-363) + SVTRv2 recognizer (20.5 M, V = 6625), 64 pages in ONE predict; five pages against the oracle
    (boxes bit-exact, texts, scores <= 1e-3)."""
    det, di = models.build_det("server_hgnet", seed=0)
    rec, ri = models.build_rec("svtrv2", vocab=6625, seed=1)
    assert di["params"] > 21e6 and ri["params"] > 20e6
    chars = api.read_dict(models.synth_dict(6623))
    imgs = [pages.make_page(100 + i, (1280, 1280), 40) for i in range(64)]
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5, limit_side_len=1280)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(64).region_batch_size(256).build()
    got = ocr.predict(imgs)
    assert len(got) == 64 and sum(len(g.text_regions) for g in got) > 1500
    n, ties = _check_pages_against_oracle(got, imgs, [0, 17, 30, 46, 63], det, rec, chars, dict(limit_side_len=1280), (0.3, 0.6, 1.5))
    assert n > 60 and ties <= 2
    ocr.close()


def test_c4_rank_0_of_8_shard_of_the_1024_page_list():
    """BASELINE C4 (1024 pages image-parallel over 8 GPUs) as ONE rank sees it: oar_shard_range(1024, 8, 0) = pages [0, 128), image_batch_size 32,
    a 2-thread geometry pool (the rank's share of the host), ONE predict over the whole shard, the result leaving as oar_ocr_pack's blob and meeting the
    other ranks' blobs in oar_packed_merge.  Six pages of the shard are checked against the oracle, every page against the object-returning entry."""
    a, b = api.shard_range(1024, 8, 0)
    assert (a, b) == (0, 128) and api.shard_range(1024, 8, 7) == (896, 1024)
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    imgs = [pages.make_page(a + i, (960, 960), 40) for i in range(b - a)]                 # bench.py --config 3 seeds its pages the same way
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).host_threads(2).build()
    _, ptrs, ws, hs = api._img_arrays(imgs)
    packed = ocr.predict_packed(ptrs, ws, hs, len(imgs), want_blob=True)
    assert len(packed.region_offsets) == 129 and packed.region_offsets[-1] > 3600 and packed.blob == packed.to_bytes()
    got = ocr.predict(imgs)
    assert packed.region_offsets.tolist() == np.concatenate([[0], np.cumsum([len(g.text_regions) for g in got])]).tolist()
    k = 0
    for g in got:
        for t in g.text_regions:
            assert np.array_equal(packed.points[k], t.bounding_box) and packed.text(k) == t.text and packed.scores[k] == np.float32(t.confidence)
            k += 1
    n, ties = _check_pages_against_oracle(got, imgs, [0, 23, 41, 64, 86, 127], det, rec, chars, {}, (0.3, 0.6, 1.5))
    assert n > 100 and ties <= 2
    # rank 0's blob + a second rank's (two pages of ITS shard) through the merge the host runs after its gather
    a1, _ = api.shard_range(1024, 8, 1)
    tail = [pages.make_page(a1 + i, (960, 960), 40) for i in range(2)]
    _, p1, w1, h1 = api._img_arrays(tail)
    other = ocr.predict_packed(p1, w1, h1, 2, want_blob=True)
    merged = api.PackedPages.merge([packed.blob, other.blob])
    nr = int(packed.region_offsets[-1])
    assert merged.region_offsets.tolist() == packed.region_offsets.tolist() + [nr + int(v) for v in other.region_offsets[1:]]
    assert np.array_equal(merged.points[:nr], packed.points) and np.array_equal(merged.points[nr:], other.points)
    assert np.array_equal(merged.scores, np.concatenate([packed.scores, other.scores])) and merged.utf8 == packed.utf8 + other.utf8
    ocr.close()


def test_c4_all_eight_shards_of_the_1024_page_list_on_one_gpu():
    """BASELINE C4 at FULL size, sequentially on one GPU (VERDICT r4 #7a): the eight oar_shard_range(1024, 8, r) shards each go through ONE
    predict_packed(want_blob=True) as rank r would run it, the eight blobs meet in oar_packed_merge in rank order -> 1024 pages in list order.
    Region counts, boxes, texts and scores of the merged result equal the per-shard results; sampled pages of shards 3 and 7 equal the
    object-returning entry point; shard 7's last page is checked against the oracle."""
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    chars = api.read_dict(models.synth_dict(6904))
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(32).region_batch_size(256).build()
    blobs, shards = [], []
    covered = 0
    for r in range(8):
        a, b = api.shard_range(1024, 8, r)
        assert a == covered and b - a == 128
        covered = b
        imgs = [pages.make_page(a + i, (960, 960), 40) for i in range(b - a)]
        _, ptrs, ws, hs = api._img_arrays(imgs)
        packed = ocr.predict_packed(ptrs, ws, hs, len(imgs), want_blob=True)
        assert len(packed.region_offsets) == 129 and packed.region_offsets[-1] > 3600
        blobs.append(packed.blob)
        shards.append(packed)
        if r in (3, 7):   # a few pages of the shard through OAROCR.predict (its own call: crops pooled over these pages only -> same boxes, texts may differ only by batch padding)
            sub = [0, 64, 127]
            got = ocr.predict([imgs[i] for i in sub])
            for j, i in enumerate(sub):
                k0, k1 = int(packed.region_offsets[i]), int(packed.region_offsets[i + 1])
                assert k1 - k0 == len(got[j].text_regions)
                for k, t in zip(range(k0, k1), got[j].text_regions):
                    assert np.array_equal(packed.points[k], t.bounding_box) and abs(float(packed.scores[k]) - t.confidence) <= 1e-3
        if r == 7:
            got_all = ocr.predict(imgs[96:])   # the last recognition batches of the list, object form, against the oracle
            n, ties = _check_pages_against_oracle(got_all, imgs[96:], [31], det, rec, chars, {}, (0.3, 0.6, 1.5))
            assert n > 20 and ties <= 2
        del imgs
    assert covered == 1024
    merged = api.PackedPages.merge(blobs)
    assert len(merged.region_offsets) == 1025
    base = 0
    for r, sh in enumerate(shards):
        nr = int(sh.region_offsets[-1])
        assert merged.region_offsets[r * 128:(r + 1) * 128 + 1].tolist() == [base + int(v) for v in sh.region_offsets]
        assert np.array_equal(merged.points[base:base + nr], sh.points) and np.array_equal(merged.scores[base:base + nr], sh.scores)
        assert merged.text(base) == sh.text(0) and merged.text(base + nr - 1) == sh.text(nr - 1)
        base += nr
    assert base == len(merged.scores) and merged.utf8 == b"".join(sh.utf8 for sh in shards)
    ocr.close()
