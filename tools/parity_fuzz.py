"""Randomised end-to-end parity sweep: OAROCR (HIP, through the C ABI) against the oracle pipeline on random page sizes,
line counts, thresholds and batch sizes.  usage: python tools/parity_fuzz.py [n_cases] [seed] [stages|server|seal|plain] [only_case]
(only_case: replay that one case of the seeded sequence -- the random draws of the earlier cases are made, nothing is computed for them --
and print the regions whose boxes differ; also dumps the pages to gpurun_out/fuzz_case_<n>.npz)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import pipeline_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
with_stages = len(sys.argv) > 3 and sys.argv[3] == "stages"   # randomly attach doc orientation / UVDoc / text-line orientation
cls4, cls2, uvdoc = models.build_cls(4, seed=5)[0], models.build_cls(2, seed=9)[0], models.build_uvdoc(seed=6)[0]
server = len(sys.argv) > 3 and sys.argv[3] == "server"       # BASELINE config 3 graphs (wide layers: large-K weight-stationary kernels);
#                                                               the torch-CPU oracle needs ~15 min per case on them: run a handful at most
seal = len(sys.argv) > 3 and sys.argv[3] == "seal"           # text_type "seal": stamp-like pages, polygon boxes, sort_poly_boxes, bounding-rectangle crops
import os
tiny_name = "tiny_full" if os.environ.get("FUZZ_GRAPHS", "") == "full" else "tiny"   # FUZZ_GRAPHS=full: the real-size graphs bench.py times by default since round 6
c3small = os.environ.get("FUZZ_GRAPHS", "") == "c3small"   # the narrow twins of the C3 graphs (PP-HGNetV2 / LK-PAN detector, SVTRv2 recognizer): 9x9 convolutions, grouped mixing, streaming attention in the whole pipeline
if c3small:
    det, _ = models.build_det("hgnet_small", seed=0)
    rec, _ = models.build_rec("svtrv2_small", vocab=6625, seed=1)
    chars = api.read_dict(models.synth_dict(6623))
else:
    det, _ = models.build_det("server" if server else tiny_name, seed=2 if server else 0)
    rec, _ = models.build_rec("server" if server else tiny_name, vocab=18710 if server else 6906, seed=3 if server else 1)
    chars = api.read_dict(models.synth_dict(18708 if server else 6904))
max_side = int(os.environ.get("FUZZ_MAX_SIDE", "0")) or (640 if server else 1100)   # FUZZ_MAX_SIDE=3000: pages far past limit_side_len (Triangle reduction by up to ~3x in front of the detector)
only_case = int(sys.argv[4]) if len(sys.argv) > 4 else None
bad = 0
t0 = time.time()
for case in range(n_cases):
    n_img = int(rng.integers(1, 5))
    same = rng.random() < 0.5
    h0, w0 = int(rng.integers(48, max_side)), int(rng.integers(48, max_side))
    imgs = []
    for i in range(n_img):
        h, w = (h0, w0) if same else (int(rng.integers(48, max_side)), int(rng.integers(48, max_side)))
        if seal:
            h, w = max(h, 260) // 2 + 130, max(w, 260) // 2 + 130      # 260 .. 680: room for at least one arc band, oracle polygons stay affordable
            imgs.append(pages.make_seal_page(int(rng.integers(0, 1 << 30)), (h, w), arcs=int(rng.integers(1, 4)), straight=int(rng.integers(0, 3))))
        else:
            imgs.append(pages.make_page(int(rng.integers(0, 1 << 30)), (h, w), int(rng.integers(0, 24))))
    thr, bthr, unclip = float(rng.choice([0.2, 0.3, 0.4])), float(rng.choice([0.5, 0.6, 0.7])), float(rng.choice([1.5, 1.8, 2.0]))
    if seal:
        unclip = float(rng.choice([0.5, 1.0, 1.5]))
    ibs, rbs = int(rng.choice([1, 2, 8])), int(rng.choice([3, 16, 64]))
    b = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(thr, bthr, unclip)).image_batch_size(ibs).region_batch_size(rbs)
    if seal:
        b = b.text_type("seal")
    stages = {}
    if with_stages:
        if rng.random() < 0.6:
            b = b.with_document_image_orientation_classification(cls4); stages["doc_orientation"] = cls4
            imgs = [np.ascontiguousarray(np.rot90(im, int(rng.integers(0, 4)))) for im in imgs]
        if rng.random() < 0.5:
            b = b.with_document_image_rectification(uvdoc); stages["rectifier"] = uvdoc
        if rng.random() < 0.6:
            b = b.with_text_line_orientation_classification(cls2); stages["line_orientation"] = cls2
    if only_case is not None and case != only_case:
        continue
    ocr = b.build()
    got = ocr.predict(imgs)
    ref = pipeline_ref.OracleOCR(det, rec, chars, thr, bthr, unclip, image_batch_size=ibs, region_batch_size=rbs, **stages, **({"text_type": "seal"} if seal else {})).predict(imgs)
    ok = True
    for g, r in zip(got, ref):
        rep = pipeline_ref.compare_results(g, r)
        if not rep["ok"] and "rectifier" in stages and abs(len(g.text_regions) - len(r)) <= 1:
            # a rectified page may differ from the oracle's by one grey level in a few pixels (the (v * 255) truncation after UVDoc,
            # tests/test_gpu_config5.py): the detector then sees a marginally different image, so a threshold-marginal region may
            # appear / vanish and boxes may move by a pixel or two -- inside the contract, reported but not counted as a failure
            print("  rectified page: marginal difference tolerated", {k: rep[k] for k in ("n_regions", "box_mismatch")})
            continue
        ok = ok and rep["ok"]
        if not rep["ok"]:
            print("  MISMATCH", rep)
            if only_case is not None:
                import os
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez_compressed(f"gpurun_out/fuzz_case_{case}.npz", **{f"page{i}": im for i, im in enumerate(imgs)})
                for q, (gr, rr) in enumerate(zip(g.text_regions, r)):
                    gb, rb = np.asarray(gr.bounding_box, np.float32), np.asarray(rr["box"], np.float32)
                    if gb.shape != rb.shape or not np.array_equal(gb, rb):
                        print(f"  region {q}: product {gb.shape} oracle {rb.shape}")
                        print("   product", gb.reshape(-1).tolist())
                        print("   oracle ", rb.reshape(-1).tolist())
    nreg = sum(len(r) for r in ref)
    print(f"case {case}: stages={sorted(stages)} {n_img} pages {[im.shape[:2] for im in imgs]} thr={thr} box={bthr} unclip={unclip} ibs={ibs} rbs={rbs} regions={nreg} {'ok' if ok else 'FAIL'}", flush=True)
    bad += 0 if ok else 1
    ocr.close()
print(f"{n_cases - bad}/{n_cases} cases identical (boxes bit-exact, scores <= 1e-3) in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
