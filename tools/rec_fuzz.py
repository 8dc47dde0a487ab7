"""The recognition adapter on random crop batches against the oracle (round 6): crop widths 4 .. 6000 px and heights 6 .. 160 px (tensor widths up to max_img_w = 3200: up to 400
tokens, where the sample-local chain no longer fits LDS), batches of 1 .. 40 crops, three recognizer graphs (the real-size one, the round-1 one, the SVTRv2 twin).
Indices must agree unless the oracle's own top-2 probabilities tie within 1e-5; probabilities within 1e-3; texts equal when the indices are.   usage: python tools/rec_fuzz.py [n] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import pipeline_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
graphs = []
for name, vocab in (("tiny_full", 6906), ("tiny", 6906), ("svtrv2_small", 6625)):
    rec, _ = models.build_rec(name, vocab=vocab, seed=1)
    chars = api.read_dict(models.synth_dict(vocab - 2))
    graphs.append((name, rec, chars, api.TextRecognitionPredictor(rec, chars), pipeline_ref.OracleRecognizer(rec, chars)))
bad = 0
t0 = time.time()
for case in range(n_cases):
    name, rec, chars, pred, orc = graphs[case % len(graphs)]
    nb = int(rng.choice([1, 2, 7, 16, 40]))
    crops = []
    for i in range(nb):
        h = int(rng.choice([int(rng.integers(6, 160)), 48, 32]))
        w = int(rng.choice([int(rng.integers(4, 700)), int(rng.integers(700, 6000)), 320]))
        if rng.random() < 0.7 and h >= 20 and w >= 40:
            crops.append(pages.make_crop(int(rng.integers(0, 1 << 30)), w, h))
        else:
            crops.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    got = pred.predict(crops)
    ref = orc.recognize(crops)
    ok = got.tensor_width == ref["Wt"] and got.indices.shape == ref["idx"].shape
    why = ""
    if ok:
        mism = got.indices != ref["idx"]
        for b, t in zip(*np.nonzero(mism)):
            pf = ref["probs_full"][b, t]
            if abs(pf[got.indices[b, t]] - pf[ref["idx"][b, t]]) >= 1e-5:
                ok, why = False, f"index ({b},{t})"
                break
        if ok and np.abs(got.probs - ref["prob"]).max() >= 1e-3:
            ok, why = False, f"probabilities {np.abs(got.probs - ref['prob']).max():.2e}"
        if ok and not mism.any() and (got.texts != ref["texts"] or not np.allclose(got.scores, ref["scores"], atol=1e-3)):
            ok, why = False, "texts / scores"
    else:
        why = f"tensor width {got.tensor_width} vs {ref['Wt']}"
    print(f"{'ok  ' if ok else 'FAIL'} case {case} [{name}] {nb} crops, widths {min(c.shape[1] for c in crops)}..{max(c.shape[1] for c in crops)}, Wt {ref['Wt']} {why}", flush=True)
    bad += 0 if ok else 1
print(f"{n_cases - bad}/{n_cases} recognition batches agree with the oracle in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
