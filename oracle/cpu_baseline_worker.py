"""One CPU-baseline worker of bench.py's `cpu_baseline` leg (TEST / MEASUREMENT INFRASTRUCTURE ONLY -- nothing under oar_ocr_amd/ imports this).

A worker owns one OracleOCR (C restatement of the pre / post stages + the torch-CPU network interpreter), warms it on one page, prints READY,
waits for GO on stdin, runs its pages one predict per page (the reference's CPU policy: image batch 1, region batch 16) and prints one JSON line:
seconds, regions, and the milliseconds per page spent in each stage.  bench.py starts W workers with T torch threads each (W x T = the host's
cores): W = 1 is the reference's default deployment (one pipeline, all intra-op threads), W > 1 is W pipelines side by side -- what an operator
with 64 idle cores and a page queue would run, and the fairer comparison for a throughput metric.

`--fast` runs the interpreter's convolutions in channels_last: same graph, oneDNN's preferred layout (the parity oracle never runs in this mode)."""
import argparse
import collections
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--pages", type=int, default=2)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--size", type=int, default=960)
    ap.add_argument("--lines", type=int, default=40)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--fast", type=int, default=1)
    ap.add_argument("--c3-graphs", default="named", help="config 2: 'named' = PP-HGNetV2 / LK-PAN detector + SVTRv2 recognizer (V = 6625), 'standin' = the rounds 1-5 graphs")
    a = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = str(a.threads)
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(a.threads)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oar_ocr_amd import api
    from oar_ocr_amd.synth import models, pages
    from oracle import cpu_ref as R, onnx_ref, pipeline_ref

    T = collections.defaultdict(float)

    def wrap(mod, name, key=None):
        f = getattr(mod, name)

        def g(*args, **kw):
            s = time.perf_counter()
            r = f(*args, **kw)
            T[key or name] += time.perf_counter() - s
            return r
        setattr(mod, name, g)

    if a.fast:
        conv = F.conv2d

        def conv_cl(inp, w, b=None, **kw):
            return conv(inp.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last), b, **kw)
        F.conv2d = conv_cl
    for n in ("det_preprocess", "db_postprocess", "rotate_crop", "rec_preprocess", "argmax_rows", "ctc_decode"):
        wrap(R, n)
    size_name, vocab = ("server", 18710) if a.config == 2 else ("tiny", 6906)
    det_name = rec_name = size_name
    if size_name == "tiny":   # graphs of the size of the files they stand for (bench.py --det-params real-size, the default since round 6)
        det_name = rec_name = "tiny_full"
    if a.config == 2 and a.c3_graphs == "named":
        det_name, rec_name, vocab = "server_hgnet", "svtrv2", 6625
    det, _ = models.build_det(det_name, seed=0)
    rec, _ = models.build_rec(rec_name, vocab=vocab, seed=1)
    chars = api.read_dict(models.synth_dict(vocab - 2))
    stages = dict(doc_orientation=models.build_cls(4, seed=5)[0], rectifier=models.build_uvdoc(seed=6)[0],
                  line_orientation=models.build_cls(2, seed=9)[0]) if a.config == 4 else {}
    kw = dict(limit_side_len=a.size) if (a.config == 2 and a.c3_graphs == "named") else {}
    oc = pipeline_ref.OracleOCR(det, rec, chars, 0.3, 0.6, 1.5, image_batch_size=1, region_batch_size=16, threads=a.threads, **stages, **kw)
    # network time by graph: the detector's and the recognizer's interpreter runs are told apart by their model dict
    run = onnx_ref.run

    def run_timed(model, feeds, want=None):
        s = time.perf_counter()
        r = run(model, feeds, want)
        T["net_det" if model is oc.det.model else "net_rec" if model is oc.rec.model else "net_other"] += time.perf_counter() - s
        return r
    onnx_ref.run = run_timed
    P = [pages.make_page(a.seed0 + i, (a.size, a.size), a.lines) for i in range(a.pages + 1)]
    oc.predict(P[:1])
    T.clear()
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    regions = 0
    for pg in P[1:]:
        regions += sum(len(r) for r in oc.predict([pg]))
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "pages": a.pages, "regions": regions, "stage_ms_per_page": {k: round(v / a.pages * 1e3, 1) for k, v in sorted(T.items(), key=lambda kv: -kv[1])}}), flush=True)


if __name__ == "__main__":
    main()
