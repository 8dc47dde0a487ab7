#!/usr/bin/env python
"""bench.py -- images/sec of the det+rec hot path (OAROCR::predict behind the C ABI) on N MI355X of one node.

The metric is SURVEY.md section 8d's: u8 pages in HOST memory -> final sorted boxes + texts + scores on the host.
A "step" = one pass of the hot path over one batch of synthetic pages per GPU: ONE `oar_ocr_predict` (H2D upload of the
pages overlapped sub-batch by sub-batch with the detector network, detection, crops, recognition, CTC argmax) + ONE
`oar_ocr_decode` (CTC collapse, strings, score filter inside the library); with N > 1 the per-rank results are gathered on
rank 0 inside the timed region.  Default workload = BASELINE.json configs[1]: PP-OCRv6-tiny-class det+rec, 32 synthetic
960x960 pages per GPU per step ("weak" scaling: every rank processes its own 32 pages; value = pages of ALL ranks /
max-over-ranks time).  `--config 3` is BASELINE configs[3]: 1024 pages block-partitioned over the ranks (128 per GPU at
N = 8, "strong" scaling).  The device-resident figure (pages already in HBM, no decode -- round 1's headline) is reported
as the extra field `device_resident`.

`--gpus N` without a torchrun environment starts its own N ranks (one process per GPU, RCCL over xGMI).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel class, hipEvent-timed on the engine's own stream inside
the timed region) and `cpu_baseline` (the oracle pipeline -- C restatement + torch-CPU network -- on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: f32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak; the bf16x6 kernels spend 6 bf16 MFMAs per f32-equivalent product


def mfma_peak_for(kernel):
    """Peak in f32-equivalent TFLOP/s (2*M*K*N per product) of the matrix pipe the kernel class runs on."""
    return (MFMA_BF16_PEAK_TF / 6.0, "bf16 dense peak / 6 (three-way split, six MFMAs per product)") if ("_x6" in kernel or kernel.startswith("dsblock")) else (MFMA_F32_PEAK_TF, "f32-input MFMA dense peak")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 100 for config 1 -- ~1.4 s of GPU time, long enough for an external sampler to see; 10 for the others)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=0, help="pages per GPU per step (0 = the config's: 32 / 64 / 1024 total / 16)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--lines", type=int, default=40)
    ap.add_argument("--region-batch", type=int, default=256, help="recognition batch (this backend's recommended_batch_size; reference adapter: 64)")
    ap.add_argument("--cpu-pages", type=int, default=3, help="pages in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-device-resident", action="store_true", help="skip the second, device-resident timing")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the third timing (two calls in flight through oar_ocr_predict_async)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs index: 1 = v6-tiny det+rec on 32 x 960^2 pages per GPU (the metric's configuration, default); "
                         "2 = PP-OCRv5-server-class detector (PP-HGNetV2 / LK-PAN, 21.7 M parameters) + SVTRv2-class recognizer (20.5 M, V = 6625) on 64 x 1280^2 pages, "
                         "detector at 1280 (limit_side_len = 1280); 3 = v6-tiny, 1024 pages block-partitioned over the ranks; "
                         "4 = full pipeline (doc orientation + UVDoc + det + rec + text-line orientation) on 16 x 960^2 pages")
    ap.add_argument("--det-params", choices=("class", "real-size"), default="real-size",
                    help="config 1 / 3 / 4 graphs: 'real-size' (default since round 6) = detector and recognizer at the parameter counts of the files they stand for "
                         "(pp-ocrv6_tiny_det.onnx 0.445 M, pp-ocrv6_tiny_rec.onnx 1.116 M: registry.rs:83-84 -> synth 'tiny_full': 0.447 M / 1.136 M); 'class' = the lighter "
                         "graphs of rounds 1-5 (0.288 M / 0.914 M).  The default line times 'real-size' and reports 'class' beside it")
    ap.add_argument("--rec-variant", choices=("head", "deep"), default="head",
                    help="real-size recognizer of config 1 / 3 / 4: 'head' = synth 'tiny_full' (the extra 0.2 M parameters in the CTC projection, 96 x 6906: 1.136 M); "
                         "'deep' = synth 'tiny_deep' (two more 5x5 blocks + one SE block in the backbone: 1.103 M) -- the first form this round measured, kept for comparison")
    ap.add_argument("--c3-graphs", choices=("named", "standin"), default="named",
                    help="config 2: 'named' = graphs of the size and kind BASELINE C3 names (synth/models.py build_det_hgnet / build_rec_svtrv2); 'standin' = the "
                         "widened LCNet detector (4.3 M) + SVTR-neck recognizer (7.3 M, V = 18710) rounds 1-5 ran config 2 on, detector input at the default 960")
    ap.add_argument("--models-dir", default="", help="real-weights mode (SURVEY 8d mode i): a directory holding the config's three files under their registry names "
                    "(config 1: pp-ocrv6_tiny_det.onnx, pp-ocrv6_tiny_rec.onnx, ppocrv6_tiny_dict.txt; config 2: pp-ocrv5_server_det.onnx, ch_svtrv2_rec.onnx, ppocr_keys_v1.txt). "
                    "Every file's size and SHA-256 is checked against the reference's registry (oar_ocr_amd/weights.py); a mismatch refuses the run.  With onnxruntime importable, "
                    "the CPU leg becomes the ORT-CPU stand-in and a network-level parity report (max |dprob|, threshold-marginal pixels) is added")
    ap.add_argument("--no-real-size", action="store_true", help="skip the extra timing of the real-size detector (config 1, one GPU)")
    ap.add_argument("--stub-engine", action="store_true", help=argparse.SUPPRESS)   # tests/test_bench_cpu.py: control flow without a GPU
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts its own ranks (0 = pick a free one)")
    args = ap.parse_args(argv)
    if args.steps <= 0:
        args.steps = 100 if args.config == 1 and not args.stub_engine else 10
    return args


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) of this same script, rendezvous on 127.0.0.1."""
    port = args.master_port
    if not port:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (the host driver's only mode): RCCL needs it
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def cgroup_cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  Polling
    worker threads beyond this budget get the whole process throttled for tens of milliseconds (seen on the 1-GPU bench box:
    cpu.max = 16 CPUs on a 256-thread host)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def pin_rank_to_its_cores(local_rank: int, local_world: int) -> int:
    """One process per GPU shares the host: give each rank a contiguous slice of the cores BEFORE its geometry pool is
    created (threads inherit the affinity), so that N spinning pools never compete for a core.  Returns the rank's CPU
    budget = min(slice size, its share of the container's CPU quota).  (Tried at the end of round 4 and dropped: pinning a rank
    whose quota share is smaller than its slice to `budget` cores from inside the process -- 1855-2037 images/s against 2108-2190
    floating on the 1-GPU box, although `taskset -c 0-15 python bench.py` from outside measures 2246-2330:
    profiles/r4/host_cores_sweep.txt.)"""
    quota = cgroup_cpu_quota()
    if not hasattr(os, "sched_getaffinity"):
        per = max(1, (os.cpu_count() or 1) // max(local_world, 1))
        return per if quota is None else max(1, min(per, int(quota // max(local_world, 1))))
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // max(local_world, 1))
    if local_world > 1:
        mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            pass
    return per if quota is None else max(1, min(per, int(quota // max(local_world, 1))))


class StubEngine:
    """Stands in for the HIP pipeline in the CPU-only control-flow test: same interface, fabricated results."""

    def __init__(self, rank):
        self.rank = rank

    def predict_packed(self, pages):
        import numpy as np
        from oar_ocr_amd import api
        n = len(pages)
        per = 3
        ro = np.arange(0, (n + 1) * per, per, dtype=np.uint32)
        pts = np.tile(np.array([[0, 0], [10, 0], [10, 5], [0, 5]], np.float32), (n * per, 1, 1)) + self.rank
        texts = [f"r{self.rank}p{i // per}k{i % per}".encode() for i in range(n * per)]
        to = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.uint64)
        time.sleep(0.002)
        return api.PackedPages(ro, pts, np.full(n * per, 0.5, np.float32), b"".join(texts), to)

    def predict_device(self):
        return (0, 0)

    def close(self):
        pass


def main():
    args = parse_args()
    real_weights = None
    if args.models_dir:   # verified before anything runs (or any rank starts): a file that is not the registry's is refused, never silently replaced by a synthetic graph
        from oar_ocr_amd import weights
        try:
            rw_det, rw_rec, rw_chars, rw_report = weights.load_config(args.models_dir, args.config)
        except weights.WeightsError as e:
            print(f"bench.py --models-dir {args.models_dir}: refused -- {e}", file=sys.stderr, flush=True)
            sys.exit(3)
        real_weights = {"dir": args.models_dir, "files": rw_report}
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    from oar_ocr_amd import api, dist as oard
    from oar_ocr_amd.synth import models, pages as synth_pages

    stub = args.stub_engine
    # one rank per GPU over RCCL; OAR_DIST_BACKEND=gloo exists so that the N > 1 path can be exercised on a box with fewer
    # GPUs than ranks (RCCL refuses two ranks on one device) and by the CPU control-flow test
    backend = os.environ.get("OAR_DIST_BACKEND", "gloo" if stub else "nccl")
    rank, local, world = oard.init_from_env(backend if args.gpus > 1 else None)
    if world > 1:
        import torch.distributed as dist
    if not stub:
        assert api.device_count() > 0, "bench.py needs a GPU: libOarMi355x has no CPU fallback"
    dev = local % max(api.device_count(), 1) if not stub else 0
    comm_dev = torch.device("cuda", dev) if (backend == "nccl" and not stub) else torch.device("cpu")
    if not stub:
        torch.cuda.set_device(dev)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cores = pin_rank_to_its_cores(local, local_world)
    pool_threads = None   # N > 1: the geometry pool size this rank asks for (below); None = the library's default rule

    # ---- workload
    size_name, vocab = ("server", 18710) if args.config == 2 else ("tiny", 6906)
    det_name = "tiny_full" if (size_name == "tiny" and args.det_params == "real-size") else size_name
    rec_name = det_name if size_name == "tiny" else size_name
    if rec_name == "tiny_full" and args.rec_variant == "deep":
        rec_name = "tiny_deep"
    c3_named = args.config == 2 and args.c3_graphs == "named"
    if c3_named:   # ch_svtrv2_rec.onnx is used with ppocr_keys_v1.txt: 6623 lines -> V = 6625
        det_name, rec_name, vocab = "server_hgnet", "svtrv2", 6625
    size = args.size or (1280 if args.config == 2 else 960)
    if args.config == 3:      # BASELINE configs[3]: 1024 pages over the ranks (block partition, 128 per GPU at N = 8)
        total_pages = args.pages or 1024
        a, b = oard.shard_range(total_pages, world, rank)
        n_pages, seed0, scaling = b - a, a, "strong"
    else:
        n_pages = args.pages or {1: 32, 2: 64, 4: 16}[args.config]
        seed0, scaling, total_pages = rank * n_pages, "weak", n_pages * world
    host_pages = [synth_pages.make_page(seed0 + i, (size, size), args.lines) for i in range(n_pages)]
    image_batch = min(n_pages, 64 if args.config == 2 else 32)

    if stub:
        eng, det_info, rec_info = StubEngine(rank), {"params": 0}, {"params": 0}
    else:
        if real_weights:
            det, rec, chars = rw_det, rw_rec, rw_chars
            det_info, rec_info, vocab = {"params": len(det) // 4}, {"params": len(rec) // 4}, len(chars) + 2     # (file bytes / 4: the files are almost all f32 initializers)
        else:
            det, det_info = models.build_det(det_name, seed=0)
            rec, rec_info = models.build_rec(rec_name, vocab=vocab, seed=1)
            chars = api.read_dict(models.synth_dict(vocab - 2))
        cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)   # examples/ocr.rs:119-133 set
        if c3_named:   # "batch=64 1280x1280": the default limit (960 / Max, src/oarocr/ocr.rs:351-363) would Triangle-downscale the pages to 960^2 first
            cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5, limit_side_len=size)
        builder = (api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(image_batch)
                   .region_batch_size(args.region_batch).device(dev))
        if world > 1:
            # this rank's geometry pool stays inside its CPU budget, and on a slice of six or more pinned cores it leaves two of them to the call's
            # uploader / enqueuer threads (profiles/r4/host_cores_sweep.txt: 8 pinned cores 1600-1750 images/s with a pool of 8, 2165-2177 with 6)
            ht = max(2, min(16, cores))
            aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0
            if aff >= 6 and ht > aff - 2:
                ht = aff - 2
            builder = builder.host_threads(ht)
            pool_threads = ht
            # (round 2 switched to the GPU border follower below 6 cores per rank; measured in round 3, profiles/r3/gpu_contours_breakeven.txt:
            # the host tracer wins at EVERY pool size -- 1 thread 1002 vs 903 images/s, 2 threads 1301 vs 958, 4 threads 1635 vs 967 -- so
            # the switch is gone; oar_det_cfg.gpu_contours / OAR_GPU_CONTOURS remain as a knob)
        if args.config == 4:
            builder = (builder.with_document_image_orientation_classification(models.build_cls(4, seed=5)[0])
                       .with_document_image_rectification(models.build_uvdoc(seed=6)[0])
                       .with_text_line_orientation_classification(models.build_cls(2, seed=9)[0]))
        ocr = builder.build()
        _, h_ptrs, h_ws, h_hs = api._img_arrays(host_pages)   # pageable host buffers, exactly what a caller's Vec<RgbImage> is

        class HipEngine:
            def predict_packed(self, pages):
                return ocr.predict_packed(h_ptrs, h_ws, h_hs, n_pages, want_blob=world > 1)   # N > 1: the rank's blob comes from oar_ocr_pack

            def close(self):
                ocr.close()
        eng = HipEngine()

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    gathered = {"pages": 0, "regions": 0, "bytes": 0}

    passes = {"n": 0}   # every oar_ocr_predict pass of this process (discovery + warm-up + timed): the divisor of a rocprofv3 --stats summary

    def step():
        """host pages -> boxes + texts + scores on the host of rank 0"""
        passes["n"] += 1
        packed = eng.predict_packed(host_pages)
        passes["regions"] = len(packed.scores)   # this rank's own regions of the step
        if world == 1:
            gathered.update(pages=len(packed.region_offsets) - 1, regions=len(packed.scores), bytes=len(packed.utf8))
            return packed
        blobs = oard.gather_bytes(getattr(packed, "blob", None) or packed.to_bytes(), 0, comm_dev)   # oar_ocr_pack's wire format
        if rank == 0:
            merged = api.PackedPages.merge(blobs)   # oar_packed_merge: block partition => concatenation in rank order restores page order
            gathered.update(pages=len(merged.region_offsets) - 1, regions=len(merged.scores), bytes=len(merged.utf8))
        return packed

    # -- untimed: discover the kernel classes with every class instrumented.  `dominant` = the class with the largest share (it is `roofline`);
    # the four largest are all sampled inside the timed region (`roofline.by_family`): one profiler class per kernel FAMILY,
    # e.g. dsblock_rs (row-streaming separable blocks, HBM-bound) and dsblock_cs (chunk-streamed ones, bound by their own instruction streams).
    dominant, dominant_launches, breakdown, families = None, 0, {}, []
    if not args.no_prof and not stub:
        step()
        api.prof_enable(True)
        api.prof_filter("")
        api.prof_reset()
        step()
        snap = api.prof_snapshot()
        api.prof_enable(False)
        if snap and snap[0]["total_ms"] > 0:
            dominant, dominant_launches = snap[0]["name"], snap[0]["launches"]
            families = [(e["name"], e["launches"]) for e in snap[:4] if e["total_ms"] > 0]
        breakdown = {e["name"]: round(e["total_ms"], 3) for e in snap[:12]}

    if dominant:
        api.prof_filter(",".join(n for n, _ in families))
        api.prof_enable(True)
    for _ in range(args.warmup):
        step()
    if dominant:
        api.prof_reset()
    # An event-bracketed kernel costs ~11 us of idle queue around it (rocprofv3 kernel trace).  So each
    # step times 1 launch in S, with the phase rotating over the steps: every launch POSITION of the class is timed in exactly
    # `balanced / S` of the timed steps, which gives the same average as timing all of them at 1/S of the overhead.
    S = min(20, max(args.steps, 1))   # (round 5: 1 in min(20, steps), was 1 in 5: ~5 instead of ~21 bracketed launches per step; every position still timed at least once)
    balanced = S * (args.steps // S)
    barrier()
    cpu0 = os.times()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if dominant:
            api.prof_sampling(S, i % S if i < balanced else -1)
        step()
    barrier()
    dt = time.perf_counter() - t0
    cpu1 = os.times()
    # CPU time of this rank's process (every thread: caller, geometry pool, uploader, enqueuer, HIP runtime) per timed step -- what a rank costs its host
    host_cpu_ms = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(args.steps, 1) * 1e3
    if not stub:
        api.prof_sampling(1, 0)
    roof = None
    cfg_sfx = "" if args.config in (1, 3) else f"_c{args.config}"   # committed counter summaries are per workload: profiles/r*/pmc_traffic[_c2].json, mfma_util[_c2].json
    from oar_ocr_amd.build import csrc_fingerprint
    csrc_now = csrc_fingerprint()

    def family_roof(e, name, launches_per_step):
        """One class against BOTH of its roofs.  `bound` is the roof that costs the launch more time at peak (the matrix roof also when it is within 20 % of the
        HBM one: such a kernel moves its bytes while it computes and is limited by its instruction streams -- dsblock_cs, profiles/r5/dsblock_pc_ablations.txt)."""
        sec = e["total_ms"] / 1e3
        mfma_peak, peak_note = mfma_peak_for(name)
        t_hbm, t_mfma = e["alg_bytes"] / (HBM_PEAK_GBS * 1e9), e["alg_flops"] / (mfma_peak * 1e12)
        frac_hbm, frac_mfma = t_hbm / sec, t_mfma / sec
        if t_mfma >= 0.8 * t_hbm:
            ach, peak, unit, bound = e["alg_flops"] / sec / 1e12, round(mfma_peak, 1), "TFLOP/s", "mfma"
        else:
            ach, peak, unit, bound = e["alg_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        # HBM bytes per launch of this kernel class from the committed PMC passes (rocprofv3 cannot run inside the timed
        # region; tools/make_profile_summaries.py writes the file from separate --pmc FETCH_SIZE / WRITE_SIZE passes)
        # The file carries the fingerprint of the csrc/ it was measured on: when the sources have changed since, the counters
        # describe other kernels -- `traffic` is then null and traffic_source says why.
        traffic, traffic_src = None, None
        for tf in sorted(ROOT.glob(f"profiles/r*/pmc_traffic{cfg_sfx}.json"), reverse=True):
            try:
                doc = json.loads(tf.read_text())
                t = doc.get(name)
                if t:
                    fresh = doc.get("_csrc_fingerprint") == csrc_now
                    traffic = t["hbm_bytes_per_launch"] if fresh else None
                    traffic_src = (f"{tf.relative_to(ROOT)} (csrc fingerprint {csrc_now}: matches the timed build)" if fresh else
                                   f"none: {tf.relative_to(ROOT)} was measured on csrc {doc.get('_csrc_fingerprint', 'unrecorded')}, this build is {csrc_now} -- re-run tools/profile_round.sh")
                    break
            except Exception:
                pass
        return {"bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": name, "peak_note": peak_note if bound == "mfma" else "HBM3E spec", "frac_hbm": round(frac_hbm, 4), "frac_mfma": round(frac_mfma, 4),
                "launches_per_step": launches_per_step, "avg_launch_us": round(e["total_ms"] * 1e3 / e["launches"], 2), "timed_launches": e["launches"],
                "alg_bytes_per_launch": e["alg_bytes"] / e["launches"], "alg_flops_per_launch": e["alg_flops"] / e["launches"],
                "share_of_step": round(e["total_ms"] / e["launches"] * launches_per_step * args.steps / (dt * 1e3), 4)}

    if dominant:
        snap = {e["name"]: e for e in api.prof_snapshot()}
        api.prof_enable(False)
        api.prof_filter("")
        fam = {}
        for name, lps in families:
            e = snap.get(name)
            if e and e["launches"] > 0 and e["total_ms"] > 0:
                fam[name] = family_roof(e, name, lps)
        if dominant in fam:
            n_sampled = sum(l for _, l in families)
            roof = dict(fam[dominant])
            roof["sampling"] = (f"the {n_sampled} launches per step of the sampled classes ({', '.join(n for n, _ in families)}) are timed 1 in {S} per step with a rotating phase: every launch "
                                f"position is timed in {balanced // S} of the {args.steps} timed steps")
            roof["by_family"] = fam

    # The second half of BASELINE.json's metric ("conv MFMA util %"): counter-measured matrix-pipe use per conv-GEMM class, from the
    # committed rocprofv3 --pmc pass over this same command (tools/profile_round.sh pass 3 -> profiles/r*/mfma_util.json; the
    # counters cannot be collected inside the timed region).  Only reported for the workload the pass was made on (config 1).
    mfma_util = None
    if rank == 0 and args.config in (1, 2):
        for mf in sorted(ROOT.glob(f"profiles/r*/mfma_util{cfg_sfx}.json"), reverse=True):
            try:
                m = json.loads(mf.read_text())
                if m.get("_csrc_fingerprint") != csrc_now:
                    mfma_util = {"classes": None, "source": f"none: {mf.relative_to(ROOT)} was measured on csrc {m.get('_csrc_fingerprint', 'unrecorded')}, this build is {csrc_now}"}
                    break
                mfma_util = {"unit": "% of the dense MFMA peak of the class's dtype (bf16 2500 / f32 157.3 TFLOP/s), time-weighted per class",
                             "source": str(mf.relative_to(ROOT)),
                             "classes": {k: v["pct_of_dense_peak"] for k, v in m.items() if isinstance(v, dict) and "pct_of_dense_peak" in v}}
                # the other bound of the same classes: counter-measured HBM bytes per launch / launch duration / 8 TB/s (same session's
                # FETCH_SIZE / WRITE_SIZE passes), so that each conv class shows how far it is from BOTH of its roofs
                tf = mf.with_name(f"pmc_traffic{cfg_sfx}.json")
                if tf.exists():
                    t = json.loads(tf.read_text())
                    if t.get("_csrc_fingerprint") == csrc_now:
                        mfma_util["hbm_frac"] = {k: t[k]["hbm_frac"] for k in mfma_util["classes"] if isinstance(t.get(k), dict) and t[k].get("hbm_frac") is not None}
                        mfma_util["hbm_frac_source"] = str(tf.relative_to(ROOT))
                break
            except Exception:
                pass

    tmax = dt
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        # what each rank saw (outside the timed region): its own wall time and host CPU per step, the cores it was pinned to, the geometry pool it
        # chose, its device and the world size the process group reports -- so that a SCALE record can be read rank by rank
        mine = {"rank": rank, "local_rank": local, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "host_cpu_ms_per_step": round(host_cpu_ms, 2),
                "host_cores": cores, "cpu_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                "geometry_pool_threads": pool_threads, "world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                "device": None if stub else torch.cuda.get_device_name(dev), "regions": int(passes.get("regions", 0))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # -- second figure: pages already resident in HBM, no decode (round 1's headline), same step count
    dev_res = None
    if not stub and not args.no_device_resident and world == 1:
        dev_pages = [api.DeviceBuffer(p, dev) for p in host_pages]
        ptrs = [int(b.ptr.value) for b in dev_pages]
        ws = hs = [size] * n_pages
        for _ in range(max(1, args.warmup)):
            ocr.predict_device(ptrs, ws, hs, raw=True)
        torch.cuda.synchronize()
        d0 = time.perf_counter()
        for _ in range(args.steps):
            ocr.predict_device(ptrs, ws, hs, raw=True)
        torch.cuda.synchronize()
        ddt = time.perf_counter() - d0
        dev_res = {"value": round(n_pages * args.steps / ddt, 2), "unit": "images/sec", "ms_per_step": round(ddt / args.steps * 1e3, 3),
                   "what": "oar_ocr_predict_device only: pages resident in HBM, CTC indices returned, no string decode (round-1 definition)"}
        for b in dev_pages:
            b.free()

    # -- third figure: two calls in flight on one handle (oar_ocr_predict_async, oar_ocr_cfg.lanes = 2): call k + 1 uploads and detects
    # while call k recognises.  Same pages, same per-call results (tests/test_gpu_async.py); reported next to the synchronous headline.
    pipelined = None
    if not stub and not args.no_pipelined and world == 1:
        eng.close()
        ocr2 = builder.lanes(2).build()
        for _ in range(max(2, args.warmup)):
            ocr2.wait_packed(ocr2.submit_packed(h_ptrs, h_ws, h_hs, n_pages), n_pages)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        tickets = [ocr2.submit_packed(h_ptrs, h_ws, h_hs, n_pages)]
        for i in range(args.steps):
            if i + 1 < args.steps:
                tickets.append(ocr2.submit_packed(h_ptrs, h_ws, h_hs, n_pages))
            ocr2.wait_packed(tickets[i], n_pages)
        torch.cuda.synchronize()
        pdt = time.perf_counter() - p0
        pipelined = {"value": round(n_pages * args.steps / pdt, 2), "unit": "images/sec", "ms_per_step": round(pdt / args.steps * 1e3, 3), "calls_in_flight": 2,
                     "what": "the same host-entry step through oar_ocr_predict_async / oar_ocr_wait with oar_ocr_cfg.lanes = 2: every call is still one "
                             "OAROCR::predict (crops pooled over its own pages), two are in flight"}
        ocr2.close()

    # -- fourth figure (config 1, one GPU): the same step on the lighter graphs rounds 1-5 quoted their headline on (the file sizes
    # pin 0.445 M / 1.116 M parameters; since round 6 the headline runs graphs of that size and the lighter ones are the side figure)
    real_size = None
    if not stub and world == 1 and args.config == 1 and det_name == "tiny_full" and not args.no_real_size:
        det_cls, det_cls_info = models.build_det("tiny", seed=0)
        rec_cls, rec_cls_info = models.build_rec("tiny", vocab=vocab, seed=1)
        ocr3 = (api.OAROCRBuilder(det_cls, rec_cls, chars).text_detection_config(cfg).image_batch_size(image_batch).region_batch_size(args.region_batch).device(dev)).build()
        for _ in range(max(2, args.warmup)):
            r3 = ocr3.predict_packed(h_ptrs, h_ws, h_hs, n_pages)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for _ in range(args.steps):
            r3 = ocr3.predict_packed(h_ptrs, h_ws, h_hs, n_pages)
        torch.cuda.synchronize()
        qdt = time.perf_counter() - q0
        real_size = {"value": round(n_pages * args.steps / qdt, 2), "unit": "images/sec", "ms_per_step": round(qdt / args.steps * 1e3, 3),
                     "det_params": det_cls_info["params"], "rec_params": rec_cls_info["params"], "regions_per_step": len(r3.scores),
                     "what": "the same step (same pages, same entry point) on the LIGHTER graphs rounds 1-5 quoted as their headline: synth models.build_det('tiny') 0.288 M + "
                             "build_rec('tiny') 0.914 M parameters.  `value` above is on graphs of the size of the files it names (0.447 M / 1.136 M)"}
        ocr3.close()

    # -- fifth figure: the recognizer's batches alternating over two streams (OAR_REC_LANES=2: a second engine instance, same weights).  The
    # headline keeps one stream: with two, kernels of different batches share the GPU and a launch's duration stops being the kernel's own
    # (the roofline accounting above would be measuring the overlap), so this stays an opt-in with its own line.
    rec_two = None
    if not stub and world == 1 and args.config == 1 and not args.no_real_size:
        os.environ["OAR_REC_LANES"] = "2"
        try:
            ocr4 = builder.lanes(1).build()
        finally:
            os.environ.pop("OAR_REC_LANES", None)
        for _ in range(max(2, args.warmup)):
            ocr4.predict_packed(h_ptrs, h_ws, h_hs, n_pages)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for _ in range(args.steps):
            ocr4.predict_packed(h_ptrs, h_ws, h_hs, n_pages)
        torch.cuda.synchronize()
        qdt = time.perf_counter() - q0
        rec_two = {"value": round(n_pages * args.steps / qdt, 2), "unit": "images/sec", "ms_per_step": round(qdt / args.steps * 1e3, 3),
                   "what": "the headline step with the recognition batches of a call alternating over two HIP streams (OAR_REC_LANES=2); same results"}
        ocr4.close()

    if rank == 0:
        value = total_pages * args.steps / tmax
        cpu = None
        if args.cpu_pages > 0 and world == 1 and not stub:   # the CPU baseline is timed on rank 0 of the single-GPU run only
            from oracle import pipeline_ref
            torch.set_num_threads(min(os.cpu_count() or 1, 64))
            # SURVEY 8d "ORT-CPU stand-in": only possible when onnxruntime AND operator-supplied weights exist on this box -- say which oracle ran
            try:
                import onnxruntime  # noqa: F401
                have_ort = True
            except Exception:
                have_ort = False
            real = sorted(str(q.relative_to(ROOT)) for q in (ROOT / "models").glob("*.onnx")) if (ROOT / "models").is_dir() else []
            ort_standin = None
            if have_ort and real_weights:   # SURVEY 8d plan (1): the closest stand-in for the reference's CPU path -- ORT CPU EP on the SAME files, this repo's CPU pre / post around it
                from oracle import ort_standin as ort_mod
                ort_standin = ort_mod.time_and_compare(rw_det, rw_rec, rw_chars, host_pages[:max(1, args.cpu_pages)], api, limit_side_len=(size if c3_named else None),
                                                       threads=min(os.cpu_count() or 1, 64))
            ort_note = ("not run: " + ("onnxruntime is not importable on this box" if not have_ort else "onnxruntime present") +
                        ("; no models/*.onnx supplied (synthetic-weight graphs only)" if not real else f"; models present: {real}") +
                        " -- the timed oracle is the torch-CPU port")
            # Two deployments of the same port, each timed on a bounded sample (oracle/cpu_baseline_worker.py; every worker is warmed before the
            # common start signal):  (a) ONE pipeline with all intra-op threads -- the reference's default CPU deployment;  (b) W pipelines of T
            # threads side by side (W x T = the cores): what fills a 64-core host when pages queue up.  `value` is the better of the two.
            ncpu = min(os.cpu_count() or 1, 64)
            cfg_id = {1: 1, 2: 2, 3: 1, 4: 4}[args.config]

            def run_workers(W, Tn, pages_each):
                env = dict(os.environ, OMP_NUM_THREADS=str(Tn), MKL_NUM_THREADS=str(Tn))
                procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline_worker", "--threads", str(Tn), "--pages", str(pages_each), "--seed0", str(1000 + 17 * w),
                                           "--size", str(size), "--lines", str(args.lines), "--config", str(cfg_id), "--c3-graphs", args.c3_graphs], cwd=str(ROOT), env=env,
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for w in range(W)]
                try:
                    for q in procs:
                        while True:
                            ln = q.stdout.readline()
                            if not ln:
                                raise RuntimeError("cpu_baseline worker died during warm-up")
                            if ln.startswith("READY"):
                                break
                    c0 = time.perf_counter()
                    for q in procs:
                        q.stdin.write("GO\n"); q.stdin.flush()
                    outs = [json.loads(q.stdout.readline()) for q in procs]
                    wall = time.perf_counter() - c0
                finally:
                    for q in procs:
                        try:
                            q.stdin.close()
                        except Exception:
                            pass
                        q.wait(timeout=60)
                stage = {}
                for o in outs:
                    for k, v in o["stage_ms_per_page"].items():
                        stage[k] = stage.get(k, 0.0) + v / len(outs)
                return {"workers": W, "threads_each": Tn, "pages": W * pages_each, "images_per_sec": round(W * pages_each / wall, 3), "wall_s": round(wall, 2),
                        "stage_ms_per_page": {k: round(v, 1) for k, v in sorted(stage.items(), key=lambda kv: -kv[1])}}
            single = run_workers(1, ncpu, args.cpu_pages)
            Wn = max(1, ncpu // 8)
            multi = run_workers(Wn, max(1, ncpu // Wn), max(2, args.cpu_pages - 1)) if Wn > 1 else single
            best = max((single, multi), key=lambda r: r["images_per_sec"])
            cpu = {"value": best["images_per_sec"], "unit": "images/sec", "cores": best["workers"] * best["threads_each"], "kind": "port",
                   "deployment": f"{best['workers']} pipeline(s) x {best['threads_each']} torch threads",
                   "one_pipeline": single, "pipelines_side_by_side": multi,
                   "parallel": "per pipeline: thread pool over the crops of a page / of a recognition batch (where the reference's rayon pool fans out: "
                               "processors.rs:113-131, crnn.rs:98-121) + torch-CPU intra-op threads for the networks (channels_last convolutions); "
                               "contour tracing / unclip serial per page as in the reference",
                   "ort_cpu_standin": ort_note,
                   "why_not_29": "stage_ms_per_page: the two networks are > 90 % of a page on the CPU and run in the torch eager interpreter (one ATen call per ONNX node, "
                                 "no graph fusion, oneDNN depthwise kernels); ONNX Runtime's fused graph on the published i9-13900KF is what reaches 34 ms/image",
                   "sample": f"{best['pages']} {size}x{size} synthetic pages (same generator as the GPU workload), one predict per page, det batch 1 / rec batch 16 "
                             "(reference CPU defaults); oracle = C restatement of pre/post + torch-CPU fp32 network; the reference's own "
                             "published CPU figure is 34 ms/image (docs/FAQ.md:22, i9-13900KF, real weights)"}
            if args.config in (1, 3):   # the torch-eager port flatters the ratio; the reference's own published CPU figure is the anchor to quote
                cpu["reference_published"] = {"images_per_sec": 29.4, "source": "docs/FAQ.md:22: 34 ms/image, PP-OCRv6 tiny det+rec, ONNX Runtime CPU on an i9-13900KF (24 cores), real weights",
                                              "gpu_over_published": round(value / 29.4, 1), "gpu_over_this_port": round(value / max(cpu["value"], 1e-9), 1),
                                              "note": "quote the LOWER ratio: the port runs the networks in the torch eager interpreter, ONNX Runtime fuses the graph"}
            if ort_standin:   # real weights + onnxruntime: the stand-in IS the baseline, the torch port stays beside it
                cpu = {"value": ort_standin["images_per_sec"], "unit": "images/sec", "cores": ort_standin["threads"], "kind": "ort-standin", "sample": ort_standin["sample"],
                       "network_parity": ort_standin["parity"], "torch_port": cpu}
        line = {
            "metric": "images/sec end-to-end PP-OCRv6 det+rec", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tmax / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic pages, real weights (registry-verified files)" if real_weights else "synthetic",
            "config": {"baseline_config": args.config,
                       "workload": f"{('PP-OCRv5-server-class det (PP-HGNetV2 + LK-PAN, input 1280) + SVTRv2-class rec' if c3_named else 'server-size LCNet det + SVTR-neck rec (rounds 1-5 stand-in, detector input 960)') if args.config == 2 else 'PP-OCRv6-tiny-class det+rec'}"
                                   f"{' + doc orientation + UVDoc + text-line orientation' if args.config == 4 else ''} "
                                   f"(synthetic-weight graphs: det {det_info['params']} params, rec {rec_info['params']} params, V={vocab}), "
                                   + (f"{total_pages} synthetic {size}x{size} pages block-partitioned over {world} GPU(s)" if args.config == 3 else
                                      f"batch={n_pages} synthetic {size}x{size} pages per GPU") +
                                   f", {args.lines} text lines/page; timed region = u8 pages in pageable HOST memory -> oar_ocr_predict -> oar_ocr_decode -> "
                                   f"sorted boxes + texts + scores on the host{' of rank 0 (RCCL gather inside the region)' if world > 1 else ''}",
                       "pages_per_gpu_per_step": n_pages, "image_batch_size": image_batch, "region_batch_size": args.region_batch,
                       "regions_per_step": gathered["regions"], "text_bytes_per_step": gathered["bytes"], "pages_gathered_per_step": gathered["pages"],
                       "parallelism": f"image-parallel x{world}", "host_cores_per_rank": cores, "host_cpu_ms_per_step": round(host_cpu_ms, 2)},
            "per_rank": per_rank,
            "real_weights": real_weights, "roofline": roof, "cpu_baseline": cpu, "device_resident": dev_res, "pipelined": pipelined, "lighter_graphs_r1_r5": real_size, "rec_two_streams": rec_two, "conv_mfma_util": mfma_util, "kernel_ms_per_step_untimed_pass": breakdown, "predict_passes_total": passes["n"], "csrc_fingerprint": csrc_now,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # release native handles explicitly: destroying them from interpreter teardown (after the HIP runtime / a
    # profiler tool has finalised) was observed to hang the process under rocprofv3
    eng.close()
    sys.stdout.flush()
    sys.stderr.flush()


if __name__ == "__main__":
    main()
