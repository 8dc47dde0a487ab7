"""bench.py's multi-rank control flow without a GPU: `python bench.py --gpus 2` must start its OWN two ranks, shard the
pages, run the step loop with the per-step gather inside the timed region, and print ONE JSON line with n_gpus = 2
(a plain `--gpus 8` silently measured one GPU).  The HIP pipeline is replaced by bench.StubEngine
(--stub-engine); everything else -- launcher, rendezvous on 127.0.0.1, core pinning, shard_range, gather_bytes over gloo,
max-over-ranks timing -- is the real code."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*extra):
    env = dict(os.environ, OAR_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--stub-engine", "--steps", "3", "--warmup", "1", "--lines", "4", "--size", "128", *extra],
                       capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_weak_scaling():
    out = _run("--gpus", "2", "--pages", "5")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    c = out["config"]
    assert c["pages_per_gpu_per_step"] == 5 and c["pages_gathered_per_step"] == 10 and c["regions_per_step"] == 30   # both ranks' results reached rank 0
    assert abs(out["value"] - 10 * 3 / (out["ms_per_step"] * 3 / 1e3)) / out["value"] < 1e-2               # whole-job pages / max-over-ranks time
    # per-rank record: every rank reports what it saw, in rank order; the line's time is the slowest rank's
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["world_size_seen"] == 2 and r["backend"] == "gloo" for r in pr)
    assert all(r["regions"] == 15 and r["host_cores"] >= 1 and r["host_cpu_ms_per_step"] >= 0 for r in pr)
    assert abs(max(r["ms_per_step"] for r in pr) - out["ms_per_step"]) < 1e-2


def test_bench_config3_block_partitions_a_fixed_page_count():
    out = _run("--gpus", "2", "--config", "3", "--pages", "11")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["pages_gathered_per_step"] == 11 and out["config"]["pages_per_gpu_per_step"] == 6   # rank 0's share of 11


def test_bench_single_rank_needs_no_launcher():
    out = _run("--pages", "4")
    assert out["n_gpus"] == 1 and out["config"]["pages_gathered_per_step"] == 4 and out["per_rank"] is None


def test_packed_pages_roundtrip():
    from oar_ocr_amd import api
    p = api.PackedPages(np.array([0, 2, 2, 3], np.uint32), np.arange(24, dtype=np.float32).reshape(3, 4, 2), np.array([.5, .25, 1], np.float32),
                        "ab中c".encode(), np.array([0, 2, 5, 6], np.uint64))
    q = api.PackedPages.from_bytes(p.to_bytes())
    assert np.array_equal(p.region_offsets, q.region_offsets) and np.array_equal(p.points, q.points) and np.array_equal(p.scores, q.scores)
    assert [q.text(k) for k in range(3)] == ["ab", "中", "c"]
