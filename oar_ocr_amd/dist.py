"""Image-parallel sharding of pages over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" for the CPU tests).

Pages are independent units: rank r owns a contiguous block of the page list and runs the
whole det -> crop -> rec path on it.  There is NO collective on the data path; the only exchange is the final
gather of the (tiny, variable-length) results to rank 0.  Crops are pooled per shard, so a recognition batch never
mixes pages of different ranks (the reference calls the resulting difference "padding-induced output drift",
domain/adapters/text_recognition_adapter.rs:118-123; it is bounded by the 1e-3 float budget).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Static block partition: rank r gets [r*n/G, (r+1)*n/G) with the remainder spread over the first ranks."""
    if world_size <= 0 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad n_items/world_size/rank")
    try:
        from . import api   # oar_shard_range (C ABI: the same partition for a Rust / C host)
        return api.shard_range(n_items, world_size, rank)
    except (OSError, ImportError):   # library not built on this host: the same arithmetic (oar_shard_range, ctc_host.cc)
        base, rem = divmod(n_items, world_size)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, local_rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        import torch
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_results(local_results: list, dst: int = 0) -> List | None:
    """Gathers each rank's per-page results on `dst`, restoring global page order (block partition => concatenation)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_results)
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local_results, out, dst=dst)
    if out is None:
        return None
    merged = []
    for part in out:
        merged.extend(part)
    return merged


def gather_bytes(blob: bytes, dst: int, device) -> List[bytes] | None:
    """Variable-length gather of one byte string per rank on `dst` (the final, tiny exchange of SURVEY 8e): an all_gather of
    the lengths, then ONE gather of the blobs padded to the longest -- two collectives on `device` ("cuda" for RCCL over xGMI,
    "cpu" for gloo), no pickling.  Returns the blobs in rank order on `dst`, None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [blob]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8)
    if blob:
        buf[:len(blob)] = torch.from_numpy(np.frombuffer(blob, np.uint8).copy())
    buf = buf.to(device)
    out = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [bytes(o[:sizes[i]].cpu().numpy().tobytes()) for i, o in enumerate(out)]


def sharded_predict(predict_fn, pages: Sequence, dst: int = 0):
    """Runs predict_fn on this rank's shard of `pages` and gathers the results on `dst` in page order."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    a, b = shard_range(len(pages), world, rank)
    local = predict_fn(pages[a:b]) if b > a else []
    return gather_results(local, dst)
