"""world_size-2 gloo test of the image-parallel sharding path (host logic only: the per-shard compute is a stub,
the real compute needs a GPU and is covered by -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oar_ocr_amd import dist as oard


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [oard.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        oard.shard_range(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, l, w = oard.init_from_env("gloo")
    pages = list(range(11))
    res = oard.sharded_predict(lambda ps: [{"page": p, "rank": r, "regions": p % 3} for p in ps], pages)
    # weak-scaling style aggregate, as bench.py does
    t = torch.tensor([float(len(pages))])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if r == 0:
        q.put(res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_predict_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [x["page"] for x in res] == list(range(11))          # global page order restored
    assert [x["rank"] for x in res] == [0] * 6 + [1] * 5        # block partition
