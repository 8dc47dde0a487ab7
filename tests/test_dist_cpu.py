"""world_size-2 gloo test of the image-parallel sharding path (host logic only: the per-shard compute is a stub,
the real compute needs a GPU and is covered by -m gpu tests)."""
import os
import socket

import numpy as np
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


def _fake_result(rank: int, n_pages: int, per_page):
    """An oar_ocr_result + oar_text_result pair built by hand (what a rank holds after predict + decode)."""
    import ctypes as C
    from oar_ocr_amd import api
    nr = int(sum(per_page))
    ro = np.concatenate([[0], np.cumsum(per_page)]).astype(np.uint32)
    pts = (np.arange(nr * 8, dtype=np.float32).reshape(nr, 4, 2) + 1000 * rank)
    texts = [f"r{rank}-région{k}-文字".encode() if k % 3 else b"" for k in range(nr)]
    to = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.uint64)
    utf8 = b"".join(texts)
    sc = (np.arange(nr, dtype=np.float32) / max(nr, 1) + rank).astype(np.float32)
    res, txt = api.OcrResult(), api.TextResult()
    keep = [ro, pts, sc, to, C.create_string_buffer(utf8, len(utf8) + 1)]
    res.n_images, res.n_regions = n_pages, nr
    res.region_offsets = ro.ctypes.data_as(C.POINTER(C.c_uint32))
    res.points = pts.ctypes.data_as(C.POINTER(C.c_float))
    txt.n = nr
    txt.text_offsets = to.ctypes.data_as(C.POINTER(C.c_uint64))
    txt.utf8 = C.cast(keep[4], C.POINTER(C.c_char))
    txt.scores = sc.ctypes.data_as(C.POINTER(C.c_float))
    return res, txt, keep, api.PackedPages(ro, pts, sc, utf8, to)


def test_c_abi_pack_and_merge_equal_the_python_wire_format():
    """oar_ocr_pack == PackedPages.to_bytes() byte for byte; oar_packed_merge of the ranks' blobs == concatenating the unpacked parts
    (offsets rebased, UTF-8 intact, empty pages / empty texts / an empty rank included); corrupt blobs are refused."""
    import ctypes as C
    from oar_ocr_amd import api
    L = api.lib()
    blobs, parts = [], []
    for rank, per_page in enumerate([[2, 0, 3], [], [1], [0, 4]]):
        res, txt, keep, ref = _fake_result(rank, len(per_page), per_page)
        blob, ln = C.POINTER(C.c_uint8)(), C.c_size_t(0)
        assert L.oar_ocr_pack(C.byref(res), C.byref(txt), C.byref(blob), C.byref(ln)) == 0
        b = C.string_at(blob, ln.value)
        L.oar_blob_free(blob)
        assert b == ref.to_bytes()
        blobs.append(b); parts.append(ref)
    m = api.PackedPages.merge(blobs)
    assert len(m.region_offsets) - 1 == sum(len(p.region_offsets) - 1 for p in parts)
    assert np.array_equal(m.points, np.concatenate([p.points for p in parts]))
    assert np.array_equal(m.scores, np.concatenate([p.scores for p in parts]))
    assert m.utf8 == b"".join(p.utf8 for p in parts)
    k = 0
    page = 0
    for p in parts:
        for i in range(len(p.region_offsets) - 1):
            assert m.region_offsets[page + 1] - m.region_offsets[page] == p.region_offsets[i + 1] - p.region_offsets[i]
            page += 1
        for j in range(len(p.scores)):
            assert m.text(k) == p.text(j)
            k += 1
    assert api.PackedPages.merge([]).scores.size == 0
    with pytest.raises(api.OCRError, match="header says"):
        api.PackedPages.merge([blobs[0][:-1]])
    bad = bytearray(blobs[0]); bad[24 + 4] = 99     # region_offsets[1] beyond the region count
    with pytest.raises(api.OCRError, match="out of range"):
        api.PackedPages.merge([bytes(bad)])


def test_c_shard_range_is_the_block_partition():
    from oar_ocr_amd import api
    for n in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [api.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(api.OCRError):
        api.shard_range(4, 2, 2)
