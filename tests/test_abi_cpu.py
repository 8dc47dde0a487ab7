"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/oar_mi355x.h declares; host-side logic mirrors the reference; and the product path fails LOUDLY
(no CPU fallback) when no GPU is visible."""
import re
from pathlib import Path

import numpy as np
import pytest

from oar_ocr_amd import api, build

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def L():
    build.build_lib()
    return api.lib()


def test_header_symbols_are_exported(L):
    hdr = (ROOT / "include" / "oar_mi355x.h").read_text()
    declared = set(re.findall(r"\b(oar_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"oar_status"}
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(api.EXPORTS) <= declared


def test_product_path_does_not_import_oracle():
    for p in (ROOT / "oar_ocr_amd").rglob("*"):
        if p.suffix in (".py", ".cc", ".h", ".hip") and "lib/obj" not in str(p):
            txt = p.read_text()
            assert "from oracle" not in txt and "import oracle" not in txt and "liboar_oracle" not in txt and "oracle/" not in txt, p


def test_fails_loudly_without_gpu(L):
    if api.device_count() > 0:
        pytest.skip("GPU present")
    from oar_ocr_amd.synth import models
    det, _ = models.build_det("tiny")
    with pytest.raises(api.OCRError) as e:
        api.OrtInfer(det)
    assert e.value.code == api.OAR_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(api.OCRError) as e:
        api.k_threshold(np.zeros((4, 4), np.float32), 0.3)
    assert e.value.code == api.OAR_DEVICE


def test_model_load_errors_are_reported(L):
    # mirrors core/inference/mod.rs:123-128 (missing/garbage model => error, not a crash)
    for blob in (b"", b"\x00\x01garbage"):
        with pytest.raises(api.OCRError) as e:
            api.OrtInfer(blob)
        assert e.value.code in (api.OAR_MODEL_LOAD, api.OAR_DEVICE, api.OAR_INVALID_INPUT)


# ---- host logic mirrored from the reference tests
def test_ctc_label_decode_k9():
    # processors/decode.rs:679-745
    winners = [[(0, .9), (1, .8), (1, .7), (0, .6), (1, .5), (2, .4), (2, .3)],
               [(3, .95), (3, .85), (4, .75), (3, .65), (0, .55), (2, .45), (0, .35)]]
    idx = np.array([[w[0] for w in s] for s in winners], np.int64)
    pr = np.array([[w[1] for w in s] for s in winners], np.float32)
    dec = api.CTCLabelDecode(["a", "b", "c"], use_space_char=False)
    texts, scores, pos, cols, lens = dec.decode_argmax(idx, pr, 2, 7)
    f = np.float32
    assert texts == ["aab", "ccb"]
    assert scores == [float((f(.8) + f(.5) + f(.4)) / f(3)), float((f(.95) + f(.65) + f(.45)) / f(3))]
    assert cols == [[1, 4, 5], [0, 3, 5]] and lens == [7, 7]
    assert dec.decode_argmax(np.zeros(0), np.zeros(0), 2, 0)[0] == []   # decode.rs:747-757


def test_ctc_word_boxes_k20():
    # src/oarocr/ocr.rs:1197-1232
    b = api.ctc_word_boxes(np.array([[0, 0], [100, 0], [100, 20], [0, 20]], np.float32), "ABC", [1, 4, 7], 10, 5.0, 5.0)
    rng = [(float(x[:, 0].min()), float(x[:, 0].max())) for x in b]
    assert np.allclose(rng, [(0, 30), (30, 60), (60, 100)], atol=1e-5)


def test_word_boxes_c_abi_equals_the_oracle_on_random_lines():
    """row a21 behind the boundary (VERDICT r2 missing #3): oar_ctc_word_boxes / oar_char_positions_to_word_boxes against the numpy
    restatement of ocr.rs:949-1072 -- Latin, CJK (all five is_cjk ranges), mixed, more columns than characters and the reverse,
    padded batches (wh_ratio < max_wh_ratio), degenerate inputs.  Bit-exact: the same f32 operations in the same order."""
    from oracle import cpu_ref as R
    rng = np.random.default_rng(5)
    alphabet = ["a", "Z", "7", " ", "é", "中", "文", "㐀", "\U00020000", "\U0002A700", "\U0002B740", "あ", "한"]
    n_boxes = 0
    for _ in range(300):
        x0, y0 = rng.uniform(0, 500, 2)
        w, h = rng.uniform(5, 600), rng.uniform(5, 60)
        ang = rng.uniform(-0.1, 0.1)
        base = np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float32)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]], np.float32)
        box = (base @ rot.T + np.array([x0, y0], np.float32)).astype(np.float32)
        T = int(rng.integers(1, 80))
        n_cols = int(rng.integers(0, min(T, 30) + 1))
        cols = np.sort(rng.choice(T, n_cols, replace=False)).astype(np.uint32)
        text = "".join(rng.choice(alphabet, int(rng.integers(0, 32))))
        wh, mwh = np.float32(rng.uniform(0.5, 20)), None
        mwh = np.float32(max(wh, rng.uniform(0.5, 25)))
        got = api.ctc_word_boxes(box, text, cols, T, float(wh), float(mwh))
        ref = R.ctc_word_boxes(box, text, [int(c) for c in cols], T, float(wh), float(mwh))
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)
        n_boxes += len(ref)
        pos = rng.random(int(rng.integers(0, 20))).astype(np.float32)
        cc = int(rng.integers(0, 25))
        got = api.char_positions_to_word_boxes(box, pos, cc)
        ref = R.char_positions_to_word_boxes(box, pos, cc)
        assert len(got) == len(ref) and all(np.array_equal(g, r) for g, r in zip(got, ref))
    assert n_boxes > 1000
    assert api.ctc_word_boxes(np.zeros((4, 2), np.float32), "abc", [0, 1], 0, 1.0, 1.0) == []          # seq_len == 0
    assert api.ctc_word_boxes(np.zeros((4, 2), np.float32), "", [0, 1], 10, 1.0, 1.0) == []             # empty text
    assert api.ctc_word_boxes(np.zeros((4, 2), np.float32), "abc", [0, 1], 10, 0.0, 1.0) == []          # effective columns <= EPSILON


def test_builder_validates_batch_sizes():
    # src/oarocr/ocr.rs:250-255,419-430
    with pytest.raises(api.OCRError) as e:
        api.OAROCRBuilder(b"x", b"y", ["a"]).region_batch_size(0).build()
    assert "region_batch_size" in str(e.value) and "1..=4096" in str(e.value)
    with pytest.raises(api.OCRError):
        api.OAROCRBuilder(b"x", b"y", ["a"]).image_batch_size(5000).build()


def test_dict_parsing():
    # decode.rs:120: first char per line, empty lines vanish; blank '\0' prepended, ' ' appended
    chars = api.read_dict("ab\n\nc\n")
    assert chars == ["a", "c"]
    dec = api.CTCLabelDecode(chars)
    assert dec.character == ["\0", "a", "c", " "]


# ---- the library-side CTC decoder (oar_ctc_decode): same vectors as the Python mirror, plus a randomised cross-check
def test_c_ctc_decode_k9_k10_and_score_filter(L):
    winners = [[(0, .9), (1, .8), (1, .7), (0, .6), (1, .5), (2, .4), (2, .3)],
               [(3, .95), (3, .85), (4, .75), (3, .65), (0, .55), (2, .45), (0, .35)]]
    idx = np.array([[w[0] for w in s] for s in winners], np.int64)
    pr = np.array([[w[1] for w in s] for s in winners], np.float32)
    d = api.CtcDict("a\nb\nc\n", use_space_char=False)
    assert d.classes == 4
    r = d.decode(idx, pr, 2, 7)
    f = np.float32
    assert r.texts == ["aab", "ccb"]                                         # decode.rs:679-745 (index 4 is outside the table)
    assert r.scores.tolist() == [float((f(.8) + f(.5) + f(.4)) / f(3)), float((f(.95) + f(.65) + f(.45)) / f(3))]
    assert [c.tolist() for c in r.char_cols] == [[1, 4, 5], [0, 3, 5]] and r.seq_len.tolist() == [7, 7]
    assert np.array_equal(r.char_positions[0], (np.array([1, 4, 5], f) / f(7)))
    assert d.decode(np.zeros(0), np.zeros(0), 2, 0).texts == []              # decode.rs:747-757: [2, 0, V] collapses to nothing
    # the adapter keeps slot and score but blanks text / positions below the threshold (text_recognition_adapter.rs:88-101)
    r = d.decode(idx, pr, 2, 7, score_threshold=0.6)
    assert r.texts == ["", "ccb"] and r.kept.tolist() == [False, True] and len(r.char_cols[0]) == 0
    assert r.scores[0] == np.float32((f(.8) + f(.5) + f(.4)) / f(3))


def test_c_ctc_dict_follows_rust_lines_semantics(L):
    text = "ab\r\n\n中文\nc\rd\n\re"            # CRLF, an empty line, a multi-byte first char, a bare \r inside / leading a line
    assert api.dict_lines(text) == ["ab", "", "中文", "c\rd", "\re"]
    assert api.read_dict(text) == ["a", "中", "c", "\r"]
    d = api.CtcDict(text, use_space_char=True)
    assert d.classes == 1 + 4 + 1
    r = d.decode(np.array([[1, 2, 0, 2, 5, 3, 4]], np.int64), np.full((1, 7), 0.5, np.float32), 1, 7)
    assert r.texts == ["a中中 c\r"]
    assert api.dict_lines("x\n") == ["x"] and api.dict_lines("") == [] and api.dict_lines("\n") == [""]


def test_c_ctc_decode_matches_the_python_mirror_and_the_oracle(L):
    from oracle import cpu_ref
    rng = np.random.default_rng(0)
    entries = [chr(0x4E00 + i) for i in range(300)] + list("abcdefghij")
    d = api.CtcDict.from_entries(entries)
    py = api.CTCLabelDecode(entries)
    charset = cpu_ref.ctc_charset(entries)
    assert d.classes == len(py.character) == len(charset)
    for T in (1, 5, 40, 133):
        n = 17
        idx = rng.integers(0, d.classes + 3, (n, T))      # includes out-of-table classes
        idx[rng.random((n, T)) < 0.5] = 0
        rep = rng.random((n, T)) < 0.3                       # repeats, so the collapse rule is exercised
        for t in range(1, T):
            idx[rep[:, t], t] = idx[rep[:, t], t - 1]
        pr = rng.random((n, T)).astype(np.float32)
        r = d.decode(idx, pr, n, T)
        texts, scores, pos, cols, lens = py.decode_argmax(idx, pr, n, T)
        assert r.texts == texts and r.scores.tolist() == [np.float32(s) for s in scores]
        assert [c.tolist() for c in r.char_cols] == cols
        ot, osc, ocols = cpu_ref.ctc_decode(idx.astype(np.int64), pr, n, T, charset)[:3]
        assert r.texts == list(ot) and np.array_equal(r.scores, np.asarray(osc, np.float32))


def test_c_ocr_decode_on_a_ragged_result_equals_the_per_batch_decoder(L):
    """oar_ocr_decode (round 5: one pass into flat buffers) over a hand-built oar_ocr_result -- regions of different lengths, empty
    regions, out-of-table classes, a score threshold that drops some -- against oar_ctc_decode run region by region."""
    rng = np.random.default_rng(3)
    entries = [chr(0x4E00 + i) for i in range(200)] + list("abcdefghij") + ["\U0001F600"]          # 1-, 3- and 4-byte characters
    d = api.CtcDict.from_entries(entries)
    Ts = [0, 1, 7, 40, 0, 133, 40, 2, 64]
    idx_l, pr_l = [], []
    for T in Ts:
        i = rng.integers(0, d.classes + 3, T)
        i[rng.random(T) < 0.5] = 0
        for t in range(1, T):
            if rng.random() < 0.3:
                i[t] = i[t - 1]
        idx_l.append(i.astype(np.int64)); pr_l.append(rng.random(T).astype(np.float32))
    offs = np.concatenate([[0], np.cumsum(Ts)]).astype(np.uint64)
    idx = np.concatenate(idx_l) if sum(Ts) else np.zeros(0, np.int64)
    pr = np.concatenate(pr_l) if sum(Ts) else np.zeros(0, np.float32)
    seq = np.array(Ts, np.uint32)
    res = api.OcrResult()
    res.n_images, res.n_regions = 1, len(Ts)
    res.ctc_offsets = offs.ctypes.data_as(api.C.POINTER(api.C.c_uint64))
    res.ctc_indices = idx.ctypes.data_as(api.C.POINTER(api.C.c_int64))
    res.ctc_probs = pr.ctypes.data_as(api.C.POINTER(api.C.c_float))
    res.seq_len = seq.ctypes.data_as(api.C.POINTER(api.C.c_uint32))
    for thr in (0.0, 0.5):
        got = d.decode_ocr(res, thr)
        for k, T in enumerate(Ts):
            if T == 0:
                assert got.texts[k] == "" and got.scores[k] == 0.0 and len(got.char_cols[k]) == 0 and got.seq_len[k] == 0
                continue
            ref = d.decode(idx_l[k][None], pr_l[k][None], 1, T, score_threshold=thr)
            assert got.texts[k] == ref.texts[0] and got.scores[k] == ref.scores[0] and bool(got.kept[k]) == bool(ref.kept[0])
            assert np.array_equal(got.char_cols[k], ref.char_cols[0]) and np.array_equal(got.char_positions[k], ref.char_positions[0])
            assert got.seq_len[k] == T
        assert any(got.kept) and (thr == 0.0 or not all(got.kept))
