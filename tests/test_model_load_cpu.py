"""Malformed / truncated `.onnx` files must end in OAR_MODEL_LOAD, never in an out-of-bounds read (the parser
trusted the dims of the file).  Host-only: oar_onnx_inspect parses and validates exactly as oar_engine_create does.
Reference behaviour: a bad model file is an error from `Session::commit_from_file` (core/inference/session.rs:30-44),
never a crash."""
import ctypes as C

import numpy as np
import pytest

from oar_ocr_amd import api, build
from oar_ocr_amd.synth import models, onnx_writer as ow


@pytest.fixture(scope="module")
def L():
    build.build_lib()
    L = api.lib()
    L.oar_onnx_inspect.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    return L


def inspect(L, blob: bytes):
    buf = C.create_string_buffer(4096)
    b = (C.c_char * max(len(blob), 1)).from_buffer_copy(blob or b"\0")
    st = L.oar_onnx_inspect(C.cast(b, C.c_void_p), len(blob), buf, 4096)
    err = C.create_string_buffer(1024)
    L.oar_last_error(err, 1024)
    return st, buf.value.decode(errors="replace"), err.value.decode(errors="replace")


def _tensor(name, dims, dt, raw=None, typed=None):
    out = b"".join(ow._f_varint(1, d) for d in dims) + ow._f_varint(2, dt) + ow._f_str(8, name)
    if raw is not None:
        out += ow._f_bytes(9, raw)
    if typed is not None:
        field, vals = typed
        out += b"".join((ow._f_float(field, v) if field == 4 else ow._f_varint(field, v)) for v in vals)
    return out


def _model(inits, nodes):
    g = b"".join(ow._f_bytes(1, n) for n in nodes) + ow._f_str(2, "g") + b"".join(ow._f_bytes(5, t) for t in inits)
    g += ow._f_bytes(11, ow.value_info("x", [1, 4, 8, 8])) + ow._f_bytes(12, ow.value_info("y", [1, 4, 8, 8]))
    return ow._f_varint(1, 8) + ow._f_bytes(7, g) + ow._f_bytes(8, ow._f_str(1, "") + ow._f_varint(2, 17))


def test_good_models_pass_and_report_their_ops(L):
    for blob in (models.build_det("tiny")[0], models.build_rec("tiny", vocab=97)[0]):
        st, summary, err = inspect(L, blob)
        assert st == api.OAR_OK, err
        assert "Conv:" in summary and "!" not in summary and "input=x" in summary


def test_round6_graph_families_load_and_have_the_files_sizes(L):
    """The graphs BASELINE's configs are timed on since round 6: every operator supported (host-only check: oar_onnx_inspect validates as oar_engine_create does), and
    parameter counts within 3 % of the files they stand for (reference registry.rs:83, :84, :77, :25 -- bytes / 4)."""
    sizes = {"det tiny_full": (models.build_det("tiny_full"), 1780590), "rec tiny_full": (models.build_rec("tiny_full", vocab=6906), 4462639),
             "det server_hgnet": (models.build_det("server_hgnet"), 88116836), "rec svtrv2": (models.build_rec("svtrv2", vocab=6625), 84196641)}
    for name, ((blob, info), file_bytes) in sizes.items():
        assert abs(info["params"] * 4 / file_bytes - 1) < 0.03, (name, info["params"], file_bytes)      # (svtrv2: -2.7 %: the real file also carries BatchNorm statistics of its stem)
        st, summary, err = inspect(L, blob)
        assert st == api.OAR_OK and "!" not in summary, (name, err, summary)
    _, summary, _ = inspect(L, sizes["rec svtrv2"][0][0])
    assert "Erf:" in summary and "LayerNormalization:" in summary and "Softmax:" in summary      # GELU arrives decomposed, attention op by op: the engine's rewrite passes fuse them
    _, summary, _ = inspect(L, sizes["det server_hgnet"][0][0])
    assert "MaxPool:" in summary and "Concat:" in summary and "GlobalAveragePool:" in summary


def test_unsupported_operator_is_named(L):
    m = _model([], [ow.node("NonMaxSuppression", ["x"], ["y"])])
    st, summary, err = inspect(L, m)
    assert st == api.OAR_UNSUPPORTED_OP and "!NonMaxSuppression:1" in summary and "NonMaxSuppression" in err


def test_truncated_file_is_a_load_error(L):
    blob = models.build_det("tiny")[0]
    for cut in (1, 7, len(blob) // 3, len(blob) // 2, len(blob) - 5):
        st, _, err = inspect(L, blob[:cut])
        assert st == api.OAR_MODEL_LOAD, (cut, st, err)
    assert inspect(L, b"")[0] == api.OAR_MODEL_LOAD


@pytest.mark.parametrize("case", ["neg_dim", "huge_dims", "raw_short_f32", "raw_short_i64", "raw_short_i32", "bool_raw_short", "typed_short_f32",
                                  "typed_short_i64", "typed_long_bool", "f64_short"])
def test_initializer_payload_must_match_dims(L, case):
    relu = [ow.node("Relu", ["x"], ["y"])]
    t = {
        "neg_dim": _tensor("w", [4, -2], 1, raw=b"\0" * 32),
        "huge_dims": _tensor("w", [1 << 31, 1 << 31, 1 << 31], 1, raw=b"\0" * 16),
        "raw_short_f32": _tensor("w", [4, 4], 1, raw=b"\0" * 60),
        "raw_short_i64": _tensor("w", [4], 7, raw=b"\0" * 24),
        "raw_short_i32": _tensor("w", [4], 6, raw=b"\0" * 12),
        "bool_raw_short": _tensor("w", [64], 9, raw=b"\1" * 8),          # the r1 parser read 64 bytes here
        "typed_short_f32": _tensor("w", [8], 1, typed=(4, [1.0, 2.0])),
        "typed_short_i64": _tensor("w", [8], 7, typed=(7, [1, 2, 3])),
        "typed_long_bool": _tensor("w", [2], 9, typed=(5, [1, 0, 1, 1])),
        "f64_short": _tensor("w", [4], 11, raw=b"\0" * 24),
    }[case]
    st, _, err = inspect(L, _model([t], relu))
    assert st == api.OAR_MODEL_LOAD, (case, st, err)
    assert "w" in err or "dimension" in err or "overflow" in err


@pytest.mark.parametrize("case", ["conv_w_rank2", "conv_bias_len", "bn_len", "bn_empty", "gemm_rank3"])
def test_initializer_ranks_are_checked_before_they_are_indexed(L, case):
    f = np.float32
    if case == "conv_w_rank2":
        inits, nodes = [ow.tensor_proto("w", np.zeros((4, 4), f))], [ow.node("Conv", ["x", "w"], ["y"])]
    elif case == "conv_bias_len":
        inits = [ow.tensor_proto("w", np.zeros((4, 4, 1, 1), f)), ow.tensor_proto("b", np.zeros(3, f))]
        nodes = [ow.node("Conv", ["x", "w", "b"], ["y"])]
    elif case == "bn_len":   # the BN fold read ga/be/mu/va[c] for c < len(scale)
        inits = [ow.tensor_proto("w", np.zeros((4, 4, 1, 1), f)), ow.tensor_proto("s", np.ones(4, f)), ow.tensor_proto("b", np.zeros(4, f)),
                 ow.tensor_proto("m", np.zeros(2, f)), ow.tensor_proto("v", np.ones(4, f))]
        nodes = [ow.node("Conv", ["x", "w"], ["c"]), ow.node("BatchNormalization", ["c", "s", "b", "m", "v"], ["y"])]
    elif case == "bn_empty":
        inits = [ow.tensor_proto(n, np.zeros(0, f)) for n in "sbmv"]
        nodes = [ow.node("BatchNormalization", ["x", "s", "b", "m", "v"], ["y"])]
    else:
        inits, nodes = [ow.tensor_proto("w", np.zeros((2, 4, 4), f))], [ow.node("Gemm", ["x", "w"], ["y"])]
    st, _, err = inspect(L, _model(inits, nodes))
    assert st == api.OAR_MODEL_LOAD, (case, st, err)


def test_fuzzed_bytes_never_crash(L):
    blob = bytearray(models.build_rec("tiny", vocab=97)[0])
    rng = np.random.default_rng(0)
    for _ in range(200):
        b = bytearray(blob)
        for pos in rng.integers(0, min(len(b), 4096), size=4):   # the head of the file holds the graph structure
            b[pos] = int(rng.integers(0, 256))
        st, _, _ = inspect(L, bytes(b))
        assert st in (api.OAR_OK, api.OAR_MODEL_LOAD, api.OAR_UNSUPPORTED_OP)
