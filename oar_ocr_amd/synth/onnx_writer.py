"""Minimal ONNX (protobuf wire-format) writer.

The `onnx` python package is not installed in this image, and the reference's model files
(`.onnx`, fetched from ModelScope -- oar-ocr-core/src/core/download/registry.rs:83-84) are not in
the container, so synthetic graphs of the same topology are serialised by hand.  Only the subset of
onnx.proto needed for inference graphs is implemented (ModelProto/GraphProto/NodeProto/
AttributeProto/TensorProto/ValueInfoProto).
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence

import numpy as np

FLOAT, INT32, INT64 = 1, 6, 7


def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _f_varint(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field: int, b: bytes) -> bytes:
    return _key(field, 2) + _varint(len(b)) + b


def _f_str(field: int, s: str) -> bytes:
    return _f_bytes(field, s.encode("utf-8"))


def _f_float(field: int, v: float) -> bytes:
    return _key(field, 5) + struct.pack("<f", float(v))


def tensor_proto(name: str, arr: np.ndarray) -> bytes:
    arr = np.asarray(arr)
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    elif arr.dtype == np.int32:
        dt = INT32
    else:
        raise TypeError(arr.dtype)
    out = b"".join(_f_varint(1, d) for d in arr.shape)
    out += _f_varint(2, dt)
    out += _f_str(8, name)
    out += _f_bytes(9, np.ascontiguousarray(arr).tobytes())
    return out


def attr(name: str, value) -> bytes:
    out = _f_str(1, name)
    if isinstance(value, float):
        out += _f_float(2, value) + _f_varint(20, 1)
    elif isinstance(value, (int, np.integer)) and not isinstance(value, bool):
        out += _f_varint(3, int(value)) + _f_varint(20, 2)
    elif isinstance(value, str):
        out += _f_bytes(4, value.encode()) + _f_varint(20, 3)
    elif isinstance(value, np.ndarray):
        out += _f_bytes(5, tensor_proto("", value)) + _f_varint(20, 4)
    elif isinstance(value, (list, tuple)) and all(isinstance(v, float) for v in value) and len(value) > 0:
        out += b"".join(_f_float(7, v) for v in value) + _f_varint(20, 6)
    elif isinstance(value, (list, tuple)):
        out += b"".join(_f_varint(8, int(v)) for v in value) + _f_varint(20, 7)
    else:
        raise TypeError(f"attr {name}: {type(value)}")
    return out


def node(op: str, inputs: Sequence[str], outputs: Sequence[str], name: str = "", **attrs) -> bytes:
    out = b"".join(_f_str(1, i) for i in inputs)
    out += b"".join(_f_str(2, o) for o in outputs)
    if name:
        out += _f_str(3, name)
    out += _f_str(4, op)
    for k, v in attrs.items():
        out += _f_bytes(5, attr(k, v))
    return out


def value_info(name: str, shape: Iterable, elem_type: int = FLOAT) -> bytes:
    dims = b""
    for d in shape:
        if isinstance(d, str):
            dims += _f_bytes(1, _f_str(2, d))
        else:
            dims += _f_bytes(1, _f_varint(1, int(d)))
    tensor_type = _f_varint(1, elem_type) + _f_bytes(2, dims)
    type_proto = _f_bytes(1, tensor_type)
    return _f_str(1, name) + _f_bytes(2, type_proto)


class GraphBuilder:
    """Accumulates nodes/initializers; `model()` returns serialised ModelProto bytes."""

    def __init__(self, name: str, opset: int = 17):
        self.name = name
        self.opset = opset
        self.nodes: list[bytes] = []
        self.inits: list[bytes] = []
        self.inputs: list[bytes] = []
        self.outputs: list[bytes] = []
        self._uid = 0
        self.n_params = 0

    def uid(self, prefix: str) -> str:
        self._uid += 1
        return f"{prefix}_{self._uid}"

    def add_input(self, name, shape, elem_type: int = FLOAT):
        self.inputs.append(value_info(name, shape, elem_type))

    def add_output(self, name, shape, elem_type: int = FLOAT):
        self.outputs.append(value_info(name, shape, elem_type))

    def init(self, arr: np.ndarray, prefix: str = "w") -> str:
        name = self.uid(prefix)
        self.inits.append(tensor_proto(name, arr))
        if arr.dtype == np.float32:
            self.n_params += arr.size
        return name

    def op(self, op_type: str, inputs: Sequence[str], n_out: int = 1, **attrs):
        outs = [self.uid(op_type.lower()) for _ in range(n_out)]
        self.nodes.append(node(op_type, inputs, outs, name=self.uid("n"), **attrs))
        return outs[0] if n_out == 1 else outs

    def model(self) -> bytes:
        g = b"".join(_f_bytes(1, n) for n in self.nodes)
        g += _f_str(2, self.name)
        g += b"".join(_f_bytes(5, t) for t in self.inits)
        g += b"".join(_f_bytes(11, i) for i in self.inputs)
        g += b"".join(_f_bytes(12, o) for o in self.outputs)
        m = _f_varint(1, 8)                      # ir_version
        m += _f_str(2, "oar_ocr_amd.synth")      # producer_name
        m += _f_bytes(7, g)
        m += _f_bytes(8, _f_str(1, "") + _f_varint(2, self.opset))
        return m
