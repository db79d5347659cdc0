"""Minimal ONNX (protobuf) writer for tests: ModelProto with metadata_props, initializers and nodes, using the naming
conventions of a torch.onnx export of the icefall wrappers (scoped node names, anonymous MatMul weights stored [in, out])."""
import struct

import numpy as np


def _varint(x: int) -> bytes:
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fn: int, payload: bytes) -> bytes:
    return _varint((fn << 3) | 2) + _varint(len(payload)) + payload


def _vi(fn: int, x: int) -> bytes:
    return _varint(fn << 3) + _varint(x)


def tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    arr = np.ascontiguousarray(arr)
    code = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.uint8): 2, np.dtype(np.int8): 3}[arr.dtype]
    b = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, code) + _ld(8, name.encode())
    if raw:
        b += _ld(9, arr.tobytes())
    elif code == 1:
        b += _ld(4, arr.astype("<f4").tobytes())
    else:
        b += _ld(7, b"".join(_varint(int(x) & ((1 << 64) - 1)) for x in arr.reshape(-1)))
    return b


def node(name: str, op: str, inputs, outputs) -> bytes:
    return b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs) + _ld(3, name.encode()) + _ld(4, op.encode())


def model(meta: dict, initializers, nodes) -> bytes:
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"k2hip-test") + b"".join(_ld(5, t) for t in initializers)
    m = _vi(1, 8) + _ld(2, b"pytorch") + _ld(7, g)
    for k, v in meta.items():
        m += _ld(14, _ld(1, k.encode()) + _ld(2, str(v).encode()))
    return m
