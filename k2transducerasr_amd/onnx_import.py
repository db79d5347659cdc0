"""ONNX initializers -> ``.k2w``.

The reference's on-disk format is three ONNX files per model (K2TransducerAsr/OfflineModel.cs:111-115: encoder, decoder,
joiner) whose custom-metadata maps carry the configuration (OfflineModel.cs:31-72, OnlineModel.cs:38-166).  This module reads
such files WITHOUT the ``onnx`` package (not installed here): a small protobuf wire-format reader pulls out
``ModelProto.metadata_props``, ``GraphProto.initializer`` and ``GraphProto.node`` and maps every weight back to its icefall
state-dict name:

* initializers that kept their parameter name (``encoder.encoders.0.layers.0.norm.bias`` ...) map directly;
* anonymous ones (``onnx::MatMul_1234``) are named after the scope of the node that consumes them
  (``/encoder/encoders.0/layers.0/feed_forward1/in_proj/MatMul`` -> ``...feed_forward1.in_proj.weight``; a MatMul weight is
  stored ``[in, out]`` in ONNX and transposed back to torch's ``[out, in]``);
* the export wrappers' names are translated: ``encoder_proj`` / ``decoder_proj`` / ``output_linear`` live under ``joiner.``
  in the state dict.

No real model file exists in this environment (no network), so the mapping is exercised on files written by
``tests/onnx_writer.py`` with the same naming conventions; ``import_onnx`` returns a report of unmapped initializers and of
tensors the engine needs but did not find, so that a first run on a real export says exactly what is missing.
Quantised (``MatMulInteger`` / ``DynamicQuantizeLinear``) models are dequantised when scale / zero-point initializers are
present next to the int8 weight; otherwise they are listed as unmapped.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from .k2w import write_k2w

_ONNX_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


# ----------------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: memoryview):
    """Yield (field_number, wire_type, value) for one message; value is int (varint / fixed) or memoryview (bytes)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos : pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _packed_varints(v) -> List[int]:
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    """TensorProto: dims=1, data_type=2, float_data=4, int32_data=5, int64_data=7, name=8, raw_data=9."""
    dims: List[int] = []
    dtype, name, raw = 1, "", None
    floats: List[float] = []
    ints: List[int] = []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif fn == 2:
            dtype = v
        elif fn == 4:
            floats += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fn in (5, 7):
            ints += _packed_varints(v) if wt == 2 else [v]
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = v
    if dtype not in _ONNX_DTYPES:
        raise ValueError(f"initializer {name}: ONNX data_type {dtype} not supported")
    np_dt = _ONNX_DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dt).newbyteorder("<")).astype(np_dt)
    elif floats:
        arr = np.asarray(floats, np_dt)
    else:
        arr = np.asarray([x - (1 << 64) if x >= 1 << 63 else x for x in ints]).astype(np_dt)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def read_onnx(path: str):
    """-> (metadata {str: str}, initializers {name: ndarray}, nodes [(name, op_type, inputs, outputs)])"""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    meta: Dict[str, str] = {}
    inits: Dict[str, np.ndarray] = {}
    nodes = []
    for fn, wt, v in _fields(buf):          # ModelProto: graph = 7, metadata_props = 14
        if fn == 14 and wt == 2:
            k = val = ""
            for f2, _, v2 in _fields(v):    # StringStringEntryProto: key = 1, value = 2
                if f2 == 1:
                    k = bytes(v2).decode()
                elif f2 == 2:
                    val = bytes(v2).decode()
            meta[k] = val
        elif fn == 7 and wt == 2:
            for f2, w2, v2 in _fields(v):   # GraphProto: node = 1, initializer = 5
                if f2 == 5 and w2 == 2:
                    name, arr = _tensor(v2)
                    inits[name] = arr
                elif f2 == 1 and w2 == 2:
                    ins, outs, nname, op = [], [], "", ""
                    for f3, _, v3 in _fields(v2):  # NodeProto: input = 1, output = 2, name = 3, op_type = 4
                        if f3 == 1:
                            ins.append(bytes(v3).decode())
                        elif f3 == 2:
                            outs.append(bytes(v3).decode())
                        elif f3 == 3:
                            nname = bytes(v3).decode()
                        elif f3 == 4:
                            op = bytes(v3).decode()
                    nodes.append((nname, op, ins, outs))
    return meta, inits, nodes


# ----------------------------------------------------------------------------------------------- name mapping
_WRAPPER_RENAMES = (("encoder_proj.", "joiner.encoder_proj."), ("decoder_proj.", "joiner.decoder_proj."),
                    ("output_linear.", "joiner.output_linear."), ("joiner.joiner.", "joiner."))


def _canonical(name: str) -> str:
    for a, b in _WRAPPER_RENAMES:
        if name.startswith(a):
            return b + name[len(a):]
    return name


def _scope_to_module(node_name: str) -> str:
    """'/encoder/encoders.0/layers.1/feed_forward1/in_proj/MatMul' -> 'encoder.encoders.0.layers.1.feed_forward1.in_proj'"""
    parts = [p for p in node_name.split("/") if p]
    return ".".join(parts[:-1])


def map_initializers(inits: Dict[str, np.ndarray], nodes) -> Tuple[Dict[str, np.ndarray], List[str]]:
    out: Dict[str, np.ndarray] = {}
    unmapped: List[str] = []
    consumers: Dict[str, Tuple[str, str, int]] = {}
    for nname, op, ins, _ in nodes:
        for pos, i in enumerate(ins):
            consumers.setdefault(i, (nname, op, pos))
    # onnxruntime's dynamic quantisation (the reference's *.int8.onnx model zoo, README.EN.md:8-35): a Linear becomes
    # DynamicQuantizeLinear -> MatMulInteger(A_q, W_quantized, a_zp, W_zero_point) -> Cast -> Mul(a_scale * W_scale); the weight is
    # stored as <w>_quantized (int8 / uint8, [in, out]) with <w>_scale and <w>_zero_point.  The weights are dequantised here,
    # W = (W_q - zp) * scale, and the engine computes in f32 -- the dynamic quantisation of the ACTIVATIONS is not reproduced
    # (int8 graphs are outside the parity scope).
    handled = set()
    for name, arr in inits.items():
        if not name.endswith("_quantized") or arr.dtype not in (np.int8, np.uint8):
            continue
        stem = name[: -len("_quantized")]
        scale = inits.get(stem + "_scale")
        if scale is None:
            continue
        zp = inits.get(stem + "_zero_point")
        c = consumers.get(name)
        if c is None or not c[0]:
            unmapped.append(name)
            continue
        nname, op, pos = c
        sc = np.asarray(scale, np.float32)
        z = np.asarray(zp if zp is not None else 0).astype(np.int32)
        if op == "ConvInteger" and arr.ndim > 1 and sc.ndim == 1 and sc.size == arr.shape[0]:
            shp = (-1,) + (1,) * (arr.ndim - 1)          # per-output-channel scales of a conv filter [O, I, ...]
            sc, z = sc.reshape(shp), (z.reshape(shp) if z.ndim == 1 and z.size == arr.shape[0] else z)
        w = ((arr.astype(np.int32) - z) * sc).astype(np.float32)
        mod = _scope_to_module(nname)
        for suffix in ("_quant",):                        # quantize_dynamic renames the node '<scope>/MatMul' -> '<scope>/MatMul_quant'
            if mod.endswith(suffix):
                mod = mod[: -len(suffix)]
        mod = _canonical(mod + ".")[:-1]
        if op == "MatMulInteger" and w.ndim == 2:
            out[mod + ".weight"] = np.ascontiguousarray(w.T)
        elif op == "ConvInteger":
            out[mod + ".weight"] = w
        else:
            unmapped.append(name)
            continue
        handled.update({name, stem + "_scale", stem + "_zero_point"})
    for name, arr in inits.items():
        if name in handled:
            continue
        anonymous = name.startswith("onnx::") or "." not in name
        if not anonymous:
            out[_canonical(name)] = arr
            continue
        c = consumers.get(name)
        if c is None or not c[0]:
            unmapped.append(name)
            continue
        nname, op, pos = c
        mod = _canonical(_scope_to_module(nname) + ".")[:-1]
        if op == "MatMul" and arr.ndim == 2:
            out[mod + ".weight"] = np.ascontiguousarray(arr.T)      # ONNX [in, out] -> torch [out, in]
        elif op == "Gemm" and arr.ndim == 2:
            out[mod + (".weight" if pos == 1 else ".bias")] = arr
        elif op == "Conv":
            out[mod + (".weight" if pos == 1 else ".bias")] = arr
        elif op == "Add" and arr.ndim == 1:
            out[mod + ".bias"] = arr
        else:
            unmapped.append(name)
    return out, unmapped


def import_onnx(paths: List[str], out_path: str, extra_meta: Dict[str, str] | None = None, required: List[str] | None = None):
    """Merge the ONNX files of one model (encoder, decoder, joiner) into a .k2w container.

    Returns a report dict: {"tensors": n, "unmapped": [...], "missing": [...], "meta": {...}}.
    """
    meta: Dict[str, str] = {}
    tensors: Dict[str, np.ndarray] = {}
    unmapped: List[str] = []
    for p in paths:
        m, inits, nodes = read_onnx(p)
        meta.update(m)
        t, u = map_initializers(inits, nodes)
        tensors.update(t)
        unmapped += [f"{p}:{x}" for x in u]
    if extra_meta:
        meta.update(extra_meta)
    keep = {}
    for k, a in tensors.items():
        if a.dtype in (np.float16, np.float64):
            a = a.astype(np.float32)
        # the engine computes in f32 and reads nothing else: integer initializers of an export (int64 shape / axes constants such as
        # '/encoder/encoders.0/Constant_output_0') are listed, not stored
        if a.dtype == np.float32:
            keep[k] = a if a.ndim > 0 else a.reshape(1)
        else:
            unmapped.append(k + f" (dtype {a.dtype})")
    missing = [r for r in (required or []) if r not in keep]
    write_k2w(out_path, meta, keep.items())
    return {"tensors": len(keep), "unmapped": unmapped, "missing": missing, "meta": meta}
