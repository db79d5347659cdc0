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


def _scope_parts(mod: str):
    """module path -> scope components as torch names them: child modules of a ModuleList keep 'name.index' together"""
    parts, out = mod.split("."), []
    i = 0
    while i < len(parts):
        if i + 1 < len(parts) and parts[i + 1].isdigit():
            out.append(parts[i] + "." + parts[i + 1])
            i += 2
        else:
            out.append(parts[i])
            i += 1
    return out


def quantise_u8(w: np.ndarray):
    """per-tensor asymmetric uint8 quantisation as onnxruntime's quantize_dynamic does it for a MatMul weight:
    scale = (max - min) / 255 over a range that contains 0, zero_point = round(-min / scale)."""
    lo, hi = min(float(w.min()), 0.0), max(float(w.max()), 0.0)
    scale = np.float32((hi - lo) / 255.0) if hi > lo else np.float32(1.0)
    zp = np.uint8(np.clip(np.round(-lo / scale), 0, 255))
    q = np.clip(np.round(w / scale) + int(zp), 0, 255).astype(np.uint8)
    return q, scale, zp


def export_triple(meta, tensors, tmp_path, int8: bool = False):
    """Split a state dict like icefall's export-onnx.py: encoder (+ encoder_proj), decoder (+ decoder_proj), joiner
    (output_linear).  Every Linear becomes a MatMul with an anonymous, transposed initializer under its scoped node name; with
    `int8` that MatMul is what quantize_dynamic leaves (MatMulInteger on <w>_quantized + <w>_scale + <w>_zero_point, node renamed
    '<scope>/MatMul_quant'), as in the reference's *.int8.onnx model zoo (README.EN.md:8-35)."""
    files = {"encoder": ([], []), "decoder": ([], []), "joiner": ([], [])}
    anon = 0
    for name, arr in tensors.items():
        arr = np.asarray(arr)
        if name.startswith("joiner.encoder_proj."):
            f, local = "encoder", name[len("joiner."):]
        elif name.startswith("joiner.decoder_proj."):
            f, local = "decoder", name[len("joiner."):]
        elif name.startswith("joiner.output_linear."):
            f, local = "joiner", name[len("joiner."):]
        elif name.startswith("decoder."):
            f, local = "decoder", name
        else:
            f, local = "encoder", name
        inits, nodes = files[f]
        if local.endswith(".weight") and arr.ndim == 2 and "embedding" not in local:
            anon += 1
            iname = f"onnx::MatMul_{1000 + anon}"
            scope = "/" + "/".join(_scope_parts(local[: -len(".weight")]))
            if int8:
                q, scale, zp = quantise_u8(np.ascontiguousarray(arr.T))
                inits += [tensor(iname + "_quantized", q), tensor(iname + "_scale", np.asarray(scale).reshape(())),
                          tensor(iname + "_zero_point", np.asarray(zp).reshape(()))]
                nodes.append(node(scope + "/MatMul_quant", "MatMulInteger", [f"x{anon}_q", iname + "_quantized", f"x{anon}_zp", iname + "_zero_point"], [f"y{anon}"]))
            else:
                inits.append(tensor(iname, np.ascontiguousarray(arr.T), raw=(anon % 2 == 0)))
                nodes.append(node(scope + "/MatMul", "MatMul", [f"x{anon}", iname], [f"y{anon}"]))
        else:
            inits.append(tensor(local, arr))
    paths = []
    for f, (inits, nodes) in files.items():
        p = tmp_path / f"{f}{'.int8' if int8 else ''}.onnx"
        md = meta if f == "encoder" else {k: meta[k] for k in ("context_size", "vocab_size", "joiner_dim") if k in meta}
        p.write_bytes(model(md, inits, nodes))
        paths.append(str(p))
    return paths
