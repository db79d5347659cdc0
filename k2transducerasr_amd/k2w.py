"""``.k2w`` weight container: metadata map + named f32/i64 tensors.

The reference loads ONNX files and reads its configuration from their
custom-metadata maps (K2TransducerAsr/OfflineModel.cs:31-72,
K2TransducerAsr/OnlineModel.cs:32-184).  ``.k2w`` keeps the same string->string
map and stores every initializer under its icefall state-dict name, in the
native torch layout (Linear ``[out, in]``, Conv ``[out, in/groups, kh, kw]``),
so an ONNX-initializer importer is a pure rename.

Layout (little endian)::

    char[4]  magic = "K2W1"
    u32      version = 1
    u32      n_meta
    u32      n_tensors
    u64      data_offset           # absolute, 64-byte aligned
    n_meta   x { u32 klen; u32 vlen; char key[klen]; char val[vlen] }
    n_tensors x { u32 namelen; char name[namelen]; u32 dtype (0=f32, 1=i64);
                  u32 ndim; u64 dims[4]; u64 offset (relative to data_offset);
                  u64 nbytes }
    data region, every tensor 64-byte aligned
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Tuple

import numpy as np

MAGIC = b"K2W1"
_DTYPES = {0: np.float32, 1: np.int64}
_DTYPE_CODES = {np.dtype(np.float32): 0, np.dtype(np.int64): 1}


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def write_k2w(path: str, meta: Dict[str, str], tensors: Iterable[Tuple[str, np.ndarray]]) -> None:
    tensors = [(n, np.ascontiguousarray(a)) for n, a in tensors]
    head = bytearray()
    for k, v in meta.items():
        kb, vb = k.encode(), str(v).encode()
        head += struct.pack("<II", len(kb), len(vb)) + kb + vb
    entries = []
    off = 0
    for name, arr in tensors:
        if arr.dtype not in _DTYPE_CODES:
            raise TypeError(f"{name}: unsupported dtype {arr.dtype}")
        if arr.ndim > 4:
            raise ValueError(f"{name}: ndim {arr.ndim} > 4")
        dims = list(arr.shape) + [1] * (4 - arr.ndim)
        nb = arr.nbytes
        entries.append((name.encode(), _DTYPE_CODES[arr.dtype], arr.ndim, dims, off, nb))
        off = _align(off + nb)
    table = bytearray()
    for nb_, code, ndim, dims, o, nbytes in entries:
        table += struct.pack("<I", len(nb_)) + nb_
        table += struct.pack("<II4QQQ", code, ndim, *dims, o, nbytes)
    fixed = 4 + 4 + 4 + 4 + 8
    data_offset = _align(fixed + len(head) + len(table))
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<IIIQ", 1, len(meta), len(tensors), data_offset))
        f.write(head)
        f.write(table)
        f.write(b"\0" * (data_offset - fixed - len(head) - len(table)))
        pos = 0
        for (name, arr), (_, _, _, _, o, nbytes) in zip(tensors, entries):
            if o != pos:
                f.write(b"\0" * (o - pos))
                pos = o
            f.write(arr.tobytes())
            pos += nbytes


def read_k2w(path: str):
    """Return (meta, {name: ndarray}).  Arrays are memory-mapped views."""
    with open(path, "rb") as f:
        hdr = f.read(24)
        if hdr[:4] != MAGIC:
            raise ValueError(f"{path}: not a K2W1 file")
        version, n_meta, n_tensors, data_offset = struct.unpack("<IIIQ", hdr[4:])
        if version != 1:
            raise ValueError(f"{path}: unsupported version {version}")
        rest = f.read(data_offset - 24)
    p = 0
    meta = {}
    for _ in range(n_meta):
        kl, vl = struct.unpack_from("<II", rest, p)
        p += 8
        k = rest[p : p + kl].decode()
        p += kl
        v = rest[p : p + vl].decode()
        p += vl
        meta[k] = v
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    out = {}
    for _ in range(n_tensors):
        (nl,) = struct.unpack_from("<I", rest, p)
        p += 4
        name = rest[p : p + nl].decode()
        p += nl
        code, ndim, d0, d1, d2, d3, off, nbytes = struct.unpack_from("<II4QQQ", rest, p)
        p += 8 + 32 + 16
        shape = (d0, d1, d2, d3)[:ndim]
        a = mm[data_offset + off : data_offset + off + nbytes].view(_DTYPES[code]).reshape(shape)
        out[name] = a
    return meta, out
