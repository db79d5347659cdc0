"""ONNX -> .k2w importer: a synthetic model is written as three ONNX files with the export's naming conventions
(named parameters, anonymous transposed MatMul weights under scoped node names, wrapper prefixes), imported back, and the
result must be the SAME model: identical tensors, identical metadata, identical oracle output."""
import numpy as np
import pytest

import onnx_writer as ow
from k2transducerasr_amd.k2w import read_k2w
from k2transducerasr_amd.onnx_import import import_onnx, read_onnx
from k2transducerasr_amd.synth import tensor_specs


def _export(meta, tensors, tmp_path):
    """split like icefall's export-onnx.py: encoder (+ encoder_proj), decoder (+ decoder_proj), joiner (output_linear)"""
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
            # Linear -> MatMul with an anonymous, transposed initializer
            anon += 1
            iname = f"onnx::MatMul_{1000 + anon}"
            inits.append(ow.tensor(iname, np.ascontiguousarray(arr.T), raw=(anon % 2 == 0)))
            scope = "/" + "/".join(_scope_parts(local[: -len(".weight")]))
            nodes.append(ow.node(scope + "/MatMul", "MatMul", [f"x{anon}", iname], [f"y{anon}"]))
        else:
            inits.append(ow.tensor(local, arr))
    paths = []
    for f, (inits, nodes) in files.items():
        p = tmp_path / f"{f}.onnx"
        md = meta if f == "encoder" else {k: meta[k] for k in ("context_size", "vocab_size", "joiner_dim") if k in meta}
        p.write_bytes(ow.model(md, inits, nodes))
        paths.append(str(p))
    return paths


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


@pytest.mark.parametrize("preset", ["zipformer2-tiny-test", "conformer-tiny-test"])
def test_round_trip_through_onnx(tmp_path, preset):
    from k2transducerasr_amd.synth import write_synthetic_model
    src = str(tmp_path / "src.k2w")
    meta = write_synthetic_model(src, preset)
    meta0, tensors0 = read_k2w(src)
    paths = _export(meta0, tensors0, tmp_path)
    m, inits, nodes = read_onnx(paths[0])
    assert m["model_type"] == meta["model_type"] and any(k.startswith("onnx::MatMul_") for k in inits)
    dst = str(tmp_path / "dst.k2w")
    rep = import_onnx(paths, dst, required=[n for n, _, _ in tensor_specs(meta)])
    assert rep["unmapped"] == [] and rep["missing"] == []
    meta1, tensors1 = read_k2w(dst)
    assert meta1 == meta0
    assert set(tensors1) == set(tensors0)
    for k in tensors0:
        assert tensors1[k].shape == tensors0[k].shape and np.array_equal(tensors1[k], tensors0[k]), k


def test_imported_model_gives_identical_oracle_output(tmp_path, utts):
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle import Oracle
    src = str(tmp_path / "src.k2w")
    write_synthetic_model(src, "zipformer2-tiny-test")
    meta0, tensors0 = read_k2w(src)
    dst = str(tmp_path / "dst.k2w")
    import_onnx(_export(meta0, tensors0, tmp_path), dst)
    a, b = Oracle(src), Oracle(dst)
    f = [a.fbank(u) for u in utts[:2]]
    assert a.recognize_batch(f) == b.recognize_batch(f)


def test_report_lists_what_is_missing(tmp_path):
    w = np.ones((4, 3), np.float32)
    p = tmp_path / "m.onnx"
    p.write_bytes(ow.model({"model_type": "zipformer2"}, [ow.tensor("onnx::MatMul_7", w), ow.tensor("onnx::Mul_9", np.ones(3, np.float32))],
                           [ow.node("/encoder_proj/MatMul", "MatMul", ["x", "onnx::MatMul_7"], ["y"])]))
    rep = import_onnx([str(p)], str(tmp_path / "o.k2w"), required=["joiner.encoder_proj.weight", "joiner.encoder_proj.bias"])
    assert rep["missing"] == ["joiner.encoder_proj.bias"]
    assert rep["unmapped"] == [f"{p}:onnx::Mul_9"]
    _, t = read_k2w(str(tmp_path / "o.k2w"))
    assert t["joiner.encoder_proj.weight"].shape == (3, 4)
