"""ONNX -> .k2w importer: a synthetic model is written as three ONNX files with the export's naming conventions
(named parameters, anonymous transposed MatMul weights under scoped node names, wrapper prefixes), imported back, and the
result must be the SAME model: identical tensors, identical metadata, identical oracle output."""
import numpy as np
import pytest

import onnx_writer as ow
from k2transducerasr_amd.k2w import read_k2w
from k2transducerasr_amd.onnx_import import import_onnx, read_onnx
from k2transducerasr_amd.synth import tensor_specs


_export = ow.export_triple


@pytest.mark.parametrize("preset", ["zipformer2-tiny-test", "conformer-tiny-test", "zipformer-tiny-test", "zipformer-streaming-tiny-test",
                                    "lstm-tiny-test", "zipformer2-ctc-tiny-test"])
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


def test_int8_dynamic_quantised_weights_are_dequantised(tmp_path):
    """A Linear as onnxruntime's quantize_dynamic leaves it (MatMulInteger on <w>_quantized with <w>_scale / <w>_zero_point,
    README.EN.md:8-35 lists the *.int8.onnx model zoo): the importer must hand the engine W = (W_q - zp) * scale in torch layout,
    and drop the scale / zero-point helpers."""
    rng = np.random.default_rng(3)
    wq = rng.integers(0, 256, (48, 24), dtype=np.uint8)                 # [in, out]
    scale, zp = np.float32(0.0123), np.uint8(131)
    sq = rng.integers(-128, 128, (16, 48), dtype=np.int8)
    s_scale = np.float32(0.5)
    inits = [ow.tensor("onnx::MatMul_77_quantized", wq), ow.tensor("onnx::MatMul_77_scale", np.asarray(scale).reshape(())),
             ow.tensor("onnx::MatMul_77_zero_point", np.asarray(zp).reshape(())),
             ow.tensor("onnx::MatMul_78_quantized", sq), ow.tensor("onnx::MatMul_78_scale", np.asarray(s_scale).reshape(())),
             ow.tensor("encoder_proj.bias", np.ones(24, np.float32))]
    nodes = [ow.node("/encoder_proj/MatMul_quant", "MatMulInteger", ["a_q", "onnx::MatMul_77_quantized", "a_zp", "onnx::MatMul_77_zero_point"], ["y"]),
             ow.node("/encoder/encoders.0/layers.1/feed_forward1/in_proj/MatMul_quant", "MatMulInteger", ["b_q", "onnx::MatMul_78_quantized", "b_zp"], ["z"])]
    p = tmp_path / "enc.int8.onnx"
    p.write_bytes(ow.model({"model_type": "zipformer2"}, inits, nodes))
    rep = import_onnx([str(p)], str(tmp_path / "m.k2w"))
    assert rep["unmapped"] == []
    _, t = read_k2w(str(tmp_path / "m.k2w"))
    want = ((wq.astype(np.int32) - 131) * scale).astype(np.float32).T
    np.testing.assert_array_equal(t["joiner.encoder_proj.weight"], want)
    np.testing.assert_array_equal(t["encoder.encoders.0.layers.1.feed_forward1.in_proj.weight"], (sq.astype(np.float32) * 0.5).T)   # no zero point
    assert set(t) == {"joiner.encoder_proj.weight", "joiner.encoder_proj.bias", "encoder.encoders.0.layers.1.feed_forward1.in_proj.weight"}


def test_int64_initializers_are_listed_not_stored(tmp_path):
    """An export carries int64 shape / axes constants beside the weights; one whose name looks like a state-dict entry used to be
    written into the container, which the engine's loader then rejected as a whole.  They are reported in `unmapped` and the
    container holds f32 tensors only."""
    w = np.ones((4, 3), np.float32)
    p = tmp_path / "m.onnx"
    p.write_bytes(ow.model({"model_type": "zipformer2"},
                           [ow.tensor("onnx::MatMul_7", w), ow.tensor("/encoder/encoders.0/Constant_output_0", np.array([1, -1], np.int64)),
                            ow.tensor("encoder.encoders.0.downsample.index", np.arange(3, dtype=np.int64), raw=False)],
                           [ow.node("/encoder_proj/MatMul", "MatMul", ["x", "onnx::MatMul_7"], ["y"])]))
    rep = import_onnx([str(p)], str(tmp_path / "o.k2w"))
    _, t = read_k2w(str(tmp_path / "o.k2w"))
    assert set(t) == {"joiner.encoder_proj.weight"} and all(a.dtype == np.float32 for a in t.values())
    assert any("Constant_output_0" in u for u in rep["unmapped"]) and any("downsample.index" in u and "int64" in u for u in rep["unmapped"])


@pytest.mark.parametrize("preset", ["zipformer2-tiny-test", "zipformer2-streaming-tiny-test"])
def test_int8_triple_of_a_whole_model_imports(tmp_path, preset):
    """The whole model as quantize_dynamic leaves it: every Linear comes back within half a quantisation step of the original,
    everything else bit-identical, nothing unmapped or missing (the GPU leg of this is tests/test_onnx_import_gpu.py)."""
    from k2transducerasr_amd.synth import write_synthetic_model
    src = str(tmp_path / "src.k2w")
    meta = write_synthetic_model(src, preset)
    meta0, t0 = read_k2w(src)
    dst = str(tmp_path / "dst.k2w")
    rep = import_onnx(ow.export_triple(meta0, t0, tmp_path, int8=True), dst, required=[n for n, _, _ in tensor_specs(meta)])
    assert rep["unmapped"] == [] and rep["missing"] == []
    meta1, t1 = read_k2w(dst)
    assert meta1 == meta0 and set(t1) == set(t0)
    for k, w in t0.items():
        if k.endswith(".weight") and w.ndim == 2 and "embedding" not in k:
            scale = (max(float(w.max()), 0.0) - min(float(w.min()), 0.0)) / 255.0
            assert float(np.abs(t1[k] - w).max()) <= 0.5 * scale * (1 + 1e-5) + 1e-12, k
        else:
            assert np.array_equal(t1[k], w), k
