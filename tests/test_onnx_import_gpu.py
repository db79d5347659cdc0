"""SURVEY 8f N1 through the GPU: the reference loads three .onnx files per model (OfflineModel.cs:84-118, OnlineModel.cs:224-228).
Here a model is written as that ONNX triple (tests/onnx_writer.py: the export's naming conventions), imported to a .k2w container,
loaded by k2hip_model_create and DECODED on the device; tokens must equal the oracle's on the ORIGINAL container.  The same with
the triple as onnxruntime's quantize_dynamic leaves it (*.int8.onnx): the importer dequantises, the engine computes in f32."""
import numpy as np
import pytest

import onnx_writer as ow
from parity import LOGIT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


def _import(tmp_path, preset, int8=False):
    from k2transducerasr_amd.k2w import read_k2w
    from k2transducerasr_amd.onnx_import import import_onnx
    from k2transducerasr_amd.synth import tensor_specs, write_synthetic_model
    src = str(tmp_path / "src.k2w")
    meta = write_synthetic_model(src, preset)
    meta0, tensors0 = read_k2w(src)
    dst = str(tmp_path / ("dst.int8.k2w" if int8 else "dst.k2w"))
    rep = import_onnx(ow.export_triple(meta0, tensors0, tmp_path, int8=int8), dst, required=[n for n, _, _ in tensor_specs(meta)])
    assert rep["unmapped"] == [] and rep["missing"] == []
    return src, dst


@pytest.mark.parametrize("preset", ["zipformer2-tiny-test", "conformer-tiny-test"])
def test_imported_offline_model_decodes_on_the_gpu(tmp_path, utts, preset):
    from k2transducerasr_amd import Model
    from oracle import Oracle
    src, dst = _import(tmp_path, preset)
    ora, hip = Oracle(src), Model(dst, 0)
    feats = [ora.fbank(u) for u in utts]
    want = ora.recognize_batch(feats)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(hip.offline_greedy_from_samples(utts), want, what=f"{preset} imported from ONNX")
    x = ora.pad_sequence(feats).reshape(len(utts), -1, 80)
    np.testing.assert_allclose(hip.encoder_proj(x), ora.encoder(x), atol=2e-4, rtol=0)
    hip.close()


def test_imported_streaming_model_decodes_on_the_gpu(tmp_path):
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance
    from oracle.online import OnlineOracle
    src, dst = _import(tmp_path, "zipformer2-streaming-tiny-test")
    rec, ora = OnlineRecognizer(dst), OnlineOracle(src)
    T, S = rec.chunk_length, rec.shift_length
    feats = [ora.fbank(synth_utterance(60 + u, 1.9)) for u in range(3)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    for k in range((feats[0].shape[0] - T) // S + 1):
        rec.get_results(hs)
        ora.step(os_, [f[k * S : k * S + T] for f in feats])
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps and h.hyp == o.hyp, k
    import parity
    parity.COMPARED[0] += len(hs)
    assert sum(len(h.tokens) - 2 for h in hs) > 0


@pytest.mark.parametrize("preset", ["zipformer2-tiny-test", "conformer-tiny-test"])
def test_imported_int8_model_decodes_on_the_gpu(tmp_path, utts, preset):
    """quantize_dynamic form: every Linear weight is uint8 + scale + zero point.  The importer hands the engine W' = (W_q - zp) * scale
    (|W' - W| <= scale / 2 per element, checked), the engine computes in f32: on the SAME dequantised container GPU and oracle must
    agree as always (tokens exact, logits 1e-3); against the f32 original the encoder output moves by the quantisation noise only."""
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.k2w import read_k2w
    from oracle import Oracle
    src, dst = _import(tmp_path, preset, int8=True)
    _, t0 = read_k2w(src)
    _, t1 = read_k2w(dst)
    assert set(t0) == set(t1)
    nq = 0
    for k, w in t0.items():
        if k.endswith(".weight") and w.ndim == 2 and "embedding" not in k:
            scale = (max(float(w.max()), 0.0) - min(float(w.min()), 0.0)) / 255.0
            assert float(np.abs(t1[k] - w).max()) <= 0.5 * scale * (1 + 1e-5) + 1e-12, k
            assert not np.array_equal(t1[k], w)
            nq += 1
        else:
            assert np.array_equal(t1[k], w), k
    assert nq > 10
    ora_q, ora_f, hip = Oracle(dst), Oracle(src), Model(dst, 0)
    feats = [ora_q.fbank(u) for u in utts]
    x = ora_q.pad_sequence(feats).reshape(len(utts), -1, 80)
    enc_q, enc_h = ora_q.encoder(x), hip.encoder_proj(x)
    np.testing.assert_allclose(enc_h, enc_q, atol=2e-4, rtol=0)
    dec = ora_q.decoder(np.array([[-1, 0]], np.int64))
    lo = ora_q.joiner(enc_q[0], np.repeat(dec, enc_q.shape[1], 0))
    lh = hip.joiner_proj(enc_h[0], np.repeat(dec, enc_q.shape[1], 0))
    assert float(np.abs(lo - lh).max()) < LOGIT_TOL
    assert_tokens_match(hip.offline_greedy_from_samples(utts), ora_q.recognize_batch(feats), what=f"{preset} int8 import")
    # against the f32 original: the dequantisation error of the weights (relative ~1/255 of each tensor's range) carried through the
    # network -- a loose sanity bound, not a parity claim (int8 arithmetic is outside the parity scope)
    enc_f = ora_f.encoder(x)
    rel = float(np.abs(enc_h - enc_f).max()) / max(float(np.abs(enc_f).max()), 1e-6)
    assert rel < 0.05, rel
    hip.close()
