"""CPU tests for the offline Conformer restatement (BASELINE.json configs[4]):
oracle/k2_oracle_conformer.c against the independent torch twin (tests/torch_twin_conformer.py),
tap by tap, plus shape / frame-count known answers."""
import numpy as np
import pytest

from k2transducerasr_amd.k2w import read_k2w


@pytest.fixture(scope="module")
def twin(conformer_tiny_path):
    from torch_twin_conformer import ConformerTwin
    meta, tensors = read_k2w(conformer_tiny_path)
    return ConformerTwin(meta, tensors)


@pytest.fixture(scope="module")
def feats(oracle_conformer, utts):
    f = [oracle_conformer.fbank(u) for u in utts[:3]]
    return oracle_conformer.pad_sequence(f).reshape(3, -1, 80)


def test_conformer_out_frames(oracle_conformer):
    # T' = ((T-1)//2 - 1)//2 (Conv2dSubsampling); C5: 2998 + 19 = 3017 -> 753
    for T, want in [(3017, 753), (1017, 253), (9, 1), (8, 1), (7, 1), (6, 0), (100, 24), (101, 24), (103, 25)]:
        assert oracle_conformer.encoder_out_frames(T) == want


@pytest.mark.parametrize("tap", [0, 1, 2, -1])
def test_conformer_oracle_matches_twin(oracle_conformer, twin, feats, tap):
    B, T, _ = feats.shape
    Tp = oracle_conformer.encoder_out_frames(T)
    if tap < 0:
        got = oracle_conformer.encoder(feats)
    else:
        got = oracle_conformer.encoder_tap(feats, tap).reshape(B, Tp, -1)
    want = twin.forward(feats, tap)
    assert got.shape == want.shape
    scale = float(np.abs(want).max())
    assert scale > 1e-2  # the activations carry signal
    np.testing.assert_allclose(got, want, atol=2e-5 * max(1.0, scale), rtol=0)


def test_conformer_batch_rows_are_independent(oracle_conformer, feats):
    # x_lens = T for every row and no masks (OfflineProjOfTransducer.cs:66-70): row b of a batch equals
    # the same utterance run alone at the same padded length
    full = oracle_conformer.encoder(feats)
    one = oracle_conformer.encoder(feats[1:2])
    np.testing.assert_allclose(full[1:2], one, atol=1e-6, rtol=0)


def test_conformer_decoder_groups_one(oracle_conformer, twin):
    # stateless2 decoder: Conv1d(groups=1); id < 0 -> zero embedding (OfflineRecognizer.cs:105)
    y = np.array([[-1, 0], [0, 0], [3, 40], [17, 5], [-1, -1]], np.int64)
    got = oracle_conformer.decoder(y)
    want = twin.decoder(y)
    np.testing.assert_allclose(got, want, atol=2e-6, rtol=0)


def test_conformer_greedy_emits(oracle_conformer, feats):
    res = oracle_conformer.recognize_batch([f.reshape(-1)[: 80 * 100] for f in feats])
    n = [len(t) for t, _ in res]
    assert sum(n) > 0 and max(n) < 30
