"""CPU tests: known-answer tests that pin the oracle's restatement of what the reference
itself defines (padding quirks, argmax tie-break, emit filters, context seeding), analytic
fbank KATs, the oracle against the independent torch twin, and against committed goldens."""
import os

import numpy as np
import pytest

from kat_model import CASES, write_kat_model

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
LOG_FLOOR = np.float32(-23.025850929940457)


# ------------------------------------------------------------------ F3 PadHelper.cs:14-60
def test_pad_sequence_known_answer(oracle_tiny):
    a = np.array([1.0, 0.0, 2.0], np.float32)  # a genuine zero feature
    b = np.array([3.0], np.float32)
    out = oracle_tiny.pad_sequence([a, b])
    assert out.shape == (2, 3 + 80 * 19)  # Q1: 80*19 floats whatever featureDim is (:17,:22)
    np.testing.assert_array_equal(out[0, :3], [1.0, LOG_FLOOR, 2.0])  # Q2 (:58)
    np.testing.assert_array_equal(out[1, :1], [3.0])
    assert (out[0, 3:] == LOG_FLOOR).all() and (out[1, 1:] == LOG_FLOOR).all()
    assert oracle_tiny.pad_sequence([a], tail_frames=0).shape == (1, 3)  # online overload (:9-13)


# ------------------------------------------------------------------ Q6 argmax
@pytest.mark.parametrize("logits,want", [
    ([1, 3, 3, 2], 2), ([5, 5, 5, 5], 3), ([9, 1, 2], 0), ([0, 1, 2, 3], 3), ([7], 0),
    ([0, np.nan, 1], 2), ([np.nan, 0, 0], 2), ([0, 1, np.nan], 2), ([-np.inf, -np.inf], 1),
])
def test_argmax_ref(oracle_tiny, logits, want):
    # token_num = logits[token_num] > logits[k] ? token_num : k   (OfflineRecognizer.cs:153)
    assert oracle_tiny.argmax_ref(np.array(logits, np.float32)) == want


# ------------------------------------------------------------------ F7 greedy KATs
@pytest.fixture(scope="module")
def kat_oracle(tmp_path_factory):
    from oracle import Oracle
    p = str(tmp_path_factory.mktemp("kat") / "kat.k2w")
    write_kat_model(p)
    return Oracle(p)


def test_kat_decoder_by_hand(kat_oracle):
    d = kat_oracle.decoder(np.array([[-1, 0], [0, 0], [0, 4], [4, 3], [-1, -1]], np.int64))
    np.testing.assert_allclose(d[:, 3], [0.05, 0.1, 0.45, 0.7, 0.0], rtol=1e-6)  # Q9: id < 0 -> zero embedding
    assert (np.delete(d, 3, axis=1) == 0).all()


@pytest.mark.parametrize("name", sorted(CASES))
def test_greedy_known_answers(kat_oracle, name):
    case = CASES[name]
    enc = np.stack(case["streams"])
    assert kat_oracle.greedy_batch(enc) == case["batch"]
    for b, s in enumerate(case["streams"]):
        assert kat_oracle.greedy_single(s) == case["single"][b]


def test_greedy_single_max_symbols(kat_oracle):
    # max_sym_per_utt = 1000 (OfflineRecognizer.cs:122): frame 1000 and later emit nothing
    from kat_model import frames
    enc = frames([{5: 1.0}] * 1003)
    tok, ts = kat_oracle.greedy_single(enc)
    assert len(tok) == 1000 and ts == list(range(1000))
    tok_b, ts_b = kat_oracle.greedy_batch(enc[None])[0]
    assert len(tok_b) == 1003  # the batch loop has no such cap (:216-288)


# ------------------------------------------------------------------ F1 fbank KATs
def test_fbank_frame_count_and_floor(oracle_tiny):
    for n, nf in [(0, 0), (399, 0), (400, 1), (559, 1), (560, 2), (16000, 98), (160000, 998)]:
        assert oracle_tiny.fbank(np.zeros(n, np.float32)).shape == (nf, 80)
    z = oracle_tiny.fbank(np.zeros(800, np.float32))
    np.testing.assert_allclose(z, np.log(np.float32(np.finfo(np.float32).eps)), atol=1e-6)


def test_fbank_pure_tone_peaks_in_the_right_mel_bin(oracle_tiny):
    sr, f0 = 16000, 1000.0
    t = np.arange(1600) / sr
    f = oracle_tiny.fbank((0.5 * np.sin(2 * np.pi * f0 * t)).astype(np.float32))

    def mel(x):
        return 1127.0 * np.log(1 + x / 700.0)

    centers = mel(20.0) + (np.arange(80) + 1) * (mel(8000.0) - mel(20.0)) / 81
    want = int(np.argmin(np.abs(centers - mel(f0))))
    assert (np.argmax(f, axis=1) == want).all()


def test_fbank_matches_float64_numpy_restatement(oracle_tiny, tiny_model_path, utts):
    from k2transducerasr_amd.k2w import read_k2w
    from torch_twin import fbank_np
    meta, _ = read_k2w(tiny_model_path)
    for u in utts[:2]:
        # the twin pre-processes in f64 from f64 window/mel tables; the oracle uses f32 tables
        np.testing.assert_allclose(oracle_tiny.fbank(u), fbank_np(u, meta), atol=2e-4, rtol=0)


# ------------------------------------------------------------------ oracle vs independent torch twin
@pytest.fixture(scope="module")
def twin(tiny_model_path):
    import torch
    from k2transducerasr_amd.k2w import read_k2w
    from torch_twin import Twin
    torch.set_num_threads(4)
    meta, tensors = read_k2w(tiny_model_path)
    return Twin(meta, tensors)


@pytest.mark.parametrize("tap", [0, 1, 2, 3, 4, 100, None])
def test_encoder_matches_torch_twin(oracle_tiny, twin, utts, tap):
    import torch
    x = oracle_tiny.pad_sequence([oracle_tiny.fbank(u) for u in utts[:3]]).reshape(3, -1, 80)
    if tap is None:
        a = oracle_tiny.encoder(x)
        b = twin.encoder(torch.from_numpy(x)).numpy()
    else:
        a = oracle_tiny.encoder_tap(x, tap)
        b = twin.encoder(torch.from_numpy(x), tap=tap).numpy().reshape(3, -1)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=5e-5, rtol=0)


def test_decoder_joiner_match_torch_twin(oracle_tiny, twin):
    import torch
    y = np.array([[-1, 0], [0, 0], [5, 7], [36, 1], [-1, -1]], np.int64)
    d = oracle_tiny.decoder(y)
    np.testing.assert_allclose(d, twin.decoder(torch.from_numpy(y)).numpy(), atol=1e-5, rtol=0)
    rng = np.random.default_rng(0)
    e = rng.standard_normal((5, 512)).astype(np.float32)
    np.testing.assert_allclose(oracle_tiny.joiner(e, d), twin.joiner(torch.from_numpy(e), torch.from_numpy(d)).numpy(),
                               atol=1e-5, rtol=0)


def test_shape_formulas(oracle_tiny):
    # T50 = (T-7)//2, T' = (T50+1)//2 ; C2: 998+19 = 1017 -> 505 -> 253
    assert oracle_tiny.encoder_out_frames(1017) == 253
    assert oracle_tiny.encoder_out_frames(3017) == 753
    assert oracle_tiny.encoder_out_frames(8) == 0 and oracle_tiny.encoder_out_frames(9) == 1


def test_batch_rows_are_independent_when_lengths_are_equal(oracle_tiny, utts):
    # Q3: no masking -> with equal lengths, a row's encoder_out is the same alone or in a batch
    f = [oracle_tiny.fbank(utts[0]), oracle_tiny.fbank(utts[3])]
    x = oracle_tiny.pad_sequence(f).reshape(2, -1, 80)
    np.testing.assert_array_equal(oracle_tiny.encoder(x)[1], oracle_tiny.encoder(x[1:2])[0])


def test_padding_length_changes_results(oracle_tiny, utts):
    # ... and a shorter utterance's output DOES depend on the batch maximum (log-floor padding is input)
    short = oracle_tiny.fbank(utts[1])
    alone = oracle_tiny.encoder(oracle_tiny.pad_sequence([short]).reshape(1, -1, 80))
    both = oracle_tiny.encoder(oracle_tiny.pad_sequence([short, oracle_tiny.fbank(utts[0])]).reshape(2, -1, 80))
    assert both.shape[1] > alone.shape[1]


# ------------------------------------------------------------------ committed golden vectors
def test_oracle_matches_committed_golden(oracle_tiny, tiny_model_path):
    g = np.load(os.path.join(GOLDEN, "tiny_golden.npz"))
    from k2transducerasr_amd.k2w import read_k2w
    _, tensors = read_k2w(tiny_model_path)
    # the generator is seeded: the weights the fixtures were made with must be the ones we regenerate
    assert float(np.asarray(tensors["joiner.output_linear.weight"], np.float64).sum()) == pytest.approx(float(g["w_checksum"]), abs=1e-9)
    np.testing.assert_allclose(oracle_tiny.fbank(g["samples"]), g["fbank"], atol=2e-4, rtol=0)
    x = g["x"]
    np.testing.assert_allclose(oracle_tiny.encoder(x), g["encoder_out"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(oracle_tiny.decoder(g["y"]), g["decoder_out"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(oracle_tiny.joiner(g["encoder_out"][0, :6], g["decoder_out"][:6]), g["logits"], atol=1e-5, rtol=0)
    got = oracle_tiny.greedy_batch(g["encoder_out"])
    want = [(g[f"tok{b}"].tolist(), g[f"ts{b}"].tolist()) for b in range(x.shape[0])]
    assert got == want
