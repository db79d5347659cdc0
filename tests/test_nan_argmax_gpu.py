"""The reference's argmax on rows that contain NaNs, through the HIP search kernels.

OfflineRecognizer.cs:150-154 / :236-240 (OnlineRecognizer.cs:159-163) scan a row with
    token_num = logits[j, token_num] > logits[j, k] ? token_num : k
A comparison with a NaN is false, so the scan jumps ONTO a NaN and OFF it again at the next index: the result is the later-wins
argmax over the indexes behind the row's LAST NaN, or that NaN's index if it ends the row ([5, NaN, 1] -> 2, [0, 1, NaN] -> 2,
all NaN -> V - 1; tests/test_oracle_kat.py::test_argmax_ref holds the oracle to these by hand).  A NaN logit needs no broken model:
one NaN sample in the audio makes a whole frame's logits NaN.  Every form of the search is held to the rule here: k_greedy (one
pass, several passes, several column slabs, with and without the decoder table), the rounds form, the t0 pass of the batch loop,
the single-stream loop."""
import numpy as np
import pytest

from kat_model import J as KJ
from kat_model import V as KV
from kat_model import frames, write_kat_model, write_wide_model

pytestmark = pytest.mark.gpu
NAN = float("nan")


def _pair(path):
    from k2transducerasr_amd import Model
    from oracle import Oracle
    return Model(path, 0), Oracle(path)


def _forms(hip, enc, single=True):
    """the same enc_out through every form of the greedy search"""
    from k2transducerasr_amd import set_switch
    out = {"batch": hip.greedy_batch(enc)}
    if single:
        out["single"] = [hip.greedy_single(e) for e in enc]
    set_switch("K2HIP_SEARCH_ROUNDS", 1)
    try:
        out["rounds"] = hip.greedy_batch(enc)
    finally:
        set_switch("K2HIP_SEARCH_ROUNDS", -1)
    return out


@pytest.mark.parametrize("nan_at,rows,want", [
    # NaN at 1: a high blank logit in FRONT of the NaN is dead; behind it 5 wins -> emitted.  Without the rule: blank, nothing.
    ((1,), [{0: 2.0, 5: 0.3}, {0: 2.0}, {0: 2.0, 2: 0.5}], ([5, 3], [0, 1])),
    #   frame 1 (context [0, 5]: token 3 carries +0.55): behind the NaN 3 = tanh(-2.45) beats the tanh(-3) of the others -> 3.
    #   frame 2 (context [5, 3]: +0.8): 2 = unk = tanh(0.5) wins behind the NaN -> skipped
    # NaN at 0: the scan leaves index 0 at once; 3 vs 4 decided normally
    ((0,), [{3: 0.2, 4: 0.1}, {0: 5.0, 6: -1.0}], ([3, 6], [0, 1])),
    # NaN at the LAST index: every frame ends on it -> token 7 on every frame, whatever else the row holds
    ((7,), [{0: 3.0}, {5: 3.0}, {2: 3.0}], ([7, 7, 7], [0, 1, 2])),
    # two NaNs: only what is behind the LAST one counts
    ((1, 4), [{0: 2.0, 3: 2.0, 6: 0.1}, {5: 0.1, 6: 0.1}], ([6, 6], [0, 1])),
])
def test_nan_logit_known_answers(tmp_path, nan_at, rows, want):
    bias = np.zeros(KV, np.float32)
    bias[list(nan_at)] = NAN
    p = str(tmp_path / "kat_nan.k2w")
    write_kat_model(p, bias=bias)
    hip, ora = _pair(p)
    enc = frames(rows)[None]
    assert ora.greedy_batch(enc) == [want], "the hand-derived answer and the oracle disagree"
    for form, got in _forms(hip, enc).items():
        assert got == ([want] if form != "single" else [want]), (form, got)
    hip.close()


def test_all_nan_frame_emits_the_last_token(tmp_path):
    """one NaN in a frame of encoder_out (a NaN sample upstream): tanh(NaN + dec) poisons every logit of that frame -> V - 1"""
    p = str(tmp_path / "kat.k2w")
    write_kat_model(p)
    hip, ora = _pair(p)
    enc = np.stack([frames([{0: 1.0}, {5: 1.0}, {0: 1.0}, {0: 1.0}]), frames([{0: 1.0}, {0: 1.0}, {4: 1.0}, {0: 1.0}])])
    enc[0, 2, 6] = NAN
    enc[1, 0, 0] = NAN
    want = [([5, KV - 1], [1, 2]), ([KV - 1, 4], [0, 2])]
    assert ora.greedy_batch(enc) == want
    for form, got in _forms(hip, enc, single=False).items():
        assert got == want, (form, got)
    assert hip.greedy_single(enc[0]) == ora.greedy_single(enc[0]) == want[0]
    hip.close()


@pytest.mark.parametrize("table_mb", [1024, 0])
@pytest.mark.parametrize("nan_at", [(), (5,), (255,), (256,), (607, 608), (1000,), (1199,), (3, 700, 1198)])
def test_wide_vocabulary_slabs_and_passes(tmp_path, nan_at, table_mb):
    """V = 1200: 300 column groups -- one part walks them in 5 passes, two parts split at column 608, four at 304 / 608 / 912; the NaNs
    sit on pass and slab boundaries.  Every form and split must give the oracle's tokens (the rule is exact, not approximate)."""
    from k2transducerasr_amd import set_switch
    p = str(tmp_path / "wide.k2w")
    write_wide_model(p, 1200, nan_at=nan_at)
    set_switch("K2HIP_DECODER_TABLE_MB", table_mb)
    try:
        hip, ora = _pair(p)
        rng = np.random.default_rng(11)
        enc = rng.standard_normal((3, 14, 64)).astype(np.float32)
        want = ora.greedy_batch(enc)
        if not nan_at:
            assert 0 < sum(len(t) for t, _ in want) < 3 * 14      # the model emits on some frames and not on others
        for parts in (1, 2, 4):
            set_switch("K2HIP_GREEDY_PARTS", parts)
            try:
                got = hip.greedy_batch(enc)
            finally:
                set_switch("K2HIP_GREEDY_PARTS", 0)
            assert got == want, (nan_at, parts, got, want)
        set_switch("K2HIP_SEARCH_ROUNDS", 1)
        try:
            assert hip.greedy_batch(enc) == want, (nan_at, "rounds")
        finally:
            set_switch("K2HIP_SEARCH_ROUNDS", -1)
        for b in range(3):
            assert hip.greedy_single(enc[b]) == ora.greedy_single(enc[b]), (nan_at, "single", b)
        hip.close()
    finally:
        set_switch("K2HIP_DECODER_TABLE_MB", 1024)
