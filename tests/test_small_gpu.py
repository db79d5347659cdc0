"""BASELINE configs[0]: zipformer-small-en offline greedy, ONE utterance through OfflineRecognizer.GetResult -- the reference's
single-stream path (OfflineRecognizer.cs:77-83 -> ForwardGreedySearch :93-187): hypList starts [-1, blank], at most one symbol
per frame, stops at 1000 symbols, the stream keeps its samples (no RemoveSamples)."""
import numpy as np
import pytest

from parity import LOGIT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("small") / "small.k2w")
    write_synthetic_model(p, "zipformer2-small-en")
    return p


def test_small_en_single_utterance_get_result(small_path):
    from k2transducerasr_amd import OfflineRecognizer
    from k2transducerasr_amd.synth import synth_utterance
    from oracle import Oracle
    ora = Oracle(small_path)
    rec = OfflineRecognizer(small_path)                 # decodingMethod = greedy_search
    u = synth_utterance(700, 6.0)
    s = rec.create_offline_stream()
    s.add_samples(u[:40000])                            # two AddSamples calls, as a file read in pieces would make
    s.add_samples(u[40000:])
    f = ora.fbank(u)
    assert s.speech_length == f.size
    np.testing.assert_allclose(s.speech.reshape(-1, 80), f, atol=2e-5, rtol=0)
    x = ora.pad_sequence([f]).reshape(1, -1, 80)
    enc_o = ora.encoder(x)
    enc_h = rec.model.encoder_proj(x)
    np.testing.assert_allclose(enc_h, enc_o, atol=5e-4, rtol=0)
    dec = ora.decoder(np.array([[-1, 0]], np.int64))
    n = enc_o.shape[1]
    assert float(np.abs(ora.joiner(enc_o[0], np.repeat(dec, n, 0)) - rec.model.joiner_proj(enc_h[0], np.repeat(dec, n, 0))).max()) < LOGIT_TOL
    (wt, wts), mg = ora.greedy_single(enc_o[0], want_margins=True)
    assert len(wt) > 0
    tok, ts = rec.get_result(s)
    assert tok[:2] == [-1, 0]                           # OfflineRecognizer.cs:115-117
    assert_tokens_match([(tok[2:], ts)], [(wt, wts)], mg, what="small-en GetResult", batch_context=False)
    assert s.speech_length == f.size                    # the single path does not call RemoveSamples
    # the same stream through the batch entry as a batch of one (GetResults :85-91): Tokens is REPLACED by the batch list with its
    # 2*B-blank prefix (:250-258, :292), Timestamps.AddRange appends to what GetResult left (:293), samples removed (:294)
    (tok_b, ts_b), = rec.get_results([s])
    want_b = ora.recognize_batch([f])[0]
    assert tok_b == [0, 0] + want_b[0] and ts_b == ts + [0, 0] + want_b[1]
    assert s.speech_length == 0


def test_decoder_table_rows_are_the_decoder_bit_for_bit(tmp_path):
    """Small vocabularies: the engine tabulates decoder(y0, y1) for EVERY context and the search loops read a row where they would
    run the decoder.  The table (built with contexts in place of frames on the matrix pipe) must hold exactly the bits the loops'
    own routine computes, and a search with the table must return what the search without it returns."""
    import ctypes as C
    import k2transducerasr_amd as pkg
    from k2transducerasr_amd.binding import load_library
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

    L = load_library()
    L.k2hip_debug_decoder_table_check.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.k2hip_debug_decoder_table_check.restype = C.c_int32
    for preset in ("zipformer2-tiny-test", "conformer-tiny-test", "zipformer2-small-en"):
        path = str(tmp_path / f"{preset}.k2w")
        write_synthetic_model(path, preset)
        rec = pkg.OfflineRecognizer(path)
        wavs = [synth_utterance(40 + u, 3.0 + 0.5 * u) for u in range(4)]

        def run(r):
            ss = []
            for w in wavs:
                s = r.create_offline_stream()
                s.add_samples(w)
                ss.append(s)
            return r.get_results(ss)

        with_table = run(rec)
        rows, bad = C.c_int64(), C.c_int64()
        assert L.k2hip_debug_decoder_table_check(rec.model.handle, 500, 7, C.byref(rows), C.byref(bad)) == 0
        V = rec.model.vocab_size
        assert rows.value == (V + 1) * V, f"{preset}: a vocabulary of {V} must get the table"
        assert bad.value == 0, f"{preset}: {bad.value} table values differ from the decoder's"
        pkg.set_switch("K2HIP_DECODER_TABLE_MB", 0)
        try:
            rec2 = pkg.OfflineRecognizer(path)
            assert L.k2hip_debug_decoder_table_check(rec2.model.handle, 4, 7, C.byref(rows), C.byref(bad)) == 0
            assert rows.value == 0
            without = run(rec2)
        finally:
            pkg.set_switch("K2HIP_DECODER_TABLE_MB", 1024)
        assert with_table == without
        assert sum(len(t) for t, _ in without) > 2 * 4 * len(wavs)
