#!/usr/bin/env python3
"""Dev tool (GPU): a longer seeded sweep than tests/test_parity_gpu.py::test_random_ragged_batches_match_oracle, over every offline
model type: random batch sizes and ragged lengths, fused samples -> tokens on the GPU against the CPU oracle (tests/parity.py
criteria).  usage: soak_offline.py [cases-per-model] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402
from oracle import Oracle  # noqa: E402
from parity import assert_beam_match, assert_tokens_match  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
# (preset, metadata overrides, blank bias): the last entry has a vocabulary wide enough for the f16-screened search (k_greedy<true>)
presets = [("zipformer2-tiny-test", None, None), ("zipformer-tiny-test", None, None), ("conformer-tiny-test", None, None), ("lstm-tiny-test", None, None),
           ("lstm-tiny-split-test", None, None), ("zipformer2-tiny-test", {"vocab_size": "3000"}, 2.4)]
tmp = tempfile.mkdtemp()
pkg.set_switch("K2HIP_BEAM_TRACE", 1)
lens_pool = [400, 401, 559, 560, 1999, 3200, 4801, 8000, 12345, 16000, 20001, 31999]
for preset, overrides, bias in presets:
    path = os.path.join(tmp, preset + ("-wide" if overrides else "") + ".k2w")
    write_synthetic_model(path, preset, blank_bias=bias, meta_overrides=overrides)
    if overrides:
        preset += " " + str(overrides)
    hip, ora = pkg.Model(path, 0), Oracle(path)
    rng = np.random.default_rng(seed)
    exact = tot = skipped = bexact = btot = 0
    for case in range(cases):
        B = int(rng.integers(1, 9))
        ns = [int(rng.choice(lens_pool)) if rng.random() < 0.5 else int(rng.integers(400, 40000)) for _ in range(B)]
        utts = [synth_utterance(5000 + 16 * case + b, n / 16000.0)[:n] for b, n in enumerate(ns)]
        feats = [ora.fbank(u) for u in utts]
        x = ora.pad_sequence(feats).reshape(B, -1, 80)
        if ora.encoder_out_frames(x.shape[1]) <= 0:
            skipped += 1
            continue
        want = ora.recognize_batch(feats)
        _, mg = ora.greedy_batch(ora.encoder(x), want_margins=True)
        got = hip.offline_greedy_from_samples(utts)
        exact += assert_tokens_match(got, want, mg, what=f"{preset} case {case} (B={B}, samples={ns})", allow_tie=True)
        tot += B
        if preset == "zipformer2-tiny-test":   # the modified beam search (one kernel per batch on this vocabulary) on the same encoder output
            enc = ora.encoder(x)
            beam = int(rng.choice([2, 4, 8]))
            bwant, bmg, btr = ora.modified_beam_search(enc, beam, want_margins=True, want_trace=True)
            bgot = hip.beam_search(enc, beam)
            bexact += assert_beam_match(bgot, bwant, bmg, what=f"{preset} case {case} beam {beam}", allow_tie=True, trace_got=hip.beam_trace(),
                                        trace_want=btr)
            btot += B
    print(f"{preset}: {cases - skipped} batches, {exact}/{tot} streams token-exact (the rest diverge on an oracle near-tie)", flush=True)
    if btot:
        print(f"{preset}: modified beam search {bexact}/{btot} streams exact", flush=True)
print("soak ok")
