#!/usr/bin/env python3
"""Dev tool (GPU): the headline shapes with OTHER audio than the tests use -- zipformer2-large-en, batches of 32 x 10 s (and a ragged
one), greedy search and modified beam search (beam 4, the two-slab one-kernel form) through the fused samples -> tokens entry, every
stream against the CPU oracle (tests/parity.py criteria; near-ties are reported, not failed).
usage: soak_full_size.py [batches] [first-utterance-seed]"""
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

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
path = os.path.join(tempfile.mkdtemp(), "large.k2w")
write_synthetic_model(path, "zipformer2-large-en")
hip, ora = pkg.Model(path, 0), Oracle(path)
rng = np.random.default_rng(seed0)
g_exact = g_tot = b_exact = b_tot = forms_differ = 0
for k in range(batches):
    B = 32
    secs = [10.0] * B if k % 2 == 0 else [float(rng.uniform(2.0, 10.0)) for _ in range(B)]   # every other batch ragged
    utts = [synth_utterance(seed0 + 64 * k + b, secs[b]) for b in range(B)]
    feats = [ora.fbank(u) for u in utts]
    x = ora.pad_sequence(feats).reshape(B, -1, 80)
    enc = ora.encoder(x)
    want = ora.recognize_batch(feats)
    _, mg = ora.greedy_batch(enc, want_margins=True)
    hip.set_decoding_method("greedy_search")
    g_exact += assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what=f"batch {k} greedy", allow_tie=True)
    g_tot += B
    bwant, bmg = ora.modified_beam_search(enc, 4, want_margins=True)
    hip.set_decoding_method("modified_beam_search", 4)
    got = hip.offline_greedy_from_samples(utts)
    # (the beam's decisions compare SUMS of up to 253 log-probabilities: 2 x the per-logit tolerance for the excuse)
    b_exact += assert_beam_match(got, bwant, bmg, tol=2e-3, what=f"batch {k} beam 4", allow_tie=True)
    pkg.set_switch("K2HIP_BEAM_LAUNCHES", 1)   # the four-launches-per-frame form on the same batch: how many streams it decides differently
    try:
        alt = hip.offline_greedy_from_samples(utts)
    finally:
        pkg.set_switch("K2HIP_BEAM_LAUNCHES", 0)
    forms_differ += sum(1 for x, y in zip(got, alt) if x != y)
    b_tot += B
    print(f"batch {k} ({'ragged' if k % 2 else '32 x 10 s'}): greedy {g_exact}/{g_tot}, beam 4 {b_exact}/{b_tot} streams exact so far ({forms_differ} decided differently by the launch form)", flush=True)
print("soak ok")
