#!/usr/bin/env python3
"""Dev tool (GPU): the headline shapes with OTHER audio than the tests use -- zipformer2-large-en, batches of 32 x 10 s (every other
one ragged), greedy search and modified beam search (beam 4) against the CPU oracle.

Greedy: the fused samples -> tokens entry, tests/parity.py criteria (a divergence must start on a frame whose own top-2 gap is a tie).
Beam search, at TWO levels, every differing stream LOCALISED by the per-frame taps of both sides (tests/parity.py localise_beam):
  operator level -- the engine's search on the oracle's encoder_out (only joiner + search differ);
  fused          -- samples -> tokens (the engine's own encoder, 5e-4 from the oracle's, feeds the search).
A miss is accepted only if the searches part on a frame at which the oracle's own scores of the candidates in question differ by less
than 1e-3; anything else raises.  Also counted: streams with EQUAL results whose searches nevertheless parted (the dropped hypothesis
was not the winner), and how many streams the four-launch form of the search decides differently from the one-kernel form.
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
import parity  # noqa: E402
from parity import assert_beam_match, assert_tokens_match, hidden_beam_divergences  # noqa: E402

TOL = 1e-3
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
path = os.path.join(tempfile.mkdtemp(), "large.k2w")
write_synthetic_model(path, "zipformer2-large-en")
hip, ora = pkg.Model(path, 0), Oracle(path)
rng = np.random.default_rng(seed0)
g_exact = op_exact = f_exact = tot = forms_differ = hidden_n = 0
forms_gaps = []
pkg.set_switch("K2HIP_BEAM_TRACE", 1)
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
    bwant, bmg, tr_w = ora.modified_beam_search(enc, 4, want_margins=True, want_trace=True)
    got_op = hip.beam_search(enc, 4)
    op_exact += assert_beam_match(got_op, bwant, bmg, tol=TOL, what=f"batch {k} beam 4 operator level", allow_tie=True,
                                  trace_got=hip.beam_trace(), trace_want=tr_w)
    hip.set_decoding_method("modified_beam_search", 4)
    got = hip.offline_greedy_from_samples(utts)
    tr_f = hip.beam_trace()
    f_exact += assert_beam_match(got, bwant, bmg, tol=TOL, what=f"batch {k} beam 4 fused", allow_tie=True, trace_got=tr_f, trace_want=tr_w)
    hid = hidden_beam_divergences(got, bwant, tr_f, tr_w)
    assert all(g < TOL for _, _, g in hid), hid
    hidden_n += len(hid)
    pkg.set_switch("K2HIP_BEAM_LAUNCHES", 1)   # the four-launches-per-frame form on the same batch: which streams it decides differently,
    try:                                        # and how far apart the oracle's candidates are where either form left the oracle
        alt = hip.offline_greedy_from_samples(utts)
        tr_a = hip.beam_trace()
    finally:
        pkg.set_switch("K2HIP_BEAM_LAUNCHES", 0)
    n_f = len(parity.NEAR_TIES)
    assert_beam_match(alt, bwant, bmg, tol=TOL, what=f"batch {k} beam 4 fused, launch form", allow_tie=True, trace_got=tr_a, trace_want=tr_w)
    gap_l = {e[1]: e[3] for e in parity.NEAR_TIES[n_f:]}
    gap_f = {e[1]: e[3] for e in parity.NEAR_TIES if e[0] == f"batch {k} beam 4 fused"}
    for b_, (a_, c_) in enumerate(zip(got, alt)):
        if a_ != c_:
            forms_differ += 1
            forms_gaps.append((k, b_, gap_f.get(b_), gap_l.get(b_)))
    tot += B
    print(f"batch {k} ({'ragged' if k % 2 else '32 x 10 s'}): greedy {g_exact}/{tot}; beam 4 operator level {op_exact}/{tot}, fused {f_exact}/{tot} "
          f"streams exact so far; {hidden_n} equal results over parted searches; {forms_differ} decided differently by the launch form", flush=True)
print(f"streams the engine's two search forms decide differently (batch, stream, oracle gap where the one-kernel form / the launch form left the oracle): {forms_gaps}")
print(f"every miss localised to a frame whose own oracle gap is < {TOL}:")
for e in parity.NEAR_TIES:
    print("  near-tie (what, stream, frame, gap):", e)
print("soak ok")
