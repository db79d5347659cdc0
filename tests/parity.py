"""Shared parity criteria.

Integer outputs (tokens, timestamps) must be identical: `assert_tokens_match` is STRICT by default
and fails on any difference.

The north star allows fp32 logits within 1e-3, so a frame on which the oracle's own top-2 gap is
below that resolution may legally resolve the other way under a different summation order, and
everything after it follows a different decoder context.  A caller that wants to tolerate exactly
that passes `allow_tie=True`; then a divergence of stream b is excused only if

  * stream b's OWN margin at the frame where b first diverges is < tol, or
  * (batch entry points) some stream's margin at the batch's first-emission frame t0 is < tol --
    the reference re-runs the decoder for every stream there (OfflineRecognizer.cs:278-286), so a
    tie on that single frame changes every stream's context from t0 + 1 on.

Every excuse is recorded in `EXCUSED`; tests/conftest.py prints the count at the end of the session
and fails the session if it is not zero (K2HIP_ALLOW_TIES=1 downgrades that to a report, for soak
runs over thousands of random streams).

Modified beam search (no reference behaviour: the oracle is the only truth): `assert_beam_match` is exact by
default; a test that decodes audio nobody has looked at before may pass allow_tie=True, and then a differing
stream must be LOCALISED by the two sides' per-frame taps to one frame whose own oracle decision gap is below
the tolerance (`localise_beam`); those are listed in `NEAR_TIES` and printed, an unlocalised one fails.
"""
import numpy as np

LOGIT_TOL = 1e-3   # north_star: fp32 logits within 1e-3
ACT_TOL = 2e-4     # encoder activations (O(1) magnitude), fp32 accumulate-order noise

EXCUSED = []       # (what, stream, frame, margin) of every tolerated divergence in this process
COMPARED = [0]     # streams compared through assert_tokens_match (all of them exact unless listed above)


def first_divergence(a, b):
    """frame index of the first emission on which two (tokens, timestamps) results differ"""
    (ta, sa), (tb, sb) = a, b
    n = min(len(ta), len(tb))
    for i in range(n):
        if ta[i] != tb[i] or sa[i] != sb[i]:
            return min(sa[i], sb[i])
    if len(ta) != len(tb):
        rest = sa[n:] if len(sa) > n else sb[n:]
        return rest[0] if rest else None
    return None


def _first_emit(results):
    ts = [r[1][0] for r in results if r[1]]
    return min(ts) if ts else None


def assert_tokens_match(got, want, margins=None, tol=LOGIT_TOL, what="", allow_tie=False, batch_context=True):
    """got/want: list of (tokens, timestamps) per stream; margins: [B, T'] (or [T'] for one stream)
    oracle top-2 gaps.  Returns the number of exactly equal streams (== len(want) unless allow_tie)."""
    assert len(got) == len(want), f"{what}: {len(got)} results for {len(want)} streams"
    exact = 0
    COMPARED[0] += len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        if g == w:
            exact += 1
            continue
        t = first_divergence(g, w)
        msg = f"{what} stream {b}: tokens differ at frame {t}: got {g} want {w}"
        assert allow_tie and margins is not None and t is not None, msg
        mg = np.asarray(margins)
        own = float(mg[b, t]) if mg.ndim == 2 else float(mg[t])
        cands = [own]
        if batch_context and mg.ndim == 2:
            # the batch's first-emission frame, as seen by either side
            for t0 in {_first_emit(got), _first_emit(want)} - {None}:
                if t0 <= t:
                    cands.append(float(mg[:, t0].min()))
        m = min(cands)
        assert m < tol, f"{msg}; the oracle's top-2 gap there is {own:.3g} (batch t0 gap {cands[1:]}) >= {tol}"
        EXCUSED.append((what, b, int(t), m))
    return exact


NEAR_TIES = []     # beam-search divergences LOCALISED to one frame whose own oracle decision gap is < tol (reported, not failed)


def localise_beam(tr_got, tr_want, b):
    """Where do two modified beam searches of stream b part?  tr_got: the engine's per-frame tap (Model.beam_trace()), tr_want: the
    oracle's (Oracle.modified_beam_search(want_trace=True)).  Equal histories up to frame t - 1 put the same hypotheses into the same
    slots, so the first frame whose ranked selections (flat index = slot * V + token) or survivor counts differ IS the frame of the
    divergence.  Returns (t, gap): gap = the largest difference, over the ranks that differ, between the ORACLE's scores of the
    candidate the oracle put at that rank and of the one the engine put there (the oracle ranks 2 * beam candidates per frame; a
    candidate of the engine's outside that list has gap = inf).  (None, None) if the taps agree on every frame -- then only the final
    length-normalised pick can differ."""
    K = tr_got["beam"]
    gi, wi, wv = tr_got["idx"][b], tr_want["idx"][b], tr_want["val"][b]
    assert gi.shape[0] == wi.shape[0], "taps of different lengths"
    sel_differs = (gi != wi[:, :K]).any(axis=1) | (tr_got["n"][b] != tr_want["n"][b])
    if not sel_differs.any():
        return None, None
    t = int(np.argmax(sel_differs))
    gap, ranks = 0.0, 0
    for r in range(K):
        if gi[t, r] == wi[t, r]:
            continue
        ranks += 1
        where = np.nonzero(wi[t] == gi[t, r])[0]
        if gi[t, r] < 0 or wi[t, r] < 0 or where.size == 0:
            return t, float("inf")
        gap = max(gap, abs(float(wv[t, r]) - float(wv[t, where[0]])))
    # (the same selection with a different survivor count would mean the MERGE decided differently on equal inputs: never a tie)
    return t, (gap if ranks else float("inf"))


def assert_beam_match(got, want, margins, tol=LOGIT_TOL, what="", allow_tie=False, trace_got=None, trace_want=None):
    """Modified beam search: exact by default.  With allow_tie a differing stream is tolerated only if the divergence is LOCALISED:
    the per-frame taps of both sides (trace_got / trace_want, see localise_beam) name the first frame at which the selections differ
    and the oracle's own scores of the candidates in question lie within `tol` of each other AT THAT FRAME; if the taps agree on
    every frame, the final length-normalised pick must have been closer than `tol` (margins[b][-1]).  A whole-stream minimum gap is
    not accepted as an excuse (with ~254 decisions per stream it is nearly always small).  Returns the number of exactly equal
    streams; every tolerated stream is recorded in NEAR_TIES with its frame and gap."""
    assert len(got) == len(want)
    exact = 0
    COMPARED[0] += len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        if g == w:
            exact += 1
            continue
        msg = f"{what} stream {b}: beam results differ: got {g} want {w}"
        assert allow_tie, msg
        assert trace_got is not None and trace_want is not None, f"{msg}; no per-frame taps to localise the divergence with"
        t, gap = localise_beam(trace_got, trace_want, b)
        if t is None:
            gap = float(np.asarray(margins)[b][-1])
            assert gap < tol, f"{msg}; every frame's selection agrees and the final pick's gap is {gap:.3g} >= {tol}"
            NEAR_TIES.append((what, b, -1, gap))
        else:
            assert gap < tol, f"{msg}; the searches part at frame {t}, where the oracle's scores of the candidates in question differ by {gap:.3g} >= {tol}"
            NEAR_TIES.append((what, b, t, gap))
    return exact


def hidden_beam_divergences(got, want, trace_got, trace_want):
    """streams whose RESULTS agree although the searches parted on some frame (the dropped hypothesis was not the winner):
    [(b, t, gap)] -- informational, for the soak tools"""
    out = []
    for b, (g, w) in enumerate(zip(got, want)):
        if g != w:
            continue
        t, gap = localise_beam(trace_got, trace_want, b)
        if t is not None:
            out.append((b, t, gap))
    return out
