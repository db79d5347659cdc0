"""Shared parity criteria.

Integer outputs (tokens, timestamps) must be identical.  The one allowed
exception is a frame where the oracle itself is undecided at fp32 resolution:
if two vocabulary entries are closer than LOGIT_TOL in the oracle's logits, a
different summation order may legally pick the other one, and everything after
that frame follows a different decoder context.  `assert_tokens_match` accepts a
divergence only if its first differing emission sits on such a frame.
"""
import numpy as np

LOGIT_TOL = 1e-3   # north_star: fp32 logits within 1e-3
ACT_TOL = 2e-4     # encoder activations (O(1) magnitude), fp32 accumulate-order noise


def first_divergence(a, b):
    (ta, sa), (tb, sb) = a, b
    n = min(len(ta), len(tb))
    for i in range(n):
        if ta[i] != tb[i] or sa[i] != sb[i]:
            return min(sa[i], sb[i])
    if len(ta) != len(tb):
        return (sa + sb)[n] if n < len(sa + sb) else None
    return None


def assert_tokens_match(got, want, margins=None, tol=LOGIT_TOL, what=""):
    """got/want: list of (tokens, timestamps) per stream; margins: [B, T'] oracle top-2 gaps."""
    assert len(got) == len(want)
    exact = 0
    for b, (g, w) in enumerate(zip(got, want)):
        if g == w:
            exact += 1
            continue
        t = first_divergence(g, w)
        assert margins is not None, f"{what} stream {b}: tokens differ at frame {t}: {g} vs {w}"
        # a near-tie anywhere up to the divergence frame (context switches propagate) excuses it
        m = float(np.min(margins[:, : t + 1])) if margins.ndim == 2 else float(np.min(margins[: t + 1]))
        assert m < tol, (f"{what} stream {b}: tokens differ at frame {t} but the oracle's smallest top-2 gap "
                         f"up to there is {m:.3g} >= {tol}: {g} vs {w}")
    return exact
