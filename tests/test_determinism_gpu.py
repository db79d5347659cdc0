"""Bit-reproducibility under timing perturbation: the same batch decoded repeatedly while a SECOND process keeps the same GPU busy
(that is how the multi-rank path is rehearsed on a one-GPU box: kernels of two processes interleave and every latency moves).
This caught a real bug: hand-written `ds_read` asm with a hand-placed `s_waitcnt lgkmcnt` in the pipelined GEMM let the compiler
copy a fragment register before its data had arrived -- 3-13 % of the runs changed a token, only with a second process on the card."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokens_do_not_depend_on_timing(switch=""):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "determinism_stress.py"), "60"] + ([switch] if switch else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "60 iterations, 0 differed" in out.stdout, out.stdout[-2000:]
