#!/usr/bin/env python3
"""Dev tool (GPU): where a round of the persistent greedy search goes (K2HIP_GREEDY_STAMPS: workgroup 0's s_memrealtime differences per
phase, printed by the library on stderr).  usage: greedy_stamps.py [preset] [batch] [seconds]"""
import os
import sys

os.environ["K2HIP_GREEDY_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "conformer-zh"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
path = f"/tmp/k2hip_stamps_{preset}.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, preset)
m = pkg.Model(path, 0)
s = np.stack([synth_utterance(u, secs) for u in range(B)])
ptr = m.device_alloc(s.nbytes)
m.device_upload(ptr, s)
for _ in range(3):
    m.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
    print({k: round(v, 3) for k, v in m.timing().items() if k.endswith("_ms")})
