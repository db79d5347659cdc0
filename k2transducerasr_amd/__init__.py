"""k2transducerasr_amd -- MI355X-native RNN-T decoding engine behind K2TransducerAsr's
IOfflineProj / OfflineRecognizer hot path.

The product is ``libk2hip.so`` (hand-written HIP for gfx950 + a C ABI, see
``include/k2hip.h``).  This package is only the ctypes view of that ABI plus the
synthetic-model helpers the tests and the benchmark need; there is no Python or
CPU implementation of the path here, and importing the compute classes without
the built library raises immediately.
"""
from __future__ import annotations

from . import config, k2w, synth  # noqa: F401
from .binding import (  # noqa: F401
    K2HipError,
    Model,
    OfflineRecognizer,
    OfflineStream,
    OnlineProj,
    OnlineRecognizer,
    OnlineStream,
    StreamBatch,
    TokenTable,
    build_library,
    library_path,
    load_library,
    set_switch,
)

__all__ = [
    "K2HipError",
    "Model",
    "OfflineRecognizer",
    "OfflineStream",
    "OnlineProj",
    "OnlineRecognizer",
    "OnlineStream",
    "StreamBatch",
    "TokenTable",
    "build_library",
    "library_path",
    "load_library",
    "set_switch",
    "config",
    "k2w",
    "synth",
]
