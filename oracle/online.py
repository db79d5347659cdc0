"""ctypes view of the streaming part of the CPU oracle (k2_oracle_online.c).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import Oracle, OracleError, _fp, _ip, _lp, lib

_KINDS = {"key": 0, "nonlin": 1, "val1": 2, "val2": 3, "conv1": 4, "conv2": 5, "embed": 6,
          # Zipformer v1 streams (OnlineProjOfZipformer.cs:56-111)
          "avg": 1, "val": 2, "len": 7}


def _bind(L):
    if getattr(L, "_online_bound", False):
        return
    fp, ip, lp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p
    for f in ("k2o_online_chunk_length", "k2o_online_shift_length", "k2o_online_frames_per_chunk"):
        getattr(L, f).argtypes = [vp]
    L.k2o_online_stream_create.restype = vp
    L.k2o_online_stream_create.argtypes = [vp]
    L.k2o_online_stream_free.argtypes = [vp]
    L.k2o_online_stream_num_layers.argtypes = [vp]
    L.k2o_online_stream_processed_len.restype = C.c_int64
    L.k2o_online_stream_processed_len.argtypes = [vp]
    L.k2o_online_stream_state.restype = C.c_int64
    L.k2o_online_stream_state.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int64]
    L.k2o_online_stream_num_tokens.argtypes = [vp]
    L.k2o_online_stream_num_timestamps.argtypes = [vp]
    L.k2o_online_stream_get_tokens.argtypes = [vp, lp]
    L.k2o_online_stream_get_timestamps.argtypes = [vp, ip]
    L.k2o_online_stream_get_hyp.argtypes = [vp, lp]
    L.k2o_online_encoder_chunk.argtypes = [vp, vp, fp, fp]
    L.k2o_online_step.argtypes = [vp, C.POINTER(vp), C.POINTER(fp), C.c_int, ip]
    L._online_bound = True


class OnlineOracleStream:
    """OnlineStream state as the reference holds it (States, Hyp, Tokens, Timestamps)."""

    def __init__(self, oracle: "OnlineOracle"):
        self._o = oracle
        self._L = oracle._L
        self._s = self._L.k2o_online_stream_create(oracle._m)
        if not self._s:
            raise OracleError(self._L.k2o_last_error().decode())
        self.num_layers = self._L.k2o_online_stream_num_layers(self._s)

    def __del__(self):
        try:
            if self._s:
                self._L.k2o_online_stream_free(self._s)
                self._s = None
        except Exception:
            pass

    def state(self, layer: int, kind: str) -> np.ndarray:
        n = self._L.k2o_online_stream_state(self._s, layer, _KINDS[kind], None, 0)
        out = np.empty(n, np.float32)
        self._L.k2o_online_stream_state(self._s, layer, _KINDS[kind], _fp(out), n)
        return out

    def lstm_state(self, layer: int, kind: str) -> np.ndarray:
        """kind 'h' ([d_model]) or 'c' ([rnn_hidden_size]) of one layer (OnlineProjOfLstm.cs:55-75)"""
        k = {"h": 0, "c": 1}[kind]
        self._L.k2o_online_stream_lstm_state.restype = C.c_int64
        self._L.k2o_online_stream_lstm_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64]
        n = self._L.k2o_online_stream_lstm_state(self._o._m, self._s, layer, k, None, 0)
        out = np.empty(n, np.float32)
        self._L.k2o_online_stream_lstm_state(self._o._m, self._s, layer, k, _fp(out), n)
        return out

    @property
    def processed_len(self) -> int:
        return self._L.k2o_online_stream_processed_len(self._s)

    @property
    def tokens(self):
        n = self._L.k2o_online_stream_num_tokens(self._s)
        out = np.zeros(n, np.int64)
        self._L.k2o_online_stream_get_tokens(self._s, _lp(out))
        return out.tolist()

    @property
    def timestamps(self):
        n = self._L.k2o_online_stream_num_timestamps(self._s)
        out = np.zeros(max(n, 1), np.int32)
        self._L.k2o_online_stream_get_timestamps(self._s, _ip(out))
        return out[:n].tolist()

    @property
    def hyp(self):
        out = np.zeros(2, np.int64)
        self._L.k2o_online_stream_get_hyp(self._s, _lp(out))
        return out.tolist()


class OnlineOracle(Oracle):
    def __init__(self, k2w_path: str):
        super().__init__(k2w_path)
        _bind(self._L)
        self.chunk_length = self._L.k2o_online_chunk_length(self._m)
        self.shift_length = self._L.k2o_online_shift_length(self._m)
        self.frames_per_chunk = self._L.k2o_online_frames_per_chunk(self._m)

    def create_stream(self) -> OnlineOracleStream:
        return OnlineOracleStream(self)

    def encoder_chunk(self, stream: OnlineOracleStream, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32).reshape(self.chunk_length, self.feature_dim)
        out = np.empty((self.frames_per_chunk, self.encoder_out_dim), np.float32)
        rc = self._L.k2o_online_encoder_chunk(self._m, stream._s, _fp(x), _fp(out))
        if rc < 0:
            raise OracleError(self._L.k2o_last_error().decode())
        return out[:rc]

    def step(self, streams, chunks):
        B = len(streams)
        cs = [np.ascontiguousarray(c, np.float32).reshape(-1) for c in chunks]
        sp = (C.c_void_p * B)(*[s._s for s in streams])
        cp = (C.POINTER(C.c_float) * B)(*[_fp(c) for c in cs])
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2o_online_step(self._m, sp, cp, B, _ip(n)))
        return n.tolist()
