"""ctypes view of oracle/libk2oracle.so -- the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from k2transducerasr_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libk2oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("k2_oracle.c", "k2_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def _cap_omp_threads():
    """The GPU box shows all host cores but gives this job a share of them; an
    uncapped OpenMP team oversubscribes and every parallel region crawls."""
    if "OMP_NUM_THREADS" not in os.environ:
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(n, 16)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _cap_omp_threads()
        L = C.CDLL(_SO)
        fp, ip, lp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.k2o_model_load.restype = C.c_void_p
        L.k2o_model_load.argtypes = [C.c_char_p]
        L.k2o_model_free.argtypes = [C.c_void_p]
        L.k2o_last_error.restype = C.c_char_p
        L.k2o_meta.restype = C.c_char_p
        L.k2o_meta.argtypes = [C.c_void_p, C.c_char_p]
        for f in ("k2o_vocab_size", "k2o_joiner_dim", "k2o_context_size", "k2o_feature_dim"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.k2o_fbank_num_frames.restype = C.c_int64
        L.k2o_fbank_num_frames.argtypes = [C.c_void_p, C.c_int64]
        L.k2o_fbank.restype = C.c_int64
        L.k2o_fbank.argtypes = [C.c_void_p, fp, C.c_int64, fp, C.c_int64]
        L.k2o_pad_sequence.restype = C.c_int64
        L.k2o_pad_sequence.argtypes = [C.POINTER(fp), lp, C.c_int, C.c_int, fp]
        L.k2o_encoder_out_frames.argtypes = [C.c_void_p, C.c_int]
        L.k2o_offline_encoder.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, fp]
        L.k2o_offline_encoder_tap.restype = C.c_int64
        L.k2o_offline_encoder_tap.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int64]
        L.k2o_decoder.argtypes = [C.c_void_p, lp, C.c_int, fp]
        L.k2o_joiner.argtypes = [C.c_void_p, fp, fp, C.c_int, fp]
        L.k2o_argmax_ref.argtypes = [fp, C.c_int]
        L.k2o_greedy_batch.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, lp, ip, ip, C.c_int, fp]
        L.k2o_greedy_single.argtypes = [C.c_void_p, fp, C.c_int, lp, ip, ip, C.c_int, fp]
        L.k2o_offline_recognize_batch.argtypes = [C.c_void_p, C.POINTER(fp), lp, C.c_int, lp, ip, ip, C.c_int]
        L.k2o_encoder_out_dim.argtypes = [C.c_void_p]
        L.k2o_ctc_greedy.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, lp, ip, ip, C.c_int, ip]
        L.k2o_modified_beam_search.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, lp, ip, ip, C.c_int, fp, fp]
        L.k2o_modified_beam_search_trace.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, lp, ip, ip, C.c_int, fp, fp, ip]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class OracleError(RuntimeError):
    pass


class Oracle:
    """The reference path on the CPU.  Method names follow the reference operators."""

    def __init__(self, k2w_path: str):
        self._L = lib()
        self._m = self._L.k2o_model_load(k2w_path.encode())
        if not self._m:
            raise OracleError(self._L.k2o_last_error().decode())
        self.vocab_size = self._L.k2o_vocab_size(self._m)
        self.joiner_dim = self._L.k2o_joiner_dim(self._m)
        self.encoder_out_dim = self._L.k2o_encoder_out_dim(self._m)  # vocab_size for a zipformer2ctc model
        self.context_size = self._L.k2o_context_size(self._m)
        self.feature_dim = self._L.k2o_feature_dim(self._m)

    def close(self):
        if self._m:
            self._L.k2o_model_free(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise OracleError(self._L.k2o_last_error().decode())
        return rc

    def meta(self, key):
        v = self._L.k2o_meta(self._m, key.encode())
        return None if v is None else v.decode()

    # F1
    def fbank(self, samples: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(samples, dtype=np.float32)
        nf = self._L.k2o_fbank_num_frames(self._m, s.size)
        out = np.empty((nf, self.feature_dim), np.float32)
        self._chk(self._L.k2o_fbank(self._m, _fp(s), s.size, _fp(out), nf))
        return out

    # F3
    def pad_sequence(self, feats, tail_frames=19) -> np.ndarray:
        feats = [np.ascontiguousarray(f, dtype=np.float32).reshape(-1) for f in feats]
        B = len(feats)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(f) for f in feats])
        n = np.array([f.size for f in feats], np.int64)
        L = self._L.k2o_pad_sequence(ptrs, _lp(n), B, tail_frames, None)
        out = np.empty((B, L), np.float32)
        self._L.k2o_pad_sequence(ptrs, _lp(n), B, tail_frames, _fp(out))
        return out

    # F4
    def encoder_out_frames(self, T: int) -> int:
        return self._L.k2o_encoder_out_frames(self._m, T)

    def encoder(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, T, _ = x.shape
        Tp = self.encoder_out_frames(T)
        out = np.empty((B, Tp, self.encoder_out_dim), np.float32)
        self._chk(self._L.k2o_offline_encoder(self._m, _fp(x), B, T, _fp(out)))
        return out

    def encoder_tap(self, x: np.ndarray, tap: int) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, T, _ = x.shape
        cap = B * T * 1024
        buf = np.empty(cap, np.float32)
        n = self._chk(self._L.k2o_offline_encoder_tap(self._m, _fp(x), B, T, tap, _fp(buf), cap))
        return buf[:n].reshape(B, -1).copy()

    # F5 / F6
    def decoder(self, y: np.ndarray) -> np.ndarray:
        y = np.ascontiguousarray(y, dtype=np.int64).reshape(-1, self.context_size)
        out = np.empty((y.shape[0], self.joiner_dim), np.float32)
        self._chk(self._L.k2o_decoder(self._m, _lp(y), y.shape[0], _fp(out)))
        return out

    def joiner(self, enc: np.ndarray, dec: np.ndarray) -> np.ndarray:
        enc = np.ascontiguousarray(enc, dtype=np.float32).reshape(-1, self.joiner_dim)
        dec = np.ascontiguousarray(dec, dtype=np.float32).reshape(-1, self.joiner_dim)
        out = np.empty((enc.shape[0], self.vocab_size), np.float32)
        self._chk(self._L.k2o_joiner(self._m, _fp(enc), _fp(dec), enc.shape[0], _fp(out)))
        return out

    def argmax_ref(self, logits: np.ndarray) -> int:
        l = np.ascontiguousarray(logits, dtype=np.float32)
        return self._L.k2o_argmax_ref(_fp(l), l.size)

    # F7
    def greedy_batch(self, enc_out: np.ndarray, want_margins=False):
        e = np.ascontiguousarray(enc_out, dtype=np.float32)
        B, Tp, _ = e.shape
        mt = Tp + 1
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        mg = np.zeros((B, Tp), np.float32) if want_margins else None
        self._chk(self._L.k2o_greedy_batch(self._m, _fp(e), B, Tp, _lp(tok), _ip(ts), _ip(n), mt,
                                           _fp(mg) if want_margins else None))
        res = [(tok[b, : n[b]].tolist(), ts[b, : n[b]].tolist()) for b in range(B)]
        return (res, mg) if want_margins else res

    def greedy_single(self, enc_out: np.ndarray, want_margins=False):
        e = np.ascontiguousarray(enc_out, dtype=np.float32).reshape(-1, self.joiner_dim)
        Tp = e.shape[0]
        mt = Tp + 1
        tok = np.zeros(mt, np.int64)
        ts = np.zeros(mt, np.int32)
        n = np.zeros(1, np.int32)
        mg = np.zeros(Tp, np.float32) if want_margins else None
        self._chk(self._L.k2o_greedy_single(self._m, _fp(e), Tp, _lp(tok), _ip(ts), _ip(n), mt,
                                            _fp(mg) if want_margins else None))
        res = (tok[: n[0]].tolist(), ts[: n[0]].tolist())
        return (res, mg) if want_margins else res

    def ctc_greedy(self, log_probs: np.ndarray, frame_offsets=None, num_trailing_blank=None):
        """ForwardBatchGreedySearchCTC over log_probs [B,T',V]; returns [(tokens, timestamps)] and trailing-blank counts"""
        lp_ = np.ascontiguousarray(log_probs, dtype=np.float32)
        B, Tp, V = lp_.shape
        tok = np.zeros((B, Tp + 1), np.int64)
        ts = np.zeros((B, Tp + 1), np.int32)
        n = np.zeros(B, np.int32)
        fo = np.zeros(B, np.int32) if frame_offsets is None else np.ascontiguousarray(frame_offsets, dtype=np.int32)
        tb = np.zeros(B, np.int32) if num_trailing_blank is None else np.ascontiguousarray(num_trailing_blank, dtype=np.int32).copy()
        self._chk(self._L.k2o_ctc_greedy(_fp(lp_), B, Tp, V, _ip(fo), _lp(tok), _ip(ts), _ip(n), Tp + 1, _ip(tb)))
        return [(tok[b, : n[b]].tolist(), ts[b, : n[b]].tolist()) for b in range(B)], tb

    def modified_beam_search(self, enc_out: np.ndarray, beam: int = 4, want_margins=False, want_scores=False, want_trace=False):
        """icefall modified_beam_search per stream (k2_oracle_beam.c); returns [(tokens, timestamps)] (+ margins [B,T'+1])
        (+ scores [B]) (+ trace: dict(idx [B,T',2 beam] flat candidate indexes in rank order, val [B,T',2 beam] their scores,
        n [B,T'] surviving hypotheses per frame) -- the per-frame tap tests/parity.py localises a divergence with)."""
        e = np.ascontiguousarray(enc_out, dtype=np.float32)
        B, Tp, _ = e.shape
        mt = Tp + 1
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        sc = np.zeros(B, np.float32)
        mg = np.zeros((B, Tp + 1), np.float32)
        tr = np.zeros((B, Tp, 4 * beam + 1), np.int32)
        self._chk(self._L.k2o_modified_beam_search_trace(self._m, _fp(e), B, Tp, beam, _lp(tok), _ip(ts), _ip(n), mt, _fp(sc), _fp(mg),
                                                         _ip(tr) if want_trace else None))
        res = [(tok[b, : n[b]].tolist(), ts[b, : n[b]].tolist()) for b in range(B)]
        out = (res,)
        if want_margins:
            out += (mg,)
        if want_scores:
            out += (sc,)
        if want_trace:
            out += (dict(idx=tr[:, :, : 2 * beam].copy(), val=tr[:, :, 2 * beam: 4 * beam].copy().view(np.float32), n=tr[:, :, 4 * beam].copy(),
                         beam=beam),)
        return out if len(out) > 1 else res

    def recognize_batch(self, feats):
        feats = [np.ascontiguousarray(f, dtype=np.float32).reshape(-1) for f in feats]
        B = len(feats)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(f) for f in feats])
        nfl = np.array([f.size for f in feats], np.int64)
        mt = int(nfl.max() // self.feature_dim) + 32
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2o_offline_recognize_batch(self._m, ptrs, _lp(nfl), B, _lp(tok), _ip(ts), _ip(n), mt))
        return [(tok[b, : n[b]].tolist(), ts[b, : n[b]].tolist()) for b in range(B)]
