"""ctypes binding of include/k2hip.h.

Class and method names follow the reference's host API so that the parity tests
read like the reference's own call sites:

    reference (C#)                                  here
    ---------------------------------------------   ---------------------------------
    new OfflineRecognizer(enc, dec, joiner, tok)    OfflineRecognizer(weights_path)
    recognizer.CreateOfflineStream()                recognizer.create_offline_stream()
    stream.AddSamples(float[])                      stream.add_samples(np.ndarray)
    recognizer.GetResults(List<OfflineStream>)      recognizer.get_results([streams])
    recognizer.GetResult(stream)                    recognizer.get_result(stream)
    IOfflineProj.EncoderProj / DecoderProj / ...    Model.encoder_proj / decoder_proj / joiner_proj

(K2TransducerAsr/OfflineRecognizer.cs:27-91, OfflineStream.cs:43-57,
IOfflineProj.cs:43-47.)  All arithmetic happens inside libk2hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libk2hip.so")
_CSRC = os.path.join(_HERE, "csrc")


class K2HipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"k2hip error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return _SO


def build_library(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libk2hip.so (in-tree)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "-s", "-j8"])
    return _SO


class TimingStruct(C.Structure):
    _fields_ = [
        ("total_ms", C.c_float), ("fbank_ms", C.c_float), ("pad_ms", C.c_float), ("encoder_ms", C.c_float),
        ("greedy_ms", C.c_float), ("d2h_ms", C.c_float), ("gemm_ms", C.c_float), ("gemm_launches", C.c_int32),
        ("gemm_flops", C.c_double), ("total_flops", C.c_double),
    ]


class InfoStruct(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "context_size", "joiner_dim", "feature_dim", "sample_rate",
                                         "num_stacks", "device", "reserved")]


_lib = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)


def load_library():
    """Load libk2hip.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise K2HipError(-2, f"{_SO} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for this package)")
    L = C.CDLL(_SO)
    vp = C.c_void_p
    L.k2hip_version.restype = C.c_char_p
    L.k2hip_last_error.restype = C.c_char_p
    L.k2hip_device_count.restype = C.c_int32
    L.k2hip_model_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(vp)]
    L.k2hip_model_destroy.argtypes = [vp]
    L.k2hip_model_get_info.argtypes = [vp, C.POINTER(InfoStruct)]
    L.k2hip_model_meta.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int32]
    L.k2hip_set_instrument.argtypes = [vp, C.c_int32]
    L.k2hip_get_timing.argtypes = [vp, C.POINTER(TimingStruct)]
    L.k2hip_fbank_num_frames.restype = C.c_int64
    L.k2hip_fbank_num_frames.argtypes = [vp, C.c_int64]
    L.k2hip_fbank.argtypes = [vp, fp, C.c_int64, fp, C.c_int64, lp]
    L.k2hip_pad_sequence.argtypes = [vp, C.POINTER(fp), lp, C.c_int32, C.c_int32, fp, C.c_int64, lp]
    L.k2hip_encoder_out_frames.argtypes = [vp, C.c_int32]
    L.k2hip_offline_encoder.argtypes = [vp, fp, lp, C.c_int32, C.c_int32, fp, C.c_int64, lp, ip]
    L.k2hip_offline_encoder_tap.argtypes = [vp, fp, C.c_int32, C.c_int32, C.c_int32, fp, C.c_int64, lp]
    L.k2hip_decoder.argtypes = [vp, lp, C.c_int32, fp]
    L.k2hip_joiner.argtypes = [vp, fp, fp, C.c_int32, fp]
    L.k2hip_greedy_batch.argtypes = [vp, fp, C.c_int32, C.c_int32, lp, ip, ip, C.c_int32]
    L.k2hip_beam_search.argtypes = [vp, fp, C.c_int32, C.c_int32, C.c_int32, lp, ip, ip, C.c_int32, fp]
    L.k2hip_set_decoding_method.argtypes = [vp, C.c_char_p, C.c_int32]
    L.k2hip_last_scores.argtypes = [vp, fp, C.c_int32]
    L.k2hip_greedy_single.argtypes = [vp, fp, C.c_int32, lp, ip, ip, C.c_int32]
    L.k2hip_offline_greedy.argtypes = [vp, C.POINTER(fp), lp, C.c_int32, lp, ip, ip, C.c_int32]
    L.k2hip_offline_greedy_single.argtypes = [vp, fp, C.c_int64, lp, ip, ip, C.c_int32]
    L.k2hip_offline_greedy_from_samples.argtypes = [vp, C.POINTER(fp), lp, C.c_int32, lp, ip, ip, C.c_int32]
    L.k2hip_offline_greedy_from_samples_dev.argtypes = [vp, vp, C.c_int64, C.c_int32, lp, ip, ip, C.c_int32]
    L.k2hip_offline_submit_samples_dev.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, ip]
    L.k2hip_offline_wait.argtypes = [vp, C.c_int32, lp, ip, ip]
    L.k2hip_offline_submit_samples.argtypes = [vp, fp, C.c_int64, C.c_int32, C.c_int32, ip]
    L.k2hip_host_alloc.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.k2hip_host_free.argtypes = [vp, vp]
    L.k2hip_device_alloc.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.k2hip_device_free.argtypes = [vp, vp]
    L.k2hip_device_upload.argtypes = [vp, vp, vp, C.c_int64]
    L.k2hip_synchronize.argtypes = [vp]
    L.k2hip_offline_stream_create.argtypes = [vp, C.POINTER(vp)]
    L.k2hip_offline_stream_destroy.argtypes = [vp]
    L.k2hip_offline_stream_accept_samples.argtypes = [vp, fp, C.c_int64]
    L.k2hip_offline_stream_speech_length.restype = C.c_int64
    L.k2hip_offline_stream_speech_length.argtypes = [vp]
    L.k2hip_offline_stream_get_speech.argtypes = [vp, fp, C.c_int64]
    L.k2hip_offline_recognizer_get_results.argtypes = [vp, C.POINTER(vp), C.c_int32]
    L.k2hip_offline_recognizer_get_result.argtypes = [vp, vp]
    L.k2hip_offline_stream_num_tokens.argtypes = [vp]
    L.k2hip_offline_stream_num_timestamps.argtypes = [vp]
    L.k2hip_offline_stream_get_tokens.argtypes = [vp, lp, C.c_int32]
    L.k2hip_offline_stream_get_timestamps.argtypes = [vp, ip, C.c_int32]
    _lib = L
    return L


def set_switch(env_name: str, value: int = 1):
    """k2hip_debug_set_switch: flip one development switch (named like its K2HIP_* environment variable) after start-up"""
    L = load_library()
    L.k2hip_debug_set_switch.argtypes = [C.c_char_p, C.c_int32]
    rc = L.k2hip_debug_set_switch(env_name.encode(), int(value))
    if rc != 0:
        raise K2HipError(rc, L.k2hip_last_error().decode())


def _f(a):
    return a.ctypes.data_as(fp)


def _i(a):
    return a.ctypes.data_as(ip)


def _l(a):
    return a.ctypes.data_as(lp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class TokenTable:
    """tokens.txt + the reference's token -> text stage (OfflineRecognizer.DecodeMulti / CheckText, :432-565); host only."""

    def __init__(self, tokens_path: str):
        self._L = load_library()
        self._L.k2hip_tokens_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        self._L.k2hip_tokens_destroy.argtypes = [C.c_void_p]
        self._L.k2hip_tokens_size.argtypes = [C.c_void_p]
        self._L.k2hip_decode_text.argtypes = [C.c_void_p, lp, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, ip]
        h = C.c_void_p()
        rc = self._L.k2hip_tokens_load(tokens_path.encode(), C.byref(h))
        if rc != 0:
            raise K2HipError(rc, self._L.k2hip_last_error().decode())
        self._h = h

    def __len__(self):
        return self._L.k2hip_tokens_size(self._h)

    def decode(self, ids, online: bool = False) -> str:
        a = np.ascontiguousarray(ids, dtype=np.int64)
        n = C.c_int32()
        rc = self._L.k2hip_decode_text(self._h, _l(a), a.size, int(online), None, 0, C.byref(n))
        if rc != 0:
            raise K2HipError(rc, self._L.k2hip_last_error().decode())
        buf = C.create_string_buffer(n.value + 1)
        rc = self._L.k2hip_decode_text(self._h, _l(a), a.size, int(online), buf, n.value + 1, C.byref(n))
        if rc != 0:
            raise K2HipError(rc, self._L.k2hip_last_error().decode())
        return buf.raw[: n.value].decode("utf-8")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.k2hip_tokens_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Model:
    """One model replica on one GPU (k2hip_model_t): the IOfflineProj operators."""

    def __init__(self, weights_path: str, device: int = 0, overrides: Optional[str] = None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.k2hip_model_create(weights_path.encode(), overrides.encode() if overrides else None, device, C.byref(h))
        if rc != 0:
            raise K2HipError(rc, self._L.k2hip_last_error().decode())
        self._h = h
        info = InfoStruct()
        self._chk(self._L.k2hip_model_get_info(self._h, C.byref(info)))
        self.vocab_size = info.vocab_size
        self.context_size = info.context_size
        self.joiner_dim = info.joiner_dim
        self.encoder_out_dim = info.reserved  # vocab_size for a zipformer2ctc model (log_probs)
        self.feature_dim = info.feature_dim
        self.sample_rate = info.sample_rate
        self.device = info.device
        self.blank_id, self.sos_eos_id, self.unk_id = 0, 1, 2  # OfflineModel.cs:18-20

    def _chk(self, rc):
        if rc != 0:
            raise K2HipError(rc, self._L.k2hip_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.k2hip_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def meta(self, key: str) -> str:
        buf = C.create_string_buffer(4096)
        self._chk(self._L.k2hip_model_meta(self._h, key.encode(), buf, 4096))
        return buf.value.decode()

    def set_instrument(self, on: bool):
        self._chk(self._L.k2hip_set_instrument(self._h, int(on)))

    def timing(self) -> dict:
        t = TimingStruct()
        self._chk(self._L.k2hip_get_timing(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in TimingStruct._fields_}

    # ---- F1
    def fbank_num_frames(self, n: int) -> int:
        return self._L.k2hip_fbank_num_frames(self._h, n)

    def fbank(self, samples) -> np.ndarray:
        s = _f32(samples).reshape(-1)
        nf = max(self.fbank_num_frames(s.size), 0)
        out = np.empty((nf, self.feature_dim), np.float32)
        got = C.c_int64()
        self._chk(self._L.k2hip_fbank(self._h, _f(s), s.size, _f(out), nf, C.byref(got)))
        return out[: got.value]

    # ---- F3
    def pad_sequence(self, feats: Sequence[np.ndarray], tail_frames: int = 19) -> np.ndarray:
        fs = [_f32(f).reshape(-1) for f in feats]
        B = len(fs)
        ptrs = (fp * B)(*[_f(f) for f in fs])
        n = np.array([f.size for f in fs], np.int64)
        L = int(n.max()) + 80 * tail_frames
        out = np.empty((B, L), np.float32)
        got = C.c_int64()
        self._chk(self._L.k2hip_pad_sequence(self._h, ptrs, _l(n), B, tail_frames, _f(out), out.size, C.byref(got)))
        assert got.value == L
        return out

    # ---- F4-F6: IOfflineProj
    def encoder_out_frames(self, T: int) -> int:
        return self._L.k2hip_encoder_out_frames(self._h, T)

    def encoder_proj(self, x) -> np.ndarray:
        x = _f32(x)
        B, T, _ = x.shape
        tp = max(self.encoder_out_frames(T), 0)
        out = np.empty((B, tp, self.encoder_out_dim), np.float32)
        lens = np.zeros(B, np.int64)
        xl = np.full(B, T, np.int64)
        got = C.c_int32()
        self._chk(self._L.k2hip_offline_encoder(self._h, _f(x), _l(xl), B, T, _f(out), out.size, _l(lens), C.byref(got)))
        return out

    def encoder_tap(self, x, tap: int) -> np.ndarray:
        x = _f32(x)
        B, T, _ = x.shape
        buf = np.empty(B * T * 1024, np.float32)
        n = C.c_int64()
        self._chk(self._L.k2hip_offline_encoder_tap(self._h, _f(x), B, T, tap, _f(buf), buf.size, C.byref(n)))
        return buf[: n.value].reshape(B, -1).copy()

    def decoder_proj(self, y=None, N: Optional[int] = None) -> np.ndarray:
        if y is None:
            out = np.empty((N, self.joiner_dim), np.float32)
            self._chk(self._L.k2hip_decoder(self._h, None, N, _f(out)))
            return out
        y = np.ascontiguousarray(y, dtype=np.int64).reshape(-1, self.context_size)
        out = np.empty((y.shape[0], self.joiner_dim), np.float32)
        self._chk(self._L.k2hip_decoder(self._h, _l(y), y.shape[0], _f(out)))
        return out

    def joiner_proj(self, enc, dec) -> np.ndarray:
        enc = _f32(enc).reshape(-1, self.joiner_dim)
        dec = _f32(dec).reshape(-1, self.joiner_dim)
        out = np.empty((enc.shape[0], self.vocab_size), np.float32)
        self._chk(self._L.k2hip_joiner(self._h, _f(enc), _f(dec), enc.shape[0], _f(out)))
        return out

    # ---- F7
    def _unpack(self, tok, ts, n):
        return [(tok[b, : n[b]].tolist(), ts[b, : n[b]].tolist()) for b in range(tok.shape[0])]

    def greedy_batch(self, enc_out):
        e = _f32(enc_out)
        B, Tp, _ = e.shape
        tok = np.zeros((B, Tp), np.int64)
        ts = np.zeros((B, Tp), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2hip_greedy_batch(self._h, _f(e), B, Tp, _l(tok), _i(ts), _i(n), Tp))
        return self._unpack(tok, ts, n)

    def ctc_greedy(self, log_probs, frame_offsets=None, num_trailing_blank=None):
        """ForwardBatchGreedySearchCTC over host log_probs [B,T',V] (k2hip_ctc_greedy)"""
        e = _f32(log_probs)
        B, Tp, _ = e.shape
        tok = np.zeros((B, Tp), np.int64)
        ts = np.zeros((B, Tp), np.int32)
        n = np.zeros(B, np.int32)
        fo = np.zeros(B, np.int32) if frame_offsets is None else np.ascontiguousarray(frame_offsets, dtype=np.int32)
        tb = np.zeros(B, np.int32) if num_trailing_blank is None else np.ascontiguousarray(num_trailing_blank, dtype=np.int32).copy()
        self._L.k2hip_ctc_greedy.argtypes = [C.c_void_p, fp, C.c_int32, C.c_int32, ip, lp, ip, ip, C.c_int32, ip]
        self._chk(self._L.k2hip_ctc_greedy(self._h, _f(e), B, Tp, _i(fo), _l(tok), _i(ts), _i(n), Tp, _i(tb)))
        return self._unpack(tok, ts, n), tb

    def beam_search(self, enc_out, beam: int = 4, want_scores: bool = False):
        """modified beam search over a host encoder_out [B,T',J] (k2hip_beam_search)"""
        e = _f32(enc_out)
        B, Tp, _ = e.shape
        tok = np.zeros((B, Tp), np.int64)
        ts = np.zeros((B, Tp), np.int32)
        n = np.zeros(B, np.int32)
        sc = np.zeros(B, np.float32)
        self._chk(self._L.k2hip_beam_search(self._h, _f(e), B, Tp, beam, _l(tok), _i(ts), _i(n), Tp, _f(sc)))
        res = self._unpack(tok, ts, n)
        return (res, sc) if want_scores else res

    def beam_trace(self):
        """k2hip_debug_beam_trace (include/k2hip_debug.h): the per-frame selection of the last synchronous modified beam search made
        with the switch K2HIP_BEAM_TRACE on -> dict(idx [B,T',beam] flat candidate indexes (slot * V + token) in rank order,
        val [B,T',beam] their scores, n [B,T'] hypotheses surviving the frame, beam)"""
        B, Tp, K = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._L.k2hip_debug_beam_trace.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64] + [C.POINTER(C.c_int32)] * 3
        self._chk(self._L.k2hip_debug_beam_trace(self._h, None, 0, C.byref(B), C.byref(Tp), C.byref(K)))
        tr = np.zeros((B.value, Tp.value, 2 * K.value + 1), np.int32)
        self._chk(self._L.k2hip_debug_beam_trace(self._h, _i(tr), tr.size, C.byref(B), C.byref(Tp), C.byref(K)))
        k = K.value
        return dict(idx=tr[:, :, :k].copy(), val=tr[:, :, k: 2 * k].copy().view(np.float32), n=tr[:, :, 2 * k].copy(), beam=k)

    def set_decoding_method(self, method: str = "greedy_search", beam: int = 4):
        """decodingMethod of the batch entry points (OfflineRecognizer.cs:54-68): greedy_search | modified_beam_search"""
        self._chk(self._L.k2hip_set_decoding_method(self._h, method.encode(), beam))

    def gemm_profile(self) -> np.ndarray:
        """[n, 8] rows (M, N, K, batch, act, has_residual, kind, us) of the last instrumented call"""
        n = C.c_int32(0)
        self._L.k2hip_get_gemm_profile.argtypes = [C.c_void_p, fp, C.c_int32, C.POINTER(C.c_int32)]
        self._chk(self._L.k2hip_get_gemm_profile(self._h, None, 0, C.byref(n)))
        rows = np.zeros((max(n.value, 1), 8), np.float32)
        self._chk(self._L.k2hip_get_gemm_profile(self._h, _f(rows), n.value, C.byref(n)))
        return rows[: n.value]

    def last_scores(self, B: int):
        sc = np.zeros(B, np.float32)
        self._chk(self._L.k2hip_last_scores(self._h, _f(sc), B))
        return sc

    def greedy_single(self, enc_out):
        e = _f32(enc_out).reshape(-1, self.joiner_dim)
        Tp = e.shape[0]
        tok = np.zeros((1, Tp), np.int64)
        ts = np.zeros((1, Tp), np.int32)
        n = np.zeros(1, np.int32)
        self._chk(self._L.k2hip_greedy_single(self._h, _f(e), Tp, _l(tok), _i(ts), _i(n), Tp))
        return self._unpack(tok, ts, n)[0]

    def offline_greedy(self, feats: Sequence[np.ndarray]):
        fs = [_f32(f).reshape(-1) for f in feats]
        B = len(fs)
        ptrs = (fp * B)(*[_f(f) for f in fs])
        nfl = np.array([f.size for f in fs], np.int64)
        mt = max(1, self.encoder_out_frames(int(nfl.max()) // self.feature_dim + 19))
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2hip_offline_greedy(self._h, ptrs, _l(nfl), B, _l(tok), _i(ts), _i(n), mt))
        return self._unpack(tok, ts, n)

    def offline_greedy_single(self, feats: np.ndarray):
        f = _f32(feats).reshape(-1)
        mt = max(1, self.encoder_out_frames(f.size // self.feature_dim + 19))
        tok = np.zeros((1, mt), np.int64)
        ts = np.zeros((1, mt), np.int32)
        n = np.zeros(1, np.int32)
        self._chk(self._L.k2hip_offline_greedy_single(self._h, _f(f), f.size, _l(tok), _i(ts), _i(n), mt))
        return self._unpack(tok, ts, n)[0]

    def offline_greedy_from_samples(self, samples: Sequence[np.ndarray]):
        ss = [_f32(s).reshape(-1) for s in samples]
        B = len(ss)
        ptrs = (fp * B)(*[_f(s) for s in ss])
        ns = np.array([s.size for s in ss], np.int64)
        mt = max(1, self.encoder_out_frames(self.fbank_num_frames(int(ns.max())) + 19))
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2hip_offline_greedy_from_samples(self._h, ptrs, _l(ns), B, _l(tok), _i(ts), _i(n), mt))
        return self._unpack(tok, ts, n)

    # ---- device-resident benchmark path
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self._L.k2hip_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr: int):
        self._chk(self._L.k2hip_device_free(self._h, C.c_void_p(ptr)))

    def device_upload(self, ptr: int, host: np.ndarray):
        h = np.ascontiguousarray(host)
        self._chk(self._L.k2hip_device_upload(self._h, C.c_void_p(ptr), h.ctypes.data_as(C.c_void_p), h.nbytes))

    def synchronize(self):
        self._chk(self._L.k2hip_synchronize(self._h))

    def offline_greedy_from_samples_dev(self, dev_ptr: int, n_each: int, B: int, max_tokens: Optional[int] = None):
        mt = max_tokens or max(1, self.encoder_out_frames(self.fbank_num_frames(n_each) + 19))
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2hip_offline_greedy_from_samples_dev(self._h, C.c_void_p(dev_ptr), n_each, B, _l(tok), _i(ts), _i(n), mt))
        return self._unpack(tok, ts, n)


    # pipelined form: submit() returns a ticket, wait(ticket) returns that batch's results
    def offline_submit_samples_dev(self, dev_ptr: int, n_each: int, B: int, max_tokens: Optional[int] = None):
        mt = max_tokens or max(1, self.encoder_out_frames(self.fbank_num_frames(n_each) + 19))
        t = C.c_int32()
        self._chk(self._L.k2hip_offline_submit_samples_dev(self._h, C.c_void_p(dev_ptr), n_each, B, mt, C.byref(t)))
        return (t.value, B, mt)

    def host_alloc(self, shape, dtype=np.float32) -> np.ndarray:
        """page-locked host array (k2hip_host_alloc); release with host_free(array)"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self._L.k2hip_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p is not None:
            self._chk(self._L.k2hip_host_free(self._h, C.c_void_p(p)))

    def offline_submit_samples(self, samples: np.ndarray, max_tokens: Optional[int] = None):
        """pipelined submit of a [B, n] f32 batch that is still in host memory (H2D inside the pipeline)"""
        assert samples.dtype == np.float32 and samples.ndim == 2 and samples.flags.c_contiguous
        B, n_each = samples.shape
        mt = max_tokens or max(1, self.encoder_out_frames(self.fbank_num_frames(n_each) + 19))
        t = C.c_int32()
        self._chk(self._L.k2hip_offline_submit_samples(self._h, _f(samples), n_each, B, mt, C.byref(t)))
        return (t.value, B, mt)

    def offline_wait(self, ticket):
        t, B, mt = ticket
        tok = np.zeros((B, mt), np.int64)
        ts = np.zeros((B, mt), np.int32)
        n = np.zeros(B, np.int32)
        self._chk(self._L.k2hip_offline_wait(self._h, t, _l(tok), _i(ts), _i(n)))
        return self._unpack(tok, ts, n)


class OfflineStream:
    """OfflineStream.cs:7-99."""

    def __init__(self, model: Model):
        self._m = model
        self._L = model._L
        h = C.c_void_p()
        model._chk(self._L.k2hip_offline_stream_create(model.handle, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.k2hip_offline_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_samples(self, samples):  # AddSamples, OfflineStream.cs:43-57
        s = _f32(samples).reshape(-1)
        self._m._chk(self._L.k2hip_offline_stream_accept_samples(self._h, _f(s), s.size))

    @property
    def speech_length(self) -> int:  # OfflineInputEntity.SpeechLength
        return self._L.k2hip_offline_stream_speech_length(self._h)

    @property
    def speech(self) -> np.ndarray:  # OfflineInputEntity.Speech
        n = self.speech_length
        out = np.empty(n, np.float32)
        self._m._chk(self._L.k2hip_offline_stream_get_speech(self._h, _f(out), n))
        return out

    @property
    def tokens(self) -> List[int]:
        n = self._L.k2hip_offline_stream_num_tokens(self._h)
        out = np.zeros(max(n, 1), np.int64)
        self._m._chk(self._L.k2hip_offline_stream_get_tokens(self._h, _l(out), n))
        return out[:n].tolist()

    @property
    def timestamps(self) -> List[int]:
        n = self._L.k2hip_offline_stream_num_timestamps(self._h)
        out = np.zeros(max(n, 1), np.int32)
        self._m._chk(self._L.k2hip_offline_stream_get_timestamps(self._h, _i(out), n))
        return out[:n].tolist()


class OfflineRecognizer:
    """OfflineRecognizer.cs:12-91 on the HIP backend; decoding_method "greedy_search" (the reference's only method) or
    "modified_beam_search" (BASELINE.json configs[2])."""

    def __init__(self, weights_path: str, device: int = 0, decoding_method: str = "greedy_search", beam: int = 4):
        self.model = Model(weights_path, device)
        self.model.set_decoding_method(decoding_method, beam)

    def create_offline_stream(self) -> OfflineStream:  # CreateOfflineStream :71-75
        return OfflineStream(self.model)

    def get_results(self, streams: Sequence[OfflineStream]):  # GetResults :85-91 (tokens, not text)
        B = len(streams)
        arr = (C.c_void_p * B)(*[s._h for s in streams])
        self.model._chk(self.model._L.k2hip_offline_recognizer_get_results(self.model.handle, arr, B))
        return [(s.tokens, s.timestamps) for s in streams]

    def get_result(self, stream: OfflineStream):  # GetResult :77-83
        self.model._chk(self.model._L.k2hip_offline_recognizer_get_result(self.model.handle, stream._h))
        return stream.tokens, stream.timestamps


# ============================== streaming: OnlineStream / OnlineRecognizer ===============================
_STATE_KINDS = {"key": 0, "nonlin": 1, "val1": 2, "val2": 3, "conv1": 4, "conv2": 5, "embed": 6, "lstm_h": 0, "lstm_c": 1, "conf_attn": 0, "conf_conv": 1,
                # Zipformer v1 streams (OnlineProjOfZipformer.cs:56-111)
                "avg": 1, "val": 2, "len": 7}


def _bind_online(L):
    if getattr(L, "_online_bound", False):
        return
    vp = C.c_void_p
    L.k2hip_online_stream_create.argtypes = [vp, C.POINTER(vp)]
    L.k2hip_online_stream_destroy.argtypes = [vp]
    L.k2hip_online_stream_reset.argtypes = [vp]
    L.k2hip_online_chunk_info.argtypes = [vp, ip, ip, ip]
    L.k2hip_online_stream_accept_samples.argtypes = [vp, fp, C.c_int64]
    L.k2hip_online_stream_accept_features.argtypes = [vp, fp, C.c_int64]
    L.k2hip_online_accept_samples_batch.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(fp), lp]
    L.k2hip_online_accept_samples_matrix.argtypes = [vp, C.POINTER(vp), C.c_int32, vp, C.c_int64, C.c_int64]
    L.k2hip_online_stream_speech_length.restype = C.c_int64
    L.k2hip_online_stream_speech_length.argtypes = [vp]
    L.k2hip_online_stream_is_finished.argtypes = [vp, C.c_int32, ip]
    L.k2hip_online_step.argtypes = [vp, C.POINTER(vp), C.c_int32, ip, ip]
    L.k2hip_online_state_create.argtypes = [vp, C.POINTER(vp)]
    L.k2hip_online_state_destroy.argtypes = [vp]
    L.k2hip_online_state_processed_len.restype = C.c_int64
    L.k2hip_online_state_processed_len.argtypes = [vp]
    L.k2hip_online_encoder.argtypes = [vp, C.POINTER(vp), C.c_int32, fp, fp, C.c_int64]
    L.k2hip_online_stream_num_tokens.argtypes = [vp]
    L.k2hip_online_stream_num_timestamps.argtypes = [vp]
    L.k2hip_online_stream_get_tokens.argtypes = [vp, lp, C.c_int32]
    L.k2hip_online_stream_get_timestamps.argtypes = [vp, ip, C.c_int32]
    L.k2hip_online_stream_get_hyp.argtypes = [vp, lp]
    L.k2hip_online_stream_state.argtypes = [vp, C.c_int32, C.c_int32, fp, C.c_int64, lp]
    L._online_bound = True


class OnlineStream:
    """OnlineStream.cs:7-199 (feature FIFO + Hyp/Tokens/Timestamps; caches live in a GPU slot)."""

    def __init__(self, model: Model):
        self._m = model
        self._L = model._L
        _bind_online(self._L)
        h = C.c_void_p()
        model._chk(self._L.k2hip_online_stream_create(model.handle, C.byref(h)))
        self._h = h

    def reset(self):
        """back to a freshly created stream (same slot): for the next utterance on the same object"""
        self._m._chk(self._L.k2hip_online_stream_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.k2hip_online_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_samples(self, samples):  # AddSamples :57-79
        s = _f32(samples).reshape(-1)
        self._m._chk(self._L.k2hip_online_stream_accept_samples(self._h, _f(s), s.size))

    def add_features(self, feats):
        f = _f32(feats).reshape(-1, self._m.feature_dim)
        self._m._chk(self._L.k2hip_online_stream_accept_features(self._h, _f(f), f.shape[0]))

    @property
    def speech_length(self) -> int:
        return self._L.k2hip_online_stream_speech_length(self._h)

    def is_finished(self, is_endpoint: bool = False) -> bool:  # IsFinished :124-161
        out = C.c_int32()
        self._m._chk(self._L.k2hip_online_stream_is_finished(self._h, int(is_endpoint), C.byref(out)))
        return bool(out.value)

    @property
    def tokens(self) -> List[int]:
        n = self._L.k2hip_online_stream_num_tokens(self._h)
        out = np.zeros(max(n, 1), np.int64)
        self._m._chk(self._L.k2hip_online_stream_get_tokens(self._h, _l(out), n))
        return out[:n].tolist()

    @property
    def timestamps(self) -> List[int]:
        n = self._L.k2hip_online_stream_num_timestamps(self._h)
        out = np.zeros(max(n, 1), np.int32)
        self._m._chk(self._L.k2hip_online_stream_get_timestamps(self._h, _i(out), n))
        return out[:n].tolist()

    @property
    def hyp(self) -> List[int]:
        out = np.zeros(2, np.int64)
        self._m._chk(self._L.k2hip_online_stream_get_hyp(self._h, _l(out)))
        return out.tolist()

    @property
    def processed_len(self) -> int:
        self._L.k2hip_online_stream_processed_len.restype = C.c_int64
        self._L.k2hip_online_stream_processed_len.argtypes = [C.c_void_p]
        return int(self._L.k2hip_online_stream_processed_len(self._h))

    def state(self, layer: int, kind: str) -> np.ndarray:
        n = C.c_int64()
        self._m._chk(self._L.k2hip_online_stream_state(self._h, layer, _STATE_KINDS[kind], None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        self._m._chk(self._L.k2hip_online_stream_state(self._h, layer, _STATE_KINDS[kind], _f(out), n.value, C.byref(n)))
        return out


class StreamBatch:
    """The handles of a fixed group of OnlineStreams as one ctypes array (what a native host keeps next to its stream objects)."""

    def __init__(self, streams):
        self.streams = list(streams)
        self.arr = (C.c_void_p * len(self.streams))(*[s._h.value for s in self.streams])

    def __len__(self):
        return len(self.streams)

    def __iter__(self):
        return iter(self.streams)

    def __getitem__(self, i):
        return self.streams[i]


class OnlineProj:
    """IOnlineProj (IOnlineProj.cs:65-71) on the HIP backend: the operator a host swaps in when it keeps the reference's own
    OnlineRecognizer loop.  States are handles (a slot of the device pool each); stack_states / unstack_states are the identity."""

    def __init__(self, weights_path: str, device: int = 0):
        self.model = Model(weights_path, device)
        _bind_online(self.model._L)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.model._chk(self.model._L.k2hip_online_chunk_info(self.model.handle, C.byref(a), C.byref(b), C.byref(c)))
        self.chunk_length, self.shift_length, self.frames_per_chunk = a.value, b.value, c.value

    def get_encoder_init_states(self):  # GetEncoderInitStates, one stream
        h = C.c_void_p()
        self.model._chk(self.model._L.k2hip_online_state_create(self.model.handle, C.byref(h)))
        return h

    def free_states(self, state):
        self.model._chk(self.model._L.k2hip_online_state_destroy(state))

    def processed_len(self, state) -> int:
        return int(self.model._L.k2hip_online_state_processed_len(state))

    def encoder_proj(self, feats, states) -> np.ndarray:  # EncoderProj: the states advance in place
        B = len(states)
        x = _f32(feats).reshape(B, self.chunk_length, -1)
        out = np.empty((B, self.frames_per_chunk, self.model.joiner_dim), np.float32)
        arr = (C.c_void_p * B)(*[s.value for s in states])
        self.model._chk(self.model._L.k2hip_online_encoder(self.model.handle, arr, B, _f(x), _f(out), out.size))
        return out

    def decoder_proj(self, y) -> np.ndarray:
        return self.model.decoder_proj(y)

    def joiner_proj(self, enc, dec) -> np.ndarray:
        return self.model.joiner_proj(enc, dec)


class OnlineRecognizer:
    """OnlineRecognizer.cs:11-84 with decodingMethod = "greedy_search" on the HIP backend."""

    def __init__(self, weights_path: str, device: int = 0):
        self.model = Model(weights_path, device)
        _bind_online(self.model._L)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.model._chk(self.model._L.k2hip_online_chunk_info(self.model.handle, C.byref(a), C.byref(b), C.byref(c)))
        self.chunk_length, self.shift_length, self.frames_per_chunk = a.value, b.value, c.value
        mt = self.model.meta("model_type")
        nl = self.model.meta("num_encoder_layers")
        self.num_layers = sum(int(x) for x in nl.split(",")) if nl else 0
        # names of the per-layer caches of this operator's GetEncoderInitStates, and the floats of embed_states (Zipformer2 only)
        self.state_kinds = {"zipformer2": ["key", "nonlin", "val1", "val2", "conv1", "conv2"], "zipformer2ctc": ["key", "nonlin", "val1", "val2", "conv1", "conv2"],
                            "zipformer": ["len", "avg", "key", "val", "val2", "conv1", "conv2"], "conformer": ["conf_attn", "conf_conv"],
                            "lstm": ["lstm_h", "lstm_c"]}.get(mt, [])
        self.embed_state_floats = 128 * 3 * 19 if mt in ("zipformer2", "zipformer2ctc") else 0

    def create_online_stream(self) -> OnlineStream:  # CreateOnlineStream :60-64
        return OnlineStream(self.model)

    def add_samples_batch(self, streams: Sequence[OnlineStream], samples: Sequence[np.ndarray]):
        """B AddSamples calls in one (one fbank launch when all streams are at the same position)."""
        B = len(streams)
        arr = self._handles(streams)
        if isinstance(samples, np.ndarray) and samples.ndim == 2 and samples.dtype == np.float32 and samples.strides[1] == 4 and samples.strides[0] % 4 == 0:
            # one [B, n] matrix (rows contiguous, any row stride): one call, no per-stream pointers
            self.model._chk(self.model._L.k2hip_online_accept_samples_matrix(self.model.handle, arr, B, C.c_void_p(samples.ctypes.data),
                                                                             samples.strides[0] // 4, samples.shape[1]))
            return
        ss = [_f32(x).reshape(-1) for x in samples]
        ptrs = (fp * B)(*[_f(x) for x in ss])
        n = np.array([x.size for x in ss], np.int64)
        self.model._chk(self.model._L.k2hip_online_accept_samples_batch(self.model.handle, arr, B, ptrs, _l(n)))

    def get_results(self, streams: Sequence[OnlineStream]):
        """GetResults :76-84 (tokens, not text).  Returns (decoded flags, new-token counts)."""
        B = len(streams)
        arr = self._handles(streams)
        dec = np.zeros(B, np.int32)
        n = np.zeros(B, np.int32)
        self.model._chk(self.model._L.k2hip_online_step(self.model.handle, arr, B, _i(dec), _i(n)))
        return dec.tolist(), n.tolist()

    def batch(self, streams: Sequence[OnlineStream]) -> "StreamBatch":
        """A fixed group of streams as the native array of handles a C# / C host would hold (IntPtr[]): pass it to
        add_samples_batch / get_results instead of the list and no per-call scan of the streams happens."""
        return StreamBatch(streams)

    def _handles(self, streams):
        """ctypes array of the streams' handles; rebuilt only when the list changes (a serving loop passes the same list every 50 ms)"""
        if isinstance(streams, StreamBatch):
            return streams.arr
        key = tuple(s._h.value for s in streams)
        if getattr(self, "_hkey", None) != key:
            self._hkey = key
            self._harr = (C.c_void_p * len(streams))(*key)
        return self._harr
